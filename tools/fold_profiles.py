#!/usr/bin/env python3
"""Folds the output of tools/gpu_profile.sh (gpurun_out/prof_<cfg>/) into profiles/: the rocprofv3 kernel_stats.csv of
the configuration and its entry in <round>_traffic.json (ROUND, default r02) (FETCH_SIZE / WRITE_SIZE are KB per launch; on gfx950 FETCH_SIZE
counts 128-byte requests of wide reads as 64, hence x 2 -- MI355X_MICROARCH.md, HBM section).
python3 tools/fold_profiles.py cfg2 cfg4s ..."""
import glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = os.environ.get("ROUND", "r02")
tp = os.path.join(ROOT, "profiles", RND + "_traffic.json")
traffic = json.load(open(tp)) if os.path.exists(tp) else {}
for cfg in sys.argv[1:]:
    d = os.path.join(ROOT, "gpurun_out", "prof_" + cfg)
    summ = json.load(open(os.path.join(d, "summary.json")))
    out = {}
    for name, v in summ.items():
        short = name.replace("void ", "").replace("evofst::", "").split("<")[0]
        if short.startswith("__amd") or short == "synth_kernel" or "avg_ns" not in v:
            continue
        e = {"avg_ns": v["avg_ns"], "calls": v["calls"]}
        if "fetch_size_raw_per_launch" in v and "write_size_raw_per_launch" in v:
            e["FETCH_SIZE_KB_raw"] = v["fetch_size_raw_per_launch"]
            e["WRITE_SIZE_KB_raw"] = v["write_size_raw_per_launch"]
            e["hbm_read_bytes"] = v["fetch_size_raw_per_launch"] * 1024 * 2
            e["hbm_write_bytes"] = v["write_size_raw_per_launch"] * 1024
        out[short] = e
    # bench.py's "walk" line is the two statistics kernels together (clean chunks / the rest)
    parts = [out[k] for k in ("clean_walk_kernel", "fused_pairs_kernel") if k in out]
    if parts:
        out["walk"] = {k: sum(x.get(k, 0) for x in parts) for k in ("avg_ns", "hbm_read_bytes", "hbm_write_bytes")}
        out["walk"]["calls"] = parts[0]["calls"]
    traffic[cfg] = out
    stats = glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True)
    if stats:  # gpurun_out/ keeps the files of earlier calls: the newest one is this run's
        stats.sort(key=os.path.getmtime)
        shutil.copy(stats[-1], os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.csv" % (RND, cfg)))
    print(cfg, {k: (round(v["avg_ns"] / 1e3, 1), round((v.get("hbm_read_bytes", 0) + v.get("hbm_write_bytes", 0)) / 1e6, 1)) for k, v in out.items()})
json.dump(traffic, open(tp, "w"), indent=1)
