# rocprofv3 kernel stats + HBM traffic counters of one bench configuration (run through gpurun)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CFG=${CFG:-cfg2}
OUT=gpurun_out/prof_$CFG
rm -rf $OUT && mkdir -p $OUT
timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --config $CFG --steps 20 --warmup 3 --cpu-sites 0 --no-hbm-leg --streams 1 > $OUT/stats.log 2>&1 || { tail -5 $OUT/stats.log; exit 1; }
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --config $CFG --steps 3 --warmup 1 --cpu-sites 0 --no-hbm-leg --streams 1 > $OUT/fetch.log 2>&1 || { tail -5 $OUT/fetch.log; exit 1; }
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --config $CFG --steps 3 --warmup 1 --cpu-sites 0 --no-hbm-leg --streams 1 > $OUT/write.log 2>&1 || { tail -5 $OUT/write.log; exit 1; }
python3 - $OUT $CFG <<'PY'
import csv, glob, collections, json, sys
out, cfg = sys.argv[1], sys.argv[2]
res = collections.defaultdict(dict)
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        res[r["Name"].split("(")[0]]["avg_ns"] = float(r["AverageNs"]); res[r["Name"].split("(")[0]]["calls"] = int(r["Calls"])
for which in ("fetch", "write"):
    for f in glob.glob(out + "/" + which + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(float); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            acc[k] += float(r["Counter_Value"]); n[k] += 1
        for k in acc:
            res[k][which + "_size_raw_per_launch"] = acc[k] / n[k]
json.dump(res, open(out + "/summary.json", "w"), indent=1)
for k, v in res.items():
    print(k[:60], v)
PY
