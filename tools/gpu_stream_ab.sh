# stream pass against the separate kernels: cfg2 (3 streams and 1), cfg4 shard (run through gpurun)
set -e
mkdir -p gpurun_out
B="timeout -k 10 240 python3 bench.py --cpu-sites 0 --no-hbm-leg"
for pp in 1 2; do
  $B --config cfg2 --steps 400 --warmup 40 --pairs-pass $pp > gpurun_out/ab_cfg2_p$pp.json 2> gpurun_out/ab_cfg2_p$pp.err
  $B --config cfg2 --steps 400 --warmup 40 --pairs-pass $pp --streams 1 > gpurun_out/ab_cfg2_s1_p$pp.json 2> gpurun_out/ab_cfg2_s1_p$pp.err
  $B --config cfg4s --steps 30 --warmup 5 --pairs-pass $pp > gpurun_out/ab_cfg4s_p$pp.json 2> gpurun_out/ab_cfg4s_p$pp.err
done
python3 - <<'PY'
import json
for n in ("cfg2", "cfg2_s1", "cfg4s"):
    for pp in (1, 2):
        d = json.load(open("gpurun_out/ab_%s_p%d.json" % (n, pp)))
        print(n, "pass", pp, "ms/step %.4f" % d["ms_per_step"], {k.split("(")[0]: round(v, 4) for k, v in d["roofline"]["kernels_ms"].items()})
PY
