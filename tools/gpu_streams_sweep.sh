# bench.py step time on cfg2 against the number of streams / resident batches (run through gpurun)
cd $GRAFT_REPO_ROOT
for S in 1 2 3 4 6 8; do
  for B in 0 8; do
    timeout -k 10 120 python3 bench.py --config cfg2 --steps 200 --warmup 20 --cpu-sites 0 --no-hbm-leg --streams $S --batches $B > gpurun_out/streams_$S_$B.json 2> gpurun_out/streams.err || { tail -5 gpurun_out/streams.err; exit 1; }
    python3 - gpurun_out/streams_$S_$B.json $S $B <<'PY'
import json, sys
r = json.load(open(sys.argv[1]))
print("streams", sys.argv[2], "batches", sys.argv[3], "->", r["config"].get("resident_batches"), "step_us %.1f" % (r["ms_per_step"] * 1e3), "value %.3g" % r["value"])
PY
  done
done
