# K2 against occupancy: extra dynamic LDS per block (run through gpurun)
cd $GRAFT_REPO_ROOT
for X in 0 4000 8000 14000 28000; do
  EVOFST_DEBUG_K2_LDS=$X timeout -k 10 200 python3 bench.py --config ${CFG:-cfg2} --steps 30 --warmup 3 --cpu-sites 0 > gpurun_out/ko.json 2> gpurun_out/ko.err || { tail -3 gpurun_out/ko.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/ko.json')); print('extra LDS $X: K2 %.1f us' % (d['roofline']['kernels_ms']['k2_window_kernel']*1e3))"
done
