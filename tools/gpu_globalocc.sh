# against-all kernel against occupancy: extra dynamic LDS per block (run through gpurun)
cd $GRAFT_REPO_ROOT
for X in 0 8000 32000 60000; do
  EVOFST_DEBUG_GLOBAL_LDS=$X timeout -k 10 200 python3 bench.py --config cfg3s --steps 10 --warmup 2 --cpu-sites 0 > gpurun_out/go.json 2> gpurun_out/go.err || { tail -3 gpurun_out/go.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/go.json')); print('extra LDS $X: ms/step %.3f' % d['ms_per_step'], {k: round(v*1e3) for k,v in d['roofline']['kernels_ms'].items()})"
done
