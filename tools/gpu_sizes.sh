# per-kernel time against the number of sites (fixed costs of the four kernels of a pairs-mode pass)
cd $GRAFT_REPO_ROOT
for S in ${SIZES:-16384 65536 262144 524288 1000000 2000000}; do
  timeout -k 10 200 python3 bench.py --config cfg2 --sites $S --steps 50 --warmup 5 --cpu-sites 0 > gpurun_out/sz.json 2> gpurun_out/sz.err || { tail -3 gpurun_out/sz.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/sz.json')); k=d['roofline']['kernels_ms']; print('sites $S: step %.4f ms  ' % d['ms_per_step'] + '  '.join('%s %.1f' % (n[:6], v*1e3) for n, v in k.items()))"
done
