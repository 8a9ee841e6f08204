# chunk size sweep: per-kernel times on cfg2 (EVOFST_CHUNK_TILES overrides the default)
cd $GRAFT_REPO_ROOT
for X in ${CTS:-2 3 4 6 8}; do
  EVOFST_CHUNK_TILES=$X timeout -k 10 200 python3 bench.py --config ${CFG:-cfg2} --steps 50 --warmup 5 --cpu-sites 0 > gpurun_out/ct.json 2> gpurun_out/ct.err || { tail -3 gpurun_out/ct.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/ct.json')); k=d['roofline']['kernels_ms']; print('chunk_tiles $X: step %.4f ms  ' % d['ms_per_step'] + '  '.join('%s %.1f' % (n[:6], v*1e3) for n, v in k.items()))"
done
