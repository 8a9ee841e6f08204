# chunk size sweep on the bench configuration (run through gpurun)
cd $GRAFT_REPO_ROOT
for CT in 1 2 3 4 5 6 8; do
  EVOFST_CHUNK_TILES=$CT timeout -k 10 120 python3 bench.py --config ${CFG:-cfg2} --steps 50 --warmup 5 --cpu-sites 0 > gpurun_out/ct.json 2> gpurun_out/ct.err || { tail -3 gpurun_out/ct.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/ct.json')); k=d['roofline']['kernels_ms']; print('chunk_tiles=$CT ms/step %.4f  K1 %.1f scan %.1f K2 %.1f stitch %.1f us' % (d['ms_per_step'], k['k1_count_kernel']*1e3, k['scan_kernel']*1e3, k['k2_window_kernel']*1e3, k['stitch_kernel']*1e3))"
done
