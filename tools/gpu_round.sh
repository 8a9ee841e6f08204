# K2 with exactly one round of blocks over the chip: 3072 chunks of 5 tiles (983 040 sites) against the default plan
cd $GRAFT_REPO_ROOT
run() {  # sites chunk_tiles
  EVOFST_CHUNK_TILES=$2 timeout -k 10 200 python3 bench.py --config cfg2 --sites $1 --steps 50 --warmup 5 --cpu-sites 0 > gpurun_out/rd.json 2> gpurun_out/rd.err || { tail -3 gpurun_out/rd.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/rd.json')); k=d['roofline']['kernels_ms']; print('sites $1 chunk_tiles $2: step %.4f ms  ' % d['ms_per_step'] + '  '.join('%s %.1f' % (n[:6], v*1e3) for n, v in k.items()), ' K2 ns/ksite %.2f' % (k['k2_window_kernel']*1e6/($1/1000.)))"
}
run 983040 5 && run 983040 4 && run 786432 4 && run 1000000 4 && run 1000000 5
