# K2 variants built beforehand into build/ (EVOFST_LIBRARY picks the library): K2 time per variant on cfg2 and the cfg4 shard
cd $GRAFT_REPO_ROOT
for V in ${VARIANTS:-base cur u2 u8}; do
  if [ $V = cur ]; then unset EVOFST_LIBRARY; else export EVOFST_LIBRARY=$GRAFT_REPO_ROOT/build/libevofst_$V.so; fi
  for C in cfg2 cfg4s; do
    timeout -k 10 200 python3 bench.py --config $C --steps 50 --warmup 5 --cpu-sites 0 > gpurun_out/kv.json 2> gpurun_out/kv.err || { tail -3 gpurun_out/kv.err; exit 1; }
    python3 -c "
import json; d=json.load(open('gpurun_out/kv.json')); print('$V $C: step %.4f ms, K2 %.1f us' % (d['ms_per_step'], d['roofline']['kernels_ms']['k2_window_kernel']*1e3))"
  done
done
