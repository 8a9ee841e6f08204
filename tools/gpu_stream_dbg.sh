# stream pass with parts switched off (-DEVOFST_EXPERIMENTS build; results are wrong when a switch is set): one stream
set -e
mkdir -p gpurun_out
export EVOFST_LIB=$PWD/evo_amd/libevofst_exp.so
for cfg in ${CFGS:-cfg2 cfg4s}; do
  steps=200; [ $cfg = cfg4s ] && steps=20
  for dbg in ${DBGS:-0 1 2 3 4 8 12}; do
    EVOFST_DEBUG_STREAM=$dbg timeout -k 10 200 python3 bench.py --config $cfg --steps $steps --warmup 10 --cpu-sites 0 --no-hbm-leg --streams 1 --pairs-pass 2 ${EXTRA:-} > gpurun_out/dbg.json 2> gpurun_out/dbg.err || { tail -3 gpurun_out/dbg.err; exit 1; }
    python3 -c "
import json
d = json.load(open('gpurun_out/dbg.json'))
print('$cfg', 'debug', $dbg, 'ms/step %.4f' % d['ms_per_step'], {k.split('(')[0]: round(v, 4) for k, v in d['roofline']['kernels_ms'].items()})"
  done
done
