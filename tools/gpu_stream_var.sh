# stream pass variants (exp build): EVOFST_STREAM_AHEAD, producers/consumers, chunk tiles, blocks
set -e
mkdir -p gpurun_out
export EVOFST_LIB=$PWD/evo_amd/libevofst_exp.so
run() {  # cfg steps label env... -- bench args
  cfg=$1; steps=$2; label=$3; shift 3
  env "$@" timeout -k 10 200 python3 bench.py --config $cfg --steps $steps --warmup 10 --cpu-sites 0 --no-hbm-leg --streams 1 --pairs-pass 2 $BARGS > gpurun_out/var.json 2> gpurun_out/var.err || { tail -3 gpurun_out/var.err; exit 1; }
  python3 -c "
import json
d = json.load(open('gpurun_out/var.json'))
print('$cfg', '$label', 'ms/step %.4f' % d['ms_per_step'], {k.split('(')[0]: round(v, 4) for k, v in d['roofline']['kernels_ms'].items()})"
}
for a in 1 2; do
  BARGS="" run cfg2 200 "ahead=$a" EVOFST_STREAM_AHEAD=$a
  BARGS="" run cfg4s 20 "ahead=$a" EVOFST_STREAM_AHEAD=$a
done
BARGS="--stream-waves 4,1" run cfg4s 20 "waves 4,1" EVOFST_STREAM_AHEAD=1
BARGS="--stream-waves 2,1" run cfg2 200 "waves 2,1" EVOFST_STREAM_AHEAD=1
BARGS="--stream-waves 3,1" run cfg2 200 "waves 3,1" EVOFST_STREAM_AHEAD=1
BARGS="--chunk-tiles 4" run cfg2 200 "chunk 4" EVOFST_STREAM_AHEAD=1
BARGS="--chunk-tiles 16" run cfg2 200 "chunk 16" EVOFST_STREAM_AHEAD=1
BARGS="--chunk-tiles 4" run cfg4s 20 "chunk 4" EVOFST_STREAM_AHEAD=1
