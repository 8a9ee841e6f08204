# stream pass variants (exp build): producers/consumers, chunk tiles
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
# VARS: producers,consumers per set:chunk tiles:buffers:consumer sets
for v in ${VARS:-4,3:8:2 8,3:16:2 6,3:12:3}; do
  IFS=: read w ct nb ns <<< "$v"
  for poll in ${POLLS:-8}; do
    BARGS="--stream-waves $w --chunk-tiles $ct --stream-buffers $nb --stream-sets ${ns:-1}" run ${CFG:-cfg4s} ${STEPS:-20} "waves $w chunk $ct buffers $nb sets ${ns:-1} poll $poll" EVOFST_STREAM_POLL=$poll
  done
done
