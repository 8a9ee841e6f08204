# phase timing of the Gram kernel with the EVOFST_DEBUG_GRAM switches (run through gpurun)
cd $GRAFT_REPO_ROOT
for CFG in cfg5s cfg5s_w1000; do
for D in 0 1 2 3 4; do
  EVOFST_DEBUG_GRAM=$D timeout -k 10 200 python3 bench.py --config $CFG --steps 2 --warmup 1 --cpu-sites 0 > gpurun_out/gd.json 2> gpurun_out/gd.err || { tail -3 gpurun_out/gd.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/gd.json')); print('$CFG debug=$D ms/step %.1f gram %.1f' % (d['ms_per_step'], d['roofline']['kernels_ms']['gram_kernel']))"
done; done
