# Gram kernel build variants (EVOFST_LIBRARY) on the cfg5 shard (run through gpurun)
# variants are built by hand, e.g.  hipcc <flags of evo_amd/csrc/Makefile> -DEVOFST_GRAM_WPB=4 -shared -o build_var/libevofst_wpb4.so ...
cd $GRAFT_REPO_ROOT
for L in "" build_var/libevofst_wpb2.so build_var/libevofst_wpb4.so build_var/libevofst_wpb16.so; do
  EVOFST_LIBRARY=${L:+$GRAFT_REPO_ROOT/$L} timeout -k 10 200 python3 bench.py --config cfg5s --steps 2 --warmup 1 --cpu-sites 0 > gpurun_out/gv.json 2> gpurun_out/gv.err || { tail -3 gpurun_out/gv.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/gv.json')); print('lib=${L:-default(wpb8)} ms/step %.1f gram %.1f frac %.3f' % (d['ms_per_step'], d['roofline']['kernels_ms']['gram_kernel'], d['roofline']['frac']))"
done
