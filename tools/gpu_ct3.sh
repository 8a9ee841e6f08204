# the against-all window kernel on the cfg3 shard against the tiles a workgroup walks (groups of K1's 8-tile chunks;
# -DEVOFST_EXPERIMENTS build) (run through gpurun)
cd $GRAFT_REPO_ROOT
for GT in 8 16 24 32 48 64; do
  EVOFST_GLOBAL_GROUP_TILES=$GT EVOFST_LIB=$GRAFT_REPO_ROOT/evo_amd/libevofst_exp.so timeout -k 10 120 python3 bench.py --config cfg3s --steps 10 --warmup 2 --cpu-sites 0 --no-hbm-leg --streams 1 > gpurun_out/ct3_$GT.json 2> gpurun_out/ct3.err || { tail -5 gpurun_out/ct3.err; exit 1; }
  echo "== cfg3s tiles per workgroup $GT"; python3 tools/bench_brief.py gpurun_out/ct3_$GT.json | tail -1
done
