# K1 on the cfg4 shard with its count stores (4) and / or its four mask words per tile (8) switched off
# (-DEVOFST_EXPERIMENTS build; clean data, so nothing reads what is missing) (run through gpurun)
cd $GRAFT_REPO_ROOT
for D in 0 4 8 12; do
  EVOFST_DEBUG_K1=$D EVOFST_LIB=$GRAFT_REPO_ROOT/evo_amd/libevofst_exp.so timeout -k 10 200 python3 bench.py --config cfg4s --steps 10 --warmup 2 --cpu-sites 0 --no-hbm-leg --streams 1 > gpurun_out/k1bits_$D.json 2> gpurun_out/k1bits.err || { tail -5 gpurun_out/k1bits.err; exit 1; }
  echo "== debug $D"; python3 tools/bench_brief.py gpurun_out/k1bits_$D.json
done
