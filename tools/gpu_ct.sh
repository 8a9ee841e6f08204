# the cfg2 / cfg4s step against the chunk size (EVOFST_CHUNK_TILES), serialized kernels and the 3-stream step (run through gpurun)
cd $GRAFT_REPO_ROOT
for CT in 4 6 8 12 16; do
  export EVOFST_CHUNK_TILES=$CT
  timeout -k 10 120 python3 bench.py --config cfg2 --steps 200 --warmup 20 --cpu-sites 0 --no-hbm-leg > gpurun_out/ct_$CT.json 2> gpurun_out/ct.err || { tail -5 gpurun_out/ct.err; exit 1; }
  echo "== cfg2 chunk_tiles $CT"; python3 tools/bench_brief.py gpurun_out/ct_$CT.json
done
