for ct in ${CTS:-1 2 4 8}; do
EVOFST_CHUNK_TILES=$ct timeout -k 10 200 python -u bench.py --config ${CFG:-cfg2} --steps 30 --warmup 3 --cpu-sites 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ct$ct', '%.3e'%d['value'], '%.4f'%d['ms_per_step'], {k:round(v,4) for k,v in d['roofline']['kernels_ms'].items()})" || exit 1
done
