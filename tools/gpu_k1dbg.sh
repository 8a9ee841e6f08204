for dbg in ${DBGS:-0 1 2 4 6 7}; do
EVOFST_DEBUG_K1=$dbg timeout -k 10 200 python -u bench.py --config ${CFG:-cfg4s} --steps 10 --warmup 2 --cpu-sites 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('k1dbg$dbg', '%.4f'%d['roofline']['kernels_ms']['k1_count_kernel'], round(d['roofline']['k1_GBps']))" || exit 1
done
