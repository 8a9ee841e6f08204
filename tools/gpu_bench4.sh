for dbg in ${DBGS:-0 1 2 3}; do
EVOFST_DEBUG_K2=$dbg timeout -k 10 200 python -u bench.py --config ${CFG:-cfg2} --steps 20 --warmup 3 --cpu-sites 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dbg$dbg', d['ms_per_step'], d['roofline']['kernels_ms']['k2_window_kernel'])" || exit 1
done
