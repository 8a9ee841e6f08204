for cfg in cfg2 cfg4s; do
timeout -k 10 200 python -u bench.py --config $cfg --steps 20 --warmup 3 --cpu-sites 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d['ms_per_step'], d['roofline']['kernels_ms'])" || exit 1
done
