timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/t8.log 2>&1; tail -4 gpurun_out/t8.log
for cfg in ${CFGS:-cfg2 cfg4s}; do
timeout -k 10 200 python -u bench.py --config $cfg --steps 30 --warmup 3 --cpu-sites 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', '%.3e'%d['value'], '%.4f'%d['ms_per_step'], {k:round(v,4) for k,v in d['roofline']['kernels_ms'].items()}, round(d['roofline']['k1_GBps']))" || exit 1
done
