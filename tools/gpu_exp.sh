B='import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernels_ms"])'
for cf in cfg2 cfg2nw; do for dbg in 0 2; do echo "cfg=$cf debug=$dbg"; EVOFST_DEBUG_K2=$dbg timeout -k 10 100 python -u bench.py --config $cf --steps 30 --warmup 3 --cpu-sites 0 2>/dev/null | python -c "$B" || exit 1; done; done
