# TCC counters of the stream pass with and without the look-back (experiments build): what the look-back costs the memory side
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export EVOFST_LIB=$GRAFT_REPO_ROOT/evo_amd/libevofst_exp.so
for dbg in 0 2; do
  for CTRS in "TCC_EA_RDREQ_sum TCC_EA_RD_UNCACHED_32B_sum TCC_EA_ATOMIC_sum TCC_TAG_STALL_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA_WRREQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum"; do
    OUT=gpurun_out/spmc_$dbg
    rm -rf $OUT
    EVOFST_DEBUG_STREAM=$dbg timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT -- python3 bench.py --config cfg4s --steps 3 --warmup 1 --cpu-sites 0 --no-hbm-leg --streams 1 --batches 1 > $OUT.log 2>&1 || { tail -5 $OUT.log; exit 1; }
    python3 - $OUT $dbg <<'PY'
import csv,glob,collections,sys
out,dbg=sys.argv[1],sys.argv[2]
for f in glob.glob(out+"/**/*counter_collection.csv", recursive=True):
    acc=collections.defaultdict(float); n=collections.Counter()
    for r in csv.DictReader(open(f)):
        if "stream_pairs" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
    print("debug", dbg, {c: round(v/max(n[c],1)) for c,v in acc.items()})
PY
  done
done
