cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CFG=${CFG:-cfg4s}
# one pass holds at most 8 SQ_ counters
CTRS=${CTRS:-GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD}
OUT=gpurun_out/${TAG:-pmc}
rm -rf $OUT
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT -- python3 bench.py --config $CFG --steps 3 --warmup 1 --cpu-sites 0 --no-hbm-leg --streams 1 > $OUT.log 2>&1 || { tail -5 $OUT.log; exit 1; }
python3 - $OUT <<'PY'
import csv,glob,collections,sys
out=sys.argv[1]
dur=collections.defaultdict(list)
for f in glob.glob(out+"/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"][:40]].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for f in glob.glob(out+"/**/*counter_collection.csv", recursive=True):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    first=None
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:40]; acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        if first is None: first=r["Counter_Name"]
        if r["Counter_Name"]==first: n[k]+=1
    for k in acc:
        d=sum(dur[k])/max(len(dur[k]),1)
        a={c:v/max(n[k],1) for c,v in acc[k].items()}
        print(k, "avg_ns=%.0f"%d, {c:round(v) for c,v in a.items()})
PY
