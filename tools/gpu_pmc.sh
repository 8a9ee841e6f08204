cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmc3 -- python3 bench.py --steps 3 --warmup 1 --cpu-sites 0 > gpurun_out/pmc3.log 2>&1 || { tail -5 gpurun_out/pmc3.log; exit 1; }
python3 - <<'PY'
import csv,glob,collections
dur=collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc3/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"][:40]].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for f in glob.glob("gpurun_out/pmc3/**/*counter_collection.csv", recursive=True):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:40]; acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        if r["Counter_Name"]=="SQ_WAVES": n[k]+=1
    for k in acc:
        d=sum(dur[k])/max(len(dur[k]),1)
        a={c:v/max(n[k],1) for c,v in acc[k].items()}
        print(k, "avg_ns=%.0f"%d, "clk_GHz=%.2f"%(a.get("GRBM_GUI_ACTIVE",0)/8/max(d,1)), {c:round(v) for c,v in a.items()})
PY
