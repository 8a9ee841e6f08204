cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc1 -- python3 bench.py --steps 3 --warmup 1 --cpu-sites 0 > gpurun_out/pmc1.log 2>&1 || { tail -5 gpurun_out/pmc1.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/pmc2 -- python3 bench.py --steps 3 --warmup 1 --cpu-sites 0 > gpurun_out/pmc2.log 2>&1 || { tail -5 gpurun_out/pmc2.log; exit 1; }
python3 - <<'PY'
import csv,glob,collections
for d in ("pmc1","pmc2"):
    for f in glob.glob("gpurun_out/%s/**/*counter_collection.csv"%d, recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"][:40]; acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
            if r["Counter_Name"]==list(acc[k].keys())[0]: n[k]+=1
        for k in acc:
            print(d,k,n[k],{c:round(v/max(n[k],1)) for c,v in acc[k].items()})
PY
