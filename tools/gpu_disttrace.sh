# kernel trace of the multi-rank pass rehearsed with one rank: what runs between K1 and K2, and the gaps
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/disttrace
rm -rf $OUT
timeout -s KILL 180 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --dist-path --steps 20 --warmup 3 --cpu-sites 0 > $OUT.log 2>&1 || { tail -5 $OUT.log; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]))
rows.sort()
# one steady-state step: from the 15th k1 launch to the 16th
k1 = [i for i, r in enumerate(rows) if "k1_count" in r[2]]
a, b = k1[14], k1[15]
t0 = rows[a][0]
for s, e, n in rows[a:b + 1]:
    print("%8.1f us  +%6.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n))
PY
