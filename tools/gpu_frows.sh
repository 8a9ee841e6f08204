# timing + kernel stats of the 8(f) consumer kernels (run through gpurun)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_frows
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 200 python3 tools/f_rows_bench.py > $OUT/wall.log 2>&1 || { tail -5 $OUT/wall.log; exit 1; }
cat $OUT/wall.log
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/f_rows_bench.py > $OUT/stats.log 2>&1 || { tail -5 $OUT/stats.log; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("%-70s calls %4s avg %10.1f us" % (r["Name"].split("(")[0][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
    import shutil; shutil.copy(f, sys.argv[1] + "/kernel_stats.csv")
PY
