B='import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernels_ms"])'
for v in "4 4" "2 4" "2 6" "2 8" "4 5"; do set -- $v
  (cd evo_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DEVOFST_K2_GRP=$1 -DEVOFST_K2_WPS=$2 -shared -o ../libevofst.so evofst_kernels.hip -x hip evofst_api.cpp 2>/dev/null) || exit 1
  for ct in 4 8; do echo "grp=$1 wps=$2 ct=$ct"; EVOFST_CHUNK_TILES=$ct timeout -k 10 100 python -u bench.py --steps 30 --warmup 3 --cpu-sites 0 2>/dev/null | python -c "$B" || exit 1; done
done
