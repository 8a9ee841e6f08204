#!/usr/bin/env python3
"""End-to-end CLI timing: evo fst on a synthetic VCF, GPU front-end vs the CPU oracle driver.
python tests/cli_e2e.py [n_sites] [n_samples]   (needs a GPU for the first leg)"""
import os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rng = np.random.default_rng(5)
d = tempfile.mkdtemp(prefix="evo_e2e_")
vcf = os.path.join(d, "big.vcf")
names = ["S%04d" % i for i in range(N)]
gts = np.array(["0/0:30:99", "0/1:25:99", "1/1:33:99", "./.:0:0"])
with open(vcf, "w") as f:
    f.write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(names) + "\n")
    pos = np.cumsum(rng.integers(1, 200, size=S))
    for b in range(0, S, 2000):
        n = min(2000, S - b)
        fr = rng.beta(0.5, 0.5, size=(n, 1))
        a = (rng.random((n, N)) < fr).astype(int) + (rng.random((n, N)) < fr).astype(int)
        a[rng.random((n, N)) < 0.02] = 3
        g = gts[a]
        f.write("".join("scaffold_0\t%d\t.\tA\tG\t999\t.\tDP=9\tGT:DP:GQ\t%s\n" % (pos[b + i], "\t".join(g[i])) for i in range(n)))
sets = os.path.join(d, "sets.txt")
open(sets, "w").write(",".join(names[: N // 2]) + "\n" + ",".join(names[N // 2:]) + "\n")
print("VCF: %d sites x %d samples, %.0f MB" % (S, N, os.path.getsize(vcf) / 1e6))
cli = os.path.join(ROOT, "evo_amd", "cli", "evo")
t0 = time.perf_counter()
r = subprocess.run([cli, "fst", "--vcf=" + vcf, "--sets=" + sets, "-w", "100,100", "-n", "gpu"], cwd=d, stderr=subprocess.PIPE, text=True,
                   env=dict(os.environ, EVO_TIMING="1"), check=True)
print([l for l in r.stderr.splitlines() if l.startswith("[timing]")])
t_gpu = time.perf_counter() - t0
from oracle import oracle as O
od = os.path.join(d, "ora"); os.makedirs(od)
t0 = time.perf_counter()
O.fst_run(vcf, sets, od, run_name="gpu", size=100, step=100)
t_cpu = time.perf_counter() - t0
same = open(os.path.join(d, "gpu_w_100_fst.txt"), "rb").read() == open(os.path.join(od, "gpu_w_100_fst.txt"), "rb").read()
print("GPU front-end: %.2f s (%.0f k sites/s);  CPU oracle driver (1 thread, closed forms): %.2f s (%.0f k sites/s);  outputs identical: %s"
      % (t_gpu, S / t_gpu / 1e3, t_cpu, S / t_cpu / 1e3, same))
