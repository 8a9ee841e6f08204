#!/usr/bin/env python3
"""Host packer throughput: VCF text lines -> packed tiles, lines per second for 1..T threads.
No GPU needed.  python tools/host_pack_bench.py [n_samples] [n_lines]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evo_amd import EvoFst
from evo_amd.api import MODE_FST

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
S = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
rng = np.random.default_rng(1)
gt = np.array([b"0/0:35:99", b"0/1:28:99", b"1/1:31:99", b"./.:0:0"], dtype=object)
base = []
for i in range(256):                                     # 256 distinct lines, repeated
    g = gt[rng.choice(4, size=N, p=[0.55, 0.25, 0.17, 0.03])]
    base.append(b"\t".join([b"chr1", str(1000 + 97 * i).encode(), b".", b"A", b"G", b"999", b".", b"DP=10", b"GT:DP:GQ"] + list(g)))
lines = [base[i & 255] for i in range(S)]
nbytes = sum(len(l) for l in base) / 256 * S
e = EvoFst(-1)
e.set_layout((np.arange(N) % 10).astype(np.int32), 10)
for th in (1, 2, 4, 8, 16):
    if th > (os.cpu_count() or 1):
        break
    t0 = time.perf_counter()
    e.pack_vcf_lines(lines, MODE_FST, N, None, n_threads=th)
    dt = time.perf_counter() - t0
    print("threads %2d: %.2f M lines/s, %.0f MB/s of VCF text (%d samples)" % (th, S / dt / 1e6, nbytes / dt / 1e6, N))
