#!/usr/bin/env python3
"""Timing of the SURVEY 8(f) consumer kernels on the cfg2 shard shape (200 samples, 10 populations, missing 2 %):
PBS and ABS per site, outgroup-to-ingroup dxy per window, per-individual pairwise differences (200 x 201 / 2 pairs).  Wall time of the C-ABI call (staging + kernel + copy back
to pageable host memory); run under `rocprofv3 --kernel-trace --stats` for the kernels' own durations.
python3 tools/f_rows_bench.py [n_sites]"""
import itertools, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from evo_amd.api import EvoFst

S = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
N, P = 200, 10
e = EvoFst(0)
e.set_layout(np.arange(N) % P, P)
e.reserve(S)
e.synth(S, missing_rate=0.02)
e.run_counts(); e.sync()
trios = np.array(list(itertools.combinations(range(P), 3))[:24], np.int32)
quartets = np.array(list(itertools.combinations(range(P), 4))[:24], np.int32)
wo = np.arange(0, S + 1, 256, dtype=np.int64)
out_idx, in_idx = [0, 1, 2], [3, 4, 5, 6, 7, 8, 9]


def timed(name, fn, units, unit, out_bytes):
    fn()
    t = []
    for _ in range(5):
        t0 = time.perf_counter(); fn(); t.append(time.perf_counter() - t0)
    dt = min(t)
    print("%-22s %8.2f ms per call  %.3g %s/s  (results: %.0f MB to the host)" % (name, dt * 1e3, units / dt, unit, out_bytes / 1e6))


timed("pbs_site_stats", lambda: e.pbs_site_stats(trios), S * len(trios), "site*trio", S * len(trios) * 25)
timed("abs_site_stats", lambda: e.abs_site_stats(quartets), S * len(quartets), "site*quartet", S * len(quartets) * 17)
timed("dist_windows", lambda: e.dist_windows(wo, out_idx, in_idx), S * len(out_idx) * len(in_idx), "site*pair",
      (len(wo) - 1) * len(out_idx) * (len(in_idx) * 16 + 8))
blocks = np.arange(0, S + 1, 4096, dtype=np.int64)
timed("indiv_diffs", lambda: e.indiv_diffs(blocks), S * (N * (N + 1) // 2), "site*pair", (len(blocks) - 1) * (N * (N + 1) // 2) * 20)
timed("indiv_diffs 1 block", lambda: e.indiv_diffs([0, S]), S * (N * (N + 1) // 2), "site*pair", (N * (N + 1) // 2) * 20)
