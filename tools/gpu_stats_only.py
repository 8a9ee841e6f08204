"""cfg2 shape: throughput of run_counts alone, run_stats alone and run() over six contexts (one stream each), back to back.
Run through gpurun: python3 tools/gpu_stats_only.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from evo_amd.api import EvoFst

sites = int(os.environ.get("K1_SITES", 1_000_000)); samples = 200; pops = 10
NE = int(os.environ.get("N_ENGINES", 6))
engines = []
for b in range(NE):
    e = EvoFst(0)
    e.set_layout(np.repeat(np.arange(pops, dtype=np.int32), samples // pops), pops)
    e.set_option("pairs_pass", 1)
    e.synth(sites, seed=1234 + b)
    e.set_pairs(); e.set_windows(100, 100)
    e.run_counts(); e.run_stats(); e.sync()
    engines.append(e)
def bench(label, fn, reps=120):
    best = 1e9
    for r in range(5):
        t0 = time.perf_counter()
        for i in range(reps): fn(engines[i % NE])
        for e in engines: e.sync()
        best = min(best, (time.perf_counter() - t0) / reps)
    print("%-28s %.2f us per call" % (label, best * 1e6))
bench("run_counts", lambda e: e.run_counts())
bench("run_stats", lambda e: e.run_stats())
bench("run_counts + run_stats", lambda e: (e.run_counts(), e.run_stats()))
bench("run", lambda e: e.run())
