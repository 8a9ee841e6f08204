"""K1 alone (evofst_run_counts) on the cfg2 shape, six data sets in turn so the Infinity Cache cannot serve them.
With the -DEVOFST_EXPERIMENTS build and EVOFST_DEBUG_K1 set, nothing consumes what K1 leaves out (only run_counts runs).
Run through gpurun:  EVOFST_LIB=$PWD/evo_amd/libevofst_exp.so EVOFST_DEBUG_K1=1 python3 tools/gpu_k1_only.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from evo_amd.api import EvoFst

sites = int(os.environ.get("K1_SITES", 1_000_000)); samples = int(os.environ.get("K1_SAMPLES", 200)); pops = int(os.environ.get("K1_POPS", 10))
miss = float(os.environ.get("K1_MISS", 0.0))
ct = int(os.environ.get("K1_CHUNK_TILES", 0))
one_stream = os.environ.get("K1_ONE_STREAM") is not None      # the launches one after the other instead of on a stream each
if one_stream:
    import torch
    shared = torch.cuda.Stream()
engines = []
for b in range(6):
    e = EvoFst(0)
    e.set_layout(np.repeat(np.arange(pops, dtype=np.int32), samples // pops), pops)
    e.set_option("pairs_pass", 0)
    if ct: e.set_option("chunk_tiles", ct)
    e.synth(sites, seed=1234 + b, missing_rate=miss)
    e.set_pairs(); e.set_windows(100, 100)
    if one_stream: e.set_stream(shared.cuda_stream)
    e.run_counts(); e.sync()
    engines.append(e)
reps = 60
best = 1e9
for r in range(5):
    t0 = time.perf_counter()
    for i in range(reps):
        engines[i % 6].run_counts()
    for e in engines: e.sync()
    best = min(best, (time.perf_counter() - t0) / reps)
print("k1 only: debug %s chunk_tiles %s  %.2f us per launch (back to back, %s)" % (os.environ.get("EVOFST_DEBUG_K1", "0"), ct or "auto", best * 1e6, "all on one stream" if one_stream else "one stream each"))
