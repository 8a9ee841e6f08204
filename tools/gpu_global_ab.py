"""A/B of the two against-all window kernels on the cfg3 shard (evofst_set_option global_walk 1 / 2)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from evo_amd import EvoFst
S, N, P = 2_000_000, 1000, 100
e = EvoFst(0)
e.set_layout((np.arange(N) % P).astype(np.int32), P)
e.set_against_all(np.arange(P, dtype=np.int32), np.full(P, N // P, np.int64))
e.set_windows(20, 10)
e.reserve(S)
e.synth(S, seed=20261110, missing_rate=0.0)
e.sync()
e.set_profiling(True)
for walk in (1, 2, 1, 2):
    e.set_option("global_walk", walk)
    ms = []
    for _ in range(6):
        e.run(); e.sync()
        ms.append((e.kernel_ms(0), e.kernel_ms(2)))
    print("global_walk", walk, "k1 %.3f ms, windows %.3f ms" % tuple(np.median(np.array(ms[1:]), axis=0)), flush=True)
e.close()
