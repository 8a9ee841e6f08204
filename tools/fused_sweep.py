"""Timing sweeps of the fused pairs pass (tools only; uses the -DEVOFST_EXPERIMENTS build, whose EVOFST_DEBUG_FUSED
switches leave out phases -- the results of such runs are wrong on purpose).

    EVOFST_LIB=evo_amd/libevofst_exp.so python tools/fused_sweep.py [cfg2|cfg4s] [sites]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from evo_amd import EvoFst  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
S, N, P = {"cfg2": (1_000_000, 200, 10), "cfg4s": (6_250_000, 2000, 20)}[name]
if len(sys.argv) > 2:
    S = int(sys.argv[2])
e = EvoFst(0)
e.set_layout((np.arange(N) % P).astype(np.int32), P)
e.set_pairs()
e.set_windows(100, 100)
e.reserve(S)
e.synth(S, seed=20261110, missing_rate=float(os.environ.get("SWEEP_MISS", "0")))
e.sync()


def timed(env, steps=20):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        for _ in range(3):
            e.run()
        e.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            e.run()
        e.sync()
        wall = (time.perf_counter() - t0) / steps * 1e6
        e.set_profiling(True)
        ks = []
        for _ in range(5):
            e.run(); e.sync()
            try:
                ks.append(e.kernel_ms(4) * 1e3)
            except Exception:
                ks.append(sum(e.kernel_ms(k) for k in (0, 1, 2)) * 1e3)
        e.set_profiling(False)
        return "%8.1f us wall %8.1f us kernel" % (wall, float(np.mean(ks)))
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


print("%s: %d sites x %d samples, %d pops" % (name, S, N, P))
for sub in (32, 16):
    for ct in (8, 16):
        for dbg, what in ((0, "all"), (1, "no waits"), (3, "A + census only")):
            print("sub %2d chunk_tiles %2d  %-26s %s" % (sub, ct, what, timed({"EVOFST_FUSED_SUB": sub, "EVOFST_CHUNK_TILES": ct, "EVOFST_DEBUG_FUSED": dbg})), flush=True)
print("unfused %s" % timed({"EVOFST_NO_FUSE": 1}))
e.close()
