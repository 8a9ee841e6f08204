"""Random shapes through the stream pass against the oracle (the checks of tests/test_gpu_pairs.py::_check_windows_once):
layouts, shard lengths (down to less than a chunk), missing-call rates, skipped sites, windows, producer counts and the
library's own or a given chunk size.  Not a test of the suite -- a soak run for a change of evofst_stream.hip.
Run through gpurun:  python3 tests/stream_stress.py [cases] [seed]   (a checker that calls the oracle: it lives under tests/; pytest does not collect it)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from evo_amd import EvoFst
from tests.helpers import options, random_codes, round_robin_pops
from tests.test_gpu_pairs import _check_windows_once

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
t0 = time.time()
for case in range(n_cases):
    P = int(rng.choice([2, 3, 5, 8, 13, 20, 23]))
    N = int(P * rng.integers(2, 9) + rng.integers(0, P))
    S = int(rng.choice([50, 700, 1792, 1793, 9_000, 40_000, 120_000]))
    miss = float(rng.choice([0.0, 0.0, 0.0005, 0.02]))
    size = int(rng.choice([1, 7, 20, 100, 1000])); step = int(rng.choice([size, max(1, size // 2), max(1, size // 5)]))
    prod = int(rng.choice([0, 1, 3, 4, 5, 8]))
    ct = int(rng.choice([0, 0, 0, prod * 2 if prod else 8, 16]))
    if prod and ct:
        ct = max(prod, ct // prod * prod)
    flags = rng.random() < 0.5
    sp = round_robin_pops(N, P)
    codes = random_codes(rng, S, sp, P, miss=miss, edge=bool(rng.random() < 0.3))
    fl = (rng.random(S) < 0.03).astype(np.uint8) if flags else None
    e = EvoFst(0)
    try:
        e.set_layout(sp, P)
        e.upload(e.pack_codes(codes), S, site_flags=fl)
        e.set_pairs()
        with options(e, pairs_pass=2, stream_producers=prod, chunk_tiles=ct):
            _check_windows_once(e, sp, codes, fl if fl is not None else np.zeros(S, np.uint8), P, size, step, None, "stress")
            took = e.last_pairs_pass()
            # and a second pass over the same data (launch parity of the look-back tables, the marked-chunk feedback)
            _check_windows_once(e, sp, codes, fl if fl is not None else np.zeros(S, np.uint8), P, size, step, None, "stress")
        print("case %2d ok: P %2d N %3d S %6d miss %.4f -w %d,%d producers %d chunk_tiles %d flags %d -> pass %d  (%.0f s)" %
              (case, P, N, S, miss, size, step, prod, ct, flags, took, time.time() - t0), flush=True)
    finally:
        e.close()
print("all %d cases passed" % n_cases)
