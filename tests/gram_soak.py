"""Randomised parity soak of the Gram path against the CPU oracle (a checker like the tests, which is why it lives here:
it imports oracle/; pytest does not collect it).

    python3 tests/gram_soak.py [cases] [seed]

Shapes the fixed tests do not enumerate: population counts around the 64-wide blocks, 1 .. 70 chromosomes a
population (the 32 x 32 and 64 x 64 operand tables, the two-stream forms beyond them), missing rates up to 40 %,
skipped sites, ragged windows with empty ones and windows longer than the one-contraction form takes (pieces),
sub-ranges of the windows, both denominators.  Every case must hold the tests' bars.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from evo_amd import EvoFst                                     # noqa: E402
from tests.helpers import options, random_codes, round_robin_pops, n_full_of   # noqa: E402
from tests.test_gpu_gram import _check_gram                    # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    e = EvoFst(0)
    worst = 0.0
    for it in range(cases):
        P = int(rng.choice([2, 3, 5, 17, 33, 63, 64, 65, 70, 129]))
        per = int(rng.choice([1, 2, 3, 5, 8, 15, 16, 20, 31, 35]))          # individuals a population: 2 .. 70 chromosomes
        if P * per > 1400:
            per = max(1, 1400 // P)
        N = P * per
        S = int(rng.integers(200, 2600))
        miss = float(rng.choice([0.0, 0.02, 0.15, 0.4]))
        flags = bool(rng.integers(2))
        sp = round_robin_pops(N, P)
        codes = random_codes(rng, S, sp, P, miss=miss, edge=bool(rng.integers(2)))
        fl = (rng.random(S) < 0.05).astype(np.uint8) if flags else None
        e.set_layout(sp, P)
        e.upload(e.pack_codes(codes), S, site_flags=fl)
        e.set_gram()
        ncut = int(rng.choice([1, 3, max(1, S // 400), max(1, S // 90), max(1, S // 20)]))
        cuts = np.sort(rng.integers(0, S, size=ncut))
        ws = np.concatenate([[0], cuts, [cuts[-1]], [S]]).astype(np.int64)
        e.set_site_windows(ws)
        e.run_counts()
        e.run_gram()
        streams = e.last_gram_streams()
        max_n = int(n_full_of(sp, P).max())
        assert streams == (1 if max_n <= 63 else 2), (it, streams, max_n)     # (pieces where windows are longer than the rule)
        w1 = _check_gram(e, codes, sp, P, fl, ws, "soak")
        with options(e, gram_denominator=2):
            e.run_gram()
            w2 = _check_gram(e, codes, sp, P, fl, ws, "soak_two")
        # a sub-range of the windows gives the same rows
        e.run_gram()
        full = e.gram()
        nw = len(ws) - 1
        a = int(rng.integers(0, nw)); n = int(rng.integers(1, nw - a + 1))
        e.run_gram(a, n)
        part = e.gram()
        for k in full:
            assert np.array_equal(full[k][a:a + n], part[k], equal_nan=True), (it, k)
        worst = max(worst, w1, w2)
        print("case %2d: P %3d x %2d chromosomes, S %4d, miss %.2f, %3d windows (longest %4d), streams %d: worst %.3f / %.3f of the bar"
              % (it, P, int(n_full_of(sp, P).max()), S, miss, nw, int(np.diff(ws).max()), streams, w1, w2), flush=True)
    e.close()
    print("gram soak: %d cases, worst %.3f of the bar" % (cases, worst))


if __name__ == "__main__":
    main()
