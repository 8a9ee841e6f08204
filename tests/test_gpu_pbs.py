"""GPU parity: PBS per site and trio (evo_PBS.cpp:69-98, 304-322) and ABS per site and quartet (evo_ABS.cpp:58-106,
243-262) through the C ABI vs the CPU oracle.  `used` flags identical; values within 1e-12 relative (the device's log
and glibc's may differ in the last bit; everything before the log is the reference's operation order).  The oracle's
per-site functions are pinned bit for bit on the reference's own calculatePBSfromAFs / calculateABSfromAFs
(tests/test_oracle.py, oracle/_ref)."""
import numpy as np
import pytest

from tests.helpers import random_codes, round_robin_pops
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("S,N,P,miss,flags", [(3000, 60, 6, 0.1, True), (2000, 24, 4, 0.5, False), (1500, 900, 3, 0.05, True)])
def test_pbs_site_stats(engine, S, N, P, miss, flags):
    rng = np.random.default_rng(S + P)
    sp = round_robin_pops(N, P)
    codes = random_codes(rng, S, sp, P, miss=miss)
    engine.set_layout(sp, P)
    fl = (rng.random(S) < 0.05).astype(np.uint8) if flags else None
    engine.upload(engine.pack_codes(codes), S, site_flags=fl)
    engine.run_counts()
    trios = np.array([[0, 1, 2], [2, 1, 0], [P - 1, 0, 1], [1, P - 2, P - 1]], np.int32)
    pbs, used = engine.pbs_site_stats(trios)
    oalt, onm = O.counts_from_codes(codes, sp, P)
    opbs, oused = O.pbs_site_stats(oalt, onm, trios, fl)
    assert np.array_equal(used, oused)
    assert used.sum() > S
    with np.errstate(all="ignore"):
        nan_same = np.isnan(pbs) == np.isnan(opbs)
        assert nan_same.all()
        fin = np.isfinite(opbs)
        assert np.array_equal(np.isinf(pbs), np.isinf(opbs))
        assert np.all(np.abs(pbs[fin] - opbs[fin]) <= 1e-12 * np.maximum(np.abs(opbs[fin]), 1e-300))


@pytest.mark.parametrize("S,N,P,miss,flags", [(3000, 60, 6, 0.1, True), (2500, 16, 4, 0.5, False), (1500, 900, 5, 0.05, True)])
def test_abs_site_stats(engine, S, N, P, miss, flags):
    rng = np.random.default_rng(7 * S + P)
    sp = round_robin_pops(N, P)
    codes = random_codes(rng, S, sp, P, miss=miss)
    engine.set_layout(sp, P)
    fl = (rng.random(S) < 0.05).astype(np.uint8) if flags else None
    engine.upload(engine.pack_codes(codes), S, site_flags=fl)
    engine.run_counts()
    quartets = np.array([[0, 1, 2, 3], [3, 2, 1, 0], [P - 1, 0, 2, 1], [1, 3, 0, P - 1], [2, 0, 3, 1]], np.int32)
    got, used = engine.abs_site_stats(quartets)
    oalt, onm = O.counts_from_codes(codes, sp, P)
    want, oused = O.abs_site_stats(oalt, onm, quartets, fl)
    assert np.array_equal(used, oused)
    assert used.sum() > S
    with np.errstate(all="ignore"):
        assert np.array_equal(np.isnan(got), np.isnan(want))
        assert np.array_equal(np.isinf(got), np.isinf(want))
        fin = np.isfinite(want)
        # ABS is a difference of sums of branch lengths: the bound is relative to the branch lengths (O(1) and
        # larger), not to a difference that may cancel to almost nothing
        assert np.all(np.abs(got[fin] - want[fin]) <= 1e-12 * np.maximum(np.abs(want[fin]), 1.0))
        big = fin & (np.abs(want) > 1e-3)
        assert big.sum() > S and np.all(np.abs(got[big] - want[big]) <= 1e-12 * np.abs(want[big]))
    if miss >= 0.5:
        assert np.isnan(want).any()          # single-allele populations are in the case
