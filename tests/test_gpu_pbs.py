"""GPU parity: PBS per site and trio (evo_PBS.cpp:69-98, 304-322) through the C ABI vs the CPU oracle.
`used` flags identical; values within 1e-12 relative (the device's log and glibc's may differ in the last bit;
everything before the log is the reference's operation order).  The oracle's PBS functions are a restatement
without a reference build or fixture behind them: parity unpinned at that level."""
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
