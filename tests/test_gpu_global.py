"""GPU parity: Fst-against-all (evo FstGlobal) through the C ABI vs the CPU oracle.

Bars: kept flags and window rows (start/end sites) identical; per-site numerators / denominators
bit-exact; window means bit-exact too (each lane sums its deque column oldest -> newest exactly as
vector_average does on the reference's deque, evo_FstAgainstAll.cpp:278).
"""
import numpy as np
import pytest

from tests.helpers import random_codes, round_robin_pops, assert_same_bits
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _run(engine, rng, S, N, P, miss, size, step, use=None, set_sizes=None, flags=False, chunk_env=None):
    sp = round_robin_pops(N, P)
    codes = random_codes(rng, S, sp, P, miss=miss)
    engine.set_layout(sp, P)
    use = np.arange(P, dtype=np.int32) if use is None else np.asarray(use, np.int32)
    if set_sizes is None:
        set_sizes = np.array([(sp == k).sum() for k in range(len(use))], np.int64)   # ALL-pops order, sic
    engine.set_against_all(use, set_sizes)
    engine.set_windows(size, step)
    fl = (rng.random(S) < 0.07).astype(np.uint8) if flags else None
    engine.upload(engine.pack_codes(codes), S, site_flags=fl)
    oalt, onm = O.counts_from_codes(codes, sp, P)
    okept, onum, oden = O.global_site_stats(oalt, onm, use, set_sizes, fl)
    ow = O.global_windows(okept, onum, oden, size, step, fl)
    # both window kernels: lane = population with a deque per lane (1), and the two-phase one -- lane = site for the
    # evaluation, thread = window x population for the rows (2: where its LDS ring fits, else the first again)
    for walk in (1, 2):
        engine.set_option("global_walk", walk)
        try:
            engine.run()
            gw = engine.global_windows()
            assert engine.last_global_walk() in ((1,) if walk == 1 else (1, 2))
        finally:
            engine.set_option("global_walk", 0)
        assert len(gw["end_site"]) == len(ow["end_site"]), walk
        assert np.array_equal(gw["end_site"], ow["end_site"]), walk
        assert np.array_equal(gw["start_site"], ow["start_site"]), walk
        with np.errstate(all="ignore"):
            gf = gw["mean_num"] / gw["mean_den"]
            gf[(gf < 0) | (gw["mean_den"] == 0)] = 0
        assert_same_bits(gf, ow["fst"], "window Fst (global_walk %d)" % walk)
    engine.run()
    gnum, gden, gkept = engine.global_site_stats()
    assert np.array_equal(gkept, okept)
    k = okept == 1
    with np.errstate(all="ignore"):
        assert_same_bits(gnum.T[k], onum[k], "global num")
        assert_same_bits(gden.T[k], oden[k], "global den")
    gw = engine.global_windows()
    assert len(gw["end_site"]) == len(ow["end_site"])
    assert np.array_equal(gw["end_site"], ow["end_site"])
    assert np.array_equal(gw["start_site"], ow["start_site"])
    with np.errstate(all="ignore"):
        gf = gw["mean_num"] / gw["mean_den"]
        gf[(gf < 0) | (gw["mean_den"] == 0)] = 0
    assert_same_bits(gf, ow["fst"], "window Fst")
    tot = engine.included_totals()
    n_used = S - (int(fl.sum()) if fl is not None else 0)
    assert tot[0] == n_used and tot[1] == okept.sum()
    return len(ow["end_site"])


@pytest.mark.parametrize("S,N,P,miss,size,step", [(3000, 60, 6, 0.05, 20, 10), (2000, 40, 4, 0.3, 10, 10),
                                                  (5000, 200, 20, 0.1, 20, 10), (1500, 30, 3, 0.6, 7, 3),
                                                  (4000, 300, 100, 0.05, 20, 10), (1500, 900, 3, 0.2, 20, 10),
                                                  # 800 chromosomes, half of them missing: complement counts far from the total take the
                                                  # real-division path of the kernel's reciprocal table
                                                  (1200, 400, 4, 0.5, 20, 10), (1200, 400, 8, 0.35, 10, 5),
                                                  (600, 20, 2, 0.1, 1, 1), (1500, 40, 4, 0.1, 50, 7), (900, 40, 4, 0.1, 5, 20),
                                                  # no missing allele / a few: whole tiles take the kernel's constant-divisor evaluation
                                                  (3000, 60, 6, 0.0, 20, 10), (6000, 120, 12, 0.0005, 20, 10)])
def test_global_windows(engine, S, N, P, miss, size, step):
    rows = _run(engine, np.random.default_rng(S + P), S, N, P, miss, size, step, flags=True)
    assert rows > 0


def test_global_subset_and_threshold(engine):
    """populationsToUse is a subset (an Outgroup exists) and the setSizes mismatch makes the
    0.8 rule fail for some populations."""
    rng = np.random.default_rng(11)
    P = 5
    use = [0, 2, 3, 4]                      # pop 1 plays 'Outgroup'
    sizes = np.array([12, 12, 40, 12], np.int64)  # third entry too large -> threshold fails often
    _run(engine, rng, 2500, 60, P, 0.15, 20, 10, use=use, set_sizes=sizes)


def test_global_small_chunks(engine):
    from tests.helpers import options
    with options(engine, chunk_tiles=1):
        _run(engine, np.random.default_rng(5), 3000, 40, 4, 0.2, 50, 25, flags=True)
        _run(engine, np.random.default_rng(6), 1000, 40, 4, 0.9, 20, 10)   # few kept sites: long replays
