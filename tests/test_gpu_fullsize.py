"""Full-size checks at BASELINE configs[1] (1 M sites x 200 samples, 10 pops = 45 pairs, -w 100,100), where the
oracle on every site would take too long for a unit test: size-independent properties of the window machinery
plus oracle parity on sampled slices.

  * chunking invariance: the segment table must not depend on how the sites are cut into chunks
    (window membership bit-exact, sums to 1e-12 of the window's terms);
  * additivity: complete segments + the carry behind the last one = whole-shard totals;
  * counts, per-site statistics and window sums on sampled site slices against the oracle.
"""
import os

import numpy as np
import pytest

from tests.helpers import round_robin_pops, n_full_of, assert_same_bits
from oracle import oracle as O

pytestmark = pytest.mark.gpu

S, N, P, SIZE, STEP = 1_000_000, 200, 10, 100, 100


def _run(engine, chunk_tiles):
    old = os.environ.get("EVOFST_CHUNK_TILES")
    os.environ["EVOFST_CHUNK_TILES"] = str(chunk_tiles)
    try:
        engine.run()
        return engine.segments(), engine.totals(), engine.boundary()
    finally:
        if old is None:
            os.environ.pop("EVOFST_CHUNK_TILES", None)
        else:
            os.environ["EVOFST_CHUNK_TILES"] = old


def test_cfg2_properties(engine):
    sp = round_robin_pops(N, P)
    engine.set_layout(sp, P)
    engine.set_pairs()
    engine.set_windows(SIZE, STEP)
    engine.reserve(S)
    engine.synth(S, seed=20261110, missing_rate=0.02)
    seg_a, tot_a, car_a = _run(engine, 4)
    seg_b, tot_b, car_b = _run(engine, 7)      # chunk edges fall inside other windows
    nq = engine.n_pairs
    assert np.array_equal(seg_a["n_segments"], seg_b["n_segments"])
    assert np.array_equal(tot_a[:, 3], tot_b[:, 3])
    for q in range(nq):
        n = int(seg_a["n_segments"][q])
        assert n > 1000
        assert np.array_equal(seg_a["end_site"][:n, q], seg_b["end_site"][:n, q])
        assert np.array_equal(seg_a["sum_dxy_k"][:n, q], seg_b["sum_dxy_k"][:n, q])          # integers: exact
        for name in ("sum_num", "sum_den"):
            a, b = seg_a[name][:n, q], seg_b[name][:n, q]
            assert np.all(np.abs(a - b) <= 1e-12 * SIZE)                                       # |term| <= 1
        # additivity: segments + carry = totals
        for col, name in ((0, "sum_num"), (1, "sum_den"), (2, "sum_dxy_k")):
            s = seg_a[name][:n, q].sum() + car_a[q, col]
            assert abs(s - tot_a[q, col]) <= 1e-12 * max(1.0, np.abs(seg_a[name][:n, q]).sum())
        assert n * SIZE + car_a[q, 3] == tot_a[q, 3]

    # oracle parity on slices: the first 20k sites (windows from the start) and two interior slices (per-site values)
    tiles = engine.download_tiles()
    codes = engine.unpack_codes(tiles, S)
    nf = n_full_of(sp, P)
    pi, pj = engine.all_pairs()
    for first, n in ((0, 20000), (500_000 - 7, 3000), (S - 4000, 4000)):
        oalt, onm = O.counts_from_codes(codes[first:first + n], sp, P)
        alt, nm = engine.counts(first, n)
        assert np.array_equal(alt, oalt) and np.array_equal(nm, onm)
        onum, oden, odxy, oinc = O.pair_site_stats(oalt, onm, nf, pi, pj)
        num, den, dxy, inc = engine.site_stats(first, n)
        assert np.array_equal(inc.T, oinc)
        with np.errstate(all="ignore"):
            assert_same_bits(num.T, onum, "num"); assert_same_bits(den.T, oden, "den"); assert_same_bits(dxy.T, odxy, "dxy")
        if first == 0:
            from evo_amd import compose_snp_windows
            for q in (0, 17, nq - 1):
                ow = O.snp_windows(onum, oden, odxy, oinc, q, SIZE, STEP)
                gw = compose_snp_windows(seg_a, q, SIZE, STEP, nf[pi[q]], nf[pj[q]])
                k = len(ow["end_site"])
                assert k > 100
                assert np.array_equal(gw["end_site"][:k], ow["end_site"])
                for name in ("sum_num", "sum_den", "sum_dxy"):
                    a, b = gw[name][:k], ow[name]
                    fin = np.isfinite(b)
                    assert (np.isfinite(a) == fin).all()
                    assert np.all(np.abs(a[fin] - b[fin]) <= 1e-12 * SIZE)
