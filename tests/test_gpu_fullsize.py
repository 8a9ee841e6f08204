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

from tests.helpers import round_robin_pops, n_full_of, assert_same_bits, options
from oracle import oracle as O

pytestmark = pytest.mark.gpu

S, N, P, SIZE, STEP = 1_000_000, 200, 10, 100, 100


def _run(engine, chunk_tiles, pairs_pass=0):
    with options(engine, chunk_tiles=chunk_tiles, pairs_pass=pairs_pass):
        engine.run()
        return engine.segments(), engine.totals(), engine.boundary()


def test_cfg2_properties(engine):
    _pairs_properties(engine, S, N, P, ((0, 20000), (500_000 - 7, 3000), (S - 4000, 4000)), 0.02)


def test_cfg4_shard_properties(engine):
    """One GPU's share of BASELINE configs[3]: 50 M sites x 2000 samples, 20 populations (190 pairs) over 8 GPUs."""
    _pairs_properties(engine, 6_250_000, 2000, 20, ((0, 8000),), 0.0)


def _pairs_properties(engine, S, N, P, slices, miss):
    from tests import parity_log
    from tests.test_gpu_pairs import _window_abs_sums
    sp = round_robin_pops(N, P)
    engine.set_layout(sp, P)
    engine.set_pairs()
    engine.set_windows(SIZE, STEP)
    engine.reserve(S)
    engine.synth(S, seed=20261110, missing_rate=miss)
    seg_a, tot_a, car_a = _run(engine, 0)      # the library's own choice: the stream pass
    nq = engine.n_pairs
    # the separate kernels with chunk edges inside other windows, and the lock-step one-kernel pass
    for seg_b, tot_b, car_b in (_run(engine, 7, pairs_pass=1), _run(engine, 12, pairs_pass=3)):
      assert np.array_equal(seg_a["n_segments"], seg_b["n_segments"])
      assert np.array_equal(tot_a[:, 3], tot_b[:, 3])
      for q in range(nq):
        n = int(seg_a["n_segments"][q])
        assert n > 1000
        assert np.array_equal(seg_a["end_site"][:n, q], seg_b["end_site"][:n, q])
        assert np.array_equal(seg_a["sum_dxy_k"][:n, q], seg_b["sum_dxy_k"][:n, q])          # integers: exact
        for name in ("sum_num", "sum_den"):
            a, b = seg_a[name][:n, q], seg_b[name][:n, q]
            # two GPU runs against each other: every |num term| and den term of a window is <= its den term, so the
            # window's denominator sum bounds the sum of |terms| of either (each run is within 1e-12 of that of the truth)
            scale = np.maximum(seg_a["sum_den"][:n, q], seg_b["sum_den"][:n, q])
            err = np.abs(a - b)
            parity_log.record("fullsize_%dx%d/chunking_invariance/%s" % (S, N, name), err, 2e-12 * scale)
            assert np.all(err <= 2e-12 * scale)
    for q in range(nq):
        n = int(seg_a["n_segments"][q])
        # additivity: segments + carry = totals
        for col, name in ((0, "sum_num"), (1, "sum_den"), (2, "sum_dxy_k")):
            s = seg_a[name][:n, q].sum() + car_a[q, col]
            assert abs(s - tot_a[q, col]) <= 1e-12 * max(1.0, np.abs(seg_a[name][:n, q]).sum())
        assert n * SIZE + car_a[q, 3] == tot_a[q, 3]

    # oracle parity on slices: the first 20k sites (windows from the start) and two interior slices (per-site values)
    nf = n_full_of(sp, P)
    pi, pj = engine.all_pairs()
    for first, n in slices:
        a0 = first - first % 64
        codes = engine.unpack_codes(engine.download_tiles(a0, first + n - a0), first + n - a0)[first - a0:]
        oalt, onm = O.counts_from_codes(codes, sp, P)
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
                assert k > 50
                assert np.array_equal(gw["end_site"][:k], ow["end_site"])
                m = oinc[:, q] == 1
                with np.errstate(all="ignore"):     # the three terms of calculateFstNumerator per site (test_gpu_pairs.py)
                    power = (oalt[m, pi[q]] / onm[m, pi[q]] - oalt[m, pj[q]] / onm[m, pj[q]]) ** 2
                    num_terms = power + np.abs(power - onum[m, q])
                for name, terms in (("sum_num", num_terms), ("sum_den", oden[m, q]), ("sum_dxy", odxy[m, q])):
                    a, b = gw[name][:k], ow[name]
                    fin = np.isfinite(b)
                    assert (np.isfinite(a) == fin).all()
                    bar = 1e-12 * _window_abs_sums(terms, ow["first_var"], SIZE)[fin]
                    err = np.abs(a[fin] - b[fin])
                    parity_log.record("fullsize_%dx%d/first_windows/%s" % (S, N, name), err, bar)
                    assert np.all(err <= bar)


def _with_chunk_tiles(engine, ct, fn):
    with options(engine, chunk_tiles=ct):
        return fn()


def test_cfg3_full_size_properties(engine):
    """BASELINE configs[2] whole on one GPU (10 M sites x 1000 samples = 5 GB of tiles, 100 populations against all,
    -w 20,10): the window rows must not depend on the chunking at all (every lane sums its deque column oldest ->
    newest, so they are bit-identical), and the first sites agree with the oracle bit for bit."""
    S3, N3, P3, size, step = 10_000_000, 1000, 100, 20, 10
    sp = round_robin_pops(N3, P3)
    engine.set_layout(sp, P3)
    use = np.arange(P3, dtype=np.int32)
    set_sizes = np.full(P3, N3 // P3, np.int64)
    engine.set_against_all(use, set_sizes)
    engine.set_windows(size, step)
    engine.reserve(S3)
    engine.synth(S3, seed=20261110, missing_rate=0.02)

    def run():
        engine.run()
        return engine.global_windows(), engine.included_totals().copy()
    ga, ta = _with_chunk_tiles(engine, 0, run)      # the library's own choice
    gb, tb = _with_chunk_tiles(engine, 9, run)
    assert np.array_equal(ta, tb)
    assert len(ga["end_site"]) == len(gb["end_site"]) > 800_000
    for k in ("start_site", "end_site"):
        assert np.array_equal(ga[k], gb[k])
    for k in ("mean_num", "mean_den"):
        assert_same_bits(ga[k], gb[k], k)

    del gb
    n = 6000
    codes = engine.unpack_codes(engine.download_tiles(0, n), n)
    oalt, onm = O.counts_from_codes(codes, sp, P3)
    alt, nm = engine.counts(0, n)
    assert np.array_equal(alt, oalt) and np.array_equal(nm, onm)
    okept, onum, oden = O.global_site_stats(oalt, onm, use, set_sizes, None)
    gnum, gden, gkept = engine.global_site_stats(0, n)
    assert np.array_equal(gkept, okept)
    k = okept == 1
    with np.errstate(all="ignore"):
        assert_same_bits(gnum.T[k], onum[k], "global num")
        assert_same_bits(gden.T[k], oden[k], "global den")
    ow = O.global_windows(okept, onum, oden, size, step, None)
    r = len(ow["end_site"])
    assert r > 400
    assert np.array_equal(ga["end_site"][:r], ow["end_site"]) and np.array_equal(ga["start_site"][:r], ow["start_site"])
    with np.errstate(all="ignore"):
        gf = ga["mean_num"][:r] / ga["mean_den"][:r]
        gf[(gf < 0) | (ga["mean_den"][:r] == 0)] = 0
    assert_same_bits(gf, ow["fst"], "window Fst")


def test_cfg5_shard_properties(engine):
    """One GPU's share of BASELINE configs[4] (20 M sites x 5000 samples over 8 GPUs = 2.5 M sites, 6.25 GB of tiles;
    500 populations = 124 750 pairs, 15 % missing genotypes): windows of 200 sites equal the sums of their two 100-site
    halves (integers exactly, doubles to 1e-12 of the terms) in the middle and at the end of the shard, and the first
    windows agree with the oracle."""
    S5, N5, P5 = 2_500_000, 5000, 500
    sp = round_robin_pops(N5, P5)
    engine.set_layout(sp, P5)
    engine.set_gram()
    engine.reserve(S5)
    engine.synth(S5, seed=20261110, missing_rate=0.15)
    engine.run_counts()
    engine.set_site_windows(np.arange(0, S5 + 1, 100, dtype=np.int64))
    from tests import parity_log
    n100 = S5 // 100
    fines = []
    for w0 in (12_000, n100 - 20):
        engine.run_gram(w0, 20)
        fines.append(engine.gram())
    engine.run_gram(0, 3)
    head = engine.gram()
    engine.set_site_windows(np.arange(0, S5 + 1, 200, dtype=np.int64))
    for w0, fine in zip((6_000, n100 // 2 - 10), fines):
        engine.run_gram(w0, 10)
        coarse = engine.gram()
        for k in ("sum_dxy_k", "n_included"):
            assert np.array_equal(coarse[k], fine[k][0::2] + fine[k][1::2])
        for k in ("sum_num", "sum_den"):
            a, b = coarse[k], fine[k][0::2] + fine[k][1::2]
            assert (np.isnan(a) == np.isnan(b)).all()
            fin = ~np.isnan(b)
            # num is formed from den's terms too, and every |term| of either is <= the den term: the window's den sum
            # bounds the sum of |terms| (two GPU results against each other: 2e-12)
            with np.errstate(all="ignore"):
                scale = np.abs(fine["sum_den"][0::2]) + np.abs(fine["sum_den"][1::2])
            err = np.abs(a[fin] - b[fin])
            parity_log.record("cfg5_shard/halves_add_up/" + k, err, 2e-12 * scale[fin] + 1e-300)
            assert np.all(err <= 2e-12 * scale[fin] + 1e-300)

    n = 300
    codes = engine.unpack_codes(engine.download_tiles(0, n), n)
    oalt, onm = O.counts_from_codes(codes, sp, P5)
    alt, nm = engine.counts(0, n)
    assert np.array_equal(alt, oalt) and np.array_equal(nm, onm)
    nf = n_full_of(sp, P5)
    pi, pj = np.triu_indices(P5, 1)
    pi = pi.astype(np.int32); pj = pj.astype(np.int32)
    onum, oden, odxy, oinc = O.pair_site_stats(oalt, onm, nf, pi, pj, None)
    odxy_k = np.rint(odxy * (nf[pi] * nf[pj])[None, :].astype(np.float64))
    for w in range(3):
        m = oinc[100 * w:100 * w + 100] == 1
        assert np.array_equal(head["n_included"][w], m.sum(axis=0))
        assert np.array_equal(head["sum_dxy_k"][w], np.where(m, odxy_k[100 * w:100 * w + 100], 0.0).sum(axis=0))
        with np.errstate(all="ignore"):
            rn = np.where(m, onum[100 * w:100 * w + 100], 0.0).sum(axis=0)
            rd = np.where(m, oden[100 * w:100 * w + 100], 0.0).sum(axis=0)
        with np.errstate(all="ignore"):
            an = np.where(m, np.abs(onum[100 * w:100 * w + 100]), 0.0).sum(axis=0)
        for name, a, b, scale in (("sum_num", head["sum_num"][w], rn, an + rd), ("sum_den", head["sum_den"][w], rd, rd)):
            assert (np.isnan(a) == np.isnan(b)).all()
            fin = ~np.isnan(b)
            # the Gram path forms num = den - sum t - sum h: its bar is 1e-12 of the |terms| of both sums
            err = np.abs(a[fin] - b[fin])
            parity_log.record("cfg5_shard/first_windows/" + name, err, 1e-12 * scale[fin] + 1e-300)
            assert np.all(err <= 1e-12 * scale[fin] + 1e-300)

    # Physical windows (10 kb for configs[4]; process_vcf_fst.cpp:589-629): cut from the positions that travel
    # with the shard.  The synthetic positions hold one site per 100 bp, so 10 kb blocks are runs of 100 sites; blocks of
    # 12 345 bases are ragged (123 or 124 sites) and their first two are checked against the oracle's per-site values.
    pos = engine.positions().astype(np.int64)
    assert np.all(np.diff(pos) > 0)
    cut = lambda phys: np.concatenate(([0], np.flatnonzero(np.diff((pos - 1) // phys)) + 1, [S5])).astype(np.int64)
    assert np.array_equal(cut(10_000), np.arange(0, S5 + 1, 100))
    ws = cut(12_345)
    assert set(np.diff(ws[:-1]).tolist()) <= {122, 123, 124, 125} and len(set(np.diff(ws[:-1]).tolist())) > 1 and ws[2] <= n
    engine.set_site_windows(ws)
    engine.run_gram(0, 2)
    bp = engine.gram()
    for w in range(2):
        a_, b_ = int(ws[w]), int(ws[w + 1])
        m = oinc[a_:b_] == 1
        assert np.array_equal(bp["n_included"][w], m.sum(axis=0))
        assert np.array_equal(bp["sum_dxy_k"][w], np.where(m, odxy_k[a_:b_], 0.0).sum(axis=0))
        with np.errstate(all="ignore"):
            rd = np.where(m, oden[a_:b_], 0.0).sum(axis=0)
        fin = ~np.isnan(rd)
        assert (np.isnan(bp["sum_den"][w]) == np.isnan(rd)).all()
        assert np.all(np.abs(bp["sum_den"][w][fin] - rd[fin]) <= 1e-12 * rd[fin] + 1e-300)
