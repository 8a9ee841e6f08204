"""GPU parity: HIP path (through the C ABI) vs the CPU oracle, pairs mode.

Bars: allele counts, include flags and window membership bit-exact; per-site doubles bit-exact
(the kernels keep the reference's operation order, -ffp-contract=off); window sums within
1e-12 relative to the sum of |terms| (summation order differs at chunk boundaries only).
"""
import math

import numpy as np
import pytest

from tests.helpers import random_codes, round_robin_pops, n_full_of, assert_same_bits, options, PAIRS_PASSES
from oracle import oracle as O

pytestmark = pytest.mark.gpu

REL_TOL = 1e-12


def _setup(engine, rng, S, N, P, miss, flags=False):
    sp = round_robin_pops(N, P)
    codes = random_codes(rng, S, sp, P, miss=miss)
    engine.set_layout(sp, P)
    tiles = engine.pack_codes(codes)
    fl = None
    if flags:
        fl = (rng.random(S) < 0.05).astype(np.uint8)
    engine.upload(tiles, S, site_flags=fl)
    return sp, codes, fl


@pytest.mark.parametrize("S,N,P,miss", [(1000, 40, 4, 0.05), (777, 200, 10, 0.0), (2048, 64, 3, 0.3),
                                        (130, 600, 2, 0.1), (5000, 90, 9, 0.15)])
def test_counts_bit_exact(engine, S, N, P, miss):
    rng = np.random.default_rng(S + N)
    sp, codes, _ = _setup(engine, rng, S, N, P, miss)
    engine.run_counts()
    alt, nm = engine.counts()
    oalt, onm = O.counts_from_codes(codes, sp, P)
    assert np.array_equal(alt, oalt)
    assert np.array_equal(nm, onm)


def test_counts_with_unassigned_samples(engine):
    rng = np.random.default_rng(5)
    N, P, S = 50, 3, 300
    sp = round_robin_pops(N, P)
    sp[::7] = -1
    codes = random_codes(rng, S, sp, P, miss=0.1, edge=False)
    engine.set_layout(sp, P)
    engine.upload(engine.pack_codes(codes), S)
    engine.run_counts()
    alt, nm = engine.counts()
    oalt, onm = O.counts_from_codes(codes, sp, P)
    assert np.array_equal(alt, oalt) and np.array_equal(nm, onm)


@pytest.mark.parametrize("S,N,P,miss,flags", [(1500, 40, 4, 0.1, False), (900, 200, 10, 0.02, True),
                                              (640, 24, 6, 0.4, True)])
def test_site_stats_bit_exact(engine, S, N, P, miss, flags):
    rng = np.random.default_rng(S * 3 + P)
    sp, codes, fl = _setup(engine, rng, S, N, P, miss, flags)
    engine.set_pairs()
    engine.run_counts()
    num, den, dxy, inc = engine.site_stats()
    pi, pj = engine.all_pairs()
    oalt, onm = O.counts_from_codes(codes, sp, P)
    nf = n_full_of(sp, P)
    onum, oden, odxy, oinc = O.pair_site_stats(oalt, onm, nf, pi, pj, fl)
    assert np.array_equal(inc.T, oinc)
    with np.errstate(all="ignore"):
        assert_same_bits(num.T, onum, "num")
        assert_same_bits(den.T, oden, "den")
        assert_same_bits(dxy.T, odxy, "dxy")
    het, hn, pis = engine.pop_site_div()
    ohet, ohn, opi = O.pop_site_div(oalt, onm, nf)
    assert_same_bits(het.T, ohet, "het")
    assert_same_bits(hn.T, ohn, "hetNei")
    assert_same_bits(pis.T, opi, "pi")
    # census == number of included sites
    assert np.array_equal(engine.included_totals(), oinc.sum(axis=0))


def _window_abs_sums(terms, first_var, size):
    """Sum of |terms| over each window: terms = the pair's included sites in order, a window covers the `size` of them
    that start at 1-based rank first_var."""
    ca = np.concatenate(([0.0], np.cumsum(np.abs(np.nan_to_num(terms, nan=0.0, posinf=0.0, neginf=0.0)))))
    lo = np.asarray(first_var, np.int64) - 1
    return ca[lo + size] - ca[lo]


def _check_windows(engine, sp, codes, fl, P, size, step, pairs=None, tag="snp_windows"):
    """Every form of the pairs pass against the oracle: counts + scan + walk + stitch as separate kernels, the lock-step
    one-kernel pass and the stream pass (evofst_run's choice where the shape allows; shapes a one-kernel pass does not
    take fall back to the separate kernels)."""
    for suffix, pass_ in PAIRS_PASSES:
        with options(engine, pairs_pass=pass_):
            _check_windows_once(engine, sp, codes, fl, P, size, step, pairs, tag + suffix)


def _check_windows_once(engine, sp, codes, fl, P, size, step, pairs, tag):
    from tests import parity_log
    pi, pj = engine.all_pairs() if pairs is None else pairs
    nf = n_full_of(sp, P)
    oalt, onm = O.counts_from_codes(codes, sp, P)
    onum, oden, odxy, oinc = O.pair_site_stats(oalt, onm, nf, pi, pj, fl)
    engine.set_windows(size, step)
    engine.run()
    seg = engine.segments()
    tot = engine.totals()
    from evo_amd import compose_snp_windows
    for q in range(len(pi)):
        ow = O.snp_windows(onum, oden, odxy, oinc, q, size, step)
        gw = compose_snp_windows(seg, q, size, step, nf[pi[q]], nf[pj[q]])
        assert np.array_equal(gw["first_var"], ow["first_var"]), q
        assert np.array_equal(gw["end_site"], ow["end_site"]), q
        m = oinc[:, q] == 1
        absn = absd = None
        # The terms of a window's numerator sum are the three terms of calculateFstNumerator (process_vcf_fst.h:19-25) at
        # every site: power = (p1 - p2)^2, fraction1, fraction2 (num = power - fraction1 - fraction2); those of the
        # denominator and of dxy are the per-site values themselves (all >= 0).
        with np.errstate(all="ignore"):
            power = (oalt[m, pi[q]] / onm[m, pi[q]] - oalt[m, pj[q]] / onm[m, pj[q]]) ** 2
            num_terms = power + np.abs(power - onum[m, q])
        for name, terms in (("sum_num", num_terms), ("sum_den", oden[m, q]), ("sum_dxy", odxy[m, q])):
            a, b = gw[name], ow[name]
            ok_nan = np.isnan(a) == np.isnan(b)
            assert ok_nan.all(), (q, name)
            fin = ~np.isnan(b)
            # the bar of the north star: 1e-12 of the sum of |terms| of that very window
            scale = _window_abs_sums(terms, ow["first_var"], size)
            if name == "sum_num": absn = scale
            if name == "sum_den": absd = scale
            err = np.abs(a[fin] - b[fin])
            parity_log.record("%s/%s" % (tag, name), err, REL_TOL * scale[fin])
            assert np.all(err <= REL_TOL * scale[fin]), (q, name, float(np.max(err / np.maximum(scale[fin], 1e-300))))
        # Fst of a window = sum num / sum den: same clamp and NaN behaviour, and 1e-12 * sum|num terms| / sum den
        fa, fb = gw["fst"], ow["fst"]
        assert (np.isnan(fa) == np.isnan(fb)).all()
        fin = np.isfinite(fb) & np.isfinite(ow["sum_num"]) & (ow["sum_den"] > 0)
        with np.errstate(all="ignore"):
            fbar = REL_TOL * (absn + absd * np.abs(fb)) / ow["sum_den"]       # d(n/d) <= (dn + |f| dd) / d
        err = np.abs(fa[fin] - fb[fin])
        parity_log.record("%s/fst" % tag, err, fbar[fin])
        assert np.all(err <= fbar[fin]), (q, "fst")
        # whole-shard sums
        assert tot[q, 3] == m.sum()
        if m.any() and np.isfinite(onum[m, q]).all():
            for col, vals, terms in ((0, onum[m, q], num_terms), (1, oden[m, q], oden[m, q])):
                err = abs(tot[q, col] - vals.sum()); bar = REL_TOL * np.abs(terms).sum() + 1e-300
                parity_log.record("%s/totals" % tag, err, bar)
                assert err <= bar


@pytest.mark.parametrize("S,N,P,miss,size,step", [(6000, 40, 4, 0.05, 100, 100), (5000, 60, 5, 0.1, 20, 10),
                                                  (3000, 30, 3, 0.0, 7, 7), (4000, 40, 4, 0.2, 30, 12),
                                                  (700, 40, 4, 0.05, 1000, 1000),
                                                  (500, 20, 2, 0.1, 1, 1), (800, 40, 4, 0.1, 3, 1), (900, 40, 4, 0.05, 64, 32),
                                                  (2500, 900, 3, 0.1, 50, 25),       # 600 chromosomes per population: 32-bit packed counts
                                                  (1200, 2100, 2, 0.02, 40, 40)])    # 2100 per population: beyond the reciprocal table
def test_snp_windows(engine, S, N, P, miss, size, step):
    rng = np.random.default_rng(S + size)
    sp, codes, fl = _setup(engine, rng, S, N, P, miss, flags=True)
    engine.set_pairs()
    _check_windows(engine, sp, codes, fl, P, size, step)


@pytest.mark.parametrize("S,N,P,size,step,dirty", [(6000, 200, 10, 100, 100, 0), (5000, 40, 4, 20, 10, 0), (4096, 60, 5, 7, 7, 40),
                                                     (3000, 250, 2, 50, 25, 3), (2500, 40, 4, 1, 1, 0), (2500, 36, 3, 3, 1, 2),
                                                     (1500, 100, 25, 40, 40, 0),      # 25 populations: the lean walk's LDS block is above 48 KB
                                                     (1500, 180, 90, 40, 40, 0)])     # 90 populations, 4005 pairs: beyond the walk, K2
def test_snp_windows_without_missing_alleles(engine, S, N, P, size, step, dirty):
    """No missing allele (but for `dirty` sites spread over the input): half tiles without one take the exact integer sums
    (v_dot4_u32_u8) instead of the per-population tables, the others the tables, and segments mix both."""
    rng = np.random.default_rng(S + size + dirty)
    sp = round_robin_pops(N, P)
    codes = random_codes(rng, S, sp, P, miss=0.0, edge=False)
    codes[0, :] = 0; codes[1, :] = 1                      # monomorphic / fixed sites
    for s_ in rng.choice(S, dirty, replace=False):
        codes[s_, rng.choice(2 * N, 3, replace=False)] = 2
    engine.set_layout(sp, P)
    engine.upload(engine.pack_codes(codes), S)
    if P == 25:                                            # a short pair list: a wave per chunk with 25 populations' counts
        pi = np.array([0, 3, 24, 17, 5, 23, 20, 1, 2, 11], np.int32)
        pj = np.array([24, 4, 1, 18, 21, 22, 19, 13, 9, 12], np.int32)
        engine.set_pairs(pi, pj)
        _check_windows(engine, sp, codes, None, P, size, step, pairs=(pi, pj), tag="snp_windows_clean")
        return
    engine.set_pairs()
    _check_windows(engine, sp, codes, None, P, size, step, tag="snp_windows_clean")


@pytest.mark.parametrize("S,N,P,dirty", [(200_000, 40, 4, 900), (100_000, 48, 12, 500)])
def test_snp_windows_clean_and_dirty_chunks(engine, S, N, P, dirty):
    """One-tile chunks, a few hundred of them with a missing allele: clean_walk_kernel takes the chunks K1 left unmarked,
    the general walk strides over the marked ones with a grid smaller than the number of regions, and every window
    straddles chunks of both kinds (6 pairs: a wave per chunk; 66 pairs: a workgroup per chunk)."""
    rng = np.random.default_rng(S + dirty)
    sp = round_robin_pops(N, P)
    codes = random_codes(rng, S, sp, P, miss=0.0, edge=False)
    for s_ in rng.choice(S, dirty, replace=False):
        codes[s_, rng.choice(2 * N, 2, replace=False)] = 2
    engine.set_layout(sp, P)
    engine.upload(engine.pack_codes(codes), S)
    engine.set_pairs()
    with options(engine, chunk_tiles=1):
        _check_windows(engine, sp, codes, None, P, 100, 50, tag="snp_windows_clean_and_dirty_chunks")


def test_pair_range_sums_in_site_order(engine):
    """evofst_get_pair_range_sums: the sums of a pair's per-site values over caller-cut ranges, added in site order from a
    carried-in start value -- bit for bit what one loop over the oracle's per-site values gives (the per-site values are
    bit-exact, test_site_stats_bit_exact, and the order of the additions is the reference's, process_vcf_utils.h:369-385)."""
    S, N, P = 3000, 46, 3
    rng = np.random.default_rng(4242)
    sp, codes, fl = _setup(engine, rng, S, N, P, 0.15, flags=True)
    engine.set_pairs()
    pi, pj = engine.all_pairs()
    nf = n_full_of(sp, P)
    oalt, onm = O.counts_from_codes(codes, sp, P)
    onum, oden, odxy, oinc = O.pair_site_stats(oalt, onm, nf, pi, pj, fl)
    ohet, ohn, opi = O.pop_site_div(oalt, onm, nf)
    lo = np.concatenate(([0, 0, S, 17], rng.integers(0, S - 200, 40)))
    hi = np.concatenate(([S, 0, S, 18], lo[4:] + rng.integers(0, 200, 40)))
    init = rng.normal(0, 3, (len(lo), 12)); init[::3] = 0.0
    for q in (0, len(pi) - 1):
        for start in (None, init):
            got = engine.pair_range_sums(q, lo, hi, start)
            for r in range(len(lo)):
                a = np.zeros(12) if start is None else start[r].copy()
                with np.errstate(all="ignore"):
                    for s_ in range(lo[r], hi[r]):
                        if not oinc[s_, q]:
                            continue
                        vals = (onum[s_, q], oden[s_, q], odxy[s_, q], ohet[s_, pi[q]], ohet[s_, pj[q]], ohn[s_, pi[q]], ohn[s_, pj[q]],
                                opi[s_, pi[q]], opi[s_, pj[q]], 1.0, float(opi[s_, pi[q]] != 0), float(opi[s_, pj[q]] != 0))
                        for k, v in enumerate(vals):
                            a[k] = a[k] + v
                assert_same_bits(got[r], a)
    with pytest.raises(Exception):
        engine.pair_range_sums(0, [0], [S + 1])
    with pytest.raises(Exception):
        engine.pair_range_sums(len(pi), [0], [1])


@pytest.mark.parametrize("prod,cons,chunk_tiles,bufs,sets", [(8, 0, 16, 2, 1), (6, 0, 12, 3, 1), (6, 0, 12, 3, 2), (3, 1, 3, 4, 2),
                                                             (1, 0, 2, 2, 1), (8, 0, 8, 4, 1),
                                                             # chunk_tiles 0: the library's own shape -- 4 tiles for a SIMD's first
                                                             # producer wave, 3 for the waves behind it (28- and 19-tile chunks)
                                                             (8, 0, 0, 2, 1), (5, 0, 0, 2, 1)])
def test_stream_pass_shapes(engine, prod, cons, chunk_tiles, bufs, sets):
    """The stream pass (evofst_stream.hip) with every knob of evofst_set_option turned: producer and consumer waves per
    workgroup, tiles per chunk, LDS buffers (chunks of a workgroup in flight) and consumer sets; 78 pairs (two groups of
    lanes), clean chunks and chunks with a missing allele, windows that straddle both, more chunks than workgroups."""
    S, N, P, dirty = 150_000, 52, 13, 300
    rng = np.random.default_rng(prod * 100 + chunk_tiles + bufs)
    sp = round_robin_pops(N, P)
    codes = random_codes(rng, S, sp, P, miss=0.0, edge=False)
    for s_ in rng.choice(S, dirty, replace=False):
        codes[s_, rng.choice(2 * N, 2, replace=False)] = 2
    engine.set_layout(sp, P)
    # 3 % of the sites skipped by a filter (and the 48 sites behind the shard's end in its last tile): they stay on the
    # integer walk -- no count, both ballots set
    fl = (rng.random(S) < 0.03).astype(np.uint8)
    engine.upload(engine.pack_codes(codes), S, site_flags=fl)
    engine.set_pairs()
    with options(engine, pairs_pass=2, stream_producers=prod, stream_consumers=cons, chunk_tiles=chunk_tiles, stream_buffers=bufs,
                 stream_sets=sets, stream_blocks=16 if prod == 1 else 0):
        _check_windows_once(engine, sp, codes, fl, P, 100, 50, None, "snp_windows_stream_shapes")
        assert engine.last_pairs_pass() == 2


@pytest.mark.parametrize("chunk_tiles", [4, 16])
def test_one_byte_counts_of_clean_chunks(chunk_tiles):
    """K1 leaves one byte per count in chunks without a skipped, irregular or missing-allele site (pairs mode, 16-bit
    counts).  Same data through a context that walks them (run, segments) and one that was asked for per-site values
    first (which keeps K1 to packed counts): segments, totals and boundary bit-identical; the per-site getters after the
    walk (K1 runs again behind them) equal the other context's.  Chunks that turn dirty at their first, a middle and
    their last tile cover the return of already written one-byte tiles to packed counts."""
    from evo_amd import EvoFst
    rng = np.random.default_rng(1000 + chunk_tiles)
    S, N, P = 64 * chunk_tiles * 9 + 37, 60, 6
    sp = round_robin_pops(N, P)
    codes = random_codes(rng, S, sp, P, miss=0.0, edge=False)
    cs = 64 * chunk_tiles                                        # sites per chunk
    for chunk, tile in ((1, 0), (3, chunk_tiles // 2), (5, chunk_tiles - 1), (6, 1)):
        codes[chunk * cs + tile * 64 + 5, 3] = 2                 # one missing allele
    fl = np.zeros(S, np.uint8); fl[7 * cs + 64 * (chunk_tiles - 2) + 9] = 1      # and one skipped site
    out = []
    for ask_first in (False, True):
        e = EvoFst(0)
        try:
            e.set_option("chunk_tiles", chunk_tiles)
            e.set_option("pairs_pass", 1)                           # counts, scan and walk as separate kernels
            e.set_layout(sp, P)
            e.upload(e.pack_codes(codes), S, site_flags=fl)
            e.set_pairs()
            e.set_windows(50, 25)
            if ask_first:
                e.run_counts()
                c0 = e.counts()
            e.run()
            seg, tot, bnd = e.segments(), e.totals(), e.boundary()
            alt, nm = e.counts()
            st = e.site_stats()
            out.append((seg, tot, bnd, alt, nm, st))
            _check_windows_once(e, sp, codes, fl, P, 50, 25, None, "one_byte_counts")
        finally:
            e.close()
    a, b = out
    assert np.array_equal(a[0]["first_segment"], b[0]["first_segment"]) and np.array_equal(a[0]["n_segments"], b[0]["n_segments"])
    for q, n in enumerate(a[0]["n_segments"]):                   # rows behind a pair's last segment are not written
        assert n > 50
        for k in ("sum_num", "sum_den", "sum_dxy_k", "end_site"):
            assert np.array_equal(a[0][k][:n, q], b[0][k][:n, q], equal_nan=True), (k, q)
    assert np.array_equal(a[1], b[1], equal_nan=True) and np.array_equal(a[2], b[2], equal_nan=True)
    assert np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])
    oalt, onm = O.counts_from_codes(codes, sp, P)
    assert np.array_equal(a[3], oalt) and np.array_equal(a[4], onm)
    for x, y in zip(a[5], b[5]):
        assert np.array_equal(x, y, equal_nan=True)


def test_snp_windows_many_pairs(engine):
    """435 pairs: more than one K2 block per chunk (blockIdx.y > 0), each deriving the same tables."""
    rng = np.random.default_rng(435)
    sp, codes, fl = _setup(engine, rng, 2000, 90, 30, 0.1, flags=True)
    engine.set_pairs()
    assert engine.n_pairs == 435
    _check_windows(engine, sp, codes, fl, 30, 25, 25)


def test_snp_windows_few_pairs_of_many_populations(engine):
    """A short pair list over 40 populations: one 64-thread K2 block has to stage 80 mask words and 160 count
    vectors per half tile (the loops that run past the block's thread count)."""
    rng = np.random.default_rng(40)
    P = 40
    sp, codes, fl = _setup(engine, rng, 1800, 120, P, 0.15, flags=True)
    pi = np.array([0, 3, 39, 17, 5, 38, 20], np.int32)
    pj = np.array([39, 4, 1, 18, 31, 37, 21], np.int32)
    engine.set_pairs(pi, pj)
    _check_windows(engine, sp, codes, fl, P, 30, 15, pairs=(pi, pj))


def test_snp_windows_lds_near_the_limit(engine):
    """130 populations: K2's per-population tables take 140 KB of the 160 KB of LDS (dynamic size above 48 KB)."""
    rng = np.random.default_rng(130)
    P = 130
    sp, codes, fl = _setup(engine, rng, 700, 260, P, 0.1, flags=False)
    pi = np.array([0, 129, 64, 7], np.int32)
    pj = np.array([129, 1, 65, 100], np.int32)
    engine.set_pairs(pi, pj)
    _check_windows(engine, sp, codes, fl, P, 20, 20, pairs=(pi, pj))


def test_snp_windows_small_chunks(engine):
    """Chunks of one tile: every window straddles chunk boundaries (stitch path)."""
    rng = np.random.default_rng(99)
    sp, codes, fl = _setup(engine, rng, 3000, 40, 4, 0.3, flags=True)
    engine.set_pairs()
    with options(engine, chunk_tiles=1):
        _check_windows(engine, sp, codes, fl, 4, 150, 150)
        _check_windows(engine, sp, codes, fl, 4, 64, 32)


def test_snp_windows_longest_chunks(engine):
    """Chunks of 256 tiles (16 k sites, the library's maximum) with short overlapping windows: the error of a window's
    sums must not grow with the position of the window in its chunk."""
    rng = np.random.default_rng(256)
    sp, codes, fl = _setup(engine, rng, 40000, 40, 4, 0.05, flags=True)
    engine.set_pairs()
    with options(engine, chunk_tiles=256):
        _check_windows(engine, sp, codes, fl, 4, 20, 10, tag="snp_windows_256_tile_chunks")
        _check_windows(engine, sp, codes, fl, 4, 100, 100, tag="snp_windows_256_tile_chunks")


def test_base_offsets_shift_windows(engine):
    """A shard that starts after `base` included sites emits the windows of the global ranks."""
    rng = np.random.default_rng(3)
    S, N, P = 4000, 40, 4
    sp = round_robin_pops(N, P)
    codes = random_codes(rng, S, sp, P, miss=0.05)
    engine.set_layout(sp, P)
    engine.set_pairs()
    engine.set_windows(50, 50)
    # whole run
    engine.upload(engine.pack_codes(codes), S)
    engine.run()
    whole = engine.segments()
    # second half only, told how many included sites precede it
    cut = 2048
    engine.upload(engine.pack_codes(codes[:cut]), cut)
    engine.run()
    first_tot = engine.included_totals().copy()
    carry = engine.boundary()
    engine.upload(engine.pack_codes(codes[cut:]), S - cut)
    engine.run_counts()
    engine.set_base_offsets(first_tot)
    engine.run_stats()
    part = engine.segments()
    for q in range(engine.n_pairs):
        f = int(part["first_segment"][q]); n = int(part["n_segments"][q])
        assert f == first_tot[q] // 50
        a = part["sum_num"][:n, q].copy()
        a[0] = carry[q, 0] + a[0]
        b = whole["sum_num"][f:f + n, q]
        assert np.allclose(a, b, rtol=0, atol=1e-12 * 50)
        assert np.array_equal(part["end_site"][:n, q] + cut, whole["end_site"][f:f + n, q])


@pytest.mark.parametrize("S", [1, 2, 63, 64, 65, 129])
def test_tiny_and_ragged_inputs(engine, S):
    """Fewer sites than a tile, exactly a tile, one more: the ragged last tile never leaks into counts or windows."""
    rng = np.random.default_rng(1000 + S)
    sp, codes, fl = _setup(engine, rng, S, 24, 3, 0.1, flags=False)
    engine.set_pairs()
    engine.run_counts()
    alt, nm = engine.counts()
    oalt, onm = O.counts_from_codes(codes, sp, 3)
    assert alt.shape[0] == S and np.array_equal(alt, oalt) and np.array_equal(nm, onm)
    _check_windows(engine, sp, codes, fl, 3, 5, 5)
    _check_windows(engine, sp, codes, fl, 3, 1000, 1000)     # no window completes: zero segments, totals still exact


def test_empty_shard_is_refused(engine):
    """Zero sites: the library refuses the upload with an error (no launch over an empty grid); callers skip empty
    batches, as the CLI and evo_amd.shard do."""
    from evo_amd.api import EvoFstError
    sp = round_robin_pops(24, 3)
    engine.set_layout(sp, 3)
    engine.set_pairs()
    engine.set_windows(10, 10)
    with pytest.raises(EvoFstError, match="n_sites must be positive"):
        engine.upload(np.zeros(0, np.uint8), 0)


def test_limits_are_refused_not_launched(engine):
    """shapes the kernels' LDS tables cannot hold come back as errors before any launch."""
    from evo_amd.api import EvoFstError
    rng = np.random.default_rng(8)
    # 200 populations in pairs mode: K2's tables would need 211 KB of LDS
    P = 200
    sp = round_robin_pops(400, P)
    codes = random_codes(rng, 300, sp, P, miss=0.05)
    engine.set_layout(sp, P)
    engine.upload(engine.pack_codes(codes), 300)
    engine.set_pairs(np.array([0, 5], np.int32), np.array([199, 6], np.int32))
    engine.set_windows(10, 10)
    with pytest.raises(EvoFstError):
        engine.run()
    # the same data goes through the Gram path
    engine.set_gram()
    engine.set_site_windows(np.array([0, 100, 300], np.int64))
    engine.run_counts()
    engine.run_gram()
    assert engine.gram()["n_included"].shape == (2, P * (P - 1) // 2)
    # against all with a deque of 400 sites x 200 populations x 16 B = 1.28 MB
    engine.set_against_all(np.arange(P, dtype=np.int32), np.full(P, 2, np.int64))
    engine.set_windows(400, 400)
    with pytest.raises(EvoFstError):
        engine.run()
