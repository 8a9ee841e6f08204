"""GPU parity of the Gram path (FP64 MFMA, all pairs at once over pair-independent windows) against the CPU
oracle: per window and pair, the sums over the window's included sites of the reference's per-site numerator,
denominator and dxy numerator (process_vcf_fst.h:19-30, process_vcf_fst.cpp:299-321, filter :543-544).

Bars: included-site counts and dxy sums (integers) exact; num / den within 1e-12 of the window's sum of
|terms| (for num: of the numerator's and the denominator's terms, since num is formed as den - t - h); NaN
exactly where the reference's sum is NaN.
"""
import numpy as np
import pytest

from tests.helpers import options, random_codes, round_robin_pops, n_full_of
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _oracle_windows(onum, oden, odxy_k, oinc, ws):
    nw, nq = len(ws) - 1, onum.shape[1]
    out = {k: np.zeros((nw, nq)) for k in ("num", "den", "k", "cnt", "absnum", "absden")}
    for w in range(nw):
        a, b = ws[w], ws[w + 1]
        m = oinc[a:b] == 1
        with np.errstate(all="ignore"):
            out["num"][w] = np.where(m, onum[a:b], 0.0).sum(axis=0)
            out["den"][w] = np.where(m, oden[a:b], 0.0).sum(axis=0)
            out["absnum"][w] = np.where(m & np.isfinite(onum[a:b]), np.abs(onum[a:b]), 0.0).sum(axis=0)
            out["absden"][w] = np.where(m & np.isfinite(oden[a:b]), np.abs(oden[a:b]), 0.0).sum(axis=0)
        out["k"][w] = np.where(m, odxy_k[a:b], 0.0).sum(axis=0)
        out["cnt"][w] = m.sum(axis=0)
    return out


def _check_gram(engine, codes, sp, P, fl, ws, tag):
    """run_gram over the windows ws of the loaded data against the oracle; returns the worst error as a fraction of the bar."""
    from tests import parity_log
    g = engine.gram()
    nf = n_full_of(sp, P)
    pi, pj = np.triu_indices(P, 1)
    pi = pi.astype(np.int32); pj = pj.astype(np.int32)
    oalt, onm = O.counts_from_codes(codes, sp, P)
    onum, oden, odxy, oinc = O.pair_site_stats(oalt, onm, nf, pi, pj, fl)
    # the oracle returns dxy = k / (n_i n_j); k itself is an integer
    odxy_k = np.rint(odxy * (nf[pi] * nf[pj])[None, :].astype(np.float64))
    ow = _oracle_windows(onum, oden, odxy_k, oinc, ws)

    assert np.array_equal(g["n_included"], ow["cnt"])
    assert np.array_equal(g["sum_dxy_k"], ow["k"])
    worst = 0.0
    for name, ref, scale in (("sum_den", ow["den"], ow["absden"]), ("sum_num", ow["num"], ow["absnum"] + ow["absden"])):
        a = g[name]
        assert (np.isnan(a) == np.isnan(ref)).all(), name
        fin = ~np.isnan(ref)
        err, bar = np.abs(a[fin] - ref[fin]), 1e-12 * np.maximum(scale[fin], 1e-300) + 1e-300
        parity_log.record("%s/%s" % (tag, name), err, bar)
        assert np.all(err <= bar), (name, float(np.max(err / bar)))
        worst = max(worst, float(np.max(err / bar))) if err.size else worst
    return worst


@pytest.mark.parametrize("S,N,P,miss,flags,edge,seed", [(1500, 48, 12, 0.0, False, False, 1), (2000, 100, 20, 0.15, True, True, 2),
                                                        (900, 150, 75, 0.1, False, True, 3), (700, 66, 33, 0.3, True, True, 4),
                                                        (1300, 520, 130, 0.0, False, False, 5), (640, 40, 2, 0.2, False, True, 6),
                                                        (800, 1200, 3, 0.1, True, True, 7), (500, 2200, 2, 0.05, False, True, 8),
                                                        (600, 126, 2, 0.2, True, True, 9),    # 126 chromosomes: the top of the int8 range
                                                        (1000, 300, 12, 0.15, True, True, 10),   # 50 chromosomes: the 64 x 64 table of the one-stream form
                                                        (700, 124, 4, 0.1, False, True, 11)])    # 62: its top
def test_gram_windows(engine, S, N, P, miss, flags, edge, seed):
    rng = np.random.default_rng(seed)
    sp = round_robin_pops(N, P)
    codes = random_codes(rng, S, sp, P, miss=miss, edge=edge)
    engine.set_layout(sp, P)
    fl = (rng.random(S) < 0.05).astype(np.uint8) if flags else None
    engine.upload(engine.pack_codes(codes), S, site_flags=fl)
    engine.set_gram()
    # ragged windows incl. empty ones, not aligned to anything; the last one ends at S
    cuts = np.sort(rng.integers(0, S, size=S // 90))
    ws = np.concatenate([[0], cuts, [cuts[-1]], [S]]).astype(np.int64)
    engine.set_site_windows(ws)
    engine.run_counts()
    engine.run_gram()
    # the denominator on one FP64 contraction (centred frequencies) where populations have <= 63 chromosomes -- most of these
    # shapes; windows longer than 2^13 / chromosomes run as pieces, (640, 40, 2) has such -- else, and when asked, on the two
    # of p q^T + q p^T
    max_n = int(n_full_of(sp, P).max())
    one = max_n <= 63
    assert engine.last_gram_streams() == (1 if one else 2)
    _check_gram(engine, codes, sp, P, fl, ws, "gram_windows_one_stream" if one else "gram_windows")
    if one:
        with options(engine, gram_denominator=2):
            engine.run_gram()
            assert engine.last_gram_streams() == 2
            _check_gram(engine, codes, sp, P, fl, ws, "gram_windows")


@pytest.mark.parametrize("per,win", [(5, 800), (25, 160), (5, 3200), (25, 1200)])
def test_gram_one_stream_hard_windows(engine, per, win):
    """The one-contraction denominator where it has least room: windows as long as its rule allows (800 sites at 10
    chromosomes a population, 160 at 50 -- the 64 x 64 table) and longer ones, which run as pieces of that length (3200 =
    four pieces, 1200 = eight), nearly every site fixed for the same allele in all populations (their terms cancel exactly:
    u_i u_j = 1/4), and a single site or two per window where one population carries one other allele -- a denominator of
    0.1 or 0.18 formed as 400 - 2 x 199.95 (in every other window near its start, where the term lives through the most
    binades of the running sum).  Populations without a called allele at a site (n_both) and fixed sites with
    missing alleles are in there too.  Every window inside 1e-12 of its own terms."""
    rng = np.random.default_rng(2026)
    S, P = 9600, 6
    N = P * per
    sp = round_robin_pops(N, P)
    codes = np.ones((S, 2 * N), np.uint8)                      # fixed for ALT everywhere ...
    codes[rng.random(S) < 0.3] = 0                             # ... or for REF
    ws = np.arange(0, S + 1, win, dtype=np.int64)
    chrom_of_pop = [np.where(np.repeat(sp, 2) == k)[0] for k in range(P)]
    for w in range(len(ws) - 1):
        for s_ in rng.choice(np.arange(ws[w], ws[w] + win // 8 if w % 2 else ws[w + 1]), 2, replace=False):
            k = int(rng.integers(P))
            codes[s_, chrom_of_pop[k][0]] ^= 1                 # one allele of one population differs
        for s_ in rng.choice(np.arange(ws[w], ws[w + 1]), win // 40, replace=False):
            k = int(rng.integers(P))
            n_miss = int(rng.choice([1, 3, 2 * per]))          # some alleles, or all, of a population missing at a fixed site
            codes[s_, chrom_of_pop[k][:n_miss]] = 2
    engine.set_layout(sp, P)
    engine.upload(engine.pack_codes(codes), S)
    engine.set_gram()
    engine.set_site_windows(ws)
    engine.run_counts()
    engine.run_gram()
    assert engine.last_gram_streams() == 1
    worst = _check_gram(engine, codes, sp, P, None, ws, "gram_one_stream_hard_windows")
    assert worst <= 0.6, worst


@pytest.mark.parametrize("S,win,first,n", [(3000, 100, 7, 11), (9000, 1500, 2, 3)])   # (1500-site windows run as two pieces each)
def test_gram_window_batches(engine, S, win, first, n):
    """run_gram over sub-ranges of the windows gives the same rows as one call."""
    rng = np.random.default_rng(11)
    N, P = 80, 16
    sp = round_robin_pops(N, P)
    codes = random_codes(rng, S, sp, P, miss=0.05)
    engine.set_layout(sp, P)
    engine.upload(engine.pack_codes(codes), S)
    engine.set_gram()
    ws = np.arange(0, S + 1, win, dtype=np.int64)
    engine.set_site_windows(ws)
    engine.run_counts()
    engine.run_gram()
    full = engine.gram()
    engine.run_gram(first, n)
    part = engine.gram()
    for k in full:
        assert np.array_equal(full[k][first:first + n], part[k], equal_nan=True)
