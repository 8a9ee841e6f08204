"""Shared generators for the parity tests (seeded, numpy only)."""
import numpy as np


def random_codes(rng, n_sites, sample_pop, n_pops, miss=0.05, edge=True):
    """Allele codes [S, 2N] (0 ref, 1 alt, 2 missing) with per-pop frequencies and the edge cases the
    reference's arithmetic is sensitive to: fixed sites, all-missing pops, pops with one allele left."""
    sample_pop = np.asarray(sample_pop)
    N = len(sample_pop)
    f = rng.beta(0.5, 0.5, size=(n_sites, 1))
    pk = np.clip(f + rng.normal(0, 0.15, size=(n_sites, n_pops)), 0, 1)
    hp = np.where(sample_pop >= 0, sample_pop, 0)
    p_h = np.repeat(pk[:, hp], 2, axis=1)                         # [S, 2N]
    codes = (rng.random((n_sites, 2 * N)) < p_h).astype(np.uint8)
    # missing: mostly whole genotypes, sometimes half calls
    mg = rng.random((n_sites, N)) < miss
    codes[np.repeat(mg, 2, axis=1)] = 2
    half = rng.random((n_sites, 2 * N)) < miss * 0.2
    codes[half] = 2
    if edge and n_sites >= 16:
        hap_pop = np.repeat(sample_pop, 2)
        codes[0, :] = 0                                            # monomorphic ref
        codes[1, :] = 1                                            # fixed alt everywhere
        codes[2, hap_pop == 0] = 2                                 # pop 0 all missing
        idx = np.where(hap_pop == (1 % n_pops))[0]
        codes[3, idx] = 2; codes[3, idx[0]] = 1                    # pop 1 has exactly one allele left (nm == 1)
        codes[4, :] = 2                                            # everything missing
        codes[5, hap_pop == 0] = 1; codes[5, hap_pop != 0] = 0     # fixed difference
        codes[6, idx] = 2; codes[6, idx[:2]] = [0, 1]              # nm == 2
    return codes


def round_robin_pops(n_samples, n_pops):
    return (np.arange(n_samples) % n_pops).astype(np.int32)


def n_full_of(sample_pop, n_pops):
    sp = np.asarray(sample_pop)
    return np.array([2 * int((sp == k).sum()) for k in range(n_pops)], np.int32)


def assert_same_bits(a, b, what=""):
    """Doubles equal bit for bit (NaN == NaN, +0 == +0 but not -0)."""
    a = np.ascontiguousarray(a, np.float64); b = np.ascontiguousarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    ia = a.view(np.uint64); ib = b.view(np.uint64)
    both_nan = np.isnan(a) & np.isnan(b)
    bad = (ia != ib) & ~both_nan
    if bad.any():
        k = np.argwhere(bad)[0]
        raise AssertionError("%s: %d of %d doubles differ, first at %s: %r vs %r" %
                             (what, int(bad.sum()), a.size, tuple(k), a[tuple(k)], b[tuple(k)]))
