"""GPU parity: per-individual pairwise differences (process_vcf_stats_functions.cpp:180-281) through the C ABI -- VCF
lines packed with EVOFST_MODE_STATS, multi-allelic sites as allele numbers -- vs the CPU oracle's literal restatement
(which tests/test_oracle.py pins bit for bit on the reference's own functions).  All five counts are integers: exact."""
import numpy as np
import pytest

from evo_amd import api
from tests.helpers import write_vcf
from oracle import oracle as O

pytestmark = pytest.mark.gpu
MODE_STATS, SKIP, MULTI = 2, 1, 2


def _oracle_inputs(lines, n):
    S = len(lines)
    iv = np.zeros((S, n), np.intc); hp = np.zeros((S, 2 * n), np.intc); ms = np.zeros((S, n), np.uint8); mu = np.zeros(S, np.uint8)
    for s, ln in enumerate(lines):
        f = ln.split("\t")
        iv[s], hp[s], ms[s], mu[s] = O.stats_variant_counts(f[4], f[9:])
    return iv, hp, ms, mu


def _load(engine, lines, n, threads=4):
    engine.set_layout(np.zeros(n, np.int32), 1)
    tiles, flags = engine.pack_vcf_lines([l.encode() for l in lines], MODE_STATS, n, n_threads=threads)
    engine.upload(tiles, len(lines), site_flags=flags)
    multi = np.flatnonzero(flags & MULTI)
    assert np.all(flags[multi] & SKIP)
    if len(multi):
        al = np.stack([api.alleles_from_vcf_line(lines[k], n) for k in multi])
        engine.set_multiallelic(multi, al)
    return flags


@pytest.mark.parametrize("S,n,miss,seed", [(5000, 37, 0.1, 1), (3000, 150, 0.02, 2), (700, 16, 0.4, 3)])
def test_indiv_diffs_blocks(engine, tmp_path, S, n, miss, seed):
    rng = np.random.default_rng(seed)
    vcf = str(tmp_path / "in.vcf")
    write_vcf(vcf, rng, S, n, miss=miss)
    lines = [l.rstrip("\n") for l in open(vcf) if not l.startswith("#")]
    flags = _load(engine, lines, n)
    iv, hp, ms, mu = _oracle_inputs(lines, n)
    assert np.array_equal((flags & MULTI) != 0, mu != 0) and mu.sum() > 10
    cuts = np.concatenate([[0], np.sort(rng.integers(0, S + 1, 9)), [S]]).astype(np.int64)
    cuts[3] = cuts[2]                                                   # an empty block
    got = engine.indiv_diffs(cuts)
    for b in range(len(cuts) - 1):
        sl = slice(cuts[b], cuts[b + 1])
        want = O.indiv_diff_counts(iv[sl], hp[sl], ms[sl], mu[sl])
        for k in want:
            assert np.array_equal(got[k][b], want[k]), (k, b)
    whole = engine.indiv_diffs([0, S])
    tot = O.indiv_diff_counts(iv, hp, ms, mu)
    for k in tot:
        assert np.array_equal(whole[k][0], tot[k]) and np.array_equal(got[k].sum(0), tot[k]), k
    assert tot["hh"].sum() > 0 and tot["hd"].sum() > 0 and tot["oh"].sum() > 0 and tot["ms"].sum() > 0 and tot["d4"].sum() > 0
    # and the matrices the reference prints follow from the counts
    lit = O.indiv_diffs(iv, hp, ms, mu)
    comp = O.matrices_from_diff_counts({k: v[0] for k, v in whole.items()}, n)
    assert np.array_equal(lit["diffs_me"], comp["diffs_me"]) and np.array_equal(lit["miss"], comp["miss"])
    assert np.array_equal(lit["hets_vs_homs"], comp["hets_vs_homs"])
    assert np.allclose(lit["diffs"], comp["diffs"], rtol=1e-12, atol=0)


def test_indiv_diffs_no_multiallelic_and_new_upload_forgets_them(engine, tmp_path):
    rng = np.random.default_rng(9)
    n = 21
    vcf = str(tmp_path / "a.vcf")
    write_vcf(vcf, rng, 900, n, miss=0.1)
    lines = [l.rstrip("\n") for l in open(vcf) if not l.startswith("#")]
    _load(engine, lines, n)
    plain = [l for l in lines if "," not in l.split("\t")[4]]            # no multi-allelic line left
    flags = _load(engine, plain, n)
    assert not (flags & MULTI).any()
    iv, hp, ms, mu = _oracle_inputs(plain, n)
    got = engine.indiv_diffs([0, len(plain)])
    want = O.indiv_diff_counts(iv, hp, ms, mu)
    for k in want:
        assert np.array_equal(got[k][0], want[k]), k
    assert got["d4"].sum() == 0


@pytest.mark.parametrize("S,n,miss,seed", [(6000, 37, 0.004, 4), (2500, 150, 0.001, 5), (700, 16, 0.4, 6), (130, 70, 0.0, 7)])
def test_indiv_hets(engine, tmp_path, S, n, miss, seed):
    """evofst_get_indiv_hets vs het_analysis as statsMain drives it (process_vcf_stats_functions.cpp:19-28,
    process_vcf_stats.cpp:239-242): integers, exact; more than 64 individuals take several per lane."""
    rng = np.random.default_rng(seed)
    vcf = str(tmp_path / "in.vcf")
    write_vcf(vcf, rng, S, n, miss=miss)
    lines = [l.rstrip("\n") for l in open(vcf) if not l.startswith("#")]
    _load(engine, lines, n)
    iv, hp, ms, mu = _oracle_inputs(lines, n)
    het, shared = engine.indiv_hets()
    want_het, want_shared = O.indiv_hets(iv, ms)
    assert np.array_equal(het, want_het) and np.array_equal(shared, want_shared)
    if miss < 0.1:
        assert want_het.min() > 0 and want_shared.sum() > 0
        assert n > 64 or (want_shared < want_het).any()                # among many individuals a heterozygote is hardly ever alone
    het2, shared2 = engine.indiv_hets()                               # a second call starts from zero again
    assert np.array_equal(het2, het) and np.array_equal(shared2, shared)
