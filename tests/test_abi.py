"""CPU tests of the C ABI: the library loads without a GPU, exports every symbol include/evofst.h
declares, the host packer (stage K0) is exact, and compute calls fail loudly without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import evo_amd
from evo_amd import api
from oracle import oracle as O
from tests.helpers import random_codes, round_robin_pops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "evofst.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(evofst_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    L = C.CDLL(evo_amd.LIB_PATH)
    names = _declared()
    assert len(names) >= 38
    for n in names:
        assert hasattr(L, n), "libevofst.so does not export %s" % n
    assert sorted(api.SIGNATURES) == names, "evo_amd.api.SIGNATURES is out of sync with include/evofst.h"
    assert b"gfx950" in api.load_library().evofst_version()


def test_no_cpu_fallback():
    """Without a GPU a device context cannot be created, and a host-only context refuses compute."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(api.EvoFstError):
        api.EvoFst(0)
    e = api.EvoFst(-1)
    e.set_layout(round_robin_pops(8, 2), 2)
    for call in (e.run_counts, e.run_stats, e.run, e.sync, lambda: e.reserve(10), lambda: e.set_pairs(), e.indiv_hets,
                 lambda: e.indiv_diffs([0, 1])):
        with pytest.raises(api.EvoFstError) as ex:
            call()
        assert ex.value.code == api.ERR_STATE


def _numpy_pack(codes, sample_pop, n_pops):
    """Independent restatement of the tile format documented in include/evofst.h."""
    S, H2 = codes.shape
    order = [i for k in range(n_pops) for i in range(len(sample_pop)) if sample_pop[i] == k]
    haps = np.array([[2 * i, 2 * i + 1] for i in order]).reshape(-1)
    H = len(haps); W = (H + 63) // 64
    T = (S + 63) // 64
    out = np.zeros((T, W, 64, 2), np.uint64)
    sub = codes[:, haps]
    for w in range(W):
        blk = sub[:, w * 64:(w + 1) * 64]
        wts = (np.uint64(1) << np.arange(blk.shape[1], dtype=np.uint64))
        alt = ((blk == 1).astype(np.uint64) * wts).sum(axis=1, dtype=np.uint64)
        mis = ((blk == 2).astype(np.uint64) * wts).sum(axis=1, dtype=np.uint64)
        for s in range(S):
            out[s // 64, w, s % 64, 0] = alt[s]; out[s // 64, w, s % 64, 1] = mis[s]
    return out.reshape(-1)


@pytest.mark.parametrize("S,N,P", [(1, 3, 1), (64, 32, 2), (65, 33, 3), (300, 100, 7), (129, 70, 70)])
def test_packer_round_trip_and_format(S, N, P):
    rng = np.random.default_rng(S * 31 + N)
    sp = round_robin_pops(N, P)
    if N > P + 2:
        sp[N - 1] = -1                      # a sample that belongs to no population
    codes = random_codes(rng, S, np.where(sp < 0, 0, sp), P, miss=0.2, edge=False)
    e = api.EvoFst(-1)
    e.set_layout(sp, P)
    g = e.geometry()
    assert g.tile_sites == 64 and g.words_per_site == (g.n_haplotypes + 63) // 64
    assert e.tiles_bytes(S) == ((S + 63) // 64) * g.words_per_site * 64 * 16
    tiles = e.pack_codes(codes)
    assert np.array_equal(tiles, _numpy_pack(codes, sp, P))
    back = e.unpack_codes(tiles, S)
    keep = np.repeat(sp >= 0, 2)
    assert np.array_equal(back[:, keep], codes[:, keep])
    assert (back[:, ~keep] == 0).all()
    nf = e.pop_chroms()
    assert list(nf) == [2 * int((sp == k).sum()) for k in range(P)]
    e.close()


def test_layout_errors():
    e = api.EvoFst(-1)
    with pytest.raises(api.EvoFstError):
        e.set_layout(np.array([0, 0, 2], np.int32), 2)          # pop index out of range
    with pytest.raises(api.EvoFstError):
        e.set_layout(np.array([0, 0, 0], np.int32), 2)          # empty population
    assert e.tiles_bytes(10) == 0                                # no layout yet
    e.close()


def test_vcf_line_codes_follow_both_rule_sets():
    line = "chr1\t100\t.\tA\tG,*\t9\t.\tDP=1\tGT:DP\t0/1:3\t1|2:4\t./.:0\t2/0:1"
    c, fl, chrom, pos = api.codes_from_vcf_line(line, api.MODE_FST, 4)
    assert (chrom, pos, fl) == ("chr1", "100", 0)
    assert list(c) == [0, 1, 1, 2, 2, 2, 2, 0]                   # allele 2 = '*' = missing
    c, fl, _, _ = api.codes_from_vcf_line(line, api.MODE_GLOBAL, 4)
    assert fl == api.SITE_SKIP and (c == 2).all()                 # len(ALT) > 1: skipped, genotypes ignored
    line = "chr1\t100\t.\tA\t*,G\t9\t.\tDP=1\tGT\t0/1\t1/2\t2/2\t0/."
    c, fl, _, _ = api.codes_from_vcf_line(line, api.MODE_FST, 4)
    assert fl == 0 and list(c) == [0, 2, 2, 1, 1, 1, 0, 2]       # allele 1 missing, allele 2 is ALT
    for alt, skip in (("GT", 1), ("G,T", 1), ("*", 1), ("G", 0)):
        line = "c\t1\t.\tA\t%s\t9\t.\t.\tGT\t0/1\t1/1" % alt
        assert api.codes_from_vcf_line(line, api.MODE_FST, 2)[1] == skip
        assert api.codes_from_vcf_line(line, api.MODE_GLOBAL, 2)[1] == skip
    line = "c\t1\t.\tAT\tG\t9\t.\t.\tGT\t0/1\t1/1"
    assert api.codes_from_vcf_line(line, api.MODE_FST, 2)[1] == 1
    # single-char ALT with a stray allele '2': ref for `fst`, missing for `FstGlobal`
    line = "c\t1\t.\tA\tG\t9\t.\t.\tGT\t2/1\t0/2"
    assert list(api.codes_from_vcf_line(line, api.MODE_FST, 2)[0]) == [0, 1, 0, 0]
    assert list(api.codes_from_vcf_line(line, api.MODE_GLOBAL, 2)[0]) == [2, 1, 0, 2]
    with pytest.raises(api.EvoFstError):
        api.codes_from_vcf_line("c\t1\t.\tA\tG\t9\t.\t.\tGT\t0/1", api.MODE_FST, 2)   # too few columns
    with pytest.raises(api.EvoFstError):
        api.codes_from_vcf_line("c\t1\t.\tA\tG\t9\t.\t.\tGT\t0\t1/1", api.MODE_FST, 2)  # haploid field


def test_vcf_line_codes_agree_with_oracle_allele_rules():
    L = O.lib()
    rng = np.random.default_rng(2)
    for _ in range(200):
        alt = rng.choice(["G", "G,*", "*,G", "*", "G,T", "GA"])
        gts = ["".join([rng.choice(list("012.")), rng.choice(list("/|")), rng.choice(list("012."))]) for _ in range(6)]
        line = "c\t5\t.\tA\t%s\t9\t.\t.\tGT\t%s" % (alt, "\t".join(gts))
        c, fl, _, _ = api.codes_from_vcf_line(line, api.MODE_FST, 6)
        ac = C.c_int(); am = C.c_int()
        ofl = L.orc_fst_site_flags(b"A", alt.encode(), C.byref(ac), C.byref(am))
        assert fl == ofl
        exp = [L.orc_fst_allele_code(g[i].encode(), ac.value, am.value) for g in gts for i in (0, 2)]
        assert list(c) == (exp if not fl else [2] * 12)      # skipped lines: genotypes are never looked at
        c, fl, _, _ = api.codes_from_vcf_line(line, api.MODE_GLOBAL, 6)
        assert fl == L.orc_general_site_flags(b"A", alt.encode())
        exp = [L.orc_general_allele_code(g[i].encode()) for g in gts for i in (0, 2)]
        assert list(c) == (exp if not fl else [2] * 12)
        c, fl, _, _ = api.codes_from_vcf_line(line, api.MODE_GLOBAL_INDELS, 6)   # -i of PBS / DistOutgroups: only "*" is skipped
        assert fl == (api.SITE_SKIP if alt == "*" else 0)
        assert list(c) == (exp if not fl else [2] * 12)
