"""CPU tests of the oracle: pinned against the reference's own inline math (oracle/_ref, built from
/root/reference headers) and against the known answers the survey recorded from the reference."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests.helpers import assert_same_bits, write_vcf

GOLD = os.path.join(os.path.dirname(__file__), "golden")
KA = json.load(open(os.path.join(GOLD, "survey_known_answers.json")))


def test_known_answer_direct_call():
    d = KA["direct_call"]
    sc, i1, i2, h1, h2 = O.fst_variant_counts(d["alt"], d["genotypes"], d["set1"], d["set2"])
    assert (sc.set1Count, sc.set1_n_withoutMissing, sc.set2Count, sc.set2_n_withoutMissing) == \
        (d["c1"], d["nm1"], d["c2"], d["nm2"])
    L = O.lib()
    p1 = sc.set1Count / sc.set1_n_withoutMissing; p2 = sc.set2Count / sc.set2_n_withoutMissing
    assert p1 == d["p1"] and p2 == d["p2"]
    assert L.orc_fst_numerator(p1, p2, d["nm1"], d["nm2"]) == d["num"]
    assert L.orc_fst_denominator(p1, p2) == d["den"]
    assert L.orc_dxy_literal(i1.ctypes.data_as(O._ip), i2.ctypes.data_as(O._ip), 8, 8) == d["dxy"]
    assert L.orc_dxy_closed(d["c1"], d["c2"], 8, 8) == d["dxy"]
    assert L.orc_pi_literal(h1.ctypes.data_as(O._ip), 8, d["nm1"]) == d["pi1"]
    assert L.orc_pi_literal(h2.ctypes.data_as(O._ip), 8, d["nm2"]) == d["pi2"]
    assert L.orc_pi_closed(d["c1"], 8, d["nm1"]) == d["pi1"] and L.orc_pi_closed(d["c2"], 8, d["nm2"]) == d["pi2"]
    g = d["general"]
    alt, alleles, aaf, allc, aafc = O.general_counts(d["genotypes"], [0] * 4 + [1] * 4, 2, [1, 1])
    assert (alt[0], alleles[0], aaf[0], aafc[0], allc[0]) == (g["alt"], g["n"], g["aaf"], g["aaf_compl"], g["n_compl"])


def test_known_answer_sc3749(tmp_path):
    vcf = os.path.join(GOLD, "sc3749.vcf")
    names = open(vcf).readline().rstrip("\n").split("\t")[9:]
    sets = tmp_path / "sets.txt"
    sets.write_text(",".join(names[:25]) + "\n" + ",".join(names[25:]) + "\n")
    summ = O.fst_run(vcf, str(sets), str(tmp_path), run_name="t", size=1, step=1)
    rows = (tmp_path / "t_w_1_fst.txt").read_text().splitlines()
    assert rows[1] == KA["sc3749"]["fst_row"]
    assert "%g" % summ[0] == KA["sc3749"]["genome_wide_fst_6sig"]
    # -w 1,1 on a scaffold that is not scaffold_0: fixed-window file is header-only
    assert len((tmp_path / "t_w_10000_fixedWindowStats.txt").read_text().splitlines()) == 1


def test_literal_loops_equal_closed_forms():
    rng = np.random.default_rng(0)
    L = O.lib()
    for _ in range(300):
        m1, m2 = int(rng.integers(1, 9)), int(rng.integers(1, 9))
        g = []
        for i in range(m1 + m2):
            a = rng.choice(["0", "1", "."], 2, p=[0.45, 0.4, 0.15])
            g.append(a[0] + "/" + a[1])
        sc, i1, i2, h1, h2 = O.fst_variant_counts("T", g, list(range(m1)), list(range(m1, m1 + m2)))
        n1, n2 = 2 * m1, 2 * m2
        lit = L.orc_dxy_literal(i1.ctypes.data_as(O._ip), i2.ctypes.data_as(O._ip), n1, n2)
        assert lit == L.orc_dxy_closed(sc.set1Count, sc.set2Count, n1, n2)
        with np.errstate(all="ignore"):
            a = L.orc_pi_literal(h1.ctypes.data_as(O._ip), n1, sc.set1_n_withoutMissing)
            b = L.orc_pi_closed(sc.set1Count, n1, sc.set1_n_withoutMissing)
        assert a == b or (np.isnan(a) and np.isnan(b))


def test_star_alt_rules():
    g = ["0/1", "1/2", "2/2", "0/."]
    sc, *_ = O.fst_variant_counts("G,*", g, [0, 1], [2, 3])        # allele 2 is missing
    assert (sc.n_alt_alleles, sc.alt_code, sc.set1Count, sc.set1_n_withoutMissing,
            sc.set2Count, sc.set2_n_withoutMissing) == (1, 1, 2, 3, 0, 1)
    sc, *_ = O.fst_variant_counts("*,G", g, [0, 1], [2, 3])        # allele 1 missing, allele 2 is ALT
    assert (sc.n_alt_alleles, sc.alt_code, sc.set1Count, sc.set1_n_withoutMissing,
            sc.set2Count, sc.set2_n_withoutMissing) == (1, 2, 1, 2, 2, 3)
    sc, *_ = O.fst_variant_counts("G,T", g, [0, 1], [2, 3])
    assert sc.n_alt_alleles == 2 and not O.lib().orc_fst_site_included(C.byref(sc))
    sc, *_ = O.fst_variant_counts("GT", g, [0, 1], [2, 3])
    assert sc.bIndel == 1 and not O.lib().orc_fst_site_included(C.byref(sc))


@pytest.mark.skipif(O.ref() is None, reason="oracle/_ref not built (no /root/reference here)")
def test_math_bit_equal_to_reference_headers():
    """The oracle's per-site math vs the reference's own inline functions (process_vcf_fst.h:19-45,
    process_vcf_utils.h:272-275, 350-385) compiled from /root/reference."""
    R, L = O.ref(), O.lib()
    rng = np.random.default_rng(1)
    for _ in range(20000):
        n1, n2 = int(rng.integers(0, 60)), int(rng.integers(0, 60))
        c1, c2 = int(rng.integers(0, n1 + 1)), int(rng.integers(0, n2 + 1))
        p1 = C.c_double(); p2 = C.c_double()
        with np.errstate(all="ignore"):
            R.ref_calculateAlleleFrequencies(c1, n1, c2, n2, C.byref(p1), C.byref(p2))
            e1 = np.float64(c1) / np.float64(n1); e2 = np.float64(c2) / np.float64(n2)
        assert_same_bits([p1.value, p2.value], [e1, e2], "allele frequencies")
        a = [R.ref_calculateFstNumerator(p1.value, p2.value, n1, n2), R.ref_calculateFstDenominator(p1.value, p2.value),
             R.ref_calculateExpectedHeterozygositySimple(p1.value), R.ref_calculateExpectedHeterozygosityNei78(p1.value, n1)]
        b = [L.orc_fst_numerator(p1.value, p2.value, n1, n2), L.orc_fst_denominator(p1.value, p2.value),
             L.orc_het_simple(p1.value), L.orc_het_nei78(p1.value, n1)]
        assert_same_bits(a, b, "per-site math (c=%d/%d, %d/%d)" % (c1, n1, c2, n2))
    for n in (1, 2, 7, 100, 1000):
        v = rng.normal(size=n)
        vp = v.ctypes.data_as(O._dp)
        assert R.ref_vector_average(vp, n) == L.orc_vector_average(vp, n)
        assert R.ref_vector_sum(vp, n) == L.orc_vector_sum(vp, n)
        assert R.ref_vector_average_withRegion(vp, n, 37) == L.orc_vector_average_with_region(vp, n, 37)


@pytest.mark.skipif(O.ref() is None, reason="oracle/_ref not built (no /root/reference here)")
def test_general_ctor_order_and_init():
    """setSizes follows std::map (sorted name) order over ALL pops and AAFs start at -1.0
    (process_vcf_utils.h:71-82, 115-119)."""
    R = O.ref()
    names = [b"zeta", b"Outgroup", b"alpha", b"xxx"]
    sizes = (C.c_int * 4)(3, 5, 7, 2)
    arr = (C.c_char_p * 4)(*names)
    out = (C.c_long * 4)()
    a = C.c_double(); b = C.c_double()
    n = R.ref_general_ctor(arr, sizes, 4, out, C.byref(a), C.byref(b))
    assert n == 4 and list(out) == [5, 7, 2, 3]          # Outgroup, alpha, xxx, zeta
    assert a.value == -1.0 and b.value == -1.0


@pytest.mark.skipif(O.ref_pbs() is None, reason="oracle/_ref not built (no /root/reference here)")
def test_pbs_bit_equal_to_reference():
    """orc_pbs_from_afs vs the reference's own calculatePBSfromAFs (evo_PBS.cpp:69-98) compiled from
    /root/reference, on every triple of (alt, called) counts up to 7 alleles -- this includes single-allele
    populations (0/0 in the variance term), fixed differences (Fst == 1) and the NaN sign that reaches the text."""
    R = O.ref_pbs()
    L = O.lib()
    L.orc_pbs_from_afs.restype = None
    L.orc_pbs_from_afs.argtypes = [C.c_double] * 6 + [O._dp]
    pn = [(c / n, float(n)) for n in range(1, 8) for c in range(n + 1)]
    a = (C.c_double * 3)(); b = (C.c_double * 3)()
    got = np.empty((len(pn) ** 3, 3)); want = np.empty_like(got)
    k = 0
    for p1, n1 in pn:
        for p2, n2 in pn:
            for p3, n3 in pn:
                R.ref_calculatePBSfromAFs(p1, p2, p3, n1, n2, n3, a)
                L.orc_pbs_from_afs(p1, p2, p3, n1, n2, n3, b)
                want[k] = a[:]; got[k] = b[:]; k += 1
    assert_same_bits(got, want, "PBS")
    assert (np.signbit(got) == np.signbit(want)).all()       # also the sign of a NaN: it is printed
    assert np.isnan(want).any() and (want > 0).any()
    rng = np.random.default_rng(5)
    for _ in range(20000):                                   # larger populations
        n = rng.integers(2, 400, 3)
        c = [int(rng.integers(0, x + 1)) for x in n]
        args = [c[0] / n[0], c[1] / n[1], c[2] / n[2], float(n[0]), float(n[1]), float(n[2])]
        R.ref_calculatePBSfromAFs(*args, a); L.orc_pbs_from_afs(*args, b)
        assert_same_bits(b[:], a[:], "PBS %r" % (args,))


@pytest.mark.skipif(O.ref_abs() is None, reason="oracle/_ref not built (no /root/reference here)")
def test_abs_bit_equal_to_reference():
    """orc_abs_from_afs vs the reference's own calculateABSfromAFs (evo_ABS.cpp:58-106) compiled from /root/reference:
    every quadruple of (alt, called) counts up to 3 alleles (single-allele populations, fixed differences, the swapped
    variance terms of :66, NaN signs) and random larger ones."""
    R = O.ref_abs()
    L = O.lib()
    L.orc_abs_from_afs.restype = None
    L.orc_abs_from_afs.argtypes = [C.c_double] * 8 + [O._dp]
    pn = [(c / n, float(n)) for n in range(1, 4) for c in range(n + 1)]
    a = (C.c_double * 2)(); b = (C.c_double * 2)()
    got = np.empty((len(pn) ** 4, 2)); want = np.empty_like(got)
    k = 0
    for pw, nw in pn:
        for px, nx in pn:
            for py, ny in pn:
                for pz, nz in pn:
                    R.ref_calculateABSfromAFs(pw, px, py, pz, nw, nx, ny, nz, a)
                    L.orc_abs_from_afs(pw, px, py, pz, nw, nx, ny, nz, b)
                    want[k] = a[:]; got[k] = b[:]; k += 1
    assert_same_bits(got, want, "ABS")
    assert (np.signbit(got) == np.signbit(want)).all()
    assert np.isnan(want).any() and (want > 0).any() and (want < 0).any()
    rng = np.random.default_rng(6)
    for _ in range(20000):
        n = rng.integers(2, 300, 4)
        c = [int(rng.integers(0, x + 1)) for x in n]
        args = [c[i] / n[i] for i in range(4)] + [float(x) for x in n]
        R.ref_calculateABSfromAFs(*args, a); L.orc_abs_from_afs(*args, b)
        assert_same_bits(b[:], a[:], "ABS %r" % (args,))


@pytest.mark.skipif(O.ref_dist() is None, reason="oracle/_ref not built (no /root/reference here)")
def test_dxy_from_afs_bit_equal_to_reference():
    """orc_dxy_from_afs vs the reference's own DxyPerSNPfromAFs (evo_distanceToOutgroups.cpp:62-65) compiled from
    /root/reference: every pair of allele frequencies c/n with n <= 24, the -1 sentinel of a population without a called
    allele (process_vcf_utils.h:75), NaN, and random frequencies of larger populations."""
    R = O.ref_dist()
    L = O.lib()
    L.orc_dxy_from_afs.restype = C.c_double
    L.orc_dxy_from_afs.argtypes = [C.c_double, C.c_double]
    afs = sorted({c / n for n in range(1, 25) for c in range(n + 1)}) + [-1.0, float("nan")]
    got = np.array([[L.orc_dxy_from_afs(a, b) for b in afs] for a in afs])
    want = np.array([[R.ref_DxyPerSNPfromAFs(a, b) for b in afs] for a in afs])
    assert_same_bits(got, want, "DxyPerSNPfromAFs")
    rng = np.random.default_rng(7)
    n = rng.integers(1, 4000, (20000, 2)); c = (rng.random((20000, 2)) * (n + 1)).astype(np.int64)
    a = c / n
    got = np.array([L.orc_dxy_from_afs(x, y) for x, y in a]); want = np.array([R.ref_DxyPerSNPfromAFs(x, y) for x, y in a])
    assert_same_bits(got, want, "DxyPerSNPfromAFs (random)")


@pytest.mark.skipif(O.ref_bed() is None, reason="oracle/_ref not built (no /root/reference here)")
def test_accessible_bp_equal_to_reference():
    """orc_accessible_bp vs the reference's own getAccessibleBPinRegion (process_vcf_annotation_tools.h:360-402) compiled
    from /root/reference: sorted disjoint features, abutting and overlapping ones, unsorted files, queries that start or
    end inside, before and behind the features, and a scaffold the map does not hold."""
    R = O.ref_bed()
    ip = C.POINTER(C.c_int)
    rng = np.random.default_rng(11)
    n_checked = 0
    for case in range(300):
        n = int(rng.integers(0, 40))
        kind = case % 3
        if kind == 0:                                              # sorted, disjoint
            cuts = np.sort(rng.choice(5000, 2 * n, replace=False)) if n else np.zeros(0, np.int64)
            st, en = cuts[0::2], cuts[1::2]
        elif kind == 1:                                            # sorted starts, lengths at random: overlaps and abutting
            st = np.sort(rng.integers(0, 5000, n)); en = st + rng.integers(0, 400, n)
        else:                                                      # unsorted file
            st = rng.integers(0, 5000, n); en = st + rng.integers(1, 300, n)
        st = np.ascontiguousarray(st, np.int32); en = np.ascontiguousarray(en, np.int32)
        for _ in range(40):
            a = int(rng.integers(0, 5200)); b = a + int(rng.integers(1, 1500))
            want = R.ref_accessible_bp(st.ctypes.data_as(ip), en.ctypes.data_as(ip), n, a, b, 0)
            assert O.accessible_bp(st, en, a, b) == want, (case, a, b)
            n_checked += 1
    st = np.array([0, 100], np.int32); en = np.array([50, 200], np.int32)
    assert R.ref_accessible_bp(st.ctypes.data_as(ip), en.ctypes.data_as(ip), 2, 1, 1000, 1) == 0
    assert n_checked == 12000


def test_bed_loader_quirks(tmp_path):
    """loadBedFeatureMap (process_vcf_annotation_tools.h:461-496): consecutive lines of a scaffold form its entry, a
    scaffold that comes back later replaces its earlier entry, an unknown scaffold has no accessible bases."""
    bed = tmp_path / "acc.bed"
    bed.write_text("s1\t0\t100\ns1\t200\t300\ns2\t10\t20\ns1\t1000\t1100\n")
    assert O.bed_accessible_bp(str(bed), "s2", 1, 1000) == 10
    assert O.bed_accessible_bp(str(bed), "s1", 1, 5000) == 100          # only the last run of s1 is left
    assert O.bed_accessible_bp(str(bed), "s3", 1, 5000) == 0
    bed.write_text("s1\t0\t100\ns1\t200\t300\n")
    assert O.bed_accessible_bp(str(bed), "s1", 1, 5000) == O.accessible_bp([0, 200], [100, 300], 1, 5000) == 200
    assert O.bed_accessible_bp(str(bed), "s1", 51, 250) == (100 - 51 + 1) + (250 - 200)


def test_annotation_category_quirks(tmp_path):
    """getCategoryOfSNP / loadAnnotationMap (process_vcf_annotation_tools.h:190-214, 255-295) as restated (parity
    unpinned): string comparison of positions, first line of a transcript as its only exon, partial transcripts, the
    dropped last scaffold."""
    an = tmp_path / "a.gffExtract"
    an.write_text("s1\t1000\t1200\t+\tg.1.1.1.1\n"        # transcript A: 1000..2500 (+), first exon 1000..1200
                  "s1\t2000\t2500\t+\tg.1.1.1.1\n"
                  "s1\t5000\t5200\t-\tg2\n"               # transcript B (-): start = last line's start 4000, end = first line's end 5200
                  "s1\t4000\t4100\t-\tg2\n"
                  "s1\t7000\t7100\t+\tg3\n"               # dropped: the partial line after it raises the flag first
                  "s1\t8000\t8100\t+\t5partial-g4\n"      # partial: dropped
                  "s1\t9000\t9100\t+\tg5\n"               # last transcript of s1: stored because s2 follows
                  "s2\t100\t900\t+\th1\n"
                  "s3\t100\t900\t+\tk1\n")                # last scaffold of the file: never stored (so s2 is, s3 is not)
    q = [("s1", 1100), ("s1", 1500), ("s1", 2200), ("s1", 3000), ("s1", 4050), ("s1", 5100), ("s1", 7050), ("s1", 8050),
         ("s1", 9050), ("s2", 500), ("s3", 500), ("nowhere", 5), ("s1", 15), ("s1", 100), ("s2", 1000), ("s2", 95)]
    got = O.annot_categories(str(an), q)
    assert got == ["exon", "intron", "intron", "other non-coding", "intron", "exon", "other non-coding", "other non-coding",
                   "exon", "exon", "other non-coding", "other non-coding",
                   "intron",               # "15" >= "1000" and "15" <= "2500" as strings; not <= "1200"
                   "other non-coding",     # "100" < "1000" (a prefix is smaller)
                   "exon",                 # s2: "1000" lies between "100" and "900"
                   "other non-coding"]     # "95" > "900"
    # getSNPgeneDetails (process_vcf_annotation_tools.cpp:39-80) compares integers, looks through every line of the
    # transcript, knows promoters (3 kb upstream by strand) and stops at the first transcript that holds the locus
    q = [("s1", 999), ("s1", 1100), ("s1", 1500), ("s1", 2200), ("s1", 2600), ("s1", 5300), ("s1", 8201), ("s1", 4050),
         ("s1", 4500), ("s1", 15), ("s2", 95), ("s2", 1000), ("s3", 500)]
    assert O.annot_gene_details(str(an), q) == [
        ("g.1.1.1.1", "promoter"), ("g.1.1.1.1", "exon"), ("g.1.1.1.1", "intron"), ("g.1.1.1.1", "exon"),
        ("", "nonCoding"), ("g2", "promoter"), ("g5", "promoter"), ("g2", "exon"), ("g2", "intron"), ("g.1.1.1.1", "promoter"),
        ("h1", "promoter"), ("", "nonCoding"), ("", "nonCoding")]


def test_dist_windows_against_numpy():
    """orc_dist_windows (evo_distanceToOutgroups.cpp:274-294) against a direct numpy evaluation, incl. carry."""
    rng = np.random.default_rng(9)
    S, P = 3000, 6
    nm = rng.integers(0, 9, (S, P)).astype(np.int32)
    nm[rng.random((S, P)) < 0.2] = 0
    alt = (rng.random((S, P)) * (nm + 1)).astype(np.int32).clip(0, nm)
    fl = (rng.random(S) < 0.1).astype(np.uint8)
    wo = np.array([0, 0, 500, 1700, 1700, S], np.int64)
    oi, ii = [0, 5], [1, 2, 5]
    r = O.dist_windows(alt, nm, wo, oi, ii, fl)
    with np.errstate(all="ignore"):
        af = alt / nm
    for w in range(len(wo) - 1):
        sl = slice(wo[w], wo[w + 1])
        keep = fl[sl] == 0
        for a, o in enumerate(oi):
            has_o = keep & (nm[sl, o] > 0)
            assert r["used"][w, a] == has_o.sum() and r["missing"][w, a] == (keep & (nm[sl, o] == 0)).sum()
            for b, j in enumerate(ii):
                both = has_o & (nm[sl, j] > 0)
                assert r["nsnp"][w, a, b] == both.sum() and r["miss"][w, a, b] == (has_o & (nm[sl, j] == 0)).sum()
                d = af[sl, o][both] * (1 - af[sl, j][both]) + af[sl, j][both] * (1 - af[sl, o][both])
                assert abs(r["sum"][w, a, b] - d.sum()) <= 1e-12 * max(1.0, d.sum())
    first = O.dist_windows(alt[:1000], nm[:1000], [0, 0, 500, 1000], oi, ii, fl[:1000])
    rest = O.dist_windows(alt[1000:], nm[1000:], [0, 700, 700, S - 1000], oi, ii, fl[1000:], carry={k: v[-1] for k, v in first.items()})
    for k in r:
        assert np.array_equal(np.concatenate([first[k][:-1], rest[k]]), r[k]), k     # bit-identical, sums included


def _random_diff_sites(rng, S, n, miss=0.15, multi=0.1):
    iv = rng.integers(0, 3, (S, n))
    haps = rng.integers(0, 4, (S, 2 * n))
    missing = (rng.random((S, n)) < miss).astype(np.uint8)
    mul = (rng.random(S) < multi).astype(np.uint8)
    return iv, haps, missing, mul


@pytest.mark.skipif(O.ref_diffs() is None, reason="oracle/_ref not built (no /root/reference here)")
def test_indiv_diffs_bit_equal_to_reference():
    """orc_indiv_diffs vs the reference's own diffs_between_individuals / ..._with_multialleleics
    (process_vcf_stats_functions.cpp:180-281) compiled from /root/reference: all six matrices, bit for bit (the 2/3
    weights are added up in site order in both)."""
    rng = np.random.default_rng(12)
    for S, n in ((1500, 9), (400, 23)):
        args = _random_diff_sites(rng, S, n)
        a = O.indiv_diffs(*args, use_reference=True)
        b = O.indiv_diffs(*args)
        for k in a:
            if a[k].dtype == np.float64:
                assert_same_bits(b[k], a[k], k)
            else:
                assert np.array_equal(a[k], b[k]), k
        assert a["diffs"].sum() > 0 and a["hets_vs_homs"][np.triu_indices(n, 1)].sum() > 0 and a["miss"].sum() > 0


@pytest.mark.skipif(O.ref_diffs() is None, reason="oracle/_ref not built (no /root/reference here)")
def test_indiv_hets_equal_to_reference():
    """orc_indiv_hets vs the reference's own het_analysis (process_vcf_stats_functions.cpp:19-28) compiled from
    /root/reference, called under statsMain's condition (process_vcf_stats.cpp:239-242): exact."""
    rng = np.random.default_rng(13)
    for S, n, miss in ((2000, 9, 0.02), (500, 23, 0.01), (300, 5, 0.5), (50, 4, 0.0)):
        iv, _, missing, mul = _random_diff_sites(rng, S, n, miss=miss)
        iv[mul != 0] = 0                                             # multi-allelic lines leave individualsWithVariant at 0
        iv[rng.random(S) < 0.3] = 0                                  # and many sites carry one or no ALT allele
        for s in range(0, S, 3):
            iv[s, rng.integers(0, n)] = 1
        a = O.indiv_hets(iv, missing, use_reference=True)
        b = O.indiv_hets(iv, missing)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        assert a[0].sum() > 0 and (miss > 0.4 or (a[1] < a[0]).any())


def test_indiv_diff_counts_compose_the_matrices():
    """The five counts per pair give back the literal matrices: diffs_me, hets/homs and missingness exactly (all terms
    are multiples of 1/4), Richard's matrix to 1e-12 (its 2/3 weights are rounded site by site in the reference)."""
    rng = np.random.default_rng(13)
    S, n = 3000, 17
    args = _random_diff_sites(rng, S, n)
    lit = O.indiv_diffs(*args)
    comp = O.matrices_from_diff_counts(O.indiv_diff_counts(*args), n)
    assert np.array_equal(lit["diffs_me"], comp["diffs_me"])
    assert np.array_equal(lit["hets_vs_homs"], comp["hets_vs_homs"])
    assert np.array_equal(lit["miss"], comp["miss"])
    assert np.allclose(lit["diffs"], comp["diffs"], rtol=1e-12, atol=0)


def test_stats_variant_counts_rules():
    """getThisVariantCountsSimple (process_vcf_utils.cpp:358-408): '*' last among two ALTs is a missing allele and the
    site stays bi-allelic; '*' first, or three ALTs, make it multi-allelic; a half call is a missing individual."""
    g = ["0/1:9", "1|1:3", "./1:0", "2/0:1", "0/0:5"]
    iv, hp, ms, multi = O.stats_variant_counts("G", g)
    assert multi == 0 and list(iv) == [1, 2, 1, 0, 0] and list(ms) == [0, 0, 1, 0, 0] and list(hp[6:8]) == [2, 0]
    iv, hp, ms, multi = O.stats_variant_counts("G,*", g)
    assert multi == 0 and list(ms) == [0, 0, 1, 1, 0] and list(iv) == [1, 2, 1, 0, 0]
    iv, hp, ms, multi = O.stats_variant_counts("*,G", g)
    assert multi == 1 and list(ms) == [0, 0, 1, 0, 0] and list(iv) == [0] * 5 and list(hp[:4]) == [0, 1, 1, 1]
    iv, hp, ms, multi = O.stats_variant_counts("G,T,*", g)
    assert multi == 1 and list(ms) == [0, 0, 1, 0, 0]
    iv, hp, ms, multi = O.stats_variant_counts("G,T,*", ["3/0", "0/2"])
    assert multi == 1 and list(ms) == [1, 0] and list(hp) == [0, 0, 0, 2]


def test_golden_vectors_of_the_reference_functions():
    """The oracle against tests/golden/ref_consumer_vectors.json: outputs of the reference's own calculatePBSfromAFs,
    calculateABSfromAFs and diffs_between_individuals*, recorded by tests/golden/make_golden_ref.py where /root/reference
    exists.  Bit for bit, NaN signs included; runs everywhere (the fixture travels, the reference does not)."""
    import json
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_consumer_vectors.json")))

    def from_bits(rows):
        return np.array([[int(x, 16) for x in r] for r in rows], np.uint64).view(np.float64)

    want = from_bits(g["pbs"]["pbs"])
    got = np.array([O.pbs_from_afs(*[c / n for c, n in t], *[float(n) for c, n in t]) for t in g["pbs"]["counts_c_n"]])
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64)) and np.isnan(want).any()
    want = from_bits(g["abs"]["abs_absalt"])
    got = np.array([O.abs_from_afs(*[c / n for c, n in t], *[float(n) for c, n in t]) for t in g["abs"]["counts_c_n"]])
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64)) and np.isnan(want).any()
    L = O.lib()
    want = from_bits(g["fst_math"]["p1_p2_num_den_het_hetNei78"])
    rows = []
    with np.errstate(all="ignore"):
        for c1, n1, c2, n2 in g["fst_math"]["c1_n1_c2_n2"]:
            p1 = float(np.float64(c1) / np.float64(n1)); p2 = float(np.float64(c2) / np.float64(n2))
            rows.append([p1, p2, L.orc_fst_numerator(p1, p2, n1, n2), L.orc_fst_denominator(p1, p2), L.orc_het_simple(p1),
                         L.orc_het_nei78(p1, n1)])
    got = np.array(rows)
    same = (got.view(np.uint64) == want.view(np.uint64)) | (np.isnan(got) & np.isnan(want))   # 0/0: numpy's NaN sign is its own
    assert same.all() and np.isnan(want).any() and len(rows) > 400
    d = g["indiv_diffs"]
    n = d["n"]
    r = O.indiv_diffs(d["ind_with_variant"], d["haps"], np.array(d["missing"], np.uint8), np.array(d["multi"], np.uint8))
    for k in ("diffs", "diffs_me", "hets_vs_homs"):
        w = np.array([int(x, 16) for x in d[k]], np.uint64).reshape(n, n)
        assert np.array_equal(r[k].view(np.uint64), w), k
    assert np.array_equal(r["miss"].ravel(), d["miss"])
    n_q = 0
    for case in g["accessible_bp"]:
        for a, b, want_bp in case["start_end_bp"]:
            assert O.accessible_bp(case["starts"], case["ends"], a, b) == want_bp, (case, a, b)
            n_q += 1
    assert n_q == 1080


def test_rcp_division_identity():
    """The kernels' reciprocal-table quotient is bit-identical to IEEE division on the whole domain."""
    assert O.lib().orc_check_rcp_division(2047) == 0


def test_text_driver_documented_quirks(tmp_path):
    rng = np.random.default_rng(7)
    vcf = tmp_path / "a.vcf"
    names = write_vcf(str(vcf), rng, 600, 24, scaffolds=("chrA", "chrB"), weird=True)
    (tmp_path / "sets.txt").write_text(",".join(names[:12]) + "\n" + ",".join(names[12:]) + "\n")
    O.fst_run(str(vcf), str(tmp_path / "sets.txt"), str(tmp_path), run_name="q", size=50, step=50)
    rows = (tmp_path / "q_w_50_fst.txt").read_text().splitlines()
    assert rows[0].startswith("var_num\tscaffold\tStart\tEnd")
    # first window swallowed (start string is "scaffold_0\t0"), the next one starts at 0
    first = rows[1].split("\t")
    assert first[0] == "51" and first[2] == "0"
    # FstGlobal: misplaced header on line 2, data row glued to it; FW rows all zero
    pops = tmp_path / "pops.txt"
    pops.write_text("".join("%s\t%s\n" % (n, ["p1", "p2", "Outgroup"][i % 3]) for i, n in enumerate(names)))
    O.fstglobal_run(str(vcf), str(pops), str(tmp_path), run_name="g", size=20, step=10)
    lines = (tmp_path / "g_FstGlobal_20_10.txt").read_text().splitlines()
    assert lines[0] == "chr\tpos1\tpos2\tFstGlobal\tp1\tp2"
    assert lines[1].startswith("\tp1\tp2chrA\t")
    fw = (tmp_path / "g_FstGlobal_FW10000.txt").read_text().splitlines()
    assert fw[0] == "chr\tpos1\tpos2\tnSNPs"
    assert all(r.split("\t")[3:] == ["0", "0", "0"] for r in fw[1:]) and len(fw) > 1


def test_text_driver_matches_code_level(tmp_path):
    """The string-level fst driver and the code-level path agree on every counted site."""
    from evo_amd import codes_from_vcf_line, MODE_FST
    rng = np.random.default_rng(3)
    vcf = tmp_path / "b.vcf"
    n = 30
    names = write_vcf(str(vcf), rng, 400, n, scaffolds=("scaffold_0",), weird=True)
    (tmp_path / "sets.txt").write_text(",".join(names[:15]) + "\n" + ",".join(names[15:]) + "\n")
    summ = O.fst_run(str(vcf), str(tmp_path / "sets.txt"), str(tmp_path), run_name="c", size=1, step=1)
    rows = (tmp_path / "c_w_1_fst.txt").read_text().splitlines()[1:]
    codes, flags = [], []
    for line in open(vcf):
        if line.startswith("#"):
            continue
        c, fl, chrom, pos = codes_from_vcf_line(line.rstrip("\n"), MODE_FST, n)
        codes.append(c); flags.append(fl)
    codes = np.array(codes); flags = np.array(flags, np.uint8)
    sp = np.array([0] * 15 + [1] * 15, np.int32)
    alt, nm = O.counts_from_codes(codes, sp, 2)
    num, den, dxy, inc = O.pair_site_stats(alt, nm, [30, 30], [0], [1], flags)
    assert inc.sum() == len(rows) == summ[5]
    with np.errstate(all="ignore"):
        f = num[inc[:, 0] == 1, 0] / den[inc[:, 0] == 1, 0]
    f[f < 0] = 0
    for r, fv, dv in zip(rows, f, dxy[inc[:, 0] == 1, 0]):
        parts = r.split("\t")
        assert parts[3] == "%g" % fv and parts[4] == "%g" % dv
