"""ctypes loader for the CPU oracle (oracle/evo_oracle.c) and the reference math shim.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (evo_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libevo_oracle.so")
_REF = os.path.join(_HERE, "_ref", "libevo_refmath.so")
_REF_PBS = os.path.join(_HERE, "_ref", "libevo_refpbs.so")
_REF_ABS = os.path.join(_HERE, "_ref", "libevo_refabs.so")
_REF_DIFFS = os.path.join(_HERE, "_ref", "libevo_refdiffs.so")
_REF_DIST = os.path.join(_HERE, "_ref", "libevo_refdist.so")
_REF_BED = os.path.join(_HERE, "_ref", "libevo_refbed.so")

_dp = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_u8p = C.POINTER(C.c_uint8)
_ip = C.POINTER(C.c_int)


def build(force=False):
    """(Re)build the oracle library; also the reference shim when /root/reference exists."""
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(
            os.path.join(_HERE, "evo_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "libevo_oracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference") and (force or not os.path.exists(_REF) or not os.path.exists(_REF_PBS)
                                             or not os.path.exists(_REF_ABS) or not os.path.exists(_REF_DIFFS)
                                             or not os.path.exists(_REF_DIST)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


class SetCounts(C.Structure):
    _fields_ = [("set1Count", C.c_int), ("set2Count", C.c_int),
                ("set1_n_withoutMissing", C.c_int), ("set2_n_withoutMissing", C.c_int),
                ("n_alt_alleles", C.c_int), ("bIndel", C.c_int),
                ("alt_code", C.c_int), ("allele_as_missing", C.c_int)]


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.orc_fst_numerator.restype = C.c_double
        L.orc_fst_numerator.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int]
        L.orc_fst_denominator.restype = C.c_double
        L.orc_fst_denominator.argtypes = [C.c_double, C.c_double]
        L.orc_het_simple.restype = C.c_double
        L.orc_het_simple.argtypes = [C.c_double]
        L.orc_het_nei78.restype = C.c_double
        L.orc_het_nei78.argtypes = [C.c_double, C.c_int]
        for f in ("orc_vector_sum", "orc_vector_average"):
            getattr(L, f).restype = C.c_double
            getattr(L, f).argtypes = [_dp, C.c_size_t]
        L.orc_vector_average_with_region.restype = C.c_double
        L.orc_vector_average_with_region.argtypes = [_dp, C.c_size_t, C.c_int]
        L.orc_calculate_fst.restype = C.c_double
        L.orc_calculate_fst.argtypes = [_dp, _dp, C.c_size_t]
        L.orc_dxy_literal.restype = C.c_double
        L.orc_dxy_literal.argtypes = [_ip, _ip, C.c_int, C.c_int]
        L.orc_pi_literal.restype = C.c_double
        L.orc_pi_literal.argtypes = [_ip, C.c_int, C.c_int]
        L.orc_dxy_closed.restype = C.c_double
        L.orc_dxy_closed.argtypes = [C.c_int] * 4
        L.orc_pi_closed.restype = C.c_double
        L.orc_pi_closed.argtypes = [C.c_int] * 3
        L.orc_fst_variant_counts.restype = None
        L.orc_fst_variant_counts.argtypes = [C.c_char_p, C.POINTER(C.c_char_p),
                                             C.POINTER(C.c_size_t), C.c_int,
                                             C.POINTER(C.c_size_t), C.c_int,
                                             C.POINTER(SetCounts), _ip, _ip, _ip, _ip]
        L.orc_fst_site_included.restype = C.c_int
        L.orc_fst_site_included.argtypes = [C.POINTER(SetCounts)]
        L.orc_general_counts.restype = None
        L.orc_general_counts.argtypes = [C.POINTER(C.c_char_p), C.c_int, _i32p, C.c_int, _u8p,
                                         _ip, _ip, _dp, _ip, _dp]
        L.orc_fst_site_flags.restype = C.c_uint32
        L.orc_fst_site_flags.argtypes = [C.c_char_p, C.c_char_p, _ip, _ip]
        L.orc_fst_allele_code.restype = C.c_uint8
        L.orc_fst_allele_code.argtypes = [C.c_char, C.c_int, C.c_int]
        L.orc_general_site_flags.restype = C.c_uint32
        L.orc_general_site_flags.argtypes = [C.c_char_p, C.c_char_p]
        L.orc_general_allele_code.restype = C.c_uint8
        L.orc_general_allele_code.argtypes = [C.c_char]
        L.orc_counts_from_codes.restype = None
        L.orc_counts_from_codes.argtypes = [_u8p, C.c_int64, C.c_int, _i32p, C.c_int, _i32p, _i32p]
        L.orc_pair_site_stats.restype = None
        L.orc_pair_site_stats.argtypes = [_i32p, _i32p, C.c_int64, C.c_int, _i32p, _i32p, _i32p,
                                          C.c_int, _u8p, _dp, _dp, _dp, _u8p]
        L.orc_pop_site_div.restype = None
        L.orc_pop_site_div.argtypes = [_i32p, _i32p, C.c_int64, C.c_int, _i32p, _dp, _dp, _dp]
        L.orc_snp_windows.restype = C.c_int64
        L.orc_snp_windows.argtypes = [_dp, _dp, _dp, _u8p, C.c_int64, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int64, _i64p, _i64p, _dp, _dp, _dp, _dp, _dp]
        L.orc_global_site_stats.restype = None
        L.orc_global_site_stats.argtypes = [_i32p, _i32p, C.c_int64, C.c_int, _i32p, C.c_int, _i64p,
                                            _u8p, _u8p, _dp, _dp]
        L.orc_global_windows.restype = C.c_int64
        L.orc_global_windows.argtypes = [_u8p, _u8p, _dp, _dp, C.c_int64, C.c_int, C.c_int, C.c_int,
                                         C.c_int64, _i64p, _i64p, _dp]
        L.orc_fst_run.restype = C.c_int
        L.orc_fst_run.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int,
                                  C.c_double, C.c_char_p, C.c_char_p, _dp]
        L.orc_af_run.restype = C.c_int
        L.orc_af_run.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p]
        L.orc_fstglobal_run.restype = C.c_int
        L.orc_fstglobal_run.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int,
                                        C.c_char_p, C.c_char_p]
        L.orc_check_rcp_division.restype = C.c_long
        L.orc_check_rcp_division.argtypes = [C.c_int]
        _lib = L
    return _lib


def ref():
    """The reference's own inline math, or None when oracle/_ref was never built."""
    global _ref
    if _ref is None:
        build()
        if not os.path.exists(_REF):
            return None
        R = C.CDLL(_REF)
        R.ref_calculateFstNumerator.restype = C.c_double
        R.ref_calculateFstNumerator.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int]
        R.ref_calculateFstDenominator.restype = C.c_double
        R.ref_calculateFstDenominator.argtypes = [C.c_double, C.c_double]
        R.ref_calculateExpectedHeterozygositySimple.restype = C.c_double
        R.ref_calculateExpectedHeterozygositySimple.argtypes = [C.c_double]
        R.ref_calculateExpectedHeterozygosityNei78.restype = C.c_double
        R.ref_calculateExpectedHeterozygosityNei78.argtypes = [C.c_double, C.c_int]
        for f in ("ref_vector_sum", "ref_vector_average"):
            getattr(R, f).restype = C.c_double
            getattr(R, f).argtypes = [_dp, C.c_size_t]
        R.ref_vector_average_withRegion.restype = C.c_double
        R.ref_vector_average_withRegion.argtypes = [_dp, C.c_size_t, C.c_int]
        R.ref_calculateAlleleFrequencies.restype = None
        R.ref_calculateAlleleFrequencies.argtypes = [C.c_int] * 4 + [_dp, _dp]
        R.ref_general_ctor.restype = C.c_int
        R.ref_general_ctor.argtypes = [C.POINTER(C.c_char_p), _ip, C.c_int, C.POINTER(C.c_long), _dp, _dp]
        R.ref_build_info.restype = C.c_char_p
        _ref = R
    return _ref


_ref_pbs = None


def ref_pbs():
    """The reference's own calculatePBSfromAFs (evo_PBS.cpp:69-98), or None when oracle/_ref was never built."""
    global _ref_pbs
    if _ref_pbs is None:
        build()
        if not os.path.exists(_REF_PBS):
            return None
        R = C.CDLL(_REF_PBS)
        R.ref_calculatePBSfromAFs.restype = None
        R.ref_calculatePBSfromAFs.argtypes = [C.c_double] * 6 + [_dp]
        _ref_pbs = R
    return _ref_pbs


_ref_dist = None


def ref_dist():
    """The reference's own DxyPerSNPfromAFs (evo_distanceToOutgroups.cpp:62-65), or None when oracle/_ref was never built."""
    global _ref_dist
    if _ref_dist is None:
        build()
        if not os.path.exists(_REF_DIST):
            return None
        R = C.CDLL(_REF_DIST)
        R.ref_DxyPerSNPfromAFs.restype = C.c_double
        R.ref_DxyPerSNPfromAFs.argtypes = [C.c_double, C.c_double]
        _ref_dist = R
    return _ref_dist


_ref_bed = None


def ref_bed():
    """The reference's own AccessibleGenome::getAccessibleBPinRegion (process_vcf_annotation_tools.h:360-402, 510-512)
    behind ref_accessible_bp(starts, ends, n, start, end, scaffold_absent), or None when oracle/_ref was never built."""
    global _ref_bed
    if _ref_bed is None:
        build()
        if not os.path.exists(_REF_BED):
            return None
        R = C.CDLL(_REF_BED)
        ip = C.POINTER(C.c_int)
        R.ref_accessible_bp.restype = C.c_int
        R.ref_accessible_bp.argtypes = [ip, ip, C.c_int, C.c_int, C.c_int, C.c_int]
        _ref_bed = R
    return _ref_bed


def accessible_bp(starts, ends, start, end):
    """Accessible bases of the 1-based region [start, end] given one scaffold's BED features in file order."""
    st = np.ascontiguousarray(starts, np.int32); en = np.ascontiguousarray(ends, np.int32)
    L = lib()
    ip = C.POINTER(C.c_int)
    L.orc_accessible_bp.restype = C.c_int
    L.orc_accessible_bp.argtypes = [ip, ip, C.c_int, C.c_int, C.c_int]
    return int(L.orc_accessible_bp(st.ctypes.data_as(ip), en.ctypes.data_as(ip), len(st), int(start), int(end)))


def bed_accessible_bp(bed_path, scaffold, start, end):
    """The same through the BED loader (loadBedFeatureMap, process_vcf_annotation_tools.h:461-496)."""
    L = lib()
    L.orc_bed_load.restype = C.c_void_p
    L.orc_bed_load.argtypes = [C.c_char_p]
    L.orc_bed_free.restype = None
    L.orc_bed_free.argtypes = [C.c_void_p]
    L.orc_bed_accessible_bp.restype = C.c_int
    L.orc_bed_accessible_bp.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int]
    b = L.orc_bed_load(bed_path.encode())
    if not b:
        raise RuntimeError("could not open " + bed_path)
    try:
        return int(L.orc_bed_accessible_bp(b, scaffold.encode(), int(start), int(end)))
    finally:
        L.orc_bed_free(b)


_ref_abs = None


def ref_abs():
    """The reference's own calculateABSfromAFs (evo_ABS.cpp:58-106), or None when oracle/_ref was never built."""
    global _ref_abs
    if _ref_abs is None:
        build()
        if not os.path.exists(_REF_ABS):
            return None
        R = C.CDLL(_REF_ABS)
        R.ref_calculateABSfromAFs.restype = None
        R.ref_calculateABSfromAFs.argtypes = [C.c_double] * 8 + [_dp]
        _ref_abs = R
    return _ref_abs


def abs_from_afs(pW, pX, pY, pZ, nW, nX, nY, nZ):
    L = lib()
    L.orc_abs_from_afs.restype = None
    L.orc_abs_from_afs.argtypes = [C.c_double] * 8 + [_dp]
    out = (C.c_double * 2)()
    L.orc_abs_from_afs(pW, pX, pY, pZ, nW, nX, nY, nZ, out)
    return np.array(out[:])


def abs_site_stats(alt, nm, quartets, site_flags=None):
    """(abs [T, 2, S], used [T, S]) from counts [S, P] (evo_ABS.cpp:58-106, 243-262)."""
    L = lib()
    alt = np.ascontiguousarray(alt, np.int32); nm = np.ascontiguousarray(nm, np.int32)
    S, P = alt.shape
    qt = np.ascontiguousarray(quartets, np.int32).reshape(-1, 4)
    fl = np.ascontiguousarray(site_flags, np.uint8) if site_flags is not None else None
    out = np.empty((len(qt), 2, S)); used = np.empty((len(qt), S), np.uint8)
    L.orc_abs_site_stats.restype = None
    L.orc_abs_site_stats.argtypes = [_i32p, _i32p, C.c_int64, C.c_int, _i32p, C.c_int, _u8p, _dp, _u8p]
    L.orc_abs_site_stats(_p(alt, _i32p), _p(nm, _i32p), S, P, _p(qt, _i32p), len(qt), _p(fl, _u8p), _p(out, _dp), _p(used, _u8p))
    return out, used


_ref_diffs = None
_u8p_ = C.POINTER(C.c_uint8)


def ref_diffs():
    """The reference's own diffs_between_individuals[_with_multialleleics] (process_vcf_stats_functions.cpp:180-281),
    or None when oracle/_ref was never built."""
    global _ref_diffs
    if _ref_diffs is None:
        build()
        if not os.path.exists(_REF_DIFFS):
            return None
        R = C.CDLL(_REF_DIFFS)
        R.ref_indiv_diffs.restype = None
        R.ref_indiv_diffs.argtypes = [C.c_int, C.c_long, _ip, _ip, _u8p_, _u8p_, _dp, _dp, _dp, _dp, _ip, _ip]
        _ref_diffs = R
    return _ref_diffs


def _diff_inputs(iv, haps, missing, multi):
    iv = np.ascontiguousarray(iv, np.intc); haps = np.ascontiguousarray(haps, np.intc)
    missing = np.ascontiguousarray(missing, np.uint8); multi = np.ascontiguousarray(multi, np.uint8)
    S, n = iv.shape
    assert haps.shape == (S, 2 * n) and missing.shape == (S, n) and multi.shape == (S,)
    return iv, haps, missing, multi, S, n


def indiv_diffs(iv, haps, missing, multi, use_reference=False):
    """Literal per-site matrix updates -> dict(diffs, diffs_me, boot, hets_vs_homs [n, n] doubles, miss, boot_miss ints)."""
    iv, haps, missing, multi, S, n = _diff_inputs(iv, haps, missing, multi)
    r = dict(diffs=np.zeros((n, n)), diffs_me=np.zeros((n, n)), boot=np.zeros((n, n)), hets_vs_homs=np.zeros((n, n)),
             miss=np.zeros((n, n), np.intc), boot_miss=np.zeros((n, n), np.intc))
    args = [n, S, _p(iv, _ip), _p(haps, _ip), _p(missing, _u8p), _p(multi, _u8p), _p(r["diffs"], _dp), _p(r["diffs_me"], _dp),
            _p(r["boot"], _dp), _p(r["hets_vs_homs"], _dp), _p(r["miss"], _ip), _p(r["boot_miss"], _ip)]
    if use_reference:
        ref_diffs().ref_indiv_diffs(*args)
    else:
        L = lib()
        L.orc_indiv_diffs.restype = None
        L.orc_indiv_diffs.argtypes = [C.c_int, C.c_int64, _ip, _ip, _u8p, _u8p, _dp, _dp, _dp, _dp, _ip, _ip]
        L.orc_indiv_diffs(*args)
    return r


def indiv_diff_counts(iv, haps, missing, multi):
    """The same sites as five counts per pair j <= i (triangular index): dict(hh, hd, oh, ms, d4) of int32 [n(n+1)/2]."""
    iv, haps, missing, multi, S, n = _diff_inputs(iv, haps, missing, multi)
    T = n * (n + 1) // 2
    r = {k: np.zeros(T, np.int32) for k in ("hh", "hd", "oh", "ms", "d4")}
    L = lib()
    L.orc_indiv_diff_counts.restype = None
    L.orc_indiv_diff_counts.argtypes = [C.c_int, C.c_int64, _ip, _ip, _u8p, _u8p] + [_i32p] * 5
    L.orc_indiv_diff_counts(n, S, _p(iv, _ip), _p(haps, _ip), _p(missing, _u8p), _p(multi, _u8p),
                            *[_p(r[k], _i32p) for k in ("hh", "hd", "oh", "ms", "d4")])
    return r


def matrices_from_diff_counts(c, n):
    """Compose the reference's four matrices from the five counts (the closed form the CLI uses)."""
    D = np.zeros((n, n)); Dme = np.zeros((n, n)); H = np.zeros((n, n)); M = np.zeros((n, n), np.intc)
    i, j = np.tril_indices(n)
    o = i * (i + 1) // 2 + j
    off = j < i
    hh, hd, oh, ms, d4 = (c[k][o].astype(np.float64) for k in ("hh", "hd", "oh", "ms", "d4"))
    D[i, j] = np.where(off, (2.0 / 3.0) * hh + (2.0 / 3.0) * hd + 0.5 * oh, hh)
    Dme[i, j] = np.where(off, 0.5 * hh + hd + 0.5 * oh + d4 / 4.0, hh + d4)
    H[i[off], j[off]] = hh[off]; H[j[off], i[off]] = hd[off]
    M[i, j] = ms
    return dict(diffs=D, diffs_me=Dme, hets_vs_homs=H, miss=M)


def stats_variant_counts(alt_field, genotypes):
    L = lib()
    n = len(genotypes)
    g = (C.c_char_p * n)(*[s.encode() for s in genotypes])
    iv = np.zeros(n, np.intc); hp = np.zeros(2 * n, np.intc); ms = np.zeros(n, np.uint8); multi = C.c_int(0)
    L.orc_stats_variant_counts.restype = None
    L.orc_stats_variant_counts.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), C.c_int, _ip, _ip, _u8p, C.POINTER(C.c_int)]
    L.orc_stats_variant_counts(alt_field.encode(), g, n, _p(iv, _ip), _p(hp, _ip), _p(ms, _u8p), C.byref(multi))
    return iv, hp, ms, multi.value


def indiv_hets(iv, missing, use_reference=False):
    """het_analysis over the sites without a missing genotype -> (het, shared) int32 [n]; use_reference: the reference's
    own function (oracle/_ref/libevo_refdiffs.so)."""
    iv = np.ascontiguousarray(iv, np.intc); missing = np.ascontiguousarray(missing, np.uint8)
    S, n = iv.shape
    het = np.zeros(n, np.int32); shared = np.zeros(n, np.int32)
    if use_reference:
        R = ref_diffs()
        R.ref_het_analysis.restype = None
        R.ref_het_analysis.argtypes = [C.c_int, C.c_long, _ip, _u8p, _ip, _ip]
        R.ref_het_analysis(n, S, _p(iv, _ip), _p(missing, _u8p), _p(het, _ip), _p(shared, _ip))
        return het, shared
    L = lib()
    L.orc_indiv_hets.restype = None
    L.orc_indiv_hets.argtypes = [C.c_int, C.c_int64, _ip, _u8p, _i32p, _i32p]
    L.orc_indiv_hets(n, S, _p(iv, _ip), _p(missing, _u8p), _p(het, _i32p), _p(shared, _i32p))
    return het, shared


def stats_run(vcf, out_dir, diffs=True, hets=False, block=0, accessible=-1):
    L = lib()
    L.orc_stats_run.restype = C.c_int
    L.orc_stats_run.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p]
    rc = L.orc_stats_run(vcf.encode(), 1 if diffs else 0, 1 if hets else 0, block, accessible, out_dir.encode())
    if rc != 0:
        raise RuntimeError("orc_stats_run failed: %d" % rc)


def stats_diffs_run(vcf, out_dir, block=0, accessible=-1):
    L = lib()
    L.orc_stats_diffs_run.restype = C.c_int
    L.orc_stats_diffs_run.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_char_p]
    rc = L.orc_stats_diffs_run(vcf.encode(), block, accessible, out_dir.encode())
    if rc != 0:
        raise RuntimeError("orc_stats_diffs_run failed: %d" % rc)


def pbs_from_afs(p1, p2, p3, n1, n2, n3):
    L = lib()
    L.orc_pbs_from_afs.restype = None
    L.orc_pbs_from_afs.argtypes = [C.c_double] * 6 + [_dp]
    out = (C.c_double * 3)()
    L.orc_pbs_from_afs(p1, p2, p3, n1, n2, n3, out)
    return np.array(out[:])


# ---------------------------------------------------------------------------
# numpy-level conveniences
# ---------------------------------------------------------------------------

def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def fst_variant_counts(alt_field, genotypes, set1, set2):
    """Literal getVariantCountsForFst on genotype strings. Returns (SetCounts, indiv1, indiv2, hap1, hap2)."""
    L = lib()
    g = (C.c_char_p * len(genotypes))(*[s.encode() for s in genotypes])
    s1 = (C.c_size_t * max(1, len(set1)))(*set1)
    s2 = (C.c_size_t * max(1, len(set2)))(*set2)
    out = SetCounts()
    i1 = np.zeros(max(1, len(set1)), np.intc); i2 = np.zeros(max(1, len(set2)), np.intc)
    h1 = np.zeros(max(2, 2 * len(set1)), np.intc); h2 = np.zeros(max(2, 2 * len(set2)), np.intc)
    L.orc_fst_variant_counts(alt_field.encode(), g, s1, len(set1), s2, len(set2), C.byref(out),
                             _p(i1, _ip), _p(i2, _ip), _p(h1, _ip), _p(h2, _ip))
    return out, i1[:len(set1)], i2[:len(set2)], h1[:2 * len(set1)], h2[:2 * len(set2)]


def general_counts(genotypes, sample_pop, n_pops, use_pop):
    L = lib()
    g = (C.c_char_p * len(genotypes))(*[s.encode() for s in genotypes])
    sp = np.ascontiguousarray(sample_pop, np.int32)
    up = np.ascontiguousarray(use_pop, np.uint8)
    alt = np.zeros(n_pops, np.intc); alleles = np.zeros(n_pops, np.intc); allc = np.zeros(n_pops, np.intc)
    aaf = np.zeros(n_pops); aafc = np.zeros(n_pops)
    L.orc_general_counts(g, len(genotypes), _p(sp, _i32p), n_pops, _p(up, _u8p),
                         _p(alt, _ip), _p(alleles, _ip), _p(aaf, _dp), _p(allc, _ip), _p(aafc, _dp))
    return alt, alleles, aaf, allc, aafc


def counts_from_codes(codes, sample_pop, n_pops):
    """codes: uint8 [S, 2N] -> (alt, nm) int32 [S, P]."""
    L = lib()
    codes = np.ascontiguousarray(codes, np.uint8)
    S, H = codes.shape
    sp = np.ascontiguousarray(sample_pop, np.int32)
    alt = np.empty((S, n_pops), np.int32); nm = np.empty((S, n_pops), np.int32)
    L.orc_counts_from_codes(_p(codes, _u8p), S, H // 2, _p(sp, _i32p), n_pops, _p(alt, _i32p), _p(nm, _i32p))
    return alt, nm


def pair_site_stats(alt, nm, n_full, pair_i, pair_j, site_flags=None):
    L = lib()
    alt = np.ascontiguousarray(alt, np.int32); nm = np.ascontiguousarray(nm, np.int32)
    S, P = alt.shape
    nf = np.ascontiguousarray(n_full, np.int32)
    pi = np.ascontiguousarray(pair_i, np.int32); pj = np.ascontiguousarray(pair_j, np.int32)
    Q = len(pi)
    fl = np.ascontiguousarray(site_flags, np.uint8) if site_flags is not None else None
    num = np.empty((S, Q)); den = np.empty((S, Q)); dxy = np.empty((S, Q)); inc = np.empty((S, Q), np.uint8)
    L.orc_pair_site_stats(_p(alt, _i32p), _p(nm, _i32p), S, P, _p(nf, _i32p), _p(pi, _i32p), _p(pj, _i32p), Q,
                          _p(fl, _u8p), _p(num, _dp), _p(den, _dp), _p(dxy, _dp), _p(inc, _u8p))
    return num, den, dxy, inc


def pop_site_div(alt, nm, n_full):
    L = lib()
    alt = np.ascontiguousarray(alt, np.int32); nm = np.ascontiguousarray(nm, np.int32)
    S, P = alt.shape
    nf = np.ascontiguousarray(n_full, np.int32)
    het = np.empty((S, P)); hn = np.empty((S, P)); pi = np.empty((S, P))
    L.orc_pop_site_div(_p(alt, _i32p), _p(nm, _i32p), S, P, _p(nf, _i32p), _p(het, _dp), _p(hn, _dp), _p(pi, _dp))
    return het, hn, pi


def snp_windows(num, den, dxy, inc, pair, size, step):
    L = lib()
    S, Q = num.shape
    mw = S // max(1, step) + 1
    fv = np.empty(mw, np.int64); es = np.empty(mw, np.int64)
    sn = np.empty(mw); sd = np.empty(mw); sx = np.empty(mw); fst = np.empty(mw); md = np.empty(mw)
    n = L.orc_snp_windows(_p(num, _dp), _p(den, _dp), _p(dxy, _dp), _p(inc, _u8p), S, Q, pair, size, step, mw,
                          _p(fv, _i64p), _p(es, _i64p), _p(sn, _dp), _p(sd, _dp), _p(sx, _dp), _p(fst, _dp), _p(md, _dp))
    return dict(first_var=fv[:n], end_site=es[:n], sum_num=sn[:n], sum_den=sd[:n], sum_dxy=sx[:n],
                fst=fst[:n], mean_dxy=md[:n])


def global_site_stats(alt, nm, use_idx, set_sizes, site_flags=None):
    L = lib()
    alt = np.ascontiguousarray(alt, np.int32); nm = np.ascontiguousarray(nm, np.int32)
    S, P = alt.shape
    ui = np.ascontiguousarray(use_idx, np.int32); ss = np.ascontiguousarray(set_sizes, np.int64)
    U = len(ui)
    fl = np.ascontiguousarray(site_flags, np.uint8) if site_flags is not None else None
    kept = np.zeros(S, np.uint8); num = np.zeros((S, U)); den = np.zeros((S, U))
    L.orc_global_site_stats(_p(alt, _i32p), _p(nm, _i32p), S, P, _p(ui, _i32p), U, _p(ss, _i64p),
                            _p(fl, _u8p), _p(kept, _u8p), _p(num, _dp), _p(den, _dp))
    return kept, num, den


def global_windows(kept, num, den, size, step, site_flags=None):
    L = lib()
    S, U = num.shape
    fl = np.ascontiguousarray(site_flags, np.uint8) if site_flags is not None else None
    mr = S // max(1, step) + 1
    ss = np.empty(mr, np.int64); es = np.empty(mr, np.int64); fst = np.empty((mr, U))
    n = L.orc_global_windows(_p(kept, _u8p), _p(fl, _u8p), _p(num, _dp), _p(den, _dp), S, U, size, step, mr,
                             _p(ss, _i64p), _p(es, _i64p), _p(fst, _dp))
    return dict(start_site=ss[:n], end_site=es[:n], fst=fst[:n])


def fst_run(vcf, sets, out_dir, run_name="", size=0, step=0, phys=10000, reg_above=0.0, samples=None, bed=None, annot=None, anc_sets=None):
    summ = np.zeros(7)
    L = lib()
    L.orc_fst_run_opts.restype = C.c_int
    L.orc_fst_run_opts.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int,
                                   C.c_double, C.c_char_p, C.c_char_p, _dp, C.c_char_p, C.c_char_p, C.c_char_p]
    rc = L.orc_fst_run_opts(vcf.encode(), sets.encode(), (samples or "").encode(), size, step, phys,
                            reg_above, run_name.encode(), out_dir.encode(), _p(summ, _dp), (bed or "").encode(),
                            (annot or "").encode(), (anc_sets or "").encode())
    if rc != 0:
        raise RuntimeError("orc_fst_run failed: %d" % rc)
    return summ


def annot_gene_details(annot_path, queries):
    """getSNPgeneDetails (process_vcf_annotation_tools.cpp:39-80) of (scaffold, position) pairs: [(gene, category)]."""
    L = lib()
    L.orc_annot_load.restype = C.c_void_p; L.orc_annot_load.argtypes = [C.c_char_p]
    L.orc_annot_free.argtypes = [C.c_void_p]
    L.orc_annot_gene_details.restype = C.c_char_p
    L.orc_annot_gene_details.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_char_p)]
    a = L.orc_annot_load(annot_path.encode())
    if not a:
        raise RuntimeError("orc_annot_load failed")
    try:
        out = []
        for sc, loc in queries:
            g = C.c_char_p()
            cat = L.orc_annot_gene_details(a, sc.encode(), int(loc), C.byref(g)).decode()
            out.append((g.value.decode(), cat))
        return out
    finally:
        L.orc_annot_free(a)


def annot_categories(annot_path, queries):
    """getCategoryOfSNP (process_vcf_annotation_tools.h:190-214) of (scaffold, position string) pairs."""
    L = lib()
    L.orc_annot_load.restype = C.c_void_p; L.orc_annot_load.argtypes = [C.c_char_p]
    L.orc_annot_free.argtypes = [C.c_void_p]
    L.orc_annot_category.restype = C.c_char_p; L.orc_annot_category.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
    a = L.orc_annot_load(annot_path.encode())
    if not a:
        raise RuntimeError("orc_annot_load failed")
    try:
        return [L.orc_annot_category(a, sc.encode(), str(loc).encode()).decode() for sc, loc in queries]
    finally:
        L.orc_annot_free(a)


def pbs_site_stats(alt, nm, trios, site_flags=None):
    """(pbs [T, 3, S], used [T, S]) from counts [S, P] (evo_PBS.cpp:69-98, 304-322)."""
    L = lib()
    alt = np.ascontiguousarray(alt, np.int32); nm = np.ascontiguousarray(nm, np.int32)
    S, P = alt.shape
    tr = np.ascontiguousarray(trios, np.int32).reshape(-1, 3)
    fl = np.ascontiguousarray(site_flags, np.uint8) if site_flags is not None else None
    pbs = np.empty((len(tr), 3, S)); used = np.empty((len(tr), S), np.uint8)
    L.orc_pbs_site_stats.restype = None
    L.orc_pbs_site_stats.argtypes = [_i32p, _i32p, C.c_int64, C.c_int, _i32p, C.c_int, _u8p, _dp, _u8p]
    L.orc_pbs_site_stats(_p(alt, _i32p), _p(nm, _i32p), S, P, _p(tr, _i32p), len(tr), _p(fl, _u8p), _p(pbs, _dp), _p(used, _u8p))
    return pbs, used


def pbs_run(vcf, pops, trios, out_dir, run_name="", size=20, step=10, fixed=10000, annot=None, allow_indels=False, af=False):
    L = lib()
    L.orc_set_allow_indels(1 if allow_indels else 0)
    L.orc_set_pbs_af(1 if af else 0)
    L.orc_pbs_run_annot.restype = C.c_int
    L.orc_pbs_run_annot.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p]
    rc = L.orc_pbs_run_annot(vcf.encode(), pops.encode(), trios.encode(), size, step, fixed, run_name.encode(), out_dir.encode(),
                             (annot or "").encode())
    L.orc_set_allow_indels(0)
    L.orc_set_pbs_af(0)
    if rc != 0:
        raise RuntimeError("orc_pbs_run failed: %d" % rc)


def abs_run(vcf, pops, quartets, out_dir, run_name="", size=20, step=10):
    L = lib()
    L.orc_abs_run.restype = C.c_int
    L.orc_abs_run.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_char_p]
    rc = L.orc_abs_run(vcf.encode(), pops.encode(), quartets.encode(), size, step, run_name.encode(), out_dir.encode())
    if rc != 0:
        raise RuntimeError("orc_abs_run failed: %d" % rc)


def dist_windows(alt, nm, win_off, out_idx, in_idx, site_flags=None, carry=None):
    """DistOutMain's accumulation per window (evo_distanceToOutgroups.cpp:274-294) from counts [S, P]."""
    L = lib()
    alt = np.ascontiguousarray(alt, np.int32); nm = np.ascontiguousarray(nm, np.int32)
    S, P = alt.shape
    wo = np.ascontiguousarray(win_off, np.int64)
    oi = np.ascontiguousarray(out_idx, np.int32); ii = np.ascontiguousarray(in_idx, np.int32)
    W, O, I = len(wo) - 1, len(oi), len(ii)
    fl = np.ascontiguousarray(site_flags, np.uint8) if site_flags is not None else None
    r = dict(sum=np.zeros((W, O, I)), nsnp=np.zeros((W, O, I), np.int32), miss=np.zeros((W, O, I), np.int32),
             used=np.zeros((W, O), np.int32), missing=np.zeros((W, O), np.int32))
    if carry is not None:
        for k in r:
            r[k][0] = carry[k]
    L.orc_dist_windows.restype = None
    L.orc_dist_windows.argtypes = [_i32p, _i32p, C.c_int, _u8p, _i64p, C.c_int, _i32p, C.c_int, _i32p, C.c_int, C.c_int,
                                   _dp, _i32p, _i32p, _i32p, _i32p]
    L.orc_dist_windows(_p(alt, _i32p), _p(nm, _i32p), P, _p(fl, _u8p), _p(wo, _i64p), W, _p(oi, _i32p), O, _p(ii, _i32p), I,
                       0 if carry is None else 1, _p(r["sum"], _dp), _p(r["nsnp"], _i32p), _p(r["miss"], _i32p),
                       _p(r["used"], _i32p), _p(r["missing"], _i32p))
    return r


def distout_run(vcf, pops, outgroups, ingroups, out_dir, run_name="", fixed=10000, bed=None, allow_indels=False):
    L = lib()
    L.orc_set_allow_indels(1 if allow_indels else 0)
    L.orc_distout_run_bed.restype = C.c_int
    L.orc_distout_run_bed.argtypes = [C.c_char_p] * 4 + [C.c_int, C.c_char_p, C.c_char_p, C.c_char_p]
    rc = L.orc_distout_run_bed(vcf.encode(), pops.encode(), outgroups.encode(), ingroups.encode(), fixed, run_name.encode(),
                               out_dir.encode(), bed.encode() if bed else None)
    L.orc_set_allow_indels(0)
    if rc != 0:
        raise RuntimeError("orc_distout_run failed: %d" % rc)


def af_run(vcf, pops, out_dir, run_name="out"):
    rc = lib().orc_af_run(vcf.encode(), pops.encode(), run_name.encode(), out_dir.encode())
    if rc != 0:
        raise RuntimeError("orc_af_run failed: %d" % rc)


def fstglobal_run(vcf, pops, out_dir, run_name="", size=20, step=10, fixed=10000, annot=None):
    if size <= 0 or step <= 0:
        raise ValueError("FstGlobal needs positive window size and step")
    L = lib()
    L.orc_fstglobal_run_annot.restype = C.c_int
    L.orc_fstglobal_run_annot.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p]
    rc = L.orc_fstglobal_run_annot(vcf.encode(), pops.encode(), size, step, fixed, run_name.encode(), out_dir.encode(),
                                   (annot or "").encode())
    if rc != 0:
        raise RuntimeError("orc_fstglobal_run failed: %d" % rc)
