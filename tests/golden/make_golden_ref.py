#!/usr/bin/env python3
"""Writes tests/golden/ref_consumer_vectors.json: inputs and outputs of the REFERENCE's own functions, called through
the shims of oracle/_ref (compiled from /root/reference where it lies; see oracle/Makefile):
    calculatePBSfromAFs                             evo_PBS.cpp:69-98
    calculateABSfromAFs                             evo_ABS.cpp:58-106
    diffs_between_individuals[_with_multialleleics] process_vcf_stats_functions.cpp:180-281
    AccessibleGenome::getAccessibleBPinRegion       process_vcf_annotation_tools.h:360-402
    calculateAlleleFrequencies, calculateFstNumerator / Denominator, calculateExpectedHeterozygositySimple / Nei78
                                                    process_vcf_utils.h:272-275, process_vcf_fst.h:19-45
Doubles are stored as the hex of their 64 bits, so the fixture pins NaN signs too.  Needs /root/reference (run in the
build container); the fixture travels, the reference does not.
python3 tests/golden/make_golden_ref.py"""
import ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O

def bits(a):
    return ["%016x" % int(x) for x in np.ascontiguousarray(a, np.float64).view(np.uint64).ravel()]

rng = np.random.default_rng(20261110)
out = {"about": "outputs of the reference's own functions via oracle/_ref (g++ -std=c++11 -O2 -ffp-contract=off); doubles as uint64 hex"}

Rp, Ra = O.ref_pbs(), O.ref_abs()
assert Rp is not None and Ra is not None and O.ref_diffs() is not None, "oracle/_ref is not built"
small = [(c, n) for n in range(1, 4) for c in range(n + 1)]
cases = [(a, b, c) for a in small for b in small for c in small]                      # every triple up to 3 alleles
cases += [tuple((int(rng.integers(0, n + 1)), int(n)) for n in rng.integers(2, 400, 3)) for _ in range(300)]
buf = (C.c_double * 3)()
pbs_in, pbs_out = [], []
for t in cases:
    args = [t[k][0] / t[k][1] for k in range(3)] + [float(t[k][1]) for k in range(3)]
    Rp.ref_calculatePBSfromAFs(*args, buf)
    pbs_in.append([list(x) for x in t]); pbs_out.append(bits(buf[:]))
out["pbs"] = {"counts_c_n": pbs_in, "pbs": pbs_out}

cases = [(a, b, c, d) for a in small[:5] + small[-2:] for b in small for c in small[::2] for d in small[1::2]]
cases += [tuple((int(rng.integers(0, n + 1)), int(n)) for n in rng.integers(2, 300, 4)) for _ in range(300)]
buf = (C.c_double * 2)()
abs_in, abs_out = [], []
for t in cases:
    args = [t[k][0] / t[k][1] for k in range(4)] + [float(t[k][1]) for k in range(4)]
    Ra.ref_calculateABSfromAFs(*args, buf)
    abs_in.append([list(x) for x in t]); abs_out.append(bits(buf[:]))
out["abs"] = {"counts_c_n": abs_in, "abs_absalt": abs_out}

S, n = 400, 7
iv = rng.integers(0, 3, (S, n)); haps = rng.integers(0, 4, (S, 2 * n))
missing = (rng.random((S, n)) < 0.15).astype(np.uint8); multi = (rng.random(S) < 0.1).astype(np.uint8)
r = O.indiv_diffs(iv, haps, missing, multi, use_reference=True)
out["indiv_diffs"] = {"n": n, "ind_with_variant": iv.tolist(), "haps": haps.tolist(), "missing": missing.tolist(), "multi": multi.tolist(),
                      "diffs": bits(r["diffs"]), "diffs_me": bits(r["diffs_me"]), "hets_vs_homs": bits(r["hets_vs_homs"]),
                      "miss": r["miss"].ravel().tolist()}
R = O.ref()
assert R is not None
rows_in, rows_out = [], []
quads = [(c1, n1, c2, n2) for n1 in range(0, 4) for c1 in range(n1 + 1) for n2 in range(0, 4) for c2 in range(n2 + 1)]
quads += [(int(rng.integers(0, a + 1)), int(a), int(rng.integers(0, b + 1)), int(b)) for a, b in rng.integers(1, 500, (400, 2))]
for c1, n1, c2, n2 in quads:
    p1 = C.c_double(); p2 = C.c_double()
    R.ref_calculateAlleleFrequencies(c1, n1, c2, n2, C.byref(p1), C.byref(p2))
    vals = [p1.value, p2.value, R.ref_calculateFstNumerator(p1.value, p2.value, n1, n2), R.ref_calculateFstDenominator(p1.value, p2.value),
            R.ref_calculateExpectedHeterozygositySimple(p1.value), R.ref_calculateExpectedHeterozygosityNei78(p1.value, n1)]
    rows_in.append([c1, n1, c2, n2]); rows_out.append(bits(vals))
out["fst_math"] = {"c1_n1_c2_n2": rows_in, "p1_p2_num_den_het_hetNei78": rows_out}
Rb = O.ref_bed()
assert Rb is not None
ip = C.POINTER(C.c_int)
bed_cases = []
for case in range(90):                       # after everything else: the earlier sections keep their random numbers
    n = int(rng.integers(0, 25)); kind = case % 3
    if kind == 0:
        cuts = np.sort(rng.choice(4000, 2 * n, replace=False)) if n else np.zeros(0, np.int64)
        st, en = cuts[0::2], cuts[1::2]
    elif kind == 1:
        st = np.sort(rng.integers(0, 4000, n)); en = st + rng.integers(0, 300, n)
    else:
        st = rng.integers(0, 4000, n); en = st + rng.integers(1, 250, n)
    st = np.ascontiguousarray(st, np.int32); en = np.ascontiguousarray(en, np.int32)
    qs = []
    for _ in range(12):
        a = int(rng.integers(0, 4200)); b = a + int(rng.integers(1, 1200))
        qs.append([a, b, int(Rb.ref_accessible_bp(st.ctypes.data_as(ip), en.ctypes.data_as(ip), n, a, b, 0))])
    bed_cases.append({"starts": st.tolist(), "ends": en.tolist(), "start_end_bp": qs})
out["accessible_bp"] = bed_cases
p = os.path.join(ROOT, "tests", "golden", "ref_consumer_vectors.json")
json.dump(out, open(p, "w"), separators=(",", ":"))
print(p, os.path.getsize(p), "bytes;", len(pbs_in), "PBS,", len(abs_in), "ABS cases")
