// ref_math_shim.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Thin extern "C" wrappers around the REFERENCE's own inline functions and templates,
// compiled from the headers where they lie (-I/root/reference): process_vcf_fst.h:19-45,
// process_vcf_utils.h:272-275 and :350-385.  No reference source is copied and no stand-in
// header is used: these two headers need only the C++ standard library and zlib.h.
// (The .cpp files of the path include Boost.Math, absent from this image, and are therefore
// unbuildable here -- see oracle/evo_oracle.h.)
//
// Output: oracle/_ref/libevo_refmath.so (git-ignored, travels to the GPU box).
#include "process_vcf_fst.h"

extern "C" {

double ref_calculateFstNumerator(double p1, double p2, int n1, int n2) {
    return calculateFstNumerator(p1, p2, n1, n2);
}
double ref_calculateFstDenominator(double p1, double p2) { return calculateFstDenominator(p1, p2); }
double ref_calculateExpectedHeterozygositySimple(double p) { return calculateExpectedHeterozygositySimple(p); }
double ref_calculateExpectedHeterozygosityNei78(double p, int n) { return calculateExpectedHeterozygosityNei78(p, n); }

double ref_vector_sum(const double* v, size_t n) { std::vector<double> x(v, v + n); return vector_sum(x); }
double ref_vector_average(const double* v, size_t n) { std::vector<double> x(v, v + n); return vector_average(x); }
double ref_vector_average_withRegion(const double* v, size_t n, int region) {
    std::vector<double> x(v, v + n); return vector_average_withRegion(x, region);
}

// SetCounts::calculateAlleleFrequencies (process_vcf_utils.h:272-275)
void ref_calculateAlleleFrequencies(int c1, int nm1, int c2, int nm2, double* p1, double* p2) {
    SetCounts s; s.set1Count = c1; s.set2Count = c2;
    s.set1_n_withoutMissing = nm1; s.set2_n_withoutMissing = nm2;
    s.calculateAlleleFrequencies(); *p1 = s.p1; *p2 = s.p2;
}

// GeneralSetCountsWithComplements constructor (process_vcf_utils.h:71-82, 115-119):
// setSizes order and the -1.0 initial AAFs.  names[k] has sizes[k] members at dummy positions.
int ref_general_ctor(const char* const* names, const int* sizes, int n_pops,
                     long* set_sizes_out, double* aaf_init_first, double* aaf_compl_init_first) {
    std::map<string, std::vector<size_t> > m;
    for (int k = 0; k < n_pops; k++) m[names[k]] = std::vector<size_t>((size_t)sizes[k], 0);
    GeneralSetCountsWithComplements c(m, 4);
    for (size_t i = 0; i < c.setSizes.size(); i++) set_sizes_out[i] = (long)c.setSizes[i];
    *aaf_init_first = c.setAAFs.begin()->second;
    *aaf_compl_init_first = c.setAAFsComplement.begin()->second;
    return (int)c.setSizes.size();
}

const char* ref_build_info() { return "g++ -std=c++11 -O2, headers from /root/reference (process_vcf_fst.h, process_vcf_utils.h)"; }
}
