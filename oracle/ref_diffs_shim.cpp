// ref_diffs_shim.cpp -- TEST INFRASTRUCTURE ONLY.
//
// extern "C" wrappers around the REFERENCE's diffs_between_individuals and
// diffs_between_individuals_with_multialleleics (process_vcf_stats_functions.cpp:180-281) and its het_analysis
// (:19-28), built like the other
// shims: that translation unit needs only the standard library, so it is compiled from where it lies under
// /root/reference by including it here; everything but the wrapper is hidden and discarded at link time.
// No reference source is copied and no stand-in header is used.
//
// Output: oracle/_ref/libevo_refdiffs.so (git-ignored, travels to the GPU box).
#include "process_vcf_stats_functions.cpp"

// Runs the reference's per-site update over S sites.  ind_with_variant [S][n], haps [S][2n], missing [S][n],
// multi [S]; the n x n row-major matrices are added to.
extern "C" __attribute__((visibility("default")))
void ref_indiv_diffs(int n, long S, const int* ind_with_variant, const int* haps, const unsigned char* missing,
                     const unsigned char* multi, double* diffs, double* diffs_me, double* diffs_me_bootstrap,
                     double* hets_vs_homs, int* pairwise_missingness, int* pairwise_missingness_bootstrap) {
    std::vector<std::vector<double> > D(n, std::vector<double>(n)), Dme(n, std::vector<double>(n)), Dboot(n, std::vector<double>(n)),
        HvH(n, std::vector<double>(n));
    std::vector<std::vector<int> > M(n, std::vector<int>(n)), Mboot(n, std::vector<int>(n));
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            D[i][j] = diffs[i * n + j]; Dme[i][j] = diffs_me[i * n + j]; Dboot[i][j] = diffs_me_bootstrap[i * n + j];
            HvH[i][j] = hets_vs_homs[i * n + j]; M[i][j] = pairwise_missingness[i * n + j]; Mboot[i][j] = pairwise_missingness_bootstrap[i * n + j];
        }
    FilterResult result;
    for (long s = 0; s < S; s++) {
        result.counts.individualsWithVariant.assign(ind_with_variant + s * n, ind_with_variant + (s + 1) * n);
        result.counts.haplotypesWithVariant.assign(haps + s * 2 * n, haps + (s + 1) * 2 * n);
        result.counts.missingGenotypesPerIndividual.assign(n, false);
        for (int i = 0; i < n; i++) result.counts.missingGenotypesPerIndividual[i] = missing[s * n + i] != 0;
        if (!multi[s]) diffs_between_individuals(D, Dme, Dboot, HvH, M, Mboot, result);                 // process_vcf_stats.cpp:258-262
        else diffs_between_individuals_with_multialleleics(Dme, M, Dboot, Mboot, result);
    }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            diffs[i * n + j] = D[i][j]; diffs_me[i * n + j] = Dme[i][j]; diffs_me_bootstrap[i * n + j] = Dboot[i][j];
            hets_vs_homs[i * n + j] = HvH[i][j]; pairwise_missingness[i * n + j] = M[i][j]; pairwise_missingness_bootstrap[i * n + j] = Mboot[i][j];
        }
}

// The reference's het_analysis (process_vcf_stats_functions.cpp:19-28) under the caller's condition of
// process_vcf_stats.cpp:239-242 (only when no genotype of the line is missing).  counts.overall is what
// getThisVariantCountsSimple leaves there: one per '1' character, i.e. the sum of individualsWithVariant
// (process_vcf_utils.cpp:397-406).  het [n] and shared [n] are added to.
extern "C" __attribute__((visibility("default")))
void ref_het_analysis(int n, long S, const int* ind_with_variant, const unsigned char* missing, int* het, int* shared) {
    std::vector<int> hetCounts(het, het + n), sharedHetCounts(shared, shared + n);
    for (long s = 0; s < S; s++) {
        FilterResult result;
        result.counts.individualsWithVariant.assign(ind_with_variant + s * n, ind_with_variant + (s + 1) * n);
        result.counts.overall = 0;
        result.counts.bAnyMissingGenotypes = false;
        for (int i = 0; i < n; i++) {
            result.counts.overall += ind_with_variant[s * n + i];
            if (missing[s * n + i]) result.counts.bAnyMissingGenotypes = true;
        }
        if (result.counts.bAnyMissingGenotypes == false) het_analysis(hetCounts, sharedHetCounts, result);
    }
    for (int i = 0; i < n; i++) { het[i] = hetCounts[i]; shared[i] = sharedHetCounts[i]; }
}
