/*
 * evo_oracle.h -- CPU restatement of millanek/evo's pairwise Fst / dxy hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under evo_amd/ (the product) may include,
 * link or execute this.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the reported CPU baseline.
 *
 * Parity status ("pinning"):
 *   - The floating-point per-site math (Fst numerator / denominator,
 *     heterozygosities, allele frequency, vector_average*) is PINNED against
 *     the reference's own inline code: oracle/ref_math_shim.cpp compiles
 *     /root/reference/process_vcf_fst.h + process_vcf_utils.h unmodified into
 *     oracle/_ref/libevo_refmath.so and tests/test_oracle.py checks
 *     bit-equality.
 *   - The PBS / ABS per-site math (orc_pbs_from_afs, orc_abs_from_afs) is
 *     PINNED the same way: oracle/ref_pbs_shim.cpp / ref_abs_shim.cpp compile
 *     /root/reference/evo_PBS.cpp / evo_ABS.cpp where they lie, keep only
 *     calculatePBSfromAFs / calculateABSfromAFs (everything else is hidden
 *     and discarded at link time) in oracle/_ref/libevo_refpbs.so /
 *     libevo_refabs.so, and tests/test_oracle.py checks bit-equality, NaN
 *     sign included.  The window text drivers (orc_pbs_run, orc_abs_run)
 *     are restatements: "parity unpinned".
 *   - The genotype counting code (process_vcf_fst.cpp, process_vcf_utils.cpp)
 *     and the two window state machines live in translation units that need
 *     Boost.Math, which this image lacks; they are therefore UNBUILDABLE here
 *     and are restated below from the source text.  They are pinned only by
 *     the known answers SURVEY.md section 8c / BASELINE.md section 2 recorded
 *     from the reference (tests/golden/survey_known_answers.json).  Beyond
 *     those: "parity unpinned".
 *
 * All file:line citations are relative to /root/reference/.
 */
#ifndef EVO_ORACLE_H
#define EVO_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* allele codes used by the packed paths (one code per haplotype) */
enum { ORC_REF = 0, ORC_ALT = 1, ORC_MISSING = 2 };

/* per-site flags */
enum {
    ORC_SITE_SKIP = 1u /* site never reaches the statistics (REF/ALT filters) */
};

/* ---------------- literal, string-level restatements ---------------- */

/* SetCounts (process_vcf_utils.h:249-284), the fields the Fst path reads */
typedef struct {
    int set1Count, set2Count;
    int set1_n_withoutMissing, set2_n_withoutMissing;
    int n_alt_alleles;
    int bIndel;
    int alt_code;          /* 1 or 2: which allele digit counts as ALT (fst.cpp:236-238) */
    int allele_as_missing; /* -1 or 1-based index of a "*" ALT entry (fst.cpp:224-229) */
} orc_setcounts;

/* getVariantCountsForFst (process_vcf_fst.cpp:207-295).
 * genotypes[i] is the i-th sample column (e.g. "0/1:12,3"); set*_loci index it.
 * set*_indiv (len*), set*_hap (2*len*) may be NULL. */
void orc_fst_variant_counts(const char *alt_field, const char *const *genotypes,
                            const size_t *set1_loci, int len1,
                            const size_t *set2_loci, int len2,
                            orc_setcounts *out,
                            int *set1_indiv, int *set2_indiv,
                            int *set1_hap, int *set2_hap);

/* site filter, process_vcf_fst.cpp:543-544 */
int orc_fst_site_included(const orc_setcounts *c);

/* calculateDxy if-chain (process_vcf_fst.cpp:297-323) */
double orc_dxy_literal(const int *set1_indiv, const int *set2_indiv, int n1, int n2);
/* calculatePiTwoSets (process_vcf_fst.cpp:325-351), one set */
double orc_pi_literal(const int *hap, int n, int n_without_missing);

/* closed forms of the two loops above (SURVEY 8a rows 8,9) */
double orc_dxy_closed(int c1, int c2, int n1, int n2);
double orc_pi_closed(int c, int n, int n_without_missing);

/* process_vcf_fst.h:19-45 */
double orc_fst_numerator(double p1, double p2, int n1, int n2);
double orc_fst_denominator(double p1, double p2);
double orc_het_simple(double p);
double orc_het_nei78(double p, int n);

/* vector_average / vector_average_withRegion / vector_sum, process_vcf_utils.h:350-385 */
double orc_vector_sum(const double *v, size_t n);
double orc_vector_average(const double *v, size_t n);
double orc_vector_average_with_region(const double *v, size_t n, int region);
/* calculateFst, process_vcf_fst.cpp:354-360 */
double orc_calculate_fst(const double *num, const double *den, size_t n);

/* getBasicCounts + getSetVariantCountsSimple + getComplementCounts
 * (process_vcf_utils.cpp:50-98).  sample_pop[i] in [0,n_pops) or -1 (sample not
 * in the populations file).  use_pop[k] != 0 when pop k is in populationsToUse.
 * Outputs have n_pops entries. aaf stays -1.0 when a pop has no alleles. */
void orc_general_counts(const char *const *genotypes, int n_samples,
                        const int32_t *sample_pop, int n_pops, const uint8_t *use_pop,
                        int *alt, int *alleles, double *aaf,
                        int *alleles_compl, double *aaf_compl);

/* ---------------- genotype text -> allele codes ---------------- */

/* Parse the ALT column the way `fst` does; returns site flags (ORC_SITE_SKIP when
 * n_alt_alleles != 1 or bIndel or len(REF) > 1) and fills alt_code / allele_as_missing. */
uint32_t orc_fst_site_flags(const char *ref_field, const char *alt_field,
                            int *alt_code, int *allele_as_missing);
/* one allele char -> code, `fst` rules (process_vcf_fst.cpp:244-260) */
uint8_t orc_fst_allele_code(char ch, int alt_code, int allele_as_missing);
/* FstGlobal rules: site skip (evo_FstAgainstAll.cpp:184-188), allele (utils.cpp:67-81) */
uint32_t orc_general_site_flags(const char *ref_field, const char *alt_field);
uint8_t orc_general_allele_code(char ch);

/* ---------------- code-level (packed-path) restatements ---------------- */

/* codes: [n_sites][2*n_samples] (haplotype 2i, 2i+1 belong to sample i).
 * alt / nm: [n_sites][n_pops] (nm = non-missing alleles). */
void orc_counts_from_codes(const uint8_t *codes, int64_t n_sites, int n_samples,
                           const int32_t *sample_pop, int n_pops,
                           int32_t *alt, int32_t *nm);

/* Per-site, per-pair statistics of the two-set path applied to pops (pi,pj).
 * n_full[k] = 2 * (#samples of pop k).  Outputs [n_sites][n_pairs]; any may be NULL.
 * include follows process_vcf_fst.cpp:543-544 and site_flags (may be NULL). */
void orc_pair_site_stats(const int32_t *alt, const int32_t *nm, int64_t n_sites, int n_pops,
                         const int32_t *n_full, const int32_t *pair_i, const int32_t *pair_j,
                         int n_pairs, const uint8_t *site_flags,
                         double *num, double *den, double *dxy, uint8_t *include);

/* Per-site per-pop het/pi of getSetHeterozygozities + calculatePiTwoSets:
 * out [n_sites][n_pops]: het, hetNei (full n), pi (nm). */
void orc_pop_site_div(const int32_t *alt, const int32_t *nm, int64_t n_sites, int n_pops,
                      const int32_t *n_full, double *het, double *het_nei, double *pi);

/* SNP-count windows of process_vcf_fst.cpp:638-650 for one pair (column `pair` of the
 * [n_sites][n_pairs] arrays): emits when counted % step == 0 && counted >= size, summing the
 * last `size` counted sites left to right.  Returns the number of windows; fills up to
 * max_windows entries of each output (any may be NULL):
 *   first_var = counted-size+1, end_site = site index of the emitting site,
 *   sum_num/sum_den/sum_dxy = sequential sums, fst = clamp(avg/avg), mean_dxy. */
int64_t orc_snp_windows(const double *num, const double *den, const double *dxy,
                        const uint8_t *include, int64_t n_sites, int n_pairs, int pair,
                        int size, int step, int64_t max_windows,
                        int64_t *first_var, int64_t *end_site,
                        double *sum_num, double *sum_den, double *sum_dxy,
                        double *fst, double *mean_dxy);

/* FstGlobal per-site evaluation (evo_FstAgainstAll.cpp:198-203, 244-261) from counts.
 * use_idx[u] = pop index of populationsToUse[u]; set_sizes[u] = the value the reference
 * reads as c->setSizes[u] (sizes in ALL-pops sorted order).
 * kept[s] = 0 when the site is skipped or dropped by the pop-0 variability test.
 * num/den: [n_sites][n_use] (0/0 where the 0.8 rule fails; untouched where !kept). */
void orc_global_site_stats(const int32_t *alt, const int32_t *nm, int64_t n_sites, int n_pops,
                           const int32_t *use_idx, int n_use, const int64_t *set_sizes,
                           const uint8_t *site_flags,
                           uint8_t *kept, double *num, double *den);

/* FstGlobal deque windows (evo_FstAgainstAll.cpp:119-122, 237-240, 276-280).
 * Returns number of emitted rows; start_site = index of the site whose coord is
 * coordDeque[0] (-1 while it is still the initial "0"), end_site = emitting site,
 * fst: [rows][n_use]. */
int64_t orc_global_windows(const uint8_t *kept, const uint8_t *site_flags,
                           const double *num, const double *den,
                           int64_t n_sites, int n_use, int size, int step, int64_t max_rows,
                           int64_t *start_site, int64_t *end_site, double *fst);

/* ---------------- text-level drivers (VCF in, evo output files out) ---------------- */

/* `evo fst --vcf --sets [-s] [-w] [--physicalWindowSize] [--regions-above] [-n]`
 * (getFstFromVCF, process_vcf_fst.cpp:374-715) without --annot/--ancSets/--accessibleGenomeBED.
 * Output files are written into out_dir with the reference's names.  Returns 0 on success.
 * summary (may be NULL) gets {Fst, hetS1, hetS2, hetNei1, hetNei2, counted, total}. */
int orc_fst_run(const char *vcf_path, const char *sets_path, const char *samples_path,
                int window_size, int window_step, int physical_window_size, double reg_above,
                const char *run_name, const char *out_dir, double *summary);
/* the same with --accessibleGenomeBED (bed_path may be NULL or ""), and the pieces behind it:
 * BedCoordinateFeatures / AccessibleGenome, process_vcf_annotation_tools.h:340-530 */
int orc_fst_run_bed(const char *vcf_path, const char *sets_path, const char *samples_path,
                    int windowSize, int windowStep, int physicalWindowSize, double regAbove,
                    const char *run_name, const char *out_dir, double *summary, const char *bed_path);
/* the same with --annot and --ancSets as well (paths may be NULL or ""): Annotation::getCategoryOfSNP,
 * process_vcf_annotation_tools.h:177-327 */
int orc_fst_run_opts(const char *vcf_path, const char *sets_path, const char *samples_path,
                     int windowSize, int windowStep, int physicalWindowSize, double regAbove,
                     const char *run_name, const char *out_dir, double *summary, const char *bed_path, const char *annot_path,
                     const char *anc_sets_path);
typedef struct orc_annot orc_annot;
orc_annot *orc_annot_load(const char *path);
void orc_annot_free(orc_annot *a);
const char *orc_annot_category(const orc_annot *a, const char *scaffold, const char *locus);
typedef struct orc_bed orc_bed;
orc_bed *orc_bed_load(const char *path);
void orc_bed_free(orc_bed *b);
int orc_bed_accessible_bp(const orc_bed *b, const char *scaffold, int start, int end);
int orc_accessible_bp(const int *starts, const int *ends, int n, int start, int end);


/* `evo FstGlobal [-w] [-f] [-n] VCF POPS` (FstGlobalMain, evo_FstAgainstAll.cpp:62-308)
 * without --annot. */
int orc_fstglobal_run(const char *vcf_path, const char *pops_path,
                      int window_size, int window_step, int fixed_window_size,
                      const char *run_name, const char *out_dir);
/* with --annot (annot_path may be NULL or ""): one row per gene, evo_FstAgainstAll.cpp:107-116, 215-219, 267-297 */
int orc_fstglobal_run_annot(const char *vcf_path, const char *pops_path, int window_size, int window_step, int fixed_window_size,
                            const char *run_name, const char *out_dir, const char *annot_path);
const char *orc_annot_gene_details(const orc_annot *a, const char *scaffold, int locus, const char **gene);

/* calculatePBSfromAFs (evo_PBS.cpp:69-98) and the per-site rules of PBSmain (:304-322) on counts:
 * pbs [n_trios][3][n_sites], used [n_trios][n_sites].  No reference build or fixture covers these: parity unpinned. */
void orc_pbs_from_afs(double p1, double p2, double p3, double n1, double n2, double n3, double *pbs);
void orc_pbs_site_stats(const int32_t *alt, const int32_t *nm, int64_t n_sites, int n_pops,
                        const int32_t *trios, int n_trios, const uint8_t *site_flags, double *pbs, uint8_t *used);

/* calculateABSfromAFs (evo_ABS.cpp:58-106; abs_out = {ABS, ABSalt}) and the per-site rules of ABSmain (:243-262) on
 * counts: abs [n_quartets][2][n_sites], used [n_quartets][n_sites] */
void orc_abs_from_afs(double pW, double pX, double pY, double pZ, double nW, double nX, double nY, double nZ, double *abs_out);
void orc_abs_site_stats(const int32_t *alt, const int32_t *nm, int64_t n_sites, int n_pops,
                        const int32_t *quartets, int n_quartets, const uint8_t *site_flags, double *abs_out, uint8_t *used);
/* `evo PBS [-w] [-f] [-n] VCF POPS TRIOS` (PBSmain, evo_PBS.cpp:125-363) without --annot / --af / -i */
int orc_pbs_run(const char *vcf_path, const char *pops_path, const char *trios_path,
                int window_size, int window_step, int fixed_window_size, const char *run_name, const char *out_dir);
void orc_set_pbs_af(int on);         /* --af of the PBS driver for the calls that follow: leaves <pops stem>_AF.txt, empty as the reference does */
void orc_set_allow_indels(int on);   /* -i of the PBS and DistOutgroups drivers for the calls that follow (0 = off, the default) */
/* with --annot (annot_path may be NULL or ""): a row per gene and trio, evo_PBS.cpp:176-181, 282-286, 333-361 */
int orc_pbs_run_annot(const char *vcf_path, const char *pops_path, const char *trios_path, int window_size, int window_step,
                      int fixed_window_size, const char *run_name, const char *out_dir, const char *annot_path);
/* `evo ABS [-w] [-n] VCF POPS QUARTETS` (ABSmain, evo_ABS.cpp:109-293) without --annot */
int orc_abs_run(const char *vcf_path, const char *pops_path, const char *quartets_path,
                int windowSize, int windowStep, const char *run_name, const char *out_dir);

/* `evo alleleFreq [-n] VCF POPS` (AFmain, evo_AlleleFeq.cpp:46-166) without -g; the output file
 * <pops base name>_<run_name>_AF.txt is written into out_dir. */
int orc_af_run(const char *vcf_path, const char *pops_path, const char *run_name, const char *out_dir);

/* checker for the kernels' reciprocal-table division; returns the number of mismatches */
long orc_check_rcp_division(int max_n);

/* DistOutgroups (evo_distanceToOutgroups.cpp): DxyPerSNPfromAFs (:62-65); the accumulation of DistOutMain between two
 * window emissions (:274-294) on counts (see evo_oracle.c for the array shapes); and the text driver
 * `evo DistOutgroups [-f] [-n] VCF POPS OUTGROUPS INGROUPS` (:68-306) without -i / --accessibleGenomeBED */
double orc_dxy_from_afs(double AF1, double AF2);
void orc_dist_windows(const int32_t *alt, const int32_t *nm, int n_pops, const uint8_t *site_flags,
                      const int64_t *win_off, int n_windows, const int32_t *out_idx, int n_out, const int32_t *in_idx, int n_in,
                      int carry_first, double *sum, int32_t *nsnp, int32_t *miss, int32_t *used, int32_t *missing_vars);
int orc_distout_run(const char *vcf_path, const char *pops_path, const char *outgroups_path, const char *ingroups_path,
                    int fixedWindowSize, const char *run_name, const char *out_dir);
int orc_distout_run_bed(const char *vcf_path, const char *pops_path, const char *outgroups_path, const char *ingroups_path,
                        int fixedWindowSize, const char *run_name, const char *out_dir, const char *bed_path);

/* stats --diff-matrix: getThisVariantCountsSimple (process_vcf_utils.cpp:358-408), the two per-site matrix updates
 * (process_vcf_stats_functions.cpp:180-281) literally, the same as five counts per pair (what the device returns), and
 * the text driver for `evo stats --diff-matrix [--block-bootstrap=B] [--numAccessibleBP=A] VCF`
 * (process_vcf_stats.cpp:98-355).  Array shapes: see evo_oracle.c. */
void orc_stats_variant_counts(const char *alt_field, const char *const *genotypes, int n,
                              int *ind_with_variant, int *haps, uint8_t *missing, int *is_multiallelic);
void orc_indiv_diffs(int n, int64_t S, const int *ind_with_variant, const int *haps, const uint8_t *missing, const uint8_t *multi,
                     double *diffs, double *diffs_me, double *diffs_me_bootstrap, double *hets_vs_homs,
                     int *pairwise_missingness, int *pairwise_missingness_bootstrap);
void orc_indiv_diff_counts(int n, int64_t S, const int *ind_with_variant, const int *haps, const uint8_t *missing, const uint8_t *multi,
                           int32_t *hh, int32_t *hd, int32_t *oh, int32_t *ms_out, int32_t *d4);
int orc_stats_diffs_run(const char *vcf_path, int bootstrapBlockSize, int numAccessibleBP, const char *out_dir);
/* het_analysis at the sites without a missing genotype (stats_functions.cpp:19-28, process_vcf_stats.cpp:239-242); ADDS */
void orc_indiv_hets(int n, int64_t S, const int *ind_with_variant, const uint8_t *missing, int32_t *het, int32_t *shared);
/* stats driver with --diff-matrix and / or --hets-per-individual (<VCF base>.hets.txt, <VCF base>.sharedHets.txt) */
int orc_stats_run(const char *vcf_path, int doDiffs, int doHets, int bootstrapBlockSize, int numAccessibleBP, const char *out_dir);

#ifdef __cplusplus
}
#endif
#endif
