/*
 * evofst.h -- C ABI of the MI355X-native engine for evo's pairwise Fst / dxy hot path.
 *
 * The reference (millanek/evo) has no plugin / FFI seam: its only entry points for this path
 * are `int fstMain(int, char**)` (process_vcf_fst.h:16, dispatched at process_vcf.cpp:171-172)
 * and `int FstGlobalMain(int, char**)` (evo_FstAgainstAll.h:16, process_vcf.cpp:217-218).
 * Each function below names the part of those two programs it replaces (file:line relative to
 * the reference tree).  A maintainer's binding is shown in INTEGRATION.md.
 *
 * Conventions: plain pointers and sizes only; every call returns EVOFST_OK (0) or a negative
 * error code and never exits or throws; evofst_last_error() gives the message.  The caller
 * owns all host buffers; the library owns device memory.  One context = one GPU = one HIP
 * stream; calls on one context must come from one thread at a time.  Compute calls are
 * asynchronous on the context's stream; evofst_sync() and every evofst_get_* call wait.
 *
 * Packed genotype format ("tiles"): haplotypes are permuted so that each population is a
 * contiguous bit range (population k = bits [off_k, off_k + 2*size_k)); a site is W =
 * ceil(H/64) pairs of 64-bit words {ALT bits, MISSING bits}; sites are grouped in tiles of 64
 * and a tile is stored word-major: tiles[t][w][s] = {alt, missing} (16 bytes) for site 64*t+s.
 * Two bits per haplotype => N/2 bytes per site (+ padding of H to 64 and of S to 64).
 */
#ifndef EVOFST_H
#define EVOFST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EVOFST_OK 0
#define EVOFST_ERR_ARG (-1)         /* bad argument */
#define EVOFST_ERR_HIP (-2)         /* a HIP runtime call failed */
#define EVOFST_ERR_STATE (-3)       /* call made in the wrong order */
#define EVOFST_ERR_NOMEM (-4)       /* host or device allocation failed */
#define EVOFST_ERR_UNSUPPORTED (-5) /* outside the limits of this build */

#define EVOFST_TILE_SITES 64
#define EVOFST_MAX_POP_CHROMS 16383 /* 2 * samples of one population */
#define EVOFST_MAX_POPS 4096

/* allele codes accepted by the packer: one per haplotype */
#define EVOFST_REF 0
#define EVOFST_ALT 1
#define EVOFST_MISSING 2

/* site flags */
#define EVOFST_SITE_SKIP 1u /* site fails the REF/ALT filters of the mode in use */
#define EVOFST_SITE_MULTI 2u /* EVOFST_MODE_STATS: multi-allelic site (always together with SKIP: its genotypes travel as
                              * allele numbers through evofst_set_multiallelic, not in the tiles) */

/* genotype-text interpretation, evofst_codes_from_vcf_line() */
#define EVOFST_MODE_FST 0    /* `evo fst` rules   (process_vcf_fst.cpp:222-280, 529-532) */
#define EVOFST_MODE_GLOBAL 1 /* `evo FstGlobal`   (process_vcf_utils.cpp:62-83, evo_FstAgainstAll.cpp:184-188) */
#define EVOFST_MODE_STATS 2  /* `evo stats`       (getThisVariantCountsSimple, process_vcf_utils.cpp:358-408): an individual
                              * with a '.' or the missing allele in either position is missing as a whole (both haplotypes
                              * EVOFST_MISSING); ALT = the character '1'; no REF/ALT length filter */
#define EVOFST_MODE_GLOBAL_INDELS 3 /* the rules of EVOFST_MODE_GLOBAL with -i / --allow-indels-and-multiallelics of `evo PBS` and
                                     `evo DistOutgroups` (evo_PBS.cpp:248-256): only ALT == "*" is skipped; packer modes only */

typedef struct evofst_ctx evofst_ctx;

typedef struct {
    int32_t n_samples;      /* samples in the input order */
    int32_t n_pops;
    int32_t n_haplotypes;   /* H: 2 * samples that belong to a population */
    int32_t words_per_site; /* W */
    int32_t tile_sites;     /* 64 */
    int32_t count_bytes;    /* 2 or 4: bytes of one packed (alt, non-missing) count */
    int64_t tile_bytes;     /* W * 64 * 16 */
} evofst_geometry;

/* ------------------------------------------------------------------ lifecycle */

/* Creates a context on HIP device `device_id` (its own stream).  NULL on failure;
 * evofst_last_error(NULL) then tells why.  device_id == -1 creates a HOST-ONLY context for the
 * packer functions (set_layout / pack / unpack / geometry); every device call on it fails with
 * EVOFST_ERR_STATE -- there is no CPU compute path. */
evofst_ctx *evofst_create(int device_id);
/* GPUs this process sees (0: none, or the runtime failed).  (No counterpart in the reference.) */
int evofst_device_count(void);
void evofst_destroy(evofst_ctx *ctx);
const char *evofst_last_error(const evofst_ctx *ctx);
const char *evofst_version(void);
/* Use an existing HIP stream (hipStream_t passed as void*) instead of the context's own. */
int evofst_set_stream(evofst_ctx *ctx, void *hip_stream);

/* ------------------------------------------------------------------ layout
 * Replaces locateSet + the per-line set loops (process_vcf_utils.cpp:923-936,
 * process_vcf_fst.cpp:489-491, evo_FstAgainstAll.cpp:153-168): sample_pop[i] is the population
 * of sample column i, or -1 when the sample is in no population. */
int evofst_set_layout(evofst_ctx *ctx, int n_samples, int n_pops, const int32_t *sample_pop);
int evofst_get_geometry(const evofst_ctx *ctx, evofst_geometry *out);
/* n_full[k] = 2 * size of population k (the `n1`,`n2` of process_vcf_fst.cpp:491) */
int evofst_get_pop_chroms(const evofst_ctx *ctx, int32_t *n_full);

/* ------------------------------------------------------------------ host packer (stage K0)
 * Replaces split() + the genotype char tests (process_vcf_utils.cpp:104-130,
 * process_vcf_fst.cpp:243-280, process_vcf_utils.cpp:62-83). */

/* Bytes of the tile buffer that holds n_sites sites. */
size_t evofst_tiles_bytes(const evofst_ctx *ctx, int64_t n_sites);
/* codes: [n_sites][2*n_samples] allele codes in the INPUT sample order -> tiles.
 * first_site lets a caller fill one tile buffer in several calls (must be a multiple of 64
 * except for the last call). */
int evofst_pack_codes(const evofst_ctx *ctx, int64_t first_site, int64_t n_sites,
                      const uint8_t *codes, void *tiles);
/* Inverse of evofst_pack_codes (used by tests and by callers that want to audit a buffer). */
int evofst_unpack_codes(const evofst_ctx *ctx, int64_t first_site, int64_t n_sites,
                        const void *tiles, uint8_t *codes);
/* One VCF data line -> allele codes [2*n_samples] + site flags, with the rules of `mode`.
 * `line` is modified in place (tabs become NULs).  chrom/pos point into `line`.
 * Returns EVOFST_ERR_ARG for lines with fewer than 9+n_samples columns or genotype fields
 * shorter than 3 characters (the reference reads chars [0] and [2] unchecked). */
int evofst_codes_from_vcf_line(char *line, int mode, int n_samples, uint8_t *codes,
                               uint32_t *site_flags, const char **chrom, const char **pos);
/* The same for a block of lines, parsed and packed straight into tiles on n_threads host threads (the reference's
 * per-line split() + char tests, process_vcf_utils.cpp:104-130 and process_vcf_fst.cpp:243-280, are its largest
 * cost: SURVEY 8a row 1).  lines[i] are NUL-terminated data lines and are not modified; layout_to_vcf[k] = VCF
 * genotype column (0-based) of sample k of evofst_set_layout (NULL: identity); tiles as evofst_pack_codes writes
 * them for sites 0 .. n_lines-1 (evofst_tiles_bytes(ctx, n_lines) bytes), site_flags[n_lines].  Returns
 * EVOFST_ERR_ARG if any line is malformed. */
int evofst_pack_vcf_lines(const evofst_ctx *ctx, int mode, int64_t n_lines, const char *const *lines,
                          int n_vcf_samples, const int32_t *layout_to_vcf, int n_threads, void *tiles,
                          uint8_t *site_flags);

/* ------------------------------------------------------------------ device data */

/* (Re)allocates device buffers for up to max_sites sites in the current layout/mode. */
int evofst_reserve(evofst_ctx *ctx, int64_t max_sites);
/* Pinned host staging memory for evofst_upload (hipHostMalloc / hipHostFree). */
void *evofst_alloc_pinned(evofst_ctx *ctx, size_t bytes);
void evofst_free_pinned(evofst_ctx *ctx, void *p);
/* Asynchronous H2D of n_sites packed sites.  site_flags may be NULL (no site skipped);
 * pos may be NULL.  Buffers must stay valid until evofst_sync(). */
int evofst_upload(evofst_ctx *ctx, int64_t n_sites, const void *tiles,
                  const uint8_t *site_flags, const int32_t *pos);
/* Use tiles that already live in HBM (device pointers; not copied, caller keeps them alive). */
int evofst_attach_device(evofst_ctx *ctx, int64_t n_sites, const void *d_tiles,
                         const uint8_t *d_site_flags, const int32_t *d_pos);
/* Fills the device tile buffer with the synthetic workload of SURVEY.md 8(d) (benchmarks and
 * full-size property tests; no reference counterpart).  first_site_global keys the generator so
 * that a shard regenerates exactly its slice.  Also fills pos (cumulative steps of 1..199). */
int evofst_synth(evofst_ctx *ctx, int64_t n_sites, int64_t first_site_global, uint64_t seed,
                 double missing_rate);
/* D2H copy of the tile buffer (tests). */
int evofst_download_tiles(evofst_ctx *ctx, void *tiles);
/* The same for tiles [first_tile, first_tile + n_tiles) only (n_tiles * tile_bytes bytes). */
int evofst_download_tile_range(evofst_ctx *ctx, int64_t first_tile, int64_t n_tiles, void *tiles);

/* ------------------------------------------------------------------ modes */

/* Two-set statistics for population pairs: the path of getFstFromVCF
 * (process_vcf_fst.cpp:535-557) applied to (pair_i[q], pair_j[q]).  n_pairs == 0 selects all
 * P*(P-1)/2 pairs i<j in lexicographic order. */
int evofst_set_pairs(evofst_ctx *ctx, const int32_t *pair_i, const int32_t *pair_j, int n_pairs);
/* Fst-against-all (evo_FstAgainstAll.cpp:191-203, 244-261): use_pops = populationsToUse as
 * population indices, set_sizes[u] = the value the reference reads as c->setSizes[u]. */
int evofst_set_against_all(evofst_ctx *ctx, const int32_t *use_pops, int n_use,
                           const int64_t *set_sizes);
/* Gram path (all pairs i<j at once on the FP64 matrix cores) for windows whose membership does not
 * depend on the pair: the physical windows of process_vcf_fst.cpp:589-629 (a site belongs to a bp window
 * whatever the pair; the per-pair filter of :543-544 only zeroes its terms) and fixed site blocks.  Meant for
 * many populations (hundreds), where a lane per pair no longer fits; results per window and pair are the same
 * sums K2 gives per segment.  Use with evofst_set_site_windows + evofst_run_counts + evofst_run_gram. */
int evofst_set_gram(evofst_ctx *ctx);
/* Windows as site ranges [win_start[w], win_start[w+1]), w < n_windows: win_start has n_windows + 1
 * non-decreasing entries in [0, n_sites].  (bp windows: the caller cuts by position, as the window loop of
 * process_vcf_fst.cpp:589-629 does; empty windows are allowed.) */
int evofst_set_site_windows(evofst_ctx *ctx, const int64_t *win_start, int64_t n_windows);
/* SNP-count windows: -w SIZE,STEP of both programs (process_vcf_fst.cpp:638-650,
 * evo_FstAgainstAll.cpp:238, 276-280).  size == 0 disables windows (totals only). */
int evofst_set_windows(evofst_ctx *ctx, int snp_size, int snp_step);
/* Multi-GPU: number of included sites per pair (pairs mode) that precede this shard
 * (n_pairs values), or {used, kept} sites preceding it (against-all mode, 2 values). */
int evofst_set_base_offsets(evofst_ctx *ctx, const int64_t *offsets);

/* Device-side forms of the same exchange, for callers that keep the whole pass on one stream (set the stream
 * with evofst_set_stream; nothing here waits for the GPU): the offsets are read from / the totals and carries
 * are written to device memory of the caller (e.g. the tensors of an RCCL all-gather) by asynchronous copies on
 * the context's stream.  Sizes as in the host forms: n_pairs (pairs mode) or 2 (against-all) int64 values;
 * evofst_copy_boundary writes what evofst_get_boundary returns, n_pairs x 4 doubles. */
int evofst_set_base_offsets_device(evofst_ctx *ctx, const void *d_offsets);
/* The same from the result of the all-gather itself: d_gathered holds one row of row_stride int64 values per rank,
 * each starting with that rank's included totals; the base offsets of rank r are the sums of rows 0 .. n_rows-1 with
 * n_rows = r (the exclusive prefix the reference's running countedVariantNumber amounts to, process_vcf_fst.cpp:544-545;
 * evo_FstAgainstAll.cpp:188,204).  The sums are formed by the offset scan of the next evofst_run_stats, on the
 * context's stream: the buffer has to keep its contents until that call's kernels have run (stream order is enough). */
int evofst_set_base_offsets_gathered(evofst_ctx *ctx, const void *d_gathered, int64_t row_stride, int n_rows);
int evofst_copy_included_totals(evofst_ctx *ctx, void *d_dst);
int evofst_copy_boundary(evofst_ctx *ctx, void *d_dst);

/* Multi-GPU, against-all mode: the last n (<= SIZE) kept sites that precede this shard, oldest first --
 * what the reference's numDeque / denomDeque (evo_FstAgainstAll.cpp:119-122, 237-240) hold when the shard's
 * first site is read: num / den [n][n_use] as evofst_get_global_site_stats returns them on the earlier shards,
 * tag[n] = the value to report as a window's start site while its oldest entry is one of them (a code of the
 * caller's below -1; -1 itself is the start of a window whose oldest entry is the zero-initialised deque).  n == 0 clears.  Call after evofst_set_windows, before evofst_run_stats. */
int evofst_set_global_halo(evofst_ctx *ctx, int n, const int32_t *tag, const double *num, const double *den);

/* ------------------------------------------------------------------ compute */

/* Gram path: window sums of windows [first_window, first_window + n_windows) for all P (P-1) / 2 pairs
 * (device buffers of n_windows * n_pairs * 32 bytes are kept until the next call). */
int evofst_run_gram(evofst_ctx *ctx, int64_t first_window, int64_t n_windows);
/* Results of the last evofst_run_gram, each [n_windows][n_pairs] (pairs i<j lexicographic): sums of the Fst
 * numerators and denominators (process_vcf_fst.h:19-30) and of sumKij (process_vcf_fst.cpp:299-321) over the
 * window's included sites, and the number of included sites (process_vcf_fst.cpp:543-544).  NaN where the
 * reference's sum is NaN.  Any pointer may be NULL. */
int evofst_get_gram(evofst_ctx *ctx, double *sum_num, double *sum_den, double *sum_dxy_k, double *n_included);

/* Stage K1: per-population allele counts (getVariantCountsForFst counting loops,
 * process_vcf_fst.cpp:243-280 / getBasicCounts, process_vcf_utils.cpp:62-83) + the per-chunk
 * census of included sites (filter process_vcf_fst.cpp:543-544 / evo_FstAgainstAll.cpp:198-203). */
int evofst_run_counts(evofst_ctx *ctx);
/* Stages K2+K3: per-pair numerator / denominator / dxy and segmented window sums
 * (process_vcf_fst.h:19-30, process_vcf_fst.cpp:297-323, 354-360, 638-650), or the
 * against-all evaluation and deque windows (evo_FstAgainstAll.cpp:244-280). */
int evofst_run_stats(evofst_ctx *ctx);
/* evofst_run_counts + evofst_run_stats. */
int evofst_run(evofst_ctx *ctx);
int evofst_sync(evofst_ctx *ctx);

/* evofst_run for callers that cannot allocate device memory -- tight serving loops, harnesses that forbid hipMalloc, code
 * that runs between the launches of a captured graph and must not disturb the allocator.  The call itself is issued from the
 * host every time and is NOT meant to be recorded into a HIP graph: each call hands the kernels a new launch epoch (a host
 * counter per GPU that tells this call's look-back granules from the last call's), which a replayed graph would not renew.
 * It has NO counterpart in the reference (which has no device at all); it is a form of evofst_run in
 * pairs mode that calls no hipMalloc / hipFree: the one-kernel pairs pass (evo_amd/csrc/evofst_fused.hip) keeps its look-back
 * granules, ticket / error words and tail table in a static device array of evofst_noalloc_scratch_bytes() = 16 MiB that every
 * GPU the library runs on holds until the process ends.  Shapes whose scratch would be larger (about n_sites / 16 x n_pairs
 * x 42 bytes: 1 M sites x 45 pairs take 8 MiB) or with more than 256 pairs come back as EVOFST_ERR_UNSUPPORTED.
 *
 * Before: evofst_set_layout, evofst_set_pairs, evofst_set_windows (and evofst_set_base_offsets for a later shard); those
 * allocate, this call does not.  The caller passes the input (device pointers to the packed tiles of n_sites sites and,
 * or NULL, their site flags) and the outputs, all device memory of its own:
 *   d_seg_rows       [seg_rows_capacity][n_pairs] rows of 32 bytes {double sum_num, sum_den, sum_dxy_k; int32 end_site, pad},
 *                    seg_rows_capacity >= n_sites / gcd(SIZE, STEP) + 2; the rows evofst_get_segments hands out as four arrays,
 *   d_first_segment  [n_pairs] index of the first segment that ends in this shard,
 *   d_incl_totals    [n_pairs] included sites of the shard (n_segments = (base + total) / g - base / g),
 *   d_carry          [n_pairs][4] {sum_num, sum_den, sum_dxy_k, n_sites} behind the last complete segment (evofst_get_boundary).
 * Asynchronous on the context's stream.  All calls on one GPU share that GPU's scratch array and its epoch counter (process
 * globals, the one exception to "no global state" behind this ABI): host threads may call concurrently (the library takes a
 * lock per GPU while it carves the scratch and launches), but the GPU work of two calls on one GPU must not overlap (two
 * contexts or streams at once race on the array): one at a time per GPU.  The pass sizes its grid for a device it has to
 * itself; when other streams hold compute units a look-back wait can run out of polls, and then evofst_sync (or the next
 * getter) returns EVOFST_ERR_HIP and the outputs must be discarded -- never silently wrong rows. */
int evofst_run_pairs_noalloc(evofst_ctx *ctx, int64_t n_sites, const void *d_tiles, const uint8_t *d_site_flags,
                             void *d_seg_rows, int64_t seg_rows_capacity, int64_t *d_first_segment, int64_t *d_incl_totals,
                             double *d_carry);
int64_t evofst_noalloc_scratch_bytes(void);

/* Tuning and test switches of one context (no counterpart in the reference; nothing here changes a result beyond the
 * summation order inside a window).  The library reads no environment variable for them.
 *   "pairs_pass"    how evofst_run takes pairs mode: 0 = the library chooses (default), 1 = counts + scan + walk as separate
 *                   kernels (what evofst_run_counts + evofst_run_stats do), 2 = the stream pass (one launch of producer /
 *                   consumer workgroups, evo_amd/csrc/evofst_stream.hip; shapes it does not take fall back to 1),
 *                   3 = the lock-step one-kernel pass of evofst_fused.hip (the pass behind evofst_run_pairs_noalloc)
 *   "chunk_tiles"   tiles of 64 sites per chunk, 0 = chosen by the library
 *   "walk"          statistics walk of pairs_pass 1: 0 = chosen by the library, 1 = general walk on every chunk,
 *                   2 = k2_window_kernel
 *   "stream_blocks" workgroups of the stream pass, 0 = as many as the device holds (any number is safe)
 *   "stream_producers", "stream_consumers"  its producer / consumer waves per workgroup (consumers: per set),
 *                   0 = chosen by the library
 *   "stream_buffers" chunks of a workgroup in flight (LDS buffers), 2 .. 4, 0 = chosen by the library
 *   "stream_sets"   sets of consumer waves that take a workgroup's chunks in turn, 1 or 2, 0 = chosen by the library
 *   "gram_denominator"  evofst_run_gram's window denominators: 0 = the library chooses (one FP64 contraction of the
 *                   centred frequencies where no population has more than 63 chromosomes, in pieces of at most
 *                   2^13 / most chromosomes of a population sites -- longer windows run as several pieces whose sums
 *                   evofst_get_gram adds -- which keeps every window inside 1e-12 of its own terms; else the two of
 *                   p q^T + q p^T), 2 = always the two
 * Unknown names and values out of range return EVOFST_ERR_ARG.  A shape an option does not fit (LDS, waves) takes
 * the next pass in the order stream, lock-step, separate kernels. */
int evofst_set_option(evofst_ctx *ctx, const char *name, int64_t value);
/* Which form of the pairs pass the last evofst_run took: 1 = separate kernels, 2 = stream pass, 3 = lock-step
 * one-kernel pass; 0 = none yet, or not pairs mode.  (No counterpart in the reference.) */
int evofst_last_pairs_pass(const evofst_ctx *ctx);
/* Which kernel made the window rows of the last against-all evofst_run_stats: 1 = lane = population with a deque per lane
 * (global_window_kernel), 2 = the two-phase kernel (global_tile_kernel: lane = site evaluation, thread = window x
 * population rows; taken where its LDS ring of SIZE + 64 kept sites per population fits); 0 = none yet.  The option
 * "global_walk" of evofst_set_option chooses: 0 = the library (default), 1, 2 (where it fits). */
int evofst_last_global_walk(const evofst_ctx *ctx);
/* How many FP64 contractions the last evofst_run_gram spent on the window denominators: 1 or 2 (see "gram_denominator");
 * 0 = none yet.  (No counterpart in the reference.) */
int evofst_last_gram_streams(const evofst_ctx *ctx);

/* When enabled, evofst_run_* bracket every kernel with HIP events on the context's stream. */
int evofst_set_profiling(evofst_ctx *ctx, int enabled);
/* Milliseconds of the last run of kernel `which` (0 = counts, 1 = offsets scan, 2 = pair/global
 * stats or the Gram contraction, 3 = stitch, or the per-population window sums of the Gram path).  Waits for the stream. */
int evofst_get_kernel_ms(evofst_ctx *ctx, int which, float *ms);

/* ------------------------------------------------------------------ results */

/* Bit-exact parity hook: alt[s*n_pops+k] = alt-allele count, nm[...] = non-missing alleles
 * (set{1,2}Count / set{1,2}_n_withoutMissing, process_vcf_utils.h:262-265). */
int evofst_get_counts(evofst_ctx *ctx, int64_t first_site, int64_t n_sites, int32_t *alt,
                      int32_t *nm);
/* Included sites of this shard per pair (n_pairs values), available after evofst_run_counts.
 * Against-all mode: 2 values {used, kept}. */
int evofst_get_included_totals(evofst_ctx *ctx, int64_t *totals);
/* Per-site statistics for pairs mode, computed on demand for sites [first_site, +n_sites):
 * arrays are [n_pairs][n_sites]; any pointer may be NULL.  num / den / dxy are the values of
 * calculateFstNumerator / calculateFstDenominator / calculateDxy; include is the site filter. */
int evofst_get_site_stats(evofst_ctx *ctx, int64_t first_site, int64_t n_sites, double *num,
                          double *den, double *dxy, uint8_t *include);
/* Population branch statistic per site for n_trios population trios (trios[3*t + 0..2] = population indices):
 * calculatePBSfromAFs (evo_PBS.cpp:69-98) on the allele frequencies of the general counting rules
 * (EVOFST_MODE_GLOBAL codes), pbs[t][0..2][site]; used[t][site] = 1 where PBSmain counts the site for the trio
 * (evo_PBS.cpp:304-313: every population has a called allele, not all three frequencies 0 or all 1, site not
 * skipped), else 0 and the three values are 0.  Any mode, after evofst_run_counts. */
int evofst_get_pbs_site_stats(evofst_ctx *ctx, int64_t first_site, int64_t n_sites, int n_trios, const int32_t *trios,
                              double *pbs, uint8_t *used);
/* Ancestral branch statistic per site for n_quartets population quartets W, X, Y, Z (quartets[4*t + 0..3] = population
 * indices): calculateABSfromAFs (evo_ABS.cpp:58-106, including the swapped variance terms of :66) on the same allele
 * frequencies, abs_out[t][0][site] = ABS, abs_out[t][1][site] = ABSalt; used[t][site] = 1 where ABSmain counts the site
 * for the quartet (evo_ABS.cpp:243-256), else 0 and both values are 0.  Any mode, after evofst_run_counts. */
int evofst_get_abs_site_stats(evofst_ctx *ctx, int64_t first_site, int64_t n_sites, int n_quartets, const int32_t *quartets,
                              double *abs_out, uint8_t *used);
/* Outgroup-to-ingroup dxy in caller-cut windows: what DistOutMain (evo_distanceToOutgroups.cpp:274-294, DxyPerSNPfromAFs
 * :62-65) accumulates between two emissions of its fixed-window state machine.  Window w covers sites
 * [win_off[w], win_off[w+1]) of the uploaded shard (n_windows + 1 non-decreasing offsets); skipped sites do not count.
 * Per window and outgroup i (arrays [n_windows][n_out]): used = sites where the outgroup has a called allele,
 * missing = sites where it has none.  Per window, outgroup and ingroup j (arrays [n_windows][n_out][n_in]):
 * sum = the per-site dxy AFout(1-AFin) + AFin(1-AFout) added up in site order (bit-identical to the reference's
 * sequential sum), nsnp = the sites in that sum, miss = used sites where the ingroup has no called allele.
 * carry_first != 0: on entry the entries of window 0 hold the running values of a window begun in an earlier shard,
 * and are continued.  Any mode, after evofst_run_counts. */
int evofst_get_dist_windows(evofst_ctx *ctx, int n_windows, const int64_t *win_off, int n_out, const int32_t *out_idx,
                            int n_in, const int32_t *in_idx, int carry_first, double *sum, int32_t *nsnp, int32_t *miss,
                            int32_t *used, int32_t *missing);
/* Per-individual pairwise differences: diffs_between_individuals and diffs_between_individuals_with_multialleleics
 * (process_vcf_stats_functions.cpp:180-281), as driven by statsMain (process_vcf_stats.cpp:257-263), as five counts
 * per pair of individuals j <= i and per caller-cut block of sites (the reference's bootstrap blocks,
 * process_vcf_stats.cpp:265-275; one block = the whole shard gives the totals).  Individuals are those of the layout, in
 * layout order (set_layout with one population keeps the VCF order); tiles packed with EVOFST_MODE_STATS.
 *   evofst_alleles_from_vcf_line: allele numbers of one multi-allelic line (flag EVOFST_SITE_MULTI from the packer),
 *       alleles[2 * n_samples], 255 255 = individual missing;
 *   evofst_set_multiallelic: the multi-allelic sites of the uploaded shard (ascending shard-relative site indices,
 *       alleles [n_multi][2 * n_samples] in VCF sample order); forgotten by the next upload;
 *   evofst_get_indiv_diffs: block b covers sites [block_off[b], block_off[b+1]); arrays [n_blocks][n_ind (n_ind + 1) / 2],
 *       pair (i, j) at i (i + 1) / 2 + j.  On bi-allelic sites: hh = both heterozygous (j == i: heterozygous),
 *       hd = opposite homozygotes, oh = one homozygous and one heterozygous; ms = either one missing (all sites);
 *       d4 = on multi-allelic sites, the allele comparisons that differ, of 4 per site (j == i: sites where the
 *       individual's own two alleles differ).  The reference's matrices follow exactly:
 *       diffs_me = hh / 2 + hd + oh / 2 + d4 / 4 (diagonal hh + d4), hets-vs-homs = hh below and hd above the diagonal,
 *       missingness = ms; Richard's matrix = 2/3 hh + 2/3 hd + oh / 2 (diagonal hh). */
int evofst_alleles_from_vcf_line(const char *line, int n_samples, uint8_t *alleles);
/* Host only: the number of rand() draws getThisVariantCountsSimple makes on these lines (randomisePhase,
 * process_vcf_utils.cpp:268-276, 384-387: one per genotype whose two allele characters differ, on lines whose first
 * genotype is not phased).  The draws do not change any statistic of this path, but they move the C library's random
 * stream from which statsMain then picks its bootstrap blocks (process_vcf_stats.cpp:302); a caller that wants the
 * reference's replicates advances rand() by this many draws first. */
int evofst_stats_phase_draws(int64_t n_lines, const char *const *lines, int n_samples, int n_threads, int64_t *draws);
int evofst_set_multiallelic(evofst_ctx *ctx, int64_t n_multi, const int64_t *site_index, const uint8_t *alleles);
int evofst_get_indiv_diffs(evofst_ctx *ctx, int n_blocks, const int64_t *block_off, int32_t *hh, int32_t *hd, int32_t *oh,
                           int32_t *ms, int32_t *d4);
/* Heterozygous calls per individual: het_analysis (process_vcf_stats_functions.cpp:19-28) as statsMain drives it
 * (process_vcf_stats.cpp:239-242: only sites where no genotype is missing; multi-allelic sites count nowhere because
 * getThisVariantCountsSimple leaves their individualsWithVariant at 0, process_vcf_utils.cpp:397-406) over the uploaded
 * shard (tiles packed with EVOFST_MODE_STATS; individuals as for evofst_get_indiv_diffs): het[n_ind] = sites where the
 * individual is heterozygous, shared[n_ind] = those of them where the ALT allele was seen more than once among all
 * individuals (counts.overall > 1). */
int evofst_get_indiv_hets(evofst_ctx *ctx, int32_t *het, int32_t *shared);
/* Per-site per-population diversity (getSetHeterozygozities + calculatePiTwoSets,
 * process_vcf_fst.cpp:325-371): arrays [n_pops][n_sites]; any pointer may be NULL. */
int evofst_get_pop_site_div(evofst_ctx *ctx, int64_t first_site, int64_t n_sites, double *het,
                            double *het_nei, double *pi);
/* The positions that travelled with the shard (evofst_upload's pos, evofst_attach_device's d_pos, or the synthetic ones of
 * evofst_synth: one site per 100 bp): pos[n_sites] for sites first_site ...  EVOFST_ERR_STATE when the shard has none.
 * For callers that cut physical windows (process_vcf_fst.cpp:589-629) over data that only the device holds. */
int evofst_get_positions(evofst_ctx *ctx, int64_t first_site, int64_t n_sites, int32_t *pos);

/* Sums of ONE pair's per-site values over ranges of sites of the uploaded shard: the window sums of the two-set `evo fst`
 * (SNP windows process_vcf_fst.cpp:638-692, physical windows :589-629, the genome-wide averages :701-709), which the
 * reference forms by adding per-site values in site order (vector_average / vector_average_withRegion,
 * process_vcf_utils.h:369-385).  Range r = sites [lo[r], hi[r]); only the sites the pair includes (filter :543-544) count.
 * out[r][12] = sums of {Fst numerator, Fst denominator, dxy, heterozygosity of population i, of j, Nei's of i, of j,
 * pi of i, of j}, then {included sites, included sites with pi(i) != 0, with pi(j) != 0}; every sum starts from init[r][..]
 * (NULL: zero) and adds the sites in order, one after the other, so a window cut in two by a batch boundary comes out as
 * one loop over it would.  The caller owns the coordinates (which sites a window holds); the device owns every sum. */
int evofst_get_pair_range_sums(evofst_ctx *ctx, int pair, int64_t n_ranges, const int64_t *lo, const int64_t *hi,
                               const double *init, double *out);
/* Segment sums.  A segment is g = gcd(SIZE, STEP) consecutive included sites of a pair (in
 * against-all mode: see evofst_get_global_windows instead).  Segment m of pair q covers the
 * pair's included sites with 1-based global rank (m*g, (m+1)*g]; window "ending at rank e" is
 * the sum of segments [(e-SIZE)/g, e/g).  n_segments[q] = complete segments of pair q that END
 * in this shard; first_segment[q] = index of the first of them.  Arrays are
 * [max_segments][n_pairs] (row = local segment index m - first_segment[q]):
 *   sum_num, sum_den : sums of the Fst numerators / denominators,
 *   sum_dxy_k        : sum of the integer dxy numerators (sumKij, process_vcf_fst.cpp:299-321);
 *                      dxy sum = sum_dxy_k / (n_full[i]*n_full[j]),
 *   end_site         : shard-local site index of the last site of the segment.
 * The first segment of a shard with a non-zero base offset lacks the previous shard's carry:
 * add carry_in (see evofst_get_boundary). */
int64_t evofst_max_segments(evofst_ctx *ctx);
int evofst_get_segments(evofst_ctx *ctx, int64_t *first_segment, int64_t *n_segments,
                        double *sum_num, double *sum_den, double *sum_dxy_k, int32_t *end_site);
/* Boundary partials of this shard, 4 doubles per pair: {sum_num, sum_den, sum_dxy_k, n_sites}
 * of the included sites after the last complete segment (the carry handed to the next shard). */
int evofst_get_boundary(evofst_ctx *ctx, double *carry_out);
/* Whole-shard sums per pair: {sum_num, sum_den, sum_dxy_k, n_included} (genome-wide Fst of
 * process_vcf_fst.cpp:701 = sum_num / sum_den). */
int evofst_get_totals(evofst_ctx *ctx, double *totals);

/* Against-all mode: per-site values [n_use][n_sites] and the kept flag [n_sites]. */
int evofst_get_global_site_stats(evofst_ctx *ctx, int64_t first_site, int64_t n_sites,
                                 double *num, double *den, uint8_t *kept);
/* Against-all deque windows (evo_FstAgainstAll.cpp:237-240, 276-280): row r has
 * start_site/end_site (shard-local site indices; start -1 = the initial "0" coordinate) and
 * mean_num / mean_den [max_rows][n_use] (= vector_average of the two deques, left to right).
 * Returns the number of rows through *n_rows. */
int64_t evofst_max_global_rows(evofst_ctx *ctx);
int evofst_get_global_windows(evofst_ctx *ctx, int64_t *n_rows, int32_t *start_site,
                              int32_t *end_site, double *mean_num, double *mean_den);

#ifdef __cplusplus
}
#endif
#endif /* EVOFST_H */
