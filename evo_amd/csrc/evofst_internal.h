// evofst_internal.h -- shared between the C-ABI host code and the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace evofst {

constexpr int kTileSites = 64;
constexpr int kK1Threads = 256;    // 4 waves = 4 consecutive chunks (one wave per chunk) sharing the layout tables in LDS
constexpr int kK2Threads = 256;    // pairs per K2 block
constexpr int kPadSites = 65;      // LDS row stride (sites + 1) -> conflict-free pop-indexed reads
constexpr int kScanParts = 16;

// One (word, population) overlap of the packed haplotype layout; sorted by bit position.
struct SegDesc {
    uint64_t mask;   // bits of `word` that belong to `pop`
    uint32_t pop;
    uint32_t flags;  // bit0: last segment of the population, bit1: pop is in populationsToUse,
                     // bit2: pop is populationsToUse[0]
};
constexpr uint32_t kSegLast = 1u, kSegUse = 2u, kSegFirstUse = 4u, kSegWordEnd = 8u;   // bit3: last segment of its word
constexpr int kSegNFullShift = 8;   // flags >> 8: n_full of the population


// packed per-(site,pop) table word of K2: a = c (alt count), b = n_full - c
constexpr uint32_t kXBits = 14;
constexpr uint32_t kXMask = (1u << kXBits) - 1u;
constexpr uint32_t kXZero = 1u << 28;   // c == 0        (or site skipped)
constexpr uint32_t kXFixed = 1u << 29;  // c == nm       (or site skipped)

enum { MODE_COUNTS = 0, MODE_PAIRS = 1, MODE_GLOBAL = 2, MODE_GRAM = 3 };

// ---- device helpers shared by evofst_kernels.hip and evofst_fused.hip
template <typename CT> struct CountTraits;
template <> struct CountTraits<uint16_t> { static constexpr int shift = 8; static constexpr uint32_t mask = 0xffu; };
template <> struct CountTraits<uint32_t> { static constexpr int shift = 16; static constexpr uint32_t mask = 0xffffu; };

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
// acc + popcount(x) as the one instruction the hardware has for it (v_bcnt_u32_b32 adds its second operand).  Left to
// itself the compiler counts into fresh registers and adds three ways (v_bcnt, v_bcnt, v_add3): a fifth more vector
// instructions in the counting loops, which a single K1 launch on short rows is bound by.
__device__ __forceinline__ uint32_t popc_acc(uint32_t x, uint32_t acc) {
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}

constexpr int kRcpMax = 1023;
constexpr int kSub = 32;            // sites per LDS table fill (half a tile)
constexpr int kSubPad = kSub + 1;   // row stride in entries: pop-indexed reads stay conflict-free
#ifndef EVOFST_K2_GRP
#define EVOFST_K2_GRP 4
#endif
#ifndef EVOFST_K2_ENT_BATCH
#define EVOFST_K2_ENT_BATCH 4
#endif
#ifndef EVOFST_K2_UNROLL
#define EVOFST_K2_UNROLL 8
#endif
#ifndef EVOFST_K2_WPS
#define EVOFST_K2_WPS 3
#endif
constexpr int kGrp = EVOFST_K2_GRP;             // sites per snapshot group
constexpr int kK2WavesPerSimd = EVOFST_K2_WPS;  // register budget: 512 / waves per SIMD
constexpr uint32_t kXBadNum = 1u << 30;         // nm < 2  (h undefined)
constexpr uint32_t kXBadDen = 1u << 31;         // nm == 0 (p undefined)

struct __align__(16) RcpEntry { double r0, rr0, r1; uint32_t flags, pad; };   // read as two 16-byte halves

// position of the n-th (1-based) set bit of m; needs 1 <= n <= popc(m)
__device__ __forceinline__ int nth_set_bit(uint32_t m, int n) {
    int pos = 0;
#pragma unroll
    for (int w = 16; w >= 1; w >>= 1) {
        const int c = __popc(m & ((1u << w) - 1u));
        if (n > c) { n -= c; pos += w; m >>= w; }
    }
    return pos;
}
// position of the n-th (1-based) set bit of the 64-bit mask {hi, lo}; needs 1 <= n <= popc
__device__ __forceinline__ int nth_set_bit64(uint32_t lo, uint32_t hi, int n) {
    const int cl = __popc(lo);
    const bool up = n > cl;
    return nth_set_bit(up ? hi : lo, up ? n - cl : n) + (up ? 32 : 0);
}



struct K1Params {
    const ulonglong2 *tiles;
    const uint8_t *site_flags;  // may be null
    int64_t n_sites;
    int32_t n_tiles, W, P;
    const SegDesc *segs;
    const int32_t *word_seg_start;  // [W+1]
    const int32_t *n_full;          // [P]
    void *counts;                   // CT [n_tiles][P][64], c | nm << (4*sizeof(CT))
    int32_t chunk_tiles, n_chunks;
    int32_t n_segs;  // entries of segs
    // pairs mode
    const uint32_t *pairs;  // i | j << 16
    int32_t n_pairs;
    uint32_t *chunk_census;  // [n_pairs][n_chunks]
    uint64_t *zf_masks;      // [n_tiles][2][P] sites with c == 0 / c == nm (skipped sites set in both)
    uint64_t *irr_mask;      // [n_tiles] sites with nm < 2 or c == nm < n_full in some population
    uint64_t *miss_mask;     // [n_tiles] sites where some population has a missing allele (pairs mode)
    uint64_t *nan_mask;      // [n_tiles] sites where some population has fewer than two called alleles (pairs mode)
    uint32_t *tile_poison;   // [n_tiles] Gram mode: != 0 when a population has fewer than two called alleles at some site of the tile
    uint32_t *chunk_dirty;   // [n_chunks] != 0: the chunk holds a skipped, irregular or missing-allele site (pairs mode)
    // against-all mode
    uint2 *totals;            // [n_tiles*64] {sum alt, sum non-missing} over populationsToUse
    uint64_t *kept_mask;      // [n_tiles]
    uint64_t *skip_mask;      // [n_tiles] (both modes)
    uint32_t *chunk_used_kept;  // [2][n_chunks]
    int32_t debug;              // timing experiments (EVOFST_EXPERIMENTS builds only; results are wrong when set)
    // pairs mode, 16-bit counts: a chunk without a skipped, irregular or missing-allele site leaves one byte per count
    // (c alone: nm is the population's n_full there), its tiles back to back from the chunk's first byte
    // ([chunk_tiles][P][64] bytes at counts + chunk * chunk_tiles * P * 128); every other chunk the packed form above.
    // chunk_dirty tells the two apart.  Writes are what K1 pays most for.
    int32_t compact;
};

// one segment of one pair: sums of the Fst numerators / denominators, sum of sumKij (dxy numerators; integers, exact),
// site of the last included site of the segment
struct __align__(32) SegRow { double num, den, k; int32_t end, pad; };

struct K2Params {
    const void *counts;
    const uint8_t *site_flags;
    int64_t n_sites;
    int32_t n_tiles, P;
    int32_t max_n;  // largest n_full
    int32_t debug;  // EVOFST_DEBUG_K2 (profiling experiments only): 1 = skip pair phase, 2 = skip derive
    const int32_t *n_full;
    const uint32_t *pairs;
    int32_t n_pairs;
    int32_t chunk_tiles, n_chunks;
    const int64_t *chunk_offsets;  // [n_pairs][n_chunks] included sites of the pair before the chunk
    const int64_t *first_segment;  // [n_pairs]
    int32_t g;
    int64_t seg_cap;
    SegRow *seg_rows;              // [seg_cap][n_pairs]: a lane writes the four values of a segment as one 32-byte
                                   // sector (four separate arrays made four partial-line writes of it)
    const uint64_t *zf_masks, *irr_mask, *skip_mask;   // from K1
    const uint64_t *miss_mask;                         // from K1: sites where some population has a missing allele
    // chunk-boundary partials, planes [chunk][pair] (a wave writes contiguous rows)
    double *tail;                  // [3][n_chunks][n_pairs] {num, den, k} after the last completed segment
    int4 *tail_n;                  // {sites summed in the tail, segments the chunk completed, row of its first segment, -}
};

// pairs mode in one streaming kernel (evofst_fused.hip): tiles -> segment rows, nothing else read from or kept in HBM
struct FusedParams {
    const ulonglong2 *tiles;
    const uint8_t *site_flags;  // may be null
    int64_t n_sites;
    int32_t n_tiles, W, P;
    const SegDesc *segs;            // [n_segs] (word, population) overlaps in bit order, flags incl. kSegWordEnd
    int32_t n_segs;
    const int32_t *n_full;          // [P]
    int32_t max_n;
    const uint32_t *pairs;          // i | j << 16
    int32_t n_pairs;
    int32_t chunk_tiles, n_chunks;
    int32_t g;
    const int64_t *base;            // [n_pairs] included sites of the pair before this shard
    int64_t *first_segment;         // [n_pairs] out: base / g
    int64_t *incl_totals;           // [n_pairs] out: included sites of the shard
    int64_t seg_cap;
    SegRow *seg_rows;               // [seg_cap][n_pairs]
    double *tail;                   // [3][fused_tail_chunks()][n_pairs], as K2 writes it
    int4 *tail_n;                   // [fused_tail_chunks()][n_pairs]
    // look-back state: 8-byte {epoch, count} granules, valid when the tag is this launch's epoch
    uint64_t *agg0;                 // [n_chunks][n_pairs] included sites of a chunk
    uint64_t *agg1;                 // [ceil(n_chunks / 32)][n_pairs] ... of a group of 32 chunks
    uint64_t *agg2;                 // [ceil(n_chunks / 1024)][n_pairs] ... of a group of 32 groups
    uint32_t *ticket;               // chunk tickets; back to 0 when the launch ends
    uint32_t *err;                  // set when a wait gave up
    uint32_t epoch;
    // statistics half on its own (counts != null): a region is one of K1's n_chunks chunks of chunk_tiles tiles
    const void *counts;             // K1's packed counts
    const uint64_t *zf_masks, *irr_mask, *skip_mask, *miss_mask;   // K1's ballots
    const uint64_t *nan_mask;       // K1: sites where some population has fewer than two called alleles
    const int64_t *chunk_offsets;   // [n_pairs][n_chunks] from the scan
    const uint32_t *chunk_dirty;    // [n_chunks] from K1, or null: the walk of evofst_fused.hip takes the chunks marked here,
                                    // clean_walk_kernel the others
    int32_t piece_tiles;            // tiles of a region held in LDS at a time
    int32_t compact;                // the chunks clean_walk_kernel takes hold one byte per count (K1Params::compact)
    int32_t debug;                  // -DEVOFST_EXPERIMENTS builds only (tools/): 1 no look-back waits, 2 no phase B, 4 no tile loads
};

// pairs mode as one streaming kernel with producer and consumer waves (evofst_stream.hip)
constexpr int kStreamMaxWaves = 12;                    // producer + consumer waves of a workgroup
constexpr int kStreamMaxThreads = 64 * kStreamMaxWaves;
constexpr int kStreamFan = 32;                         // look-back fan-out per level
constexpr int kStreamAggShift = 26;                    // levels 1, 2: {contributions << 26 | included sites}
constexpr uint32_t kStreamAggMask = (1u << kStreamAggShift) - 1u;
constexpr int kStreamRow = 80;                         // bytes per (tile, population) row of alt counts in LDS: 64 sites + 16 of padding
constexpr int kStreamMaxPairs = 256;
#ifndef EVOFST_STREAM_WPS
#define EVOFST_STREAM_WPS 4                            // waves per SIMD the register allocation aims at
#endif
struct StreamParams {
    const ulonglong2 *tiles;
    const uint8_t *site_flags;      // may be null
    int64_t n_sites;
    int32_t n_tiles, W, P;
    const SegDesc *segs;            // [n_segs + 2] descriptor stream, two spare entries behind it (upload_layout)
    const int32_t *n_full;          // [P]
    const uint32_t *pairs;          // i | j << 16
    int32_t n_pairs;
    int32_t n_prod, n_cons;         // producer and consumer waves of a workgroup
    int32_t poll;                   // s_sleep(16)s between two polls of the look-back flags
    int32_t n_sets;                 // sets of consumer waves (1 or 2) that take the workgroup's chunks in turn; n_cons = all of them
    int32_t n_buf;                  // LDS buffers (chunks of a workgroup in flight): 2 .. kStreamMaxBufs
    int32_t tr;                     // tiles a producer wave counts per chunk: waves 0 .. 3 (the oldest wave of their SIMD, which
    int32_t tr2;                    // the hardware serves first), and the waves behind them (tr2 <= tr: both kinds then finish together)
    int32_t chunk_tiles, n_chunks;  // chunk_tiles = n_prod * tr
    int32_t g;
    const int64_t *base;            // [n_pairs] included sites of the pair before this shard
    int64_t *first_segment;         // [n_pairs] out: base / g
    int64_t *incl_totals;           // [n_pairs] out: included sites of the shard
    int64_t seg_cap;
    SegRow *seg_rows;               // [seg_cap][n_pairs]
    double *tail;                   // [3][n_chunks][n_pairs]
    int4 *tail_n;                   // [n_chunks][n_pairs]
    // look-back (evofst_stream.hip, "Look-back"): included sites per pair of a chunk (+ 1), of a group of 32 chunks and of a
    // group of 32 groups ({contributions, sum} in 64 bits), and one flag word per row (rows 0.., l1_row.., l2_row..) that
    // counts the row's contributions.  Two copies: a launch uses copy (launch counter & 1) and zeroes the other one
    uint32_t *agg0[2];              // [ceil(n_chunks / 2)][n_pairs] two chunks' 16-bit words each
    uint32_t *agg1[2];              // [ceil(n_chunks / 32)][n_pairs] {contributions << 26 | sites}
    uint32_t *agg2[2];              // [ceil(n_chunks / 1024)][n_pairs]
    uint32_t *flags[2];             // [n_chunks + ceil(n_chunks / 32) + ceil(n_chunks / 1024)]
    int32_t l1_row, l2_row;
    uint32_t *ticket;               // chunk tickets; back to 0 when the launch ends
    uint32_t *err;                  // set when a wait gave up
    uint32_t *launches;             // launch counter, bumped by the launch's last ticket draw
    uint32_t *n_dirty;              // chunks left to the general walk, counted on through the launches (the host takes differences)
    // what a chunk the integer walk cannot take leaves for the general walk (K1's and the scan's outputs)
    void *counts;                   // u16 [n_tiles][P][64]
    uint64_t *zf_masks, *irr_mask, *skip_mask, *miss_mask, *nan_mask;
    int64_t *chunk_offsets;         // [n_pairs][n_chunks]
    uint32_t *chunk_dirty;          // [n_chunks]
    int32_t debug;                  // -DEVOFST_EXPERIMENTS builds only (tools/): 1 no walk, 2 no look-back, 4 no counting, 8 no consumer work
    unsigned long long *stamps;     // -DEVOFST_EXPERIMENTS: [workgroups][kStreamStampIters][16] s_memrealtime stamps, or null
};
constexpr int kStreamStampIters = 48;
constexpr int kStreamMaxBufs = 4;                      // LDS buffers of a workgroup (chunks in flight)

struct SiteStatParams {
    const void *counts;
    const uint8_t *site_flags;
    int64_t n_sites_total, first_site, n_sites;
    int32_t P;
    const int32_t *n_full;
    const uint32_t *pairs;
    int32_t n_pairs;
    double *num, *den, *dxy;  // [n_pairs][n_sites] (device), may be null
    uint8_t *include;
};

struct PbsParams {
    const void *counts;
    const uint8_t *site_flags;
    int64_t first_site, n_sites;
    int32_t P, n_trios;
    const int32_t *trios;   // [n_trios][3] population indices (device)
    double *pbs;            // [n_trios][3][n_sites]
    uint8_t *used;          // [n_trios][n_sites]
};

// windowed outgroup-to-ingroup dxy (DistOutMain, evo_distanceToOutgroups.cpp:274-294)
struct DistParams {
    const void *counts;
    const uint8_t *site_flags;
    const int64_t *win_off;     // [n_windows + 1] site offsets (device)
    int32_t P, n_windows, n_out, n_in, carry_first;
    const int32_t *out_idx;     // [n_out] population indices (device)
    const int32_t *in_idx;      // [n_in]
    double *sum;                // [n_windows][n_out][n_in]
    int32_t *nsnp, *miss;       // [n_windows][n_out][n_in]
    int32_t *used, *missing;    // [n_windows][n_out]
};

// per-individual pairwise differences (process_vcf_stats_functions.cpp:180-281)
struct IndivParams {
    const ulonglong2 *tiles;
    const uint8_t *site_flags;
    int64_t n_sites;
    int32_t W, n_tiles, n_ind;       // n_ind = layout individuals (H / 2)
    uint64_t *planes;                // [n_tiles][n_ind][4] = {het, hom-alt, hom-ref, missing} bits over the tile's sites
    const int64_t *block_off;        // [n_blocks + 1] site offsets
    int32_t n_blocks, n_pair_blocks; // pair blocks: 16 x 16 individuals on or below the diagonal
    int64_t T;                       // n_ind (n_ind + 1) / 2
    int32_t *hh, *hd, *oh, *ms, *d4; // [n_blocks][T]
    const int64_t *multi_site;       // [n_multi] site index of the multi-allelic sites, ascending
    const uint8_t *multi_alleles;    // [n_multi][n_ind][2], 255 = missing individual
    int64_t n_multi;
    int32_t *het_cnt, *shared_cnt;   // [n_ind] each (indiv_hets_kernel; zeroed by the caller)
};

struct PopDivParams {
    const void *counts;
    int64_t first_site, n_sites;
    int32_t P;
    const int32_t *n_full;
    double *het, *het_nei, *pi;  // [P][n_sites]
};

// evofst_get_pair_range_sums: per-site arrays of ONE pair ([n_sites] each) and the caller's ranges
constexpr int kRangeSumCols = 12;
struct RangeSumParams {
    const double *num, *den, *dxy, *het_i, *het_j, *nei_i, *nei_j, *pi_i, *pi_j;
    const uint8_t *include;
    int64_t n_sites, n_ranges;
    const int64_t *lo, *hi;      // [n_ranges] sites [lo, hi)
    const double *init;          // [n_ranges][kRangeSumCols] or null
    double *out;                 // [n_ranges][kRangeSumCols]
};
hipError_t launch_pair_range_sums(const RangeSumParams &p, hipStream_t st);

struct GlobalParams {
    const void *counts;
    const uint2 *totals;
    const uint64_t *kept_mask, *skip_mask;
    int64_t n_sites;
    int32_t n_tiles, P;
    const int32_t *use_idx;   // [n_use]
    const double *thresholds; // [n_use] set_sizes[u] * 0.8
    int32_t n_use;
    int32_t chunk_tiles, n_chunks;
    const int64_t *chunk_used_kept_off;  // [2][n_chunks] used / kept sites before the chunk
    int32_t size, step;
    int32_t tn_max;       // total chromosomes of populationsToUse
    int32_t rcp_lo, rcp_hi; // reciprocal table: 1/n for n <= rcp_lo, 1/(tn_max - d) for d <= rcp_hi (global_rcp_split)
    const int64_t *base;  // {used, kept} sites that precede this shard (device)
    int32_t halo_n;       // the last halo_n (<= size) of them, oldest first: what the deques hold when the shard starts
    const double *halo_num, *halo_den;   // [halo_n][n_use]
    const int32_t *halo_tag;             // [halo_n] reported as a row's start site while the oldest entry is a halo entry
    int64_t row_cap;
    int32_t *row_start, *row_end;  // [row_cap]
    double *row_num, *row_den;     // [row_cap][n_use]
    const int64_t *chunk_row_off;  // [n_chunks+1] first output row of the chunk
    int32_t group;                 // global_window_kernel: K1 chunks a workgroup walks in a row (0: launch_global chooses)
    const int32_t *n_full;         // [P] chromosomes per population (global_tile_kernel)
    int32_t kw;                    // global_tile_kernel: entries per population of its LDS ring (odd, >= size + 64)
    // per-site output (on demand)
    double *site_num, *site_den;   // [n_use][n_sites] or null
    uint8_t *site_kept;
    int64_t first_site, n_out_sites;
};

// K4: window sums for every pair through Gram matrices (windows whose membership does not depend on the pair)
struct GramParams {
    const void *counts;
    int64_t n_sites;
    int32_t n_tiles, P;
    int32_t max_n;
    const int32_t *n_full;
    const uint64_t *masks;       // [n_tiles][4][P] from K1 (Gram mode): c == 0, c == nm, nm < 2, nm == 0
    const uint64_t *skip_mask;   // [n_tiles]
    const uint32_t *tile_poison; // [n_tiles] from K1 (Gram mode): != 0 when one of the nm < 2 / nm == 0 masks of the tile has a bit set
    const int64_t *win_start;    // [n_windows_total + 1] first site of every window (device)
    int64_t first_window;        // this launch covers windows [first_window, first_window + n_windows)
    int64_t max_window_sites;    // the longest window of the set (0: not known): picks the one-stream denominator
    int32_t n_windows;
    double *U, *C;               // [n_windows][P] per-population sums of t + h and of c over the window
    double *D;                   // [n_windows][P] the window's included-by-flags sites where the population has a called allele (16-bit counts, <= 63 chromosomes)
    int64_t n_pairs;             // P (P - 1) / 2, pair (i < j) at i (2P - i - 1) / 2 + j - i - 1
    // results as the workgroups hold them: [n_windows][block pairs][A|B][wave 4][slot 16][lane 64] x 16 bytes, A = {num, den},
    // B = {k, n_included}; every store instruction of the epilogue writes 1 KiB of consecutive, aligned bytes
    // (gram_unpack_kernel turns the table into the four [n_windows][n_pairs] arrays the getter hands out)
    double2 *out_raw;
    int32_t two_streams;         // never the one-contraction form (evofst_set_option "gram_denominator" 2)
    int32_t fmt;                 // entries of out_raw (gram_result_format): 0 = 16 B {num, den} / {k, n_included}, 1 = 8 B, the one-contraction
                                 // form's, 2 = 8 B {den} / {k, n_included | num is NaN << 31} as 32-bit integers, num left to the getter
    int32_t debug;               // EVOFST_DEBUG_GRAM (timing experiments): 1 no contraction loop, 2 no epilogue, 4 no popsums
};

struct SynthParams {
    ulonglong2 *tiles;
    int32_t *pos;
    int64_t n_sites, first_site_global;
    int32_t n_tiles, W, P;
    const SegDesc *segs;
    const int32_t *word_seg_start;
    uint64_t seed;
    uint32_t miss_thr;  // missing_rate * 2^32
};

// launchers (evofst_kernels.hip)
hipError_t launch_k1(const K1Params &p, int count_bytes, int mode, hipStream_t st);
hipError_t launch_scan(const uint32_t *census, const int64_t *base, int64_t *offsets, int64_t *totals,
                       int n_chunks, int n_cols, int64_t g, int64_t *first_seg, hipStream_t st,
                       const int64_t *gathered = nullptr, int64_t row_stride = 0, int n_rows = 0, int64_t *base_out = nullptr,
                       int64_t *totals_copy = nullptr);
hipError_t launch_k2(const K2Params &p, int count_bytes, hipStream_t st);
hipError_t launch_stitch(const K2Params &p, hipStream_t st);
hipError_t launch_totals(const K2Params &p, const int64_t *base, const int64_t *incl_totals, double *totals, double *carry,
                         hipStream_t st);
hipError_t launch_site_stats(const SiteStatParams &p, int count_bytes, hipStream_t st);
hipError_t launch_pbs_sites(const PbsParams &p, int count_bytes, hipStream_t st);
hipError_t launch_abs_sites(const PbsParams &p, int count_bytes, hipStream_t st);   // trios = [n][4], pbs = [n][2][n_sites]
hipError_t launch_dist_windows(const DistParams &p, int count_bytes, hipStream_t st);
hipError_t launch_indiv_planes(const IndivParams &p, hipStream_t st);
hipError_t launch_indiv_pairs(const IndivParams &p, hipStream_t st);
hipError_t launch_indiv_multi(const IndivParams &p, hipStream_t st);
hipError_t launch_indiv_hets(const IndivParams &p, hipStream_t st);
hipError_t launch_unpack_segments(const SegRow *rows, size_t n, double *num, double *den, double *k, int32_t *end, hipStream_t st);
hipError_t launch_pop_div(const PopDivParams &p, int count_bytes, hipStream_t st);
hipError_t launch_unpack_counts(const void *counts, int count_bytes, int P, int64_t first_site, int64_t n_sites,
                                int32_t *alt, int32_t *nm, hipStream_t st);
hipError_t launch_global_rows(const uint64_t *kept_mask, const uint64_t *skip_mask, const int64_t *chunk_off,
                              int n_chunks, int chunk_tiles, int n_tiles, int size, int step,
                              uint32_t *chunk_rows, hipStream_t st);
hipError_t launch_global(const GlobalParams &p, int count_bytes, size_t lds_bytes, hipStream_t st);
hipError_t launch_count_marks(const uint32_t *marks, int n, uint32_t *out, hipStream_t st);
hipError_t launch_global_sites(const GlobalParams &p, int count_bytes, hipStream_t st);
hipError_t launch_synth(const SynthParams &p, hipStream_t st);
size_t gram_raw_bytes(int P, int64_t n_windows);
int gram_result_format(const GramParams &p, int count_bytes);
int64_t gram_one_rule_sites(int count_bytes, int max_n);   // longest window of the one-contraction form (0: not for these populations)
hipError_t launch_gram_unpack(const double2 *raw, int P, int64_t n_windows, bool i8, int fmt, const int64_t *sub_off, int64_t sub_base, const double *U, double *num, double *den, double *k,
                              double *cnt, hipStream_t st);
bool gram_uses_i8(int count_bytes, int max_n);
bool gram_one_stream(const GramParams &p, int count_bytes);     // the denominator as one FP64 contraction (gram_kernel<..., ONE>)
hipError_t launch_gram_popsums(const GramParams &p, int count_bytes, hipStream_t st);
hipError_t launch_gram(const GramParams &p, int count_bytes, bool has_missing, hipStream_t st);
size_t k2_lds_bytes(int P, int max_n);
size_t fused_lds_bytes(int P, int n_segs, int max_n, int n_pairs, int chunk_tiles, int count_bytes, int sub);
int fused_threads(int n_pairs);
int fused_tail_chunks(int n_pairs, int n_chunks);   // entries per pair of the tail table (sub-chunks when every wave owns a part of a chunk)
// sub: sites per table fill (16 or 32); max_blocks <= 0: as many workgroups as the device holds at once (occupancy query)
hipError_t launch_fused(const FusedParams &p, int count_bytes, int sub, int max_blocks, hipStream_t st);
size_t clean_walk_lds_bytes(int P, int n_pairs, int piece_tiles, bool compact);
size_t stream_buffer_bytes(int P, int chunk_tiles);
size_t stream_lds_bytes(int P, int n_pairs, int chunk_tiles, int n_buf);
// max_blocks <= 0: as many workgroups as the device holds at once (occupancy query); any number is safe
hipError_t launch_stream(const StreamParams &p, int max_blocks, hipStream_t st);
hipError_t launch_clean_walk(const FusedParams &p, hipStream_t st);
size_t global_lds_bytes(int tn_max, int max_n, int n_use, int size);
size_t global_tile_lds_bytes(int tn_max, int max_n, int n_use, int size, int *kw_out);
hipError_t launch_global_tile(const GlobalParams &p, int count_bytes, size_t lds_bytes, hipStream_t st);
void global_rcp_split(int tn_max, int max_n, int *lo, int *hi);

}  // namespace evofst
