// evofst_api.cpp -- the C ABI declared in include/evofst.h (host side, HIP runtime calls only).
//
// There is no CPU fallback: every compute entry point launches the kernels of
// evofst_kernels.hip and fails with EVOFST_ERR_HIP when no GPU is available.
#include "../../include/evofst.h"
#include "evofst_internal.h"

#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <numeric>
#include <stdexcept>
#include <mutex>
#include <string>
#include <atomic>
#include <thread>
#include <vector>

using namespace evofst;

// ---- scratch of evofst_run_pairs_noalloc: the look-back granules, ticket / error words and tail table of the one-kernel
// pairs pass, for callers that cannot allocate.  One copy per GPU, held until the process ends; host code reaches it only
// through the address hipGetSymbolAddress returned for that GPU (g_scratch_ptr), never through the array's name.
constexpr size_t kNoAllocScratchBytes = (size_t)16 << 20;
static __device__ __align__(256) unsigned char scratch[kNoAllocScratchBytes];

namespace {

constexpr int kMaxDevices = 64;
void *g_scratch_ptr[kMaxDevices] = {};        // per device, set by evofst_create
uint32_t g_scratch_epoch[kMaxDevices] = {};   // per device: launch epochs of the granules in that GPU's scratch
std::mutex g_scratch_mu[kMaxDevices];         // per device: one host thread at a time carves the scratch, takes an epoch and launches

thread_local std::string g_create_error;

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

constexpr int kNumKernels = 5;   // 0 counts, 1 offsets scan, 2 pair / against-all / Gram statistics, 3 stitch, 4 fused pairs pass

}  // namespace

struct evofst_ctx {
    int device = 0;
    bool host_only = false;  // evofst_create(-1): packer-only context, no GPU is touched
    hipStream_t stream = nullptr;
    bool own_stream = true;
    std::string err;

    // layout
    int n_samples = 0, n_pops = 0, H = 0, W = 0, count_bytes = 2;
    std::vector<int32_t> sample_pop, n_full, pop_off;
    std::vector<int32_t> src_hap;  // packed haplotype -> input haplotype (2*sample + allele)
    std::vector<SegDesc> segs;
    std::vector<int32_t> word_seg_start;
    DevBuf d_segs, d_wss, d_nfull;

    // mode
    int mode = MODE_COUNTS;
    std::vector<uint32_t> pairs;
    std::vector<int32_t> use_idx;
    std::vector<double> thresholds;
    DevBuf d_pairs, d_use, d_thr;
    int snp_size = 0, snp_step = 0, g = INT_MAX;
    std::vector<int64_t> base;  // n_pairs (pairs mode) or 2 (against-all)
    DevBuf d_base, d_first_seg;
    bool offsets_dirty = true;

    // data
    int64_t cap_sites = 0, n_sites = 0;
    int n_tiles = 0, chunk_tiles = 0, n_chunks = 0;
    DevBuf d_tiles, d_flags, d_pos, d_counts;
    DevBuf d_planes, d_multi_site, d_multi_alleles;   // per-individual differences
    int64_t n_multi = 0;
    const void *tiles = nullptr;  // d_tiles.p or an attached pointer
    const uint8_t *flags = nullptr;
    const int32_t *pos = nullptr;
    bool have_flags = false, have_pos = false;

    // pairs-mode work buffers
    DevBuf d_census, d_offsets, d_incl_totals, d_seg_rows, d_seg_num, d_seg_den, d_seg_k, d_seg_end, d_zf, d_irr,
        d_tail, d_tail_n, d_carry, d_win, d_halo_num, d_halo_den, d_halo_tag, d_gU, d_gC, d_gnum, d_gden, d_gk, d_gcnt, d_graw,
        d_nrows, d_totals;
    int64_t seg_cap = 0;
    std::vector<int64_t> win_start;   // Gram mode: site windows
    // ... and their pieces, where windows are longer than the one-contraction form of the Gram path takes (evofst_run_gram):
    // piece starts (+ the end of the last), and the first piece of every window (+ the number of pieces)
    std::vector<int64_t> gram_sub, gram_suboff;
    int64_t gram_sub_rule = -1;       // the piece length they were cut for (-1: not cut yet)
    DevBuf d_gsub, d_gsuboff;
    bool gram_split = false;          // the last evofst_run_gram ran pieces
    int64_t gram_sub_base = 0, gram_first_window = 0;   // its first piece and first window
    bool base_host_valid = true;      // c->base mirrors d_base (false after a device-to-device set_base_offsets)
    const int64_t *gathered = nullptr; // pending evofst_set_base_offsets_gathered: the next scan sums these rows into d_base
    int64_t gathered_stride = 0;
    int gathered_rows = 0;
    int halo_n = 0;                   // against-all mode: kept sites handed over by the shards before this one
    int64_t gram_windows = 0;         // windows of the last run_gram
    int gram_missing = -1;            // -1 unknown, else whether any site needs the missing-allele correction
    // against-all work buffers
    DevBuf d_gtotals, d_kept, d_skip, d_used_kept, d_uk_off, d_uk_tot, d_chunk_rows, d_row_off, d_row_tot, d_row_start,
        d_row_end, d_row_num, d_row_den;
    int64_t row_cap = 0;

    // fused pairs pass (evofst_fused.hip): look-back granules, ticket + error words, launch epoch
    DevBuf d_agg0, d_agg1, d_agg2, d_fsync, d_miss, d_dirty, d_poison, d_nan;
    uint32_t epoch = 0;
    bool plain_counts = false;        // d_counts holds the packed counts of the loaded data, every tile (from any K1 launch)
    bool compact_counts = false;      // d_counts comes from a pairs-mode K1 that left one byte per count in its clean chunks
    bool prefer_plain = false;        // a getter has asked for per-site values on this context: K1 keeps to packed counts
    bool counts_done = false, stats_done = false;
    bool profiling = false;

    // options (evofst_set_option): read when a pass is planned, never from the process environment
    int opt_pairs_pass = 0;           // 0 auto, 1 K1 + scan + walk, 2 the producer / consumer stream pass, 3 the lock-step one-kernel pass
    int opt_chunk_tiles = 0;          // tiles per chunk (0: the library chooses)
    int opt_gram_den = 0;             // Gram denominator: 0 the library chooses, 2 always both FP64 streams
    int opt_walk = 0;                 // K1 + scan + walk: 0 auto, 1 general walk on every chunk, 2 k2_window_kernel
    int opt_stream_blocks = 0;        // workgroups of the stream pass (0: as many as the device holds)
    bool check_dirty = false;                       // the check word is the stream pass's: its count of marked chunks lies behind it
    uint32_t stream_dirty_cum = 0, stream_dirty_last = 0, stream_launches = 0; bool stream_dirty_known = false;   // (what the last synchronised passes marked)
    // chunks K1 marked for the general walk on the data that is loaded: asked once per data set at evofst_sync, so that later
    // passes over the same data skip (or size) the general walk's launch instead of letting it find out
    bool k1_marks_valid = false, k1_marks_known = false; uint32_t k1_marks = 0;
    int last_global_walk = 0;                       // evofst_last_global_walk
    int last_gram_streams = 0;                      // evofst_last_gram_streams
    int last_gram_fmt = 0;                          // GramParams::fmt of the last launch (what evofst_get_gram unpacks)
    int last_pairs_pass = 0;                        // evofst_last_pairs_pass
    int opt_global_walk = 0;                        // against-all rows: 0 auto, 1 lane = population kernel, 2 two-phase kernel where it fits
    int opt_stream_sets = 0;                        // its sets of consumer waves (0: the library chooses)
    int opt_stream_bufs = 0;                        // its LDS buffers per workgroup (0: the library chooses)
    int opt_stream_prod = 0, opt_stream_cons = 0;   // its producer / consumer waves per workgroup (0: the library chooses)
    // stream pairs pass (evofst_stream.hip): granule rows and their flags (two copies each), ticket / error / launch words
    DevBuf d_sagg, d_sflag, d_ssync;
    int64_t sagg_rows = -1, sagg_pairs = -1;   // shape of the tables at the last launch (another shape: zero them)
    // a one-kernel pass has not had its error word looked at yet: where the word is, and what to zero when it is set
    const uint32_t *check_word = nullptr;
    void *check_zero[3] = {nullptr, nullptr, nullptr};
    size_t check_zero_bytes[3] = {0, 0, 0};
    hipEvent_t ev[kNumKernels][2] = {};
    bool ev_valid[kNumKernels] = {};
};

namespace {

int fail(evofst_ctx *c, int code, const std::string &msg) {
    if (c) c->err = msg;
    return code;
}
// No C++ exception leaves the C ABI: every `int` entry point is a function-try-block that ends here.
int abi_exception(evofst_ctx *c) {
    try {
        throw;
    } catch (const std::bad_alloc &) {
        return fail(c, EVOFST_ERR_NOMEM, "out of host memory");
    } catch (const std::exception &e) {
        return fail(c, EVOFST_ERR_STATE, std::string("internal error: ") + e.what());
    } catch (...) {
        return fail(c, EVOFST_ERR_STATE, "internal error: unknown exception");
    }
}
int hip_fail(evofst_ctx *c, hipError_t e, const char *what) {
    return fail(c, EVOFST_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define NEED_GPU(c)                                                                                   \
    do {                                                                                              \
        if ((c)->host_only) return fail(c, EVOFST_ERR_STATE, "host-only context (evofst_create(-1)): no GPU work possible"); \
    } while (0)
#define HIPCHK(c, call)                                        \
    do {                                                       \
        hipError_t e__ = (call);                               \
        if (e__ != hipSuccess) return hip_fail(c, e__, #call); \
    } while (0)

int ensure(evofst_ctx *c, DevBuf &b, size_t bytes) {
    if (bytes == 0) bytes = 16;
    if (b.bytes >= bytes) return EVOFST_OK;
    if (b.p) { HIPCHK(c, hipFree(b.p)); b.p = nullptr; b.bytes = 0; }
    hipError_t e = hipMalloc(&b.p, bytes);
    if (e != hipSuccess) { b.p = nullptr; return fail(c, EVOFST_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e)); }
    b.bytes = bytes;
    return EVOFST_OK;
}
void release(DevBuf &b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr; b.bytes = 0;
}
template <typename T>
int upload_vec(evofst_ctx *c, DevBuf &b, const std::vector<T> &v) {
    int rc = ensure(c, b, v.size() * sizeof(T));
    if (rc) return rc;
    if (!v.empty()) {
        HIPCHK(c, hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));  // v may be a temporary
    }
    return EVOFST_OK;
}

int n_tiles_of(int64_t n_sites) { return (int)((n_sites + kTileSites - 1) / kTileSites); }

int upload_layout(evofst_ctx *c) {
    // flags of the segment table depend on the against-all selection
    std::vector<uint8_t> use(c->n_pops, 0);
    int first_use = -1;
    if (c->mode == MODE_GLOBAL) {
        for (size_t u = 0; u < c->use_idx.size(); ++u) use[c->use_idx[u]] = 1;
        if (!c->use_idx.empty()) first_use = c->use_idx[0];
    }
    for (auto &d : c->segs) {
        d.flags &= kSegLast | kSegWordEnd;
        if (use[d.pop]) d.flags |= kSegUse;
        if ((int)d.pop == first_use) d.flags |= kSegFirstUse;
        d.flags |= (uint32_t)c->n_full[d.pop] << kSegNFullShift;
    }
    int rc;
    // K1 reads the stream two entries ahead of its use: two spare entries (end of a word) behind it
    std::vector<SegDesc> stream(c->segs);
    for (int i = 0; i < 2; ++i) { SegDesc e{}; e.flags = kSegWordEnd; stream.push_back(e); }
    if ((rc = upload_vec(c, c->d_segs, stream))) return rc;
    if ((rc = upload_vec(c, c->d_wss, c->word_seg_start))) return rc;
    if ((rc = upload_vec(c, c->d_nfull, c->n_full))) return rc;
    return EVOFST_OK;
}

void record(evofst_ctx *c, int which, int end) {
    if (!c->profiling) return;
    if (!c->ev[which][end]) (void)hipEventCreate(&c->ev[which][end]);
    (void)hipEventRecord(c->ev[which][end], c->stream);
    if (end) c->ev_valid[which] = true;
}

int plan_chunks(evofst_ctx *c) {
    c->n_tiles = n_tiles_of(c->n_sites);
    int ct = c->opt_chunk_tiles;
    // ~4096 chunks (4 waves of the walk per SIMD), and no fewer than 8 tiles a chunk: at 1 M sites chunks of 8 tiles cost
    // K1 and the walk nothing against chunks of 4, halve the scan's and the stitch's work, and leave other streams'
    // kernels more room (the 3-stream step of cfg2: 48.8 against 53.7 us; tools/gpu_ct.sh)
    if (ct <= 0) ct = std::max(8, (c->n_tiles + 4095) / 4096);
    ct = std::min(ct, 256);   // K2's accumulators run from the start of a chunk: 16k sites keep their rounding
                              // error 1e-13 of a window's terms (and the 32-bit count prefixes far from wrapping)
    c->chunk_tiles = ct;
    c->n_chunks = (c->n_tiles + ct - 1) / ct;
    return EVOFST_OK;
}

int run_scan(evofst_ctx *c, int64_t *totals_copy = nullptr) {
    if (c->mode == MODE_PAIRS) {
        const int np = (int)c->pairs.size();
        int rc;
        if ((rc = ensure(c, c->d_offsets, (size_t)c->n_chunks * np * 8))) return rc;
        if ((rc = ensure(c, c->d_incl_totals, (size_t)np * 8))) return rc;
        if ((rc = ensure(c, c->d_first_seg, (size_t)np * 8))) return rc;
        record(c, 1, 0);
        HIPCHK(c, launch_scan((const uint32_t *)c->d_census.p, (const int64_t *)c->d_base.p, (int64_t *)c->d_offsets.p,
                              (int64_t *)c->d_incl_totals.p, c->n_chunks, np, std::max(1, c->g), (int64_t *)c->d_first_seg.p,
                              c->stream, c->gathered, c->gathered_stride, c->gathered_rows, (int64_t *)c->d_base.p, totals_copy));
        record(c, 1, 1);
    } else if (c->mode == MODE_GLOBAL) {
        int rc;
        if ((rc = ensure(c, c->d_uk_off, (size_t)c->n_chunks * 2 * 8))) return rc;
        if ((rc = ensure(c, c->d_uk_tot, 2 * 8))) return rc;
        record(c, 1, 0);
        HIPCHK(c, launch_scan((const uint32_t *)c->d_used_kept.p, (const int64_t *)c->d_base.p, (int64_t *)c->d_uk_off.p,
                              (int64_t *)c->d_uk_tot.p, c->n_chunks, 2, 1, nullptr, c->stream, c->gathered, c->gathered_stride,
                              c->gathered_rows, (int64_t *)c->d_base.p, totals_copy));
        record(c, 1, 1);
    }
    c->offsets_dirty = false;
    c->gathered = nullptr;            // d_base holds the sums from here on
    return EVOFST_OK;
}

int set_base_internal(evofst_ctx *c, const int64_t *offsets) {
    size_t n = c->mode == MODE_PAIRS ? c->pairs.size() : 2;
    c->base.assign(n, 0);
    if (offsets) std::copy(offsets, offsets + n, c->base.begin());
    int rc = upload_vec(c, c->d_base, c->base);
    if (rc) return rc;
    c->gathered = nullptr;
    c->base_host_valid = true;
    c->offsets_dirty = true;
    c->stats_done = false;
    return EVOFST_OK;
}

K2Params make_k2(evofst_ctx *c) {
    K2Params p{};
    p.counts = c->d_counts.p;
    p.site_flags = c->have_flags ? c->flags : nullptr;
    p.n_sites = c->n_sites;
    p.n_tiles = c->n_tiles; p.P = c->n_pops;
    p.max_n = *std::max_element(c->n_full.begin(), c->n_full.end());
#ifdef EVOFST_EXPERIMENTS
    if (const char *e = getenv("EVOFST_DEBUG_K2")) p.debug = atoi(e);
#endif
    p.n_full = (const int32_t *)c->d_nfull.p;
    p.pairs = (const uint32_t *)c->d_pairs.p;
    p.n_pairs = (int)c->pairs.size();
    p.chunk_tiles = c->chunk_tiles; p.n_chunks = c->n_chunks;
    p.chunk_offsets = (const int64_t *)c->d_offsets.p;
    p.first_segment = (const int64_t *)c->d_first_seg.p;
    p.g = c->g; p.seg_cap = c->seg_cap;
    p.seg_rows = (SegRow *)c->d_seg_rows.p;
    p.zf_masks = (const uint64_t *)c->d_zf.p; p.irr_mask = (const uint64_t *)c->d_irr.p; p.skip_mask = (const uint64_t *)c->d_skip.p;
    p.miss_mask = (const uint64_t *)c->d_miss.p;
    p.tail = (double *)c->d_tail.p;
    p.tail_n = (int4 *)c->d_tail_n.p;
    return p;
}

GlobalParams make_global(evofst_ctx *c) {
    GlobalParams p{};
    p.counts = c->d_counts.p;
    p.totals = (const uint2 *)c->d_gtotals.p;
    p.kept_mask = (const uint64_t *)c->d_kept.p; p.skip_mask = (const uint64_t *)c->d_skip.p;
    p.n_sites = c->n_sites; p.n_tiles = c->n_tiles; p.P = c->n_pops;
    p.use_idx = (const int32_t *)c->d_use.p; p.thresholds = (const double *)c->d_thr.p;
    p.n_use = (int)c->use_idx.size();
    p.chunk_tiles = c->chunk_tiles; p.n_chunks = c->n_chunks;
    p.chunk_used_kept_off = (const int64_t *)c->d_uk_off.p;
    p.size = c->snp_size; p.step = c->snp_step;
    p.tn_max = 0;
    int max_n_use = 0;
    for (int32_t k : c->use_idx) { p.tn_max += c->n_full[k]; max_n_use = std::max(max_n_use, c->n_full[k]); }
    { int lo, hi; global_rcp_split(p.tn_max, max_n_use, &lo, &hi); p.rcp_lo = lo; p.rcp_hi = hi; }
    p.base = (const int64_t *)c->d_base.p;
    p.halo_n = c->halo_n;
    p.n_full = (const int32_t *)c->d_nfull.p;
    p.halo_num = (const double *)c->d_halo_num.p; p.halo_den = (const double *)c->d_halo_den.p;
    p.halo_tag = (const int32_t *)c->d_halo_tag.p;
    p.row_cap = c->row_cap;
    p.row_start = (int32_t *)c->d_row_start.p; p.row_end = (int32_t *)c->d_row_end.p;
    p.row_num = (double *)c->d_row_num.p; p.row_den = (double *)c->d_row_den.p;
    p.chunk_row_off = (const int64_t *)c->d_row_off.p;
    return p;
}

template <typename T>
int d2h(evofst_ctx *c, T *dst, const void *src, size_t n) {
    if (!dst || n == 0) return EVOFST_OK;
    HIPCHK(c, hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyDeviceToHost, c->stream));
    return EVOFST_OK;
}


// (Re)allocates a buffer whose words a kernel polls: new memory is zeroed (a tag of a past epoch is as good as zero).
int ensure_zeroed(evofst_ctx *c, DevBuf &b, size_t bytes) {
    if (bytes == 0) bytes = 16;
    if (b.bytes >= bytes) return EVOFST_OK;
    int rc = ensure(c, b, bytes);
    if (rc) return rc;
    HIPCHK(c, hipMemsetAsync(b.p, 0, b.bytes, c->stream));
    return EVOFST_OK;
}

constexpr int kFusedFan = 32, kFusedMaxChunks = 32 * 32 * 64;

// Tiles per chunk of the fused pairs pass and sites per table fill, or 0 when this shape goes through K1 + scan + K2.
int fused_plan(evofst_ctx *c, int64_t n_sites, bool always, int *sub_out) {
    if (c->mode != MODE_PAIRS || c->pairs.size() > (size_t)kK2Threads || c->gathered) return 0;
    // Opt-in (evofst_set_option "pairs_pass" 3) and the pass behind evofst_run_pairs_noalloc: its workgroups move through
    // their phases in lock-step; evofst_run takes the stream pass of evofst_stream.hip instead.
    if (!always && c->opt_pairs_pass != 3) return 0;
    const int n_tiles = n_tiles_of(n_sites);
    const int np = (int)c->pairs.size();
    const int max_n = *std::max_element(c->n_full.begin(), c->n_full.end());
    const int unit = np <= 64 ? 4 : 1;                    // up to 64 pairs: every wave of the workgroup owns a quarter of a chunk
    int sub = 32;
#ifdef EVOFST_EXPERIMENTS
    if (const char *e = getenv("EVOFST_FUSED_SUB")) sub = atoi(e) == 16 ? 16 : 32;
#endif
    auto lds = [&](int ct) { return fused_lds_bytes(c->n_pops, (int)c->segs.size(), max_n, np, ct, c->count_bytes, sub); };
    int ct = c->opt_chunk_tiles;
    if (ct <= 0) ct = 4 * unit;
    ct = std::max(ct, (n_tiles + kFusedMaxChunks - 1) / kFusedMaxChunks);
    ct = (ct + unit - 1) / unit * unit;
    const size_t budget = 72 * 1024;                      // two workgroups per CU at the very least
    while (ct > unit && lds(ct) > budget && (n_tiles + (ct - unit) - 1) / (ct - unit) <= kFusedMaxChunks) ct -= unit;
    if (lds(ct) > 150 * 1024) return 0;
    *sub_out = sub;
    return ct;
}

int run_fused(evofst_ctx *c, int ct, int sub) {
    const size_t np = c->pairs.size();
    c->n_tiles = n_tiles_of(c->n_sites);
    c->chunk_tiles = ct;
    const int n_chunks = (c->n_tiles + ct - 1) / ct;
    c->n_chunks = fused_tail_chunks((int)np, n_chunks);   // entries of the tail table, which stitch_kernel / totals_kernel walk
    const int64_t g = c->g;
    c->seg_cap = c->n_sites / g + 2;
    int rc;
    if ((rc = ensure(c, c->d_incl_totals, np * 8)) || (rc = ensure(c, c->d_first_seg, np * 8))) return rc;
    if ((rc = ensure(c, c->d_seg_rows, (size_t)c->seg_cap * np * sizeof(SegRow)))) return rc;
    const size_t cq = (size_t)c->n_chunks * np;
    if ((rc = ensure(c, c->d_tail, cq * 3 * 8)) || (rc = ensure(c, c->d_tail_n, cq * 16))) return rc;
    const size_t n1 = (n_chunks + kFusedFan - 1) / kFusedFan, n2 = (n1 + kFusedFan - 1) / kFusedFan;
    if (c->epoch == 0xffffffffu) {     // the epoch counter wraps: forget every tag
        release(c->d_agg0); release(c->d_agg1); release(c->d_agg2);
        c->epoch = 0;
    }
    if ((rc = ensure_zeroed(c, c->d_agg0, (size_t)n_chunks * np * 8)) || (rc = ensure_zeroed(c, c->d_agg1, n1 * np * 8)) ||
        (rc = ensure_zeroed(c, c->d_agg2, n2 * np * 8)) || (rc = ensure_zeroed(c, c->d_fsync, 256))) return rc;
    FusedParams p{};
    p.tiles = (const ulonglong2 *)c->tiles;
    p.site_flags = c->have_flags ? c->flags : nullptr;
    p.n_sites = c->n_sites; p.n_tiles = c->n_tiles; p.W = c->W; p.P = c->n_pops;
    p.segs = (const SegDesc *)c->d_segs.p; p.n_segs = (int)c->segs.size();
    p.n_full = (const int32_t *)c->d_nfull.p;
    p.max_n = *std::max_element(c->n_full.begin(), c->n_full.end());
    p.pairs = (const uint32_t *)c->d_pairs.p; p.n_pairs = (int)np;
    p.chunk_tiles = ct; p.n_chunks = n_chunks;
    p.g = c->g;
    p.base = (const int64_t *)c->d_base.p;
    p.first_segment = (int64_t *)c->d_first_seg.p; p.incl_totals = (int64_t *)c->d_incl_totals.p;
    p.seg_cap = c->seg_cap; p.seg_rows = (SegRow *)c->d_seg_rows.p;
    p.tail = (double *)c->d_tail.p; p.tail_n = (int4 *)c->d_tail_n.p;
    p.agg0 = (uint64_t *)c->d_agg0.p; p.agg1 = (uint64_t *)c->d_agg1.p; p.agg2 = (uint64_t *)c->d_agg2.p;
    p.ticket = (uint32_t *)c->d_fsync.p; p.err = (uint32_t *)c->d_fsync.p + 32;
    p.epoch = ++c->epoch;
    int max_blocks = 0;
#ifdef EVOFST_EXPERIMENTS
    if (const char *e = getenv("EVOFST_FUSED_BLOCKS")) max_blocks = atoi(e);
    if (const char *e = getenv("EVOFST_DEBUG_FUSED")) p.debug = atoi(e);
#endif
    record(c, 4, 0);
    HIPCHK(c, launch_fused(p, c->count_bytes, sub, max_blocks, c->stream));
    record(c, 4, 1);
    K2Params k2 = make_k2(c);
    record(c, 3, 0);
    HIPCHK(c, launch_stitch(k2, c->stream));
    record(c, 3, 1);
    c->check_word = (const uint32_t *)c->d_fsync.p + 32; c->check_dirty = false;
    c->check_zero[0] = c->d_fsync.p; c->check_zero_bytes[0] = 256;
    c->check_zero[1] = c->check_zero[2] = nullptr; c->check_zero_bytes[1] = c->check_zero_bytes[2] = 0;
    c->offsets_dirty = false;
    c->counts_done = false;          // K1's census / masks are not there (plain_counts says whether its counts are)
    c->stats_done = true;
    return EVOFST_OK;
}

// After the stream has been waited for: did a wait inside the last one-kernel pass give up?  (Bounded spins raise an
// error word instead of hanging; the pass then leaves rows and tails unwritten.)
int fused_verdict(evofst_ctx *c) {
    if (!c->check_word) return EVOFST_OK;
    uint32_t w[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // [0] error word; [8] (stream pass) chunks left to the general walk, all launches so far
    HIPCHK(c, hipMemcpyAsync(w, c->check_word, c->check_dirty ? sizeof(w) : 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->check_word = nullptr;
    const uint32_t err = w[0];
    if (c->check_dirty && !err) {
        // marked chunks per launch of the passes since the last look (rounded up)
        const uint32_t n = std::max(1u, c->stream_launches);
        c->stream_dirty_last = (w[8] - c->stream_dirty_cum + n - 1) / n; c->stream_dirty_cum = w[8]; c->stream_dirty_known = true;
        c->stream_launches = 0;
    }
    if (err) {
        c->stream_dirty_cum = 0; c->stream_dirty_known = false; c->stream_launches = 0;
        for (int i = 0; i < 3; ++i)
            if (c->check_zero[i]) HIPCHK(c, hipMemsetAsync(c->check_zero[i], 0, c->check_zero_bytes[i], c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->stats_done = false;
        return fail(c, EVOFST_ERR_HIP, "one-kernel pairs pass: a look-back wait gave up (results discarded)");
    }
    return EVOFST_OK;
}

int launch_k1_counts(evofst_ctx *c, int mode);

// Getters that need the per-site counts: K1 runs on demand (the fused pairs pass keeps no counts in HBM).
int need_counts(evofst_ctx *c, const char *who) {
    if (c->plain_counts) return EVOFST_OK;
    if (c->n_sites == 0 || !c->tiles) return fail(c, EVOFST_ERR_STATE, std::string(who) + ": no data");
    if (c->compact_counts) c->prefer_plain = true;   // this caller reads per-site values: later batches skip the one-byte form
    return launch_k1_counts(c, MODE_COUNTS);
}

// The walk of run_stats that takes K1's clean chunks (clean_walk_kernel): the condition K1 and run_stats share.
bool clean_walk_applies(const evofst_ctx *c) {
    const size_t np = c->pairs.size();
    if (c->mode != MODE_PAIRS || c->count_bytes != 2 || np == 0 || np > (size_t)kK2Threads) return false;
    if (c->opt_walk != 0) return false;
    const int piece_tiles = std::max(1, std::min(2, c->chunk_tiles));
    const int max_n = *std::max_element(c->n_full.begin(), c->n_full.end());
    if (fused_lds_bytes(c->n_pops, 0, max_n, (int)np, 2 * piece_tiles * (np <= 64 ? 4 : 1), c->count_bytes, 32) > 100 * 1024) return false;
    return clean_walk_lds_bytes(c->n_pops, (int)np, piece_tiles, false) <= 64 * 1024;
}


// The statistics walk over counts, ballots and offsets that lie in HBM (K1 + scan, or the chunks the stream pass left to
// it).  Up to 256 pairs: the walk of evofst_fused.hip (regions in LDS, exact integer sums where no allele is missing);
// more pairs, or layouts its LDS regions cannot hold: k2_window_kernel.  marked_only: chunk_dirty says which chunks the
// general walk takes; with_clean: clean_walk_kernel takes the others (else somebody else has).
bool lds_walk_fits(const evofst_ctx *c, int chunk_tiles) {
    const size_t np = c->pairs.size();
    if (np == 0 || np > (size_t)kK2Threads || c->opt_walk == 2) return false;
    const int max_n = *std::max_element(c->n_full.begin(), c->n_full.end());
    const int piece_tiles = std::max(1, std::min(2, chunk_tiles));
    return fused_lds_bytes(c->n_pops, 0, max_n, (int)np, 2 * piece_tiles * (np <= 64 ? 4 : 1), c->count_bytes, 32) <= 100 * 1024;
}
int launch_walks(evofst_ctx *c, const K2Params &p, bool with_clean, bool compact, bool marked_only = false, int max_blocks = 0) {
    const size_t np = c->pairs.size();
    if (lds_walk_fits(c, c->chunk_tiles)) {
        FusedParams f{};
        f.site_flags = p.site_flags; f.n_sites = c->n_sites; f.n_tiles = c->n_tiles; f.W = c->W; f.P = c->n_pops;
        f.n_segs = 0; f.n_full = p.n_full; f.max_n = p.max_n; f.pairs = p.pairs; f.n_pairs = p.n_pairs;
        f.chunk_tiles = c->chunk_tiles; f.n_chunks = c->n_chunks; f.g = c->g;
        f.base = (const int64_t *)c->d_base.p;
        f.first_segment = (int64_t *)c->d_first_seg.p; f.incl_totals = (int64_t *)c->d_incl_totals.p;
        f.seg_cap = c->seg_cap; f.seg_rows = p.seg_rows; f.tail = p.tail; f.tail_n = p.tail_n;
        f.counts = p.counts; f.zf_masks = p.zf_masks; f.irr_mask = p.irr_mask; f.skip_mask = p.skip_mask; f.miss_mask = p.miss_mask;
        f.chunk_offsets = p.chunk_offsets;
        f.nan_mask = (const uint64_t *)c->d_nan.p;
        f.piece_tiles = std::max(1, std::min(2, c->chunk_tiles));     // two buffers of two tiles per region
        int wsub = 32;
#ifdef EVOFST_EXPERIMENTS
        if (const char *e = getenv("EVOFST_WALK_SUB")) wsub = atoi(e) == 16 ? 16 : 32;
        if (const char *e = getenv("EVOFST_DEBUG_WALK")) f.debug = atoi(e);
#endif
        // chunks without a skipped, irregular or missing-allele site (K1 marks the others) go through the lean
        // integer walk, the rest through the general one; each kernel leaves the other's chunks alone
        if (with_clean || marked_only) f.chunk_dirty = (const uint32_t *)c->d_dirty.p;
        if (with_clean) {
            f.compact = compact ? 1 : 0;
            HIPCHK(c, launch_clean_walk(f, c->stream));
            // (only for data the library owns: memory attached with evofst_attach_device may change between two runs)
            if (c->k1_marks_known && c->k1_marks_valid && c->tiles == c->d_tiles.p) {   // K1's marks of this very data are known
                if (c->k1_marks == 0) return EVOFST_OK;             // nothing for the general walk: no launch at all
                if (c->k1_marks < 128 && max_blocks == 0) max_blocks = 8 + 2 * (int)c->k1_marks;
            }
        }
        HIPCHK(c, launch_fused(f, c->count_bytes, wsub, max_blocks, c->stream));
        return EVOFST_OK;
    }
    if (marked_only) return fail(c, EVOFST_ERR_STATE, "internal: the stream pass was planned for a layout the LDS walk cannot take");
    (void)np;
    HIPCHK(c, launch_k2(p, c->count_bytes, c->stream));
    return EVOFST_OK;
}


// ---- the stream pairs pass (evofst_stream.hip): tiles -> segment rows in one launch of producer / consumer workgroups
constexpr int kStreamMaxChunks = 1 << 17;

// Shape of the stream pass's workgroups: producer waves, consumer waves, tiles a producer counts per chunk.
struct StreamPlan { int n_prod = 0, n_cons = 0, tr = 0, tr2 = 0, n_buf = 2, n_sets = 1; };
// tr == 0 when this shape goes through K1 + scan + walk.
StreamPlan stream_plan(const evofst_ctx *c, int64_t n_sites) {
    StreamPlan pl;
    if (c->mode != MODE_PAIRS || c->count_bytes != 2 || c->n_pops > 64 || c->pairs.empty() || c->pairs.size() > (size_t)kStreamMaxPairs ||
        c->gathered) return pl;
    if (c->opt_pairs_pass != 0 && c->opt_pairs_pass != 2) return pl;
    const int n_tiles = n_tiles_of(n_sites);
    const int np = (int)c->pairs.size();
    // Left to itself the library takes this pass where a tile row is long and the shard large: there K1's store stream and
    // the walk's re-read cost more than the look-back does (cfg4 shard: 1.57 against 1.74 ms); on short rows (cfg2:
    // 7 words) the three small kernels of the split pass are the faster ones (0.07 against 0.11 ms).
    const bool wide = c->W >= 32;
    if (c->opt_pairs_pass == 0 && (!wide || n_tiles < 16 * 1024)) return pl;
    // a consumer wave per 64 pairs; producers: two per SIMD on long rows, else four, or what the options say
    pl.n_cons = c->opt_stream_cons > 0 ? std::min(c->opt_stream_cons, (np + 63) / 64) : (np + 63) / 64;
    pl.n_prod = c->opt_stream_prod > 0 ? c->opt_stream_prod : (wide && pl.n_cons <= 4 ? 8 : 4);
    pl.n_sets = c->opt_stream_sets > 0 ? c->opt_stream_sets : 1;
    pl.n_cons *= pl.n_sets;
    if (pl.n_prod + pl.n_cons > kStreamMaxWaves) return StreamPlan();
    pl.n_buf = c->opt_stream_bufs > 0 ? c->opt_stream_bufs : 2;
    // as many workgroups per CU as 16 waves allow (two at the very least); LDS must not be what cuts that number
    // (a chunk size the options ask for may take the whole CU's LDS for one workgroup)
    const int wgs = c->opt_chunk_tiles > 0 ? 1 : std::max(pl.n_prod + pl.n_cons > 8 ? 1 : 2, 16 / (pl.n_prod + pl.n_cons));
    const size_t lds_cap = (size_t)(156 * 1024) / wgs;
    const int nf = std::min(pl.n_prod, 4), ns = pl.n_prod - nf;        // producers that are their SIMD's oldest wave, and the others
    auto chunk_of = [&](int a, int b) { return nf * a + ns * b; };
    auto fits = [&](int a, int b) {
        const int ct = chunk_of(a, b);
        return ct <= 64 && stream_lds_bytes(c->n_pops, np, ct, pl.n_buf) <= lds_cap && (n_tiles + ct - 1) / ct <= kStreamMaxChunks;
    };
    int tr = 0, tr2 = 0;
    if (c->opt_chunk_tiles > 0) {
        tr = tr2 = std::max(1, c->opt_chunk_tiles / pl.n_prod);
        if (!fits(tr, tr2)) return StreamPlan();
    } else {
        // Chunks as long as the LDS holds two of: every chunk costs a look-back and a tail, and a chunk is in flight from its
        // ticket to its last producer's arrival, which every later chunk's consumer waits for.  The second wave of a SIMD
        // counts about a fifth slower than the first (the hardware serves the oldest wave first), so it takes 3 tiles to
        // the other's 4 where 28-tile chunks fit: both kinds arrive together.
        static const int shapes[][2] = {{4, 3}, {3, 3}, {3, 2}, {2, 2}, {2, 1}, {1, 1}};
        for (const auto &sh : shapes)
            if ((ns > 0 || sh[0] == sh[1]) && fits(sh[0], sh[1])) { tr = sh[0]; tr2 = sh[1]; break; }
        if (tr == 0) return StreamPlan();
    }
    // the chunks the integer walk cannot take go to the walk of evofst_fused.hip over counts in HBM
    if (c->opt_walk == 2 || !lds_walk_fits(c, chunk_of(tr, tr2))) return StreamPlan();
    pl.tr = tr; pl.tr2 = tr2;
    return pl;
}

int run_stream(evofst_ctx *c, const StreamPlan &pl) {
    const int tr = pl.tr;
    const size_t np = c->pairs.size();
    c->n_tiles = n_tiles_of(c->n_sites);
    c->chunk_tiles = tr * std::min(pl.n_prod, 4) + pl.tr2 * std::max(pl.n_prod - 4, 0);
    c->n_chunks = (c->n_tiles + c->chunk_tiles - 1) / c->chunk_tiles;
    c->seg_cap = c->n_sites / c->g + 2;
    int rc;
    if ((rc = ensure(c, c->d_incl_totals, np * 8)) || (rc = ensure(c, c->d_first_seg, np * 8))) return rc;
    if ((rc = ensure(c, c->d_seg_rows, (size_t)c->seg_cap * np * sizeof(SegRow)))) return rc;
    const size_t cq = (size_t)c->n_chunks * np;
    if ((rc = ensure(c, c->d_tail, cq * 3 * 8)) || (rc = ensure(c, c->d_tail_n, cq * 16))) return rc;
    // what a chunk with a skipped site or a missing allele leaves for the general walk
    if ((rc = ensure(c, c->d_counts, (size_t)c->n_tiles * c->n_pops * kTileSites * c->count_bytes))) return rc;
    if ((rc = ensure(c, c->d_zf, (size_t)c->n_tiles * c->n_pops * 16))) return rc;
    if ((rc = ensure(c, c->d_irr, (size_t)c->n_tiles * 8)) || (rc = ensure(c, c->d_skip, (size_t)c->n_tiles * 8)) ||
        (rc = ensure(c, c->d_miss, (size_t)c->n_tiles * 8)) || (rc = ensure(c, c->d_nan, (size_t)c->n_tiles * 8))) return rc;
    if ((rc = ensure(c, c->d_dirty, (size_t)c->n_chunks * 4)) || (rc = ensure(c, c->d_offsets, cq * 8))) return rc;
    // granule tables: rows of level 0, 1, 2 one after the other, two copies
    const int64_t n1 = (c->n_chunks + kStreamFan - 1) / kStreamFan, n2 = (n1 + kStreamFan - 1) / kStreamFan;
    const int64_t rows = (int64_t)c->n_chunks + n1 + n2;
    auto a256 = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t b0 = a256((size_t)((c->n_chunks + 1) / 2) * np * 4), b1 = a256((size_t)n1 * np * 4), b2 = a256((size_t)n2 * np * 4);
    const size_t copy_bytes = b0 + b1 + b2, flag_bytes = a256((size_t)rows * 4);
    if ((rc = ensure_zeroed(c, c->d_ssync, 256))) return rc;
    const bool fresh = c->d_sflag.bytes < 2 * flag_bytes || c->d_sagg.bytes < 2 * copy_bytes;
    if ((rc = ensure(c, c->d_sagg, 2 * copy_bytes)) || (rc = ensure(c, c->d_sflag, 2 * flag_bytes))) return rc;
    if (fresh || c->sagg_rows != rows || c->sagg_pairs != (int64_t)np) {
        // another shape than the last launch zeroed the rows and flags for (or none yet): zero both copies
        HIPCHK(c, hipMemsetAsync(c->d_sagg.p, 0, c->d_sagg.bytes, c->stream));
        HIPCHK(c, hipMemsetAsync(c->d_sflag.p, 0, c->d_sflag.bytes, c->stream));
        c->sagg_rows = rows; c->sagg_pairs = (int64_t)np;
    }
    StreamParams p{};
    p.tiles = (const ulonglong2 *)c->tiles;
    p.site_flags = c->have_flags ? c->flags : nullptr;
    p.n_sites = c->n_sites; p.n_tiles = c->n_tiles; p.W = c->W; p.P = c->n_pops;
    p.segs = (const SegDesc *)c->d_segs.p;
    p.n_full = (const int32_t *)c->d_nfull.p;
    p.pairs = (const uint32_t *)c->d_pairs.p; p.n_pairs = (int)np;
    p.n_prod = pl.n_prod; p.n_cons = pl.n_cons;
    p.poll = 8; p.n_buf = pl.n_buf; p.n_sets = pl.n_sets;
#ifdef EVOFST_EXPERIMENTS
    if (const char *e = getenv("EVOFST_STREAM_POLL")) p.poll = std::max(0, atoi(e));
#endif
    p.tr = tr; p.tr2 = pl.tr2; p.chunk_tiles = c->chunk_tiles; p.n_chunks = c->n_chunks;
    p.g = c->g;
    p.base = (const int64_t *)c->d_base.p;
    p.first_segment = (int64_t *)c->d_first_seg.p; p.incl_totals = (int64_t *)c->d_incl_totals.p;
    p.seg_cap = c->seg_cap;
    p.seg_rows = (SegRow *)c->d_seg_rows.p;
    p.tail = (double *)c->d_tail.p; p.tail_n = (int4 *)c->d_tail_n.p;
    for (int k = 0; k < 2; ++k) {
        unsigned char *base = (unsigned char *)c->d_sagg.p + (size_t)k * copy_bytes;
        p.agg0[k] = (uint32_t *)base; p.agg1[k] = (uint32_t *)(base + b0); p.agg2[k] = (uint32_t *)(base + b0 + b1);
    }
    p.flags[0] = (uint32_t *)c->d_sflag.p; p.flags[1] = (uint32_t *)((unsigned char *)c->d_sflag.p + flag_bytes);
    p.l1_row = c->n_chunks; p.l2_row = c->n_chunks + (int)n1;
    p.ticket = (uint32_t *)c->d_ssync.p; p.err = (uint32_t *)c->d_ssync.p + 32; p.n_dirty = (uint32_t *)c->d_ssync.p + 40; p.launches = (uint32_t *)c->d_ssync.p + 48;
    p.counts = c->d_counts.p;
    p.zf_masks = (uint64_t *)c->d_zf.p; p.irr_mask = (uint64_t *)c->d_irr.p; p.skip_mask = (uint64_t *)c->d_skip.p;
    p.miss_mask = (uint64_t *)c->d_miss.p; p.nan_mask = (uint64_t *)c->d_nan.p;
    p.chunk_offsets = (int64_t *)c->d_offsets.p; p.chunk_dirty = (uint32_t *)c->d_dirty.p;
    c->k1_marks_valid = c->k1_marks_known = false;      // the marks are this pass's now (its own chunks)
#ifdef EVOFST_EXPERIMENTS
    if (const char *e = getenv("EVOFST_DEBUG_STREAM")) p.debug = atoi(e);
    // in-kernel time stamps of one launch (the EVOFST_STREAM_STAMPS_AT-th of this context), written to the file named
    static thread_local int stamp_launch = 0;
    const char *stamp_file = getenv("EVOFST_STREAM_STAMPS");
    const int stamp_at = getenv("EVOFST_STREAM_STAMPS_AT") ? atoi(getenv("EVOFST_STREAM_STAMPS_AT")) : 5;
    const size_t stamp_bytes = (size_t)4096 * kStreamStampIters * 16 * 8;
    const bool stamping = stamp_file && stamp_launch++ == stamp_at;
    if (stamping) {
        if ((rc = ensure(c, c->d_gnum, stamp_bytes))) return rc;     // a buffer of the Gram getters, idle in pairs mode
        HIPCHK(c, hipMemsetAsync(c->d_gnum.p, 0, stamp_bytes, c->stream));
        p.stamps = (unsigned long long *)c->d_gnum.p;
    }
#endif
    record(c, 4, 0);
    HIPCHK(c, launch_stream(p, c->opt_stream_blocks, c->stream));
    record(c, 4, 1);
#ifdef EVOFST_EXPERIMENTS
    if (stamping) {
        std::vector<unsigned long long> h(stamp_bytes / 8);
        HIPCHK(c, hipMemcpyAsync(h.data(), c->d_gnum.p, stamp_bytes, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (FILE *fp = fopen(stamp_file, "wb")) { fwrite(h.data(), 1, stamp_bytes, fp); fclose(fp); }
    }
#endif
    K2Params k2 = make_k2(c);
#ifdef EVOFST_EXPERIMENTS
    if (p.debug) return EVOFST_OK;   // timing switches: rows, tails and marks are not all there, nothing may read them
#endif
    record(c, 2, 0);
    // the chunks the stream pass marked, if any: a launch that finds none costs 40 us with a grid that fills the device and
    // a few with a small one, so the grid follows what the passes before this one marked (known at evofst_sync)
    // (a shard whose last tile is short always marks its last chunk: a workgroup or two per marked chunk, up to the device)
    ++c->stream_launches;
    const int walk_blocks = c->stream_dirty_known && c->stream_dirty_last < 128 ? 8 + 2 * (int)c->stream_dirty_last : 0;
    // (nothing marked by the passes before this one over the same data in the library's own memory -- which only
    // evofst_upload / evofst_synth change, and they reset what is known: nothing to launch; attached memory may have changed)
    const bool none_marked = c->stream_dirty_known && c->stream_dirty_last == 0 && c->tiles == c->d_tiles.p;
    if (!none_marked && (rc = launch_walks(c, k2, false, false, true, walk_blocks))) return rc;
    record(c, 2, 1);
    record(c, 3, 0);
    HIPCHK(c, launch_stitch(k2, c->stream));
    record(c, 3, 1);
    c->check_word = (const uint32_t *)c->d_ssync.p + 32; c->check_dirty = true;
    c->check_zero[0] = c->d_ssync.p; c->check_zero_bytes[0] = 192;           // ticket, error and marked-chunk words; the launch counter stays
    c->check_zero[1] = c->d_sflag.p; c->check_zero_bytes[1] = c->d_sflag.bytes;
    c->check_zero[2] = c->d_sagg.p; c->check_zero_bytes[2] = c->d_sagg.bytes;
    c->offsets_dirty = false;
    c->counts_done = false;          // K1's census is not there, and its counts only for the marked chunks
    c->plain_counts = false; c->compact_counts = false;
    c->stats_done = true;
    return EVOFST_OK;
}

}  // namespace

// ================================================================== lifecycle

extern "C" int evofst_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess && n > 0 ? n : 0;
}

extern "C" evofst_ctx *evofst_create(int device_id) {
    if (device_id == -1) {
        evofst_ctx *h = new (std::nothrow) evofst_ctx();
        if (!h) { g_create_error = "out of host memory"; return nullptr; }
        h->host_only = true; h->device = -1; h->own_stream = false;
        return h;
    }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_create_error = std::string("no HIP device available: ") + hipGetErrorString(e);
        return nullptr;
    }
    if (device_id < 0 || device_id >= n) { g_create_error = "device_id out of range"; return nullptr; }
    if ((e = hipSetDevice(device_id)) != hipSuccess) {
        g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e);
        return nullptr;
    }
    if (device_id < kMaxDevices) {
        // this GPU's copy of the no-allocation scratch array (evofst_run_pairs_noalloc); two threads may create contexts at once
        std::lock_guard<std::mutex> scratch_lock(g_scratch_mu[device_id]);
        void *sp = g_scratch_ptr[device_id];
        if (!sp) {
        if ((e = hipGetSymbolAddress(&sp, HIP_SYMBOL(scratch))) != hipSuccess) {
            g_create_error = std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e);
            return nullptr;
        }
        g_scratch_ptr[device_id] = sp;
        }
    }
    evofst_ctx *c = new (std::nothrow) evofst_ctx();
    if (!c) { g_create_error = "out of host memory"; return nullptr; }
    c->device = device_id;
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) {
        g_create_error = std::string("hipStreamCreate: ") + hipGetErrorString(e);
        delete c;
        return nullptr;
    }
    return c;
}

extern "C" void evofst_destroy(evofst_ctx *c) {
    if (!c) return;
    if (c->host_only) { delete c; return; }
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    DevBuf *bufs[] = {&c->d_segs, &c->d_wss, &c->d_nfull, &c->d_pairs, &c->d_use, &c->d_thr, &c->d_base, &c->d_first_seg,
                      &c->d_tiles, &c->d_flags, &c->d_pos, &c->d_counts, &c->d_census, &c->d_offsets, &c->d_incl_totals,
                      &c->d_seg_rows, &c->d_seg_num, &c->d_seg_den, &c->d_seg_k, &c->d_seg_end, &c->d_zf, &c->d_irr,
                      &c->d_tail, &c->d_tail_n, &c->d_win, &c->d_halo_num, &c->d_halo_den, &c->d_halo_tag, &c->d_gU, &c->d_gC, &c->d_gsub, &c->d_gsuboff, &c->d_gnum, &c->d_gden, &c->d_gk, &c->d_gcnt, &c->d_graw, &c->d_carry,
                      &c->d_nrows, &c->d_totals, &c->d_gtotals, &c->d_kept, &c->d_skip, &c->d_used_kept, &c->d_uk_off,
                      &c->d_uk_tot, &c->d_chunk_rows, &c->d_row_off, &c->d_row_tot, &c->d_row_start, &c->d_row_end,
                      &c->d_row_num, &c->d_row_den, &c->d_planes, &c->d_multi_site, &c->d_multi_alleles, &c->d_agg0, &c->d_agg1,
                      &c->d_agg2, &c->d_fsync, &c->d_miss, &c->d_dirty, &c->d_poison, &c->d_nan, &c->d_sagg, &c->d_sflag, &c->d_ssync};
    for (DevBuf *b : bufs) release(*b);
    for (auto &pair : c->ev)
        for (auto &e : pair)
            if (e) (void)hipEventDestroy(e);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" const char *evofst_last_error(const evofst_ctx *c) { return c ? c->err.c_str() : g_create_error.c_str(); }
extern "C" const char *evofst_version(void) { return "evofst 0.1 (gfx950)"; }

extern "C" int evofst_set_stream(evofst_ctx *c, void *hip_stream) try {
    if (!c) return EVOFST_ERR_ARG;
    NEED_GPU(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    c->stream = (hipStream_t)hip_stream;
    c->own_stream = false;
    return EVOFST_OK;
} catch (...) {
    return abi_exception(c);
}

// ================================================================== layout

extern "C" int evofst_set_layout(evofst_ctx *c, int n_samples, int n_pops, const int32_t *sample_pop) try {
    if (!c) return EVOFST_ERR_ARG;
    if (n_samples <= 0 || n_pops <= 0 || !sample_pop) return fail(c, EVOFST_ERR_ARG, "set_layout: bad arguments");
    if (n_pops > EVOFST_MAX_POPS) return fail(c, EVOFST_ERR_UNSUPPORTED, "set_layout: too many populations");
    c->prefer_plain = false;          // a new job: K1 may leave one-byte counts until a getter asks for per-site values
    std::vector<int32_t> size(n_pops, 0);
    for (int i = 0; i < n_samples; ++i) {
        if (sample_pop[i] >= n_pops) return fail(c, EVOFST_ERR_ARG, "set_layout: population index out of range");
        if (sample_pop[i] >= 0) size[sample_pop[i]]++;
    }
    for (int k = 0; k < n_pops; ++k) {
        if (size[k] == 0) return fail(c, EVOFST_ERR_ARG, "set_layout: population without samples");
        if (2 * size[k] > EVOFST_MAX_POP_CHROMS) return fail(c, EVOFST_ERR_UNSUPPORTED, "set_layout: population too large");
    }
    if (!c->host_only) HIPCHK(c, hipSetDevice(c->device));
    c->n_samples = n_samples; c->n_pops = n_pops;
    c->sample_pop.assign(sample_pop, sample_pop + n_samples);
    c->n_full.resize(n_pops); c->pop_off.resize(n_pops + 1);
    c->pop_off[0] = 0;
    int max_n = 0;
    for (int k = 0; k < n_pops; ++k) {
        c->n_full[k] = 2 * size[k];
        c->pop_off[k + 1] = c->pop_off[k] + c->n_full[k];
        max_n = std::max(max_n, c->n_full[k]);
    }
    c->H = c->pop_off[n_pops];
    c->W = (c->H + 63) / 64;
    c->count_bytes = max_n <= 255 ? 2 : 4;
    // stable permutation: samples ordered by population, input order inside a population
    c->src_hap.assign(c->H, 0);
    std::vector<int32_t> fill(c->pop_off.begin(), c->pop_off.end() - 1);
    for (int i = 0; i < n_samples; ++i) {
        int k = sample_pop[i];
        if (k < 0) continue;
        c->src_hap[fill[k]] = 2 * i; c->src_hap[fill[k] + 1] = 2 * i + 1;
        fill[k] += 2;
    }
    // (word, pop) overlap table
    c->segs.clear(); c->word_seg_start.assign(c->W + 1, 0);
    for (int w = 0; w < c->W; ++w) {
        c->word_seg_start[w] = (int)c->segs.size();
        const int b0 = w * 64, b1 = b0 + 64;
        for (int k = 0; k < n_pops; ++k) {
            const int lo = std::max(b0, c->pop_off[k]), hi = std::min(b1, c->pop_off[k + 1]);
            if (lo >= hi) continue;
            SegDesc d{};
            const int len = hi - lo;
            d.mask = (len == 64 ? ~0ull : ((1ull << len) - 1ull)) << (lo - b0);
            d.pop = (uint32_t)k;
            d.flags = (hi == c->pop_off[k + 1]) ? kSegLast : 0u;
            c->segs.push_back(d);
        }
        if ((int)c->segs.size() > c->word_seg_start[w]) c->segs.back().flags |= kSegWordEnd;
    }
    c->word_seg_start[c->W] = (int)c->segs.size();
    c->mode = MODE_COUNTS;
    c->pairs.clear(); c->use_idx.clear(); c->thresholds.clear();
    c->counts_done = c->stats_done = false;
    c->plain_counts = false; c->compact_counts = false;
    c->cap_sites = 0; c->n_sites = 0;
    if (c->host_only) return EVOFST_OK;
    return upload_layout(c);
} catch (...) {
    return abi_exception(c);
}

extern "C" int evofst_get_geometry(const evofst_ctx *c, evofst_geometry *out) try {
    if (!c || !out) return EVOFST_ERR_ARG;
    if (c->n_pops == 0) return EVOFST_ERR_STATE;
    out->n_samples = c->n_samples; out->n_pops = c->n_pops; out->n_haplotypes = c->H;
    out->words_per_site = c->W; out->tile_sites = kTileSites; out->count_bytes = c->count_bytes;
    out->tile_bytes = (int64_t)c->W * kTileSites * 16;
    return EVOFST_OK;
} catch (...) {
    return abi_exception(nullptr);
}

extern "C" int evofst_get_pop_chroms(const evofst_ctx *c, int32_t *n_full) try {
    if (!c || !n_full) return EVOFST_ERR_ARG;
    std::copy(c->n_full.begin(), c->n_full.end(), n_full);
    return EVOFST_OK;
} catch (...) {
    return abi_exception(nullptr);
}

// ================================================================== host packer (K0)

extern "C" size_t evofst_tiles_bytes(const evofst_ctx *c, int64_t n_sites) {
    if (!c || c->n_pops == 0 || n_sites < 0) return 0;
    return (size_t)n_tiles_of(n_sites) * c->W * kTileSites * 16;
}

extern "C" int evofst_pack_codes(const evofst_ctx *c, int64_t first_site, int64_t n_sites, const uint8_t *codes,
                                 void *tiles) try {
    if (!c || !codes || !tiles || first_site < 0 || n_sites < 0) return EVOFST_ERR_ARG;
    if (c->n_pops == 0) return EVOFST_ERR_STATE;
    uint64_t *out = (uint64_t *)tiles;
    const size_t row = (size_t)2 * c->n_samples;
    const int W = c->W, H = c->H;
    for (int64_t s = 0; s < n_sites; ++s) {
        const uint8_t *r = codes + (size_t)s * row;
        const int64_t site = first_site + s;
        uint64_t *tp = out + ((size_t)(site >> 6) * W * kTileSites + (size_t)(site & 63)) * 2;
        for (int w = 0; w < W; ++w) {
            uint64_t alt = 0, miss = 0;
            const int b1 = std::min(64, H - w * 64);
            const int32_t *src = c->src_hap.data() + (size_t)w * 64;
            for (int b = 0; b < b1; ++b) {
                const uint8_t code = r[src[b]];
                alt |= (uint64_t)(code == EVOFST_ALT) << b;
                miss |= (uint64_t)(code == EVOFST_MISSING) << b;
            }
            tp[(size_t)w * kTileSites * 2] = alt;
            tp[(size_t)w * kTileSites * 2 + 1] = miss;
        }
    }
    // zero the padding sites of the last tile
    const int64_t end = first_site + n_sites;
    for (int64_t site = end; (site & 63) != 0; ++site) {
        uint64_t *tp = out + ((size_t)(site >> 6) * W * kTileSites + (size_t)(site & 63)) * 2;
        for (int w = 0; w < W; ++w) { tp[(size_t)w * kTileSites * 2] = 0; tp[(size_t)w * kTileSites * 2 + 1] = 0; }
    }
    return EVOFST_OK;
} catch (...) {
    return abi_exception(nullptr);
}

extern "C" int evofst_unpack_codes(const evofst_ctx *c, int64_t first_site, int64_t n_sites, const void *tiles,
                                   uint8_t *codes) try {
    if (!c || !codes || !tiles || first_site < 0 || n_sites < 0) return EVOFST_ERR_ARG;
    if (c->n_pops == 0) return EVOFST_ERR_STATE;
    const uint64_t *in = (const uint64_t *)tiles;
    const size_t row = (size_t)2 * c->n_samples;
    for (int64_t s = 0; s < n_sites; ++s) {
        uint8_t *r = codes + (size_t)s * row;
        std::memset(r, EVOFST_REF, row);
        const int64_t site = first_site + s;
        const uint64_t *tp = in + ((size_t)(site >> 6) * c->W * kTileSites + (size_t)(site & 63)) * 2;
        for (int h = 0; h < c->H; ++h) {
            const int w = h >> 6, b = h & 63;
            const uint64_t alt = tp[(size_t)w * kTileSites * 2], miss = tp[(size_t)w * kTileSites * 2 + 1];
            uint8_t code = EVOFST_REF;
            if ((miss >> b) & 1ull) code = EVOFST_MISSING;
            else if ((alt >> b) & 1ull) code = EVOFST_ALT;
            r[c->src_hap[h]] = code;
        }
    }
    return EVOFST_OK;
} catch (...) {
    return abi_exception(nullptr);
}

namespace {

// ALT column as `evo stats` reads it (getThisVariantCountsSimple, process_vcf_utils.cpp:365-381): returns the allele
// number that stands for "missing" (-1: none) and adds SKIP | MULTI to *fl for a multi-allelic site.  split() drops a
// trailing empty field and yields no field for an empty string (process_vcf_utils.cpp:104-110).
int stats_missing_allele(const char *alt, const char *alt_end, uint32_t *fl) {
    int n_alt = 0;
    const char *last = alt;
    size_t last_len = 0;
    const char *a = alt;
    if (a != alt_end) {
        for (;;) {
            const char *e = (const char *)std::memchr(a, ',', (size_t)(alt_end - a));
            ++n_alt; last = a; last_len = e ? (size_t)(e - a) : (size_t)(alt_end - a);
            if (!e) break;
            a = e + 1;
            if (a == alt_end) break;
        }
    }
    int miss_allele = -1;
    bool multi = false;
    if (n_alt > 1) {
        if (last_len == 1 && last[0] == '*') { miss_allele = n_alt; multi = (n_alt - 1) > 1; }
        else multi = true;
    }
    if (multi) *fl |= EVOFST_SITE_SKIP | EVOFST_SITE_MULTI;
    return miss_allele;
}

// One VCF data line -> allele codes of its n_samples genotype columns + site flags, without touching the text.
// cols[0..1] receive the starts of CHROM and POS, col_len their lengths.
int parse_vcf_line(const char *line, int mode, int n_samples, uint8_t *codes, uint32_t *site_flags,
                   const char **cols, size_t *col_len) {
    // columns: CHROM POS ID REF ALT QUAL FILTER INFO FORMAT samples...
    const char *col[9];
    size_t len[9];
    const char *p = line;
    for (int i = 0; i < 9; ++i) {
        col[i] = p;
        const char *t = std::strchr(p, '\t');
        if (!t) return EVOFST_ERR_ARG;
        len[i] = (size_t)(t - p);
        p = t + 1;
    }
    if (cols) { cols[0] = col[0]; cols[1] = col[1]; col_len[0] = len[0]; col_len[1] = len[1]; }
    const size_t ref_len = len[3], alt_len = len[4];
    const char *alt = col[4], *alt_end = col[4] + alt_len;
    uint32_t fl = 0;
    int alt_code = 1, allele_as_missing = -1;
    if (mode == EVOFST_MODE_FST) {
        // ALT parse of getVariantCountsForFst (process_vcf_fst.cpp:222-238); split() drops a trailing
        // empty field and yields no field for an empty string (process_vcf_utils.cpp:104-110)
        int n_alt = 0, idx = 0;
        bool indel = false;
        const char *a = alt;
        if (a != alt_end) {
            for (;;) {
                const char *e = (const char *)std::memchr(a, ',', (size_t)(alt_end - a));
                const size_t l = e ? (size_t)(e - a) : (size_t)(alt_end - a);
                ++idx; ++n_alt;
                if (l == 1 && a[0] == '*') { --n_alt; allele_as_missing = idx; }
                if (l > 1) indel = true;
                if (!e) break;
                a = e + 1;
                if (a == alt_end) break;
            }
        }
        if (n_alt == 1 && allele_as_missing == 1) alt_code = 2;
        if (ref_len > 1 || n_alt != 1 || indel) fl |= EVOFST_SITE_SKIP;  // :529-532, :543
    } else if (mode == EVOFST_MODE_STATS) {
        // ALT parse of getThisVariantCountsSimple (process_vcf_utils.cpp:365-381): a '*' that comes last among
        // several ALT alleles is a missing allele; more than one real ALT allele makes the site multi-allelic
        const int miss_allele = stats_missing_allele(alt, alt_end, &fl);
        allele_as_missing = miss_allele;
    } else if (mode == EVOFST_MODE_GLOBAL_INDELS) {
        if (alt_len == 1 && alt[0] == '*') fl |= EVOFST_SITE_SKIP;        // evo_PBS.cpp:248-256 with -i
    } else {
        if (ref_len > 1 || alt_len > 1 || (alt_len == 1 && alt[0] == '*'))
            fl |= EVOFST_SITE_SKIP;                                       // FstAgainstAll.cpp:185
    }
    *site_flags = fl;
    if (fl & EVOFST_SITE_SKIP) {
        // the reference never looks at the genotype columns of such a line
        std::memset(codes, EVOFST_MISSING, (size_t)2 * n_samples);
        return EVOFST_OK;
    }
    for (int i = 0; i < n_samples; ++i) {
        // genotype column i: chars [0] and [2] (process_vcf_fst.cpp:244, process_vcf_utils.cpp:67,75)
        if (p[0] == '\0' || p[0] == '\t' || p[1] == '\0' || p[1] == '\t' || p[2] == '\0' || p[2] == '\t')
            return EVOFST_ERR_ARG;
        for (int h = 0; h < 2; ++h) {
            const char ch = p[2 * h];
            uint8_t code;
            if (mode == EVOFST_MODE_FST) {
                const int v = ch - '0';
                if (v == alt_code) code = EVOFST_ALT;
                else if (ch == '.' || v == allele_as_missing) code = EVOFST_MISSING;
                else code = EVOFST_REF;
            } else if (mode == EVOFST_MODE_STATS) {
                // utils.cpp:386-405: either allele '.' or the missing allele -> the individual is missing; '1' counts
                const char c0 = p[0], c2 = p[2];
                if (c0 == '.' || c2 == '.' || c0 - '0' == allele_as_missing || c2 - '0' == allele_as_missing) code = EVOFST_MISSING;
                else code = ch == '1' ? EVOFST_ALT : EVOFST_REF;
            } else {
                code = ch == '1' ? EVOFST_ALT : (ch == '0' ? EVOFST_REF : EVOFST_MISSING);
            }
            codes[2 * i + h] = code;
        }
        // to the next column: genotype fields are a dozen characters, so a plain scan beats a library call
        const char *t = p + 3;
        while (*t != '\t' && *t != '\0') ++t;
        if (*t == '\0') {
            if (i != n_samples - 1) return EVOFST_ERR_ARG;
            break;
        }
        p = t + 1;
    }
    return EVOFST_OK;
}

}  // namespace

extern "C" int evofst_codes_from_vcf_line(char *line, int mode, int n_samples, uint8_t *codes, uint32_t *site_flags,
                                          const char **chrom, const char **pos) try {
    if (!line || !codes || !site_flags || n_samples <= 0) return EVOFST_ERR_ARG;
    const char *cols[2];
    size_t len[2];
    const int rc = parse_vcf_line(line, mode, n_samples, codes, site_flags, cols, len);
    if (rc != EVOFST_OK) return rc;
    // CHROM and POS are handed back as C strings inside the caller's buffer
    line[(cols[0] - line) + len[0]] = '\0';
    line[(cols[1] - line) + len[1]] = '\0';
    if (chrom) *chrom = cols[0];
    if (pos) *pos = cols[1];
    return EVOFST_OK;
} catch (...) {
    return abi_exception(nullptr);
}

// K0 in bulk: parse + pack a block of VCF lines on n_threads host threads (a thread owns whole tiles of 64 sites).
extern "C" int evofst_pack_vcf_lines(const evofst_ctx *c, int mode, int64_t n_lines, const char *const *lines,
                                     int n_vcf_samples, const int32_t *layout_to_vcf, int n_threads, void *tiles,
                                     uint8_t *site_flags) try {
    if (!c || !lines || !tiles || !site_flags || n_lines < 0 || n_vcf_samples <= 0) return EVOFST_ERR_ARG;
    if (c->n_pops == 0) return EVOFST_ERR_STATE;
    if (mode != EVOFST_MODE_FST && mode != EVOFST_MODE_GLOBAL && mode != EVOFST_MODE_STATS && mode != EVOFST_MODE_GLOBAL_INDELS) return EVOFST_ERR_ARG;
    // haplotype of the packed layout -> index into a line's code row
    std::vector<int32_t> src((size_t)c->H);
    for (int h = 0; h < c->H; ++h) {
        const int32_t lh = c->src_hap[h];                   // layout sample * 2 + hap
        const int32_t vs = layout_to_vcf ? layout_to_vcf[lh >> 1] : (lh >> 1);
        if (vs < 0 || vs >= n_vcf_samples) return EVOFST_ERR_ARG;
        src[h] = 2 * vs + (lh & 1);
    }
    const int64_t n_tiles = (n_lines + kTileSites - 1) / kTileSites;
    n_threads = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(n_threads, 64), n_tiles));
    std::atomic<int64_t> next{0};
    std::atomic<int> status{EVOFST_OK};
    uint64_t *out = (uint64_t *)tiles;
    const int W = c->W, H = c->H;
    auto work = [&]() noexcept {
        std::vector<uint8_t> codes;
        try { codes.resize((size_t)2 * n_vcf_samples); } catch (...) { status.store(EVOFST_ERR_NOMEM); return; }
        for (;;) {
            const int64_t t = next.fetch_add(1);
            if (t >= n_tiles || status.load() != EVOFST_OK) break;
            for (int s = 0; s < kTileSites; ++s) {
                const int64_t site = t * kTileSites + s;
                uint64_t *tp = out + ((size_t)t * W * kTileSites + (size_t)s) * 2;
                if (site >= n_lines) {                      // padding sites of the last tile
                    for (int w = 0; w < W; ++w) { tp[(size_t)w * kTileSites * 2] = 0; tp[(size_t)w * kTileSites * 2 + 1] = 0; }
                    continue;
                }
                uint32_t fl = 0;
                if (!lines[site] || parse_vcf_line(lines[site], mode, n_vcf_samples, codes.data(), &fl, nullptr, nullptr) != EVOFST_OK) {
                    status.store(EVOFST_ERR_ARG);
                    break;
                }
                site_flags[site] = (uint8_t)fl;
                for (int w = 0; w < W; ++w) {
                    uint64_t alt = 0, miss = 0;
                    const int b1 = std::min(64, H - w * 64);
                    const int32_t *sp = src.data() + (size_t)w * 64;
                    for (int b = 0; b < b1; ++b) {
                        const uint8_t code = codes[sp[b]];
                        alt |= (uint64_t)(code == EVOFST_ALT) << b;
                        miss |= (uint64_t)(code == EVOFST_MISSING) << b;
                    }
                    tp[(size_t)w * kTileSites * 2] = alt;
                    tp[(size_t)w * kTileSites * 2 + 1] = miss;
                }
            }
        }
    };
    std::vector<std::thread> th;
    th.reserve((size_t)n_threads);
    for (int i = 1; i < n_threads; ++i) {
        try { th.emplace_back(work); } catch (...) { break; }   // fewer threads than asked for: the work queue does not care
    }
    work();
    for (auto &x : th) x.join();
    return status.load();
} catch (...) {
    return abi_exception(nullptr);
}

// ================================================================== device data

extern "C" int evofst_reserve(evofst_ctx *c, int64_t max_sites) try {
    if (!c || max_sites <= 0) return EVOFST_ERR_ARG;
    NEED_GPU(c);
    if (c->n_pops == 0) return fail(c, EVOFST_ERR_STATE, "reserve: set_layout first");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t nt = (size_t)n_tiles_of(max_sites);
    int rc;
    if ((rc = ensure(c, c->d_tiles, nt * c->W * kTileSites * 16))) return rc;
    if ((rc = ensure(c, c->d_flags, nt * kTileSites))) return rc;
    if ((rc = ensure(c, c->d_pos, nt * kTileSites * 4))) return rc;
    if ((rc = ensure(c, c->d_counts, nt * c->n_pops * kTileSites * c->count_bytes))) return rc;
    c->cap_sites = (int64_t)nt * kTileSites;
    return EVOFST_OK;
} catch (...) {
    return abi_exception(c);
}

extern "C" void *evofst_alloc_pinned(evofst_ctx *c, size_t bytes) {
    void *p = nullptr;
    if (!c || c->host_only) return nullptr;
    (void)hipSetDevice(c->device);
    hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
    if (e != hipSuccess) { c->err = std::string("hipHostMalloc: ") + hipGetErrorString(e); return nullptr; }
    return p;
}
extern "C" void evofst_free_pinned(evofst_ctx *c, void *p) {
    if (c && p) (void)hipHostFree(p);
}

static int begin_data(evofst_ctx *c, int64_t n_sites) {
    if (n_sites <= 0) return fail(c, EVOFST_ERR_ARG, "n_sites must be positive");
    if (c->n_pops == 0) return fail(c, EVOFST_ERR_STATE, "set_layout first");
    if ((int64_t)n_tiles_of(n_sites) * kTileSites > (int64_t)INT_MAX) return fail(c, EVOFST_ERR_UNSUPPORTED, "too many sites for one shard");
    HIPCHK(c, hipSetDevice(c->device));
    c->counts_done = c->stats_done = false;
    c->plain_counts = false; c->compact_counts = false;
    c->offsets_dirty = true;
    c->n_multi = 0;
    c->stream_dirty_known = false;       // new data: what the passes before marked says nothing about it
    c->k1_marks_valid = c->k1_marks_known = false;
    return EVOFST_OK;
}

extern "C" int evofst_upload(evofst_ctx *c, int64_t n_sites, const void *tiles, const uint8_t *site_flags,
                             const int32_t *pos) try {
    if (!c || !tiles) return EVOFST_ERR_ARG;
    NEED_GPU(c);
    int rc = begin_data(c, n_sites);
    if (rc) return rc;
    if (n_sites > c->cap_sites && (rc = evofst_reserve(c, n_sites))) return rc;
    c->n_sites = n_sites;
    HIPCHK(c, hipMemcpyAsync(c->d_tiles.p, tiles, evofst_tiles_bytes(c, n_sites), hipMemcpyHostToDevice, c->stream));
    c->tiles = c->d_tiles.p;
    c->have_flags = site_flags != nullptr;
    if (site_flags) HIPCHK(c, hipMemcpyAsync(c->d_flags.p, site_flags, (size_t)n_sites, hipMemcpyHostToDevice, c->stream));
    c->flags = (const uint8_t *)c->d_flags.p;
    c->have_pos = pos != nullptr;
    if (pos) HIPCHK(c, hipMemcpyAsync(c->d_pos.p, pos, (size_t)n_sites * 4, hipMemcpyHostToDevice, c->stream));
    c->pos = (const int32_t *)c->d_pos.p;
    return EVOFST_OK;
} catch (...) {
    return abi_exception(c);
}

extern "C" int evofst_attach_device(evofst_ctx *c, int64_t n_sites, const void *d_tiles, const uint8_t *d_site_flags,
                                    const int32_t *d_pos) try {
    if (!c || !d_tiles) return EVOFST_ERR_ARG;
    NEED_GPU(c);
    int rc = begin_data(c, n_sites);
    if (rc) return rc;
    const size_t nt = (size_t)n_tiles_of(n_sites);
    if ((rc = ensure(c, c->d_counts, nt * c->n_pops * kTileSites * c->count_bytes))) return rc;
    c->n_sites = n_sites;
    c->tiles = d_tiles;
    c->flags = d_site_flags; c->have_flags = d_site_flags != nullptr;
    c->pos = d_pos; c->have_pos = d_pos != nullptr;
    return EVOFST_OK;
} catch (...) {
    return abi_exception(c);
}

extern "C" int evofst_synth(evofst_ctx *c, int64_t n_sites, int64_t first_site_global, uint64_t seed,
                            double missing_rate) try {
    if (!c || missing_rate < 0.0 || missing_rate >= 1.0) return EVOFST_ERR_ARG;
    NEED_GPU(c);
    int rc = begin_data(c, n_sites);
    if (rc) return rc;
    if (n_sites > c->cap_sites && (rc = evofst_reserve(c, n_sites))) return rc;
    c->n_sites = n_sites;
    SynthParams p{};
    p.tiles = (ulonglong2 *)c->d_tiles.p; p.pos = (int32_t *)c->d_pos.p;
    p.n_sites = n_sites; p.first_site_global = first_site_global;
    p.n_tiles = n_tiles_of(n_sites); p.W = c->W; p.P = c->n_pops;
    p.segs = (const SegDesc *)c->d_segs.p; p.word_seg_start = (const int32_t *)c->d_wss.p;
    p.seed = seed;
    p.miss_thr = (uint32_t)std::min(4294967295.0, missing_rate * 4294967296.0);
    HIPCHK(c, launch_synth(p, c->stream));
    c->tiles = c->d_tiles.p;
    c->flags = nullptr; c->have_flags = false;
    c->pos = (const int32_t *)c->d_pos.p; c->have_pos = true;
    return EVOFST_OK;
} catch (...) {
    return abi_exception(c);
}

extern "C" int evofst_download_tiles(evofst_ctx *c, void *tiles) try {
    if (!c || !tiles) return EVOFST_ERR_ARG;
    NEED_GPU(c);
    if (c->n_sites == 0 || !c->tiles) return fail(c, EVOFST_ERR_STATE, "no data on the device");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(tiles, c->tiles, evofst_tiles_bytes(c, c->n_sites), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return EVOFST_OK;
} catch (...) {
    return abi_exception(c);
}

extern "C" int evofst_download_tile_range(evofst_ctx *c, int64_t first_tile, int64_t n_tiles, void *tiles) try {
    if (!c || !tiles || first_tile < 0 || n_tiles <= 0) return EVOFST_ERR_ARG;
    NEED_GPU(c);
    if (c->n_sites == 0 || !c->tiles) return fail(c, EVOFST_ERR_STATE, "no data on the device");
    if (first_tile + n_tiles > (int64_t)n_tiles_of(c->n_sites)) return fail(c, EVOFST_ERR_ARG, "download_tile_range: tile range out of bounds");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t tb = (size_t)c->W * kTileSites * 16;
    HIPCHK(c, hipMemcpyAsync(tiles, (const char *)c->tiles + (size_t)first_tile * tb, (size_t)n_tiles * tb, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return EVOFST_OK;
} catch (...) {
    return abi_exception(c);
}

// ================================================================== modes

extern "C" int evofst_set_pairs(evofst_ctx *c, const int32_t *pair_i, const int32_t *pair_j, int n_pairs) try {
    if (!c || n_pairs < 0) return EVOFST_ERR_ARG;
    NEED_GPU(c);
    if (c->n_pops == 0) return fail(c, EVOFST_ERR_STATE, "set_pairs: set_layout first");
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<uint32_t> pr;
    if (n_pairs == 0) {
        for (int i = 0; i < c->n_pops; ++i)
            for (int j = i + 1; j < c->n_pops; ++j) pr.push_back((uint32_t)i | ((uint32_t)j << 16));
    } else {
        if (!pair_i || !pair_j) return fail(c, EVOFST_ERR_ARG, "set_pairs: null pair arrays");
        for (int q = 0; q < n_pairs; ++q) {
            if (pair_i[q] < 0 || pair_i[q] >= c->n_pops || pair_j[q] < 0 || pair_j[q] >= c->n_pops)
                return fail(c, EVOFST_ERR_ARG, "set_pairs: population index out of range");
            pr.push_back((uint32_t)pair_i[q] | ((uint32_t)pair_j[q] << 16));
        }
    }
    if (pr.empty()) return fail(c, EVOFST_ERR_ARG, "set_pairs: no pairs");
    c->pairs.swap(pr);
    c->mode = MODE_PAIRS;
    int rc;
    if ((rc = upload_layout(c))) return rc;
    if ((rc = upload_vec(c, c->d_pairs, c->pairs))) return rc;
    c->counts_done = c->stats_done = false;
    return set_base_internal(c, nullptr);
} catch (...) {
    return abi_exception(c);
}

extern "C" int evofst_set_against_all(evofst_ctx *c, const int32_t *use_pops, int n_use, const int64_t *set_sizes) try {
    if (!c || !use_pops || !set_sizes || n_use <= 0) return EVOFST_ERR_ARG;
    NEED_GPU(c);
    if (c->n_pops == 0) return fail(c, EVOFST_ERR_STATE, "set_against_all: set_layout first");
    if (n_use > 1024) return fail(c, EVOFST_ERR_UNSUPPORTED, "set_against_all: more than 1024 populations");
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<int32_t> ui(use_pops, use_pops + n_use);
    for (int u = 0; u < n_use; ++u)
        if (ui[u] < 0 || ui[u] >= c->n_pops) return fail(c, EVOFST_ERR_ARG, "set_against_all: population index out of range");
    c->use_idx.swap(ui);
    c->thresholds.resize(n_use);
    for (int u = 0; u < n_use; ++u) c->thresholds[u] = (double)(uint64_t)set_sizes[u] * 0.8;  // FstAgainstAll.cpp:253
    c->mode = MODE_GLOBAL;
    c->halo_n = 0;
    int rc;
    if ((rc = upload_layout(c))) return rc;
    if ((rc = upload_vec(c, c->d_use, c->use_idx))) return rc;
    if ((rc = upload_vec(c, c->d_thr, c->thresholds))) return rc;
    c->counts_done = c->stats_done = false;
    return set_base_internal(c, nullptr);
} catch (...) {
    return abi_exception(c);
}

extern "C" int evofst_set_global_halo(evofst_ctx *c, int n, const int32_t *tag, const double *num, const double *den) try {
    if (!c || n < 0) return EVOFST_ERR_ARG;
    NEED_GPU(c);
    if (c->mode != MODE_GLOBAL) return fail(c, EVOFST_ERR_STATE, "set_global_halo: against-all mode first");
    if (n > 0 && (!tag || !num || !den)) return fail(c, EVOFST_ERR_ARG, "set_global_halo: null arrays");
    if (c->snp_size > 0 && n > c->snp_size) return fail(c, EVOFST_ERR_ARG, "set_global_halo: more than SIZE sites");
    HIPCHK(c, hipSetDevice(c->device));
    c->halo_n = n;
    c->stats_done = false;
    if (n == 0) return EVOFST_OK;
    const size_t nu = c->use_idx.size();
    int rc;
    if ((rc = upload_vec(c, c->d_halo_tag, std::vector<int32_t>(tag, tag + n)))) return rc;
    if ((rc = upload_vec(c, c->d_halo_num, std::vector<double>(num, num + (size_t)n * nu)))) return rc;
    return upload_vec(c, c->d_halo_den, std::vector<double>(den, den + (size_t)n * nu));
} catch (...) {
    return abi_exception(c);
}

extern "C" int evofst_set_gram(evofst_ctx *c) try {
    if (!c) return EVOFST_ERR_ARG;
    NEED_GPU(c);
    if (c->n_pops < 2) return fail(c, EVOFST_ERR_STATE, "set_gram: set_layout with at least two populations first");
    HIPCHK(c, hipSetDevice(c->device));
    c->mode = MODE_GRAM;
    c->pairs.clear();
    int rc;
    if ((rc = upload_layout(c))) return rc;
    c->counts_done = c->stats_done = false;
    c->gram_missing = -1;
    return EVOFST_OK;
} catch (...) {
    return abi_exception(c);
}

extern "C" int evofst_set_site_windows(evofst_ctx *c, const int64_t *win_start, int64_t n_windows) try {
    if (!c || !win_start || n_windows <= 0) return EVOFST_ERR_ARG;
    NEED_GPU(c);
    for (int64_t w = 0; w < n_windows; ++w)
        if (win_start[w] < 0 || win_start[w + 1] < win_start[w]) return fail(c, EVOFST_ERR_ARG, "set_site_windows: starts must be non-negative and non-decreasing");
    HIPCHK(c, hipSetDevice(c->device));
    c->win_start.assign(win_start, win_start + n_windows + 1);
    c->gram_sub_rule = -1;
    return upload_vec(c, c->d_win, c->win_start);
} catch (...) {
    return abi_exception(c);
}

extern "C" int evofst_set_windows(evofst_ctx *c, int snp_size, int snp_step) try {
    if (!c) return EVOFST_ERR_ARG;
    if (snp_size < 0 || (snp_size > 0 && snp_step <= 0)) return fail(c, EVOFST_ERR_ARG, "set_windows: bad size/step");
    c->snp_size = snp_size; c->snp_step = snp_size > 0 ? snp_step : 0;
    c->g = snp_size > 0 ? std::gcd(snp_size, snp_step) : INT_MAX;
    c->stats_done = false;
    c->offsets_dirty = true;   // the scan also derives each pair's first segment row from g
    return EVOFST_OK;
} catch (...) {
    return abi_exception(c);
}

extern "C" int evofst_set_base_offsets(evofst_ctx *c, const int64_t *offsets) try {
    if (!c || !offsets) return EVOFST_ERR_ARG;
    NEED_GPU(c);
    if (c->mode == MODE_COUNTS) return fail(c, EVOFST_ERR_STATE, "set_base_offsets: choose a mode first");
    HIPCHK(c, hipSetDevice(c->device));
    return set_base_internal(c, offsets);
} catch (...) {
    return abi_exception(c);
}

static int launch_totals_and_carry(evofst_ctx *c, bool want_totals, double *d_carry_out = nullptr);

// Device-side forms of the multi-GPU exchange: nothing here waits for the GPU, so a pass over a shard stays one
// stream of kernels, copies and collectives (bench.py, N > 1).
extern "C" int evofst_set_base_offsets_device(evofst_ctx *c, const void *d_offsets) try {
    if (!c || !d_offsets) return EVOFST_ERR_ARG;
    NEED_GPU(c);
    if (c->mode != MODE_PAIRS && c->mode != MODE_GLOBAL) return fail(c, EVOFST_ERR_STATE, "set_base_offsets_device: choose a mode first");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t n = c->mode == MODE_PAIRS ? c->pairs.size() : 2;
    c->base.assign(n, 0);
    int rc = ensure(c, c->d_base, n * 8);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->d_base.p, d_offsets, n * 8, hipMemcpyDeviceToDevice, c->stream));
    c->gathered = nullptr;
    c->base_host_valid = false;
    c->offsets_dirty = true;
    c->stats_done = false;
    return EVOFST_OK;
} catch (...) {
    return abi_exception(c);
}

extern "C" int evofst_set_base_offsets_gathered(evofst_ctx *c, const void *d_gathered, int64_t row_stride, int n_rows) try {
    if (!c || !d_gathered || n_rows < 0) return EVOFST_ERR_ARG;
    NEED_GPU(c);
    if (c->mode != MODE_PAIRS && c->mode != MODE_GLOBAL) return fail(c, EVOFST_ERR_STATE, "set_base_offsets_gathered: choose a mode first");
    const size_t n = c->mode == MODE_PAIRS ? c->pairs.size() : 2;
    if (row_stride < (int64_t)n) return fail(c, EVOFST_ERR_ARG, "set_base_offsets_gathered: row_stride is smaller than a row of totals");
    HIPCHK(c, hipSetDevice(c->device));
    c->base.assign(n, 0);
    int rc = ensure(c, c->d_base, n * 8);
    if (rc) return rc;
    // the next scan (run_stats) forms the sums itself, in stream order behind the caller's collective: the buffer
    // must keep its contents until then
    c->gathered = (const int64_t *)d_gathered; c->gathered_stride = row_stride; c->gathered_rows = n_rows;
    c->base_host_valid = false;
    c->offsets_dirty = true;
    c->stats_done = false;
    return EVOFST_OK;
} catch (...) {
    return abi_exception(c);
}

extern "C" int evofst_copy_included_totals(evofst_ctx *c, void *d_dst) try {
    if (!c || !d_dst) return EVOFST_ERR_ARG;
    NEED_GPU(c);
    if (!c->counts_done || (c->mode != MODE_PAIRS && c->mode != MODE_GLOBAL)) return fail(c, EVOFST_ERR_STATE, "copy_included_totals: run_counts in a mode first");
    HIPCHK(c, hipSetDevice(c->device));
    // right after run_counts the scan is still to do: it writes the totals into the caller's buffer as well (no copy)
    if (c->offsets_dirty) return run_scan(c, (int64_t *)d_dst);
    const void *src = c->mode == MODE_PAIRS ? c->d_incl_totals.p : c->d_uk_tot.p;
    const size_t n = c->mode == MODE_PAIRS ? c->pairs.size() : 2;
    HIPCHK(c, hipMemcpyAsync(d_dst, src, n * 8, hipMemcpyDeviceToDevice, c->stream));
    return EVOFST_OK;
} catch (...) {
    return abi_exception(c);
}

extern "C" int evofst_copy_boundary(evofst_ctx *c, void *d_dst) try {
    if (!c || !d_dst) return EVOFST_ERR_ARG;
    NEED_GPU(c);
    if (!c->stats_done || c->mode != MODE_PAIRS) return fail(c, EVOFST_ERR_STATE, "copy_boundary: pairs mode + run_stats first");
    HIPCHK(c, hipSetDevice(c->device));
    return launch_totals_and_carry(c, false, (double *)d_dst);   // the kernel writes the caller's buffer: no copy behind it
} catch (...) {
    return abi_exception(c);
}

// ================================================================== compute

namespace {
// K1 in the given mode.  MODE_COUNTS (the on-demand form behind the getters) leaves the chunk plan of a finished
// statistics pass alone: its tails are indexed by it.
int launch_k1_counts(evofst_ctx *c, int mode) {
    const int keep_ct = c->chunk_tiles, keep_nc = c->n_chunks;
    plan_chunks(c);
    K1Params p{};
    p.tiles = (const ulonglong2 *)c->tiles;
    p.site_flags = c->have_flags ? c->flags : nullptr;
    p.n_sites = c->n_sites; p.n_tiles = c->n_tiles; p.W = c->W; p.P = c->n_pops;
    p.segs = (const SegDesc *)c->d_segs.p; p.word_seg_start = (const int32_t *)c->d_wss.p; p.n_segs = (int)c->segs.size();
    p.n_full = (const int32_t *)c->d_nfull.p;
    int rc;
    if ((rc = ensure(c, c->d_counts, (size_t)c->n_tiles * c->n_pops * kTileSites * c->count_bytes))) return rc;
    p.counts = c->d_counts.p;
    p.chunk_tiles = c->chunk_tiles; p.n_chunks = c->n_chunks;
    if (mode == MODE_PAIRS) {
        p.pairs = (const uint32_t *)c->d_pairs.p; p.n_pairs = (int)c->pairs.size();
        if ((size_t)p.n_pairs * 20 + (size_t)c->n_pops * 128 + (size_t)c->segs.size() * 16 > 140 * 1024)
            return fail(c, EVOFST_ERR_UNSUPPORTED, "run_counts: too many pairs for the per-chunk census in LDS (use the Gram path: evofst_set_gram)");
        if ((rc = ensure(c, c->d_census, (size_t)c->n_chunks * p.n_pairs * 4))) return rc;
        p.chunk_census = (uint32_t *)c->d_census.p;
        if ((rc = ensure(c, c->d_zf, (size_t)c->n_tiles * c->n_pops * 16))) return rc;
        if ((rc = ensure(c, c->d_irr, (size_t)c->n_tiles * 8))) return rc;
        if ((rc = ensure(c, c->d_skip, (size_t)c->n_tiles * 8))) return rc;
        if ((rc = ensure(c, c->d_miss, (size_t)c->n_tiles * 8))) return rc;
        p.zf_masks = (uint64_t *)c->d_zf.p; p.irr_mask = (uint64_t *)c->d_irr.p; p.skip_mask = (uint64_t *)c->d_skip.p;
        p.miss_mask = (uint64_t *)c->d_miss.p;
        if ((rc = ensure(c, c->d_dirty, (size_t)c->n_chunks * 4))) return rc;
        p.chunk_dirty = (uint32_t *)c->d_dirty.p;
        if (!c->k1_marks_valid) { c->k1_marks_valid = true; c->k1_marks_known = false; }
        if ((rc = ensure(c, c->d_nan, (size_t)c->n_tiles * 8))) return rc;
        p.nan_mask = (uint64_t *)c->d_nan.p;
        p.compact = !c->prefer_plain && (size_t)c->n_pops * c->count_bytes <= 64 && clean_walk_applies(c) ? 1 : 0;
    } else if (mode == MODE_GRAM) {
        if ((rc = ensure(c, c->d_zf, (size_t)c->n_tiles * c->n_pops * 32))) return rc;
        if ((rc = ensure(c, c->d_irr, (size_t)c->n_tiles * 8))) return rc;
        if ((rc = ensure(c, c->d_skip, (size_t)c->n_tiles * 8))) return rc;
        if ((rc = ensure(c, c->d_poison, (size_t)c->n_tiles * 4))) return rc;
        p.zf_masks = (uint64_t *)c->d_zf.p; p.irr_mask = (uint64_t *)c->d_irr.p; p.skip_mask = (uint64_t *)c->d_skip.p;
        p.tile_poison = (uint32_t *)c->d_poison.p;
        c->gram_missing = -1;
    } else if (mode == MODE_GLOBAL) {
        if ((rc = ensure(c, c->d_gtotals, (size_t)c->n_tiles * kTileSites * 8))) return rc;
        if ((rc = ensure(c, c->d_kept, (size_t)c->n_tiles * 8))) return rc;
        if ((rc = ensure(c, c->d_skip, (size_t)c->n_tiles * 8))) return rc;
        if ((rc = ensure(c, c->d_used_kept, (size_t)c->n_chunks * 2 * 4))) return rc;
        p.totals = (uint2 *)c->d_gtotals.p; p.kept_mask = (uint64_t *)c->d_kept.p; p.skip_mask = (uint64_t *)c->d_skip.p;
        p.chunk_used_kept = (uint32_t *)c->d_used_kept.p;
    }
    record(c, 0, 0);
    HIPCHK(c, launch_k1(p, c->count_bytes, mode, c->stream));
    record(c, 0, 1);
    c->plain_counts = !p.compact;
    c->compact_counts = p.compact != 0;
    if (mode == MODE_COUNTS && c->stats_done) { c->chunk_tiles = keep_ct; c->n_chunks = keep_nc; }
    return EVOFST_OK;
}
}  // namespace

extern "C" int evofst_run_counts(evofst_ctx *c) try {
    if (!c) return EVOFST_ERR_ARG;
    NEED_GPU(c);
    if (c->n_sites == 0 || !c->tiles) return fail(c, EVOFST_ERR_STATE, "run_counts: no data");
    HIPCHK(c, hipSetDevice(c->device));
    c->stats_done = false;
    int rc = launch_k1_counts(c, c->mode);
    if (rc) return rc;
    c->counts_done = true;
    c->offsets_dirty = true;     // the offset scan runs with the first call that needs it (run_stats, the totals getters)
    return EVOFST_OK;
} catch (...) {
    return abi_exception(c);
}

extern "C" int evofst_run_stats(evofst_ctx *c) try {
    if (!c) return EVOFST_ERR_ARG;
    NEED_GPU(c);
    if (!c->counts_done) return fail(c, EVOFST_ERR_STATE, "run_stats: run_counts first");
    if (c->mode == MODE_COUNTS) return fail(c, EVOFST_ERR_STATE, "run_stats: choose a mode (set_pairs / set_against_all)");
    if (c->mode == MODE_GRAM) return fail(c, EVOFST_ERR_STATE, "run_stats: Gram mode computes through evofst_run_gram");
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    if (c->offsets_dirty && (rc = run_scan(c))) return rc;
    if (c->mode == MODE_PAIRS) {
        // K1's one-byte counts are for clean_walk_kernel alone: should it not run after all, the counts are made again
        if (c->compact_counts && !clean_walk_applies(c) && (rc = launch_k1_counts(c, MODE_COUNTS))) return rc;
        const size_t np = c->pairs.size();
        const int64_t g = c->g;
        c->seg_cap = c->n_sites / g + 2;
        const size_t seg_elems = (size_t)c->seg_cap * np;
        if ((rc = ensure(c, c->d_seg_rows, seg_elems * sizeof(SegRow)))) return rc;
        const size_t cq = (size_t)c->n_chunks * np;
        if ((rc = ensure(c, c->d_tail, cq * 3 * 8))) return rc;
        if ((rc = ensure(c, c->d_tail_n, cq * 16))) return rc;
        if (k2_lds_bytes(c->n_pops, *std::max_element(c->n_full.begin(), c->n_full.end())) > 160 * 1024) return fail(c, EVOFST_ERR_UNSUPPORTED, "run_stats: too many populations for the LDS table");
        K2Params p = make_k2(c);
        record(c, 2, 0);
        if ((rc = launch_walks(c, p, clean_walk_applies(c), c->compact_counts))) return rc;
        record(c, 2, 1);
        record(c, 3, 0);
        HIPCHK(c, launch_stitch(p, c->stream));
        record(c, 3, 1);
    } else {
        if (c->snp_size <= 0) return fail(c, EVOFST_ERR_ARG, "run_stats: against-all mode needs set_windows(size, step)");
        const int nu = (int)c->use_idx.size();
        int tn_max = 0, max_n_use = 0;
        for (int32_t k : c->use_idx) { tn_max += c->n_full[k]; max_n_use = std::max(max_n_use, c->n_full[k]); }
        size_t lds = global_lds_bytes(tn_max, max_n_use, nu, c->snp_size);
#ifdef EVOFST_EXPERIMENTS
        if (const char *e = getenv("EVOFST_DEBUG_GLOBAL_LDS")) lds += (size_t)atoi(e);   // occupancy experiments
#endif
        // the two-phase kernel (lane = site for the evaluation, thread = window x population for the rows) where its
        // ring of SIZE + 64 kept sites per population fits in LDS; the lane = population kernel otherwise
        int kw = 0;
        const size_t tile_lds = global_tile_lds_bytes(tn_max, max_n_use, nu, c->snp_size, &kw);
        const bool tile_walk = c->opt_global_walk != 1 && tile_lds <= 150 * 1024 && nu <= 160 && c->halo_n <= c->snp_size;
        if (!tile_walk && lds > 160 * 1024) return fail(c, EVOFST_ERR_UNSUPPORTED, "run_stats: SIZE * populations too large for the LDS deque");
        if ((rc = ensure(c, c->d_chunk_rows, (size_t)c->n_chunks * 4))) return rc;
        if ((rc = ensure(c, c->d_row_off, (size_t)c->n_chunks * 8))) return rc;
        if ((rc = ensure(c, c->d_row_tot, 8))) return rc;
        c->row_cap = c->n_sites / c->snp_step + 2;
        if ((rc = ensure(c, c->d_row_start, (size_t)c->row_cap * 4))) return rc;
        if ((rc = ensure(c, c->d_row_end, (size_t)c->row_cap * 4))) return rc;
        if ((rc = ensure(c, c->d_row_num, (size_t)c->row_cap * nu * 8))) return rc;
        if ((rc = ensure(c, c->d_row_den, (size_t)c->row_cap * nu * 8))) return rc;
        record(c, 3, 0);
        HIPCHK(c, launch_global_rows((const uint64_t *)c->d_kept.p, (const uint64_t *)c->d_skip.p,
                                     (const int64_t *)c->d_uk_off.p, c->n_chunks, c->chunk_tiles, c->n_tiles, c->snp_size,
                                     c->snp_step, (uint32_t *)c->d_chunk_rows.p, c->stream));
        HIPCHK(c, launch_scan((const uint32_t *)c->d_chunk_rows.p, nullptr, (int64_t *)c->d_row_off.p,
                              (int64_t *)c->d_row_tot.p, c->n_chunks, 1, 1, nullptr, c->stream));
        record(c, 3, 1);
        GlobalParams p = make_global(c);
        p.group = 0;   // the launcher cuts K1's chunks into groups that fill whole rounds of workgroups (launch_global)
#ifdef EVOFST_EXPERIMENTS
        if (const char *e = getenv("EVOFST_GLOBAL_GROUP_TILES")) p.group = std::max(1, atoi(e) / std::max(1, c->chunk_tiles));
#endif
        record(c, 2, 0);
        c->last_global_walk = tile_walk ? 2 : 1;
        if (tile_walk) { p.kw = kw; HIPCHK(c, launch_global_tile(p, c->count_bytes, tile_lds, c->stream)); }
        else HIPCHK(c, launch_global(p, c->count_bytes, lds, c->stream));
        record(c, 2, 1);
    }
    c->stats_done = true;
    return EVOFST_OK;
} catch (...) {
    return abi_exception(c);
}

extern "C" int evofst_run_gram(evofst_ctx *c, int64_t first_window, int64_t n_windows) try {
    if (!c) return EVOFST_ERR_ARG;
    NEED_GPU(c);
    if (c->mode != MODE_GRAM || !c->counts_done) return fail(c, EVOFST_ERR_STATE, "run_gram: set_gram + run_counts first");
    const int64_t nw_total = (int64_t)c->win_start.size() - 1;
    if (nw_total <= 0) return fail(c, EVOFST_ERR_STATE, "run_gram: set_site_windows first");
    if (first_window < 0 || n_windows <= 0 || first_window + n_windows > nw_total || n_windows > 0x7fffffff)
        return fail(c, EVOFST_ERR_ARG, "run_gram: window range out of bounds");
    if (c->win_start.back() > c->n_sites) return fail(c, EVOFST_ERR_ARG, "run_gram: windows reach beyond the loaded sites");
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    if (c->gram_missing < 0) {
        // does any site need the missing-allele terms?  (K1 flagged them; one small copy per run_counts)
        std::vector<uint64_t> irr((size_t)c->n_tiles);
        HIPCHK(c, hipMemcpyAsync(irr.data(), c->d_irr.p, irr.size() * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->gram_missing = 0;
        for (uint64_t m : irr) if (m) { c->gram_missing = 1; break; }
    }
    const int64_t P = c->n_pops, np = P * (P - 1) / 2;
    GramParams p{};
    p.max_n = *std::max_element(c->n_full.begin(), c->n_full.end());
    for (int64_t w = first_window; w < first_window + n_windows; ++w)
        p.max_window_sites = std::max(p.max_window_sites, c->win_start[(size_t)w + 1] - c->win_start[(size_t)w]);
    // Windows longer than the one-contraction form takes (gram_kernel: longest window x chromosomes <= 2^13) run as pieces
    // that are short enough, and evofst_get_gram adds the pieces' sums: every piece is inside the bar of its own terms and so
    // is their sum.  The pieces of all windows are cut once per set of windows.
    const int64_t rule = c->opt_gram_den == 2 ? 0 : gram_one_rule_sites(c->count_bytes, p.max_n);
    const bool split = rule > 0 && p.max_window_sites > rule;
    const int64_t n_run = n_windows;          // windows of the call
    if (split) {
        if (c->gram_sub_rule != rule) {
            c->gram_sub.clear(); c->gram_suboff.clear();
            for (size_t w = 0; w + 1 < c->win_start.size(); ++w) {
                c->gram_suboff.push_back((int64_t)c->gram_sub.size());
                const int64_t a = c->win_start[w], b = c->win_start[w + 1];
                c->gram_sub.push_back(a);                                  // (an empty window is one empty piece)
                for (int64_t x = a + rule; x < b; x += rule) c->gram_sub.push_back(x);
            }
            c->gram_suboff.push_back((int64_t)c->gram_sub.size());
            c->gram_sub.push_back(c->win_start.back());
            if ((rc = upload_vec(c, c->d_gsub, c->gram_sub)) || (rc = upload_vec(c, c->d_gsuboff, c->gram_suboff))) return rc;
            c->gram_sub_rule = rule;
        }
        const int64_t s0 = c->gram_suboff[(size_t)first_window], s1 = c->gram_suboff[(size_t)(first_window + n_windows)];
        if (s1 - s0 > 0x7fffffff) return fail(c, EVOFST_ERR_ARG, "run_gram: too many window pieces in one call");
        c->gram_sub_base = s0; c->gram_first_window = first_window;
        first_window = s0; n_windows = s1 - s0;           // the kernels run pieces
        p.max_window_sites = std::min(p.max_window_sites, rule);
    }
    c->gram_split = split;
    p.two_streams = c->opt_gram_den == 2;                     // evofst_set_option "gram_denominator" 2
    p.fmt = gram_result_format(p, c->count_bytes);
    if ((rc = ensure(c, c->d_gU, (size_t)n_windows * P * 8))) return rc;
    if ((rc = ensure(c, c->d_gC, (size_t)n_windows * P * 8 * 2))) return rc;   // C, then D (sites with a called allele)
    if ((rc = ensure(c, c->d_graw, gram_raw_bytes(c->n_pops, n_windows) >> (p.fmt ? 1 : 0)))) return rc;   // (8-byte entries: half the table)
    p.counts = c->d_counts.p; p.n_sites = c->n_sites; p.n_tiles = c->n_tiles; p.P = c->n_pops;
    p.n_full = (const int32_t *)c->d_nfull.p;
    p.masks = (const uint64_t *)c->d_zf.p; p.skip_mask = (const uint64_t *)c->d_skip.p;
    p.tile_poison = (const uint32_t *)c->d_poison.p;
    p.win_start = (const int64_t *)(split ? c->d_gsub.p : c->d_win.p);
    p.first_window = first_window; p.n_windows = (int32_t)n_windows;
    p.U = (double *)c->d_gU.p; p.C = (double *)c->d_gC.p; p.D = p.C + (size_t)n_windows * P;
    p.n_pairs = np;
    p.out_raw = (double2 *)c->d_graw.p;
#ifdef EVOFST_EXPERIMENTS
    if (const char *e = getenv("EVOFST_DEBUG_GRAM")) p.debug = atoi(e);
#endif
    record(c, 3, 0);
    HIPCHK(c, launch_gram_popsums(p, c->count_bytes, c->stream));
    record(c, 3, 1);
    record(c, 2, 0);
    HIPCHK(c, launch_gram(p, c->count_bytes, c->gram_missing != 0, c->stream));
    c->last_gram_streams = gram_one_stream(p, c->count_bytes) ? 1 : 2;
    c->last_gram_fmt = p.fmt;
    record(c, 2, 1);
    c->gram_windows = n_run;
    return EVOFST_OK;
} catch (...) {
    return abi_exception(c);
}

extern "C" int evofst_get_gram(evofst_ctx *c, double *sum_num, double *sum_den, double *sum_dxy_k, double *n_included) try {
    if (!c) return EVOFST_ERR_ARG;
    NEED_GPU(c);
    if (c->mode != MODE_GRAM || c->gram_windows <= 0) return fail(c, EVOFST_ERR_STATE, "get_gram: run_gram first");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t n = (size_t)c->gram_windows * ((size_t)c->n_pops * (c->n_pops - 1) / 2);
    int rc;
    // the kernel's table of results -> the four arrays of the interface
    if ((rc = ensure(c, c->d_gnum, n * 8)) || (rc = ensure(c, c->d_gden, n * 8)) || (rc = ensure(c, c->d_gk, n * 8)) ||
        (rc = ensure(c, c->d_gcnt, n * 8))) return rc;
    HIPCHK(c, launch_gram_unpack((const double2 *)c->d_graw.p, c->n_pops, c->gram_windows,
                                 gram_uses_i8(c->count_bytes, *std::max_element(c->n_full.begin(), c->n_full.end())),
                                 c->last_gram_fmt, c->gram_split ? (const int64_t *)c->d_gsuboff.p + c->gram_first_window : nullptr, c->gram_sub_base,
                                 (const double *)c->d_gU.p, (double *)c->d_gnum.p, (double *)c->d_gden.p, (double *)c->d_gk.p, (double *)c->d_gcnt.p, c->stream));
    if ((rc = d2h(c, sum_num, c->d_gnum.p, n)) || (rc = d2h(c, sum_den, c->d_gden.p, n)) ||
        (rc = d2h(c, sum_dxy_k, c->d_gk.p, n)) || (rc = d2h(c, n_included, c->d_gcnt.p, n))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return EVOFST_OK;
} catch (...) {
    return abi_exception(c);
}

extern "C" int evofst_run(evofst_ctx *c) try {
    if (!c) return EVOFST_ERR_ARG;
    NEED_GPU(c);
    if (c->mode == MODE_PAIRS && c->n_sites > 0 && c->tiles) {
        // pairs mode: one streaming kernel from the tiles to the segment rows (no counts kept in HBM; the getters
        // that want them run K1 on demand)
        HIPCHK(c, hipSetDevice(c->device));
        const StreamPlan pl = stream_plan(c, c->n_sites);
        if (pl.tr > 0) { c->last_pairs_pass = 2; return run_stream(c, pl); }
        int sub = 32;
        const int ct = fused_plan(c, c->n_sites, false, &sub);
        if (ct > 0) { c->last_pairs_pass = 3; return run_fused(c, ct, sub); }
    }
    c->last_pairs_pass = c->mode == MODE_PAIRS ? 1 : 0;
    int rc = evofst_run_counts(c);
    if (rc) return rc;
    return evofst_run_stats(c);
} catch (...) {
    return abi_exception(c);
}

// One more form of evofst_run for callers that cannot allocate (no counterpart in the reference): the one-kernel pairs pass
// on this GPU's static scratch array.  Calls no hipMalloc / hipFree.
extern "C" int evofst_run_pairs_noalloc(evofst_ctx *c, int64_t n_sites, const void *d_tiles, const uint8_t *d_site_flags,
                                        void *d_seg_rows, int64_t seg_rows_capacity, int64_t *d_first_segment,
                                        int64_t *d_incl_totals, double *d_carry) try {
    if (!c || !d_tiles || !d_seg_rows || !d_first_segment || !d_incl_totals || !d_carry || n_sites <= 0) return EVOFST_ERR_ARG;
    NEED_GPU(c);
    if (c->mode != MODE_PAIRS) return fail(c, EVOFST_ERR_STATE, "run_pairs_noalloc: set_pairs first");
    if (c->snp_size <= 0) return fail(c, EVOFST_ERR_STATE, "run_pairs_noalloc: set_windows first");
    if (c->gathered) return fail(c, EVOFST_ERR_STATE, "run_pairs_noalloc: base offsets must be set with evofst_set_base_offsets[_device]");
    if ((int64_t)n_tiles_of(n_sites) * kTileSites > (int64_t)INT_MAX) return fail(c, EVOFST_ERR_UNSUPPORTED, "run_pairs_noalloc: too many sites for one shard");
    const size_t np = c->pairs.size();
    if (np > (size_t)kK2Threads) return fail(c, EVOFST_ERR_UNSUPPORTED, "run_pairs_noalloc: more than 256 pairs");
    if (c->device >= kMaxDevices || !g_scratch_ptr[c->device]) return fail(c, EVOFST_ERR_STATE, "run_pairs_noalloc: no scratch address for this device");
    HIPCHK(c, hipSetDevice(c->device));
    int sub = 32;
    const int ct = fused_plan(c, n_sites, true, &sub);
    if (ct <= 0) return fail(c, EVOFST_ERR_UNSUPPORTED, "run_pairs_noalloc: this layout does not fit the one-kernel pass");
    const int n_tiles = n_tiles_of(n_sites);
    const int n_chunks = (n_tiles + ct - 1) / ct;
    const int n_sub = fused_tail_chunks((int)np, n_chunks);
    const int64_t seg_cap = n_sites / c->g + 2;
    if (seg_rows_capacity < seg_cap) return fail(c, EVOFST_ERR_ARG, "run_pairs_noalloc: seg_rows_capacity is below n_sites / gcd(SIZE, STEP) + 2 rows");
    // carve the scratch: [sync words 256 B][agg0][agg1][agg2][tail 3 planes][tail_n]
    auto a256 = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t n1 = (n_chunks + kFusedFan - 1) / kFusedFan, n2 = (n1 + kFusedFan - 1) / kFusedFan;
    const size_t o_agg0 = 256, o_agg1 = o_agg0 + a256((size_t)n_chunks * np * 8), o_agg2 = o_agg1 + a256(n1 * np * 8),
                 o_tail = o_agg2 + a256(n2 * np * 8), o_tailn = o_tail + a256((size_t)n_sub * np * 24),
                 need = o_tailn + a256((size_t)n_sub * np * 16);
    if (need > kNoAllocScratchBytes)
        return fail(c, EVOFST_ERR_UNSUPPORTED, "run_pairs_noalloc: " + std::to_string(need) + " bytes of scratch needed for this many sites x pairs, the static array holds " +
                                                   std::to_string(kNoAllocScratchBytes) + " (use evofst_run)");
    std::lock_guard<std::mutex> scratch_lock(g_scratch_mu[c->device]);
    unsigned char *sp = (unsigned char *)g_scratch_ptr[c->device];     // the address, never the array's name
    uint32_t &epoch = g_scratch_epoch[c->device];
    if (epoch == 0xffffffffu) {            // the epoch counter wraps: forget every tag
        HIPCHK(c, hipMemsetAsync(sp, 0, kNoAllocScratchBytes, c->stream));
        epoch = 0;
    }
    HIPCHK(c, hipMemsetAsync(sp, 0, 256, c->stream));                  // ticket and error words start at zero
    FusedParams p{};
    p.tiles = (const ulonglong2 *)d_tiles;
    p.site_flags = d_site_flags;
    p.n_sites = n_sites; p.n_tiles = n_tiles; p.W = c->W; p.P = c->n_pops;
    p.segs = (const SegDesc *)c->d_segs.p; p.n_segs = (int)c->segs.size();
    p.n_full = (const int32_t *)c->d_nfull.p;
    p.max_n = *std::max_element(c->n_full.begin(), c->n_full.end());
    p.pairs = (const uint32_t *)c->d_pairs.p; p.n_pairs = (int)np;
    p.chunk_tiles = ct; p.n_chunks = n_chunks;
    p.g = c->g;
    p.base = (const int64_t *)c->d_base.p;
    p.first_segment = d_first_segment; p.incl_totals = d_incl_totals;
    p.seg_cap = seg_cap; p.seg_rows = (SegRow *)d_seg_rows;
    p.tail = (double *)(sp + o_tail); p.tail_n = (int4 *)(sp + o_tailn);
    p.agg0 = (uint64_t *)(sp + o_agg0); p.agg1 = (uint64_t *)(sp + o_agg1); p.agg2 = (uint64_t *)(sp + o_agg2);
    p.ticket = (uint32_t *)sp; p.err = (uint32_t *)sp + 32;
    p.epoch = ++epoch;
    HIPCHK(c, launch_fused(p, c->count_bytes, sub, 0, c->stream));
    K2Params k2{};
    k2.n_pairs = (int)np; k2.n_chunks = n_sub; k2.g = c->g; k2.seg_cap = seg_cap;
    k2.seg_rows = (SegRow *)d_seg_rows; k2.tail = p.tail; k2.tail_n = p.tail_n;
    HIPCHK(c, launch_stitch(k2, c->stream));
    HIPCHK(c, launch_totals(k2, (const int64_t *)c->d_base.p, d_incl_totals, nullptr, d_carry, c->stream));
    // evofst_sync looks at the pass's error word (a look-back wait that gave up leaves rows, tails and totals unwritten)
    c->check_word = (const uint32_t *)sp + 32; c->check_dirty = false;
    c->check_zero[0] = sp; c->check_zero_bytes[0] = 256;
    c->check_zero[1] = c->check_zero[2] = nullptr; c->check_zero_bytes[1] = c->check_zero_bytes[2] = 0;
    return EVOFST_OK;
} catch (...) {
    return abi_exception(c);
}

extern "C" int64_t evofst_noalloc_scratch_bytes(void) { return (int64_t)kNoAllocScratchBytes; }

extern "C" int evofst_sync(evofst_ctx *c) try {
    if (!c) return EVOFST_ERR_ARG;
    NEED_GPU(c);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->mode == MODE_PAIRS && c->counts_done && c->k1_marks_valid && !c->k1_marks_known && c->d_dirty.p && c->n_chunks > 0) {
        int rc = ensure(c, c->d_row_tot, 8);
        if (rc) return rc;
        uint32_t n = 0;
        HIPCHK(c, launch_count_marks((const uint32_t *)c->d_dirty.p, c->n_chunks, (uint32_t *)c->d_row_tot.p, c->stream));
        HIPCHK(c, hipMemcpyAsync(&n, c->d_row_tot.p, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->k1_marks = n; c->k1_marks_known = true;
    }
    return fused_verdict(c);
} catch (...) {
    return abi_exception(c);
}

extern "C" int evofst_set_option(evofst_ctx *c, const char *name, int64_t value) try {
    if (!c || !name) return EVOFST_ERR_ARG;
    const std::string n(name);
    if (n == "pairs_pass") {
        if (value < 0 || value > 3) return fail(c, EVOFST_ERR_ARG, "set_option: pairs_pass is 0 (auto), 1 (counts + scan + walk), 2 (stream pass) or 3 (lock-step one-kernel pass)");
        c->opt_pairs_pass = (int)value;
    } else if (n == "gram_denominator") {
        if (value != 0 && value != 2) return fail(c, EVOFST_ERR_ARG, "set_option: gram_denominator is 0 (auto) or 2 (always two FP64 streams)");
        c->opt_gram_den = (int)value;
    } else if (n == "chunk_tiles") {
        if (value < 0 || value > 256) return fail(c, EVOFST_ERR_ARG, "set_option: chunk_tiles is 0 (auto) .. 256");
        c->opt_chunk_tiles = (int)value;
    } else if (n == "walk") {
        if (value < 0 || value > 2) return fail(c, EVOFST_ERR_ARG, "set_option: walk is 0 (auto), 1 (general walk only) or 2 (k2_window_kernel)");
        c->opt_walk = (int)value;
    } else if (n == "stream_blocks") {
        if (value < 0 || value > (1 << 20)) return fail(c, EVOFST_ERR_ARG, "set_option: stream_blocks is 0 (auto) or a number of workgroups");
        c->opt_stream_blocks = (int)value;
    } else if (n == "global_walk") {
        if (value < 0 || value > 2) return fail(c, EVOFST_ERR_ARG, "set_option: global_walk is 0 (auto), 1 (lane = population kernel) or 2 (two-phase kernel where it fits)");
        c->opt_global_walk = (int)value;
    } else if (n == "stream_sets") {
        if (value < 0 || value > 2) return fail(c, EVOFST_ERR_ARG, "set_option: stream_sets is 0 (auto), 1 or 2");
        c->opt_stream_sets = (int)value;
    } else if (n == "stream_buffers") {
        if (value != 0 && (value < 2 || value > kStreamMaxBufs)) return fail(c, EVOFST_ERR_ARG, "set_option: stream_buffers is 0 (auto) or 2 .. 4");
        c->opt_stream_bufs = (int)value;
    } else if (n == "stream_producers" || n == "stream_consumers") {
        if (value < 0 || value >= kStreamMaxWaves) return fail(c, EVOFST_ERR_ARG, "set_option: " + n + " is 0 (auto) .. 11");
        (n == "stream_producers" ? c->opt_stream_prod : c->opt_stream_cons) = (int)value;
    } else {
        return fail(c, EVOFST_ERR_ARG, "set_option: unknown option '" + n + "'");
    }
    c->stats_done = false;
    return EVOFST_OK;
} catch (...) {
    return abi_exception(c);
}

extern "C" int evofst_last_global_walk(const evofst_ctx *c) {
    return c ? c->last_global_walk : 0;
}

extern "C" int evofst_last_gram_streams(const evofst_ctx *c) {
    return c ? c->last_gram_streams : 0;
}

extern "C" int evofst_last_pairs_pass(const evofst_ctx *c) {
    return c ? c->last_pairs_pass : 0;
}

extern "C" int evofst_set_profiling(evofst_ctx *c, int enabled) try {
    if (!c) return EVOFST_ERR_ARG;
    c->profiling = enabled != 0;
    return EVOFST_OK;
} catch (...) {
    return abi_exception(c);
}

extern "C" int evofst_get_kernel_ms(evofst_ctx *c, int which, float *ms) try {
    if (!c || !ms || which < 0 || which >= kNumKernels) return EVOFST_ERR_ARG;
    NEED_GPU(c);
    if (!c->ev_valid[which]) return fail(c, EVOFST_ERR_STATE, "get_kernel_ms: kernel was not profiled");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipEventSynchronize(c->ev[which][1]));
    HIPCHK(c, hipEventElapsedTime(ms, c->ev[which][0], c->ev[which][1]));
    return EVOFST_OK;
} catch (...) {
    return abi_exception(c);
}

// ================================================================== results

static int check_range(evofst_ctx *c, int64_t first, int64_t n) {
    if (first < 0 || n <= 0 || first + n > c->n_sites) return fail(c, EVOFST_ERR_ARG, "site range out of bounds");
    return EVOFST_OK;
}

extern "C" int evofst_get_counts(evofst_ctx *c, int64_t first_site, int64_t n_sites, int32_t *alt, int32_t *nm) try {
    if (!c || !alt || !nm) return EVOFST_ERR_ARG;
    NEED_GPU(c);
    int rc = check_range(c, first_site, n_sites);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    if ((rc = need_counts(c, "get_counts"))) return rc;
    const size_t n = (size_t)n_sites * c->n_pops;
    DevBuf da, dn;
    if ((rc = ensure(c, da, n * 4)) || (rc = ensure(c, dn, n * 4))) { release(da); release(dn); return rc; }
    hipError_t e = launch_unpack_counts(c->d_counts.p, c->count_bytes, c->n_pops, first_site, n_sites, (int32_t *)da.p,
                                        (int32_t *)dn.p, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(alt, da.p, n * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(nm, dn.p, n * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    release(da); release(dn);
    if (e != hipSuccess) return hip_fail(c, e, "get_counts");
    return EVOFST_OK;
} catch (...) {
    return abi_exception(c);
}

extern "C" int evofst_get_included_totals(evofst_ctx *c, int64_t *totals) try {
    if (!c || !totals) return EVOFST_ERR_ARG;
    NEED_GPU(c);
    const bool fused = c->stats_done && !c->counts_done && c->mode == MODE_PAIRS;   // the fused pass wrote the totals itself
    if ((!c->counts_done && !fused) || c->mode == MODE_COUNTS) return fail(c, EVOFST_ERR_STATE, "get_included_totals: run_counts in a mode first");
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    if (!fused && c->offsets_dirty && (rc = run_scan(c))) return rc;
    if (c->mode == MODE_PAIRS) { if ((rc = d2h(c, totals, c->d_incl_totals.p, c->pairs.size()))) return rc; }
    else { if ((rc = d2h(c, totals, c->d_uk_tot.p, 2))) return rc; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return EVOFST_OK;
} catch (...) {
    return abi_exception(c);
}

extern "C" int evofst_get_site_stats(evofst_ctx *c, int64_t first_site, int64_t n_sites, double *num, double *den,
                                     double *dxy, uint8_t *include) try {
    if (!c) return EVOFST_ERR_ARG;
    NEED_GPU(c);
    if (c->mode != MODE_PAIRS) return fail(c, EVOFST_ERR_STATE, "get_site_stats: pairs mode first");
    int rc = check_range(c, first_site, n_sites);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    if ((rc = need_counts(c, "get_site_stats"))) return rc;
    const size_t n = (size_t)n_sites * c->pairs.size();
    DevBuf b[4];
    SiteStatParams p{};
    p.counts = c->d_counts.p; p.site_flags = c->have_flags ? c->flags : nullptr;
    p.n_sites_total = c->n_sites; p.first_site = first_site; p.n_sites = n_sites;
    p.P = c->n_pops; p.n_full = (const int32_t *)c->d_nfull.p;
    p.pairs = (const uint32_t *)c->d_pairs.p; p.n_pairs = (int)c->pairs.size();
    rc = EVOFST_OK;
    if (num && !rc) { rc = ensure(c, b[0], n * 8); p.num = (double *)b[0].p; }
    if (den && !rc) { rc = ensure(c, b[1], n * 8); p.den = (double *)b[1].p; }
    if (dxy && !rc) { rc = ensure(c, b[2], n * 8); p.dxy = (double *)b[2].p; }
    if (include && !rc) { rc = ensure(c, b[3], n); p.include = (uint8_t *)b[3].p; }
    hipError_t e = hipSuccess;
    if (!rc) {
        e = launch_site_stats(p, c->count_bytes, c->stream);
        if (e == hipSuccess && num) e = hipMemcpyAsync(num, b[0].p, n * 8, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess && den) e = hipMemcpyAsync(den, b[1].p, n * 8, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess && dxy) e = hipMemcpyAsync(dxy, b[2].p, n * 8, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess && include) e = hipMemcpyAsync(include, b[3].p, n, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    for (auto &x : b) release(x);
    if (rc) return rc;
    if (e != hipSuccess) return hip_fail(c, e, "get_site_stats");
    return EVOFST_OK;
} catch (...) {
    return abi_exception(c);
}

namespace {
// PBS (arity 3, 3 values per site) and ABS (arity 4, 2 values) share the staging: population tuples up, values down
int branch_site_stats(evofst_ctx *c, const char *what, int arity, int n_out, int64_t first_site, int64_t n_sites, int n_sets,
                      const int32_t *sets, double *out, uint8_t *used) {
    if (!c || !sets || !out || !used || n_sets <= 0) return EVOFST_ERR_ARG;
    NEED_GPU(c);
    int rc = check_range(c, first_site, n_sites);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    if ((rc = need_counts(c, what))) return rc;
    for (int i = 0; i < arity * n_sets; ++i)
        if (sets[i] < 0 || sets[i] >= c->n_pops) return fail(c, EVOFST_ERR_ARG, std::string(what) + ": population index out of range");
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf b[3];
    PbsParams p{};
    p.counts = c->d_counts.p; p.site_flags = c->have_flags ? c->flags : nullptr;
    p.first_site = first_site; p.n_sites = n_sites; p.P = c->n_pops; p.n_trios = n_sets;
    const size_t n = (size_t)n_sets * n_sites, set_bytes = (size_t)n_sets * arity * sizeof(int32_t);
    rc = ensure(c, b[0], set_bytes);
    if (!rc) rc = ensure(c, b[1], n * n_out * sizeof(double));
    if (!rc) rc = ensure(c, b[2], n);
    hipError_t e = hipSuccess;
    if (!rc) {
        p.trios = (const int32_t *)b[0].p; p.pbs = (double *)b[1].p; p.used = (uint8_t *)b[2].p;
        e = hipMemcpyAsync(b[0].p, sets, set_bytes, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = arity == 3 ? launch_pbs_sites(p, c->count_bytes, c->stream) : launch_abs_sites(p, c->count_bytes, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(out, b[1].p, n * n_out * sizeof(double), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(used, b[2].p, n, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    for (auto &x : b) release(x);
    if (rc) return rc;
    if (e != hipSuccess) return hip_fail(c, e, what);
    return EVOFST_OK;
}
}  // namespace

extern "C" int evofst_get_pbs_site_stats(evofst_ctx *c, int64_t first_site, int64_t n_sites, int n_trios, const int32_t *trios,
                                         double *pbs, uint8_t *used) try {
    return branch_site_stats(c, "get_pbs_site_stats", 3, 3, first_site, n_sites, n_trios, trios, pbs, used);
} catch (...) {
    return abi_exception(c);
}

extern "C" int evofst_get_abs_site_stats(evofst_ctx *c, int64_t first_site, int64_t n_sites, int n_quartets, const int32_t *quartets,
                                         double *abs_out, uint8_t *used) try {
    return branch_site_stats(c, "get_abs_site_stats", 4, 2, first_site, n_sites, n_quartets, quartets, abs_out, used);
} catch (...) {
    return abi_exception(c);
}

extern "C" int evofst_get_dist_windows(evofst_ctx *c, int n_windows, const int64_t *win_off, int n_out, const int32_t *out_idx,
                                      int n_in, const int32_t *in_idx, int carry_first, double *sum, int32_t *nsnp, int32_t *miss,
                                      int32_t *used, int32_t *missing) try {
    if (!c || !win_off || !out_idx || !used || !missing || n_windows <= 0 || n_out <= 0 || n_in < 0) return EVOFST_ERR_ARG;
    if (n_in > 0 && (!in_idx || !sum || !nsnp || !miss)) return EVOFST_ERR_ARG;
    NEED_GPU(c);
    if (!c->host_only) { (void)hipSetDevice(c->device); int rc0 = need_counts(c, "get_dist_windows"); if (rc0) return rc0; }
    if (win_off[0] < 0 || win_off[n_windows] > c->n_sites) return fail(c, EVOFST_ERR_ARG, "get_dist_windows: window offsets out of bounds");
    for (int w = 0; w < n_windows; ++w)
        if (win_off[w] > win_off[w + 1]) return fail(c, EVOFST_ERR_ARG, "get_dist_windows: window offsets must not decrease");
    for (int i = 0; i < n_out; ++i)
        if (out_idx[i] < 0 || out_idx[i] >= c->n_pops) return fail(c, EVOFST_ERR_ARG, "get_dist_windows: population index out of range");
    for (int j = 0; j < n_in; ++j)
        if (in_idx[j] < 0 || in_idx[j] >= c->n_pops) return fail(c, EVOFST_ERR_ARG, "get_dist_windows: population index out of range");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t nw = (size_t)n_windows * n_out, nwj = nw * (size_t)n_in;
    // one staging block: sum | win_off | nsnp | miss | used | missing | out_idx | in_idx  (8-byte items first)
    const size_t o_sum = 0, o_off = o_sum + nwj * 8, o_nsnp = o_off + ((size_t)n_windows + 1) * 8, o_miss = o_nsnp + nwj * 4,
                 o_used = o_miss + nwj * 4, o_missing = o_used + nw * 4, o_out = o_missing + nw * 4, o_in = o_out + (size_t)n_out * 4,
                 total = o_in + (size_t)n_in * 4;
    DevBuf b;
    int rc = ensure(c, b, total);
    if (rc) return rc;
    char *d = (char *)b.p;
    DistParams p{};
    p.counts = c->d_counts.p; p.site_flags = c->have_flags ? c->flags : nullptr;
    p.win_off = (const int64_t *)(d + o_off); p.P = c->n_pops; p.n_windows = n_windows; p.n_out = n_out; p.n_in = n_in;
    p.carry_first = carry_first ? 1 : 0;
    p.out_idx = (const int32_t *)(d + o_out); p.in_idx = (const int32_t *)(d + o_in);
    p.sum = (double *)(d + o_sum); p.nsnp = (int32_t *)(d + o_nsnp); p.miss = (int32_t *)(d + o_miss);
    p.used = (int32_t *)(d + o_used); p.missing = (int32_t *)(d + o_missing);
    hipError_t e = hipMemcpyAsync(d + o_off, win_off, ((size_t)n_windows + 1) * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d + o_out, out_idx, (size_t)n_out * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && n_in) e = hipMemcpyAsync(d + o_in, in_idx, (size_t)n_in * 4, hipMemcpyHostToDevice, c->stream);
    if (carry_first) {   // the running values of window 0
        const size_t r = (size_t)n_out * n_in;
        if (e == hipSuccess && r) e = hipMemcpyAsync(d + o_sum, sum, r * 8, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess && r) e = hipMemcpyAsync(d + o_nsnp, nsnp, r * 4, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess && r) e = hipMemcpyAsync(d + o_miss, miss, r * 4, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d + o_used, used, (size_t)n_out * 4, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d + o_missing, missing, (size_t)n_out * 4, hipMemcpyHostToDevice, c->stream);
    }
    if (e == hipSuccess) e = launch_dist_windows(p, c->count_bytes, c->stream);
    if (e == hipSuccess && nwj) e = hipMemcpyAsync(sum, d + o_sum, nwj * 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && nwj) e = hipMemcpyAsync(nsnp, d + o_nsnp, nwj * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && nwj) e = hipMemcpyAsync(miss, d + o_miss, nwj * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(used, d + o_used, nw * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(missing, d + o_missing, nw * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    else (void)hipStreamSynchronize(c->stream);
    release(b);
    if (e != hipSuccess) return hip_fail(c, e, "get_dist_windows");
    return EVOFST_OK;
} catch (...) {
    return abi_exception(c);
}

// Allele numbers of one multi-allelic line as `evo stats` reads them (process_vcf_utils.cpp:385-395)
extern "C" int evofst_alleles_from_vcf_line(const char *line, int n_samples, uint8_t *alleles) try {
    if (!line || !alleles || n_samples <= 0) return EVOFST_ERR_ARG;
    const char *col[9];
    size_t len[9];
    const char *p = line;
    for (int i = 0; i < 9; ++i) {
        col[i] = p;
        const char *t = std::strchr(p, '\t');
        if (!t) return EVOFST_ERR_ARG;
        len[i] = (size_t)(t - p);
        p = t + 1;
    }
    uint32_t fl = 0;
    const int miss_allele = stats_missing_allele(col[4], col[4] + len[4], &fl);
    for (int i = 0; i < n_samples; ++i) {
        if (p[0] == '\0' || p[0] == '\t' || p[1] == '\0' || p[1] == '\t' || p[2] == '\0' || p[2] == '\t') return EVOFST_ERR_ARG;
        const char c0 = p[0], c2 = p[2];
        const int v1 = c0 - '0', v2 = c2 - '0';
        if (c0 == '.' || c2 == '.' || v1 == miss_allele || v2 == miss_allele) { alleles[2 * i] = 255; alleles[2 * i + 1] = 255; }
        else {
            if (v1 < 0 || v1 > 254 || v2 < 0 || v2 > 254) return EVOFST_ERR_ARG;
            alleles[2 * i] = (uint8_t)v1; alleles[2 * i + 1] = (uint8_t)v2;
        }
        const char *t = std::strchr(p + 3, '\t');
        if (!t) {
            if (i != n_samples - 1) return EVOFST_ERR_ARG;
            break;
        }
        p = t + 1;
    }
    return EVOFST_OK;
} catch (...) {
    return abi_exception(nullptr);
}

// How many times getThisVariantCountsSimple draws from rand() on these lines: once per genotype whose two allele
// characters differ, on lines whose first genotype is not written with '|' (process_vcf_utils.cpp:384-387, 268-276)
extern "C" int evofst_stats_phase_draws(int64_t n_lines, const char *const *lines, int n_samples, int n_threads, int64_t *draws) try {
    if (!lines || !draws || n_lines < 0 || n_samples <= 0) return EVOFST_ERR_ARG;
    n_threads = std::max(1, std::min(n_threads, 64));
    std::vector<int64_t> part((size_t)n_threads, 0);
    std::vector<int> bad((size_t)n_threads, 0);
    auto work = [&](int tid) {
        int64_t acc = 0;
        for (int64_t l = tid; l < n_lines; l += n_threads) {
            const char *p = lines[l];
            for (int i = 0; i < 9 && p; ++i) { p = std::strchr(p, '\t'); if (p) ++p; }
            if (!p) { bad[tid] = 1; return; }
            if (p[0] == '\0' || p[1] == '|') continue;
            for (int i = 0; i < n_samples; ++i) {
                if (p[0] == '\0' || p[0] == '\t' || p[1] == '\0' || p[1] == '\t' || p[2] == '\0' || p[2] == '\t') { bad[tid] = 1; return; }
                acc += p[0] != p[2];
                const char *t = std::strchr(p + 3, '\t');
                if (!t) {
                    if (i != n_samples - 1) { bad[tid] = 1; return; }
                    break;
                }
                p = t + 1;
            }
        }
        part[tid] = acc;
    };
    if (n_threads == 1) work(0);
    else {
        std::vector<std::thread> th;
        th.reserve((size_t)n_threads);
        int started = 0;
        for (; started < n_threads - 1; ++started) {
            try { th.emplace_back(work, started); } catch (...) { break; }
        }
        for (int t = started; t < n_threads; ++t) work(t);       // the strides no thread took run here
        for (auto &t : th) t.join();
    }
    int64_t total = 0;
    for (int t = 0; t < n_threads; ++t) { if (bad[t]) return EVOFST_ERR_ARG; total += part[t]; }
    *draws = total;
    return EVOFST_OK;
} catch (...) {
    return abi_exception(nullptr);
}

extern "C" int evofst_set_multiallelic(evofst_ctx *c, int64_t n_multi, const int64_t *site_index, const uint8_t *alleles) try {
    if (!c || n_multi < 0 || (n_multi > 0 && (!site_index || !alleles))) return EVOFST_ERR_ARG;
    NEED_GPU(c);
    if (c->n_sites <= 0) return fail(c, EVOFST_ERR_STATE, "set_multiallelic: upload first");
    for (int64_t k = 0; k < n_multi; ++k)
        if (site_index[k] < 0 || site_index[k] >= c->n_sites || (k && site_index[k] <= site_index[k - 1]))
            return fail(c, EVOFST_ERR_ARG, "set_multiallelic: site indices must ascend inside the shard");
    c->n_multi = 0;
    if (n_multi == 0) return EVOFST_OK;
    HIPCHK(c, hipSetDevice(c->device));
    // from VCF sample order to the layout's individual order (dropped samples leave)
    const int n_ind = c->H / 2;
    std::vector<uint8_t> lay((size_t)n_multi * n_ind * 2);
    for (int64_t k = 0; k < n_multi; ++k)
        for (int h = 0; h < c->H; ++h) lay[(size_t)k * c->H + h] = alleles[(size_t)k * 2 * c->n_samples + c->src_hap[h]];
    int rc = ensure(c, c->d_multi_site, (size_t)n_multi * 8);
    if (!rc) rc = ensure(c, c->d_multi_alleles, lay.size());
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->d_multi_site.p, site_index, (size_t)n_multi * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_multi_alleles.p, lay.data(), lay.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));   // lay goes out of scope
    c->n_multi = n_multi;
    return EVOFST_OK;
} catch (...) {
    return abi_exception(c);
}

extern "C" int evofst_get_indiv_diffs(evofst_ctx *c, int n_blocks, const int64_t *block_off, int32_t *hh, int32_t *hd, int32_t *oh,
                                     int32_t *ms, int32_t *d4) try {
    if (!c || !block_off || !hh || !hd || !oh || !ms || !d4 || n_blocks <= 0) return EVOFST_ERR_ARG;
    NEED_GPU(c);
    if (c->n_sites <= 0 || !c->tiles) return fail(c, EVOFST_ERR_STATE, "get_indiv_diffs: upload first");
    if (n_blocks > 65535) return fail(c, EVOFST_ERR_UNSUPPORTED, "get_indiv_diffs: at most 65535 blocks per call");
    if (block_off[0] < 0 || block_off[n_blocks] > c->n_sites) return fail(c, EVOFST_ERR_ARG, "get_indiv_diffs: block offsets out of bounds");
    for (int b = 0; b < n_blocks; ++b)
        if (block_off[b] > block_off[b + 1]) return fail(c, EVOFST_ERR_ARG, "get_indiv_diffs: block offsets must not decrease");
    HIPCHK(c, hipSetDevice(c->device));
    IndivParams p{};
    p.tiles = (const ulonglong2 *)c->tiles; p.site_flags = c->have_flags ? c->flags : nullptr;
    p.n_sites = c->n_sites; p.W = c->W; p.n_tiles = n_tiles_of(c->n_sites); p.n_ind = c->H / 2;
    p.n_blocks = n_blocks;
    const int nb16 = (p.n_ind + 15) / 16;
    p.n_pair_blocks = nb16 * (nb16 + 1) / 2;
    p.T = (int64_t)p.n_ind * (p.n_ind + 1) / 2;
    p.n_multi = c->n_multi;
    p.multi_site = (const int64_t *)c->d_multi_site.p; p.multi_alleles = (const uint8_t *)c->d_multi_alleles.p;
    const size_t out_n = (size_t)n_blocks * (size_t)p.T;
    int rc = ensure(c, c->d_planes, (size_t)p.n_tiles * p.n_ind * 32);
    DevBuf b;
    const size_t off_bytes = ((size_t)n_blocks + 1) * 8;
    if (!rc) rc = ensure(c, b, off_bytes + out_n * 5 * 4);
    if (rc) { release(b); return rc; }
    char *d = (char *)b.p;
    p.planes = (uint64_t *)c->d_planes.p;
    p.block_off = (const int64_t *)d;
    int32_t *o = (int32_t *)(d + off_bytes);
    p.hh = o; p.hd = o + out_n; p.oh = o + 2 * out_n; p.ms = o + 3 * out_n; p.d4 = o + 4 * out_n;
    hipError_t e = hipMemcpyAsync(d, block_off, off_bytes, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = launch_indiv_planes(p, c->stream);
    if (e == hipSuccess) e = launch_indiv_pairs(p, c->stream);
    if (e == hipSuccess) {
        if (p.n_multi > 0) e = launch_indiv_multi(p, c->stream);
        else e = hipMemsetAsync(p.d4, 0, out_n * 4, c->stream);
    }
    int32_t *host[5] = {hh, hd, oh, ms, d4};
    for (int k = 0; k < 5 && e == hipSuccess; ++k)
        e = hipMemcpyAsync(host[k], o + (size_t)k * out_n, out_n * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    else (void)hipStreamSynchronize(c->stream);
    release(b);
    if (e != hipSuccess) return hip_fail(c, e, "get_indiv_diffs");
    return EVOFST_OK;
} catch (...) {
    return abi_exception(c);
}

extern "C" int evofst_get_indiv_hets(evofst_ctx *c, int32_t *het, int32_t *shared) try {
    if (!c || !het || !shared) return EVOFST_ERR_ARG;
    NEED_GPU(c);
    if (c->n_sites <= 0 || !c->tiles) return fail(c, EVOFST_ERR_STATE, "get_indiv_hets: upload first");
    HIPCHK(c, hipSetDevice(c->device));
    IndivParams p{};
    p.tiles = (const ulonglong2 *)c->tiles; p.site_flags = c->have_flags ? c->flags : nullptr;
    p.n_sites = c->n_sites; p.W = c->W; p.n_tiles = n_tiles_of(c->n_sites); p.n_ind = c->H / 2;
    if (p.n_ind <= 0) return fail(c, EVOFST_ERR_STATE, "get_indiv_hets: the layout has no individual");
    int rc = ensure(c, c->d_planes, (size_t)p.n_tiles * p.n_ind * 32);
    DevBuf b;
    const size_t out_bytes = (size_t)p.n_ind * 4;
    if (!rc) rc = ensure(c, b, 2 * out_bytes);
    if (rc) { release(b); return rc; }
    p.planes = (uint64_t *)c->d_planes.p;
    p.het_cnt = (int32_t *)b.p; p.shared_cnt = p.het_cnt + p.n_ind;
    hipError_t e = hipMemsetAsync(b.p, 0, 2 * out_bytes, c->stream);
    if (e == hipSuccess) e = launch_indiv_planes(p, c->stream);
    if (e == hipSuccess) e = launch_indiv_hets(p, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(het, p.het_cnt, out_bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(shared, p.shared_cnt, out_bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    else (void)hipStreamSynchronize(c->stream);
    release(b);
    if (e != hipSuccess) return hip_fail(c, e, "get_indiv_hets");
    return EVOFST_OK;
} catch (...) {
    return abi_exception(c);
}

extern "C" int evofst_get_pop_site_div(evofst_ctx *c, int64_t first_site, int64_t n_sites, double *het, double *het_nei,
                                       double *pi) try {
    if (!c) return EVOFST_ERR_ARG;
    NEED_GPU(c);
    int rc = check_range(c, first_site, n_sites);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    if ((rc = need_counts(c, "get_pop_site_div"))) return rc;
    const size_t n = (size_t)n_sites * c->n_pops;
    DevBuf b[3];
    PopDivParams p{};
    p.counts = c->d_counts.p; p.first_site = first_site; p.n_sites = n_sites; p.P = c->n_pops;
    p.n_full = (const int32_t *)c->d_nfull.p;
    rc = EVOFST_OK;
    if (het && !rc) { rc = ensure(c, b[0], n * 8); p.het = (double *)b[0].p; }
    if (het_nei && !rc) { rc = ensure(c, b[1], n * 8); p.het_nei = (double *)b[1].p; }
    if (pi && !rc) { rc = ensure(c, b[2], n * 8); p.pi = (double *)b[2].p; }
    hipError_t e = hipSuccess;
    if (!rc) {
        e = launch_pop_div(p, c->count_bytes, c->stream);
        if (e == hipSuccess && het) e = hipMemcpyAsync(het, b[0].p, n * 8, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess && het_nei) e = hipMemcpyAsync(het_nei, b[1].p, n * 8, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess && pi) e = hipMemcpyAsync(pi, b[2].p, n * 8, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    for (auto &x : b) release(x);
    if (rc) return rc;
    if (e != hipSuccess) return hip_fail(c, e, "get_pop_site_div");
    return EVOFST_OK;
} catch (...) {
    return abi_exception(c);
}

extern "C" int evofst_get_positions(evofst_ctx *c, int64_t first_site, int64_t n_sites, int32_t *pos) try {
    if (!c || !pos) return EVOFST_ERR_ARG;
    NEED_GPU(c);
    int rc = check_range(c, first_site, n_sites);
    if (rc) return rc;
    if (!c->have_pos || !c->pos) return fail(c, EVOFST_ERR_STATE, "get_positions: the shard was loaded without positions");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(pos, c->pos + first_site, (size_t)n_sites * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return EVOFST_OK;
} catch (...) {
    return abi_exception(c);
}

extern "C" int evofst_get_pair_range_sums(evofst_ctx *c, int pair, int64_t n_ranges, const int64_t *lo, const int64_t *hi,
                                          const double *init, double *out) try {
    if (!c || !lo || !hi || !out || n_ranges < 0) return EVOFST_ERR_ARG;
    NEED_GPU(c);
    if (c->mode != MODE_PAIRS) return fail(c, EVOFST_ERR_STATE, "get_pair_range_sums: pairs mode first");
    if (pair < 0 || (size_t)pair >= c->pairs.size()) return fail(c, EVOFST_ERR_ARG, "get_pair_range_sums: no such pair");
    if (n_ranges == 0) return EVOFST_OK;
    if (c->n_sites <= 0 || !c->tiles) return fail(c, EVOFST_ERR_STATE, "get_pair_range_sums: no data");
    for (int64_t r = 0; r < n_ranges; ++r)
        if (lo[r] < 0 || hi[r] > c->n_sites || lo[r] > hi[r]) return fail(c, EVOFST_ERR_ARG, "get_pair_range_sums: range outside the shard");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = need_counts(c, "get_pair_range_sums");
    if (rc) return rc;
    const size_t n = (size_t)c->n_sites, P = (size_t)c->n_pops;
    const uint32_t ij = c->pairs[(size_t)pair];
    const size_t pi = ij & 0xffffu, pj = ij >> 16;
    DevBuf b[11];
    rc = EVOFST_OK;
    const size_t sizes[11] = {n * 8, n * 8, n * 8, n, P * n * 8, P * n * 8, P * n * 8, (size_t)n_ranges * 8, (size_t)n_ranges * 8,
                              (size_t)n_ranges * kRangeSumCols * 8, (size_t)n_ranges * kRangeSumCols * 8};
    for (int i = 0; i < 11 && !rc; ++i) rc = ensure(c, b[i], sizes[i]);
    hipError_t e = hipSuccess;
    if (!rc) {
        // the pair's per-site values (site_stats_kernel on a one-pair list: the pair's entry of the context's list) and the
        // per-population diversities, left on the device
        SiteStatParams sp{};
        sp.counts = c->d_counts.p; sp.site_flags = c->have_flags ? c->flags : nullptr;
        sp.n_sites_total = c->n_sites; sp.first_site = 0; sp.n_sites = c->n_sites;
        sp.P = c->n_pops; sp.n_full = (const int32_t *)c->d_nfull.p;
        sp.pairs = (const uint32_t *)c->d_pairs.p + pair; sp.n_pairs = 1;
        sp.num = (double *)b[0].p; sp.den = (double *)b[1].p; sp.dxy = (double *)b[2].p; sp.include = (uint8_t *)b[3].p;
        e = launch_site_stats(sp, c->count_bytes, c->stream);
        PopDivParams dp{};
        dp.counts = c->d_counts.p; dp.first_site = 0; dp.n_sites = c->n_sites; dp.P = c->n_pops;
        dp.n_full = (const int32_t *)c->d_nfull.p;
        dp.het = (double *)b[4].p; dp.het_nei = (double *)b[5].p; dp.pi = (double *)b[6].p;
        if (e == hipSuccess) e = launch_pop_div(dp, c->count_bytes, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(b[7].p, lo, (size_t)n_ranges * 8, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(b[8].p, hi, (size_t)n_ranges * 8, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess && init) e = hipMemcpyAsync(b[9].p, init, sizes[9], hipMemcpyHostToDevice, c->stream);
        RangeSumParams rp{};
        rp.num = sp.num; rp.den = sp.den; rp.dxy = sp.dxy; rp.include = sp.include;
        rp.het_i = dp.het + pi * n; rp.het_j = dp.het + pj * n; rp.nei_i = dp.het_nei + pi * n; rp.nei_j = dp.het_nei + pj * n;
        rp.pi_i = dp.pi + pi * n; rp.pi_j = dp.pi + pj * n;
        rp.n_sites = c->n_sites; rp.n_ranges = n_ranges;
        rp.lo = (const int64_t *)b[7].p; rp.hi = (const int64_t *)b[8].p; rp.init = init ? (const double *)b[9].p : nullptr;
        rp.out = (double *)b[10].p;
        if (e == hipSuccess) e = launch_pair_range_sums(rp, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(out, b[10].p, sizes[10], hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    for (auto &x : b) release(x);
    if (rc) return rc;
    if (e != hipSuccess) return hip_fail(c, e, "get_pair_range_sums");
    return EVOFST_OK;
} catch (...) {
    return abi_exception(c);
}

extern "C" int64_t evofst_max_segments(evofst_ctx *c) {
    if (!c || !c->stats_done || c->mode != MODE_PAIRS) return 0;
    return c->seg_cap;
}

static int segment_rows(evofst_ctx *c, std::vector<int64_t> &first, std::vector<int64_t> &rows) {
    const size_t np = c->pairs.size();
    { int rc0 = fused_verdict(c); if (rc0) return rc0; }
    std::vector<int64_t> tot(np);
    int rc = d2h(c, tot.data(), c->d_incl_totals.p, np);
    if (rc) return rc;
    if (!c->base_host_valid) {
        if ((rc = d2h(c, c->base.data(), c->d_base.p, c->base.size()))) return rc;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->base_host_valid = true;
    first.resize(np); rows.resize(np);
    for (size_t q = 0; q < np; ++q) {
        first[q] = c->base[q] / c->g;
        rows[q] = (c->base[q] + tot[q]) / c->g - first[q];
    }
    return EVOFST_OK;
}

extern "C" int evofst_get_segments(evofst_ctx *c, int64_t *first_segment, int64_t *n_segments, double *sum_num,
                                   double *sum_den, double *sum_dxy_k, int32_t *end_site) try {
    if (!c) return EVOFST_ERR_ARG;
    NEED_GPU(c);
    if (!c->stats_done || c->mode != MODE_PAIRS) return fail(c, EVOFST_ERR_STATE, "get_segments: pairs mode + run_stats first");
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<int64_t> first, rows;
    int rc = segment_rows(c, first, rows);
    if (rc) return rc;
    const size_t np = c->pairs.size();
    if (first_segment) std::copy(first.begin(), first.end(), first_segment);
    if (n_segments) std::copy(rows.begin(), rows.end(), n_segments);
    const size_t n = (size_t)c->seg_cap * np;
    // the device keeps one 32-byte row per segment and pair; the getter hands out the four arrays
    if (sum_num && (rc = ensure(c, c->d_seg_num, n * 8))) return rc;
    if (sum_den && (rc = ensure(c, c->d_seg_den, n * 8))) return rc;
    if (sum_dxy_k && (rc = ensure(c, c->d_seg_k, n * 8))) return rc;
    if (end_site && (rc = ensure(c, c->d_seg_end, n * 4))) return rc;
    HIPCHK(c, launch_unpack_segments((const SegRow *)c->d_seg_rows.p, n, sum_num ? (double *)c->d_seg_num.p : nullptr,
                                     sum_den ? (double *)c->d_seg_den.p : nullptr, sum_dxy_k ? (double *)c->d_seg_k.p : nullptr,
                                     end_site ? (int32_t *)c->d_seg_end.p : nullptr, c->stream));
    if ((rc = d2h(c, sum_num, c->d_seg_num.p, n))) return rc;
    if ((rc = d2h(c, sum_den, c->d_seg_den.p, n))) return rc;
    if ((rc = d2h(c, sum_dxy_k, c->d_seg_k.p, n))) return rc;
    if ((rc = d2h(c, end_site, c->d_seg_end.p, n))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return EVOFST_OK;
} catch (...) {
    return abi_exception(c);
}

static int launch_totals_and_carry(evofst_ctx *c, bool want_totals, double *d_carry_out) {
    const size_t np = c->pairs.size();
    int rc;
    if ((rc = ensure(c, c->d_totals, np * 4 * 8))) return rc;
    if ((rc = ensure(c, c->d_carry, np * 4 * 8))) return rc;
    K2Params p = make_k2(c);
    HIPCHK(c, launch_totals(p, (const int64_t *)c->d_base.p, (const int64_t *)c->d_incl_totals.p,
                            want_totals ? (double *)c->d_totals.p : nullptr, d_carry_out ? d_carry_out : (double *)c->d_carry.p, c->stream));
    return EVOFST_OK;
}

static int totals_and_carry(evofst_ctx *c, double *totals, double *carry) {
    const size_t np = c->pairs.size();
    int rc = fused_verdict(c);
    if (rc) return rc;
    rc = launch_totals_and_carry(c, totals != nullptr);
    if (rc) return rc;
    if ((rc = d2h(c, totals, c->d_totals.p, np * 4))) return rc;
    if ((rc = d2h(c, carry, c->d_carry.p, np * 4))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return EVOFST_OK;
}

extern "C" int evofst_get_boundary(evofst_ctx *c, double *carry_out) try {
    if (!c || !carry_out) return EVOFST_ERR_ARG;
    NEED_GPU(c);
    if (!c->stats_done || c->mode != MODE_PAIRS) return fail(c, EVOFST_ERR_STATE, "get_boundary: pairs mode + run_stats first");
    HIPCHK(c, hipSetDevice(c->device));
    return totals_and_carry(c, nullptr, carry_out);
} catch (...) {
    return abi_exception(c);
}

extern "C" int evofst_get_totals(evofst_ctx *c, double *totals) try {
    if (!c || !totals) return EVOFST_ERR_ARG;
    NEED_GPU(c);
    if (!c->stats_done || c->mode != MODE_PAIRS) return fail(c, EVOFST_ERR_STATE, "get_totals: pairs mode + run_stats first");
    HIPCHK(c, hipSetDevice(c->device));
    return totals_and_carry(c, totals, nullptr);
} catch (...) {
    return abi_exception(c);
}

extern "C" int evofst_get_global_site_stats(evofst_ctx *c, int64_t first_site, int64_t n_sites, double *num, double *den,
                                            uint8_t *kept) try {
    if (!c) return EVOFST_ERR_ARG;
    NEED_GPU(c);
    if (!c->counts_done || c->mode != MODE_GLOBAL) return fail(c, EVOFST_ERR_STATE, "get_global_site_stats: against-all mode + run_counts first");
    int rc = check_range(c, first_site, n_sites);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t nu = c->use_idx.size();
    const size_t n = (size_t)n_sites * nu;
    DevBuf b[3];
    GlobalParams p = make_global(c);
    p.first_site = first_site; p.n_out_sites = n_sites;
    rc = EVOFST_OK;
    if (num && !rc) { rc = ensure(c, b[0], n * 8); p.site_num = (double *)b[0].p; }
    if (den && !rc) { rc = ensure(c, b[1], n * 8); p.site_den = (double *)b[1].p; }
    if (kept && !rc) { rc = ensure(c, b[2], (size_t)n_sites); p.site_kept = (uint8_t *)b[2].p; }
    hipError_t e = hipSuccess;
    if (!rc) {
        e = launch_global_sites(p, c->count_bytes, c->stream);
        if (e == hipSuccess && num) e = hipMemcpyAsync(num, b[0].p, n * 8, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess && den) e = hipMemcpyAsync(den, b[1].p, n * 8, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess && kept) e = hipMemcpyAsync(kept, b[2].p, (size_t)n_sites, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    for (auto &x : b) release(x);
    if (rc) return rc;
    if (e != hipSuccess) return hip_fail(c, e, "get_global_site_stats");
    return EVOFST_OK;
} catch (...) {
    return abi_exception(c);
}

extern "C" int64_t evofst_max_global_rows(evofst_ctx *c) {
    if (!c || !c->stats_done || c->mode != MODE_GLOBAL) return 0;
    return c->row_cap;
}

extern "C" int evofst_get_global_windows(evofst_ctx *c, int64_t *n_rows, int32_t *start_site, int32_t *end_site,
                                         double *mean_num, double *mean_den) try {
    if (!c || !n_rows) return EVOFST_ERR_ARG;
    NEED_GPU(c);
    if (!c->stats_done || c->mode != MODE_GLOBAL) return fail(c, EVOFST_ERR_STATE, "get_global_windows: against-all mode + run_stats first");
    HIPCHK(c, hipSetDevice(c->device));
    int64_t rows = 0;
    int rc = d2h(c, &rows, c->d_row_tot.p, 1);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *n_rows = rows;
    const size_t nu = c->use_idx.size();
    if ((rc = d2h(c, start_site, c->d_row_start.p, (size_t)rows))) return rc;
    if ((rc = d2h(c, end_site, c->d_row_end.p, (size_t)rows))) return rc;
    if ((rc = d2h(c, mean_num, c->d_row_num.p, (size_t)rows * nu))) return rc;
    if ((rc = d2h(c, mean_den, c->d_row_den.p, (size_t)rows * nu))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return EVOFST_OK;
} catch (...) {
    return abi_exception(c);
}
