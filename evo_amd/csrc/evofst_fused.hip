// evofst_fused.hip -- pairs mode as ONE streaming kernel (gfx950, wave64).
//
// Replaces, for every population pair at once, the per-line work of getFstFromVCF: the counting loops
// (process_vcf_fst.cpp:243-280), the site filter (:543-544), calculateFstNumerator / Denominator
// (process_vcf_fst.h:19-30), calculateDxy (process_vcf_fst.cpp:297-323) and the SNP windows (:638-650).
//
// A workgroup works through chunks of a few tiles: its first chunk is its own index, the later ones come in ticket
// order (one atomic per chunk).  Per chunk:
//   phase A  lane = site: 16-byte coalesced loads of the tile rows, popcounts under wave-uniform masks that the
//            scalar unit forms from the population sizes while the vector unit counts, packed {alt, non-missing}
//            counts and the zero / fixed ballots of every population INTO LDS (nothing but the tiles is read
//            from HBM and no counts are written to it);
//   census   lane = pair: included sites of the chunk from the ballots (filter :543-544);
//   look-back the chunk publishes its census as 8-byte {epoch, count} granules and sums those of the chunks
//            before it (three levels of fan-out 32: chunks of its group, groups of its super-group,
//            super-groups; the owner of a group's last chunk has read all the others of the group for its
//            own prefix, so it publishes the group's aggregate for free).  A chunk only ever waits for chunks
//            with smaller indices and phase A waits for nothing, so the waits end as long as workgroups start
//            in index order or all fit on the device (the grid is sized by the occupancy query); spins are
//            bounded and give up through an error word that every workgroup polls;
//   phase B  the counts in LDS -> per-population tables (p, prefix sums of h, t, c) per group of SUB sites,
//            lane = pair walks the sites, segment sums at the window boundaries (K2's scheme, see
//            evofst_kernels.hip), with accumulators that restart with every (sub-)chunk of a few hundred sites.
// Two shapes: up to 64 pairs -- every wave of the workgroup owns a quarter of the chunk's tiles through phases A
// and B with tables of its own (no workgroup barrier inside them); more pairs -- the waves share one set of tables
// and split the pairs.  The segment a (sub-)chunk starts in the middle of is completed by stitch_kernel from the
// tails, as before.
#include "evofst_internal.h"

#include <algorithm>

namespace evofst {

constexpr int kFan = 32;                  // look-back fan-out per level
constexpr int kFuseNB = 8;                // tile rows per load batch (8 KiB per wave)
constexpr int kFuseWaves = 4;             // waves per workgroup in the wave-private shape
#ifndef EVOFST_FUSED_WPS
#define EVOFST_FUSED_WPS 2             // waves per SIMD the register allocation aims at
#endif
constexpr uint32_t kSpinLimit = 1u << 22; // polls before a wait gives up (seconds; a healthy wait takes microseconds)

static __host__ __device__ constexpr size_t a16(size_t x) { return (x + 15) & ~(size_t)15; }

// LDS of one region: the tables of phase B and the counts / ballots of `tiles` tiles.
static size_t fused_region_bytes(int P, int tiles, int count_bytes, int sub) {
    const size_t subp = (size_t)sub + 1;
    size_t b = 0;
    b += 3 * a16((size_t)P * subp * 8);         // pt, PH, PT
    b += 2 * a16((size_t)P * 8);                // carH, carT
    b += 2 * a16((size_t)P * subp * 4);         // xt, PC
    b += a16((size_t)P * 4);                    // carC
    b += a16((size_t)P * ((size_t)sub + 8));    // c of the fill as bytes
    b += a16((size_t)tiles * P * kTileSites * count_bytes);
    b += a16((size_t)tiles * 2 * P * 8);        // zero / fixed ballots
    b += 4 * a16((size_t)tiles * 8);            // irregular, skip, any-missing, NaN sites
    return b;
}

size_t fused_lds_bytes(int P, int n_segs, int max_n, int n_pairs, int chunk_tiles, int count_bytes, int sub) {
    const bool wp = n_pairs <= 64;
    size_t b = a16(max_n <= kRcpMax ? (size_t)(max_n + 1) * sizeof(RcpEntry) : 0) + 64 + a16((size_t)(n_segs + 1) * sizeof(SegDesc));
    if (wp) {
        b += a16((size_t)kFuseWaves * 64 * 4) + a16((size_t)kFuseWaves * 3 * 64 * 4);   // census and look-back partials of the waves
        b += kFuseWaves * fused_region_bytes(P, chunk_tiles / kFuseWaves, count_bytes, sub);
    } else {
        b += fused_region_bytes(P, chunk_tiles, count_bytes, sub);
    }
    return b;
}

typedef __attribute__((address_space(1))) uint64_t gu64;
typedef __attribute__((address_space(1))) uint32_t gu32;

__device__ __forceinline__ void store_granule(uint64_t *g, uint32_t epoch, uint32_t value) {
    __hip_atomic_store((gu64 *)g, ((uint64_t)epoch << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t load_granule(const uint64_t *g) {
    return __hip_atomic_load((gu64 *)g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The granule rows a chunk adds up for its prefix, as one list: chunks [grp0 * 32, chunk) of level 0, groups
// [grp1 * 32, grp0) of level 1, super-groups [0, grp1) of level 2.  This thread takes rows part, part + nparts, ...
// of the list and column q; s[level] get the sums.  Waits for rows that are not yet published; false when the wait
// gave up (error word set, by this or by another thread).
__device__ __forceinline__ bool sum_granule_rows(const FusedParams &p, int chunk, int q, int part, int nparts, uint32_t (&s)[3]) {
    constexpr int B = 8;
    const int grp0 = chunk / kFan, grp1 = grp0 / kFan;
    const int n0 = chunk - grp0 * kFan, n1 = grp0 - grp1 * kFan, n2 = grp1;
    const int total = n0 + n1 + n2;
    const size_t np = (size_t)p.n_pairs;
    uint32_t spins = 0;
    for (int r0 = part; r0 < total; r0 += B * nparts) {
        const uint64_t *ptr[B];
        int lev[B];
#pragma unroll
        for (int u = 0; u < B; ++u) {
            const int r = min(r0 + u * nparts, total - 1);
            lev[u] = r < n0 ? 0 : (r < n0 + n1 ? 1 : 2);
            ptr[u] = lev[u] == 0 ? p.agg0 + (size_t)(grp0 * kFan + r) * np + q
                   : lev[u] == 1 ? p.agg1 + (size_t)(grp1 * kFan + r - n0) * np + q
                                 : p.agg2 + (size_t)(r - n0 - n1) * np + q;
        }
        uint64_t x[B];
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int u = 0; u < B; ++u) x[u] = load_granule(ptr[u]);
#pragma unroll
            for (int u = 0; u < B; ++u) ok = ok && (uint32_t)(x[u] >> 32) == p.epoch;
            if (ok) break;
            if (++spins > kSpinLimit) { __hip_atomic_store((gu32 *)p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
            if ((spins & 255u) == 0u && __hip_atomic_load((gu32 *)p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
            __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int u = 0; u < B; ++u) {
            const uint32_t v = r0 + u * nparts < total ? (uint32_t)x[u] : 0u;
            s[0] += lev[u] == 0 ? v : 0u; s[1] += lev[u] == 1 ? v : 0u; s[2] += lev[u] == 2 ? v : 0u;
        }
    }
    return true;
}

// Phase A for the tiles of one wave: tile_first, tile_first + tile_stride, ... (n_my of them) go to slots slot_first,
// slot_first + slot_stride, ... of a region.  All arguments but `lane` are wave-uniform.
//
// A tile is W rows of 64 x 16 bytes {ALT word, MISSING word}; the wave reads them in batches of 8 rows (8 KiB) and
// requests the next batch -- of this tile or of the next -- before it counts the current one.  Counting walks the
// (word, population) overlaps of the layout, a stream of 16-byte descriptors in LDS that is read one descriptor ahead
// of its use, so the loop waits for nothing but the tile rows.
template <typename CT>
__device__ __forceinline__ void count_wave_tiles(const FusedParams &p, int lane, int n_my, int tile_first, int tile_stride,
                                                 int slot_first, int slot_stride, CT *lcnt, uint64_t *lzf, uint64_t *lirr,
                                                 uint64_t *lskip, uint64_t *lmiss, uint64_t *lnan, const uint4 *lsegs) {
    const int P = p.P;
    const int nb = (p.W + kFuseNB - 1) / kFuseNB;      // load batches per tile
    const int n_items = n_my * nb;
    ulonglong2 cur[kFuseNB], nxt[kFuseNB];
    auto load_item = [&](ulonglong2 (&r)[kFuseNB], int item) {
        const int tj = item / nb, wb = (item - tj * nb) * kFuseNB;
        const ulonglong2 *tp = p.tiles + (size_t)(tile_first + tj * tile_stride) * p.W * kTileSites + lane;
#pragma unroll
        for (int u = 0; u < kFuseNB; ++u)   // unconditional (clamped): all loads of a batch are in flight together
            r[u] = tp[(size_t)min(wb + u, p.W - 1) * kTileSites];
    };
#ifdef EVOFST_EXPERIMENTS
    if (p.debug & 4) {
#pragma unroll
        for (int u = 0; u < kFuseNB; ++u) cur[u] = nxt[u] = make_ulonglong2(0x5555555555555555ull * (lane & 1), 0);
    }
#define EVOFST_FUSED_LOAD(r, item) do { if (!(p.debug & 4)) load_item(r, item); } while (0)
#else
#define EVOFST_FUSED_LOAD(r, item) load_item(r, item)
#endif
    if (n_items > 0) EVOFST_FUSED_LOAD(cur, 0);
    uint32_t c = 0, mc = 0;
    uint32_t irregular = 0, anymiss = 0, nanany = 0;   // per-lane 0 / 1 words, not lane masks merged at every join (see K1)
    bool skip = false;
    uint64_t skipmask = 0;
    int sidx = 0;                          // next descriptor of the stream
    uint4 d = lsegs[0];                    // {mask lo, mask hi, pop, flags | n_full << 8}
    for (int item = 0; item < n_items; ++item) {
        const int tj = item / nb, bi = item - tj * nb;
        const int slot = slot_first + tj * slot_stride, wb = bi * kFuseNB;
        if (item + 1 < n_items) EVOFST_FUSED_LOAD(nxt, item + 1);
        if (bi == 0) {
            const int64_t site = (int64_t)(tile_first + tj * tile_stride) * kTileSites + lane;
            skip = site >= p.n_sites;
            if (!skip && p.site_flags) skip = (p.site_flags[site] & 1u) != 0;
            skipmask = __ballot(skip);
            c = 0; mc = 0; irregular = 0; anymiss = 0; nanany = 0;
            sidx = 0; d = lsegs[0];
        }
        CT *cp = lcnt + (size_t)slot * P * kTileSites + lane;
        uint64_t *zp = lzf + (size_t)slot * 2 * P;
#ifdef EVOFST_EXPERIMENTS
        if (!(p.debug & 32))
#endif
#pragma unroll
        for (int u = 0; u < kFuseNB; ++u) {
            if (wb + u < p.W) {
                const uint32_t xlo = (uint32_t)cur[u].x, xhi = (uint32_t)(cur[u].x >> 32);
                const uint32_t ylo = (uint32_t)cur[u].y, yhi = (uint32_t)(cur[u].y >> 32);
                bool more = true;
                while (more) {
                    const uint4 dn = lsegs[sidx + 1];      // the table ends with one spare entry
                    c = popc_acc(xlo & d.x, c); c = popc_acc(xhi & d.y, c);
                    mc = popc_acc(ylo & d.x, mc); mc = popc_acc(yhi & d.y, mc);
                    const uint32_t fl = d.w;
                    if (fl & kSegLast) {                   // population d.z is complete
                        const uint32_t nfull = fl >> kSegNFullShift;
                        const uint32_t nm = nfull - mc;
                        cp[(size_t)d.z * kTileSites] = (CT)(c | (nm << CountTraits<CT>::shift));
                        const uint64_t z = __ballot(c == 0) | skipmask;
                        const uint64_t f = __ballot(c == nm) | skipmask;
                        if (lane == 0) { zp[d.z] = z; zp[P + d.z] = f; }
                        // sites phase B must take through its general loop: a NaN in the reference (nm < 2) or an
                        // excluded site with a dxy term (fixed with missing alleles)
                        const uint32_t few = nm < 2u ? 1u : 0u;
                        irregular |= few | ((c == nm && nm < nfull) ? 1u : 0u);
                        nanany |= few;
                        anymiss |= mc != 0 ? 1u : 0u;
                        c = 0; mc = 0;
                    }
                    more = (fl & kSegWordEnd) == 0u;
                    d = dn; ++sidx;
                }
            }
        }
        if (bi == nb - 1) {
            const uint64_t im = __ballot(irregular != 0 && !skip), mm = __ballot(anymiss != 0), nn = __ballot(nanany != 0 && !skip);
            if (lane == 0) { lirr[slot] = im; lskip[slot] = skipmask; lmiss[slot] = mm; lnan[slot] = nn; }
        }
        if (item + 1 < n_items) {
#pragma unroll
            for (int u = 0; u < kFuseNB; ++u) cur[u] = nxt[u];
        }
    }
}

// WP: wave-private shape (n_pairs <= 64).  SUB: sites per table fill (16 or 32).
// HBM: the statistics half on its own (evofst_run_stats after evofst_run_counts, the multi-GPU pass): a region is one of
// K1's chunks, its counts and ballots come from HBM in pieces of a few tiles (the next piece is requested before the
// current one is walked) and its offsets from the scan kernel; no phase A, no look-back, no tickets.
// W3 (HBM mode, shared tables): registers for three waves per SIMD instead of two.  The walk is bound by the latency of
// its LDS round trips, not by issue slots: where the LDS holds three workgroups per CU the third wave per SIMD is worth
// more than the two dozen registers it spills (cfg4 shard with missing calls: 1.81 -> 1.40 ms); where the LDS holds two
// anyway (wave-private tables, many populations) it would only add the spills (cfg2 with missing calls: 93 -> 106 us).
#ifndef EVOFST_WALK_WPS
#define EVOFST_WALK_WPS 2
#endif
template <typename CT, bool use_rcp, bool WP, int SUB, bool HBM, bool W3 = false>
__global__ __launch_bounds__(kK2Threads, HBM ? (W3 ? 3 : EVOFST_WALK_WPS) : EVOFST_FUSED_WPS) void fused_pairs_kernel(FusedParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    // HBM mode with K1's / the stream pass's marks: which turns of this workgroup (regions a grid's stride apart) hold a
    // marked chunk, as a bit map in LDS -- one thread per turn, all loads at once, so a launch that finds nothing costs one
    // memory round trip and a workgroup with many turns (a small grid over a long shard) walks straight to its marked ones
    constexpr int kMarkWords = 512;
    __shared__ uint32_t marked_turns[HBM ? kMarkWords : 1];
    int n_turns = 0;
    bool have_marks = false;
    if (HBM && p.chunk_dirty != nullptr) {
        const int per = WP ? (int)(blockDim.x >> 6) : 1;
        const int n_regions = (p.n_chunks + per - 1) / per;
        n_turns = (int)blockIdx.x < n_regions ? (n_regions - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
        have_marks = n_turns <= kMarkWords * 32;
        for (int i = threadIdx.x; i < kMarkWords; i += blockDim.x) marked_turns[i] = 0u;
        __syncthreads();
        int any = 0;
        for (int it = threadIdx.x; it < n_turns; it += blockDim.x) {
            const int64_t kc_ = ((int64_t)blockIdx.x + (int64_t)it * gridDim.x) * per;
            int m = 0;
            for (int w = 0; w < per; ++w) m |= (kc_ + w < p.n_chunks && p.chunk_dirty[kc_ + w] != 0u) ? 1 : 0;
            if (m && have_marks) atomicOr(&marked_turns[it >> 5], 1u << (it & 31));
            any |= m;
        }
        if (!__syncthreads_or(any)) return;
    }
    constexpr int SUBP = SUB + 1;                 // row stride in entries: pop-indexed reads stay conflict-free
    constexpr int SPT = kTileSites / SUB;         // table fills per tile
    constexpr uint32_t FULL = SUB == 32 ? 0xffffffffu : ((1u << (SUB & 31)) - 1u);
    constexpr int EPT = 16 / (int)sizeof(CT);     // sites per 16-byte count vector
    constexpr int VPR = SUB / EPT;                // vectors per population row of a fill
    constexpr int EB = 8 * (int)sizeof(CT);
    const int P = p.P;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nthreads = blockDim.x, NWV = nthreads >> 6;
    const int TR = HBM ? p.piece_tiles : (WP ? p.chunk_tiles / NWV : p.chunk_tiles);     // tiles of a region (HBM: of one piece of it)
    const int TRL = HBM ? 2 * TR : TR;            // tiles a region holds in LDS (HBM: the piece being walked and the one arriving)
    size_t off = 0;
    auto carve = [&](size_t bytes) { unsigned char *r = smem + off; off += (bytes + 15) & ~(size_t)15; return r; };
    RcpEntry *rcp = reinterpret_cast<RcpEntry *>(carve(use_rcp ? (size_t)(p.max_n + 1) * sizeof(RcpEntry) : 0));
    int32_t *lmisc = reinterpret_cast<int32_t *>(carve(64));
    uint4 *lsegs = reinterpret_cast<uint4 *>(carve((size_t)(p.n_segs + 1) * sizeof(SegDesc)));   // descriptor stream + one spare entry
    uint32_t *wcen = nullptr, *lb = nullptr;
    if (WP) {
        wcen = reinterpret_cast<uint32_t *>(carve((size_t)kFuseWaves * 64 * 4));        // [wave][pair] census of the wave's tiles
        lb = reinterpret_cast<uint32_t *>(carve((size_t)kFuseWaves * 3 * 64 * 4));      // [wave][level][pair] look-back partial sums
        const size_t rb = ((size_t)3 * (((size_t)P * SUBP * 8 + 15) & ~(size_t)15)) + 2 * (((size_t)P * 8 + 15) & ~(size_t)15) +
                          2 * (((size_t)P * SUBP * 4 + 15) & ~(size_t)15) + (((size_t)P * 4 + 15) & ~(size_t)15) +
                          (((size_t)P * (SUB + 8) + 15) & ~(size_t)15) +
                          (((size_t)TRL * P * kTileSites * sizeof(CT) + 15) & ~(size_t)15) + (((size_t)TRL * 2 * P * 8 + 15) & ~(size_t)15) +
                          4 * (((size_t)TRL * 8 + 15) & ~(size_t)15);
        off += (size_t)wave * rb;
    }
    double *pt = reinterpret_cast<double *>(carve((size_t)P * SUBP * 8));   // [P][SUB + 1] p = c / nm
    double *PH = reinterpret_cast<double *>(carve((size_t)P * SUBP * 8));   // sum of h from the start of the (sub-)chunk
    double *PT = reinterpret_cast<double *>(carve((size_t)P * SUBP * 8));   // sum of t
    double *carH = reinterpret_cast<double *>(carve((size_t)P * 8));        // [P] the same sums after the last derived fill
    double *carT = reinterpret_cast<double *>(carve((size_t)P * 8));
    uint32_t *xt = reinterpret_cast<uint32_t *>(carve((size_t)P * SUBP * 4));   // c | flags
    uint32_t *PC = reinterpret_cast<uint32_t *>(carve((size_t)P * SUBP * 4));   // sum of c
    uint32_t *carC = reinterpret_cast<uint32_t *>(carve((size_t)P * 4));
    // the fill's c as bytes, [P][SUB + 8] (rows 8-byte aligned): the regular sums loop takes c_i c_j of four sites with one
    // v_dot4_u32_u8 on two 4-byte reads instead of four 24-bit multiplies on eight (the loop is bound by the LDS bytes it reads)
    constexpr int SUBB = SUB + 8;
    unsigned char *cbt = carve((size_t)P * SUBB);
    CT *lcnt = reinterpret_cast<CT *>(carve((size_t)TRL * P * kTileSites * sizeof(CT)));  // [tile][P][64]
    uint64_t *lzf = reinterpret_cast<uint64_t *>(carve((size_t)TRL * 2 * P * 8));         // [tile][z|f][P]
    uint64_t *lirr = reinterpret_cast<uint64_t *>(carve((size_t)TRL * 8));
    uint64_t *lskip = reinterpret_cast<uint64_t *>(carve((size_t)TRL * 8));
    uint64_t *lmiss = reinterpret_cast<uint64_t *>(carve((size_t)TRL * 8));   // sites where some population has a missing allele
    uint64_t *lnan = reinterpret_cast<uint64_t *>(carve((size_t)TRL * 8));    // sites where some population has fewer than two called alleles

    const int dtid = WP ? lane : tid;          // thread index and count of the group that shares one region
    const int dn = WP ? 64 : nthreads;
    const int q = dtid;
    const bool active = q < p.n_pairs;
    uint32_t pi = 0, pj = 0, ni = 0, nj = 0;
    int64_t b0 = 0;
    if (active) {
        const uint32_t ij = p.pairs[q];
        pi = ij & 0xffffu; pj = ij >> 16;
        ni = (uint32_t)p.n_full[pi]; nj = (uint32_t)p.n_full[pj];
        b0 = p.base[q];
    }
    const int64_t first_seg = b0 / p.g;
    const uint32_t io = pi * SUBP, jo = pj * SUBP;
    // clean fills (no missing allele, no skipped site; 8-bit counts): exact integer sums, scaled once per segment
    constexpr bool kInt = sizeof(CT) == 2;
    const uint64_t ni2 = (uint64_t)ni * ni, nj2 = (uint64_t)nj * nj, ninj = (uint64_t)ni * nj;
    const double rE = active ? 1.0 / ((double)ninj * (double)ninj) : 0.0;          // d^2 = (c_i n_j - c_j n_i)^2 / (n_i n_j)^2
    const double rTi = active ? 1.0 / (double)ni2 : 0.0, rTj = active ? 1.0 / (double)nj2 : 0.0;   // t = c (n - c) / n^2
    const double rHi = ni > 1 ? rTi / (double)(ni - 1) : 0.0, rHj = nj > 1 ? rTj / (double)(nj - 1) : 0.0;   // h = t / (n - 1)
    if (!HBM)
        for (int i = tid; i <= p.n_segs; i += nthreads)
            lsegs[i] = i < p.n_segs ? reinterpret_cast<const uint4 *>(p.segs)[i] : make_uint4(0u, 0u, 0u, kSegWordEnd);
    if (use_rcp)
        for (int i = tid; i <= p.max_n; i += nthreads) {
            RcpEntry e;
            e.r0 = i ? 1.0 / (double)i : 0.0;
            e.rr0 = e.r0 * e.r0;
            e.r1 = i > 1 ? 1.0 / (double)(i - 1) : 0.0;
            e.flags = (i < 2 ? kXBadNum : 0u) | (i == 0 ? kXBadDen : 0u);
            e.pad = 0;
            rcp[i] = e;
        }
    const double kNaN = __longlong_as_double(0x7ff8000000000000ll);
    bool dead = false;                                 // a wait gave up somewhere: draw the remaining tickets, do nothing
    auto region_sync = [&]() {                         // orders the LDS traffic of the threads that share a region
        if (WP) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
        else __syncthreads();
    };

    for (int iter = 0;; ++iter) {
        if (HBM && have_marks) {                       // on to the next marked turn
            while (iter < n_turns) {
                const uint32_t w = marked_turns[iter >> 5] >> (iter & 31);
                if (w & 1u) break;
                iter = w == 0u ? (iter | 31) + 1 : iter + __ffs((int)w) - 1;
            }
            if (iter >= n_turns) break;
        }
        if (HBM) {                                     // regions in turn, a grid's stride apart; no ticket, nothing to wait for
            const int region = (int)blockIdx.x + iter * (int)gridDim.x, per = WP ? NWV : 1;
            if (region >= (p.n_chunks + per - 1) / per) break;
            if (p.chunk_dirty != nullptr) {            // clean_walk_kernel takes the chunks K1 left unmarked
                bool any = false;
                for (int w = 0; w < per; ++w) { const int kc_ = region * per + w; any = any || (kc_ < p.n_chunks && p.chunk_dirty[kc_] != 0u); }
                if (!any) continue;
            }
        }
        __syncthreads();                               // the tables above; the previous chunk's readers of lmisc / LDS
        if (tid == 0) {
            int32_t chunk = (int32_t)blockIdx.x;       // first chunk: the workgroup's own index (the grid is not larger than n_chunks)
            if (HBM) chunk += iter * (int32_t)gridDim.x;   // HBM: regions (one chunk, or four of K1's) in turn, a grid's stride apart
            if (!HBM && iter > 0) {
                // n_chunks draws in all: every workgroup draws until its chunk is past the end
                const uint32_t t = __hip_atomic_fetch_add((gu32 *)p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // the last draw of the launch: nobody adds after it, so the counter can go back to zero for the next launch
                if (t == (uint32_t)p.n_chunks - 1u) __hip_atomic_store((gu32 *)p.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                chunk = (int32_t)min((uint32_t)gridDim.x + t, (uint32_t)p.n_chunks);
            }
            lmisc[0] = chunk;
            lmisc[1] = HBM ? 0 : (int32_t)__hip_atomic_load((gu32 *)p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        const int chunk = lmisc[0];
        dead = dead || lmisc[1] != 0;
        if (chunk >= (HBM && WP ? (p.n_chunks + NWV - 1) / NWV : p.n_chunks)) break;
        if (dead) continue;
        const int t0 = chunk * p.chunk_tiles;
        const int nt = min(p.chunk_tiles, p.n_tiles - t0);           // tiles of the chunk
        // tiles of this region: WP -- tiles [wave * TR, wave * TR + TR) of the chunk; shared -- all of them
        const int rt0 = WP ? min(wave * TR, nt) : 0;
        // HBM: the region is K1's chunk kc
        const int kc = WP ? chunk * NWV + wave : chunk;
        const bool kvalid = kc < p.n_chunks && (!HBM || p.chunk_dirty == nullptr || p.chunk_dirty[min(kc, p.n_chunks - 1)] != 0u);
        const int rg0 = HBM ? (kvalid ? kc * p.chunk_tiles : 0) : t0 + rt0;     // first tile of the region
        const int rnt = HBM ? (kvalid ? min(p.chunk_tiles, p.n_tiles - rg0) : 0) : (WP ? min(TR, nt - rt0) : nt);

        // ------------------------------------------------------------------ phase A: tiles -> counts and ballots in LDS
#ifdef EVOFST_EXPERIMENTS
        if (!(p.debug & 8))
#endif
        if (!HBM) {
            if (WP) count_wave_tiles<CT>(p, lane, rnt, t0 + rt0, 1, 0, 1, lcnt, lzf, lirr, lskip, lmiss, lnan, lsegs);
            else count_wave_tiles<CT>(p, lane, wave < nt ? (nt - wave + NWV - 1) / NWV : 0, t0 + wave, NWV, wave, NWV, lcnt, lzf, lirr, lskip, lmiss, lnan, lsegs);
        }
        for (int i = dtid; i < P; i += dn) { carH[i] = 0.0; carT[i] = 0.0; carC[i] = 0u; }
        region_sync();

        // ------------------------------------------------------------------ census, publish, look back
        uint32_t cen = 0;                      // included sites of the region's tiles
#ifdef EVOFST_EXPERIMENTS
        if (p.debug & 64) continue;
#endif
        int64_t g0 = b0;
        if (HBM) {
            if (active && kvalid) g0 = p.chunk_offsets[(size_t)q * p.n_chunks + kc];
        } else {
        if (active)
            for (int ti = 0; ti < rnt; ++ti) {
                const uint64_t *zp = lzf + (size_t)ti * 2 * P;
                const uint64_t ex = (zp[pi] & zp[pj]) | (zp[P + pi] & zp[P + pj]);     // fst.cpp:543-544
                cen += 64u - (uint32_t)__popcll(ex);
            }
        uint32_t cen_chunk = cen, cen_before = 0;      // of the whole chunk; of the chunk's regions before this one
        if (WP) {
            wcen[wave * 64 + lane] = cen;
            __syncthreads();
            cen_chunk = 0;
            for (int w = 0; w < NWV; ++w) {
                const uint32_t v = wcen[w * 64 + lane];
                cen_chunk += v;
                cen_before += w < wave ? v : 0u;
            }
        }
        const bool publisher = active && (!WP || wave == 0);
        if (publisher) store_granule(p.agg0 + (size_t)chunk * p.n_pairs + q, p.epoch, cen_chunk);
        uint32_t s[3] = {0u, 0u, 0u};
        bool ok = true;
#ifdef EVOFST_EXPERIMENTS
        if (p.debug & 1) s[0] = (uint32_t)chunk * 100u / (WP ? NWV : 1); else
#endif
        if (active) ok = sum_granule_rows(p, chunk, q, WP ? wave : 0, WP ? NWV : 1, s);
        if (WP) {
            lb[(wave * 3 + 0) * 64 + lane] = s[0]; lb[(wave * 3 + 1) * 64 + lane] = s[1]; lb[(wave * 3 + 2) * 64 + lane] = s[2];
        }
        if (__syncthreads_or(!ok)) { dead = true; continue; }
        if (WP) {
            s[0] = s[1] = s[2] = 0u;
            for (int w = 0; w < NWV; ++w) { s[0] += lb[(w * 3 + 0) * 64 + lane]; s[1] += lb[(w * 3 + 1) * 64 + lane]; s[2] += lb[(w * 3 + 2) * 64 + lane]; }
        }
        if (publisher) {
            const int grp0 = chunk / kFan, grp1 = grp0 / kFan;
            if (chunk % kFan == kFan - 1) {
                store_granule(p.agg1 + (size_t)grp0 * p.n_pairs + q, p.epoch, s[0] + cen_chunk);
                if (grp0 % kFan == kFan - 1) store_granule(p.agg2 + (size_t)grp1 * p.n_pairs + q, p.epoch, s[1] + s[0] + cen_chunk);
            }
            if (chunk == p.n_chunks - 1) p.incl_totals[q] = (int64_t)s[2] + (int64_t)s[1] + (int64_t)s[0] + (int64_t)cen_chunk;
            if (chunk == 0) p.first_segment[q] = first_seg;
        }
        g0 = b0 + (int64_t)s[2] + (int64_t)s[1] + (int64_t)s[0] + (int64_t)cen_before;
        }
#ifdef EVOFST_EXPERIMENTS
        if (p.debug & 2) continue;
#endif

        // ------------------------------------------------------------------ phase B: K2's walk over the region's tiles
        int64_t m = g0 / p.g;
        int32_t left = p.g - (int32_t)(g0 - m * p.g);
        const int32_t left0 = left;
        const int32_t row0 = (int32_t)(m - first_seg);
        size_t orow = (size_t)row0 * p.n_pairs + q;
        double D = 0.0;                       // sum d^2 over the sites of the current fill (masked by the site filter in general fills)
        uint64_t cc = 0;                      // sum c_i c_j
        uint32_t exi = 0, exj = 0;            // sum of c_i / c_j over excluded sites (general fills only)
        uint32_t bn = 0, bd = 0;              // included sites whose numerator / denominator is NaN in the reference
        // Numerator / denominator sums since the last boundary, complete fills only.  Every floating-point sum here starts at a
        // fill (D, the per-population tables PH / PT) or at a boundary (segN, segD), never at the chunk: a segment's error
        // scales with the segment's own sums, whatever the chunk length (256-tile chunks and 20-site windows included).
        double segN = 0.0, segD = 0.0;
        uint32_t prevCi = 0, prevCj = 0, prevBn = 0, prevBd = 0;
        uint64_t prevCC = 0;
        uint64_t kx = 0, prevKx = 0;          // sumKij terms of excluded sites the regular loop's sums include (fixed in both, alleles missing)
        int32_t nflush = 0;
        uint64_t aCC = 0, aCi = 0, aCj = 0, aQi = 0, aQj = 0;   // clean fills since the last boundary: sums of c_i c_j, c, c^2
        const int nvec = P * VPR;

        // integer sums of clean fills -> their share of the numerator / denominator sums (added to numI / denI) and of sumKij
        auto int_parts = [&](uint64_t iCC, uint64_t iCi, uint64_t iCj, uint64_t iQi, uint64_t iQj, double &numI, double &denI, uint64_t &kI) {
            const uint64_t E = nj2 * iQi + ni2 * iQj - 2ull * ninj * iCC;       // sum (c_i n_j - c_j n_i)^2, exact
            const double dTi = (double)((uint64_t)ni * iCi - iQi), dTj = (double)((uint64_t)nj * iCj - iQj);   // sum c (n - c)
            const double dE = (double)E * rE;
            numI = dE - dTi * rHi - dTj * rHj;
            denI = dE + dTi * rTi + dTj * rTj;
            kI += (uint64_t)nj * iCi + (uint64_t)ni * iCj - 2ull * iCC;
        };

        // One vector = EPT consecutive sites of population k.
        auto derive_vec = [&](int vi, const uint4 &vv, uint32_t skipfill) {
            const int k = vi / VPR, r = vi - k * VPR;
            const int o = k * SUBP + r * EPT;
            const uint32_t skipbits = skipfill >> (r * EPT);
            uint32_t w[4] = {vv.x, vv.y, vv.z, vv.w};
            if (skipfill != 0u) {                 // uniform and rare: a skipped site contributes 0 everywhere (c = 0)
#pragma unroll
                for (int j = 0; j < EPT; ++j)
                    if ((skipbits >> j) & 1u) w[(j * EB) / 32] &= ~(CountTraits<CT>::mask << ((j * EB) % 32));
            }
            double tv[EPT], hv[EPT];
            uint32_t cv[EPT];
            double rh = 0.0, rt = 0.0;
            uint32_t rc = 0;
            RcpEntry ent[use_rcp ? EPT : 1];
            constexpr int HB = EVOFST_K2_ENT_BATCH;
            auto read_ent = [&](int j0) {
#pragma unroll
                for (int j = j0; j < j0 + HB && j < EPT; ++j) {
                    const uint32_t v = (w[(j * EB) / 32] >> ((j * EB) % 32)) & (uint32_t)((1ull << EB) - 1ull);
                    const uint4 *ep = reinterpret_cast<const uint4 *>(rcp + (v >> CountTraits<CT>::shift));
                    const uint4 lo = ep[0], hi = ep[1];
                    ent[j].r0 = __hiloint2double((int)lo.y, (int)lo.x); ent[j].rr0 = __hiloint2double((int)lo.w, (int)lo.z);
                    ent[j].r1 = __hiloint2double((int)hi.y, (int)hi.x); ent[j].flags = hi.z;
                }
            };
#pragma unroll
            for (int j = 0; j < EPT; ++j) {
                if (use_rcp && j % HB == 0) read_ent(j);
                const uint32_t v = (w[(j * EB) / 32] >> ((j * EB) % 32)) & (uint32_t)((1ull << EB) - 1ull);
                const uint32_t c = v & CountTraits<CT>::mask;
                const uint32_t nm = v >> CountTraits<CT>::shift;
                const double dc = (double)c;
                double pp, tt, hh;
                uint32_t x = c;
                if (use_rcp) {
                    const RcpEntry e = ent[j];
                    pp = dc * e.r0;                                       // c / nm to 1 ulp (0 when nm == 0)
                    tt = (double)(c * (nm - c)) * e.rr0;                  // p(1-p), exactly 0 when c is 0 or nm
                    hh = tt * e.r1;                                       // t / (nm - 1)     (0 when nm < 2)
                    x |= e.flags;
                } else {
                    const double dn_ = (double)nm;
                    pp = nm ? dc / dn_ : 0.0;                             // utils.h:272-275
                    tt = pp * (1.0 - pp);
                    hh = nm > 1 ? tt / (dn_ - 1.0) : 0.0;                 // fst.h:21-22
                    if (nm < 2) x |= kXBadNum;
                    if (nm == 0) x |= kXBadDen;
                }
                pt[o + j] = pp; xt[o + j] = x;
                tv[j] = tt; hv[j] = hh; cv[j] = c;
                rh += hh; rt += tt; rc += c;
            }
            // exclusive scan of the vector totals over the VPR lanes of the row (consecutive lanes of one wave)
            double eh = rh, et = rt;
            uint32_t ec = rc;
#pragma unroll
            for (int d = 1; d < VPR; d <<= 1) {
                const double yh = __shfl_up(eh, d, VPR), yt = __shfl_up(et, d, VPR);
                const uint32_t yc = __shfl_up(ec, d, VPR);
                if (r >= d) { eh += yh; et += yt; ec += yc; }
            }
            const double ch = 0.0, ct = 0.0;                          // the h / t tables start at the fill (see segN below)
            if (sizeof(CT) == 2) {                                   // (EPT = 8 counts of at most 255)
                *reinterpret_cast<uint2 *>(cbt + k * SUBB + r * EPT) =
                    make_uint2(cv[0] | (cv[1] << 8) | (cv[2] << 16) | (cv[3] << 24), cv[4 % EPT] | (cv[5 % EPT] << 8) | (cv[6 % EPT] << 16) | (cv[7 % EPT] << 24));
            }
            const uint32_t ccar = carC[k];
            if (r == VPR - 1) carC[k] = ccar + ec;
            double ah = ch + (eh - rh), at = ct + (et - rt);       // sums before this vector
            uint32_t ac = ccar + (ec - rc);
#pragma unroll
            for (int j = 0; j < EPT; ++j) {
                ah += hv[j]; at += tv[j]; ac += cv[j];
                PH[o + j] = ah; PT[o + j] = at; PC[o + j] = ac;
            }
        };

        // HBM: the region's counts and ballots come in pieces of TR tiles, copied from HBM straight into one of two LDS
        // buffers (global_load_lds, 16 bytes per lane, no registers in between) while the piece in the other one is walked
        typedef __attribute__((address_space(3))) void *LdsPtr;
        const CT *lcntp = lcnt;
        const uint64_t *lzfp = lzf, *lirrp = lirr, *lskipp = lskip, *lmissp = lmiss, *lnanp = lnan;
        uint64_t prem[4] = {0, 0, 0, 0};
        auto piece_tiles_of = [&](int piece) { return min(TR, rnt - piece * TR); };
        auto request_piece = [&](int piece) {
            const int g = rg0 + piece * TR, pnt = piece_tiles_of(piece), buf = piece & 1;
            const int v0 = WP ? 0 : wave * 64;           // first vector of this wave's row of 64; uniform
            const int nvp = pnt * P * kTileSites * (int)sizeof(CT) / 16;
            const uint4 *src = reinterpret_cast<const uint4 *>(reinterpret_cast<const CT *>(p.counts) + (size_t)g * P * kTileSites);
            uint4 *dst = reinterpret_cast<uint4 *>(lcnt + (size_t)buf * TR * P * kTileSites);
            for (int v = v0; v < nvp; v += dn)
                if (v + lane < nvp) __builtin_amdgcn_global_load_lds(src + v + lane, (LdsPtr)(dst + v), 16, 0, 0);
            const int nz = pnt * P;                      // 16-byte vectors of the ballots: [tile][z|f][P] u64
            const uint4 *zsrc = reinterpret_cast<const uint4 *>(p.zf_masks + (size_t)g * 2 * P);
            uint4 *zdst = reinterpret_cast<uint4 *>(lzf + (size_t)buf * TR * 2 * P);
            for (int v = v0; v < nz; v += dn)
                if (v + lane < nz) __builtin_amdgcn_global_load_lds(zsrc + v + lane, (LdsPtr)(zdst + v), 16, 0, 0);
            const int tt = g + min(dtid, pnt - 1);
            prem[0] = p.irr_mask[tt]; prem[1] = p.skip_mask[tt]; prem[2] = p.miss_mask[tt]; prem[3] = p.nan_mask[tt];
        };
        const int n_pieces = HBM ? (rnt + TR - 1) / TR : 1;
        if (HBM && rnt > 0) request_piece(0);
        for (int piece = 0; piece < n_pieces; ++piece) {
        const int pnt = HBM ? piece_tiles_of(piece) : rnt;     // tiles in LDS
        const int gtile0 = HBM ? rg0 + piece * TR : rg0;       // and the first of them
        if (HBM) {
            const int buf = piece & 1;
            lcntp = lcnt + (size_t)buf * TR * P * kTileSites; lzfp = lzf + (size_t)buf * TR * 2 * P;
            lirrp = lirr + buf * TR; lskipp = lskip + buf * TR; lmissp = lmiss + buf * TR; lnanp = lnan + buf * TR;
            for (int v = dtid; v < pnt; v += dn) {
                if (v >= dn) { prem[0] = p.irr_mask[gtile0 + v]; prem[1] = p.skip_mask[gtile0 + v]; prem[2] = p.miss_mask[gtile0 + v]; prem[3] = p.nan_mask[gtile0 + v]; }
                lirr[buf * TR + v] = prem[0]; lskip[buf * TR + v] = prem[1]; lmiss[buf * TR + v] = prem[2]; lnan[buf * TR + v] = prem[3];
            }
            __builtin_amdgcn_s_waitcnt(0x0f70);                // vmcnt(0): this thread's copies of the piece have landed
            region_sync();                                     // ... and everybody else's
            if (piece + 1 < n_pieces) request_piece(piece + 1);   // into the buffer whose walk ended at the last region_sync
        }
        for (int st = 0; st < SPT * pnt; ++st) {
            const int ti = st / SPT, sh = (st - ti * SPT) * SUB;
            const uint32_t skipfill = (uint32_t)(lskipp[ti] >> sh) & FULL;
            // general: a site of the fill is NaN in the reference (a population with fewer than two called alleles).  Sites
            // both populations are fixed at with alleles missing -- excluded, but with a sumKij term -- are taken out of the
            // regular loop's sums afterwards (kx below)
            const bool general = ((uint32_t)(lnanp[ti] >> sh) & FULL) != 0u;   // uniform
            const bool fill_missing = ((uint32_t)(lmissp[ti] >> sh) & FULL) != 0u;   // uniform
            const bool clean = kInt && ((uint32_t)((lmissp[ti] | lskipp[ti] | lirrp[ti]) >> sh) & FULL) == 0u;   // uniform
            if (!clean) {
                const uint4 *vp = reinterpret_cast<const uint4 *>(lcntp + (size_t)ti * P * kTileSites);
                for (int vi = dtid; vi < nvec; vi += dn) {
                    const int k = vi / VPR, r = vi - k * VPR;
                    derive_vec(vi, vp[k * (kTileSites / EPT) + sh / EPT + r], skipfill);
                }
            }
            region_sync();
#ifdef EVOFST_EXPERIMENTS
            if (!(p.debug & 256))
#endif
            if (active) {
                const uint64_t *zp = lzfp + (size_t)ti * 2 * P;
                const uint32_t incl = ~(uint32_t)(((zp[pi] & zp[pj]) | (zp[P + pi] & zp[P + pj])) >> sh) & FULL;   // fst.cpp:543-544
                int b = SUB;                                   // site of this lane's next boundary in the fill (SUB: none)
                {
                    const int cnt = __popc(incl);
                    if (left <= cnt) b = nth_set_bit(incl, left);
                    else left -= cnt;
                }
                // emit: the accumulators right after boundary site b -> one row of the segment table, then the next boundary.
                // H, T, PCi, PCj: the per-population prefix sums at b; i*: the integer sums of the segment's clean fills.
                auto emit = [&](double cD, uint64_t ccc, uint64_t ckx, uint32_t cexi, uint32_t cexj, uint32_t cbn, uint32_t cbd, double H, double T,
                                uint32_t PCi, uint32_t PCj, uint64_t iCC, uint64_t iCi, uint64_t iCj, uint64_t iQi, uint64_t iQj) {
                    const uint32_t Ci = PCi - cexi, Cj = PCj - cexj;
                    const double cumN = cD - H, cumD = cD + T;
                    // sumKij (fst.cpp:299-321) of the segment, exact: differences are taken modulo 2^32 / 2^64
                    uint64_t kk = (uint64_t)nj * (uint64_t)(Ci - prevCi) + (uint64_t)ni * (uint64_t)(Cj - prevCj) - 2ull * (ccc - prevCC) - (ckx - prevKx);
                    double numI = 0.0, denI = 0.0;
                    // (nothing to add where the segment has no clean fill -- data with missing alleles in every fill -- and the
                    //  64-bit products and conversions are most of this function's instructions)
                    if (kInt && (iCC | iCi | iCj | iQi | iQj) != 0ull) int_parts(iCC, iCi, iCj, iQi, iQj, numI, denI, kk);
                    SegRow row;
                    row.num = cbn != prevBn ? kNaN : (segN + cumN) + numI;      // cumN, cumD: this fill's sums up to b
                    row.den = cbd != prevBd ? kNaN : (segD + cumD) + denI;
                    row.k = (double)kk;
                    row.end = (gtile0 + ti) * kTileSites + sh + b; row.pad = 0;
                    {   // two 16-byte stores into one 32-byte sector
                        uint4 *dst = reinterpret_cast<uint4 *>(p.seg_rows + orow);
                        const uint4 *src = reinterpret_cast<const uint4 *>(&row);
                        dst[0] = src[0]; dst[1] = src[1];
                    }
                    segN = -cumN; segD = -cumD;                     // the rest of the fill comes in with the fill's totals
                    prevCi = Ci; prevCj = Cj; prevCC = ccc; prevKx = ckx; prevBn = cbn; prevBd = cbd;
                    ++nflush;
                    orow += p.n_pairs;
                    const uint32_t rem = b >= SUB - 1 ? 0u : incl & ~((2u << b) - 1u);
                    const int c2 = __popc(rem);
                    if (p.g <= c2) b = nth_set_bit(rem, p.g);
                    else { left = p.g - c2; b = SUB; }
                };
                if (clean) {
                    // ---- no missing allele and no skipped site in this fill: p = c / n with constant n, so the sums of d^2, t
                    // and h over a range of sites follow exactly from the integer sums of c_i c_j, c and c^2 (4 sites per
                    // v_dot4_u32_u8); no tables are derived
                    constexpr int NW4 = SUB / 4;
                    constexpr uint32_t kOnes = 0x01010101u;
                    uint32_t wi[NW4], wj[NW4];
                    {
                        const uint4 *ri = reinterpret_cast<const uint4 *>(lcntp + ((size_t)ti * P + pi) * kTileSites + sh);
                        const uint4 *rj = reinterpret_cast<const uint4 *>(lcntp + ((size_t)ti * P + pj) * kTileSites + sh);
#pragma unroll
                        for (int v = 0; v < SUB / 8; ++v) {       // 8 packed {c, nm} pairs -> two words of 4 c bytes
                            const uint4 x = ri[v], y = rj[v];
                            wi[2 * v] = __builtin_amdgcn_perm(x.y, x.x, 0x06040200u); wi[2 * v + 1] = __builtin_amdgcn_perm(x.w, x.z, 0x06040200u);
                            wj[2 * v] = __builtin_amdgcn_perm(y.y, y.x, 0x06040200u); wj[2 * v + 1] = __builtin_amdgcn_perm(y.w, y.z, 0x06040200u);
                        }
                    }
                    uint32_t tCC = 0, tCi = 0, tCj = 0, tQi = 0, tQj = 0;     // over the whole fill (< 2^22)
#pragma unroll
                    for (int m_ = 0; m_ < NW4; ++m_) {
                        tCC = __builtin_amdgcn_udot4(wi[m_], wj[m_], tCC, false);
                        tCi = __builtin_amdgcn_udot4(wi[m_], kOnes, tCi, false); tCj = __builtin_amdgcn_udot4(wj[m_], kOnes, tCj, false);
                        tQi = __builtin_amdgcn_udot4(wi[m_], wi[m_], tQi, false); tQj = __builtin_amdgcn_udot4(wj[m_], wj[m_], tQj, false);
                    }
                    uint32_t dCC = 0, dCi = 0, dCj = 0, dQi = 0, dQj = 0;     // of them, up to the last boundary
#ifdef EVOFST_EXPERIMENTS
                    if (p.debug & 128) b = SUB;
#endif
                    while (b < SUB) {
                        uint32_t pCC = 0, pCi = 0, pCj = 0, pQi = 0, pQj = 0;  // sites 0 .. b
#pragma unroll
                        for (int m_ = 0; m_ < NW4; ++m_) {
                            const int nbv = b + 1 - 4 * m_;
                            const uint32_t mk = nbv >= 4 ? 0xffffffffu : (nbv <= 0 ? 0u : ((1u << (8 * nbv)) - 1u));
                            const uint32_t x = wi[m_] & mk, y = wj[m_] & mk;
                            pCC = __builtin_amdgcn_udot4(x, y, pCC, false);
                            pCi = __builtin_amdgcn_udot4(x, kOnes, pCi, false); pCj = __builtin_amdgcn_udot4(y, kOnes, pCj, false);
                            pQi = __builtin_amdgcn_udot4(x, x, pQi, false); pQj = __builtin_amdgcn_udot4(y, y, pQj, false);
                        }
                        emit(0.0, cc, kx, exi, exj, bn, bd, 0.0, 0.0, carC[pi], carC[pj],
                             aCC + (pCC - dCC), aCi + (pCi - dCi), aCj + (pCj - dCj), aQi + (pQi - dQi), aQj + (pQj - dQj));
                        aCC = aCi = aCj = aQi = aQj = 0;
                        dCC = pCC; dCi = pCi; dCj = pCj; dQi = pQi; dQj = pQj;
                    }
                    aCC += tCC - dCC; aCi += tCi - dCi; aCj += tCj - dCj; aQi += tQi - dQi; aQj += tQj - dQj;
                } else {
                // sites of the fill both populations are fixed at (c == nm), where the fill has missing alleles at all
                const uint32_t E = (!general && fill_missing) ? ((uint32_t)((zp[P + pi] & zp[P + pj]) >> sh) & FULL) : 0u;
                const uint64_t kx0 = kx;
                auto kx_sum = [&](uint32_t bits) {             // their sumKij terms (0 where no allele is missing: c = n)
                    uint64_t sum = 0;
                    while (bits) {
                        const int e_ = __ffs((int)bits) - 1;
                        bits &= bits - 1u;
                        const uint64_t ci = xt[io + e_] & kXMask, cj = xt[jo + e_] & kXMask;
                        sum += (uint64_t)nj * ci + (uint64_t)ni * cj - 2ull * ci * cj;
                    }
                    return sum;
                };
                const int bgrp = b & ~(kGrp - 1);
                D = 0.0;
                double sD = D;                                 // snapshots at the end of the boundary's group
                uint64_t scc = cc;
                uint32_t sexi = exi, sexj = exj, sbn = bn, sbd = bd;   // (the regular loop leaves these four alone)
                if (!general) {
                    constexpr bool kBytes = sizeof(CT) == 2 && kGrp == 4;
#pragma unroll
                    for (int sb = 0; sb < SUB; sb += kGrp) {
                        double a[kGrp], bb[kGrp];
                        uint32_t xi[kGrp], xj[kGrp];
                        uint32_t ci4 = 0, cj4 = 0;
                        if (kBytes) {
                            ci4 = *reinterpret_cast<const uint32_t *>(cbt + pi * SUBB + sb);
                            cj4 = *reinterpret_cast<const uint32_t *>(cbt + pj * SUBB + sb);
                        }
#pragma unroll
                        for (int u = 0; u < kGrp; ++u) {
                            a[u] = pt[io + sb + u]; bb[u] = pt[jo + sb + u];
                            if (!kBytes) { xi[u] = xt[io + sb + u]; xj[u] = xt[jo + sb + u]; }
                        }
                        uint32_t cch = kBytes ? __builtin_amdgcn_udot4(ci4, cj4, 0u, false) : 0u;   // c_i c_j of kGrp sites (< 2^30)
#pragma unroll
                        for (int u = 0; u < kGrp; ++u) {
                            const double d = a[u] - bb[u];
                            D = fma(d, d, D);
                            if (!kBytes) cch += __umul24(xi[u], xj[u]);     // low 24 bits: c (the flags sit above)
                        }
                        cc += cch;
                        if (sb == bgrp) { sD = D; scc = cc; }
                    }
                } else {
#pragma unroll 1
                    for (int sb = 0; sb < SUB; sb += kGrp) {
                        double a[kGrp], bb[kGrp];
                        uint32_t xi[kGrp], xj[kGrp];
#pragma unroll
                        for (int u = 0; u < kGrp; ++u) {
                            a[u] = pt[io + sb + u]; bb[u] = pt[jo + sb + u];
                            xi[u] = xt[io + sb + u]; xj[u] = xt[jo + sb + u];
                        }
                        uint32_t cch = 0;
#pragma unroll
                        for (int u = 0; u < kGrp; ++u) {
                            const bool inc = ((incl >> (sb + u)) & 1u) != 0u;
                            const double d = inc ? a[u] - bb[u] : 0.0;
                            D = fma(d, d, D);
                            const uint32_t ci = xi[u] & kXMask, cj = xj[u] & kXMask;
                            cch += inc ? ci * cj : 0u;
                            exi += inc ? 0u : ci; exj += inc ? 0u : cj;
                            const uint32_t bad = inc ? (xi[u] | xj[u]) >> 30 : 0u;
                            bn += bad & 1u; bd += bad >> 1;
                        }
                        cc += cch;
                        if (sb == bgrp) { sD = D; scc = cc; sexi = exi; sexj = exj; sbn = bn; sbd = bd; }
                    }
                }
                auto emit_general = [&](double cD, uint64_t ccc, uint32_t cexi, uint32_t cexj, uint32_t cbn, uint32_t cbd) {
                    const uint64_t ckx = E ? kx0 + kx_sum(E & (b >= 31 ? 0xffffffffu : ((2u << b) - 1u))) : kx0;   // up to the boundary
                    emit(cD, ccc, ckx, cexi, cexj, cbn, cbd, PH[io + b] + PH[jo + b], PT[io + b] + PT[jo + b], PC[io + b], PC[jo + b],
                         aCC, aCi, aCj, aQi, aQj);
                    aCC = aCi = aCj = aQi = aQj = 0;
                };
                if (b < SUB) {
                    // first boundary: the group snapshot minus the (at most kGrp - 1) sites of the group behind b
                    {
                        double cD = sD;
                        uint64_t ccc = scc;
                        uint32_t cexi = sexi, cexj = sexj, cbn = sbn, cbd = sbd;
                        double pa[kGrp - 1], pb[kGrp - 1];
                        uint32_t xa[kGrp - 1], xb[kGrp - 1];
                        const int room = (kGrp - 1) - (b & (kGrp - 1));   // sites of the group behind b
#pragma unroll
                        for (int u = 1; u < kGrp; ++u) {
                            const int s_ = u <= room ? b + u : b;
                            pa[u - 1] = pt[io + s_]; pb[u - 1] = pt[jo + s_]; xa[u - 1] = xt[io + s_]; xb[u - 1] = xt[jo + s_];
                        }
#pragma unroll
                        for (int u = 1; u < kGrp; ++u) {
                            const bool there = u <= room;
                            const bool inc = !general || ((incl >> ((b + u) & 31)) & 1u) != 0u;
                            const double d = there && inc ? pa[u - 1] - pb[u - 1] : 0.0;
                            cD = fma(-d, d, cD);
                            const uint32_t ci = there ? xa[u - 1] & kXMask : 0u, cj = there ? xb[u - 1] & kXMask : 0u;
                            ccc -= inc ? (uint64_t)(ci * cj) : 0ull;
                            if (general) {
                                cexi -= inc ? 0u : ci; cexj -= inc ? 0u : cj;
                                const uint32_t bad = there && inc ? (xa[u - 1] | xb[u - 1]) >> 30 : 0u;
                                cbn -= bad & 1u; cbd -= bad >> 1;
                            }
                        }
                        emit_general(cD, ccc, cexi, cexj, cbn, cbd);
                    }
                    // further boundaries of the same fill (windows shorter than it): from the end of the fill
                    while (b < SUB) {
                        double cD = D;
                        uint64_t ccc = cc;
                        uint32_t cexi = exi, cexj = exj, cbn = bn, cbd = bd;
                        for (int s_ = b + 1; s_ < SUB; ++s_) {
                            const uint32_t xi = xt[io + s_], xj = xt[jo + s_];
                            const double d = pt[io + s_] - pt[jo + s_];
                            const uint32_t ci = xi & kXMask, cj = xj & kXMask;
                            const bool inc = !general || ((incl >> s_) & 1u) != 0u;
                            if (inc) {
                                cD = fma(-d, d, cD);
                                ccc -= (uint64_t)(ci * cj);
                                if (general) { const uint32_t bad = (xi | xj) >> 30; cbn -= bad & 1u; cbd -= bad >> 1; }
                            } else {
                                cexi -= ci; cexj -= cj;
                            }
                        }
                        emit_general(cD, ccc, cexi, cexj, cbn, cbd);
                    }
                }
                if (E) kx = kx0 + kx_sum(E);                   // the whole fill
                // the fill's totals join the segment's sums (PH / PT at the last site = the populations' totals over the fill)
                segN += D - (PH[io + SUB - 1] + PH[jo + SUB - 1]);
                segD += D + (PT[io + SUB - 1] + PT[jo + SUB - 1]);
                }   // general fill
            }
            region_sync();
        }
        }   // pieces
        if (active && (!HBM || kvalid)) {
            // what the (sub-)chunk holds behind its last boundary
            const int n_sub = HBM ? p.n_chunks : (WP ? p.n_chunks * NWV : p.n_chunks);
            const int sub = HBM ? kc : (WP ? chunk * NWV + wave : chunk);
            const size_t cq = (size_t)sub * p.n_pairs + q;
            const size_t plane = (size_t)p.n_pairs * n_sub;
            const uint32_t Ci = carC[pi] - exi, Cj = carC[pj] - exj;
            uint64_t kk = (uint64_t)nj * (uint64_t)(Ci - prevCi) + (uint64_t)ni * (uint64_t)(Cj - prevCj) - 2ull * (cc - prevCC) - (kx - prevKx);
            double numI = 0.0, denI = 0.0;
            if (kInt) int_parts(aCC, aCi, aCj, aQi, aQj, numI, denI, kk);
            p.tail[cq] = bn != prevBn ? kNaN : segN + numI;
            p.tail[plane + cq] = bd != prevBd ? kNaN : segD + denI;
            p.tail[2 * plane + cq] = (double)kk;
            p.tail_n[cq] = make_int4((nflush ? p.g : left0) - left, nflush, row0, 0);   // {sites summed in the tail, segments completed, row of the first}
        }
    }
}

// =====================================================================================
// clean_walk_kernel: the statistics walk over chunks in which no site is skipped and no population has a missing
// allele (K1 marks the others in chunk_dirty; the walk above takes those).  There p = c / n with the population's full
// n, so the window sums of d^2, p(1-p) and sumKij follow exactly from integer sums of c_i c_j, c and c^2 (8-bit counts,
// four sites per v_dot4_u32_u8); none of the per-population tables of the general walk exist here, which leaves room for
// four waves per SIMD.  Same regions, pieces, rows and tails as the walk above (its HBM mode), same arithmetic as its
// clean fills: the two kernels are interchangeable chunk by chunk.
// =====================================================================================
constexpr int kCleanWps = 4;

static size_t clean_region_bytes(int P, int piece_tiles, bool compact) {
    return a16((size_t)2 * piece_tiles * P * kTileSites * (compact ? 1 : 2)) + a16((size_t)2 * piece_tiles * 2 * P * 8);
}
size_t clean_walk_lds_bytes(int P, int n_pairs, int piece_tiles, bool compact) {
    return 32 * 32 + (n_pairs <= 64 ? kFuseWaves : 1) * clean_region_bytes(P, piece_tiles, compact);
}

// CB = bytes per count in the chunks this kernel takes: 2 = K1's packed {c, nm}, 1 = c alone, the chunk's tiles back to
// back from the chunk's first byte (K1Params::compact)
template <bool WP, int CB>
__global__ __launch_bounds__(kK2Threads, kCleanWps) void clean_walk_kernel(FusedParams p) {
    typedef unsigned char CT;                          // LDS and HBM offsets below are in bytes
    constexpr int SUB = 32, SPT = kTileSites / SUB, NW4 = SUB / 4;
    extern __shared__ __align__(16) unsigned char smem[];
    const int P = p.P;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NWV = blockDim.x >> 6;
    const int TR = p.piece_tiles;
    // prefix masks: row b holds, for each of the 8 words of a fill, the bytes of sites 0 .. b
    uint32_t *pmask = reinterpret_cast<uint32_t *>(smem);
    for (int i = tid; i < 32 * NW4; i += blockDim.x) {
        const int b = i / NW4, m = i - b * NW4, nbv = b + 1 - 4 * m;
        pmask[i] = nbv >= 4 ? 0xffffffffu : (nbv <= 0 ? 0u : ((1u << (8 * nbv)) - 1u));
    }
    __syncthreads();
    const size_t cnt_bytes = a16((size_t)2 * TR * P * kTileSites * CB);
    unsigned char *region = smem + 32 * 32 + (WP ? (size_t)wave * (cnt_bytes + a16((size_t)2 * TR * 2 * P * 8)) : 0);
    CT *lcnt = reinterpret_cast<CT *>(region);                            // [2][TR][P][64]
    uint64_t *lzf = reinterpret_cast<uint64_t *>(region + cnt_bytes);     // [2][TR][z|f][P]
    const int dtid = WP ? lane : tid, dn = WP ? 64 : (int)blockDim.x;
    const int kc = WP ? (int)blockIdx.x * NWV + wave : (int)blockIdx.x;   // the region: K1's chunk kc
    if (kc >= p.n_chunks || p.chunk_dirty[kc] != 0u) return;              // uniform over the threads that share the region
    const int q = dtid;
    const bool active = q < p.n_pairs;
    uint32_t pi = 0, pj = 0, ni = 1, nj = 1;
    int64_t b0 = 0, g0 = 0;
    if (active) {
        const uint32_t ij = p.pairs[q];
        pi = ij & 0xffffu; pj = ij >> 16;
        ni = (uint32_t)p.n_full[pi]; nj = (uint32_t)p.n_full[pj];
        b0 = p.base[q];
        g0 = p.chunk_offsets[(size_t)q * p.n_chunks + kc];
    }
    const int64_t first_seg = b0 / p.g;
    const double dni = (double)ni, dnj = (double)nj;
    const double ni2 = dni * dni, nj2 = dnj * dnj, ninj = dni * dnj;      // exact
    const double rE = 1.0 / (ninj * ninj);                                // d^2 = (c_i n_j - c_j n_i)^2 / (n_i n_j)^2
    const double rTi = 1.0 / ni2, rTj = 1.0 / nj2;                        // t = c (n - c) / n^2
    const double rHi = ni > 1 ? rTi / (double)(ni - 1) : 0.0, rHj = nj > 1 ? rTj / (double)(nj - 1) : 0.0;   // h = t / (n - 1)
    auto region_sync = [&]() {
        if (WP) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
        else __syncthreads();
    };
    const int rg0 = kc * p.chunk_tiles;
    const int rnt = min(p.chunk_tiles, p.n_tiles - rg0);

    int64_t m = g0 / p.g;
    int32_t left = p.g - (int32_t)(g0 - m * p.g);
    const int32_t left0 = left;
    const int32_t row0 = (int32_t)(m - first_seg);
    size_t orow = (size_t)row0 * p.n_pairs + q;
    int32_t nflush = 0;
    // integer sums since the last boundary: c_i c_j, c_i, c_j, c_i^2, c_j^2.  A chunk has at most 2^14 sites and
    // c <= 255, so every sum stays below 2^31 and every product below 2^53: the doubles below are exact integers.
    uint32_t aCC = 0, aCi = 0, aCj = 0, aQi = 0, aQj = 0;
    auto parts = [&](uint32_t iCC, uint32_t iCi, uint32_t iCj, uint32_t iQi, uint32_t iQj, double &num, double &den, double &k) {
        const double CC = (double)iCC, Ci = (double)iCi, Cj = (double)iCj, Qi = (double)iQi, Qj = (double)iQj;
        const double E = fma(nj2, Qi, fma(ni2, Qj, -2.0 * ninj * CC));    // sum (c_i n_j - c_j n_i)^2
        const double dTi = fma(dni, Ci, -Qi), dTj = fma(dnj, Cj, -Qj);    // sum c (n - c)
        const double dE = E * rE;
        num = dE - dTi * rHi - dTj * rHj;
        den = dE + dTi * rTi + dTj * rTj;
        k = fma(dnj, Ci, fma(dni, Cj, -2.0 * CC));                        // sumKij (fst.cpp:299-321)
    };

    typedef __attribute__((address_space(3))) void *LdsPtr;
    auto piece_tiles_of = [&](int piece) { return min(TR, rnt - piece * TR); };
    auto request_piece = [&](int piece) {
        const int g = rg0 + piece * TR, pnt = piece_tiles_of(piece), buf = piece & 1;
        const int v0 = WP ? 0 : wave * 64;
        const int nvp = pnt * P * kTileSites * CB / 16;
        // one-byte tiles lie back to back from the chunk's first byte, packed ones at their tile's own place
        const uint4 *src = reinterpret_cast<const uint4 *>(reinterpret_cast<const CT *>(p.counts) +
                                                           (CB == 1 ? (size_t)rg0 * P * kTileSites * 2 + (size_t)piece * TR * P * kTileSites : (size_t)g * P * kTileSites * 2));
        uint4 *dst = reinterpret_cast<uint4 *>(lcnt + (size_t)buf * TR * P * kTileSites * CB);
        for (int v = v0; v < nvp; v += dn)
            if (v + lane < nvp) __builtin_amdgcn_global_load_lds(src + v + lane, (LdsPtr)(dst + v), 16, 0, 0);
        const int nz = pnt * P;
        const uint4 *zsrc = reinterpret_cast<const uint4 *>(p.zf_masks + (size_t)g * 2 * P);
        uint4 *zdst = reinterpret_cast<uint4 *>(lzf + (size_t)buf * TR * 2 * P);
        for (int v = v0; v < nz; v += dn)
            if (v + lane < nz) __builtin_amdgcn_global_load_lds(zsrc + v + lane, (LdsPtr)(zdst + v), 16, 0, 0);
    };
    const int n_pieces = (rnt + TR - 1) / TR;
    if (rnt > 0) request_piece(0);
    for (int piece = 0; piece < n_pieces; ++piece) {
        const int pnt = piece_tiles_of(piece), gtile0 = rg0 + piece * TR, buf = piece & 1;
        const CT *lcntp = lcnt + (size_t)buf * TR * P * kTileSites * CB;
        const uint64_t *lzfp = lzf + (size_t)buf * TR * 2 * P;
        __builtin_amdgcn_s_waitcnt(0x0f70);                    // vmcnt(0): this thread's copies of the piece have landed
        region_sync();
        if (piece + 1 < n_pieces) request_piece(piece + 1);
        if (active)
        for (int st = 0; st < SPT * pnt; ++st) {
            const int ti = st / SPT, sh = (st - ti * SPT) * SUB;
            const uint64_t *zp = lzfp + (size_t)ti * 2 * P;
            const uint32_t incl = ~(uint32_t)(((zp[pi] & zp[pj]) | (zp[P + pi] & zp[P + pj])) >> sh);   // fst.cpp:543-544
            int b = SUB;                                       // site of this lane's next boundary in the fill (SUB: none)
            {
                const int cnt = __popc(incl);
                if (left <= cnt) b = nth_set_bit(incl, left);
                else left -= cnt;
            }
            constexpr uint32_t kOnes = 0x01010101u;
            uint32_t wi[NW4], wj[NW4];
            if (CB == 1) {
                const uint4 *ri = reinterpret_cast<const uint4 *>(lcntp + ((size_t)ti * P + pi) * kTileSites + sh);
                const uint4 *rj = reinterpret_cast<const uint4 *>(lcntp + ((size_t)ti * P + pj) * kTileSites + sh);
#pragma unroll
                for (int v = 0; v < SUB / 16; ++v) {           // 16 counts a vector, as the dot products want them
                    const uint4 x = ri[v], y = rj[v];
                    wi[4 * v] = x.x; wi[4 * v + 1] = x.y; wi[4 * v + 2] = x.z; wi[4 * v + 3] = x.w;
                    wj[4 * v] = y.x; wj[4 * v + 1] = y.y; wj[4 * v + 2] = y.z; wj[4 * v + 3] = y.w;
                }
            } else {
                const uint4 *ri = reinterpret_cast<const uint4 *>(lcntp + (((size_t)ti * P + pi) * kTileSites + sh) * 2);
                const uint4 *rj = reinterpret_cast<const uint4 *>(lcntp + (((size_t)ti * P + pj) * kTileSites + sh) * 2);
#pragma unroll
                for (int v = 0; v < SUB / 8; ++v) {            // 8 packed {c, nm} pairs -> two words of 4 c bytes
                    const uint4 x = ri[v], y = rj[v];
                    wi[2 * v] = __builtin_amdgcn_perm(x.y, x.x, 0x06040200u); wi[2 * v + 1] = __builtin_amdgcn_perm(x.w, x.z, 0x06040200u);
                    wj[2 * v] = __builtin_amdgcn_perm(y.y, y.x, 0x06040200u); wj[2 * v + 1] = __builtin_amdgcn_perm(y.w, y.z, 0x06040200u);
                }
            }
            uint32_t tCC = 0, tCi = 0, tCj = 0, tQi = 0, tQj = 0;     // over the whole fill
#pragma unroll
            for (int m_ = 0; m_ < NW4; ++m_) {
                tCC = __builtin_amdgcn_udot4(wi[m_], wj[m_], tCC, false);
                tCi = __builtin_amdgcn_udot4(wi[m_], kOnes, tCi, false); tCj = __builtin_amdgcn_udot4(wj[m_], kOnes, tCj, false);
                tQi = __builtin_amdgcn_udot4(wi[m_], wi[m_], tQi, false); tQj = __builtin_amdgcn_udot4(wj[m_], wj[m_], tQj, false);
            }
            uint32_t dCC = 0, dCi = 0, dCj = 0, dQi = 0, dQj = 0;     // of them, up to the last boundary
            while (b < SUB) {
                uint32_t pCC = 0, pCi = 0, pCj = 0, pQi = 0, pQj = 0;  // sites 0 .. b
                const uint4 *mk4 = reinterpret_cast<const uint4 *>(pmask + b * NW4);
                const uint4 mka = mk4[0], mkb = mk4[1];
                const uint32_t mk[NW4] = {mka.x, mka.y, mka.z, mka.w, mkb.x, mkb.y, mkb.z, mkb.w};
#pragma unroll
                for (int m_ = 0; m_ < NW4; ++m_) {
                    const uint32_t x = wi[m_] & mk[m_], y = wj[m_] & mk[m_];
                    pCC = __builtin_amdgcn_udot4(x, y, pCC, false);
                    pCi = __builtin_amdgcn_udot4(x, kOnes, pCi, false); pCj = __builtin_amdgcn_udot4(y, kOnes, pCj, false);
                    pQi = __builtin_amdgcn_udot4(x, x, pQi, false); pQj = __builtin_amdgcn_udot4(y, y, pQj, false);
                }
                SegRow row;
                parts(aCC + (pCC - dCC), aCi + (pCi - dCi), aCj + (pCj - dCj), aQi + (pQi - dQi), aQj + (pQj - dQj), row.num, row.den, row.k);
                row.end = (gtile0 + ti) * kTileSites + sh + b; row.pad = 0;
                {   // two 16-byte stores into one 32-byte sector
                    uint4 *dst = reinterpret_cast<uint4 *>(p.seg_rows + orow);
                    const uint4 *src = reinterpret_cast<const uint4 *>(&row);
                    dst[0] = src[0]; dst[1] = src[1];
                }
                aCC = aCi = aCj = aQi = aQj = 0;
                dCC = pCC; dCi = pCi; dCj = pCj; dQi = pQi; dQj = pQj;
                ++nflush;
                orow += p.n_pairs;
                const uint32_t rem = b >= SUB - 1 ? 0u : incl & ~((2u << b) - 1u);
                const int c2 = __popc(rem);
                if (p.g <= c2) b = nth_set_bit(rem, p.g);
                else { left = p.g - c2; b = SUB; }
            }
            aCC += tCC - dCC; aCi += tCi - dCi; aCj += tCj - dCj; aQi += tQi - dQi; aQj += tQj - dQj;
        }
        region_sync();                                         // the buffer is free for the piece after next
    }
    if (active) {
        // what the chunk holds behind its last boundary
        const size_t cq = (size_t)kc * p.n_pairs + q;
        const size_t plane = (size_t)p.n_pairs * p.n_chunks;
        double num, den, k;
        parts(aCC, aCi, aCj, aQi, aQj, num, den, k);
        p.tail[cq] = num; p.tail[plane + cq] = den; p.tail[2 * plane + cq] = k;
        p.tail_n[cq] = make_int4((nflush ? p.g : left0) - left, nflush, row0, 0);   // {sites summed in the tail, segments completed, row of the first}
    }
}

template <bool WP, int CB>
static hipError_t launch_clean_walk_t(const FusedParams &p, size_t lds, int threads, hipStream_t st) {
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&clean_walk_kernel<WP, CB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((clean_walk_kernel<WP, CB>), dim3(WP ? (p.n_chunks + kFuseWaves - 1) / kFuseWaves : p.n_chunks), dim3(threads), lds, st, p);
    return hipGetLastError();
}
hipError_t launch_clean_walk(const FusedParams &p, hipStream_t st) {
    if (p.n_chunks <= 0 || p.n_pairs <= 0) return hipSuccess;
    if (p.n_pairs > kK2Threads || p.chunk_dirty == nullptr || p.counts == nullptr) return hipErrorInvalidValue;
    const bool wp = p.n_pairs <= 64;
    const int threads = fused_threads(p.n_pairs);
    size_t lds = clean_walk_lds_bytes(p.P, p.n_pairs, p.piece_tiles, p.compact != 0);
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    // At most three workgroups per CU, by the LDS asked for: the kernel is no slower for it (24.3 us on cfg2 at six, three or
    // two), and the registers the others would hold are what the next batch's K1 runs in beside it (at five workgroups of 88
    // registers a SIMD has 72 left, at three 248: a K1 wave of 128 instead of none once the walk has filled a CU): the
    // overlapped cfg2 step 43.1 -> 41.4 us.
    lds = std::max(lds, (size_t)41000);
    if (p.compact) return wp ? launch_clean_walk_t<true, 1>(p, lds, threads, st) : launch_clean_walk_t<false, 1>(p, lds, threads, st);
    return wp ? launch_clean_walk_t<true, 2>(p, lds, threads, st) : launch_clean_walk_t<false, 2>(p, lds, threads, st);
}

template <typename CT, bool RCP, bool WP, int SUB, bool HBM, bool W3>
static hipError_t launch_fused_t(const FusedParams &p, size_t lds, int threads, int max_blocks, hipStream_t st) {
    const void *fn = reinterpret_cast<const void *>(&fused_pairs_kernel<CT, RCP, WP, SUB, HBM, W3>);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    int grid = max_blocks;
    if (grid <= 0) {
        // workgroups the device holds at once; asked once per shape (the query costs more than a launch)
        static thread_local struct { const void *fn; int threads; size_t lds; int dev; int grid; } memo = {nullptr, 0, 0, -1, 0};
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (memo.fn != fn || memo.threads != threads || memo.lds != lds || memo.dev != dev) {
            int per_cu = 0, cus = 0;
            if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, lds)) != hipSuccess) return e;
            if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
            memo = {fn, threads, lds, dev, std::max(1, per_cu) * std::max(1, cus)};
        }
        grid = memo.grid;
    }
    grid = std::max(1, std::min(grid, HBM && WP ? (p.n_chunks + kFuseWaves - 1) / kFuseWaves : p.n_chunks));
    hipLaunchKernelGGL((fused_pairs_kernel<CT, RCP, WP, SUB, HBM, W3>), dim3(grid), dim3(threads), lds, st, p);
    return hipGetLastError();
}

int fused_threads(int n_pairs) { return n_pairs <= 64 ? 64 * kFuseWaves : 64 * ((n_pairs + 63) / 64); }
int fused_tail_chunks(int n_pairs, int n_chunks) { return n_pairs <= 64 ? n_chunks * kFuseWaves : n_chunks; }

hipError_t launch_fused(const FusedParams &p, int count_bytes, int sub, int max_blocks, hipStream_t st) {
    if (p.n_chunks <= 0 || p.n_pairs <= 0) return hipSuccess;
    if (p.n_pairs > kK2Threads || (sub != 16 && sub != 32)) return hipErrorInvalidValue;
    const bool wp = p.n_pairs <= 64;
    const bool hbm = p.counts != nullptr;
    if (!hbm && wp && p.chunk_tiles % kFuseWaves != 0) return hipErrorInvalidValue;
    const int threads = fused_threads(p.n_pairs);
    const size_t lds = fused_lds_bytes(p.P, p.n_segs, p.max_n, p.n_pairs, hbm ? 2 * p.piece_tiles * (wp ? kFuseWaves : 1) : p.chunk_tiles, count_bytes, sub);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const bool rcp = p.max_n <= kRcpMax;
    // three waves per SIMD where three workgroups of four waves fit a CU's LDS (see W3 above)
    const bool w3 = hbm && !wp && threads <= 256 && 3 * (lds + 1024) <= 160 * 1024;
#define EVOFST_FUSED_GO(CT, R, W_, S_) do { if (hbm && !W_ && w3) return launch_fused_t<CT, R, W_, S_, true, !W_>(p, lds, threads, max_blocks, st); \
                                            if (hbm) return launch_fused_t<CT, R, W_, S_, true, false>(p, lds, threads, max_blocks, st); \
                                            return launch_fused_t<CT, R, W_, S_, false, false>(p, lds, threads, max_blocks, st); } while (0)
#define EVOFST_FUSED_SUB(CT, R, W_) do { if (sub == 16) EVOFST_FUSED_GO(CT, R, W_, 16); else EVOFST_FUSED_GO(CT, R, W_, 32); } while (0)
#define EVOFST_FUSED_WP(CT, R) do { if (wp) EVOFST_FUSED_SUB(CT, R, true); else EVOFST_FUSED_SUB(CT, R, false); } while (0)
    if (count_bytes == 2) { if (rcp) EVOFST_FUSED_WP(uint16_t, true); else EVOFST_FUSED_WP(uint16_t, false); }
    else { if (rcp) EVOFST_FUSED_WP(uint32_t, true); else EVOFST_FUSED_WP(uint32_t, false); }
#undef EVOFST_FUSED_GO
#undef EVOFST_FUSED_SUB
#undef EVOFST_FUSED_WP
    return hipErrorInvalidValue;
}

}  // namespace evofst
