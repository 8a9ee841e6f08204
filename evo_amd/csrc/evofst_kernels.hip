// evofst_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for evo's Fst / dxy hot path.
//
// Built with -ffp-contract=off: every per-site double is evaluated in the reference's operation
// order (process_vcf_fst.h:19-45, process_vcf_fst.cpp:321,346-347) with IEEE division, so per-site
// values are bit-identical to the CPU; only window sums may differ in summation order.
//
//   K1  k1_count_kernel     tiles -> per-(site,pop) {alt, non-missing} counts (+ census of included
//                           sites per chunk and pair, or kept/used masks in against-all mode).
//                           HBM-bound: lane = site, 16-byte coalesced loads, wave-uniform masks.
//   scan_kernel             exclusive prefix of the census over chunks.
//   K2  k2_window_kernel    counts -> per-pop table in LDS -> lane = population pair walks the sites
//                           of its chunk in order, accumulating segment sums (K3 fused).
//   stitch_kernel           completes the segment that straddles a chunk boundary.
//   site_stats / pop_div    per-site outputs on demand (parity hooks and the CLI front-end).
//   global_*                Fst-against-all evaluation and deque windows.
#include <type_traits>
#include "evofst_internal.h"

#include <algorithm>

namespace evofst {


// =====================================================================================
// K1: allele counting.  Replaces the per-sample loops of getVariantCountsForFst
// (process_vcf_fst.cpp:243-280) and getBasicCounts (process_vcf_utils.cpp:62-83).
// =====================================================================================
//
// One wave per chunk (a block = 4 consecutive chunks), lane = site.
// A tile is W rows of 64 x 16 bytes {ALT word, MISSING word}; the wave reads them in batches of 8 rows
// (8 KiB per batch) and requests the next batch -- of this tile or of the chunk's next tile -- before it
// counts the current one.  Counting walks the (word, population) descriptors of the layout, a stream in bit
// order that comes through the scalar cache two entries ahead of its use (wave-uniform masks and flags in SGPRs,
// scalar branches).  The counting loop holds no vector-memory instruction: with ST the packed counts of a tile
// collect in the wave's LDS block and leave as 16-byte stores together with the tile's masks right before the
// wave starts on its next tile, so the only wait on the vector-memory counter is the one for the next batch.
// The zero / fixed ballots of up to 64 populations collect in four VGPRs (v_writelane, lane = population).
typedef uint32_t SegWords __attribute__((ext_vector_type(4)));                  // {mask lo, mask hi, pop, flags | n_full << 8}
typedef const __attribute__((address_space(4))) SegWords *SegStream;
// v_writelane_b32: lane `lane` of the result is `val` (both wave-uniform), the other lanes keep `old`.  This clang has
// no builtin for it; the declaration binds to the LLVM intrinsic, which also routes the lane select through M0.
extern "C" __device__ int evofst_write_lane(int val, int lane, int old) __asm("llvm.amdgcn.writelane.i32");
__device__ __forceinline__ void write_lane4(int &a, int &b, int &c, int &d, uint64_t x, uint64_t y, int lane) {
    a = evofst_write_lane((int)(uint32_t)x, lane, a); b = evofst_write_lane((int)(uint32_t)(x >> 32), lane, b);
    c = evofst_write_lane((int)(uint32_t)y, lane, c); d = evofst_write_lane((int)(uint32_t)(y >> 32), lane, d);
}

#ifdef EVOFST_EXPERIMENTS
#define EVOFST_K1_DBG(bit) && !(p.debug & (bit))
#else
#define EVOFST_K1_DBG(bit)
#endif
template <typename CT, int MODE, bool ST>
__global__ __launch_bounds__(kK1Threads) void k1_count_kernel(K1Params p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int NW = kK1Threads / 64;
    constexpr int NB = 4;   // rows per batch
    constexpr bool kBallots = MODE == MODE_PAIRS || MODE == MODE_GRAM;
    const int P = p.P;
    // LDS: (pairs and Gram mode) per wave [P] zero-masks, [P] fixed-masks (+ [P] nm < 2, [P] nm == 0 in Gram mode), then per
    // wave [n_pairs] census counters, the pair list, and (ST) per wave the [P][64] packed counts of the current tile.
    // The other modes keep no ballots: their workgroups take next to no LDS and fit beside a kernel that fills a CU's
    // (the against-all tile kernel of another batch on another stream).
    uint64_t *zm = reinterpret_cast<uint64_t *>(smem) + (size_t)wave * 4 * P;
    uint64_t *fm = zm + P;
    uint32_t *cen_all = reinterpret_cast<uint32_t *>(reinterpret_cast<uint64_t *>(smem) + (kBallots ? (size_t)NW * 4 * P : 0));
    const int npq = MODE == MODE_PAIRS ? p.n_pairs : 0;
    uint32_t *cen = cen_all + (size_t)wave * npq;
    uint32_t *lpairs = cen_all + (size_t)NW * npq;                                   // [n_pairs]
    CT *stage = reinterpret_cast<CT *>(smem + (((reinterpret_cast<unsigned char *>(lpairs + npq) - smem) + 15) & ~(size_t)15)) +
                (size_t)wave * P * kTileSites;
    const SegStream segs = (SegStream)(uintptr_t)p.segs;                             // [n_segs + 2]: two spare entries behind the stream

    const int chunk = blockIdx.x * NW + wave;
    const bool have = chunk < p.n_chunks;
    const int t0 = min(chunk, p.n_chunks - 1) * p.chunk_tiles;
    const int t1 = have ? min(t0 + p.chunk_tiles, p.n_tiles) : t0;
    const int nb = (p.W + NB - 1) / NB;                 // batches per tile
    const int n_items = (t1 - t0) * nb;
    // four batch buffers in turn: the one being counted and the three requested after it (12 KiB in flight per wave)
    struct Batch { ulonglong2 r[NB]; uint32_t flag; };  // flag: site_flags of the tile, requested with its first batch
    Batch b0, b1, b2, b3;
    b0.flag = b1.flag = b2.flag = b3.flag = 0u;
    // Every request is unconditional -- rows past W, batches past the chunk's end and the flag byte of a run without
    // site_flags all read the first bytes of the wave's chunk (one cache line per instruction, results ignored) -- so the
    // number of loads behind the batch a wave waits for is known at compile time and the wait is a count, not
    // "everything outstanding".
    const bool have_flags = p.site_flags != nullptr;
    const ulonglong2 *dummy = p.tiles + (size_t)t0 * p.W * kTileSites;   // the wave's own first line: no address shared by all waves
    // items (tile, batch) are requested and counted in order: two cursors instead of a division by nb per item
    const int n_own = t1 - t0;
    int l_ti = 0, l_bi = 0, c_ti = 0, c_bi = 0;
    auto load_item = [&](Batch &b) {
        const bool past = l_ti >= n_own;
        const int t = t0 + (past ? 0 : l_ti), wb = (past ? 0 : l_bi) * NB;
        if (++l_bi == nb) { l_bi = 0; ++l_ti; }
        const ulonglong2 *tp = p.tiles + (size_t)t * p.W * kTileSites + lane;
#pragma unroll
        for (int u = 0; u < NB; ++u)
            b.r[u] = (past || wb + u >= p.W) ? *dummy : tp[(size_t)(wb + u) * kTileSites];
        const int64_t site = (int64_t)t * kTileSites + lane;
        const uint8_t *fp = (have_flags && !past && site < p.n_sites) ? p.site_flags + site : reinterpret_cast<const uint8_t *>(dummy);
        b.flag = *fp;
    };
    load_item(b0);                                      // the first batches go out before the pair list is staged
    load_item(b1);
    load_item(b2);
    if (MODE == MODE_PAIRS) {
        if ((int)threadIdx.x < npq) lpairs[threadIdx.x] = p.pairs[threadIdx.x];       // one round trip for up to 256 pairs
        for (int i = threadIdx.x + kK1Threads; i < npq; i += kK1Threads) lpairs[i] = p.pairs[i];
        for (int q = lane; q < p.n_pairs; q += 64) cen[q] = 0;
    }
    uint32_t used_cnt = 0, kept_cnt = 0;
    __syncthreads();

    // per-tile state
    uint32_t c = 0, mc = 0;
    uint32_t Tc = 0, Tn = 0, c0 = 0, nm0 = 0;
    // per-lane flags as 0 / 1 words: as bools they live in scalar register pairs, merged with exec at every join of
    // the segment loop (a dozen scalar instructions per segment word, in a kernel that is short of scalar registers)
    uint32_t irregular = 0, anymiss = 0;
    bool skip = false;
    uint64_t skipmask = 0;
    bool dirty = false;                                 // wave-uniform: the chunk has a site clean_walk_kernel cannot take
    int n_compact = 0;                                  // wave-uniform: the chunk's tiles so far, all written one byte per count
    uint32_t nanany = 0;                                // this lane's site: some population has fewer than two called alleles (0 / 1)
    bool poison = false;                                // wave-uniform, Gram mode: the tile has a site where a population has < 2 called alleles
    int zlo = 0, zhi = 0, flo = 0, fhi = 0, ylo1 = 0, yhi1 = 0, ylo0 = 0, yhi0 = 0;   // ballots by population, lane = pop & 63
    auto wave_lds_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto flush_ballots = [&](int base, int n) {         // populations base .. base + n - 1 -> the wave's mask tables
        if (lane < n) {
            zm[base + lane] = ((uint64_t)(uint32_t)zhi << 32) | (uint32_t)zlo;
            fm[base + lane] = ((uint64_t)(uint32_t)fhi << 32) | (uint32_t)flo;
            if (MODE == MODE_GRAM) {
                zm[2 * P + base + lane] = ((uint64_t)(uint32_t)yhi1 << 32) | (uint32_t)ylo1;
                zm[3 * P + base + lane] = ((uint64_t)(uint32_t)yhi0 << 32) | (uint32_t)ylo0;
            }
        }
        if (MODE == MODE_GRAM) poison = poison || __any(lane < n && (ylo1 | yhi1 | ylo0 | yhi0) != 0);
    };
    // Everything a finished tile writes to HBM.  With ST (P * sizeof(CT) <= 64) the number of store instructions is
    // bounded at compile time, so the wait for the next batch stays a count (not "everything outstanding").
    auto epilogue = [&](int t) {
        if (kBallots) { const int base = ((P - 1) >> 6) << 6; flush_ballots(base, P - base); }
        if (ST || kBallots) wave_lds_sync();
        uint64_t pim = 0, pmm = 0, pnn = 0;                 // pairs mode: the tile's irregular / missing / NaN sites
        bool compact_tile = false;
        if (MODE == MODE_PAIRS) {
            pim = __ballot(irregular != 0 && !skip); pmm = __ballot(anymiss != 0); pnn = __ballot(nanany != 0 && !skip);
            const bool was_dirty = dirty;
            dirty = dirty || (pim | pmm | skipmask) != 0;
            if (ST && sizeof(CT) == 2 && p.compact) {
                compact_tile = !dirty;
                // the chunk's first dirty tile: the tiles before it become packed counts again, last one first (the
                // packed tile k lies behind every one-byte tile below k).  A wave reads back what it stored itself, after
                // its stores have been acknowledged, with loads that pass the CU's cache by.
                if (dirty && !was_dirty && n_compact > 0) {
                    __builtin_amdgcn_s_waitcnt(0x0f70);     // vmcnt(0)
                    unsigned char *cbase = reinterpret_cast<unsigned char *>(p.counts) + (size_t)t0 * P * kTileSites * 2;
                    const int ndw = P * kTileSites / 4;     // one-byte tile in dwords, <= 512
                    for (int k = n_compact - 1; k >= 0; --k) {
                        const uint32_t *src = reinterpret_cast<const uint32_t *>(cbase + (size_t)k * P * kTileSites);
                        uint2 *dst = reinterpret_cast<uint2 *>(cbase + (size_t)k * P * kTileSites * 2);
                        uint32_t r[8], nf[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int dw = lane + 64 * j;
                            r[j] = dw < ndw ? __hip_atomic_load(src + dw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                            nf[j] = dw < ndw ? (uint32_t)p.n_full[dw >> 4] : 0u;
                        }
                        asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));   // every load before the first store
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int dw = lane + 64 * j;
                            const uint32_t hi = (nf[j] << 8) * 0x00010001u;
                            if (dw < ndw) dst[dw] = make_uint2(hi | (r[j] & 0xffu) | ((r[j] & 0xff00u) << 8), hi | ((r[j] >> 16) & 0xffu) | ((r[j] >> 8) & 0xff0000u));
                        }
                        __builtin_amdgcn_s_waitcnt(0x0f70); // the packed tile k is out before the tile below it is read
                    }
                    n_compact = 0;
                }
            }
        }
        if (ST) {
            const uint4 *s4 = reinterpret_cast<const uint4 *>(stage);
            if (compact_tile) {
                uint4 *dst = reinterpret_cast<uint4 *>(reinterpret_cast<unsigned char *>(p.counts) + (size_t)t0 * P * kTileSites * 2 + (size_t)(t - t0) * P * kTileSites);
                const int nv = P * kTileSites / 16;                   // <= 128
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int v = lane + 64 * j;
                    if (v < nv EVOFST_K1_DBG(4)) {
                        const uint4 a = s4[2 * v], b = s4[2 * v + 1];   // 16 packed counts -> their 16 low bytes
                        dst[v] = make_uint4(__builtin_amdgcn_perm(a.y, a.x, 0x06040200u), __builtin_amdgcn_perm(a.w, a.z, 0x06040200u),
                                            __builtin_amdgcn_perm(b.y, b.x, 0x06040200u), __builtin_amdgcn_perm(b.w, b.z, 0x06040200u));
                    }
                }
                ++n_compact;
            } else {
                uint4 *dst = reinterpret_cast<uint4 *>(reinterpret_cast<CT *>(p.counts) + (size_t)t * P * kTileSites);
                const int nv = P * kTileSites * (int)sizeof(CT) / 16;     // <= 256
#pragma unroll
                for (int j = 0; j < 4; ++j) { const int v = lane + 64 * j; if (v < nv EVOFST_K1_DBG(4)) dst[v] = s4[v]; }
            }
        }
        if (MODE == MODE_GRAM) {
            const uint64_t im = __ballot(irregular != 0 && !skip);
            if (lane == 0) { p.irr_mask[t] = im; p.skip_mask[t] = skipmask; p.tile_poison[t] = poison ? 1u : 0u; }
            if (ST) {
#pragma unroll
                for (int j = 0; j < 2; ++j) { const int i = lane + 64 * j; if (i < 4 * P) p.zf_masks[(size_t)t * 4 * P + i] = zm[i]; }
            } else {
                for (int i = lane; i < 4 * P; i += 64) p.zf_masks[(size_t)t * 4 * P + i] = zm[i];   // [tile][z|f|b1|b0][P]
            }
        }
        if (MODE == MODE_PAIRS) {
            if (lane == 0 EVOFST_K1_DBG(8)) { p.irr_mask[t] = pim; p.skip_mask[t] = skipmask; p.miss_mask[t] = pmm; p.nan_mask[t] = pnn; }
            // [tile][z|f][P]: zm, fm are adjacent
            if (ST) { if (lane < 2 * P EVOFST_K1_DBG(16)) p.zf_masks[(size_t)t * 2 * P + lane] = zm[lane]; }
            else for (int i = lane; i < 2 * P; i += 64) p.zf_masks[(size_t)t * 2 * P + i] = zm[i];
            // census of the site filter (process_vcf_fst.cpp:543-544): a site is excluded for pair
            // (i,j) iff both counts are 0 or both pops are fixed for ALT (or the site is skipped).
            for (int q = lane; q < p.n_pairs EVOFST_K1_DBG(32); q += 64) {
                const uint32_t ij = lpairs[q];
                const uint32_t i = ij & 0xffffu, j = ij >> 16;
                const uint64_t ex = (zm[i] & zm[j]) | (fm[i] & fm[j]);
                cen[q] += 64u - (uint32_t)__popcll(ex);
            }
        }
        if (MODE == MODE_GLOBAL) {
            // evo_FstAgainstAll.cpp:198-203 on populationsToUse[0], in exact integer form:
            // AAF == 0 <=> nm > 0 && c == 0 ; AAF == 1 <=> nm > 0 && c == nm (0/0 is NaN, -1 if own nm == 0)
            const uint32_t cc = Tc - c0, cn = Tn - nm0;
            const bool both0 = (nm0 > 0 && c0 == 0) && (cn > 0 && cc == 0);
            const bool both1 = (nm0 > 0 && c0 == nm0) && (cn > 0 && cc == cn);
            const bool kept = !skip && !both0 && !both1;
            const uint64_t km = __ballot(kept);
            p.totals[(int64_t)t * kTileSites + lane] = make_uint2(Tc, Tn);
            if (lane == 0) { p.kept_mask[t] = km; p.skip_mask[t] = skipmask; }
            used_cnt += 64u - (uint32_t)__popcll(skipmask);
            kept_cnt += (uint32_t)__popcll(km);
        }
        if (ST || kBallots) wave_lds_sync();
    };

    int sidx = 0;
    SegWords d = segs[0], dn = segs[1];
    // One batch: the finished tile's stores first, then the request for the batch three ahead (into the buffer that was
    // counted last), then the counting -- so what is outstanding behind the batch waited for is only ever younger loads.
    // Past the chunk's end (the last round of four) a step still requests and takes its batch: with `break`s between
    // the steps the loop header is reached with the buffers requested in different orders, and the compiler's wait there
    // becomes "everything outstanding" (two of the three batches in flight drained at every fourth batch).
    auto step = [&](Batch &cur, Batch &spare) {
        const int ti = c_ti, bi = c_bi;
        if (++c_bi == nb) { c_bi = 0; ++c_ti; }
        const bool live = ti < n_own;
        const int t = t0 + ti, wb = bi * NB;
        if (bi == 0 && live) {
#ifdef EVOFST_EXPERIMENTS
            if (!(p.debug & 2))
#endif
            if (ti > 0) epilogue(t - 1);
            skip = (int64_t)t * kTileSites + lane >= p.n_sites || (have_flags && (cur.flag & 1u) != 0);
            skipmask = __ballot(skip);
            c = 0; mc = 0; Tc = 0; Tn = 0; c0 = 0; nm0 = 0; irregular = 0; anymiss = 0; poison = false; nanany = 0;
            sidx = 0; d = segs[0]; dn = segs[1];
        }
        load_item(spare);
        // the flag byte is the last load of its batch: taking it here marks the whole batch as arrived, including rows
        // past W that nothing reads (their registers would otherwise count as outstanding when they are reused)
        asm volatile("" :: "v"(cur.flag));
        if (!live) return;
        CT *cp = reinterpret_cast<CT *>(p.counts) + (size_t)t * P * kTileSites + lane;
#ifdef EVOFST_EXPERIMENTS
        if (p.debug & 1) { c += (uint32_t)cur.r[0].x ^ (uint32_t)cur.r[1].y ^ (uint32_t)cur.r[2].x ^ (uint32_t)cur.r[3].y; return; }
#endif
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            if (wb + u < p.W) {
                const uint32_t xlo = (uint32_t)cur.r[u].x, xhi = (uint32_t)(cur.r[u].x >> 32);
                const uint32_t ylo = (uint32_t)cur.r[u].y, yhi = (uint32_t)(cur.r[u].y >> 32);
                bool more = true;
                while (more) {
                    const SegWords dnn = segs[sidx + 2];
                    c = popc_acc(xlo & d.x, c); c = popc_acc(xhi & d.y, c);
                    mc = popc_acc(ylo & d.x, mc); mc = popc_acc(yhi & d.y, mc);
                    const uint32_t fl = d.w, pop = d.z;
                    if (fl & kSegLast) {
                        const uint32_t nfull = fl >> kSegNFullShift;
                        const uint32_t nm = nfull - mc;
                        const CT packed = (CT)(c | (nm << CountTraits<CT>::shift));
                        if (ST) stage[pop * kTileSites + lane] = packed;
                        else cp[(size_t)pop * kTileSites] = packed;
                        if (kBallots) {
                            const int pl = (int)(pop & 63u);
                            const uint64_t z = __ballot(c == 0) | skipmask;
                            const uint64_t f = __ballot(c == nm) | skipmask;
                            write_lane4(zlo, zhi, flo, fhi, z, f, pl);
                            if (MODE == MODE_GRAM) {
                                const uint64_t b1 = __ballot(nm < 2) & ~skipmask, b0_ = __ballot(nm == 0) & ~skipmask;
                                write_lane4(ylo1, yhi1, ylo0, yhi0, b1, b0_, pl);
                            }
                            if (pl == 63) flush_ballots((int)pop - 63, 64);
                            // sites K2 must take through its general loop: a NaN in the reference (nm < 2) or an excluded
                            // site with a dxy term (fixed with missing alleles).  Both need a missing allele unless n_full < 2.
                            const bool anym = mc != 0;
                            if (__any(anym) || nfull < 2u) {
                                const uint32_t few = nm < 2u ? 1u : 0u;
                                irregular |= few | ((c == nm && nm < nfull) ? 1u : 0u);
                                nanany |= few;
                                anymiss |= anym ? 1u : 0u;
                            }
                        }
                        if (MODE == MODE_GLOBAL) {
                            if (fl & kSegUse) { Tc += c; Tn += nm; }
                            if (fl & kSegFirstUse) { c0 = c; nm0 = nm; }
                        }
                        c = 0; mc = 0;
                    }
                    more = (fl & kSegWordEnd) == 0u;
                    d = dn; dn = dnn; ++sidx;
                }
            }
        }
    };
    for (int item = 0; item < n_items; item += 4) {
        step(b0, b3);
        step(b1, b0);
        step(b2, b1);
        step(b3, b2);
    }
    if (n_items > 0) epilogue(t1 - 1);
    if (have) {
        if (MODE == MODE_PAIRS) {
            for (int q = lane; q < p.n_pairs; q += 64) p.chunk_census[(size_t)q * p.n_chunks + chunk] = cen[q];
            if (lane == 0) p.chunk_dirty[chunk] = dirty ? 1u : 0u;
        }
        if (MODE == MODE_GLOBAL && lane == 0) {
            p.chunk_used_kept[chunk] = used_cnt;
            p.chunk_used_kept[(size_t)p.n_chunks + chunk] = kept_cnt;
        }
    }
}

hipError_t launch_k1(const K1Params &p_, int count_bytes, int mode, hipStream_t st) {
    if (p_.n_chunks <= 0) return hipSuccess;
    K1Params p = p_;
    p.debug = 0;
#ifdef EVOFST_EXPERIMENTS
    if (const char *e = getenv("EVOFST_DEBUG_K1")) p.debug = atoi(e);
#endif
    constexpr int NW = kK1Threads / 64;
    const size_t npq = mode == MODE_PAIRS ? p.n_pairs : 0;
    const bool ballots = mode == MODE_PAIRS || mode == MODE_GRAM;
    const size_t base = (ballots ? (size_t)NW * 4 * p.P * 8 : 0) + (NW + 1) * npq * 4 + 16;
    const size_t stage = (size_t)NW * p.P * kTileSites * count_bytes;
    // staged: a tile's counts leave as at most four 16-byte stores per lane (more populations: they go out as they complete)
    const bool staged = (size_t)p.P * count_bytes <= 64;
    const size_t lds = base + (staged ? stage : 0) + 64;
    if (lds > 150 * 1024) return hipErrorInvalidValue;   // layouts with thousands of populations: not expected
    dim3 grid((p.n_chunks + NW - 1) / NW), block(kK1Threads);
#define EVOFST_K1S(CT, M, S) do { \
        if (lds > 48 * 1024) { \
            hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&k1_count_kernel<CT, M, S>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e_ != hipSuccess) return e_; \
        } \
        hipLaunchKernelGGL((k1_count_kernel<CT, M, S>), grid, block, lds, st, p); } while (0)
#define EVOFST_K1(CT, M) do { if (staged) EVOFST_K1S(CT, M, true); else EVOFST_K1S(CT, M, false); } while (0)
    if (count_bytes == 2) {
        if (mode == MODE_COUNTS) EVOFST_K1(uint16_t, MODE_COUNTS);
        else if (mode == MODE_PAIRS) EVOFST_K1(uint16_t, MODE_PAIRS);
        else if (mode == MODE_GRAM) EVOFST_K1(uint16_t, MODE_GRAM);
        else EVOFST_K1(uint16_t, MODE_GLOBAL);
    } else {
        if (mode == MODE_COUNTS) EVOFST_K1(uint32_t, MODE_COUNTS);
        else if (mode == MODE_PAIRS) EVOFST_K1(uint32_t, MODE_PAIRS);
        else if (mode == MODE_GRAM) EVOFST_K1(uint32_t, MODE_GRAM);
        else EVOFST_K1(uint32_t, MODE_GLOBAL);
    }
#undef EVOFST_K1
#undef EVOFST_K1S
    return hipGetLastError();
}

// =====================================================================================
// Exclusive scan of the census over chunks, one 1024-thread block per column (pair):
// offsets[q][c] = base[q] + sum_{c'<c} census[q][c'].  A thread owns 4 consecutive chunks, so a
// pass covers 4096 chunks with coalesced loads and stores; wave shuffles + one LDS exchange.
// =====================================================================================
constexpr int kScanThreads = 1024, kScanPer = 4;
__global__ __launch_bounds__(kScanThreads) void scan_kernel(const uint32_t *__restrict__ census,
                                                            const int64_t *base,      // may be base_out (run_scan passes d_base for both)
                                                            int64_t *__restrict__ offsets,
                                                            int64_t *__restrict__ totals, int n_chunks, int n_cols,
                                                            int64_t g, int64_t *__restrict__ first_seg,
                                                            const int64_t *__restrict__ gathered, int64_t row_stride, int n_rows,
                                                            int64_t *base_out, int64_t *__restrict__ totals_copy) {
    __shared__ uint32_t wsum[kScanThreads / 64];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int q = blockIdx.x;
    const uint32_t *in = census + (size_t)q * n_chunks;
    int64_t *out = offsets + (size_t)q * n_chunks;
    int64_t b0 = (base && !gathered) ? base[q] : 0;
    if (gathered) {   // the base offset is the sum of the earlier ranks' totals, straight from the all-gather buffer
        b0 = 0;
        for (int r = 0; r < n_rows; ++r) b0 += gathered[(size_t)r * row_stride + q];
        if (threadIdx.x == 0) base_out[q] = b0;
    }
    int64_t carry = b0;
    for (int c0 = 0; c0 < n_chunks; c0 += kScanThreads * kScanPer) {
        const int first = c0 + threadIdx.x * kScanPer;
        uint32_t v[kScanPer];
#pragma unroll
        for (int i = 0; i < kScanPer; ++i) v[i] = in[min(first + i, n_chunks - 1)];   // unconditional: no branch per load
        uint32_t mine = 0;
#pragma unroll
        for (int i = 0; i < kScanPer; ++i) { v[i] = first + i < n_chunks ? v[i] : 0u; mine += v[i]; }
        uint32_t x = mine;  // inclusive scan of the thread totals inside the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x, d);
            if (lane >= d) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        uint32_t woff = 0, all = 0;
#pragma unroll
        for (int w = 0; w < kScanThreads / 64; ++w) { const uint32_t t = wsum[w]; woff += w < wave ? t : 0u; all += t; }
        int64_t run = carry + (int64_t)woff + (int64_t)(x - mine);
#pragma unroll
        for (int i = 0; i < kScanPer; ++i) {
            if (first + i < n_chunks) out[first + i] = run;
            run += v[i];
        }
        carry += (int64_t)all;
        __syncthreads();
    }
    if (threadIdx.x == 0 && totals) totals[q] = carry - b0;
    if (threadIdx.x == 0 && totals_copy) totals_copy[q] = carry - b0;   // the caller's buffer (evofst_copy_included_totals)
    if (threadIdx.x == 0 && first_seg) first_seg[q] = b0 / g;    // segment-table row 0 of the pair in this shard
}

hipError_t launch_scan(const uint32_t *census, const int64_t *base, int64_t *offsets, int64_t *totals,
                       int n_chunks, int n_cols, int64_t g, int64_t *first_seg, hipStream_t st,
                       const int64_t *gathered, int64_t row_stride, int n_rows, int64_t *base_out, int64_t *totals_copy) {
    if (n_cols <= 0 || n_chunks <= 0) return hipSuccess;
    hipLaunchKernelGGL(scan_kernel, dim3(n_cols), dim3(kScanThreads), 0, st, census, base, offsets, totals, n_chunks, n_cols,
                       g, first_seg, gathered, row_stride, n_rows, base_out, totals_copy);
    return hipGetLastError();
}

// =====================================================================================
// K2 (+K3): per-pair window sums of the Hudson numerator / denominator and of the dxy numerator.
// Replaces process_vcf_fst.cpp:543-553 + 638-650 for every population pair at once.
//
// Window sums only have to agree with the reference to 1e-12 (the per-site values, which are
// bit-exact, come from site_stats_kernel), so the sums are restated so that a lane (= one pair)
// does as little as possible per site.  With d = p_i - p_j, t = p(1-p), h = t/(nm-1):
//     num = d^2 - h_i - h_j                      (process_vcf_fst.h:19-25)
//     den = p_i(1-p_j) + p_j(1-p_i) = d^2 + t_i + t_j      (process_vcf_fst.h:27-30; all terms >= 0)
//     k   = c_i(n_j-c_j) + c_j(n_i-c_i) = n_j c_i + n_i c_j - 2 c_i c_j   (process_vcf_fst.cpp:297-321)
// A site the filter (fst.cpp:543-544) excludes for a pair has p_i = p_j in {0,1}: d, t and h are 0
// there (t and h exactly: t = c(nm-c)/nm^2 is formed from the integer product), so nothing has to be
// masked.  Per site a lane therefore accumulates only D = sum d^2 (one FMA) and CC = sum c_i c_j (one
// integer MAD); sum h, sum t and sum c are per-population prefix sums from the start of the chunk,
// kept in LDS and read only at window boundaries.  All accumulators run from the start of the chunk; a
// segment sum is the difference of two boundary snapshots.  The boundaries of a half tile come from
// K1's zero / fixed masks per population (include = ~((Z_i & Z_j) | (F_i & F_j))), a popcount and an
// n-th-set-bit search, so the site loop has no boundary test: it snapshots the accumulators once per
// group of kGrp sites, and the few sites behind the boundary are taken out again when the segment is
// written (once per half tile and lane).
//
// Half tiles in which some population has < 2 called alleles (the reference's 0/0 = NaN cases) or is
// fixed with missing alleles (c == nm < n: excluded sites with a non-zero dxy term) -- K1 flags them --
// take the general loop, which masks D and the integer sums by the site filter and counts the sites
// whose value is NaN in the reference.
//
// Reciprocals come from a per-block LDS table {1/nm, 1/nm^2, 1/(nm-1), flags} (0 where undefined).
// =====================================================================================
size_t k2_lds_bytes(int P, int max_n) {
    return (size_t)P * kSubPad * (3 * 8 + 2 * 4) + (size_t)P * (2 * 8 + 4 + 8) +
           (max_n <= kRcpMax ? (size_t)(max_n + 1) * sizeof(RcpEntry) : 0) + 64;
}


template <typename CT, bool use_rcp>
__global__ __launch_bounds__(kK2Threads, kK2WavesPerSimd) void k2_window_kernel(K2Params p) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int EPT = 16 / (int)sizeof(CT);     // sites per 16-byte count vector
    constexpr int VPR = kSub / EPT;               // vectors per population row of a half tile
    constexpr int EB = 8 * (int)sizeof(CT);
    const int P = p.P;
    RcpEntry *rcp = reinterpret_cast<RcpEntry *>(smem);                           // [max_n+1] when use_rcp
    double *pt = reinterpret_cast<double *>(rcp + (use_rcp ? p.max_n + 1 : 0));   // [P][33] p = c/nm
    double *PH = pt + (size_t)P * kSubPad;                    // [P][33] sum of h from the start of the chunk
    double *PT = PH + (size_t)P * kSubPad;                    // [P][33] sum of t
    double *carH = PT + (size_t)P * kSubPad;                  // [P] the same sums after the last derived half tile
    double *carT = carH + P;
    uint32_t *xt = reinterpret_cast<uint32_t *>(carT + P);    // [P][33] c | flags
    uint32_t *PC = xt + (size_t)P * kSubPad;                  // [P][33] sum of c
    uint32_t *carC = PC + (size_t)P * kSubPad;                // [P]
    uint32_t *zfm = carC + P;                                 // [2][P] zero / fixed masks of the half tile

    const int nthreads = blockDim.x;
    const int chunk = blockIdx.x;
    const int q = blockIdx.y * nthreads + threadIdx.x;
    const bool active = q < p.n_pairs;
    uint32_t pi = 0, pj = 0;
    int64_t m = 0;
    int32_t left = p.g;
    uint32_t ni = 0, nj = 0;
    if (active) {
        const uint32_t ij = p.pairs[q];
        pi = ij & 0xffffu; pj = ij >> 16;
        ni = (uint32_t)p.n_full[pi]; nj = (uint32_t)p.n_full[pj];
        const int64_t g0 = p.chunk_offsets[(size_t)q * p.n_chunks + chunk];
        m = g0 / p.g;
        left = p.g - (int32_t)(g0 - m * p.g);
    }
    const int32_t left0 = left;
    // clean half tiles (no missing allele, no skipped site; 8-bit counts): exact integer sums, scaled once per segment
    constexpr bool kInt = sizeof(CT) == 2;
    const uint64_t ni2 = (uint64_t)ni * ni, nj2 = (uint64_t)nj * nj, ninj = (uint64_t)ni * nj;
    const double rE = active ? 1.0 / ((double)ninj * (double)ninj) : 0.0;          // d^2 = (c_i n_j - c_j n_i)^2 / (n_i n_j)^2
    const double rTi = active ? 1.0 / (double)ni2 : 0.0, rTj = active ? 1.0 / (double)nj2 : 0.0;   // t = c (n - c) / n^2
    const double rHi = ni > 1 ? rTi / (double)(ni - 1) : 0.0, rHj = nj > 1 ? rTj / (double)(nj - 1) : 0.0;   // h = t / (n - 1)
    uint64_t aCC = 0, aCi = 0, aCj = 0, aQi = 0, aQj = 0;   // clean half tiles since the last boundary: sums of c_i c_j, c, c^2
    // integer sums of clean half tiles -> their share of the numerator / denominator sums and of sumKij
    auto int_parts = [&](uint64_t iCC, uint64_t iCi, uint64_t iCj, uint64_t iQi, uint64_t iQj, double &numI, double &denI, uint64_t &kI) {
        const uint64_t E = nj2 * iQi + ni2 * iQj - 2ull * ninj * iCC;       // sum (c_i n_j - c_j n_i)^2, exact
        const double dTi = (double)((uint64_t)ni * iCi - iQi), dTj = (double)((uint64_t)nj * iCj - iQj);   // sum c (n - c)
        const double dE = (double)E * rE;
        numI = dE - dTi * rHi - dTj * rHj;
        denI = dE + dTi * rTi + dTj * rTj;
        kI += (uint64_t)nj * iCi + (uint64_t)ni * iCj - 2ull * iCC;
    };
    // Segment-table element of the next segment this lane completes.  The chunk's first one is only a part
    // (the stitch kernel adds what earlier chunks hold of it).
    const int32_t row0 = active ? (int32_t)(m - p.first_segment[q]) : 0;
    size_t orow = (size_t)row0 * p.n_pairs + q;
    // accumulators from the start of the chunk
    double D = 0.0;                       // sum d^2 (masked by the site filter in general half tiles)
    uint64_t cc = 0;                      // sum c_i c_j
    uint32_t exi = 0, exj = 0;            // sum of c_i / c_j over excluded sites (general half tiles only)
    uint32_t bn = 0, bd = 0;              // included sites whose numerator / denominator is NaN in the reference
    // their values at the last boundary
    double prevN = 0.0, prevD = 0.0;
    uint32_t prevCi = 0, prevCj = 0, prevBn = 0, prevBd = 0;
    uint64_t prevCC = 0;
    int32_t nflush = 0;
    for (int i = threadIdx.x; i < P; i += nthreads) { carH[i] = 0.0; carT[i] = 0.0; carC[i] = 0u; }
    if (use_rcp)
        for (int i = threadIdx.x; i <= p.max_n; i += nthreads) {
            RcpEntry e;
            e.r0 = i ? 1.0 / (double)i : 0.0;
            e.rr0 = e.r0 * e.r0;
            e.r1 = i > 1 ? 1.0 / (double)(i - 1) : 0.0;
            e.flags = (i < 2 ? kXBadNum : 0u) | (i == 0 ? kXBadDen : 0u);
            e.pad = 0;
            rcp[i] = e;
        }

    const int t0 = chunk * p.chunk_tiles;
    const int t1 = min(t0 + p.chunk_tiles, p.n_tiles);
    const uint32_t io = pi * kSubPad, jo = pj * kSubPad;
    __syncthreads();
    // The counts of a tile ([P][64] packed CT) are fetched as 16-byte vectors (EPT consecutive sites of one
    // population); the vectors of the next half tile are requested before the pair phase of this one.
    constexpr int NV = 2;
    const int nvec = P * VPR;
    uint4 pre[NV];
    uint64_t pre_zf = 0, pre_skip = 0, pre_irr = 0, pre_miss = 0;   // K1's masks of the tile of the prefetched half tile
    auto fetch = [&](int st) {               // st = 2*tile + half
        if ((st & 1) == 0 || st == 2 * t0) {
            const int tile = st >> 1;
            pre_zf = p.zf_masks[(size_t)tile * 2 * P + min((int)threadIdx.x, 2 * P - 1)];   // (P > nthreads / 2: see below)
            pre_skip = p.skip_mask[tile]; pre_irr = p.irr_mask[tile]; pre_miss = p.miss_mask[tile];
        }
        const uint4 *vp = reinterpret_cast<const uint4 *>(reinterpret_cast<const CT *>(p.counts) + (size_t)(st >> 1) * P * kTileSites);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int vi = min((int)threadIdx.x + j * nthreads, nvec - 1);   // unconditional: loads stay in flight together
            const int k = vi / VPR, r = vi - k * VPR;
            pre[j] = vp[k * (kTileSites / EPT) + (st & 1) * VPR + r];
        }
    };
    // One vector = EPT consecutive sites of population k.  skipbits: the vector's bits of the tile's skip mask.
    auto derive_vec = [&](int vi, const uint4 &vv, uint32_t skiphalf) {
        const int k = vi / VPR, r = vi - k * VPR;
        const int o = k * kSubPad + r * EPT;
        const uint32_t skipbits = skiphalf >> (r * EPT);
        uint32_t w[4] = {vv.x, vv.y, vv.z, vv.w};
        if (skiphalf != 0u) {                 // uniform and rare: a skipped site contributes 0 everywhere (c = 0)
#pragma unroll
            for (int j = 0; j < EPT; ++j)
                if ((skipbits >> j) & 1u) w[(j * EB) / 32] &= ~(CountTraits<CT>::mask << ((j * EB) % 32));
        }
        double tv[EPT], hv[EPT];
        uint32_t cv[EPT];
        double rh = 0.0, rt = 0.0;           // sums over this vector's sites
        uint32_t rc = 0;
        // the table entries of 4 sites at a time: the compiler cannot move an LDS read of rcp[] above an LDS write of
        // pt[] / xt[] (same shared array), so reads issued site by site would wait for each other; all 8 at once cost
        // 56 registers at the kernel's pressure peak (spills: 10 MB of scratch traffic on cfg2)
        RcpEntry ent[use_rcp ? EPT : 1];
        constexpr int HB = EVOFST_K2_ENT_BATCH;      // entries requested at a time
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
                const double dn = (double)nm;
                pp = nm ? dc / dn : 0.0;                              // utils.h:272-275
                tt = pp * (1.0 - pp);
                hh = nm > 1 ? tt / (dn - 1.0) : 0.0;                  // fst.h:21-22
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
        const double ch = carH[k], ct = carT[k];
        const uint32_t ccar = carC[k];
        if (r == VPR - 1) { carH[k] = ch + eh; carT[k] = ct + et; carC[k] = ccar + ec; }
        double ah = ch + (eh - rh), at = ct + (et - rt);       // sums before this vector
        uint32_t ac = ccar + (ec - rc);
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            ah += hv[j]; at += tv[j]; ac += cv[j];
            PH[o + j] = ah; PT[o + j] = at; PC[o + j] = ac;
        }
    };
    const double kNaN = __longlong_as_double(0x7ff8000000000000ll);
    const int st0 = 2 * t0, st1 = 2 * t1;
    if (st0 < st1) fetch(st0);
    for (int st = st0; st < st1; ++st) {
        const int tile = st >> 1, sh = (st & 1) * kSub;
        // ---- derive the per-(pop, site) tables of this half tile
        const uint32_t skiphalf = (uint32_t)(pre_skip >> sh);
        const bool general = (uint32_t)(pre_irr >> sh) != 0u;   // uniform
        const bool clean = kInt && (uint32_t)((pre_miss | pre_skip | pre_irr) >> sh) == 0u;   // uniform
        uint4 *stage = reinterpret_cast<uint4 *>(pt);           // clean half tiles: the raw count vectors [P][VPR] instead of tables
        if (clean) {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int vi = threadIdx.x + j * nthreads;
                if (vi < nvec) stage[vi] = pre[j];
            }
            const uint4 *vp = reinterpret_cast<const uint4 *>(reinterpret_cast<const CT *>(p.counts) + (size_t)tile * P * kTileSites);
            for (int vi = threadIdx.x + NV * nthreads; vi < nvec; vi += nthreads) {
                const int k = vi / VPR, r = vi - k * VPR;
                stage[vi] = vp[k * (kTileSites / EPT) + (st & 1) * VPR + r];
            }
        }
        if (!clean) {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int vi = threadIdx.x + j * nthreads;
                if (vi < nvec) derive_vec(vi, pre[j], skiphalf);
            }
            const uint4 *vp = reinterpret_cast<const uint4 *>(reinterpret_cast<const CT *>(p.counts) + (size_t)tile * P * kTileSites);
            for (int vi = threadIdx.x + NV * nthreads; vi < nvec; vi += nthreads) {
                const int k = vi / VPR, r = vi - k * VPR;
                derive_vec(vi, vp[k * (kTileSites / EPT) + (st & 1) * VPR + r], skiphalf);
            }
        }
        if ((int)threadIdx.x < 2 * P) zfm[threadIdx.x] = (uint32_t)(pre_zf >> sh);
        for (int i = threadIdx.x + nthreads; i < 2 * P; i += nthreads)        // more than nthreads / 2 populations
            zfm[i] = (uint32_t)(p.zf_masks[(size_t)tile * 2 * P + i] >> sh);
        if (st + 1 < st1) fetch(st + 1);
        __syncthreads();
        // ---- pairs: lane = pair
        if (active) {
            const uint32_t incl = ~((zfm[pi] & zfm[pj]) | (zfm[P + pi] & zfm[P + pj]));   // fst.cpp:543-544
            int b = kSub;                                  // site of this lane's next boundary in the half tile (kSub: none)
            {
                const int cnt = __popc(incl);
                if (left <= cnt) b = nth_set_bit(incl, left);
                else left -= cnt;
            }
            // emit: the accumulators right after boundary site b -> one row of the segment table, then the next boundary.
            // H, T, PCi, PCj: the per-population prefix sums at b; i*: the integer sums of the segment's clean half tiles.
            auto emit = [&](double cD, uint64_t ccc, uint32_t cexi, uint32_t cexj, uint32_t cbn, uint32_t cbd, double H, double T,
                            uint32_t PCi, uint32_t PCj, uint64_t iCC, uint64_t iCi, uint64_t iCj, uint64_t iQi, uint64_t iQj) {
                const uint32_t Ci = PCi - cexi, Cj = PCj - cexj;
                const double cumN = cD - H, cumD = cD + T;
                // sumKij (fst.cpp:299-321) of the segment, exact: differences are taken modulo 2^32 / 2^64
                uint64_t kk = (uint64_t)nj * (uint64_t)(Ci - prevCi) + (uint64_t)ni * (uint64_t)(Cj - prevCj) - 2ull * (ccc - prevCC);
                double numI = 0.0, denI = 0.0;
                if (kInt) int_parts(iCC, iCi, iCj, iQi, iQj, numI, denI, kk);
                SegRow row;
                row.num = cbn != prevBn ? kNaN : (cumN - prevN) + numI;
                row.den = cbd != prevBd ? kNaN : (cumD - prevD) + denI;
                row.k = (double)kk;
                row.end = st * kSub + b; row.pad = 0;
                {   // two 16-byte stores into one 32-byte sector
                    uint4 *dst = reinterpret_cast<uint4 *>(p.seg_rows + orow);
                    const uint4 *src = reinterpret_cast<const uint4 *>(&row);
                    dst[0] = src[0]; dst[1] = src[1];
                }
                prevN = cumN; prevD = cumD; prevCi = Ci; prevCj = Cj; prevCC = ccc; prevBn = cbn; prevBd = cbd;
                ++nflush;
                orow += p.n_pairs;
                const uint32_t rem = b >= kSub - 1 ? 0u : incl & ~((2u << b) - 1u);
                const int c2 = __popc(rem);
                if (p.g <= c2) b = nth_set_bit(rem, p.g);
                else { left = p.g - c2; b = kSub; }
            };
            if (clean) {
                // ---- no missing allele and no skipped site in this half tile: p = c / n with constant n, so the sums of d^2,
                // t and h over a range of sites follow exactly from the integer sums of c_i c_j, c and c^2 (4 sites per
                // v_dot4_u32_u8); no tables are derived
                constexpr int NW4 = kSub / 4;
                constexpr uint32_t kOnes = 0x01010101u;
                uint32_t wi[NW4], wj[NW4];
#pragma unroll
                for (int v = 0; v < kSub / 8; ++v) {       // 8 packed {c, nm} pairs -> two words of 4 c bytes
                    const uint4 x = stage[pi * VPR + v], y = stage[pj * VPR + v];
                    wi[2 * v] = __builtin_amdgcn_perm(x.y, x.x, 0x06040200u); wi[2 * v + 1] = __builtin_amdgcn_perm(x.w, x.z, 0x06040200u);
                    wj[2 * v] = __builtin_amdgcn_perm(y.y, y.x, 0x06040200u); wj[2 * v + 1] = __builtin_amdgcn_perm(y.w, y.z, 0x06040200u);
                }
                uint32_t tCC = 0, tCi = 0, tCj = 0, tQi = 0, tQj = 0;     // over the whole half tile (< 2^22)
#pragma unroll
                for (int m_ = 0; m_ < NW4; ++m_) {
                    tCC = __builtin_amdgcn_udot4(wi[m_], wj[m_], tCC, false);
                    tCi = __builtin_amdgcn_udot4(wi[m_], kOnes, tCi, false); tCj = __builtin_amdgcn_udot4(wj[m_], kOnes, tCj, false);
                    tQi = __builtin_amdgcn_udot4(wi[m_], wi[m_], tQi, false); tQj = __builtin_amdgcn_udot4(wj[m_], wj[m_], tQj, false);
                }
                uint32_t dCC = 0, dCi = 0, dCj = 0, dQi = 0, dQj = 0;     // of them, up to the last boundary
                while (b < kSub) {
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
                    emit(D, cc, exi, exj, bn, bd, carH[pi] + carH[pj], carT[pi] + carT[pj], carC[pi], carC[pj],
                         aCC + (pCC - dCC), aCi + (pCi - dCi), aCj + (pCj - dCj), aQi + (pQi - dQi), aQj + (pQj - dQj));
                    aCC = aCi = aCj = aQi = aQj = 0;
                    dCC = pCC; dCi = pCi; dCj = pCj; dQi = pQi; dQj = pQj;
                }
                aCC += tCC - dCC; aCi += tCi - dCi; aCj += tCj - dCj; aQi += tQi - dQi; aQj += tQj - dQj;
            } else {
            const int bgrp = b & ~(kGrp - 1);
            double sD = D;                                 // snapshots at the end of the boundary's group
            uint64_t scc = cc;
            uint32_t sexi = exi, sexj = exj, sbn = bn, sbd = bd;   // (the regular loop leaves these four alone)
            if (!general) {
#pragma unroll EVOFST_K2_UNROLL
                for (int sb = 0; sb < kSub; sb += kGrp) {
                    double a[kGrp], bb[kGrp];
                    uint32_t xi[kGrp], xj[kGrp];
#pragma unroll
                    for (int u = 0; u < kGrp; ++u) {
                        a[u] = pt[io + sb + u]; bb[u] = pt[jo + sb + u];
                        xi[u] = xt[io + sb + u]; xj[u] = xt[jo + sb + u];
                    }
                    uint32_t cch = 0;                      // c_i c_j of kGrp sites (< 2^30)
#pragma unroll
                    for (int u = 0; u < kGrp; ++u) {
                        const double d = a[u] - bb[u];
                        D = fma(d, d, D);
                        cch += __umul24(xi[u], xj[u]);     // low 24 bits: c (the flags sit above)
                    }
                    cc += cch;
                    if (sb == bgrp) { sD = D; scc = cc; }
                }
            } else {
#pragma unroll 1
                for (int sb = 0; sb < kSub; sb += kGrp) {
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
                emit(cD, ccc, cexi, cexj, cbn, cbd, PH[io + b] + PH[jo + b], PT[io + b] + PT[jo + b], PC[io + b], PC[jo + b],
                     aCC, aCi, aCj, aQi, aQj);
                aCC = aCi = aCj = aQi = aQj = 0;
            };
            if (b < kSub) {
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
                        const int s = u <= room ? b + u : b;
                        pa[u - 1] = pt[io + s]; pb[u - 1] = pt[jo + s]; xa[u - 1] = xt[io + s]; xb[u - 1] = xt[jo + s];
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
                // further boundaries of the same half tile (windows shorter than it): from the end of the half tile
                while (b < kSub) {
                    double cD = D;
                    uint64_t ccc = cc;
                    uint32_t cexi = exi, cexj = exj, cbn = bn, cbd = bd;
                    for (int s = b + 1; s < kSub; ++s) {
                        const uint32_t xi = xt[io + s], xj = xt[jo + s];
                        const double d = pt[io + s] - pt[jo + s];
                        const uint32_t ci = xi & kXMask, cj = xj & kXMask;
                        const bool inc = !general || ((incl >> s) & 1u) != 0u;
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
                    }   // general half tile
        }
        __syncthreads();
    }
    if (active) {
        // what the chunk holds behind its last boundary
        const size_t cq = (size_t)chunk * p.n_pairs + q;        // [chunk][pair] planes: a wave writes a contiguous row
        const size_t plane = (size_t)p.n_pairs * p.n_chunks;
        const double H = carH[pi] + carH[pj], T = carT[pi] + carT[pj];
        const uint32_t Ci = carC[pi] - exi, Cj = carC[pj] - exj;
        uint64_t kk = (uint64_t)nj * (uint64_t)(Ci - prevCi) + (uint64_t)ni * (uint64_t)(Cj - prevCj) - 2ull * (cc - prevCC);
        double numI = 0.0, denI = 0.0;
        if (kInt) int_parts(aCC, aCi, aCj, aQi, aQj, numI, denI, kk);
        p.tail[cq] = bn != prevBn ? kNaN : ((D - H) - prevN) + numI;
        p.tail[plane + cq] = bd != prevBd ? kNaN : ((D + T) - prevD) + denI;
        p.tail[2 * plane + cq] = (double)kk;
        p.tail_n[cq] = make_int4((nflush ? p.g : left0) - left, nflush, row0, 0);   // {sites summed in the tail, segments completed, row of the first}
    }
}

template <typename CT, bool RCP>
static hipError_t launch_k2_t(const K2Params &p, size_t lds, dim3 grid, dim3 block, hipStream_t st) {
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k2_window_kernel<CT, RCP>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k2_window_kernel<CT, RCP>), grid, block, lds, st, p);
    return hipGetLastError();
}

hipError_t launch_k2(const K2Params &p, int count_bytes, hipStream_t st) {
    if (p.n_chunks <= 0 || p.n_pairs <= 0) return hipSuccess;
    const int threads = 64 * ((std::min(p.n_pairs, kK2Threads) + 63) / 64);
    dim3 grid(p.n_chunks, (p.n_pairs + threads - 1) / threads), block(threads);
    size_t lds = k2_lds_bytes(p.P, p.max_n);
#ifdef EVOFST_EXPERIMENTS
    if (const char *e = getenv("EVOFST_DEBUG_K2_LDS")) lds += (size_t)atoi(e);   // occupancy experiments
#endif
    const bool rcp = p.max_n <= kRcpMax;
    if (count_bytes == 2) return rcp ? launch_k2_t<uint16_t, true>(p, lds, grid, block, st) : launch_k2_t<uint16_t, false>(p, lds, grid, block, st);
    return rcp ? launch_k2_t<uint32_t, true>(p, lds, grid, block, st) : launch_k2_t<uint32_t, false>(p, lds, grid, block, st);
}

// The first segment a chunk completes started in an earlier chunk: K2 wrote the chunk's own part into
// the segment's row; add, in site order, the tail of the last chunk that completed a segment and the
// whole-chunk sums of the chunks in between.
__global__ __launch_bounds__(256) void stitch_kernel(K2Params p) {
    const size_t cq = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // [chunk][pair], flat
    const size_t plane = (size_t)p.n_pairs * p.n_chunks;
    if (cq >= plane) return;
    const int c = (int)(cq / p.n_pairs), q = (int)(cq - (size_t)c * p.n_pairs);
    // The usual case is that the chunk before completed a segment too, so its tail is all there is to add: ask for
    // it together with this chunk's own entry instead of after it (the kernel is a chain of dependent loads).
    const bool has_prev = c > 0;
    const size_t op = has_prev ? cq - (size_t)p.n_pairs : cq;
    const int4 me = p.tail_n[cq];
    const int4 pv = p.tail_n[op];
    const double pn = p.tail[op], pd = p.tail[plane + op], pk = p.tail[2 * plane + op];
    if (me.y == 0) return;                                  // this chunk completed no segment
    if ((uint64_t)(int64_t)me.z >= (uint64_t)p.seg_cap) return;      // (never from a healthy walk: no store outside the table whatever the words say)
    const size_t o = (size_t)me.z * p.n_pairs + q;
    SegRow &row = p.seg_rows[o];
    const double rn = row.num, rd = row.den, rk = row.k;
    double an = 0.0, ad = 0.0, ak = 0.0;
    if (has_prev && pv.y != 0) {
        an += pn; ad += pd; ak += pk;
    } else if (has_prev) {
        int cp = c - 2;
        while (cp >= 0 && p.tail_n[(size_t)cp * p.n_pairs + q].y == 0) --cp;
        for (int cc_ = (cp < 0 ? 0 : cp); cc_ < c; ++cc_) {
            const size_t oc = (size_t)cc_ * p.n_pairs + q;
            an += p.tail[oc]; ad += p.tail[plane + oc]; ak += p.tail[2 * plane + oc];
        }
    }
    row.num = an + rn; row.den = ad + rd; row.k = ak + rk;
}

// the segment table as the four arrays the host getter hands out
__global__ __launch_bounds__(256) void unpack_segments_kernel(const SegRow *__restrict__ rows, size_t n, double *__restrict__ num,
                                                              double *__restrict__ den, double *__restrict__ k, int32_t *__restrict__ end) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const SegRow r = rows[i];
    if (num) num[i] = r.num;
    if (den) den[i] = r.den;
    if (k) k[i] = r.k;
    if (end) end[i] = r.end;
}

hipError_t launch_unpack_segments(const SegRow *rows, size_t n, double *num, double *den, double *k, int32_t *end, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(unpack_segments_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, rows, n, num, den, k, end);
    return hipGetLastError();
}

// how many chunks K1 marked for the general walk (one block; asked once per data set, at evofst_sync)
__global__ __launch_bounds__(256) void count_marks_kernel(const uint32_t *__restrict__ marks, int n, uint32_t *__restrict__ out) {
    __shared__ uint32_t part[256];
    uint32_t v = 0;
    for (int i = threadIdx.x; i < n; i += 256) v += marks[i] != 0u ? 1u : 0u;
    part[threadIdx.x] = v;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = part[0];
}
hipError_t launch_count_marks(const uint32_t *marks, int n, uint32_t *out, hipStream_t st) {
    hipLaunchKernelGGL(count_marks_kernel, dim3(1), dim3(256), 0, st, marks, n, out);
    return hipGetLastError();
}

hipError_t launch_stitch(const K2Params &p, hipStream_t st) {
    if (p.n_chunks <= 0 || p.n_pairs <= 0) return hipSuccess;
    const size_t n = (size_t)p.n_pairs * p.n_chunks;
    hipLaunchKernelGGL(stitch_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p);
    return hipGetLastError();
}

// Whole-shard sums per pair: complete segments in order, then the carry after the last one.
// totals[q] = {num, den, k, n_included}; carry[q] = {num, den, k, n_sites} behind the last complete segment.
// (k = sumKij is a sum of integers < 2^53 held in doubles: exact in any order.)
__global__ void totals_kernel(K2Params p, const int64_t *__restrict__ base, const int64_t *__restrict__ incl_totals,
                              double *__restrict__ totals, double *__restrict__ carry) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= p.n_pairs) return;
    const size_t plane = (size_t)p.n_pairs * p.n_chunks;
    double an = 0.0, ad = 0.0, ak = 0.0;
    const int64_t rows = (base[q] + incl_totals[q]) / p.g - base[q] / p.g;   // complete segments of this shard
    if (totals)
        for (int64_t r = 0; r < rows; ++r) {
            const size_t o = (size_t)r * p.n_pairs + q;
            const SegRow row = p.seg_rows[o];
            an += row.num; ad += row.den; ak += row.k;
        }
    int cl = p.n_chunks - 1;
    while (cl >= 0 && p.tail_n[(size_t)cl * p.n_pairs + q].y == 0) --cl;
    double tn = 0.0, td = 0.0, tk = 0.0, cnt = 0.0;
    for (int c = (cl < 0 ? 0 : cl); c < p.n_chunks; ++c) {
        const size_t o = (size_t)c * p.n_pairs + q;
        tn += p.tail[o]; td += p.tail[plane + o]; tk += p.tail[2 * plane + o];
        cnt += (double)p.tail_n[o].x;
    }
    if (carry) { carry[(size_t)q * 4] = tn; carry[(size_t)q * 4 + 1] = td; carry[(size_t)q * 4 + 2] = tk; carry[(size_t)q * 4 + 3] = cnt; }
    if (totals) {
        totals[(size_t)q * 4 + 0] = an + tn; totals[(size_t)q * 4 + 1] = ad + td;
        totals[(size_t)q * 4 + 2] = ak + tk;
        totals[(size_t)q * 4 + 3] = (double)rows * (double)p.g + cnt;
    }
}

hipError_t launch_totals(const K2Params &p, const int64_t *base, const int64_t *incl_totals, double *totals, double *carry,
                         hipStream_t st) {
    if (p.n_pairs <= 0) return hipSuccess;
    dim3 block(64), grid((p.n_pairs + 63) / 64);
    hipLaunchKernelGGL(totals_kernel, grid, block, 0, st, p, base, incl_totals, totals, carry);
    return hipGetLastError();
}

// =====================================================================================
// K4: Gram path.  For windows whose membership does not depend on the pair (bp windows, fixed
// site blocks) the window sums of every pair follow from a few P x P contractions over the
// window's sites, done on the FP64 matrix cores (v_mfma_f64_16x16x4_f64):
//     sum den_ij = sum_s p_i q_j + q_i p_j                 (q = 1 - p; process_vcf_fst.h:27-30, term by term)
//     sum num_ij = sum den_ij - U_i - U_j,   U = sum_s t + h   (num = den - t_i - t_j - h_i - h_j)
//     sum k_ij   = n_j C_i + n_i C_j - 2 sum_s c_i c_j - sum_s [F_i F_j] (nm_i m_j + m_i nm_j)
// with C = sum_s c, m = n - nm the missing alleles and F = [c == nm].  A site the filter
// (process_vcf_fst.cpp:543-544) excludes contributes exactly 0 to all of them: p_i = p_j in {0, 1}
// there (populations without a called allele get p = q = 0), and the last term removes the dxy part
// of sites where both populations are fixed but have missing alleles.  The number of included sites
// and the windows the reference would poison with NaN (an included site with nm < 2 resp. nm == 0)
// come from K1's bit masks.
//
// A block of 4 waves owns a 64 x 64 block of pairs (only blocks on or above the diagonal), a wave a
// 32 x 32 sub-block = 2 x 2 MFMA tiles.  Lane l holds population (l & 15) of a tile and the 8 sites
// 8 (l >> 4) .. + 7 of every 32-site step -- the MFMA's k index is only a summation index, so each lane
// can take 8 consecutive sites = one 16-byte vector of packed counts.
// =====================================================================================
typedef double double4_t __attribute__((ext_vector_type(4)));

template <typename CT>
__global__ __launch_bounds__(256) void gram_popsums_kernel(GramParams p) {
    // one wave per (window, population): lanes stride over the window's sites, then a wave reduction.  (One block per
    // window with a wave walking every 4th population was tried: 12 ms instead of 5 on the cfg5 shard -- a chain of
    // dependent load latencies per wave.)
    const int w = blockIdx.x;
    const int k = blockIdx.y * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (k >= p.P) return;
    const int lane = lane_id();
    const int64_t a = p.win_start[p.first_window + w], b = min(p.win_start[p.first_window + w + 1], p.n_sites);
    double U = 0.0, C = 0.0;
    const CT *cb = reinterpret_cast<const CT *>(p.counts);
    for (int64_t s = a + lane; s < b; s += 64) {
        const int t = (int)(s >> 6), o = (int)(s & 63);
        if ((p.skip_mask[t] >> o) & 1ull) continue;
        const uint32_t v = cb[((size_t)t * p.P + k) * kTileSites + o];
        const uint32_t c = v & CountTraits<CT>::mask, nm = v >> CountTraits<CT>::shift;
        const double pp = nm ? (double)c / (double)nm : 0.0;
        const double tt = pp * (1.0 - pp);                     // exactly 0 at c == 0 and c == nm
        const double hh = nm > 1 ? tt / (double)(nm - 1) : 0.0;
        U += tt + hh;
        C += (double)c;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { U += __shfl_xor(U, d); C += __shfl_xor(C, d); }
    if (lane == 0) {
        p.U[(size_t)w * p.P + k] = U;
        p.C[(size_t)w * p.P + k] = C;
    }
}

// The same sums for populations of at most 31 chromosomes (16-bit counts): one workgroup per window.  Eight lanes take
// one population's 64 sites of a tile as eight 16-byte vectors (a tile's counts of all populations are one contiguous
// run, so the wave's loads are 1 KiB each), t + h comes from a 1024-entry LDS table by packed count (built once per
// workgroup: the two divisions per entry that the kernel above makes per site and population), and a group walks
// populations g, g + 32, ... with the window's two or three tiles each.  12.5 M one-wave work items of a few
// instructions each become 25 k workgroups of streaming loads.
// TB: bits of a count in the table's index (5: up to 31 chromosomes a population, 6: up to 63).
template <int TB>
__global__ __launch_bounds__(256) void gram_popsums_tab_kernel(GramParams p) {
    __shared__ double th[1 << (2 * TB)];
    for (int i = threadIdx.x; i < (1 << (2 * TB)); i += blockDim.x) {
        const int c = i & ((1 << TB) - 1), nm = i >> TB;
        double v = 0.0;
        if (nm > 0 && c <= nm) {
            const double pp = (double)c / (double)nm;
            const double tt = pp * (1.0 - pp);
            const double hh = nm > 1 ? tt / (double)(nm - 1) : 0.0;
            v = tt + hh;
        }
        th[i] = v;
    }
    __syncthreads();
    const int w = blockIdx.x;
    const int64_t a = p.win_start[p.first_window + w], b = min(p.win_start[p.first_window + w + 1], p.n_sites);
    const int group = threadIdx.x >> 3, j = threadIdx.x & 7;            // 32 groups of 8 lanes; lane j: sites 8 j .. 8 j + 7 of a tile
    const int tw0 = (int)(a >> 6), tw1 = b > a ? (int)((b - 1) >> 6) : tw0 - 1;
    const uint16_t *cb = reinterpret_cast<const uint16_t *>(p.counts);
    // KU populations of a group at a time: their count vectors of a tile are requested together (the loop waits on the
    // loads' latency: one in flight per group took 100 us per launch of 1 820 windows where four take ~40)
    constexpr int KU = 4;
    for (int kb = group; kb < p.P; kb += 32 * KU) {
        double U[KU];
        uint32_t C[KU], D[KU];                                            // D: sites where the population has a called allele
#pragma unroll
        for (int u = 0; u < KU; ++u) { U[u] = 0.0; C[u] = 0; D[u] = 0; }
        for (int t = tw0; t <= tw1; ++t) {
            uint64_t vm = ~p.skip_mask[t];                                // sites of the tile inside the window and not skipped
            if (t == tw0) vm &= ~0ull << (a & 63);
            if (t == tw1 && (b & 63)) vm &= ~0ull >> (64 - (b & 63));
            const uint32_t bits = (uint32_t)(vm >> (8 * j)) & 0xffu;
            if (__builtin_amdgcn_ballot_w64(bits != 0u) == 0) continue;
            uint4 v[KU];
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                const int k = min(kb + 32 * u, p.P - 1);
                v[u] = *reinterpret_cast<const uint4 *>(cb + ((size_t)t * p.P + k) * kTileSites + 8 * j);
            }
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                const uint32_t wv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const uint32_t x = ((bits >> e) & 1u) ? (wv[e >> 1] >> (16 * (e & 1))) & 0xffffu : 0u;   // packed count, 0 outside
                    const uint32_t c = x & 0xffu, nm = x >> 8;
                    U[u] += th[(nm << TB) | c];
                    C[u] += c;
                    D[u] += nm ? 1u : 0u;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int k = kb + 32 * u;
            double Us = U[u], Cd = (double)C[u];
            uint32_t Ds = D[u];
#pragma unroll
            for (int d = 4; d >= 1; d >>= 1) { Us += __shfl_xor(Us, d); Cd += __shfl_xor(Cd, d); Ds += (uint32_t)__shfl_xor((int)Ds, d); }
            if (j == 0 && k < p.P) { p.U[(size_t)w * p.P + k] = Us; p.C[(size_t)w * p.P + k] = Cd; p.D[(size_t)w * p.P + k] = (double)Ds; }
        }
    }
}

#ifndef EVOFST_GRAM_WPB
#define EVOFST_GRAM_WPB 8
#endif
constexpr int kGramWPB = EVOFST_GRAM_WPB;   // windows per block
constexpr int kGramMaskTiles = 4;    // tiles of a window whose masks are staged in LDS for the epilogue (longer windows read HBM)

// TAB (every population has <= 31 chromosomes): the operands of a packed count {c, nm} come from a 1024-entry LDS
// table {p, q}, {c, F nm}, [c == nm] instead of being derived with ~25 VALU instructions per element.
// I8 (8-bit counts, every population has <= 127 chromosomes): the integer contractions -- sum c_i c_j and the
// fixed-with-missing term of sumKij -- run on the int8 matrix cores (v_mfma_i32_16x16x32_i8, one instruction per 32
// sites and tile pair, exact in 32 bits), which leaves the FP64 cores the two streams of the denominator.
// ONE (with I8 and a table of its own, up to 63 chromosomes a population, short windows -- launch_gram): the denominator on ONE FP64 stream.  With
// u = p - 1/2, p_i (1 - p_j) + p_j (1 - p_i) = 1/2 - 2 u_i u_j, so sum den = n_both / 2 - 2 sum u_i u_j, n_both = the
// window's sites where both populations have a called allele (an int8 contraction of 0 / 1 bytes, exact; u = 0 where a
// population has none, as P = Q = 0 there).  Sites both populations are fixed at the same allele give u_i u_j = 1/4 exactly
// and sums of quarters are exact, so the rounding errors are those of the polymorphic sites: half an ulp of the running sum
// when the site's term is added and half an ulp of every binade the sum enters afterwards, in all at most one ulp(sites / 4)
// per such site, twice that in the denominator = sites x 2^-53, against a bar of 1e-12 of a denominator that is at least
// 1 / max_n per such site -- inside the bar for every window (<= 0.91 of it) as long as sites x max_n <= 2^13, which is
// what launch_gram asks before it takes this form (kGramOneRule).  evofst_run_gram cuts longer windows into pieces of that
// length (the kernel sees the pieces as windows) and the getter adds the pieces' sums.
constexpr int64_t kGramOneRule = 1 << 13;
typedef int int4_t __attribute__((ext_vector_type(4)));
template <typename CT, bool MISS, bool TAB, bool I8, bool ONE = false, int TB = 5>
__global__ __launch_bounds__(256, 2) void gram_kernel(GramParams p) {
    static_assert(TB == 5 || (ONE && TB == 6), "the 64 x 64 table exists for the one-stream form only");
    static_assert(!ONE || (TAB && I8), "the one-stream denominator reads the table and the int8 contractions");
    __shared__ uint64_t spread[I8 ? 256 : 1];            // bit e of the index -> byte e 0xff
    __shared__ double rcp[TAB ? 2 : 2048];
    __shared__ double2 tab_pq[TAB && !ONE ? 1024 : 1], tab_cf[TAB && !ONE ? 1024 : 1];
    __shared__ double tab_fi[TAB && !ONE ? 1024 : 1];
    __shared__ double tab_u[ONE ? (1 << (2 * TB)) : 1];  // entry (nm << TB) | c: (2 c - nm) / (2 nm), p - 1/2 rounded once; 0 without a called allele
                                                         // (TB = 6: populations of up to 63 chromosomes, 32 KB)
    __shared__ uint64_t lmask[kGramMaskTiles][4][128];   // [tile][z|f|nm<2|nm==0][64 row pops, 64 column pops]
    __shared__ double lU[ONE ? 1 : kGramWPB][128], lC[ONE ? 1 : kGramWPB][128], lN[128];   // U, C of the block's windows and n_full, same 128 populations
    __shared__ int lCi[ONE ? kGramWPB : 1][128], lDi[ONE ? kGramWPB : 1][128], lNi[ONE ? 128 : 1];   // ONE: C, n_full and the sites with a called allele
                                                         // as integers (the getter subtracts U)
    constexpr int EV = 8;                               // sites per lane and step
    constexpr int NVEC = sizeof(CT) == 2 ? 1 : 2;       // 16-byte vectors per 8 packed counts
    const int lane = lane_id(), wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (uniform, and known to be: scalar branches below)
    const int r = lane & 15, kq = lane >> 4;
    // block pair (bi <= bj) from the linear index
    const int nblk = (p.P + 63) / 64;
    int bi = 0, rem = blockIdx.y;
    while (rem >= nblk - bi) { rem -= nblk - bi; ++bi; }
    const int bj = bi + rem;
    const int r0 = bi * 64 + (wave >> 1) * 32, c0 = bj * 64 + (wave & 1) * 32;
    const bool use_rcp = p.max_n < 2048;
    if (I8)
        for (int i = threadIdx.x; i < 256; i += blockDim.x) {
            uint64_t m = 0;
            for (int e = 0; e < 8; ++e) if ((i >> e) & 1) m |= 0xffull << (8 * e);
            spread[i] = m;
        }
    if (TAB) {
        for (int i = threadIdx.x; i < (1 << (2 * TB)); i += blockDim.x) {      // entry (nm << TB) | c; impossible ones (c > nm) are 0: entry 1 serves sites outside the window
            const int c = i & ((1 << TB) - 1), nm = i >> TB;
            const bool okc = c <= nm && nm <= p.max_n;
            if (ONE) { tab_u[i] = okc && nm ? (double)(2 * c - nm) / (double)(2 * nm) : 0.0; continue; }
            const double pp = okc && nm ? (double)c / (double)nm : 0.0;
            tab_pq[i] = make_double2(pp, okc && nm ? 1.0 - pp : 0.0);
            tab_cf[i] = make_double2(okc ? (double)c : 0.0, okc && c == nm ? (double)nm : 0.0);
            tab_fi[i] = okc && c == nm ? 1.0 : 0.0;
        }
    } else if (use_rcp) {
        for (int i = threadIdx.x; i <= p.max_n; i += blockDim.x) rcp[i] = i ? 1.0 / (double)i : 0.0;
    }
    if (threadIdx.x < 128) {
        const int x = threadIdx.x, pop_ = x < 64 ? bi * 64 + x : bj * 64 + x - 64;
        lN[x] = pop_ < p.P ? (double)p.n_full[pop_] : 0.0;
        if (ONE) lNi[x] = pop_ < p.P ? p.n_full[pop_] : 0;
    }
    const bool idle = r0 > c0 + 31 || r0 >= p.P || c0 >= p.P;   // a wave below the diagonal or outside: only joins the barriers
    int pop[4];                                          // row tile 0, 1, column tile 0, 1
    pop[0] = r0 + r; pop[1] = r0 + 16 + r; pop[2] = c0 + r; pop[3] = c0 + 16 + r;
    uint32_t nfull[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) nfull[x] = pop[x] < p.P ? (uint32_t)p.n_full[pop[x]] : 0u;
    const CT *cb = reinterpret_cast<const CT *>(p.counts);
    const double kNaN = __longlong_as_double(0x7ff8000000000000ll);

    // a block works through kGramWPB consecutive windows: the tables above are built once for all of them, and what a
    // window needs before its first site -- its bounds, whether K1 announced a NaN poison in it, the per-population sums
    // for its epilogue -- is fetched here for all of them at once (one memory latency per block instead of three per
    // window, and no barrier between windows that stage no masks)
    const int w_first = blockIdx.x * kGramWPB;
    const int nw_blk = (int)min((int64_t)kGramWPB, p.n_windows - w_first);
    if (nw_blk <= 0) return;                             // (uniform; before the first barrier)
    const int64_t ws_l = p.win_start[p.first_window + w_first + min(lane, nw_blk)];   // lane l: start of the block's window l (l >= nw_blk: the end of its last)
    uint32_t need_bits = 0;                              // bit l: window l stages K1's masks
    if (I8 && MISS) {
        // the included sites of every pair come out of the int8 contraction below; K1's masks are only needed where the
        // reference would poison a window with NaN (a population with fewer than two called alleles at an included site),
        // which the two masks behind it announce -- all zero in all but pathological windows
        const int64_t al = ws_l, bl = min((int64_t)__shfl_down((long long)ws_l, 1), p.n_sites);
        bool need = false;
        if (lane < nw_blk && bl > al)
            for (int t = (int)(al >> 6); t <= (int)((bl - 1) >> 6); ++t) need = need || p.tile_poison[t] != 0u;
        need_bits = (uint32_t)__builtin_amdgcn_ballot_w64(need);
    } else if (!I8) {
        need_bits = ~0u;
    }
    for (int i = threadIdx.x; i < nw_blk * 128; i += blockDim.x) {
        const int x = i & 127, ww = i >> 7, pop_ = x < 64 ? bi * 64 + x : bj * 64 + x - 64;
        if (ONE) {
            lCi[ww][x] = pop_ < p.P ? (int)p.C[(size_t)(w_first + ww) * p.P + pop_] : 0;
            if (MISS) lDi[ww][x] = pop_ < p.P ? (int)p.D[(size_t)(w_first + ww) * p.P + pop_] : 0;
            continue;
        }
        lC[ww][x] = pop_ < p.P ? p.C[(size_t)(w_first + ww) * p.P + pop_] : 0.0;
        lU[ww][x] = pop_ < p.P ? p.U[(size_t)(w_first + ww) * p.P + pop_] : 0.0;
    }
    __syncthreads();
    auto window_start = [&](int l) {                     // l uniform, <= nw_blk
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)ws_l, l), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)ws_l >> 32), l);
        return (int64_t)(((uint64_t)hi << 32) | lo);
    };
    uint32_t popc[4];                                    // (rows of populations >= P read population P - 1)
#pragma unroll
    for (int x = 0; x < 4; ++x) popc[x] = (uint32_t)min(pop[x], p.P - 1);
    uint4 nxt[4][NVEC] = {};                             // (live across windows: see prefetched)
    uint64_t nxt_skip = 0;                               // skip mask of the tile of those sites, requested with them
    auto fetch = [&](int64_t base) {                     // the 4 populations' count vectors of this lane's 8 sites
        const int64_t sv = base + (int64_t)kq * EV;
        const int t = (int)min(sv >> 6, (int64_t)p.n_tiles - 1), o = (int)(sv & 63);
        nxt_skip = p.skip_mask[t];
        const uint32_t tp = (uint32_t)t * (uint32_t)p.P;  // (tiles x populations x 128 B is the table in HBM: far below 2^32 rows)
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const CT *src = cb + (size_t)(tp + popc[x]) * kTileSites + o;
#pragma unroll
            for (int v = 0; v < NVEC; ++v) nxt[x][v] = reinterpret_cast<const uint4 *>(src)[v];
        }
    };
    bool prefetched = false;                             // nxt holds the first step of the window (requested before the last one's epilogue)

    for (int wi = 0; wi < nw_blk; ++wi) {
    const int w = w_first + wi;
    const int64_t a = window_start(wi), b = min(window_start(wi + 1), p.n_sites);
    const int tw0 = (int)(a >> 6), tw1 = b > a ? (int)((b - 1) >> 6) : tw0 - 1;
    // K1's masks of the block's 128 populations for ng tiles from tile tg on
    // (m_first = 2: only the two masks behind the NaN poison; returns whether one of them has a bit set)
    auto stage_masks = [&](int tg, int ng, int m_first, int m_count) {
        bool any = false;
        for (int i = threadIdx.x; i < ng * m_count * 128; i += blockDim.x) {
            const int x = i & 127, mt = i >> 7;
            const int t = mt / m_count, m = m_first + (mt - t * m_count);
            const int pop = x < 64 ? bi * 64 + x : bj * 64 + x - 64;
            const uint64_t v = pop < p.P ? p.masks[((size_t)(tg + t) * 4 + m) * p.P + pop] : (m < 2 ? ~0ull : 0ull);
            lmask[t][m][x] = v;
            any = any || (m >= 2 && v != 0ull);
        }
        return any;
    };
    const bool masks_needed = (need_bits >> wi) & 1u;    // uniform over the block
    if (masks_needed) {
        __syncthreads();                                 // the windows before are done with lmask
        stage_masks(tw0, min(kGramMaskTiles, tw1 - tw0 + 1), 0, 4);
        __syncthreads();
    }

    // included sites and NaN poison of every pair from K1's masks, kGramMaskTiles tiles at a time through LDS.  The
    // first group is counted here, before the contraction (18 registers carried through it are cheaper than counting
    // with the 96 accumulator registers live); longer windows do their other groups after it.
    const int lr0 = (wave >> 1) * 32, lc0 = 64 + (wave & 1) * 32;     // this wave's rows / columns inside lmask
    uint32_t cnt_in[2][2][4];
    uint32_t poison_n = 0, poison_d = 0;                              // bit (j * 2 + i) * 4 + v
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int v = 0; v < 4; ++v) cnt_in[j][i][v] = 0;
    auto count_group = [&](int tg, int ng) {
        for (int t = 0; t < ng; ++t) {
            const int tw = tg + t;
            uint64_t wm = ~0ull;
            if (tw == tw0) wm &= ~0ull << (a & 63);
            if (tw == tw1 && (b & 63)) wm &= ~0ull >> (64 - (b & 63));
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int xj = lc0 + j * 16 + r;
                const uint64_t zj = lmask[t][0][xj], fj = lmask[t][1][xj];
                const uint64_t b1j = MISS ? lmask[t][2][xj] : 0ull, b0j = MISS ? lmask[t][3][xj] : 0ull;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int xi = lr0 + i * 16 + kq + 4 * v;
                        const uint64_t incl = ~((lmask[t][0][xi] & zj) | (lmask[t][1][xi] & fj)) & wm;
                        cnt_in[j][i][v] += (uint32_t)__popcll(incl);
                        if (MISS) {
                            const uint32_t bit = 1u << ((j * 2 + i) * 4 + v);
                            if ((lmask[t][2][xi] | b1j) & incl) poison_n |= bit;
                            if ((lmask[t][3][xi] | b0j) & incl) poison_d |= bit;
                        }
                    }
            }
        }
    };
    if (!idle && masks_needed && !(p.debug & 2)) count_group(tw0, min(kGramMaskTiles, tw1 - tw0 + 1));

    double4_t acc_den[2][2], acc_cc[2][2], acc_m[2][2];
    int4_t icc[2][2], im[2][2], iex[2][2];               // I8: the integer sums (C/D layout: col = lane & 15, row = 4 (lane >> 4) + v);
                                                         // iex: sites of the window the filter excludes for the pair
    int4_t inb[2][2];                                    // ONE, MISS: sites of the window where both populations have a called allele
    uint32_t vcount = 0;                                 // this lane's valid (in the window, not skipped) sites
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { icc[i][j] = int4_t{0, 0, 0, 0}; im[i][j] = int4_t{0, 0, 0, 0}; iex[i][j] = int4_t{0, 0, 0, 0}; inb[i][j] = int4_t{0, 0, 0, 0}; }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { acc_den[i][j] = double4_t{0, 0, 0, 0}; acc_cc[i][j] = double4_t{0, 0, 0, 0}; acc_m[i][j] = double4_t{0, 0, 0, 0}; }

    const int64_t base0 = a & ~(int64_t)7;
    const int64_t bend = (idle || (p.debug & 1)) ? base0 : b;        // an idle wave contracts nothing and writes nothing
    if (base0 < bend && !prefetched) fetch(base0);
    prefetched = false;
    for (int64_t base = base0; base < bend; base += 4 * EV) {
        const int64_t sv = base + (int64_t)kq * EV;      // this lane's 8 sites (inside one tile)
        const int o = (int)(sv & 63);
        // sites of the vector that belong to the window and are not skipped
        // (branch-free: the window's bounds relative to this lane's first site, clamped to its 8)
        const int da = (int)(a - base) - kq * EV, db = (int)(b - base) - kq * EV;
        const int lo = min(max(da, 0), EV), hi = min(max(db, 0), EV);
        uint32_t valid = hi > lo ? (((1u << hi) - 1u) & ~((1u << lo) - 1u)) : 0u;
        valid &= ~(uint32_t)(nxt_skip >> o);
        uint32_t cnt[4][EV];                             // packed counts of the 4 populations
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            if (sizeof(CT) == 2) {
                const uint32_t wv[4] = {nxt[x][0].x, nxt[x][0].y, nxt[x][0].z, nxt[x][0].w};
#pragma unroll
                for (int e = 0; e < EV; ++e) cnt[x][e] = (wv[e >> 1] >> (16 * (e & 1))) & 0xffffu;
            } else {
                const uint4 v0 = nxt[x][0], v1 = nxt[x][NVEC - 1];
                cnt[x][0] = v0.x; cnt[x][1] = v0.y; cnt[x][2] = v0.z; cnt[x][3] = v0.w;
                cnt[x][4] = v1.x; cnt[x][5] = v1.y; cnt[x][6] = v1.z; cnt[x][7] = v1.w;
            }
        }
        // the site indices some lane of the wave has inside the window, as a scalar (the 16 lanes of a group share their sites)
        const uint32_t emask = (uint32_t)(__builtin_amdgcn_readlane((int)valid, 0) | __builtin_amdgcn_readlane((int)valid, 16) |
                                          __builtin_amdgcn_readlane((int)valid, 32) | __builtin_amdgcn_readlane((int)valid, 48));
        const bool any_valid = emask != 0u;
        // ONE: the table offsets of the step's packed counts, two sites per word and no branch: entry (nm << 5) | c of tab_u
        // is at byte nm * 256 + c * 8, both in place in a 16-bit packed count but for c's shift (TB = 6: nm * 512 + c * 8,
        // at most 32 760: still 16 bits), and the sites outside the window (or skipped) go to entry 0 (= 0.0) through the
        // masks.  (Rows of populations >= P contract a copy of population P - 1: their elements of the result are never
        // read.)  The two-stream form with the {p, q} table takes its operands the same way: 16-byte entries at
        // nm * 512 + c * 16.  The offsets are formed between the int8 contractions below, one population per tile pair:
        // vector work the matrix cores' int8 instructions run beside.
        constexpr bool TPQ = TAB && I8 && !ONE && sizeof(CT) == 2;
        uint32_t T[4][4];
        if (I8 && any_valid) {
            // the step's 32 sites at once: lane = population (lane & 15), bytes = its 8 sites
            long C8[4], FN8[4], FM8[4], Z8[4], F8[4], B8[4], BN8[4], D8[4];
            vcount += (uint32_t)__popc(valid);
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const uint4 v = nxt[x][0];
                const uint64_t vm = spread[valid];                // (rows of populations >= P contract a copy of population P - 1:
                                                                  //  their elements of the result are never read)
                const uint32_t n4 = nfull[x] * 0x01010101u;
                // the operand bytes of four sites.  Counts are below 128 here, so bit 7 of 0x80 - t says "byte t is zero"
                // without a borrow between bytes, and a 0x7f mask keeps every value a fixed site contributes.
                // ONE: a site where both of a pair have no called allele (Z and F both set, B below) is counted through D
                // instead -- B = valid - D, so sum B_i B_j = n_valid - nD_i - nD_j + sum D_i D_j (the epilogue) -- and
                // without missing data there is no such site.
                uint32_t c_[2], z_[2], f_[2], b_[2], bn_[2], d_[2], fn_[2], fm_[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t m = h ? (uint32_t)(vm >> 32) : (uint32_t)vm, m80 = m & 0x80808080u;
                    const uint32_t wa = h ? v.z : v.x, wb = h ? v.w : v.y;
                    const uint32_t c = __builtin_amdgcn_perm(wb, wa, 0x06040200u) & m, n = __builtin_amdgcn_perm(wb, wa, 0x07050301u) & m;
                    const uint32_t z7 = (0x80808080u - c) & m80;          // Z = [c == 0], sites of the window only
                    const uint32_t f7 = (0x80808080u - (c ^ n)) & m80;    // F = [c == nm]
                    const uint32_t f1 = f7 >> 7, f7f = f7 - f1;
                    c_[h] = c; z_[h] = z7 >> 7; f_[h] = f1;
                    if (!ONE) {
                        const uint32_t b7 = z7 & f7, b1 = b7 >> 7;
                        b_[h] = b1; bn_[h] = b7 | (b7 - b1);              // -1 where both
                    }
                    d_[h] = ((n + 0x7f7f7f7fu) & 0x80808080u) >> 7;       // D = [nm > 0] (n is masked already)
                    fn_[h] = n & f7f;                                     // F nm
                    fm_[h] = (n4 - n) & f7f;                              // F (n - nm); bytewise, no borrows
                }
                auto pack = [](const uint32_t (&t)[2]) { return (long)(((uint64_t)t[1] << 32) | t[0]); };
                C8[x] = pack(c_);
                // the site filter (process_vcf_fst.cpp:543-544) as 0/1 bytes: excluded = z_i z_j + f_i f_j - (z f)_i (z f)_j
                Z8[x] = pack(z_); F8[x] = pack(f_);
                if (!ONE) { B8[x] = pack(b_); BN8[x] = pack(bn_); }
                if (ONE && MISS) D8[x] = pack(d_);
                if (MISS) { FN8[x] = pack(fn_); FM8[x] = pack(fm_); }
            }
            uint32_t mk[4] = {0u, 0u, 0u, 0u};
            if (ONE || TPQ) {
                const uint64_t vm = spread[valid];
                const uint32_t vlo = (uint32_t)vm, vhi = (uint32_t)(vm >> 32);
                mk[0] = __builtin_amdgcn_perm(vlo, vlo, 0x01010000u); mk[1] = __builtin_amdgcn_perm(vlo, vlo, 0x03030202u);
                mk[2] = __builtin_amdgcn_perm(vhi, vhi, 0x01010000u); mk[3] = __builtin_amdgcn_perm(vhi, vhi, 0x03030202u);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    icc[i][j] = __builtin_amdgcn_mfma_i32_16x16x32_i8(C8[i], C8[2 + j], icc[i][j], 0, 0, 0);
                    iex[i][j] = __builtin_amdgcn_mfma_i32_16x16x32_i8(Z8[i], Z8[2 + j], iex[i][j], 0, 0, 0);
                    iex[i][j] = __builtin_amdgcn_mfma_i32_16x16x32_i8(F8[i], F8[2 + j], iex[i][j], 0, 0, 0);
                    if (!ONE) iex[i][j] = __builtin_amdgcn_mfma_i32_16x16x32_i8(BN8[i], B8[2 + j], iex[i][j], 0, 0, 0);
                    if (MISS) {
                        im[i][j] = __builtin_amdgcn_mfma_i32_16x16x32_i8(FN8[i], FM8[2 + j], im[i][j], 0, 0, 0);
                        im[i][j] = __builtin_amdgcn_mfma_i32_16x16x32_i8(FM8[i], FN8[2 + j], im[i][j], 0, 0, 0);
                    }
                    if (ONE && MISS) inb[i][j] = __builtin_amdgcn_mfma_i32_16x16x32_i8(D8[i], D8[2 + j], inb[i][j], 0, 0, 0);
                    if (ONE || TPQ) {
                        const int x = i * 2 + j;
                        const uint32_t wv[4] = {nxt[x][0].x, nxt[x][0].y, nxt[x][0].z, nxt[x][0].w};
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            T[x][k] = TPQ ? ((wv[k] << 1) & (mk[k] & 0x3e003e00u)) | ((wv[k] << 4) & (mk[k] & 0x01f001f0u))       // nm * 512 + c * 16
                                    : TB == 5 ? (wv[k] & (mk[k] & 0x1f001f00u)) | ((wv[k] << 3) & (mk[k] & 0x00f800f8u))
                                              : ((wv[k] << 1) & (mk[k] & 0x7e007e00u)) | ((wv[k] << 3) & (mk[k] & 0x01f801f8u));   // nm * 512 + c * 8
                    }
                }
        }
        if (base + 4 * EV < b) fetch(base + 4 * EV);     // the next step's vectors travel while this one is contracted
        if (ONE) {
            if (!any_valid) continue;
            // the operands of site index e + 1 are read before the contractions of e are issued (the reads' round trip then
            // passes behind the matrix instructions); an index no lane of the wave has inside the window (the ragged ends of a
            // window) is read all the same and its four contractions skipped
            double Un[4];
            auto read_u = [&](int e, double (&u)[4]) {
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    const uint32_t off = (e & 1) ? T[x][e >> 1] >> 16 : T[x][e >> 1] & 0xffffu;
                    u[x] = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(tab_u) + off);
                }
            };
            read_u(0, Un);
#pragma unroll
            for (int e = 0; e < EV; ++e) {
                double U_[4] = {Un[0], Un[1], Un[2], Un[3]};
                if (e + 1 < EV) read_u(e + 1, Un);
                if (!((emask >> e) & 1u)) continue;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc_den[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(U_[i], U_[2 + j], acc_den[i][j], 0, 0, 0);   // sum u_i u_j
            }
            continue;
        }
        if (TPQ) {
            if (!any_valid) continue;
            double2 PQn[4];                               // (read one site index ahead, as in the one-contraction form)
            auto read_pq = [&](int e, double2 (&u)[4]) {
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    const uint32_t off = (e & 1) ? T[x][e >> 1] >> 16 : T[x][e >> 1] & 0xffffu;
                    u[x] = *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(tab_pq) + off);
                }
            };
            read_pq(0, PQn);
#pragma unroll
            for (int e = 0; e < EV; ++e) {
                double2 PQ[4] = {PQn[0], PQn[1], PQn[2], PQn[3]};
                if (e + 1 < EV) read_pq(e + 1, PQn);
                if (!((emask >> e) & 1u)) continue;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc_den[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(PQ[i].x, PQ[2 + j].y, acc_den[i][j], 0, 0, 0);
                        acc_den[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(PQ[i].y, PQ[2 + j].x, acc_den[i][j], 0, 0, 0);
                    }
            }
            continue;
        }
#pragma unroll
        for (int e = 0; e < EV; ++e) {
            double P_[4], Q_[4], C_[4], FN[4], FM[4];
            const bool ok = (valid >> e) & 1u;
            // a site index no lane of the wave has inside the window (the ragged ends of a window): all operands would be
            // zero, so the 12 or 20 MFMAs of this index are skipped -- exact, and ~6 % of the work with 100-site windows
            if (!((emask >> e) & 1u)) continue;
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                uint32_t c = cnt[x][e] & CountTraits<CT>::mask, nm = cnt[x][e] >> CountTraits<CT>::shift;
                if (TAB) {
                    const uint32_t idx = ok && pop[x] < p.P ? (nm << TB) | c : 1u;
                    const double2 pq = tab_pq[idx], cf = tab_cf[idx];
                    P_[x] = pq.x; Q_[x] = pq.y; C_[x] = cf.x;
                    if (MISS) { FN[x] = cf.y; FM[x] = fma((double)nfull[x], tab_fi[idx], -cf.y); }   // F (n - nm)
                    continue;
                }
                if (!ok || pop[x] >= p.P) { c = 0; nm = 0; }
                const double dc = (double)c, dn = (double)nm;
                double pp;
                if (use_rcp) {
                    const double rr = rcp[nm];
                    const double q0 = dc * rr;
                    pp = fma(fma(-q0, dn, dc), rr, q0);          // == c / nm (Markstein, exact on this domain); 0 when nm == 0
                } else {
                    pp = nm ? dc / dn : 0.0;
                }
                P_[x] = pp; Q_[x] = nm ? 1.0 - pp : 0.0; C_[x] = dc;
                if (MISS) {
                    const bool fixed = c == nm && nm > 0;
                    FN[x] = fixed ? dn : 0.0;
                    FM[x] = c == nm && ok && pop[x] < p.P ? (double)(nfull[x] - nm) : 0.0;   // nm == 0: all n alleles missing
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc_den[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(P_[i], Q_[2 + j], acc_den[i][j], 0, 0, 0);
                    acc_den[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(Q_[i], P_[2 + j], acc_den[i][j], 0, 0, 0);
                    if (I8) continue;
                    acc_cc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(C_[i], C_[2 + j], acc_cc[i][j], 0, 0, 0);
                    if (MISS) {
                        acc_m[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(FN[i], FM[2 + j], acc_m[i][j], 0, 0, 0);
                        acc_m[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(FM[i], FN[2 + j], acc_m[i][j], 0, 0, 0);
                    }
                }
        }
    }

    if (!idle && wi + 1 < nw_blk && !(p.debug & 1)) {    // the next window's first vectors travel during this one's epilogue
        const int64_t a2 = window_start(wi + 1), b2 = min(window_start(wi + 2), p.n_sites);
        if ((a2 & ~(int64_t)7) < b2) { fetch(a2 & ~(int64_t)7); prefetched = true; }
    }

    // ---- epilogue: C/D layout col = lane & 15, row = (lane >> 4) + 4 v
    if (p.debug & 2) {                                   // timing experiment: keep the accumulators alive, nothing else
        double s = 0;
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int v = 0; v < 4; ++v) s += acc_den[i][j][v] + acc_cc[i][j][v] + acc_m[i][j][v] + (double)(icc[i][j][v] + im[i][j][v] + iex[i][j][v]);
        if (s == 1.2345e-300) p.out_raw[0].x = s;
        continue;
    }
    if (masks_needed)
    for (int tg = tw0 + kGramMaskTiles; tg <= tw1; tg += kGramMaskTiles) {   // windows longer than one group of tiles
        const int ng = min(kGramMaskTiles, tw1 - tg + 1);
        __syncthreads();
        stage_masks(tg, ng, 0, 4);
        __syncthreads();
        if (!idle) count_group(tg, ng);
    }
    uint32_t nvalid = vcount;                            // over the four lanes that hold the same population: the window's valid sites
    nvalid += __shfl_xor(nvalid, 16); nvalid += __shfl_xor(nvalid, 32);
    if (idle) continue;                                  // (uniform over the wave; the barriers are at the top of the loop)
    const int lrow = (wave >> 1) * 32, lcol = 64 + (wave & 1) * 32;   // this wave's rows / columns inside lU, lC, lN
    const int nbp = nblk * (nblk + 1) / 2;
    double2 *rawA = p.out_raw + (((size_t)w * nbp + blockIdx.y) * 2 * 4 + wave) * 1024 + lane;
    double2 *rawB = rawA + 4 * 1024;
    // ONE: the same table with 8-byte entries and what is left of the arithmetic to the getter -- A = sum u_i u_j (NaN where
    // the denominator is poisoned), B = {k, n_included | n_both << 15 | num is NaN << 31} as 32-bit integers in the layout
    // their accumulators have (windows of this form have <= 2^13 sites, k <= sites x n_i n_j < 2^24); the getter forms
    // den = n_both / 2 - 2 A and num = den - U_i - U_j, the very operations done here otherwise.  Half the bytes of the
    // result stream, no exchange between the two accumulator layouts, integer arithmetic for k.
    double *rawA1 = reinterpret_cast<double *>(p.out_raw) + (((size_t)w * nbp + blockIdx.y) * 2 * 4 + wave) * 1024 + lane;
    uint2 *rawB1 = reinterpret_cast<uint2 *>(rawA1 + 4 * 1024);
    if (ONE) {
        // (a window without a poison announcement -- nearly all -- takes the copy without the NaN selects)
        auto emit = [&](auto with_poison) {
            constexpr bool PZ = decltype(with_poison)::value;
            uint32_t pn8[4] = {0u, 0u, 0u, 0u};              // poison_n of the lanes whose FP64 element is this lane's 32-bit one
            if (PZ) {
#pragma unroll
                for (int v = 0; v < 4; ++v) pn8[v] = (uint32_t)__shfl((int)poison_n, r + 16 * v);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int xj = lcol + j * 16 + r;
                const int Cj = lCi[wi][xj], nj = lNi[xj], Dj = MISS ? lDi[wi][xj] : 0;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int slot = (j * 2 + i) * 4 + v;
                        rawA1[slot * 64] = PZ && (poison_d & (1u << slot)) ? kNaN : acc_den[i][j][v];   // FP64 accumulators: row kq + 4 v
                        const int l8 = lrow + i * 16 + 4 * kq + v;                                 // 32-bit accumulators: row 4 kq + v
                        int kk = (int)__umul24((uint32_t)nj, (uint32_t)lCi[wi][l8]) + (int)__umul24((uint32_t)lNi[l8], (uint32_t)Cj) - 2 * icc[i][j][v];
                        int nb = (int)nvalid, nin = (int)nvalid - iex[i][j][v];
                        if (MISS) {
                            kk -= im[i][j][v];
                            nb = inb[i][j][v];
                            nin += (int)nvalid - lDi[wi][l8] - Dj + nb;                            // sites neither has a called allele at: in both Z and F
                        }
                        const uint32_t nan_num = PZ ? (pn8[v] >> ((j * 2 + i) * 4 + kq)) & 1u : 0u;
                        rawB1[slot * 64] = make_uint2((uint32_t)kk, (uint32_t)nin | ((uint32_t)nb << 15) | (nan_num << 31));
                    }
            }
        };
        if (MISS && masks_needed) emit(std::true_type{}); else emit(std::false_type{});
        continue;
    }
    if (!ONE && I8 && p.fmt == 2) {
        // 8-byte entries as in the one-contraction form: A = den (NaN where it is poisoned), B = {k, n_included | num is NaN << 31}
        // in the layout of the int8 accumulators; num = den - U_i - U_j is the getter's (the same subtractions in the same order)
        uint32_t pn8[4] = {0u, 0u, 0u, 0u};
        if (MISS && masks_needed) {
#pragma unroll
            for (int v = 0; v < 4; ++v) pn8[v] = (uint32_t)__shfl((int)poison_n, r + 16 * v);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const double Cj = lC[ONE ? 0 : wi][lcol + j * 16 + r], nj = lN[lcol + j * 16 + r];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int slot = (j * 2 + i) * 4 + v;
                    rawA1[slot * 64] = (poison_d & (1u << slot)) ? kNaN : acc_den[i][j][v];
                    const int l8 = lrow + i * 16 + 4 * kq + v;
                    double kk = nj * lC[ONE ? 0 : wi][l8] + lN[l8] * Cj - 2.0 * (double)icc[i][j][v];
                    if (MISS) kk -= (double)im[i][j][v];
                    const uint32_t nan_num = (pn8[v] >> ((j * 2 + i) * 4 + kq)) & 1u;
                    rawB1[slot * 64] = make_uint2((uint32_t)(long long)kk, (uint32_t)((int)nvalid - iex[i][j][v]) | (nan_num << 31));
                }
        }
        continue;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const double Uj = lU[ONE ? 0 : wi][lcol + j * 16 + r], Cj = lC[ONE ? 0 : wi][lcol + j * 16 + r], nj = lN[lcol + j * 16 + r];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int slot = (j * 2 + i) * 4 + v;
                const int li = lrow + i * 16 + kq + 4 * v;          // FP64 accumulators: row kq + 4 v
                const uint32_t bit = 1u << slot;
                const double den = acc_den[i][j][v];
                const double num = den - lU[ONE ? 0 : wi][li] - Uj;
                rawA[slot * 64] = make_double2((poison_n & bit) ? kNaN : num, (poison_d & bit) ? kNaN : den);
                if (!I8) {
                    double kk = nj * lC[ONE ? 0 : wi][li] + lN[li] * Cj - 2.0 * acc_cc[i][j][v];
                    if (MISS) kk -= acc_m[i][j][v];
                    rawB[slot * 64] = make_double2(kk, (double)cnt_in[j][i][v]);
                } else {
                    const int l8 = lrow + i * 16 + 4 * kq + v;      // 32-bit accumulators: row 4 kq + v
                    double kk = nj * lC[ONE ? 0 : wi][l8] + lN[l8] * Cj - 2.0 * (double)icc[i][j][v];
                    if (MISS) kk -= (double)im[i][j][v];
                    rawB[slot * 64] = make_double2(kk, (double)((int)nvalid - iex[i][j][v]));
                }
            }
    }
    }   // windows of this block
}

// [n_windows][block pairs][A|B][wave][slot][lane] -> the four [n_windows][n_pairs] arrays (pair (i < j) at
// i (2P - i - 1) / 2 + j - i - 1).  i8: B's rows follow the 32-bit accumulator layout (row 4 kq + v), else the FP64 one.
// one (the one-contraction denominator): 8-byte entries, A = den, B = {k, n_included | num is NaN << 31}; num = den - U_i - U_j
// as gram_kernel forms it in the other case (U: the launch's [window][P] per-population sums).
// sub_off (fmt 1 only): the table holds pieces of the windows -- window w is pieces sub_off[w] .. sub_off[w + 1] - 1, the
// table's first being piece sub_base -- each short enough for the one-contraction form; their sums are added here (evofst_run_gram).
__global__ __launch_bounds__(256) void gram_unpack_kernel(const double2 *__restrict__ raw, int P, int64_t n_windows, bool i8, int fmt,
                                                           const int64_t *__restrict__ sub_off, int64_t sub_base, const double *__restrict__ U,
                                                           double *__restrict__ num, double *__restrict__ den,
                                                           double *__restrict__ k, double *__restrict__ cnt) {
    const int pj = blockIdx.x * blockDim.x + threadIdx.x, pi = blockIdx.y;
    const int64_t w = blockIdx.z;
    if (pj >= P || pi >= pj) return;
    const int nblk = (P + 63) / 64, nbp = nblk * (nblk + 1) / 2;
    const int bi = pi >> 6, bj = pj >> 6;
    const int bp = bi * nblk - bi * (bi - 1) / 2 + (bj - bi);
    const int rho = pi & 63, chi = pj & 63;
    const int wave = (rho >> 5) * 2 + (chi >> 5);
    const int i = (rho >> 4) & 1, j = (chi >> 4) & 1, r16 = rho & 15, r = chi & 15;
    size_t base = (((size_t)w * nbp + bp) * 2 * 4 + wave) * 1024;
    const int slotA = (j * 2 + i) * 4 + (r16 >> 2), laneA = (r16 & 3) * 16 + r;             // row kq + 4 v
    const int slotB = i8 ? (j * 2 + i) * 4 + (r16 & 3) : slotA, laneB = i8 ? (r16 >> 2) * 16 + r : laneA;   // row 4 kq + v
    const size_t np = (size_t)P * (P - 1) / 2;
    const size_t o = (size_t)w * np + (size_t)pi * (2 * (size_t)P - pi - 1) / 2 + (size_t)(pj - pi - 1);
    if (fmt == 2) {                                    // A = den, B = {k, n_included | num is NaN << 31}
        const double d = reinterpret_cast<const double *>(raw)[base + slotA * 64 + laneA];
        const uint2 b = reinterpret_cast<const uint2 *>(raw)[base + 4 * 1024 + slotB * 64 + laneB];
        num[o] = (b.y >> 31) ? __longlong_as_double(0x7ff8000000000000ll) : d - U[(size_t)w * P + pi] - U[(size_t)w * P + pj];
        den[o] = d; k[o] = (double)b.x; cnt[o] = (double)(b.y & 0x7fffffffu);
        return;
    }
    if (fmt == 1 && sub_off) {
        double dn = 0.0, Ui = 0.0, Uj = 0.0, kk = 0.0;
        uint32_t cn = 0, nan = 0;
        for (int64_t s_ = sub_off[w] - sub_base; s_ < sub_off[w + 1] - sub_base; ++s_) {
            base = (((size_t)s_ * nbp + bp) * 2 * 4 + wave) * 1024;
            const double d = reinterpret_cast<const double *>(raw)[base + slotA * 64 + laneA];
            const uint2 b = reinterpret_cast<const uint2 *>(raw)[base + 4 * 1024 + slotB * 64 + laneB];
            dn += 0.5 * (double)((b.y >> 15) & 0x7fffu) - 2.0 * d;
            kk += (double)b.x; cn += b.y & 0x7fffu; nan |= b.y >> 31;
            Ui += U[(size_t)s_ * P + pi]; Uj += U[(size_t)s_ * P + pj];
        }
        num[o] = nan ? __longlong_as_double(0x7ff8000000000000ll) : dn - Ui - Uj;
        den[o] = dn; k[o] = kk; cnt[o] = (double)cn;
        return;
    }
    if (fmt == 1) {
        const double d = reinterpret_cast<const double *>(raw)[base + slotA * 64 + laneA];
        const uint2 b = reinterpret_cast<const uint2 *>(raw)[base + 4 * 1024 + slotB * 64 + laneB];
        const double dn = 0.5 * (double)((b.y >> 15) & 0x7fffu) - 2.0 * d;      // n_both / 2 - 2 sum u_i u_j
        num[o] = (b.y >> 31) ? __longlong_as_double(0x7ff8000000000000ll) : dn - U[(size_t)w * P + pi] - U[(size_t)w * P + pj];
        den[o] = dn; k[o] = (double)b.x; cnt[o] = (double)(b.y & 0x7fffu);
        return;
    }
    const double2 a = raw[base + slotA * 64 + laneA], b = raw[base + 4 * 1024 + slotB * 64 + laneB];
    num[o] = a.x; den[o] = a.y; k[o] = b.x; cnt[o] = b.y;
}

size_t gram_raw_bytes(int P, int64_t n_windows) {
    const size_t nblk = (P + 63) / 64;
    return (size_t)n_windows * (nblk * (nblk + 1) / 2) * 2 * 4 * 1024 * sizeof(double2);
}
bool gram_uses_i8(int count_bytes, int max_n) {
    bool i8 = count_bytes == 2 && max_n <= 127;
#ifdef EVOFST_EXPERIMENTS
    if (getenv("EVOFST_GRAM_NO_I8")) i8 = false;
#endif
    return i8;
}
hipError_t launch_gram_unpack(const double2 *raw, int P, int64_t n_windows, bool i8, int fmt, const int64_t *sub_off, int64_t sub_base, const double *U, double *num, double *den, double *k,
                              double *cnt, hipStream_t st) {
    if (n_windows <= 0 || P < 2) return hipSuccess;
    for (int64_t w0 = 0; w0 < n_windows; w0 += 65535) {          // gridDim.z
        const int64_t nw = std::min<int64_t>(65535, n_windows - w0);
        const size_t nblk = (P + 63) / 64, per_w = (nblk * (nblk + 1) / 2) * 2 * 4 * 1024, np = (size_t)P * (P - 1) / 2;
        dim3 grid((P + 255) / 256, P, (unsigned)nw), block(256);
        // (with 8-byte entries a window's part of the table is half as long)
        const double2 *rw = fmt ? reinterpret_cast<const double2 *>(reinterpret_cast<const double *>(raw) + (size_t)w0 * per_w) : raw + (size_t)w0 * per_w;
        if (sub_off)      // (pieces are counted from the table's first: neither the table nor U moves with w0)
            hipLaunchKernelGGL(gram_unpack_kernel, grid, block, 0, st, raw, P, nw, i8, fmt, sub_off + w0, sub_base, U, num + (size_t)w0 * np,
                               den + (size_t)w0 * np, k + (size_t)w0 * np, cnt + (size_t)w0 * np);
        else
        hipLaunchKernelGGL(gram_unpack_kernel, grid, block, 0, st, rw, P, nw, i8, fmt, (const int64_t *)nullptr, (int64_t)0, U + (size_t)w0 * P, num + (size_t)w0 * np,
                           den + (size_t)w0 * np, k + (size_t)w0 * np, cnt + (size_t)w0 * np);
    }
    return hipGetLastError();
}

int64_t gram_one_rule_sites(int count_bytes, int max_n) {
    return gram_uses_i8(count_bytes, max_n) && count_bytes == 2 && max_n >= 1 && max_n <= 63 ? kGramOneRule / max_n : 0;
}
// the denominator on one FP64 stream where every window stays inside the 1e-12 bar by construction (see gram_kernel)
bool gram_one_stream(const GramParams &p, int count_bytes) {
    return !p.two_streams && gram_uses_i8(count_bytes, p.max_n) && count_bytes == 2 && p.max_n <= 63 && p.max_window_sites > 0 &&
           p.max_window_sites * (int64_t)p.max_n <= kGramOneRule;
}
// the windows' per-population sums (U, C, D of GramParams), which the contraction's epilogue and the getter read
// The entries of the result table a launch writes (GramParams::fmt): the one-contraction form has its own 8-byte entries;
// the other forms with int8 contractions write 8-byte entries too where k = sumKij <= sites x n_i n_j and the site count
// fit 32 and 31 bits, and leave num = den - U_i - U_j to the getter; 16-byte {num, den} / {k, n_included} otherwise.
int gram_result_format(const GramParams &p, int count_bytes) {
    if (gram_one_stream(p, count_bytes)) return 1;
    if (gram_uses_i8(count_bytes, p.max_n) && p.max_window_sites > 0 &&
        (double)p.max_window_sites * (double)p.max_n * (double)p.max_n < 4.0e9 && p.max_window_sites < (1ll << 30)) return 2;
    return 0;
}
hipError_t launch_gram_popsums(const GramParams &p, int count_bytes, hipStream_t st) {
    if (p.n_windows <= 0 || p.P < 2 || (p.debug & 4)) return hipSuccess;
    dim3 grid(p.n_windows, (p.P + 3) / 4), block(256);
    if (count_bytes == 2 && p.max_n <= 31) hipLaunchKernelGGL(gram_popsums_tab_kernel<5>, dim3(p.n_windows), block, 0, st, p);
    else if (count_bytes == 2 && p.max_n <= 63) hipLaunchKernelGGL(gram_popsums_tab_kernel<6>, dim3(p.n_windows), block, 0, st, p);
    else if (count_bytes == 2) hipLaunchKernelGGL((gram_popsums_kernel<uint16_t>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((gram_popsums_kernel<uint32_t>), grid, block, 0, st, p);
    return hipGetLastError();
}
hipError_t launch_gram(const GramParams &p, int count_bytes, bool has_missing, hipStream_t st) {
    if (p.n_windows <= 0 || p.P < 2) return hipSuccess;
    const int nblk = (p.P + 63) / 64;
    dim3 grid((p.n_windows + kGramWPB - 1) / kGramWPB, nblk * (nblk + 1) / 2), block(256);
    const bool i8 = gram_uses_i8(count_bytes, p.max_n);
#define EVOFST_GRAM_GO(CT, M, T) do { if (i8) hipLaunchKernelGGL((gram_kernel<CT, M, T, sizeof(CT) == 2>), grid, block, 0, st, p); \
                                      else hipLaunchKernelGGL((gram_kernel<CT, M, T, false>), grid, block, 0, st, p); } while (0)
    const bool one = gram_one_stream(p, count_bytes);
    if (count_bytes == 2) {
        if (one && p.max_n <= 31) {
            if (has_missing) hipLaunchKernelGGL((gram_kernel<uint16_t, true, true, true, true>), grid, block, 0, st, p);
            else hipLaunchKernelGGL((gram_kernel<uint16_t, false, true, true, true>), grid, block, 0, st, p);
        } else if (one) {                            // 32 .. 63 chromosomes: the 64 x 64 table
            if (has_missing) hipLaunchKernelGGL((gram_kernel<uint16_t, true, true, true, true, 6>), grid, block, 0, st, p);
            else hipLaunchKernelGGL((gram_kernel<uint16_t, false, true, true, true, 6>), grid, block, 0, st, p);
        } else if (p.max_n <= 31) {
            if (has_missing) EVOFST_GRAM_GO(uint16_t, true, true);
            else EVOFST_GRAM_GO(uint16_t, false, true);
        } else {
            if (has_missing) EVOFST_GRAM_GO(uint16_t, true, false);
            else EVOFST_GRAM_GO(uint16_t, false, false);
        }
    } else {
        if (has_missing) hipLaunchKernelGGL((gram_kernel<uint32_t, true, false, false>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((gram_kernel<uint32_t, false, false, false>), grid, block, 0, st, p);
    }
#undef EVOFST_GRAM_GO
    return hipGetLastError();
}

// =====================================================================================
// Per-site outputs on demand (lane = site).  These are the parity hooks and what the CLI front-end
// feeds to its replay of the reference's window state machines.
// =====================================================================================
template <typename CT>
__global__ __launch_bounds__(256) void site_stats_kernel(SiteStatParams p) {
    const int64_t ls = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // index in the request
    if (ls >= p.n_sites) return;
    const int64_t site = p.first_site + ls;
    const int64_t t = site >> 6;
    const int s = (int)(site & 63);
    const CT *cp = reinterpret_cast<const CT *>(p.counts) + (size_t)t * p.P * kTileSites + s;
    bool skip = p.site_flags ? ((p.site_flags[site] & 1u) != 0) : false;
    for (int q = blockIdx.y; q < p.n_pairs; q += gridDim.y) {
        const uint32_t ij = p.pairs[q];
        const uint32_t i = ij & 0xffffu, j = ij >> 16;
        const uint32_t vi = cp[(size_t)i * kTileSites], vj = cp[(size_t)j * kTileSites];
        const int c1 = (int)(vi & CountTraits<CT>::mask), nm1 = (int)(vi >> CountTraits<CT>::shift);
        const int c2 = (int)(vj & CountTraits<CT>::mask), nm2 = (int)(vj >> CountTraits<CT>::shift);
        const int n1 = p.n_full[i], n2 = p.n_full[j];
        const double p1 = (double)c1 / (double)nm1, p2 = (double)c2 / (double)nm2;
        const double power = (p1 - p2) * (p1 - p2);
        const double f1 = (p1 * (1.0 - p1)) / (double)(nm1 - 1);
        const double f2 = (p2 * (1.0 - p2)) / (double)(nm2 - 1);
        const double num = power - f1 - f2;
        const double den = (p1 * (1.0 - p2)) + (p2 * (1.0 - p1));
        const int sumKij = c1 * (n2 - c2) + c2 * (n1 - c1);
        const double dxy = (double)sumKij / (double)(n1 * n2);
        const bool inc = !skip && (c1 > 0 || c2 > 0) && (c1 < nm1 || c2 < nm2);
        const size_t o = (size_t)q * p.n_sites + ls;
        if (p.num) p.num[o] = num;
        if (p.den) p.den[o] = den;
        if (p.dxy) p.dxy[o] = dxy;
        if (p.include) p.include[o] = inc ? 1 : 0;
    }
}

// Population branch statistic per site and trio (calculatePBSfromAFs, evo_PBS.cpp:69-98, and the per-site rules of
// PBSmain, :304-322): lane = site, blockIdx.y walks the trios.  AAF = -1 for a population without a called allele.
template <typename CT>
__global__ __launch_bounds__(256) void pbs_site_kernel(PbsParams p) {
    const int64_t ls = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ls >= p.n_sites) return;
    const int64_t site = p.first_site + ls;
    const CT *cp = reinterpret_cast<const CT *>(p.counts) + (size_t)(site >> 6) * p.P * kTileSites + (int)(site & 63);
    const bool skip = p.site_flags ? ((p.site_flags[site] & 1u) != 0) : false;
    for (int q = blockIdx.y; q < p.n_trios; q += gridDim.y) {
        double pf[3], nf[3];
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            const uint32_t v = cp[(size_t)p.trios[3 * q + x] * kTileSites];
            const int c = (int)(v & CountTraits<CT>::mask), nm = (int)(v >> CountTraits<CT>::shift);
            pf[x] = nm > 0 ? (double)c / (double)nm : -1.0;                 // utils.cpp:56-58, utils.h:75
            nf[x] = (double)nm;
        }
        const double p1 = pf[0], p2 = pf[1], p3 = pf[2];
        bool used = !skip && p1 != -1.0 && p2 != -1.0 && p3 != -1.0;        // PBS.cpp:304-309
        if (used && ((p1 == 0 && p2 == 0 && p3 == 0) || (p1 == 1 && p2 == 1 && p3 == 1))) used = false;   // :311-312
        double out[3] = {0.0, 0.0, 0.0};
        if (used) {
            const double power12 = (p1 - p2) * (p1 - p2), power13 = (p1 - p3) * (p1 - p3), power23 = (p2 - p3) * (p2 - p3);
            const double fraction1 = (p1 * (1 - p1)) / (nf[0] - 1), fraction2 = (p2 * (1 - p2)) / (nf[1] - 1), fraction3 = (p3 * (1 - p3)) / (nf[2] - 1);
            const double numerator12 = power12 - fraction1 - fraction2, numerator13 = power13 - fraction1 - fraction3, numerator23 = power23 - fraction2 - fraction3;
            const double denominator12 = (p1 * (1 - p2)) + (p2 * (1 - p1)), denominator13 = (p1 * (1 - p3)) + (p3 * (1 - p1)), denominator23 = (p2 * (1 - p3)) + (p3 * (1 - p2));
            double Fst12, Fst13, Fst23;
            if ((p1 == 0 && p2 == 0) || (p1 == 1 && p2 == 1)) Fst12 = 0.0; else Fst12 = numerator12 / denominator12;
            if ((p1 == 0 && p3 == 0) || (p1 == 1 && p3 == 1)) Fst13 = 0.0; else Fst13 = numerator13 / denominator13;
            if ((p2 == 0 && p3 == 0) || (p2 == 1 && p3 == 1)) Fst23 = 0.0; else Fst23 = numerator23 / denominator23;
            if (Fst12 < 0) Fst12 = 0; if (Fst13 < 0) Fst13 = 0; if (Fst23 < 0) Fst23 = 0;
            if (Fst12 == 1) Fst12 = 1 - (Fst12 / nf[0]); if (Fst13 == 1) Fst13 = 1 - (Fst13 / nf[0]); if (Fst23 == 1) Fst23 = 1 - (Fst23 / nf[1]);   // :87 (sic)
            const double T12 = -log(1 - Fst12), T13 = -log(1 - Fst13), T23 = -log(1 - Fst23);
            out[0] = (T12 + T13 - T23) / 2.0; out[1] = (T12 + T23 - T13) / 2.0; out[2] = (T13 + T23 - T12) / 2.0;
#pragma unroll
            for (int x = 0; x < 3; ++x) if (out[x] < 0) out[x] = 0;
        }
#pragma unroll
        for (int x = 0; x < 3; ++x) p.pbs[((size_t)q * 3 + x) * p.n_sites + ls] = out[x];
        p.used[(size_t)q * p.n_sites + ls] = used ? 1 : 0;
    }
}

// Hudson Fst of one population pair as calculateABSfromAFs forms it (evo_ABS.cpp:63-91); fa, fb = the two variance
// terms the reference subtracts for this pair
__device__ __forceinline__ double abs_pair_fst(double pa, double pb, double fa, double fb, double na, double nb) {
    const double power = (pa - pb) * (pa - pb);
    const double numerator = power - fa - fb;
    const double denominator = (pa * (1 - pb)) + (pb * (1 - pa));
    double Fst;
    if ((pa == 0 && pb == 0) || (pa == 1 && pb == 1)) Fst = 0.0; else Fst = numerator / denominator;
    if (Fst < 0) Fst = 0;
    if (Fst == 1) Fst = 1 - (Fst / ((na + nb) / 2.0));
    return Fst;
}

// Ancestral branch statistic per site and quartet W, X, Y, Z (calculateABSfromAFs, evo_ABS.cpp:58-106, and the
// per-site rules of ABSmain, :243-262): lane = site, blockIdx.y walks the quartets; out[q][0] = ABS, out[q][1] = ABSalt.
// PbsParams is reused: trios = [n][4] population indices, pbs = [n][2][n_sites].
template <typename CT>
__global__ __launch_bounds__(256) void abs_site_kernel(PbsParams p) {
    const int64_t ls = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ls >= p.n_sites) return;
    const int64_t site = p.first_site + ls;
    const CT *cp = reinterpret_cast<const CT *>(p.counts) + (size_t)(site >> 6) * p.P * kTileSites + (int)(site & 63);
    const bool skip = p.site_flags ? ((p.site_flags[site] & 1u) != 0) : false;
    for (int q = blockIdx.y; q < p.n_trios; q += gridDim.y) {
        double pf[4], nf[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const uint32_t v = cp[(size_t)p.trios[4 * q + x] * kTileSites];
            const int c = (int)(v & CountTraits<CT>::mask), nm = (int)(v >> CountTraits<CT>::shift);
            pf[x] = nm > 0 ? (double)c / (double)nm : -1.0;
            nf[x] = (double)nm;
        }
        const double pW = pf[0], pX = pf[1], pY = pf[2], pZ = pf[3];
        bool used = !skip && pW != -1.0 && pX != -1.0 && pY != -1.0 && pZ != -1.0;                        // ABS.cpp:245-252
        if (used && ((pW == 0 && pX == 0 && pY == 0 && pZ == 0) || (pW == 1 && pX == 1 && pY == 1 && pZ == 1))) used = false;   // :254-255
        double ABSmain = 0.0, ABSalt = 0.0;
        if (used) {
            // :66 (sic) the variable named fractionX holds W's term and fractionW holds X's
            const double fractionX = (pW * (1 - pW)) / (nf[0] - 1), fractionW = (pX * (1 - pX)) / (nf[1] - 1);
            const double fractionY = (pY * (1 - pY)) / (nf[2] - 1), fractionZ = (pZ * (1 - pZ)) / (nf[3] - 1);
            const double TWX = -log(1 - abs_pair_fst(pW, pX, fractionW, fractionX, nf[0], nf[1]));
            const double TWY = -log(1 - abs_pair_fst(pW, pY, fractionW, fractionY, nf[0], nf[2]));
            const double TWZ = -log(1 - abs_pair_fst(pW, pZ, fractionW, fractionZ, nf[0], nf[3]));
            const double TXY = -log(1 - abs_pair_fst(pX, pY, fractionX, fractionY, nf[1], nf[2]));
            const double TXZ = -log(1 - abs_pair_fst(pX, pZ, fractionX, fractionZ, nf[1], nf[3]));
            const double TYZ = -log(1 - abs_pair_fst(pY, pZ, fractionY, fractionZ, nf[2], nf[3]));
            const double TWYTXZ = TWY + TXZ, TWZTXY = TWZ + TXY, TWXTYZ = TWX + TYZ;
            const double twoMax = (TWYTXZ >= TWZTXY) ? TWYTXZ : TWZTXY;                                      // :95
            ABSalt = (twoMax - TWXTYZ) / 2.0;
            double threeMax, threeMin;                                                                     // :97-100
            if (TWYTXZ >= TWZTXY && TWYTXZ >= TWXTYZ) { threeMax = TWYTXZ; threeMin = (TWZTXY <= TWXTYZ) ? TWZTXY : TWXTYZ; }
            else if (TWZTXY >= TWYTXZ && TWZTXY >= TWXTYZ) { threeMax = TWZTXY; threeMin = (TWYTXZ <= TWXTYZ) ? TWYTXZ : TWXTYZ; }
            else { threeMax = TWXTYZ; threeMin = (TWYTXZ <= TWZTXY) ? TWYTXZ : TWZTXY; }
            ABSmain = (threeMax - threeMin) / 2.0;
        }
        p.pbs[((size_t)q * 2 + 0) * p.n_sites + ls] = ABSmain;
        p.pbs[((size_t)q * 2 + 1) * p.n_sites + ls] = ABSalt;
        p.used[(size_t)q * p.n_sites + ls] = used ? 1 : 0;
    }
}

hipError_t launch_abs_sites(const PbsParams &p, int count_bytes, hipStream_t st) {
    if (p.n_sites <= 0 || p.n_trios <= 0) return hipSuccess;
    dim3 block(256), grid((unsigned)((p.n_sites + 255) / 256), (unsigned)min(p.n_trios, 1024));
    if (count_bytes == 2) hipLaunchKernelGGL((abs_site_kernel<uint16_t>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((abs_site_kernel<uint32_t>), grid, block, 0, st, p);
    return hipGetLastError();
}

hipError_t launch_pbs_sites(const PbsParams &p, int count_bytes, hipStream_t st) {
    if (p.n_sites <= 0 || p.n_trios <= 0) return hipSuccess;
    dim3 block(256), grid((unsigned)((p.n_sites + 255) / 256), (unsigned)min(p.n_trios, 1024));
    if (count_bytes == 2) hipLaunchKernelGGL((pbs_site_kernel<uint16_t>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((pbs_site_kernel<uint32_t>), grid, block, 0, st, p);
    return hipGetLastError();
}

// Outgroup-to-ingroup dxy per caller-cut window (DistOutMain, evo_distanceToOutgroups.cpp:274-294, DxyPerSNPfromAFs :62-65).
// One lane per (window, outgroup, ingroup): the lane walks its window's sites in order, so its sum is accumulated
// exactly in the order of the reference's vector_average_withRegion -- bit-identical, not just close.  Four sites are
// in flight per lane.  The lane of ingroup 0 also writes the outgroup's own counters.  carry_first: window 0
// continues the running values already in the arrays.  (The first version gave a whole wave to one window and
// outgroup, with as many lanes busy as there were ingroups.)
template <typename CT>
__global__ __launch_bounds__(256) void dist_window_kernel(DistParams p) {
    const int nin = p.n_in > 0 ? p.n_in : 1;              // without ingroups one lane per (window, outgroup) still counts
    const int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= (int64_t)p.n_windows * p.n_out * nin) return;
    const int w = (int)(item / ((int64_t)p.n_out * nin));
    const int rem = (int)(item - (int64_t)w * p.n_out * nin);
    const int i = rem / nin, j = rem - i * nin;
    const bool act = j < p.n_in;
    const int64_t s0 = p.win_off[w], s1 = p.win_off[w + 1];
    const CT *counts = reinterpret_cast<const CT *>(p.counts);
    const int ko = p.out_idx[i];
    const int kj = act ? p.in_idx[j] : ko;
    const bool carry = p.carry_first && w == 0;
    const size_t wi = (size_t)w * p.n_out + i;
    const size_t o = wi * p.n_in + (act ? j : 0);
    double sum = (carry && act) ? p.sum[o] : 0.0;
    int nsnp = (carry && act) ? p.nsnp[o] : 0, miss = (carry && act) ? p.miss[o] : 0;
    int used = carry ? p.used[wi] : 0, missing = carry ? p.missing[wi] : 0;
    for (int64_t s = s0; s < s1; s += 4) {
        uint32_t vo[4], vj[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t su = s + u;
            const int64_t sc = su < s1 ? su : s1 - 1;
            ok[u] = su < s1 && !(p.site_flags && (p.site_flags[sc] & 1u));
            const CT *cp = counts + (size_t)(sc >> 6) * p.P * kTileSites + (int)(sc & 63);
            vo[u] = cp[(size_t)ko * kTileSites];
            vj[u] = cp[(size_t)kj * kTileSites];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!ok[u]) continue;
            const int co = (int)(vo[u] & CountTraits<CT>::mask), mo = (int)(vo[u] >> CountTraits<CT>::shift);
            if (mo == 0) { ++missing; continue; }                       // AFout == -1 (:278-282)
            ++used;
            const int cj = (int)(vj[u] & CountTraits<CT>::mask), mj = (int)(vj[u] >> CountTraits<CT>::shift);
            if (mj != 0) {
                const double AFout = (double)co / (double)mo, AFin = (double)cj / (double)mj;
                sum += AFout * (1 - AFin) + AFin * (1 - AFout);
                ++nsnp;
            } else ++miss;
        }
    }
    if (act) { p.sum[o] = sum; p.nsnp[o] = nsnp; p.miss[o] = miss; }
    if (j == 0) { p.used[wi] = used; p.missing[wi] = missing; }
}

hipError_t launch_dist_windows(const DistParams &p, int count_bytes, hipStream_t st) {
    if (p.n_windows <= 0 || p.n_out <= 0) return hipSuccess;
    const int64_t items = (int64_t)p.n_windows * p.n_out * (p.n_in > 0 ? p.n_in : 1);
    dim3 block(256), grid((unsigned)((items + 255) / 256));
    if (count_bytes == 2) hipLaunchKernelGGL((dist_window_kernel<uint16_t>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((dist_window_kernel<uint32_t>), grid, block, 0, st, p);
    return hipGetLastError();
}

// =====================================================================================
// Per-individual pairwise differences (diffs_between_individuals[_with_multialleleics],
// process_vcf_stats_functions.cpp:180-281).  Three kernels:
//   indiv_planes  tiles (site-major: one word = 64 haplotypes of a site) -> per individual and tile four 64-bit planes
//                 over the tile's sites {heterozygous, hom-alt, hom-ref, missing}; a 64 x 64 bit transpose by ballots
//   indiv_pairs   thread = pair j <= i, walks the tiles of a site block: the reference's if-chain becomes ANDs and
//                 popcounts of the planes, 64 sites at a time
//   indiv_multi   thread = pair, walks the (rare) multi-allelic sites of the block on their allele codes
// =====================================================================================
__global__ __launch_bounds__(256) void indiv_planes_kernel(IndivParams p) {
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);   // (tile, word)
    if (item >= (int64_t)p.n_tiles * p.W) return;
    const int t = (int)(item / p.W), w = (int)(item % p.W);
    const int64_t site = (int64_t)t * kTileSites + lane;
    const bool valid = site < p.n_sites && !(p.site_flags && (p.site_flags[site < p.n_sites ? site : 0] & 1u));
    ulonglong2 v = p.tiles[((size_t)t * p.W + w) * kTileSites + lane];
    if (!valid) v = make_ulonglong2(0, 0);
    const uint64_t incl = __builtin_amdgcn_ballot_w64(valid);
    // lane b of the result holds, for haplotype 64 w + b, one bit per site of the tile
    uint64_t ta = 0, tm = 0;
    const bool any_missing = __builtin_amdgcn_ballot_w64(v.y != 0) != 0;
#pragma unroll 8
    for (int b = 0; b < 64; ++b) {
        const uint64_t m = __builtin_amdgcn_ballot_w64((v.x >> b) & 1ull);
        if (lane == b) ta = m;
    }
    if (any_missing) {
#pragma unroll 8
        for (int b = 0; b < 64; ++b) {
            const uint64_t m = __builtin_amdgcn_ballot_w64((v.y >> b) & 1ull);
            if (lane == b) tm = m;
        }
    }
    const uint64_t pa = __shfl_xor(ta, 1), pm = __shfl_xor(tm, 1);   // the individual's other haplotype
    const int d = (w * 64 + lane) >> 1;
    if ((lane & 1) == 0 && d < p.n_ind) {
        const uint64_t miss = (tm | pm) & incl;
        const uint64_t het = (ta ^ pa) & ~miss & incl, alt = ta & pa & ~miss & incl;
        const uint64_t ref = incl & ~(het | alt | miss);
        ulonglong2 *o = reinterpret_cast<ulonglong2 *>(p.planes + ((size_t)t * p.n_ind + d) * 4);
        o[0] = make_ulonglong2(het, alt);
        o[1] = make_ulonglong2(ref, miss);
    }
}

// pair block pb -> (bi, bj), bj <= bi, in row-major order of the lower triangle
__device__ __forceinline__ void pair_block_of(int pb, int &bi, int &bj) {
    int r = (int)((sqrtf(8.0f * (float)pb + 1.0f) - 1.0f) * 0.5f);
    while ((r + 1) * (r + 2) / 2 <= pb) ++r;
    while (r * (r + 1) / 2 > pb) --r;
    bi = r; bj = pb - r * (r + 1) / 2;
}

constexpr int kIndivStage = 16;   // tiles staged in LDS at a time by the pair kernel

__global__ __launch_bounds__(256) void indiv_pair_kernel(IndivParams p) {
    // planes of the block's 16 row individuals and 16 column individuals for kIndivStage tiles: every plane is read
    // by 16 threads, so it comes from global memory once per block and from LDS after that
    __shared__ ulonglong2 sp[kIndivStage][2][16][2];       // [tile][rows | columns][individual][{het, alt} | {ref, miss}]
    int bi, bj;
    pair_block_of((int)blockIdx.x, bi, bj);
    const int ty = (int)(threadIdx.x >> 4), tx = (int)(threadIdx.x & 15);
    const int i = bi * 16 + ty, j = bj * 16 + tx;
    const int b = blockIdx.y;
    const int64_t s0 = p.block_off[b], s1 = p.block_off[b + 1];
    const bool act = i < p.n_ind && j <= i;
    int hh = 0, hd = 0, oh = 0, ms = 0;
    if (s1 > s0) {
        const int t0 = (int)(s0 >> 6), t1 = (int)((s1 - 1) >> 6);
        for (int tb = t0; tb <= t1; tb += kIndivStage) {
            const int nt = min(kIndivStage, t1 - tb + 1);
            if (tb != t0) __syncthreads();
            // nt tiles x 32 individuals x 2 vectors of 16 bytes
            for (int x = threadIdx.x; x < nt * 64; x += 256) {
                const int t = x >> 6, side = (x >> 5) & 1, ind = (x >> 1) & 15, half = x & 1;
                const int d = (side ? bj : bi) * 16 + ind;
                sp[t][side][ind][half] = d < p.n_ind
                    ? reinterpret_cast<const ulonglong2 *>(p.planes + ((size_t)(tb + t) * p.n_ind + d) * 4)[half]
                    : make_ulonglong2(0, 0);
            }
            __syncthreads();
            if (act)
                for (int t = 0; t < nt; ++t) {
                    // the sites of this tile that belong to the block
                    const int64_t base = (int64_t)(tb + t) << 6;
                    const int lo = (int)(s0 > base ? s0 - base : 0), hi = (int)(s1 - base < 64 ? s1 - base : 64);
                    const uint64_t tmask = (hi == 64 ? ~0ull : ((1ull << hi) - 1ull)) & ~((1ull << lo) - 1ull);
                    const ulonglong2 i0 = sp[t][0][ty][0], i1 = sp[t][0][ty][1], j0 = sp[t][1][tx][0], j1 = sp[t][1][tx][1];
                    const uint64_t Hi = i0.x & tmask, Ai = i0.y & tmask, Ri = i1.x & tmask, Mi = i1.y;
                    const uint64_t Hj = j0.x, Aj = j0.y, Rj = j1.x, Mj = j1.y;
                    hh += __popcll(Hi & Hj);                                         // :201 both heterozygous (j == i: heterozygous, :222)
                    hd += __popcll((Ai & Rj) | (Ri & Aj));                           // :205 opposite homozygotes
                    oh += __popcll((Hi & (Rj | Aj)) | ((Ri | Ai) & Hj));             // :211 one homozygous, one heterozygous
                    ms += __popcll((Mi | Mj) & tmask);                               // :186-197 either one missing
                }
        }
    }
    if (!act) return;
    const size_t o = (size_t)b * p.T + (size_t)i * (i + 1) / 2 + j;
    p.hh[o] = hh; p.hd[o] = hd; p.oh[o] = oh; p.ms[o] = ms;
}

// multi-allelic sites (:235-281): adds its missing pairs to ms (the pair kernel has already written it), writes d4
__global__ __launch_bounds__(256) void indiv_multi_kernel(IndivParams p) {
    int bi, bj;
    pair_block_of((int)blockIdx.x, bi, bj);
    const int i = bi * 16 + (int)(threadIdx.x >> 4), j = bj * 16 + (int)(threadIdx.x & 15);
    const int b = blockIdx.y;
    if (i >= p.n_ind || j > i) return;
    const int64_t s0 = p.block_off[b], s1 = p.block_off[b + 1];
    // first multi-allelic site at or after s0
    int64_t lo = 0, hi = p.n_multi;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (p.multi_site[mid] < s0) lo = mid + 1; else hi = mid; }
    int d4 = 0, ms = 0;
    for (int64_t k = lo; k < p.n_multi && p.multi_site[k] < s1; ++k) {
        const uint8_t *a = p.multi_alleles + ((size_t)k * p.n_ind) * 2;
        const int i1 = a[2 * i], i2 = a[2 * i + 1], j1 = a[2 * j], j2 = a[2 * j + 1];
        if (i1 == 255 || j1 == 255) { ++ms; continue; }
        if (j < i) d4 += (i1 != j1) + (i1 != j2) + (i2 != j1) + (i2 != j2);  // :257-260
        else d4 += (i1 != i2);                                              // :272
    }
    const size_t o = (size_t)b * p.T + (size_t)i * (i + 1) / 2 + j;
    p.d4[o] = d4;
    if (ms) p.ms[o] += ms;
}

// het_analysis (process_vcf_stats_functions.cpp:19-28) at the sites where no genotype is missing
// (process_vcf_stats.cpp:239-242): per individual its heterozygous sites, and those of them where the ALT allele was
// seen more than once among all individuals (counts.overall > 1: some individual is hom-alt, or two are heterozygous).
// One wave per tile; lanes take the individuals in turn.  "Seen in two individuals" is kept as a pair of masks
// {one, two} that combines over disjoint groups: two = two_a | two_b | (one_a & one_b).  Multi-allelic and skipped
// sites have empty planes (indiv_planes_kernel) and count nowhere, as in the reference (individualsWithVariant stays 0).
__global__ __launch_bounds__(256) void indiv_hets_kernel(IndivParams p) {
    const int lane = threadIdx.x & 63;
    const int t = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (t >= p.n_tiles) return;                                  // wave-uniform
    const uint64_t *pl = p.planes + (size_t)t * p.n_ind * 4;
    uint64_t any_miss = 0, any_alt = 0, one = 0, two = 0;
    for (int d = lane; d < p.n_ind; d += 64) {
        const ulonglong2 ha = reinterpret_cast<const ulonglong2 *>(pl + (size_t)d * 4)[0];   // {het, hom-alt}
        any_miss |= pl[(size_t)d * 4 + 3];
        any_alt |= ha.y;
        two |= one & ha.x;
        one |= ha.x;
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const uint64_t o1 = __shfl_xor(one, m), o2 = __shfl_xor(two, m);
        two |= o2 | (one & o1);
        one |= o1;
        any_miss |= __shfl_xor(any_miss, m);
        any_alt |= __shfl_xor(any_alt, m);
    }
    const uint64_t ok = ~any_miss, shared = any_alt | two;
    for (int d = lane; d < p.n_ind; d += 64) {
        const uint64_t h = pl[(size_t)d * 4] & ok;
        if (h) {
            atomicAdd(p.het_cnt + d, (int32_t)__popcll(h));
            const int sh = (int)__popcll(h & shared);
            if (sh) atomicAdd(p.shared_cnt + d, (int32_t)sh);
        }
    }
}

hipError_t launch_indiv_hets(const IndivParams &p, hipStream_t st) {
    if (p.n_tiles <= 0 || p.n_ind <= 0) return hipSuccess;
    hipLaunchKernelGGL(indiv_hets_kernel, dim3((unsigned)((p.n_tiles + 3) / 4)), dim3(256), 0, st, p);
    return hipGetLastError();
}

hipError_t launch_indiv_planes(const IndivParams &p, hipStream_t st) {
    const int64_t items = (int64_t)p.n_tiles * p.W;
    if (items <= 0) return hipSuccess;
    hipLaunchKernelGGL(indiv_planes_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, p);
    return hipGetLastError();
}
hipError_t launch_indiv_pairs(const IndivParams &p, hipStream_t st) {
    if (p.n_blocks <= 0 || p.n_pair_blocks <= 0) return hipSuccess;
    hipLaunchKernelGGL(indiv_pair_kernel, dim3((unsigned)p.n_pair_blocks, (unsigned)p.n_blocks), dim3(256), 0, st, p);
    return hipGetLastError();
}
hipError_t launch_indiv_multi(const IndivParams &p, hipStream_t st) {
    if (p.n_blocks <= 0 || p.n_pair_blocks <= 0) return hipSuccess;
    hipLaunchKernelGGL(indiv_multi_kernel, dim3((unsigned)p.n_pair_blocks, (unsigned)p.n_blocks), dim3(256), 0, st, p);
    return hipGetLastError();
}

hipError_t launch_site_stats(const SiteStatParams &p, int count_bytes, hipStream_t st) {
    if (p.n_sites <= 0 || p.n_pairs <= 0) return hipSuccess;
    dim3 block(256), grid((unsigned)((p.n_sites + 255) / 256), (unsigned)min(p.n_pairs, 1024));
    if (count_bytes == 2) hipLaunchKernelGGL((site_stats_kernel<uint16_t>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((site_stats_kernel<uint32_t>), grid, block, 0, st, p);
    return hipGetLastError();
}

template <typename CT>
__global__ __launch_bounds__(256) void pop_div_kernel(PopDivParams p) {
    const int64_t ls = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ls >= p.n_sites) return;
    const int64_t site = p.first_site + ls;
    const CT *cp = reinterpret_cast<const CT *>(p.counts) + (size_t)(site >> 6) * p.P * kTileSites + (site & 63);
    for (int k = blockIdx.y; k < p.P; k += gridDim.y) {
        const uint32_t v = cp[(size_t)k * kTileSites];
        const int c = (int)(v & CountTraits<CT>::mask), nm = (int)(v >> CountTraits<CT>::shift);
        const int n = p.n_full[k];
        const double pp = (double)c / (double)n;                  // fst.cpp:364-365 (full n)
        const double qq = 1.0 - pp;
        const double het = 1.0 - (pp * pp + qq * qq);             // fst.h:33-37
        const double nei = ((double)n * het) / (double)(n - 1);   // fst.h:40-45
        const double kk = (double)(c * (n - c));
        const double pi = (2.0 / (double)(nm * (nm - 1))) * kk;   // fst.cpp:346
        const size_t o = (size_t)k * p.n_sites + ls;
        if (p.het) p.het[o] = het;
        if (p.het_nei) p.het_nei[o] = nei;
        if (p.pi) p.pi[o] = pi;
    }
}

hipError_t launch_pop_div(const PopDivParams &p, int count_bytes, hipStream_t st) {
    if (p.n_sites <= 0 || p.P <= 0) return hipSuccess;
    dim3 block(256), grid((unsigned)((p.n_sites + 255) / 256), (unsigned)min(p.P, 1024));
    if (count_bytes == 2) hipLaunchKernelGGL((pop_div_kernel<uint16_t>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((pop_div_kernel<uint32_t>), grid, block, 0, st, p);
    return hipGetLastError();
}

// Sums of one pair's per-site values over ranges of sites, one thread per range, the sites in order: exactly the order in
// which vector_average / vector_average_withRegion add a window's values (process_vcf_utils.h:369-385), so a window that
// continues a partial sum of an earlier batch (init) comes out as the reference's one loop over it would.  The ranges are
// the caller's: SNP windows, physical windows, the whole shard (process_vcf_fst.cpp:589-692, 701-709).
__global__ __launch_bounds__(64) void pair_range_sums_kernel(RangeSumParams p) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= p.n_ranges) return;
    double a[kRangeSumCols];
#pragma unroll
    for (int k = 0; k < kRangeSumCols; ++k) a[k] = p.init ? p.init[r * kRangeSumCols + k] : 0.0;
    const int64_t lo = max((int64_t)0, p.lo[r]), hi = min(p.n_sites, p.hi[r]);
    for (int64_t s = lo; s < hi; ++s) {
        if (!p.include[s]) continue;                         // filter, fst.cpp:543-544
        a[0] += p.num[s]; a[1] += p.den[s]; a[2] += p.dxy[s];
        const double pi_i = p.pi_i[s], pi_j = p.pi_j[s];
        a[3] += p.het_i[s]; a[4] += p.het_j[s]; a[5] += p.nei_i[s]; a[6] += p.nei_j[s];
        a[7] += pi_i; a[8] += pi_j;
        a[9] += 1.0;
        a[10] += pi_i != 0.0 ? 1.0 : 0.0; a[11] += pi_j != 0.0 ? 1.0 : 0.0;   // (std::count(.., 0) of fst.cpp:609-610, complemented)
    }
#pragma unroll
    for (int k = 0; k < kRangeSumCols; ++k) p.out[r * kRangeSumCols + k] = a[k];
}

hipError_t launch_pair_range_sums(const RangeSumParams &p, hipStream_t st) {
    if (p.n_ranges <= 0) return hipSuccess;
    hipLaunchKernelGGL(pair_range_sums_kernel, dim3((unsigned)((p.n_ranges + 63) / 64)), dim3(64), 0, st, p);
    return hipGetLastError();
}

template <typename CT>
__global__ __launch_bounds__(256) void unpack_counts_kernel(const void *counts, int P, int64_t first_site,
                                                            int64_t n_sites, int32_t *alt, int32_t *nm) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_sites * P) return;
    const int64_t ls = e / P;
    const int k = (int)(e - ls * P);
    const int64_t site = first_site + ls;
    const uint32_t v = reinterpret_cast<const CT *>(counts)[((size_t)(site >> 6) * P + k) * kTileSites + (site & 63)];
    alt[e] = (int32_t)(v & CountTraits<CT>::mask);
    nm[e] = (int32_t)(v >> CountTraits<CT>::shift);
}

hipError_t launch_unpack_counts(const void *counts, int count_bytes, int P, int64_t first_site, int64_t n_sites,
                                int32_t *alt, int32_t *nm, hipStream_t st) {
    const int64_t n = n_sites * P;
    if (n <= 0) return hipSuccess;
    dim3 block(256), grid((unsigned)((n + 255) / 256));
    if (count_bytes == 2)
        hipLaunchKernelGGL((unpack_counts_kernel<uint16_t>), grid, block, 0, st, counts, P, first_site, n_sites, alt, nm);
    else
        hipLaunchKernelGGL((unpack_counts_kernel<uint32_t>), grid, block, 0, st, counts, P, first_site, n_sites, alt, nm);
    return hipGetLastError();
}

// =====================================================================================
// Fst-against-all (evo FstGlobal).
// =====================================================================================

// Number of window rows each chunk emits (evo_FstAgainstAll.cpp:238): a kept site emits when
// (used > SIZE || SIZE == STEP) && used % STEP == 0.  One thread per chunk walks its masks.
__global__ __launch_bounds__(256) void global_rows_kernel(const uint64_t *__restrict__ kept_mask, const uint64_t *__restrict__ skip_mask,
                                                          const int64_t *__restrict__ chunk_off, int n_chunks, int chunk_tiles,
                                                          int n_tiles, int size, int step, uint32_t *__restrict__ chunk_rows) {
    // one wave per chunk, lane = site of the tile: the site's `used` is the count of unskipped sites up to it
    const int lane = lane_id();
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= n_chunks) return;
    int64_t used0 = chunk_off[c];
    uint32_t r0 = (uint32_t)(used0 % step);           // used modulo STEP, carried in 32 bits
    uint32_t rows = 0;
    const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0ull;
    const int t0 = c * chunk_tiles, t1 = min(t0 + chunk_tiles, n_tiles);
    for (int t = t0; t < t1; ++t) {
        const uint64_t km = kept_mask[t], ns = ~skip_mask[t];
        const uint32_t k = (uint32_t)__popcll(ns & below) + 1u;   // rank of this site among the tile's unskipped sites
        const bool mine = ((ns >> lane) & 1ull) && ((km >> lane) & 1ull);
        const int64_t used = used0 + k;
        const bool emit = mine && (used > size || size == step) && ((r0 + k) % (uint32_t)step == 0u);
        rows += (uint32_t)__popcll(__ballot(emit));
        const uint32_t n = (uint32_t)__popcll(ns);
        used0 += n; r0 = (r0 + n) % (uint32_t)step;
    }
    if (lane == 0) chunk_rows[c] = rows;
}

hipError_t launch_global_rows(const uint64_t *kept_mask, const uint64_t *skip_mask, const int64_t *chunk_off,
                              int n_chunks, int chunk_tiles, int n_tiles, int size, int step,
                              uint32_t *chunk_rows, hipStream_t st) {
    if (n_chunks <= 0) return hipSuccess;
    dim3 block(256), grid((n_chunks + 3) / 4);
    hipLaunchKernelGGL(global_rows_kernel, grid, block, 0, st, kept_mask, skip_mask, chunk_off, n_chunks,
                       chunk_tiles, n_tiles, size, step, chunk_rows);
    return hipGetLastError();
}

// Reciprocal table of the against-all kernel.  The divisors are a population's called alleles n1 (and n1 - 1): at most
// max_n; and the complement's n2 = Tn - n1 (and n2 - 1): close below the total chromosome count tn_max unless a site
// has lost most of its genotypes.  So the table holds 1/n for n <= lo and 1/(tn_max - d) for d <= hi, and the rare
// divisor in between is divided for real -- the same correctly rounded value.  (A full table of tn_max + 1 doubles
// was 16 KB of LDS per block on the cfg3 shard, a quarter of what limits this latency-bound kernel's occupancy.)
constexpr int kRcpHiGlobal = 255;
void global_rcp_split(int tn_max, int max_n, int *lo, int *hi) {
    *lo = std::min(tn_max, max_n + 1);
    *hi = std::min(kRcpHiGlobal, tn_max - *lo - 1);      // < 0: the low part already covers everything
    if (*hi < 0) *hi = -1;
}
size_t global_lds_bytes(int tn_max, int max_n, int n_use, int size) {
    // ring [size][n_use] x {num, den} + ring of site ids + reciprocal table
    int lo, hi;
    global_rcp_split(tn_max, max_n, &lo, &hi);
    return (size_t)size * n_use * 16 + (size_t)size * 4 + (size_t)(lo + 1 + hi + 1) * 8 + 64;
}

// a / b through the reciprocal table (bit-identical to IEEE a/b on this path's domain, see K2)
__device__ __forceinline__ double rcp_div(double a, double b, double r) {
    const double q0 = a * r;
    return fma(fma(-q0, b, a), r, q0);
}

struct GlobalRcp {
    const double *tab;    // [lo + 1] then [hi + 1]
    int lo, hi, tn_max;
    // divisor known to be a population's own allele count (or one less): always in the low part
    __device__ __forceinline__ double small(int n) const { return tab[n]; }
    // divisor from the complement: in the high part unless the site lost most of its genotypes -- then, for the whole
    // wave at once (the branch is wave-uniform and all but never taken), the low part or a real division
    __device__ __forceinline__ double large(int n) const {
        const int d = tn_max - n;
        double r = tab[lo + 1 + (d <= hi ? d : 0)];
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(d > hi) != 0, 0)) {
            if (d > hi) r = n <= lo ? tab[n] : 1.0 / (double)n;
        }
        return r;
    }
};

template <bool RCP>
__device__ __forceinline__ void global_eval(uint32_t c, uint32_t nm, uint32_t Tc, uint32_t Tn, double thr,
                                            const GlobalRcp &rcp, double &num, double &den) {
    // evo_FstAgainstAll.cpp:245-261 with p1 = setAAFs (-1.0 when the pop has no alleles,
    // utils.h:75 + utils.cpp:56-58), p2 = complement AAF (utils.cpp:95)
    const int n1 = (int)nm, n2 = (int)(Tn - nm);
    if ((double)n1 >= thr) {
        double p1, p2, f1, f2;
        const double dc2 = (double)(int)(Tc - c), dn1 = (double)n1, dn2 = (double)n2;
        if (RCP) {
            p1 = nm > 0 ? rcp_div((double)c, dn1, rcp.small(n1)) : -1.0;
            p2 = rcp_div(dc2, dn2, rcp.large(n2));
            f1 = rcp_div(p1 * (1.0 - p1), (double)(n1 - 1), rcp.small(n1 > 0 ? n1 - 1 : 0));
            f2 = rcp_div(p2 * (1.0 - p2), (double)(n2 - 1), rcp.large(n2 > 0 ? n2 - 1 : 0));
        } else {
            p1 = nm > 0 ? (double)c / dn1 : -1.0;
            p2 = dc2 / dn2;
            f1 = (p1 * (1.0 - p1)) / (double)(n1 - 1);
            f2 = (p2 * (1.0 - p2)) / (double)(n2 - 1);
        }
        const double power = (p1 - p2) * (p1 - p2);
        num = power - f1 - f2;
        den = (p1 * (1.0 - p2)) + (p2 * (1.0 - p1));
    } else {
        num = 0.0; den = 0.0;
    }
}

// lane = population of populationsToUse; one block per chunk.  Each lane streams the count row of its
// own population (16-byte vectors = EPT consecutive sites, next vector prefetched) and owns one column of
// the two deques (a ring in LDS), so the site loop has neither a barrier nor a staging pass; the chunk
// first replays the last SIZE kept sites before it (the deque contents at its first site).
template <typename CT, bool RCP>
__global__ __launch_bounds__(1024) void global_window_kernel(GlobalParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    double *ring_n = reinterpret_cast<double *>(smem);                       // [size][n_use]
    double *ring_d = ring_n + (size_t)p.size * p.n_use;
    double *rcp_tab = ring_d + (size_t)p.size * p.n_use;                     // [rcp_lo + 1] + [rcp_hi + 1]
    int32_t *ring_s = reinterpret_cast<int32_t *>(rcp_tab + (p.rcp_lo + 1 + p.rcp_hi + 1));  // [size]
    const GlobalRcp rcp{rcp_tab, p.rcp_lo, p.rcp_hi, p.tn_max};

    // a workgroup walks `group` consecutive chunks of K1's plan: the replay of the deque contents before its first site
    // (up to a tile) and the tables below are paid once for all of them
    const int group = max(1, p.group);
    const int c = blockIdx.x * group;
    const int u = threadIdx.x;
    const bool active = u < p.n_use;
    const int k = active ? p.use_idx[u] : p.use_idx[0];
    const double thr = active ? p.thresholds[u] : 0.0;
    for (int i = threadIdx.x; i < p.size * p.n_use; i += blockDim.x) { ring_n[i] = 0.0; ring_d[i] = 0.0; }
    for (int i = threadIdx.x; i < p.size; i += blockDim.x) ring_s[i] = -1;
    for (int i = threadIdx.x; i <= p.rcp_lo; i += blockDim.x) rcp_tab[i] = 1.0 / (double)i;
    for (int i = threadIdx.x; i <= p.rcp_hi; i += blockDim.x) rcp_tab[p.rcp_lo + 1 + i] = 1.0 / (double)(p.tn_max - i);

    const int t0 = c * p.chunk_tiles, t1 = min(t0 + group * p.chunk_tiles, p.n_tiles);
    // ---- where the replay starts: walk back until SIZE kept sites of this shard are found
    int th = t0;
    int need = p.size;
    const int64_t kept_before = p.chunk_used_kept_off[(size_t)p.n_chunks + c] - p.base[1];
    if (kept_before < need) need = (int)kept_before;   // the rest is the zero-initialised deque
    int drop = 0;         // kept sites of tile th to ignore (older than the window)
    while (need > 0 && th > 0) {
        --th;
        const int n = __popcll(p.kept_mask[th]);
        if (n >= need) { drop = n - need; need = 0; }
        else need -= n;
    }
    int64_t used = p.chunk_used_kept_off[c];
    int64_t row = p.chunk_row_off[c];
    int head = 0;  // ring slot of the oldest element
    __syncthreads();
    {   // multi-GPU: kept sites of the shards before this one stand where this shard has fewer than SIZE before the chunk
        const int n_loc = (int)min((int64_t)p.size, kept_before);
        const int n_h = min(p.size - n_loc, p.halo_n);
        for (int e = p.halo_n - n_h; e < p.halo_n; ++e) {
            if (active) {
                ring_n[(size_t)head * p.n_use + u] = p.halo_num[(size_t)e * p.n_use + u];
                ring_d[(size_t)head * p.n_use + u] = p.halo_den[(size_t)e * p.n_use + u];
            }
            if (threadIdx.x == 0) ring_s[head] = p.halo_tag[e];
            head = head + 1 == p.size ? 0 : head + 1;
        }
    }

    constexpr int EPT = 16 / (int)sizeof(CT);
    constexpr int VPT = kTileSites / EPT;
    constexpr int EB = 8 * (int)sizeof(CT);
    const CT *cbase = reinterpret_cast<const CT *>(p.counts);
    auto vec_at = [&](int t, int j) {
        return *reinterpret_cast<const uint4 *>(cbase + ((size_t)t * p.P + k) * kTileSites + (size_t)j * EPT);
    };
    // totals {sum alt, sum nm} of the same EPT sites: EPT x 8 bytes, wave-uniform
    constexpr int TV = EPT / 2;                        // 16-byte vectors of totals per count vector
    auto tot_at = [&](int t, int j, uint4 (&dst)[TV]) {
        const uint4 *src = reinterpret_cast<const uint4 *>(p.totals + (size_t)t * kTileSites + (size_t)j * EPT);
#pragma unroll
        for (int x = 0; x < TV; ++x) dst[x] = src[x];
    };
    uint4 nxt = vec_at(th, 0);
    uint4 nxt_tot[TV];
    tot_at(th, 0, nxt_tot);
    uint32_t urem = (uint32_t)(used % p.step);         // used modulo STEP, carried in 32 bits
    const int size = p.size, nu = p.n_use;
    const uint32_t ustep = (uint32_t)p.step;
    double *my_n = ring_n + u, *my_d = ring_d + u;      // this lane's column of the two rings
    // window row of this lane's population: the deque oldest -> newest as vector_average walks it, i.e. ring slots
    // hd .. size-1, then 0 .. hd-1 (two straight runs, loads of four slots ahead of their adds)
    auto window_row = [&](int hd, int64_t row_) {
        double an = 0.0, ad = 0.0;
        auto run = [&](int from, int to) {
            int i = from;
            for (; i + 4 <= to; i += 4) {
                const double n0 = my_n[i * nu], n1 = my_n[(i + 1) * nu], n2 = my_n[(i + 2) * nu], n3 = my_n[(i + 3) * nu];
                const double d0 = my_d[i * nu], d1 = my_d[(i + 1) * nu], d2 = my_d[(i + 2) * nu], d3 = my_d[(i + 3) * nu];
                an += n0; an += n1; an += n2; an += n3;
                ad += d0; ad += d1; ad += d2; ad += d3;
            }
            for (; i < to; ++i) { an += my_n[i * nu]; ad += my_d[i * nu]; }
        };
        run(hd, size);
        run(0, hd);
        p.row_num[(size_t)row_ * nu + u] = an / (double)size;
        p.row_den[(size_t)row_ * nu + u] = ad / (double)size;
    };
    for (int t = th; t < t1; ++t) {
        const uint64_t km = p.kept_mask[t], sm = p.skip_mask[t];
        const bool replay = t < t0;
#pragma unroll 1
        for (int j = 0; j < VPT; ++j) {
            const uint4 cur = nxt;
            uint4 cur_tot[TV];
#pragma unroll
            for (int x = 0; x < TV; ++x) cur_tot[x] = nxt_tot[x];
            {   // prefetch the next vector of this lane's row (clamped at the end of the chunk)
                int tn = t, jn = j + 1;
                if (jn == VPT) { jn = 0; tn = min(t + 1, t1 - 1); }
                nxt = vec_at(tn, jn);
                tot_at(tn, jn, nxt_tot);
            }
            const uint32_t kb = (uint32_t)(km >> (j * EPT)) & ((1u << EPT) - 1u);
            const uint32_t sb = (uint32_t)(sm >> (j * EPT)) & ((1u << EPT) - 1u);
            if (kb == 0u) {                       // nothing kept among these sites: only the used counter moves
                if (!replay) { const uint32_t n = EPT - __popc(sb); used += n; urem = (urem + n) % ustep; }
                continue;
            }
            // 1) the EPT sites' values, independent of each other (no branches: the chains overlap)
            double numv[EPT], denv[EPT];
            {
                const uint32_t cw[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
                for (int e = 0; e < EPT; ++e) {
                    const uint32_t v = (cw[(e * EB) / 32] >> ((e * EB) % 32)) & (uint32_t)((1ull << EB) - 1ull);
                    const uint4 tv = cur_tot[e / 2];
                    const uint32_t Tc = (e & 1) ? tv.z : tv.x, Tn = (e & 1) ? tv.w : tv.y;
                    global_eval<RCP>(v & CountTraits<CT>::mask, v >> CountTraits<CT>::shift, Tc, Tn, thr, rcp, numv[e], denv[e]);
                }
            }
            // 2a) the usual vector -- every site kept, none skipped, windows longer than it: the EPT inserts with one
            // wave-uniform test per site (at most one window ends inside the vector)
            if (!replay && kb == ((1u << EPT) - 1u) && sb == 0u && size > EPT && ustep >= (uint32_t)EPT) {
                const int estar = (int)(ustep - 1u - urem);        // the site of the vector that completes a STEP, if < EPT
                const bool emits = estar < EPT && (used + estar + 1 > size || size == (int)ustep);
                const int site0 = t * kTileSites + j * EPT;
                int start_site = 0;
                if (emits && threadIdx.x == 0) {                   // the window's oldest site, before its slot is reused below
                    int os = head + estar + 1; if (os >= size) os -= size;
                    start_site = ring_s[os];
                }
                if (threadIdx.x < EPT) { int hs = head + (int)threadIdx.x; if (hs >= size) hs -= size; ring_s[hs] = site0 + (int)threadIdx.x; }
#pragma unroll
                for (int e = 0; e < EPT; ++e) {
                    int h = head + e; if (h >= size) h -= size;
                    if (active) { my_n[h * nu] = numv[e]; my_d[h * nu] = denv[e]; }
                    if (emits && e == estar) {
                        if (active) window_row(h + 1 == size ? 0 : h + 1, row);
                        if (threadIdx.x == 0) { p.row_start[row] = start_site; p.row_end[row] = site0 + e; }
                        ++row;
                    }
                }
                head += EPT; if (head >= size) head -= size;
                used += EPT; urem += EPT; if (urem >= ustep) urem -= ustep;
                continue;
            }
            // 2b) the deque walk, in site order (LDS offsets in 32 bits; the ring is walked as its two straight runs)
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                if ((sb >> e) & 1u) continue;
                if (!replay) { ++used; urem = urem + 1 == ustep ? 0u : urem + 1; }
                if (!((kb >> e) & 1u)) continue;
                if (replay && drop > 0) { --drop; continue; }
                const int site = t * kTileSites + j * EPT + e;
                if (active) {
                    my_n[head * nu] = numv[e];
                    my_d[head * nu] = denv[e];
                }
                if (threadIdx.x == 0) ring_s[head] = site;
                head = head + 1 == size ? 0 : head + 1;
                if (!replay && (used > size || size == (int)ustep) && urem == 0u) {
                    if (active) window_row(head, row);
                    if (threadIdx.x == 0) {
                        p.row_start[row] = ring_s[head];
                        p.row_end[row] = site;
                    }
                    ++row;
                }
            }
        }
    }
}

// ---- the same rows, tile by tile in two phases (what launch_global takes when the tables fit in LDS)
// Sixteen waves per workgroup, one workgroup per group of K1 chunks.  Per tile of 64 sites:
//   phase E   lane = site, the waves share the populations: every lane evaluates its site for one population after the
//             other (64 of 64 lanes busy, the chains of different lanes independent) and writes {numerator,
//             denominator} of the kept sites into the population's row of a ring in LDS, K[population][kept site];
//   phase W   thread = (window that ends in the tile, population): the window's SIZE entries of its row, oldest to
//             newest, added one after the other as vector_average does (evo_FstAgainstAll.cpp:265-280) -- the rows are
//             bit for bit those of global_window_kernel.
// The ring holds SIZE + 64 kept sites per population (odd row length: the lanes of phase W, one population each, hit
// different LDS banks).  The deque before the first tile: zeros, then the halo of the shards before this one, then the
// last kept sites before the group (replayed through phase E), as in the kernel above.
constexpr int kGlobalTileThreads = 1024;
constexpr int kGlobalTilePops = 10;          // populations per wave of global_tile_kernel at most (16 waves: 160 populations)
size_t global_tile_lds_bytes(int tn_max, int max_n, int n_use, int size, int *kw_out) {
    int lo, hi;
    global_rcp_split(tn_max, max_n, &lo, &hi);
    const int kw = (size + kTileSites) | 1;
    if (kw_out) *kw_out = kw;
    return (size_t)n_use * kw * 16 + (size_t)(lo + 1 + hi + 1) * 8 + (size_t)n_use * 64 + (size_t)kw * 4 + 64;
}

// global_eval for a population that has every allele called at all 64 sites of a tile whose sites have every allele of
// every population called: n1, n2 and their reciprocals are the same for the whole wave (k = {n1, 1/n1, n2, 1/n2, n1 - 1,
// 1/(n1 - 1), n2 - 1, 1/(n2 - 1)} as doubles, n1 >= 2, n2 >= 2, the 0.8 rule passed).  Same operations on the same values as
// global_eval<true>, so the same bits.
__device__ __forceinline__ void global_eval_full(uint32_t c, uint32_t Tc, const double *k, double &num, double &den) {
    const double dc = (double)c, dc2 = (double)(int)(Tc - c);
    const double p1 = rcp_div(dc, k[0], k[1]);
    const double p2 = rcp_div(dc2, k[2], k[3]);
    const double f1 = rcp_div(p1 * (1.0 - p1), k[4], k[5]);
    const double f2 = rcp_div(p2 * (1.0 - p2), k[6], k[7]);
    const double power = (p1 - p2) * (p1 - p2);
    num = power - f1 - f2;
    den = (p1 * (1.0 - p2)) + (p2 * (1.0 - p1));
}

// MP: populations per wave at most (the loops over them are unrolled and their counts live in registers one tile ahead: a
// kernel cut for 7 -- 112 populations -- carries less than one cut for kGlobalTilePops)
template <typename CT, bool RCP, int MP>
__global__ __launch_bounds__(kGlobalTileThreads) void global_tile_kernel(GlobalParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int nu = p.n_use, KW = p.kw, size = p.size;
    double *K_n = reinterpret_cast<double *>(smem);                          // [n_use][KW]
    double *K_d = K_n + (size_t)nu * KW;
    double *rcp_tab = K_d + (size_t)nu * KW;                                 // [rcp_lo + 1] + [rcp_hi + 1]
    double *full_k = rcp_tab + (p.rcp_lo + 1 + p.rcp_hi + 1);                // [n_use][8] global_eval_full's constants ([0] = 0: not for this population)
    int32_t *K_s = reinterpret_cast<int32_t *>(full_k + (size_t)nu * 8);     // [KW] site of the entry (-1: the zero-initialised deque)
    const GlobalRcp rcp{rcp_tab, p.rcp_lo, p.rcp_hi, p.tn_max};
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NW = kGlobalTileThreads / 64;
    const int group = max(1, p.group);
    const int c = blockIdx.x * group;
    for (int i = tid; i < nu * KW; i += kGlobalTileThreads) { K_n[i] = 0.0; K_d[i] = 0.0; }
    for (int i = tid; i < KW; i += kGlobalTileThreads) K_s[i] = -1;
    for (int i = tid; i <= p.rcp_lo; i += kGlobalTileThreads) rcp_tab[i] = 1.0 / (double)i;
    for (int i = tid; i <= p.rcp_hi; i += kGlobalTileThreads) rcp_tab[p.rcp_lo + 1 + i] = 1.0 / (double)(p.tn_max - i);
    for (int u = tid; u < nu; u += kGlobalTileThreads) {
        const int n1 = p.n_full[p.use_idx[u]], n2 = p.tn_max - n1;
        const bool ok = n1 >= 2 && n2 >= 2 && (double)n1 >= p.thresholds[u];
        const double v[8] = {(double)n1, 1.0 / (double)n1, (double)n2, 1.0 / (double)n2, (double)(n1 - 1), 1.0 / (double)(n1 - 1), (double)(n2 - 1), 1.0 / (double)(n2 - 1)};
        for (int i = 0; i < 8; ++i) full_k[(size_t)u * 8 + i] = ok ? v[i] : 0.0;
    }

    const int t0 = c * p.chunk_tiles, t1 = min(t0 + group * p.chunk_tiles, p.n_tiles);
    // where the replay starts: walk back until SIZE kept sites of this shard are found
    int th = t0;
    int need = size;
    const int64_t kept_before = p.chunk_used_kept_off[(size_t)p.n_chunks + c] - p.base[1];
    if (kept_before < need) need = (int)kept_before;   // the rest is the zero-initialised deque
    int drop = 0;         // kept sites of tile th to ignore (older than the window)
    while (need > 0 && th > 0) {
        --th;
        const int n = __popcll(p.kept_mask[th]);
        if (n >= need) { drop = n - need; need = 0; }
        else need -= n;
    }
    int64_t used = p.chunk_used_kept_off[c];
    int64_t row = p.chunk_row_off[c];
    int pos = 0;          // ring position of the next kept site
    __syncthreads();
    {   // multi-GPU: kept sites of the shards before this one stand where this shard has fewer than SIZE before the group
        const int n_loc = (int)min((int64_t)size, kept_before);
        const int n_h = min(size - n_loc, p.halo_n);
        for (int e = p.halo_n - n_h; e < p.halo_n; ++e) {
            for (int u = tid; u < nu; u += kGlobalTileThreads) {
                K_n[(size_t)u * KW + pos] = p.halo_num[(size_t)e * nu + u];
                K_d[(size_t)u * KW + pos] = p.halo_den[(size_t)e * nu + u];
            }
            if (tid == 0) K_s[pos] = p.halo_tag[e];
            ++pos;                                       // (n_h <= SIZE < KW: no wrap yet)
        }
    }
    const CT *cbase = reinterpret_cast<const CT *>(p.counts);
    const uint64_t below = (1ull << lane) - 1ull;        // the sites before this lane's
    const uint64_t upto = below | (1ull << lane);
    const uint32_t ustep = (uint32_t)p.step;
    // this wave's populations (wave, wave + 16, ...: at most kGlobalTilePops of them) and, one tile ahead of its use, the
    // counts of this lane's site in each of them and the site's totals: the loads of tile t + 1 fly while tile t is evaluated
    int kk[MP];
    double thr[MP];
#pragma unroll
    for (int i = 0; i < MP; ++i) {
        const int u = min(wave + i * NW, nu - 1);
        kk[i] = p.use_idx[u]; thr[i] = p.thresholds[u];
    }
    const int npw = wave < nu ? (nu - wave + NW - 1) / NW : 0;     // populations of this wave
    uint32_t cv[MP], nv[MP];
    uint2 ctot, ntot;
    auto request = [&](int t, uint32_t (&dst)[MP], uint2 &tot) {
        const CT *crow = cbase + (size_t)t * p.P * kTileSites + lane;
#pragma unroll
        for (int i = 0; i < MP; ++i) dst[i] = i < npw ? (uint32_t)crow[(size_t)kk[i] * kTileSites] : 0u;
        tot = p.totals[(size_t)t * kTileSites + lane];
    };
    if (th < t1) request(th, nv, ntot);
    uint64_t nsm = th < t1 ? p.skip_mask[th] : 0ull, nkm = th < t1 ? p.kept_mask[th] : 0ull;   // the masks too come a tile ahead
    for (int t = th; t < t1; ++t) {
        const bool replay = t < t0;
        const uint64_t sm = nsm;
        uint64_t km = nkm;
        { const int tn = min(t + 1, t1 - 1); nsm = p.skip_mask[tn]; nkm = p.kept_mask[tn]; }
        if (replay && drop > 0) {                        // the oldest kept sites of the first replayed tile are older than the deque
            for (; drop > 0 && km != 0ull; --drop) km &= km - 1ull;
        }
        const int nk = __popcll(km);
#pragma unroll
        for (int i = 0; i < MP; ++i) cv[i] = nv[i];
        ctot = ntot;
        request(min(t + 1, t1 - 1), nv, ntot);
        // ---- phase E
        const bool kept = (km >> lane) & 1ull;
        int wpos = pos + __popcll(km & below);
        if (wpos >= KW) wpos -= KW;
        if (nk > 0) {
            const bool tile_full = __builtin_amdgcn_ballot_w64(ctot.y != (uint32_t)p.tn_max) == 0ull;   // every allele of every population called
#pragma unroll
            for (int i = 0; i < MP; ++i) {
                if (i < npw) {
                    double num, den;
                    const uint32_t cc_ = cv[i] & CountTraits<CT>::mask, nm_ = cv[i] >> CountTraits<CT>::shift;
                    const double *fk = full_k + (size_t)(wave + i * NW) * 8;
                    // (tile_full: the totals are the sum of the populations' own counts, so this one is complete too)
                    if (tile_full && fk[0] != 0.0) global_eval_full(cc_, ctot.x, fk, num, den);
                    else global_eval<RCP>(cc_, nm_, ctot.x, ctot.y, thr[i], rcp, num, den);
                    if (kept) { const size_t o = (size_t)(wave + i * NW) * KW + wpos; K_n[o] = num; K_d[o] = den; }
                }
            }
            if (wave == 0 && kept) K_s[wpos] = t * kTileSites + lane;
        }
        // the sites of this tile that complete a window (the same in every wave): kept, and the count of used sites up to
        // and including it is a multiple of STEP beyond the first SIZE (evo_FstAgainstAll.cpp:263-264)
        uint64_t emit = 0ull;
        if (!replay && nk > 0) {
            const int64_t used_here = used + (int64_t)__popcll(~sm & upto);
            const bool e_ = kept && !((sm >> lane) & 1ull) && (used_here > size || size == (int)ustep) && (uint32_t)(used_here % ustep) == 0u;
            emit = __builtin_amdgcn_ballot_w64(e_);
        }
        __syncthreads();
        // ---- phase W
        // a wave takes one window at a time (wave-uniform ring positions) and 64 populations of it
        const int n_emit = __popcll(emit);
        const int ublocks = (nu + 63) >> 6;                       // groups of 64 populations
        for (int item = wave; item < n_emit * ublocks; item += NW) {
            const int w = item / ublocks, u = (item - w * ublocks) * 64 + lane;
            const int e = nth_set_bit64((uint32_t)emit, (uint32_t)(emit >> 32), w + 1);
            int j = pos + __popcll(km & ((2ull << e) - 1ull)) - 1;        // ring position of site e's entry
            if (j >= KW) j -= KW;
            int i0 = j - size + 1;
            if (i0 < 0) i0 += KW;
            if (u < nu && row + w < p.row_cap) {                  // (row_cap: never reached, launch_global_rows counted the same rows)
                const double *rn = K_n + (size_t)u * KW, *rd = K_d + (size_t)u * KW;
                double an = 0.0, ad = 0.0;
                // oldest to newest: ring positions i0 .. (two straight runs when the window wraps), four loads ahead of their adds
                auto run = [&](int from, int to) {
                    int i = from;
                    for (; i + 4 <= to; i += 4) {
                        const double n0 = rn[i], n1 = rn[i + 1], n2 = rn[i + 2], n3 = rn[i + 3];
                        const double d0 = rd[i], d1 = rd[i + 1], d2 = rd[i + 2], d3 = rd[i + 3];
                        an += n0; an += n1; an += n2; an += n3;
                        ad += d0; ad += d1; ad += d2; ad += d3;
                    }
                    for (; i < to; ++i) { an += rn[i]; ad += rd[i]; }
                };
                if (i0 + size <= KW) run(i0, i0 + size);
                else { run(i0, KW); run(0, i0 + size - KW); }
                p.row_num[(size_t)(row + w) * nu + u] = an / (double)size;
                p.row_den[(size_t)(row + w) * nu + u] = ad / (double)size;
                if (u == 0) { p.row_start[row + w] = K_s[i0]; p.row_end[row + w] = t * kTileSites + e; }
            }
        }
        row += n_emit;
        pos += nk; if (pos >= KW) pos -= KW;
        if (!replay) used += kTileSites - __popcll(sm);
        __syncthreads();                                 // the next tile's entries may take the place of the oldest ones
    }
}

template <typename CT, bool RCP, int MP>
static hipError_t launch_global_tile_t(const GlobalParams &p_, size_t lds_bytes, hipStream_t st) {
    const void *fn = reinterpret_cast<const void *>(&global_tile_kernel<CT, RCP, MP>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    GlobalParams p = p_;
    if (p.group <= 0) {
        // one workgroup per CU at a time: the groups are cut so that the workgroups fill whole rounds of the device
        static thread_local struct { int dev; int cus; } memo = {-1, 0};
        int dev = 0;
        if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
        if (memo.dev != dev) {
            int cus = 0;
            if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
            memo = {dev, std::max(1, cus)};
        }
        const int64_t cap = memo.cus;
        double best = -1.0;
        p.group = 1;
        for (int r = 1; r <= 8; ++r) {
            const int g = (int)std::max<int64_t>(1, (p.n_chunks + cap * r - 1) / (cap * r));
            const int64_t blocks = (p.n_chunks + g - 1) / g, rounds = (blocks + cap - 1) / cap;
            const double tiles = (double)g * p.chunk_tiles;
            const double score = (double)blocks / (double)(rounds * cap) * tiles / (tiles + 1.0);
            if (score > best + 1e-9) { best = score; p.group = g; }
        }
    }
    dim3 grid((p.n_chunks + p.group - 1) / p.group);
    hipLaunchKernelGGL((global_tile_kernel<CT, RCP, MP>), grid, dim3(kGlobalTileThreads), lds_bytes, st, p);
    return hipGetLastError();
}

hipError_t launch_global_tile(const GlobalParams &p, int count_bytes, size_t lds_bytes, hipStream_t st) {
    if (p.n_chunks <= 0 || p.n_use <= 0) return hipSuccess;
    if (p.kw < p.size + kTileSites || (p.kw & 1) == 0 || p.halo_n > p.size || lds_bytes > 160 * 1024 ||
        p.n_use > kGlobalTilePops * (kGlobalTileThreads / 64)) return hipErrorInvalidValue;
    const int per_wave = (p.n_use + kGlobalTileThreads / 64 - 1) / (kGlobalTileThreads / 64);
    if (count_bytes == 2) {
        if (per_wave <= 4) return launch_global_tile_t<uint16_t, true, 4>(p, lds_bytes, st);
        if (per_wave <= 7) return launch_global_tile_t<uint16_t, true, 7>(p, lds_bytes, st);
        return launch_global_tile_t<uint16_t, true, kGlobalTilePops>(p, lds_bytes, st);
    }
    if (per_wave <= 4) return launch_global_tile_t<uint32_t, true, 4>(p, lds_bytes, st);
    if (per_wave <= 7) return launch_global_tile_t<uint32_t, true, 7>(p, lds_bytes, st);
    return launch_global_tile_t<uint32_t, true, kGlobalTilePops>(p, lds_bytes, st);
}

template <typename CT, bool RCP>
static hipError_t launch_global_t(const GlobalParams &p_, size_t lds_bytes, dim3 block, hipStream_t st) {
    const void *fn = reinterpret_cast<const void *>(&global_window_kernel<CT, RCP>);
    if (lds_bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    GlobalParams p = p_;
    if (p.group <= 0) {
        // K1 chunks per workgroup: the kernel's time follows the number of rounds its workgroups need on the device (on
        // the cfg3 shard 8 / 16 / 24 / 32 / 48 tiles per workgroup take 1.07 / 1.12 / 1.31 / 0.89 / 1.33 ms: 977 workgroups are
        // one round of the 1024 the device holds), so the groups are cut to fill whole rounds, the fewer the better
        // (every workgroup first replays up to a tile before its own sites).
        static thread_local struct { const void *fn; int threads; size_t lds; int dev; int cap; } memo = {nullptr, 0, 0, -1, 0};
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (memo.fn != fn || memo.threads != (int)block.x || memo.lds != lds_bytes || memo.dev != dev) {
            int per_cu = 0, cus = 0;
            if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, (int)block.x, lds_bytes)) != hipSuccess) return e;
            if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
            memo = {fn, (int)block.x, lds_bytes, dev, std::max(1, per_cu) * std::max(1, cus)};
        }
        const int64_t cap = memo.cap;
        double best = -1.0;
        p.group = 1;
        for (int r = 1; r <= 8; ++r) {
            const int g = (int)std::max<int64_t>(1, (p.n_chunks + cap * r - 1) / (cap * r));
            const int64_t blocks = (p.n_chunks + g - 1) / g, rounds = (blocks + cap - 1) / cap;
            const double tiles = (double)g * p.chunk_tiles;
            const double score = (double)blocks / (double)(rounds * cap) * tiles / (tiles + 1.0);
            if (score > best + 1e-9) { best = score; p.group = g; }
        }
    }
    dim3 grid((p.n_chunks + p.group - 1) / p.group);
    hipLaunchKernelGGL((global_window_kernel<CT, RCP>), grid, block, lds_bytes, st, p);
    return hipGetLastError();
}

hipError_t launch_global(const GlobalParams &p, int count_bytes, size_t lds_bytes, hipStream_t st) {
    if (p.n_chunks <= 0 || p.n_use <= 0) return hipSuccess;
    int threads = ((p.n_use + 63) / 64) * 64;
    dim3 block(threads);
    const bool rcp = true;   // the table is small whatever tn_max is
    if (count_bytes == 2) return rcp ? launch_global_t<uint16_t, true>(p, lds_bytes, block, st)
                                     : launch_global_t<uint16_t, false>(p, lds_bytes, block, st);
    return rcp ? launch_global_t<uint32_t, true>(p, lds_bytes, block, st)
               : launch_global_t<uint32_t, false>(p, lds_bytes, block, st);
}

// per-site against-all values on demand: lane = site
template <typename CT>
__global__ __launch_bounds__(256) void global_site_kernel(GlobalParams p) {
    const int64_t ls = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ls >= p.n_out_sites) return;
    const int64_t site = p.first_site + ls;
    const int64_t t = site >> 6;
    const int s = (int)(site & 63);
    const CT *cp = reinterpret_cast<const CT *>(p.counts) + (size_t)t * p.P * kTileSites + s;
    const uint2 T = p.totals[site];
    if (blockIdx.y == 0 && p.site_kept) p.site_kept[ls] = (uint8_t)((p.kept_mask[t] >> s) & 1ull);
    for (int u = blockIdx.y; u < p.n_use; u += gridDim.y) {
        const uint32_t v = cp[(size_t)p.use_idx[u] * kTileSites];
        double num, den;
        global_eval<false>(v & CountTraits<CT>::mask, v >> CountTraits<CT>::shift, T.x, T.y, p.thresholds[u], GlobalRcp{nullptr, 0, 0, 0}, num, den);
        if (p.site_num) p.site_num[(size_t)u * p.n_out_sites + ls] = num;
        if (p.site_den) p.site_den[(size_t)u * p.n_out_sites + ls] = den;
    }
}

hipError_t launch_global_sites(const GlobalParams &p, int count_bytes, hipStream_t st) {
    if (p.n_out_sites <= 0 || p.n_use <= 0) return hipSuccess;
    dim3 block(256), grid((unsigned)((p.n_out_sites + 255) / 256), (unsigned)min(p.n_use, 1024));
    if (count_bytes == 2) hipLaunchKernelGGL((global_site_kernel<uint16_t>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((global_site_kernel<uint32_t>), grid, block, 0, st, p);
    return hipGetLastError();
}

// =====================================================================================
// Synthetic workload generator, device side so that full-size inputs never cross
// PCIe.  Counter-based: every bit is a hash of (seed, global site, haplotype).
// =====================================================================================
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ float u01(uint64_t h) { return (float)((h >> 40) + 1) * (1.0f / 16777217.0f); }

__global__ __launch_bounds__(256) void synth_kernel(SynthParams p) {
    const int64_t site = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t t = site >> 6;
    if (t >= p.n_tiles) return;
    const int s = (int)(site & 63);
    const bool valid = site < p.n_sites;
    const uint64_t gs = (uint64_t)(p.first_site_global + site);
    const uint64_t key = mix64(p.seed ^ (gs * 0xD1B54A32D192ED03ull));
    // base frequency f ~ Beta(0.5, 0.5) = sin^2(pi*u/2)
    const float sb = __sinf(1.57079632679f * u01(mix64(key ^ 0x1111ull)));
    const float f = sb * sb;
    ulonglong2 *tp = p.tiles + (size_t)t * p.W * kTileSites + s;
    uint32_t thr = 0;
    for (int w = 0; w < p.W; ++w) {
        uint64_t alt = 0, miss = 0;
        for (int sg = p.word_seg_start[w]; sg < p.word_seg_start[w + 1]; ++sg) {
            const SegDesc d = p.segs[sg];
            // per-pop frequency p_k = clip(f + N(0, 0.15))
            const uint64_t hk = mix64(key ^ (0x2222ull + (uint64_t)d.pop * 0x9E3779B1ull));
            const float g = sqrtf(-2.0f * __logf(u01(hk))) * __cosf(6.28318530718f * u01(mix64(hk)));
            float pk = f + 0.15f * g;
            pk = fminf(fmaxf(pk, 0.0f), 1.0f);
            thr = (uint32_t)fminf(pk * 4294967296.0f, 4294967295.0f);
            uint64_t mm = d.mask;
            while (mm) {
                const int b = __ffsll((unsigned long long)mm) - 1;
                mm &= mm - 1;
                const uint64_t hap = (uint64_t)w * 64 + b;
                const uint64_t hh = mix64(key ^ (0x3333ull + hap * 0xC2B2AE3D27D4EB4Full));
                if ((uint32_t)hh < thr && pk > 0.0f) alt |= 1ull << b;
                const uint64_t hm = mix64(key ^ (0x4444ull + (hap >> 1) * 0x165667B19E3779F9ull));
                if ((uint32_t)hm < p.miss_thr) miss |= 1ull << b;
            }
        }
        alt &= ~miss;
        if (!valid) { alt = 0; miss = 0; }
        tp[(size_t)w * kTileSites] = make_ulonglong2(alt, miss);
    }
    if (valid && p.pos) p.pos[site] = (int32_t)(((uint64_t)(p.first_site_global + site)) * 100ull + (mix64(key ^ 0x5555ull) % 100ull) + 1ull);
}

hipError_t launch_synth(const SynthParams &p, hipStream_t st) {
    if (p.n_tiles <= 0) return hipSuccess;
    const int64_t n = (int64_t)p.n_tiles * kTileSites;
    dim3 block(256), grid((unsigned)((n + 255) / 256));
    hipLaunchKernelGGL(synth_kernel, grid, block, 0, st, p);
    return hipGetLastError();
}

}  // namespace evofst
