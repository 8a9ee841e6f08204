// evofst_stream.hip -- pairs mode as ONE streaming kernel with producer and consumer waves (gfx950, wave64).
//
// Replaces, for every population pair at once, the per-line work of getFstFromVCF: the counting loops
// (process_vcf_fst.cpp:243-280), the site filter (:543-544), calculateFstNumerator / Denominator
// (process_vcf_fst.h:19-30), calculateDxy (process_vcf_fst.cpp:297-323) and the SNP windows (:638-650).
//
// A workgroup is a few PRODUCER waves and one or more CONSUMER waves (one per 64 pairs at most; optionally two sets of
// them that take the chunks in turn) and works through chunks of a few tiles in ticket order (one atomic per chunk,
// drawn by the producer that finishes the chunk before it first).  For chunk k of a workgroup, in LDS buffer k mod NB:
//   the producers  count it: lane = site, 16-byte coalesced loads of the tile rows (three batches of four rows requested
//                  ahead of the one being counted), popcounts under the wave-uniform masks of the layout's descriptor
//                  stream (scalar cache), one alt-count byte per site and population and the zero / fixed ballots of
//                  every population INTO LDS -- nothing but the tiles is read from HBM and, on data without missing
//                  alleles, no count is written to it (a (tile, population) row with a missing allele leaves as packed
//                  counts right away: its chunk goes to the general walk); then the census of included sites per pair
//                  (filter :543-544, lane = pair), which the producer that finishes last publishes as one row of
//                  2-byte words.  A SIMD's first producer wave takes tr tiles of the chunk, its second tr2 <= tr: the
//                  hardware serves the older wave first, and with 4 : 3 both arrive together;
//   the consumers  take it once it is published: look-back over the rows of the chunks before it (three levels of
//                  fan-out 32: chunks of its group, groups of its super-group, super-groups); then the statistics
//                  walk over the chunk's counts in LDS (exact integer sums, four sites per v_dot4_u32_u8, segment rows
//                  at the window boundaries) -- the arithmetic of clean_walk_kernel.
// No workgroup barrier after the set-up: the waves hand buffers over through counters in LDS (producers arrived, chunk
// published, consumers done), so a wave only waits for a wave that is ahead of it.  The producers never wait for
// another workgroup, and a consumer only for chunks with smaller tickets, whose producers are already running:
// whatever the dispatch order or the number of workgroups resident, the smallest unpublished chunk is always being
// counted by somebody who waits for nothing.  While a consumer waits for its predecessors the producers beside it keep
// the tile loads going, which is what the lock-step one-kernel pass of evofst_fused.hip could not do.
//
// A chunk with a missing allele (the walk above cannot take it; skipped sites it can) leaves its packed counts, ballots
// and its pairs' offsets in HBM, exactly as K1 + scan would have, and is walked by fused_pairs_kernel's HBM mode in a
// second launch that finds nothing to do on clean data.  stitch_kernel completes the segments that straddle chunks.
//
// The look-back state carries no launch tag: a row of the granule table is complete when its flag word has counted all
// groups of pairs.  The flag words exist twice; launch L uses copy L & 1 (of flags and rows) and zeroes, chunk by chunk,
// the flags of the other copy for launch L + 1.  The launch counter is a device word that the launch's last ticket draw
// bumps, so a replayed launch (HIP graph) alternates too.
#include "evofst_internal.h"

#include <algorithm>

namespace evofst {

typedef uint32_t SegWords __attribute__((ext_vector_type(4)));                  // {mask lo, mask hi, pop, flags | n_full << 8}
typedef const __attribute__((address_space(4))) SegWords *SegStream;
extern "C" __device__ int evofst_write_lane(int val, int lane, int old) __asm("llvm.amdgcn.writelane.i32");

typedef __attribute__((address_space(1))) uint32_t gu32;
typedef __attribute__((address_space(1))) uint64_t gu64;
typedef __attribute__((address_space(1))) uint16_t gu16;

#ifdef EVOFST_EXPERIMENTS
// slot k of (workgroup, iteration): the 100 MHz real-time counter, written by one lane
#define STREAM_STAMP(p, it, k) do { if ((p).stamps && (it) < kStreamStampIters && (threadIdx.x & 63) == 0) \
    (p).stamps[((size_t)blockIdx.x * kStreamStampIters + (it)) * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define STREAM_STAMP(p, it, k) do {} while (0)
#endif
constexpr uint32_t kStreamSpinLimit = 1u << 20;   // polls (with a sleep each) before a wait gives up: seconds

static __host__ __device__ constexpr size_t s16(size_t x) { return (x + 15) & ~(size_t)15; }

// One LDS buffer = one counted chunk: alt counts, one byte each, rows of kStreamRow bytes per (tile, population) (64 sites
// + padding: rows of different populations start in different banks); zero / fixed ballots [tile][z|f][P]; five words per
// tile (the four site masks and the populations whose row has a missing allele); the quarter table of the integer walk
// (stream_quarter_table), which the consumers fill in.  The non-missing counts are not kept: a row without a missing
// allele has n_full of them at every site, and a row with one -- its chunk goes to the general walk -- leaves the
// producer as packed counts straight for HBM (stream_count_piece); the consumer writes the chunk's other rows.
struct StreamBuf { unsigned char *c; uint64_t *zf, *flg; uint32_t *qt; };
constexpr int kStreamFlg = 5;
static __host__ __device__ size_t stream_c_bytes(int P, int chunk_tiles) { return s16((size_t)chunk_tiles * P * kStreamRow); }
static __host__ __device__ size_t stream_buf_bytes(int P, int chunk_tiles) {
    return stream_c_bytes(P, chunk_tiles) + s16((size_t)chunk_tiles * 2 * P * 8) + s16((size_t)chunk_tiles * kStreamFlg * 8) +
           s16((size_t)chunk_tiles * P * 32);
}
size_t stream_buffer_bytes(int P, int chunk_tiles) { return stream_buf_bytes(P, chunk_tiles); }
static __device__ __forceinline__ StreamBuf stream_buf(unsigned char *base, int P, int chunk_tiles) {
    StreamBuf b;
    b.c = base;
    b.zf = reinterpret_cast<uint64_t *>(b.c + stream_c_bytes(P, chunk_tiles));
    b.flg = b.zf + (size_t)chunk_tiles * 2 * P;
    b.qt = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(b.flg) + s16((size_t)chunk_tiles * kStreamFlg * 8));
    return b;
}
// + the table of byte masks, the workgroup's words, three per-pair arrays
size_t stream_lds_bytes(int P, int n_pairs, int chunk_tiles, int n_buf) {
    return 1024 + 128 + (size_t)(1 + n_buf) * s16((size_t)n_pairs * 4) + (size_t)n_buf * stream_buffer_bytes(P, chunk_tiles);
}

// ------------------------------------------------------------------------------------------------ producer
// Counts tiles [t0, t0 + n_own) of the shard into slots slot0, slot0 + 1, ... of an LDS buffer.  All arguments but
// `lane` are wave-uniform.  The batch pipeline is K1's (evofst_kernels.hip): every request is unconditional, so the
// wait for a batch is a count of younger loads, never "everything outstanding".
__device__ __forceinline__ void stream_count_piece(const StreamParams &p, int lane, int t0, int n_own, int slot0, const StreamBuf &buf, SegStream segs) {
    uint64_t *zf = buf.zf, *flg = buf.flg;
    constexpr int NB = 4;                                // rows per batch
    const int P = p.P;
    const int nb = (p.W + NB - 1) / NB;                  // batches per tile
    const int n_items = n_own * nb;
    struct Batch { ulonglong2 r[NB]; uint32_t flag; };
    Batch b0, b1, b2, b3;
    b0.flag = b1.flag = b2.flag = b3.flag = 0u;
    const bool have_flags = p.site_flags != nullptr;
    const ulonglong2 *dummy = p.tiles + (size_t)t0 * p.W * kTileSites;   // the wave's own first line
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
    load_item(b0);
    load_item(b1);
    load_item(b2);

    uint32_t c = 0, mc = 0;
    uint32_t irregular = 0, anymiss = 0, nanany = 0;     // per-lane 0 / 1 words (see K1)
    bool skip = false;
    uint64_t skipmask = 0;
    int rowmiss = 0;                                     // lane = population: its row of this tile has a missing allele (set on the rare path only)
    int zlo = 0, zhi = 0, flo = 0, fhi = 0;              // ballots by population, lane = population
    auto tile_done = [&](int slot) {                     // the tile's ballots -> LDS
        if (lane < P) {
            zf[(size_t)slot * 2 * P + lane] = ((uint64_t)(uint32_t)zhi << 32) | (uint32_t)zlo;
            zf[(size_t)slot * 2 * P + P + lane] = ((uint64_t)(uint32_t)fhi << 32) | (uint32_t)flo;
        }
        const uint64_t im = __ballot(irregular != 0 && !skip), mm = __ballot(anymiss != 0 && !skip), nn = __ballot(nanany != 0 && !skip);
        if (lane == 0) {
            flg[slot * kStreamFlg + 0] = im; flg[slot * kStreamFlg + 1] = skipmask; flg[slot * kStreamFlg + 2] = mm; flg[slot * kStreamFlg + 3] = nn;
        }
        const uint64_t rm = __ballot(rowmiss != 0);
        if (lane == 0) flg[slot * kStreamFlg + 4] = rm;
    };
    int sidx = 0;
    SegWords d = segs[0], dn = segs[1];
    auto step = [&](Batch &cur, Batch &spare) {
        const int ti = c_ti, bi = c_bi;
        if (++c_bi == nb) { c_bi = 0; ++c_ti; }
        const bool live = ti < n_own;
        const int t = t0 + ti, wb = bi * NB;
        if (bi == 0 && live) {
            if (ti > 0) tile_done(slot0 + ti - 1);
            skip = (int64_t)t * kTileSites + lane >= p.n_sites || (have_flags && (cur.flag & 1u) != 0);
            skipmask = __ballot(skip);
            c = 0; mc = 0; irregular = 0; anymiss = 0; nanany = 0; rowmiss = 0;
            sidx = 0; d = segs[0]; dn = segs[1];
        }
        load_item(spare);
        asm volatile("" :: "v"(cur.flag));               // the flag byte is the last load of its batch: the batch has arrived
        if (!live) return;
        unsigned char *cp = buf.c + (size_t)(slot0 + ti) * P * kStreamRow + lane;
#ifdef EVOFST_EXPERIMENTS
        if (p.debug & 4) { c += (uint32_t)cur.r[0].x ^ (uint32_t)cur.r[1].y ^ (uint32_t)cur.r[2].x ^ (uint32_t)cur.r[3].y; if (c == 0x12345u) { cp[0] = 1; } return; }
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
                        const uint32_t cs = skip ? 0u : c;               // a skipped site: no count, both ballots set -- excluded, and 0 in every sum
                        cp[pop * kStreamRow] = (unsigned char)cs;
                        const uint64_t z = __ballot(c == 0) | skipmask;
                        const uint64_t f = __ballot(c == nm) | skipmask;
                        const int pl = (int)pop;
                        zlo = evofst_write_lane((int)(uint32_t)z, pl, zlo); zhi = evofst_write_lane((int)(uint32_t)(z >> 32), pl, zhi);
                        flo = evofst_write_lane((int)(uint32_t)f, pl, flo); fhi = evofst_write_lane((int)(uint32_t)(f >> 32), pl, fhi);
                        // sites the integer walk cannot take: a NaN in the reference (nm < 2) or an excluded site with a
                        // dxy term (fixed with missing alleles).  Both need a missing allele unless n_full < 2.
                        const bool anym = mc != 0;
                        const bool rowm = __any(anym);
                        if (rowm) {                                      // rare: the chunk goes to the general walk, which reads HBM
                            // (as an instruction of its own: a store the compiler sees inside this loop turns every wait
                            // for a prefetched batch into "everything outstanding"; unseen, it can only make one such
                            // wait longer, and nothing in this kernel reads what it writes)
                            const uint16_t *dst16 = reinterpret_cast<const uint16_t *>(p.counts) + (((size_t)t * P + pop) * kTileSites + lane);
                            asm volatile("global_store_short %0, %1, off" :: "v"(dst16), "v"(cs | (nm << 8)) : "memory");
                            rowmiss = evofst_write_lane(1, (int)pop, rowmiss);
                        }
                        if (rowm || nfull < 2u) {
                            const uint32_t few = nm < 2u ? 1u : 0u;
                            irregular |= few | ((c == nm && nm < nfull) ? 1u : 0u);
                            nanany |= few;
                            anymiss |= anym ? 1u : 0u;
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
    if (n_own > 0) tile_done(slot0 + n_own - 1);
}

// ------------------------------------------------------------------------------------------------ consumer
// Look-back.  Three tables, all written by the producer that publishes a chunk (stream_publish) and read by the
// consumers of later chunks:
//   level 0  agg0[chunk / 2][pair][chunk & 1]  16-bit halves of a word, included sites of the chunk + 1 (0: not there
//            yet): a reader's 4-byte load brings two chunks;
//   level 1  agg1[group of 32 chunks][pair]: 32-bit words that every chunk of the group adds {1 << 26 | its included
//            sites} to (one atomic add): the sum is complete when the top six bits say 32 (a group holds at most
//            32 * 32 * 4096 < 2^26 sites) -- no chunk waits for another to form the
//            aggregate, and it is there as soon as the group's last chunk has been counted;
//   level 2  agg2[group of 32 groups][pair]: the same with a group's total as the contribution, added by a consumer of
//            the first chunk of the next group (stream_consume) -- 32 adds per word over a long time, where 1024 chunks
//            adding straight into one row queue up at the memory side.
// The rows before a chunk: chunks [grp0 * 32, chunk) of level 0, groups [grp1 * 32, grp0) of level 1, super-groups
// [0, grp1) of level 2 (all of them full groups).  Flag word r counts the chunks that have published into row r of
// {level 0 | level 1 | level 2} (rows 0.., l1_row.., l2_row..): a consumer polls those -- one or two 4-byte loads per
// lane and poll -- and reads the rows when the flags say they are worth reading.  The flags are raised without waiting
// for the row's stores or adds, so a row can lag its flag: every word read is checked and read again if need be.
struct StreamRows {
    int grp0, grp1, n0, n1, n2;
    __device__ __forceinline__ StreamRows(int chunk) {
        grp0 = chunk / kStreamFan; grp1 = grp0 / kStreamFan;
        n0 = chunk - grp0 * kStreamFan; n1 = grp0 - grp1 * kStreamFan; n2 = grp1;
    }
};
// Polls the flags until every row before the chunk has all its contributions; false when the wait gave up (error word
// set, here or elsewhere).
__device__ __forceinline__ bool stream_wait_flags(const StreamParams &p, const StreamRows &r, int lane, int par, uint32_t &polls) {
    constexpr int F = kStreamFan;
    const uint32_t *flags = p.flags[par];
    const int idx = lane < F ? (lane < r.n0 ? r.grp0 * F + lane : -1) : (lane - F < r.n1 ? p.l1_row + r.grp1 * F + lane - F : -1);
    const uint32_t want = lane < F ? 1u : (uint32_t)F;
    uint32_t spins = 0;
    for (;;) {
        bool ok = true;
        if (idx >= 0) ok = __hip_atomic_load((gu32 *)(flags + idx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == want;
        for (int r0 = 0; r0 < r.n2; r0 += 64)
            if (r0 + lane < r.n2) ok = ok && __hip_atomic_load((gu32 *)(flags + p.l2_row + r0 + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (uint32_t)F;
        if (__all(ok)) { polls = spins; return true; }
        if (++spins > kStreamSpinLimit) { __hip_atomic_store((gu32 *)p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
        if ((spins & 63u) == 0u && __hip_atomic_load((gu32 *)p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
        // polls are uncached loads of lines many waves want: a few per microsecond and CU slow the tile stream of every CU
        // down.  The consumer has most of an iteration to spare, so it asks again only every few microseconds.
        for (int z = 0; z < p.poll; ++z) __builtin_amdgcn_s_sleep(16);
    }
}
// Included sites of column q in the rows before the chunk.  False when a row stayed incomplete and the wait gave up.
__device__ __forceinline__ bool stream_sum_rows(const StreamParams &p, const StreamRows &r, int par, int q, uint64_t &before, uint32_t &retries) {
    constexpr int F = kStreamFan, B = 32;
    const uint32_t np = (uint32_t)p.n_pairs;
    uint32_t spins = 0;
    uint64_t sum = 0;
    {   // level 0: at most 31 rows, two to a word (the group starts at an even chunk)
        const uint32_t *a0 = p.agg0[par];                 // uniform base + 32-bit offsets: one address register per load
        const uint32_t o0 = (uint32_t)(r.grp0 * (F / 2)) * np + (uint32_t)q;
        const int nw = (r.n0 + 1) >> 1;
        uint32_t x[B / 2];
        while (r.n0 > 0) {
#pragma unroll
            for (int u = 0; u < B / 2; ++u) x[u] = __hip_atomic_load((gu32 *)(a0 + (o0 + (uint32_t)min(u, nw - 1) * np)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool ok = true;
#pragma unroll
            for (int u = 0; u < B / 2; ++u) ok = ok && (x[u] & 0xffffu) != 0u && (2 * min(u, nw - 1) + 1 >= r.n0 || (x[u] >> 16) != 0u);
            if (__all(ok)) break;
            if (++spins > kStreamSpinLimit) { __hip_atomic_store((gu32 *)p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
            if ((spins & 63u) == 0u && __hip_atomic_load((gu32 *)p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
#ifdef EVOFST_EXPERIMENTS
            if (p.debug & 512) __builtin_amdgcn_s_sleep(4); else
#endif
            __builtin_amdgcn_s_sleep(96);
        }
        if (r.n0 > 0) {
#pragma unroll
            for (int u = 0; u < B / 2; ++u) sum += (2 * u < r.n0 ? (x[u] & 0xffffu) - 1u : 0u) + (2 * u + 1 < r.n0 ? (x[u] >> 16) - 1u : 0u);
        }
    }
    // levels 1 and 2 as one list of 32-bit words {contributions << 26 | sites}, 16 in flight (the two tables lie one
    // behind the other: one base)
    constexpr int B2 = 16;
    const int total = r.n1 + r.n2;
    const uint32_t *a1 = p.agg1[par];
    const uint32_t o1 = (uint32_t)(r.grp1 * F) * np + (uint32_t)q, o2 = (uint32_t)(p.agg2[par] - p.agg1[par]) + (uint32_t)q;
    for (int r0 = 0; r0 < total; r0 += B2) {
        uint32_t x[B2];
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int u = 0; u < B2; ++u) {
                const int i = min(r0 + u, total - 1);
                const uint32_t off = i < r.n1 ? o1 + (uint32_t)i * np : o2 + (uint32_t)(i - r.n1) * np;
                x[u] = __hip_atomic_load((gu32 *)(a1 + off), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = ok && (x[u] >> kStreamAggShift) == (uint32_t)F;
            }
            if (__all(ok)) break;
            if (++spins > kStreamSpinLimit) { __hip_atomic_store((gu32 *)p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
            if ((spins & 63u) == 0u && __hip_atomic_load((gu32 *)p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
#ifdef EVOFST_EXPERIMENTS
            if (p.debug & 512) __builtin_amdgcn_s_sleep(4); else
#endif
            __builtin_amdgcn_s_sleep(96);
        }
#pragma unroll
        for (int u = 0; u < B2; ++u) sum += r0 + u < total ? (uint64_t)(x[u] & kStreamAggMask) : 0ull;
    }
    before = sum;
    retries = spins;
    return true;
}

// Per (tile, population) of a counted chunk: the sums of c and of c^2 over the tile's first 16, 32, 48 and 64 sites
// ([4] x {C, Q}).  Every consumer wave fills the whole table itself before it reads it (the waves write the same words).
__device__ __forceinline__ void stream_quarter_table(int lane, int nt, int P, const unsigned char *cpl, uint32_t *qtab) {
    constexpr uint32_t kOnes = 0x01010101u;
    for (int item = lane; item < nt * P; item += 64) {
        const uint4 *row = reinterpret_cast<const uint4 *>(cpl + (size_t)item * kStreamRow);
        uint32_t C = 0, Q = 0;
        uint32_t out[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint4 x = row[k];
            C = __builtin_amdgcn_udot4(x.x, kOnes, C, false); C = __builtin_amdgcn_udot4(x.y, kOnes, C, false);
            C = __builtin_amdgcn_udot4(x.z, kOnes, C, false); C = __builtin_amdgcn_udot4(x.w, kOnes, C, false);
            Q = __builtin_amdgcn_udot4(x.x, x.x, Q, false); Q = __builtin_amdgcn_udot4(x.y, x.y, Q, false);
            Q = __builtin_amdgcn_udot4(x.z, x.z, Q, false); Q = __builtin_amdgcn_udot4(x.w, x.w, Q, false);
            out[2 * k] = C; out[2 * k + 1] = Q;
        }
        uint4 *dst = reinterpret_cast<uint4 *>(qtab + (size_t)item * 8);
        dst[0] = make_uint4(out[0], out[1], out[2], out[3]);
        dst[1] = make_uint4(out[4], out[5], out[6], out[7]);
    }
}

// The statistics walk of one pair (this lane) over the nt tiles of a chunk whose alt counts and ballots lie in LDS:
// clean_walk_kernel's arithmetic (evofst_fused.hip), a tile of 64 sites per step.  Skipped sites hold count 0 and
// no allele is missing, so p = c / n with the population's full n, and the window sums of d^2, p(1-p) and sumKij follow
// exactly from integer sums of c_i c_j, c and c^2 (a chunk has at most 2^12 sites and c <= 255: every sum stays below
// 2^32 and every product below 2^53).  Per tile a lane forms the sum of c_i c_j (sixteen v_dot4_u32_u8, kept per
// quarter of 16 sites) and takes the sums of c and c^2 from the quarter table; at a window boundary the sums up to the
// boundary site are the quarter-table entry before its quarter plus masked dot products over that quarter alone.
__device__ __forceinline__ void stream_walk_pair(const StreamParams &p, const unsigned char *cpl, const uint64_t *zf, const uint32_t *qtab,
                                                 const uint32_t *pmask, int nt, int gtile0, int chunk, int q, uint32_t pi, uint32_t pj,
                                                 int64_t b0, int64_t g0) {
    constexpr int SUB = kTileSites;
    constexpr uint32_t kOnes = 0x01010101u;
    const int P = p.P;
    const uint32_t ni = (uint32_t)p.n_full[pi], nj = (uint32_t)p.n_full[pj];
    const int64_t first_seg = b0 / p.g;
    const double dni = (double)ni, dnj = (double)nj;
    const double ni2 = dni * dni, nj2 = dnj * dnj, ninj = dni * dnj;      // exact
    const double rE = 1.0 / (ninj * ninj);                                // d^2 = (c_i n_j - c_j n_i)^2 / (n_i n_j)^2
    const double rTi = 1.0 / ni2, rTj = 1.0 / nj2;                        // t = c (n - c) / n^2
    const double rHi = ni > 1 ? rTi / (double)(ni - 1) : 0.0, rHj = nj > 1 ? rTj / (double)(nj - 1) : 0.0;   // h = t / (n - 1)
    int64_t m = g0 / p.g;
    int32_t left = p.g - (int32_t)(g0 - m * p.g);
    const int32_t left0 = left;
    const int32_t row0 = (int32_t)(m - first_seg);
    size_t orow = (size_t)row0 * p.n_pairs + q;
    int32_t nflush = 0;
    uint32_t aCC = 0, aCi = 0, aCj = 0, aQi = 0, aQj = 0;                 // integer sums since the last boundary
    auto parts = [&](uint32_t iCC, uint32_t iCi, uint32_t iCj, uint32_t iQi, uint32_t iQj, double &num, double &den, double &k) {
        const double CC = (double)iCC, Ci = (double)iCi, Cj = (double)iCj, Qi = (double)iQi, Qj = (double)iQj;
        const double E = fma(nj2, Qi, fma(ni2, Qj, -2.0 * ninj * CC));    // sum (c_i n_j - c_j n_i)^2
        const double dTi = fma(dni, Ci, -Qi), dTj = fma(dnj, Cj, -Qj);    // sum c (n - c)
        const double dE = E * rE;
        num = dE - dTi * rHi - dTj * rHj;
        den = dE + dTi * rTi + dTj * rTj;
        k = fma(dnj, Ci, fma(dni, Cj, -2.0 * CC));                        // sumKij (fst.cpp:299-321)
    };
    for (int ti = 0; ti < nt; ++ti) {
        const uint64_t *zp = zf + (size_t)ti * 2 * P;
        const uint64_t incl = ~((zp[pi] & zp[pj]) | (zp[P + pi] & zp[P + pj]));   // fst.cpp:543-544
        int b = SUB;                                       // site of this lane's next boundary in the tile (SUB: none)
        {
            const int n = __popcll(incl);
            if (left <= n) b = nth_set_bit64((uint32_t)incl, (uint32_t)(incl >> 32), left);
            else left -= n;
        }
        const unsigned char *ri = cpl + ((size_t)ti * P + pi) * kStreamRow, *rj = cpl + ((size_t)ti * P + pj) * kStreamRow;
        uint32_t cq[4];                                    // sum of c_i c_j over the tile's quarters
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#ifdef EVOFST_EXPERIMENTS
            const uint4 x = (p.debug & 32) ? make_uint4(pi, pj, ti, k) : reinterpret_cast<const uint4 *>(ri)[k];
            const uint4 y = (p.debug & 32) ? make_uint4(pj, pi, k, ti) : reinterpret_cast<const uint4 *>(rj)[k];
#else
            const uint4 x = reinterpret_cast<const uint4 *>(ri)[k], y = reinterpret_cast<const uint4 *>(rj)[k];
#endif
            uint32_t v = __builtin_amdgcn_udot4(x.x, y.x, 0u, false);
            v = __builtin_amdgcn_udot4(x.y, y.y, v, false); v = __builtin_amdgcn_udot4(x.z, y.z, v, false);
            cq[k] = __builtin_amdgcn_udot4(x.w, y.w, v, false);
        }
        const uint2 *qi = reinterpret_cast<const uint2 *>(qtab + ((size_t)ti * P + pi) * 8), *qj = reinterpret_cast<const uint2 *>(qtab + ((size_t)ti * P + pj) * 8);
        const uint2 Ti = qi[3], Tj = qj[3];                // {sum c, sum c^2} over the tile
        const uint32_t tCC = (cq[0] + cq[1]) + (cq[2] + cq[3]);
        uint32_t dCC = 0, dCi = 0, dCj = 0, dQi = 0, dQj = 0;     // of the tile's sums, the part up to the last boundary
        while (b < SUB) {
            const int qb = b >> 4;                         // the boundary's quarter: the quarters before it come whole
            const uint2 Bi = qi[max(qb - 1, 0)], Bj = qj[max(qb - 1, 0)];
            uint32_t pCi = qb ? Bi.x : 0u, pQi = qb ? Bi.y : 0u, pCj = qb ? Bj.x : 0u, pQj = qb ? Bj.y : 0u;
            uint32_t pCC = qb == 0 ? 0u : (qb == 1 ? cq[0] : (qb == 2 ? cq[0] + cq[1] : cq[0] + cq[1] + cq[2]));
            const uint4 mk = reinterpret_cast<const uint4 *>(pmask)[b & 15];     // bytes of sites 0 .. b & 15 of a quarter
            uint4 x = reinterpret_cast<const uint4 *>(ri)[qb], y = reinterpret_cast<const uint4 *>(rj)[qb];
            x.x &= mk.x; x.y &= mk.y; x.z &= mk.z; x.w &= mk.w; y.x &= mk.x; y.y &= mk.y; y.z &= mk.z; y.w &= mk.w;
            pCC = __builtin_amdgcn_udot4(x.x, y.x, pCC, false); pCC = __builtin_amdgcn_udot4(x.y, y.y, pCC, false);
            pCC = __builtin_amdgcn_udot4(x.z, y.z, pCC, false); pCC = __builtin_amdgcn_udot4(x.w, y.w, pCC, false);
            pCi = __builtin_amdgcn_udot4(x.x, kOnes, pCi, false); pCi = __builtin_amdgcn_udot4(x.y, kOnes, pCi, false);
            pCi = __builtin_amdgcn_udot4(x.z, kOnes, pCi, false); pCi = __builtin_amdgcn_udot4(x.w, kOnes, pCi, false);
            pCj = __builtin_amdgcn_udot4(y.x, kOnes, pCj, false); pCj = __builtin_amdgcn_udot4(y.y, kOnes, pCj, false);
            pCj = __builtin_amdgcn_udot4(y.z, kOnes, pCj, false); pCj = __builtin_amdgcn_udot4(y.w, kOnes, pCj, false);
            pQi = __builtin_amdgcn_udot4(x.x, x.x, pQi, false); pQi = __builtin_amdgcn_udot4(x.y, x.y, pQi, false);
            pQi = __builtin_amdgcn_udot4(x.z, x.z, pQi, false); pQi = __builtin_amdgcn_udot4(x.w, x.w, pQi, false);
            pQj = __builtin_amdgcn_udot4(y.x, y.x, pQj, false); pQj = __builtin_amdgcn_udot4(y.y, y.y, pQj, false);
            pQj = __builtin_amdgcn_udot4(y.z, y.z, pQj, false); pQj = __builtin_amdgcn_udot4(y.w, y.w, pQj, false);
            SegRow row;
#ifdef EVOFST_EXPERIMENTS
            if (p.debug & 8192) { row.num = 1.0; row.den = 2.0; row.k = 3.0; } else   // (timing: rows stored, not computed)
#endif
            parts(aCC + (pCC - dCC), aCi + (pCi - dCi), aCj + (pCj - dCj), aQi + (pQi - dQi), aQj + (pQj - dQj), row.num, row.den, row.k);
            row.end = (gtile0 + ti) * kTileSites + b; row.pad = 0;
#ifdef EVOFST_EXPERIMENTS
            if (p.debug & 4096) { asm volatile("" :: "v"(row.num), "v"(row.den), "v"(row.k)); } else   // (timing: rows computed, not stored)
            if ((p.debug & 16) && row.end != -12345) {} else
#endif
            if ((uint64_t)((int64_t)row0 + nflush) >= (uint64_t)p.seg_cap) {   // never with a sound prefix; no store outside the table whatever the look-back handed over
                __hip_atomic_store((gu32 *)p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {   // two 16-byte stores into one 32-byte sector
                uint4 *dst = reinterpret_cast<uint4 *>(p.seg_rows + orow);
                const uint4 *src = reinterpret_cast<const uint4 *>(&row);
                dst[0] = src[0]; dst[1] = src[1];
            }
            aCC = aCi = aCj = aQi = aQj = 0;
            dCC = pCC; dCi = pCi; dCj = pCj; dQi = pQi; dQj = pQj;
            ++nflush;
            orow += p.n_pairs;
            const uint64_t rem = b >= SUB - 1 ? 0ull : incl & ~((2ull << b) - 1ull);
            const int c2 = __popcll(rem);
            if (p.g <= c2) b = nth_set_bit64((uint32_t)rem, (uint32_t)(rem >> 32), p.g);
            else { left = p.g - c2; b = SUB; }
        }
        aCC += tCC - dCC; aCi += Ti.x - dCi; aCj += Tj.x - dCj; aQi += Ti.y - dQi; aQj += Tj.y - dQj;
    }
    // what the chunk holds behind its last boundary
    const size_t cq_ = (size_t)chunk * p.n_pairs + q;
    const size_t plane = (size_t)p.n_pairs * p.n_chunks;
    double num, den, k;
    parts(aCC, aCi, aCj, aQi, aQj, num, den, k);
    p.tail[cq_] = num; p.tail[plane + cq_] = den; p.tail[2 * plane + cq_] = k;
    p.tail_n[cq_] = make_int4((nflush ? p.g : left0) - left, nflush, row0, 0);   // {sites summed in the tail, segments completed, row of the first}
}

// The census of a producer's tiles (included sites per pair, filter fst.cpp:543-544, lane = pair), added to the
// workgroup's census of the chunk in LDS.
__device__ __forceinline__ void stream_census(const StreamParams &p, int lane, int slot0, int n_own, const uint64_t *zf, const uint32_t *lpairs, uint32_t *cen) {
    const int P = p.P;
    const int NG = (p.n_pairs + 63) >> 6;
    if (n_own > 0)
        for (int g = 0; g < NG; ++g) {
            const int q = g * 64 + lane;
            if (q < p.n_pairs) {
                const uint32_t ij = lpairs[q];
                const uint32_t pi = ij & 0xffffu, pj = ij >> 16;
                uint32_t n = 0;
                for (int ti = 0; ti < n_own; ++ti) {
                    const uint64_t *zp = zf + (size_t)(slot0 + ti) * 2 * P;
                    const uint64_t ex = (zp[pi] & zp[pj]) | (zp[P + pi] & zp[P + pj]);
                    n += 64u - (uint32_t)__popcll(ex);
                }
                __hip_atomic_fetch_add(cen + q, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
}

// The producer that arrives last at a chunk publishes the chunk's row -- straight after the counting, whatever the
// consumers are busy with.  The flag is raised without waiting for the row's stores: it only tells pollers when reading
// the row is worth it.  `pub` = uses of the buffer whose census has been read for publishing: what the consumers wait
// for, since they clear the census.
__device__ __forceinline__ void stream_publish(const StreamParams &p, int lane, int par, int chunk, const uint32_t *cen, uint32_t *pub, int use) {
    constexpr int F = kStreamFan;
    const int NG = (p.n_pairs + 63) >> 6;
#ifdef EVOFST_EXPERIMENTS
    if (p.debug & 256) {                                   // nothing published at all (timing only, with bit 2)
        if (lane == 0) __hip_atomic_store(pub, (uint32_t)use + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        return;
    }
#endif
    const size_t np = (size_t)p.n_pairs;
    const int grp0 = chunk / F, grp1 = grp0 / F;
    uint16_t *row0 = reinterpret_cast<uint16_t *>(p.agg0[par] + (size_t)(chunk >> 1) * np) + (chunk & 1);   // column q: [2 * q]
    uint16_t *orow0 = reinterpret_cast<uint16_t *>(p.agg0[par ^ 1] + (size_t)(chunk >> 1) * np) + (chunk & 1);
    uint32_t *row1 = p.agg1[par] + (size_t)grp0 * np;
    for (int g = 0; g < NG; ++g) {
        const int q = g * 64 + lane;
        if (q < p.n_pairs) {
            const uint32_t n = cen[q];
            __hip_atomic_store((gu16 *)(row0 + 2 * q), (uint16_t)(n + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t one = (1u << kStreamAggShift) | n;   // {one more chunk, its included sites}
#ifdef EVOFST_EXPERIMENTS
            if (!(p.debug & 128))
#endif
            __hip_atomic_fetch_add((gu32 *)(row1 + q), one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // the other copy's rows, for the next launch
            orow0[2 * q] = 0;
            if (chunk % F == 0) p.agg1[par ^ 1][(size_t)grp0 * np + q] = 0u;
            if (chunk % (F * F) == 0) p.agg2[par ^ 1][(size_t)grp1 * np + q] = 0u;
        }
    }
    // (the release makes this wave's reads of the census complete before the consumers may clear it)
    if (lane == 0) __hip_atomic_store(pub, (uint32_t)use + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (lane == 0) {
        uint32_t *flags = p.flags[par], *oflags = p.flags[par ^ 1];
        __hip_atomic_fetch_add((gu32 *)(flags + chunk), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add((gu32 *)(flags + p.l1_row + grp0), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        oflags[chunk] = 0u;
        if (chunk % F == 0) oflags[p.l1_row + grp0] = 0u;
        if (chunk % (F * F) == 0) oflags[p.l2_row + grp1] = 0u;
    }
}

// Everything a consumer wave does with a counted chunk.  Returns false when a look-back wait gave up.
// Consumer wave `cw` of `ncw` takes the groups of 64 pairs cw, cw + ncw, ...; the first one also hands a chunk the integer
// walk cannot take to the general walk.
// `lready` = chunks of this workgroup whose look-back flags consumer 0 has seen complete; `dead` != 0: some wave gave up.
__device__ __forceinline__ bool stream_consume(const StreamParams &p, int it, int lane, int cw, int ncw, int par, int chunk, const StreamBuf &buf,
                                               const uint32_t *lpairs, uint32_t *lcen, int32_t *lready, const int32_t *dead, const uint32_t *pmask) {
    uint32_t *qtab = buf.qt;
    const uint64_t *zf = buf.zf, *flg = buf.flg;
    const int NG = (p.n_pairs + 63) >> 6;
    const int my_groups = cw < NG ? (NG - cw + ncw - 1) / ncw : 0;
    const int P = p.P;
    const int gtile0 = chunk * p.chunk_tiles;
    const int nt = min(p.chunk_tiles, p.n_tiles - gtile0);
    // an allele missing, a population with fewer than two chromosomes: the general walk's chunk.  (Skipped sites -- the
    // packer's filters, the sites behind the shard's end in its last tile -- stay here: their counts were left out of the
    // count plane and both of their ballots are set, so they are excluded and add exactly 0 to every integer sum.)
    uint64_t fw = 0;
    if (lane < nt) fw = flg[lane * kStreamFlg + 0] | flg[lane * kStreamFlg + 2];
    const bool dirty = __any(fw != 0);
    if (my_groups == 0) return true;
    // ---- the rows before this chunk: wait until their flags say they are complete, read them once per group of 64 pairs;
    // then the walk (or the chunk's offsets for the general walk)
    const StreamRows rows(chunk);
    uint32_t polls = 0, retries = 0;
    if (!dirty) {   // (needs nothing from other chunks: done before the wait)
        stream_quarter_table(lane, nt, P, buf.c, qtab);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // The first chunk of a group of 32 hands the total of the group before it to level 2 -- before it waits for level 2
    // itself, which its own hand-over may complete (one add per group and row: no contention there).
#ifdef EVOFST_EXPERIMENTS
    if (!(p.debug & 2))
#endif
    if (chunk % kStreamFan == 0 && chunk > 0) {
        const int gprev = rows.grp0 - 1;
        const uint32_t np = (uint32_t)p.n_pairs;
        uint32_t spins = 0;
#pragma unroll 1
        for (int g = cw; g < NG; g += ncw) {
            const int q = g * 64 + lane;
            const uint32_t *src = p.agg1[par] + ((uint32_t)gprev * np + (uint32_t)(q < p.n_pairs ? q : 0));
            uint32_t v;
            for (;;) {
                v = __hip_atomic_load((gu32 *)src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all((v >> kStreamAggShift) == (uint32_t)kStreamFan)) break;
                if (++spins > kStreamSpinLimit) { __hip_atomic_store((gu32 *)p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
                if ((spins & 63u) == 0u && __hip_atomic_load((gu32 *)p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
                __builtin_amdgcn_s_sleep(8);
            }
            if (q < p.n_pairs)
                __hip_atomic_fetch_add((gu32 *)(p.agg2[par] + ((uint32_t)(gprev / kStreamFan) * np + (uint32_t)q)), (1u << kStreamAggShift) | (v & kStreamAggMask),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (cw == 0 && lane == 0) __hip_atomic_fetch_add((gu32 *)(p.flags[par] + p.l2_row + gprev / kStreamFan), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#ifdef EVOFST_EXPERIMENTS
    if (!(p.debug & 2))
#endif
    {   // one consumer wave of the workgroup polls the flags; the others wait for its word in LDS
        if (cw == 0) {
            if (!stream_wait_flags(p, rows, lane, par, polls)) return false;       // (the caller raises `dead`)
            if (lane == 0) __hip_atomic_store(lready, it + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            uint32_t spins = 0;
            for (;;) {
                const int32_t v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(lready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (v > it) break;
                if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != 0 ||
                    ++spins > (kStreamSpinLimit << 4)) return false;
                __builtin_amdgcn_s_sleep(8);
            }
        }
    }
    if (cw == 0) STREAM_STAMP(p, it, 5);                  // flags complete
#pragma unroll 1
    for (int g = cw; g < NG; g += ncw) {
        const int q = g * 64 + lane;
        uint64_t before = 0;
#ifdef EVOFST_EXPERIMENTS
        if (!(p.debug & (2 | 2048)))                       // (2048: flags waited for, rows not read -- prefix 0, rows stay inside the table)
#endif
        if (!stream_sum_rows(p, rows, par, q < p.n_pairs ? q : 0, before, retries)) return false;
        if (cw == 0 && g == 0) {
            STREAM_STAMP(p, it, 6);     // first sums there
#ifdef EVOFST_EXPERIMENTS
            if (p.stamps != nullptr && it < kStreamStampIters && lane == 0) p.stamps[((size_t)blockIdx.x * kStreamStampIters + it) * 16 + 10] = ((unsigned long long)polls << 32) | retries;
#endif
        }
        if (q < p.n_pairs) {
            const uint32_t ij = lpairs[q];
            const uint32_t pi = ij & 0xffffu, pj = ij >> 16;
            const uint32_t cen = lcen[q];
            lcen[q] = 0u;                                  // for the chunk that takes this buffer next
            const int64_t b0 = p.base[q];
            if (chunk == p.n_chunks - 1) p.incl_totals[q] = (int64_t)before + (int64_t)cen;
            if (chunk == 0) p.first_segment[q] = b0 / p.g;
            const int64_t g0 = b0 + (int64_t)before;
            if (dirty) p.chunk_offsets[(size_t)q * p.n_chunks + chunk] = g0;
#ifdef EVOFST_EXPERIMENTS
            else if (p.debug & 1) {}
#endif
            else stream_walk_pair(p, buf.c, zf, qtab, pmask, nt, gtile0, chunk, q, pi, pj, b0, g0);
        }
    }
    if (cw == 0) STREAM_STAMP(p, it, 8);                  // walked
    if (cw != 0) return true;
    if (dirty) {
        // K1's outputs for this chunk: packed counts [tile][P][64] (the rows without a missing allele: the producers
        // wrote the others), ballots [tile][z|f][P], the four site masks
        const int nv = nt * P * 4;                         // groups of 16 sites: 16 alt bytes + n_full -> 16 packed counts
        uint4 *dst = reinterpret_cast<uint4 *>(reinterpret_cast<uint16_t *>(p.counts) + (size_t)gtile0 * P * kTileSites);
        for (int v = lane; v < nv; v += 64) {
            const int row = v >> 2, part = v & 3;          // row = slot * P + population
            const int slot = row / P, pop = row - slot * P;
            if ((flg[slot * kStreamFlg + 4] >> pop) & 1ull) continue;
            const uint4 c16 = *reinterpret_cast<const uint4 *>(buf.c + (size_t)row * kStreamRow + part * 16);
            const uint32_t nn = (uint32_t)p.n_full[pop] * 0x01010101u;
            const uint4 n16 = make_uint4(nn, nn, nn, nn);
            dst[2 * v] = make_uint4(__builtin_amdgcn_perm(n16.x, c16.x, 0x05010400u), __builtin_amdgcn_perm(n16.x, c16.x, 0x07030602u),
                                    __builtin_amdgcn_perm(n16.y, c16.y, 0x05010400u), __builtin_amdgcn_perm(n16.y, c16.y, 0x07030602u));
            dst[2 * v + 1] = make_uint4(__builtin_amdgcn_perm(n16.z, c16.z, 0x05010400u), __builtin_amdgcn_perm(n16.z, c16.z, 0x07030602u),
                                        __builtin_amdgcn_perm(n16.w, c16.w, 0x05010400u), __builtin_amdgcn_perm(n16.w, c16.w, 0x07030602u));
        }
        for (int i = lane; i < nt * 2 * P; i += 64) p.zf_masks[(size_t)gtile0 * 2 * P + i] = zf[i];
        if (lane < nt) {
            p.irr_mask[gtile0 + lane] = flg[lane * kStreamFlg + 0]; p.skip_mask[gtile0 + lane] = flg[lane * kStreamFlg + 1];
            p.miss_mask[gtile0 + lane] = flg[lane * kStreamFlg + 2]; p.nan_mask[gtile0 + lane] = flg[lane * kStreamFlg + 3];
        }
    }
    if (lane == 0) p.chunk_dirty[chunk] = dirty ? 1u : 0u;
    if (lane == 0 && dirty) __hip_atomic_fetch_add((gu32 *)p.n_dirty, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

__global__ __launch_bounds__(kStreamMaxThreads, EVOFST_STREAM_WPS) void stream_pairs_kernel(StreamParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NP = p.n_prod, NC = p.n_cons;                                    // waves 0 .. NP-1 count, NP .. NP+NC-1 consume
    const bool consumer = wave >= NP;
    const int nthreads = 64 * (NP + NC);
    uint32_t *pmask = reinterpret_cast<uint32_t *>(smem);                      // [16][4] byte masks of sites 0 .. r of a quarter of 16 sites
    // the workgroup's words: [k & 7] ticket of its chunk k; [8] a wait gave up; [9] granule copy of this launch; [10], [11] lready
    // of consumer set 0 / 1; [24] chunks drawn; [12 + b] producers arrived at buffer b; [16 + b] uses of buffer b published; [20 + b] consumers
    // done with buffer b (all counters run on through the launch)
    int32_t *lmisc = reinterpret_cast<int32_t *>(smem + 1024);
    uint32_t *lpairs = reinterpret_cast<uint32_t *>(smem + 1024 + 128);
    const int NB = p.n_buf;
    const size_t pw = s16((size_t)p.n_pairs * 4) / 4;                          // words of one per-pair array
    uint32_t *lcen = lpairs + pw;                                              // [NB][pw] census of the chunk in buffer b (the producers add, the consumers read and clear)
    int32_t *ldead = lmisc + 8, *lready = lmisc + 10;
    uint32_t *ldrawn = reinterpret_cast<uint32_t *>(lmisc + 24), *ldone = reinterpret_cast<uint32_t *>(lmisc + 12),
             *lpub = reinterpret_cast<uint32_t *>(lmisc + 16), *lfree = reinterpret_cast<uint32_t *>(lmisc + 20);
    unsigned char *bufs = reinterpret_cast<unsigned char *>(lpairs) + (size_t)(1 + NB) * s16((size_t)p.n_pairs * 4);
    const size_t buf_bytes = stream_buf_bytes(p.P, p.chunk_tiles);
    const SegStream segs = (SegStream)(uintptr_t)p.segs;
    const int n_chunks = p.n_chunks;
    const uint32_t last_draw = (uint32_t)n_chunks + gridDim.x - 1u;            // every workgroup draws until its ticket is past the end
    auto draw = [&]() -> int32_t {
        const uint32_t t = __hip_atomic_fetch_add((gu32 *)p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == last_draw) {   // nobody draws after this one: the counter goes back to zero and the launch counter on
            __hip_atomic_store((gu32 *)p.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add((gu32 *)p.launches, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return (int32_t)min(t, (uint32_t)n_chunks);
    };
    // which copy of the granule arrays this launch uses: the launch counter changes only with the launch's last draw,
    // which comes after every workgroup's first, and a workgroup reads the counter before its first draw
    if (tid == 0) lmisc[9] = (int32_t)(__hip_atomic_load((gu32 *)p.launches, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u);
    for (int i = tid; i < 16 * 4; i += nthreads) {
        const int b = i >> 2, m = i & 3, nbv = b + 1 - 4 * m;
        pmask[i] = nbv >= 4 ? 0xffffffffu : (nbv <= 0 ? 0u : ((1u << (8 * nbv)) - 1u));
    }
    for (int i = tid; i < p.n_pairs; i += nthreads) lpairs[i] = p.pairs[i];
    for (int i = tid; i < NB * (int)pw; i += nthreads) lcen[i] = 0u;
    if (tid == 0) {   // after the launch counter has been read (same thread, above)
        for (int i = 10; i < 25; ++i) lmisc[i] = 0;
        const int32_t t = draw();
        lmisc[0] = lmisc[1] = lmisc[2] = lmisc[3] = t;
        *ldrawn = t >= n_chunks ? 4u : 1u;
        *ldead = 0;
    }
    __syncthreads();
    const int par = __builtin_amdgcn_readfirstlane(lmisc[9]);   // (what comes out of LDS is wave-uniform here; the compiler is told so)
    auto is_dead = [&]() -> bool { return __builtin_amdgcn_readfirstlane(__hip_atomic_load(ldead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != 0; };
    // waits until an LDS counter has reached `want`; false when some wave of the workgroup has given up
    auto wait_word = [&](const uint32_t *w, uint32_t want) -> bool {
        for (uint32_t spins = 0;; ++spins) {
            const uint32_t v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(w, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
            if (v >= want) return true;
            if (is_dead()) return false;
            if (spins > (kStreamSpinLimit << 5)) {             // seconds: never on a healthy launch
                if (lane == 0) {
                    __hip_atomic_store(ldead, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_store((gu32 *)p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                return false;
            }
            __builtin_amdgcn_s_sleep(4);
        }
    };

    // No barrier from here on.  Chunk k of this workgroup (ticket c_k) lives in buffer k mod NB:
    //   a producer  counts its tiles of c_k as soon as every consumer is done with c_(k-NB), whatever the other
    //               producers are doing; the producer that finishes c_k first draws the ticket of c_(k+1);
    //   a consumer  takes c_k once the producer that finished last has published its row.
    // So a wave only ever waits for a wave that is ahead of it in this order, and the ticket ring of eight cannot be
    // overrun: slot (k + 1) & 7 is written when every wave is past c_(k-NB), NB <= 4.
    if (!consumer) {
        for (int k = 0, kb = 0, use = 0;; ++k, kb = kb + 1 == NB ? 0 : kb + 1, use += kb == 0 ? 1 : 0) {
            if (wave == 0) STREAM_STAMP(p, k, 0);             // iteration begins (producer 0)
            if (!wait_word(ldrawn, (uint32_t)k + 1u)) break;
            const int32_t ci = __builtin_amdgcn_readfirstlane(lmisc[k & 7]);
            if (ci >= n_chunks) break;
            const StreamBuf bk = stream_buf(bufs + (size_t)kb * buf_bytes, p.P, p.chunk_tiles);
            if (use > 0 && !wait_word(lfree + kb, (uint32_t)(NC / p.n_sets * use))) break;
            if (wave == 0) STREAM_STAMP(p, k, 1);             // buffer free
            const int slot0 = wave < 4 ? wave * p.tr : min(NP, 4) * p.tr + (wave - 4) * p.tr2;
            const int t0 = __builtin_amdgcn_readfirstlane(ci * p.chunk_tiles + slot0);
            const int n_own = __builtin_amdgcn_readfirstlane(min(wave < 4 ? p.tr : p.tr2, p.n_tiles - t0));
#ifdef EVOFST_EXPERIMENTS
            const unsigned long long tc0 = __builtin_amdgcn_s_memrealtime();
#endif
            if (n_own > 0) stream_count_piece(p, lane, t0, n_own, slot0, bk, segs);
            if (wave == 0) STREAM_STAMP(p, k, 2);             // counted
#ifdef EVOFST_EXPERIMENTS
            if (p.stamps && k < kStreamStampIters && lane == 0 && wave < 8)   // every producer's counting time, 10 ns units, 16 bits each
                reinterpret_cast<unsigned short *>(p.stamps + ((size_t)blockIdx.x * kStreamStampIters + k) * 16 + 14)[wave] =
                    (unsigned short)min(__builtin_amdgcn_s_memrealtime() - tc0, 65535ull);
#endif
            stream_census(p, lane, slot0, n_own, bk.zf, lpairs, lcen + (size_t)kb * pw);
            // arrival: the first producer there draws the next ticket (the others are still counting, nobody waits for
            // the round trip, and a chunk is counted straight after its ticket is drawn: chunks with smaller tickets
            // are never far behind); the last one publishes -- unless it is also the first, then it publishes first
            uint32_t old = 0;
            if (lane == 0) old = __hip_atomic_fetch_add(ldone + kb, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
            old = __builtin_amdgcn_readfirstlane(old) - (uint32_t)(NP * use);
#ifdef EVOFST_EXPERIMENTS
            if (!(p.debug & 8))
#endif
            if (old == (uint32_t)NP - 1u) { STREAM_STAMP(p, k, 11); stream_publish(p, lane, par, ci, lcen + (size_t)kb * pw, lpub + kb, use); STREAM_STAMP(p, k, 13); }
            if (old == 0u) STREAM_STAMP(p, k, 12);
            if (old == 0u) {
                // ... once the next chunk's buffer is free: a ticket held through a wait would be a chunk that every
                // later chunk of the shard waits for
                const int kb1 = kb + 1 == NB ? 0 : kb + 1, use1 = use + (kb1 == 0 ? 1 : 0);
                const bool live = use1 == 0 || wait_word(lfree + kb1, (uint32_t)(NC / p.n_sets * use1));
                if (lane == 0) {
                const int32_t t = live ? draw() : n_chunks;
                lmisc[(k + 1) & 7] = t;
                if (t >= n_chunks) lmisc[(k + 2) & 7] = lmisc[(k + 3) & 7] = lmisc[(k + 4) & 7] = t;   // the end, for every set of consumers
                __hip_atomic_store(ldrawn, (uint32_t)k + (t >= n_chunks ? 5u : 2u), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    } else {
        // The consumers are the youngest waves of their workgroup and would get the issue slots the producers leave;
        // the producers mostly wait for memory, so the consumers go first.
#ifdef EVOFST_EXPERIMENTS
        if (!(p.debug & 64)) __builtin_amdgcn_s_setprio(3);
#else
        __builtin_amdgcn_s_setprio(3);
#endif
        // NS sets of consumer waves take the workgroup's chunks in turn: a set has NS chunk times for the wait on its chunk's
        // predecessors plus the walk
        const int NS = p.n_sets, ncw = NC / NS;
        const int set = (wave - NP) / ncw, cw = (wave - NP) - set * ncw;
        for (int k = set;; k += NS) {
            const int use = k / NB, kb = k - use * NB;
            if (!wait_word(ldrawn, (uint32_t)k + 1u)) break;
            const int32_t ck = __builtin_amdgcn_readfirstlane(lmisc[k & 7]);
            if (ck >= n_chunks) break;
            const StreamBuf bk = stream_buf(bufs + (size_t)kb * buf_bytes, p.P, p.chunk_tiles);
            if (wave == NP) STREAM_STAMP(p, k, 3);            // consumer 0 begins
#ifdef EVOFST_EXPERIMENTS
            if (p.debug & 8) {                                // no consumer work: the buffer is free as soon as it is counted
                if (!wait_word(ldone + kb, (uint32_t)(NP * (use + 1)))) break;
                if (lane == 0) __hip_atomic_fetch_add(lfree + kb, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                continue;
            }
#endif
            if (!wait_word(lpub + kb, (uint32_t)use + 1u)) break;
            if (wave == NP) STREAM_STAMP(p, k, 4);            // chunk counted and published
            const bool ok = stream_consume(p, k, lane, cw, ncw, par, ck, bk, lpairs, lcen + (size_t)kb * pw, lready + set, ldead, pmask);
            if (!ok) {                                        // whatever made this wave give up: every workgroup stops, the host is told
                if (lane == 0) {
                    __hip_atomic_store(ldead, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_store((gu32 *)p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                break;
            }
            if (lane == 0) __hip_atomic_fetch_add(lfree + kb, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (wave == NP) STREAM_STAMP(p, k, 9);            // consumer 0 done with the chunk
        }
    }
}

hipError_t launch_stream(const StreamParams &p, int max_blocks, hipStream_t st) {
    if (p.n_chunks <= 0 || p.n_pairs <= 0) return hipSuccess;
    if (p.n_pairs > kStreamMaxPairs || p.P > 64 || p.tr <= 0 || p.tr2 <= 0 || p.tr2 > p.tr || p.n_prod < 1 || p.n_cons < 1 || p.n_prod + p.n_cons > kStreamMaxWaves ||
        p.chunk_tiles != p.tr * std::min(p.n_prod, 4) + p.tr2 * std::max(p.n_prod - 4, 0) || p.chunk_tiles > 64 || p.n_buf < 2 || p.n_buf > kStreamMaxBufs ||
        p.n_sets < 1 || p.n_sets > 2 || p.n_cons % p.n_sets != 0) return hipErrorInvalidValue;
    const int threads = 64 * (p.n_prod + p.n_cons);
    const void *fn = reinterpret_cast<const void *>(&stream_pairs_kernel);
    const size_t lds = stream_lds_bytes(p.P, p.n_pairs, p.chunk_tiles, p.n_buf);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    int grid = max_blocks;
    if (grid <= 0) {
        // workgroups the device holds at once; asked once per shape (the query costs more than a launch).  Any grid is
        // safe -- chunks go by ticket -- this one just leaves no workgroup waiting for a slot.
        static thread_local struct { size_t lds; int threads; int dev; int grid; } memo = {0, 0, -1, 0};
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (memo.lds != lds || memo.threads != threads || memo.dev != dev) {
            int per_cu = 0, cus = 0;
            if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, lds)) != hipSuccess) return e;
            if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
            memo = {lds, threads, dev, std::max(1, per_cu) * std::max(1, cus)};
        }
        grid = memo.grid;
    }
    grid = std::max(1, std::min(grid, p.n_chunks));
    hipLaunchKernelGGL(stream_pairs_kernel, dim3(grid), dim3(threads), lds, st, p);
    return hipGetLastError();
}

}  // namespace evofst
