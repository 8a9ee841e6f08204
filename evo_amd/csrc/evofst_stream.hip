// evofst_stream.hip -- pairs mode as ONE streaming kernel with producer and consumer waves (gfx950, wave64).
//
// Replaces, for every population pair at once, the per-line work of getFstFromVCF: the counting loops
// (process_vcf_fst.cpp:243-280), the site filter (:543-544), calculateFstNumerator / Denominator
// (process_vcf_fst.h:19-30), calculateDxy (process_vcf_fst.cpp:297-323) and the SNP windows (:638-650).
//
// A workgroup is four PRODUCER waves and one CONSUMER wave and works through chunks of a few tiles in ticket order
// (one atomic per chunk).  In iteration i
//   the producers  count chunk c_i: lane = site, 16-byte coalesced loads of the tile rows (three batches of four
//                  rows requested ahead of the one being counted), popcounts under the wave-uniform masks of the
//                  layout's descriptor stream (scalar cache), packed {alt, non-missing} counts and the zero / fixed
//                  ballots of every population INTO LDS buffer i & 1 -- nothing but the tiles is read from HBM and no
//                  count is written to it;
//   the consumer   takes chunk c_(i-1) from LDS buffer (i-1) & 1: census of included sites per pair (filter :543-544,
//                  lane = pair), published as one 4-byte granule per pair; look-back over the chunks before it (three
//                  levels of fan-out 32: chunks of its group, groups of its super-group, super-groups); then the
//                  statistics walk over the chunk's counts in LDS (exact integer sums, four sites per
//                  v_dot4_u32_u8, segment rows at the window boundaries) -- the arithmetic of clean_walk_kernel.
// One workgroup barrier per iteration.  The producers never wait for another workgroup, and a consumer only for
// chunks with smaller tickets, whose producers are already running: whatever the dispatch order or the number of
// workgroups resident, the smallest unpublished chunk is always being counted by somebody who waits for nothing.
// While a consumer waits for its predecessors' granules the producers beside it keep the tile loads going, which
// is what the lock-step one-kernel pass of evofst_fused.hip could not do.
//
// A chunk with a skipped site or a missing allele (the walk above cannot take it) leaves its packed counts, ballots
// and its pairs' offsets in HBM, exactly as K1 + scan would have, and is walked by fused_pairs_kernel's HBM mode in a
// second launch that finds nothing to do on clean data.  stitch_kernel completes the segments that straddle chunks.
//
// Granules carry no launch tag: a granule is value + 1, and 0 means "not there yet".  The granule arrays exist
// twice; launch L uses copy L & 1 and zeroes, chunk by chunk, the rows of the other copy for launch L + 1.  The launch
// counter is a device word that the launch's last ticket draw bumps, so a replayed launch (HIP graph) alternates too.
#include "evofst_internal.h"

#include <algorithm>

namespace evofst {

typedef uint32_t SegWords __attribute__((ext_vector_type(4)));                  // {mask lo, mask hi, pop, flags | n_full << 8}
typedef const __attribute__((address_space(4))) SegWords *SegStream;
extern "C" __device__ int evofst_write_lane(int val, int lane, int old) __asm("llvm.amdgcn.writelane.i32");

typedef __attribute__((address_space(1))) uint32_t gu32;

constexpr uint32_t kStreamSpinLimit = 1u << 20;   // polls (with a sleep each) before a wait gives up: seconds

static __host__ __device__ constexpr size_t s16(size_t x) { return (x + 15) & ~(size_t)15; }

size_t stream_buffer_bytes(int P, int chunk_tiles) {
    return s16((size_t)chunk_tiles * P * kStreamRow * 2) + s16((size_t)chunk_tiles * 2 * P * 8) + s16((size_t)chunk_tiles * 4 * 8);
}
size_t stream_lds_bytes(int P, int n_pairs, int chunk_tiles) {
    return 1024 + 64 + 2 * s16((size_t)n_pairs * 4) + 2 * stream_buffer_bytes(P, chunk_tiles);
}

// ------------------------------------------------------------------------------------------------ producer
// Counts tiles [t0, t0 + n_own) of the shard into slots slot0, slot0 + 1, ... of an LDS buffer.  All arguments but
// `lane` are wave-uniform.  The batch pipeline is K1's (evofst_kernels.hip): every request is unconditional, so the
// wait for a batch is a count of younger loads, never "everything outstanding".
__device__ __forceinline__ void stream_count_piece(const StreamParams &p, int lane, int t0, int n_own, int slot0, uint16_t *cnt,
                                                   uint64_t *zf, uint64_t *flg, SegStream segs) {
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
    int zlo = 0, zhi = 0, flo = 0, fhi = 0;              // ballots by population, lane = population
    auto tile_done = [&](int slot) {                     // the tile's ballots -> LDS
        if (lane < P) {
            zf[(size_t)slot * 2 * P + lane] = ((uint64_t)(uint32_t)zhi << 32) | (uint32_t)zlo;
            zf[(size_t)slot * 2 * P + P + lane] = ((uint64_t)(uint32_t)fhi << 32) | (uint32_t)flo;
        }
        const uint64_t im = __ballot(irregular != 0 && !skip), mm = __ballot(anymiss != 0), nn = __ballot(nanany != 0 && !skip);
        if (lane == 0) { flg[slot * 4 + 0] = im; flg[slot * 4 + 1] = skipmask; flg[slot * 4 + 2] = mm; flg[slot * 4 + 3] = nn; }
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
            c = 0; mc = 0; irregular = 0; anymiss = 0; nanany = 0;
            sidx = 0; d = segs[0]; dn = segs[1];
        }
        load_item(spare);
        asm volatile("" :: "v"(cur.flag));               // the flag byte is the last load of its batch: the batch has arrived
        if (!live) return;
        uint16_t *cp = cnt + (size_t)(slot0 + ti) * P * kStreamRow + lane;
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            if (wb + u < p.W) {
                const uint32_t xlo = (uint32_t)cur.r[u].x, xhi = (uint32_t)(cur.r[u].x >> 32);
                const uint32_t ylo = (uint32_t)cur.r[u].y, yhi = (uint32_t)(cur.r[u].y >> 32);
                bool more = true;
                while (more) {
                    const SegWords dnn = segs[sidx + 2];
                    c = __popc(xlo & d.x) + c; c = __popc(xhi & d.y) + c;
                    mc = __popc(ylo & d.x) + mc; mc = __popc(yhi & d.y) + mc;
                    const uint32_t fl = d.w, pop = d.z;
                    if (fl & kSegLast) {
                        const uint32_t nfull = fl >> kSegNFullShift;
                        const uint32_t nm = nfull - mc;
                        cp[pop * kStreamRow] = (uint16_t)(c | (nm << 8));
                        const uint64_t z = __ballot(c == 0) | skipmask;
                        const uint64_t f = __ballot(c == nm) | skipmask;
                        const int pl = (int)pop;
                        zlo = evofst_write_lane((int)(uint32_t)z, pl, zlo); zhi = evofst_write_lane((int)(uint32_t)(z >> 32), pl, zhi);
                        flo = evofst_write_lane((int)(uint32_t)f, pl, flo); fhi = evofst_write_lane((int)(uint32_t)(f >> 32), pl, fhi);
                        // sites the integer walk cannot take: a NaN in the reference (nm < 2) or an excluded site with a
                        // dxy term (fixed with missing alleles).  Both need a missing allele unless n_full < 2.
                        const bool anym = mc != 0;
                        if (__any(anym) || nfull < 2u) {
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
// Sums of the granules before `chunk` for column q, as one list of rows: chunks [grp0 * 32, chunk) of level 0, groups
// [grp1 * 32, grp0) of level 1, super-groups [0, grp1) of level 2; s[level] get the sums.  Sixteen rows are requested at
// a time (one round trip when they are there); a batch with a row that is not there yet is asked for again after a
// sleep.  False when the wait gave up (error word set, by this wave or by another).
__device__ __forceinline__ bool stream_lookback(const StreamParams &p, int par, int chunk, int q, uint32_t (&s)[3]) {
    constexpr int F = kStreamFan, B = 16;
    const int grp0 = chunk / F, grp1 = grp0 / F;
    const int n0 = chunk - grp0 * F, n1 = grp0 - grp1 * F, n2 = grp1;
    const int total = n0 + n1 + n2;
    const uint32_t np = (uint32_t)p.n_pairs;
    const uint32_t *agg = p.agg[par];
    s[0] = s[1] = s[2] = 0u;
    uint32_t spins = 0;
    for (int r0 = 0; r0 < total; r0 += B) {
        uint32_t off[B];
        int lev[B];
#pragma unroll
        for (int u = 0; u < B; ++u) {
            const int r = min(r0 + u, total - 1);
            lev[u] = r < n0 ? 0 : (r < n0 + n1 ? 1 : 2);
            const int row = lev[u] == 0 ? grp0 * F + r : (lev[u] == 1 ? p.l1_row + grp1 * F + r - n0 : p.l2_row + r - n0 - n1);
            off[u] = (uint32_t)row * np + (uint32_t)q;
        }
        uint32_t x[B];
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int u = 0; u < B; ++u) x[u] = __hip_atomic_load((gu32 *)(agg + off[u]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int u = 0; u < B; ++u) ok = ok && x[u] != 0u;
            if (ok) break;
            if (++spins > kStreamSpinLimit) { __hip_atomic_store((gu32 *)p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
            if ((spins & 63u) == 0u && __hip_atomic_load((gu32 *)p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
            __builtin_amdgcn_s_sleep(16);
        }
#pragma unroll
        for (int u = 0; u < B; ++u) {
            const uint32_t v = r0 + u < total ? x[u] - 1u : 0u;
            s[0] += lev[u] == 0 ? v : 0u; s[1] += lev[u] == 1 ? v : 0u; s[2] += lev[u] == 2 ? v : 0u;
        }
    }
    return true;
}

// The statistics walk of one pair (this lane) over the nt tiles of a chunk whose packed counts and ballots lie in LDS:
// clean_walk_kernel's arithmetic (evofst_fused.hip).  No site of the chunk is skipped and no allele is missing, so
// p = c / n with the population's full n, and the window sums of d^2, p(1-p) and sumKij follow exactly from integer
// sums of c_i c_j, c and c^2 (a chunk has at most 2^10 sites and c <= 255: every sum stays below 2^31 and every
// product below 2^53).
__device__ __forceinline__ void stream_walk_pair(const StreamParams &p, const uint16_t *cnt, const uint64_t *zf, const uint32_t *pmask, int nt,
                                                 int gtile0, int chunk, int q, uint32_t pi, uint32_t pj, int64_t b0, int64_t g0) {
    constexpr int SUB = 32, NW4 = SUB / 4;
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
    for (int st = 0; st < 2 * nt; ++st) {
        const int ti = st >> 1, sh = (st & 1) * SUB;
        const uint64_t *zp = zf + (size_t)ti * 2 * P;
        const uint32_t incl = ~(uint32_t)(((zp[pi] & zp[pj]) | (zp[P + pi] & zp[P + pj])) >> sh);   // fst.cpp:543-544
        int b = SUB;                                       // site of this lane's next boundary in the fill (SUB: none)
        {
            const int n = __popc(incl);
            if (left <= n) b = nth_set_bit(incl, left);
            else left -= n;
        }
        constexpr uint32_t kOnes = 0x01010101u;
        uint32_t wi[NW4], wj[NW4];
        {
            const uint4 *ri = reinterpret_cast<const uint4 *>(cnt + ((size_t)ti * P + pi) * kStreamRow + sh);
            const uint4 *rj = reinterpret_cast<const uint4 *>(cnt + ((size_t)ti * P + pj) * kStreamRow + sh);
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
    // what the chunk holds behind its last boundary
    const size_t cq = (size_t)chunk * p.n_pairs + q;
    const size_t plane = (size_t)p.n_pairs * p.n_chunks;
    double num, den, k;
    parts(aCC, aCi, aCj, aQi, aQj, num, den, k);
    p.tail[cq] = num; p.tail[plane + cq] = den; p.tail[2 * plane + cq] = k;
    p.tail_n[cq] = make_int4((nflush ? p.g : left0) - left, nflush, row0, 0);   // {sites summed in the tail, segments completed, row of the first}
}

// Everything the consumer wave does with a counted chunk.  Returns false when a look-back wait gave up.
__device__ __forceinline__ bool stream_consume(const StreamParams &p, int lane, int par, int chunk, const uint16_t *cnt, const uint64_t *zf,
                                               const uint64_t *flg, const uint32_t *lpairs, uint32_t *lcen, const uint32_t *pmask) {
    constexpr int F = kStreamFan;
    const int NG = (p.n_pairs + 63) >> 6;
    const int P = p.P;
    const int gtile0 = chunk * p.chunk_tiles;
    const int nt = min(p.chunk_tiles, p.n_tiles - gtile0);
    const size_t np = (size_t)p.n_pairs;
    // a site skipped, an allele missing, a population with fewer than two chromosomes: the general walk's chunk
    uint64_t fw = 0;
    if (lane < nt) fw = flg[lane * 4 + 0] | flg[lane * 4 + 1] | flg[lane * 4 + 2];
    const bool dirty = __any(fw != 0);
    const int grp0 = chunk / F, grp1 = grp0 / F;

    // ---- census of every pair, published before anything is waited for
#pragma unroll 1
    for (int g = 0; g < NG; ++g) {
        const int q = g * 64 + lane;
        if (q < p.n_pairs) {
            const uint32_t ij = lpairs[q];
            const uint32_t pi = ij & 0xffffu, pj = ij >> 16;
            uint32_t n = 0;
            for (int ti = 0; ti < nt; ++ti) {
                const uint64_t *zp = zf + (size_t)ti * 2 * P;
                const uint64_t ex = (zp[pi] & zp[pj]) | (zp[P + pi] & zp[P + pj]);     // fst.cpp:543-544
                n += 64u - (uint32_t)__popcll(ex);
            }
            lcen[q] = n;                                   // this lane reads it back below
            __hip_atomic_store((gu32 *)(p.agg[par] + (size_t)chunk * np + q), n + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // the other copy's rows, for the next launch
            uint32_t *other = p.agg[par ^ 1];
            other[(size_t)chunk * np + q] = 0u;
            if (chunk % F == 0) other[(size_t)(p.l1_row + grp0) * np + q] = 0u;
            if (chunk % (F * F) == 0) other[(size_t)(p.l2_row + grp1) * np + q] = 0u;
        }
    }
    // ---- per group of 64 pairs: look back, then walk (or leave the chunk to the general walk)
    bool all_ok = true;
#pragma unroll 1
    for (int g = 0; g < NG; ++g) {
        const int q = g * 64 + lane;
        const bool active = q < p.n_pairs;
        uint32_t s[3];
#ifndef STREAM_NO_LB
        const bool ok = stream_lookback(p, par, chunk, active ? q : 0, s);
#else
        const bool ok = true; s[0] = s[1] = s[2] = chunk;
#endif
        if (__any(!ok)) { all_ok = false; break; }
        if (active) {
            const uint32_t ij = lpairs[q];
            const uint32_t pi = ij & 0xffffu, pj = ij >> 16;
            const uint32_t cen = lcen[q];
            if (chunk % F == F - 1) {
                __hip_atomic_store((gu32 *)(p.agg[par] + (size_t)(p.l1_row + grp0) * np + q), s[0] + cen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (grp0 % F == F - 1)
                    __hip_atomic_store((gu32 *)(p.agg[par] + (size_t)(p.l2_row + grp1) * np + q), s[1] + s[0] + cen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const int64_t b0 = p.base[q];
            const int64_t before = (int64_t)s[2] + (int64_t)s[1] + (int64_t)s[0];
            if (chunk == p.n_chunks - 1) p.incl_totals[q] = before + (int64_t)cen;
            if (chunk == 0) p.first_segment[q] = b0 / p.g;
            const int64_t g0 = b0 + before;
            if (dirty) p.chunk_offsets[(size_t)q * p.n_chunks + chunk] = g0;
#ifndef STREAM_NO_WALK
            else stream_walk_pair(p, cnt, zf, pmask, nt, gtile0, chunk, q, pi, pj, b0, g0);
#endif
        }
    }
    if (!all_ok) return false;
    if (dirty) {
        // K1's outputs for this chunk: packed counts [tile][P][64], ballots [tile][z|f][P], the four site masks
        const int nv = nt * P * 8;                         // 16-byte vectors of the counts
        uint4 *dst = reinterpret_cast<uint4 *>(reinterpret_cast<uint16_t *>(p.counts) + (size_t)gtile0 * P * kTileSites);
        for (int v = lane; v < nv; v += 64) {
            const int row = v >> 3, part = v & 7;          // row = slot * P + population
            dst[v] = *reinterpret_cast<const uint4 *>(cnt + (size_t)row * kStreamRow + part * 8);
        }
        for (int i = lane; i < nt * 2 * P; i += 64) p.zf_masks[(size_t)gtile0 * 2 * P + i] = zf[i];
        if (lane < nt) {
            p.irr_mask[gtile0 + lane] = flg[lane * 4 + 0]; p.skip_mask[gtile0 + lane] = flg[lane * 4 + 1];
            p.miss_mask[gtile0 + lane] = flg[lane * 4 + 2]; p.nan_mask[gtile0 + lane] = flg[lane * 4 + 3];
        }
    }
    if (lane == 0) p.chunk_dirty[chunk] = dirty ? 1u : 0u;
    return true;
}

__global__ __launch_bounds__(kStreamThreads, EVOFST_STREAM_WPS) void stream_pairs_kernel(StreamParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool consumer = wave == kStreamProd;
    uint32_t *pmask = reinterpret_cast<uint32_t *>(smem);                      // [32][8] byte masks of sites 0 .. b of a fill
    volatile int32_t *lmisc = reinterpret_cast<volatile int32_t *>(smem + 1024); // [0], [1]: chunk of iteration i & 1; [2]: a wait gave up; [3]: granule copy of this launch
    uint32_t *lpairs = reinterpret_cast<uint32_t *>(smem + 1024 + 64);
    uint32_t *lcen = reinterpret_cast<uint32_t *>(smem + 1024 + 64 + s16((size_t)p.n_pairs * 4));   // the consumer's: census of the chunk it holds
    unsigned char *bufs = smem + 1024 + 64 + 2 * s16((size_t)p.n_pairs * 4);
    const size_t cnt_bytes = s16((size_t)p.chunk_tiles * p.P * kStreamRow * 2), zf_bytes = s16((size_t)p.chunk_tiles * 2 * p.P * 8);
    const size_t buf_bytes = cnt_bytes + zf_bytes + s16((size_t)p.chunk_tiles * 4 * 8);
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
    if (tid == 0) lmisc[3] = (int32_t)(__hip_atomic_load((gu32 *)p.launches, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u);
    for (int i = tid; i < 32 * 8; i += kStreamThreads) {
        const int b = i >> 3, m = i & 7, nbv = b + 1 - 4 * m;
        pmask[i] = nbv >= 4 ? 0xffffffffu : (nbv <= 0 ? 0u : ((1u << (8 * nbv)) - 1u));
    }
    for (int i = tid; i < p.n_pairs; i += kStreamThreads) lpairs[i] = p.pairs[i];
    if (tid == 0) { lmisc[0] = draw(); lmisc[2] = 0; }   // after the launch counter has been read (same thread, above)
    __syncthreads();
    const int par = lmisc[3];

    bool dead = false;
    int32_t prev = n_chunks;
    for (int i = 0;; ++i) {
        const int32_t ci = lmisc[i & 1];
        const bool have = ci < n_chunks;
        unsigned char *bi_ = bufs + (size_t)(i & 1) * buf_bytes, *bp_ = bufs + (size_t)((i & 1) ^ 1) * buf_bytes;
        if (!consumer) {
#ifndef STREAM_NO_COUNT
            if (have && !dead) {
                const int t0 = ci * p.chunk_tiles + wave * p.tr;
                const int n_own = min(p.tr, p.n_tiles - t0);
                if (n_own > 0)
                stream_count_piece(p, lane, t0, n_own, wave * p.tr, reinterpret_cast<uint16_t *>(bi_),
                                   reinterpret_cast<uint64_t *>(bi_ + cnt_bytes), reinterpret_cast<uint64_t *>(bi_ + cnt_bytes + zf_bytes), segs);
            }
#endif
        } else {
            if (lane == 0) {
                lmisc[(i + 1) & 1] = have ? draw() : n_chunks;
                if (__hip_atomic_load((gu32 *)p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) lmisc[2] = 1;
            }
#ifndef STREAM_NO_CONSUME
            if (prev < n_chunks && !dead) {
                const bool ok = stream_consume(p, lane, par, prev, reinterpret_cast<const uint16_t *>(bp_),
                                               reinterpret_cast<const uint64_t *>(bp_ + cnt_bytes),
                                               reinterpret_cast<const uint64_t *>(bp_ + cnt_bytes + zf_bytes), lpairs, lcen, pmask);
                if (!ok && lane == 0) lmisc[2] = 1;
            }
#endif
        }
        __syncthreads();
        dead = dead || lmisc[2] != 0;
        if (!have) break;
        prev = ci;
    }
}

hipError_t launch_stream(const StreamParams &p, int max_blocks, hipStream_t st) {
    if (p.n_chunks <= 0 || p.n_pairs <= 0) return hipSuccess;
    if (p.n_pairs > kStreamMaxPairs || p.P > 64 || p.tr <= 0 || p.chunk_tiles != p.tr * kStreamProd || p.chunk_tiles > 64) return hipErrorInvalidValue;
    const void *fn = reinterpret_cast<const void *>(&stream_pairs_kernel);
    const size_t lds = stream_lds_bytes(p.P, p.n_pairs, p.chunk_tiles);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    int grid = max_blocks;
    if (grid <= 0) {
        // workgroups the device holds at once; asked once per shape (the query costs more than a launch).  Any grid is
        // safe -- chunks go by ticket -- this one just leaves no workgroup waiting for a slot.
        static thread_local struct { size_t lds; int dev; int grid; } memo = {0, -1, 0};
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (memo.lds != lds || memo.dev != dev) {
            int per_cu = 0, cus = 0;
            if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kStreamThreads, lds)) != hipSuccess) return e;
            if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
            memo = {lds, dev, std::max(1, per_cu) * std::max(1, cus)};
        }
        grid = memo.grid;
    }
    grid = std::max(1, std::min(grid, p.n_chunks));
    hipLaunchKernelGGL(stream_pairs_kernel, dim3(grid), dim3(kStreamThreads), lds, st, p);
    return hipGetLastError();
}

}  // namespace evofst
