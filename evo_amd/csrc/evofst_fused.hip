// evofst_fused.hip -- pairs mode as ONE streaming kernel (gfx950, wave64).
//
// Replaces, for every population pair at once, the per-line work of getFstFromVCF: the counting loops
// (process_vcf_fst.cpp:243-280), the site filter (:543-544), calculateFstNumerator / Denominator
// (process_vcf_fst.h:19-30), calculateDxy (process_vcf_fst.cpp:297-323) and the SNP windows (:638-650).
//
// A workgroup takes chunks of a few tiles in ticket order (one atomic per chunk) and does, per chunk:
//   phase A  lane = site: 16-byte coalesced loads of the tile rows, popcounts under wave-uniform masks, packed
//            {alt, non-missing} counts and the zero / fixed ballots of every population INTO LDS (nothing
//            but the tiles is read from HBM and no counts are written to it);
//   census   thread = pair: included sites of the chunk from the ballots (filter :543-544);
//   look-back the chunk publishes its census as 8-byte {epoch, count} granules and sums those of the
//            chunks before it (three levels of fan-out 32: chunks of its group, groups of its super-group,
//            super-groups; the owner of a group's last chunk has read all the others of the group for its
//            own prefix, so it publishes the group's aggregate for free).  Tickets are taken by running
//            workgroups only, so everything a chunk waits for belongs to a workgroup that is running
//            or done, and phase A waits for nothing: the waits always end.  Spins are bounded anyway and
//            give up through an error word that every workgroup polls;
//   phase B  the counts in LDS -> per-population tables (p, prefix sums of h, t, c) per half tile, thread =
//            pair walks the sites, segment sums at the window boundaries (K2's scheme, see evofst_kernels.hip),
//            with accumulators that restart with every chunk of at most a few hundred sites.
// The segment a chunk starts in the middle of is completed by stitch_kernel from the chunks' tails, as before.
#include "evofst_internal.h"

#include <algorithm>

namespace evofst {

constexpr int kFan = 32;                  // look-back fan-out per level
constexpr int kFuseNB = 8;                // tile rows per load batch (8 KiB per wave)
constexpr uint32_t kSpinLimit = 1u << 22; // polls before a wait gives up (seconds; a healthy wait takes microseconds)

size_t fused_lds_bytes(int P, int W, int n_segs, int max_n, int chunk_tiles, int count_bytes) {
    auto a16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    size_t b = 0;
    b += a16(max_n <= kRcpMax ? (size_t)(max_n + 1) * sizeof(RcpEntry) : 0);
    b += 3 * a16((size_t)P * kSubPad * 8);      // pt, PH, PT
    b += 2 * a16((size_t)P * 8);                // carH, carT
    b += 2 * a16((size_t)P * kSubPad * 4);      // xt, PC
    b += a16((size_t)P * 4);                    // carC
    b += a16((size_t)n_segs * sizeof(SegDesc));
    b += a16((size_t)(W + 1) * 4);
    b += a16((size_t)chunk_tiles * P * kTileSites * count_bytes);
    b += a16((size_t)chunk_tiles * 2 * P * 8);  // zero / fixed ballots
    b += 2 * a16((size_t)chunk_tiles * 8);      // irregular, skip
    b += 16;                                    // ticket broadcast
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

// Sum of the values of granule rows [lo, hi) of one column (stride = granules per row); waits for rows that are
// not yet published.  Returns false when the wait gave up (error word set, by this or by another thread).
__device__ __forceinline__ bool sum_granules(const uint64_t *col, size_t stride, int lo, int hi, uint32_t epoch,
                                             uint32_t *err, uint64_t &sum) {
    constexpr int B = 8;
    uint32_t spins = 0;
    for (int r0 = lo; r0 < hi; r0 += B) {
        uint64_t x[B];
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int u = 0; u < B; ++u) x[u] = load_granule(col + (size_t)min(r0 + u, hi - 1) * stride);
#pragma unroll
            for (int u = 0; u < B; ++u) ok = ok && (uint32_t)(x[u] >> 32) == epoch;
            if (ok) break;
            if (++spins > kSpinLimit) { __hip_atomic_store((gu32 *)err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
            if ((spins & 255u) == 0u && __hip_atomic_load((gu32 *)err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
            __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int u = 0; u < B; ++u) sum += r0 + u < hi ? (uint64_t)(uint32_t)x[u] : 0ull;
    }
    return true;
}

template <typename CT, bool use_rcp>
__global__ __launch_bounds__(kK2Threads, kK2WavesPerSimd) void fused_pairs_kernel(FusedParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int EPT = 16 / (int)sizeof(CT);     // sites per 16-byte count vector
    constexpr int VPR = kSub / EPT;               // vectors per population row of a half tile
    constexpr int EB = 8 * (int)sizeof(CT);
    const int P = p.P;
    size_t off = 0;
    auto carve = [&](size_t bytes) { unsigned char *r = smem + off; off += (bytes + 15) & ~(size_t)15; return r; };
    RcpEntry *rcp = reinterpret_cast<RcpEntry *>(carve(use_rcp ? (size_t)(p.max_n + 1) * sizeof(RcpEntry) : 0));
    double *pt = reinterpret_cast<double *>(carve((size_t)P * kSubPad * 8));   // [P][33] p = c / nm
    double *PH = reinterpret_cast<double *>(carve((size_t)P * kSubPad * 8));   // [P][33] sum of h from the start of the chunk
    double *PT = reinterpret_cast<double *>(carve((size_t)P * kSubPad * 8));   // [P][33] sum of t
    double *carH = reinterpret_cast<double *>(carve((size_t)P * 8));           // [P] the same sums after the last derived half tile
    double *carT = reinterpret_cast<double *>(carve((size_t)P * 8));
    uint32_t *xt = reinterpret_cast<uint32_t *>(carve((size_t)P * kSubPad * 4));   // [P][33] c | flags
    uint32_t *PC = reinterpret_cast<uint32_t *>(carve((size_t)P * kSubPad * 4));   // [P][33] sum of c
    uint32_t *carC = reinterpret_cast<uint32_t *>(carve((size_t)P * 4));
    SegDesc *lsegs = reinterpret_cast<SegDesc *>(carve((size_t)p.n_segs * sizeof(SegDesc)));
    int32_t *lwss = reinterpret_cast<int32_t *>(carve((size_t)(p.W + 1) * 4));
    CT *lcnt = reinterpret_cast<CT *>(carve((size_t)p.chunk_tiles * P * kTileSites * sizeof(CT)));   // [tile][P][64]
    uint64_t *lzf = reinterpret_cast<uint64_t *>(carve((size_t)p.chunk_tiles * 2 * P * 8));          // [tile][z|f][P]
    uint64_t *lirr = reinterpret_cast<uint64_t *>(carve((size_t)p.chunk_tiles * 8));
    uint64_t *lskip = reinterpret_cast<uint64_t *>(carve((size_t)p.chunk_tiles * 8));
    int32_t *lmisc = reinterpret_cast<int32_t *>(carve(16));

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nthreads = blockDim.x, NWV = nthreads >> 6;
    const int q = tid;
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
    const uint32_t io = pi * kSubPad, jo = pj * kSubPad;
    for (int i = tid; i <= p.W; i += nthreads) lwss[i] = p.word_seg_start[i];
    for (int i = tid; i < p.n_segs; i += nthreads) lsegs[i] = p.segs[i];
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
    const int nb = (p.W + kFuseNB - 1) / kFuseNB;      // load batches per tile
    const uint32_t n_tickets = (uint32_t)p.n_chunks + gridDim.x;   // every workgroup draws exactly one ticket past the end
    bool dead = false;                                 // a wait gave up somewhere: draw the remaining tickets, do nothing

    for (;;) {
        __syncthreads();                               // the tables above; the previous chunk's readers of lmisc / LDS
        if (tid == 0) {
            const uint32_t t = __hip_atomic_fetch_add((gu32 *)p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // the last ticket of the launch: nobody adds after it, so the counter can go back to zero for the next launch
            if (t == n_tickets - 1u) __hip_atomic_store((gu32 *)p.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lmisc[0] = (int32_t)min(t, (uint32_t)p.n_chunks);
            lmisc[1] = (int32_t)__hip_atomic_load((gu32 *)p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        const int chunk = lmisc[0];
        dead = dead || lmisc[1] != 0;
        if (chunk >= p.n_chunks) break;
        if (dead) continue;
        const int t0 = chunk * p.chunk_tiles;
        const int nt = min(p.chunk_tiles, p.n_tiles - t0);

        // ------------------------------------------------------------------ phase A: tiles -> counts and ballots in LDS
        {
            const int my_nt = wave < nt ? (nt - wave + NWV - 1) / NWV : 0;   // tiles wave, wave + NWV, ...
            const int n_items = my_nt * nb;
            ulonglong2 cur[kFuseNB], nxt[kFuseNB];
            auto load_item = [&](ulonglong2 (&r)[kFuseNB], int item) {
                const int tj = item / nb, wb = (item - tj * nb) * kFuseNB;
                const ulonglong2 *tp = p.tiles + (size_t)(t0 + wave + tj * NWV) * p.W * kTileSites + lane;
#pragma unroll
                for (int u = 0; u < kFuseNB; ++u)   // unconditional (clamped): all loads of a batch are in flight together
                    r[u] = tp[(size_t)min(wb + u, p.W - 1) * kTileSites];
            };
            if (n_items > 0) load_item(cur, 0);
            uint32_t c = 0, mc = 0;
            bool irregular = false, skip = false;
            uint64_t skipmask = 0;
            for (int item = 0; item < n_items; ++item) {
                const int tj = item / nb, bi = item - tj * nb;
                const int ti = wave + tj * NWV, wb = bi * kFuseNB;
                if (item + 1 < n_items) load_item(nxt, item + 1);
                if (bi == 0) {
                    const int64_t site = (int64_t)(t0 + ti) * kTileSites + lane;
                    skip = site >= p.n_sites;
                    if (!skip && p.site_flags) skip = (p.site_flags[site] & 1u) != 0;
                    skipmask = __ballot(skip);
                    c = 0; mc = 0; irregular = false;
                }
                CT *cp = lcnt + (size_t)ti * P * kTileSites + lane;
                uint64_t *zp = lzf + (size_t)ti * 2 * P;
#pragma unroll
                for (int u = 0; u < kFuseNB; ++u) {
                    if (wb + u < p.W) {
                        const int s0 = lwss[wb + u], s1 = lwss[wb + u + 1];
                        for (int s = s0; s < s1; ++s) {
                            const SegDesc d = lsegs[s];
                            c += __popcll(cur[u].x & d.mask);
                            mc += __popcll(cur[u].y & d.mask);
                            if (d.flags & kSegLast) {
                                const uint32_t nfull = d.flags >> kSegNFullShift;
                                const uint32_t nm = nfull - mc;
                                cp[(size_t)d.pop * kTileSites] = (CT)(c | (nm << CountTraits<CT>::shift));
                                const uint64_t z = __ballot(c == 0) | skipmask;
                                const uint64_t f = __ballot(c == nm) | skipmask;
                                if (lane == 0) { zp[d.pop] = z; zp[P + d.pop] = f; }
                                // sites phase B must take through its general loop: a NaN in the reference (nm < 2) or an
                                // excluded site with a dxy term (fixed with missing alleles)
                                irregular = irregular || nm < 2 || (c == nm && nm < nfull);
                                c = 0; mc = 0;
                            }
                        }
                    }
                }
                if (bi == nb - 1) {
                    const uint64_t im = __ballot(irregular && !skip);
                    if (lane == 0) { lirr[ti] = im; lskip[ti] = skipmask; }
                }
                if (item + 1 < n_items) {
#pragma unroll
                    for (int u = 0; u < kFuseNB; ++u) cur[u] = nxt[u];
                }
            }
        }
        for (int i = tid; i < P; i += nthreads) { carH[i] = 0.0; carT[i] = 0.0; carC[i] = 0u; }
        __syncthreads();

        // ------------------------------------------------------------------ census, publish, look back
        uint32_t cen = 0;
        int64_t g0 = b0;
        bool ok = true;
        if (active) {
            for (int ti = 0; ti < nt; ++ti) {
                const uint64_t *zp = lzf + (size_t)ti * 2 * P;
                const uint64_t ex = (zp[pi] & zp[pj]) | (zp[P + pi] & zp[P + pj]);     // fst.cpp:543-544
                cen += 64u - (uint32_t)__popcll(ex);
            }
            const size_t np = (size_t)p.n_pairs;
            store_granule(p.agg0 + (size_t)chunk * np + q, p.epoch, cen);
            const int grp0 = chunk / kFan, grp1 = grp0 / kFan;
            uint64_t sum0 = 0, sum1 = 0, sum2 = 0;
            ok = sum_granules(p.agg0 + q, np, grp0 * kFan, chunk, p.epoch, p.err, sum0);
            if (ok && chunk % kFan == kFan - 1) store_granule(p.agg1 + (size_t)grp0 * np + q, p.epoch, (uint32_t)(sum0 + cen));
            ok = ok && sum_granules(p.agg1 + q, np, grp1 * kFan, grp0, p.epoch, p.err, sum1);
            if (ok && chunk % kFan == kFan - 1 && grp0 % kFan == kFan - 1)
                store_granule(p.agg2 + (size_t)grp1 * np + q, p.epoch, (uint32_t)(sum1 + sum0 + cen));
            ok = ok && sum_granules(p.agg2 + q, np, 0, grp1, p.epoch, p.err, sum2);
            g0 += (int64_t)(sum2 + sum1 + sum0);
            if (chunk == p.n_chunks - 1) p.incl_totals[q] = g0 - b0 + (int64_t)cen;
            if (chunk == 0) p.first_segment[q] = first_seg;
        }
        if (__syncthreads_or(!ok)) { dead = true; continue; }

        // ------------------------------------------------------------------ phase B: K2's walk over the chunk's half tiles
        int64_t m = g0 / p.g;
        int32_t left = p.g - (int32_t)(g0 - m * p.g);
        const int32_t left0 = left;
        const int32_t row0 = (int32_t)(m - first_seg);
        size_t orow = (size_t)row0 * p.n_pairs + q;
        double D = 0.0;                       // sum d^2 (masked by the site filter in general half tiles)
        uint64_t cc = 0;                      // sum c_i c_j
        uint32_t exi = 0, exj = 0;            // sum of c_i / c_j over excluded sites (general half tiles only)
        uint32_t bn = 0, bd = 0;              // included sites whose numerator / denominator is NaN in the reference
        double prevN = 0.0, prevD = 0.0;      // their values at the last boundary
        uint32_t prevCi = 0, prevCj = 0, prevBn = 0, prevBd = 0;
        uint64_t prevCC = 0;
        int32_t nflush = 0;
        const int nvec = P * VPR;

        // One vector = EPT consecutive sites of population k.
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
            double ah = ch + (eh - rh), at = ct + (et - rt);
            uint32_t ac = ccar + (ec - rc);
#pragma unroll
            for (int j = 0; j < EPT; ++j) {
                ah += hv[j]; at += tv[j]; ac += cv[j];
                PH[o + j] = ah; PT[o + j] = at; PC[o + j] = ac;
            }
        };

        for (int st = 0; st < 2 * nt; ++st) {
            const int ti = st >> 1, sh = (st & 1) * kSub;
            const uint32_t skiphalf = (uint32_t)(lskip[ti] >> sh);
            const bool general = (uint32_t)(lirr[ti] >> sh) != 0u;   // uniform
            {
                const uint4 *vp = reinterpret_cast<const uint4 *>(lcnt + (size_t)ti * P * kTileSites);
                for (int vi = tid; vi < nvec; vi += nthreads) {
                    const int k = vi / VPR, r = vi - k * VPR;
                    derive_vec(vi, vp[k * (kTileSites / EPT) + (st & 1) * VPR + r], skiphalf);
                }
            }
            __syncthreads();
            if (active) {
                const uint64_t *zp = lzf + (size_t)ti * 2 * P;
                const uint32_t incl = ~(uint32_t)(((zp[pi] & zp[pj]) | (zp[P + pi] & zp[P + pj])) >> sh);   // fst.cpp:543-544
                int b = kSub;                                  // site of this lane's next boundary in the half tile (kSub: none)
                {
                    const int cnt = __popc(incl);
                    if (left <= cnt) b = nth_set_bit(incl, left);
                    else left -= cnt;
                }
                const int bgrp = b & ~(kGrp - 1);
                double sD = D;
                uint64_t scc = cc;
                uint32_t sexi = exi, sexj = exj, sbn = bn, sbd = bd;
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
                        uint32_t cch = 0;
#pragma unroll
                        for (int u = 0; u < kGrp; ++u) {
                            const double d = a[u] - bb[u];
                            D = fma(d, d, D);
                            cch += __umul24(xi[u], xj[u]);
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
                // emit: the accumulators right after boundary site b -> one row of the segment table, then the next boundary
                auto emit = [&](double cD, uint64_t ccc, uint32_t cexi, uint32_t cexj, uint32_t cbn, uint32_t cbd) {
                    const double H = PH[io + b] + PH[jo + b], T = PT[io + b] + PT[jo + b];
                    const uint32_t Ci = PC[io + b] - cexi, Cj = PC[jo + b] - cexj;
                    const double cumN = cD - H, cumD = cD + T;
                    const uint64_t kk = (uint64_t)nj * (uint64_t)(Ci - prevCi) + (uint64_t)ni * (uint64_t)(Cj - prevCj) - 2ull * (ccc - prevCC);
                    SegRow row;
                    row.num = cbn != prevBn ? kNaN : cumN - prevN;
                    row.den = cbd != prevBd ? kNaN : cumD - prevD;
                    row.k = (double)kk;
                    row.end = (t0 * 2 + st) * kSub + b; row.pad = 0;
                    {
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
                if (b < kSub) {
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
                        emit(cD, ccc, cexi, cexj, cbn, cbd);
                    }
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
                        emit(cD, ccc, cexi, cexj, cbn, cbd);
                    }
                }
            }
            __syncthreads();
        }
        if (active) {
            // what the chunk holds behind its last boundary
            const size_t cq = (size_t)chunk * p.n_pairs + q;
            const size_t plane = (size_t)p.n_pairs * p.n_chunks;
            const double H = carH[pi] + carH[pj], T = carT[pi] + carT[pj];
            const uint32_t Ci = carC[pi] - exi, Cj = carC[pj] - exj;
            const uint64_t kk = (uint64_t)nj * (uint64_t)(Ci - prevCi) + (uint64_t)ni * (uint64_t)(Cj - prevCj) - 2ull * (cc - prevCC);
            p.tail[cq] = bn != prevBn ? kNaN : (D - H) - prevN;
            p.tail[plane + cq] = bd != prevBd ? kNaN : (D + T) - prevD;
            p.tail[2 * plane + cq] = (double)kk;
            p.tail_n[cq] = make_int4((nflush ? p.g : left0) - left, nflush, row0, 0);
        }
    }
}

template <typename CT, bool RCP>
static hipError_t launch_fused_t(const FusedParams &p, size_t lds, int threads, int max_blocks, hipStream_t st, int *grid_out) {
    const void *fn = reinterpret_cast<const void *>(&fused_pairs_kernel<CT, RCP>);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    int grid = max_blocks;
    if (grid <= 0) {
        int per_cu = 0, dev = 0, cus = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, lds);
        if (e != hipSuccess) return e;
        if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
        if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
        grid = std::max(1, per_cu) * std::max(1, cus);
    }
    grid = std::max(1, std::min(grid, p.n_chunks));
    if (grid_out) *grid_out = grid;
    hipLaunchKernelGGL((fused_pairs_kernel<CT, RCP>), dim3(grid), dim3(threads), lds, st, p);
    return hipGetLastError();
}

hipError_t launch_fused(const FusedParams &p, int count_bytes, int max_blocks, hipStream_t st, int *grid_out) {
    if (p.n_chunks <= 0 || p.n_pairs <= 0) return hipSuccess;
    if (p.n_pairs > kK2Threads) return hipErrorInvalidValue;
    const int threads = 64 * ((p.n_pairs + 63) / 64);
    const size_t lds = fused_lds_bytes(p.P, p.W, p.n_segs, p.max_n, p.chunk_tiles, count_bytes);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const bool rcp = p.max_n <= kRcpMax;
    if (count_bytes == 2) return rcp ? launch_fused_t<uint16_t, true>(p, lds, threads, max_blocks, st, grid_out)
                                     : launch_fused_t<uint16_t, false>(p, lds, threads, max_blocks, st, grid_out);
    return rcp ? launch_fused_t<uint32_t, true>(p, lds, threads, max_blocks, st, grid_out)
               : launch_fused_t<uint32_t, false>(p, lds, threads, max_blocks, st, grid_out);
}

}  // namespace evofst
