// What a small write stream costs inside a large read stream (the stream pass writes 0.38 GB of 32-byte segment rows while
// it reads 6.2 GB of tiles, and the rows cost 0.3 of its 1.47 ms; K1's count stores showed the same ratio).
// One persistent workgroup per CU (256 threads); every wave streams its share of the source (8 rows of 1 KiB in flight)
// and, after every `every` rows read, writes `wbytes` per lane in one of these patterns:
//   0  nothing
//   1  coalesced: 64 lanes x 16 bytes = 1 KiB contiguous
//   2  32-byte rows, one per lane, a lane's rows one behind the other (stride between lanes: its own region)
//   3  32-byte rows, one per lane, lanes adjacent ([row][lane] table: a wave writes 2 KiB contiguous)
//   4  as 2, but only every 8th lane writes (8 x as often): few active lanes per store instruction
//   5  as 1, into 4 KiB per wave over and over (the writes stay in the L2: what of the cost is the memory side?)
//   6  as 1, but only one workgroup in eight writes, eight times as often (the writes come from few CUs)
//   9  as 1 with 4 bytes per lane (256 bytes per instruction): the same number of store instructions, a quarter of the bytes
//   8  as 1 in bursts: eight stores of 1 KiB back to back, an eighth as often (16: sixteen)
//   7  the fourth wave of every workgroup writes and reads nothing (1 KiB, then a sleep, as often as makes the same volume);
//      the other three waves read everything: no wave ever waits for a store of its own
// hipcc --offload-arch=gfx950 -O3 -o tools/hbm_write_mix_probe tools/hbm_write_mix_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void mix_kernel(const uint4 *src, size_t n_rows, uint4 *dst, size_t dst_vec, int pattern, int every, uint32_t *out) {
    constexpr int NB = 8;
    const int lane = threadIdx.x & 63;
    size_t n_waves = (size_t)gridDim.x * 4, wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pattern == 7) {
        const int w = threadIdx.x >> 6;
        if (w == 3) {
            const size_t slice7 = dst_vec / gridDim.x;
            uint4 *my7 = dst + (size_t)blockIdx.x * slice7;
            const size_t events = n_rows / (size_t)every / gridDim.x;      // the volume of pattern 1 at this `every`
            for (size_t e = 0; e < events; ++e) {
                if ((e + 1) * 64 <= slice7) my7[e * 64 + lane] = make_uint4((uint32_t)e, 1u, lane, 7u);
                for (int z = 0; z < (every * 3) / 16; ++z) __builtin_amdgcn_s_sleep(16);
            }
            return;
        }
        n_waves = (size_t)gridDim.x * 3; wave = (size_t)blockIdx.x * 3 + w;
        pattern = 0;
    }
    const size_t per = (n_rows + n_waves - 1) / n_waves;
    size_t r0 = wave * per, r1 = r0 + per;
    if (r1 > n_rows) r1 = n_rows;
    uint4 cur[NB], nxt[NB];
    uint32_t acc = 0;
    auto load = [&](uint4 (&r)[NB], size_t row) {
#pragma unroll
        for (int u = 0; u < NB; ++u) { size_t rr = row + u < r1 ? row + u : r1 - 1; r[u] = src[rr * 64 + lane]; }
    };
    // this wave's slice of the destination, in 16-byte vectors
    const size_t slice = dst_vec / n_waves;
    uint4 *my = dst + wave * slice;
    size_t wpos = 0;          // rows written so far
    int since = 0, turn = 0;
    if (r0 < r1) load(cur, r0);
    for (size_t row = r0; row < r1; row += NB) {
        if (row + NB < r1) load(nxt, row + NB);
#pragma unroll
        for (int u = 0; u < NB; ++u) acc += __popc(cur[u].x) + __popc(cur[u].y) + __popc(cur[u].z) + __popc(cur[u].w);
        since += NB;
        if (pattern && since >= every) {
            since = 0;
            const uint4 v = make_uint4(acc, (uint32_t)row, lane, 7u);
            if (pattern == 1) {                              // 1 KiB contiguous
                if ((wpos + 1) * 64 <= slice) my[wpos * 64 + lane] = v;
                ++wpos;
            } else if (pattern == 9) {
                if ((wpos + 1) * 64 <= slice) reinterpret_cast<uint32_t *>(my + wpos * 64)[lane] = acc;
                ++wpos;
            } else if (pattern == 8 || pattern == 16) {
                if (++turn == pattern) {
                    turn = 0;
#pragma unroll 1
                    for (int k = 0; k < pattern; ++k) { if ((wpos + 1) * 64 <= slice) my[wpos * 64 + lane] = v; ++wpos; }
                }
            } else if (pattern == 5) {
                my[(wpos & 3) * 64 + lane] = v;
                ++wpos;
            } else if (pattern == 6) {
                if ((blockIdx.x & 7) == 0) {
#pragma unroll 1
                    for (int k = 0; k < 8; ++k) { if ((wpos + 1) * 64 <= slice) my[wpos * 64 + lane] = v; ++wpos; }
                }
            } else if (pattern == 2) {                       // lane-major 32-byte rows
                const size_t lane_slice = slice / 64;        // vectors per lane
                if ((wpos + 1) * 2 <= lane_slice) { uint4 *d = my + lane * lane_slice + wpos * 2; d[0] = v; d[1] = v; }
                ++wpos;
            } else if (pattern == 3) {                       // [row][lane] 32-byte rows
                if ((wpos + 1) * 128 <= slice) { uint4 *d = my + wpos * 128 + lane * 2; d[0] = v; d[1] = v; }
                ++wpos;
            } else if (pattern == 4) {                       // lane-major, 8 lanes a time
                const size_t lane_slice = slice / 64;
                const size_t k = wpos / 8;
                if ((lane & 7) == (turn & 7) && (k + 1) * 2 <= lane_slice) { uint4 *d = my + lane * lane_slice + k * 2; d[0] = v; d[1] = v; }
                ++turn; ++wpos;
            }
        }
        if (row + NB < r1) {
#pragma unroll
            for (int u = 0; u < NB; ++u) cur[u] = nxt[u];
        }
    }
    if (acc == 0xffffffffu) out[wave] = acc;
}

int main(int argc, char **argv) {
    const size_t mib = argc > 1 ? (size_t)atol(argv[1]) : 6000;
    const size_t bytes = mib << 20, n_rows = bytes / 1024;
    uint32_t *out; CHK(hipMalloc(&out, 1 << 24));
    uint4 *src, *dst; CHK(hipMalloc(&src, bytes)); CHK(hipMemset(src, 0x5a, bytes));
    const size_t dst_bytes = (size_t)1 << 30;
    CHK(hipMalloc(&dst, dst_bytes)); CHK(hipMemset(dst, 0, dst_bytes));
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int wgs_per_cu : {2}) {
        const int grid = prop.multiProcessorCount * wgs_per_cu;
        for (int pattern : {0, 1, 9}) {
            // every: rows read per write event; bytes written per event: pattern 1: 1 KiB, 2/3: 2 KiB, 4: 256 B
            for (int every : {16, 32, 64}) {
                if (pattern == 0 && every != 16) continue;
                int ev = every;
                if (pattern == 2 || pattern == 3) ev = every * 2;      // same write fraction as pattern 1
                if (pattern == 4) ev = every / 4 < 8 ? 8 : every / 4;
                hipLaunchKernelGGL(mix_kernel, dim3(grid), dim3(256), 0, 0, src, n_rows, dst, dst_bytes / 16, pattern, ev, out);
                CHK(hipDeviceSynchronize());
                const int iters = 5;
                CHK(hipEventRecord(e0));
                for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(mix_kernel, dim3(grid), dim3(256), 0, 0, src, n_rows, dst, dst_bytes / 16, pattern, ev, out);
                CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
                const double us = ms * 1e3 / iters;
                const double wper = pattern == 0 ? 0.0 : pattern == 9 ? 256.0 : (pattern == 1 || pattern >= 5) ? 1024.0 : pattern == 4 ? 256.0 : 2048.0;
                const double wfrac = pattern ? wper / (ev * 1024.0) : 0.0;
                printf("wgs/cu %d pattern %d every %3d rows (writes %.1f %% of the reads, %.0f MB): %.1f us  read rate %.2f TB/s\n", wgs_per_cu, pattern, ev,
                       100.0 * wfrac, wfrac * bytes / 1e6, us, bytes / us / 1e6);
            }
        }
    }
    return 0;
}
