// Achievable HBM read rate of a streaming kernel shaped like K1 (wave = 8 rows of 1 KiB in flight, next batch requested
// before the current one is consumed), without K1's arithmetic: the ceiling K1's GB/s is compared with in DESIGN.md.
// hipcc --offload-arch=gfx950 -O3 -o tools/hbm_read_probe tools/hbm_read_probe.hip ; tools/hbm_read_probe [MiB ...]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NB>
__global__ __launch_bounds__(256) void read_kernel(const uint4 *src, size_t rows_per_wave, size_t n_rows, uint32_t *out) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    size_t r0 = wave * rows_per_wave, r1 = r0 + rows_per_wave;
    if (r1 > n_rows) r1 = n_rows;
    uint4 cur[NB], nxt[NB];
    uint32_t acc = 0;
    auto load = [&](uint4 (&r)[NB], size_t row) {
#pragma unroll
        for (int u = 0; u < NB; ++u) { size_t rr = row + u < r1 ? row + u : r1 - 1; r[u] = src[rr * 64 + lane]; }
    };
    if (r0 < r1) load(cur, r0);
    for (size_t row = r0; row < r1; row += NB) {
        if (row + NB < r1) load(nxt, row + NB);
#pragma unroll
        for (int u = 0; u < NB; ++u) acc += __popc(cur[u].x) + __popc(cur[u].y) + __popc(cur[u].z) + __popc(cur[u].w);
        if (row + NB < r1) {
#pragma unroll
            for (int u = 0; u < NB; ++u) cur[u] = nxt[u];
        }
    }
    if (acc == 0xffffffffu) out[wave] = acc;
}

int main(int argc, char **argv) {
    std::vector<size_t> sizes;
    for (int i = 1; i < argc; ++i) sizes.push_back((size_t)atol(argv[i]));
    if (sizes.empty()) sizes = {117, 1024, 6231};
    uint32_t *out; CHK(hipMalloc(&out, 1 << 24));
    for (size_t mib : sizes) {
        const size_t bytes = mib << 20, n_rows = bytes / 1024;
        const int nbuf = (int)(700 / mib) + 2;
        std::vector<uint4 *> bufs(nbuf);
        for (auto &b : bufs) { CHK(hipMalloc(&b, bytes)); CHK(hipMemset(b, 0x5a, bytes)); }
        for (size_t rpw : {28, 56, 112, 448}) {
            const size_t waves = (n_rows + rpw - 1) / rpw;
            const int grid = (int)((waves + 3) / 4);
            hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
            for (int i = 0; i < nbuf; ++i) hipLaunchKernelGGL(read_kernel<8>, dim3(grid), dim3(256), 0, 0, bufs[i], rpw, n_rows, out);
            CHK(hipDeviceSynchronize());
            const int iters = 40;
            CHK(hipEventRecord(e0));
            for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(read_kernel<8>, dim3(grid), dim3(256), 0, 0, bufs[i % nbuf], rpw, n_rows, out);
            CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / iters;
            printf("%zu MiB, %zu rows/wave, grid %d: %.1f us  %.2f TB/s\n", mib, rpw, grid, us, bytes / us / 1e6);
        }
        for (auto &b : bufs) CHK(hipFree(b));
    }
    return 0;
}
