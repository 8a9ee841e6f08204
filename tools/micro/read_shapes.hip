#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
// MODE 0: NB=8 depth 1 (the first probe). MODE 1: NB=4, three batches ahead, unconditional. MODE 2: MODE 1 + flag byte load per batch.
// MODE 3: MODE 2 + 11 KB LDS + a staging load and __syncthreads at the start.
template <int MODE>
__global__ __launch_bounds__(256) void read_kernel(const uint4 *src, int rows_per_wave, size_t n_rows, uint32_t *out, const uint32_t *aux) {
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t r0 = wave * rows_per_wave;
    size_t r1 = r0 + rows_per_wave; if (r1 > n_rows) r1 = n_rows;
    uint32_t acc = 0;
    if (MODE == 0) {
        constexpr int NB = 8;
        uint4 cur[NB], nxt[NB];
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
    } else {
        constexpr int NB = 4;
        struct B { uint4 r[NB]; uint32_t flag; };
        B b0, b1, b2, b3;
        const int n_items = r0 < r1 ? (int)((r1 - r0 + NB - 1) / NB) : 0;
        const uint4 *dummy = src + r0 * 64;
        auto load = [&](B &b, int item) {
            const bool past = item >= n_items;
            const size_t row = r0 + (size_t)(past ? 0 : item) * NB;
#pragma unroll
            for (int u = 0; u < NB; ++u) b.r[u] = (past || row + u >= r1) ? *dummy : src[(row + u) * 64 + lane];
            b.flag = MODE >= 2 ? *reinterpret_cast<const unsigned char *>(dummy) : 0u;
        };
        load(b0, 0); load(b1, 1); load(b2, 2);
        if (MODE == 3) { lds[threadIdx.x] = aux[threadIdx.x]; __syncthreads(); acc += lds[(threadIdx.x * 7) & 255]; }
        auto step = [&](B &cur, B &spare, int item) {
            load(spare, item + 3);
            asm volatile("" :: "v"(cur.flag), "v"(cur.r[3].x));
#pragma unroll
            for (int u = 0; u < NB; ++u) acc += __popc(cur.r[u].x) + __popc(cur.r[u].y) + __popc(cur.r[u].z) + __popc(cur.r[u].w);
        };
        for (int item = 0; item < n_items; item += 4) {
            step(b0, b3, item); if (item + 1 >= n_items) break;
            step(b1, b0, item + 1); if (item + 2 >= n_items) break;
            step(b2, b1, item + 2); if (item + 3 >= n_items) break;
            step(b3, b2, item + 3);
        }
    }
    if (acc == 0xffffffffu) out[wave] = acc;
}
template <int MODE> void run(size_t mib, int rpw) {
    const size_t bytes = mib << 20, n_rows = bytes / 1024;
    const int nbuf = (int)(700 / mib) + 2;
    std::vector<uint4 *> bufs(nbuf);
    for (auto &b : bufs) { CHK(hipMalloc(&b, bytes)); CHK(hipMemset(b, 0x5a, bytes)); }
    uint32_t *out, *aux; CHK(hipMalloc(&out, 1 << 24)); CHK(hipMalloc(&aux, 4096)); CHK(hipMemset(aux, 0, 4096));
    const size_t waves = (n_rows + rpw - 1) / rpw; const int grid = (int)((waves + 3) / 4);
    const size_t lds = MODE == 3 ? 11 * 1024 : 0;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int i = 0; i < nbuf; ++i) hipLaunchKernelGGL(read_kernel<MODE>, dim3(grid), dim3(256), lds, 0, bufs[i], rpw, n_rows, out, aux);
    CHK(hipDeviceSynchronize());
    const int iters = 40;
    CHK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(read_kernel<MODE>, dim3(grid), dim3(256), lds, 0, bufs[i % nbuf], rpw, n_rows, out, aux);
    CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("mode %d, %zu MiB, %d rows/wave, grid %d: %.1f us  %.2f TB/s\n", MODE, mib, rpw, grid, ms * 1e3 / iters, bytes / (ms * 1e3 / iters) / 1e6);
    for (auto &b : bufs) CHK(hipFree(b));
    CHK(hipFree(out)); CHK(hipFree(aux));
}
int main() {
    run<0>(117, 28); run<1>(117, 28); run<2>(117, 28); run<3>(117, 28);
    run<0>(6231, 1512); run<1>(6231, 1512); run<3>(6231, 1512);
    return 0;
}
