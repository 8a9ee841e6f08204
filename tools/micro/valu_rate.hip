// Microbenchmark: SIMD cycles per wave64 instruction for fp64 add / mul / fma, int32 add and v_cndmask on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int OP>
__global__ __launch_bounds__(256) void k(double *out, int iters, double seed) {
    double a[8]; uint32_t b[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 1e-3 + i; b[i] = threadIdx.x + i; }
    const double c = seed * 0.5, d = 1.0000001;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) a[i] = a[i] + c;
            if (OP == 1) a[i] = a[i] * d;
            if (OP == 2) a[i] = __builtin_fma(a[i], d, c);
            if (OP == 3) b[i] = b[i] + 0x9e3779b9u + it;
            if (OP == 4) a[i] = (b[i] & 1) ? a[i] : c;   // cndmask x2 (+and/cmp)
        }
    }
    double s = 0; for (int i = 0; i < 8; ++i) s += a[i] + b[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    double *out; hipMalloc(&out, 8 * 256 * 4096);
    const int iters = 20000, blocks = 256 * 8;   // 8 blocks of 4 waves per CU = 8 waves per SIMD
    const char *names[] = {"v_add_f64", "v_mul_f64", "v_fma_f64", "v_add_u32(+it)", "select f64"};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int op = 0; op < 5; ++op) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (op == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.5);
            if (op == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.5);
            if (op == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.5);
            if (op == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.5);
            if (op == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.5);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // wave-instructions per SIMD = (blocks*4 waves / 1024 SIMDs) * iters * 8
        double winstr = (double)blocks * 4 / 1024.0 * iters * 8;
        printf("%-16s %.3f ms  -> %.2f ns per wave-instr per SIMD (x2.4 GHz = %.2f cycles)\n", names[op], ms, ms * 1e6 / winstr, ms * 1e6 / winstr * 2.4);
    }
    return 0;
}
