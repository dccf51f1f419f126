#include <hip/hip_runtime.h>
#include <cstdio>
#define F "v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4\n"
#define C32 "v_cndmask_b32 %5, %5, %6, vcc\n"
#define C64 "v_cndmask_b32_e64 %5, %5, %6, vcc\n"
#define XOR "v_xor_b32 %5, %5, %6\n"
template <int MODE, int NF> __global__ void k(long long *out, int iters) {
    double a0 = threadIdx.x + 1.0, a1 = a0 + 1.0, a2 = a0 + 2.0, a3 = a0 + 3.0, b = 1.0000001;
    unsigned c = threadIdx.x, e = 12345u;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (NF == 2) asm volatile(F F : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c), "v"(e));
            if (NF == 1) asm volatile(F : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c), "v"(e));
            if (MODE == 0) asm volatile("v_cndmask_b32 %4, %4, %6, vcc\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(c) : "v"(b), "v"(e) : "vcc");
            if (MODE == 1) asm volatile("v_cndmask_b32_e64 %4, %4, %6, vcc\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(c) : "v"(b), "v"(e) : "vcc");
            if (MODE == 2) asm volatile("v_xor_b32 %4, %4, %6\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(c) : "v"(b), "v"(e) : "vcc");
        }
    }
    long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
    if (a0 == 12345.678 && a1 == a2 && a3 == a0 && c == 77) out[31] = 1;
}
template <int MODE, int NF> void run(const char *name, long long *d) {
    long long h[16]; const int iters = 20000;
    hipLaunchKernelGGL((k<MODE, NF>), dim3(1), dim3(1024), 0, 0, d, iters);
    hipLaunchKernelGGL((k<MODE, NF>), dim3(1), dim3(1024), 0, 0, d, iters);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    long long m = 0; for (int w = 0; w < 16; ++w) m = h[w] > m ? h[w] : m;
    const double n_inst = (double)iters * 8 * (4 * NF + 1) * 4;
    printf("%-44s %7.3f cycles per wave64 instruction on the SIMD (four waves)\n", name, (double)m / n_inst);
}
int main() {
    long long *d; hipMalloc(&d, 32 * 8);
    run<0, 2>("8 fma_f64 : 1 v_cndmask_b32_e32 (vcc)", d);
    run<1, 2>("8 fma_f64 : 1 v_cndmask_b32_e64 (vcc)", d);
    run<2, 2>("8 fma_f64 : 1 v_xor_b32", d);
    run<0, 1>("4 fma_f64 : 1 v_cndmask_b32_e32 (vcc)", d);
    run<1, 1>("4 fma_f64 : 1 v_cndmask_b32_e64 (vcc)", d);
    run<2, 1>("4 fma_f64 : 1 v_xor_b32", d);
    return 0;
}
