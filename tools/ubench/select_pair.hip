// An f64 select is two v_cndmask_b32 on one condition, back to back. What does the pair cost among f64 work, in its VOP2
// (condition in vcc) and VOP3 encodings, with and without the compare that sets vcc in front of it?
// hipcc --offload-arch=gfx950 -O2 -o select_pair select_pair.hip && ./select_pair
#include <hip/hip_runtime.h>
#include <cstdio>
#define F "v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4\n"
template <int MODE, int NF> __global__ void k(long long *out, int iters) {
    double a0 = threadIdx.x + 1.0, a1 = a0 + 1.0, a2 = a0 + 2.0, a3 = a0 + 3.0, b = 1.0000001;
    unsigned c = threadIdx.x, c2 = threadIdx.x * 3u, e = 12345u;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (NF == 2) asm volatile(F F : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
            if (NF == 1) asm volatile(F : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
            if (MODE == 0) asm volatile("v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n" : "+v"(c), "+v"(c2) : "v"(e) : "vcc");
            if (MODE == 1) asm volatile("v_cndmask_b32_e64 %0, %0, %2, vcc\n v_cndmask_b32_e64 %1, %1, %2, vcc\n" : "+v"(c), "+v"(c2) : "v"(e) : "vcc");
            if (MODE == 2) asm volatile("v_xor_b32 %0, %0, %2\n v_xor_b32 %1, %1, %2\n" : "+v"(c), "+v"(c2) : "v"(e) : "vcc");
            if (MODE == 3) asm volatile("v_cmp_lt_f64 vcc, %3, %4\n v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n" : "+v"(c), "+v"(c2) : "v"(e), "v"(a0), "v"(b) : "vcc");
            if (MODE == 4) asm volatile("v_cmp_lt_f64 s[20:21], %3, %4\n v_cndmask_b32_e64 %0, %0, %2, s[20:21]\n v_cndmask_b32_e64 %1, %1, %2, s[20:21]\n" : "+v"(c), "+v"(c2) : "v"(e), "v"(a0), "v"(b) : "s20", "s21");
            if (MODE == 5) asm volatile("v_cmp_lt_f64 vcc, %3, %4\n v_xor_b32 %0, %0, %2\n v_xor_b32 %1, %1, %2\n" : "+v"(c), "+v"(c2) : "v"(e), "v"(a0), "v"(b) : "vcc");
        }
    }
    long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
    if (a0 == 12345.678 && a1 == a2 && a3 == a0 && c == 77 && c2 == 78) out[31] = 1;
}
template <int MODE, int NF> void run(const char *name, long long *d) {
    long long h[16]; const int iters = 20000;
    hipLaunchKernelGGL((k<MODE, NF>), dim3(1), dim3(1024), 0, 0, d, iters);
    hipLaunchKernelGGL((k<MODE, NF>), dim3(1), dim3(1024), 0, 0, d, iters);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    long long m = 0; for (int w = 0; w < 16; ++w) m = h[w] > m ? h[w] : m;
    const double groups = (double)iters * 8 * 4; // four waves per SIMD
    printf("%-58s %8.2f cycles per group on the SIMD (four waves)\n", name, (double)m / groups);
}
int main() {
    long long *d; hipMalloc(&d, 32 * 8);
    run<2, 1>("4 fma_f64 + 2 v_xor_b32", d);
    run<0, 1>("4 fma_f64 + 2 v_cndmask_b32_e32 (vcc)", d);
    run<1, 1>("4 fma_f64 + 2 v_cndmask_b32_e64 (vcc)", d);
    run<5, 1>("4 fma_f64 + v_cmp_lt_f64 vcc + 2 v_xor_b32", d);
    run<3, 1>("4 fma_f64 + v_cmp_lt_f64 vcc + 2 v_cndmask_b32_e32", d);
    run<4, 1>("4 fma_f64 + v_cmp_lt_f64 s[20:21] + 2 v_cndmask_b32_e64", d);
    run<2, 2>("8 fma_f64 + 2 v_xor_b32", d);
    run<0, 2>("8 fma_f64 + 2 v_cndmask_b32_e32 (vcc)", d);
    run<3, 2>("8 fma_f64 + v_cmp_lt_f64 vcc + 2 v_cndmask_b32_e32", d);
    run<4, 2>("8 fma_f64 + v_cmp_lt_f64 s[20:21] + 2 v_cndmask_b32_e64", d);
    return 0;
}
