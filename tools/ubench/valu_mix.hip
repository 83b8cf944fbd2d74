// Micro-benchmark: the instruction mix of winblock's softmax / p*v inner loop (4 sub, 4 exp, 4 add, 16 fmac per key tile)
// on 1, 2 and 4 waves per SIMD, to see whether waves sharing a SIMD overlap their vector issue on this mix.
// build: hipcc -O3 --offload-arch=gfx950 -o valu_mix valu_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* cyc, int iters) {
    float s0 = threadIdx.x * 1e-3f, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3, mx = 0.5f;
    float v0 = 1.f, v1 = 1.5f, v2 = 0.25f, v3 = 0.75f;
    float l = 0, o0 = 0, o1 = 0, o2 = 0, o3 = 0, p0, p1, p2, p3;
    const unsigned long long q0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#define SUBS "v_sub_f32 %5, %9, %13\n v_sub_f32 %6, %10, %13\n v_sub_f32 %7, %11, %13\n v_sub_f32 %8, %12, %13\n"
#define EXPS "v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n v_exp_f32 %8, %8\n"
#define MULS "v_mul_f32 %5, %5, %14\n v_mul_f32 %6, %6, %14\n v_mul_f32 %7, %7, %14\n v_mul_f32 %8, %8, %14\n"
#define ACC(p) "v_add_f32 %0, %0, " p "\n v_fmac_f32 %1, " p ", %14\n v_fmac_f32 %2, " p ", %15\n v_fmac_f32 %3, " p ", %16\n v_fmac_f32 %4, " p ", %17\n"
#define OPS : "+v"(l), "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3), "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3) \
            : "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(mx), "v"(v0), "v"(v1), "v"(v2), "v"(v3)
        if (OP == 0) {          // as in winblock
            asm volatile(SUBS EXPS ACC("%5") ACC("%6") ACC("%7") ACC("%8") OPS);
        } else if (OP == 1) {   // exponentials replaced by multiplies
            asm volatile(SUBS MULS ACC("%5") ACC("%6") ACC("%7") ACC("%8") OPS);
        } else if (OP == 2) {   // no accumulate dependence on the exponentials: the FMAs use the s registers
            asm volatile(SUBS EXPS ACC("%9") ACC("%10") ACC("%11") ACC("%12") OPS);
        } else if (OP == 3) {   // exponentials only
            asm volatile(SUBS EXPS OPS);
        } else if (OP == 4) {   // accumulates only
            asm volatile(ACC("%9") ACC("%10") ACC("%11") ACC("%12") OPS);
        }
    }
    const unsigned long long q1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = l + o0 + o1 + o2 + o3 + p0 + p1 + p2 + p3;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = q1 - q0;
}

template <typename K>
void run(const char* name, K kern, int threads, int ninstr) {
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4);
    hipMalloc(&cyc, 256 * 16 * 8);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, cyc, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    static unsigned long long h[256 * 16];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const int nw = threads / 64;
    double mn = 1e30, mxv = 0;
    for (int b = 0; b < 256; ++b)
        for (int w = 0; w < nw; ++w) { const double t = (double)h[b * 16 + w] / 100.0; mn = t < mn ? t : mn; mxv = t > mxv ? t : mxv; }
    const double per_simd = (double)iters * ninstr * (threads / 256);
    printf("%-28s waves/SIMD=%d  event %.0f us = %.2f ns per wave-instruction per SIMD; a wave's own span %.0f..%.0f us\n", name, threads / 256,
           ms * 1e3, ms * 1e6 / per_simd, mn, mxv);
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int threads : {256, 512, 1024}) {
        run("winblock mix (28)", k<0>, threads, 28);
        run("exp -> mul (28)", k<1>, threads, 28);
        run("fma independent of exp (28)", k<2>, threads, 28);
        run("sub + exp only (8)", k<3>, threads, 8);
        run("accumulates only (20)", k<4>, threads, 20);
    }
    return 0;
}
