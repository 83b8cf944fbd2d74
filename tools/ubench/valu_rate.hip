// Micro-benchmark: issue cost of the vector-ALU instructions winblock's softmax / p*v loop is made of
// (registers only).  Prints SIMD cycles per wave instruction from s_memtime deltas and from the event time.
// build: hipcc -O3 --offload-arch=gfx950 -o valu_rate valu_rate.hip ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
template <int OP>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* cyc, int iters) {
    float r0 = threadIdx.x * 1e-3f, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {r0, r1}, p1 = {r2, r3}, p2 = {r4, r5}, p3 = {r6, r7}, m = {0.999f, 1.001f}, c = {1e-3f, 2e-3f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), q0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) {            // v_fma_f32, eight independent chains
            asm volatile(REP8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(m[0]), "v"(c[0]));
        } else if (OP == 1) {     // v_pk_fma_f32, four independent chains (8 instructions x 8)
            asm volatile(REP8("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                              "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(m), "v"(c));
        } else if (OP == 2) {     // v_exp_f32
            asm volatile(REP8("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                              "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n")
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7));
        } else if (OP == 3) {     // v_pk_add_f32
            asm volatile(REP8("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                              "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(c));
        } else if (OP == 4) {     // v_max3_f32
            asm volatile(REP8("v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n"
                              "v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9\n")
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(m[0]), "v"(c[0]));
        } else if (OP == 5) {     // v_pk_fma_f32 with the first source broadcast from one half (op_sel)
            asm volatile(REP8("v_pk_fma_f32 %0, %4, %5, %0 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %1, %4, %5, %1 op_sel:[1,0,0]\n"
                              "v_pk_fma_f32 %2, %4, %5, %2 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %3, %4, %5, %3 op_sel:[1,0,0]\n"
                              "v_pk_fma_f32 %0, %4, %5, %0 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %1, %4, %5, %1 op_sel:[1,0,0]\n"
                              "v_pk_fma_f32 %2, %4, %5, %2 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %3, %4, %5, %3 op_sel:[1,0,0]\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(m), "v"(c));
        } else if (OP == 6) {     // v_pk_mul_f32
            asm volatile(REP8("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                              "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(m));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), q1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + p0[0] + p0[1] + p1[0] + p1[1] + p2[0] + p2[1] + p3[0] + p3[1];
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; cyc[1024 + blockIdx.x] = q1 - q0; }
}

template <typename K>
void run(const char* name, K kern, int threads, int wgs = 256) {
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4);
    hipMalloc(&cyc, 2048 * 8);
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(threads), 0, 0, out, cyc, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2048];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0, real = 0;
    for (int i = 0; i < wgs; ++i) { avg += (double)h[i]; real += (double)h[1024 + i]; }
    avg /= wgs; real /= wgs;
    const double instr_per_simd = (double)iters * 64 * (threads / 256);     // wave instructions issued per SIMD
    printf("%-30s wgs=%3d waves/SIMD=%d  %.2f ticks/instr/SIMD  in-kernel %.1f us (s_memrealtime) -> clock %.2f GHz, %.2f ns/instr/SIMD; event %.1f us\n",
           name, wgs, threads / 256, avg / instr_per_simd, real / 100.0, avg / (real * 10.0), real * 10.0 / instr_per_simd, ms * 1e3);
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int threads : {256, 1024}) {
        run("v_fma_f32", k<0>, threads);
        run("v_pk_fma_f32", k<1>, threads);
        run("v_pk_fma_f32 op_sel broadcast", k<5>, threads);
        run("v_pk_add_f32", k<3>, threads);
        run("v_pk_mul_f32", k<6>, threads);
        run("v_exp_f32", k<2>, threads);
        run("v_max3_f32", k<4>, threads);
    }
    run("v_fma_f32", k<0>, 1024, 128);
    run("v_fma_f32", k<0>, 1024, 512);
    run("v_fma_f32", k<0>, 512, 256);
    return 0;
}
