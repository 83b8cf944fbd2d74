// Micro-benchmark: issue rate of the fp32 MFMA shapes used by this repo (registers only, no memory).
// build: hipcc -O3 --offload-arch=gfx950 -o mfma_rate mfma_rate.hip ; run: ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k16(float* out, int iters, float a, float b) {
    f4 acc[NACC];
    for (int q = 0; q < NACC; ++q) acc[q] = f4{0, 0, 0, 0};
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[q], 0, 0, 0);
    float s = 0;
    for (int q = 0; q < NACC; ++q) s += acc[q][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ void k32(float* out, int iters, float a, float b) {
    f16 acc[NACC];
    for (int q = 0; q < NACC; ++q)
        for (int r = 0; r < 16; ++r) acc[q][r] = 0;
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
    float s = 0;
    for (int q = 0; q < NACC; ++q) s += acc[q][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename K>
void run(const char* name, K kern, int nacc, double flop_per_mfma, int waves_per_cu) {
    float* out;
    hipMalloc(&out, 256 * 1024 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid(256 * waves_per_cu / 4), block(256);
    hipLaunchKernelGGL(kern, grid, block, 0, 0, out, 100, 1.0f, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, grid, block, 0, 0, out, iters, 1.0f, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mfmas = (double)grid.x * 4 * iters * nacc;
    printf("%-28s waves/CU=%2d  %.1f TFLOP/s  (%.1f ns per MFMA per wave-stream -> %.1f cycles @2.4GHz per SIMD-slot)\n", name,
           waves_per_cu, mfmas * flop_per_mfma / ms / 1e9, ms * 1e6 / (iters * nacc), ms * 1e6 / (iters * nacc) * 2.4 / (waves_per_cu / 4.0));
    hipFree(out);
}
int main() {
    for (int w : {4, 8, 16}) {
        run("16x16x4 f32, 1 acc", k16<1>, 1, 2048, w);
        run("16x16x4 f32, 2 acc", k16<2>, 2, 2048, w);
        run("16x16x4 f32, 4 acc", k16<4>, 4, 2048, w);
        run("32x32x2 f32, 1 acc", k32<1>, 1, 4096, w);
        run("32x32x2 f32, 2 acc", k32<2>, 2, 4096, w);
        run("32x32x2 f32, 4 acc", k32<4>, 4, 4096, w);
    }
    return 0;
}
