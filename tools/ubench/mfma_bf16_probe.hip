// Operand layout and rate probe of v_mfma_f32_16x16x32_bf16 on gfx950 (for the bf16-split emulation of fp32 products).
// build: hipcc -O3 --offload-arch=gfx950 -o ab_build/mfma_bf16_probe tools/ubench/mfma_bf16_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef float f4 __attribute__((ext_vector_type(4)));
// A[16][32], B[32][16] row-major floats holding bf16-representable values; D = A*B
__global__ void probe(const float* A, const float* B, float* D) {
    const int lane = threadIdx.x, m = lane & 15, kg = lane >> 4;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (__bf16)A[m * 32 + kg * 8 + j];          // assumed: lane (row m, k-group kg) holds k = 8kg..8kg+7
        b[j] = (__bf16)B[(kg * 8 + j) * 16 + m];        // assumed: lane (col m, k-group kg)
    }
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((lane >> 4) * 4 + r) * 16 + (lane & 15)] = acc[r];   // assumed C/D map of 16x16x4
}
__global__ void rate(float* out, int iters) {
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * threadIdx.x); b[j] = (__bf16)1.0f; }
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
    }
    long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    if (threadIdx.x == 0) out[64] = (float)(t1 - t0) / (4.f * iters);
}
__global__ void rate_f32(float* out, int iters) {
    float a = 0.001f * threadIdx.x, b = 1.0f;
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
    }
    long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    if (threadIdx.x == 0) out[64] = (float)(t1 - t0) / (4.f * iters);
}
int main() {
    float hA[16 * 32], hB[32 * 16], hD[256], ref[256];
    for (int i = 0; i < 512; ++i) { hA[i] = (float)((i * 7) % 13 - 6); hB[i] = (float)((i * 5) % 11 - 5) * 0.5f; }
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { double s = 0; for (int k = 0; k < 32; ++k) s += (double)hA[m * 32 + k] * hB[k * 16 + n]; ref[m * 16 + n] = (float)s; }
    float *dA, *dB, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, 4096);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    double err = 0; for (int i = 0; i < 256; ++i) err = fmax(err, fabs(hD[i] - ref[i]));
    printf("layout check: max |D - A*B| = %g (0 = the assumed operand layout is right)\n", err);
    hipLaunchKernelGGL(rate, dim3(1), dim3(64), 0, 0, dD, 10000);
    hipMemcpy(hD, dD, 65 * 4, hipMemcpyDeviceToHost);
    printf("cycles per 16x16x32 bf16 mfma (one wave, 4 accumulators): %.2f\n", hD[64]);
    hipLaunchKernelGGL(rate_f32, dim3(1), dim3(64), 0, 0, dD, 10000);
    hipMemcpy(hD, dD, 65 * 4, hipMemcpyDeviceToHost);
    printf("cycles per 16x16x4 f32 mfma, same loop: %.2f  (K = 4 against K = 32 per instruction)\n", hD[64]);
    return 0;
}
