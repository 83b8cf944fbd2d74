// v_mfma_f32_32x32x16_f16 / 16x16x32_f16 on gfx950 for the two-term fp16 split (x = hi + lo, 11 + 11 significant bits, three
// MFMAs per fp32 block: hi*lo' + lo*hi' + hi*hi'): (1) do fp16 SUBNORMAL operands survive (a value below 2^-3 has a subnormal
// low term -- flushing would cut it to 11 bits), (2) cycles per MFMA against the bf16 form, (3) error of a K = 576 contraction
// against float64, three-term bf16 (six MFMAs) vs two-term fp16 (three MFMAs) vs fp32 MFMA.
// build: hipcc -O3 --offload-arch=gfx950 -o ab_build/mfma_f16_probe tools/ubench/mfma_f16_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));

// D[32][32] = A[32][K] B[K][32]; modes: 0 fp32 mfma (32x32x2), 1 bf16 three-term (6), 2 fp16 two-term (3), 3 fp16 hi only
__global__ void contract(const float* A, const float* B, float* D, int K, int mode) {
    const int lane = threadIdx.x, m = lane & 31, hl = lane >> 5;
    f16v acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (mode == 0) {
        for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[m * K + k + hl], B[(k + hl) * 32 + m], acc, 0, 0, 0);
    } else {
        for (int k0 = 0; k0 < K; k0 += 16) {
            float a[8], b[8];
            for (int j = 0; j < 8; ++j) { a[j] = A[m * K + k0 + hl * 8 + j]; b[j] = B[(k0 + hl * 8 + j) * 32 + m]; }
            if (mode == 1) {
                b8 at[3], bt[3];
                for (int j = 0; j < 8; ++j) {
                    float x = a[j]; for (int t = 0; t < 3; ++t) { at[t][j] = (__bf16)x; x -= (float)at[t][j]; }
                    x = b[j];       for (int t = 0; t < 3; ++t) { bt[t][j] = (__bf16)x; x -= (float)bt[t][j]; }
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[0], bt[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[2], bt[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[1], bt[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[0], bt[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[1], bt[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[0], bt[0], acc, 0, 0, 0);
            } else {
                h8 at[2], bt[2];
                for (int j = 0; j < 8; ++j) {
                    at[0][j] = (_Float16)a[j]; at[1][j] = (_Float16)(a[j] - (float)at[0][j]);
                    bt[0][j] = (_Float16)b[j]; bt[1][j] = (_Float16)(b[j] - (float)bt[0][j]);
                }
                if (mode == 2) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(at[0], bt[1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(at[1], bt[0], acc, 0, 0, 0);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(at[0], bt[0], acc, 0, 0, 0);
            }
        }
    }
    // D layout of the 32x32 forms: register r of lane l = row 8 (r / 4) + 4 (l / 32) + r % 4, column l % 32
    for (int r = 0; r < 16; ++r) D[(8 * (r >> 2) + 4 * hl + (r & 3)) * 32 + m] = acc[r];
}
// subnormal operands: A = 2^-20 (fp16 subnormal, 16 quanta), B = 1: D = K * 2^-20 if they survive, 0 if flushed
__global__ void subnormal(float* out) {
    h8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)9.5367431640625e-07f; b[j] = (_Float16)1.0f; }
    f16v acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = c[0]; }
}
template <int F16>
__global__ void rate(float* out, int iters) {
    h8 ah, bh; b8 ab, bb;
    for (int j = 0; j < 8; ++j) { ah[j] = (_Float16)(0.001f * threadIdx.x); bh[j] = (_Float16)1.0f; ab[j] = (__bf16)(0.001f * threadIdx.x); bb[j] = (__bf16)1.0f; }
    f16v c0, c1; for (int r = 0; r < 16; ++r) c0[r] = c1[r] = 0.f;
    long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (F16) { c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c1, 0, 0, 0); }
        else     { c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c1, 0, 0, 0); }
    }
    long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = c0[0] + c1[1];
    if (threadIdx.x == 0) out[64] = (float)(t1 - t0) / (2.f * iters);
}
int main() {
    const int K = 576;
    std::mt19937 rng(1);
    std::normal_distribution<float> nd;
    std::vector<float> A(32 * K), B(K * 32), D(1024);
    for (auto& v : A) v = 0.05f * nd(rng);
    for (auto& v : B) v = std::fmax(0.f, nd(rng));                  // post-ReLU activations
    std::vector<double> ref(1024);
    double mx = 0;
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { double s = 0; for (int k = 0; k < K; ++k) s += (double)A[m * K + k] * B[k * 32 + n]; ref[m * 32 + n] = s; mx = fmax(mx, fabs(s)); }
    float *dA, *dB, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 4096);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    const char* names[4] = {"fp32 mfma 32x32x2", "bf16 three terms (6 mfma)", "fp16 two terms (3 mfma)", "fp16 one term"};
    for (int mode = 0; mode < 4; ++mode) {
        hipLaunchKernelGGL(contract, dim3(1), dim3(64), 0, 0, dA, dB, dD, K, mode);
        hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
        double e = 0, s2 = 0; for (int i = 0; i < 1024; ++i) { e = fmax(e, fabs(D[i] - ref[i])); s2 += (D[i] - ref[i]) * (D[i] - ref[i]); }
        printf("%-28s K = %d: max err / max|ref| = %.3g, rms %.3g\n", names[mode], K, e / mx, sqrt(s2 / 1024) / mx);
    }
    hipLaunchKernelGGL(subnormal, dim3(1), dim3(64), 0, 0, dD);
    hipMemcpy(D.data(), dD, 8, hipMemcpyDeviceToHost);
    printf("subnormal fp16 operand 2^-20 x 1, K = 16 / 32: 32x32x16 gives %g (kept: %g), 16x16x32 gives %g (kept: %g)\n", D[0], 16 * 9.5367431640625e-07, D[1], 32 * 9.5367431640625e-07);
    hipLaunchKernelGGL(rate<1>, dim3(1), dim3(64), 0, 0, dD, 10000);
    hipMemcpy(D.data(), dD, 65 * 4, hipMemcpyDeviceToHost);
    printf("cycles per 32x32x16 f16 mfma: %.2f\n", D[64]);
    hipLaunchKernelGGL(rate<0>, dim3(1), dim3(64), 0, 0, dD, 10000);
    hipMemcpy(D.data(), dD, 65 * 4, hipMemcpyDeviceToHost);
    printf("cycles per 32x32x16 bf16 mfma: %.2f\n", D[64]);
    return 0;
}
