// Layout and rate probe of v_mfma_f32_4x4x1_16B_f32 (16 blocks of 4x4 outer products) on gfx950.
// build: hipcc -O3 --offload-arch=gfx950 -o gpurun_out/mfma4x4 tools/ubench/mfma4x4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out) {
    const int lane = threadIdx.x;
    // A value encodes (lane), B value encodes (lane): a = 1 + lane, b = 100 + lane  -> d = a*b identifies the pair
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(1 + lane), (float)(100 + lane), acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = acc[r];
}
__global__ void rate(float* out, int iters) {
    f4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    float x = threadIdx.x * 0.001f, y = 1.0f;
    long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a3, 0, 0, 0);
    }
    long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
    if (threadIdx.x == 0) out[64] = (float)(t1 - t0) / (4.f * iters);
}
int main() {
    float* d; hipMalloc(&d, 4096);
    float h[260];
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, 256 * 4, hipMemcpyDeviceToHost);
    // decode: d = (1+la)*(100+lb)  ->  find la, lb
    for (int lane = 0; lane < 64; lane += 1) {
        printf("lane %2d:", lane);
        for (int r = 0; r < 4; ++r) {
            int v = (int)h[lane * 4 + r], fa = -1, fb = -1;
            for (int la = 0; la < 64 && fa < 0; ++la)
                for (int lb = 0; lb < 64; ++lb)
                    if ((1 + la) * (100 + lb) == v) { fa = la; fb = lb; break; }
            printf("  r%d=A[l%d]*B[l%d]", r, fa, fb);
        }
        printf("\n");
        if (lane == 7) lane = 55;
    }
    hipLaunchKernelGGL(rate, dim3(1), dim3(64), 0, 0, d, 10000);
    hipMemcpy(h, d, 65 * 4, hipMemcpyDeviceToHost);
    printf("cycles per 4x4x1 mfma (one wave, 4 independent accumulators): %.2f\n", h[64]);
    return 0;
}
