// What v_permlane16_swap / v_permlane32_swap and the DPP row controls do on gfx950, through the clang builtins: prints, per lane, the
// two results of swap(x, y) with x = lane, y = 100 + lane, and the row reductions / cross-row sums of winblock.h built on them.
//   hipcc -O3 --offload-arch=gfx950 -I bde2vid_amd/csrc -o ab_build/permlane_probe tools/ubench/permlane_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL>
__device__ float dpp(float v) {
    const int x = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(x, x, CTRL, 0xf, 0xf, true));
}
__global__ void k(float* out) {
    const unsigned lane = threadIdx.x;
    const auto a = __builtin_amdgcn_permlane16_swap(lane, 100 + lane, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(lane, 100 + lane, false, false);
    float* o = out + lane * 8;
    o[0] = a[0]; o[1] = a[1]; o[2] = b[0]; o[3] = b[1];
    o[4] = dpp<0xB1>((float)lane); o[5] = dpp<0x4E>((float)lane); o[6] = dpp<0x141>((float)lane); o[7] = dpp<0x140>((float)lane);
}
int main() {
    float* d;
    hipMalloc(&d, 64 * 8 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[512];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 3)
        printf("lane %2d: swap16 -> (%3.0f, %3.0f)  swap32 -> (%3.0f, %3.0f)  quad[1,0,3,2] %2.0f  quad[2,3,0,1] %2.0f  half_mirror %2.0f  mirror %2.0f\n", l,
               h[l * 8], h[l * 8 + 1], h[l * 8 + 2], h[l * 8 + 3], h[l * 8 + 4], h[l * 8 + 5], h[l * 8 + 6], h[l * 8 + 7]);
    return 0;
}
