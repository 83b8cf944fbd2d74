// What v_permlane16_swap / v_permlane32_swap return through the clang builtins (gfx950): prints, per lane, the two results
// of swap(v, v) with v = lane id.   hipcc --offload-arch=gfx950 -o permlane_probe permlane_probe.hip && ./permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    const unsigned v = threadIdx.x;
    const auto a = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    out[threadIdx.x * 4 + 0] = a[0];
    out[threadIdx.x * 4 + 1] = a[1];
    out[threadIdx.x * 4 + 2] = b[0];
    out[threadIdx.x * 4 + 3] = b[1];
}
int main() {
    unsigned* d;
    hipMalloc(&d, 64 * 4 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[256];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 5) printf("lane %2d: swap16 -> (%2u, %2u)   swap32 -> (%2u, %2u)\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    return 0;
}
