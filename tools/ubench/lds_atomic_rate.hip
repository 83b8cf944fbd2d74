// LDS atomic throughput on gfx950: float add vs u32 add vs u64 add, random cells of a 27000-entry tile (the voxel tile), one
// workgroup of 1024 threads per CU.   build: hipcc -O3 --offload-arch=gfx950 -o ab_build/lds_atomic_rate tools/ubench/lds_atomic_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ __launch_bounds__(1024) void k(const unsigned* idx, int n_per_thread, int cells, float* out) {
    extern __shared__ float tile[];
    for (int i = threadIdx.x; i < cells * (MODE == 2 ? 2 : 1); i += blockDim.x) tile[i] = 0.f;
    __syncthreads();
    const unsigned* p = idx + (size_t)blockIdx.x * n_per_thread * 1024 + threadIdx.x;
    for (int j = 0; j < n_per_thread; j += 4) {
        unsigned a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = p[(j + u) * 1024];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE == 0) atomicAdd(tile + a[u], 1.0f);
            if (MODE == 1) atomicAdd(reinterpret_cast<unsigned*>(tile) + a[u], 3u);
            if (MODE == 2) atomicAdd(reinterpret_cast<unsigned long long*>(tile) + a[u], 3ull);
            if (MODE == 3) tile[a[u]] = 1.0f;                      // plain store: the LDS access pattern without the atomic
        }
    }
    __syncthreads();
    float s = 0.f;
    for (int i = threadIdx.x; i < cells; i += blockDim.x) s += tile[i];
    if (s == 12345.f) out[0] = s;
}
int main() {
    const int cells = 13000, npt = 256, blocks = 256;
    std::vector<unsigned> h((size_t)blocks * npt * 1024);
    unsigned x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (x >> 8) % cells; }
    unsigned* d; float* o;
    hipMalloc(&d, h.size() * 4); hipMalloc(&o, 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto kern, const char* name) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        for (int r = 0; r < 2; ++r) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), cells * 8, 0, d, npt, cells, o);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double ops = (double)blocks * npt * 1024;
        printf("%-10s %.3f ms  %.1f G ops/s chip-wide, %.2f lane-ops per cycle per CU at 2.1 GHz\n", name, ms, ops / ms / 1e6, ops / blocks / (ms * 1e-3 * 2.1e9));
    };
    run(k<0>, "f32 add"); run(k<1>, "u32 add"); run(k<2>, "u64 add"); run(k<3>, "store");
    return 0;
}
