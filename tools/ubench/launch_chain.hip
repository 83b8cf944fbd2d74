// Cost of a dependent kernel in a linear chain: eager launches vs one hipGraph replay (gfx950).
// build: hipcc -O3 --offload-arch=gfx950 -o ab_build/launch_chain tools/ubench/launch_chain.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void tiny(float* p, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0001f + 1.f;
}
int main() {
    const int N = 176 * 256, CH = 500;
    float* d; hipMalloc(&d, N * 4); hipMemset(d, 0, N * 4);
    hipStream_t s; hipStreamCreate(&s);
    for (int grid : {1, 176, 704}) {
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, s, d, N);
        hipStreamSynchronize(s);
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < CH; ++i) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, s, d, N);
        hipStreamSynchronize(s);
        double eager = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / CH;
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < CH; ++i) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, s, d, N);
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, s); hipStreamSynchronize(s);
        t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, s);
        hipStreamSynchronize(s);
        double graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (5 * CH);
        printf("grid %4d x 256 threads: %.2f us per dependent kernel eager, %.2f us in a graph\n", grid, eager, graph);
    }
    return 0;
}
