// Quality metrics of eval_model's scoring loop (eval_models_seq.py:229-258) on the device: the frames never leave HBM.
//
//   mse_loss              = F.mse_loss(pred, target)                       (evaluate/metrics.py:42-43)
//   structural_similarity = skimage.metrics.structural_similarity(a, b)    (evaluate/metrics.py:46-65), called on float32
//                           images with NO data_range: only scikit-image <= 0.18 accepts that call, and there it means
//                           float64 arithmetic, data_range = 2 (dtype range of floats, -1..1), 7x7 uniform window,
//                           K1 = 0.01, K2 = 0.03, sample covariance (NP / (NP - 1)), mean of S over the image cropped by 3.
// scikit-image is not installed in this image: the SSIM here follows the published algorithm (oracle/metrics_oracle.py
// restates it on scipy.ndimage.uniform_filter, the primitive scikit-image itself calls) -- PARITY UNPINNED against
// scikit-image's own output until a fixture from it exists.  LPIPS needs network weights that are absent from the mount.
//
// Both are HBM-bound reductions (8 B read per pixel); partial sums per workgroup in float64, summed in a fixed order by
// a second one-workgroup pass, so results are deterministic.
#pragma once
#include <hip/hip_runtime.h>
#include "common.h"

namespace bde {

__device__ __forceinline__ double block_sum_256(double v, double* sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0)
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w];
    __syncthreads();
    return t;       // valid in thread 0
}

// grid (blocks, N): partial[n][block] = sum over the block's elements of (a - b)^2
__global__ __launch_bounds__(256) void mse_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, long n,
                                                          double* __restrict__ partial) {
    __shared__ double sh[4];
    const float* pa = a + (long)blockIdx.y * n;
    const float* pb = b + (long)blockIdx.y * n;
    double acc = 0.0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float d = pa[i] - pb[i];            // float32 difference and square, like the reference's float32 tensors
        acc += (double)(d * d);
    }
    const double t = block_sum_256(acc, sh);
    if (threadIdx.x == 0) partial[(long)blockIdx.y * gridDim.x + blockIdx.x] = t;
}

// grid (blocks, N): partial sums of the SSIM map S over the valid (H-6) x (W-6) region of image n
__global__ __launch_bounds__(256) void ssim_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, int H, int W,
                                                           double data_range, double* __restrict__ partial) {
    __shared__ double sh[4];
    const int vh = H - 6, vw = W - 6;
    const float* pa = a + (long)blockIdx.y * H * W;
    const float* pb = b + (long)blockIdx.y * H * W;
    const double C1 = (0.01 * data_range) * (0.01 * data_range), C2 = (0.03 * data_range) * (0.03 * data_range);
    const double NP = 49.0, cov_norm = NP / (NP - 1.0);
    double acc = 0.0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < (long)vh * vw; i += (long)gridDim.x * blockDim.x) {
        const int y = (int)(i / vw), x = (int)(i - (long)y * vw);
        double sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
        for (int dy = 0; dy < 7; ++dy) {
            const float* ra = pa + (long)(y + dy) * W + x;
            const float* rb = pb + (long)(y + dy) * W + x;
#pragma unroll
            for (int dx = 0; dx < 7; ++dx) {
                const double u = ra[dx], v = rb[dx];
                sx += u; sy += v; sxx += u * u; syy += v * v; sxy += u * v;
            }
        }
        const double ux = sx / NP, uy = sy / NP, uxx = sxx / NP, uyy = syy / NP, uxy = sxy / NP;
        const double vx = cov_norm * (uxx - ux * ux), vy = cov_norm * (uyy - uy * uy), vxy = cov_norm * (uxy - ux * uy);
        const double A1 = 2 * ux * uy + C1, A2 = 2 * vxy + C2, B1 = ux * ux + uy * uy + C1, B2 = vx + vy + C2;
        acc += (A1 * A2) / (B1 * B2);
    }
    const double t = block_sum_256(acc, sh);
    if (threadIdx.x == 0) partial[(long)blockIdx.y * gridDim.x + blockIdx.x] = t;
}

// one workgroup per image: out[n] = (sum of its partials in index order) / count
__global__ __launch_bounds__(64) void metric_finish_kernel(const double* __restrict__ partial, int nblocks, double count,
                                                           double* __restrict__ out) {
    if (threadIdx.x != 0) return;
    double t = 0.0;
    for (int i = 0; i < nblocks; ++i) t += partial[(long)blockIdx.x * nblocks + i];
    out[blockIdx.x] = t / count;
}

constexpr int METRIC_BLOCKS = 64;

// scratch: device double [N * METRIC_BLOCKS]; out: device double [N]
static int metric_mse_launch(const float* a, const float* b, long n, int N, double* scratch, double* out, hipStream_t s) {
    hipLaunchKernelGGL(mse_partial_kernel, dim3(METRIC_BLOCKS, N), dim3(256), 0, s, a, b, n, scratch);
    hipLaunchKernelGGL(metric_finish_kernel, dim3(N), dim3(64), 0, s, scratch, METRIC_BLOCKS, (double)n, out);
    BDE_HIP(hipGetLastError());
    return BDE_OK;
}
static int metric_ssim_launch(const float* a, const float* b, int H, int W, int N, double data_range, double* scratch, double* out,
                              hipStream_t s) {
    hipLaunchKernelGGL(ssim_partial_kernel, dim3(METRIC_BLOCKS, N), dim3(256), 0, s, a, b, H, W, data_range, scratch);
    hipLaunchKernelGGL(metric_finish_kernel, dim3(N), dim3(64), 0, s, scratch, METRIC_BLOCKS, (double)(H - 6) * (W - 6), out);
    BDE_HIP(hipGetLastError());
    return BDE_OK;
}

}  // namespace bde
