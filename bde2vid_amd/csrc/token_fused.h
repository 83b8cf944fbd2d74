// Fused per-pixel part of a temporal-attention block (everything after the softmax that is
// pointwise over pixels), one launch per block instead of four:
//
//   x1 = shortcut + proj(attn)                 DTransformer.py:204,299   (uncovered pixels of a dilated
//                                                                          block get the shortcut only, :79-82)
//   x2 = x1 + fc2(GELU(fc1(LayerNorm2(x1))))   DTransformer.py:279-283,304
//   [last block]  x2 += merged[t]              V5.py:166
//   [not last]    q|k|v of the NEXT block = W'(LayerNorm_q/kv(x2))   DTransformer.py:183-190
//
// A workgroup owns 32 consecutive pixels and all channels; the intermediates (x1, the 4C hidden
// activations, x2) never leave LDS.  Contractions run on v_mfma_f32_16x16x4_f32 (exact fp32):
//   A = packed weights [co16 tile][k/4][64 lanes] straight from L2 (each fragment feeds both 16-pixel
//       tiles), B = LDS [k][pixel] (row pitch 48 floats -> the four k-rows of a fragment hit
//       disjoint banks), D: row = co, col = pixel.
// The LayerNorms are folded into the weights exactly as in pw_gemm.h; the per-pixel statistics are
// taken from the LDS tile.  Used where a level has enough pixels to fill the chip with 32-pixel
// tiles (level 0 of config A: 345 tiles); small maps (level 2: 22 tiles) keep the split-K GEMMs.
#pragma once
#include <hip/hip_runtime.h>
#include "common.h"
#include "conv_mfma.h"

namespace bde {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct TokenArgs {
    const float* ao;        // [B][C][HW] attention output (pre-proj)
    const float* x;         // [B][C][HW] shortcut (block input)
    const float* addres;    // [B][C][HW] added to x2 (merged[t]) or nullptr
    float* x2;              // [B][C][HW] block output
    float* qkv;             // [B][3C][HW] q|k|v of the next block, or nullptr
    const float *wproj, *bproj;                 // packed16 [C/16][C/4][64], [C]
    const float *wfc1, *bfc1, *sfc1;            // packed16 [4C/16][C/4][64], [4C] bias', [4C] row sums
    const float *wfc2, *bfc2;                   // packed16 [C/16][4C/4][64], [C]
    const float *wqkv, *bqkv, *sqkv;            // packed16 [3C/16][C/4][64], [3C], [3C] (next block)
    long bs_c;              // batch stride of the C-channel tensors (C*HW)
    long bs_qkv;            // batch stride of qkv (3C*HW)
    int C, HW;
    int mask_w, mask_pt, mask_pl;               // dilated coverage mask (0 = plain block)
};

constexpr int TOK_PT = 32;       // pixels per workgroup
constexpr int TOK_PITCH = 48;    // LDS row pitch (floats)

// One contraction phase: rows [0, nct*16) x 32 pixels, K = 4*nk4, B operand in LDS.
// Wave w owns co-tiles w, w+4, ...; `epi(row, px, value)` receives every output element.
template <typename Epi>
__device__ __forceinline__ void tok_gemm(const float* __restrict__ wpk, int nct, int nk4, const float* ldsB, int wave,
                                         int lane, Epi epi) {
    const int krow = lane >> 4, col = lane & 15;
    const float* bptr = ldsB + krow * TOK_PITCH + col;
    for (int ct = wave; ct < nct; ct += 4) {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        const float* wp = wpk + ((long)ct * nk4) * 64 + lane;
        int k4 = 0;
        for (; k4 + 8 <= nk4; k4 += 8) {
            float av[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) av[u] = wp[(long)(k4 + u) * 64];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float b0 = bptr[(k4 + u) * 4 * TOK_PITCH];
                const float b1 = bptr[(k4 + u) * 4 * TOK_PITCH + 16];
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], b1, acc1, 0, 0, 0);
            }
        }
        for (; k4 < nk4; ++k4) {
            const float av = wp[(long)k4 * 64];
            const float b0 = bptr[k4 * 4 * TOK_PITCH];
            const float b1 = bptr[k4 * 4 * TOK_PITCH + 16];
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1, acc1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = ct * 16 + krow * 4 + r;     // C/D map of 16x16x4: row = (lane>>4)*4 + reg
            epi(row, col, acc0[r]);
            epi(row, col + 16, acc1[r]);
        }
    }
}

// per-pixel mean / rstd over C channels of an LDS tile [C][TOK_PITCH]; result in stat[0..31], stat[32..63]
__device__ __forceinline__ void tok_ln_stats(const float* tile, int C, float* stat, int tid) {
    const int px = tid & 31, part = tid >> 5;          // 8 parts of the channel range
    float s1 = 0.f, s2 = 0.f;
    for (int c = part; c < C; c += 8) {
        const float v = tile[c * TOK_PITCH + px];
        s1 += v;
        s2 += v * v;
    }
    // lanes px and px+32 of a wave hold two parts; fold them, then reduce the 4 waves through LDS
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 32);
    const int wave = tid >> 6;
    if ((tid & 63) < 32) {
        stat[64 + wave * 64 + px] = s1;
        stat[64 + wave * 64 + 32 + px] = s2;
    }
    __syncthreads();
    if (tid < 32) {
        float u = 0.f, v = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            u += stat[64 + w * 64 + tid];
            v += stat[64 + w * 64 + 32 + tid];
        }
        const float mean = u / (float)C;
        const float var = fmaxf(v / (float)C - mean * mean, 0.f);
        stat[tid] = mean;
        stat[32 + tid] = 1.0f / sqrtf(var + 1e-5f);
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void token_fused_kernel(const TokenArgs a) {
    extern __shared__ __align__(16) float lds[];
    const int C = a.C, HID = 4 * a.C, HW = a.HW;
    float* tA = lds;                                   // [C][PITCH]   attention output
    float* tX = tA + C * TOK_PITCH;                    // [C][PITCH]   x -> x1 -> x2 (in place)
    float* tH = tX + C * TOK_PITCH;                    // [4C][PITCH]  hidden
    float* stat = tH + HID * TOK_PITCH;                // 64 + 4*64 floats
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const int p0 = blockIdx.x * TOK_PT;
    const float* aob = a.ao + b * a.bs_c;
    const float* xb = a.x + b * a.bs_c;

    // ---- load the two input tiles (rows of 32 pixels = 128 B) ------------------------------------
    {
        const int px = tid & 31;
        const int p = min(p0 + px, HW - 1);
        for (int c = tid >> 5; c < C; c += 8) {
            tA[c * TOK_PITCH + px] = aob[(long)c * HW + p];
            tX[c * TOK_PITCH + px] = xb[(long)c * HW + p];
        }
    }
    __syncthreads();

    // ---- x1 = x + proj(ao) (masked) ----------------------------------------------------------------
    {
        unsigned covered = 0xffffffffu;                 // bit per pixel column handled by this lane
        if (a.mask_w > 0) {
            covered = 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int p = p0 + (lane & 15) + 16 * h;
                const int y = p / a.mask_w, xx = p - y * a.mask_w;
                const int rr = y + a.mask_pt, cc = xx + a.mask_pl;
                if (!((rr < 7 && (rr & 1)) || (cc < 7 && (cc & 1)))) covered |= 1u << h;
            }
        }
        tok_gemm(a.wproj, C / 16, C / 4, tA, wave, lane, [&](int row, int px, float v) {
            const bool cov = (covered >> (px >> 4)) & 1u;
            float* d = tX + row * TOK_PITCH + px;
            *d = *d + (cov ? v + a.bproj[row] : 0.f);
        });
    }
    __syncthreads();

    // ---- hidden = GELU(fc1(LN2(x1))) -------------------------------------------------------------
    tok_ln_stats(tX, C, stat, tid);
    tok_gemm(a.wfc1, HID / 16, C / 4, tX, wave, lane, [&](int row, int px, float v) {
        const float y = stat[32 + px] * (v - stat[px] * a.sfc1[row]) + a.bfc1[row];
        tH[row * TOK_PITCH + px] = 0.5f * y * (1.f + erff(y * 0.70710678118654752440f));
    });
    __syncthreads();

    // ---- x2 = x1 + fc2(hidden) (+ merged[t]) -------------------------------------------------------
    {
        float* x2b = a.x2 + b * a.bs_c;
        const float* adb = a.addres ? a.addres + b * a.bs_c : nullptr;
        tok_gemm(a.wfc2, C / 16, HID / 4, tH, wave, lane, [&](int row, int px, float v) {
            float* d = tX + row * TOK_PITCH + px;
            const float x2 = *d + v + a.bfc2[row];
            *d = x2;                                   // LN statistics / q|k|v of the next block read this
            const int p = p0 + px;
            if (p < HW) {
                const long o = (long)row * HW + p;
                x2b[o] = adb ? x2 + adb[o] : x2;
            }
        });
    }
    if (a.qkv == nullptr) return;
    __syncthreads();

    // ---- q|k|v of the next block from x2 ------------------------------------------------------------
    tok_ln_stats(tX, C, stat, tid);
    {
        float* qb = a.qkv + b * a.bs_qkv;
        tok_gemm(a.wqkv, 3 * C / 16, C / 4, tX, wave, lane, [&](int row, int px, float v) {
            const int p = p0 + px;
            if (p < HW) qb[(long)row * HW + p] = stat[32 + px] * (v - stat[px] * a.sqkv[row]) + a.bqkv[row];
        });
    }
}

static inline size_t token_lds_bytes(int C) {
    return ((size_t)(6 * C) * TOK_PITCH + 64 + 4 * 64) * sizeof(float);
}

static inline int token_launch(const TokenArgs& a, int B, hipStream_t stream) {
    const size_t lds = token_lds_bytes(a.C);
    if (lds > 64 * 1024) {
        static bool raised = false;
        if (!raised) {
            BDE_HIP(hipFuncSetAttribute((const void*)token_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        160 * 1024));
            raised = true;
        }
    }
    dim3 grid(cdiv(a.HW, TOK_PT), B);
    hipLaunchKernelGGL(token_fused_kernel, grid, dim3(256), lds, stream, a);
    BDE_HIP(hipGetLastError());
    return BDE_OK;
}

}  // namespace bde
