// Temporal window-attention core (softmax(q k^T + bias) v per 7x7 window and head).
//
// Restates the middle of WindowAttention3D.forward (DTransformer.py:192-203) together with the
// addressing of window_partition / window_reverse (DTransformer.py:40-83) and the zero padding of
// SwinTransformerBlock3D.forward_part1 (:260-264): windows are never materialised.  q/k/v live as
// NCHW planes produced by the 1x1-conv GEMMs (conv_mfma.h, LayerNorm folded); a token that falls
// on a pad pixel, on the +7 dilation border, or in an out-of-range temporal slot is the constant
// vector Linear(LayerNorm(0)) = W*beta + b, passed in as `kvpad`.
//
// One wave per (window, head); lane m < 49 owns query token m and walks the D*49 keys with an
// online softmax.  K/V of the window are staged in LDS ([n][hd+4], broadcast float4 reads).
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include "common.h"

namespace bde {

constexpr int ATT_MAXD = 8;      // max frames in the temporal buffer
constexpr int ATT_WS = 7;
constexpr int ATT_TOK = 49;

struct AttnArgs {
    const float* q;               // [B][Cq_total][HW] query planes, channel offset already applied
    const float* kv[ATT_MAXD];    // per slot: [B][2C(+...)][HW] base of the K planes; nullptr = zero frame
    long q_bs;                    // batch stride of q (elements)
    long kv_bs[ATT_MAXD];         // batch stride per slot
    long v_off[ATT_MAXD];         // element offset from K planes to V planes (C*HW)
    const float* kvpad;           // [2C] constant K|V vector of a zero token
    const float* biasT;           // [heads][D*49][49]  relative-position bias, query index fastest
    float* out;                   // [B][C][HW]
    long out_bs;
    int D, C, heads, H, W, Hp, Wp, pt, pl, nWw, dilated;
};

template <int HD>
__global__ __launch_bounds__(256) void attn_core_kernel(const AttnArgs a) {
    constexpr int HS = HD + 4;                       // padded LDS row (keeps float4 alignment)
    extern __shared__ __align__(16) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wpb = blockDim.x >> 6;                 // waves (= heads) per block, sized by LDS
    const int head = blockIdx.y * wpb + wave;
    const int win = blockIdx.x, b = blockIdx.z;
    if (head >= a.heads) return;                     // whole wave exits; no block-level barrier below
    const int N = a.D * ATT_TOK;
    float* kl = lds + (size_t)wave * 2 * N * HS;
    float* vl = kl + (size_t)N * HS;

    const int wi = win / a.nWw, wj = win - wi * a.nWw;
    const int HW = a.H * a.W;
    // token -> pixel (lane = token for the staging loops and for the query)
    const int tok = lane < ATT_TOK ? lane : ATT_TOK - 1;
    const int ta = tok / ATT_WS, tb = tok - ta * ATT_WS;
    const int step = a.dilated ? 2 : 1;
    const int rp = wi * ATT_WS + ta * step, cp = wj * ATT_WS + tb * step;   // padded-map coords
    const bool inmap = (rp < a.Hp) && (cp < a.Wp);
    const int ry = rp - a.pt, rx = cp - a.pl;
    const bool valid = inmap && ry >= 0 && ry < a.H && rx >= 0 && rx < a.W;
    const long pixoff = valid ? (long)ry * a.W + rx : 0;
    const int c0 = head * HD;

    // ---- stage K,V of all D slots ---------------------------------------------------------------
    if (lane < ATT_TOK) {
        for (int d = 0; d < a.D; ++d) {
            const float* kp = a.kv[d];
            const bool use = valid && (kp != nullptr);
            const float* kb = use ? kp + b * a.kv_bs[d] + (long)c0 * HW + pixoff : nullptr;
            const float* vb = use ? kb + a.v_off[d] : nullptr;
            float* krow = kl + (d * ATT_TOK + lane) * HS;
            float* vrow = vl + (d * ATT_TOK + lane) * HS;
#pragma unroll
            for (int c = 0; c < HD; ++c) {
                krow[c] = use ? kb[(long)c * HW] : a.kvpad[c0 + c];
                vrow[c] = use ? vb[(long)c * HW] : a.kvpad[a.C + c0 + c];
            }
        }
    }
    // query of this lane (already scaled by head_dim^-0.5 through the packed weights)
    float q[HD];
    {
        const float* qb = a.q + b * a.q_bs + (long)c0 * HW + pixoff;
#pragma unroll
        for (int c = 0; c < HD; ++c) q[c] = valid ? qb[(long)c * HW] : 0.f;
    }
    __builtin_amdgcn_s_waitcnt(0);   // LDS writes of this wave complete before its own reads
    __builtin_amdgcn_wave_barrier();

    // ---- online softmax over the N keys -----------------------------------------------------------
    const float* bias = a.biasT + (long)head * N * ATT_TOK + tok;
    float mx = -INFINITY, l = 0.f;
    float o[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) o[c] = 0.f;
    for (int nidx = 0; nidx < N; ++nidx) {
        float s = bias[(long)nidx * ATT_TOK];
        const float4* k4 = reinterpret_cast<const float4*>(kl + nidx * HS);
#pragma unroll
        for (int c4 = 0; c4 < HD / 4; ++c4) {
            float4 kk = k4[c4];
            s += q[4 * c4] * kk.x + q[4 * c4 + 1] * kk.y + q[4 * c4 + 2] * kk.z + q[4 * c4 + 3] * kk.w;
        }
        if constexpr (HD % 4 != 0) {
#pragma unroll
            for (int c = (HD / 4) * 4; c < HD; ++c) s += q[c] * kl[nidx * HS + c];
        }
        float mnew = fmaxf(mx, s);
        float corr = expf(mx - mnew);
        float p = expf(s - mnew);
        l = l * corr + p;
        const float4* v4 = reinterpret_cast<const float4*>(vl + nidx * HS);
#pragma unroll
        for (int c4 = 0; c4 < HD / 4; ++c4) {
            float4 vv = v4[c4];
            o[4 * c4] = o[4 * c4] * corr + p * vv.x;
            o[4 * c4 + 1] = o[4 * c4 + 1] * corr + p * vv.y;
            o[4 * c4 + 2] = o[4 * c4 + 2] * corr + p * vv.z;
            o[4 * c4 + 3] = o[4 * c4 + 3] * corr + p * vv.w;
        }
        if constexpr (HD % 4 != 0) {
#pragma unroll
            for (int c = (HD / 4) * 4; c < HD; ++c) o[c] = o[c] * corr + p * vl[nidx * HS + c];
        }
        mx = mnew;
    }
    if (lane < ATT_TOK && valid) {
        const float inv = 1.f / l;
        float* ob = a.out + b * a.out_bs + (long)c0 * HW + pixoff;
#pragma unroll
        for (int c = 0; c < HD; ++c) ob[(long)c * HW] = o[c] * inv;
    }
}

static inline int attn_launch(const AttnArgs& a, int B, hipStream_t stream) {
    const int hd = a.C / a.heads;
    const int nW = (a.Hp / ATT_WS) * (a.Wp / ATT_WS);
    const int N = a.D * ATT_TOK;
    const size_t per_wave = (size_t)2 * N * (hd + 4) * sizeof(float);
    int wpb = (int)std::min<size_t>(4, (160 * 1024) / per_wave);
    if (wpb < 1) return fail(BDE_ERR_UNSUPPORTED, "attention: head_dim %d x %d keys exceeds LDS", hd, N);
    wpb = std::min(wpb, a.heads);
    dim3 grid(nW, cdiv(a.heads, wpb), B), block(64 * wpb);
    const size_t lds = per_wave * wpb;
#define BDE_ATT_CASE(HDV)                                                                             \
    case HDV: {                                                                                       \
        auto kern = attn_core_kernel<HDV>;                                                            \
        if (lds > 64 * 1024) {                                                                        \
            static bool raised = false;                                                               \
            if (!raised) {                                                                            \
                BDE_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                            160 * 1024));                                             \
                raised = true;                                                                        \
            }                                                                                         \
        }                                                                                             \
        hipLaunchKernelGGL(kern, grid, block, lds, stream, a);                                        \
        break;                                                                                        \
    }
    switch (hd) {
        BDE_ATT_CASE(1)
        BDE_ATT_CASE(2)
        BDE_ATT_CASE(4)
        BDE_ATT_CASE(8)
        BDE_ATT_CASE(16)
        BDE_ATT_CASE(32)
        default:
            return fail(BDE_ERR_UNSUPPORTED, "attention head_dim %d not built (1,2,4,8,16,32)", hd);
    }
#undef BDE_ATT_CASE
    BDE_HIP(hipGetLastError());
    return BDE_OK;
}

}  // namespace bde
