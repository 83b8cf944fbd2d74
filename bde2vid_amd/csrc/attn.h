// Temporal window-attention core (softmax(q k^T + bias) v per 7x7 window and head).
//
// Restates the middle of WindowAttention3D.forward (DTransformer.py:192-203) together with the
// addressing of window_partition / window_reverse (DTransformer.py:40-83) and the zero padding of
// SwinTransformerBlock3D.forward_part1 (:260-264): windows are never materialised.  q/k/v live as
// NCHW planes produced by the 1x1 GEMMs (pw_gemm.h, LayerNorm folded); a token that falls on a pad
// pixel, on the +7 dilation border, or in an out-of-range temporal slot is the constant vector
// Linear(LayerNorm(0)) = W*beta + b, passed in as `kvpad`.
//
// Work decomposition (latency first: this kernel sits on the sequential chain V5.py:154-169):
//   one WAVE per (window, head, temporal slot d): lane m < 49 owns query token m and scores the 49
//   keys of slot d, seven at a time (one running-max update and one rescale per seven keys, the
//   seven bias loads issued together).  The D partial softmaxes (m, l, o[hd]) of a (window, head)
//   are merged through LDS by the slot-0 wave.  Level 2 of config A has only 20 windows x 16 heads;
//   splitting the keys over D waves triples the waves in flight there.
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include "common.h"

namespace bde {

constexpr int ATT_MAXD = 8;      // max frames in the temporal buffer
constexpr int ATT_WS = 7;
constexpr int ATT_TOK = 49;

struct AttnArgs {
    const float* q;               // [B][..][HW] query planes (channel 0 of the q rows)
    const float* kv[ATT_MAXD];    // per slot: base of the K planes [B][..][HW]; nullptr = zero frame
    long q_bs;                    // batch stride of q (elements)
    long kv_bs[ATT_MAXD];         // batch stride per slot
    long v_off[ATT_MAXD];         // element offset from the K planes to the V planes
    const float* kvpad;           // [2C] constant K|V vector of a zero token
    const float* biasT;           // [heads][D*49][49]  relative-position bias, query index fastest
    float* out;                   // [B][C][HW]
    long out_bs;
    int D, C, heads, H, W, Hp, Wp, pt, pl, nWw, dilated, hpb;   // hpb = heads per block
};

template <int HD>
__global__ __launch_bounds__(1024) void attn_core_kernel(const AttnArgs a) {
    constexpr int HS = (HD + 3) / 4 * 4 + 4;        // padded LDS row, float4-aligned
    constexpr int PS = HD + 2;                       // partial record: m, l, o[HD]
    extern __shared__ __align__(16) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hl = wave / a.D, d = wave - hl * a.D;  // local head, temporal slot
    const int head = blockIdx.y * a.hpb + hl;
    const int win = blockIdx.x, b = blockIdx.z;
    const bool active = head < a.heads;              // uniform per wave
    float* kl = lds + (size_t)wave * 2 * ATT_TOK * HS;
    float* vl = kl + ATT_TOK * HS;
    float* part = lds + (size_t)blockDim.x / 64 * 2 * ATT_TOK * HS;   // [hpb][D][PS][64]

    const int wi = win / a.nWw, wj = win - wi * a.nWw;
    const int HW = a.H * a.W;
    const int tok = lane < ATT_TOK ? lane : ATT_TOK - 1;
    const int ta = tok / ATT_WS, tb = tok - ta * ATT_WS;
    const int step = a.dilated ? 2 : 1;
    const int rp = wi * ATT_WS + ta * step, cp = wj * ATT_WS + tb * step;   // padded-map coords
    const bool inmap = (rp < a.Hp) && (cp < a.Wp);
    const int ry = rp - a.pt, rx = cp - a.pl;
    const bool valid = inmap && ry >= 0 && ry < a.H && rx >= 0 && rx < a.W;
    const long pixoff = valid ? (long)ry * a.W + rx : 0;
    const int c0 = head * HD;

    float mx = -INFINITY, l = 0.f;
    float o[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) o[c] = 0.f;

    if (active) {
        // bias of the next seven keys is always loaded one batch ahead; the first batch goes out
        // together with the K/V/q loads
        const float* bias = a.biasT + ((long)head * a.D + d) * ATT_TOK * ATT_TOK + tok;
        float sn[7];
#pragma unroll
        for (int u = 0; u < 7; ++u) sn[u] = bias[(long)u * ATT_TOK];
        // ---- stage this slot's 49 keys / values (lane = key token) -------------------------------
        if (lane < ATT_TOK) {
            const float* kp = a.kv[d];
            const bool use = valid && (kp != nullptr);
            const float* kb = use ? kp + b * a.kv_bs[d] + (long)c0 * HW + pixoff : nullptr;
            const float* vb = use ? kb + a.v_off[d] : nullptr;
            float kreg[HD], vreg[HD];
            // pointer select + unconditional load (a load under a per-lane branch costs a vmcnt(0) join)
            const float* ksrc = use ? kb : a.kvpad + c0;
            const float* vsrc = use ? vb : a.kvpad + a.C + c0;
            const long cstride = use ? (long)HW : 1;
#pragma unroll
            for (int c = 0; c < HD; ++c) {
                kreg[c] = ksrc[c * cstride];
                vreg[c] = vsrc[c * cstride];
            }
#pragma unroll
            for (int c = 0; c < HD; ++c) {
                kl[lane * HS + c] = kreg[c];
                vl[lane * HS + c] = vreg[c];
            }
        }
        // query of this lane; the packed q weights carry head_dim^-0.5 * log2(e) and the bias table
        // log2(e), so the softmax runs on v_exp_f32 (2^x) directly
        float q[HD];
        {
            const float* qb = a.q + b * a.q_bs + (long)c0 * HW + pixoff;
#pragma unroll
            for (int c = 0; c < HD; ++c) {
                const float v = qb[(long)c * HW];          // pixoff is 0 for invalid tokens: in bounds
                q[c] = valid ? v : 0.f;
            }
        }
        __builtin_amdgcn_s_waitcnt(0);   // this wave's LDS writes land before its own reads
        __builtin_amdgcn_wave_barrier();

#pragma unroll 1
        for (int j0 = 0; j0 < ATT_TOK; j0 += 7) {
            float s[7];
#pragma unroll
            for (int u = 0; u < 7; ++u) s[u] = sn[u];
            if (j0 + 7 < ATT_TOK) {
#pragma unroll
                for (int u = 0; u < 7; ++u) sn[u] = bias[(long)(j0 + 7 + u) * ATT_TOK];
            }
#pragma unroll
            for (int u = 0; u < 7; ++u) {
                const float* kr = kl + (j0 + u) * HS;
                float acc = 0.f;
#pragma unroll
                for (int c = 0; c < HD; ++c) acc += q[c] * kr[c];
                s[u] += acc;
            }
            float mb = s[0];
#pragma unroll
            for (int u = 1; u < 7; ++u) mb = fmaxf(mb, s[u]);
            const float mnew = fmaxf(mx, mb);
            const float corr = __builtin_amdgcn_exp2f(mx - mnew);
            l *= corr;
#pragma unroll
            for (int c = 0; c < HD; ++c) o[c] *= corr;
#pragma unroll
            for (int u = 0; u < 7; ++u) {
                const float p = __builtin_amdgcn_exp2f(s[u] - mnew);
                l += p;
                const float* vr = vl + (j0 + u) * HS;
#pragma unroll
                for (int c = 0; c < HD; ++c) o[c] += p * vr[c];
            }
            mx = mnew;
        }
        // publish the partial (m, l, o) of this slot
        float* pr = part + ((size_t)(hl * a.D + d) * PS) * 64 + lane;
        pr[0] = mx;
        pr[64] = l;
#pragma unroll
        for (int c = 0; c < HD; ++c) pr[(2 + c) * 64] = o[c];
    }
    __syncthreads();
    if (active && d == 0 && lane < ATT_TOK && valid) {
        // merge the D partial softmaxes of this (window, head)
        const float* p0 = part + ((size_t)(hl * a.D) * PS) * 64 + lane;
        float M = p0[0];
        for (int dd = 1; dd < a.D; ++dd) M = fmaxf(M, p0[(size_t)dd * PS * 64]);
        float L = 0.f;
        float O[HD];
#pragma unroll
        for (int c = 0; c < HD; ++c) O[c] = 0.f;
        for (int dd = 0; dd < a.D; ++dd) {
            const float* pp = p0 + (size_t)dd * PS * 64;
            const float w = __builtin_amdgcn_exp2f(pp[0] - M);
            L += pp[64] * w;
#pragma unroll
            for (int c = 0; c < HD; ++c) O[c] += pp[(2 + c) * 64] * w;
        }
        const float inv = 1.f / L;
        float* ob = a.out + b * a.out_bs + (long)c0 * HW + pixoff;
#pragma unroll
        for (int c = 0; c < HD; ++c) ob[(long)c * HW] = O[c] * inv;
    }
}

static inline int attn_launch(AttnArgs a, int B, hipStream_t stream) {
    const int hd = a.C / a.heads;
    const int nW = (a.Hp / ATT_WS) * (a.Wp / ATT_WS);
    int hpb = std::max(1, 16 / a.D);                 // <= 16 waves (1024 threads) per block
    int p2 = 1;
    while (p2 * 2 <= hpb) p2 *= 2;
    hpb = std::min(p2, a.heads);
    a.hpb = hpb;
    const int waves = hpb * a.D;
    const int HS = (hd + 3) / 4 * 4 + 4;
    const size_t lds = ((size_t)waves * 2 * ATT_TOK * HS + (size_t)waves * (hd + 2) * 64) * sizeof(float);
    if (lds > 160 * 1024) return fail(BDE_ERR_UNSUPPORTED, "attention: head_dim %d needs %zu B of LDS", hd, lds);
    dim3 grid(nW, cdiv(a.heads, hpb), B), block(64 * waves);
#define BDE_ATT_CASE(HDV)                                                                             \
    case HDV: {                                                                                       \
        auto kern = attn_core_kernel<HDV>;                                                            \
        if (lds > 64 * 1024) {                                                                        \
            static unsigned char raised[BDE_MAX_DEVICES];                                                               \
            BDE_HIP(raise_dynamic_lds(raised, (const void*)kern));                                    \
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
