// Temporal window-attention core for head_dim 16 on the matrix cores -- the variant of attn.h used on
// the level-2 chain of config A (256 channels, 16 heads: 20 windows per frame, one frame at a time,
// V5.py:154-169), where the launch is short and its latency is what counts.
//
// A workgroup owns one (window, head); its four waves each take a tile of 16 queries:
//   scores^T[key][query] = k q^T + bias   one 16x16x4 MFMA per 4 channels, the bias as the C operand
//   p = 2^(s - max)                        (log2(e) is folded into q and the bias table, as in attn.h)
//   out^T[channel][query] += v^T p^T       four MFMAs per 16 keys: the score registers of a lane are the
//                                          B operand as they stand (row = key 4*(lane>>4)+r, column = query)
// so a query tile costs 80 MFMAs and ~200 vector instructions, against ~49 * 40 vector instructions per
// lane in attn.h.  K|V of the window are staged once per workgroup:
//   KL [16-key tile][channel][16]  (A operand of k q^T: four channel rows x 16 keys = 64 consecutive floats)
//   VL [key][channel 16]           (A operand of v^T p^T: four keys x 16 channels = 64 consecutive floats)
// Window addressing, zero padding, dilation and zero frames exactly as in attn.h.
#pragma once
#include <hip/hip_runtime.h>
#include "attn.h"

namespace bde {

typedef float am_f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void attn_mfma16_kernel(const AttnArgs a) {
    constexpr int HD = 16, NT = 10;                       // key tiles of 16 (D*49 <= 160)
    __shared__ __align__(16) float KL[NT * HD * 16];
    __shared__ __align__(16) float VL[NT * 16 * HD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g4 = lane >> 4, col = lane & 15;
    const int win = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
    const int wi = win / a.nWw, wj = win - wi * a.nWw;
    const int HW = a.H * a.W;
    const int step = a.dilated ? 2 : 1;
    const int c0 = head * HD;
    const int nkey = a.D * ATT_TOK;
    auto token_pixel = [&](int tok) {                     // pixel of window token `tok`, -1 = zero token
        const int ta = tok / ATT_WS, tb = tok - ta * ATT_WS;
        const int rp = wi * ATT_WS + ta * step, cp = wj * ATT_WS + tb * step;
        const int ry = rp - a.pt, rx = cp - a.pl;
        return (rp < a.Hp && cp < a.Wp && ry >= 0 && ry < a.H && rx >= 0 && rx < a.W) ? ry * a.W + rx : -1;
    };

    // ---- bias of the first two key tiles and the query fragments go out first ------------------------
    const int qi = wave * 16 + col;                        // query token of this lane's column
    const int qpix = qi < ATT_TOK ? token_pixel(qi) : -1;
    float qf[4];
    {
        const float* qb = a.q + b * a.q_bs + (long)c0 * HW + max(qpix, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const float v = qb[(long)(ks * 4 + g4) * HW];
            qf[ks] = qpix >= 0 ? v : 0.f;
        }
    }
    const float* bias = a.biasT + (long)head * nkey * ATT_TOK + min(qi, ATT_TOK - 1);

    // ---- bias tiles (the C operands of pass 1) are independent of the staging: issue them first ----------
    am_f32x4 sc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int u = j * 16 + g4 * 4 + r;
            const float bv = bias[(long)min(u, nkey - 1) * ATT_TOK];
            sc[j][r] = u < nkey ? bv : -1e30f;
        }
    }

    // ---- stage K|V: thread = (key u, 4-channel group cg); 160 keys x 4 groups = 640 items, all loads of a
    //      thread's three items in flight together ------------------------------------------------------
    {
        float kv[3][4], vv[3][4];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int it = min(tid + t * 256, NT * 16 * 4 - 1);
            const int u = it >> 2, cg = it & 3;
            const int d = u / ATT_TOK, tok = u - d * ATT_TOK;
            const int pix = u < nkey ? token_pixel(tok) : -1;
            const float* kp = u < nkey ? a.kv[d] : nullptr;
            const bool use = pix >= 0 && kp != nullptr;
            // pointer select + unconditional loads (a load under a per-lane branch costs a vmcnt(0) join)
            const float* ksrc = use ? kp + b * a.kv_bs[d] + (long)(c0 + cg * 4) * HW + pix : a.kvpad + c0 + cg * 4;
            const float* vsrc = use ? ksrc + a.v_off[d] : a.kvpad + a.C + c0 + cg * 4;
            const long cs = use ? (long)HW : 1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                kv[t][e] = ksrc[e * cs];
                vv[t][e] = vsrc[e * cs];
            }
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int it = tid + t * 256;
            if (it < NT * 16 * 4) {
                const int u = it >> 2, cg = it & 3;
#pragma unroll
                for (int e = 0; e < 4; ++e) KL[((u >> 4) * HD + cg * 4 + e) * 16 + (u & 15)] = kv[t][e];
                *reinterpret_cast<float4*>(VL + u * HD + cg * 4) = float4{vv[t][0], vv[t][1], vv[t][2], vv[t][3]};
            }
        }
    }
    __syncthreads();

    // ---- pass 1: scores of all key tiles (registers), maximum per query ------------------------------
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            sc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(KL[(j * HD + ks * 4 + g4) * 16 + col], qf[ks], sc[j], 0, 0, 0);
    }
    float mx = sc[0][0];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        mx = fmaxf(mx, fmaxf(sc[j][0], sc[j][1]));
        mx = fmaxf(mx, fmaxf(sc[j][2], sc[j][3]));
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));

    // ---- pass 2: p = 2^(s - max), out^T += v^T p^T ------------------------------------------------------
    am_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float l = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        float p[4], va[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            p[r] = __builtin_amdgcn_exp2f(sc[j][r] - mx);
            va[r] = VL[(j * 16 + g4 * 4 + r) * HD + col];
            l += p[r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(va[r], p[r], acc, 0, 0, 0);
    }
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    if (qpix >= 0) {
        const float inv = 1.f / l;
        float* ob = a.out + b * a.out_bs + (long)(c0 + g4 * 4) * HW + qpix;   // rows = channels 4*g4 + r
#pragma unroll
        for (int r = 0; r < 4; ++r) ob[(long)r * HW] = acc[r] * inv;
    }
}

static inline int attn_mfma16_launch(AttnArgs a, int B, hipStream_t stream) {
    const int nW = (a.Hp / ATT_WS) * (a.Wp / ATT_WS);
    hipLaunchKernelGGL(attn_mfma16_kernel, dim3(nW, a.heads, B), dim3(256), 0, stream, a);
    BDE_HIP(hipGetLastError());
    return BDE_OK;
}

}  // namespace bde
