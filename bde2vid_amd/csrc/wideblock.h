// Temporal window attention for WIDE levels (head_dim 16: 256 channels / 16 heads at level 2 of the canonical config),
// where a frame is only a few hundred tokens (23 x 30 at 184 x 240) and the five launches of a block sit on the sequential
// chain of V5.py:154-169.  The split path (pw_gemm.h + attn_mfma.h) ran each of them in 10-13.5 us for < 1 us of matrix
// work: NCHW planes make every operand fetch a gather of 28-byte runs, and every wave fetched its operands with dword
// loads.  Here the chain keeps its activations in the layout the matrix cores consume:
//
//   FRAG16   [token tile of 16][channel group of 16][64 lanes][4]:  lane (col, g4), element j  =  x[token 16*tile + col]
//            [channel 16*group + 4*j + g4] -- i.e. the tensor IS the sequence of B fragments of v_mfma_f32_16x16x4_f32,
//            four k-steps per 16-byte load, 1 KiB per wave instruction, no LDS staging and no transposition anywhere.
//            The D fragments of the producing GEMM land in it with four 256-byte stores per 16 x 16 tile.
//   token-major [token][channels] for q|k|v, which the attention core gathers per window token (64 contiguous bytes per
//            head instead of 16 scattered dwords).
//
//   tokgemm_kernel<MT, NT, KSPLIT>   out = epilogue(W x): weights pre-packed in the same four-k-steps-per-load order;
//            LayerNorm folded as in pw_gemm.h (Linear(LN(x)) = rstd (W'x - mu s) + b', statistics from the streamed B
//            fragments); epilogues: token-major store (q|k|v, K|V stacks) or FRAG16 store with GELU / residual /
//            dilated-coverage mask / merged[t] / a second NCHW copy for the decoder.
//   attn_tok16_kernel   softmax(q k^T + bias) v per (window, head) on the matrix cores as attn_mfma.h, reading token-major
//            q|k|v with 16-byte loads and writing its output straight into FRAG16.
//
// Restates DTransformer.py:165-207 (WindowAttention3D), :254-306 (SwinTransformerBlock3D) and :19-37 (Mlp) exactly as the
// split path does (same folded weights, same bias tables, same padding / dilation rules).
#pragma once
#include <hip/hip_runtime.h>
#include "attn.h"
#include "common.h"
#include "conv_mfma.h"
#include "token_fused.h"
#include "winblock_sb.h"        // ws_split_pair_g

namespace bde {

typedef float wf4 __attribute__((ext_vector_type(4)));

__host__ __device__ __forceinline__ long frag16_index(long tile, int ngroups, int channel, int col) {
    // element (token 16*tile + col, channel) of a FRAG16 tensor with `ngroups` = C / 16 channel groups
    const int grp = channel >> 4, j = (channel >> 2) & 3, g4 = channel & 3;
    return ((tile * ngroups + grp) * 64 + (col + 16 * g4)) * 4 + j;
}

// [N][C][HW] planes -> FRAG16 [N][ntile][C/16][256]; tokens past HW are zero
__global__ __launch_bounds__(256) void nchw_to_frag_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int HW,
                                                           int ntile) {
    const int tile = blockIdx.x, grp = blockIdx.y;
    const long n = blockIdx.z;
    const int lane = threadIdx.x & 63, j = threadIdx.x >> 6;          // 4 waves = 4 elements j of a lane's float4
    const int col = lane & 15, g4 = lane >> 4;
    const int tok = tile * 16 + col, ch = grp * 16 + 4 * j + g4;
    const float v = (tok < HW && ch < C) ? in[(n * C + ch) * HW + tok] : 0.f;
    out[((n * ntile + tile) * (C / 16) + grp) * 256 + lane * 4 + j] = v;
}
static int nchw_to_frag(const float* in, float* out, int N, int C, int HW, hipStream_t s) {
    const int ntile = cdiv(HW, 16);
    hipLaunchKernelGGL(nchw_to_frag_kernel, dim3(ntile, C / 16, N), dim3(256), 0, s, in, out, C, HW, ntile);
    BDE_HIP(hipGetLastError());
    return BDE_OK;
}

// FRAG16 -> [N][C][HW] planes (the op-level entry point hands results back as NCHW)
__global__ __launch_bounds__(256) void frag_to_nchw_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int HW,
                                                           int ntile) {
    const int tile = blockIdx.x, grp = blockIdx.y;
    const long n = blockIdx.z;
    const int lane = threadIdx.x & 63, j = threadIdx.x >> 6;
    const int col = lane & 15, g4 = lane >> 4;
    const int tok = tile * 16 + col, ch = grp * 16 + 4 * j + g4;
    if (tok < HW && ch < C) out[(n * C + ch) * HW + tok] = in[((n * ntile + tile) * (C / 16) + grp) * 256 + lane * 4 + j];
}
static int frag_to_nchw(const float* in, float* out, int N, int C, int HW, hipStream_t s) {
    const int ntile = cdiv(HW, 16);
    hipLaunchKernelGGL(frag_to_nchw_kernel, dim3(ntile, C / 16, N), dim3(256), 0, s, in, out, C, HW, ntile);
    BDE_HIP(hipGetLastError());
    return BDE_OK;
}

struct TokGemmArgs {
    const float* x;        // FRAG16 [B][ntile][K/16][256]
    const float* w;        // packed [M/16][K/16][256]: lane l, element j = W[16*rt + (l&15)][16*kg + 4*j + (l>>4)]
    const float* bias;     // [M]
    const float* lnsum;    // [M] row sums of the LayerNorm-folded weights, nullptr = no LayerNorm
    float* out_tok;        // token-major [B][HW][M]                       (one of out_tok / out_frag)
    float* out_frag;       // FRAG16 [B][ntile][M/16][256]
    float* out_nchw;       // optional second copy of the FRAG16 result as [B][M][HW]
    const float* res;      // optional FRAG16 residual, shape of out_frag
    const float* addres;   // optional second FRAG16 residual (merged[t], V5.py:166)
    long x_bs, out_bs, res_bs, addres_bs, nchw_bs;     // batch strides (elements)
    int K, M, HW, ntile;
    int act;               // ACT_NONE | ACT_GELU
    int mask_w, mask_pt, mask_pl;    // dilated-window coverage mask (DTransformer.py:79-82): uncovered pixels get the residuals only
    // tokgemm_sb_kernel: the weights as two fp16 terms (split.h) in A-fragment order of the 16x16x32 MFMA with the k order of a
    // PAIR of FRAG16 channel groups, [M/16][K/32][2 terms][64 lanes][8]: lane l = (row m = l & 15, g4 = l >> 4), element jj =
    // W[16 rt + m][32 ks + (jj < 4 ? 4 jj + g4 : 16 + 4 (jj - 4) + g4)] * 2^e;  *w_unscale = 2^-e
    const unsigned short* wS;
    const float* w_unscale;
    unsigned* ovf;         // ... and the forward's overflow word: range guard of the two-term format (split.h)
};

// A wave computes MT row tiles x NT token tiles over 1/KSPLIT of K.  KSPLIT == 1: the four waves of a workgroup take
// four consecutive groups of MT row tiles for the same tokens (the B fragments of three of them are L1 hits);
// KSPLIT == 4: they take the four quarters of K of the same tiles and meet in LDS.
// PRE > 0: the wave's whole operand stream (up to PRE channel groups, K <= 16 PRE per wave) is requested before the first
// MFMA -- the launches on the sequential chain have one or two waves per SIMD, so what counts is that the L2 round trip
// is paid once, not once per batch of fragments; PRE == 0 streams through two register buffers (the T-batched launches).
template <int MT, int NT, int KSPLIT, int PRE>
__global__ __launch_bounds__(256) void tokgemm_kernel(const TokGemmArgs a) {
    static_assert(KSPLIT == 1 || KSPLIT == 4, "");
    constexpr int U = 4;                                   // channel groups (of 16) fetched per batch
    __shared__ float red[KSPLIT == 4 ? 3 * MT * NT * 256 : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g4 = lane >> 4, col = lane & 15;
    const int b = blockIdx.z;
    const int ngk = a.K >> 4, ngm = a.M >> 4;
    const int rt0 = (KSPLIT == 1 ? (blockIdx.y * 4 + wave) : blockIdx.y) * MT;
    const int tt0 = blockIdx.x * NT;
    if (KSPLIT == 1 && rt0 >= ngm) return;
    const int kq = ngk / KSPLIT;                           // channel groups of this wave
    const int kg0 = KSPLIT == 1 ? 0 : wave * kq;

    const wf4* xp[NT];
    const wf4* wp[MT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int tile = min(tt0 + t, a.ntile - 1);
        xp[t] = reinterpret_cast<const wf4*>(a.x + b * a.x_bs) + ((long)tile * ngk + kg0) * 64 + lane;
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int rt = min(rt0 + m, ngm - 1);
        wp[m] = reinterpret_cast<const wf4*>(a.w) + ((long)rt * ngk + kg0) * 64 + lane;
    }
    f32x4 acc[MT][NT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[m][t][0] = acc[m][t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    float s1[NT], s2[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) s1[t] = s2[t] = 0.f;
    const bool want_ln = a.lnsum != nullptr;
    // epilogue operands go out with the operand stream (a load issued after the MFMA loop is a round trip on the
    // critical path of a launch that lives for a few microseconds)
    float bb[MT][4], ss[MT][4], rs[MT][NT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int row0 = min(rt0 + m, ngm - 1) * 16 + g4 * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            bb[m][r] = a.bias[row0 + r];
            ss[m][r] = want_ln ? a.lnsum[row0 + r] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const long fo = ((long)min(tt0 + t, a.ntile - 1) * ngm + min(rt0 + m, ngm - 1)) * 256 + col * 4 + g4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = 0.f;
                if (a.res) v = a.res[b * a.res_bs + fo + r * 64];
                if (a.addres) v += a.addres[b * a.addres_bs + fo + r * 64];
                rs[m][t][r] = v;
            }
        }
    }

    if constexpr (PRE > 0) {
        wf4 av[PRE][MT], bv[PRE][NT];
#pragma unroll
        for (int k = 0; k < PRE; ++k) {
            const int kk = min(k, kq - 1);
#pragma unroll
            for (int m = 0; m < MT; ++m) av[k][m] = wp[m][(long)kk * 64];
#pragma unroll
            for (int t = 0; t < NT; ++t) bv[k][t] = xp[t][(long)kk * 64];
        }
#pragma unroll
        for (int k = 0; k < PRE; ++k) {
            if (k < kq) {
                if (want_ln) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const wf4 x = bv[k][t];
                        s1[t] += (x[0] + x[1]) + (x[2] + x[3]);
                        s2[t] += (x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3]);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            acc[m][t][j & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[k][m][j], bv[k][t][j], acc[m][t][j & 1], 0, 0, 0);
            }
        }
    } else {
    wf4 av[2][U][MT], bv[2][U][NT];
    auto fetch = [&](int buf, int kg) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = min(kg + u, kq - 1);             // (kq is a multiple of U for every layer of the chain)
#pragma unroll
            for (int m = 0; m < MT; ++m) av[buf][u][m] = wp[m][(long)k * 64];
#pragma unroll
            for (int t = 0; t < NT; ++t) bv[buf][u][t] = xp[t][(long)k * 64];
        }
    };
    auto compute = [&](int buf, int kg) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (kg + u < kq) {
                if (want_ln) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const wf4 x = bv[buf][u][t];
                        s1[t] += (x[0] + x[1]) + (x[2] + x[3]);
                        s2[t] += (x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3]);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            acc[m][t][j & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[buf][u][m][j], bv[buf][u][t][j], acc[m][t][j & 1], 0, 0, 0);
            }
        }
    };
    // two register buffers, indexed by literals only (a run-time buffer index would put the fragments in scratch)
    fetch(0, 0);
    for (int kg = 0; kg < kq; kg += 2 * U) {
        if (kg + U < kq) fetch(1, kg + U);
        compute(0, kg);
        if (kg + 2 * U < kq) fetch(0, kg + 2 * U);
        if (kg + U < kq) compute(1, kg + U);
    }
    }
    f32x4 fin[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) fin[m][t] = acc[m][t][0] + acc[m][t][1];

    if (KSPLIT == 4) {
        // K quarters meet in LDS: waves 1..3 publish, wave 0 adds them in a fixed order and runs the epilogue
        if (wave > 0) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    *reinterpret_cast<f32x4*>(red + (((wave - 1) * MT + m) * NT + t) * 256 + lane * 4) = fin[m][t];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int t = 0; t < NT; ++t) fin[m][t] += *reinterpret_cast<const f32x4*>(red + ((w * MT + m) * NT + t) * 256 + lane * 4);
    }

    // LayerNorm statistics of each lane's token: the four g4 lanes of a column hold the four channel residues
    float mu[NT], rstd[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        mu[t] = 0.f;
        rstd[t] = 1.f;
        if (want_ln) {
            float u = s1[t], v = s2[t];
            u += __shfl_xor(u, 16); v += __shfl_xor(v, 16);
            u += __shfl_xor(u, 32); v += __shfl_xor(v, 32);
            const float mean = u / (float)a.K;
            const float var = fmaxf(v / (float)a.K - mean * mean, 0.f);
            mu[t] = mean;
            rstd[t] = __builtin_amdgcn_rsqf(var + 1e-5f);       // v_rsq_f32 (1 ulp): this sits on the tail of a 5-10 us launch
        }
    }

#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int rt = rt0 + m;
        if (rt >= ngm) continue;
        const int row0 = rt * 16 + g4 * 4;                 // this lane's four output rows
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int tile = tt0 + t;
            if (tile >= a.ntile) continue;
            const int tok = tile * 16 + col;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float y = fin[m][t][r];
                if (want_ln) y = rstd[t] * (y - mu[t] * ss[m][r]);
                v[r] = act_apply(y + bb[m][r], a.act);
            }
            if (a.out_tok) {
                if (tok < a.HW)
                    *reinterpret_cast<float4*>(a.out_tok + b * a.out_bs + (long)tok * a.M + row0) = float4{v[0], v[1], v[2], v[3]};
                continue;
            }
            if (a.mask_w > 0 && tok < a.HW) {
                const int y = tok / a.mask_w, x = tok - y * a.mask_w;
                const int rr = y + a.mask_pt, cc = x + a.mask_pl;
                if ((rr < 7 && (rr & 1)) || (cc < 7 && (cc & 1))) v[0] = v[1] = v[2] = v[3] = 0.f;
            }
            // FRAG16 position of (token, row0 + r): group rt, lane (col + 16 r), element g4
            const long fo = ((long)tile * ngm + rt) * 256 + col * 4 + g4;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += rs[m][t][r];
            float* op = a.out_frag + b * a.out_bs + fo;
#pragma unroll
            for (int r = 0; r < 4; ++r) op[r * 64] = v[r];
            if (a.out_nchw && tok < a.HW) {
                float* np = a.out_nchw + b * a.nchw_bs + (long)row0 * a.HW + tok;
#pragma unroll
                for (int r = 0; r < 4; ++r) np[(long)r * a.HW] = v[r];
            }
        }
    }
}

template <int MT, int NT, int KSPLIT, int PRE>
static int tokgemm_launch_t(const TokGemmArgs& a, int B, hipStream_t s) {
    const int ngm = a.M / 16;
    dim3 grid(cdiv(a.ntile, NT), KSPLIT == 1 ? cdiv(ngm, 4 * MT) : cdiv(ngm, MT), B);
    hipLaunchKernelGGL((tokgemm_kernel<MT, NT, KSPLIT, PRE>), grid, dim3(256), 0, s, a);
    BDE_HIP(hipGetLastError());
    return BDE_OK;
}

// The K|V GEMMs of the chain (token-major output, LayerNorm folded; K = C = 256, M = depth * 2C = 3072 at level 2 of config A) on
// the fp16 matrix cores with two-term split operands (split.h): a wave = MT row tiles x one token tile.  The tokens stay in
// FRAG16 (fp32): a lane's float4s of two consecutive channel groups ARE eight k values of the 16x16x32 MFMA's B operand once the
// weights are packed in that k order, so the only extra work is their split into two fp16 terms (three vector instructions
// per value, shared by the wave's MT row tiles) -- 3 MFMAs of 16 cycles per 32 k against 8 of 32 cycles.
template <int MT>
__global__ __launch_bounds__(256) void tokgemm_sb_kernel(const TokGemmArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g4 = lane >> 4, col = lane & 15;
    const int b = blockIdx.z;
    const int ngk = a.K >> 4, ngm = a.M >> 4, nks = a.K >> 5;
    // the four waves of a workgroup take four token tiles for the SAME row tiles: their weight fragments are one L2 -> L1 fetch
    // (weights are 4 x the bytes of a token tile's operand; with the waves on different rows the launch pulled 135 MB through L2)
    const int rt0 = blockIdx.y * MT;
    const int tile = blockIdx.x * 4 + wave;
    if (rt0 >= ngm || tile >= a.ntile) return;
    const wf4* xp = reinterpret_cast<const wf4*>(a.x + b * a.x_bs) + ((long)tile * ngk) * 64 + lane;
    const sb8* wp[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) wp[m] = reinterpret_cast<const sb8*>(a.wS) + ((long)min(rt0 + m, ngm - 1) * nks * 2) * 64 + lane;
    // epilogue operands: with the operand stream on the short launches of the sequential chain (a load issued after the MFMA
    // loop is a round trip on their critical path), after it on the T-batched one (64 registers it cannot spare)
    float bb[MT][4], ss[MT][4];
    auto load_rows = [&]() {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int row0 = min(rt0 + m, ngm - 1) * 16 + g4 * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                bb[m][r] = a.bias[row0 + r];
                ss[m][r] = a.lnsum ? a.lnsum[row0 + r] : 0.f;
            }
        }
    };
    if (MT < 8) load_rows();
    const float unscale = a.w_unscale[0];
    constexpr int KS = 8;                                   // k-steps of 32 (K = 256); longer K loops over groups of KS
    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    float s1 = 0.f, s2 = 0.f, gm = 0.f;
    for (int k0 = 0; k0 < nks; k0 += KS) {
        wf4 xv[KS][2];                                      // the token operand of the whole group goes out first
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const int ks = min(k0 + k, nks - 1);
            xv[k][0] = xp[(long)(2 * ks) * 64];
            xv[k][1] = xp[(long)(2 * ks + 1) * 64];
        }
        sb8 av[2][MT][2];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int t = 0; t < 2; ++t) av[0][m][t] = wp[m][((long)k0 * 2 + t) * 64];
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            if (k + 1 < KS) {
                const int ks = min(k0 + k + 1, nks - 1);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int t = 0; t < 2; ++t) av[(k + 1) & 1][m][t] = wp[m][((long)ks * 2 + t) * 64];
            }
            if (k0 + k < nks) {
                const wf4 x0 = xv[k][0], x1 = xv[k][1];
                s1 += ((x0[0] + x0[1]) + (x0[2] + x0[3])) + ((x1[0] + x1[1]) + (x1[2] + x1[3]));
                s2 += ((x0[0] * x0[0] + x0[1] * x0[1]) + (x0[2] * x0[2] + x0[3] * x0[3])) +
                      ((x1[0] * x1[0] + x1[1] * x1[1]) + (x1[2] * x1[2] + x1[3] * x1[3]));
                unsigned t[4][2];
                ws_split_pair_g<2>(x0[0], x0[1], t[0], gm);
                ws_split_pair_g<2>(x0[2], x0[3], t[1], gm);
                ws_split_pair_g<2>(x1[0], x1[1], t[2], gm);
                ws_split_pair_g<2>(x1[2], x1[3], t[3], gm);
                sb8 bfr[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) bfr[q] = sb8{(int)t[0][q], (int)t[1][q], (int)t[2][q], (int)t[3][q]};
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = sb_mma16<2>(av[k & 1][m], bfr, acc[m]);
            }
        }
    }
    if (MT >= 8) load_rows();
    sb_guard_flush(gm, a.ovf);
    // LayerNorm statistics of each lane's token: the four g4 lanes of a column hold the four channel residues
    float mu = 0.f, rstd = 1.f;
    if (a.lnsum) {
        float u = s1, v = s2;
        u += __shfl_xor(u, 16); v += __shfl_xor(v, 16);
        u += __shfl_xor(u, 32); v += __shfl_xor(v, 32);
        const float mean = u / (float)a.K;
        const float var = fmaxf(v / (float)a.K - mean * mean, 0.f);
        mu = mean;
        rstd = __builtin_amdgcn_rsqf(var + 1e-5f);
    }
    const int tok = tile * 16 + col;
    if (tok >= a.HW) return;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int rt = rt0 + m;
        if (rt >= ngm) continue;
        const int row0 = rt * 16 + g4 * 4;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float y = acc[m][r] * unscale;
            if (a.lnsum) y = rstd * (y - mu * ss[m][r]);
            v[r] = act_apply(y + bb[m][r], a.act);
        }
        *reinterpret_cast<float4*>(a.out_tok + b * a.out_bs + (long)tok * a.M + row0) = float4{v[0], v[1], v[2], v[3]};
    }
}
// The T-batched form of the same GEMM (the K|V rows of the unrefined frames: 11 040 tokens x 3 072 rows at config A).  In
// tokgemm_sb_kernel a workgroup is 4 token tiles x 4 row tiles: the tokens of a tile are read and split by the 48 workgroups that
// share them, and a workgroup lives for ~100 MFMAs per wave behind a prologue of its own.  Here a wave splits its token tile ONCE
// (sixteen B fragments, 64 registers), computes its LayerNorm statistics once and then walks RGN groups of four row tiles: one stream of
// weight fragments (the four waves of a workgroup ask for the same ones: L1), MFMAs and 16-byte stores.  The sum over k runs in the
// same order as in tokgemm_sb_kernel: identical results.  K = 256.
#ifndef TOKGEMM_SB_RGN
#define TOKGEMM_SB_RGN 6
#endif
#ifndef TOKGEMM_SB_PD
#define TOKGEMM_SB_PD 3
#endif
#ifndef TOKGEMM_SB_MT
#define TOKGEMM_SB_MT 4
#endif
#ifndef TOKGEMM_SB_NTT
#define TOKGEMM_SB_NTT 2
#endif
#ifndef TOKGEMM_SB_OCC
#define TOKGEMM_SB_OCC 1
#endif
template <int RGN, int MT, int NTT>
__global__ __launch_bounds__(256, TOKGEMM_SB_OCC) void tokgemm_sb_rows_kernel(const TokGemmArgs a) {
    // NTT token tiles per wave: every weight fragment feeds NTT MFMAs -- the four waves of a workgroup walk the same rows, and each pulls
    // its own copy of the fragments through the CU's vector-memory return path (64 B/clk): that path, not the matrix pipe, is what one
    // token tile per wave was bound by (128 us at any occupancy, prefetch depth or store pattern).
    constexpr int KS = 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g4 = lane >> 4, col = lane & 15;
    const int b = blockIdx.z;
    const int ngk = a.K >> 4, ngm = a.M >> 4;
    const int tile0 = ((int)blockIdx.x * 4 + wave) * NTT;                   // (a tile past the last one repeats it and stores nothing)
    const int rtg0 = blockIdx.y * MT * RGN;
    const float unscale_v = a.w_unscale[0];
    const sb8* wbase = reinterpret_cast<const sb8*>(a.wS) + lane;
    auto wfrag = [&](int rt, int k, int t) { return wbase[(((long)min(rt, ngm - 1) * KS + k) * 2 + t) * 64]; };
    const float* lsum = a.lnsum ? a.lnsum : a.bias;        // (pointer select: no load under a branch)
    // weight fragments run PD k-steps ahead of the MFMAs in a ring of PD + 1 register sets (8 steps per group: a step's ring position is
    // a compile-time constant); one wave per SIMD at this register count, so the distance has to cover an L2 round trip by itself
    constexpr int PD = TOKGEMM_SB_PD, RING = PD + 1;
    static_assert(KS % RING == 0, "ring position of a k-step must not depend on the group");
    sb8 av[RING][MT][2];
#pragma unroll
    for (int j = 0; j < PD; ++j)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int t = 0; t < 2; ++t) av[j][m][t] = wfrag(rtg0 + m, j, t);
    __builtin_amdgcn_sched_barrier(0);
    // the token tiles: statistics and split, once
    float gm = 0.f, mu[NTT], rstd[NTT];
    sb8 bfr[NTT][KS][2];
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
        const int tile = min(tile0 + tt, a.ntile - 1);
        const wf4* xp = reinterpret_cast<const wf4*>(a.x + b * a.x_bs) + ((long)tile * ngk) * 64 + lane;
        wf4 xv[KS][2];
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            xv[k][0] = xp[(long)(2 * k) * 64];
            xv[k][1] = xp[(long)(2 * k + 1) * 64];
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const wf4 x0 = xv[k][0], x1 = xv[k][1];
            s1 += ((x0[0] + x0[1]) + (x0[2] + x0[3])) + ((x1[0] + x1[1]) + (x1[2] + x1[3]));
            s2 += ((x0[0] * x0[0] + x0[1] * x0[1]) + (x0[2] * x0[2] + x0[3] * x0[3])) +
                  ((x1[0] * x1[0] + x1[1] * x1[1]) + (x1[2] * x1[2] + x1[3] * x1[3]));
            unsigned t[4][2];
            ws_split_pair_g<2>(x0[0], x0[1], t[0], gm);
            ws_split_pair_g<2>(x0[2], x0[3], t[1], gm);
            ws_split_pair_g<2>(x1[0], x1[1], t[2], gm);
            ws_split_pair_g<2>(x1[2], x1[3], t[3], gm);
#pragma unroll
            for (int q = 0; q < 2; ++q) bfr[tt][k][q] = sb8{(int)t[0][q], (int)t[1][q], (int)t[2][q], (int)t[3][q]};
        }
        mu[tt] = 0.f;
        rstd[tt] = 1.f;
        if (a.lnsum) {
            float u = s1, v = s2;
            u += __shfl_xor(u, 16); v += __shfl_xor(v, 16);
            u += __shfl_xor(u, 32); v += __shfl_xor(v, 32);
            const float mean = u / (float)a.K;
            const float var = fmaxf(v / (float)a.K - mean * mean, 0.f);
            mu[tt] = mean;
            rstd[tt] = __builtin_amdgcn_rsqf(var + 1e-5f);
        }
    }
    sb_guard_flush(gm, a.ovf);
    const float unscale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, unscale_v)));
    constexpr int SP = MT * 16 + 4;                       // floats per token in the staging tile (+ 4: bank spread)
    __shared__ __align__(16) float stg_all[4][16 * (MT * 16 + 4)];
    float* stg = stg_all[wave];
#pragma unroll 1
    for (int rg = 0; rg < RGN; ++rg) {
        const int rt0 = rtg0 + rg * MT;
        if (rt0 >= ngm) break;
        float bb[MT][4], ss[MT][4];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int row0 = min(rt0 + m, ngm - 1) * 16 + g4 * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                bb[m][r] = a.bias[row0 + r];
                ss[m][r] = lsum[row0 + r];
            }
        }
        f32x4 acc[NTT][MT];
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[tt][m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            // the fragments of k-step k + PD (the next group's first ones behind the last steps): requested here, used PD steps later
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    av[(k + PD) % RING][m][t] = k + PD < KS ? wfrag(rt0 + m, k + PD, t) : wfrag(rt0 + MT + m, k + PD - KS, t);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[tt][m] = sb_mma16<2>(av[k % RING][m], bfr[tt][k], acc[tt][m]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // a group's 16 tokens x (MT x 16) rows per token tile go out through the wave's corner of LDS: a lane holds four rows of ONE token,
        // so stored directly an instruction is sixteen 64-byte halves of sixteen different lines (the launch then FETCHES 240 MB to fill the
        // lines it half-writes: rocprofv3 FETCH_SIZE); read back token-major, sixteen lanes write the 16 MT rows x 4 bytes of a token.
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
            const int tile = tile0 + tt;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float y = acc[tt][m][r] * unscale;
                    if (a.lnsum) y = rstd[tt] * (y - mu[tt] * ss[m][r]);
                    v[r] = act_apply(y + bb[m][r], a.act);
                }
                *reinterpret_cast<f32x4*>(stg + col * SP + m * 16 + g4 * 4) = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            constexpr int LPT = MT * 4;                       // lanes (16-byte pieces) per token
#pragma unroll
            for (int j = 0; j < 16 * LPT / 64; ++j) {
                const int tk = (j * 64 + lane) / LPT, pc = (j * 64 + lane) % LPT;
                const f32x4 v = *reinterpret_cast<const f32x4*>(stg + tk * SP + pc * 4);
                const int tokk = tile * 16 + tk, rt = rt0 + (pc >> 2);
                if (tile < a.ntile && tokk < a.HW && rt < ngm)
                    *reinterpret_cast<f32x4*>(a.out_tok + b * a.out_bs + (long)tokk * a.M + (long)rt0 * 16 + pc * 4) = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}
static bool tokgemm_sb_fits(const TokGemmArgs& a) {
    return a.wS != nullptr && a.out_tok != nullptr && a.res == nullptr && a.addres == nullptr && a.K % 32 == 0 && a.M % 16 == 0 && a.mask_w == 0;
}
static int tokgemm_sb_launch(const TokGemmArgs& a, int B, hipStream_t s) {
    const int ngm = a.M / 16;
    const long tiles = (long)a.ntile * ngm * B;
    // (eight row tiles per wave would halve the split work per MFMA on the T-batched launch, but need 239 + 32 registers: one wave
    //  per SIMD; four row tiles: 212, two waves)
    (void)tiles;
    if (TOKGEMM_SB_RGN > 0 && a.K == 256 && (long)a.ntile * B >= 256 && ngm >= TOKGEMM_SB_MT * TOKGEMM_SB_RGN) {
        // the T-batched launch: a wave keeps its split token tiles and walks TOKGEMM_SB_RGN groups of TOKGEMM_SB_MT row tiles
        hipLaunchKernelGGL((tokgemm_sb_rows_kernel<TOKGEMM_SB_RGN, TOKGEMM_SB_MT, TOKGEMM_SB_NTT>),
                           dim3(cdiv(a.ntile, 4 * TOKGEMM_SB_NTT), cdiv(ngm, TOKGEMM_SB_MT * TOKGEMM_SB_RGN), B), dim3(256), 0, s, a);
        BDE_HIP(hipGetLastError());
        return BDE_OK;
    }
    hipLaunchKernelGGL(tokgemm_sb_kernel<4>, dim3(cdiv(a.ntile, 4), cdiv(ngm, 4), B), dim3(256), 0, s, a);
    BDE_HIP(hipGetLastError());
    return BDE_OK;
}

// x1 = x + proj(attention output) AND hidden = GELU(fc1(LayerNorm2(x1))) in ONE launch of the sequential chain
// (DTransformer.py:299, 279-283): fc1 needs every channel of a token's x1, so the two GEMMs used to be two launches of 8-9 us for
// 0.09 + 0.36 GFLOP.  With two-term operands (split.h) the proj of a 16-token tile is 384 MFMAs of 16 cycles: cheap enough to
// REPEAT in each of the four workgroups that share a token tile's fc1 rows (grid = token tiles x 4 quarters of the hidden rows).
// A workgroup: proj of its token tile (16 row tiles over four waves) -> x1 into LDS in FRAG16 order (quarter 0 also stores it: the
// residual of fc2) + LayerNorm sums -> barrier -> its 16 fc1 row tiles with x1 as the token operand, GELU, FRAG16 store.
// Token operands are split in registers as in tokgemm_sb_kernel; weights in the k order of FRAG16 group pairs (TokGemmArgs::wS).
// C = 256, hidden = 1024 (level 2 of config A).
struct ProjFc1Args {
    const float* ao;              // FRAG16 [B][ntile][C/16][256]: attention output
    const float* x;               // FRAG16, same shape: the block input (shortcut)
    const unsigned short *wprojS, *wfc1S;
    const float *proj_unscale, *fc1_unscale;
    const float* bproj;           // [C]
    const float *bfc1, *sfc1;     // [hidden] folded bias, row sums of the LayerNorm-folded weights
    float* x1;                    // FRAG16 [B][ntile][C/16][256]
    float* hid;                   // FRAG16 [B][ntile][hidden/16][256]
    long x_bs, hid_bs;
    int C, hidden, HW, ntile;
    int mask_w, mask_pt, mask_pl; // dilated-window coverage mask on the proj output (uncovered pixels: shortcut only)
    unsigned* ovf;                // range guard of the two-term format (split.h)
};
// acc[m] += W[row tiles rt0 .. rt0 + 3] x over K = 32 nks; xp = the lane's wf4 of channel group 0 (consecutive groups 64 wf4 apart)
template <typename XP>
__device__ __forceinline__ void frag_gemm4(XP xp, const sb8* wbase, int rt0, int nks, f32x4 (&acc)[4], float& gm) {
    constexpr int KS = 8;
    const sb8* wp[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) wp[m] = wbase + ((long)(rt0 + m) * nks * 2) * 64;
    for (int k0 = 0; k0 < nks; k0 += KS) {
        wf4 xv[KS][2];
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const int ks = min(k0 + k, nks - 1);
            xv[k][0] = xp[(2 * ks) * 64];
            xv[k][1] = xp[(2 * ks + 1) * 64];
        }
        sb8 av[2][4][2];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int t = 0; t < 2; ++t) av[0][m][t] = wp[m][((long)k0 * 2 + t) * 64];
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            if (k + 1 < KS) {
                const int ks = min(k0 + k + 1, nks - 1);
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int t = 0; t < 2; ++t) av[(k + 1) & 1][m][t] = wp[m][((long)ks * 2 + t) * 64];
            }
            if (k0 + k < nks) {
                const wf4 x0 = xv[k][0], x1 = xv[k][1];
                unsigned t[4][2];
                ws_split_pair_g<2>(x0[0], x0[1], t[0], gm);
                ws_split_pair_g<2>(x0[2], x0[3], t[1], gm);
                ws_split_pair_g<2>(x1[0], x1[1], t[2], gm);
                ws_split_pair_g<2>(x1[2], x1[3], t[3], gm);
                sb8 bfr[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) bfr[q] = sb8{(int)t[0][q], (int)t[1][q], (int)t[2][q], (int)t[3][q]};
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m] = sb_mma16<2>(av[k & 1][m], bfr, acc[m]);
            }
        }
    }
}
__global__ __launch_bounds__(256) void projfc1_sb_kernel(const ProjFc1Args a) {
    __shared__ __align__(16) float X1[16 * 256];           // x1 of the token tile, FRAG16 order [C/16][64 lanes][4]
    __shared__ float ST[4][16][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g4 = lane >> 4, col = lane & 15;
    const int tile = blockIdx.x, quarter = blockIdx.y, b = blockIdx.z;
    const int ngc = a.C >> 4, ngh = a.hidden >> 4, nks = a.C >> 5;
    const int tok = tile * 16 + col;
    // ---- x1 = x + proj(ao): row tiles 4 wave .. 4 wave + 3 -------------------------------------------------------------------
    {
        const wf4* ap = reinterpret_cast<const wf4*>(a.ao + b * a.x_bs) + ((long)tile * ngc) * 64 + lane;
        f32x4 acc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        float xr[4][4], bp[4][4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int rt = 4 * wave + m;
            const long fo = ((long)tile * ngc + rt) * 256 + col * 4 + g4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                xr[m][r] = a.x[b * a.x_bs + fo + r * 64];
                bp[m][r] = a.bproj[rt * 16 + g4 * 4 + r];
            }
        }
        float gm = 0.f;
        frag_gemm4(ap, reinterpret_cast<const sb8*>(a.wprojS) + lane, 4 * wave, nks, acc, gm);
        sb_guard_flush(gm, a.ovf);
        bool covered = true;
        if (a.mask_w > 0 && tok < a.HW) {
            const int y = tok / a.mask_w, x = tok - y * a.mask_w;
            const int rr = y + a.mask_pt, cc = x + a.mask_pl;
            covered = !((rr < 7 && (rr & 1)) || (cc < 7 && (cc & 1)));
        }
        const float us = a.proj_unscale[0];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int rt = 4 * wave + m;
            const long fo = ((long)tile * ngc + rt) * 256 + col * 4 + g4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = (covered ? acc[m][r] * us + bp[m][r] : 0.f) + xr[m][r];
                X1[(rt * 64 + col + 16 * r) * 4 + g4] = v;
                if (quarter == 0) a.x1[b * a.x_bs + fo + r * 64] = v;
                s1 += v;
                s2 += v * v;
            }
        }
        s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
        s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
        if (lane < 16) { ST[wave][lane][0] = s1; ST[wave][lane][1] = s2; }
    }
    __syncthreads();
    // ---- hidden = GELU(fc1(LayerNorm2(x1))): row tiles 16 quarter + 4 wave .. + 3 ---------------------------------------------
    {
        const float u = (ST[0][col][0] + ST[1][col][0]) + (ST[2][col][0] + ST[3][col][0]);       // (fixed order: every workgroup
        const float v = (ST[0][col][1] + ST[1][col][1]) + (ST[2][col][1] + ST[3][col][1]);       //  of a tile gets the same sums)
        const float mean = u / (float)a.C;
        const float rstd = __builtin_amdgcn_rsqf(fmaxf(v / (float)a.C - mean * mean, 0.f) + 1e-5f);
        const int rt0 = 16 * quarter + 4 * wave;
        float bb[4][4], ss[4][4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                bb[m][r] = a.bfc1[(rt0 + m) * 16 + g4 * 4 + r];
                ss[m][r] = a.sfc1[(rt0 + m) * 16 + g4 * 4 + r];
            }
        f32x4 acc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        float gm = 0.f;
        frag_gemm4(reinterpret_cast<const wf4*>(X1) + lane, reinterpret_cast<const sb8*>(a.wfc1S) + lane, rt0, nks, acc, gm);
        sb_guard_flush(gm, a.ovf);
        const float us = a.fc1_unscale[0];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const long fo = ((long)tile * ngh + rt0 + m) * 256 + col * 4 + g4;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                a.hid[b * a.hid_bs + fo + r * 64] = gelu_f(rstd * (acc[m][r] * us - mean * ss[m][r]) + bb[m][r]);
        }
    }
}
static int projfc1_sb_launch(const ProjFc1Args& a, int B, hipStream_t s) {
    if (a.C != 256 || a.hidden != 1024) return fail(BDE_ERR_UNSUPPORTED, "fused proj + fc1: C = %d, hidden = %d", a.C, a.hidden);
    hipLaunchKernelGGL(projfc1_sb_kernel, dim3(a.ntile, 4, B), dim3(256), 0, s, a);
    BDE_HIP(hipGetLastError());
    return BDE_OK;
}

inline bool wide_nt2() { return tuning().pw_force == 121; }   // (experiment switch: two token tiles per wave)
// Decomposition of one GEMM of the chain: enough waves for the 1024 SIMDs, each with as long an MFMA chain as that allows.
static int tokgemm_launch(const TokGemmArgs& a, int B, hipStream_t s) {
    if (a.K % 64 != 0 || a.M % 16 != 0)
        return fail(BDE_ERR_UNSUPPORTED, "token GEMM: K=%d must be a multiple of 64, M=%d of 16", a.K, a.M);
    const long tiles = (long)a.ntile * (a.M / 16) * B;     // 16 x 16 output tiles
    const int ngk = a.K / 16;
    if (a.K >= 1024 && tiles < 4096) {                     // few tiles, long K: the four waves of a workgroup split K
        if (ngk / 4 <= 16) return tokgemm_launch_t<1, 1, 4, 16>(a, B, s);
        return tokgemm_launch_t<1, 1, 4, 0>(a, B, s);
    }
    if (tiles >= 32768) return tokgemm_launch_t<2, 2, 1, 0>(a, B, s);       // T-batched: throughput, two register buffers
    if (tiles >= 4096) return ngk <= 16 ? tokgemm_launch_t<1, 2, 1, 16>(a, B, s) : tokgemm_launch_t<2, 2, 1, 0>(a, B, s);
    if (tiles >= 1536 && wide_nt2()) return ngk <= 16 ? tokgemm_launch_t<1, 2, 1, 16>(a, B, s) : tokgemm_launch_t<1, 2, 1, 0>(a, B, s);
    return ngk <= 16 ? tokgemm_launch_t<1, 1, 1, 16>(a, B, s) : tokgemm_launch_t<1, 1, 1, 0>(a, B, s);
}

// ---------------------------------------------------------------------------------------------------------------------
struct AttnTokArgs {
    const float* q;               // token-major [B][HW][q_ld]: the query frame's q | k | v rows
    const float* kv[ATT_MAXD];    // per slot: token-major rows holding K at +k_off[d], V at +v_off[d]; nullptr = zero frame
    long q_bs, kv_bs[ATT_MAXD];
    int q_ld, kv_ld[ATT_MAXD], k_off[ATT_MAXD], v_off[ATT_MAXD];
    const float* kvpad;           // [2C] K | V of a zero token
    const float* biasT;           // [heads][D*49][49], query index fastest, log2(e) folded
    float* out;                   // FRAG16 [B][ntile][C/16][256]
    long out_bs;
    int D, C, heads, H, W, Hp, Wp, pt, pl, nWw, dilated, ntile;
    // FUSE: q | k | v of the query frame are computed here from the block input instead of read from `q` / kv[q_slot]
    const float* x;               // FRAG16 [B][ntile][C/16][256]: the block input (current x of the query frame)
    long x_bs;
    const float* wqkv;            // packed as for tokgemm_kernel: [3C/16][C/16][256]; rows q | k | v, LayerNorms folded
    const float* bqkv;            // [3C] folded biases
    const float* sqkv;            // [3C] row sums of the folded weights
    int q_slot;                   // buffer slot of the query frame
    // SPLIT: the same rows as two fp16 terms in the k order of FRAG16 group pairs (TokGemmArgs::wS), *wqkv_unscale their inverse scale
    const unsigned short* wqkvS;
    const float* wqkv_unscale;
    unsigned* ovf;                // range guard of the two-term format (split.h)
};

// softmax(q k^T + bias) v for one (window, head), head_dim 16; four waves = four tiles of 16 queries (attn_mfma.h).
// The 16 channels of the head are contracted in the order (4 g4 + ks): a lane's 16-byte load of q is then its four
// B-operand values as they stand, and K is staged into LDS rows in the same order.
// FUSE: the workgroup first computes q | k | v of its head for the window's 49 tokens of the query frame -- three row tiles of the
// q|k|v GEMM, K = C, wave = token tile, operand fragments gathered straight from the FRAG16 block input (a window token's
// fragment is one 16-byte load per lane and channel group) -- instead of a GEMM launch of its own in front of this one
// (8.9 us on the sequential chain for 0.36 GFLOP).  The D fragments ARE what the attention wants: q as the score MFMA's B
// operand, K and V as the LDS rows below; the per-(window, head) split repeats no work.
// SPLIT (with FUSE): that GEMM on the fp16 matrix cores with two-term operands (split.h), the window tokens split in registers as in
// tokgemm_sb_kernel: 72 MFMAs of 16 cycles per wave instead of 192 of 32.
template <bool FUSE, bool SPLIT>
__global__ __launch_bounds__(256) void attn_tok16_kernel(const AttnTokArgs a) {
    constexpr int HD = 16, NT = 10;
    __shared__ __align__(16) float KL[NT * HD * 16];
    __shared__ __align__(16) float VL[NT * 16 * HD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g4 = lane >> 4, col = lane & 15;
    const int win = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
    const int wi = win / a.nWw, wj = win - wi * a.nWw;
    const int step = a.dilated ? 2 : 1;
    const int c0 = head * HD;
    const int nkey = a.D * ATT_TOK;
    auto token_pixel = [&](int tok) {
        const int ta = tok / ATT_WS, tb = tok - ta * ATT_WS;
        const int rp = wi * ATT_WS + ta * step, cp = wj * ATT_WS + tb * step;
        const int ry = rp - a.pt, rx = cp - a.pl;
        return (rp < a.Hp && cp < a.Wp && ry >= 0 && ry < a.H && rx >= 0 && rx < a.W) ? ry * a.W + rx : -1;
    };
    const int qi = wave * 16 + col;
    const int qpix = qi < ATT_TOK ? token_pixel(qi) : -1;
    wf4 qv;
    if constexpr (FUSE) {
        // rows head (q), C/16 + head (k), 2C/16 + head (v) of the packed q|k|v weights; this wave's token tile = queries 16 wave ..
        const int ngk = a.C >> 4;
        const wf4* xw = reinterpret_cast<const wf4*>(a.x + b * a.x_bs) + ((long)(max(qpix, 0) >> 4) * ngk) * 64 + (max(qpix, 0) & 15) + 16 * g4;
        const wf4* wq = reinterpret_cast<const wf4*>(a.wqkv) + ((long)head * ngk) * 64 + lane;
        const wf4* wk = reinterpret_cast<const wf4*>(a.wqkv) + ((long)(ngk + head) * ngk) * 64 + lane;
        const wf4* wv = reinterpret_cast<const wf4*>(a.wqkv) + ((long)(2 * ngk + head) * ngk) * 64 + lane;
        f32x4 aq = {0.f, 0.f, 0.f, 0.f}, ak = aq, av = aq;
        float s1 = 0.f, s2 = 0.f;
        const bool any = wave * 16 < ATT_TOK;                      // (tile 3 holds token 48 only; a tile past the window is skipped)
        if (SPLIT && any) {
            const int nks = a.C >> 5;
            const sb8* wqS = reinterpret_cast<const sb8*>(a.wqkvS) + ((long)head * nks * 2) * 64 + lane;
            const sb8* wkS = reinterpret_cast<const sb8*>(a.wqkvS) + ((long)(ngk + head) * nks * 2) * 64 + lane;
            const sb8* wvS = reinterpret_cast<const sb8*>(a.wqkvS) + ((long)(2 * ngk + head) * nks * 2) * 64 + lane;
            constexpr int U = 2;                                   // k-steps of 32 (two channel groups each) per register buffer
            float gm = 0.f;
            wf4 xb[2][2 * U];
            sb8 fq[2][U][2], fk[2][U][2], fv[2][U][2];
            auto fetch = [&](int buf, int ks0) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const long ks = min(ks0 + u, nks - 1);
                    xb[buf][2 * u] = xw[(2 * ks) * 64];
                    xb[buf][2 * u + 1] = xw[(2 * ks + 1) * 64];
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        fq[buf][u][t] = wqS[(ks * 2 + t) * 64];
                        fk[buf][u][t] = wkS[(ks * 2 + t) * 64];
                        fv[buf][u][t] = wvS[(ks * 2 + t) * 64];
                    }
                }
            };
            auto compute = [&](int buf, int ks0) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (ks0 + u >= nks) continue;
                    wf4 x0 = xb[buf][2 * u], x1 = xb[buf][2 * u + 1];
                    if (qpix < 0) x0 = x1 = wf4{0.f, 0.f, 0.f, 0.f};
                    s1 += ((x0[0] + x0[1]) + (x0[2] + x0[3])) + ((x1[0] + x1[1]) + (x1[2] + x1[3]));
                    s2 += ((x0[0] * x0[0] + x0[1] * x0[1]) + (x0[2] * x0[2] + x0[3] * x0[3])) +
                          ((x1[0] * x1[0] + x1[1] * x1[1]) + (x1[2] * x1[2] + x1[3] * x1[3]));
                    unsigned t[4][2];
                    ws_split_pair_g<2>(x0[0], x0[1], t[0], gm);
                    ws_split_pair_g<2>(x0[2], x0[3], t[1], gm);
                    ws_split_pair_g<2>(x1[0], x1[1], t[2], gm);
                    ws_split_pair_g<2>(x1[2], x1[3], t[3], gm);
                    sb8 bfr[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) bfr[q] = sb8{(int)t[0][q], (int)t[1][q], (int)t[2][q], (int)t[3][q]};
                    aq = sb_mma16<2>(fq[buf][u], bfr, aq);
                    ak = sb_mma16<2>(fk[buf][u], bfr, ak);
                    av = sb_mma16<2>(fv[buf][u], bfr, av);
                }
            };
            fetch(0, 0);
            for (int ks = 0; ks < nks; ks += 2 * U) {
                if (ks + U < nks) fetch(1, ks + U);
                compute(0, ks);
                if (ks + 2 * U < nks) fetch(0, ks + 2 * U);
                if (ks + U < nks) compute(1, ks + U);
            }
            sb_guard_flush(gm, a.ovf);
            const float us = a.wqkv_unscale[0];
#pragma unroll
            for (int r = 0; r < 4; ++r) { aq[r] *= us; ak[r] *= us; av[r] *= us; }
        }
        if (!SPLIT && any) {
            // two register buffers of U channel groups (literal indices only): the next batch of operand fragments is in flight
            // during the MFMAs of the current one -- a workgroup has the CU almost to itself, nobody else covers an L2 round trip
            constexpr int U = 4;
            wf4 xb[2][U], fq[2][U], fk[2][U], fv[2][U];
            auto fetch = [&](int buf, int kg) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const long o = (long)min(kg + u, ngk - 1) * 64;
                    xb[buf][u] = xw[o]; fq[buf][u] = wq[o]; fk[buf][u] = wk[o]; fv[buf][u] = wv[o];
                }
            };
            auto compute = [&](int buf, int kg) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (kg + u >= ngk) continue;
                    wf4 xx = xb[buf][u];
                    if (qpix < 0) xx = wf4{0.f, 0.f, 0.f, 0.f};    // a zero token: LayerNorm(0) = beta, i.e. the folded bias alone
                    s1 += (xx[0] + xx[1]) + (xx[2] + xx[3]);
                    s2 += (xx[0] * xx[0] + xx[1] * xx[1]) + (xx[2] * xx[2] + xx[3] * xx[3]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        aq = __builtin_amdgcn_mfma_f32_16x16x4f32(fq[buf][u][j], xx[j], aq, 0, 0, 0);
                        ak = __builtin_amdgcn_mfma_f32_16x16x4f32(fk[buf][u][j], xx[j], ak, 0, 0, 0);
                        av = __builtin_amdgcn_mfma_f32_16x16x4f32(fv[buf][u][j], xx[j], av, 0, 0, 0);
                    }
                }
            };
            fetch(0, 0);
            for (int kg = 0; kg < ngk; kg += 2 * U) {
                if (kg + U < ngk) fetch(1, kg + U);
                compute(0, kg);
                if (kg + 2 * U < ngk) fetch(0, kg + 2 * U);
                if (kg + U < ngk) compute(1, kg + U);
            }
        }
        s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
        s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
        const float mean = s1 / (float)a.C;
        const float rstd = __builtin_amdgcn_rsqf(fmaxf(s2 / (float)a.C - mean * mean, 0.f) + 1e-5f);
        float kq[4], vq[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rq = c0 + 4 * g4 + r, rk = a.C + rq, rv = 2 * a.C + rq;
            qv[r] = rstd * (aq[r] - mean * a.sqkv[rq]) + a.bqkv[rq];
            kq[r] = rstd * (ak[r] - mean * a.sqkv[rk]) + a.bqkv[rk];
            vq[r] = rstd * (av[r] - mean * a.sqkv[rv]) + a.bqkv[rv];
        }
        if (qi < ATT_TOK) {
            // key index of this token in the slot-major key order; K row (k-step e = r, lanes g4) and the V row of the core below
            const int u = a.q_slot * ATT_TOK + qi;
#pragma unroll
            for (int r = 0; r < 4; ++r) KL[((u >> 4) * HD + r * 4 + g4) * 16 + (u & 15)] = kq[r];
            *reinterpret_cast<wf4*>(VL + u * HD + g4 * 4) = wf4{vq[0], vq[1], vq[2], vq[3]};
        }
    } else {
        qv = *reinterpret_cast<const wf4*>(a.q + b * a.q_bs + (long)max(qpix, 0) * a.q_ld + c0 + 4 * g4);
    }
    if (qpix < 0) qv = wf4{0.f, 0.f, 0.f, 0.f};
    const float* bias = a.biasT + (long)head * nkey * ATT_TOK + min(qi, ATT_TOK - 1);
    f32x4 sc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int u = j * 16 + g4 * 4 + r;
            const float bvv = bias[(long)min(u, nkey - 1) * ATT_TOK];
            sc[j][r] = u < nkey ? bvv : -1e30f;
        }
    {
        wf4 kk[3], vv[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int it = min(tid + t * 256, NT * 16 * 4 - 1);
            const int u = it >> 2, cg = it & 3;
            const int d = min(u / ATT_TOK, a.D - 1), tok = u - (u / ATT_TOK) * ATT_TOK;
            const int pix = u < nkey ? token_pixel(tok) : -1;
            const float* kp = (u < nkey && !(FUSE && d == a.q_slot)) ? a.kv[d] : nullptr;
            const bool use = pix >= 0 && kp != nullptr;
            const float* ksrc = use ? kp + b * a.kv_bs[d] + (long)pix * a.kv_ld[d] + a.k_off[d] + c0 + cg * 4 : a.kvpad + c0 + cg * 4;
            const float* vsrc = use ? kp + b * a.kv_bs[d] + (long)pix * a.kv_ld[d] + a.v_off[d] + c0 + cg * 4 : a.kvpad + a.C + c0 + cg * 4;
            kk[t] = *reinterpret_cast<const wf4*>(ksrc);
            vv[t] = *reinterpret_cast<const wf4*>(vsrc);
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int it = tid + t * 256;
            if (it < NT * 16 * 4 && !(FUSE && (it >> 2) / ATT_TOK == a.q_slot)) {     // (fused: the query frame's rows are written above)
                const int u = it >> 2, cg = it & 3;
                // channel cg*4 + e of the head is contracted at k-step e by the lanes g4 = cg: LDS row e*4 + cg
#pragma unroll
                for (int e = 0; e < 4; ++e) KL[((u >> 4) * HD + e * 4 + cg) * 16 + (u & 15)] = kk[t][e];
                *reinterpret_cast<wf4*>(VL + u * HD + cg * 4) = vv[t];
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            sc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(KL[(j * HD + ks * 4 + g4) * 16 + col], qv[ks], sc[j], 0, 0, 0);
    float mx = sc[0][0];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        mx = fmaxf(mx, fmaxf(sc[j][0], sc[j][1]));
        mx = fmaxf(mx, fmaxf(sc[j][2], sc[j][3]));
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
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
        // rows of acc = channels c0 + 4 g4 + r of query pixel qpix: FRAG16 group `head`, lane (pixel column + 16 r), element g4
        float* op = a.out + b * a.out_bs + ((long)(qpix >> 4) * (a.C >> 4) + head) * 256 + (qpix & 15) * 4 + g4;
#pragma unroll
        for (int r = 0; r < 4; ++r) op[r * 64] = acc[r] * inv;
    }
}

static int attn_tok16_launch(const AttnTokArgs& a, int B, hipStream_t s) {
    const int nW = (a.Hp / ATT_WS) * (a.Wp / ATT_WS);
    if (a.x && a.wqkvS && a.C % 32 == 0) hipLaunchKernelGGL((attn_tok16_kernel<true, true>), dim3(nW, a.heads, B), dim3(256), 0, s, a);
    else if (a.x) hipLaunchKernelGGL((attn_tok16_kernel<true, false>), dim3(nW, a.heads, B), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((attn_tok16_kernel<false, false>), dim3(nW, a.heads, B), dim3(256), 0, s, a);
    BDE_HIP(hipGetLastError());
    return BDE_OK;
}

}  // namespace bde
