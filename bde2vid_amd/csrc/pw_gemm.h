// Pointwise (1x1) contraction  out[co][p] = sum_ci W[co][ci] * x[ci][p]  on the fp32 matrix cores,
// for every Linear of the temporal attention (DTransformer.py:188-190,204) and of its MLP
// (DTransformer.py:31-37) applied to [C][H*W] planes, LayerNorm folded (see conv_mfma.h).
//
// These GEMMs are small (0.1-0.4 GFLOP) and sit on the sequential attention chain (V5.py:154-169),
// so the kernel is built for latency, not for operand reuse: NO LDS staging and NO barrier in the
// main loop.  Both MFMA operands go straight from L2 to VGPRs --
//   A: packed weight fragment, one 256-B coalesced load per (32 co x 2 ci),
//   B: x[2*kp + (lane>>5)][p0 + (lane&31)], two 128-B segments per load --
// eight k-pairs of loads are in flight per wave while the previous eight feed the MFMAs.
//   SPLIT=false: 4 waves per block, each with its own pixel tiles (T-batched launches).
//   SPLIT=true : the 4 waves of a block split K for one tile and reduce through LDS once at the
//                end (per-step launches, where one frame has to fill 1024 SIMDs).
#pragma once
#include "conv_mfma.h"

namespace bde {

template <int MT, int NT, int U>
struct PwFrag {
    float av[U][MT], bv[U][NT];
};

template <int MT, int NT, int U>
__device__ __forceinline__ void pw_load(PwFrag<MT, NT, U>& f, const float* const (&wq)[MT], const float* const (&xb)[NT],
                                        int kp, long HW2) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int m = 0; m < MT; ++m) f.av[u][m] = wq[m][(long)(kp + u) * 64];
#pragma unroll
        for (int t = 0; t < NT; ++t) f.bv[u][t] = xb[t][(long)(kp + u) * HW2];
    }
}

template <int MT, int NT, int U>
__device__ __forceinline__ void pw_mac(const PwFrag<MT, NT, U>& f, f32x16 (&acc)[MT][NT], float (&s1)[NT], float (&s2)[NT]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            s1[t] += f.bv[u][t];
            s2[t] += f.bv[u][t] * f.bv[u][t];
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.av[u][m], f.bv[u][t], acc[m][t], 0, 0, 0);
    }
}

// WS = 1: 4 waves per block, each with its own pixel tiles, no reduction.
// WS = 4 | 8: WS waves per block split K of one tile and reduce through LDS.
template <int MT, int NT, int WS>
__global__ __launch_bounds__(WS == 1 ? 256 : 64 * WS) void pw_gemm_kernel(const ConvArgs a) {
    constexpr bool SPLIT = WS > 1;
    constexpr int U = (MT * NT <= 2) ? 16 : 8;    // k-pairs per register buffer (two buffers in flight)
    constexpr int WN = SPLIT ? 1 : 4;
    constexpr int BN = WN * NT * 32;
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int z = blockIdx.z;
    const int g = z / a.N, n = z - g * a.N;
    const int HW = a.Wo;                          // flattened plane: Ho == 1
    const int p0 = blockIdx.x * BN;

    int pix[NT];
    const float* xb[NT];
    const float* inb = a.in + g * a.in_gs + n * a.in_ns;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        int p = p0 + ((SPLIT ? 0 : wave * NT) + t) * 32 + (lane & 31);
        pix[t] = p;
        xb[t] = inb + (long)(lane >> 5) * HW + min(p, HW - 1);
    }
    const int KP = a.nchunks * 8;                 // k-pairs in the packed weights (CK = 16)
    const int kp_real = a.Cin >> 1;               // Cin is even on this path
    int k0 = 0, k1 = kp_real;
    if constexpr (SPLIT) {
        const int per = KP / WS;
        k0 = min(wave * per, kp_real);
        k1 = min(k0 + per, kp_real);
    }
    const float* wq[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) wq[m] = a.wpk + g * a.w_gs + ((long)(blockIdx.y * MT + m) * KP) * 64 + lane;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;
    float s1[NT], s2[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) s1[t] = s2[t] = 0.f;
    const bool want_ln = (a.lnsum != nullptr);
    const long HW2 = 2L * HW;

    // software pipeline: while buffer A feeds the MFMAs, buffer B's loads are in flight
    int kp = k0;
    if (kp + U <= k1) {
        PwFrag<MT, NT, U> fa, fb;
        pw_load(fa, wq, xb, kp, HW2);
        kp += U;
        while (kp + 2 * U <= k1) {
            pw_load(fb, wq, xb, kp, HW2);
            pw_mac(fa, acc, s1, s2);
            pw_load(fa, wq, xb, kp + U, HW2);
            pw_mac(fb, acc, s1, s2);
            kp += 2 * U;
        }
        if (kp + U <= k1) {
            pw_load(fb, wq, xb, kp, HW2);
            pw_mac(fa, acc, s1, s2);
            pw_mac(fb, acc, s1, s2);
            kp += U;
        } else {
            pw_mac(fa, acc, s1, s2);
        }
    }
    for (; kp < k1; ++kp) {
        PwFrag<MT, NT, 1> f1;
        pw_load(f1, wq, xb, kp, HW2);
        pw_mac(f1, acc, s1, s2);
    }

    constexpr int RPW = SPLIT ? 4 : 16;
    const int r0 = SPLIT ? (wave & 3) * 4 : 0;
    float fin[MT][NT][RPW];
    if constexpr (SPLIT) {
        // every wave publishes its whole partial tile once; after ONE barrier wave q sums register
        // quarter [4q, 4q+4) over the WS copies (these launches sit on the sequential attention chain:
        // the earlier quarter-per-phase scheme cost eight barriers for a quarter of the LDS)
        constexpr int LNOFF = WS * MT * NT * 1024;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int rr = 0; rr < 16; ++rr)
                    lds[((((wave * MT + m) * NT + t) * 16) + rr) * 64 + lane] = acc[m][t][rr];
        if (want_ln) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                lds[LNOFF + (wave * NT + t) * 128 + lane] = s1[t];
                lds[LNOFF + (wave * NT + t) * 128 + 64 + lane] = s2[t];
            }
        }
        __syncthreads();
        if (wave >= 4) return;                    // only waves 0..3 finish rows
        const int q = wave & 3;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    float v = 0.f;
#pragma unroll
                    for (int w = 0; w < WS; ++w) v += lds[((((w * MT + m) * NT + t) * 16) + 4 * q + rr) * 64 + lane];
                    fin[m][t][rr] = v;
                }
        if (want_ln) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float u = 0.f, v = 0.f;
#pragma unroll
                for (int w = 0; w < WS; ++w) {
                    u += lds[LNOFF + (w * NT + t) * 128 + lane];
                    v += lds[LNOFF + (w * NT + t) * 128 + 64 + lane];
                }
                s1[t] = u;
                s2[t] = v;
            }
        }
    } else {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) fin[m][t][rr] = acc[m][t][rr];
    }

    float mu[NT], rstd[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) mu[t] = rstd[t] = 0.f;
    if (want_ln) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float u = s1[t] + __shfl_xor(s1[t], 32);
            float v = s2[t] + __shfl_xor(s2[t], 32);
            float mean = u / (float)a.Cin;
            float var = fmaxf(v / (float)a.Cin - mean * mean, 0.f);
            mu[t] = mean;
            rstd[t] = 1.0f / sqrtf(var + 1e-5f);
        }
    }
    generic_epilogue<MT, NT, RPW>(a, fin, pix, r0, lane, g, n, HW, HW, want_ln, mu, rstd);
}

template <int MT, int NT, int WS>
static int pw_launch_t(const ConvArgs& a, int G, hipStream_t stream) {
    constexpr int BN = (WS > 1 ? 1 : 4) * NT * 32;
    const size_t lds = WS > 1 ? (size_t)(WS * MT * NT * 1024 + WS * NT * 128) * sizeof(float) : 0;
    dim3 grid(cdiv(a.Wo, BN), cdiv(a.Cout, MT * 32), G * a.N);
    if (grid.x == 0 || grid.y == 0 || grid.z == 0) return BDE_OK;
    hipLaunchKernelGGL((pw_gemm_kernel<MT, NT, WS>), grid, dim3(WS > 1 ? 64 * WS : 256), lds, stream, a);
    BDE_HIP(hipGetLastError());
    return BDE_OK;
}

// Pick the decomposition of one pointwise GEMM launch: enough waves to cover the 1024 SIMDs,
// as little reduction as possible.
// (tuning().pw_force / pw_batched: mt*100 + nt*10 + ws overrides for the per-frame / T-batched launches)
#ifdef BDE_CONV_TU
int pw_launch_auto(const ConvArgs& a, int G, hipStream_t stream) {
    if (const int f = tuning().pw_batched; f != 0 && (long)G * a.N > 4) {
        switch (f) {
            case 221: return pw_launch_t<2, 2, 1>(a, G, stream);
            case 211: return pw_launch_t<2, 1, 1>(a, G, stream);
            case 111: return pw_launch_t<1, 1, 1>(a, G, stream);
            default: break;
        }
    }
    if (const int f = tuning().pw_force; f != 0 && (long)G * a.N <= 4) {
        switch (f) {
            case 211: return pw_launch_t<2, 1, 1>(a, G, stream);
            case 111: return pw_launch_t<1, 1, 1>(a, G, stream);
            case 114: return pw_launch_t<1, 1, 4>(a, G, stream);
            case 118: return pw_launch_t<1, 1, 8>(a, G, stream);
            default: break;
        }
    }
    const long px_tiles = cdiv(a.Wo, 32);
    const long frames = (long)G * a.N;
    const int kpairs = a.Cin / 2;
    const long t1 = px_tiles * cdiv(a.Cout, 32) * frames;     // wave tasks with MT = 1
    // T-batched launches are bound by their output stores (K is only 64..256): the lean one-tile-per-
    // wave variant (4 waves/SIMD) measured ahead of the register-heavy 2x2 and 2x1 ones
    if (t1 >= 768) return pw_launch_t<1, 1, 1>(a, G, stream);
    // few tiles (one frame of a small map): split K over the waves of a block; one 32x32 tile per
    // wave measured faster than two on the level-2 chain (more blocks, shorter per-wave chains)
    if (kpairs >= 256 && t1 * 4 < 1024) return pw_launch_t<1, 1, 8>(a, G, stream);
    if (kpairs >= 32) return pw_launch_t<1, 1, 4>(a, G, stream);
    return pw_launch_t<1, 1, 1>(a, G, stream);
}

#else
int pw_launch_auto(const ConvArgs& a, int G, hipStream_t stream);   // conv_tu.hip
#endif

}  // namespace bde
