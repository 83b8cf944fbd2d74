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
    unsigned long long* stamps;                 // diagnostic build only: s_memtime per phase, [block][wave][8]
    int debug;                                  // timing experiments: bit0 skip proj, 1 fc1, 2 fc2, 3 qkv, 4 LN stats, 5 input tiles
};

// NPT = 16-pixel tiles per workgroup (2: each weight fragment feeds two MFMAs; 1: twice the
// workgroups, for maps that would otherwise leave most CUs with a single resident block)
constexpr int tok_pitch(int npt) { return npt == 2 ? 48 : 16; }   // LDS row pitch (floats)

// One contraction phase: rows [0, nct*16) x 32 pixels, K = 4*nk4, B operand in LDS.
// Wave w owns co-tiles w, w+4, ...; `epi(row0, col, acc0, acc1)` receives a finished tile.
// The weight fragments come straight from L2, so the (tile, k) iteration space of a wave is
// flattened into batches of U fragments kept in a four-deep register ring: batches i+1..i+3 are in
// flight while batch i feeds the MFMAs, across tile boundaries as well.
template <int U, int NPT, typename Epi>
__device__ __forceinline__ void tok_gemm_u(const float* __restrict__ wpk, int nct, int nk4, const float* ldsB, int wave,
                                           int lane, Epi epi, int dbg = 0) {
    constexpr int TOK_PITCH = tok_pitch(NPT);
    const int krow = lane >> 4, col = lane & 15;
    const float* bptr = ldsB + krow * TOK_PITCH + col;
    const int ntile = (nct - wave + 3) / 4;            // co-tiles of this wave
    const int bpt = nk4 / U;                           // batches per tile
    const int nb = ntile * bpt;
    if (nb <= 0) return;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    // Loads and MACs both walk the batches strictly in order, so their (tile, k) positions are
    // running counters (no division per batch).  Loads past the end re-load the last batch:
    // they must stay UNCONDITIONAL -- a load under a branch makes hipcc's waitcnt pass fall back
    // to vmcnt(0..5) at the join and drain the ring every batch.
    const float* lwp = wpk + ((long)wave * nk4) * 64 + lane;     // next batch to load
    int lkb = 0, lleft = nb;
    const long tile_skip = ((long)4 * nk4 - (long)(bpt - 1) * U) * 64;   // last batch of a tile -> first of the next
    auto load = [&](float (&av)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) av[u] = lwp[(long)u * 64];
        if (lleft > 1) {                               // advance (pointer arithmetic only)
            --lleft;
            if (++lkb == bpt) { lkb = 0; lwp += tile_skip; } else { lwp += U * 64; }
        }
    };
    int mkb = 0, mct = wave;
    auto mac = [&](const float (&av)[U]) {
        float b0[U], b1[U];
        const float* bp = bptr + mkb * U * 4 * TOK_PITCH;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            b0[u] = bp[u * 4 * TOK_PITCH];
            b1[u] = NPT == 2 ? bp[u * 4 * TOK_PITCH + 16] : 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);             // keep the LDS reads ahead of the MFMA group
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], b0[u], acc0, 0, 0, 0);
            if constexpr (NPT == 2) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], b1[u], acc1, 0, 0, 0);
        }
        if (++mkb == bpt) {
            // C/D map of 16x16x4: row = (lane>>4)*4 + reg, col = lane&15.  The epilogue gets the whole
            // tile (rows row0..row0+3, pixels col and col+16) so it can batch its LDS reads.
            epi(mct * 16 + krow * 4, col, acc0, acc1);
            acc0 = f32x4{0.f, 0.f, 0.f, 0.f};
            acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
            mkb = 0;
            mct += 4;
        }
    };
    float f0[U], f1[U], f2[U], f3[U];
    load(f0);
    load(f1);
    load(f2);
    for (int bi = 0; bi < nb; bi += 4) {
        load(f3);
        mac(f0);
        load(f0);
        if (bi + 1 < nb) mac(f1);
        load(f1);
        if (bi + 2 < nb) mac(f2);
        load(f2);
        if (bi + 3 < nb) mac(f3);
    }
}

template <int NPT, typename Epi>
__device__ __forceinline__ void tok_gemm(const float* __restrict__ wpk, int nct, int nk4, const float* ldsB, int wave,
                                         int lane, Epi epi, int dbg = 0) {
    if ((nk4 & 7) == 0) tok_gemm_u<8, NPT>(wpk, nct, nk4, ldsB, wave, lane, epi, dbg);
    else tok_gemm_u<4, NPT>(wpk, nct, nk4, ldsB, wave, lane, epi, dbg);    // C = 16: K = 16 -> four k4 steps
}

// per-pixel mean / rstd over C channels of an LDS tile [C][TOK_PITCH]; result in stat[0..31], stat[32..63]
template <int NPT>
__device__ __forceinline__ void tok_ln_stats(const float* tile, int C, float* stat, int tid) {
    constexpr int TOK_PITCH = tok_pitch(NPT);
    const int px = (NPT == 2) ? (tid & 31) : (tid & 15), part = tid >> 5;   // 8 parts of the channel range
    // (NPT == 1: lanes 16..31 of each half duplicate pixels 0..15; their sums are simply not used)
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

#define TOK_STAMP(i)                                                                              \
    do {                                                                                          \
        if (a.stamps && lane == 0 && blockIdx.x < 64)                                             \
            a.stamps[(blockIdx.x * 4 + wave) * 8 + (i)] = __builtin_amdgcn_s_memtime();           \
    } while (0)

template <int NPT>
__global__ __launch_bounds__(256) void token_fused_kernel(const TokenArgs a) {
    constexpr int TOK_PT = 16 * NPT;
    constexpr int TOK_PITCH = tok_pitch(NPT);
    extern __shared__ __align__(16) float lds[];
    const int C = a.C, HID = 4 * a.C, HW = a.HW;
    float* tA = lds;                                   // [C][PITCH]   attention output
    float* tX = tA + C * TOK_PITCH;                    // [C][PITCH]   x -> x1 -> x2 (in place)
    float* tH = tX + C * TOK_PITCH;                    // [4C][PITCH]  hidden
    float* stat = tH + HID * TOK_PITCH;                // 64 + 4*64 floats
    // per-row epilogue parameters, staged once so that no epilogue waits on L2:
    float* pbproj = stat + 320;                        // [C]
    float* pbfc1 = pbproj + C;                         // [4C]
    float* psfc1 = pbfc1 + HID;                        // [4C]
    float* pbfc2 = psfc1 + HID;                        // [C]
    float* pbqkv = pbfc2 + C;                          // [3C]
    float* psqkv = pbqkv + 3 * C;                      // [3C]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const int p0 = blockIdx.x * TOK_PT;
    const float* aob = a.ao + b * a.bs_c;
    const float* xb = a.x + b * a.bs_c;

    TOK_STAMP(0);
    // ---- load the two input tiles (rows of 32 pixels = 128 B) ------------------------------------
    {
        const int px = tid & 31;                       // (NPT == 1: columns 16..31 are loaded but never used)
        const int p = min(p0 + (NPT == 2 ? px : (px & 15)), HW - 1);
        // eight channels per pass, all sixteen loads in flight before the first LDS store
        for (int c0 = tid >> 5; c0 < C && !(a.debug & 32); c0 += 64) {
            float va[8], vx[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int c = min(c0 + 8 * k, C - 1);
                va[k] = aob[(long)c * HW + p];
                vx[k] = xb[(long)c * HW + p];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int c = c0 + 8 * k;
                if (c < C) {
                    if (NPT == 2 || px < 16) {
                        tA[c * TOK_PITCH + px] = va[k];
                        tX[c * TOK_PITCH + px] = vx[k];
                    }
                }
            }
        }
        for (int i = tid; i < HID; i += 256) {
            pbfc1[i] = a.bfc1[i];
            psfc1[i] = a.sfc1[i];
        }
        for (int i = tid; i < C; i += 256) {
            pbproj[i] = a.bproj[i];
            pbfc2[i] = a.bfc2[i];
        }
        if (a.qkv) {
            for (int i = tid; i < 3 * C; i += 256) {
                pbqkv[i] = a.bqkv[i];
                psqkv[i] = a.sqkv[i];
            }
        }
    }
    __syncthreads();

    TOK_STAMP(1);
    // ---- x1 = x + proj(ao) (masked) ----------------------------------------------------------------
    {
        unsigned covered = 0xffffffffu;                 // bit per pixel column handled by this lane
        if (a.mask_w > 0) {
            covered = 0;
#pragma unroll
            for (int h = 0; h < NPT; ++h) {
                const int p = p0 + (lane & 15) + 16 * h;
                const int y = p / a.mask_w, xx = p - y * a.mask_w;
                const int rr = y + a.mask_pt, cc = xx + a.mask_pl;
                if (!((rr < 7 && (rr & 1)) || (cc < 7 && (cc & 1)))) covered |= 1u << h;
            }
        }
        if (!(a.debug & 1)) tok_gemm<NPT>(a.wproj, C / 16, C / 4, tA, wave, lane, [&](int row0, int col, f32x4 v0, f32x4 v1) {
            float bb[4], x0[4], x1[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                bb[r] = pbproj[row0 + r];
                x0[r] = tX[(row0 + r) * TOK_PITCH + col];
                x1[r] = NPT == 2 ? tX[(row0 + r) * TOK_PITCH + col + 16] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                tX[(row0 + r) * TOK_PITCH + col] = x0[r] + ((covered & 1u) ? v0[r] + bb[r] : 0.f);
                if constexpr (NPT == 2) tX[(row0 + r) * TOK_PITCH + col + 16] = x1[r] + ((covered & 2u) ? v1[r] + bb[r] : 0.f);
            }
        });
    }
    __syncthreads();

    TOK_STAMP(2);
    // ---- hidden = GELU(fc1(LN2(x1))) -------------------------------------------------------------
    if (!(a.debug & 16)) tok_ln_stats<NPT>(tX, C, stat, tid);
    {
        const int col = lane & 15;
        const float mu0 = stat[col], mu1 = stat[col + 16], rs0 = stat[32 + col], rs1 = stat[48 + col];   // (tile 1 unused when NPT == 1)
        if (!(a.debug & 2)) tok_gemm<NPT>(a.wfc1, HID / 16, C / 4, tX, wave, lane, [&](int row0, int c_, f32x4 v0, f32x4 v1) {
            float ss[4], bb[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ss[r] = psfc1[row0 + r];
                bb[r] = pbfc1[row0 + r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                tH[(row0 + r) * TOK_PITCH + c_] = gelu_f(rs0 * (v0[r] - mu0 * ss[r]) + bb[r]);
                if constexpr (NPT == 2) tH[(row0 + r) * TOK_PITCH + c_ + 16] = gelu_f(rs1 * (v1[r] - mu1 * ss[r]) + bb[r]);
            }
        });
    }
    __syncthreads();

    TOK_STAMP(3);
    // ---- x2 = x1 + fc2(hidden) (+ merged[t]) -------------------------------------------------------
    {
        float* x2b = a.x2 + b * a.bs_c;
        const float* adb = a.addres ? a.addres + b * a.bs_c : nullptr;
        if (!(a.debug & 4)) tok_gemm<NPT>(a.wfc2, C / 16, HID / 4, tH, wave, lane, [&](int row0, int col, f32x4 v0, f32x4 v1) {
            float bb[4], x0[4], x1[4], r0[4], r1[4];
            const int pa = p0 + col, pb = p0 + col + 16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                bb[r] = pbfc2[row0 + r];
                x0[r] = tX[(row0 + r) * TOK_PITCH + col];
                x1[r] = NPT == 2 ? tX[(row0 + r) * TOK_PITCH + col + 16] : 0.f;
                r0[r] = (adb && pa < HW) ? adb[(long)(row0 + r) * HW + pa] : 0.f;
                r1[r] = (NPT == 2 && adb && pb < HW) ? adb[(long)(row0 + r) * HW + pb] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float y0 = x0[r] + v0[r] + bb[r], y1 = x1[r] + v1[r] + bb[r];
                tX[(row0 + r) * TOK_PITCH + col] = y0;      // LN statistics / q|k|v of the next block read this
                if constexpr (NPT == 2) tX[(row0 + r) * TOK_PITCH + col + 16] = y1;
                if (pa < HW) x2b[(long)(row0 + r) * HW + pa] = y0 + r0[r];
                if (NPT == 2 && pb < HW) x2b[(long)(row0 + r) * HW + pb] = y1 + r1[r];
            }
        }, a.debug);
    }
    TOK_STAMP(4);
    if (a.qkv == nullptr) return;
    __syncthreads();

    // ---- q|k|v of the next block from x2 ------------------------------------------------------------
    if (!(a.debug & 16)) tok_ln_stats<NPT>(tX, C, stat, tid);
    {
        float* qb = a.qkv + b * a.bs_qkv;
        const int col = lane & 15;
        const float mu0 = stat[col], mu1 = stat[col + 16], rs0 = stat[32 + col], rs1 = stat[48 + col];   // (tile 1 unused when NPT == 1)
        if (!(a.debug & 8)) tok_gemm<NPT>(a.wqkv, 3 * C / 16, C / 4, tX, wave, lane, [&](int row0, int c_, f32x4 v0, f32x4 v1) {
            float ss[4], bb[4];
            const int pa = p0 + c_, pb = p0 + c_ + 16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ss[r] = psqkv[row0 + r];
                bb[r] = pbqkv[row0 + r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (pa < HW) qb[(long)(row0 + r) * HW + pa] = rs0 * (v0[r] - mu0 * ss[r]) + bb[r];
                if (NPT == 2 && pb < HW) qb[(long)(row0 + r) * HW + pb] = rs1 * (v1[r] - mu1 * ss[r]) + bb[r];
            }
        });
    }
    TOK_STAMP(5);
}

static inline size_t token_lds_bytes(int C, int npt = 2) {
    return ((size_t)(6 * C) * tok_pitch(npt) + 64 + 4 * 64 + 16 * C) * sizeof(float);
}

template <int NPT>
static int token_launch_t(const TokenArgs& a, int B, hipStream_t stream) {
    const size_t lds = token_lds_bytes(a.C, NPT);
    if (lds > 64 * 1024) {
        static unsigned char raised[BDE_MAX_DEVICES];
        BDE_HIP(raise_dynamic_lds(raised, (const void*)token_fused_kernel<NPT>));
    }
    dim3 grid(cdiv(a.HW, 16 * NPT), B);
    hipLaunchKernelGGL(token_fused_kernel<NPT>, grid, dim3(256), lds, stream, a);
    BDE_HIP(hipGetLastError());
    return BDE_OK;
}

static inline int token_launch(const TokenArgs& a, int B, hipStream_t stream) {
    int npt = tuning().tok_npt;
    if (npt == 0) npt = (cdivl(a.HW, 32) * B >= 1024) ? 2 : 1;   // enough 32-pixel tiles for ~4 blocks per CU
    return npt == 2 ? token_launch_t<2>(a, B, stream) : token_launch_t<1>(a, B, stream);
}

}  // namespace bde
