// Recurrent ConvLSTM step on v_mfma_f32_16x16x4_f32 (exact fp32) -- the kernel on the sequential
// critical path of every level (V5.py:122-135; ConvLSTM.forward, submodules.py:293-334).
//
//   gates[4][Ch] = conv3x3(h_prev; W[:, C:])  (+ gx = conv3x3(x; W[:, :C]) + bias, hoisted and batched
//   over T by conv_mfma.h);  i,f,o = sigmoid, g = tanh;  c = f*c + i*g;  h = o*tanh(c).
//
// Why 16x16x4 here: a workgroup owns 64 pixels of one image row x 16 hidden channels x 4 gates, so a
// wave's accumulators are 4 tiles x 4 registers (16 VGPRs, against 64 for the 32x32x2 shape).  With
// the weight fragments and the halo tile both staged through LDS (shared by the four waves, next stage
// prefetched into ~25 registers) the kernel runs at 5+ waves/SIMD, and a level-0 step of config A is
// 1472 workgroups of equal cost -- the 32x32 version was register-bound at 2 waves/SIMD.
//   A = packed weights  [hidden16 block][channel chunk 8][tap 9][k4 2][gate 4][64 lanes],
//       lane l = W[gate*Ch + hb*16 + (l&15)][ci = chunk*8 + k4*4 + (l>>4)][tap]
//   B = LDS halo tile [8 channels][3 rows][66 cols], plane stride padded to 16 mod 32 floats so the
//       four channel rows of a fragment fall on disjoint banks
//   D: row = hidden channel, col = pixel -> the four gates of a (channel, pixel) sit in one lane.
#pragma once
#include <hip/hip_runtime.h>
#include "conv_mfma.h"

namespace bde {

typedef float f32x4_ __attribute__((ext_vector_type(4)));

constexpr int L16_CK = 8;                        // channels per stage
constexpr int L16_AFL = 9 * 2 * 4 * 64;          // floats of one stage's weight fragments (18 KiB)
// plane stride of the halo tile, padded to 16 (mod 32) floats
constexpr int l16_ps(int rows, int pxw) { return ((rows + 2) * (pxw + 2) + 15) / 32 * 32 + 16; }

// A workgroup covers ROWS image rows x PXW pixels, one 16-pixel segment per wave (ROWS*PXW/16 waves:
// 4 or 8), and HB blocks of 16 hidden channels.  <1,64> for wide maps, <2,32> / <4,16> for narrow ones
// so that all waves have pixels.  In-kernel stamps (level 0 of config A): ~80 % MFMA duty while four
// workgroups share a CU, 54 % for a workgroup alone; the launch is 1472 workgroups on 1024 resident
// slots, and the second, thinly populated round costs about a quarter of it.  HB = 2 (half the waves,
// one round) was built to fix that but needs 176 registers -> 2 waves/SIMD, and measured slower.
// Uses the ConvArgs fields of EPI_LSTM (in = h_prev, out = h, gx, cstate, first, strides).
template <int ROWS, int PXW, int HB>
__global__ __launch_bounds__(ROWS * PXW * 4) void lstm16_step_kernel(const ConvArgs a) {
    constexpr int NT = ROWS * PXW * 4;               // threads per workgroup (64 per 16-pixel segment)
    constexpr int AFL4 = HB * L16_AFL / 4;           // float4 of one stage's weight fragments
    constexpr int L16_AK4 = (AFL4 + NT - 1) / NT;    // 16-byte weight loads per thread and stage
    constexpr int L16_IW = PXW + 2;
    constexpr int L16_R = ROWS + 2;
    constexpr int L16_PS = l16_ps(ROWS, PXW);
    constexpr int L16_BK = (L16_CK * L16_R * L16_IW + NT - 1) / NT;
    constexpr int XT = PXW / 16;                     // waves along x
    __shared__ __align__(16) float ldsA[HB * L16_AFL];
    __shared__ __align__(16) float ldsB[L16_CK * L16_PS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int z = blockIdx.z;
    const int g = z / a.N, n = z - g * a.N;
    const int Ch = a.Cout / 4, W = a.Wo, H = a.Ho, HW = H * W;
    const int tiles_x = (W + PXW - 1) / PXW;
    const int ty = blockIdx.x / tiles_x;
    const int y0 = ty * ROWS, x0 = (blockIdx.x - ty * tiles_x) * PXW;
    const int hb0 = blockIdx.y * HB;                         // first hidden16 block of this workgroup
    const int nhb = (Ch + 15) / 16;
    const int wrow = wave / XT, wx = wave - wrow * XT;       // this wave's row and 16-pixel segment
    const int pxl = wx * 16 + (lane & 15);                   // x of this lane inside the tile
    const int y = y0 + wrow;
    const bool pvalid = (x0 + pxl) < W && y < H;
    const bool wave_active = (x0 + wx * 16) < W && y < H;

    f32x4_ acc[HB][4];
#pragma unroll
    for (int h = 0; h < HB; ++h)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[h][q] = f32x4_{0.f, 0.f, 0.f, 0.f};

    if (!a.first) {
        const float* inb = a.in + g * a.in_gs + n * a.in_ns;
        const float* wgrp = a.wpk + g * a.w_gs;
        // fixed staging slots of this thread for the halo tile: element e = (ci, r, col)
        int boffs[L16_BK];
        unsigned bsrc[L16_BK];
        unsigned bmask = 0;
#pragma unroll
        for (int k = 0; k < L16_BK; ++k) {
            const int e = tid + k * NT;
            const int ci = e / (L16_R * L16_IW), rem = e - ci * (L16_R * L16_IW);
            const int r = rem / L16_IW, col = rem - r * L16_IW;
            const int iy = y0 - 1 + r, ix = x0 - 1 + col;
            const bool item = e < L16_CK * L16_R * L16_IW;
            boffs[k] = item ? ci * L16_PS + r * L16_IW + col : -1;
            const bool ok = item && iy >= 0 && iy < H && ix >= 0 && ix < W;
            if (ok) bmask |= 1u << k;
            bsrc[k] = ok ? (unsigned)(ci * HW + iy * W + ix) : 0u;
        }
        // weight staging slot k of this thread: float4 index i4 inside the HB stacked fragment blocks
        float4 aw[L16_AK4];
        float bw[L16_BK];
        auto stage_load = [&](int st) {
            // 16-byte loads (a stage of dword loads queued for thousands of cycles on the VMEM path)
#pragma unroll
            for (int k = 0; k < L16_AK4; ++k) {
                const int i4 = tid + k * NT;
                const int h = i4 / (L16_AFL / 4), r4 = i4 - h * (L16_AFL / 4);
                const int hbc = min(hb0 + h, nhb - 1);       // (a padded hidden block re-reads the last one)
                const float4* wsrc = reinterpret_cast<const float4*>(wgrp + ((long)hbc * a.nchunks + st) * L16_AFL);
                aw[k] = (i4 < AFL4) ? wsrc[r4] : float4{0.f, 0.f, 0.f, 0.f};
            }
            const float* cb = inb + (long)st * L16_CK * HW;
            // exec-masked loads: fastest of the three zero-padding forms tried (select at load time,
            // select at LDS-store time, masked load)
#pragma unroll
            for (int k = 0; k < L16_BK; ++k) bw[k] = ((bmask >> k) & 1u) ? cb[bsrc[k]] : 0.f;
        };
        const int bofl = (lane >> 4) * L16_PS + wrow * L16_IW + pxl;   // tap (0,0), k4 = 0
        const int nst = a.nchunks;
        stage_load(0);
        for (int st = 0; st < nst; ++st) {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < L16_AK4; ++k) {
                const int i4 = tid + k * NT;
                if (i4 < AFL4) reinterpret_cast<float4*>(ldsA)[i4] = aw[k];
            }
#pragma unroll
            for (int k = 0; k < L16_BK; ++k)
                if (boffs[k] >= 0) ldsB[boffs[k]] = bw[k];
            __syncthreads();
            if (st + 1 < nst) stage_load(st + 1);
            if (!wave_active) continue;                       // this wave's 16 pixels lie outside the map
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    // all LDS reads of a tap first, then its MFMAs (hipcc otherwise emits
                    // read -> lgkmcnt(0) -> MFMA chains and exposes the LDS latency per k-step)
                    float bq[2], aq[HB][2][4];
#pragma unroll
                    for (int k4 = 0; k4 < 2; ++k4) {
                        bq[k4] = ldsB[bofl + k4 * 4 * L16_PS + ky * L16_IW + kx];
#pragma unroll
                        for (int h = 0; h < HB; ++h) {
                            const float* ap = ldsA + h * L16_AFL + (((ky * 3 + kx) * 2 + k4) * 4) * 64 + lane;
#pragma unroll
                            for (int q = 0; q < 4; ++q) aq[h][k4][q] = ap[q * 64];
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k4 = 0; k4 < 2; ++k4)
#pragma unroll
                        for (int h = 0; h < HB; ++h)
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                acc[h][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[h][k4][q], bq[k4], acc[h][q], 0, 0, 0);
                }
        }
    }

    // ---- pointwise (submodules.py:320-332); gate order i, f, o, g ------------------------------
    if (!pvalid) return;
    const long p = (long)y * W + x0 + pxl;
    float* hout = a.out + g * a.out_gs + n * a.out_ns;
    float* cst = a.cstate + g * a.c_gs + n * a.c_ns;
    const float* gxb = a.gx + g * a.gx_gs + n * a.gx_ns;
#pragma unroll
    for (int h = 0; h < HB; ++h) {
        float gv[4][4], cprev[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int hc = min((hb0 + h) * 16 + (lane >> 4) * 4 + r, Ch - 1);
            const long o = (long)hc * HW + p;
#pragma unroll
            for (int q = 0; q < 4; ++q) gv[q][r] = gxb[(long)q * Ch * HW + o];
            cprev[r] = a.first ? 0.f : cst[o];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int hc = (hb0 + h) * 16 + (lane >> 4) * 4 + r;
            if (hc >= Ch) continue;
            const long o = (long)hc * HW + p;
            const float gi = acc[h][0][r] + gv[0][r], gf = acc[h][1][r] + gv[1][r];
            const float go = acc[h][2][r] + gv[2][r], gg = acc[h][3][r] + gv[3][r];
            const float c = sigmoidf_(gf) * cprev[r] + sigmoidf_(gi) * tanhf(gg);
            cst[o] = c;
            hout[o] = sigmoidf_(go) * tanhf(c);
        }
    }
}

inline int& lstm16_shape_ref() { static int v = 0; return v; }    // tuning: hb*100000 + rows*1000 + pxw, 0 = auto

template <int ROWS, int PXW, int HB>
static int lstm16_launch_t(const ConvArgs& a, hipStream_t stream) {
    const int Ch = a.Cout / 4;
    dim3 grid(cdiv(a.Ho, ROWS) * cdiv(a.Wo, PXW), cdiv(Ch, 16 * HB), 2 * a.N);
    hipLaunchKernelGGL((lstm16_step_kernel<ROWS, PXW, HB>), grid, dim3(ROWS * PXW * 4), 0, stream, a);
    BDE_HIP(hipGetLastError());
    return BDE_OK;
}

static int lstm16_launch(const ConvArgs& a, hipStream_t stream) {
    if (a.Cin % L16_CK != 0) return fail(BDE_ERR_UNSUPPORTED, "recurrent step: %d hidden channels is not a multiple of %d", a.Cin, L16_CK);
    // tile shape with the most active 16-pixel wave segments per launched wave
    auto fill = [&](int rows, int pxw) {
        const double segs = (double)a.Ho * cdiv(a.Wo, 16);
        return segs / ((double)cdiv(a.Ho, rows) * cdiv(a.Wo, pxw) * (rows * pxw / 16));
    };
    const int shapes[3][2] = {{1, 64}, {2, 32}, {4, 16}};
    int rows = 1, pxw = 64;
    double bf = -1;
    for (auto& sh : shapes) { const double f = fill(sh[0], sh[1]); if (f > bf + 0.02) { bf = f; rows = sh[0]; pxw = sh[1]; } }
    // one hidden16 block per workgroup gives 4 waves/SIMD of residency (1024 workgroups on the chip);
    // above that, two blocks per workgroup (3 waves/SIMD, 768 workgroups) keep the launch in one round
    const long wg1 = (long)cdiv(a.Ho, rows) * cdiv(a.Wo, pxw) * cdiv(a.Cout / 4, 16) * 2 * a.N;
    (void)wg1;
    int hb = 1;   // measured: HB = 2 needs 176 registers (2 waves/SIMD) and loses 6-12 % at every level of config A
    if (const int f = lstm16_shape_ref()) { hb = f / 100000; rows = (f / 1000) % 100; pxw = f % 1000; }
#define BDE_L16(R_, P_)                                                                  \
    if (rows == R_ && pxw == P_)                                                         \
        return hb == 2 ? lstm16_launch_t<R_, P_, 2>(a, stream) : lstm16_launch_t<R_, P_, 1>(a, stream);
    BDE_L16(1, 64)
    BDE_L16(2, 32)
    BDE_L16(4, 16)
#undef BDE_L16
    return fail(BDE_ERR_ARG, "recurrent step: tile shape %dx%d not built", rows, pxw);
}

}  // namespace bde
