// Direct (im2col-free) convolution on the gfx950 fp32 matrix cores.
//
//   D[co][p] = sum_k A[co][k] * B[k][p],   k = (ci, ky, kx),  p = linear output pixel y*Wo+x
//
// One kernel template serves every dense contraction of the BDE2VID path:
//   * ConvLayer / head / encoder convs            (submodules.py:105-114)     KS 5|3, stride 1|2
//   * ConvLSTM gates, x-part and recurrent h-part (submodules.py:316-332)     KS 3, LSTM epilogue
//   * UpsampleConvLayer: bilinear x2 done while staging the LDS tile (submodules.py:137-147)
//   * every Linear of the window attention / MLP as a 1x1 conv over [C][H*W] planes, with the
//     LayerNorm folded into the weights and finished in the epilogue (DTransformer.py:183-190,279-283)
//
// Mapping onto v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD):
//   A operand = packed weights, one coalesced 256-B global load per (32 co x 2 ci) fragment,
//               pre-packed on the host in exactly the lane order the instruction wants;
//   B operand = input halo tile staged once per channel chunk into LDS ([ci][row][col], planar,
//               pixels contiguous -> conflict-free ds_read_b32 for stride 1), each lane reading
//               its own pixel at a wave-uniform (ci,ky,kx) offset: no im2col buffer anywhere;
//   D         = 32 co x 32 pixels per tile, pixel on the lane -> NCHW stores in 128-B segments.
// Activations stay NCHW (the reference layout), so the boundary needs no layout conversion.
//
// Two wave arrangements (4 waves / 256 threads per workgroup):
//   SPLITK=false : waves tile the pixel axis  -> block = (MT*32 co) x (4*NT*32 px), no reduction;
//                  used for the T-batched, non-recurrent launches (thousands of blocks).
//   SPLITK=true  : waves split the channel chunks of one (MT*32 co) x (NT*32 px) tile and reduce
//                  through LDS; 4x more, 4x smaller blocks for the sequential per-step launches
//                  (ConvLSTM step, attention GEMMs) where one frame must fill 256 CUs.
#pragma once
#include <hip/hip_runtime.h>
#include "common.h"
#include "split.h"

namespace bde {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_RELU6 = 2, ACT_GELU = 3 };
enum { EPI_GENERIC = 0, EPI_LSTM = 1 };

struct ConvArgs {
    // ---- tensors; z = g * N + n selects (group g, frame n) -----------------------------
    const float* in;     // [G?][N][Cin][Hs][Ws]      (Hs,Ws = source dims; = Hin,Win unless UP2)
    const float* in2;    // optional, summed with `in` while staging (skip_sum, V5.py:289-293)
    const float* wpk;    // packed weights [co_tile][chunk][tap][CK/2][64]
    const float* bias;   // [Cout] (plain) -- already Wb+b when the LayerNorm is folded
    const float* lnsum;  // [Cout] row sums of the LN-folded weights, nullptr = no LayerNorm
    const float* res1;   // optional residuals added after the activation, shape of `out`
    const float* res2;
    float* out;          // [G?][N][Cout][Ho][Wo]
    long in_gs, in2_gs, w_gs, bias_gs, res1_gs, res2_gs, out_gs;   // group strides (elements)
    long in_ns, in2_ns, res1_ns, res2_ns, out_ns;                   // frame strides (elements)
    int N;               // frames per group (grid.z = G*N)
    int Cin, Hin, Win;   // logical conv input (after the optional x2 upsample)
    int Hs, Ws;          // stored input dims
    int Cout, Ho, Wo;
    int nchunks;         // channel chunks in the packed weights (multiple of 4 when SPLITK)
    int act;
    // dilated-window coverage mask (DTransformer.py:79-82): pixels the fold never writes get
    // res1+res2 only.  mask_w = map width used to recover (y,x) from p; 0 = off.
    int mask_w, mask_pt, mask_pl;
    // >0: pixel tiles are aligned to image rows (row_tiles tiles of BN pixels per row, the last one
    // partial) so a tile never straddles two rows and its halo is KS rows x (BN*S+KS-1) columns.
    int row_tiles;
    // conv_sb.h: > 0 = 2-D pixel tiles of (BN / tile_cols) rows x tile_cols columns instead (row_tiles unused)
    int tile_cols;
    // ---- EPI_LSTM (group = direction) ---------------------------------------------------
    const float* gx;     // [G][N][4*Ch][HW]  x-part of the gates incl. bias
    float* cstate;       // [G][N][Ch][HW]    cell state, updated in place
    long gx_gs, gx_ns, c_gs, c_ns;
    int first;           // 1: h_prev == 0 -> skip the contraction entirely
    // ---- fused predI (V5.py:195-197): when pred_out is set and one workgroup holds all Cout rows of a pixel, the
    // epilogue emits act(sum_co pred_w[co] * (y[co] + pred_head[co]) + pred_b) per pixel INSTEAD of storing y
    const float* pred_w;     // [Cout]
    const float* pred_b;     // [1]
    const float* pred_head;  // [N][Cout][Ho*Wo] or nullptr
    float* pred_out;         // [N][Ho*Wo]
    int pred_sigmoid;
    const float* zeros;      // >= 16 bytes of zeros in device memory (source of out-of-image pixels for LDS-DMA staging)
    // ---- split-bf16 output: when sb_out is set the result is stored ONLY as SB16 [G?][N][Cout/16][HW][3 terms][16] bf16
    // (conv_sb.h: the input layout of the convolution that consumes it), not as fp32 planes; Cout % 32 == 0, no residuals
    unsigned short* sb_out;
    long sb_out_gs, sb_out_ns;   // group / frame strides in 16-bit elements
    // split format of sb_out and of a conv_sb input (split.h): 3 = three bf16 terms, 2 = two fp16 terms; *acc_scale undoes
    // the power-of-two scale the two-term weights were packed with (it lives in the packed image beside them)
    int sb_terms;
    const float* acc_scale;
    unsigned* sb_ovf;            // range guard of the two-term format (split.h): the forward's overflow word, written by sb_out stores
    int sb_stage_ok;             // conv_sb.h: the workgroup's LDS holds its SB16 output tile (staged, piece-major stores); set by the launcher
    int xcd_remap;               // conv_sb.h: workgroup order that keeps neighbouring tiles in one XCD's L2
    // one sweep direction per GPU (bde_split_*): the launch covers `groups` of the layer's groups, but every launch-shape
    // choice is made as if all `decide_groups` were present, so each direction computes exactly what the joint launch computes
    int lstm_groups;             // recurrent step: groups in this launch (0 = both directions)
    int decide_groups;           // 0 = the launch's own group count
};

// erf with |error| <= 1.5e-7 (Abramowitz & Stegun 7.1.26) on v_rcp_f32 / v_exp_f32: the exact-GELU
// epilogues sit on the sequential attention chain, where libm's erff (a ~60-instruction branchy
// routine) cost more than the GEMM it follows.  GELU(x) = 0.5 x (1 + erf(x/sqrt2)) then deviates
// from nn.GELU() by < 1e-7 |x|, three orders below the parity tolerance.
__device__ __forceinline__ float erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
    float p = 1.061405429f;
    p = p * t - 1.453152027f;
    p = p * t + 1.421413741f;
    p = p * t - 0.284496736f;
    p = p * t + 0.254829592f;
    const float e = __builtin_amdgcn_exp2f(-ax * ax * 1.4426950408889634f);
    const float r = 1.0f - p * t * e;
    return copysignf(r, x);
}
__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.f + erf_fast(v * 0.70710678118654752440f)); }

__device__ __forceinline__ float act_apply(float v, int act) {
    // compare + select, not fmaxf / fminf: those return the non-NaN operand, and torch's ReLU / ReLU6 hand a NaN on
    // (a voxel grid of a zero-duration event window is NaN, event_utils.py:489-495: the reference's frames then are too)
    if (act == ACT_RELU) return v < 0.f ? 0.f : v;
    if (act == ACT_RELU6) return v < 0.f ? 0.f : (v > 6.f ? 6.f : v);
    if (act == ACT_GELU) return gelu_f(v);
    return v;
}
__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

// Row of accumulator register r inside a 32x32 tile (cdna_hip_programming.md §3).
// Workgroup -> (pixel tile bx, channel group by, frame z) of the batched convolutions.  The dispatcher deals consecutive
// workgroups round-robin to the 8 XCDs, each with an L2 of its own: in the plain (x fastest) order the row tiles above and
// below a tile -- most of its halo -- and the other channel groups of its pixels live in seven other L2s.  With `remap`
// every XCD gets one contiguous range of the (frame, pixel tile, channel group) order instead, channel group fastest.
__device__ __forceinline__ void conv_block_coords(int remap, int& bx, int& by, int& z) {
    bx = blockIdx.x; by = blockIdx.y; z = blockIdx.z;
    if (remap) {
        const unsigned gx = gridDim.x, gy = gridDim.y;
        const unsigned total = gx * gy * gridDim.z;
        const unsigned lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
        const unsigned xcd = lin & 7u, q = total >> 3, rem = total & 7u;
        const unsigned L = xcd * q + min(xcd, rem) + (lin >> 3);
        by = (int)(L % gy);
        const unsigned r = L / gy;
        bx = (int)(r % gx);
        z = (int)(r / gx);
    }
}
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// Shared epilogue of the generic contractions: LayerNorm fold, bias, activation, dilated-window
// coverage mask, residual adds, NCHW store (lane = pixel -> 128-B segments).
template <int MT, int NT, int RPW>
__device__ __forceinline__ void generic_epilogue(const ConvArgs& a, const float (&fin)[MT][NT][RPW], const int (&pix)[NT],
                                                 int r0, int lane, int g, int n, int HW, int p_end, bool want_ln,
                                                 const float (&mu)[NT], const float (&rstd)[NT], int cot0 = -1) {
    if (cot0 < 0) cot0 = blockIdx.y * MT;        // first 32-row output tile of this wave (conv_sb.h passes its own)
    float* outb = a.out + g * a.out_gs + n * a.out_ns;
    const float* r1 = a.res1 ? a.res1 + g * a.res1_gs + n * a.res1_ns : nullptr;
    const float* r2 = a.res2 ? a.res2 + g * a.res2_gs + n * a.res2_ns : nullptr;
    const float* biasg = a.bias + g * a.bias_gs;
    // per-row parameters first (all loads in flight together), then the per-element work
    float bv[MT][RPW], sv[MT][RPW];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int co = min((cot0 + m) * 32 + acc_row(r0 + rr, lane), a.Cout - 1);
            bv[m][rr] = biasg[co];
            sv[m][rr] = want_ln ? a.lnsum[co] : 0.f;
        }
    if (a.pred_out != nullptr) {
        // rows of this lane: acc_row(rr, lane); the other 16 rows of a pixel sit in lane ^ 32
        const float* hb = a.pred_head ? a.pred_head + (long)n * a.Cout * HW : nullptr;
        float* po = a.pred_out + (long)n * HW;
        const float pb = a.pred_b[0];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int p = min(pix[t], p_end - 1);
            float s = 0.f;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int rr = 0; rr < RPW; ++rr) {
                    const int co = (cot0 + m) * 32 + acc_row(r0 + rr, lane);
                    if (co >= a.Cout) continue;
                    float v = act_apply(fin[m][t][rr] + bv[m][rr], a.act);
                    if (hb) v += hb[(long)co * HW + p];
                    s += a.pred_w[co] * v;
                }
            s += __shfl_xor(s, 32);
            s += pb;
            if ((lane >> 5) == 0 && pix[t] < p_end) po[p] = a.pred_sigmoid ? 1.f / (1.f + expf(-s)) : s;
        }
        return;
    }
    if (a.sb_out != nullptr) {
        // a lane's registers 4q .. 4q+3 are four consecutive output channels (acc_row): 8 bytes of each of the three terms
        static_assert(RPW % 4 == 0, "split-bf16 store takes registers in groups of four");
        unsigned short* sbb = a.sb_out + g * a.sb_out_gs + n * a.sb_out_ns;
        float gm = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int p = pix[t];
            if (p >= p_end) continue;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int q = 0; q < RPW / 4; ++q) {
                    const int co0 = (cot0 + m) * 32 + acc_row(r0 + 4 * q, lane);
                    if (co0 >= a.Cout) continue;
                    float v4[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = fin[m][t][4 * q + i];
                        if (want_ln) v = rstd[t] * (v - mu[t] * sv[m][4 * q + i]);
                        v4[i] = act_apply(v + bv[m][4 * q + i], a.act);
                    }
                    if (a.sb_terms == 2) {
                        unsigned short* d = sbb + ((long)(co0 >> 4) * HW + p) * 32 + (co0 & 15);
                        uint2 hi, lo;
                        gm = sb_guard_max2(sb_guard_max2(gm, v4[0], v4[1]), v4[2], v4[3]);
                        split2_quad(v4, hi, lo);
                        *reinterpret_cast<uint2*>(d) = hi;
                        *reinterpret_cast<uint2*>(d + 16) = lo;
                    } else {
                        unsigned short* d = sbb + ((long)(co0 >> 4) * HW + p) * 48 + (co0 & 15);
                        uint2 hi, mid, lo;
                        split3_quad(v4, hi, mid, lo);
                        *reinterpret_cast<uint2*>(d) = hi;
                        *reinterpret_cast<uint2*>(d + 16) = mid;
                        *reinterpret_cast<uint2*>(d + 32) = lo;
                    }
                }
        }
        if (a.sb_terms == 2) sb_guard_flush(gm, a.sb_ovf);
        return;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int p = pix[t];
        if (p >= p_end) continue;
        bool covered = true;
        if (a.mask_w > 0) {
            int y = p / a.mask_w, x = p - y * a.mask_w;
            int rr = y + a.mask_pt, cc = x + a.mask_pl;
            covered = !((rr < 7 && (rr & 1)) || (cc < 7 && (cc & 1)));
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int cob = (cot0 + m) * 32;
            float rv[RPW];
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr) {
                const int co = cob + acc_row(r0 + rr, lane);
                const long o = (long)min(co, a.Cout - 1) * HW + p;
                rv[rr] = 0.f;
                if (r1) rv[rr] = r1[o];
                if (r2) rv[rr] += r2[o];
            }
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr) {
                const int co = cob + acc_row(r0 + rr, lane);
                if (co >= a.Cout) continue;
                float v = fin[m][t][rr];
                if (want_ln) v = rstd[t] * (v - mu[t] * sv[m][rr]);
                v = act_apply(v + bv[m][rr], a.act);
                if (!covered) v = 0.f;
                outb[(long)co * HW + p] = v + rv[rr];
            }
        }
    }
}

template <int KS, int STRIDE, int MT, int NT, int CK, bool SPLITK, int EPI, int MAXI>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvArgs a) {
    constexpr int PAD = KS / 2;
    constexpr int TAPS = KS * KS;
    constexpr int PAIRS = CK / 2;
    constexpr int WN = SPLITK ? 1 : 4;        // waves along the pixel axis
    constexpr int BN = WN * NT * 32;          // pixels per block
    constexpr int STAGE_C = SPLITK ? 4 * CK : CK;   // channels staged per barrier pair
    constexpr int CG = SPLITK ? 8 : 1;        // channel groups of the staging work items
    constexpr int CPI = STAGE_C / CG;         // channels per work item
    // Without split-K the four waves of a block share the weight fragments of a stage: they are
    // staged through LDS as well (A region in front of the B halo tile), 4x less L2->L1 traffic
    // and no global load inside the MFMA loop.
    constexpr bool ALDS = !SPLITK;
    constexpr int FRAG = TAPS * PAIRS * 64;   // floats of one co-tile's fragments for one chunk
    constexpr int ASZ = ALDS ? MT * FRAG : 0;
    extern __shared__ __align__(16) float lds_all[];
    float* const lds = lds_all + ASZ;         // B halo tile (and split-K scratch)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int z = blockIdx.z;
    const int g = z / a.N, n = z - g * a.N;
    const int HW = a.Ho * a.Wo;
    int p0, p_end;
    if (a.row_tiles > 0) {
        const int yy = blockIdx.x / a.row_tiles, xt = blockIdx.x - yy * a.row_tiles;
        p0 = yy * a.Wo + xt * BN;
        p_end = min(p0 + BN, (yy + 1) * a.Wo);
    } else {
        p0 = blockIdx.x * BN;
        p_end = min(p0 + BN, HW);
    }
    const int p_last = p_end - 1;
    const int y_first = p0 / a.Wo, y_last = p_last / a.Wo;
    const bool one_row = (y_first == y_last);
    const int x_first = p0 - y_first * a.Wo;
    const int iy0 = y_first * STRIDE - PAD;
    const int ix0 = one_row ? x_first * STRIDE - PAD : -PAD;
    const int R = (y_last - y_first) * STRIDE + KS;
    const int IW = one_row ? (p_last - p0) * STRIDE + KS : a.Win + 2 * PAD;
    const int PS = R * IW;                    // LDS plane stride (one input channel)

    // per-lane LDS offset of each of this wave's pixel tiles (tap (0,0), even channel of a pair)
    int boff[NT];
    int pix[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        int p = p0 + ((SPLITK ? 0 : wave * NT) + t) * 32 + (lane & 31);
        pix[t] = p;
        int pc = min(p, p_last);
        int y = pc / a.Wo, x = pc - y * a.Wo;
        boff[t] = (y - y_first) * STRIDE * IW + (x * STRIDE - PAD - ix0) + (lane >> 5) * PS;
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;
    float s1[NT], s2[NT];   // per-pixel sum / sum of squares of the B operand (LayerNorm fold)
#pragma unroll
    for (int t = 0; t < NT; ++t) s1[t] = s2[t] = 0.f;
    constexpr bool want_ln = false;   // LayerNorm-folded layers are pointwise: pw_gemm.h

    const bool skip_mac = (EPI == EPI_LSTM) && a.first;
    if (!skip_mac) {
        const float* inb = a.in + g * a.in_gs + n * a.in_ns;
        const float* wg = a.wpk + g * a.w_gs;
        const long HsWs = (long)a.Hs * a.Ws;
        const int tile_elems = R * IW;
        const int nstages = SPLITK ? a.nchunks / 4 : a.nchunks;
        // ---- staging work items of this thread: (tile element, channel group), fixed for all
        //      stages.  voff = element offset of the work item inside the stage's first channel
        //      plane (so every load is uniform base + 32-bit lane offset); vmask bit it = the
        //      element lies inside the image (otherwise zero padding).
        unsigned voff[MAXI];
        int lde[MAXI], cgi[MAXI];
        unsigned vmask = 0;
        {
            const float inv_iw = 1.0f / (float)IW;
            const float inv_te = 1.0f / (float)tile_elems;
#pragma unroll
            for (int it = 0; it < MAXI; ++it) {
                const int wi = tid + it * 256;
                const int cg = (CG == 1) ? 0 : (int)(((float)wi + 0.5f) * inv_te);
                const int e = wi - cg * tile_elems;
                const int r = (int)(((float)e + 0.5f) * inv_iw);
                const int col = e - r * IW;
                const int iy = iy0 + r, ix = ix0 + col;
                const bool in_img = (iy >= 0) && (iy < a.Hin) && (ix >= 0) && (ix < a.Win);
                const bool item = wi < tile_elems * CG;
                lde[it] = item ? e + cg * PS : -1;
                cgi[it] = cg;
                if (item && in_img) {
                    vmask |= 1u << it;
                    voff[it] = (unsigned)(cg * (int)HsWs + iy * a.Ws + ix);
                } else {
                    voff[it] = 0;
                }
            }
        }
        constexpr int AKR = ALDS ? (FRAG + 255) / 256 : 0;   // loads per thread and co-tile
        float sv[MAXI][CPI];
        float aw[ALDS ? MT * AKR : 1];
        auto stage_load = [&](int st) {
            const int c0 = st * STAGE_C;
            if constexpr (ALDS) {
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const float* wb = wg + (((long)(blockIdx.y * MT + m) * a.nchunks + st) * FRAG);   // uniform
                    // all staging loads are UNCONDITIONAL (index clamped, value selected afterwards): a
                    // load under a per-lane condition makes hipcc branch around it and wait vmcnt(0)
                    // at the join -- one full memory round trip per load
#pragma unroll
                    for (int k = 0; k < AKR; ++k) aw[m * AKR + k] = wb[min(tid + k * 256, FRAG - 1)];
                }
            }
#pragma unroll
            for (int j = 0; j < CPI; ++j) {
                // channel base clamped so that even padded channels read inside the tensor
                const int cbase = min(c0 + j * CG, a.Cin - CG);
                const float* cb = inb + (long)cbase * HsWs;                                         // uniform
#pragma unroll
                for (int it = 0; it < MAXI; ++it) sv[it][j] = cb[voff[it]];   // masked when stored to LDS
            }
        };
        auto stage_store = [&](int st) {
            const int c0 = st * STAGE_C;
            if constexpr (ALDS) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int k = 0; k < AKR; ++k)
                        if (tid + k * 256 < FRAG) lds_all[m * FRAG + tid + k * 256] = aw[m * AKR + k];
            }
#pragma unroll
            for (int it = 0; it < MAXI; ++it)
                if (lde[it] >= 0) {
                    const bool in_img = (vmask >> it) & 1u;
#pragma unroll
                    for (int j = 0; j < CPI; ++j)
                        lds[lde[it] + j * CG * PS] = (in_img && (c0 + j * CG + cgi[it] < a.Cin)) ? sv[it][j] : 0.f;
                }
        };
        // software pipeline (register staged, cdna_hip_programming.md T14): the global loads of
        // stage s+1 are issued before the MFMA loop of stage s and land in LDS after it.
        stage_load(0);
        for (int st = 0; st < nstages; ++st) {
            __syncthreads();   // previous stage fully consumed
            stage_store(st);
            __syncthreads();
            if (st + 1 < nstages) stage_load(st + 1);
            // ---- contraction over this stage's channels ---------------------------------
            const int chunk = SPLITK ? st * 4 + wave : st;
            const float* ldsw = lds + (SPLITK ? wave * CK * PS : 0);
            const float* wch[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m)
                wch[m] = wg + (((long)(blockIdx.y * MT + m) * a.nchunks + chunk) * TAPS * PAIRS) * 64 + lane;
#pragma unroll 1
            for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    const int tap = ky * KS + kx;
#pragma unroll
                    for (int pr = 0; pr < PAIRS; ++pr) {
                        float av[MT], bv[NT];
#pragma unroll
                        for (int m = 0; m < MT; ++m)
                            av[m] = ALDS ? lds_all[(m * TAPS * PAIRS + tap * PAIRS + pr) * 64 + lane]
                                         : wch[m][(tap * PAIRS + pr) * 64];
#pragma unroll
                        for (int t = 0; t < NT; ++t) bv[t] = ldsw[boff[t] + pr * 2 * PS + ky * IW + kx];
#pragma unroll
                        for (int m = 0; m < MT; ++m)
#pragma unroll
                            for (int t = 0; t < NT; ++t)
                                acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[t], acc[m][t], 0, 0, 0);
                    }
                }
        }
    }

    // ---- split-K reduction through LDS, one register quarter per phase: in phase q every wave
    //      publishes registers [4q,4q+4) of all its tiles and wave q sums the four copies, so wave
    //      w finishes rows acc_row(4w..4w+3) of every tile (scratch: MT*NT*4 KiB).
    constexpr int RPW = SPLITK ? 4 : 16;     // accumulator registers this wave finalises
    const int r0 = SPLITK ? wave * 4 : 0;
    float fin[MT][NT][RPW];
    if constexpr (SPLITK) {
        if (skip_mac) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) fin[m][t][rr] = 0.f;
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                __syncthreads();
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr)
                            lds[((((wave * MT + m) * NT + t) * 4) + rr) * 64 + lane] = acc[m][t][4 * q + rr];
                if (want_ln && q == 0) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        lds[MT * NT * 1024 + (wave * NT + t) * 128 + lane] = s1[t];
                        lds[MT * NT * 1024 + (wave * NT + t) * 128 + 64 + lane] = s2[t];
                    }
                }
                __syncthreads();
                if (wave == q) {
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int t = 0; t < NT; ++t)
#pragma unroll
                            for (int rr = 0; rr < 4; ++rr) {
                                float v = 0.f;
#pragma unroll
                                for (int w = 0; w < 4; ++w)
                                    v += lds[((((w * MT + m) * NT + t) * 4) + rr) * 64 + lane];
                                fin[m][t][rr] = v;
                            }
                }
                if (want_ln && q == 0) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        float u = 0.f, v = 0.f;
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            u += lds[MT * NT * 1024 + (w * NT + t) * 128 + lane];
                            v += lds[MT * NT * 1024 + (w * NT + t) * 128 + 64 + lane];
                        }
                        s1[t] = u;
                        s2[t] = v;
                    }
                }
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

    // ---- LayerNorm statistics of each lane's pixel (both k-parities summed) -------------------
    float mu[NT], rstd[NT];
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

    // ---- epilogue ----------------------------------------------------------------------------
    if constexpr (EPI == EPI_GENERIC) {
        generic_epilogue<MT, NT, RPW>(a, fin, pix, r0, lane, g, n, HW, p_end, want_ln, mu, rstd);
    } else {
        // ConvLSTM pointwise (submodules.py:320-332): tiles m = gate i,f,o,g of the same 32 hidden
        // channels; acc holds W_h * h_prev, gx holds W_x * x + bias.
        static_assert(EPI != EPI_LSTM || MT == 4, "LSTM epilogue needs the four gate tiles");
        const int Ch = a.Cout / 4;
        float* hout = a.out + g * a.out_gs + n * a.out_ns;
        float* cst = a.cstate + g * a.c_gs + n * a.c_ns;
        const float* gxb = a.gx + g * a.gx_gs + n * a.gx_ns;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int p = pix[t];
            if (p >= p_end) continue;
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr) {
                const int hc = blockIdx.y * 32 + acc_row(r0 + rr, lane);
                if (hc >= Ch) continue;
                const long o = (long)hc * HW + p;
                float gi = fin[0][t][rr] + gxb[o];
                float gf = fin[1][t][rr] + gxb[(long)Ch * HW + o];
                float go = fin[2][t][rr] + gxb[(long)2 * Ch * HW + o];
                float gg = fin[3][t][rr] + gxb[(long)3 * Ch * HW + o];
                float cprev = a.first ? 0.f : cst[o];
                float c = sigmoidf_(gf) * cprev + sigmoidf_(gi) * tanhf(gg);
                float h = sigmoidf_(go) * tanhf(c);
                cst[o] = c;
                hout[o] = h;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// host side: geometry, LDS sizing, launch
// ------------------------------------------------------------------------------------------
struct ConvGeom {
    int KS, STRIDE, MT, NT, CK;
    bool up2, splitk;
    int epi;
};

// Largest halo tile (R*IW elements of one channel) over the blocks of a launch.
static inline long conv_tile_elems(const ConvGeom& gm, int Win, int Ho, int Wo, int row_tiles) {
    const int PAD = gm.KS / 2;
    const int BN = (gm.splitk ? 1 : 4) * gm.NT * 32;
    const int HW = Ho * Wo;
    if (row_tiles > 0) {
        const int npx = BN < Wo ? BN : Wo;
        return (long)gm.KS * ((npx - 1) * gm.STRIDE + gm.KS);
    }
    long best = 0;
    for (int p0 = 0; p0 < HW; p0 += BN) {
        int pl = (p0 + BN < HW ? p0 + BN : HW) - 1;
        int yf = p0 / Wo, yl = pl / Wo;
        int R = (yl - yf) * gm.STRIDE + gm.KS;
        int IW = (yf == yl) ? (pl - p0) * gm.STRIDE + gm.KS : Win + 2 * PAD;
        long e = (long)R * IW;
        if (e > best) best = e;
    }
    return best;
}

static inline size_t conv_lds_bytes(const ConvGeom& gm, long tile_elems, bool ln) {
    long stage = (long)(gm.splitk ? 4 * gm.CK : gm.CK) * tile_elems;
    if (!gm.splitk) stage += (long)gm.MT * gm.KS * gm.KS * (gm.CK / 2) * 64;   // weight fragments of the stage
    long red = gm.splitk ? (long)gm.MT * gm.NT * 1024 + (ln ? 4 * gm.NT * 128 : 0) : 0;
    long fl = stage > red ? stage : red;
    return (size_t)fl * sizeof(float);
}

// staging work items per thread for a tile
static inline int conv_stage_items(const ConvGeom& gm, long tile_elems) {
    return (int)((tile_elems * (gm.splitk ? 8 : 1) + 255) / 256);
}

template <int KS, int STRIDE, int MT, int NT, int CK, bool SPLITK, int EPI, int MAXI>
static int conv_launch_t(const ConvArgs& a, int G, hipStream_t stream) {
    ConvGeom gm{KS, STRIDE, MT, NT, CK, false, SPLITK, EPI};
    const long tile = conv_tile_elems(gm, a.Win, a.Ho, a.Wo, a.row_tiles);
    const size_t lds = conv_lds_bytes(gm, tile, a.lnsum != nullptr);
    if (lds > 160 * 1024)
        return fail(BDE_ERR_UNSUPPORTED, "conv tile needs %zu B of LDS (> 160 KiB): Win=%d KS=%d CK=%d", lds,
                    a.Win, KS, CK);
    if (conv_stage_items(gm, tile) > MAXI)
        return fail(BDE_ERR_UNSUPPORTED, "conv tile of %ld elements needs %d staging slots (> %d): Win=%d KS=%d",
                    tile, conv_stage_items(gm, tile), MAXI, a.Win, KS);
    auto kern = conv_mfma_kernel<KS, STRIDE, MT, NT, CK, SPLITK, EPI, MAXI>;
    if (lds > 64 * 1024) {
        static unsigned char raised[BDE_MAX_DEVICES];
        BDE_HIP(raise_dynamic_lds(raised, (const void*)kern));
    }
    constexpr int BN = (SPLITK ? 1 : 4) * NT * 32;
    const int co_rows = (EPI == EPI_LSTM) ? a.Cout / 4 : a.Cout;          // LSTM: 32 hidden ch per block
    const int co_per_block = (EPI == EPI_LSTM) ? 32 : MT * 32;
    dim3 grid(a.row_tiles > 0 ? a.Ho * a.row_tiles : cdiv(a.Ho * a.Wo, BN), cdiv(co_rows, co_per_block), G * a.N);
    if (grid.x == 0 || grid.y == 0 || grid.z == 0) return BDE_OK;
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, a);
    BDE_HIP(hipGetLastError());
    return BDE_OK;
}

// staging register slots per thread: 40 prefetched values = slots x CK channels
constexpr int conv_maxi(int KS) { return KS >= 5 ? 10 : 5; }
// channels per stage of the generic convs: 5x5 has 25 taps per channel, so half the channels of 3x3
constexpr int conv_ck(int KS) { return KS >= 5 ? 4 : 8; }
constexpr int LSTM_CK = 16;    // channels per wave and stage of the recurrent step (4 waves -> 32 per stage)

template <int KS, int STRIDE>
static int conv_launch_ks(ConvArgs a, int G, hipStream_t stream) {
    // choose (NT, tiling mode): most useful pixels per launched pixel among the shapes whose halo
    // tile fits the staging registers; NT = 2 reuses each weight fragment twice.
    constexpr int CK = conv_ck(KS);
    constexpr int CONV_MAXI = conv_maxi(KS);
    const int MT = a.Cout > 32 ? 2 : 1;
    double best = -1.0;
    int bnt = 1, brow = 0;
    for (int nt = 2; nt >= 1; --nt)
        for (int row = 1; row >= 0; --row) {
            ConvGeom gm{KS, STRIDE, MT, nt, CK, false, false, EPI_GENERIC};
            const int BN = 4 * nt * 32;
            const int rt = row ? cdiv(a.Wo, BN) : 0;
            const long tile = conv_tile_elems(gm, a.Win, a.Ho, a.Wo, rt);
            if (conv_stage_items(gm, tile) > CONV_MAXI || conv_lds_bytes(gm, tile, false) > 64 * 1024) continue;
            const double launched = row ? (double)a.Ho * rt * BN : (double)cdiv(a.Ho * a.Wo, BN) * BN;
            double score = (double)a.Ho * a.Wo / launched * (nt == 2 ? 1.0 : 0.92);
            const long blocks = (long)(launched / BN) * cdiv(a.Cout, MT * 32) * G * a.N;
            if (blocks < 512) score *= 0.5 + 0.5 * blocks / 512.0;      // keep 256 CUs busy
            if (score > best) { best = score; bnt = nt; brow = rt; }
        }
    if (best < 0) return fail(BDE_ERR_UNSUPPORTED, "no conv tiling fits: KS=%d stride=%d Win=%d Wo=%d", KS, STRIDE, a.Win, a.Wo);
    a.row_tiles = brow;
    if (MT == 2) {
        if (bnt == 2) return conv_launch_t<KS, STRIDE, 2, 2, CK, false, EPI_GENERIC, CONV_MAXI>(a, G, stream);
        return conv_launch_t<KS, STRIDE, 2, 1, CK, false, EPI_GENERIC, CONV_MAXI>(a, G, stream);
    }
    if (bnt == 2) return conv_launch_t<KS, STRIDE, 1, 2, CK, false, EPI_GENERIC, CONV_MAXI>(a, G, stream);
    return conv_launch_t<KS, STRIDE, 1, 1, CK, false, EPI_GENERIC, CONV_MAXI>(a, G, stream);
}

#ifdef BDE_CONV_TU
static int conv_launch_auto(int KS, int stride, const ConvArgs& a, int G, hipStream_t stream) {
    if (KS == 5 && stride == 1) return conv_launch_ks<5, 1>(a, G, stream);
    if (KS == 5 && stride == 2) return conv_launch_ks<5, 2>(a, G, stream);
    if (KS == 3 && stride == 1) return conv_launch_ks<3, 1>(a, G, stream);
    if (KS == 3 && stride == 2) return conv_launch_ks<3, 2>(a, G, stream);
    return fail(BDE_ERR_UNSUPPORTED, "conv KS=%d stride=%d not built", KS, stride);
}

static int lstm_launch(ConvArgs a, hipStream_t stream) {
    a.row_tiles = cdiv(a.Wo, 32);
    return conv_launch_t<3, 1, 4, 1, LSTM_CK, true, EPI_LSTM, 4>(a, 2, stream);
}

#endif

}  // namespace bde
