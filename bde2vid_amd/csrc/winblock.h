// One temporal-attention block (SwinTransformerBlock3D.forward, DTransformer.py:286-306) of a 64-channel /
// 16-head level as ONE launch: a workgroup owns a 7x7 window and takes it from the block input to the
// block output without leaving the CU.
//
//   gather   the window's 49 tokens of the query frame and of the other D-1 buffer frames (zero padding,
//            dilation and zero frames as in window_partition / forward_part1, DTransformer.py:40-60,255-275)
//   q|k|v    = W'(LayerNorm(x)) for the query frame, k|v for the other frames (DTransformer.py:183-190),
//            on v_mfma_f32_16x16x4_f32 with the LayerNorms folded as in pw_gemm.h.  A zero token gives the
//            constant W*beta + b by itself (mean 0, variance 0), so no pad vector is needed.
//   softmax(q k^T + bias) v   per head (DTransformer.py:192-203): one wave per head; the 16 x 16 score tiles
//            come from one MFMA each (K = head_dim = 4) with the relative-position bias as its C operand,
//            the exponentials and p*v run on the vector ALU, lane = (query, quarter of the keys); the 49th
//            query (a tile of its own otherwise) is taken with the keys on the lanes instead
//   x1 = x + proj(.)          DTransformer.py:204,299
//   x2 = x1 + fc2(GELU(fc1(LayerNorm2(x1))))   (+ merged[t] after the last block, V5.py:166)
//
// Against the split path (attn.h + token_fused.h + the K|V GEMMs of pw_gemm.h) nothing but x2 is written
// to memory: level 0 of config A moved 1.2 GB of K|V planes per sequence through HBM before.
// Pixels a dilated block's fold never writes (DTransformer.py:79-82) take x1 = x and only need the MLP half:
// they ride in the 15 token columns a window leaves free (49 tokens in four 16-column tiles), a few per
// window; only if there were more than 15 per window would extra workgroups of the launch take them.
//
// Frames are read and written TOKEN-MAJOR, [pixel][64 channels]: a token is 256 contiguous bytes, so the
// window gather moves exactly the bytes it needs (the 7-pixel rows of a window in NCHW planes cost 6.8x
// their size in 64-byte lines).  nchw_to_tok_kernel converts a level's frames once per sequence; the last
// block of a frame also writes the NCHW plane the convolutions downstream read.
//
// LDS ([tile][row][16] = the B-fragment order of the 16x16x4 MFMA: four k-rows x 16 tokens are 64
// consecutive floats):  XT tokens x channels (x, then x1 in place)   KL keys   QL queries
//                       VL [token][channel] (the p*v loop reads four channels of a key at once)
//                       AO attention output (over XT's tiles 4..7)   HID hidden (over KL|VL)
#pragma once
#include <hip/hip_runtime.h>
#include "common.h"
#include "conv_mfma.h"
#include "token_fused.h"

namespace bde {

constexpr int WB_C = 64;            // channels
constexpr int WB_NH = 16;           // heads = waves per workgroup
constexpr int WB_HD = 4;            // head dim
constexpr int WB_HID = 256;
constexpr int WB_TOK = 49;
constexpr int WB_NT = 10;           // 16-token tiles of the D*49 <= 160 tokens (query frame first)
constexpr int WB_VP = 68;           // row pitch of VL
constexpr int WB_MAXD = 3;

struct WinArgs {
    const float* slot[WB_MAXD];     // [0] = query frame (block input), then the other buffer frames in slot
                                    // order; nullptr = zero frame.  Token-major [B][HW][C]
    long slot_bs[WB_MAXD];
    const float* addres;            // token-major, added to the result (merged[t]) or nullptr
    float* out;                     // token-major [B][HW][C]
    float* out_nchw;                // optional second copy of the result as [B][C][HW]
    long out_bs, addres_bs;
    const float *wqkv, *bqkv, *sqkv;            // packed16 [12][16][64], [192] bias', [192] row sums
    const float *wproj, *bproj;                 // packed16 [4][16][64], [64]
    const float *wfc1, *bfc1, *sfc1;            // packed16 [16][16][64], [256], [256]
    const float *wfc2, *bfc2;                   // packed16 [4][64][64], [64]
    // winblock_sb.h: the same four weight matrices as three bf16 terms in A-fragment order of the 16x16x32 MFMA,
    // [row tile 16][k-step 32][term][64 lanes][8]
    const unsigned short *wqkvS, *wprojS, *wfc1S, *wfc2S;
    int terms;                      // split format of the four (split.h); unscale[i] undoes the packing scale of GEMM i (two terms;
    const float* unscale;           // four floats in the packed image)
    const float* biasF;             // [16 heads][4 query tiles][10 key tiles][64 lanes][4]: a lane's four C-operand values of a score tile as one 16-byte load, log2(e) folded,
                                    // keys beyond D*49 = -1e30
    int nslots;                     // D
    int H, W, Hp, Wp, pt, pl, nWw, nWin, dilated;
    // pixels outside every window of a dilated block: nA in full rows rowsA[], then nB in columns colsB[]
    int nA, nB, nrowsA, ncolsB, rowsA[3], colsB[3];
    int ke;                         // such pixels carried per window (0: extra workgroups take them)
    unsigned long long* stamps;     // diagnostics only: s_memtime per phase, [block < 64][wave < 4][8]
    unsigned* ovf;                  // winblock_sb.h, two-term operands: the forward's overflow word (range guard, split.h)
};

#define WB_STAMP(i)                                                                               \
    do {                                                                                          \
        if (a.stamps && lane == 0 && wave < 4 && blockIdx.x < 64)                                 \
            a.stamps[(blockIdx.x * 4 + wave) * 8 + (i)] = __builtin_amdgcn_s_memtime();           \
    } while (0)

__device__ __forceinline__ float wb_shfl_xor(float v, int m) { return __shfl_xor(v, m); }
// Workgroup barrier for data handed over through LDS only: waits for this wave's LDS traffic, not for its
// global loads, so weight fragments fetched a phase ahead stay in flight across it (__syncthreads() drains
// vmcnt as well).
__device__ __forceinline__ void wb_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
typedef float f32x2 __attribute__((ext_vector_type(2)));
// ---- cross-lane exchanges on the vector ALU ------------------------------------------------------------------------------------
// __shfl_xor is ds_bpermute_b32: an LDS-pipe instruction (one per ~8 cycles for the whole CU, ~100 cycles of latency).  The attention
// phase's merges issued ~70 of them per wave -- 1100 per window, a third of the phase (measured: the phase without its score loop
// still took 9.3 k of 28.7 k cycles).  gfx950's v_permlane16_swap / v_permlane32_swap exchange the odd 16-lane rows (the upper 32
// lanes) of one register with the even rows (the lower 32 lanes) of another: with a copy as the second operand that is the value of
// lane ^ 16 (lane ^ 32) in one vector instruction.
// Inline asm, not __builtin_amdgcn_permlane{16,32}_swap: with operands that hold equal values ROCm 7.2 reads the first result for
// both (v_permlane16_swap v53, v50 ; v_add_f32 v50, v53, v53 -- tools/ubench/permlane_probe.hip prints the instruction's real
// behaviour).  The s_nop covers the two wait states a vector write needs before a permlane swap reads it; the hazard recognizer does
// not look into asm.
__device__ __forceinline__ void wb_swap16(float& x, float& y) {         // rows 1, 3 of x <-> rows 0, 2 of y
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
}
__device__ __forceinline__ void wb_swap32(float& x, float& y) {         // lanes 32..63 of x <-> lanes 0..31 of y
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
}
// sum / max over the four lanes col, col + 16, col + 32, col + 48: every one of them gets the result (a + b = b + a exactly, so the
// four copies agree bit for bit)
__device__ __forceinline__ float wb_rows_sum(float v) {
    float y = v;
    wb_swap16(v, y);
    v += y;
    y = v;
    wb_swap32(v, y);
    return v + y;
}
__device__ __forceinline__ float wb_rows_max(float v) {
    float y = v;
    wb_swap16(v, y);
    v = fmaxf(v, y);
    y = v;
    wb_swap32(v, y);
    return fmaxf(v, y);
}
// sum / max over the 16 lanes of a row through DPP (quad xor 1, quad xor 2, mirror within 8, mirror within 16)
template <int CTRL>
__device__ __forceinline__ float wb_dpp(float v) {
    const int x = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(x, x, CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wb_row16_sum(float v) {
    v += wb_dpp<0xB1>(v);
    v += wb_dpp<0x4E>(v);
    v += wb_dpp<0x141>(v);
    return v + wb_dpp<0x140>(v);
}
__device__ __forceinline__ float wb_row16_max(float v) {
    v = fmaxf(v, wb_dpp<0xB1>(v));
    v = fmaxf(v, wb_dpp<0x4E>(v));
    v = fmaxf(v, wb_dpp<0x141>(v));
    return fmaxf(v, wb_dpp<0x140>(v));
}
__device__ __forceinline__ float wb_wave_sum(float v) { return wb_rows_sum(wb_row16_sum(v)); }
__device__ __forceinline__ float wb_wave_max(float v) { return wb_rows_max(wb_row16_max(v)); }
// max of three without the quieting copies fmaxf() adds per operand (the scores are finite by construction)
__device__ __forceinline__ float wb_max3(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

__global__ __launch_bounds__(1024) void winblock_kernel(const WinArgs a) {
    extern __shared__ __align__(16) float lds[];
    float* XT = lds;                           // [10][64][16]
    float* KL = XT + WB_NT * WB_C * 16;        // [10][64][16]
    float* VL = KL + WB_NT * WB_C * 16;        // [160][68]
    float* QL = VL + 160 * WB_VP;              // [4][64][16]
    float* ST = QL + 4 * WB_C * 16;            // mu[160] | rstd[160]
    float* PR = ST + 320;                      // bqkv 192 | sqkv 192 | bproj 64 | bfc1 256 | sfc1 256 | bfc2 64
    int* PIX = reinterpret_cast<int*>(PR + 1024);   // [64] pixel of the workgroup's output tokens, -1 = none
    float* XE = PR + 1024 + 64;                // [64][16] x of the carried pixels (columns 49..63)
    float* S2 = XE + WB_C * 16;                // [4 row tiles][64 tokens][2] partial sums of x1, x1^2 for LayerNorm2
    float* AO = XT + 4 * WB_C * 16;            // [4][64][16]
    float* HID = KL;                           // [4][256][16]
    float* pbqkv = PR, *psqkv = PR + 192, *pbproj = PR + 384, *pbfc1 = PR + 448, *psfc1 = PR + 704, *pbfc2 = PR + 960;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g4 = lane >> 4, col = lane & 15;
    const int b = blockIdx.z;
    const int HW = a.H * a.W;
    const bool mlp_only = (int)blockIdx.x >= a.nWin;
    const int ntok = mlp_only ? 64 : WB_TOK * a.nslots;

    WB_STAMP(0);
    // ---- weight fragments of the first contraction, in flight while the tokens are gathered ---------
    const int rtkv = 4 + (wave & 7), rtq = wave & 3;
    float akv[16], aqw[16];
    if (!mlp_only) {
#pragma unroll
        for (int k4 = 0; k4 < 16; ++k4) {
            akv[k4] = a.wqkv[(rtkv * 16 + k4) * 64 + lane];
            aqw[k4] = a.wqkv[(rtq * 16 + k4) * 64 + lane];
        }
    }
    for (int i = tid; i < 1024; i += 1024) {
        float v;
        if (i < 192) v = a.bqkv[i];
        else if (i < 384) v = a.sqkv[i - 192];
        else if (i < 448) v = a.bproj[i - 384];
        else if (i < 704) v = a.bfc1[i - 448];
        else if (i < 960) v = a.sfc1[i - 704];
        else v = a.bfc2[i - 960];
        PR[i] = v;
    }

    // ---- gather: thread = (token u, cq); 16-byte chunks cq, cq+4, cq+8, cq+12 of the token's 256 bytes ----
    {
        const int u = tid >> 2, cq = tid & 3;
        float s1 = 0.f, s2 = 0.f;
        const int nextra = a.nA + a.nB;
        auto uncovered_pixel = [&](int e) {      // e-th pixel outside every dilated window
            int y, x;
            if (e < a.nA) {
                const int ri = e / a.W;
                y = a.rowsA[ri];
                x = e - ri * a.W;
            } else {
                const int e2 = e - a.nA;
                const int yi = e2 / a.ncolsB;
                x = a.colsB[e2 - yi * a.ncolsB];
                y = yi;
                for (int k = 0; k < a.nrowsA; ++k)
                    if (y >= a.rowsA[k]) ++y;
            }
            return y * a.W + x;
        };
        if (u < 160 + 15) {
            int sl = 0, pix = -1;
            const bool carried = u >= 160;             // columns 49..63 of the output tiles
            if (carried) {
                const int xi = u - 160;
                const int e = (int)blockIdx.x * a.ke + xi;
                if (!mlp_only && xi < a.ke && e < nextra) pix = uncovered_pixel(e);
            } else if (mlp_only) {
                const int e = ((int)blockIdx.x - a.nWin) * 64 + u;
                if (u < 64 && e < nextra) pix = uncovered_pixel(e);
            } else if (u < ntok) {
                int tok = u;
                if (u >= WB_TOK) {
                    const int v = u - WB_TOK;
                    sl = 1 + v / WB_TOK;
                    tok = v - (sl - 1) * WB_TOK;
                }
                const int win = blockIdx.x;
                const int wi = win / a.nWw, wj = win - wi * a.nWw;
                const int ta = tok / 7, tb = tok - ta * 7;
                const int step = a.dilated ? 2 : 1;
                const int rp = wi * 7 + ta * step, cp = wj * 7 + tb * step;
                const int ry = rp - a.pt, rx = cp - a.pl;
                if (rp < a.Hp && cp < a.Wp && ry >= 0 && ry < a.H && rx >= 0 && rx < a.W) pix = ry * a.W + rx;
            }
            if (cq == 0) {
                if (carried) { if (!mlp_only) PIX[WB_TOK + (u - 160)] = pix; }
                else if (u < 64 && (mlp_only || u < WB_TOK)) PIX[u] = pix;
            }
            const float* sp = a.slot[sl];
            const bool live = pix >= 0 && sp != nullptr;
            // pointer select + unconditional loads (a load under a per-lane branch costs a vmcnt(0) join)
            const float4* src = reinterpret_cast<const float4*>(live ? sp + b * a.slot_bs[sl] + (long)pix * WB_C : a.slot[0]) + (live ? cq : 0);
            float4 v[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) v[f] = src[live ? 4 * f : 0];
            float* dst = carried ? XE + (u - 160) : XT + (u >> 4) * WB_C * 16 + (u & 15);
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const float w[4] = {live ? v[f].x : 0.f, live ? v[f].y : 0.f, live ? v[f].z : 0.f, live ? v[f].w : 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dst[(4 * (cq + 4 * f) + e) * 16] = w[e];
                    s1 += w[e];
                    s2 += w[e] * w[e];
                }
            }
        }
        s1 += wb_shfl_xor(s1, 1);
        s2 += wb_shfl_xor(s2, 1);
        s1 += wb_shfl_xor(s1, 2);
        s2 += wb_shfl_xor(s2, 2);
        if (mlp_only && u < 64) {                  // x1 = x: LayerNorm2 sums in the layout the proj epilogue leaves
            S2[(cq * 64 + u) * 2] = cq == 0 ? s1 : 0.f;
            S2[(cq * 64 + u) * 2 + 1] = cq == 0 ? s2 : 0.f;
        }
        if (cq == 0 && u < 160) {
            const float mean = s1 * (1.f / WB_C);
            const float var = fmaxf(s2 * (1.f / WB_C) - mean * mean, 0.f);
            ST[u] = mean;
            ST[160 + u] = __builtin_amdgcn_rsqf(var + 1e-5f);
        }
    }
    wb_sync();
    WB_STAMP(1);

    float a1[16];                              // fc1 fragments of this wave's row tile, fetched a phase early
    if (mlp_only) {
#pragma unroll
        for (int k4 = 0; k4 < 16; ++k4) a1[k4] = a.wfc1[(wave * 16 + k4) * 64 + lane];
    }
    if (!mlp_only) {
        // ---- k|v of all frames (8 row tiles x 10 token tiles) and q of the query frame (4 x 4) -----------
        {
            const int j0 = (wave >> 3) * 5;
#pragma unroll 1
            for (int jj = 0; jj < 5; ++jj) {
                const int j = j0 + jj;
                float bq[16];
#pragma unroll
                for (int k4 = 0; k4 < 16; ++k4) bq[k4] = XT[(j * WB_C + k4 * 4 + g4) * 16 + col];
                __builtin_amdgcn_sched_barrier(0);
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k4 = 0; k4 < 16; ++k4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(akv[k4], bq[k4], acc, 0, 0, 0);
                const int u = j * 16 + col;
                const float mu = ST[u], rs = ST[160 + u];
                const int row0 = rtkv * 16 + g4 * 4;           // row in q|k|v
                float val[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) val[r] = rs * (acc[r] - mu * psqkv[row0 + r]) + pbqkv[row0 + r];
                if (rtkv < 8) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) KL[(j * WB_C + (row0 - 64) + r) * 16 + col] = val[r];
                } else {
                    *reinterpret_cast<float4*>(VL + u * WB_VP + (row0 - 128)) = float4{val[0], val[1], val[2], val[3]};
                }
            }
            {
                const int j = wave >> 2;
                float bq[16];
#pragma unroll
                for (int k4 = 0; k4 < 16; ++k4) bq[k4] = XT[(j * WB_C + k4 * 4 + g4) * 16 + col];
                __builtin_amdgcn_sched_barrier(0);
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k4 = 0; k4 < 16; ++k4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aqw[k4], bq[k4], acc, 0, 0, 0);
                const int u = j * 16 + col;
                const float mu = ST[u], rs = ST[160 + u];
                const int row0 = rtq * 16 + g4 * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    QL[(j * WB_C + row0 + r) * 16 + col] = rs * (acc[r] - mu * psqkv[row0 + r]) + pbqkv[row0 + r];
            }
        }
        wb_sync();
        WB_STAMP(2);

        // ---- attention: wave = head ------------------------------------------------------------------
        float ap[16];                          // proj fragments (4 row tiles x 4 token tiles, one per wave)
        {
            const int h = wave;
            constexpr int NQT = 3;                 // query tiles on the MFMA path: queries 0..47; query 48 below
            float kf[WB_NT], qf[NQT];
#pragma unroll
            for (int j = 0; j < WB_NT; ++j) kf[j] = KL[(j * WB_C + h * WB_HD + g4) * 16 + col];
#pragma unroll
            for (int i = 0; i < NQT; ++i) qf[i] = QL[(i * WB_C + h * WB_HD + g4) * 16 + col];
            const f32x4* bf = reinterpret_cast<const f32x4*>(a.biasF + (long)h * 4 * WB_NT * 256) + lane;
            // Half a query tile (5 key tiles = 20 scores per lane) at a time, the bias of the next half in
            // flight meanwhile; every lane keeps its own running maximum over its quarter of the keys and
            // the four quarters of all query tiles are merged at the end.
            constexpr int HT = WB_NT / 2;
            f32x4 sc[2][HT];
#pragma unroll
            for (int j = 0; j < HT; ++j) sc[0][j] = bf[j * 64];
            // The 49th query would be a tile of its own (a quarter of the softmax work for one token): it is taken
            // with the KEYS on the lanes instead, lane (col, g4) and register r <-> key 16*col + 4*g4 + r (col < 10).
            // Its bias values sit in column 0 of query tile 3 of the table.
            float s48[4];
            {
                const int jt = min(col, WB_NT - 1);
                // (tile (3, jt): the values of query column 0 and key quarter g4 sit in lane 16 * g4)
                const f32x4 b48 = *(reinterpret_cast<const f32x4*>(a.biasF + ((long)(h * 4 + 3) * WB_NT + jt) * 256) + g4 * 16);
#pragma unroll
                for (int r = 0; r < 4; ++r) s48[r] = col < WB_NT ? b48[r] : -1e30f;
            }
            const float* vbase = VL + (g4 * 4) * WB_VP + h * WB_HD;
            float pm[NQT], pl[NQT], po[NQT][4];                        // per query tile: max, sum, p*v of this lane's keys
#pragma unroll
            for (int i = 0; i < NQT; ++i) {
                float mx = -INFINITY, l = 0.f, o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int cb = hf, nx = hf ^ 1;
                    const int nt = (i * 2 + hf + 1);                     // next half-tile overall
                    if (nt < 2 * NQT) {
#pragma unroll
                        for (int j = 0; j < HT; ++j) sc[nx][j] = bf[(nt * HT + j) * 64];
                    }
                    f32x4 vb[2][4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) vb[0][r] = *reinterpret_cast<const f32x4*>(vbase + ((hf * HT) * 16 + r) * WB_VP);
                    __builtin_amdgcn_sched_barrier(0);
                    // scores^T tile: rows = keys (4*g4 + r), column = query; bias enters as the C operand
#pragma unroll
                    for (int j = 0; j < HT; ++j)
                        sc[cb][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[hf * HT + j], qf[i], sc[cb][j], 0, 0, 0);
                    float m2 = mx;
#pragma unroll
                    for (int j = 0; j < HT; ++j) {
                        m2 = wb_max3(m2, sc[cb][j][0], sc[cb][j][1]);
                        m2 = wb_max3(m2, sc[cb][j][2], sc[cb][j][3]);
                    }
                    if (hf == 1) {
                        const float corr = __builtin_amdgcn_exp2f(mx - m2);
                        l *= corr; o0 *= corr; o1 *= corr; o2 *= corr; o3 *= corr;
                    }
                    mx = m2;
#pragma unroll
                    for (int j = 0; j < HT; ++j) {
                        if (j + 1 < HT) {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                vb[(j + 1) & 1][r] = *reinterpret_cast<const f32x4*>(vbase + ((hf * HT + j + 1) * 16 + r) * WB_VP);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        float pr[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) pr[r] = __builtin_amdgcn_exp2f(sc[cb][j][r] - mx);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const f32x4 v = vb[j & 1][r];
                            l += pr[r];
                            o0 += pr[r] * v[0];
                            o1 += pr[r] * v[1];
                            o2 += pr[r] * v[2];
                            o3 += pr[r] * v[3];
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                pm[i] = mx; pl[i] = l; po[i][0] = o0; po[i][1] = o1; po[i][2] = o2; po[i][3] = o3;
            }
#pragma unroll
            for (int k4 = 0; k4 < 16; ++k4) ap[k4] = a.wproj[((wave & 3) * 16 + k4) * 64 + lane];
            // ---- query 48: keys on the lanes ----------------------------------------------------------
            float l48, o48[4];
            {
                const int jt = min(col, WB_NT - 1);
                float q48[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) q48[c] = QL[(3 * WB_C + h * WB_HD + c) * 16];     // token 48 = tile 3, column 0
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) s48[r] += q48[c] * KL[(jt * WB_C + h * WB_HD + c) * 16 + g4 * 4 + r];
                float m48 = wb_max3(s48[0], s48[1], fmaxf(s48[2], s48[3]));
#pragma unroll
                for (int sh = 1; sh < 64; sh <<= 1) m48 = fmaxf(m48, wb_shfl_xor(m48, sh));
                l48 = 0.f;
                o48[0] = o48[1] = o48[2] = o48[3] = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pr = __builtin_amdgcn_exp2f(s48[r] - m48);
                    const f32x4 v = *reinterpret_cast<const f32x4*>(VL + (jt * 16 + g4 * 4 + r) * WB_VP + h * WB_HD);
                    l48 += pr;
#pragma unroll
                    for (int c = 0; c < 4; ++c) o48[c] += pr * v[c];
                }
#pragma unroll
                for (int sh = 1; sh < 64; sh <<= 1) {
                    l48 += wb_shfl_xor(l48, sh);
#pragma unroll
                    for (int c = 0; c < 4; ++c) o48[c] += wb_shfl_xor(o48[c], sh);
                }
            }
            // merge the four key quarters of each query (lanes col, col+16, col+32, col+48)
            float M[NQT], f[NQT];
#pragma unroll
            for (int i = 0; i < NQT; ++i) M[i] = fmaxf(pm[i], wb_shfl_xor(pm[i], 16));
#pragma unroll
            for (int i = 0; i < NQT; ++i) M[i] = fmaxf(M[i], wb_shfl_xor(M[i], 32));
#pragma unroll
            for (int i = 0; i < NQT; ++i) {
                f[i] = __builtin_amdgcn_exp2f(pm[i] - M[i]);
                pl[i] *= f[i];
#pragma unroll
                for (int c = 0; c < 4; ++c) po[i][c] *= f[i];
            }
#pragma unroll
            for (int i = 0; i < NQT; ++i) {
                pl[i] += wb_shfl_xor(pl[i], 16);
#pragma unroll
                for (int c = 0; c < 4; ++c) po[i][c] += wb_shfl_xor(po[i][c], 16);
            }
#pragma unroll
            for (int i = 0; i < NQT; ++i) {
                pl[i] += wb_shfl_xor(pl[i], 32);
#pragma unroll
                for (int c = 0; c < 4; ++c) po[i][c] += wb_shfl_xor(po[i][c], 32);
            }
            if (lane < 16) {
#pragma unroll
                for (int i = 0; i < NQT; ++i) {
                    const float inv = 1.f / pl[i];
                    float* ao = AO + (i * WB_C + h * WB_HD) * 16 + lane;
#pragma unroll
                    for (int c = 0; c < 4; ++c) ao[c * 16] = po[i][c] * inv;
                }
                // token tile 3: column 0 = query 48, the other columns carry no attention output
                const float inv48 = 1.f / l48;
                float* ao3 = AO + (3 * WB_C + h * WB_HD) * 16 + lane;
#pragma unroll
                for (int c = 0; c < 4; ++c) ao3[c * 16] = lane == 0 ? o48[c] * inv48 : 0.f;
            }
        }
        wb_sync();
        WB_STAMP(3);

        // ---- x1 = x + proj(ao): 4 row tiles x 4 token tiles, one per wave -------------------------------
        {
            const int rt = wave & 3, i = wave >> 2;
            float bq[16];
#pragma unroll
            for (int k4 = 0; k4 < 16; ++k4) a1[k4] = a.wfc1[(wave * 16 + k4) * 64 + lane];
#pragma unroll
            for (int k4 = 0; k4 < 16; ++k4) bq[k4] = AO[(i * WB_C + k4 * 4 + g4) * 16 + col];
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k4 = 0; k4 < 16; ++k4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[k4], bq[k4], acc, 0, 0, 0);
            const int row0 = rt * 16 + g4 * 4;
            const bool carried = i == 3 && col >= 1;       // token columns 49..63: x1 = x of a pixel outside every window
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float* xp = XT + (i * WB_C + row0 + r) * 16 + col;
                const float v = carried ? XE[(row0 + r) * 16 + col - 1] : *xp + acc[r] + pbproj[row0 + r];
                *xp = v;
                s1 += v;
                s2 += v * v;
            }
            // LayerNorm2 sums over this wave's 16 rows; the four row tiles are added in a fixed order by fc1
            s1 += wb_shfl_xor(s1, 16);
            s2 += wb_shfl_xor(s2, 16);
            s1 += wb_shfl_xor(s1, 32);
            s2 += wb_shfl_xor(s2, 32);
            if (lane < 16) {
                S2[(rt * 64 + i * 16 + col) * 2] = s1;
                S2[(rt * 64 + i * 16 + col) * 2 + 1] = s2;
            }
        }
        wb_sync();
        WB_STAMP(4);
    }

    // ---- hidden = GELU(fc1(LayerNorm2(x1))): wave = row tile, all four token tiles ------------------
    {
        const int row0 = wave * 16 + g4 * 4;
        float ss[4], bb[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ss[r] = psfc1[row0 + r];
            bb[r] = pbfc1[row0 + r];
        }
#pragma unroll 2
        for (int i = 0; i < 4; ++i) {
            float bq[16];
#pragma unroll
            for (int k4 = 0; k4 < 16; ++k4) bq[k4] = XT[(i * WB_C + k4 * 4 + g4) * 16 + col];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float2 pp = *reinterpret_cast<const float2*>(S2 + (t * 64 + i * 16 + col) * 2);
                s1 += pp.x;
                s2 += pp.y;
            }
            __builtin_amdgcn_sched_barrier(0);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k4 = 0; k4 < 16; ++k4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[k4], bq[k4], acc, 0, 0, 0);
            const float mu = s1 * (1.f / WB_C);
            // (v_rsq_f32, 1 ulp: the IEEE sqrt + divide pair costs ~50 vector instructions per token tile in a phase that is
            //  vector-issue bound beside its MFMAs)
            const float rs = __builtin_amdgcn_rsqf(fmaxf(s2 * (1.f / WB_C) - mu * mu, 0.f) + 1e-5f);
#pragma unroll
            for (int r = 0; r < 4; ++r) HID[(i * WB_HID + row0 + r) * 16 + col] = gelu_f(rs * (acc[r] - mu * ss[r]) + bb[r]);
        }
    }
    wb_sync();
    WB_STAMP(5);

    // ---- x2 = x1 + fc2(hidden) (+ merged[t]): 4 row tiles x 4 token tiles, K = 256 ----------------------
    {
        const int rt = wave & 3, i = wave >> 2;
        const float* wp = a.wfc2 + (long)rt * 64 * 64 + lane;
        const float* hp = HID + (i * WB_HID + g4) * 16 + col;
        float w0[16], w1[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) w0[k] = wp[k * 64];
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ch = 0; ch < 4; ch += 2) {
#pragma unroll
            for (int k = 0; k < 16; ++k) w1[k] = wp[((ch + 1) * 16 + k) * 64];
            {
                float bq[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) bq[k] = hp[(ch * 16 + k) * 4 * 16];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[k], bq[k], acc, 0, 0, 0);
            }
            if (ch + 2 < 4) {
#pragma unroll
                for (int k = 0; k < 16; ++k) w0[k] = wp[((ch + 2) * 16 + k) * 64];
            }
            {
                float bq[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) bq[k] = hp[((ch + 1) * 16 + k) * 4 * 16];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[k], bq[k], acc, 0, 0, 0);
            }
        }
        const int pix = PIX[i * 16 + col];
        if (pix >= 0) {
            const int row0 = rt * 16 + g4 * 4;
            float y[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = XT[(i * WB_C + row0 + r) * 16 + col] + acc[r] + pbfc2[row0 + r];
            if (a.addres) {
                const float4 ad = *reinterpret_cast<const float4*>(a.addres + b * a.addres_bs + (long)pix * WB_C + row0);
                y[0] += ad.x; y[1] += ad.y; y[2] += ad.z; y[3] += ad.w;
            }
            *reinterpret_cast<float4*>(a.out + b * a.out_bs + (long)pix * WB_C + row0) = float4{y[0], y[1], y[2], y[3]};
            if (a.out_nchw) {
                float* ob = a.out_nchw + b * a.out_bs + pix;
#pragma unroll
                for (int r = 0; r < 4; ++r) ob[(long)(row0 + r) * HW] = y[r];
            }
        }
    }
    WB_STAMP(6);
}

// [N][C][HW] planes -> token-major [N][HW][C]; 64 pixels x 64 channels per block through LDS
__global__ __launch_bounds__(256) void nchw_to_tok_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int HW) {
    __shared__ float tile[64][65];
    const long base = (long)blockIdx.z * C * HW;
    const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int k = ty; k < 64; k += 4) {
        const int c = c0 + k, p = p0 + tx;
        tile[k][tx] = (c < C && p < HW) ? in[base + (long)c * HW + p] : 0.f;
    }
    wb_sync();
    for (int k = ty; k < 64; k += 4) {
        const int p = p0 + k, c = c0 + tx;
        if (p < HW && c < C) out[base + (long)p * C + c] = tile[tx][k];
    }
}
static int nchw_to_tok(const float* in, float* out, int N, int C, int HW, hipStream_t s) {
    hipLaunchKernelGGL(nchw_to_tok_kernel, dim3(cdiv(HW, 64), cdiv(C, 64), N), dim3(256), 0, s, in, out, C, HW);
    BDE_HIP(hipGetLastError());
    return BDE_OK;
}

constexpr size_t winblock_lds_bytes() {
    return (size_t)(2 * WB_NT * WB_C * 16 + 160 * WB_VP + 4 * WB_C * 16 + 320 + 1024 + 64 + WB_C * 16 + 512) * sizeof(float);
}

// window grid and the pixels a dilated block's fold never writes (host side of WinArgs)
static void winblock_geometry(WinArgs& a) {
    a.nWw = a.Wp / 7;
    a.nWin = (a.Hp / 7) * a.nWw;
    a.nA = a.nB = a.nrowsA = a.ncolsB = 0;
    if (a.dilated) {
        // padded coordinates (rr, cc) with an odd value below 7 lie outside every dilated window
        for (int o = 1; o < 7; o += 2) {
            const int y = o - a.pt, x = o - a.pl;
            if (y >= 0 && y < a.H) a.rowsA[a.nrowsA++] = y;
            if (x >= 0 && x < a.W) a.colsB[a.ncolsB++] = x;
        }
        a.nA = a.nrowsA * a.W;
        a.nB = (a.H - a.nrowsA) * a.ncolsB;
    }
    a.ke = cdiv(a.nA + a.nB, a.nWin);
}

static int winblock_launch(WinArgs a, int B, hipStream_t stream) {
    static unsigned char raised[BDE_MAX_DEVICES];
    BDE_HIP(raise_dynamic_lds(raised, (const void*)winblock_kernel));
    winblock_geometry(a);
    int extra = 0;
    if (a.ke > 15) { a.ke = 0; extra = cdiv(a.nA + a.nB, 64); }
    hipLaunchKernelGGL(winblock_kernel, dim3(a.nWin + extra, 1, B), dim3(1024), winblock_lds_bytes(), stream, a);
    BDE_HIP(hipGetLastError());
    return BDE_OK;
}

}  // namespace bde
