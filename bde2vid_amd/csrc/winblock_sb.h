// winblock_kernel (winblock.h) with its four GEMM phases -- q|k|v, proj, fc1, fc2: 1.53 of the block's 2.0 GFLOP and half of
// its cycles on v_mfma_f32_16x16x4_f32 -- moved to the 16-bit matrix cores with split operands (split.h: fp32 accumulate,
// fp32-equivalent accuracy): per 16 x 16 output tile and K = 64, six v_mfma_f32_16x16x32_f16 with two fp16 terms (TERMS = 2, the
// default) or twelve ..._bf16 with three bf16 terms (16 cycles each) instead of 16 fp32 MFMAs (32 cycles each).
// Scores, softmax and p*v are winblock.h's, unchanged (K = head_dim = 4 is the fp32 16x16x4 MFMA's own shape).
//
// What changes around the GEMMs:
//   * weights: split at pack time into A-fragment order of the 16x16x32 MFMA, [row tile 16][k-step 32][term][64 lanes][8]
//     (lane l = W[16 tile + (l & 15)][32 kstep + 8 (l >> 4) + j]), 16-byte loads L2 -> registers, a phase ahead;
//   * activations: every GEMM's token operand lives in LDS as TERMS 16-bit images in B-fragment order
//     [term][token tile 16][chunk of 8 channels][16 tokens][8 channels]: a lane's fragment is one ds_read_b128, the 16
//     lanes of a read group hit 16 different bank groups.  The gather splits the window's tokens once as they arrive
//     (v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32: 3 / 5.5 vector instructions per element), the attention phase writes its output already split,
//     the proj / fc1 epilogues split x1 / the hidden activations for the GEMM that follows;
//   * LDS: the split tokens (40 / 60 KB for 10 token tiles) do not fit beside K, V and Q (100 KB): the q|k|v GEMM runs in two
//     passes over a 36 KB operand buffer (laid out for three terms; two-term operands use the first two thirds) -- token tiles 0..5 with the query projection, then tiles 6..9, whose tokens wait
//     in registers meanwhile; both passes give every wave the same number of tiles (4 + 2).
//   * the query frame is kept in fp32 as well ([4][64][16], the residual x + proj(.)); pixels outside every dilated
//     window ride in token columns 49..63 as in winblock.h.
#pragma once
#include <hip/hip_runtime.h>
#include "winblock.h"
#include "conv_sb.h"

namespace bde {

constexpr int WS_XT = 0;                                   // f32 [4][64][16]: x of the query tokens (+ carried pixels), then x1
constexpr int WS_XS = WS_XT + 4 * WB_C * 16 * 4;           // bf16 x 3 operand tiles [term][6 tiles][8 chunks][16 tokens][8 ch]
constexpr int WS_XS_TILE = 2048;
constexpr int WS_XS_TERM = 6 * WS_XS_TILE;
constexpr int WS_KL = WS_XS + 3 * WS_XS_TERM;              // f32 [10][64][16]
constexpr int WS_VL = WS_KL + WB_NT * WB_C * 16 * 4;       // f32 [160][68]
constexpr int WS_QL = WS_VL + 160 * WB_VP * 4;             // f32 [4][64][16]
constexpr int WS_ST = WS_QL + 4 * WB_C * 16 * 4;           // mu[160] | rstd[160]
constexpr int WS_PR = WS_ST + 320 * 4;                     // biases and LayerNorm row sums, as in winblock.h
constexpr int WS_PIX = WS_PR + 1024 * 4;
constexpr int WS_S2 = WS_PIX + 64 * 4;
constexpr int WS_END = WS_S2 + 512 * 4;
constexpr int WS_HID = WS_KL;                              // bf16 x 3 hidden activations [term][4 tiles][32 chunks][16][8] over K | V | Q
constexpr int WS_HID_TILE = 32 * 256;
constexpr int WS_HID_TERM = 4 * WS_HID_TILE;
static_assert(WS_HID + 3 * WS_HID_TERM <= WS_ST, "hidden activations overlay K | V | Q only");
static_assert(WS_END <= 160 * 1024, "LDS budget of one workgroup per CU");

// two fp32 values -> their 16-bit roundings in one dword (round to nearest even).  Inline asm on purpose: as a vector
// conversion it seeds the SLP vectorizer, which then turns the softmax / p*v accumulators feeding it into v_pk_*_f32
// pairs -- 120 spilled registers in a phase that sits at the 128-register cap.
__device__ __forceinline__ unsigned ws_pk(float a, float b) {
    unsigned d;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ unsigned ws_pkh(float a, float b) {
    unsigned d;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ float ws_h_lo(unsigned d) {          // fp16 in bits 0..15 / 16..31 of a dword -> fp32
    float f;
    asm("v_cvt_f32_f16 %0, %1" : "=v"(f) : "v"(d));
    return f;
}
__device__ __forceinline__ float ws_h_hi(unsigned d) {
    float f;
    asm("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(f) : "v"(d));
    return f;
}
// (x0, x1) = the sum of TERMS terms, each term a packed 16-bit pair; the remainders are exact in fp32.  (No special case for
// Inf or, with fp16 terms, |x| >= 65520: the later terms become NaN; ws_split_pair_g reports such a value through the range guard
// of split.h, and the forward is recomputed with three bf16 terms.)
template <int TERMS>
__device__ __forceinline__ void ws_split_pair(float x0, float x1, unsigned (&t)[TERMS]) {
    if constexpr (TERMS == 2) {
        t[0] = ws_pkh(x0, x1);
        t[1] = ws_pkh(x0 - ws_h_lo(t[0]), x1 - ws_h_hi(t[0]));
    } else {
        t[0] = ws_pk(x0, x1);
        const float r0 = x0 - __builtin_bit_cast(float, t[0] << 16), r1 = x1 - __builtin_bit_cast(float, t[0] & 0xffff0000u);
        t[1] = ws_pk(r0, r1);
        const float q0 = r0 - __builtin_bit_cast(float, t[1] << 16), q1 = r1 - __builtin_bit_cast(float, t[1] & 0xffff0000u);
        t[2] = ws_pk(q0, q1);
    }
}

// ... with the range guard of the two-term format: gm = max(gm, |x0|, |x1|) (one v_max3_f32), flushed by the caller per phase so
// that no guard register lives across the attention phase (128-register cap)
template <int TERMS>
__device__ __forceinline__ void ws_split_pair_g(float x0, float x1, unsigned (&t)[TERMS], float& gm) {
    if constexpr (TERMS == 2) gm = sb_guard_max2(gm, x0, x1);
    ws_split_pair<TERMS>(x0, x1, t);
}

// Measured and removed (rounds 3 / 4, same-box A/B):
//   * key tile outermost (a key tile's V rows read once for the three query tiles, 40 LDS reads per wave instead of 120, online softmax
//     per tile): attention phase 28.8 k -> 30.7 k cycles, 32.2 -> 32.9 us per launch -- the V reads are not what the phase waits on;
//   * packed fp32 vector math (v_pk_fma_f32 / v_pk_add_f32 through vector builtins; as inline asm the consumers of a fresh v_exp_f32
//     result are invisible to the hazard recognizer and read garbage): 18 instead of 30 vector instructions per score tile,
//     2123 / 1756 frames/s against 2151 / 1764 -- no gain.
// p . v on the matrix pipe (default): v_mfma_f32_4x4x1_16b_f32 = sixteen independent 4x4 outer products per instruction, lane 4 b + i
// gives row i of block b's A column and column i of its B row, D[i][j] of block b sits in register i of lane 4 b + j (probe:
// tools/ubench/mfma4x4.hip).  A score tile leaves P^T[key 4 g4 + r][query col] in register r of lane (col, g4): with a = p[r] and
// b = V[key 4 g4 + r][channel col & 3], block (g4, col >> 2) accumulates O[query 4 (col >> 2) + i][channel j] over this lane's key
// quarter -- four MFMAs per score tile instead of sixteen v_fma_f32, and V is read as one 16-byte LDS load per tile (four keys of one
// channel) from a transposed image instead of four.  The per-query factors (running-maximum corrections, 1 / sum) reach the
// transposed accumulator through quad broadcasts (DPP), the sum over the key quarters is a reduce-scatter (3 exchanges, not 8).
// Measured (tools/win_stamps.py, one box): attention phase 27.5 k cycles against 28.1 k with the vector FMAs (-DWS_PV_MFMA=0) -- a
// dependent 4x4x1 MFMA costs the SIMD ~12 cycles, four of them what sixteen FMAs cost; two or four independent accumulation chains
// change nothing (28.0 k).  What it does save is registers (98 instead of 128) and 90 of the 120 V reads per wave.
#ifndef WS_PV_MFMA
#define WS_PV_MFMA 1
#endif
// V^T [64 channels][160 keys] fp32 in the VL region; the key's tile parity is flipped by two channel bits so that the reads of a
// wave (channels 4 h .. + 3 x four 16-byte key groups) touch 64 different banks and the q|k|v epilogue's writes conflict 2-way, not 4-way
__device__ __forceinline__ int ws_vt(int ch, int key) { return ch * 160 + (key ^ ((((ch >> 1) ^ (ch >> 2)) & 1) << 4)); }
template <int I>
__device__ __forceinline__ float ws_quad_bcast(float v) {          // lane 4 q + I's value in the four lanes of quad q
    const int x = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(x, x, I * 0x55, 0xf, 0xf, true));
}
__device__ __forceinline__ void ws_quad_scale(f32x4& o, float f) {  // o[i] *= f of lane 4 q + i
    o[0] *= ws_quad_bcast<0>(f);
    o[1] *= ws_quad_bcast<1>(f);
    o[2] *= ws_quad_bcast<2>(f);
    o[3] *= ws_quad_bcast<3>(f);
}
// two-term weights are packed times a power of two (split.h): the accumulator of GEMM i is multiplied by a.unscale[i]
#define WS_US(x, i) (TERMS == 2 ? (x) * us_s[i] : (x))
template <int TERMS>
__global__ __launch_bounds__(1024) void winblock_sb_kernel(const WinArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    float* XT = reinterpret_cast<float*>(smem + WS_XT);
    unsigned char* XS = smem + WS_XS;
    float* KL = reinterpret_cast<float*>(smem + WS_KL);
    float* VL = reinterpret_cast<float*>(smem + WS_VL);
    float* QL = reinterpret_cast<float*>(smem + WS_QL);
    float* ST = reinterpret_cast<float*>(smem + WS_ST);
    float* PR = reinterpret_cast<float*>(smem + WS_PR);
    int* PIX = reinterpret_cast<int*>(smem + WS_PIX);
    float* S2 = reinterpret_cast<float*>(smem + WS_S2);
    unsigned char* HS = smem + WS_HID;
    float* pbqkv = PR, *psqkv = PR + 192, *pbproj = PR + 384, *pbfc1 = PR + 448, *psfc1 = PR + 704, *pbfc2 = PR + 960;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave-uniform: lives in a scalar register
    const int g4 = lane >> 4, col = lane & 15;
    const int b = blockIdx.z;
    const int HW = a.H * a.W;
    const bool mlp_only = (int)blockIdx.x >= a.nWin;
    const int ntok = mlp_only ? 64 : WB_TOK * a.nslots;

    WB_STAMP(0);
    // ---- weight fragments of the first contraction, in flight while the tokens are gathered ---------
    const int rtkv = 4 + (wave & 7), rtq = wave & 3;
    sb8 akv[2][TERMS], aqw[2][TERMS];
    if (!mlp_only) {
        const sb8* wq = reinterpret_cast<const sb8*>(a.wqkvS) + lane;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int k = 0; k < TERMS; ++k) {
                akv[ks][k] = wq[((rtkv * 2 + ks) * TERMS + k) * 64];
                aqw[ks][k] = wq[((rtq * 2 + ks) * TERMS + k) * 64];
            }
    }
    // Everything below is requested in ONE round trip.  (ISA of round 4: the six-way `if` that filled PR was six loads under
    // per-lane branches, each joined by s_waitcnt vmcnt(0); a.unscale[i] read where it is used was a load + vmcnt(0) inside the
    // GEMM phases; a.slot[sl] / a.rowsA[ri] indexed by a per-lane value are vector loads from the kernel-argument segment in front
    // of the token loads -- nine dependent round trips before the first token arrived.)
    float us_v[4] = {1.f, 1.f, 1.f, 1.f};
    if constexpr (TERMS == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) us_v[i] = a.unscale[i];
    }
    float pr_v;
    {
        const int i = tid;                          // 1024 threads, 1024 values: pointer select, one unconditional load
        const float* src = i < 192 ? a.bqkv + i : i < 384 ? a.sqkv + (i - 192) : i < 448 ? a.bproj + (i - 384)
                         : i < 704 ? a.bfc1 + (i - 448) : i < 960 ? a.sfc1 + (i - 704) : a.bfc2 + (i - 960);
        pr_v = *src;
    }
    // the per-slot and uncovered-pixel arguments as opaque scalars, picked by selects (D <= 3 slots, <= 3 rows / columns)
    const float* slot_s[3] = {a.slot[0], a.slot[1], a.slot[2]};
    long slot_bs_s[3] = {a.slot_bs[0], a.slot_bs[1], a.slot_bs[2]};
    int rowsA_s[3] = {a.rowsA[0], a.rowsA[1], a.rowsA[2]}, colsB_s[3] = {a.colsB[0], a.colsB[1], a.colsB[2]};
#pragma unroll
    for (int i = 0; i < 3; ++i) asm volatile("" : "+s"(slot_s[i]), "+s"(slot_bs_s[i]), "+s"(rowsA_s[i]), "+s"(colsB_s[i]));
    auto pick3 = [](const auto& arr, int i) { return i == 0 ? arr[0] : (i == 1 ? arr[1] : arr[2]); };

    // ---- gather: wave = token tile, lane = (token col, quarter g4): channels 8 g4 .. + 7 and 32 + 8 g4 .. + 7 (chunks g4, 4 + g4) ----
    // waves 0..9: the window's tokens of all frames (query frame first); wave 10: pixels outside every dilated window
    // carried in token columns 49..63; mlp-only workgroups: waves 0..3 = 64 such pixels.
    float xv[16];                                   // this thread's 16 channels of its token
    bool have_tok = false;                          // the thread holds a token of tiles 0..9 (or an mlp-only pixel)
    {
        const int nextra = a.nA + a.nB;
        auto uncovered_pixel = [&](int e) {         // e-th pixel outside every dilated window
            int y, x;
            if (e < a.nA) {
                const int ri = e / a.W;
                y = pick3(rowsA_s, ri);
                x = e - ri * a.W;
            } else {
                const int e2 = e - a.nA;
                const int yi = e2 / a.ncolsB;
                x = pick3(colsB_s, e2 - yi * a.ncolsB);
                y = yi;
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    if (k < a.nrowsA && y >= rowsA_s[k]) ++y;
            }
            return y * a.W + x;
        };
        const int u = wave * 16 + col;
        const bool carried = !mlp_only && wave == 10;
        int sl = 0, pix = -1;
        bool active = false;
        if (carried) {
            const int e = (int)blockIdx.x * a.ke + col;
            active = col < 15;
            if (col < a.ke && e < nextra) pix = uncovered_pixel(e);
        } else if (mlp_only) {
            const int e = ((int)blockIdx.x - a.nWin) * 64 + u;
            active = wave < 4;
            if (active && e < nextra) pix = uncovered_pixel(e);
        } else if (wave < 10) {
            active = true;
            if (u < ntok) {
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
        }
        if (g4 == 0 && active) {
            if (carried) PIX[WB_TOK + col] = pix;
            else if (u < 64 && (mlp_only || u < WB_TOK)) PIX[u] = pix;
        }
        const float* sp = pick3(slot_s, sl);
        const bool live = active && pix >= 0 && sp != nullptr;
        // pointer select + unconditional loads (a load under a per-lane branch costs a vmcnt(0) join)
        // (explicitly global: a pointer that went through an opaque asm operand is a generic one, and its loads flat_load)
        typedef const __attribute__((address_space(1))) f32x4 gf4;
        gf4* src = (gf4*)reinterpret_cast<const f32x4*>(live ? sp + b * pick3(slot_bs_s, sl) + (long)pix * WB_C : slot_s[0]);
        f32x4 v[4];
        v[0] = src[live ? 2 * g4 : 0];
        v[1] = src[live ? 2 * g4 + 1 : 0];
        v[2] = src[live ? 8 + 2 * g4 : 0];
        v[3] = src[live ? 9 + 2 * g4 : 0];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const float w[4] = {live ? v[f][0] : 0.f, live ? v[f][1] : 0.f, live ? v[f][2] : 0.f, live ? v[f][3] : 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xv[4 * f + e] = w[e];
                s1 += w[e];
                s2 += w[e] * w[e];
            }
        }
        s1 = wb_rows_sum(s1);
        s2 = wb_rows_sum(s2);
        have_tok = active && !carried;
        // fp32 copy for the residual: the query frame's tokens (tiles 0..2 and token 48), the carried pixels (tile 3,
        // columns 1..15), all four tiles of an mlp-only workgroup
        int xt_col = -1;
        if (carried) xt_col = 48 + 1 + col;
        else if (mlp_only) { if (active) xt_col = u; }
        else if (u < WB_TOK) xt_col = u;
        if (xt_col >= 0 && active) {
            float* dst = XT + (xt_col >> 4) * WB_C * 16 + (xt_col & 15);
#pragma unroll
            for (int k = 0; k < 16; ++k) dst[((k < 8 ? 8 * g4 : 24 + 8 * g4) + k) * 16] = xv[k];
        }
        if (mlp_only && active) {                  // x1 = x: LayerNorm2 sums in the layout the proj epilogue leaves
            S2[(g4 * 64 + u) * 2] = g4 == 0 ? s1 : 0.f;
            S2[(g4 * 64 + u) * 2 + 1] = g4 == 0 ? s2 : 0.f;
        }
        if (g4 == 0 && have_tok && !mlp_only) {
            const float mean = s1 * (1.f / WB_C);
            const float var = fmaxf(s2 * (1.f / WB_C) - mean * mean, 0.f);
            ST[u] = mean;
            ST[160 + u] = __builtin_amdgcn_rsqf(var + 1e-5f);
        }
    }
    // this thread's token as split terms into operand tile `jl` of XS (chunks g4 and 4 + g4)
    auto write_split = [&](int jl) {
        unsigned t[2][4][TERMS];
        float gm = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int p = 0; p < 4; ++p) ws_split_pair_g<TERMS>(xv[8 * h + 2 * p], xv[8 * h + 2 * p + 1], t[h][p], gm);
        if (TERMS == 2) sb_guard_flush(gm, a.ovf);
#pragma unroll
        for (int k = 0; k < TERMS; ++k)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                *reinterpret_cast<uint4*>(XS + k * WS_XS_TERM + jl * WS_XS_TILE + (4 * h + g4) * 256 + col * 16) =
                    uint4{t[h][0][k], t[h][1][k], t[h][2][k], t[h][3][k]};
    };
    PR[tid] = pr_v;
    float us_s[4];                                  // (readfirstlane: uniform values into scalar registers)
#pragma unroll
    for (int i = 0; i < 4; ++i) us_s[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, us_v[i])));
    if (have_tok && (mlp_only || wave < 6)) write_split(wave);
    wb_sync();
    WB_STAMP(1);

    // B-operand fragments of token tile `jl` of an operand region (chunk stride 256 B, tile stride `tile_b`, term stride `term_b`)
    auto load_b = [&](const unsigned char* base, int term_b, int tile_b, int jl, int ks, sb8 (&bf)[TERMS]) {
#pragma unroll
        for (int k = 0; k < TERMS; ++k) bf[k] = *reinterpret_cast<const sb8*>(base + k * term_b + jl * tile_b + (ks * 4 + g4) * 256 + col * 16);
    };

    sb8 a1[2][TERMS];                                  // fc1 fragments of this wave's row tile, fetched a phase early
    if (mlp_only) {
        const sb8* w1 = reinterpret_cast<const sb8*>(a.wfc1S) + lane;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int k = 0; k < TERMS; ++k) a1[ks][k] = w1[((wave * 2 + ks) * TERMS + k) * 64];
    } else {                                       // (else, not a second if: a1 must not be live across the attention phase)
        // ---- k|v of all frames (8 row tiles x 10 token tiles) and q of the query frame (4 x 4), two passes ------------
        const int gp = wave >> 3;
        auto kv_tile = [&](int j, int jl) {
            sb8 b0[TERMS], b1[TERMS];
            load_b(XS, WS_XS_TERM, WS_XS_TILE, jl, 0, b0);
            load_b(XS, WS_XS_TERM, WS_XS_TILE, jl, 1, b1);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = sb_mma16<TERMS>(akv[0], b0, acc);
            acc = sb_mma16<TERMS>(akv[1], b1, acc);
            const int u = j * 16 + col;
            const float mu = ST[u], rs = ST[160 + u];
            const int row0 = rtkv * 16 + g4 * 4;           // row in q|k|v
            float val[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) val[r] = rs * (WS_US(acc[r], 0) - mu * psqkv[row0 + r]) + pbqkv[row0 + r];
            if (rtkv < 8) {
#pragma unroll
                for (int r = 0; r < 4; ++r) KL[(j * WB_C + (row0 - 64) + r) * 16 + col] = val[r];
            } else {
#if WS_PV_MFMA
#pragma unroll
                for (int r = 0; r < 4; ++r) VL[ws_vt(row0 - 128 + r, u)] = val[r];
#else
                *reinterpret_cast<float4*>(VL + u * WB_VP + (row0 - 128)) = float4{val[0], val[1], val[2], val[3]};
#endif
            }
        };
#pragma unroll 1
        for (int jj = 0; jj < 3; ++jj) kv_tile(gp * 3 + jj, gp * 3 + jj);
        {
            const int j = wave >> 2;
            sb8 b0[TERMS], b1[TERMS];
            load_b(XS, WS_XS_TERM, WS_XS_TILE, j, 0, b0);
            load_b(XS, WS_XS_TERM, WS_XS_TILE, j, 1, b1);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = sb_mma16<TERMS>(aqw[0], b0, acc);
            acc = sb_mma16<TERMS>(aqw[1], b1, acc);
            const int u = j * 16 + col;
            const float mu = ST[u], rs = ST[160 + u];
            const int row0 = rtq * 16 + g4 * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                QL[(j * WB_C + row0 + r) * 16 + col] = rs * (WS_US(acc[r], 0) - mu * psqkv[row0 + r]) + pbqkv[row0 + r];
        }
        wb_sync();                                  // every wave is done with operand tiles 0..5
        if (wave >= 6 && wave < 10) write_split(wave - 6);
        wb_sync();
#pragma unroll 1
        for (int jj = 0; jj < 2; ++jj) kv_tile(6 + gp * 2 + jj, gp * 2 + jj);
        wb_sync();
        WB_STAMP(2);

        // ---- attention: wave = head (winblock.h, unchanged but for the store of its output) ----------------------------
        sb8 ap[2][TERMS];                              // proj fragments (4 row tiles x 4 token tiles, one per wave)
        {
            const int h = wave;
            constexpr int NQT = 3;                 // query tiles on the MFMA path: queries 0..47; query 48 below
            float kf[WB_NT], qf[NQT];
#pragma unroll
            for (int j = 0; j < WB_NT; ++j) kf[j] = KL[(j * WB_C + h * WB_HD + g4) * 16 + col];
#pragma unroll
            for (int i = 0; i < NQT; ++i) qf[i] = QL[(i * WB_C + h * WB_HD + g4) * 16 + col];
            const f32x4* bf = reinterpret_cast<const f32x4*>(a.biasF + (long)h * 4 * WB_NT * 256) + lane;
            constexpr int HT = WB_NT / 2;
            f32x4 sc[2][HT];
#pragma unroll
            for (int j = 0; j < HT; ++j) sc[0][j] = bf[j * 64];
            float s48[4];
            {
                const int jt = min(col, WB_NT - 1);
                const f32x4 b48 = *(reinterpret_cast<const f32x4*>(a.biasF + ((long)(h * 4 + 3) * WB_NT + jt) * 256) + g4 * 16);
#pragma unroll
                for (int r = 0; r < 4; ++r) s48[r] = col < WB_NT ? b48[r] : -1e30f;
            }
            float pm[NQT], pl[NQT];                                    // per query tile: max and sum over this lane's keys
#if WS_PV_MFMA
            // V^T rows of channel 4 h + (col & 3): keys 4 g4 .. + 3 of tile j at vte / vto + 16 j (even / odd j: ws_vt's parity flip)
            const int vch = h * WB_HD + (col & 3);
            const int vsw = (((vch >> 1) ^ (vch >> 2)) & 1) << 4;
            const float* vte = VL + vch * 160 + 4 * g4 + vsw;
            const float* vto = VL + vch * 160 + 4 * g4 - vsw;
            auto vload = [&](int j) { return *reinterpret_cast<const f32x4*>(((j & 1) ? vto : vte) + 16 * j); };
            f32x4 po[NQT];                                             // O[query 4 (col >> 2) + i][channel col & 3] over this lane's key quarter
#else
            const float* vbase = VL + (g4 * 4) * WB_VP + h * WB_HD;
            float po[NQT][4];                                          // p*v of this lane's keys, query col
#endif
#pragma unroll
            for (int i = 0; i < NQT; ++i) {
                float mx = -INFINITY, l = 0.f;
#if WS_PV_MFMA
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
#else
                float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
#endif
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int cb = hf, nx = hf ^ 1;
                    const int nt = (i * 2 + hf + 1);                     // next half-tile overall
                    if (nt < 2 * NQT) {
#pragma unroll
                        for (int j = 0; j < HT; ++j) sc[nx][j] = bf[(nt * HT + j) * 64];
                    }
#if WS_PV_MFMA
                    f32x4 vb[2];
                    vb[0] = vload(hf * HT);
#else
                    f32x4 vb[2][4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) vb[0][r] = *reinterpret_cast<const f32x4*>(vbase + ((hf * HT) * 16 + r) * WB_VP);
#endif
                    __builtin_amdgcn_sched_barrier(0);
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
                        l *= corr;
#if WS_PV_MFMA
                        ws_quad_scale(o, corr);
#else
                        o0 *= corr; o1 *= corr; o2 *= corr; o3 *= corr;
#endif
                    }
                    mx = m2;
#pragma unroll
                    for (int j = 0; j < HT; ++j) {
                        if (j + 1 < HT) {
#if WS_PV_MFMA
                            vb[(j + 1) & 1] = vload(hf * HT + j + 1);
#else
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                vb[(j + 1) & 1][r] = *reinterpret_cast<const f32x4*>(vbase + ((hf * HT + j + 1) * 16 + r) * WB_VP);
#endif
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        float pr[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) pr[r] = __builtin_amdgcn_exp2f(sc[cb][j][r] - mx);
#if WS_PV_MFMA
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            l += pr[r];
                            o = __builtin_amdgcn_mfma_f32_4x4x1f32(pr[r], vb[j & 1][r], o, 0, 0, 0);
                        }
#else
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const f32x4 v = vb[j & 1][r];
                            l += pr[r];
                            o0 += pr[r] * v[0];
                            o1 += pr[r] * v[1];
                            o2 += pr[r] * v[2];
                            o3 += pr[r] * v[3];
                        }
#endif
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                pm[i] = mx; pl[i] = l;
#if WS_PV_MFMA
                po[i] = o;
#else
                po[i][0] = o0; po[i][1] = o1; po[i][2] = o2; po[i][3] = o3;
#endif
            }
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
                const float m48 = wb_wave_max(wb_max3(s48[0], s48[1], fmaxf(s48[2], s48[3])));
                l48 = 0.f;
                o48[0] = o48[1] = o48[2] = o48[3] = 0.f;
#if WS_PV_MFMA
                f32x4 v48[4];                      // [channel][key 4 g4 + r of tile jt]
#pragma unroll
                for (int c = 0; c < 4; ++c) v48[c] = *reinterpret_cast<const f32x4*>(VL + ws_vt(h * WB_HD + c, jt * 16 + g4 * 4));
#endif
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pr = __builtin_amdgcn_exp2f(s48[r] - m48);
#if WS_PV_MFMA
                    const float v[4] = {v48[0][r], v48[1][r], v48[2][r], v48[3][r]};
#else
                    const f32x4 v = *reinterpret_cast<const f32x4*>(VL + (jt * 16 + g4 * 4 + r) * WB_VP + h * WB_HD);
#endif
                    l48 += pr;
#pragma unroll
                    for (int c = 0; c < 4; ++c) o48[c] += pr * v[c];
                }
                l48 = wb_wave_sum(l48);
#pragma unroll
                for (int c = 0; c < 4; ++c) o48[c] = wb_wave_sum(o48[c]);
            }
            // merge the four key quarters of each query (lanes col, col+16, col+32, col+48)
            float M[NQT], f[NQT];
#pragma unroll
            for (int i = 0; i < NQT; ++i) M[i] = wb_rows_max(pm[i]);
#if WS_PV_MFMA
            // register ii of po[i] belongs to query 4 (col >> 2) + ii: its quarter's factor comes from that lane of the quad.  The sum over
            // the quarters is a reduce-scatter (3 exchanges instead of 8): lane (col, g4) ends with query 4 (col >> 2) + g4, channel col & 3.
            float gm_o = 0.f;
#pragma unroll
            for (int i = 0; i < NQT; ++i) {
                f[i] = __builtin_amdgcn_exp2f(pm[i] - M[i]);
                const float lsum = wb_rows_sum(pl[i] * f[i]);          // the query's softmax denominator (the same in its four lanes)
                ws_quad_scale(po[i], f[i] * __builtin_amdgcn_rcpf(lsum));
                // lanes 0..31 keep queries 4 (col >> 2) + {0, 1}, lanes 32..63 + {2, 3}; then even rows the first of the two, odd rows the second
                float x0 = po[i][0], x1 = po[i][1], y0 = po[i][2], y1 = po[i][3];
                wb_swap32(x0, y0);
                wb_swap32(x1, y1);
                float k0 = x0 + y0, k1 = x1 + y1;
                wb_swap16(k0, k1);
                const float kk = k0 + k1;
                unsigned t[TERMS];
                ws_split_pair_g<TERMS>(kk, 0.f, t, gm_o);
                unsigned char* ao = XS + i * WS_XS_TILE + (h >> 1) * 256 + (4 * (col >> 2) + g4) * 16 + (h & 1) * 8 + (col & 3) * 2;
#pragma unroll
                for (int k = 0; k < TERMS; ++k) *reinterpret_cast<unsigned short*>(ao + k * WS_XS_TERM) = (unsigned short)t[k];
            }
            if (TERMS == 2) sb_guard_flush(gm_o, a.ovf);
            if (lane < 16) {
                // token tile 3: column 0 = query 48, the other columns carry no attention output
                unsigned char* ao = XS + (h >> 1) * 256 + lane * 16 + (h & 1) * 8;
                float gm = 0.f;
                const float inv48 = 1.f / l48;
                unsigned t[2][TERMS];
                ws_split_pair_g<TERMS>(lane == 0 ? o48[0] * inv48 : 0.f, lane == 0 ? o48[1] * inv48 : 0.f, t[0], gm);
                ws_split_pair_g<TERMS>(lane == 0 ? o48[2] * inv48 : 0.f, lane == 0 ? o48[3] * inv48 : 0.f, t[1], gm);
#pragma unroll
                for (int k = 0; k < TERMS; ++k) *reinterpret_cast<uint2*>(ao + k * WS_XS_TERM + 3 * WS_XS_TILE) = uint2{t[0][k], t[1][k]};
                if (TERMS == 2) sb_guard_flush(gm, a.ovf);
            }
#else
#pragma unroll
            for (int i = 0; i < NQT; ++i) {
                f[i] = __builtin_amdgcn_exp2f(pm[i] - M[i]);
                pl[i] *= f[i];
#pragma unroll
                for (int c = 0; c < 4; ++c) po[i][c] *= f[i];
            }
#pragma unroll
            for (int i = 0; i < NQT; ++i) {
                pl[i] = wb_rows_sum(pl[i]);
#pragma unroll
                for (int c = 0; c < 4; ++c) po[i][c] = wb_rows_sum(po[i][c]);
            }
            if (lane < 16) {
                // the head's four output channels of a token = half of operand chunk h >> 1: 8 bytes per term, already split
                unsigned char* ao = XS + (h >> 1) * 256 + lane * 16 + (h & 1) * 8;
                float gm = 0.f;
#pragma unroll
                for (int i = 0; i < NQT; ++i) {
                    const float inv = 1.f / pl[i];
                    unsigned t[2][TERMS];
                    ws_split_pair_g<TERMS>(po[i][0] * inv, po[i][1] * inv, t[0], gm);
                    ws_split_pair_g<TERMS>(po[i][2] * inv, po[i][3] * inv, t[1], gm);
#pragma unroll
                    for (int k = 0; k < TERMS; ++k) *reinterpret_cast<uint2*>(ao + k * WS_XS_TERM + i * WS_XS_TILE) = uint2{t[0][k], t[1][k]};
                }
                // token tile 3: column 0 = query 48, the other columns carry no attention output
                const float inv48 = 1.f / l48;
                unsigned t[2][TERMS];
                ws_split_pair_g<TERMS>(lane == 0 ? o48[0] * inv48 : 0.f, lane == 0 ? o48[1] * inv48 : 0.f, t[0], gm);
                ws_split_pair_g<TERMS>(lane == 0 ? o48[2] * inv48 : 0.f, lane == 0 ? o48[3] * inv48 : 0.f, t[1], gm);
#pragma unroll
                for (int k = 0; k < TERMS; ++k) *reinterpret_cast<uint2*>(ao + k * WS_XS_TERM + 3 * WS_XS_TILE) = uint2{t[0][k], t[1][k]};
                if (TERMS == 2) sb_guard_flush(gm, a.ovf);
            }
#endif
            // proj fragments: requested only now -- the phase above sits at the 128-register cap, and a spilled register there costs
            // more than this load's latency (part of it passes in the barrier)
            {
                const sb8* wp = reinterpret_cast<const sb8*>(a.wprojS) + lane;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int k = 0; k < TERMS; ++k) ap[ks][k] = wp[(((wave & 3) * 2 + ks) * TERMS + k) * 64];
            }
        }
        wb_sync();
        WB_STAMP(3);

        // ---- x1 = x + proj(ao): 4 row tiles x 4 token tiles, one per wave -------------------------------
        {
            // lane coordinates recomputed from an opaque copy: index expressions shared with the phases before the attention
            // would otherwise be kept alive across it (the compiler spilled them: seven registers, each reload a scratch round trip)
            int lane_p;
            asm volatile("v_and_b32 %0, 63, %1" : "=v"(lane_p) : "v"(tid));
            const int lane = lane_p, g4 = lane_p >> 4, col = lane_p & 15;
            auto load_b = [&](const unsigned char* base, int term_b, int tile_b, int jl, int ks, sb8 (&bf)[TERMS]) {
#pragma unroll
                for (int k = 0; k < TERMS; ++k) bf[k] = *reinterpret_cast<const sb8*>(base + k * term_b + jl * tile_b + (ks * 4 + g4) * 256 + col * 16);
            };
            const int rt = wave & 3, i = wave >> 2;
            {
                const sb8* w1 = reinterpret_cast<const sb8*>(a.wfc1S) + lane;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int k = 0; k < TERMS; ++k) a1[ks][k] = w1[((wave * 2 + ks) * TERMS + k) * 64];
            }
            sb8 b0[TERMS], b1[TERMS];
            load_b(XS, WS_XS_TERM, WS_XS_TILE, i, 0, b0);
            load_b(XS, WS_XS_TERM, WS_XS_TILE, i, 1, b1);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = sb_mma16<TERMS>(ap[0], b0, acc);
            acc = sb_mma16<TERMS>(ap[1], b1, acc);
            const int row0 = rt * 16 + g4 * 4;
            const bool carried = i == 3 && col >= 1;       // token columns 49..63: x1 = x of a pixel outside every window
            float s1 = 0.f, s2 = 0.f, x1v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float* xp = XT + (i * WB_C + row0 + r) * 16 + col;
                const float v = carried ? *xp : *xp + WS_US(acc[r], 1) + pbproj[row0 + r];
                *xp = v;
                x1v[r] = v;
                s1 += v;
                s2 += v * v;
            }
            // LayerNorm2 sums over this wave's 16 rows; the four row tiles are added in a fixed order by fc1
            s1 = wb_rows_sum(s1);
            s2 = wb_rows_sum(s2);
            if (lane < 16) {
                S2[(rt * 64 + i * 16 + col) * 2] = s1;
                S2[(rt * 64 + i * 16 + col) * 2 + 1] = s2;
            }
            wb_sync();                              // every wave has read its attention-output fragments: x1 takes their place
            unsigned t[2][TERMS];
            float gm = 0.f;
            ws_split_pair_g<TERMS>(x1v[0], x1v[1], t[0], gm);
            ws_split_pair_g<TERMS>(x1v[2], x1v[3], t[1], gm);
            if (TERMS == 2) sb_guard_flush(gm, a.ovf);
            unsigned char* d = XS + i * WS_XS_TILE + (rt * 2 + (g4 >> 1)) * 256 + col * 16 + (g4 & 1) * 8;
#pragma unroll
            for (int k = 0; k < TERMS; ++k) *reinterpret_cast<uint2*>(d + k * WS_XS_TERM) = uint2{t[0][k], t[1][k]};
        }
        wb_sync();
        WB_STAMP(4);
    }

    // ---- hidden = GELU(fc1(LayerNorm2(x1))): wave = row tile, all four token tiles ------------------
    int lane_q;
    asm volatile("v_and_b32 %0, 63, %1" : "=v"(lane_q) : "v"(tid));
    const int lane2 = lane_q, g42 = lane_q >> 4, col2 = lane_q & 15;
    auto load_b2 = [&](const unsigned char* base, int term_b, int tile_b, int jl, int ks, sb8 (&bf)[TERMS]) {
#pragma unroll
        for (int k = 0; k < TERMS; ++k) bf[k] = *reinterpret_cast<const sb8*>(base + k * term_b + jl * tile_b + (ks * 4 + g42) * 256 + col2 * 16);
    };
    const sb8* w2 = reinterpret_cast<const sb8*>(a.wfc2S) + ((long)(wave & 3) * 8 * TERMS) * 64 + lane2;   // fc2 fragments [rt][8 k-steps][terms]
    sb8 wa[2][TERMS];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int k = 0; k < TERMS; ++k) wa[ks][k] = w2[(ks * TERMS + k) * 64];
    {
        const int row0 = wave * 16 + g42 * 4;
        float ss[4], bb[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ss[r] = psfc1[row0 + r];
            bb[r] = pbfc1[row0 + r];
        }
        float gm = 0.f;
#pragma unroll 2
        for (int i = 0; i < 4; ++i) {
            sb8 b0[TERMS], b1[TERMS];
            load_b2(XS, WS_XS_TERM, WS_XS_TILE, i, 0, b0);
            load_b2(XS, WS_XS_TERM, WS_XS_TILE, i, 1, b1);
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float2 pp = *reinterpret_cast<const float2*>(S2 + (t * 64 + i * 16 + col2) * 2);
                s1 += pp.x;
                s2 += pp.y;
            }
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = sb_mma16<TERMS>(a1[0], b0, acc);
            acc = sb_mma16<TERMS>(a1[1], b1, acc);
            const float mu = s1 * (1.f / WB_C);
            const float rs = __builtin_amdgcn_rsqf(fmaxf(s2 * (1.f / WB_C) - mu * mu, 0.f) + 1e-5f);
            float hv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) hv[r] = gelu_f(rs * (WS_US(acc[r], 2) - mu * ss[r]) + bb[r]);
            unsigned t[2][TERMS];
            ws_split_pair_g<TERMS>(hv[0], hv[1], t[0], gm);
            ws_split_pair_g<TERMS>(hv[2], hv[3], t[1], gm);
            unsigned char* d = HS + i * WS_HID_TILE + (wave * 2 + (g42 >> 1)) * 256 + col2 * 16 + (g42 & 1) * 8;
#pragma unroll
            for (int k = 0; k < TERMS; ++k) *reinterpret_cast<uint2*>(d + k * WS_HID_TERM) = uint2{t[0][k], t[1][k]};
        }
        if (TERMS == 2) sb_guard_flush(gm, a.ovf);
    }
    // fc2's remaining fragments (k-steps 2..7 of this wave's row tile) and the frame's residual (merged[t]) are requested HERE, in
    // front of the barrier that ends fc1: requested round by round inside the fc2 loop the scheduler sank every load to its use
    // (ISA of round 4: load, s_waitcnt vmcnt(1), MFMA -- three dependent round trips in a phase of sixteen MFMAs), and the residual
    // was a round trip of its own in front of the stores.
    sb8 wr[3][2][TERMS];
#pragma unroll
    for (int kp = 1; kp < 4; ++kp)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int k = 0; k < TERMS; ++k) wr[kp - 1][ks][k] = w2[((kp * 2 + ks) * TERMS + k) * 64];
    const int pix_o = PIX[(wave >> 2) * 16 + col2];
    f32x4 ad_v = {0.f, 0.f, 0.f, 0.f};
    {
        typedef const __attribute__((address_space(1))) f32x4 gf4;
        const bool has = a.addres != nullptr && pix_o >= 0;
        const float* adp = has ? a.addres + b * a.addres_bs + (long)pix_o * WB_C + (wave & 3) * 16 + g42 * 4 : a.bfc2;
        ad_v = *(gf4*)reinterpret_cast<const f32x4*>(adp);
        if (!has) ad_v = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __builtin_amdgcn_sched_barrier(0);
    wb_sync();
    { const int lane = lane2; WB_STAMP(5); }

    // ---- x2 = x1 + fc2(hidden) (+ merged[t]): 4 row tiles x 4 token tiles, K = 256 = 8 k-steps ----------------------
    {
        const int rt = wave & 3, i = wave >> 2;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kp = 0; kp < 4; ++kp) {                // two k-steps per round
            sb8 b0[TERMS], b1[TERMS];
            load_b2(HS, WS_HID_TERM, WS_HID_TILE, i, 2 * kp, b0);
            load_b2(HS, WS_HID_TERM, WS_HID_TILE, i, 2 * kp + 1, b1);
            if (kp == 0) {
                acc = sb_mma16<TERMS>(wa[0], b0, acc);
                acc = sb_mma16<TERMS>(wa[1], b1, acc);
            } else {
                acc = sb_mma16<TERMS>(wr[kp - 1][0], b0, acc);
                acc = sb_mma16<TERMS>(wr[kp - 1][1], b1, acc);
            }
        }
        const int pix = pix_o;
        if (pix >= 0) {
            const int row0 = rt * 16 + g42 * 4;
            float y[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = XT[(i * WB_C + row0 + r) * 16 + col2] + WS_US(acc[r], 3) + pbfc2[row0 + r];
            if (a.addres) { y[0] += ad_v[0]; y[1] += ad_v[1]; y[2] += ad_v[2]; y[3] += ad_v[3]; }
            *reinterpret_cast<float4*>(a.out + b * a.out_bs + (long)pix * WB_C + row0) = float4{y[0], y[1], y[2], y[3]};
            if (a.out_nchw) {
                float* ob = a.out_nchw + b * a.out_bs + pix;
#pragma unroll
                for (int r = 0; r < 4; ++r) ob[(long)(row0 + r) * HW] = y[r];
            }
        }
    }
    { const int lane = lane2; WB_STAMP(6); }
}

static int winblock_sb_launch(WinArgs a, int B, hipStream_t stream) {
    static unsigned char raised2[BDE_MAX_DEVICES], raised3[BDE_MAX_DEVICES];
    if (a.terms == 2) BDE_HIP(raise_dynamic_lds(raised2, (const void*)winblock_sb_kernel<2>));
    else BDE_HIP(raise_dynamic_lds(raised3, (const void*)winblock_sb_kernel<3>));
    winblock_geometry(a);
    int extra = 0;
    if (a.ke > 15) { a.ke = 0; extra = cdiv(a.nA + a.nB, 64); }
    if (a.terms == 2) hipLaunchKernelGGL(winblock_sb_kernel<2>, dim3(a.nWin + extra, 1, B), dim3(1024), WS_END, stream, a);
    else hipLaunchKernelGGL(winblock_sb_kernel<3>, dim3(a.nWin + extra, 1, B), dim3(1024), WS_END, stream, a);
    BDE_HIP(hipGetLastError());
    return BDE_OK;
}

}  // namespace bde
