// Level-2 attention chain (head_dim 16, 256 channels, wideblock.h): the token half of a block -- x1 = x + proj(attention output),
// hidden = GELU(fc1(LayerNorm2(x1))), x2 = x1 + fc2(hidden) (+ merged[t]) (DTransformer.py:299, 279-283, 304; V5.py:166) -- as ONE
// launch on the sequential chain of V5.py:154-169 instead of two (projfc1_sb_kernel + the fc2 GEMM).
//
// What bounds these launches is not arithmetic (0.8 GFLOP) but how many weight bytes a CU can have in flight: a workgroup of four
// waves that owns 16 tokens streams 0.75 MB of two-term weight fragments (proj 256 KB, a quarter of fc1 and of fc2 256 KB each)
// from L2, and with 8 fragments (8 KB) per wave in flight a CU pulls ~35 GB/s (512 KB in 13 us: what projfc1_sb_kernel measured).
//   * Every wave runs its three GEMM phases as ONE stream of 24 k-step groups (8 per phase; a group = the wave's four 16-row
//     tiles x two terms = 8 fragments of 1 KB) with the fragments of PD groups ahead in registers: 24-32 KB per wave, ~100 KB per
//     CU in flight, across the phase boundaries (the barriers between phases wait for LDS only, never for the stream).
//   * x1 and the hidden activations go from one GEMM to the next through LDS already split into two fp16 terms in B-fragment
//     order of the 16x16x32 MFMA (natural k order: a D-fragment lane's four consecutive channels are 8 bytes of one B lane), so
//     the consumer's operand is two 16-byte LDS reads per k-step and nobody splits a value twice.
//   * fc2 needs the 1024 hidden values of a token, which four workgroups (quarters) produce: each contracts ITS quarter of K and
//     leaves a partial x2; the workgroup whose counter add comes last sums the four partials IN A FIXED ORDER (q = 0, 1, 2, 3:
//     the result does not depend on which workgroup that is), adds merged[t] and stores x2.  Hand-off as MI355X_MICROARCH.md
//     prescribes for a last-arriver: 16-byte sc1 stores, every storing wave's vmcnt(0), workgroup barrier, ONE agent-scope atomic
//     add per workgroup on one counter per token tile, sc1 loads by the workgroup whose add returned 3.  Nobody waits for anybody.
// Results are those of the two launches it replaces up to fp32 summation order (fc2's K is summed per quarter).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "wideblock.h"
#include "wide_core.h"           // SPL16

namespace bde {

struct MlpFusedArgs {
    const float* ao;              // FRAG16 [B][ntile][16][256]: attention output
    const float* x;               // FRAG16, same shape: the block input (shortcut)
    const unsigned short* wprojS; // proj, two fp16 terms, k in FRAG16 group-pair order (TokGemmArgs::wS): its operand comes as fp32 FRAG16
    const unsigned short* wfc1S;  // fc1 / fc2, two fp16 terms, natural k order [row tile 16][k-step 32][term][64 lanes][8]
    const unsigned short* wfc2S;
    const float* unscale_proj;    // inverse packing scales (split.h)
    const float* unscale_mlp;     // {fc1, fc2}
    const float *bproj, *bfc1, *sfc1, *bfc2;
    float* part;                  // [4 quarters][B][ntile][16 row tiles][64 lanes][4]: partial x2 of a quarter of the hidden rows
    int* count;                   // [B][ntile], zero between launches: quarters that have stored their partial
    float* out;                   // FRAG16 [B][ntile][16][256]: x2
    float* out_nchw;              // optional second copy as [B][C][HW]
    const float* addres;          // optional FRAG16 tensor added to x2 (merged[t], V5.py:166)
    long x_bs, nchw_bs;
    int HW, ntile, B;
    int mask_w, mask_pt, mask_pl; // dilated-window coverage mask on the proj output (uncovered pixels: shortcut only)
    unsigned* ovf;                // range guard of the two-term format (split.h)
    unsigned long long* stamps;   // diagnostics only: s_memtime per phase, [workgroup < 64][wave][8]
    unsigned short* out_spl;      // optional: x2 also as SPL16 (wide_core.h) [B][ntile][8][2][64][8] ...
    float* out_stats;             // ... with its LayerNorm statistics [B][ntile * 16][2] = (mean, rstd): the next core's operand
    long spl_bs, st_bs;
};
#define WM_STAMP(i)                                                                                                  \
    do {                                                                                                             \
        if (a.stamps && lane == 0 && blockIdx.z == 0 && blockIdx.x * 4 + blockIdx.y < 64)                            \
            a.stamps[((blockIdx.x * 4 + blockIdx.y) * 4 + wave) * 8 + (i)] = __builtin_amdgcn_s_memtime();           \
    } while (0)

// compile-time loop: f(std::integral_constant<int, I>) for I = 0 .. N - 1 (the 24-group stream below must be straight-line code:
// its register ring is indexed by the group number)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ void st_sc1_f4(float* p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ f32x4 ld_sc1_f4(const float* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}

#ifndef WM_PD
#define WM_PD 3
#endif
#ifndef WM_XCD_REMAP
#define WM_XCD_REMAP 1
#endif
// C = 256 channels, hidden = 1024.  grid (token tiles, 4 quarters of the hidden rows, B), 256 threads.
__global__ __launch_bounds__(256) void mlp_fused_kernel(const MlpFusedArgs a) {
    constexpr int NKS = 8;                 // k-steps of 32 per phase (K = 256 per workgroup in all three GEMMs)
    constexpr int NG = 3 * NKS;            // k-step groups of the wave's fragment stream
    constexpr int PD = WM_PD;              // groups in flight ahead of the one being contracted
    constexpr int RING = PD + 1;
    __shared__ __align__(16) unsigned char X1S[NKS * 2 * 1024];   // x1 as split B fragments [k-step][term][64 lanes][16 B]
    __shared__ __align__(16) unsigned char HS[NKS * 2 * 1024];    // this quarter's hidden activations likewise
    __shared__ __align__(16) float PR[1024];                      // bproj | bfc1 (quarter) | sfc1 (quarter) | bfc2
    __shared__ float ST[4][16][2];
    __shared__ int last_flag;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g4 = lane >> 4, col = lane & 15;
    // Workgroup -> (token tile, quarter, batch), quarter-major inside each XCD (see wide_core.h): an XCD then streams proj and ONE
    // quarter of fc1 / fc2 (768 KB) through its L2 instead of all four (2.3 MB).
    int tile, quarter, b;
    {
        const unsigned gx = gridDim.x, total = gx * 4u * gridDim.z;
        const unsigned lin = blockIdx.x + gx * (blockIdx.y + 4u * blockIdx.z);
        unsigned L = lin;
        if (WM_XCD_REMAP) {
            const unsigned xcd = lin & 7u, q = total >> 3, rem = total & 7u;
            L = xcd * q + min(xcd, rem) + (lin >> 3);
        }
        const unsigned q1 = (unsigned)(((float)L + 0.5f) * __builtin_amdgcn_rcpf((float)gx));   // (exact: wide_core.h)
        tile = (int)(L - q1 * gx);
        quarter = (int)(q1 & 3u);
        b = (int)(q1 >> 2);
    }
    const int tok = tile * 16 + col;
    WM_STAMP(0);
    // The accumulator scales of the three GEMMs (split.h) are requested FIRST and turned into scalars behind the first fragment
    // groups (a counted wait: they are the oldest loads).  Read where they are used -- the end of a phase -- each was the YOUNGEST
    // load in flight, and the wait for it (vmcnt(0), ISA of round 4) drained the whole fragment ring twice per launch.
    const float us_proj_v = a.unscale_proj[0], us_fc1_v = a.unscale_mlp[0], us_fc2_v = a.unscale_mlp[1];
    __builtin_amdgcn_sched_barrier(0);

    // ---- operands that do not depend on anything computed here: all requested before the first MFMA ------------------------------
    // Every workgroup of a launch streams the same weight bytes (all of them proj, the 44 of a quarter its fc1 / fc2 rows) and
    // the workgroups run in lockstep: walking K in the same order they would all ask the same L2 channel for the same lines at the
    // same time.  Each workgroup therefore starts its K loop at a k-step of its own (`rot`, order rot, rot + 1, ... mod 8): the
    // sum over K is the same set of products in a rotated order -- fixed per workgroup, so results stay bit-reproducible.
    const int rot = __builtin_amdgcn_readfirstlane((tile + 3 * quarter) & (NKS - 1));
    const wf4* ap = reinterpret_cast<const wf4*>(a.ao + b * a.x_bs) + ((long)tile * 16) * 64 + lane;
    wf4 aov[16];                           // aov[2 j], aov[2 j + 1]: the operand of k-step (j + rot) mod 8
#pragma unroll
    for (int kg = 0; kg < 16; ++kg) aov[kg] = ap[((kg + 2 * rot) & 15) * 64];
    float xr[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const long fo = ((long)tile * 16 + 4 * wave + m) * 256 + col * 4 + g4;
#pragma unroll
        for (int r = 0; r < 4; ++r) xr[m][r] = a.x[b * a.x_bs + fo + r * 64];
    }
    {
        const int i = tid * 4;
        const float* src = i < 256 ? a.bproj + i : i < 512 ? a.bfc1 + 256 * quarter + (i - 256)
                                   : i < 768 ? a.sfc1 + 256 * quarter + (i - 512) : a.bfc2 + (i - 768);
        *reinterpret_cast<float4*>(PR + i) = *reinterpret_cast<const float4*>(src);
    }
    // the wave's fragment stream: group G = (phase G / 8, k-step G % 8), fragments [row tile m of the wave][term]
    const sb8* wp0 = reinterpret_cast<const sb8*>(a.wprojS) + ((long)(4 * wave) * NKS * 2) * 64 + lane;
    const sb8* wp1 = reinterpret_cast<const sb8*>(a.wfc1S) + ((long)(16 * quarter + 4 * wave) * NKS * 2) * 64 + lane;
    const sb8* wp2 = reinterpret_cast<const sb8*>(a.wfc2S) + (((long)(4 * wave) * 4 * NKS + NKS * quarter) * 2) * 64 + lane;
    sb8 ring[RING][4][2];
    auto fetch = [&](auto Gc) {
        constexpr int G = decltype(Gc)::value;
        constexpr int ph = G / NKS;
        const int ks = (G % NKS + rot) & (NKS - 1);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                // row tile stride in fragments: 2 NKS (proj, fc1: K = 256), 8 NKS (fc2: K = 1024)
                const sb8* p = ph == 0 ? wp0 + ((m * NKS + ks) * 2 + t) * 64
                             : ph == 1 ? wp1 + ((m * NKS + ks) * 2 + t) * 64
                                       : wp2 + ((m * 4 * NKS + ks) * 2 + t) * 64;
                ring[G % RING][m][t] = *p;
            }
    };
    static_for<0, PD>(fetch);
    __builtin_amdgcn_sched_barrier(0);
    const float us_proj = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, us_proj_v)));
    const float us_fc1 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, us_fc1_v)));
    const float us_fc2 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, us_fc2_v)));
    __builtin_amdgcn_sched_barrier(0);

    f32x4 acc[4];
    float x1v[4][4];                       // x1 of this wave's rows (= the rows of its fc2 output tiles)
    float mean = 0.f, rstd = 1.f;
    float gm = 0.f;                        // range guard (split.h)
    static_for<0, NG>([&](auto Gc) {
        constexpr int G = decltype(Gc)::value;
        constexpr int ph = G / NKS, ks = G % NKS;
        if constexpr (ks == 0) {
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // (scheduling fences: left alone, the machine scheduler sinks every load to just in front of its first use -- the whole
        //  point is that the loads of group G + PD are issued HERE)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (G + PD < NG) fetch(std::integral_constant<int, G + PD>{});
        __builtin_amdgcn_sched_barrier(0);
        sb8 bfr[2];
        if constexpr (ph == 0) {
            const wf4 x0 = aov[2 * ks], x1 = aov[2 * ks + 1];
            unsigned t[4][2];
            ws_split_pair_g<2>(x0[0], x0[1], t[0], gm);
            ws_split_pair_g<2>(x0[2], x0[3], t[1], gm);
            ws_split_pair_g<2>(x1[0], x1[1], t[2], gm);
            ws_split_pair_g<2>(x1[2], x1[3], t[3], gm);
#pragma unroll
            for (int q = 0; q < 2; ++q) bfr[q] = sb8{(int)t[0][q], (int)t[1][q], (int)t[2][q], (int)t[3][q]};
        } else {
            const unsigned char* base = ph == 1 ? X1S : HS;
            const int kr = (ks + rot) & (NKS - 1);
#pragma unroll
            for (int q = 0; q < 2; ++q) bfr[q] = *reinterpret_cast<const sb8*>(base + ((kr * 2 + q) * 64 + lane) * 16);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = sb_mma16<2>(ring[G % RING][m], bfr, acc[m]);
        // ---- end of a phase -----------------------------------------------------------------------------------------------------------
        if constexpr (ks == NKS - 1 && ph == 0) {
            // x1 = x + proj(ao) (uncovered pixels of a dilated block: shortcut only; DTransformer.py:299, 79-82)
            bool covered = true;
            if (a.mask_w > 0 && tok < a.HW) {
                const int y = tok / a.mask_w, x = tok - y * a.mask_w;
                const int rr = y + a.mask_pt, cc = x + a.mask_pl;
                covered = !((rr < 7 && (rr & 1)) || (cc < 7 && (cc & 1)));
            }
            const float us = us_proj;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int rt = 4 * wave + m;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = (covered ? acc[m][r] * us + PR[rt * 16 + g4 * 4 + r] : 0.f) + xr[m][r];
                    x1v[m][r] = v;
                    s1 += v;
                    s2 += v * v;
                }
                // a lane's four consecutive channels 16 rt + 4 g4 + r = half of B lane (col, (rt & 1) 2 + (g4 >> 1)) of k-step rt >> 1
                unsigned t[2][2];
                ws_split_pair_g<2>(x1v[m][0], x1v[m][1], t[0], gm);
                ws_split_pair_g<2>(x1v[m][2], x1v[m][3], t[1], gm);
                unsigned char* d = X1S + (((rt >> 1) * 2) * 64 + col + 16 * ((rt & 1) * 2 + (g4 >> 1))) * 16 + (g4 & 1) * 8;
#pragma unroll
                for (int q = 0; q < 2; ++q) *reinterpret_cast<uint2*>(d + q * 1024) = uint2{t[0][q], t[1][q]};
            }
            s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
            s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
            if (lane < 16) { ST[wave][lane][0] = s1; ST[wave][lane][1] = s2; }
            wb_sync();                     // (LDS only: the fragment stream stays in flight)
            const float u = (ST[0][col][0] + ST[1][col][0]) + (ST[2][col][0] + ST[3][col][0]);   // (fixed order: every workgroup
            const float v = (ST[0][col][1] + ST[1][col][1]) + (ST[2][col][1] + ST[3][col][1]);   //  of a tile gets the same sums)
            mean = u * (1.f / 256.f);
            rstd = __builtin_amdgcn_rsqf(fmaxf(v * (1.f / 256.f) - mean * mean, 0.f) + 1e-5f);
            WM_STAMP(1);
        }
        if constexpr (ks == NKS - 1 && ph == 1) {
            // hidden = GELU(fc1(LayerNorm2(x1))), rows 256 quarter + 64 wave + 16 m + 4 g4 + r
            const float us = us_fc1;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int hrt = 4 * wave + m;
                float hv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = hrt * 16 + g4 * 4 + r;
                    hv[r] = gelu_f(rstd * (acc[m][r] * us - mean * PR[512 + row]) + PR[256 + row]);
                }
                unsigned t[2][2];
                ws_split_pair_g<2>(hv[0], hv[1], t[0], gm);
                ws_split_pair_g<2>(hv[2], hv[3], t[1], gm);
                unsigned char* d = HS + (((hrt >> 1) * 2) * 64 + col + 16 * ((hrt & 1) * 2 + (g4 >> 1))) * 16 + (g4 & 1) * 8;
#pragma unroll
                for (int q = 0; q < 2; ++q) *reinterpret_cast<uint2*>(d + q * 1024) = uint2{t[0][q], t[1][q]};
            }
            wb_sync();
            WM_STAMP(2);
        }
    });
    WM_STAMP(3);
    sb_guard_flush(gm, a.ovf);

    // ---- this quarter's share of x2 (quarter 0 carries x1 and the bias) -> scratch, 16-byte sc1 stores ----------------------------
    {
        const float us = us_fc2;
        float* pq = a.part + ((((long)quarter * a.B + b) * a.ntile + tile) * 16 + 4 * wave) * 256 + lane * 4;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            f32x4 p;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                p[r] = acc[m][r] * us;
                if (quarter == 0) p[r] += x1v[m][r] + PR[768 + (4 * wave + m) * 16 + g4 * 4 + r];
            }
            st_sc1_f4(pq + m * 256, p);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave, before the counter add that signals for it
    WM_STAMP(4);
    __syncthreads();
    if (tid == 0) {
        int* cnt = a.count + b * a.ntile + tile;
        const int old = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_flag = old == 3;
        if (old == 3) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (next user: a later launch)
    }
    __syncthreads();
    WM_STAMP(5);
    if (!last_flag) return;
    // ---- last quarter to arrive: x2 = ((p0 + p1) + p2) + p3 (+ merged[t]) for the wave's four row tiles ----------------------------
    {
        f32x4 pv[4][4];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
#pragma unroll
            for (int m = 0; m < 4; ++m)
                pv[qq][m] = ld_sc1_f4(a.part + ((((long)qq * a.B + b) * a.ntile + tile) * 16 + 4 * wave + m) * 256 + lane * 4);
        float ad[4][4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const long fo = ((long)tile * 16 + 4 * wave + m) * 256 + col * 4 + g4;
#pragma unroll
            for (int r = 0; r < 4; ++r) ad[m][r] = a.addres ? a.addres[b * a.x_bs + fo + r * 64] : 0.f;
        }
        // the sc1 loads are inline asm: the compiler does not count them.  The wait names the loaded registers as operands, so no
        // use of them can be scheduled in front of it.
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(pv[qq][0]), "+v"(pv[qq][1]), "+v"(pv[qq][2]), "+v"(pv[qq][3])::"memory");
        float s1 = 0.f, s2 = 0.f, gmo = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int rt = 4 * wave + m;
            const long fo = ((long)tile * 16 + rt) * 256 + col * 4 + g4;
            float* op = a.out + b * a.x_bs + fo;
            float yv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float y = ((pv[0][m][r] + pv[1][m][r]) + pv[2][m][r]) + pv[3][m][r];
                y += ad[m][r];
                yv[r] = y;
                op[r * 64] = y;
                if (a.out_nchw && tok < a.HW) a.out_nchw[b * a.nchw_bs + (long)(rt * 16 + g4 * 4 + r) * a.HW + tok] = y;
                s1 += y;
                s2 += y * y;
            }
            if (a.out_spl) {
                // the same four channels as 8 bytes of each term of B lane (col, (rt & 1) 2 + (g4 >> 1)) of k-step rt >> 1
                unsigned t[2][2];
                ws_split_pair_g<2>(yv[0], yv[1], t[0], gmo);
                ws_split_pair_g<2>(yv[2], yv[3], t[1], gmo);
                unsigned char* d = reinterpret_cast<unsigned char*>(a.out_spl + b * a.spl_bs) +
                                   (spl16_frag(tile, 8, rt >> 1, 0) + col + 16 * ((rt & 1) * 2 + (g4 >> 1))) * 16 + (g4 & 1) * 8;
#pragma unroll
                for (int q = 0; q < 2; ++q) *reinterpret_cast<uint2*>(d + q * 1024) = uint2{t[0][q], t[1][q]};
            }
        }
        if (a.out_spl) {
            // LayerNorm statistics of x2 for its consumers: this wave's 64 channels, then the four waves in a fixed order
            sb_guard_flush(gmo, a.ovf);
            s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
            s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
            if (lane < 16) { ST[wave][lane][0] = s1; ST[wave][lane][1] = s2; }
            __syncthreads();                   // (the whole workgroup is the last arriver)
            if (tid < 16) {
                const float u = (ST[0][tid][0] + ST[1][tid][0]) + (ST[2][tid][0] + ST[3][tid][0]);
                const float v = (ST[0][tid][1] + ST[1][tid][1]) + (ST[2][tid][1] + ST[3][tid][1]);
                const float mean2 = u * (1.f / 256.f);
                const float rstd2 = __builtin_amdgcn_rsqf(fmaxf(v * (1.f / 256.f) - mean2 * mean2, 0.f) + 1e-5f);
                *reinterpret_cast<float2*>(a.out_stats + b * a.st_bs + ((long)tile * 16 + tid) * 2) = float2{mean2, rstd2};
            }
        }
    }
    WM_STAMP(6);
}

static int mlp_fused_launch(const MlpFusedArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(mlp_fused_kernel, dim3(a.ntile, 4, a.B), dim3(256), 0, s, a);
    BDE_HIP(hipGetLastError());
    return BDE_OK;
}

}  // namespace bde
