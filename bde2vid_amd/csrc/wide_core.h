// Level-2 attention chain (head_dim 16, wideblock.h): the window half of a block -- q | k | v of the query frame's window tokens,
// K | V of the REFINED neighbour frame's window tokens, softmax(q k^T + bias) v per (window, head) -- as one launch, second
// generation of attn_tok16_kernel<true, true> (DTransformer.py:165-207; V5.py:154-169).
//
// What changed, and why (attn_tok16_kernel measured 15.9 us per launch for ~1 us of matrix work):
//   * the head's q | k | v weight fragments (48 KB, two fp16 terms) come ONCE per workgroup, by LDS-DMA, all of them requested in
//     the first instructions of the kernel together with every other operand (token fragments, bias tile, the other frames' K | V
//     rows): one L2 round trip for everything instead of four register double-buffer rounds per wave, each wave re-fetching the
//     same 48 KB;
//   * the relative-position bias arrives as the score tiles' C operands (16-byte loads of a pre-packed table, as winblock.h)
//     instead of 40 strided dword loads per lane;
//   * K | V of the frame refined just before (buffer offset < 0, V5.py:166-169) are computed HERE from that frame's block output --
//     norm_kv and the kv projection are the query frame's own k | v rows (DTransformer.py:183-190), so the neighbour's window
//     tokens are one more token tile for fragments that are in LDS anyway -- which removes the K|V GEMM launch (26 us, 15 per
//     forward) that sat between two frames of the sequential chain and its [T][HW][depth 2C] buffer.
// Frames at offsets > 0 are still unrefined when they are needed: their K | V of all blocks come from one T-batched GEMM
// (tokgemm_sb_kernel) as before.  Same arithmetic as attn_tok16_kernel<true, true>: two-term q|k|v GEMM, fp32 scores / softmax / p v.
#pragma once
#include <hip/hip_runtime.h>
#include "wideblock.h"

namespace bde {

// ---- SPL16: a frame of the chain as the B fragments of the 16x16x32 fp16 MFMA, already split into two terms -----------------------------
//   [B][token tile 16][k-step 32 channels][term][64 lanes][8 x fp16]: lane (token & 15) + 16 g, element j = channel 32 ks + 8 g + j
//   (16 KB per token tile at 256 channels, the bytes of the fp32 tile), and beside it the LayerNorm statistics of every token,
//   [B][token][2] = (mean, rstd) over the channels.  Written ONCE by whoever produces a frame (the conversion below for the merged
//   frames, the last arriver of mlp_fused_kernel for a block's output); the attention core then loads its operand fragments as
//   they stand -- no split, no running sums, no zero masking in its GEMM loop (6.5 vector instructions per value and token set,
//   two thirds of that loop's cycles).  The fp32 FRAG16 twin stays: residuals and the final NCHW copy are fp32.
__host__ __device__ __forceinline__ long spl16_frag(long tile, int nks, int ks, int term) { return ((tile * nks + ks) * 2 + term) * 64; }   // in 16-byte units

// [N][C][HW] planes -> FRAG16 + SPL16 + statistics.  grid (token tiles, N), 256 threads: thread = (token, 16-channel group).
__global__ __launch_bounds__(256) void nchw_to_frag_spl_kernel(const float* __restrict__ in, float* __restrict__ frag,
                                                               unsigned short* __restrict__ spl, float* __restrict__ stats, int C, int HW,
                                                               int ntile, unsigned* ovf) {
    __shared__ float S1[16][17], S2[16][17];
    const int tile = blockIdx.x;
    const long n = blockIdx.y;
    const int t = threadIdx.x & 15, g = threadIdx.x >> 4;             // token of the tile, channel group (C = 256: 16 groups)
    const int tok = tile * 16 + t;
    const int ngc = C >> 4, nks = C >> 5;
    float v[16];
    float s1 = 0.f, s2 = 0.f, gm = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        v[k] = tok < HW ? in[(n * C + 16 * g + k) * HW + tok] : 0.f;
        s1 += v[k];
        s2 += v[k] * v[k];
    }
    // FRAG16: lane (t, g4), element j of group g = channel 16 g + 4 j + g4
    float* fo = frag + ((n * ntile + tile) * ngc + g) * 256;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4)
        *reinterpret_cast<float4*>(fo + (t + 16 * g4) * 4) = float4{v[g4], v[4 + g4], v[8 + g4], v[12 + g4]};
    // SPL16: channels 16 g + k = k-step g >> 1, B lane t + 16 ((g & 1) 2 + (k >> 3)), element k & 7
    uint4* so = reinterpret_cast<uint4*>(spl) + spl16_frag(n * ntile + tile, nks, g >> 1, 0) + t + 16 * ((g & 1) * 2);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        unsigned tt[4][2];
#pragma unroll
        for (int p = 0; p < 4; ++p) ws_split_pair_g<2>(v[8 * h + 2 * p], v[8 * h + 2 * p + 1], tt[p], gm);
#pragma unroll
        for (int q = 0; q < 2; ++q) so[q * 64 + 16 * h] = uint4{tt[0][q], tt[1][q], tt[2][q], tt[3][q]};
    }
    sb_guard_flush(gm, ovf);
    S1[g][t] = s1;
    S2[g][t] = s2;
    __syncthreads();
    if (g == 0) {
        float u1 = 0.f, u2 = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) { u1 += S1[k][t]; u2 += S2[k][t]; }          // fixed order
        const float mean = u1 / (float)C;
        const float rstd = __builtin_amdgcn_rsqf(fmaxf(u2 / (float)C - mean * mean, 0.f) + 1e-5f);
        *reinterpret_cast<float2*>(stats + ((n * ntile + tile) * 16 + t) * 2) = float2{mean, rstd};
    }
}
static int nchw_to_frag_spl(const float* in, float* frag, unsigned short* spl, float* stats, int N, int C, int HW, unsigned* ovf, hipStream_t s) {
    if (C != 256) return fail(BDE_ERR_UNSUPPORTED, "SPL16 conversion: C = %d", C);
    const int ntile = cdiv(HW, 16);
    hipLaunchKernelGGL(nchw_to_frag_spl_kernel, dim3(ntile, N), dim3(256), 0, s, in, frag, spl, stats, C, HW, ntile, ovf);
    BDE_HIP(hipGetLastError());
    return BDE_OK;
}

struct WideCoreArgs {
    const float* x;               // FRAG16 [B][ntile][C/16][256]: block input of the query frame
    const float* xp;              // FRAG16: the refined neighbour frame (slot p_slot), nullptr = none handled here
    long x_bs;
    int q_slot, p_slot;           // buffer slots (key ranges slot * 49 ..) of the two
    const float* kv[ATT_MAXD];    // other slots: token-major rows holding K at +k_off[d], V at +v_off[d]; nullptr = zero frame
    long kv_bs[ATT_MAXD];
    int kv_ld[ATT_MAXD], k_off[ATT_MAXD], v_off[ATT_MAXD];
    const float* kvpad;           // [2C] K | V of a zero token
    const float* biasW;           // [heads][4 query tiles][10 key tiles][64 lanes][4]: score-tile C operands, keys slot-major, log2(e) folded
    const unsigned short* wqkvS;  // q|k|v rows as two fp16 terms, k in FRAG16 group-pair order (TokGemmArgs::wS)
    const float* wqkv_unscale;
    const float *bqkv, *sqkv;     // [3C] folded biases / row sums of the LayerNorm-folded weights
    float* out;                   // FRAG16 [B][ntile][C/16][256]: attention output
    int D, C, heads, H, W, Hp, Wp, pt, pl, nWw, dilated, ntile;
    unsigned* ovf;                // range guard of the two-term format (split.h)
    unsigned long long* stamps;   // diagnostics only
    // SPL = true: the two frames as SPL16 + statistics (above) instead of fp32 FRAG16; wqkvS then in natural k order
    const unsigned short *xS, *xpS;
    const float *xSt, *xpSt;      // [B][ntile * 16][2]
    long spl_bs, st_bs;           // batch strides: 16-bit elements / floats
    const float* zeros;           // >= 16 bytes of zeros (operand of a padding token)
    // L2 warm-up for the launch that follows on the chain (mlp_fused_kernel): its proj / fc1 weight fragments were last read one
    // frame ago and have left the 4 MB L2s since.  Each workgroup touches its share of [pf_ptr, pf_ptr + pf_bytes) -- the workgroups of
    // one XCD (every eighth in dispatch order) cover the whole range between them -- with loads nobody waits for.
    const unsigned char* pf_ptr[2];
    long pf_bytes[2];
};
#define WC_STAMP(i)                                                                                                  \
    do {                                                                                                             \
        if (a.stamps && lane == 0 && blockIdx.z == 0 && blockIdx.y == 0 && blockIdx.x < 16)                           \
            a.stamps[(blockIdx.x * 4 + wave) * 8 + (i)] = __builtin_amdgcn_s_memtime();                               \
    } while (0)

#ifndef WC_XCD_REMAP
#define WC_XCD_REMAP 1
#endif
constexpr int WC_WL_BYTES = 3 * 8 * 2 * 1024;
constexpr int WC_LDS_BYTES = WC_WL_BYTES + (2 * 10 * 16 * 16 + 2 * 48) * 4;      // 68.5 KB: two workgroups per CU

// C = 256 (8 k-steps of 32), head_dim 16, D * 49 <= 160 keys.  grid (windows, heads, B), 256 threads = four query tiles.
template <bool PREV, bool SPL>
__global__ __launch_bounds__(256, 2) void wide_core_kernel(const WideCoreArgs a) {
    constexpr int HD = 16, NT = 10, NKS = 8;
    extern __shared__ __align__(16) unsigned char wc_lds[];
    unsigned char* WL = wc_lds;                                       // weight fragments [q | k | v][k-step][term][64 lanes][16 B]
    float* KL = reinterpret_cast<float*>(wc_lds + WC_WL_BYTES);       // [NT][HD][16 keys]
    float* VL = KL + NT * HD * 16;                                    // [NT * 16 keys][HD]
    float* PR = VL + NT * 16 * HD;                                    // folded bias | row sums of the head's q, k, v rows
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g4 = lane >> 4, col = lane & 15;
    // Workgroup -> (window, head, batch), head-major inside each XCD.  The dispatcher deals consecutive workgroups round-robin to the 8
    // XCDs: in grid order every XCD sees every head and pulls all 786 KB of a block's q|k|v fragments into its L2 in every launch --
    // the level's 19 MB of weights cycle through the 4 MB L2s once per frame, so that is a fill from the Infinity Cache each time.
    // With a contiguous range of the (batch, head, window) order per XCD, an XCD holds two heads: an eighth of the fragments.
    int win, head, b;
    {
        const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
        const unsigned lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
        unsigned L = lin;
        if (WC_XCD_REMAP) {
            const unsigned xcd = lin & 7u, q = total >> 3, rem = total & 7u;
            L = xcd * q + min(xcd, rem) + (lin >> 3);
        }
        // (L < 2^20, divisors < 2^10: floor((L + 0.5) / d) through one v_rcp_f32 is exact -- an integer division by a run-time
        //  divisor is ~40 instructions, three of them at the head of every launch of the sequential chain)
        const unsigned q1 = (unsigned)(((float)L + 0.5f) * __builtin_amdgcn_rcpf((float)gx));
        win = (int)(L - q1 * gx);
        const unsigned q2 = (unsigned)(((float)q1 + 0.5f) * __builtin_amdgcn_rcpf((float)gy));
        head = (int)(q1 - q2 * gy);
        b = (int)q2;
    }
    const int wi = (int)(((float)win + 0.5f) * __builtin_amdgcn_rcpf((float)a.nWw)), wj = win - wi * a.nWw;   // (exact, as above)
    const int step = a.dilated ? 2 : 1;
    const int c0 = head * HD;
    const int nkey = a.D * ATT_TOK;
    WC_STAMP(0);
    auto token_pixel = [&](int tok) {
        const int ta = tok / ATT_WS, tb = tok - ta * ATT_WS;
        const int rp = wi * ATT_WS + ta * step, cp = wj * ATT_WS + tb * step;
        const int ry = rp - a.pt, rx = cp - a.pl;
        return (rp < a.Hp && cp < a.Wp && ry >= 0 && ry < a.H && rx >= 0 && rx < a.W) ? ry * a.W + rx : -1;
    };
    const int qi = wave * 16 + col;
    const int qpix = qi < ATT_TOK ? token_pixel(qi) : -1;
    const int ngk = a.C >> 4;

    // ---- every operand of the launch is requested here ---------------------------------------------------------------------------
    const long xo = ((long)(max(qpix, 0) >> 4) * ngk) * 64 + (max(qpix, 0) & 15) + 16 * g4;
    const wf4* xw = reinterpret_cast<const wf4*>(a.x + b * a.x_bs) + xo;
    // (fp32 token fragments: a rolling window of LA k-steps per token set is in flight -- all sixteen k-steps of both sets from
    //  the start are 128 registers, which with two workgroups per CU is more than a wave has)
    constexpr int LA = 4;
    const wf4* xpw = (PREV && !SPL) ? reinterpret_cast<const wf4*>(a.xp + b * a.x_bs) + xo : xw;
    wf4 xq[SPL ? 1 : 2 * NKS], xpv[(PREV && !SPL) ? 2 * NKS : 1];
    // SPL16 operands: the lane's 16 bytes of (k-step, term) of its token, 16 per token set, all requested here (a padding token
    // reads zeros: LayerNorm(0) = beta, i.e. the folded bias alone)
    sb8 bqS[SPL ? NKS : 1][2], bpS[(SPL && PREV) ? NKS : 1][2];
    float2 stq = {0.f, 1.f}, stp = {0.f, 1.f};
    if constexpr (SPL) {
        const bool live = qpix >= 0;
        const long so = spl16_frag(max(qpix, 0) >> 4, NKS, 0, 0) + (max(qpix, 0) & 15) + 16 * g4;
        const int stride = live ? 64 : 0;
        const sb8* sq = live ? reinterpret_cast<const sb8*>(a.xS + b * a.spl_bs) + so : reinterpret_cast<const sb8*>(a.zeros);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int t = 0; t < 2; ++t) bqS[ks][t] = sq[(ks * 2 + t) * stride];
        // (pointer select + unconditional load: a load under a per-lane branch is joined by s_waitcnt vmcnt(0) -- here a full round
        //  trip in front of the second half of the prologue's requests; a padding token reads (0, 0) and takes rstd = 1 where it is used)
        stq = *reinterpret_cast<const float2*>(live ? a.xSt + b * a.st_bs + (long)qpix * 2 : reinterpret_cast<const float*>(a.zeros));
        if constexpr (PREV) {
            const sb8* sp = live ? reinterpret_cast<const sb8*>(a.xpS + b * a.spl_bs) + so : reinterpret_cast<const sb8*>(a.zeros);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                for (int t = 0; t < 2; ++t) bpS[ks][t] = sp[(ks * 2 + t) * stride];
            stp = *reinterpret_cast<const float2*>(live ? a.xpSt + b * a.st_bs + (long)qpix * 2 : reinterpret_cast<const float*>(a.zeros));
        }
    } else {
#pragma unroll
        for (int kg = 0; kg < 2 * LA; ++kg) {
            xq[kg] = xw[kg * 64];
            if constexpr (PREV) xpv[kg] = xpw[kg * 64];
        }
    }
    {
        // weight fragments: 48 blocks of 1 KiB, twelve per wave; fragment (r, ks, t) of the packed rows: row tile r * ngk + head
        const sb8* wsrc = reinterpret_cast<const sb8*>(a.wqkvS) + lane;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int f = wave * 12 + i;
            const int r = f / (2 * NKS), rem = f - r * 2 * NKS;      // rem = ks * 2 + t
            const sb8* src = wsrc + (((long)(r * ngk + head) * NKS * 2) + rem) * 64;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(WL + f * 1024), 16, 0, 0);
        }
    }
    // K | V rows of the keys whose rows are not computed here -- the slots that are neither the query frame's nor the refined
    // neighbour's, and the padding keys beyond D * 49 (rows of a zero token, as zero frames and padding pixels): token-major
    // gathers, thread = (key, 4-channel piece).  GK keys: 62 with the neighbour in the core (one pass), 111 without (two).
    constexpr int GK = NT * 16 - (PREV ? 2 : 1) * ATT_TOK, GIT = (GK * 4 + 255) / 256;
    const int s_lo = PREV ? min(a.q_slot, a.p_slot) : a.q_slot, s_hi = PREV ? max(a.q_slot, a.p_slot) : a.q_slot;
    auto gather_key = [&](int item) {                                 // item -> key index u, skipping the slots computed here
        int u = item >> 2;
        if (u >= s_lo * ATT_TOK) u += ATT_TOK;
        if (PREV && u >= s_hi * ATT_TOK) u += ATT_TOK;
        return u;
    };
    // The per-slot arguments as opaque scalars: indexed by the per-lane d -- or picked by selects the compiler folds back into a
    // load from a selected address -- they are vector loads from the kernel-argument segment, dependent round trips in front of
    // the gathers (ISA of round 4).  D * 49 <= 160: three slots.
    const float* kv_s[3] = {a.kv[0], a.kv[1], a.kv[2]};
    long bs_s[3] = {a.kv_bs[0], a.kv_bs[1], a.kv_bs[2]};
    int ld_s[3] = {a.kv_ld[0], a.kv_ld[1], a.kv_ld[2]}, ko_s[3] = {a.k_off[0], a.k_off[1], a.k_off[2]}, vo_s[3] = {a.v_off[0], a.v_off[1], a.v_off[2]};
#pragma unroll
    for (int i = 0; i < 3; ++i) asm volatile("" : "+s"(kv_s[i]), "+s"(bs_s[i]), "+s"(ld_s[i]), "+s"(ko_s[i]), "+s"(vo_s[i]));
    wf4 kk[GIT], vv[GIT];
#pragma unroll
    for (int t = 0; t < GIT; ++t) {
        const int item = min(tid + t * 256, GK * 4 - 1);
        const int u = gather_key(item), cg = item & 3;
        const int d = min(u / ATT_TOK, a.D - 1), tok = u - (u / ATT_TOK) * ATT_TOK;
        const int pix = u < nkey ? token_pixel(tok) : -1;
        auto pick = [&](const auto& arr) { return d == 0 ? arr[0] : (d == 1 ? arr[1] : arr[2]); };
        const float* kp = u < nkey ? pick(kv_s) : nullptr;
        const bool use = pix >= 0 && kp != nullptr;
        const long rowo = b * pick(bs_s) + (long)pix * pick(ld_s);
        const float* ksrc = use ? kp + rowo + pick(ko_s) + c0 + cg * 4 : a.kvpad + c0 + cg * 4;
        const float* vsrc = use ? kp + rowo + pick(vo_s) + c0 + cg * 4 : a.kvpad + a.C + c0 + cg * 4;
        typedef const __attribute__((address_space(1))) wf4 gwf4;   // (explicitly global: pointers out of an opaque asm operand are generic)
        kk[t] = *(gwf4*)reinterpret_cast<const wf4*>(ksrc);
        vv[t] = *(gwf4*)reinterpret_cast<const wf4*>(vsrc);
    }
    // bias and row sums of rows c0 .. c0 + 15 of q, k, v: 24 float4, loaded by every thread (index clamped) so that the load is
    // not under a per-lane branch (whose join is a vmcnt(0) of its own), stored to LDS behind the one wait
    // (the accumulator scale of the q|k|v weights rides in the same round trip: read where it is used, behind the GEMM phase, it
    //  was a dependent kernarg load + global load + vmcnt(0) of its own in every launch -- ISA of round 4)
    const float us_v = a.wqkv_unscale[0];
    const int pr_t = min(tid, 23), pr_which = pr_t / 12, pr_i = pr_t - pr_which * 12;
    wf4 prv = *reinterpret_cast<const wf4*>((pr_which ? a.sqkv : a.bqkv) + (pr_i >> 2) * a.C + c0 + (pr_i & 3) * 4);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(prv)::"memory");        // one round trip for all of the above
    if (tid < 24) *reinterpret_cast<wf4*>(PR + pr_which * 48 + (pr_i >> 2) * 16 + (pr_i & 3) * 4) = prv;
    __syncthreads();
    const float us_qkv = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, us_v)));
    WC_STAMP(1);
    f32x4 pf_sink = {0.f, 0.f, 0.f, 0.f};     // destination of the warm-up loads: stays reserved until they have landed (end of the kernel)
    if (a.pf_ptr[0] != nullptr && wave == 3) {
        // (wave 3 holds one query of the window: the idlest of the four)
        const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const unsigned per_xcd = (gridDim.x * gridDim.y * gridDim.z + 7u) >> 3;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (a.pf_ptr[r] == nullptr) continue;
            const long share = ((a.pf_bytes[r] / per_xcd) + 1023) & ~1023L;        // whole 1-KiB wave loads
            const long beg = (long)(lin >> 3) * share;
            for (long o = beg + lane * 16; o < min(beg + share, a.pf_bytes[r]); o += 1024)
                asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(pf_sink) : "v"(a.pf_ptr[r] + o) : "memory");
        }
    }

    // ---- q | k | v of this wave's token tile (query frame) and k | v of the same window tokens of the refined neighbour -------------
    f32x4 aq = {0.f, 0.f, 0.f, 0.f}, ak = aq, av = aq, akp = aq, avp = aq;
    float s1 = 0.f, s2 = 0.f, p1 = 0.f, p2 = 0.f, gm = 0.f;
    auto split_tile = [&](wf4 x0, wf4 x1, float& u1, float& u2, sb8 (&bfr)[2]) {
        if (qpix < 0) x0 = x1 = wf4{0.f, 0.f, 0.f, 0.f};               // a zero token: LayerNorm(0) = beta, i.e. the folded bias alone
        u1 += ((x0[0] + x0[1]) + (x0[2] + x0[3])) + ((x1[0] + x1[1]) + (x1[2] + x1[3]));
        u2 += ((x0[0] * x0[0] + x0[1] * x0[1]) + (x0[2] * x0[2] + x0[3] * x0[3])) +
              ((x1[0] * x1[0] + x1[1] * x1[1]) + (x1[2] * x1[2] + x1[3] * x1[3]));
        unsigned t[4][2];
        ws_split_pair_g<2>(x0[0], x0[1], t[0], gm);
        ws_split_pair_g<2>(x0[2], x0[3], t[1], gm);
        ws_split_pair_g<2>(x1[0], x1[1], t[2], gm);
        ws_split_pair_g<2>(x1[2], x1[3], t[3], gm);
#pragma unroll
        for (int q = 0; q < 2; ++q) bfr[q] = sb8{(int)t[0][q], (int)t[1][q], (int)t[2][q], (int)t[3][q]};
        // (pins the running sums here: left free, the compiler sinks both chains of adds below the loop -- nothing needs them
        //  earlier -- and keeps the eight fp32 values of every k-step alive for it: 64 registers per token set)
        asm volatile("" : "+v"(u1), "+v"(u2), "+v"(gm));
    };
    // The relative-position bias of this wave's score tiles (their C operands as 16-byte loads of the pre-packed table) is
    // requested half-way through the contraction: by then half of the token fragments' registers are free again, and the scores
    // are still a microsecond away.  Held from the start it costs 40 registers through the GEMM phase.
    f32x4 sc[NT];
    const f32x4* biasp = reinterpret_cast<const f32x4*>(a.biasW) + ((long)(head * 4 + wave) * NT) * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        __builtin_amdgcn_sched_barrier(0);                             // (keeps the loads and LDS reads of later k-steps out of this one)
        if constexpr (!SPL) {
            if (ks + LA < NKS) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    xq[2 * (ks + LA) + h] = xw[(2 * (ks + LA) + h) * 64];
                    if constexpr (PREV) xpv[2 * (ks + LA) + h] = xpw[(2 * (ks + LA) + h) * 64];
                }
            }
        }
        if (ks == NKS / 2) {
#pragma unroll
            for (int j = 0; j < NT; ++j) sc[j] = biasp[j * 64];
        }
        sb8 fq[2], fk[2], fv[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            fq[t] = *reinterpret_cast<const sb8*>(WL + ((0 * NKS + ks) * 2 + t) * 1024 + lane * 16);
            fk[t] = *reinterpret_cast<const sb8*>(WL + ((1 * NKS + ks) * 2 + t) * 1024 + lane * 16);
            fv[t] = *reinterpret_cast<const sb8*>(WL + ((2 * NKS + ks) * 2 + t) * 1024 + lane * 16);
        }
        if constexpr (SPL) {
            aq = sb_mma16<2>(fq, bqS[ks], aq);
            ak = sb_mma16<2>(fk, bqS[ks], ak);
            av = sb_mma16<2>(fv, bqS[ks], av);
            if constexpr (PREV) {
                akp = sb_mma16<2>(fk, bpS[ks], akp);
                avp = sb_mma16<2>(fv, bpS[ks], avp);
            }
        } else {
            sb8 bq[2];
            split_tile(xq[2 * ks], xq[2 * ks + 1], s1, s2, bq);
            aq = sb_mma16<2>(fq, bq, aq);
            ak = sb_mma16<2>(fk, bq, ak);
            av = sb_mma16<2>(fv, bq, av);
            if constexpr (PREV) {
                sb8 bp[2];
                split_tile(xpv[2 * ks], xpv[2 * ks + 1], p1, p2, bp);
                akp = sb_mma16<2>(fk, bp, akp);
                avp = sb_mma16<2>(fv, bp, avp);
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!SPL) sb_guard_flush(gm, a.ovf);       // (SPL16 frames were checked when they were written)
    WC_STAMP(2);
    wf4 qv;
    {
        const float us = us_qkv;
        auto stats = [&](float u1, float u2, float& mean, float& rstd) {
            u1 += __shfl_xor(u1, 16); u2 += __shfl_xor(u2, 16);
            u1 += __shfl_xor(u1, 32); u2 += __shfl_xor(u2, 32);
            mean = u1 / (float)a.C;
            rstd = __builtin_amdgcn_rsqf(fmaxf(u2 / (float)a.C - mean * mean, 0.f) + 1e-5f);
        };
        float mean, rstd;
        if constexpr (SPL) { mean = stq.x; rstd = qpix >= 0 ? stq.y : 1.f; }
        else stats(s1, s2, mean, rstd);
        float kq[4], vq[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * g4 + r;
            qv[r] = rstd * (aq[r] * us - mean * PR[48 + i]) + PR[i];
            kq[r] = rstd * (ak[r] * us - mean * PR[48 + 16 + i]) + PR[16 + i];
            vq[r] = rstd * (av[r] * us - mean * PR[48 + 32 + i]) + PR[32 + i];
        }
        if (qi < ATT_TOK) {
            // key index of this token in the slot-major key order; K row (k-step e = r, lanes g4) and the V row of the core below
            const int u = a.q_slot * ATT_TOK + qi;
#pragma unroll
            for (int r = 0; r < 4; ++r) KL[((u >> 4) * HD + r * 4 + g4) * 16 + (u & 15)] = kq[r];
            *reinterpret_cast<wf4*>(VL + u * HD + g4 * 4) = wf4{vq[0], vq[1], vq[2], vq[3]};
        }
        if constexpr (PREV) {
            float meanp, rstdp;
            if constexpr (SPL) { meanp = stp.x; rstdp = qpix >= 0 ? stp.y : 1.f; }
            else stats(p1, p2, meanp, rstdp);
            if (qi < ATT_TOK) {
                const int u = a.p_slot * ATT_TOK + qi;
                float vp[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 4 * g4 + r;
                    KL[((u >> 4) * HD + r * 4 + g4) * 16 + (u & 15)] = rstdp * (akp[r] * us - meanp * PR[48 + 16 + i]) + PR[16 + i];
                    vp[r] = rstdp * (avp[r] * us - meanp * PR[48 + 32 + i]) + PR[32 + i];
                }
                *reinterpret_cast<wf4*>(VL + u * HD + g4 * 4) = wf4{vp[0], vp[1], vp[2], vp[3]};
            }
        }
    }
    if (qpix < 0) qv = wf4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < GIT; ++t) {
        const int item = tid + t * 256;
        if (item >= GK * 4) continue;
        const int u = gather_key(item), cg = item & 3;
        // channel cg*4 + e of the head is contracted at k-step e by the lanes g4 = cg: LDS row e*4 + cg
#pragma unroll
        for (int e = 0; e < 4; ++e) KL[((u >> 4) * HD + e * 4 + cg) * 16 + (u & 15)] = kk[t][e];
        *reinterpret_cast<wf4*>(VL + u * HD + cg * 4) = vv[t];
    }
    __syncthreads();
    WC_STAMP(3);
    // ---- scores^T = k q^T + bias, softmax over the keys, out^T = v^T p^T (attn_mfma.h) ---------------------------------------------
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
    WC_STAMP(4);
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
    WC_STAMP(5);
    if (qpix >= 0) {
        const float inv = 1.f / l;
        // rows of acc = channels c0 + 4 g4 + r of query pixel qpix: FRAG16 group `head`, lane (pixel column + 16 r), element g4
        float* op = a.out + b * a.x_bs + ((long)(qpix >> 4) * ngk + head) * 256 + (qpix & 15) * 4 + g4;
#pragma unroll
        for (int r = 0; r < 4; ++r) op[r * 64] = acc[r] * inv;
    }
    WC_STAMP(6);
    // (an asynchronous load may not outlive the reservation of its destination register)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(pf_sink)::"memory");
}

static int wide_core_launch(const WideCoreArgs& a, int B, hipStream_t s) {
    if (a.C != 256 || a.heads * 16 != a.C || a.D * ATT_TOK > 160)
        return fail(BDE_ERR_UNSUPPORTED, "wide attention core: C = %d, %d heads, D = %d", a.C, a.heads, a.D);
    const int nW = (a.Hp / ATT_WS) * (a.Wp / ATT_WS);
    static unsigned char raised[4][BDE_MAX_DEVICES];
    const bool spl = a.xS != nullptr;
    const dim3 grid(nW, a.heads, B);
#define WC_LAUNCH(P, S, i)                                                                                          \
    do {                                                                                                            \
        BDE_HIP(raise_dynamic_lds(raised[i], (const void*)wide_core_kernel<P, S>, WC_LDS_BYTES));                    \
        hipLaunchKernelGGL((wide_core_kernel<P, S>), grid, dim3(256), WC_LDS_BYTES, s, a);                          \
    } while (0)
    if (a.xp && spl) WC_LAUNCH(true, true, 0);
    else if (a.xp) WC_LAUNCH(true, false, 1);
    else if (spl) WC_LAUNCH(false, true, 2);
    else WC_LAUNCH(false, false, 3);
#undef WC_LAUNCH
    BDE_HIP(hipGetLastError());
    return BDE_OK;
}

}  // namespace bde
