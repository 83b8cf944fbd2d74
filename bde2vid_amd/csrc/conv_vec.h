// Batched direct convolution, 16-byte staging variant of conv_mfma.h (same math, same weight packing,
// same epilogue), used whenever the input row length is a multiple of 4 floats.
//
// In-kernel stamps on the recurrent kernel showed what a stage costs besides its MFMAs: the
// global->register->LDS staging is paid per INSTRUCTION (a dword load or ds_write_b32 costs about as
// many issue cycles as a 16-byte one).  conv_mfma.h stages 40 halo dwords + 18 weight dwords per
// thread and stage; here the halo tile is laid out with a 4-float left pad so that every image row
// segment starts 16-byte aligned both in HBM and in LDS, and everything moves as float4:
// 8-10 loads + 8-10 ds_write_b128 per thread and stage instead of 58 + 58.
#pragma once
#include "conv_mfma.h"

namespace bde {

typedef float cvf4 __attribute__((ext_vector_type(4)));   // native vector: stays in registers
constexpr int CV_LP = 4;   // left pad (floats) of the LDS halo tile: keeps row data 16-byte aligned

struct ConvVecGeom {
    int R, IW;             // halo tile rows / row pitch (floats, multiple of 4)
};

// Largest halo tile over the blocks of a launch (host side; mirrors the kernel's geometry).
static inline long conv_vec_tile_elems(int KS, int STRIDE, int NT, int Win, int Ho, int Wo, int row_tiles) {
    const int PAD = KS / 2;
    const int BN = 4 * NT * 32;
    const int HW = Ho * Wo;
    auto iw_one = [&](int npx) { return ((npx - 1) * STRIDE + KS - PAD + CV_LP + 3) / 4 * 4; };
    const int iw_full = (Win + CV_LP + PAD + 3) / 4 * 4;
    if (row_tiles > 0) return (long)KS * iw_one(BN < Wo ? BN : Wo);
    long best = 0;
    for (int p0 = 0; p0 < HW; p0 += BN) {
        const int pl = (p0 + BN < HW ? p0 + BN : HW) - 1;
        const int yf = p0 / Wo, yl = pl / Wo;
        const int R = (yl - yf) * STRIDE + KS;
        const long e = (long)R * (yf == yl ? iw_one(pl - p0 + 1) : iw_full);
        if (e > best) best = e;
    }
    return best;
}

template <int KS, int STRIDE, int MT, int NT, int CK, int MAXI4>
__global__ __launch_bounds__(256) void conv_vec_kernel(const ConvArgs a) {
    constexpr int PAD = KS / 2;
    constexpr int TAPS = KS * KS;
    constexpr int PAIRS = CK / 2;
    constexpr int BN = 4 * NT * 32;
    constexpr int FRAG = TAPS * PAIRS * 64;           // floats of one co-tile's fragments for one chunk
    constexpr int ASZ = MT * FRAG;                    // weight region of a stage (floats, multiple of 4)
    constexpr int AK4 = (ASZ / 4 + 255) / 256;
    extern __shared__ __align__(16) float lds_all[];
    float* const lds = lds_all + ASZ;                 // halo tile [CK][R][IW]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bx, by, z;
    conv_block_coords(a.xcd_remap, bx, by, z);
    const int g = z / a.N, n = z - g * a.N;
    const int HW = a.Ho * a.Wo;
    int p0, p_end;
    if (a.row_tiles > 0) {
        const int yy = bx / a.row_tiles, xt = bx - yy * a.row_tiles;
        p0 = yy * a.Wo + xt * BN;
        p_end = min(p0 + BN, (yy + 1) * a.Wo);
    } else {
        p0 = bx * BN;
        p_end = min(p0 + BN, HW);
    }
    const int p_last = p_end - 1;
    const int y_first = p0 / a.Wo, y_last = p_last / a.Wo;
    const bool one_row = (y_first == y_last);
    const int x_first = p0 - y_first * a.Wo;
    const int iy0 = y_first * STRIDE - PAD;
    const int ix0 = one_row ? x_first * STRIDE - CV_LP : -CV_LP;      // multiple of 4 (x_first is a multiple of BN)
    const int R = (y_last - y_first) * STRIDE + KS;
    const int IW = (one_row ? (p_last - p0) * STRIDE + KS - PAD + CV_LP : a.Win + CV_LP + PAD) + 3 & ~3;
    const int PS = R * IW;
    const int IW4 = IW >> 2;

    int boff[NT], pix[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int p = p0 + (wave * NT + t) * 32 + (lane & 31);
        pix[t] = p;
        const int pc = min(p, p_last);
        const int y = pc / a.Wo, x = pc - y * a.Wo;
        boff[t] = (y - y_first) * STRIDE * IW + (x * STRIDE - PAD - ix0) + (lane >> 5) * PS;
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;

    const float* inb = a.in + g * a.in_gs + n * a.in_ns;
    const float* wg = a.wpk + g * a.w_gs;
    const long HsWs = (long)a.Hs * a.Ws;
    // staging slots of this thread: float4 (r, c4) of the halo tile, fixed for all stages
    unsigned voff[MAXI4];
    int lde[MAXI4];
    unsigned vmask = 0;
    {
        const int n4 = R * IW4;
        const float inv = 1.0f / (float)IW4;
#pragma unroll
        for (int it = 0; it < MAXI4; ++it) {
            const int i4 = tid + it * 256;
            const int r = (int)(((float)i4 + 0.5f) * inv);
            const int c4 = i4 - r * IW4;
            const int iy = iy0 + r, ix = ix0 + 4 * c4;
            const bool item = i4 < n4;
            // rows are multiples of 4 long and ix is a multiple of 4: a float4 is entirely in or out
            const bool in_img = iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
            lde[it] = item ? r * IW + 4 * c4 : -1;
            voff[it] = (item && in_img) ? (unsigned)(iy * a.Ws + ix) : 0u;
            if (item && in_img) vmask |= 1u << it;
        }
    }
    cvf4 sv[MAXI4][CK];
    cvf4 aw[AK4];
    auto stage_load = [&](int st) {
        const int c0 = st * CK;
#pragma unroll
        for (int k = 0; k < AK4; ++k) {
            const int i4 = min(tid + k * 256, ASZ / 4 - 1);
            const int m = i4 / (FRAG / 4), r4 = i4 - m * (FRAG / 4);
            aw[k] = reinterpret_cast<const cvf4*>(wg + ((long)(by * MT + m) * a.nchunks + st) * FRAG)[r4];
        }
#pragma unroll
        for (int j = 0; j < CK; ++j) {
            const float* cb = inb + (long)min(c0 + j, a.Cin - 1) * HsWs;      // uniform; padded channels masked at store
#pragma unroll
            for (int it = 0; it < MAXI4; ++it)
                sv[it][j] = ((vmask >> it) & 1u) ? *reinterpret_cast<const cvf4*>(cb + voff[it]) : cvf4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stage_store = [&](int st) {
        const int c0 = st * CK;
#pragma unroll
        for (int k = 0; k < AK4; ++k)
            if (tid + k * 256 < ASZ / 4) reinterpret_cast<cvf4*>(lds_all)[tid + k * 256] = aw[k];
#pragma unroll
        for (int it = 0; it < MAXI4; ++it)
            if (lde[it] >= 0) {
#pragma unroll
                for (int j = 0; j < CK; ++j)
                    *reinterpret_cast<cvf4*>(lds + lde[it] + j * PS) = (c0 + j < a.Cin) ? sv[it][j] : cvf4{0.f, 0.f, 0.f, 0.f};
            }
    };

    stage_load(0);
    for (int st = 0; st < a.nchunks; ++st) {
        __syncthreads();
        stage_store(st);
        __syncthreads();
        if (st + 1 < a.nchunks) stage_load(st + 1);
#pragma unroll 1
        for (int ky = 0; ky < KS; ++ky)
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                if constexpr (KS == 5) __builtin_amdgcn_iglp_opt(1);   // measured: -2..3 % on the 5x5 convs, +3..5 % on the 3x3 ones
                const int tap = ky * KS + kx;
#pragma unroll
                for (int pr = 0; pr < PAIRS; ++pr) {
                    float av[MT], bv[NT];
#pragma unroll
                    for (int m = 0; m < MT; ++m) av[m] = lds_all[(m * TAPS * PAIRS + tap * PAIRS + pr) * 64 + lane];
#pragma unroll
                    for (int t = 0; t < NT; ++t) bv[t] = lds[boff[t] + pr * 2 * PS + ky * IW + kx];
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[t], acc[m][t], 0, 0, 0);
                }
            }
    }

    float fin[MT][NT][16];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) fin[m][t][rr] = acc[m][t][rr];
    float mu[NT], rstd[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) mu[t] = rstd[t] = 0.f;
    generic_epilogue<MT, NT, 16>(a, fin, pix, 0, lane, g, n, HW, p_end, false, mu, rstd, by * MT);
}

constexpr int CV_MAXI4 = 3;    // max float4 staging slots per thread and channel (1, 2 or 3 are built)

template <int KS, int STRIDE, int MT, int NT, int MAXI4>
static int conv_vec_launch_i(const ConvArgs& a, int G, hipStream_t stream, long tile) {
    constexpr int CK = conv_ck(KS);
    const size_t lds = ((size_t)MT * KS * KS * (CK / 2) * 64 + (size_t)CK * tile) * sizeof(float);
    auto kern = conv_vec_kernel<KS, STRIDE, MT, NT, CK, MAXI4>;
    if (lds > 64 * 1024) {
        static unsigned char raised[BDE_MAX_DEVICES];
        BDE_HIP(raise_dynamic_lds(raised, (const void*)kern));
    }
    constexpr int BN = 4 * NT * 32;
    dim3 grid(a.row_tiles > 0 ? a.Ho * a.row_tiles : cdiv(a.Ho * a.Wo, BN), cdiv(a.Cout, MT * 32), G * a.N);
    if (grid.x == 0 || grid.y == 0 || grid.z == 0) return BDE_OK;
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, a);
    BDE_HIP(hipGetLastError());
    return BDE_OK;
}

template <int KS, int STRIDE, int MT, int NT>
static int conv_vec_launch_t(const ConvArgs& a, int G, hipStream_t stream) {
    const long tile = conv_vec_tile_elems(KS, STRIDE, NT, a.Win, a.Ho, a.Wo, a.row_tiles);
    const long items = (tile / 4 + 255) / 256;          // staging registers follow the tile actually used
    if (items <= 1) return conv_vec_launch_i<KS, STRIDE, MT, NT, 1>(a, G, stream, tile);
    if (items == 2) return conv_vec_launch_i<KS, STRIDE, MT, NT, 2>(a, G, stream, tile);
    return conv_vec_launch_i<KS, STRIDE, MT, NT, 3>(a, G, stream, tile);
}

// Same tiling choice as conv_launch_ks; returns BDE_ERR_UNSUPPORTED when no vector tiling fits
// (the caller then falls back to the dword kernel).
template <int KS, int STRIDE>
static int conv_vec_launch_ks(ConvArgs a, int G, hipStream_t stream, bool* launched) {
    *launched = false;
    if (a.Win % 4 != 0 || a.Ws % 4 != 0) return BDE_OK;
    constexpr int CK = conv_ck(KS);
    const int MT = a.Cout > 32 ? 2 : 1;
    double best = -1.0;
    int bnt = 1, brow = 0;
    for (int nt = 2; nt >= 1; --nt)
        for (int row = 1; row >= 0; --row) {
            if (tuning().conv_nt && nt != tuning().conv_nt) continue;
            const int BN = 4 * nt * 32;
            const int rt = row ? cdiv(a.Wo, BN) : 0;
            if (row && rt > 1 && (BN * STRIDE) % 4 != 0) continue;
            const long tile = conv_vec_tile_elems(KS, STRIDE, nt, a.Win, a.Ho, a.Wo, rt);
            const long lds = ((long)MT * KS * KS * (CK / 2) * 64 + (long)CK * tile) * 4;
            if ((tile / 4 + 255) / 256 > CV_MAXI4 || lds > 64 * 1024) continue;
            const double launched_px = row ? (double)a.Ho * rt * BN : (double)cdiv(a.Ho * a.Wo, BN) * BN;
            // two 32-pixel tiles per wave reuse each weight fragment twice, but with 64-row tiles they cost a third
            // of the residency (2 waves/SIMD instead of 3): measured ahead for the 3x3 gate convs, behind for every
            // 5x5 conv with more than 32 output channels (decoder 128->64 at 92x120: 850 vs 680 us)
            const double pref = (KS == 5 && MT == 2) ? (nt == 1 ? 1.0 : 0.9) : (nt == 2 ? 1.0 : 0.92);
            double score = (double)a.Ho * a.Wo / launched_px * pref;
            const long blocks = (long)(launched_px / BN) * cdiv(a.Cout, MT * 32) * (a.decide_groups ? a.decide_groups : G) * a.N;
            if (blocks < 512) score *= 0.5 + 0.5 * blocks / 512.0;
            if (score > best) { best = score; bnt = nt; brow = rt; }
        }
    if (best < 0) return BDE_OK;
    a.row_tiles = brow;
    *launched = true;
    if (MT == 2) return bnt == 2 ? conv_vec_launch_t<KS, STRIDE, 2, 2>(a, G, stream) : conv_vec_launch_t<KS, STRIDE, 2, 1>(a, G, stream);
    return bnt == 2 ? conv_vec_launch_t<KS, STRIDE, 1, 2>(a, G, stream) : conv_vec_launch_t<KS, STRIDE, 1, 1>(a, G, stream);
}

#ifdef BDE_CONV_TU
int conv_launch_best(int KS, int stride, const ConvArgs& a, int G, hipStream_t stream) {
    bool done = false;
    if (tuning().conv_vec) {
        int st = BDE_OK;
        if (KS == 5 && stride == 1) st = conv_vec_launch_ks<5, 1>(a, G, stream, &done);
        else if (KS == 5 && stride == 2) st = conv_vec_launch_ks<5, 2>(a, G, stream, &done);
        else if (KS == 3 && stride == 1) st = conv_vec_launch_ks<3, 1>(a, G, stream, &done);
        else if (KS == 3 && stride == 2) st = conv_vec_launch_ks<3, 2>(a, G, stream, &done);
        if (st != BDE_OK || done) return st;
    }
    return conv_launch_auto(KS, stride, a, G, stream);
}

#else
// (defined in conv_tu.hip: the convolution kernels are compiled in a translation unit of their own)
int conv_launch_best(int KS, int stride, const ConvArgs& a, int G, hipStream_t stream);
#endif

}  // namespace bde
