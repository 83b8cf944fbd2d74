// Batched direct convolution with fp32-equivalent arithmetic on the 16-bit matrix cores (split operands, split.h).
//
// The convolutions of the path are compute-bound on the fp32 matrix rate (64 FLOP/clk/SIMD), and that rate is 1/16 of the
// bf16 / fp16 one.  split.h writes an fp32 operand as two fp16 terms (three MFMAs per fp32 block: 5.3x the fp32 matrix rate) or
// as three bf16 terms (six MFMAs: 2.67x; fp32's exponent range), at the accuracy of an fp32 contraction either way
// (tools/ubench/mfma_f16_probe.hip, tools/bf16_split_error.py; every parity test of the fp32 kernels holds at its tolerance).
// dtype of the path stays "f32": inputs, outputs and accumulation are fp32.  TERMS is a template parameter of every kernel here.
//
// Operands:
//   activations  SB16 [N][ceil(C/16)][H][W][TERMS][16 channels] (32 TERMS bytes per pixel and 16-channel chunk), written by
//                split_bf16_kernel from fp32 NCHW planes or directly by the producing kernel's epilogue;
//   weights      split once at pack time, in A-fragment order of the 32x32x16 MFMA:
//                [co tile 32][chunk][tap][term][64 lanes][8]: lane l = W[32 tile + (l & 31)][16 chunk + 8 (l >> 5) + j][tap].
// Workgroup = WM x WN waves, a wave = MT x NT tiles of 32 output channels x 32 pixels; per 16-channel chunk the halo tile of
// the workgroup's BN pixels is staged by LDS-DMA (global_load_lds_dwordx4: no staging registers, which is what lets the
// weight fragments be double-buffered and a third workgroup fit a CU; pixel pitch 32 TERMS + 16 bytes: conflict-free 16-byte
// fragment reads), the workgroups of a CU covering each other's staging; weight fragments go L2 -> registers two (3x3) or four (5x5)
// taps ahead.
// Epilogue = generic_epilogue of conv_mfma.h (bias, ReLU / ReLU6, residuals; same D layout as the fp32 32x32x2 MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include "conv_mfma.h"

namespace bde {

// (formats, splits and the MFMA wrappers: split.h)

#ifdef BDE_SB_TU
// fp32 [N][C][H][W] -> SB16 [N][C16][H][W][terms][16].  grid (ceil(HW / 128), C16, N), 256 threads: thread = (pixel, half of
// the chunk), a wave = 64 consecutive pixels of one half: 8 plane loads of 256 contiguous bytes each, one 16-byte store per term.
template <int TERMS>
__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ in, unsigned short* __restrict__ out, int C, long HW,
                                                         unsigned* ovf) {
    const int half = threadIdx.x >> 7, pl = threadIdx.x & 127;
    const long p0 = (long)blockIdx.x * 128;
    const long p = min(p0 + pl, HW - 1);
    const int c16 = blockIdx.y;
    const long n = blockIdx.z;
    const int C16 = gridDim.y;
    unsigned short t[8][TERMS];
    float gm = 0.f;                                        // range guard of the two-term format (split.h)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = c16 * 16 + half * 8 + j;
        const float x = c < C ? in[(n * C + c) * HW + p] : 0.f;
        if (TERMS == 2) gm = sb_guard_max(gm, x);
        sb_split_dev<TERMS>(x, t[j]);
    }
    if (TERMS == 2) sb_guard_flush(gm, ovf);
    // through LDS, so that a wave stores 1 KiB of consecutive bytes per instruction (stored by their producers, a wave's 16-byte
    // pieces are 64 partial writes into 64 different lines: api_elementwise.h, upsample2x_sum_split_quad_kernel)
    constexpr int PB = 32 * TERMS, PITCH = PB + 16;
    __shared__ __align__(16) unsigned char stg[128 * (32 * TERMS + 16)];
#pragma unroll
    for (int k = 0; k < TERMS; ++k) {
        uint4 v;
        v.x = t[0][k] | ((unsigned)t[1][k] << 16);
        v.y = t[2][k] | ((unsigned)t[3][k] << 16);
        v.z = t[4][k] | ((unsigned)t[5][k] << 16);
        v.w = t[6][k] | ((unsigned)t[7][k] << 16);
        *reinterpret_cast<uint4*>(stg + pl * PITCH + k * 32 + half * 16) = v;
    }
    __syncthreads();
    const int npix = (int)min(128L, HW - p0);
    unsigned char* ob = reinterpret_cast<unsigned char*>(out + (((n * C16 + c16) * HW + p0) * TERMS) * 16);
    constexpr int PPP = 2 * TERMS;                         // 16-byte pieces per pixel
    for (int it = threadIdx.x; it < npix * PPP; it += 256) {
        const int q = it % PPP, px = it / PPP;
        *reinterpret_cast<uint4*>(ob + (long)px * PB + q * 16) = *reinterpret_cast<const uint4*>(stg + px * PITCH + q * 16);
    }
}
int split_bf16(const float* in, void* out, long N, int C, long HW, int terms, unsigned* ovf, hipStream_t s) {
    const int C16 = cdiv(C, 16);
    const dim3 grid((unsigned)cdivl(HW, 128), C16, (unsigned)N);
    if (terms == 2) hipLaunchKernelGGL(split_bf16_kernel<2>, grid, dim3(256), 0, s, in, (unsigned short*)out, C, HW, ovf);
    else hipLaunchKernelGGL(split_bf16_kernel<3>, grid, dim3(256), 0, s, in, (unsigned short*)out, C, HW, ovf);
    BDE_HIP(hipGetLastError());
    return BDE_OK;
}
// fp32 voxel grids [N][C <= 5][H][W] -> the head3 image (api_pack.h, pack_head3): SB16 with ONE chunk and W + 3 columns,
// [N][H][W + 3][terms][16]: channel k = 5 jj + b of pixel (y, e) = in[n][b][y][e - 2 + jj] (0 outside the row; k = 15: 0) -- three
// columns of the grid per chunk, so that a 5-wide kernel row is two MFMA taps.  grid (ceil(H (W + 3) / 128), 1, N), 256 threads =
// 128 pixels x 2 halves of the chunk; pieces staged through LDS as in split_bf16_kernel.
template <int TERMS>
__global__ __launch_bounds__(256) void split_head3_kernel(const float* __restrict__ in, unsigned short* __restrict__ out, int C, int H, int W,
                                                          unsigned* ovf) {
    const int WE = W + 3;
    const long HWE = (long)H * WE;
    const int half = threadIdx.x >> 7, pl = threadIdx.x & 127;
    const long p0 = (long)blockIdx.x * 128;
    const long p = min(p0 + pl, HWE - 1);
    const long n = blockIdx.z;
    const int y = (int)(p / WE), e = (int)(p - (long)y * WE);
    unsigned short t[8][TERMS];
    float gm = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = half * 8 + j, jj = k / 5, b = k - jj * 5, x = e - 2 + jj;
        const bool ok = k < 15 && b < C && x >= 0 && x < W;
        const float v = ok ? in[((n * C + b) * H + y) * (long)W + x] : 0.f;
        if (TERMS == 2) gm = sb_guard_max(gm, v);
        sb_split_dev<TERMS>(v, t[j]);
    }
    if (TERMS == 2) sb_guard_flush(gm, ovf);
    constexpr int PB = 32 * TERMS, PITCH = PB + 16, PPP = 2 * TERMS;
    __shared__ __align__(16) unsigned char stg[128 * (32 * TERMS + 16)];
#pragma unroll
    for (int k = 0; k < TERMS; ++k) {
        uint4 v;
        v.x = t[0][k] | ((unsigned)t[1][k] << 16);
        v.y = t[2][k] | ((unsigned)t[3][k] << 16);
        v.z = t[4][k] | ((unsigned)t[5][k] << 16);
        v.w = t[6][k] | ((unsigned)t[7][k] << 16);
        *reinterpret_cast<uint4*>(stg + pl * PITCH + k * 32 + half * 16) = v;
    }
    __syncthreads();
    const int npix = (int)min(128L, HWE - p0);
    unsigned char* ob = reinterpret_cast<unsigned char*>(out + ((n * HWE + p0) * TERMS) * 16);
    for (int it = threadIdx.x; it < npix * PPP; it += 256) {
        const int q = it % PPP, px = it / PPP;
        *reinterpret_cast<uint4*>(ob + (long)px * PB + q * 16) = *reinterpret_cast<const uint4*>(stg + px * PITCH + q * 16);
    }
}
int split_head3(const float* in, void* out, long N, int C, int H, int W, int terms, unsigned* ovf, hipStream_t s) {
    const dim3 grid((unsigned)cdivl((long)H * (W + 3), 128), 1, (unsigned)N);
    if (terms == 2) hipLaunchKernelGGL(split_head3_kernel<2>, grid, dim3(256), 0, s, in, (unsigned short*)out, C, H, W, ovf);
    else hipLaunchKernelGGL(split_head3_kernel<3>, grid, dim3(256), 0, s, in, (unsigned short*)out, C, H, W, ovf);
    BDE_HIP(hipGetLastError());
    return BDE_OK;
}
#else
int split_bf16(const float* in, void* out, long N, int C, long HW, int terms, unsigned* ovf, hipStream_t s);   // sb_tu.hip
int split_head3(const float* in, void* out, long N, int C, int H, int W, int terms, unsigned* ovf, hipStream_t s);
#endif
// ConvLSTM pointwise tail for the split-bf16 recurrent step (submodules.py:320-332): gates = gx (x-part incl. bias) + gh
// (h-part, from conv_sb_kernel; nullptr at the first step, h = 0), chunk order i, f, o, g; c = sigma(f) c + sigma(i) tanh(g);
// h = sigma(o) tanh(c) -- |h| < 1, so this producer of an SB16 image needs no range guard (split.h).  Writes c in place, h as fp32 planes (the level's hidden sequence) and as SB16 (the next step's
// convolution input).  grid (ceil(HW / 128), C16, 2 * B), thread = (pixel, half of a 16-channel chunk).
struct LstmPointArgs {
    const float* gx;       // direction g, frame n: gx + g * gx_gs + n * gx_ns, [4C][HW]
    const float* gh;       // [2][B][4C][HW] or nullptr
    float* cstate;         // [2][B][C][HW]
    float* hout;           // direction g, frame n: hout + g * h_gs + n * h_ns, [C][HW]
    unsigned short* hsb;   // [2][B][C16][HW][terms][16]
    long gx_gs, gx_ns, h_gs, h_ns;
    int C, B;
    long HW;
    int first;             // 1: c_prev = 0 as well
    int terms;
};
#ifdef BDE_SB_TU
__device__ __forceinline__ float sb_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f)); }
__device__ __forceinline__ float sb_tanh(float v) { return 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(v * -2.8853900817779268f)) - 1.f; }
template <int TERMS>
__global__ __launch_bounds__(256) void lstm_point_kernel(const LstmPointArgs a) {
    const int half = threadIdx.x & 1;
    const long p = (long)blockIdx.x * 128 + (threadIdx.x >> 1);
    const int c16 = blockIdx.y;
    const int g = blockIdx.z / a.B, n = blockIdx.z - g * a.B;
    if (p >= a.HW) return;
    const int C16 = gridDim.y, C = a.C;
    const float* gx = a.gx + g * a.gx_gs + n * a.gx_ns;
    const float* gh = a.gh ? a.gh + ((long)(g * a.B + n) * 4 * C) * a.HW : nullptr;
    float* cs = a.cstate + ((long)(g * a.B + n) * C) * a.HW;
    float* ho = a.hout + g * a.h_gs + n * a.h_ns;
    unsigned short t[8][TERMS];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ch = c16 * 16 + half * 8 + j;
        float h = 0.f;
        if (ch < C) {
            const long o = (long)ch * a.HW + p;
            float vi = gx[o], vf = gx[o + (long)C * a.HW], vo = gx[o + 2L * C * a.HW], vg = gx[o + 3L * C * a.HW];
            if (gh) { vi += gh[o]; vf += gh[o + (long)C * a.HW]; vo += gh[o + 2L * C * a.HW]; vg += gh[o + 3L * C * a.HW]; }
            const float cp = a.first ? 0.f : cs[o];
            const float c = sb_sigmoid(vf) * cp + sb_sigmoid(vi) * sb_tanh(vg);
            h = sb_sigmoid(vo) * sb_tanh(c);
            cs[o] = c;
            ho[o] = h;
        }
        sb_split_dev<TERMS>(h, t[j]);
    }
    unsigned short* o = a.hsb + ((((long)(g * a.B + n) * C16 + c16) * a.HW + p) * TERMS) * 16 + half * 8;
#pragma unroll
    for (int k = 0; k < TERMS; ++k) {
        uint4 v;
        v.x = t[0][k] | ((unsigned)t[1][k] << 16);
        v.y = t[2][k] | ((unsigned)t[3][k] << 16);
        v.z = t[4][k] | ((unsigned)t[5][k] << 16);
        v.w = t[6][k] | ((unsigned)t[7][k] << 16);
        *reinterpret_cast<uint4*>(o + k * 16) = v;
    }
}
int lstm_point_launch(const LstmPointArgs& a, hipStream_t s) {
    const dim3 grid((unsigned)cdivl(a.HW, 128), cdiv(a.C, 16), 2 * a.B);
    if (a.terms == 2) hipLaunchKernelGGL(lstm_point_kernel<2>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(lstm_point_kernel<3>, grid, dim3(256), 0, s, a);
    BDE_HIP(hipGetLastError());
    return BDE_OK;
}
#else
int lstm_point_launch(const LstmPointArgs& a, hipStream_t s);   // sb_tu.hip
#endif

// bytes of an SB16 image; buffers are sized for the wider format so the format can change without reallocation
static inline long split_bf16_bytes(long N, int C, long HW, int terms = 3) { return N * cdiv(C, 16) * HW * sb_pix_bytes(terms); }

// a.in = SB16 activations (as float*), a.wpk = split packed weights; group / frame strides of `in` in BYTES / 4 (floats).
// DB: two halo buffers in LDS, the next chunk's DMA in flight during the MFMAs of the current one -- for the small launches
// of the recurrent step, where a CU holds one or two workgroups and nobody else covers the staging.
constexpr int KS_HEAD3 = 53;                              // conv_sb_kernel's KS for the head3 form (see the kernel)
#ifndef CONV_SB_STAGGER
#define CONV_SB_STAGGER 0
#endif
#ifndef CONV_SB_T32_NT
#define CONV_SB_T32_NT 2
#endif
#ifndef CONV_SB_STAGED_EPI
#define CONV_SB_STAGED_EPI 1
#endif
// timing experiments only (results are wrong): 1 = no weight-fragment loads in the tap loop, 2 = pixel fragments read at the first tap
// of a chunk only, 4 = halo tiles staged for the first chunk only
#ifndef CONV_SB_DBG
#define CONV_SB_DBG 0
#endif
template <int KS, int STRIDE, int MT, int NT, int WM, int WN, int MAXI, bool DB, int TERMS>
__global__ __launch_bounds__(64 * WM * WN, 2) void conv_sb_kernel(const ConvArgs a) {
    // KS == KS_HEAD3: the head's 5x5 convolution on its three-columns-per-chunk image (api_pack.h, pack_head3; split_head3_kernel
    // below): ten taps (ky, g) at rows y + ky - 2 and image columns x + 3 g -- a footprint of 5 rows x 4 columns, no column padding
    // (the image carries its own two columns of left context).  2-D pixel tiles only.
    constexpr bool HEAD3 = KS == KS_HEAD3;
    constexpr int KY = HEAD3 ? 5 : KS, KX = HEAD3 ? 4 : KS;            // footprint of the taps in input pixels
    constexpr int PAD = KY / 2, PADX = HEAD3 ? 0 : KS / 2, TAPS = HEAD3 ? 10 : KS * KS;
    constexpr int SB_PIX_BYTES = sb_pix_bytes(TERMS), SB_LDS_PITCH = sb_lds_pitch(TERMS), SLOTS = sb_lds_slots(TERMS);
    constexpr int BN = WN * NT * 32;
    extern __shared__ __align__(16) unsigned char sb_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int hl = lane >> 5;
    int bx, by, z;
    conv_block_coords(a.xcd_remap, bx, by, z);
    if (CONV_SB_STAGGER > 0 && !DB) {
        // The workgroups of a launch start together and, without a second halo buffer, stop together at every chunk boundary
        // (barrier, DMA, wait, barrier).  The dispatch rounds -- one workgroup per CU each -- start a fraction of a chunk apart so
        // that the two or three workgroups of a CU cover each other's staging.
        const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const unsigned round = (lin >> 8) % 3u;
        for (unsigned i = 0; i < round; ++i) __builtin_amdgcn_s_sleep(CONV_SB_STAGGER);       // CONV_SB_STAGGER x 64 cycles per round
    }
    const int g = z / a.N, n = z - g * a.N;
    const int HW = a.Ho * a.Wo;
    int p_end, iy0, ix0, R, IW;
    int boff[NT], pix[NT];
    int ep_p0 = 0, ep_y0 = 0, ep_x0 = 0;                    // the tile's first pixel / corner, for the staged SB16 epilogue
    if (a.tile_cols > 0) {
        // 2-D pixel tile of tile_rows x tile_cols output pixels (= BN): the halo is a small rectangle instead of KS whole
        // row segments.  Pixel q of the tile = (q / tile_cols, q % tile_cols); lanes outside the image compute on halo zeros
        // and store nothing (pix = HW).
        const int TC = a.tile_cols, TR = BN / TC;
        const int tiles_x = (a.Wo + TC - 1) / TC;
        const int ty = bx / tiles_x, tx = bx - ty * tiles_x;
        const int y0 = ty * TR, x0 = tx * TC;
        ep_y0 = y0; ep_x0 = x0;
        p_end = HW;
        iy0 = y0 * STRIDE - PAD;
        ix0 = x0 * STRIDE - PADX;
        R = (TR - 1) * STRIDE + KY;
        IW = (TC - 1) * STRIDE + KX;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int q = (wn * NT + t) * 32 + (lane & 31);
            const int qy = q / TC, qx = q - qy * TC;
            const int y = y0 + qy, x = x0 + qx;
            pix[t] = (y < a.Ho && x < a.Wo) ? y * a.Wo + x : HW;
            boff[t] = (qy * STRIDE * IW + qx * STRIDE) * SB_LDS_PITCH + hl * 16;
        }
    } else {
        int p0;
        if (a.row_tiles > 0) {
            const int yy = bx / a.row_tiles, xt = bx - yy * a.row_tiles;
            p0 = yy * a.Wo + xt * BN;
            p_end = min(p0 + BN, (yy + 1) * a.Wo);
        } else {
            p0 = bx * BN;
            p_end = min(p0 + BN, HW);
        }
        ep_p0 = p0;
        const int p_last = p_end - 1;
        const int y_first = p0 / a.Wo, y_last = p_last / a.Wo;
        const bool one_row = (y_first == y_last);
        const int x_first = p0 - y_first * a.Wo;
        iy0 = y_first * STRIDE - PAD;
        ix0 = one_row ? x_first * STRIDE - PAD : -PAD;
        R = (y_last - y_first) * STRIDE + KS;
        IW = one_row ? (p_last - p0) * STRIDE + KS : a.Win + 2 * PAD;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int p = p0 + (wn * NT + t) * 32 + (lane & 31);
            pix[t] = p;
            const int pc = min(p, p_last);
            const int y = pc / a.Wo, x = pc - y * a.Wo;
            boff[t] = ((y - y_first) * STRIDE * IW + (x * STRIDE - PAD - ix0)) * SB_LDS_PITCH + hl * 16;
        }
    }
    // (the accumulator scale is requested HERE, not behind the K loop where it is used: there it was a dependent load and a
    //  vmcnt(0) at the end of every workgroup)
    const float unscale_v = TERMS == 2 ? a.acc_scale[0] : 1.f;
    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;

    const int C16 = a.nchunks;
    const long plane = (long)a.Hs * a.Ws * SB_PIX_BYTES;                  // bytes of one 16-channel chunk of a frame
    const unsigned char* inb = reinterpret_cast<const unsigned char*>(a.in) + (g * a.in_gs + n * a.in_ns) * 4;
    // Halo staging by LDS-DMA (global_load_lds_dwordx4: 16 bytes per lane straight into LDS, no staging registers): the
    // tile is R * IW pixels x SLOTS sixteen-byte slots (2 per term + the pad slot); block b = 64 consecutive slots = 1 KiB of LDS
    // = one wave instruction, the waves take blocks wave, wave + NW, ...  A lane's source is its pixel's piece, or 16 bytes
    // of zeros for pixels outside the image and for the pad slot.
    constexpr int NW = WM * WN;
    const int nslots = R * IW * SLOTS;
    const int nblk = (nslots + 63) >> 6;
    unsigned goff[MAXI];
    unsigned vmask = 0;
#pragma unroll
    for (int it = 0; it < MAXI; ++it) {
        const int i = (wave + it * NW) * 64 + lane;
        const int px = i / SLOTS, q = i - px * SLOTS;
        const int r = px / IW, c = px - r * IW;
        // (stride 2 puts sixteen lanes' 16-byte fragment reads two pixels apart, i.e. into eight bank groups; storing a tile
        //  row as even columns | odd columns makes them adjacent -- measured: no faster, 157 -> 162 us at level 1; not kept)
        const int iy = iy0 + r, ix = ix0 + c;
        const bool ok = i < nslots && q < SLOTS - 1 && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
        goff[it] = ok ? (unsigned)((iy * a.Ws + ix) * SB_PIX_BYTES + q * 16) : 0u;
        if (ok) vmask |= 1u << it;
    }
    const unsigned char* zero16 = reinterpret_cast<const unsigned char*>(a.zeros);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int tile_bytes = nblk * 1024;                    // one halo buffer
    auto stage = [&](int c16) {
        const unsigned char* cb = inb + c16 * plane;
        const unsigned dstb = sb_dyn_lds_base() + (DB ? (c16 & 1) * tile_bytes : 0);
#pragma unroll
        for (int it = 0; it < MAXI; ++it) {
            const int blk = wave_u + it * NW;
            if (blk < nblk) {
                const unsigned char* src = ((vmask >> it) & 1u) ? cb + goff[it] : zero16;
                sb_lds_dma16(src, dstb + blk * 1024);       // (asm: invisible to the compiler's wait counting, split.h)
            }
        }
    };
    // weight fragments: [co tile][chunk][tap][term][64][8] x 16 bit
    const int cot0 = (by * WM + wm) * MT;
    const int ncot = (a.Cout + 31) / 32;
    const sb8* wfr[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
        wfr[m] = reinterpret_cast<const sb8*>(a.wpk + g * a.w_gs) + ((long)min(cot0 + m, ncot - 1) * C16 * TAPS * TERMS) * 64 + lane;

    // Weight fragments run PF taps ahead of the MFMAs that use them, in a ring of RING = PF + 1 register sets; the taps
    // of a chunk are a whole number of ring turns, so the ring position of a tap is a compile-time constant and the
    // stream simply continues across chunks (the fragments of consecutive (chunk, tap) pairs are consecutive in memory).
    // A launch with one or two workgroups per CU (the recurrent step) has nobody else to cover an L2 round trip.
    constexpr int RING = (TAPS % 3 == 0) ? 3 : 5, PF = RING - 1;
    static_assert(TAPS % RING == 0, "ring position of a tap must not depend on the chunk");
    const int S = C16 * TAPS;                               // (chunk, tap) pairs of the launch
    sb8 af[RING][MT][TERMS];
    if (DB) stage(0);                                       // BEFORE the fragment prefetch: the counted wait below relies on the DMA being older
#pragma unroll
    for (int q = 0; q < PF; ++q)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int k = 0; k < TERMS; ++k) af[q][m][k] = wfr[m][((long)min(q, S - 1) * TERMS + k) * 64];
    for (int c16 = 0; c16 < C16; ++c16) {
        if (!DB) {
            __syncthreads();                               // every wave is done with the previous chunk's tile
            if (!(CONV_SB_DBG & 4) || c16 == 0) stage(c16);
        }
        const unsigned char* tile = sb_lds + (DB ? (c16 & 1) * tile_bytes : 0);
        if (DB) {
            // This wave's DMA blocks of chunk c16 have landed: they were issued before the PF taps of weight fragments that
            // are still in flight (stage(0) ahead of the prefetch above, stage(c16) ahead of the tap loop of chunk c16 - 1),
            // so the counted wait covers them and leaves the fragment ring in flight.  A bare s_barrier follows
            // (__syncthreads() would drain vmcnt to 0 and the ring with it); LDS reads of the other buffer were waited
            // for by the MFMAs that consumed them.
            // (the barrier as asm with a memory clobber: the intrinsic is IntrNoMem, LDS reads or the next stage's DMA may not cross it)
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(TERMS * PF * MT) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }
        if (DB && c16 + 1 < C16) stage(c16 + 1);            // (every wave left the other buffer before that barrier)
        sb8 bfr[NT][TERMS];
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int cur = tap % RING, nxt = (tap + PF) % RING;
            {
                const long sp = min(c16 * TAPS + tap + PF, S - 1);
                if (!(CONV_SB_DBG & 1))
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int k = 0; k < TERMS; ++k) af[nxt][m][k] = wfr[m][(sp * TERMS + k) * 64];
            }
            // The prefetch stays HERE: left alone, the scheduler sinks every fragment load to just in front of the tap that uses
            // it (fewer live registers, one more wave per SIMD) -- ISA of round 4: `global_load; s_waitcnt vmcnt(0); v_mfma` at
            // every tap, an L2 round trip in front of each six MFMAs.
            __builtin_amdgcn_sched_barrier(0);
            const int ky = HEAD3 ? tap >> 1 : tap / KS, kx = HEAD3 ? 3 * (tap & 1) : tap - (tap / KS) * KS;
            if (!(CONV_SB_DBG & 2) || tap == 0)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int k = 0; k < TERMS; ++k)
                    bfr[t][k] = *reinterpret_cast<const sb8*>(tile + boff[t] + (ky * IW + kx) * SB_LDS_PITCH + k * 32);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[m][t] = sb_mma32<TERMS>(af[cur][m], bfr[t], acc[m][t]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // two-term weights are packed times a power of two (readfirstlane: into a scalar register, and the wait for it sits here)
    const float unscale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, unscale_v)));
    float fin[MT][NT][16];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) fin[m][t][rr] = TERMS == 2 ? acc[m][t][rr] * unscale : acc[m][t][rr];
    if (CONV_SB_STAGED_EPI && a.sb_out != nullptr && a.pred_out == nullptr && a.sb_stage_ok) {
        // ---- SB16 output through LDS ------------------------------------------------------------------------------------------------
        // generic_epilogue stores a lane's four consecutive channels of one term: 8 bytes, 64 bytes apart from lane to lane -- a wave's
        // store instruction is 64 partial writes into 32 different lines, eight such instructions per line (11 M write requests for
        // encoder 0's 90 MB).  The halo tiles are dead: the workgroup's output tile is assembled there in the image's own layout
        // ([chunk][pixel][term][16 channels]) and leaves as 16-byte pieces, consecutive lanes = consecutive bytes.
        constexpr int LC = WM * MT * 2, NTHR = 64 * WM * WN;  // 16-channel chunks of the workgroup's rows
        constexpr int PB = 32 * TERMS, PITCH = PB + 16, PPP = 2 * TERMS;
        __syncthreads();                                   // every wave is done with the last chunk's tile
        unsigned char* stg = sb_lds;
        const float* biasg = a.bias + g * a.bias_gs;
        float gm = 0.f;
        if (cot0 * 32 < a.Cout) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int co0 = (cot0 + m) * 32 + 8 * q + 4 * hl;          // = acc_row(4 q, lane): four consecutive channels
                    float b4[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) b4[i] = biasg[min(co0 + i, a.Cout - 1)];
                    const int lc = (wm * MT + m) * 2 + (q >> 1), c16 = 8 * (q & 1) + 4 * hl;
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        float v4[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) v4[i] = act_apply(fin[m][t][4 * q + i] + b4[i], a.act);
                        unsigned char* d = stg + (lc * BN + (wn * NT + t) * 32 + (lane & 31)) * PITCH + c16 * 2;
                        if constexpr (TERMS == 2) {
                            uint2 hi, lo;
                            gm = sb_guard_max2(sb_guard_max2(gm, v4[0], v4[1]), v4[2], v4[3]);
                            split2_quad(v4, hi, lo);
                            *reinterpret_cast<uint2*>(d) = hi;
                            *reinterpret_cast<uint2*>(d + 32) = lo;
                        } else {
                            uint2 hi, mid, lo;
                            split3_quad(v4, hi, mid, lo);
                            *reinterpret_cast<uint2*>(d) = hi;
                            *reinterpret_cast<uint2*>(d + 32) = mid;
                            *reinterpret_cast<uint2*>(d + 64) = lo;
                        }
                    }
                }
        }
        if (TERMS == 2) sb_guard_flush(gm, a.sb_ovf);
        __syncthreads();
        unsigned char* sbb = reinterpret_cast<unsigned char*>(a.sb_out + g * a.sb_out_gs + n * a.sb_out_ns);
        const int chunk0 = by * LC;
        const int TC = a.tile_cols;
        for (int it = tid; it < LC * BN * PPP; it += NTHR) {
            const int q = it % PPP, r = it / PPP;
            const int pw = r % BN, lc = r / BN;
            int p;
            if (TC > 0) {
                const int qy = pw / TC, qx = pw - qy * TC;
                const int y = ep_y0 + qy, x = ep_x0 + qx;
                p = (y < a.Ho && x < a.Wo) ? y * a.Wo + x : HW;
            } else {
                p = ep_p0 + pw;
            }
            if (p >= p_end || (chunk0 + lc) * 16 >= a.Cout) continue;
            *reinterpret_cast<uint4*>(sbb + ((long)(chunk0 + lc) * HW + p) * PB + q * 16) =
                *reinterpret_cast<const uint4*>(stg + (lc * BN + pw) * PITCH + q * 16);
        }
        return;
    }
    float mu[NT], rstd[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) mu[t] = rstd[t] = 0.f;
    if (cot0 * 32 < a.Cout) generic_epilogue<MT, NT, 16>(a, fin, pix, 0, lane, g, n, HW, p_end, false, mu, rstd, cot0);
}

// host-side geometry (mirrors the kernel)
static inline long conv_sb_halo_pixels(int KS, int STRIDE, int BN, int Win, int Ho, int Wo, int row_tiles) {
    const int PAD = KS / 2;
    const int HW = Ho * Wo;
    if (row_tiles > 0) return (long)KS * ((std::min(BN, Wo) - 1) * STRIDE + KS);
    long best = 0;
    for (int p0 = 0; p0 < HW; p0 += BN) {
        const int pl = std::min(p0 + BN, HW) - 1;
        const int yf = p0 / Wo, yl = pl / Wo;
        const long R = (long)(yl - yf) * STRIDE + KS;
        const long e = R * (yf == yl ? (pl - p0) * STRIDE + KS : Win + 2 * PAD);
        best = std::max(best, e);
    }
    return best;
}

// Workgroup shapes (output channels x pixels; all four waves):
//   SB_128x128  four waves of 32 x 128 sharing the pixel fragments (the 3x3 gate convolutions, decoder 0)
//   SB_128x64   four waves of 32 x 64: rows that 128-pixel tiles fill badly / stride-2 halos that do not fit at 128
//   SB_64x128   2 x 2 waves of 32 x 64 for 64 output channels (5x5 stride 1: decoder 1)
//   SB_32x256T  four waves of 32 channels x 64 pixels of ONE 2-D tile (5x5 stride 1, 32 output channels: decoder 2 + predI, head)
//   SB_64x128T  2 x 2 waves of 32 x 64 on a 2-D tile (5x5 stride 2, 64 output channels: encoder 0)
enum { SB_NONE = 0, SB_128x128, SB_128x64, SB_64x128, SB_32x256T, SB_64x128T };
// 2-D tiles: columns per tile (rows = BN / columns) -- the fuller cover of the map, 16 columns unless 8 or 32 cover it 5 % better
static inline int conv_sb_tile_cols(int BN, int Ho, int Wo) {
    auto fill = [&](int tc) { const int tr = BN / tc; return (double)Ho * Wo / ((double)cdiv(Ho, tr) * tr * cdiv(Wo, tc) * tc); };
    int best = 16;
    for (int tc : {8, 32})
        if (fill(tc) > fill(best) + 0.05) best = tc;
    return best;
}
// pixel tiles aligned to image rows when a row is at least 3/4 of a tile (row_tiles > 0), linear over rows otherwise
static inline int conv_sb_row_tiles(int BN, int Ho, int Wo) {
    const double fill_lin = (double)Ho * Wo / ((double)cdiv(Ho * Wo, BN) * BN);
    const int rt = cdiv(Wo, BN);
    const double fill_row = (double)Wo / ((double)rt * BN);
    return fill_row >= fill_lin - 0.1 ? rt : 0;
}
// How the pixel tiles of BN pixels are laid over the map: row_tiles > 0 (aligned to image rows), 0 (linear over rows), or
// -1 when the halo fits LDS neither way (two workgroups per CU, <= 24 DMA blocks per wave).  The fuller layout is tried
// first; a 320-pixel row does not take linear 128-pixel tiles (a tile across two rows stages four full rows) but does take
// 128 + 128 + 64.
static inline int conv_sb_tile_mode(int KS, int stride, int BN, int Win, int Ho, int Wo, int terms) {
    auto fits = [&](int rt) {
        const long halo = conv_sb_halo_pixels(KS, stride, BN, Win, Ho, Wo, rt);
        return halo * sb_lds_pitch(terms) <= 78 * 1024 && (halo * sb_lds_slots(terms) + 63) / 64 <= 4 * 24;
    };
    const int pref = conv_sb_row_tiles(BN, Ho, Wo);
    if (fits(pref)) return pref;
    const int other = pref ? 0 : cdiv(Wo, BN);
    return fits(other) ? other : -1;
}
static inline double conv_sb_fill(int BN, int Ho, int Wo, int rt) {
    return rt > 0 ? (double)Wo / ((double)rt * BN) : (double)Ho * Wo / ((double)cdiv(Ho * Wo, BN) * BN);
}
// Which shape conv_sb_launch takes for this convolution (asked before the input is split).  Among the pixel-tile widths
// that fit, the better filled one; 128 pixels reuse every weight fragment twice as often as 64, so 64 has to be 15 % fuller.
static inline int conv_sb_pick(int KS, int stride, int Cout, int Win, int Ho, int Wo, int terms) {
    if (!((KS == 3 && stride == 1) || (KS == 5 && (stride == 1 || stride == 2)))) return SB_NONE;
    const int m128 = conv_sb_tile_mode(KS, stride, 128, Win, Ho, Wo, terms);
    const double f128 = m128 >= 0 ? conv_sb_fill(128, Ho, Wo, m128) : 0.0;
    if (Cout >= 128) {
        const int m64 = conv_sb_tile_mode(KS, stride, 64, Win, Ho, Wo, terms);
        const double f64 = m64 >= 0 ? 0.85 * conv_sb_fill(64, Ho, Wo, m64) : 0.0;
        if (f128 < 0.6 && f64 < 0.6 * 0.85) return SB_NONE;
        return f128 >= f64 ? SB_128x128 : SB_128x64;
    }
    if (Cout == 64 && KS == 5 && stride == 1 && f128 >= 0.6) return SB_64x128;
    if (Cout == 64 && KS == 5 && stride == 2 && Ho >= 16 && Wo >= 16) return SB_64x128T;
    if (Cout == 32 && KS == 5 && stride == 1 && Ho >= 16 && Wo >= 16) return SB_32x256T;
    return SB_NONE;
}
static inline bool conv_sb_fits(int KS, int stride, int Cout, int Win, int Ho, int Wo, int terms) {
    return conv_sb_pick(KS, stride, Cout, Win, Ho, Wo, terms) != SB_NONE;
}

#ifdef BDE_SB_TU
template <int KS, int STRIDE, int MT, int NT, int WM, int WN, int MAXI, bool DB, int TERMS>
static int conv_sb_launch_t(const ConvArgs& a, int G, hipStream_t stream, long halo_px) {
    constexpr int BN = WN * NT * 32;
    const size_t lds = (DB ? 2 : 1) * (((size_t)halo_px * sb_lds_slots(TERMS) + 63) / 64 * 1024);   // whole 1-KiB DMA blocks
    auto kern = conv_sb_kernel<KS, STRIDE, MT, NT, WM, WN, MAXI, DB, TERMS>;
    static unsigned char raised[BDE_MAX_DEVICES];
    if (lds > 64 * 1024) BDE_HIP(raise_dynamic_lds(raised, (const void*)kern));
    dim3 grid(a.row_tiles > 0 ? a.Ho * a.row_tiles : cdiv(a.Ho * a.Wo, BN), cdiv(a.Cout, WM * MT * 32), G * a.N);
    if (a.tile_cols > 0) grid.x = cdiv(a.Wo, a.tile_cols) * cdiv(a.Ho, BN / a.tile_cols);
    if (grid.x == 0 || grid.y == 0 || grid.z == 0) return BDE_OK;
    ConvArgs b = a;
    b.sb_stage_ok = (size_t)(WM * MT * 2) * BN * (32 * TERMS + 16) <= lds ? 1 : 0;
    hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), lds, stream, b);
    BDE_HIP(hipGetLastError());
    return BDE_OK;
}

template <int KS, int STRIDE, int MT, int NT, int WM, int WN, int TERMS>
static int conv_sb_launch_shape(ConvArgs a, int G, hipStream_t stream, bool* launched) {
    constexpr int BN = WN * NT * 32;
    constexpr int SB_LDS_PITCH = sb_lds_pitch(TERMS);
    *launched = false;
    const int best_rt = conv_sb_tile_mode(KS, STRIDE, BN, a.Win, a.Ho, a.Wo, TERMS);
    if (best_rt < 0) return BDE_OK;
    a.row_tiles = best_rt;
    const long halo = conv_sb_halo_pixels(KS, STRIDE, BN, a.Win, a.Ho, a.Wo, best_rt);
    const long blocks = (halo * sb_lds_slots(TERMS) + 63) / 64;             // 1-KiB DMA blocks of the tile
    const long per_wave = (blocks + WM * WN - 1) / (WM * WN);
    if (halo * SB_LDS_PITCH > 78 * 1024 || per_wave > 24) return BDE_OK;    // (two workgroups per CU) else the fp32 kernels
    *launched = true;
    // few workgroups (the recurrent step): double-buffered halo, one workgroup per CU covers its own staging
    const long wgs = (long)(best_rt > 0 ? a.Ho * best_rt : cdiv(a.Ho * a.Wo, BN)) * cdiv(a.Cout, WM * MT * 32) * G * a.N;
    if (KS == 3 && wgs <= 768 && halo * SB_LDS_PITCH * 2 <= 150 * 1024) {
        if (per_wave <= 8) return conv_sb_launch_t<KS, STRIDE, MT, NT, WM, WN, 8, true, TERMS>(a, G, stream, halo);
        if (per_wave <= 16) return conv_sb_launch_t<KS, STRIDE, MT, NT, WM, WN, 16, true, TERMS>(a, G, stream, halo);
    }
    if (per_wave <= 8) return conv_sb_launch_t<KS, STRIDE, MT, NT, WM, WN, 8, false, TERMS>(a, G, stream, halo);
    if (per_wave <= 16) return conv_sb_launch_t<KS, STRIDE, MT, NT, WM, WN, 16, false, TERMS>(a, G, stream, halo);
    return conv_sb_launch_t<KS, STRIDE, MT, NT, WM, WN, 24, false, TERMS>(a, G, stream, halo);
}

// 2-D pixel tiles (conv_sb_kernel, a.tile_cols > 0)
template <int KS, int STRIDE, int MT, int NT, int WM, int WN, int TERMS>
static int conv_sb_launch_2d(ConvArgs a, int G, hipStream_t stream, bool* launched) {
    constexpr int BN = WN * NT * 32;
    *launched = false;
    a.row_tiles = 0;
    a.tile_cols = conv_sb_tile_cols(BN, a.Ho, a.Wo);
    const int TR = BN / a.tile_cols;
    constexpr int KY = KS == KS_HEAD3 ? 5 : KS, KX = KS == KS_HEAD3 ? 4 : KS;       // footprint of the taps (conv_sb_kernel)
    const long halo = (long)((TR - 1) * STRIDE + KY) * ((a.tile_cols - 1) * STRIDE + KX);
    const long blocks = (halo * sb_lds_slots(TERMS) + 63) / 64;
    const long per_wave = (blocks + WM * WN - 1) / (WM * WN);
    if (halo * sb_lds_pitch(TERMS) > 78 * 1024 || per_wave > 24) return BDE_OK;
    *launched = true;
    if (per_wave <= 8) return conv_sb_launch_t<KS, STRIDE, MT, NT, WM, WN, 8, false, TERMS>(a, G, stream, halo);
    if (per_wave <= 16) return conv_sb_launch_t<KS, STRIDE, MT, NT, WM, WN, 16, false, TERMS>(a, G, stream, halo);
    return conv_sb_launch_t<KS, STRIDE, MT, NT, WM, WN, 24, false, TERMS>(a, G, stream, halo);
}

template <int KS, int STRIDE, int TERMS>
static int conv_sb_launch_ks(const ConvArgs& a, int G, hipStream_t stream, bool* launched) {
    // SB_128x128: four waves of 32 channels x 128 pixels share the pixel fragments through LDS and each stream their own
    // weight fragments (12 KB per tap and workgroup from L2); 2 x 2 waves of 64 x 64 fetch every weight fragment twice
    // (24 KB per tap), which is what saturated the L1 return path.
    int shape = conv_sb_pick(KS, STRIDE, a.Cout, a.Win, a.Ho, a.Wo, TERMS);
    if (STRIDE == 1 && shape == SB_128x128 && tuning().conv_nt == 0) {
        // too few 128-pixel workgroups to give every CU two (a small map, few frames): 64-pixel tiles double them
        // (decoder 0 at 46 x 60, 16 frames: 275 -> 239 us; the stride-2 encoder of level 2 lost: 169 -> 200 us, kept at 128)
        const int groups = a.decide_groups ? a.decide_groups : G;
        const long wgs = (long)cdiv(a.Ho * a.Wo, 128) * cdiv(a.Cout, 128) * groups * a.N;
        if (wgs < 2 * 256 && conv_sb_tile_mode(KS, STRIDE, 64, a.Win, a.Ho, a.Wo, TERMS) >= 0) shape = SB_128x64;
    }
    if (shape == SB_128x128) {
        if (tuning().conv_nt == 2) return conv_sb_launch_shape<KS, STRIDE, 2, 2, 2, 2, TERMS>(a, G, stream, launched);
        return conv_sb_launch_shape<KS, STRIDE, 1, 4, 4, 1, TERMS>(a, G, stream, launched);
    }
    if (shape == SB_128x64) return conv_sb_launch_shape<KS, STRIDE, 1, 2, 4, 1, TERMS>(a, G, stream, launched);
    if constexpr (KS == 5 && STRIDE == 1) {
        if (shape == SB_64x128) return conv_sb_launch_shape<KS, STRIDE, 1, 2, 2, 2, TERMS>(a, G, stream, launched);
        // (CONV_SB_T32_NT = 4: 128 pixels per wave, a 512-pixel tile -- the four waves of this shape read the SAME weight fragments, each
        //  through the vector memory path: twice the pixels per wave halve those bytes per MFMA; needs >= 4 workgroups per CU of such tiles)
        if (shape == SB_32x256T) {
            if (CONV_SB_T32_NT == 4 && (long)cdiv(a.Ho * a.Wo, 512) * G * a.N >= 4 * 256)
                return conv_sb_launch_2d<KS, STRIDE, 1, 4, 1, 4, TERMS>(a, G, stream, launched);
            return conv_sb_launch_2d<KS, STRIDE, 1, 2, 1, 4, TERMS>(a, G, stream, launched);
        }
    }
    if constexpr (KS == 5 && STRIDE == 2)
        if (shape == SB_64x128T) return conv_sb_launch_2d<KS, STRIDE, 1, 2, 2, 2, TERMS>(a, G, stream, launched);
    *launched = false;
    return BDE_OK;
}

// a.sb_terms = the format of a.in / a.wpk (and of a.sb_out)
int conv_sb_launch(int KS, int stride, const ConvArgs& a, int G, hipStream_t stream, bool* launched) {
    *launched = false;
    if (a.sb_terms == 2) {
        if (KS == 3 && stride == 1) return conv_sb_launch_ks<3, 1, 2>(a, G, stream, launched);
        if (KS == 5 && stride == 1) return conv_sb_launch_ks<5, 1, 2>(a, G, stream, launched);
        if (KS == 5 && stride == 2) return conv_sb_launch_ks<5, 2, 2>(a, G, stream, launched);
        return BDE_OK;
    }
    if (KS == 3 && stride == 1) return conv_sb_launch_ks<3, 1, 3>(a, G, stream, launched);
    if (KS == 5 && stride == 1) return conv_sb_launch_ks<5, 1, 3>(a, G, stream, launched);
    if (KS == 5 && stride == 2) return conv_sb_launch_ks<5, 2, 3>(a, G, stream, launched);
    return BDE_OK;
}
// the head's convolution on its head3 image: a.in = that image (a.Win = a.Ws = W + 3, one chunk), a.wpk = the head3 packing
int conv_sb_launch_head3(const ConvArgs& a, hipStream_t stream, bool* launched) {
    *launched = false;
    if (a.Cout != 32 || a.Ho < 16 || a.Wo < 16) return BDE_OK;
    if (a.sb_terms == 2) return conv_sb_launch_2d<KS_HEAD3, 1, 1, 2, 1, 4, 2>(a, 1, stream, launched);
    return conv_sb_launch_2d<KS_HEAD3, 1, 1, 2, 1, 4, 3>(a, 1, stream, launched);
}
#else
int conv_sb_launch(int KS, int stride, const ConvArgs& a, int G, hipStream_t stream, bool* launched);   // sb_tu.hip
int conv_sb_launch_head3(const ConvArgs& a, hipStream_t stream, bool* launched);
#endif

}  // namespace bde
