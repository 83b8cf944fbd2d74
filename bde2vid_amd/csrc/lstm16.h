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
//   A = packed weights  [hidden16 block][channel chunk 8][tap 9][k4 2][64 lanes][gate 4]  (ds_read_b128 = four fragments),
//       lane l = W[gate*Ch + hb*16 + (l&15)][ci = chunk*8 + k4*4 + (l>>4)][tap]
//   B = LDS halo tile [8 channels][3 rows][66 cols], plane stride padded to 16 mod 32 floats so the
//       four channel rows of a fragment fall on disjoint banks
//   D: row = hidden channel, col = pixel -> the four gates of a (channel, pixel) sit in one lane.
#pragma once
#include <hip/hip_runtime.h>
#include "conv_mfma.h"

namespace bde {

typedef float f32x4_ __attribute__((ext_vector_type(4)));

// sigmoid / tanh on the hardware 2^x and reciprocal (1 ulp each): absolute error ~2e-7, against ~25 and
// ~35 instructions for the libm forms -- the pointwise tail of a step is 5 of these per (channel, pixel)
// and every wave of a launch reaches it at the same time, so it is not hidden behind other waves' MFMAs
__device__ __forceinline__ float sigmoid_fast(float v) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f));
}
__device__ __forceinline__ float tanh_fast(float v) {
    return 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(v * -2.8853900817779268f)) - 1.f;
}

constexpr int L16_CK = 8;                        // channels per stage
constexpr int L16_AFL = 9 * 2 * 4 * 64;          // floats of one stage's weight fragments (18 KiB)
// LDS row stride of the halo tile [channel][tile row][IWP]: >= PXW + 2 and chosen so that the plane
// stride (ROWS + 2) * IWP is 16 (mod 32) floats -- the four channel rows of a B fragment then fall on
// disjoint banks
constexpr int l16_iwp(int rows, int pxw) {
    int v = pxw + 2;
    while (((rows + 2) * v) % 32 != 16) ++v;
    return v;
}

// A workgroup covers ROWS image rows x PXW pixels of 16 hidden channels; a wave owns SEG 16-pixel
// segments of one row (ROWS*PXW/(16*SEG) waves = 4).  <1,64,1> / <2,32,1> / <4,16,1> keep every wave
// busy on narrow maps; <1,128,2> halves the waves of a wide level: level 0 of config A is then 736
// workgroups that are all resident at once (the one-segment shape needs 1472 on 1024 slots, and its
// thinly populated second round cost about a quarter of the launch), each weight fragment read from
// LDS feeds two MFMAs, and the weights are staged once per 128 pixels instead of once per 64.
// The halo tile is staged row-wise: a pass of all threads copies NT/PXW tile rows of PXW floats
// (coalesced, LDS address affine in the pass number), one more pass the two right-hand halo columns.
// Uses the ConvArgs fields of EPI_LSTM (in = h_prev, out = h, gx, cstate, first, strides).
// HC8: a workgroup owns 8 hidden channels instead of 16; an MFMA tile then stacks two gates (rows 0-7 gate a,
// rows 8-15 gate b of the same 8 channels), two tiles (i|f, o|g) per k-step instead of four.  Waves of half the
// size: a level whose 16-channel launch leaves a CU with one or two workgroups (level 2 of config A: 384 on 256
// CUs, the busiest SIMD carrying two 74 k-cycle waves) becomes 768 workgroups = three 37 k-cycle waves per SIMD.
template <int ROWS, int PXW, int SEG, bool HC8>
__global__ __launch_bounds__(ROWS * PXW * 4 / SEG) __attribute__((amdgpu_waves_per_eu(SEG == 1 ? 4 : 3, 8))) void lstm16_step_kernel(const ConvArgs a) {
    static_assert(!HC8 || SEG == 1, "8-channel workgroups are built for one-segment waves");
    constexpr int NG = HC8 ? 2 : 4;                  // MFMA tiles per k-step
    constexpr int AFL = 9 * 2 * 64 * NG;             // floats of one stage's weight fragments
    constexpr int NT = ROWS * PXW * 4 / SEG;         // threads per workgroup (64 per wave)
    constexpr int AFL4 = AFL / 4;                    // float4 of one stage's weight fragments
    constexpr int AK4 = (AFL4 + NT - 1) / NT;        // 16-byte weight loads per thread and stage
    constexpr int R = ROWS + 2;
    constexpr int IWP = l16_iwp(ROWS, PXW);
    constexpr int PLS = R * IWP;                     // plane (channel) stride in LDS
    constexpr int NR = L16_CK * R;                   // tile rows per stage
    constexpr int RPP = NT / PXW;                    // tile rows per staging pass
    constexpr int NPASS = NR / RPP;
    static_assert(NR % RPP == 0 && NT % PXW == 0 && 2 * NR <= NT, "halo staging shape");
    constexpr int XT = PXW / (16 * SEG);             // waves along x
    __shared__ __align__(16) float ldsA[AFL];
    __shared__ __align__(16) float ldsB[L16_CK * PLS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int z = blockIdx.z;
    const int g = z / a.N, n = z - g * a.N;
    const int Ch = a.Cout / 4, W = a.Wo, H = a.Ho, HW = H * W;
    const int tiles_x = (W + PXW - 1) / PXW;
    const int ty = blockIdx.x / tiles_x;
    const int y0 = ty * ROWS, x0 = (blockIdx.x - ty * tiles_x) * PXW;
    const int hb = blockIdx.y;                               // hidden-channel block (HCB channels) of this workgroup
    const int wrow = wave / XT, wx = wave - wrow * XT;       // this wave's row and first 16-pixel segment
    const int pxl = wx * 16 * SEG + (lane & 15);             // x of this lane (segment 0) inside the tile
    const int y = y0 + wrow;
    const bool wave_active = (x0 + wx * 16 * SEG) < W && y < H;

    f32x4_ acc[SEG][NG];
#pragma unroll
    for (int sg = 0; sg < SEG; ++sg)
#pragma unroll
        for (int q = 0; q < NG; ++q) acc[sg][q] = f32x4_{0.f, 0.f, 0.f, 0.f};

    // gx / c_prev of this lane's (pixel, hidden channels) x SEG, fetched during the last stage.  16-channel
    // workgroups: the lane finishes channels hb*16 + 4*g4 + r, r < 4; 8-channel ones: hb*8 + 4*(g4&1) + r0 + e,
    // e < 2 with r0 = 0 on the lanes holding the first gate of a pair (g4 < 2) and 2 on the others.
    constexpr int NE = HC8 ? 2 : 4;
    const int g4 = lane >> 4;
    const int hc0 = HC8 ? hb * 8 + 4 * (g4 & 1) + (g4 < 2 ? 0 : 2) : hb * 16 + g4 * 4;
    float gv[SEG][4][NE], cprev[SEG][NE];
    const float* gxb = a.gx + g * a.gx_gs + n * a.gx_ns;
    float* cst = a.cstate + g * a.c_gs + n * a.c_ns;
    auto epi_load = [&]() {
#pragma unroll
        for (int sg = 0; sg < SEG; ++sg) {
            const unsigned p = (unsigned)(min(y, H - 1) * W + min(x0 + pxl + sg * 16, W - 1));   // clamped: in bounds
#pragma unroll
            for (int r = 0; r < NE; ++r) {
                const unsigned hc = (unsigned)min(hc0 + r, Ch - 1);
                const unsigned o = hc * (unsigned)HW + p;
#pragma unroll
                for (int q = 0; q < 4; ++q) gv[sg][q][r] = gxb[(unsigned)(q * Ch) * (unsigned)HW + o];
                cprev[sg][r] = a.first ? 0.f : cst[o];
            }
        }
    };

    if (a.first) epi_load();
    if (!a.first) {
        const float* inb = a.in + g * a.in_gs + n * a.in_ns;
        const float4* wsrc = reinterpret_cast<const float4*>(a.wpk + g * a.w_gs + (long)hb * a.nchunks * AFL);
        // halo staging slots of this thread
        const int sub = tid / PXW, col = tid - sub * PXW;
        const int ixm = x0 - 1 + col;
        const bool colok = ixm >= 0 && ixm < W;
        const int xrow = tid >> 1, ixx = x0 - 1 + PXW + (tid & 1);     // extra pass: right-hand halo columns
        const bool xok = tid < 2 * NR && ixx < W;
        // source offsets of this thread's halo slots inside a stage (the stage only moves the base pointer)
        unsigned bsrc[NPASS + 1];
        unsigned bmask = 0;
#pragma unroll
        for (int k = 0; k <= NPASS; ++k) {
            const int row = k < NPASS ? k * RPP + sub : xrow;
            const int ci = row / R, r = row - ci * R;
            const int iy = y0 - 1 + r;
            const bool ok = (k < NPASS ? colok : xok) && iy >= 0 && iy < H;
            if (ok) bmask |= 1u << k;
            bsrc[k] = ok ? (unsigned)(ci * HW + iy * W + (k < NPASS ? ixm : ixx)) : 0u;
        }
        float4 aw[AK4];
        float bw[NPASS + 1];
        auto stage_load = [&](int st) {
            // 16-byte loads (a stage of dword loads queued for thousands of cycles on the VMEM path)
#pragma unroll
            for (int k = 0; k < AK4; ++k) {
                const int i4 = tid + k * NT;
                aw[k] = (i4 < AFL4) ? wsrc[(long)st * AFL4 + i4] : float4{0.f, 0.f, 0.f, 0.f};
            }
            const float* cb = inb + (long)st * L16_CK * HW;
            // exec-masked loads: fastest of the three zero-padding forms tried (select at load time,
            // select at LDS-store time, masked load)
#pragma unroll
            for (int k = 0; k <= NPASS; ++k) bw[k] = ((bmask >> k) & 1u) ? cb[bsrc[k]] : 0.f;
        };
        const int bofl = (lane >> 4) * PLS + wrow * IWP + pxl;   // tap (0,0), k4 = 0, segment 0
        const int nst = a.nchunks;
        auto stage_store = [&]() {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < AK4; ++k) {
                const int i4 = tid + k * NT;
                if (i4 < AFL4) reinterpret_cast<float4*>(ldsA)[i4] = aw[k];
            }
#pragma unroll
            for (int k = 0; k < NPASS; ++k) ldsB[(k * RPP + sub) * IWP + col] = bw[k];
            if (tid < 2 * NR) ldsB[xrow * IWP + PXW + (tid & 1)] = bw[NPASS];
            __syncthreads();
        };
        auto stage_mfma = [&]() {
            // Fragment reads are written one tap ahead of the MFMAs (two register sets); their final interleaving
            // is left to the compiler's iglp_opt strategies.
            float bq[2][2][SEG], aq[2][2][NG];
            auto frag_read = [&](int tap, int buf) {
                const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
                for (int k4 = 0; k4 < 2; ++k4) {
#pragma unroll
                    for (int sg = 0; sg < SEG; ++sg) bq[buf][k4][sg] = ldsB[bofl + k4 * 4 * PLS + ky * IWP + kx + sg * 16];
                    if constexpr (HC8) {
                        const float2 av = *reinterpret_cast<const float2*>(ldsA + ((tap * 2 + k4) * 64 + lane) * 2);
                        aq[buf][k4][0] = av.x;
                        aq[buf][k4][1] = av.y;
                    } else {
                        const f32x4_ av = *reinterpret_cast<const f32x4_*>(ldsA + ((tap * 2 + k4) * 64 + lane) * 4);
#pragma unroll
                        for (int q = 0; q < 4; ++q) aq[buf][k4][q] = av[q];
                    }
                }
            };
            frag_read(0, 0);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                // the compiler's MFMA/DS interleaving (iglp_opt) measured ahead of pinning the reads with
                // sched_barrier(0): two-segment shape -7 % with strategy 0, 8-channel shape -3 % with strategy 1
                __builtin_amdgcn_iglp_opt(SEG == 2 ? 0 : 1);
                const int buf = tap & 1;
                if (tap + 1 < 9) frag_read(tap + 1, buf ^ 1);
#pragma unroll
                for (int k4 = 0; k4 < 2; ++k4)
#pragma unroll
                    for (int sg = 0; sg < SEG; ++sg)
#pragma unroll
                        for (int q = 0; q < NG; ++q)
                            acc[sg][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[buf][k4][q], bq[buf][k4][sg], acc[sg][q], 0, 0, 0);
            }
        };
        stage_load(0);
        for (int st = 0; st + 1 < nst; ++st) {
            stage_store();
            stage_load(st + 1);
            if (wave_active) stage_mfma();                    // (else this wave's pixels lie outside the map)
        }
        // last stage, peeled: the staging registers are free, the pointwise operands take their place
        stage_store();
        epi_load();
        if (wave_active) stage_mfma();
    }

    // ---- pointwise (submodules.py:320-332); gate order i, f, o, g ------------------------------
    if (y >= H) return;
    float* hout = a.out + g * a.out_gs + n * a.out_ns;
#pragma unroll
    for (int sg = 0; sg < SEG; ++sg) {
        const int x = x0 + pxl + sg * 16;
        float gi[NE], gf[NE], go[NE], gg[NE];
        if constexpr (HC8) {
            // tile 0 = i | f, tile 1 = o | g: this lane holds one gate of each pair for channels 4*(g4&1) + r,
            // lane ^ 32 the other one; each of the two finishes two of the four channels
            const bool lo = g4 < 2;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float own0 = lo ? acc[sg][0][e] : acc[sg][0][2 + e], own1 = lo ? acc[sg][1][e] : acc[sg][1][2 + e];
                // the partner needs this lane's values of ITS two channels (the other pair of registers)
                const float snd0 = lo ? acc[sg][0][2 + e] : acc[sg][0][e], snd1 = lo ? acc[sg][1][2 + e] : acc[sg][1][e];
                const float oth0 = __shfl_xor(snd0, 32), oth1 = __shfl_xor(snd1, 32);
                gi[e] = lo ? own0 : oth0;
                gf[e] = lo ? oth0 : own0;
                go[e] = lo ? own1 : oth1;
                gg[e] = lo ? oth1 : own1;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) { gi[r] = acc[sg][0][r]; gf[r] = acc[sg][1][r]; go[r] = acc[sg][2][r]; gg[r] = acc[sg][3][r]; }
        }
        if (x >= W) continue;
        const long p = (long)y * W + x;
#pragma unroll
        for (int r = 0; r < NE; ++r) {
            const int hc = hc0 + r;
            if (hc >= Ch) continue;
            const long o = (long)hc * HW + p;
            const float vi = gi[r] + gv[sg][0][r], vf = gf[r] + gv[sg][1][r];
            const float vo = go[r] + gv[sg][2][r], vg = gg[r] + gv[sg][3][r];
            const float c = sigmoid_fast(vf) * cprev[sg][r] + sigmoid_fast(vi) * tanh_fast(vg);
            cst[o] = c;
            hout[o] = sigmoid_fast(vo) * tanh_fast(c);
        }
    }
}

template <int ROWS, int PXW, int SEG, bool HC8>
static int lstm16_launch_t(const ConvArgs& a, hipStream_t stream) {
    const int Ch = a.Cout / 4;
    dim3 grid(cdiv(a.Ho, ROWS) * cdiv(a.Wo, PXW), cdiv(Ch, HC8 ? 8 : 16), (a.lstm_groups ? a.lstm_groups : 2) * a.N);
    hipLaunchKernelGGL((lstm16_step_kernel<ROWS, PXW, SEG, HC8>), grid, dim3(ROWS * PXW * 4 / SEG), 0, stream, a);
    BDE_HIP(hipGetLastError());
    return BDE_OK;
}

// 16-channel launch geometry of a step: shape and workgroup count (the host picks the 8-channel weights and
// kernel from it)
static void lstm16_geometry(const ConvArgs& a, int* rows_, int* pxw_, int* seg_, long* wg16_) {
    auto fill = [&](int rows, int pxw) {
        const double segs = (double)a.Ho * cdiv(a.Wo, 16);
        return segs / ((double)cdiv(a.Ho, rows) * cdiv(a.Wo, pxw) * (rows * pxw / 16));
    };
    const int shapes[3][2] = {{1, 64}, {2, 32}, {4, 16}};
    int rows = 1, pxw = 64, seg = 1;
    double bf = -1;
    for (auto& sh : shapes) { const double f = fill(sh[0], sh[1]); if (f > bf + 0.02) { bf = f; rows = sh[0]; pxw = sh[1]; } }
    // more one-segment workgroups than the chip holds at once (4 per CU): two segments per wave
    const long wg1 = (long)cdiv(a.Ho, rows) * cdiv(a.Wo, pxw) * cdiv(a.Cout / 4, 16) * 2 * a.N;
    if (wg1 > 1024 && fill(1, 128) >= bf - 0.08) { rows = 1; pxw = 128; seg = 2; }
    *rows_ = rows; *pxw_ = pxw; *seg_ = seg; *wg16_ = wg1;
}

// 8-channel workgroups pay off when the 16-channel launch cannot give every CU the same number of workgroups
// and is small enough that doubling the count still fits the chip at once
static bool lstm16_wants_hc8(const ConvArgs& a) {
    int rows, pxw, seg; long wg16;
    lstm16_geometry(a, &rows, &pxw, &seg, &wg16);
    return seg == 1 && wg16 < 640 && wg16 % 256 != 0 && (a.Cout / 4) % 8 == 0;
}

#ifdef BDE_CONV_TU
int lstm16_launch(const ConvArgs& a, hipStream_t stream, bool hc8 = false) {
    if (a.Cin % L16_CK != 0) return fail(BDE_ERR_UNSUPPORTED, "recurrent step: %d hidden channels is not a multiple of %d", a.Cin, L16_CK);
    int rows, pxw, seg;
    long wg16;
    lstm16_geometry(a, &rows, &pxw, &seg, &wg16);
    if (const int f = tuning().lstm_shape) { seg = (f / 100000) % 100; rows = (f / 1000) % 100; pxw = f % 1000; }
    if (hc8 && seg != 1) return fail(BDE_ERR_ARG, "recurrent step: 8-channel workgroups need one-segment waves");
#define BDE_L16(R_, P_) \
    if (rows == R_ && pxw == P_ && seg == 1) return hc8 ? lstm16_launch_t<R_, P_, 1, true>(a, stream) : lstm16_launch_t<R_, P_, 1, false>(a, stream);
    BDE_L16(1, 64)
    BDE_L16(2, 32)
    BDE_L16(4, 16)
#undef BDE_L16
    if (rows == 1 && pxw == 128 && seg == 2) return lstm16_launch_t<1, 128, 2, false>(a, stream);
    return fail(BDE_ERR_ARG, "recurrent step: tile shape %dx%d (x%d segments) not built", rows, pxw, seg);
}

#else
int lstm16_launch(const ConvArgs& a, hipStream_t stream, bool hc8 = false);   // conv_tu.hip
#endif

}  // namespace bde
