// Recurrent ConvLSTM step with fp32-equivalent arithmetic on the 16-bit matrix cores (split operands, split.h: two fp16 terms /
// three MFMAs per fp32 block by default, three bf16 terms / six MFMAs as TERMS = 3) AND the pointwise tail in the epilogue -- the
// kernel on the sequential critical path of every level (V5.py:122-135; ConvLSTM.forward, submodules.py:293-334).
//
//   gates[4][Ch] = conv3x3([x_t | h_prev]; W) + b                         (the reference's stacked input, :316-317)
//   i, f, o = sigmoid, g = tanh;  c = f c + i g;  h = o tanh(c)            (chunk order i, f, o, g, :320-332)
//
// K runs over the 16-channel chunks of x_t (its SB16 image, written by the level's encoder convolution) and then those of h_prev
// (xchunks > 0); with xchunks = 0 the x-part arrives precomputed as gx (batched over T by conv_sb.h) and K is h_prev alone.
//
// lstm16.h runs the h-part on the fp32 matrix cores: 6.5 GFLOP per step at 64 FLOP/clk/SIMD is 100 k cycles when perfectly
// balanced (46-50 us); it measures 67 us.  What decides the step on the 16-bit cores is (a) every SIMD getting the same number of
// MFMAs, (b) the weights crossing the L2 -> CU path few times, (c) no second launch for the pointwise half.  A level's step is
// therefore cut to ONE work shape,
//   a wave = 32 gate rows x (NT x 32) pixels x a quarter .. all of K,
// by choosing, per level, how the four waves of a workgroup divide rows and K:
//   RTW x KW = 4:  RTW row tiles of 32 (weights differ per wave) x KW parts of K (waves of a row tile split the channel chunks
//   and are summed through LDS).  Config A, 184 x 240:  level 0: 4 x 1, NT 2 (692 workgroups, 3 per CU);  level 1: 2 x 2, NT 2
//   (704, 3 per CU);  level 2: 1 x 4, NT 3 = three image rows (512, 2 per CU): 1296 MFMAs per SIMD at every level with two-term
//   operands and K = [x | h], each weight fragment used for 64-96 pixels.
//
// Rows are packed GATE-INTERLEAVED: row 8 q + 4 hl + gate of a 32-row tile = that gate of hidden channel 8 tile + 4 hl + q, so the
// four gates of a (channel, pixel) are four consecutive accumulator registers of one lane (acc_row: rows 8 (r >> 2) + 4 (lane >> 5)
// + (r & 3)), a lane's four register groups are four CONSECUTIVE channels (one 8-byte store per term into the SB16 staging), and
// c / h are finished in registers.  h leaves as fp32 planes (the level's output sequence) and as the SB16 image the next
// step's convolution reads, assembled through LDS into 16-byte pieces.
//
// Operands as in conv_sb.h: SB16 [B][C/16][H][W][TERMS][16 ch], halo tiles staged by LDS-DMA at a pixel pitch of 32 TERMS + 16
// bytes (one tile per K part and stage); weights [row tile][chunk][tap][term][64 lanes][8] L2 -> registers, two taps ahead.
#pragma once
#include <hip/hip_runtime.h>
#include "conv_sb.h"
#include "lstm16.h"

namespace bde {

// 16-byte slots of a pixel in the step kernel's halo tiles (see the kernel): no pad slot in the two-term format
#ifndef LSB_SWZ
#define LSB_SWZ 1
#endif
// which shapes read their pixel fragments one unit ahead of the MFMAs (see the stage loop): where a wave takes all of K (level 0 of
// config A) it costs no register; with K split two ways (level 1) it cost 40 and the third workgroup of a CU; with K split four
// ways (level 2: two workgroups per CU) the 30 registers it costs are there
#ifndef LSB_PIPE
#define LSB_PIPE(KW) ((KW) != 2)
#endif
__host__ __device__ constexpr int lsb_slots(int terms) { return (terms == 2 && LSB_SWZ) ? 4 : sb_lds_slots(terms); }

struct LstmSbArgs {
    const unsigned char* hin;       // SB16 image of h_prev, group g, frame n: hin + g * hin_gs + n * hin_ns (bytes)
    long hin_gs, hin_ns;
    const unsigned short* wpk;      // split weights [G][row tile][chunk][tap][terms][64][8]
    long w_gs;                      // in bf16 elements
    const float* gx;                // x-part of the gates incl. bias, gate-major [4 Ch][HW]: gx + g * gx_gs + n * gx_ns
    long gx_gs, gx_ns;
    // ... or the x-part contracted HERE (xchunks > 0): K = [x | h] as in the reference's stacked input (submodules.py:316-317),
    // x's SB16 image (written by the encoder convolution's epilogue) staged like h's, the gates' bias added in the tail; gx unused
    const unsigned char* xin;       // SB16 image of x_t, group g, frame n: xin + g * xin_gs + n * xin_ns (bytes)
    long xin_gs, xin_ns;
    int xchunks;                    // 16-channel chunks of x (= Ch / 16), 0 = gx mode
    const float* bias;              // [G][4 Ch] gate-major (xchunks > 0)
    float* cstate;                  // [G][B][Ch][HW], updated in place
    long c_gs, c_ns;
    float* hout;                    // fp32 planes [Ch][HW]: hout + g * ho_gs + n * ho_ns
    long ho_gs, ho_ns;
    unsigned char* hsb;             // SB16 image of h (the next step's hin), same strides as hin
    const float* zeros;             // >= 16 bytes of zeros (source of out-of-image halo pixels)
    int B, Ch, H, W;
    int first;                      // 1: h_prev = c_prev = 0, no contraction
    int terms;                      // split format of hin / hsb / wpk (split.h)
    const float* acc_scale;         // two-term weights: -> the inverse of the power of two they were packed with (packed image)
    int TR, TC, tiles_x;            // a workgroup's pixels: TR image rows x TC columns (TR * TC <= NT * 32), tiles per row
    unsigned long long* stamps;     // diagnostics only: s_memtime per phase, [block < 64][wave < 4][8]
    int stamp_mode;                 // 1: instead, s_memrealtime start / end of every workgroup < 1000 ([wg][2])
};
#ifndef LSB_PF1
#define LSB_PF1 5
#endif
#ifndef LSB_PFK
#define LSB_PFK 8
#endif
#ifndef LSB_COUNTED_WAIT
#define LSB_COUNTED_WAIT 1
#endif
#ifndef LSB_STAGGER
#define LSB_STAGGER 0
#endif
// timing experiments only (results are wrong): 1 = no weight-fragment loads in the tap loop, 2 = pixel fragments of the first unit only
#ifndef LSB_DBG
#define LSB_DBG 0
#endif
#define LSB_STAMP(i)                                                                              \
    do {                                                                                          \
        if (a.stamps && !a.stamp_mode && lane == 0 && blockIdx.x < 64 && blockIdx.y == 0 && blockIdx.z == 0)       \
            a.stamps[(blockIdx.x * 4 + wave) * 8 + (i)] = __builtin_amdgcn_s_memtime();           \
    } while (0)

// DB: two sets of halo tiles; stage s + 1 is requested before the MFMAs of stage s (where three workgroups' worth of LDS allows:
// the workgroups of a launch run in lockstep, so their halo waits coincide and nobody covers them).
template <int RTW, int KW, int NT, int MAXI, bool DB, int TERMS>
__global__ __launch_bounds__(256, 2) void lstm_sb_step_kernel(const LstmSbArgs a) {
    static_assert(RTW * KW == 4, "four waves per workgroup");
    // Halo tiles in LDS.  Three terms: a pixel's 16-byte slots (two per term) + one pad slot, the pitch conv_sb.h uses.  Two terms:
    // NO pad slot -- 64 bytes per pixel, 20 % less LDS, which is what lets the tiles of level 1 be double-buffered with three
    // workgroups on a CU -- and the four slots of pixel p stored at slot ^ ((p >> 2) & 3) instead: sixteen lanes reading the same
    // logical slot of sixteen consecutive pixels then still hit sixteen different 16-byte bank groups (4 (p & 3) + (slot ^ (p >> 2 & 3))).
    constexpr bool SWZ = TERMS == 2 && LSB_SWZ;
    constexpr int SB_PIX_BYTES = sb_pix_bytes(TERMS), SLOTS = lsb_slots(TERMS), SB_LDS_PITCH = 16 * SLOTS;
    constexpr int QW = 4 / KW;                          // register groups (= hidden channels x 2) a wave finishes per tile
    extern __shared__ __align__(16) unsigned char lsb[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rtl = wave / KW, kp = wave - rtl * KW;    // this wave's row tile inside the workgroup and its part of K
    const int hl = lane >> 5;
    // Workgroup -> (pixel tile, row-tile group, direction x frame).  The dispatcher deals consecutive workgroups round-robin to
    // the 8 XCDs; in grid order the pixel tiles of one row tile -- which stream the same weights -- would sit in eight different
    // L2s and each would pull the slab from memory (level 2: 227 MB fetched per step for 28 MB of weights).  Each XCD gets a
    // contiguous range of the (frame, row tile, pixel tile) order instead, pixel tile fastest.
    int bx, by, z;
    {
        const unsigned gx = gridDim.x, gy = gridDim.y;
        const unsigned total = gx * gy * gridDim.z;
        const unsigned lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
        const unsigned xcd = lin & 7u, q = total >> 3, rem = total & 7u;
        const unsigned L = xcd * q + min(xcd, rem) + (lin >> 3);
        bx = (int)(L % gx);
        const unsigned r = L / gx;
        by = (int)(r % gy);
        z = (int)(r / gy);
    }
    const int g = z / a.B, n = z - g * a.B;
    const int HW = a.H * a.W;
    const int C16 = a.Ch / 16, nrt = a.Ch / 8;          // 16-channel chunks of h; 32-row tiles (8 hidden channels each)
    const int XC = a.xchunks;                           // chunks of x ahead of them in K
    const int KC = XC + C16;                            // chunks per row tile in the packed weights
    const int rt = by * RTW + rtl;
    const bool rt_live = rt < nrt;
    const int ty = bx / a.tiles_x, tx = bx - ty * a.tiles_x;
    const int y0 = ty * a.TR, x0 = tx * a.TC;
    const int tc = min(a.TC, a.W - x0), tr = min(a.TR, a.H - y0);   // live extent of this tile
    const int IW = a.TC + 2, IR = a.TR + 2;
    if (LSB_STAGGER) {
        // The workgroups of a launch start together and meet the same barriers at the same cadence, so the two or three that share
        // a CU wait for their halo tiles at the same time and nobody computes meanwhile.  The dispatch rounds (one workgroup per
        // CU each) start a third of a stage apart instead.
        const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const unsigned round = (lin >> 8) % 3u;
        for (unsigned i = 0; i < round; ++i) __builtin_amdgcn_s_sleep(LSB_STAGGER);         // LSB_STAGGER x 64 cycles per round
    }
    LSB_STAMP(0);
    const unsigned long long real0 = a.stamps ? __builtin_amdgcn_s_memrealtime() : 0ull;   // (diagnostics: 100 MHz reference clock)

    // this lane's pixel of each 32-pixel tile: q = t * 32 + (lane & 31) -> (row q / TC, column q % TC)
    // (computed BEHIND the first halo request and the fragment prefetch -- lane_setup() below -- so that the DMA's flight covers it)
    const float inv_tc = 1.0f / (float)a.TC;
    int boff[NT], pix[NT];
    f32x16 acc[NT];
    auto lane_setup = [&]() {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int q = t * 32 + (lane & 31);
            const int py = (int)(((float)q + 0.5f) * inv_tc), px = q - py * a.TC;
            const bool ok = py < tr && px < tc;
            pix[t] = ok ? (y0 + py) * a.W + x0 + px : -1;
            const int cy = min(py, a.TR - 1), cx = min(px, a.TC - 1);   // (dead lanes read inside the tile)
            boff[t] = SWZ ? cy * IW + cx : (cy * IW + cx) * SB_LDS_PITCH + hl * 16;      // (swizzled: the pixel index, see the tap loop)
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    };
    // (the accumulator scale is requested HERE, not in the tail where it is used: there it was a dependent load + vmcnt(0))
    const float unscale_v = TERMS == 2 ? a.acc_scale[0] : 1.f;

    // pointwise operands of the (channel, pixel)s this wave finishes: register groups q = kp * QW .. + QW - 1 of its row tile
    float gv[NT][QW][4], cprev[NT][QW];
    const float* gxb = XC > 0 ? a.bias + (long)g * 4 * a.Ch : a.gx + g * a.gx_gs + n * a.gx_ns;
    float* cst = a.cstate + g * a.c_gs + n * a.c_ns;
    auto epi_load = [&]() {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int qq = 0; qq < QW; ++qq) {
                const int hc = min(rt * 8 + 4 * hl + kp * QW + qq, a.Ch - 1);
                const long o = (long)hc * HW + max(pix[t], 0);
#pragma unroll
                for (int gate = 0; gate < 4; ++gate) gv[t][qq][gate] = XC > 0 ? gxb[gate * a.Ch + hc] : gxb[(long)gate * a.Ch * HW + o];
                cprev[t][qq] = a.first ? 0.f : cst[o];
            }
    };

    const int kchunks = XC + (a.first ? 0 : C16);        // chunks of K this step contracts (h_prev = 0 at the first step)
    const int stages = kchunks / KW;                     // chunks per part of K
    const int nslots = IR * IW * SLOTS;                  // 16-byte slots of one halo tile (2 per term + the pad slot per pixel)
    const int nblk = (nslots + 63) >> 6;                 // 1-KiB DMA blocks of one tile
    const int tile_bytes = nblk * 1024;
    if (kchunks == 0) {
        lane_setup();
        epi_load();
    } else {
        const unsigned char* inb = a.hin + g * a.hin_gs + n * a.hin_ns;
        const unsigned char* xb = XC > 0 ? a.xin + g * a.xin_gs + n * a.xin_ns : inb;
        const long plane = (long)HW * SB_PIX_BYTES;
        // halo staging: KW tiles per stage (one per part of K), KW * nblk blocks dealt to the four waves
        unsigned goff[MAXI];
        const float inv_iw = 1.0f / (float)IW, inv_nblk = 1.0f / (float)nblk;
        unsigned vmask = 0;
        unsigned long long pmask = 0;                    // two bits per block of this wave = the part of K the block belongs to
#pragma unroll
        for (int it = 0; it < MAXI; ++it) {
            const int blk = wave + it * 4;
            const int part = (int)(((float)blk + 0.5f) * inv_nblk), b_in = blk - part * nblk;
            const int i = b_in * 64 + lane;
            const int px = i / SLOTS, qs = i - px * SLOTS;
            const int r = (int)(((float)px + 0.5f) * inv_iw), c = px - r * IW;      // (exact for px < 2^20; an integer division by a
            const int iy = y0 - 1 + r, ix = x0 - 1 + c;                           //  run-time divisor is ~40 instructions, MAXI times)
            const int piece = SWZ ? (qs ^ ((px >> 2) & 3)) : qs;                  // the 16-byte piece of the pixel that lives in this slot
            const bool ok = part < KW && i < nslots && (SWZ || qs < SLOTS - 1) && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            goff[it] = ok ? (unsigned)((iy * a.W + ix) * SB_PIX_BYTES + piece * 16) : 0u;
            if (ok) vmask |= 1u << it;
            pmask |= (unsigned long long)(part & 3) << (2 * it);
        }
        const unsigned char* zero16 = reinterpret_cast<const unsigned char*>(a.zeros);
        const int set_bytes = KW * tile_bytes;            // one set of halo tiles (all parts of K)
        auto stage = [&](int s) {
            const unsigned dst = sb_dyn_lds_base() + (DB ? (s & 1) * set_bytes : 0);
#pragma unroll
            for (int it = 0; it < MAXI; ++it) {
                const int blk = wave + it * 4;
                if (blk < KW * nblk) {
                    const int part = (int)((pmask >> (2 * it)) & 3ull);
                    const int chunk = part * stages + s;
                    const unsigned char* cb = chunk < XC ? xb + (long)chunk * plane : inb + (long)(chunk - XC) * plane;
                    const unsigned char* src = ((vmask >> it) & 1u) ? cb + goff[it] : zero16;
                    sb_lds_dma16(src, dst + blk * 1024);       // (asm: invisible to the compiler's wait counting, split.h)
                }
            }
        };
        // weight fragments of this wave: row tile rt, chunks kp * stages .. + stages - 1 (consecutive in memory)
        // Weight fragments run PF taps ahead of the MFMAs that use them.  The ring has one position per tap of a stage (the position
        // of a tap is a compile-time constant), of which PF + 1 are live at a time.  Two taps ahead (round 3) left the matrix pipe
        // waiting for L2: in-kernel, level 0, the stage loop went 54.9 k -> 40.6 k cycles with the whole stage ahead -- 91 % matrix
        // duty -- but 72 registers of fragments cost the third workgroup per CU (692 workgroups: a second dispatch round, 61 us
        // instead of 54).  PF = 5 keeps three workgroups per CU where a wave takes all of K (KW = 1), PF = 8 elsewhere.
        constexpr int TAPS = 9, RING = 9, PF = KW == 1 ? LSB_PF1 : LSB_PFK;
        static_assert(PF >= 1 && PF < RING, "");
        const int S = stages * TAPS;
        const sb8* wfr = reinterpret_cast<const sb8*>(a.wpk + g * a.w_gs) +
                         (((long)min(rt, nrt - 1) * KC + (long)kp * stages) * TAPS * TERMS) * 64 + lane;
        sb8 af[RING][TERMS];
        if (DB) stage(0);                                // BEFORE the fragment prefetch: the counted wait below relies on the DMA being older
#pragma unroll
        for (int q = 0; q < PF; ++q)
#pragma unroll
            for (int k = 0; k < TERMS; ++k) af[q][k] = wfr[((long)min(q, S - 1) * TERMS + k) * 64];
        __builtin_amdgcn_sched_barrier(0);
        lane_setup();
        LSB_STAMP(1);
        for (int s = 0; s < stages; ++s) {
            if (!DB) {
                __syncthreads();                         // every wave is done with the previous stage's tiles
                stage(s);
            }
            if (LSB_COUNTED_WAIT && DB) {
                // The tiles of stage s were requested before the last PF taps of weight fragments, which are all that may still be in
                // flight: the counted wait covers the tiles (loads return in order) and leaves the ring alone; the barrier as asm
                // (memory clobber: LDS reads may not cross it)
                asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PF * TERMS) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");   // stage s has landed for every wave (and, DB: stage s - 1's tiles are free)
            }
            if (DB && s + 1 < stages) stage(s + 1);
            if (s == stages - 1) epi_load();             // the tail's operands travel during the last stage
            const unsigned char* tile = lsb + (DB ? (s & 1) * set_bytes : 0) + kp * tile_bytes;
            if (s == 0) LSB_STAMP(2);
            if constexpr (SWZ) {
                // (the swizzled fragment addresses of the nine taps are a few vector instructions each; hoisted out of the stage
                //  loop -- they do not depend on the stage -- they are 18 NT registers and the third workgroup of a CU)
#pragma unroll
                for (int t = 0; t < NT; ++t) asm volatile("" : "+v"(boff[t]));
            }
            // The pixel fragments of unit u + 1 (a unit = one tap of one 32-pixel tile: two 16-byte LDS reads, three dependent MFMAs
            // = 96 cycles) are read BEFORE the MFMAs of unit u: left to the compiler every unit was read, waited for
            // (s_waitcnt lgkmcnt(0)) and then multiplied, an LDS round trip in front of every three MFMAs.
            auto read_unit = [&](int u, sb8 (&b)[TERMS]) {
                const int tap = u / NT, t = u - tap * NT;
                const int ky = tap / 3, kx = tap - ky * 3;
                if constexpr (SWZ) {
                    const int p = boff[t] + ky * IW + kx;                            // pixel of the tile this lane reads for this tap
                    const int o0 = p * 64 + ((hl ^ ((p >> 2) & 3)) << 4);            // term 0 = pieces hl, term 1 = pieces 2 + hl: slot ^ 2
                    b[0] = *reinterpret_cast<const sb8*>(tile + o0);
                    b[1] = *reinterpret_cast<const sb8*>(tile + (o0 ^ 32));
                } else {
#pragma unroll
                    for (int k = 0; k < TERMS; ++k)
                        b[k] = *reinterpret_cast<const sb8*>(tile + boff[t] + (ky * IW + kx) * SB_LDS_PITCH + k * 32);
                }
            };
            if constexpr (!LSB_PIPE(KW)) {
                // (as the compiler orders it: every tap's fragments read, waited for, multiplied)
#pragma unroll
                for (int tap = 0; tap < TAPS; ++tap) {
                    const int nxt = (tap + PF) % RING;
                    const long sp = min(s * TAPS + tap + PF, S - 1);
#pragma unroll
                    for (int k = 0; k < TERMS; ++k) af[nxt][k] = wfr[(sp * TERMS + k) * 64];
                    sb8 bfr[NT][TERMS];
#pragma unroll
                    for (int t = 0; t < NT; ++t) read_unit(tap * NT + t, bfr[t]);
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = sb_mma32<TERMS>(af[tap % RING], bfr[t], acc[t]);
                }
                continue;
            }
            sb8 bb[2][TERMS];
            read_unit(0, bb[0]);
#pragma unroll
            for (int u = 0; u < TAPS * NT; ++u) {
                const int tap = u / NT, t = u - tap * NT;
                if (t == 0 && !(LSB_DBG & 1)) {
                    const int nxt = (tap + PF) % RING;
                    const long sp = min(s * TAPS + tap + PF, S - 1);
#pragma unroll
                    for (int k = 0; k < TERMS; ++k) af[nxt][k] = wfr[(sp * TERMS + k) * 64];
                }
                if (u + 1 < TAPS * NT && !(LSB_DBG & 2)) read_unit(u + 1, bb[(u + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                acc[t] = sb_mma32<TERMS>(af[tap % RING], bb[(LSB_DBG & 2) ? 0 : (u & 1)], acc[t]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        LSB_STAMP(3);
        __syncthreads();                                 // the halo tiles are dead: LDS becomes the reduction / output staging area
    }
    LSB_STAMP(4);

    // ---- sum over the parts of K: every wave leaves its accumulators in LDS, wave (row tile, kp) collects register groups
    //      kp * QW .. of its row tile from the KW waves of that row tile -----------------------------------------------------
    float gsum[NT][QW][4];
    // two-term weights are packed times a power of two (readfirstlane: into a scalar register)
    const float unscale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, unscale_v)));
    if (KW > 1 && kchunks > 0) {
        float* red = reinterpret_cast<float*>(lsb);      // [wave][tile][reg][64 lanes]
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((wave * NT + t) * 16 + r) * 64 + lane] = acc[t][r];
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int qq = 0; qq < QW; ++qq)
#pragma unroll
                for (int gate = 0; gate < 4; ++gate) {
                    float s = 0.f;
#pragma unroll
                    for (int p = 0; p < KW; ++p) s += red[(((rtl * KW + p) * NT + t) * 16 + 4 * (kp * QW + qq) + gate) * 64 + lane];
                    gsum[t][qq][gate] = TERMS == 2 ? s * unscale : s;
                }
        __syncthreads();
    } else {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int qq = 0; qq < QW; ++qq)
#pragma unroll
                for (int gate = 0; gate < 4; ++gate) gsum[t][qq][gate] = TERMS == 2 ? acc[t][4 * (kp * QW + qq) + gate] * unscale : acc[t][4 * (kp * QW + qq) + gate];
    }

    LSB_STAMP(5);
    // ---- pointwise tail (submodules.py:320-332) and the outputs -----------------------------------------------------------------
    // h as SB16 goes through LDS: [row tile of the workgroup][pixel of the tile][term][8 channels] bf16, 16-byte pieces out
    unsigned short* hst = reinterpret_cast<unsigned short*>(lsb);
    float* hob = a.hout + g * a.ho_gs + n * a.ho_ns;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        unsigned short t3[QW][TERMS];
        const bool live = rt_live && pix[t] >= 0;
#pragma unroll
        for (int qq = 0; qq < QW; ++qq) {
            const int hc = rt * 8 + 4 * hl + kp * QW + qq;   // hidden channel: 4 hl + (register group) inside the row tile
            const float vi = gsum[t][qq][0] + gv[t][qq][0], vf = gsum[t][qq][1] + gv[t][qq][1];
            const float vo = gsum[t][qq][2] + gv[t][qq][2], vg = gsum[t][qq][3] + gv[t][qq][3];
            const float c = sigmoid_fast(vf) * cprev[t][qq] + sigmoid_fast(vi) * tanh_fast(vg);
            const float h = sigmoid_fast(vo) * tanh_fast(c);
            if (live) {
                const long o = (long)hc * HW + pix[t];
                cst[o] = c;
                hob[o] = h;
            }
            sb_split_dev<TERMS>(live ? h : 0.f, t3[qq]);
        }
        const int q = t * 32 + (lane & 31);
        unsigned short* d = hst + ((rtl * (NT * 32) + q) * TERMS) * 8 + 4 * hl + kp * QW;
#pragma unroll
        for (int k = 0; k < TERMS; ++k) {
            if (QW == 4) *reinterpret_cast<uint2*>(d + k * 8) = uint2{t3[0][k] | ((unsigned)t3[1][k] << 16), t3[2][k] | ((unsigned)t3[3][k] << 16)};
            else if (QW == 2) *reinterpret_cast<unsigned*>(d + k * 8) = t3[0][k] | ((unsigned)t3[QW - 1][k] << 16);
            else d[k * 8] = t3[0][k];
        }
    }
    __syncthreads();
    {
        unsigned char* hsbo = a.hsb + g * a.hin_gs + n * a.hin_ns;
        const int pieces = RTW * NT * 32 * TERMS;            // 16-byte pieces: (row tile, pixel, term)
        for (int i = tid; i < pieces; i += 256) {
            const int k = i % TERMS, r2 = i / TERMS;
            const int q = r2 % (NT * 32), rl = r2 / (NT * 32);
            const int py = q / a.TC, px = q - py * a.TC;
            const int rtt = by * RTW + rl;
            if (py >= tr || px >= tc || rtt >= nrt) continue;
            const long p = (long)(y0 + py) * a.W + x0 + px;
            const uint4 v = *reinterpret_cast<const uint4*>(hst + ((rl * (NT * 32) + q) * TERMS + k) * 8);
            // chunk rtt / 2, pixel p, term k, half (rtt & 1) of the 16 channels
            *reinterpret_cast<uint4*>(hsbo + (((long)(rtt >> 1) * HW + p) * TERMS + k) * 32 + (rtt & 1) * 16) = v;
        }
    }
    LSB_STAMP(6);
    if (a.stamps && !a.stamp_mode && lane == 0 && blockIdx.x < 64 && blockIdx.y == 0 && blockIdx.z == 0)
        a.stamps[(blockIdx.x * 4 + wave) * 8 + 7] = __builtin_amdgcn_s_memrealtime() - real0;
    if (a.stamps && a.stamp_mode && lane == 0 && wave == 0) {
        const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (lin < 1000) { a.stamps[2 * lin] = real0; a.stamps[2 * lin + 1] = __builtin_amdgcn_s_memrealtime(); }
    }
    if (a.stamps && !a.stamp_mode && lane == 0 && wave == 0) {            // diagnostics: first start / last end over ALL workgroups (100 MHz ticks)
        atomicMin(a.stamps + 2046, real0);
        atomicMax(a.stamps + 2047, (unsigned long long)__builtin_amdgcn_s_memrealtime());
    }
}

// ---- host side --------------------------------------------------------------------------------------------------------------
struct LstmSbShape { int rtw, kw, nt, TR, TC, tiles_x, tiles_y, maxi; size_t lds; bool ok, db; };

// How a level's step is cut (see the header): KW parts of K per workgroup by the number of 16-channel chunks, NT by the map width.
// GB = directions x batch of the launch (sizes the grid: how many workgroups a CU has to hold at once)
static inline LstmSbShape lstm_sb_shape_kw(int Ch, int H, int W, int kw, int terms, int GB = 2) {
    LstmSbShape s{};
    s.ok = false;
    const int C16 = Ch / 16;
    if (C16 % kw != 0) return s;
    s.kw = kw;
    s.rtw = 4 / s.kw;
    s.nt = s.kw == 4 ? 3 : 2;
    const int px = s.nt * 32;
    if (W >= px) { s.TR = 1; s.TC = px; }                 // row segments
    else { s.TC = W; s.TR = std::max(1, std::min(px / W, H)); }
    s.tiles_x = cdiv(W, s.TC);
    s.tiles_y = cdiv(H, s.TR);
    const long halo = (long)(s.TR + 2) * (s.TC + 2);
    const long nblk = (halo * lsb_slots(terms) + 63) / 64;
    const long blocks = nblk * s.kw;
    s.maxi = (int)((blocks + 3) / 4);
    const size_t stage_b = (size_t)blocks * 1024;
    const size_t red_b = s.kw > 1 ? (size_t)4 * s.nt * 16 * 64 * 4 : 0;
    const size_t hst_b = (size_t)s.rtw * s.nt * 32 * terms * 16;
    // double-buffered halo tiles while every workgroup of the launch is still resident: three per CU, or two where the grid has
    // at most 512 workgroups (level 2 of config A: 2 x 80 KB = the whole LDS of a CU, nothing static in the kernel)
    const long wgs = (long)s.tiles_x * s.tiles_y * cdiv(Ch / 8, s.rtw) * GB;
    s.db = wgs <= 512 ? 2 * stage_b * 2 <= 160 * 1024 : 2 * stage_b * 3 <= 150 * 1024;
    s.lds = std::max((s.db ? 2 : 1) * stage_b, std::max(red_b, hst_b));
    s.ok = s.maxi <= 24 && s.lds <= 80 * 1024;            // (two workgroups per CU at least)
    return s;
}
static inline LstmSbShape lstm_sb_shape(int Ch, int H, int W, int terms, int GB = 2) {
    LstmSbShape none{};
    none.ok = false;
    if (Ch % 16 != 0 || Ch < 16) return none;
    const int C16 = Ch / 16;
    // the more chunks of K, the more of them a workgroup's waves split; a map whose rows do not fill the wider tile of the
    // four-way split (LDS of its four halo tiles) falls back to the next shape
    for (int kw = C16 >= 16 ? 4 : (C16 >= 8 ? 2 : 1); kw >= 1; kw >>= 1) {
        const LstmSbShape s = lstm_sb_shape_kw(Ch, H, W, kw, terms, GB);
        if (s.ok) return s;
    }
    return none;
}

#ifdef BDE_SB_TU
template <int RTW, int KW, int NT, int MAXI, bool DB, int TERMS>
static int lstm_sb_launch_t(const LstmSbArgs& a, const LstmSbShape& s, int G, hipStream_t stream) {
    auto kern = lstm_sb_step_kernel<RTW, KW, NT, MAXI, DB, TERMS>;
    static unsigned char raised[BDE_MAX_DEVICES];
    if (s.lds > 64 * 1024) BDE_HIP(raise_dynamic_lds(raised, (const void*)kern));
    dim3 grid(s.tiles_x * s.tiles_y, cdiv(a.Ch / 8, RTW), G * a.B);
    hipLaunchKernelGGL(kern, grid, dim3(256), s.lds, stream, a);
    BDE_HIP(hipGetLastError());
    return BDE_OK;
}
template <int RTW, int KW, int NT, int TERMS>
static int lstm_sb_launch_m(const LstmSbArgs& a, const LstmSbShape& s, int G, hipStream_t stream) {
    if (s.db) {
        if (s.maxi <= 6) return lstm_sb_launch_t<RTW, KW, NT, 6, true, TERMS>(a, s, G, stream);
        if (s.maxi <= 12) return lstm_sb_launch_t<RTW, KW, NT, 12, true, TERMS>(a, s, G, stream);
        return lstm_sb_launch_t<RTW, KW, NT, 24, true, TERMS>(a, s, G, stream);
    }
    if (s.maxi <= 6) return lstm_sb_launch_t<RTW, KW, NT, 6, false, TERMS>(a, s, G, stream);
    if (s.maxi <= 12) return lstm_sb_launch_t<RTW, KW, NT, 12, false, TERMS>(a, s, G, stream);
    return lstm_sb_launch_t<RTW, KW, NT, 24, false, TERMS>(a, s, G, stream);
}
// the kernel lstm_sb_step_launch takes for a shape (occupancy queries)
template <int RTW, int KW, int NT, int TERMS>
static const void* lstm_sb_ptr_m(const LstmSbShape& s) {
    if (s.db) {
        if (s.maxi <= 6) return (const void*)lstm_sb_step_kernel<RTW, KW, NT, 6, true, TERMS>;
        if (s.maxi <= 12) return (const void*)lstm_sb_step_kernel<RTW, KW, NT, 12, true, TERMS>;
        return (const void*)lstm_sb_step_kernel<RTW, KW, NT, 24, true, TERMS>;
    }
    if (s.maxi <= 6) return (const void*)lstm_sb_step_kernel<RTW, KW, NT, 6, false, TERMS>;
    if (s.maxi <= 12) return (const void*)lstm_sb_step_kernel<RTW, KW, NT, 12, false, TERMS>;
    return (const void*)lstm_sb_step_kernel<RTW, KW, NT, 24, false, TERMS>;
}
static const void* lstm_sb_kernel_ptr(const LstmSbShape& s, int terms) {
    if (terms == 2) return s.kw == 4 ? lstm_sb_ptr_m<1, 4, 3, 2>(s) : s.kw == 2 ? lstm_sb_ptr_m<2, 2, 2, 2>(s) : lstm_sb_ptr_m<4, 1, 2, 2>(s);
    return s.kw == 4 ? lstm_sb_ptr_m<1, 4, 3, 3>(s) : s.kw == 2 ? lstm_sb_ptr_m<2, 2, 2, 3>(s) : lstm_sb_ptr_m<4, 1, 2, 3>(s);
}
int lstm_sb_step_launch(LstmSbArgs a, int G, hipStream_t stream) {
    // (the shape is chosen as for both directions, whatever this launch covers: bde_split_sweep computes what the joint launch does)
    const LstmSbShape s = lstm_sb_shape(a.Ch, a.H, a.W, a.terms, 2 * a.B);
    if (!s.ok) return fail(BDE_ERR_UNSUPPORTED, "split recurrent step: no shape for %d channels on a %dx%d map", a.Ch, a.H, a.W);
    a.TR = s.TR; a.TC = s.TC; a.tiles_x = s.tiles_x;
    if (a.terms == 2) {
        if (s.kw == 4) return lstm_sb_launch_m<1, 4, 3, 2>(a, s, G, stream);
        if (s.kw == 2) return lstm_sb_launch_m<2, 2, 2, 2>(a, s, G, stream);
        return lstm_sb_launch_m<4, 1, 2, 2>(a, s, G, stream);
    }
    if (s.kw == 4) return lstm_sb_launch_m<1, 4, 3, 3>(a, s, G, stream);
    if (s.kw == 2) return lstm_sb_launch_m<2, 2, 2, 3>(a, s, G, stream);
    return lstm_sb_launch_m<4, 1, 2, 3>(a, s, G, stream);
}
#else
int lstm_sb_step_launch(LstmSbArgs a, int G, hipStream_t stream);   // sb_tu.hip
#endif

}  // namespace bde
