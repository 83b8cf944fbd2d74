// libbde2vid.so -- C ABI (include/bde2vid.h), weight packing and the forward schedule.
//
// The schedule restates BDE2VIDCrossscalePropogationV5.forward (V5.py:100-241) as a sequence of
// launches of three hand-written gfx950 kernels (conv_mfma.h, attn.h, small element-wise ones):
//   head conv (all T) -> per level { encoder conv x2 dirs (all T), gate x-part conv (all T),
//   T recurrent ConvLSTM steps (both directions per launch), merge, temporal window attention
//   (sequential in t, V5.py:154-169) } -> decoder (all T) -> predI + sigmoid (all T).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/bde2vid.h"
#include "attn.h"
#include "common.h"
#include "conv_mfma.h"
#include "pw_gemm.h"
#include "token_fused.h"
#include "lstm16.h"
#include "lstm_sb.h"
#include "winblock.h"
#include "winblock_sb.h"
#include "wideblock.h"
#include "wide_mlp.h"
#include "wide_core.h"
#include "attn_mfma.h"
#include "conv_vec.h"
#include "conv_sb.h"
#include "voxel.h"
#include "metrics.h"

namespace bde {

int conv_tu_occupancy(const char* kernel);   // conv_tu.hip

std::string& last_error_ref() {
    static thread_local std::string e;
    return e;
}
int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    last_error_ref() = buf;
    return code;
}

// ------------------------------------------------------------------------------------------
// small element-wise kernels
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void add2_kernel(const float4* __restrict__ a, const float4* __restrict__ b,
                                                   float4* __restrict__ o, long n4) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 x = a[i], y = b[i];
        o[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    }
}
__global__ __launch_bounds__(256) void add2_tail_kernel(const float* a, const float* b, float* o, long beg, long n) {
    long i = beg + blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i < n) o[i] = a[i] + b[i];
}
// The caller's T frame tensors <-> one contiguous [T][n] stack, one launch per direction (the frame
// pointers travel by value; 2T separate hipMemcpyAsync calls cost ~7 us each on the stream).
constexpr int FRAME_PTRS = 64;
struct FramePtrs { const float* p[FRAME_PTRS]; };
__global__ __launch_bounds__(256) void gather_frames_kernel(FramePtrs fp, float* __restrict__ dst, long n4, long n) {
    const float* src = fp.p[blockIdx.y];
    float* d = dst + (long)blockIdx.y * n;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
        reinterpret_cast<float4*>(d)[i] = reinterpret_cast<const float4*>(src)[i];
    if (blockIdx.x == 0)
        for (long i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) d[i] = src[i];
}
__global__ __launch_bounds__(256) void scatter_frames_kernel(FramePtrs fp, const float* __restrict__ srcs, long n4, long n) {
    float* d = const_cast<float*>(fp.p[blockIdx.y]);
    const float* src = srcs + (long)blockIdx.y * n;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
        reinterpret_cast<float4*>(d)[i] = reinterpret_cast<const float4*>(src)[i];
    if (blockIdx.x == 0)
        for (long i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) d[i] = src[i];
}
// dir 0: frames -> stack, 1: stack -> frames.  Frame pointers must be 16-byte aligned for the float4 path.
static int copy_frames(const float* const* frames, float* stack, int T, long n, int dir, hipStream_t s) {
    bool aligned = (n % 4 == 0);
    for (int t = 0; t < T && aligned; ++t) aligned = ((uintptr_t)frames[t] % 16) == 0;
    const long n4 = aligned ? n / 4 : 0;
    for (int t0 = 0; t0 < T; t0 += FRAME_PTRS) {
        const int nt = std::min(FRAME_PTRS, T - t0);
        FramePtrs fp;
        for (int k = 0; k < FRAME_PTRS; ++k) fp.p[k] = frames[t0 + std::min(k, nt - 1)];
        const unsigned bx = (unsigned)std::min<long>(std::max<long>(cdivl(std::max<long>(n4, n / 4), 256), 1), 256);
        if (dir == 0) hipLaunchKernelGGL(gather_frames_kernel, dim3(bx, nt), dim3(256), 0, s, fp, stack + (long)t0 * n, n4, n);
        else hipLaunchKernelGGL(scatter_frames_kernel, dim3(bx, nt), dim3(256), 0, s, fp, stack + (long)t0 * n, n4, n);
    }
    BDE_HIP(hipGetLastError());
    return BDE_OK;
}

static int add2(const float* a, const float* b, float* o, long n, hipStream_t s) {
    long n4 = n / 4;
    if (n4 > 0) {
        long blocks = std::min<long>(cdivl(n4, 256), 2048);
        hipLaunchKernelGGL(add2_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const float4*)a, (const float4*)b,
                           (float4*)o, n4);
    }
    if (n4 * 4 < n) hipLaunchKernelGGL(add2_tail_kernel, dim3(1), dim3(256), 0, s, a, b, o, n4 * 4, n);
    BDE_HIP(hipGetLastError());
    return BDE_OK;
}

// Bilinear x2 (align_corners=False) of (a + b): the input of UpsampleConvLayer's conv
// (submodules.py:138) with the skip_sum (V5.py:289-293) folded in.  src = dst/2 - 0.25 clamped at 0:
// even dst 2k -> 0.25*in[k-1] + 0.75*in[k]; odd dst 2k+1 -> 0.75*in[k] + 0.25*in[k+1]; edges clamp.
// One thread = two source columns of one source row -> a 2 x 4 block of outputs (two 16-byte stores); the 3 x 4
// source neighbourhood is read once (the one-output-per-thread form made 8 scalar loads and two integer
// divisions per output).  Same expression per output as the reference's bilinear weights.
__global__ __launch_bounds__(256) void upsample2x_sum_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                             float* __restrict__ out, int Hs, int Ws, long planes) {
    const int Wo = 2 * Ws, W2 = (Ws + 1) / 2;            // W2 column pairs per source row
    const long total = planes * Hs * W2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int jp = (int)(i % W2);
        const long t = i / W2;
        const int k = (int)(t % Hs);
        const long pl = t / Hs;
        const int j0 = 2 * jp;
        const float* pa = a + pl * Hs * Ws;
        const float* pb = b ? b + pl * Hs * Ws : nullptr;
        const int ym = max(k - 1, 0), yp = min(k + 1, Hs - 1);
        int xc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) xc[c] = min(max(j0 - 1 + c, 0), Ws - 1);
        float v[3][4];
        const int yr[3] = {ym, k, yp};
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float x = pa[yr[r] * Ws + xc[c]];
                if (pb) x += pb[yr[r] * Ws + xc[c]];
                v[r][c] = x;
            }
        float* ob = out + pl * 4 * Hs * Ws;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const int y = 2 * k + dy;
            // rows (ya, yb) and weights exactly as the per-output form: even y -> (k-1, k) with (0.25, 0.75),
            // odd y -> (k, k+1) with (0.75, 0.25); a clamped pair collapses to weight 1 on one row
            const int ra = dy == 0 ? 0 : 1, rb = dy == 0 ? 1 : 2;
            float wyb = dy ? 0.25f : 0.75f;
            if (yr[ra] == yr[rb]) wyb = 1.f;
            const float wya = 1.f - wyb;
            float o[4];
#pragma unroll
            for (int dx = 0; dx < 4; ++dx) {
                const int x = 2 * j0 + dx;               // output column; source pair (xa, xb) = columns ca, cb of v
                const int ca = (dx + 1) / 2, cb = ca + 1;   // dx 0: (j0-1, j0); 1: (j0, j0+1); 2: (j0, j0+1); 3: (j0+1, j0+2)
                float wxb = (x & 1) ? 0.25f : 0.75f;
                if (xc[ca] == xc[cb]) wxb = 1.f;
                const float wxa = 1.f - wxb;
                o[dx] = wya * (wxa * v[ra][ca] + wxb * v[ra][cb]) + wyb * (wxa * v[rb][ca] + wxb * v[rb][cb]);
            }
            float* op = ob + (long)y * Wo + 2 * j0;
            if (j0 + 1 < Ws && (Ws & 1) == 0) *reinterpret_cast<float4*>(op) = float4{o[0], o[1], o[2], o[3]};   // rows 16-byte aligned
            else {
                op[0] = o[0]; op[1] = o[1];
                if (j0 + 1 < Ws) { op[2] = o[2]; op[3] = o[3]; }
            }
        }
    }
}
// The same bilinear x2 of (a + b), stored as the SB16 image a split-bf16 decoder convolution reads (conv_sb.h) instead of fp32
// planes: [N][C/16][2Hs][2Ws][3 terms][16 channels] bf16.  grid (ceil(4 Hs Ws / 128), C/16 chunks, N), thread = (output
// pixel, half of the chunk): 2 x 2 source pixels of 8 channels of both tensors, three 16-byte stores.  Same expression and
// weights per output as upsample2x_sum_kernel.
template <int TERMS>
__global__ __launch_bounds__(256) void upsample2x_sum_split_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                   unsigned short* __restrict__ out, int C, int Hs, int Ws,
                                                                   unsigned* ovf) {
    const int Wo = 2 * Ws;
    const long HWo = 4L * Hs * Ws, HWs = (long)Hs * Ws;
    // (a wave = 64 consecutive pixels of one half: every plane load is one or two full 128-byte segments)
    const int half = threadIdx.x >> 7;
    const long p = (long)blockIdx.x * 128 + (threadIdx.x & 127);
    if (p >= HWo) return;
    const int c16 = blockIdx.y, C16 = gridDim.y;
    const long n = blockIdx.z;
    const int y = (int)(p / Wo), x = (int)(p - (long)y * Wo);
    const int k = y >> 1, j = x >> 1;
    const int ya = (y & 1) ? k : max(k - 1, 0), yb = (y & 1) ? min(k + 1, Hs - 1) : k;
    const int xa = (x & 1) ? j : max(j - 1, 0), xb = (x & 1) ? min(j + 1, Ws - 1) : j;
    float wyb = (y & 1) ? 0.25f : 0.75f, wxb = (x & 1) ? 0.25f : 0.75f;
    if (ya == yb) wyb = 1.f;
    if (xa == xb) wxb = 1.f;
    const float wya = 1.f - wyb, wxa = 1.f - wxb;
    const int iaa = ya * Ws + xa, iab = ya * Ws + xb, iba = yb * Ws + xa, ibb = yb * Ws + xb;
    unsigned short t[8][TERMS];
    float gm = 0.f;                                        // range guard of the two-term format (split.h)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int c = c16 * 16 + half * 8 + q;
        float o = 0.f;
        if (c < C) {
            const float* pa = a + (n * C + c) * HWs;
            float vaa = pa[iaa], vab = pa[iab], vba = pa[iba], vbb = pa[ibb];
            if (b) {
                const float* pb = b + (n * C + c) * HWs;
                vaa += pb[iaa]; vab += pb[iab]; vba += pb[iba]; vbb += pb[ibb];
            }
            o = wya * (wxa * vaa + wxb * vab) + wyb * (wxa * vba + wxb * vbb);
        }
        if (TERMS == 2) gm = sb_guard_max(gm, o);
        sb_split_dev<TERMS>(o, t[q]);
    }
    if (TERMS == 2) sb_guard_flush(gm, ovf);
    unsigned short* d = out + (((n * C16 + c16) * HWo + p) * TERMS) * 16 + half * 8;
#pragma unroll
    for (int kk = 0; kk < TERMS; ++kk) {
        uint4 v;
        v.x = t[0][kk] | ((unsigned)t[1][kk] << 16);
        v.y = t[2][kk] | ((unsigned)t[3][kk] << 16);
        v.z = t[4][kk] | ((unsigned)t[5][kk] << 16);
        v.w = t[6][kk] | ((unsigned)t[7][kk] << 16);
        *reinterpret_cast<uint4*>(d + kk * 16) = v;
    }
}
static int upsample2x_sum_split(const float* a, const float* b, void* out_sb, int N, int C, int Hs, int Ws, int terms, unsigned* ovf,
                                hipStream_t s) {
    const dim3 grid((unsigned)cdivl(4L * Hs * Ws, 128), cdiv(C, 16), (unsigned)N);
    if (terms == 2) hipLaunchKernelGGL(upsample2x_sum_split_kernel<2>, grid, dim3(256), 0, s, a, b, (unsigned short*)out_sb, C, Hs, Ws, ovf);
    else hipLaunchKernelGGL(upsample2x_sum_split_kernel<3>, grid, dim3(256), 0, s, a, b, (unsigned short*)out_sb, C, Hs, Ws, ovf);
    BDE_HIP(hipGetLastError());
    return BDE_OK;
}
static int upsample2x_sum(const float* a, const float* b, float* out, int Hs, int Ws, long planes, hipStream_t s) {
    const long total = planes * Hs * ((Ws + 1) / 2);
    long blocks = std::min<long>(cdivl(total, 256), 8192);
    hipLaunchKernelGGL(upsample2x_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, b, out, Hs, Ws, planes);
    BDE_HIP(hipGetLastError());
    return BDE_OK;
}

// predI (1x1 conv C->1) on (x + head) followed by the output activation (V5.py:195-197).
__global__ __launch_bounds__(256) void pred_kernel(const float* __restrict__ x, const float* __restrict__ head,
                                                   const float* __restrict__ w, const float* __restrict__ bias,
                                                   float* __restrict__ out, int C, long HW, long total, int sigmoid) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long n = i / HW, p = i - n * HW;
        const float* xb = x + n * C * HW + p;
        const float* hb = head ? head + n * C * HW + p : nullptr;
        float acc = 0.f;
        for (int c = 0; c < C; ++c) {
            float v = xb[c * HW];
            if (hb) v += hb[c * HW];
            acc += w[c] * v;
        }
        acc += bias[0];
        out[i] = sigmoid ? 1.f / (1.f + expf(-acc)) : acc;
    }
}

// ConvGRU step, element-wise halves (submodules.py:368-375).  gx = x-parts of update | reset | out incl. biases, [3C][HW] per
// (direction, frame); gh_ur = h-parts of update | reset, [2][B][2C][HW]; gh_o = h-part of the candidate, [2][B][C][HW]; both
// nullptr at the first step of a sweep (h = 0).  Plain expf / tanhf: a correctness path, not a tuned one.
struct GruArgs {
    const float *gx, *gh_ur, *gh_o, *hprev;
    float *ubuf, *hr, *hout;
    long gx_gs, gx_ns, hp_gs, hp_ns, ho_gs, ho_ns;
    int C, B, G;
    long HW;
};
__global__ __launch_bounds__(256) void gru_gate_kernel(const GruArgs a) {
    const long per = (long)a.C * a.HW, total = (long)a.G * a.B * per;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long z = i / per, e = i - z * per;
        const int g = (int)(z / a.B), n = (int)(z - (long)g * a.B);
        const float* gx = a.gx + g * a.gx_gs + n * a.gx_ns;
        float vu = gx[e], vr = gx[per + e];
        if (a.gh_ur) {
            const float* gh = a.gh_ur + z * 2 * per;
            vu += gh[e];
            vr += gh[per + e];
        }
        const float u = 1.f / (1.f + expf(-vu)), r = 1.f / (1.f + expf(-vr));
        a.ubuf[i] = u;
        a.hr[i] = a.hprev ? a.hprev[g * a.hp_gs + n * a.hp_ns + e] * r : 0.f;
    }
}
__global__ __launch_bounds__(256) void gru_out_kernel(const GruArgs a) {
    const long per = (long)a.C * a.HW, total = (long)a.G * a.B * per;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long z = i / per, e = i - z * per;
        const int g = (int)(z / a.B), n = (int)(z - (long)g * a.B);
        float vo = a.gx[g * a.gx_gs + n * a.gx_ns + 2 * per + e];
        if (a.gh_o) vo += a.gh_o[i];
        const float u = a.ubuf[i], o = tanhf(vo);
        const float hp = a.hprev ? a.hprev[g * a.hp_gs + n * a.hp_ns + e] : 0.f;
        a.hout[g * a.ho_gs + n * a.ho_ns + e] = hp * (1.f - u) + o * u;      // submodules.py:374
    }
}
// out[n] = cat(a[n], b[n]) along channels: two strided copies (skip_concat, V5.py:285-286)
static int concat_channels(const float* a, const float* b, float* out, long N, long ca_hw, long cb_hw, hipStream_t s) {
    BDE_HIP(hipMemcpy2DAsync(out, sizeof(float) * (ca_hw + cb_hw), a, sizeof(float) * ca_hw, sizeof(float) * ca_hw, (size_t)N,
                             hipMemcpyDeviceToDevice, s));
    BDE_HIP(hipMemcpy2DAsync(out + ca_hw, sizeof(float) * (ca_hw + cb_hw), b, sizeof(float) * cb_hw, sizeof(float) * cb_hw, (size_t)N,
                             hipMemcpyDeviceToDevice, s));
    return BDE_OK;
}

// ------------------------------------------------------------------------------------------
// packed layers
// ------------------------------------------------------------------------------------------
// A dense host-side layer before packing: rows x (Cin*KS*KS), row-major [row][ci][ky][kx].
struct DenseLayer {
    int rows = 0, Cin = 0, KS = 1;
    std::vector<float> w, bias, lnsum;   // lnsum empty unless the LayerNorm is folded
};

// One packed layer inside the device arena (offsets in floats).
struct PackedLayer {
    int Cin = 0, Cout = 0, KS = 1, CK = 8, nchunks = 0, ntiles = 0;
    bool lstm = false;
    long w_off = -1, b_off = -1, s_off = -1;   // weights / bias / lnsum
    long w_sz = 0;                              // floats of one group's packed weights
    int G = 1;                                  // groups packed back to back (fwd, bwd)
    long sb_off = -1, sb_sz = 0;                // split packing, three bf16 terms (conv_sb.h), floats; one group = sb_sz
    long sh_off = -1, sh_sz = 0;                // split packing, two fp16 terms, weights times 1 / sh_unscale (split.h)
    long sh_unscale_off = -1;                   // ... and the inverse scale, one float in the packed image (a receiver of the image has no weights)
    long split_off(int terms) const { return terms == 2 ? sh_off : sb_off; }
    long split_sz(int terms) const { return terms == 2 ? sh_sz : sb_sz; }
    int sb_chunks = 0;                          // 16-channel chunks
    mutable int sb_used = 0;                    // the latest launch of this layer ran on conv_sb_kernel (bde_get_info "sb_*")
    int G_decide = 0;                           // a one-group view of a grouped layer: choose launch shapes as for this many groups
};

// Group `g` of a grouped packed layer as a layer of its own (one sweep direction: bde_split_sweep, bde_op_encoder_conv)
static PackedLayer group_view(const PackedLayer& pl, int g) {
    PackedLayer v = pl;
    v.G_decide = pl.G;
    v.G = 1;
    v.w_off += g * pl.w_sz;
    if (v.b_off >= 0) v.b_off += (long)g * pl.Cout;
    if (v.s_off >= 0) v.s_off += (long)g * pl.Cout;
    if (v.sb_off >= 0) v.sb_off += g * pl.sb_sz;
    if (v.sh_off >= 0) v.sh_off += g * pl.sh_sz;
    return v;
}

struct Arena {
    std::vector<float> host;
    long alloc(long n) {
        long off = (long)host.size();
        long n4 = (n + 3) / 4 * 4;   // keep every segment 16-B aligned
        host.resize(off + n4, 0.f);
        return off;
    }
};

// Pack rows into [tile][chunk][tap][pair][64] MFMA A-fragment order (conv_mfma.h):
// lane l of fragment (tile, chunk, tap, pair) = W[row = tile*32 + (l&31)][ci = chunk*CK + 2*pair + (l>>5)][tap].
// `rowmap[packed_row]` = source row or -1 (zero).
static void pack_rows(const DenseLayer& d, const std::vector<int>& rowmap, int CK, int nchunks, float* dst) {
    const int taps = d.KS * d.KS, pairs = CK / 2;
    const int ntiles = (int)rowmap.size() / 32;
    for (int tile = 0; tile < ntiles; ++tile)
        for (int ch = 0; ch < nchunks; ++ch)
            for (int tap = 0; tap < taps; ++tap)
                for (int pr = 0; pr < pairs; ++pr) {
                    float* f = dst + ((((long)tile * nchunks + ch) * taps + tap) * pairs + pr) * 64;
                    for (int l = 0; l < 64; ++l) {
                        int row = rowmap[tile * 32 + (l & 31)];
                        int ci = ch * CK + 2 * pr + (l >> 5);
                        f[l] = (row >= 0 && ci < d.Cin) ? d.w[((long)row * d.Cin + ci) * taps + tap] : 0.f;
                    }
                }
}

// Append G dense layers (same shape) to the arena as one grouped packed layer.
// Fragment order of v_mfma_f32_16x16x4_f32 for the fused token kernel (token_fused.h):
// [co16 tile][k/4][64 lanes], lane l = W[tile*16 + (l&15)][k4*4 + (l>>4)]; rows and K zero-padded.
static long pack16(Arena& ar, const float* w, int rows, int K) {
    const int nct = cdiv(rows, 16), nk4 = cdiv(K, 4);
    const long off = ar.alloc((long)nct * nk4 * 64);
    float* dst = ar.host.data() + off;
    for (int ct = 0; ct < nct; ++ct)
        for (int k4 = 0; k4 < nk4; ++k4)
            for (int l = 0; l < 64; ++l) {
                const int r = ct * 16 + (l & 15), k = k4 * 4 + (l >> 4);
                dst[((long)ct * nk4 + k4) * 64 + l] = (r < rows && k < K) ? w[(long)r * K + k] : 0.f;
            }
    return off;
}

// The same rows four k-steps per 16-byte load for wideblock.h: [co16 tile][k/16][64 lanes][4],
// lane l, element j = W[tile*16 + (l&15)][kg*16 + 4*j + (l>>4)]; K must be a multiple of 16.
static long pack16x4(Arena& ar, const float* w, int rows, int K) {
    const int nct = cdiv(rows, 16), nkg = K / 16;
    const long off = ar.alloc((long)nct * nkg * 256);
    float* dst = ar.host.data() + off;
    for (int ct = 0; ct < nct; ++ct)
        for (int kg = 0; kg < nkg; ++kg)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 4; ++j) {
                    const int r = ct * 16 + (l & 15), k = kg * 16 + 4 * j + (l >> 4);
                    dst[((long)ct * nkg + kg) * 256 + l * 4 + j] = r < rows ? w[(long)r * K + k] : 0.f;
                }
    return off;
}

// w * scale as `terms` 16-bit terms (split.h)
static inline void split_terms(float w, int terms, float scale, unsigned short (&t)[3]) {
    if (terms == 2) { sb_split2(w * scale, t[0], t[1]); t[2] = 0; }
    else sb_split3(w, t[0], t[1], t[2]);
}
// the power-of-two packing scale of a group of layers in the two-term format (1 for three terms)
static float split_scale(const std::vector<const DenseLayer*>& groups, int terms) {
    if (terms != 2) return 1.f;
    float sc = 0.f;
    for (const DenseLayer* d : groups) {
        const float v = sb_weight_scale(d->w.data(), (long)d->w.size());
        sc = sc == 0.f ? v : std::min(sc, v);
    }
    return sc > 0.f ? sc : 1.f;
}

// winblock_sb.h: rows x K as split terms in A-fragment order of v_mfma_f32_16x16x32_{bf16,f16}:
// [row tile 16][k-step 32][term][64 lanes][8]: lane l = W[16 tile + (l & 15)][32 kstep + 8 (l >> 4) + j]; K % 32 == 0.
static long pack16_split(Arena& ar, const float* w, int rows, int K, int terms, long unscale_off) {
    const int nrt = cdiv(rows, 16), nks = K / 32;
    const long n_u16 = (long)nrt * nks * terms * 64 * 8;
    const long off = ar.alloc(n_u16 / 2);
    unsigned short* dst = reinterpret_cast<unsigned short*>(ar.host.data() + off);
    const float scale = terms == 2 ? sb_weight_scale(w, (long)rows * K) : 1.f;
    if (unscale_off >= 0) ar.host[unscale_off] = 1.f / scale;
    for (int rt = 0; rt < nrt; ++rt)
        for (int ks = 0; ks < nks; ++ks)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int r = rt * 16 + (l & 15), k = ks * 32 + 8 * (l >> 4) + j;
                    unsigned short t3[3];
                    split_terms(r < rows ? w[(long)r * K + k] : 0.f, terms, scale, t3);
                    for (int t = 0; t < terms; ++t) dst[((((long)rt * nks + ks) * terms + t) * 64 + l) * 8 + j] = t3[t];
                }
    return off;
}

// tokgemm_sb_kernel (wideblock.h): rows x K as two fp16 terms, A-fragment order of the 16x16x32 MFMA, k in the order a lane of a
// FRAG16 tensor holds two consecutive channel groups: element jj of lane (m, g4) of k-step ks = W[16 rt + m][32 ks + (jj < 4 ?
// 4 jj + g4 : 16 + 4 (jj - 4) + g4)].  [row tile 16][k-step 32][term][64 lanes][8]; K % 32 == 0.
static long pack16_split_frag(Arena& ar, const float* w, int rows, int K, long unscale_off) {
    const int nrt = cdiv(rows, 16), nks = K / 32;
    const long n_u16 = (long)nrt * nks * 2 * 64 * 8;
    const long off = ar.alloc(n_u16 / 2);
    unsigned short* dst = reinterpret_cast<unsigned short*>(ar.host.data() + off);
    const float scale = sb_weight_scale(w, (long)rows * K);
    ar.host[unscale_off] = 1.f / scale;
    for (int rt = 0; rt < nrt; ++rt)
        for (int ks = 0; ks < nks; ++ks)
            for (int l = 0; l < 64; ++l)
                for (int jj = 0; jj < 8; ++jj) {
                    const int r = rt * 16 + (l & 15), g4 = l >> 4;
                    const int k = ks * 32 + (jj < 4 ? 4 * jj + g4 : 16 + 4 * (jj - 4) + g4);
                    unsigned short t3[3];
                    split_terms(r < rows ? w[(long)r * K + k] : 0.f, 2, scale, t3);
                    for (int t = 0; t < 2; ++t) dst[((((long)rt * nks + ks) * 2 + t) * 64 + l) * 8 + jj] = t3[t];
                }
    return off;
}

// Weight fragments of the recurrent step kernel (lstm16.h):
// [hidden16 block][chunk of 8 channels][tap][k4][64 lanes][gate], lane l = W[gate*Ch + hb*16 + (l&15)][chunk*8 + k4*4 + (l>>4)][tap]
// (the four gate fragments of a lane are adjacent: one 16-byte LDS read fetches them).
static PackedLayer pack_lstm16(Arena& ar, const std::vector<const DenseLayer*>& groups) {
    const DenseLayer& d0 = *groups[0];
    PackedLayer pl;
    pl.Cin = d0.Cin;
    pl.Cout = d0.rows;
    pl.KS = 3;
    pl.lstm = true;
    pl.G = (int)groups.size();
    pl.CK = L16_CK;
    pl.nchunks = cdiv(d0.Cin, L16_CK);
    const int Ch = d0.rows / 4, nhb = cdiv(Ch, 16);
    pl.ntiles = nhb;
    pl.w_sz = (long)nhb * pl.nchunks * L16_AFL;
    pl.w_off = ar.alloc(pl.w_sz * pl.G);
    pl.b_off = ar.alloc((long)d0.rows * pl.G);
    for (int g = 0; g < pl.G; ++g) {
        const DenseLayer& d = *groups[g];
        float* dst = ar.host.data() + pl.w_off + g * pl.w_sz;
        for (int hb = 0; hb < nhb; ++hb)
            for (int ch = 0; ch < pl.nchunks; ++ch)
                for (int tap = 0; tap < 9; ++tap)
                    for (int k4 = 0; k4 < 2; ++k4)
                        for (int gate = 0; gate < 4; ++gate)
                            for (int l = 0; l < 64; ++l) {
                                const int hc = hb * 16 + (l & 15), ci = ch * 8 + k4 * 4 + (l >> 4);
                                const long o = ((((long)(hb * pl.nchunks + ch) * 9 + tap) * 2 + k4) * 64 + l) * 4 + gate;
                                dst[o] = (hc < Ch && ci < d.Cin) ? d.w[((long)(gate * Ch + hc) * d.Cin + ci) * 9 + tap] : 0.f;
                            }
        std::copy(d.bias.begin(), d.bias.end(), ar.host.begin() + pl.b_off + (long)g * d0.rows);
    }
    return pl;
}

// The same weights for 8-channel workgroups (lstm16.h, HC8): [hidden8 block][chunk][tap][k4][64 lanes][tile 2],
// tile t stacks gates 2t and 2t+1: lane l -> row m = l&15: gate 2t + (m>>3), hidden channel hb*8 + (m&7).
static PackedLayer pack_lstm8(Arena& ar, const std::vector<const DenseLayer*>& groups) {
    const DenseLayer& d0 = *groups[0];
    PackedLayer pl;
    pl.Cin = d0.Cin;
    pl.Cout = d0.rows;
    pl.KS = 3;
    pl.lstm = true;
    pl.G = (int)groups.size();
    pl.CK = L16_CK;
    pl.nchunks = cdiv(d0.Cin, L16_CK);
    const int Ch = d0.rows / 4, nhb = cdiv(Ch, 8);
    pl.ntiles = nhb;
    const long afl = 9 * 2 * 64 * 2;
    pl.w_sz = (long)nhb * pl.nchunks * afl;
    pl.w_off = ar.alloc(pl.w_sz * pl.G);
    pl.b_off = -1;
    for (int g = 0; g < pl.G; ++g) {
        const DenseLayer& d = *groups[g];
        float* dst = ar.host.data() + pl.w_off + g * pl.w_sz;
        for (int hb = 0; hb < nhb; ++hb)
            for (int ch = 0; ch < pl.nchunks; ++ch)
                for (int tap = 0; tap < 9; ++tap)
                    for (int k4 = 0; k4 < 2; ++k4)
                        for (int l = 0; l < 64; ++l)
                            for (int t = 0; t < 2; ++t) {
                                const int m = l & 15, gate = 2 * t + (m >> 3), hc = hb * 8 + (m & 7), ci = ch * 8 + k4 * 4 + (l >> 4);
                                const long o = ((((long)(hb * pl.nchunks + ch) * 9 + tap) * 2 + k4) * 64 + l) * 2 + t;
                                dst[o] = (hc < Ch && ci < d.Cin) ? d.w[((long)(gate * Ch + hc) * d.Cin + ci) * 9 + tap] : 0.f;
                            }
    }
    return pl;
}

// lstm_sb.h: h-part of the gates as split terms, rows GATE-INTERLEAVED (row 8 q + 4 hl + gate of tile rt = that gate of hidden
// channel 8 rt + 4 hl + q), A-fragment order of the 32x32x16 MFMA: [group][row tile][chunk 16][tap][term][64 lanes][8]
static void pack_lstm_sbk_terms(Arena& ar, PackedLayer& pl, const std::vector<const DenseLayer*>& groups, int terms) {
    const DenseLayer& d0 = *groups[0];
    const int Ch = d0.rows / 4, nrt = cdiv(Ch, 8), C16 = cdiv(d0.Cin, 16);
    const long per_group_u16 = (long)nrt * C16 * 9 * terms * 64 * 8;
    pl.Cin = d0.Cin; pl.Cout = d0.rows; pl.KS = 3; pl.G = (int)groups.size();
    const long sz = per_group_u16 / 2;
    const long off = ar.alloc(sz * (long)groups.size());
    const float scale = split_scale(groups, terms);
    pl.sb_chunks = C16;
    if (terms == 2) { pl.sh_off = off; pl.sh_sz = sz; pl.sh_unscale_off = ar.alloc(4); ar.host[pl.sh_unscale_off] = 1.f / scale; }
    else { pl.sb_off = off; pl.sb_sz = sz; }
    for (size_t g = 0; g < groups.size(); ++g) {
        const DenseLayer& d = *groups[g];
        unsigned short* dst = reinterpret_cast<unsigned short*>(ar.host.data() + off + (long)g * sz);
        for (int rt = 0; rt < nrt; ++rt)
            for (int ch = 0; ch < C16; ++ch)
                for (int tap = 0; tap < 9; ++tap)
                    for (int l = 0; l < 64; ++l)
                        for (int j = 0; j < 8; ++j) {
                            // row rho = 8 q + 4 hl + gate  <->  hidden channel 8 rt + 4 hl + q (lstm_sb.h)
                            const int rho = l & 31, hc = rt * 8 + 4 * ((rho >> 2) & 1) + (rho >> 3), gate = rho & 3, ci = ch * 16 + 8 * (l >> 5) + j;
                            const float w = (hc < Ch && ci < d.Cin) ? d.w[((long)(gate * Ch + hc) * d.Cin + ci) * 9 + tap] : 0.f;
                            unsigned short t3[3];
                            split_terms(w, terms, scale, t3);
                            for (int k = 0; k < terms; ++k)
                                dst[(((((long)rt * C16 + ch) * 9 + tap) * terms + k) * 64 + l) * 8 + j] = t3[k];
                        }
    }
}
static void pack_lstm_sbk(Arena& ar, PackedLayer& pl, const std::vector<const DenseLayer*>& groups) {
    pack_lstm_sbk_terms(ar, pl, groups, 3);
    pack_lstm_sbk_terms(ar, pl, groups, 2);
}

// conv_sb.h: the weights as split terms in A-fragment order of the 32x32x16 MFMA:
// [group][co tile 32][chunk 16][tap][term][64 lanes][8]: lane l = W[32 tile + (l & 31)][16 chunk + 8 (l >> 5) + j][tap]
static void pack_split_terms(Arena& ar, PackedLayer& pl, const std::vector<const DenseLayer*>& groups, int terms) {
    const DenseLayer& d0 = *groups[0];
    const int taps = d0.KS * d0.KS, ncot = cdiv(d0.rows, 32), C16 = cdiv(d0.Cin, 16);
    const long per_group_u16 = (long)ncot * C16 * taps * terms * 64 * 8;
    const long sz = per_group_u16 / 2;                              // in floats
    const long off = ar.alloc(sz * (long)groups.size());
    const float scale = split_scale(groups, terms);
    pl.sb_chunks = C16;
    if (terms == 2) { pl.sh_off = off; pl.sh_sz = sz; pl.sh_unscale_off = ar.alloc(4); ar.host[pl.sh_unscale_off] = 1.f / scale; }
    else { pl.sb_off = off; pl.sb_sz = sz; }
    for (size_t g = 0; g < groups.size(); ++g) {
        const DenseLayer& d = *groups[g];
        unsigned short* dst = reinterpret_cast<unsigned short*>(ar.host.data() + off + (long)g * sz);
        for (int ct = 0; ct < ncot; ++ct)
            for (int ch = 0; ch < C16; ++ch)
                for (int tap = 0; tap < taps; ++tap)
                    for (int l = 0; l < 64; ++l)
                        for (int j = 0; j < 8; ++j) {
                            const int row = ct * 32 + (l & 31), ci = ch * 16 + 8 * (l >> 5) + j;
                            const float w = (row < d.rows && ci < d.Cin) ? d.w[((long)row * d.Cin + ci) * taps + tap] : 0.f;
                            unsigned short t3[3];
                            split_terms(w, terms, scale, t3);
                            for (int k = 0; k < terms; ++k)
                                dst[(((((long)ct * C16 + ch) * taps + tap) * terms + k) * 64 + l) * 8 + j] = t3[k];
                        }
    }
}
static void pack_split_bf16(Arena& ar, PackedLayer& pl, const std::vector<const DenseLayer*>& groups) {
    pack_split_terms(ar, pl, groups, 3);
    pack_split_terms(ar, pl, groups, 2);
}

// Channel chunking: generic convs CK = 8; the recurrent gate conv CK = 16 with chunks in groups of
// four (one per wave); pointwise layers CK = 16 in groups of eight (any pw_gemm split-K factor).
static PackedLayer pack_layer(Arena& ar, const std::vector<const DenseLayer*>& groups, bool lstm) {
    const DenseLayer& d0 = *groups[0];
    PackedLayer pl;
    pl.Cin = d0.Cin;
    pl.Cout = d0.rows;
    pl.KS = d0.KS;
    pl.lstm = lstm;
    pl.G = (int)groups.size();
    pl.CK = lstm ? LSTM_CK : (d0.KS == 1 ? 16 : conv_ck(d0.KS));
    pl.nchunks = cdiv(d0.Cin, pl.CK);
    if (lstm) pl.nchunks = cdiv(pl.nchunks, 4) * 4;
    if (d0.KS == 1) pl.nchunks = cdiv(pl.nchunks, 8) * 8;
    std::vector<int> rowmap;
    if (lstm) {
        // packed tile (cb*4 + gate) holds gate rows gate*Ch + cb*32 .. +32  (conv_mfma.h EPI_LSTM)
        const int Ch = d0.rows / 4, ncb = cdiv(Ch, 32);
        rowmap.assign((size_t)ncb * 4 * 32, -1);
        for (int cb = 0; cb < ncb; ++cb)
            for (int gate = 0; gate < 4; ++gate)
                for (int j = 0; j < 32; ++j)
                    if (cb * 32 + j < Ch) rowmap[((size_t)cb * 4 + gate) * 32 + j] = gate * Ch + cb * 32 + j;
    } else {
        const int rows_pad = cdiv(d0.rows, 64) * 64;        // MT (1 or 2 tiles per wave) is chosen at launch
        rowmap.assign(rows_pad, -1);
        for (int r = 0; r < d0.rows; ++r) rowmap[r] = r;
    }
    pl.ntiles = (int)rowmap.size() / 32;
    pl.w_sz = (long)pl.ntiles * pl.nchunks * d0.KS * d0.KS * (pl.CK / 2) * 64;
    pl.w_off = ar.alloc(pl.w_sz * pl.G);
    pl.b_off = ar.alloc((long)d0.rows * pl.G);
    const bool ln = !d0.lnsum.empty();
    if (ln) pl.s_off = ar.alloc((long)d0.rows * pl.G);
    for (int g = 0; g < pl.G; ++g) {
        const DenseLayer& d = *groups[g];
        pack_rows(d, rowmap, pl.CK, pl.nchunks, ar.host.data() + pl.w_off + g * pl.w_sz);
        std::copy(d.bias.begin(), d.bias.end(), ar.host.begin() + pl.b_off + (long)g * d0.rows);
        if (ln) std::copy(d.lnsum.begin(), d.lnsum.end(), ar.host.begin() + pl.s_off + (long)g * d0.rows);
    }
    return pl;
}

struct AttnBlock {
    PackedLayer qkv, proj, fc1, fc2;
    long proj16 = -1, fc1_16 = -1, fc2_16 = -1, qkv16 = -1;   // 16x16x4 packings for token_fused.h
    long projW = -1, fc1W = -1, fc2W = -1, qkvW = -1;         // four-k-steps-per-load packings for wideblock.h
    long projS = -1, fc1S = -1, fc2S = -1, qkvS = -1;         // split packings for winblock_sb.h, three bf16 terms
    long projH = -1, fc1H = -1, fc2H = -1, qkvH = -1;         // two fp16 terms; unscaleH -> {q|k|v, proj, fc1, fc2} inverse scales
    long unscaleH = -1;                                       // (four floats in the packed image)
    long qkvHF = -1, qkvHF_unscale = -1;                      // q|k|v as two fp16 terms in FRAG16 k order (attn_tok16_kernel<true, true>)
    long projHF = -1, fc1HF = -1, mlpHF_unscale = -1;         // proj, fc1 likewise (projfc1_sb_kernel); unscale: {proj, fc1}
    long fc1N = -1, fc2N = -1, mlpN_unscale = -1;             // fc1, fc2 as two fp16 terms in natural k order (mlp_fused_kernel); unscale: {fc1, fc2}
    long qkvN = -1, qkvN_unscale = -1;                        // q|k|v likewise (wide_core_kernel on SPL16 operands)
    long kvpad_off = -1;    // [2C]
    long bias_off = -1;     // [heads][D*49][49]
    long biasF_off = -1;    // the same bias in the score-tile order of winblock.h (64 channels, 16 heads only)
    long biasW_off = -1;    // ... and in score-tile order with the keys slot-major (wide_core.h: head_dim 16 levels)
};
struct AttnLevel {
    int depth = 0, C = 0;
    std::vector<AttnBlock> blocks;
    PackedLayer kvall;      // rows = depth*2C: K|V of every block for a non-query frame
    long kvallW = -1;       // the same rows packed for wideblock.h
    long kvallH = -1, kvallH_unscale = -1;   // ... and as two fp16 terms for tokgemm_sb_kernel (k order of FRAG16 group pairs)
};

struct Workspace {
    int T = 0, B = 0, H = 0, W = 0;
    std::vector<void*> allocs;
    float* ev = nullptr;
    float* head = nullptr;
    float* out = nullptr;
    std::vector<float*> xenc, gx, hseq, cst, merged, mergedT, kvun, kvref, dec, qkv0;
    float *qkv = nullptr, *ao = nullptr, *x1 = nullptr, *hid = nullptr, *xa = nullptr, *xb = nullptr;
    int* tile_count = nullptr;    // wide_mlp.h: per (batch, token tile) arrival counters of the fused MLP launch, zero between launches
    // SPL16 twins (wide_core.h) of the frames of a head_dim-16 level and of the block intermediates, with their LayerNorm statistics
    std::vector<float*> mergedS, mstats;
    float *xaS = nullptr, *xbS = nullptr, *stA = nullptr, *stB = nullptr;
    float* up = nullptr;          // upsampled (+skip) decoder input, largest decoder
    float* sb = nullptr;          // split-bf16 image of the input of the convolution in flight (conv_sb.h)
    float* sb2 = nullptr;         // split-bf16 encoder output of a level, written by the encoder conv's epilogue for its gate conv
    long sb2_bytes = 0;
    std::vector<float*> gur, ghr, gou, gub;   // ConvGRU per level: h-parts of update | reset, h * reset, h-part of the candidate, update gate
    float *cat = nullptr, *fuse = nullptr;    // skip_concat: cat(skip, x) and the 1x1 fusion's output
    float *rbA = nullptr, *rbX[2] = {nullptr, nullptr}, *zero_l = nullptr;   // bottleneck: conv1 output, block outputs, a zero frame
    std::vector<float*> hsk;      // per level: hidden state as SB16, two buffers [2][2 dirs][B][C16][hw] (lstm_sb.h)
    std::vector<float*> hsb, ghb; // per level: hidden state as SB16, two buffers [2][2 dirs][B][C16][hw] / h-part of the gates [2][B][4C][hw]
    long sb_bytes = 0;
    hipGraphExec_t graph_exec = nullptr;   // captured launch sequence of forward_body for this shape
    int graph_part = 0;                    // ... PART_ALL, or PART_MAIN when the forward's tail is launched behind the graph
    bool warm = false;
    void release() {
        if (graph_exec) { (void)hipGraphExecDestroy(graph_exec); graph_exec = nullptr; }
        warm = false;
        for (void* p : allocs) (void)hipFree(p);
        allocs.clear();
        // no pointer outlives its allocation: the op-level entry points test them (ws.sb, ws.hsk[l], ...) before use
        ev = head = out = qkv = ao = x1 = hid = xa = xb = up = sb = sb2 = cat = fuse = rbA = rbX[0] = rbX[1] = zero_l = nullptr;
        tile_count = nullptr;
        xaS = xbS = stA = stB = nullptr;
        mergedS.clear(); mstats.clear();
        sb_bytes = sb2_bytes = 0;
        for (auto* v : {&xenc, &gx, &hseq, &cst, &merged, &mergedT, &kvun, &kvref, &dec, &qkv0, &gur, &ghr, &gou, &gub, &hsk, &hsb, &ghb}) v->clear();
        T = B = H = W = 0;
    }
};

}  // namespace bde

using namespace bde;

struct bde_model {
    bde_config cfg;
    int L = 0;
    std::map<std::string, std::pair<std::vector<int64_t>, std::vector<float>>> raw;
    bool finalized = false;
    Arena arena;
    float* dev = nullptr;   // device image of the arena
    long dev_numel = 0;
    PackedLayer head, pred_dummy;
    std::vector<PackedLayer> enc, gx, lstm, lstm8, dec;   // enc/gx/lstm: G=2 (fwd,bwd); lstm8 = the 8-channel-workgroup packing
    std::vector<PackedLayer> lstm_sb;                     // h-part of the gates, split-bf16 packing only (conv_sb.h)
    std::vector<PackedLayer> lstm_sbk;                    // ... gate-interleaved rows for the fused step kernel (lstm_sb.h)
    std::vector<PackedLayer> lstm_sbx;                    // ... the same with K = [x | h]: the x-part of the gates inside the step
    std::vector<PackedLayer> gru_ur, gru_o;               // ConvGRU: h-parts of update | reset and of the candidate (G = 2)
    std::vector<PackedLayer> dec_fuse;                    // skip_concat: 1x1 fusion conv in front of decoder j
    PackedLayer pred_fuse;                                // ... and in front of predI
    std::vector<PackedLayer> rb1, rb2;                    // ResidualBlockNoBN bottleneck: conv1 / conv2 of block k
    std::vector<AttnLevel> attn;
    long predw_off = -1, predb_off = -1, zero_off = -1;
    // Workspace slots: slot 0 always; with pipeline depth 2 consecutive forward calls alternate between
    // two workspaces and two internal streams, so the latency-bound attention chain of one sequence
    // overlaps the batched convolutions of the next (the sequences are independent, bde2vid.py:31).
    static constexpr int MAX_SLOTS = 4;
    Workspace wslots[MAX_SLOTS];
    int cur = 0;
    Workspace& W() { return wslots[cur]; }
    hipStream_t cap_stream = nullptr;
    int use_graph = 1;                   // replay the captured launch sequence from the second call of a shape on
    int pipeline = 1;                    // 1 = every call runs on the caller's stream (default); 2 = double-buffered
    hipStream_t pstream[MAX_SLOTS] = {};
    hipEvent_t pin[MAX_SLOTS] = {}, pout[MAX_SLOTS] = {};
    bool pbusy[MAX_SLOTS] = {};
    hipStream_t last_stream = nullptr;
    long ncalls = 0;
    int device = 0;
    // optional HIP-event timing of tagged launches / stages (bde_profile_*)
    // side stream: per-frame work that only depends on already-refined frames (next level's encoder /
    // gate convs, or the decoder) runs beside the sequential attention chain
    // one set per workspace slot for eager forwards, and one more (index MAX_SLOTS) used only while a graph is being captured: a
    // stream that still holds eager work of an earlier call cannot join a capture
    hipStream_t side[MAX_SLOTS + 1] = {};
    std::vector<hipEvent_t> frame_ev[MAX_SLOTS + 1];
    hipEvent_t join_ev[MAX_SLOTS + 1] = {};
    int overlap = 0;              // 1: decoder of the frames already refined beside the last level's attention chain, a forked branch
                                  // of the captured graph (forward_body).  Bit-identical frames; measured on one box, config A: 1098 vs
                                  // 2179 frames/s with three sequences in flight, 1459 vs 1701 with one -- a hipGraph with a fork does
                                  // not replay as one batch of packets on ROCm 7.2: off
    int eager_cut = 1;            // default mode: head + first encoder convolution and the last convolution launched outside the graph (forward_on)
    int overlap_chunk = 4;        // frames handed to the side stream per launch set
    int debug_skip = 0;           // diagnostic what-if timing only (results are wrong): bit0 attention level 0, bit1 attention levels >= 1,
                                  // bit2 recurrent steps, bit3 decoder, bit4 encoder + gate convs
    int tok_debug = 0;
    unsigned long long* tok_stamps = nullptr;
    Tuning tune;                  // launch-shape overrides (common.h), per model
    int lstm_hc8 = -1;            // recurrent step with 8-channel workgroups: -1 auto (lstm16_wants_hc8), 0 never, 1 always
    int dir_mask = 3;             // sweep directions a recurrent level runs: bit 0 forward, bit 1 backward (bde_split_sweep sets one)
    int winblock = 1;             // one launch per attention block (winblock.h) where the level qualifies
    int winblock_sb = 1;          // ... with its GEMM phases on the bf16 matrix cores, three-term split operands (winblock_sb.h)
    int fuse_pred = 1;            // predI + sigmoid in the last decoder conv's epilogue
    int wide = 1;                 // head_dim-16 attention levels on the fragment-layout chain (wideblock.h)
    int wide_fuse_qkv = 1;        // ... with the query frame's q | k | v computed inside the attention core (no GEMM launch of its own)
    int xcd_remap = 1;            // conv_sb workgroup order by XCD (conv_sb.h)
    int fuse_enc_sb = 1;          // encoder conv epilogue writes the SB16 input of its gate conv (no fp32 planes, no conversion pass)
    int conv_sb = 1;              // batched convolutions on the 16-bit matrix cores with split operands (conv_sb.h)
    int sb_terms = BDE_DEFAULT_SB_TERMS;   // format of every split operand (split.h): 2 = two fp16 terms (three MFMAs per fp32 block;
                                  // activations must stay below 65520), 3 = three bf16 terms (six MFMAs; fp32's exponent range)
    int lstm_two_streams = 0;     // the two sweep directions of a level as two launch chains on two streams (independent until the merge);
                                  // measured: 1208 vs 1444 frames/s pipelined, 1135 vs 1161 single stream -- half-size launches take almost as long: off
    hipStream_t dir_stream[4] = {};         // per workspace slot: the second direction's stream and its fork / join events
    hipEvent_t dir_fork[4] = {}, dir_join[4] = {};
    int use_lstm_sbk = 1;         // recurrent step on the 16-bit matrix cores with the pointwise tail fused (lstm_sb.h) where a shape fits
    int wide_fuse_mlp = 1;        // ... and x1 = x + proj(.) together with GELU(fc1(LN(x1))) in one launch (projfc1_sb_kernel)
    int wide_fuse_fc2 = 1;        // ... and fc2 + both residuals in the same launch (mlp_fused_kernel, wide_mlp.h): two launches per block
    int wide_spl = 1;             // ... on frames kept as SPL16 (pre-split operand fragments + LayerNorm statistics, wide_core.h)
    int wide_core2 = 1;           // the window half of such a block as wide_core_kernel (wide_core.h): weights by LDS-DMA, K | V of the
                                  // refined neighbour frame computed inside (no K|V GEMM launch between two frames)
    int wide_kv_sb = 1;           // K|V GEMMs of the head_dim-16 chain on two-term split operands (tokgemm_sb_kernel, wideblock.h)
    int lstm_fuse_x = 1;          // ... and the x-part of the gates in the same contraction (no batched gate convolution, no gx round trip)
    int lstm_sb_mode = 0;         // recurrent step as conv_sb + pointwise kernel: 0 off (default: measured slower), 1 wherever it fits, -1 by estimate
    long fused_min_tiles = 160;   // token_fused.h is used when a level has at least this many 32-pixel tiles
    // ---- range guard of the two-term operand format (split.h) ----------------------------------------------------------------
    // Every kernel that splits fp32 activations into two fp16 terms ORs bit 0 into the overflow word of the workspace slot its
    // forward runs in when a value reaches 65520 (fp16's infinity).  The word is copied to pinned host memory behind the last
    // such kernel; settle_overflow() reads it when the frames are handed over: "sb_auto" = 1 (default) recomputes the forward in
    // the three-term bf16 format (fp32's exponent range) and keeps that format for the model, 0 fails with BDE_ERR_RANGE.
    unsigned* ovf_dev = nullptr;  // [MAX_SLOTS] device words
    unsigned* ovf_host = nullptr; // [MAX_SLOTS] pinned host mirror
    int sb_auto = 1;
    long sb_overflows = 0;        // forwards whose two-term operands left fp16's range
    int sb_latched = 0;           // 1: such a forward switched the model to three bf16 terms
    struct Pending {              // a forward whose overflow word has not been looked at yet
        bool on = false;
        hipEvent_t done = nullptr;
        hipStream_t stream = nullptr;
        int T = 0, B = 0, H = 0, W = 0;
        std::vector<float*> images;
    } pend[MAX_SLOTS];
    unsigned* ovf() const { return (ovf_dev && sb_terms == 2) ? ovf_dev + cur : nullptr; }
    bool prof_on = false;
    struct ProfSpan { std::string name; hipEvent_t a, b; };
    std::vector<ProfSpan> prof;
    std::vector<hipEvent_t> prof_pool;

    int cin(int l) const { return cfg.basechannels << l; }
    int cout(int l) const { return cfg.basechannels << (l + 1); }
    const float* P(long off) const { return dev + off; }
    long lstm_sb_off(int l) const { return (size_t)l < lstm_sb.size() ? lstm_sb[l].sb_off : -1; }
    long zero_off_long() const { return zero_off; }
};

namespace bde {

// ---- event-pair profiling ---------------------------------------------------------------------
static hipEvent_t prof_event(bde_model* m) {
    hipEvent_t e;
    if (!m->prof_pool.empty()) { e = m->prof_pool.back(); m->prof_pool.pop_back(); return e; }
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
struct ProfScope {
    bde_model* m; hipStream_t s; hipEvent_t a = nullptr, b = nullptr; const char* name;
    ProfScope(bde_model* m_, const char* n, hipStream_t s_) : m(m_), s(s_), name(n) {
        if (!m->prof_on) return;
        a = prof_event(m); b = prof_event(m);
        if (a) (void)hipEventRecord(a, s);
    }
    ~ProfScope() {
        if (!m->prof_on || !a || !b) return;
        (void)hipEventRecord(b, s);
        m->prof.push_back({name, a, b});
    }
};

// span names with an index ("lstm0", "dec_conv2"): interned, the spans keep the pointer
static const char* pname(const char* base, int i) {
    static std::map<std::string, std::string> names;
    static std::mutex mu;                               // forwards of two models may run on two host threads
    const std::string k = std::string(base) + std::to_string(i);
    std::lock_guard<std::mutex> lock(mu);
    return names.emplace(k, k).first->second.c_str();   // (map nodes never move: the pointer stays valid)
}

static const std::string GP = "generator.";

static int get_raw(bde_model* m, const std::string& key, std::vector<int64_t> shape, const float** out) {
    auto it = m->raw.find(GP + key);
    if (it == m->raw.end()) return fail(BDE_ERR_STATE, "missing weight '%s%s'", GP.c_str(), key.c_str());
    if (it->second.first != shape) {
        std::string got, want;
        for (auto v : it->second.first) got += std::to_string(v) + ",";
        for (auto v : shape) want += std::to_string(v) + ",";
        return fail(BDE_ERR_ARG, "weight '%s': shape [%s] but the config implies [%s]", key.c_str(), got.c_str(),
                    want.c_str());
    }
    *out = it->second.second.data();
    return BDE_OK;
}

static const float* get_raw_opt(bde_model* m, const std::string& key, int64_t n) {
    auto it = m->raw.find(GP + key);
    if (it == m->raw.end()) return nullptr;
    int64_t have = 1;
    for (auto v : it->second.first) have *= v;
    return have == n ? it->second.second.data() : nullptr;
}

// ConvLayer / UpsampleConvLayer (submodules.py:85-147) as ONE dense convolution: conv2d (no bias under BN, :91) followed by
// BatchNorm2d or InstanceNorm2d(track_running_stats=True) in eval mode (:96-109) is the affine y -> (y - mean) * s + beta with
// s = gamma / sqrt(var + eps) per output channel, folded into the weights and the bias (fp64).
static int dense_conv(bde_model* m, const std::string& wkey, const std::string& bkey, int rows, int cin_total,
                      int ci_off, int cin, int ks, bool with_bias, DenseLayer* d);
static int dense_convlayer(bde_model* m, const std::string& prefix, int rows, int cin, int ks, DenseLayer* d) {
    const int norm = m->cfg.norm;
    const float* w;
    BDE_TRY(get_raw(m, prefix + "conv2d.weight", {rows, cin, ks, ks}, &w));
    const float* b = nullptr;
    if (norm != 1) BDE_TRY(get_raw(m, prefix + "conv2d.bias", {rows}, &b));
    d->rows = rows; d->Cin = cin; d->KS = ks;
    d->w.assign(w, w + (size_t)rows * cin * ks * ks);
    d->bias.assign(rows, 0.f);
    if (b) std::copy(b, b + rows, d->bias.begin());
    if (norm == 0) return BDE_OK;
    const float *mean, *var, *gamma = nullptr, *beta = nullptr;
    BDE_TRY(get_raw(m, prefix + "norm_layer.running_mean", {rows}, &mean));
    BDE_TRY(get_raw(m, prefix + "norm_layer.running_var", {rows}, &var));
    if (norm == 1) {
        BDE_TRY(get_raw(m, prefix + "norm_layer.weight", {rows}, &gamma));
        BDE_TRY(get_raw(m, prefix + "norm_layer.bias", {rows}, &beta));
    }
    const size_t per_row = (size_t)cin * ks * ks;
    for (int r = 0; r < rows; ++r) {
        const double sc = (gamma ? (double)gamma[r] : 1.0) / std::sqrt((double)var[r] + 1e-5);
        for (size_t i = 0; i < per_row; ++i) d->w[r * per_row + i] = (float)((double)d->w[r * per_row + i] * sc);
        d->bias[r] = (float)(((double)d->bias[r] - (double)mean[r]) * sc + (beta ? (double)beta[r] : 0.0));
    }
    return BDE_OK;
}

static int dense_conv(bde_model* m, const std::string& wkey, const std::string& bkey, int rows, int cin_total,
                      int ci_off, int cin, int ks, bool with_bias, DenseLayer* d) {
    const float *w, *b;
    BDE_TRY(get_raw(m, wkey, {rows, cin_total, ks, ks}, &w));
    BDE_TRY(get_raw(m, bkey, {rows}, &b));
    d->rows = rows;
    d->Cin = cin;
    d->KS = ks;
    d->w.resize((size_t)rows * cin * ks * ks);
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cin; ++c)
            for (int t = 0; t < ks * ks; ++t)
                d->w[((size_t)r * cin + c) * ks * ks + t] = w[((size_t)r * cin_total + ci_off + c) * ks * ks + t];
    d->bias.assign(rows, 0.f);
    if (with_bias) std::copy(b, b + rows, d->bias.begin());
    return BDE_OK;
}

// Linear(LayerNorm(x)) = rstd * (W' x - mu * s) + b'  with  W' = W diag(gamma), s = W' 1, b' = W beta + b.
// `scale` multiplies the whole output (query scale, DTransformer.py:192).
static void fold_ln_rows(const float* W, const float* b, const float* gamma, const float* beta, int rows, int C,
                         float scale, DenseLayer* d, int row_off) {
    for (int r = 0; r < rows; ++r) {
        double s = 0.0, bb = b[r];
        for (int c = 0; c < C; ++c) {
            float wf = W[(size_t)r * C + c] * gamma[c];
            d->w[(size_t)(row_off + r) * C + c] = wf * scale;
            s += (double)wf;
            bb += (double)W[(size_t)r * C + c] * (double)beta[c];
        }
        d->lnsum[row_off + r] = (float)(s * scale);
        d->bias[row_off + r] = (float)(bb * scale);
    }
}

static int build_packed(bde_model* m) {
    const bde_config& c = m->cfg;
    const int L = c.num_encoders, ks = c.ks, bc = c.basechannels;
    Arena& ar = m->arena;
    ar.host.clear();
    m->enc.assign(L, PackedLayer());
    m->gx.assign(L, PackedLayer());
    m->lstm.assign(L, PackedLayer());
    m->lstm8.assign(L, PackedLayer());
    m->lstm_sb.assign(L, PackedLayer());
    m->lstm_sbk.assign(L, PackedLayer());
    m->lstm_sbx.assign(L, PackedLayer());
    m->dec.assign(L, PackedLayer());
    m->attn.assign(L, AttnLevel());
    m->gru_ur.assign(L, PackedLayer());
    m->gru_o.assign(L, PackedLayer());
    m->dec_fuse.assign(L, PackedLayer());
    m->rb1.clear();
    m->rb2.clear();
    {
        DenseLayer d;
        BDE_TRY(dense_convlayer(m, "head.", bc, c.num_bins, ks, &d));
        m->head = pack_layer(ar, {&d}, false);
        pack_split_bf16(ar, m->head, {&d});
    }
    const char* dirs[2] = {"forward_encoder", "backward_encoder"};
    for (int l = 0; l < L; ++l) {
        const int ci = m->cin(l), co = m->cout(l);
        DenseLayer e[2];
        for (int d = 0; d < 2; ++d) {
            // RecurrentConv.conv (submodules.py:186-187) or, with useRC = False, the encoder itself (V5.py:256-258)
            std::string p = std::string(dirs[d]) + "." + std::to_string(l) + (c.use_rc ? ".conv." : ".");
            BDE_TRY(dense_convlayer(m, p, co, ci, ks, &e[d]));
        }
        m->enc[l] = pack_layer(ar, {&e[0], &e[1]}, false);
        pack_split_bf16(ar, m->enc[l], {&e[0], &e[1]});
        if (!c.use_rc) continue;
        if (c.recurrent_type == 1) {
            // ConvGRU (submodules.py:348-376): three 3x3 convolutions on cat(x, h) / cat(x, h * reset); in-channel order [x | h].
            // x-parts (rows update | reset | out, with the biases) batched over T like the LSTM's; h-parts per step.
            DenseLayer gxd[2], gur[2], go[2];
            const char* gates[3] = {"update_gate", "reset_gate", "out_gate"};
            for (int d = 0; d < 2; ++d) {
                std::string p = std::string(dirs[d]) + "." + std::to_string(l) + ".recurrent_block.";
                gxd[d].rows = 3 * co; gxd[d].Cin = co; gxd[d].KS = 3;
                gur[d].rows = 2 * co; gur[d].Cin = co; gur[d].KS = 3;
                go[d].rows = co; go[d].Cin = co; go[d].KS = 3;
                for (int q = 0; q < 3; ++q) {
                    DenseLayer xs, hs;
                    BDE_TRY(dense_conv(m, p + gates[q] + ".weight", p + gates[q] + ".bias", co, 2 * co, 0, co, 3, true, &xs));
                    BDE_TRY(dense_conv(m, p + gates[q] + ".weight", p + gates[q] + ".bias", co, 2 * co, co, co, 3, false, &hs));
                    gxd[d].w.insert(gxd[d].w.end(), xs.w.begin(), xs.w.end());
                    gxd[d].bias.insert(gxd[d].bias.end(), xs.bias.begin(), xs.bias.end());
                    DenseLayer& hd = q < 2 ? gur[d] : go[d];
                    hd.w.insert(hd.w.end(), hs.w.begin(), hs.w.end());
                    hd.bias.insert(hd.bias.end(), hs.bias.begin(), hs.bias.end());
                }
            }
            m->gx[l] = pack_layer(ar, {&gxd[0], &gxd[1]}, false);
            pack_split_bf16(ar, m->gx[l], {&gxd[0], &gxd[1]});
            m->gru_ur[l] = pack_layer(ar, {&gur[0], &gur[1]}, false);
            m->gru_o[l] = pack_layer(ar, {&go[0], &go[1]}, false);
            continue;
        }
        DenseLayer gxd[2], gh[2];
        for (int d = 0; d < 2; ++d) {
            std::string p = std::string(dirs[d]) + "." + std::to_string(l) + ".";
            // Gates weight in-channel order is [x | h] (submodules.py:316)
            BDE_TRY(dense_conv(m, p + "recurrent_block.Gates.weight", p + "recurrent_block.Gates.bias", 4 * co, 2 * co, 0,
                               co, 3, true, &gxd[d]));
            BDE_TRY(dense_conv(m, p + "recurrent_block.Gates.weight", p + "recurrent_block.Gates.bias", 4 * co, 2 * co, co,
                               co, 3, false, &gh[d]));
        }
        m->gx[l] = pack_layer(ar, {&gxd[0], &gxd[1]}, false);
        pack_split_bf16(ar, m->gx[l], {&gxd[0], &gxd[1]});
        m->lstm[l] = pack_lstm16(ar, {&gh[0], &gh[1]});
        m->lstm8[l] = pack_lstm8(ar, {&gh[0], &gh[1]});
        {
            PackedLayer& ps = m->lstm_sb[l];
            ps.Cin = co; ps.Cout = 4 * co; ps.KS = 3; ps.G = 2;
            pack_split_bf16(ar, ps, {&gh[0], &gh[1]});
        }
        if (co % 16 == 0) {
            pack_lstm_sbk(ar, m->lstm_sbk[l], {&gh[0], &gh[1]});
            // ... and with the x-part in the same contraction: K = [x | h], the order of the reference's stacked input
            DenseLayer gf[2];
            for (int d = 0; d < 2; ++d) {
                std::string p = std::string(dirs[d]) + "." + std::to_string(l) + ".";
                BDE_TRY(dense_conv(m, p + "recurrent_block.Gates.weight", p + "recurrent_block.Gates.bias", 4 * co, 2 * co, 0,
                                   2 * co, 3, true, &gf[d]));
            }
            PackedLayer& px = m->lstm_sbx[l];
            pack_lstm_sbk(ar, px, {&gf[0], &gf[1]});
            px.b_off = ar.alloc(2L * 4 * co);
            for (int d = 0; d < 2; ++d) std::copy(gf[d].bias.begin(), gf[d].bias.end(), ar.host.begin() + px.b_off + (long)d * 4 * co);
        }
    }
    if (c.depths[L - 1] == 0) {
        // Sequential(ParseLayer, ResidualBlockNoBN x num_res_blocks) in place of the last level's attention (V5.py:77-80)
        const int C = m->cout(L - 1);
        for (int k = 0; k < c.num_res_blocks; ++k) {
            std::string p = "feat_attns." + std::to_string(L - 1) + "." + std::to_string(1 + k) + ".";
            DenseLayer c1, c2;
            BDE_TRY(dense_conv(m, p + "conv1.weight", p + "conv1.bias", C, C, 0, C, 3, true, &c1));
            BDE_TRY(dense_conv(m, p + "conv2.weight", p + "conv2.bias", C, C, 0, C, 3, true, &c2));
            m->rb1.push_back(pack_layer(ar, {&c1}, false));
            m->rb2.push_back(pack_layer(ar, {&c2}, false));
        }
    }
    const int D = c.frame_num, heads = c.num_heads;
    const int tbl_rows = (2 * D - 1) * 13 * 13;
    for (int l = 0; l < L; ++l) {
        AttnLevel& al = m->attn[l];
        al.depth = c.depths[l];
        al.C = m->cout(l);
        if (al.depth == 0) continue;
        const int C = al.C, hid = 4 * C, hd = C / heads;
        // softmax(x) = 2^(x*log2e - max): fold log2(e) into the query scale and the bias table (attn.h)
        const float LOG2E = 1.4426950408889634f;
        const float scale = LOG2E / std::sqrt((float)hd);
        DenseLayer kvall;
        kvall.rows = al.depth * 2 * C;
        kvall.Cin = C;
        kvall.KS = 1;
        kvall.w.resize((size_t)kvall.rows * C);
        kvall.bias.resize(kvall.rows);
        kvall.lnsum.resize(kvall.rows);
        al.blocks.resize(al.depth);
        for (int i = 0; i < al.depth; ++i) {
            AttnBlock& ab = al.blocks[i];
            std::string p = "feat_attns." + std::to_string(l) + ".blocks." + std::to_string(i) + ".";
            const float *tbl, *gq, *bq, *gkv, *bkv, *wq, *biq, *wkv, *bikv, *wp, *bp, *g2, *b2, *w1, *b1, *w2, *b2b;
            BDE_TRY(get_raw(m, p + "attn.relative_position_bias_table", {tbl_rows, heads}, &tbl));
            BDE_TRY(get_raw(m, p + "attn.norm_q.weight", {C}, &gq));
            BDE_TRY(get_raw(m, p + "attn.norm_q.bias", {C}, &bq));
            BDE_TRY(get_raw(m, p + "attn.norm_kv.weight", {C}, &gkv));
            BDE_TRY(get_raw(m, p + "attn.norm_kv.bias", {C}, &bkv));
            BDE_TRY(get_raw(m, p + "attn.q.weight", {C, C}, &wq));
            BDE_TRY(get_raw(m, p + "attn.q.bias", {C}, &biq));
            BDE_TRY(get_raw(m, p + "attn.kv.weight", {2 * C, C}, &wkv));
            BDE_TRY(get_raw(m, p + "attn.kv.bias", {2 * C}, &bikv));
            BDE_TRY(get_raw(m, p + "attn.proj.weight", {C, C}, &wp));
            BDE_TRY(get_raw(m, p + "attn.proj.bias", {C}, &bp));
            BDE_TRY(get_raw(m, p + "norm2.weight", {C}, &g2));
            BDE_TRY(get_raw(m, p + "norm2.bias", {C}, &b2));
            BDE_TRY(get_raw(m, p + "mlp.fc1.weight", {hid, C}, &w1));
            BDE_TRY(get_raw(m, p + "mlp.fc1.bias", {hid}, &b1));
            BDE_TRY(get_raw(m, p + "mlp.fc2.weight", {C, hid}, &w2));
            BDE_TRY(get_raw(m, p + "mlp.fc2.bias", {C}, &b2b));
            // q | k | v stacked: one GEMM on the query frame; the three LayerNorms share (mu, rstd)
            DenseLayer qkv;
            qkv.rows = 3 * C;
            qkv.Cin = C;
            qkv.KS = 1;
            qkv.w.resize((size_t)3 * C * C);
            qkv.bias.resize(3 * C);
            qkv.lnsum.resize(3 * C);
            fold_ln_rows(wq, biq, gq, bq, C, C, scale, &qkv, 0);
            fold_ln_rows(wkv, bikv, gkv, bkv, 2 * C, C, 1.f, &qkv, C);
            fold_ln_rows(wkv, bikv, gkv, bkv, 2 * C, C, 1.f, &kvall, i * 2 * C);
            ab.qkv = pack_layer(ar, {&qkv}, false);
            // K|V of an all-zero token: LayerNorm(0) = beta  ->  W beta + b  (DTransformer.py:183-190)
            ab.kvpad_off = ar.alloc(2 * C);
            std::copy(qkv.bias.begin() + C, qkv.bias.end(), ar.host.begin() + ab.kvpad_off);
            // dense relative-position bias of the query frame's rows, transposed to [head][n][m]
            // (DTransformer.py:139-153,195-199): index = ((dd+D-1)*13 + (dh+6))*13 + (dw+6)
            const int N = D * 49;
            ab.bias_off = ar.alloc((long)heads * N * 49);
            float* bt = ar.host.data() + ab.bias_off;
            for (int mq = 0; mq < 49; ++mq) {
                int qh = mq / 7, qw = mq % 7;
                for (int n = 0; n < N; ++n) {
                    int kd = n / 49, kh = (n % 49) / 7, kw = n % 7;
                    int idx = ((c.q_idx - kd + D - 1) * 13 + (qh - kh + 6)) * 13 + (qw - kw + 6);
                    for (int h = 0; h < heads; ++h) bt[((long)h * N + n) * 49 + mq] = LOG2E * tbl[(long)idx * heads + h];
                }
            }
            if (C == WB_C && heads == WB_NH && D <= WB_MAXD) {
                // winblock.h: keys reordered query frame first, score tile (query tile i, key tile j) in the
                // C/D register order of the 16x16x4 MFMA: [head][i][j][lane][r], key = 16j + 4(lane>>4) + r
                ab.biasF_off = ar.alloc((long)heads * 4 * WB_NT * 256);
                float* bfp = ar.host.data() + ab.biasF_off;
                bt = ar.host.data() + ab.bias_off;               // (the arena may have moved: alloc() grows a std::vector)
                for (int h = 0; h < heads; ++h)
                    for (int qi = 0; qi < 4; ++qi)
                        for (int j = 0; j < WB_NT; ++j)
                            for (int r = 0; r < 4; ++r)
                                for (int ln = 0; ln < 64; ++ln) {
                                    const int u = 16 * j + 4 * (ln >> 4) + r;
                                    const int mq = std::min(16 * qi + (ln & 15), 48);
                                    float v = -1e30f;
                                    if (u < N) {
                                        int n;                       // key row of the reference order (slot-major)
                                        if (u < 49) n = c.q_idx * 49 + u;
                                        else {
                                            const int w = u - 49;
                                            int d = w / 49;              // index among the non-query slots
                                            if (d >= c.q_idx) ++d;
                                            n = d * 49 + w % 49;
                                        }
                                        v = bt[((long)h * N + n) * 49 + mq];
                                    }
                                    bfp[((((long)h * 4 + qi) * WB_NT + j) * 64 + ln) * 4 + r] = v;
                                }
            }
            if (C % 64 == 0 && hd == 16 && N <= 160) {
                // wide_core.h: score tile (query tile qi, key tile j) in the C/D register order of the 16x16x4 MFMA, keys in the
                // reference's slot-major order: [head][qi][j][lane][r], key = 16 j + 4 (lane >> 4) + r, query = 16 qi + (lane & 15)
                ab.biasW_off = ar.alloc((long)heads * 4 * 10 * 256);
                float* bwp = ar.host.data() + ab.biasW_off;
                bt = ar.host.data() + ab.bias_off;
                for (int h = 0; h < heads; ++h)
                    for (int qi = 0; qi < 4; ++qi)
                        for (int j = 0; j < 10; ++j)
                            for (int ln = 0; ln < 64; ++ln)
                                for (int r = 0; r < 4; ++r) {
                                    const int u = 16 * j + 4 * (ln >> 4) + r;
                                    const int mq = std::min(16 * qi + (ln & 15), 48);
                                    bwp[((((long)h * 4 + qi) * 10 + j) * 64 + ln) * 4 + r] = u < N ? bt[((long)h * N + u) * 49 + mq] : -1e30f;
                                }
            }
            DenseLayer proj;
            proj.rows = C; proj.Cin = C; proj.KS = 1;
            proj.w.assign(wp, wp + (size_t)C * C);
            proj.bias.assign(bp, bp + C);
            ab.proj = pack_layer(ar, {&proj}, false);
            DenseLayer fc1;
            fc1.rows = hid; fc1.Cin = C; fc1.KS = 1;
            fc1.w.resize((size_t)hid * C);
            fc1.bias.resize(hid);
            fc1.lnsum.resize(hid);
            fold_ln_rows(w1, b1, g2, b2, hid, C, 1.f, &fc1, 0);
            ab.fc1 = pack_layer(ar, {&fc1}, false);
            DenseLayer fc2;
            fc2.rows = C; fc2.Cin = hid; fc2.KS = 1;
            fc2.w.assign(w2, w2 + (size_t)C * hid);
            fc2.bias.assign(b2b, b2b + C);
            ab.fc2 = pack_layer(ar, {&fc2}, false);
            if (C % 64 == 0 && hd == 16) {
                ab.projW = pack16x4(ar, proj.w.data(), C, C);
                ab.fc1W = pack16x4(ar, fc1.w.data(), hid, C);
                ab.fc2W = pack16x4(ar, fc2.w.data(), C, hid);
                ab.qkvW = pack16x4(ar, qkv.w.data(), 3 * C, C);
                ab.qkvHF_unscale = ar.alloc(4);
                ab.qkvHF = pack16_split_frag(ar, qkv.w.data(), 3 * C, C, ab.qkvHF_unscale);
                ab.mlpHF_unscale = ar.alloc(4);
                ab.projHF = pack16_split_frag(ar, proj.w.data(), C, C, ab.mlpHF_unscale);
                ab.fc1HF = pack16_split_frag(ar, fc1.w.data(), hid, C, ab.mlpHF_unscale + 1);
                ab.mlpN_unscale = ar.alloc(4);
                ab.fc1N = pack16_split(ar, fc1.w.data(), hid, C, 2, ab.mlpN_unscale);
                ab.fc2N = pack16_split(ar, fc2.w.data(), C, hid, 2, ab.mlpN_unscale + 1);
                ab.qkvN_unscale = ar.alloc(4);
                ab.qkvN = pack16_split(ar, qkv.w.data(), 3 * C, C, 2, ab.qkvN_unscale);
            }
            if (C == WB_C && heads == WB_NH && D <= WB_MAXD) {
                ab.projS = pack16_split(ar, proj.w.data(), C, C, 3, -1);
                ab.fc1S = pack16_split(ar, fc1.w.data(), hid, C, 3, -1);
                ab.fc2S = pack16_split(ar, fc2.w.data(), C, hid, 3, -1);
                ab.qkvS = pack16_split(ar, qkv.w.data(), 3 * C, C, 3, -1);
                ab.unscaleH = ar.alloc(4);
                ab.qkvH = pack16_split(ar, qkv.w.data(), 3 * C, C, 2, ab.unscaleH);
                ab.projH = pack16_split(ar, proj.w.data(), C, C, 2, ab.unscaleH + 1);
                ab.fc1H = pack16_split(ar, fc1.w.data(), hid, C, 2, ab.unscaleH + 2);
                ab.fc2H = pack16_split(ar, fc2.w.data(), C, hid, 2, ab.unscaleH + 3);
            }
            if (C % 16 == 0 && token_lds_bytes(C) <= 150 * 1024) {
                ab.proj16 = pack16(ar, proj.w.data(), C, C);
                ab.fc1_16 = pack16(ar, fc1.w.data(), hid, C);
                ab.fc2_16 = pack16(ar, fc2.w.data(), C, hid);
                ab.qkv16 = pack16(ar, qkv.w.data(), 3 * C, C);
            }
        }
        al.kvall = pack_layer(ar, {&kvall}, false);
        if (C % 64 == 0 && hd == 16) {
            al.kvallW = pack16x4(ar, kvall.w.data(), kvall.rows, C);
            al.kvallH_unscale = ar.alloc(4);
            al.kvallH = pack16_split_frag(ar, kvall.w.data(), kvall.rows, C, al.kvallH_unscale);
        }
    }
    for (int j = 0; j < L; ++j) {
        const int cin = m->cout(L - 1 - j), cout = m->cin(L - 1 - j);
        DenseLayer d;
        BDE_TRY(dense_convlayer(m, "decoders." + std::to_string(j) + ".1.", cout, cin, ks, &d));
        m->dec[j] = pack_layer(ar, {&d}, false);
        pack_split_bf16(ar, m->dec[j], {&d});
        if (c.skip_concat) {                        // 1x1 fusion of cat(skip, x) (V5.py:86-89)
            DenseLayer f;
            std::string p = "decoders." + std::to_string(j) + ".0.";
            BDE_TRY(dense_conv(m, p + "weight", p + "bias", cin, 2 * cin, 0, 2 * cin, 1, true, &f));
            m->dec_fuse[j] = pack_layer(ar, {&f}, false);
        }
    }
    if (c.skip_concat) {                            // V5.py:92-93
        DenseLayer f;
        BDE_TRY(dense_conv(m, "predI.0.weight", "predI.0.bias", bc, 2 * bc, 0, 2 * bc, 1, true, &f));
        m->pred_fuse = pack_layer(ar, {&f}, false);
    }
    {
        const float *w, *b;
        BDE_TRY(get_raw(m, "predI.1.weight", {1, bc, 1, 1}, &w));
        BDE_TRY(get_raw(m, "predI.1.bias", {1}, &b));
        m->predw_off = ar.alloc(bc);
        std::copy(w, w + bc, ar.host.begin() + m->predw_off);
        m->predb_off = ar.alloc(1);
        ar.host[m->predb_off] = b[0];
        m->zero_off = ar.alloc(64);                 // 256 bytes of zeros (conv_sb.h: out-of-image pixels)
    }
    return BDE_OK;
}

static int upload(bde_model* m) {
    // captured graphs hold pointers into the old packed image: drop them (and the workspaces) with it
    for (auto& w : m->wslots) w.release();
    if (m->dev) (void)hipFree(m->dev);
    m->dev = nullptr;
    m->dev_numel = (long)m->arena.host.size();
    BDE_HIP(hipMalloc((void**)&m->dev, sizeof(float) * m->dev_numel));
    BDE_HIP(hipMemcpy(m->dev, m->arena.host.data(), sizeof(float) * m->dev_numel, hipMemcpyHostToDevice));
    std::vector<float>().swap(m->arena.host);
    m->raw.clear();
    m->finalized = true;
    if (!m->ovf_dev) {                      // overflow words of the range guard (split.h), one per workspace slot
        BDE_HIP(hipMalloc((void**)&m->ovf_dev, sizeof(unsigned) * bde_model::MAX_SLOTS));
        BDE_HIP(hipMemset(m->ovf_dev, 0, sizeof(unsigned) * bde_model::MAX_SLOTS));
        BDE_HIP(hipHostMalloc((void**)&m->ovf_host, sizeof(unsigned) * bde_model::MAX_SLOTS, hipHostMallocDefault));
        for (int i = 0; i < bde_model::MAX_SLOTS; ++i) m->ovf_host[i] = 0;
    }
    return BDE_OK;
}

// ------------------------------------------------------------------------------------------
// conv launch helper
// ------------------------------------------------------------------------------------------
struct ConvCall {
    const PackedLayer* pl = nullptr;
    const float* in = nullptr;
    float* out = nullptr;
    const float* res1 = nullptr;
    const float* res2 = nullptr;
    int N = 1, Hs = 0, Ws = 0;   // input dims
    int stride = 1;
    int act = ACT_NONE;
    long in_gs = 0, out_gs = 0;  // group strides (0 = shared input)
    int mask_w = 0, mask_pt = 0, mask_pl = 0;
    int cout_rows = -1;          // override (kvall uses all rows)
    const float* pred_head = nullptr;   // fused predI (conv_mfma.h): set pred_out to enable
    float* pred_out = nullptr;
    float* out_sb = nullptr;     // store the result as SB16 here INSTEAD of fp32 planes in `out` (conv_mfma.h sb_out)
    long out_sb_gs = 0;          // its group stride, floats
    bool in_sb = false;          // `in` already is the SB16 image (in_gs in floats of that image): conv_sb or fail
    int decide_N = 0;            // > 0: choose the kernel as for a launch of this many frames (a chunk of a batched launch computes
                                 // exactly what the whole launch computes for its frames)
};

// Will run_conv take the split-bf16 kernels for this layer at this size?  (decided before the producer of its input runs)
static bool conv_takes_sb(const bde_model* m, const PackedLayer& pl, int stride, int N, int Hs, int Ws) {
    const int pad = pl.KS / 2;
    const int Ho = (Hs + 2 * pad - pl.KS) / stride + 1, Wo = (Ws + 2 * pad - pl.KS) / stride + 1;
    return m->conv_sb && pl.split_off(m->sb_terms) >= 0 && conv_sb_fits(pl.KS, stride, pl.Cout, Ws, Ho, Wo, m->sb_terms) &&
           (long)(pl.G_decide ? pl.G_decide : pl.G) * N * Ho * Wo >= 16384;
}

static int run_conv(const bde_model* m, const ConvCall& cc, hipStream_t s) {
    const PackedLayer& pl = *cc.pl;
    ConvArgs a;
    memset(&a, 0, sizeof a);
    a.in = cc.in;
    a.wpk = m->P(pl.w_off);
    a.bias = m->P(pl.b_off);
    a.lnsum = pl.s_off >= 0 ? m->P(pl.s_off) : nullptr;
    a.res1 = cc.res1;
    a.res2 = cc.res2;
    a.out = cc.out;
    a.N = cc.N;
    a.Cin = pl.Cin;
    a.Hs = cc.Hs;
    a.Ws = cc.Ws;
    a.Hin = cc.Hs;
    a.Win = cc.Ws;
    a.Cout = pl.Cout;
    const int pad = pl.KS / 2;
    a.Ho = (a.Hin + 2 * pad - pl.KS) / cc.stride + 1;
    a.Wo = (a.Win + 2 * pad - pl.KS) / cc.stride + 1;
    a.nchunks = pl.nchunks;
    a.act = cc.act;
    a.mask_w = cc.mask_w;
    a.mask_pt = cc.mask_pt;
    a.mask_pl = cc.mask_pl;
    const long in_fs = (long)pl.Cin * cc.Hs * cc.Ws, out_fs = (long)pl.Cout * a.Ho * a.Wo;
    a.in_ns = in_fs;
    a.out_ns = a.res1_ns = a.res2_ns = out_fs;
    a.in_gs = cc.in_gs;
    a.out_gs = cc.out_gs;
    a.res1_gs = a.res2_gs = cc.out_gs;
    a.w_gs = pl.w_sz;
    a.bias_gs = pl.Cout;
    a.xcd_remap = m->xcd_remap;
    a.decide_groups = pl.G_decide;
    if (cc.out_sb) {
        a.sb_out = reinterpret_cast<unsigned short*>(cc.out_sb);
        a.sb_out_ns = (long)cdiv(pl.Cout, 16) * a.Ho * a.Wo * (16 * m->sb_terms);
        a.sb_out_gs = cc.out_sb_gs * 2;
        a.sb_ovf = m->ovf();
    }
    a.sb_terms = m->sb_terms;
    if (cc.pred_out) {
        a.pred_w = m->P(m->predw_off);
        a.pred_b = m->P(m->predb_off);
        a.pred_head = cc.pred_head;
        a.pred_out = cc.pred_out;
        a.pred_sigmoid = m->cfg.activation;
    }
    if (pl.KS == 1) return pw_launch_auto(a, pl.G, s);
    // (measured at the canonical sizes, us: 3x3 gate convs 530 / 497 / 505 against 954 / 954 / 989 on the fp32 matrix path;
    //  5x5: decoder 0 488 vs 633, decoder 1 (64 channels) 502 vs 641, encoder 1 / 2 (stride 2) 271 / 300 vs 345 / 335;
    //  conv_sb_pick has no shape for 32 output channels or for the stride-2 halo of level 0, those stay on the fp32 kernels)
    // (the fused predI epilogue needs every output channel of a pixel in one wave: 32 channels)
    if (cc.in_sb || ((!cc.pred_out || pl.Cout <= 32) && conv_takes_sb(m, pl, cc.stride, cc.decide_N > 0 ? cc.decide_N : cc.N, cc.Hs, cc.Ws))) {
        // split the input into three bf16 terms (SB16) unless its producer already wrote it that way, then the convolution
        // on the bf16 matrix cores; the small launches (a few frames of a small map) stay on the fp32 kernels
        Workspace& ws = const_cast<bde_model*>(m)->W();
        const bool grouped_in = cc.in_gs != 0;
        const long frames = (grouped_in ? pl.G : 1) * (long)cc.N;
        const long need = split_bf16_bytes(frames, pl.Cin, (long)cc.Hs * cc.Ws);
        if (cc.in_sb || (ws.sb && need <= ws.sb_bytes)) {
            if (!cc.in_sb) BDE_TRY(split_bf16(cc.in, ws.sb, frames, pl.Cin, (long)cc.Hs * cc.Ws, m->sb_terms, m->ovf(), s));
            ConvArgs b = a;
            b.in = cc.in_sb ? cc.in : ws.sb;
            b.wpk = m->P(pl.split_off(m->sb_terms));
            b.w_gs = pl.split_sz(m->sb_terms);
            b.acc_scale = m->P(pl.sh_unscale_off);
            b.nchunks = pl.sb_chunks;
            b.zeros = m->P(m->zero_off);
            b.in_ns = (long)pl.sb_chunks * cc.Hs * cc.Ws * sb_pix_bytes(m->sb_terms) / 4;
            b.in_gs = cc.in_sb ? cc.in_gs : (grouped_in ? b.in_ns * cc.N : 0);
            bool launched = false;
            BDE_TRY(conv_sb_launch(pl.KS, cc.stride, b, pl.G, s, &launched));
            pl.sb_used = launched ? 1 : 0;
            if (launched) return BDE_OK;
        }
        if (cc.in_sb) return fail(BDE_ERR_UNSUPPORTED, "convolution on a split-bf16 input has no split-bf16 launch at this size");
    }
    pl.sb_used = 0;
    return conv_launch_best(pl.KS, cc.stride, a, pl.G, s);
}

// 1x1 conv over flattened [C][HW] planes
static int run_pw(const bde_model* m, const PackedLayer* pl, const float* in, float* out, int N, long HW, int act,
                  const float* res1, const float* res2, int mask_w, int mask_pt, int mask_pl, hipStream_t s) {
    ConvCall cc;
    cc.pl = pl;
    cc.in = in;
    cc.out = out;
    cc.N = N;
    cc.Hs = 1;
    cc.Ws = (int)HW;
    cc.act = act;
    cc.res1 = res1;
    cc.res2 = res2;
    cc.mask_w = mask_w;
    cc.mask_pt = mask_pt;
    cc.mask_pl = mask_pl;
    return run_conv(m, cc, s);
}

// ------------------------------------------------------------------------------------------
// workspace
// ------------------------------------------------------------------------------------------
static int ws_alloc(Workspace& ws, float** p, long numel) {
    void* q = nullptr;
    BDE_HIP(hipMalloc(&q, sizeof(float) * (size_t)std::max<long>(numel, 4)));
    ws.allocs.push_back(q);
    *p = (float*)q;
    return BDE_OK;
}

static bool winblock_ok(const bde_model* m, int l);
static bool wide_ok(const bde_model* m, int l);
static bool wide_core2_ok(const bde_model* m, int l);
static bool lstm_sb_ok(const bde_model* m, int l, int B, int h, int w);
static bool lstm_sbk_ok(const bde_model* m, int l, int h, int w);
static bool lstm_sbx_ok(const bde_model* m, int l, int h, int w);

static int ensure_workspace(bde_model* m, int T, int B, int H, int W) {
    Workspace& ws = m->W();
    if (ws.T == T && ws.B == B && ws.H == H && ws.W == W) return BDE_OK;
    ws.release();
    const bde_config& c = m->cfg;
    const int L = c.num_encoders;
    const long TB = (long)T * B;
    BDE_TRY(ws_alloc(ws, &ws.ev, TB * c.num_bins * H * W));
    BDE_TRY(ws_alloc(ws, &ws.head, TB * c.basechannels * H * W));
    BDE_TRY(ws_alloc(ws, &ws.out, TB * H * W));
    ws.xenc.assign(L, nullptr); ws.gx.assign(L, nullptr); ws.hseq.assign(L, nullptr); ws.cst.assign(L, nullptr);
    ws.hsb.assign(L, nullptr); ws.ghb.assign(L, nullptr); ws.hsk.assign(L, nullptr);
    ws.gur.assign(L, nullptr); ws.ghr.assign(L, nullptr); ws.gou.assign(L, nullptr); ws.gub.assign(L, nullptr);
    const bool gru = c.use_rc && c.recurrent_type == 1;
    ws.mergedS.assign(L, nullptr); ws.mstats.assign(L, nullptr);
    ws.merged.assign(L, nullptr); ws.mergedT.assign(L, nullptr); ws.kvun.assign(L, nullptr); ws.kvref.assign(L, nullptr); ws.dec.assign(L, nullptr); ws.qkv0.assign(L, nullptr);
    long max_attn = 0;
    for (int l = 0; l < L; ++l) {
        const long C = m->cout(l), hw = (long)(H >> (l + 1)) * (W >> (l + 1));
        const long hwp = cdivl(hw, 16) * 16;              // token tiles of 16 (wideblock.h)
        BDE_TRY(ws_alloc(ws, &ws.xenc[l], 2 * TB * C * hw));
        BDE_TRY(ws_alloc(ws, &ws.hseq[l], 2 * TB * C * hw));
        BDE_TRY(ws_alloc(ws, &ws.cst[l], 2 * (long)B * C * hw));
        if (gru) {
            BDE_TRY(ws_alloc(ws, &ws.gur[l], 2L * B * 2 * C * hw));
            BDE_TRY(ws_alloc(ws, &ws.ghr[l], 2L * B * C * hw));
            BDE_TRY(ws_alloc(ws, &ws.gou[l], 2L * B * C * hw));
            BDE_TRY(ws_alloc(ws, &ws.gub[l], 2L * B * C * hw));
        }
        if (!gru && c.use_rc && lstm_sbk_ok(m, l, H >> (l + 1), W >> (l + 1)))
            BDE_TRY(ws_alloc(ws, &ws.hsk[l], 2 * split_bf16_bytes(2L * B, (int)C, hw) / 4 + 4));
        // x-part of the gates of all frames, both directions -- the largest buffer of a level (20 GB at 480 x 640, T = 32, B = 4);
        // not needed where the recurrent step contracts [x | h] itself (lstm_fuse_x)
        if (!lstm_sbx_ok(m, l, H >> (l + 1), W >> (l + 1))) BDE_TRY(ws_alloc(ws, &ws.gx[l], 2 * TB * 4 * C * hw));
        if (!gru && c.use_rc && lstm_sb_ok(m, l, B, H >> (l + 1), W >> (l + 1))) {
            BDE_TRY(ws_alloc(ws, &ws.hsb[l], 2 * split_bf16_bytes(2L * B, (int)C, hw) / 4 + 4));
            BDE_TRY(ws_alloc(ws, &ws.ghb[l], 2L * B * 4 * C * hw));
        }
        BDE_TRY(ws_alloc(ws, &ws.merged[l], TB * C * hw));
        if (c.depths[l] > 0) {
            if (winblock_ok(m, l)) {
                // one-launch blocks recompute the neighbours' K|V: only the token-major twin of merged is staged
                // (the K|V stacks of the split path are 71 GB at 1280x720, T = 64)
                BDE_TRY(ws_alloc(ws, &ws.mergedT[l], TB * C * hw));
            } else if (wide_ok(m, l)) {
                // fragment-layout twin of merged (token tiles of 16) + token-major K|V stacks and first-block q|k|v
                BDE_TRY(ws_alloc(ws, &ws.mergedT[l], TB * C * hwp));
                if (wide_core2_ok(m, l)) {                    // ... and its SPL16 twin + statistics (wide_core.h)
                    BDE_TRY(ws_alloc(ws, &ws.mergedS[l], TB * C * hwp));
                    BDE_TRY(ws_alloc(ws, &ws.mstats[l], TB * hwp * 2));
                }
                BDE_TRY(ws_alloc(ws, &ws.kvun[l], TB * c.depths[l] * 2 * C * hw));
                BDE_TRY(ws_alloc(ws, &ws.kvref[l], TB * c.depths[l] * 2 * C * hw));
                BDE_TRY(ws_alloc(ws, &ws.qkv0[l], TB * 3 * C * hw));
            } else {
                BDE_TRY(ws_alloc(ws, &ws.kvun[l], TB * c.depths[l] * 2 * C * hw));
                BDE_TRY(ws_alloc(ws, &ws.kvref[l], TB * c.depths[l] * 2 * C * hw));
                BDE_TRY(ws_alloc(ws, &ws.qkv0[l], TB * 3 * C * hw));
            }
            max_attn = std::max(max_attn, (long)B * C * hwp);
        }
        const int j = L - 1 - l;   // decoder j writes the map of level (L-1-j)'s input resolution
        BDE_TRY(ws_alloc(ws, &ws.dec[j], TB * m->cin(l) * (long)(H >> l) * (W >> l)));
    }
    BDE_TRY(ws_alloc(ws, &ws.up, TB * m->cout(0) * (long)H * W));   // dec L-1: cout(0) channels at full resolution
    if (c.skip_concat) {
        // largest cat(skip, x): decoder inputs 2 * cout(l) at level l, predI input 2 * basechannels at full resolution
        long mc = TB * 2 * c.basechannels * (long)H * W;
        for (int l = 0; l < L; ++l) mc = std::max(mc, TB * 2 * m->cout(l) * (long)(H >> (l + 1)) * (W >> (l + 1)));
        BDE_TRY(ws_alloc(ws, &ws.cat, mc));
        BDE_TRY(ws_alloc(ws, &ws.fuse, mc / 2));
    }
    if (c.depths[L - 1] == 0) {
        const long n = (long)B * m->cout(L - 1) * (long)(H >> L) * (W >> L);
        BDE_TRY(ws_alloc(ws, &ws.rbA, n));
        BDE_TRY(ws_alloc(ws, &ws.rbX[0], n));
        BDE_TRY(ws_alloc(ws, &ws.rbX[1], n));
        BDE_TRY(ws_alloc(ws, &ws.zero_l, n));
        BDE_HIP(hipMemset(ws.zero_l, 0, sizeof(float) * n));
    }
    {
        // split-bf16 image of one convolution's input (6 B per element, channels padded to 16): the largest of the
        // encoder inputs, gate-conv inputs (both directions) and upsampled decoder inputs
        long mx = 0;
        for (int l = 0; l < L; ++l) {
            const long hw_in = (long)(H >> l) * (W >> l), hw = (long)(H >> (l + 1)) * (W >> (l + 1));
            mx = std::max(mx, split_bf16_bytes(TB, m->cin(l), hw_in));               // encoder conv input
            mx = std::max(mx, split_bf16_bytes(2 * TB, m->cout(l), hw));             // gate conv input, both directions
            mx = std::max(mx, split_bf16_bytes(TB, m->cout(l), 4 * hw));             // decoder conv input (upsampled)
        }
        ws.sb_bytes = mx;
        BDE_TRY(ws_alloc(ws, &ws.sb, mx / 4 + 4));
        long mx2 = 0;
        for (int l = 0; l < L; ++l) mx2 = std::max(mx2, split_bf16_bytes(2 * TB, m->cout(l), (long)(H >> (l + 1)) * (W >> (l + 1))));
        ws.sb2_bytes = mx2;
        BDE_TRY(ws_alloc(ws, &ws.sb2, mx2 / 4 + 4));
    }
    if (max_attn > 0) {
        BDE_TRY(ws_alloc(ws, &ws.qkv, 3 * max_attn));
        BDE_TRY(ws_alloc(ws, &ws.ao, max_attn));
        BDE_TRY(ws_alloc(ws, &ws.x1, max_attn));
        BDE_TRY(ws_alloc(ws, &ws.hid, 4 * max_attn));
        BDE_TRY(ws_alloc(ws, &ws.xa, max_attn));
        BDE_TRY(ws_alloc(ws, &ws.xb, max_attn));
        BDE_TRY(ws_alloc(ws, &ws.xaS, max_attn));
        BDE_TRY(ws_alloc(ws, &ws.xbS, max_attn));
        BDE_TRY(ws_alloc(ws, &ws.stA, max_attn / 64 + 64));       // (two floats per token of at least 128 channels)
        BDE_TRY(ws_alloc(ws, &ws.stB, max_attn / 64 + 64));
        float* cnt = nullptr;                              // one counter per 16 tokens of the largest attention frame
        BDE_TRY(ws_alloc(ws, &cnt, max_attn / 16 + 64));
        BDE_HIP(hipMemset(cnt, 0, sizeof(int) * (size_t)(max_attn / 16 + 64)));
        ws.tile_count = reinterpret_cast<int*>(cnt);
    }
    ws.T = T; ws.B = B; ws.H = H; ws.W = W;
    return BDE_OK;
}

// ------------------------------------------------------------------------------------------
// stages
// ------------------------------------------------------------------------------------------
// RecurrentConv sweep of one level for both directions (V5.py:122-135; submodules.py:191-195).
//   in: [TB][Cin][H][W].  Results: ws.hseq[l] = [2][TB][C][h][w]; ws.cst[l] final cell states.
// dir_mask: bit0 forward, bit1 backward (the op-level test runs a single direction).
// Non-recurrent part of a level for frames [f0, f0+nf) of the [TB] stack: encoder conv (both
// directions read the same sequence, V5.py:124-130) and the x-part of the gates.
static int run_enc_gx(bde_model* m, int l, const float* in, int f0, int nf, int T, int B, int H, int W, hipStream_t s) {
    Workspace& ws = m->W();
    const int Cin = m->cin(l), C = m->cout(l), h = H / 2, w = W / 2;
    const long TB = (long)T * B, hw = (long)h * w;
    // one direction only (bde_split_sweep): one-group views of the layers, pointers moved to that direction's half
    const int dmask = m->dir_mask, dsel = dmask == 2 ? 1 : 0;
    const bool one_dir = dmask != 3;
    const PackedLayer enc_v = one_dir ? group_view(m->enc[l], dsel) : m->enc[l];
    const PackedLayer gx_v = (one_dir && m->cfg.use_rc) ? group_view(m->gx[l], dsel) : m->gx[l];
    ConvCall e;
    e.pl = &enc_v;
    e.in = in + (long)f0 * Cin * H * W;
    e.out = ws.xenc[l] + (long)f0 * C * hw;
    e.N = nf;
    e.Hs = H;
    e.Ws = W;
    e.stride = 2;
    e.act = ACT_RELU;
    e.in_gs = 0;
    e.out_gs = TB * C * hw;
    if (!m->cfg.use_rc) {
        // bare ConvLayer encoders (V5.py:256-258): the convolution's output IS the level's feature sequence
        e.out = ws.hseq[l] + (long)f0 * C * hw + (one_dir ? dsel * e.out_gs : 0);
        ProfScope ps(m, pname("enc_conv", l), s);
        const int st = run_conv(m, e, s);
        m->enc[l].sb_used = enc_v.sb_used;
        return st;
    }
    if (one_dir) e.out += dsel * e.out_gs;
    // the gate convolution reads its input as SB16 (conv_sb.h): the encoder conv's epilogue then writes that image directly
    // (6 B per element) and the fp32 planes + their conversion pass are skipped
    const long sb_fs = (long)cdiv(C, 16) * hw * sb_pix_bytes(m->sb_terms) / 4;   // floats of one SB16 frame
    // (lstm_fuse_x: the recurrent step contracts [x | h] itself and reads x from that image: no gate convolution at all)
    const bool step_x = lstm_sbx_ok(m, l, h, w);
    const bool fuse = (step_x || (m->fuse_enc_sb && C % 32 == 0 && conv_takes_sb(m, gx_v, 1, nf, h, w))) &&
                      ws.sb2 && split_bf16_bytes(2 * TB, C, hw) <= ws.sb2_bytes;
    BDE_REQUIRE(fuse || !step_x, "recurrent step with the x-part: no room for the split image of x");
    if (fuse) {
        e.out_sb = ws.sb2 + (long)f0 * sb_fs + (one_dir ? dsel * TB * sb_fs : 0);
        e.out_sb_gs = TB * sb_fs;
    }
    { ProfScope ps(m, pname("enc_conv", l), s); BDE_TRY(run_conv(m, e, s)); }
    m->enc[l].sb_used = enc_v.sb_used;                                   // (the launch ran on a copy / one-direction view of the layer)
    if (step_x) { m->gx[l].sb_used = 0; return BDE_OK; }
    // gx = conv3x3(x; W[:, :C]) + bias   (submodules.py:316-317, x half of the stacked input)
    BDE_REQUIRE(ws.gx[l] != nullptr, "gate convolution: no x-part buffer at level %d", l);
    ConvCall gxc;
    gxc.pl = &gx_v;
    gxc.in = ws.xenc[l] + (long)f0 * C * hw + (one_dir ? dsel * TB * C * hw : 0);
    gxc.out = ws.gx[l] + (long)f0 * m->gx[l].Cout * hw + (one_dir ? dsel * TB * 4 * C * hw : 0);   // rows: 4C (ConvLSTM gates) or 3C (ConvGRU)
    gxc.N = nf;
    gxc.Hs = h;
    gxc.Ws = w;
    gxc.in_gs = TB * C * hw;
    gxc.out_gs = TB * 4 * C * hw;
    if (fuse) {
        gxc.in = ws.sb2 + (long)f0 * sb_fs + (one_dir ? dsel * TB * sb_fs : 0);
        gxc.in_gs = TB * sb_fs;
        gxc.in_sb = true;
    }
    { ProfScope ps(m, pname("gates_x", l), s); BDE_TRY(run_conv(m, gxc, s)); }
    m->gx[l].sb_used = gx_v.sb_used;
    return BDE_OK;
}

// The recurrent step on the bf16 matrix cores: h-part of the gates by conv_sb_kernel on the SB16 image of h_prev, then the
// pointwise tail as an element-wise kernel that also writes the next step's SB16 h.  Built, parity-tested
// (set_tuning("lstm_sb", 1)) and measured at the canonical config: a step is one small launch (368 / 176 / 96 workgroups at
// levels 0 / 1 / 2), each workgroup walks its 4 / 8 / 16 channel chunks with the halo staging exposed (one or two
// workgroups per CU; two LDS buffers leave a single workgroup per CU and two rounds at level 0): 69 us of convolution +
// 16.5 us of pointwise kernel per step against 65 us for lstm16_step_kernel.  Off by default.
static bool lstm_sb_ok(const bde_model* m, int l, int B, int h, int w) {
    if (m->lstm_sb_mode == 0 || !m->conv_sb) return false;
    const int C = m->cout(l);
    if (m->lstm_sb_off(l) < 0 || !conv_sb_fits(3, 1, 4 * C, w, h, w, m->sb_terms)) return false;
    if (m->lstm_sb_mode == 1) return true;
    const long wgs = cdivl((long)h * w, 128) * cdivl(4 * C, 128) * 2 * B;
    return wgs >= 160 && cdiv(C, 16) <= 8;
}

// The fused split-bf16 step (lstm_sb.h): one launch per time step, h carried as SB16 between steps (two buffers).
static bool lstm_sbk_ok(const bde_model* m, int l, int h, int w) {
    if (!m->use_lstm_sbk || m->lstm_sb_mode != 0 || !m->conv_sb) return false;
    if ((size_t)l >= m->lstm_sbk.size() || m->lstm_sbk[l].sb_off < 0) return false;
    return lstm_sb_shape(m->cout(l), h, w, m->sb_terms).ok;
}

// ... with the x-part of the gates in the same contraction: the encoder convolution of the level leaves x as an SB16 image
// (its epilogue writes it, any kernel), 32 | C so that whole 32-channel tiles are written
static bool lstm_sbx_ok(const bde_model* m, int l, int h, int w) {
    if (!m->lstm_fuse_x || !lstm_sbk_ok(m, l, h, w)) return false;
    const Workspace& ws = const_cast<bde_model*>(m)->W();
    if ((size_t)l >= ws.hsk.size() || ws.hsk[l] == nullptr) return false;      // the step that will run is not the split one
    return (size_t)l < m->lstm_sbx.size() && m->lstm_sbx[l].split_off(m->sb_terms) >= 0 && m->cout(l) % 32 == 0;
}

static int run_recurrent_steps_sbk(bde_model* m, int l, int T, int B, int h, int w, hipStream_t s) {
    Workspace& ws = m->W();
    const int C = m->cout(l);
    const long TB = (long)T * B, hw = (long)h * w;
    const bool step_x = lstm_sbx_ok(m, l, h, w);
    const PackedLayer& pl = step_x ? m->lstm_sbx[l] : m->lstm_sbk[l];
    float* hs = ws.hseq[l];
    const long dstride = TB * C * hw, fs = (long)B * C * hw;
    const int terms = m->sb_terms;
    const long sb_ns = (long)cdiv(C, 16) * hw * sb_pix_bytes(terms);       // bytes of one frame's SB16 image
    const long sb_buf = 2L * B * sb_ns;                                    // one buffer: both directions
    const int dsel = m->dir_mask == 2 ? 1 : 0;
    const bool one_dir = m->dir_mask != 3;
    unsigned char* hsk = reinterpret_cast<unsigned char*>(ws.hsk[l]);
    for (int st = 0; st < T; ++st) {
        const int tf = st, tb = T - 1 - st;
        ProfScope ps(m, pname("lstm", l), s);
        LstmSbArgs a;
        memset(&a, 0, sizeof a);
        a.hin = hsk + ((st + 1) & 1) * sb_buf;
        a.hsb = hsk + (st & 1) * sb_buf;
        a.hin_gs = (long)B * sb_ns;
        a.hin_ns = sb_ns;
        a.terms = terms;
        a.acc_scale = m->P(pl.sh_unscale_off);
        a.wpk = reinterpret_cast<const unsigned short*>(m->P(pl.split_off(terms)));
        a.w_gs = pl.split_sz(terms) * 2;
        if (!step_x) {
            BDE_REQUIRE(ws.gx[l] != nullptr, "recurrent step: no x-part buffer at level %d", l);
            a.gx = ws.gx[l] + (long)tf * B * 4 * C * hw;
            a.gx_gs = (ws.gx[l] + TB * 4 * C * hw + (long)tb * B * 4 * C * hw) - a.gx;
            a.gx_ns = (long)4 * C * hw;
        } else {
            // x_t of both directions in ws.sb2 (run_enc_gx): [direction][TB frames], forward reads frame tf, backward frame tb
            const unsigned char* xs = reinterpret_cast<const unsigned char*>(ws.sb2);
            a.xin = xs + (long)tf * B * sb_ns;
            a.xin_gs = (TB + (long)tb * B - (long)tf * B) * sb_ns;
            a.xin_ns = sb_ns;
            a.xchunks = C / 16;
            a.bias = m->P(pl.b_off);
        }
        a.cstate = ws.cst[l];
        a.c_gs = (long)B * C * hw;
        a.c_ns = (long)C * hw;
        a.hout = hs + (long)tf * fs;
        a.ho_gs = (hs + dstride + (long)tb * fs) - a.hout;
        a.ho_ns = (long)C * hw;
        a.zeros = m->P(m->zero_off);
        a.B = B; a.Ch = C; a.H = h; a.W = w;
        a.first = st == 0;
        a.stamps = (st == T / 2) ? m->tok_stamps : nullptr;                // diagnostics (bde_debug_token_stamps): one mid-sweep step
        a.stamp_mode = m->tok_debug == 9 ? 1 : 0;
        if (one_dir) {                                                     // group 0 of a one-group launch = the chosen direction
            a.hin += dsel * a.hin_gs; a.hsb += dsel * a.hin_gs;
            a.wpk += dsel * a.w_gs;
            if (step_x) { a.xin += dsel * a.xin_gs; a.bias += (long)dsel * 4 * C; }
            a.gx += dsel * a.gx_gs;
            a.cstate += dsel * a.c_gs;
            a.hout += dsel * a.ho_gs;
        }
        BDE_TRY(lstm_sb_step_launch(a, one_dir ? 1 : 2, s));
    }
    return BDE_OK;
}

static int run_recurrent_steps_sb(bde_model* m, int l, int T, int B, int h, int w, hipStream_t s) {
    Workspace& ws = m->W();
    const int C = m->cout(l);
    const long TB = (long)T * B, hw = (long)h * w;
    const PackedLayer& pl = m->lstm_sb[l];
    float* hs = ws.hseq[l];
    const long dstride = TB * C * hw, fs = (long)B * C * hw;
    const int terms = m->sb_terms;
    const long sbf = split_bf16_bytes(2L * B, C, hw, terms) / 4;   // floats of one SB16 hidden-state buffer (both directions)
    for (int st = 0; st < T; ++st) {
        const int tf = st, tb = T - 1 - st;
        ProfScope ps(m, pname("lstm", l), s);
        float* hsb_prev = ws.hsb[l] + ((st + 1) & 1) * sbf;
        float* hsb_next = ws.hsb[l] + (st & 1) * sbf;
        if (st > 0) {
            ConvArgs a;
            memset(&a, 0, sizeof a);
            a.in = hsb_prev;
            a.in_ns = (long)pl.sb_chunks * hw * sb_pix_bytes(terms) / 4;
            a.in_gs = a.in_ns * B;
            a.sb_terms = terms;
            a.acc_scale = m->P(pl.sh_unscale_off);
            a.wpk = m->P(pl.split_off(terms));
            a.w_gs = pl.split_sz(terms);
            a.bias = m->P(m->zero_off_long());                       // the gates' bias rides in gx
            a.bias_gs = 0;
            a.out = ws.ghb[l];
            a.out_ns = (long)4 * C * hw;
            a.out_gs = a.out_ns * B;
            a.res1_ns = a.res2_ns = a.out_ns;
            a.N = B; a.Cin = C; a.Hin = a.Hs = h; a.Win = a.Ws = w; a.Cout = 4 * C; a.Ho = h; a.Wo = w;
            a.nchunks = pl.sb_chunks;
            a.act = ACT_NONE;
            a.zeros = m->P(m->zero_off);
            bool launched = false;
            BDE_TRY(conv_sb_launch(3, 1, a, 2, s, &launched));
            BDE_REQUIRE(launched, "recurrent step: the split-bf16 convolution does not fit %dx%d", h, w);
        }
        LstmPointArgs p;
        memset(&p, 0, sizeof p);
        p.gx = ws.gx[l] + (long)tf * B * 4 * C * hw;
        p.gx_gs = (ws.gx[l] + TB * 4 * C * hw + (long)tb * B * 4 * C * hw) - p.gx;
        p.gx_ns = (long)4 * C * hw;
        p.gh = st > 0 ? ws.ghb[l] : nullptr;
        p.cstate = ws.cst[l];
        p.hout = hs + (long)tf * fs;
        p.h_gs = (hs + dstride + (long)tb * fs) - p.hout;
        p.h_ns = (long)C * hw;
        p.hsb = reinterpret_cast<unsigned short*>(hsb_next);
        p.C = C; p.B = B; p.HW = hw;
        p.first = st == 0;
        p.terms = terms;
        BDE_TRY(lstm_point_launch(p, s));
    }
    return BDE_OK;
}

// ConvGRU sweep of a level, both directions per launch (submodules.py:358-376; RecurrentConv.forward :191-195 returns the
// state itself).  Per step: h-parts of update | reset (one 3x3 convolution, 2C rows), gates + h * reset, h-part of the
// candidate on h * reset, blend.  The x-parts sit in ws.gx[l] as [2][TB][3C][hw] (update | reset | out).
static int run_gru_steps(bde_model* m, int l, int T, int B, int h, int w, hipStream_t s) {
    Workspace& ws = m->W();
    const int C = m->cout(l);
    const long TB = (long)T * B, hw = (long)h * w;
    float* hs = ws.hseq[l];
    const long dstride = TB * C * hw, fs = (long)B * C * hw;
    const int dsel = m->dir_mask == 2 ? 1 : 0;
    const bool one_dir = m->dir_mask != 3;
    const int G = one_dir ? 1 : 2;
    const PackedLayer ur_v = one_dir ? group_view(m->gru_ur[l], dsel) : m->gru_ur[l];
    const PackedLayer o_v = one_dir ? group_view(m->gru_o[l], dsel) : m->gru_o[l];
    const long total = (long)G * B * C * hw;
    const unsigned blocks = (unsigned)std::min<long>(cdivl(total, 256), 4096);
    for (int st = 0; st < T; ++st) {
        const int tf = st, tb = T - 1 - st;
        ProfScope ps(m, pname("gru", l), s);
        const float* hprev_f = hs + (long)(tf - 1) * fs;
        const float* hprev_b = hs + dstride + (long)(tb + 1) * fs;
        GruArgs g;
        memset(&g, 0, sizeof g);
        g.gx = ws.gx[l] + (long)tf * B * 3 * C * hw;
        g.gx_gs = (ws.gx[l] + TB * 4 * C * hw + (long)tb * B * 3 * C * hw) - g.gx;
        g.gx_ns = 3L * C * hw;
        g.ubuf = ws.gub[l];
        g.hr = ws.ghr[l];
        g.hout = hs + (long)tf * fs;
        g.ho_gs = (hs + dstride + (long)tb * fs) - g.hout;
        g.ho_ns = (long)C * hw;
        g.C = C; g.B = B; g.HW = hw; g.G = G;
        if (one_dir) {                                                 // group 0 of the launches = the chosen direction
            g.gx += dsel * g.gx_gs;
            g.hout += dsel * g.ho_gs;
        }
        if (st > 0) {
            g.hprev = one_dir && dsel ? hprev_b : hprev_f;
            g.hp_gs = hprev_b - hprev_f;
            g.hp_ns = (long)C * hw;
            ConvCall ur;
            ur.pl = &ur_v;
            ur.in = g.hprev; ur.in_gs = hprev_b - hprev_f;
            ur.out = ws.gur[l]; ur.out_gs = (long)B * 2 * C * hw;
            ur.N = B; ur.Hs = h; ur.Ws = w;
            BDE_TRY(run_conv(m, ur, s));
            g.gh_ur = ws.gur[l];
        }
        hipLaunchKernelGGL(gru_gate_kernel, dim3(blocks), dim3(256), 0, s, g);
        if (st > 0) {
            ConvCall oc;
            oc.pl = &o_v;
            oc.in = ws.ghr[l]; oc.in_gs = (long)B * C * hw;
            oc.out = ws.gou[l]; oc.out_gs = (long)B * C * hw;
            oc.N = B; oc.Hs = h; oc.Ws = w;
            BDE_TRY(run_conv(m, oc, s));
            g.gh_o = ws.gou[l];
        }
        hipLaunchKernelGGL(gru_out_kernel, dim3(blocks), dim3(256), 0, s, g);
        BDE_HIP(hipGetLastError());
    }
    return BDE_OK;
}

static int run_recurrent_level(bde_model* m, int l, const float* in, int T, int B, int H, int W, hipStream_t s,
                               bool enc_done = false) {
    Workspace& ws = m->W();
    const int C = m->cout(l), h = H / 2, w = W / 2;
    const long TB = (long)T * B, hw = (long)h * w;
    if (!enc_done && !(m->debug_skip & 16)) BDE_TRY(run_enc_gx(m, l, in, 0, (int)TB, T, B, H, W, s));
    if (!m->cfg.use_rc) return BDE_OK;
    if (m->cfg.recurrent_type == 1) return run_gru_steps(m, l, T, B, h, w, s);
    if ((size_t)l < m->lstm_sbk.size()) m->lstm_sbk[l].sb_used = ws.hsk[l] != nullptr ? 1 : 0;
    if (!(m->debug_skip & 4) && ws.hsk[l] != nullptr) {
        if (m->lstm_two_streams && m->dir_mask == 3 && !m->prof_on) {
            // A step launch's workgroups run in lockstep: prologue, halo wait and the pointwise tail (a quarter of the cycles)
            // leave the matrix cores idle chip-wide.  The forward and the backward sweep are independent until the merge
            // (V5.py:122-147): as two chains of one-direction launches on two streams they drift apart and one direction's
            // MFMA phases cover the other's tails.  Same launches per direction as bde_split_sweep: results are bit-identical.
            const int slot = m->cur;
            if (!m->dir_stream[slot]) {
                BDE_HIP(hipStreamCreateWithFlags(&m->dir_stream[slot], hipStreamNonBlocking));
                BDE_HIP(hipEventCreateWithFlags(&m->dir_fork[slot], hipEventDisableTiming));
                BDE_HIP(hipEventCreateWithFlags(&m->dir_join[slot], hipEventDisableTiming));
            }
            hipStream_t s2 = m->dir_stream[slot];
            BDE_HIP(hipEventRecord(m->dir_fork[slot], s));
            BDE_HIP(hipStreamWaitEvent(s2, m->dir_fork[slot], 0));
            m->dir_mask = 1;
            int st = run_recurrent_steps_sbk(m, l, T, B, h, w, s);
            m->dir_mask = 2;
            if (st == BDE_OK) st = run_recurrent_steps_sbk(m, l, T, B, h, w, s2);
            m->dir_mask = 3;
            BDE_HIP(hipEventRecord(m->dir_join[slot], s2));
            BDE_HIP(hipStreamWaitEvent(s, m->dir_join[slot], 0));
            return st;
        }
        return run_recurrent_steps_sbk(m, l, T, B, h, w, s);
    }
    if (!(m->debug_skip & 4) && ws.hsb[l] != nullptr) {
        BDE_REQUIRE(m->dir_mask == 3, "the split-bf16 recurrent step (lstm_sb) runs both directions only");
        return run_recurrent_steps_sb(m, l, T, B, h, w, s);
    }
    // T recurrent steps; group 0 = forward at t = s, group 1 = backward at t = T-1-s
    const int dsel = m->dir_mask == 2 ? 1 : 0;
    const bool one_dir = m->dir_mask != 3;
    const PackedLayer& pl = m->lstm[l];
    float* hs = ws.hseq[l];
    const long dstride = TB * C * hw;          // direction stride inside hseq
    const long fs = (long)B * C * hw;          // one time step (B frames)
    for (int st = 0; st < T && !(m->debug_skip & 4); ++st) {
        const int tf = st, tb = T - 1 - st;
        ConvArgs a;
        memset(&a, 0, sizeof a);
        a.first = (st == 0);
        const float* hprev_f = hs + (long)(tf - 1) * fs;               // unused when first
        const float* hprev_b = hs + dstride + (long)(tb + 1) * fs;
        if (a.first) { hprev_f = hs; hprev_b = hs; }
        a.in = hprev_f;
        a.in_gs = hprev_b - hprev_f;
        a.in_ns = (long)C * hw;
        a.wpk = m->P(pl.w_off);
        a.w_gs = pl.w_sz;
        a.bias = m->P(pl.b_off);
        a.out = hs + (long)tf * fs;
        a.out_gs = (hs + dstride + (long)tb * fs) - a.out;
        a.out_ns = (long)C * hw;
        a.gx = ws.gx[l] + (long)tf * B * 4 * C * hw;
        a.gx_gs = (ws.gx[l] + TB * 4 * C * hw + (long)tb * B * 4 * C * hw) - a.gx;
        a.gx_ns = (long)4 * C * hw;
        a.cstate = ws.cst[l];
        a.c_gs = (long)B * C * hw;
        a.c_ns = (long)C * hw;
        if (one_dir) {                                                 // group 0 of a one-group launch = the chosen direction
            a.lstm_groups = 1;
            a.in += dsel * a.in_gs;
            a.wpk += dsel * a.w_gs;
            a.bias += (long)dsel * 4 * C;
            a.out += dsel * a.out_gs;
            a.gx += dsel * a.gx_gs;
            a.cstate += dsel * a.c_gs;
        }
        a.N = B;
        a.Cin = C;
        a.Hin = a.Hs = h;
        a.Win = a.Ws = w;
        a.Cout = 4 * C;
        a.Ho = h;
        a.Wo = w;
        a.nchunks = pl.nchunks;
        {
            ProfScope ps(m, pname("lstm", l), s);
            const bool hc8 = m->lstm_hc8 == 1 || (m->lstm_hc8 < 0 && lstm16_wants_hc8(a));
            if (hc8) {                                   // 8-channel workgroups: their own weight packing
                a.w_gs = m->lstm8[l].w_sz;
                a.wpk = m->P(m->lstm8[l].w_off) + (one_dir ? dsel * a.w_gs : 0);
            }
            BDE_TRY(lstm16_launch(a, s, hc8));
        }
    }
    return BDE_OK;
}

// DFrameAttention + in-place refinement for one target frame (V5.py:154-169; DTransformer.py:376-389).
//   xq      : query frame [B][C][HW] (slot q_idx)
//   kvslot  : per slot, base of the [B][depth*2C][HW] K|V stack of that frame (nullptr = zero frame);
//             ignored for slot q_idx
//   addres  : tensor added to the result (merged[t], V5.py:166) or nullptr
//   out     : [B][C][HW]
//   qkv_first: q|k|v of block blk0 for xq if already computed (batched over T), else nullptr
static int run_attention_frame(bde_model* m, int l, const float* xq, const float* const* kvslot, const float* addres,
                               float* out, int B, int H, int W, int blk0, int nblk, const float* qkv_first,
                               hipStream_t s) {
    const bde_config& c = m->cfg;
    Workspace& ws = m->W();
    const AttnLevel& al = m->attn[l];
    const int C = al.C, D = c.frame_num;
    const long HW = (long)H * W;
    const int ph = (7 - H % 7) % 7, pw = (7 - W % 7) % 7;   // DTransformer.py:260-263
    const int pt = ph / 2, plft = pw / 2;
    const int Hp = H + ph, Wp = W + pw;
    const float* x = xq;
    const bool fused = al.blocks[0].proj16 >= 0 && cdivl(HW, 32) * B >= m->fused_min_tiles;
    bool have_qkv = false;                                   // the fused kernel leaves the next block's q|k|v in ws.qkv
    for (int i = blk0; i < blk0 + nblk; ++i) {
        const AttnBlock& ab = al.blocks[i];
        const bool dil = (i % 2) == 1;                       // DTransformer.py:362
        const bool last = (i == blk0 + nblk - 1);
        // q | k | v of the current x
        const float* qkv = ws.qkv;
        if (i == blk0 && qkv_first) qkv = qkv_first;
        else if (!have_qkv) { ProfScope ps(m, pname("chain_qkv", l), s); BDE_TRY(run_pw(m, &ab.qkv, x, ws.qkv, B, HW, ACT_NONE, nullptr, nullptr, 0, 0, 0, s)); }
        AttnArgs a;
        memset(&a, 0, sizeof a);
        a.q = qkv;
        a.q_bs = 3 * C * HW;
        for (int d = 0; d < D; ++d) {
            if (d == c.q_idx) {
                a.kv[d] = qkv + (long)C * HW;
                a.kv_bs[d] = 3 * C * HW;
            } else if (kvslot[d]) {
                a.kv[d] = kvslot[d] + (long)i * 2 * C * HW;
                a.kv_bs[d] = (long)al.depth * 2 * C * HW;
            } else {
                a.kv[d] = nullptr;
            }
            a.v_off[d] = (long)C * HW;
        }
        a.kvpad = m->P(ab.kvpad_off);
        a.biasT = m->P(ab.bias_off);
        a.out = ws.ao;
        a.out_bs = C * HW;
        a.D = D; a.C = C; a.heads = c.num_heads; a.H = H; a.W = W; a.Hp = Hp; a.Wp = Wp;
        a.pt = pt; a.pl = plft; a.nWw = Wp / 7; a.dilated = dil ? 1 : 0;
        {
            ProfScope ps(m, pname("chain_core", l), s);
            if (C / c.num_heads == 16 && D * ATT_TOK <= 160 && tuning().attn_mfma) BDE_TRY(attn_mfma16_launch(a, B, s));
            else BDE_TRY(attn_launch(a, B, s));
        }
        float* dst = last ? out : (x == ws.xa ? ws.xb : ws.xa);
        if (fused) {
            TokenArgs ta;
            memset(&ta, 0, sizeof ta);
            ta.ao = ws.ao;
            ta.x = x;
            ta.addres = last ? addres : nullptr;
            ta.x2 = dst;
            ta.wproj = m->P(ab.proj16);  ta.bproj = m->P(ab.proj.b_off);
            ta.wfc1 = m->P(ab.fc1_16);   ta.bfc1 = m->P(ab.fc1.b_off);  ta.sfc1 = m->P(ab.fc1.s_off);
            ta.wfc2 = m->P(ab.fc2_16);   ta.bfc2 = m->P(ab.fc2.b_off);
            if (!last) {
                const AttnBlock& nb = al.blocks[i + 1];
                ta.qkv = ws.qkv;
                ta.wqkv = m->P(nb.qkv16);  ta.bqkv = m->P(nb.qkv.b_off);  ta.sqkv = m->P(nb.qkv.s_off);
            }
            ta.bs_c = (long)C * HW;
            ta.bs_qkv = 3L * C * HW;
            ta.C = C;
            ta.HW = (int)HW;
            ta.mask_w = dil ? W : 0;
            ta.mask_pt = pt;
            ta.mask_pl = plft;
            ta.debug = m->tok_debug;
            ta.stamps = m->tok_stamps;
            { ProfScope ps(m, pname("chain_token", l), s); BDE_TRY(token_launch(ta, B, s)); }
            have_qkv = !last;
            x = dst;
            continue;
        }
        // x1 = shortcut + proj(attn)   (uncovered pixels of a dilated block: shortcut only)
        { ProfScope ps(m, pname("chain_proj", l), s); BDE_TRY(run_pw(m, &ab.proj, ws.ao, ws.x1, B, HW, ACT_NONE, x, nullptr, dil ? W : 0, pt, plft, s)); }
        // x2 = x1 + fc2(GELU(fc1(LN(x1))))  (+ merged[t] after the last block)
        { ProfScope ps(m, pname("chain_mlp_in", l), s); BDE_TRY(run_pw(m, &ab.fc1, ws.x1, ws.hid, B, HW, ACT_GELU, nullptr, nullptr, 0, 0, 0, s)); }
        { ProfScope ps(m, pname("chain_mlp_out", l), s); BDE_TRY(run_pw(m, &ab.fc2, ws.hid, dst, B, HW, ACT_NONE, ws.x1, last ? addres : nullptr, 0, 0, 0, s)); }
        x = dst;
    }
    return BDE_OK;
}

static bool winblock_ok(const bde_model* m, int l) {
    const AttnLevel& al = m->attn[l];
    return m->winblock && al.depth > 0 && al.blocks[0].biasF_off >= 0 && al.blocks[0].qkv16 >= 0;
}

// One frame through the blocks of a level with winblock.h.  Everything is token-major [B][HW][C]:
//   frames[d]: frame of slot d (nullptr = zero frame), frames[q_idx] = the query frame
//   addres   : added to the result (merged[t], V5.py:166) or nullptr
//   out_tok  : result, token-major;  out_nchw: the same result as [B][C][HW] planes (may be nullptr)
static int run_attention_frame_win(bde_model* m, int l, const float* const* frames, const float* addres, float* out_tok,
                                   float* out_nchw, int B, int H, int W, int blk0, int nblk, hipStream_t s) {
    const bde_config& c = m->cfg;
    Workspace& ws = m->W();
    const AttnLevel& al = m->attn[l];
    const int C = al.C, D = c.frame_num;
    const long HW = (long)H * W;
    const int ph = (7 - H % 7) % 7, pw = (7 - W % 7) % 7;   // DTransformer.py:260-263
    const float* x = frames[c.q_idx];
    for (int i = blk0; i < blk0 + nblk; ++i) {
        const AttnBlock& ab = al.blocks[i];
        const bool last = (i == blk0 + nblk - 1);
        float* dst = last ? out_tok : (x == ws.xa ? ws.xb : ws.xa);
        WinArgs a;
        memset(&a, 0, sizeof a);
        a.slot[0] = x;
        a.slot_bs[0] = C * HW;
        int k = 1;
        for (int d = 0; d < D; ++d) {
            if (d == c.q_idx) continue;
            a.slot[k] = frames[d];
            a.slot_bs[k] = C * HW;
            ++k;
        }
        a.nslots = D;
        a.addres = last ? addres : nullptr;
        a.addres_bs = C * HW;
        a.out = dst;
        a.out_nchw = last ? out_nchw : nullptr;
        a.out_bs = C * HW;
        a.wqkv = m->P(ab.qkv16);   a.bqkv = m->P(ab.qkv.b_off);  a.sqkv = m->P(ab.qkv.s_off);
        a.wproj = m->P(ab.proj16); a.bproj = m->P(ab.proj.b_off);
        a.wfc1 = m->P(ab.fc1_16);  a.bfc1 = m->P(ab.fc1.b_off);  a.sfc1 = m->P(ab.fc1.s_off);
        a.wfc2 = m->P(ab.fc2_16);  a.bfc2 = m->P(ab.fc2.b_off);
        a.biasF = m->P(ab.biasF_off);
        const bool sbk = m->winblock_sb && ab.qkvS >= 0;
        if (sbk) {
            const bool two = m->sb_terms == 2;
            a.terms = two ? 2 : 3;
            a.wqkvS = reinterpret_cast<const unsigned short*>(m->P(two ? ab.qkvH : ab.qkvS));
            a.wprojS = reinterpret_cast<const unsigned short*>(m->P(two ? ab.projH : ab.projS));
            a.wfc1S = reinterpret_cast<const unsigned short*>(m->P(two ? ab.fc1H : ab.fc1S));
            a.wfc2S = reinterpret_cast<const unsigned short*>(m->P(two ? ab.fc2H : ab.fc2S));
            a.unscale = m->P(ab.unscaleH);
            a.ovf = m->ovf();
        }
        a.stamps = m->tok_stamps;
        a.H = H; a.W = W; a.Hp = H + ph; a.Wp = W + pw; a.pt = ph / 2; a.pl = pw / 2;
        a.dilated = (i % 2) == 1 ? 1 : 0;                    // DTransformer.py:362
        { ProfScope ps(m, pname("winblock", l), s); BDE_TRY(sbk ? winblock_sb_launch(a, B, s) : winblock_launch(a, B, s)); }
        x = dst;
    }
    return BDE_OK;
}

static bool wide_ok(const bde_model* m, int l) {
    const AttnLevel& al = m->attn[l];
    return m->wide && al.depth > 0 && !winblock_ok(m, l) && al.kvallW >= 0 && al.blocks[0].qkvW >= 0 &&
           m->cfg.frame_num * ATT_TOK <= 160;
}

// One GEMM of the wide chain (wideblock.h): x FRAG16 [B][ntile][K/16][256] -> token-major or FRAG16
static int run_tokgemm(bde_model* m, const char* span, int l, long w_off, const PackedLayer& pl, int M, int K, const float* x, int B,
                       long HW, float* out_tok, float* out_frag, int act, const float* res, const float* addres, float* out_nchw,
                       int mask_w, int mask_pt, int mask_pl, long row_off, hipStream_t s, long wH_off = -1, long wH_unscale = -1) {
    TokGemmArgs a;
    memset(&a, 0, sizeof a);
    const int ntile = (int)cdivl(HW, 16);
    a.x = x;
    a.w = m->P(w_off) + row_off / 16 * (K / 16) * 256;
    a.bias = m->P(pl.b_off) + row_off;
    a.lnsum = pl.s_off >= 0 ? m->P(pl.s_off) + row_off : nullptr;
    a.out_tok = out_tok;
    a.out_frag = out_frag;
    a.out_nchw = out_nchw;
    a.res = res;
    a.addres = addres;
    a.x_bs = (long)ntile * 16 * K;
    a.out_bs = out_tok ? HW * M : (long)ntile * 16 * M;
    a.res_bs = a.addres_bs = (long)ntile * 16 * M;
    a.nchw_bs = HW * M;
    a.K = K; a.M = M; a.HW = (int)HW; a.ntile = ntile;
    a.act = act;
    a.mask_w = mask_w; a.mask_pt = mask_pt; a.mask_pl = mask_pl;
    ProfScope ps(m, pname(span, l), s);
    if (wH_off >= 0 && m->wide_kv_sb && m->sb_terms == 2 && row_off == 0) {      // two fp16 terms on the matrix cores (tokgemm_sb_kernel)
        a.wS = reinterpret_cast<const unsigned short*>(m->P(wH_off));
        a.w_unscale = m->P(wH_unscale);
        a.ovf = m->ovf();
        if (tokgemm_sb_fits(a)) return tokgemm_sb_launch(a, B, s);
    }
    return tokgemm_launch(a, B, s);
}

// DFrameAttention + refinement for one target frame on the wide chain.  Everything FRAG16 / token-major:
//   xq       : query frame, FRAG16 [B][ntile][C/16][256]
//   kvslot[d]: token-major K|V stack [B][HW][depth*2C] of slot d's frame (nullptr = zero frame; ignored for q_idx)
//   addres   : FRAG16 tensor added to the result (merged[t]) or nullptr;   out: FRAG16;   out_nchw: optional [B][C][HW]
//   qkv_first: token-major q|k|v [B][HW][3C] of block blk0 for xq if already computed (batched over T)
//   prev_frag / prev_slot: FRAG16 frame of ONE refined neighbour (buffer slot prev_slot) whose K | V the attention core computes
//              itself (wide_core.h) instead of reading kvslot[prev_slot]; nullptr = none
static bool wide_core2_ok(const bde_model* m, int l) {
    const AttnLevel& al = m->attn[l];
    return m->wide_core2 && m->wide_fuse_qkv && m->wide_kv_sb && m->sb_terms == 2 && al.C == 256 && m->cfg.num_heads * 16 == al.C &&
           al.depth > 0 && al.blocks[0].biasW_off >= 0 && al.blocks[0].qkvHF >= 0;
}
struct WideTwin { float* s = nullptr; float* st = nullptr; };     // SPL16 image (as float*) and statistics of a FRAG16 frame, or nothing
static int run_attention_frame_wide(bde_model* m, int l, const float* xq, const float* const* kvslot, const float* addres, float* out,
                                    float* out_nchw, int B, int H, int W, int blk0, int nblk, const float* qkv_first, hipStream_t s,
                                    const float* prev_frag = nullptr, int prev_slot = -1, WideTwin xq_twin = WideTwin(),
                                    WideTwin out_twin = WideTwin(), WideTwin prev_twin = WideTwin()) {
    const bde_config& c = m->cfg;
    Workspace& ws = m->W();
    const AttnLevel& al = m->attn[l];
    const int C = al.C, D = c.frame_num;
    const long HW = (long)H * W;
    const int ph = (7 - H % 7) % 7, pw = (7 - W % 7) % 7;   // DTransformer.py:260-263
    const int pt = ph / 2, plft = pw / 2;
    const int ntile = (int)cdivl(HW, 16);
    const float* x = xq;
    WideTwin xt = xq_twin;
    // SPL16 operands for the core: every frame it reads has its twin, and every block that produces a frame writes one (mlp_fused_kernel)
    const bool will_fuse_mlp = m->wide_fuse_mlp && m->wide_fuse_fc2 && m->sb_terms == 2 && C == 256 && ws.tile_count && ws.xaS;
    const bool spl = m->wide_spl && wide_core2_ok(m, l) && will_fuse_mlp && xq_twin.s && (!prev_frag || prev_twin.s) && al.blocks[0].qkvN >= 0;
    for (int i = blk0; i < blk0 + nblk; ++i) {
        const AttnBlock& ab = al.blocks[i];
        const bool dil = (i % 2) == 1;                       // DTransformer.py:362
        const bool last = (i == blk0 + nblk - 1);
        if (wide_core2_ok(m, l)) {
            WideCoreArgs a;
            memset(&a, 0, sizeof a);
            a.x = x;
            a.xp = prev_frag;
            a.x_bs = (long)ntile * 16 * C;
            a.q_slot = c.q_idx;
            a.p_slot = prev_frag ? prev_slot : -1;
            for (int d = 0; d < D; ++d) {
                a.kv[d] = nullptr;
                if (d == c.q_idx || (prev_frag && d == prev_slot) || !kvslot[d]) continue;
                a.kv[d] = kvslot[d]; a.kv_bs[d] = HW * al.depth * 2 * C; a.kv_ld[d] = al.depth * 2 * C;
                a.k_off[d] = i * 2 * C; a.v_off[d] = i * 2 * C + C;
            }
            a.kvpad = m->P(ab.kvpad_off);
            a.biasW = m->P(ab.biasW_off);
            a.wqkvS = reinterpret_cast<const unsigned short*>(m->P(spl ? ab.qkvN : ab.qkvHF));
            a.wqkv_unscale = m->P(spl ? ab.qkvN_unscale : ab.qkvHF_unscale);
            if (spl) {
                a.xS = reinterpret_cast<const unsigned short*>(xt.s);
                a.xSt = xt.st;
                a.xpS = prev_frag ? reinterpret_cast<const unsigned short*>(prev_twin.s) : nullptr;
                a.xpSt = prev_frag ? prev_twin.st : nullptr;
                a.spl_bs = (long)ntile * 16 * C * 2;          // 16-bit elements: two terms per value
                a.st_bs = (long)ntile * 16 * 2;
                a.zeros = m->P(m->zero_off);
            }
            a.bqkv = m->P(ab.qkv.b_off);
            a.sqkv = m->P(ab.qkv.s_off);
            a.out = ws.ao;
            a.D = D; a.C = C; a.heads = c.num_heads; a.H = H; a.W = W; a.Hp = H + ph; a.Wp = W + pw;
            a.pt = pt; a.pl = plft; a.nWw = (W + pw) / 7; a.dilated = dil ? 1 : 0; a.ntile = ntile;
            a.ovf = m->ovf();
            a.stamps = m->tok_debug == 22 ? m->tok_stamps : nullptr;
            ProfScope ps(m, pname("wide_core", l), s);
            BDE_TRY(wide_core_launch(a, B, s));
        } else {
        BDE_REQUIRE(prev_frag == nullptr, "wide chain: K | V of the refined frame are expected from the attention core");
        const float* qkv = ws.qkv;
        const bool fuse_qkv = m->wide_fuse_qkv != 0;       // q | k | v of the query frame inside the attention core (wideblock.h)
        if (fuse_qkv) qkv = nullptr;
        else if (i == blk0 && qkv_first) qkv = qkv_first;
        else BDE_TRY(run_tokgemm(m, "wide_qkv", l, ab.qkvW, ab.qkv, 3 * C, C, x, B, HW, ws.qkv, nullptr, ACT_NONE, nullptr, nullptr,
                                 nullptr, 0, 0, 0, 0, s));
        AttnTokArgs a;
        memset(&a, 0, sizeof a);
        if (fuse_qkv) {
            a.x = x;
            a.x_bs = (long)ntile * 16 * C;
            a.wqkv = m->P(ab.qkvW);
            a.bqkv = m->P(ab.qkv.b_off);
            a.sqkv = m->P(ab.qkv.s_off);
            a.q_slot = c.q_idx;
            if (m->wide_kv_sb && m->sb_terms == 2 && ab.qkvHF >= 0) {          // q|k|v on two fp16 terms (wideblock.h)
                a.wqkvS = reinterpret_cast<const unsigned short*>(m->P(ab.qkvHF));
                a.wqkv_unscale = m->P(ab.qkvHF_unscale);
                a.ovf = m->ovf();
            }
        }
        a.q = qkv;
        a.q_bs = HW * 3 * C;
        a.q_ld = 3 * C;
        for (int d = 0; d < D; ++d) {
            if (d == c.q_idx) {
                a.kv[d] = qkv; a.kv_bs[d] = HW * 3 * C; a.kv_ld[d] = 3 * C; a.k_off[d] = C; a.v_off[d] = 2 * C;
            } else if (kvslot[d]) {
                a.kv[d] = kvslot[d]; a.kv_bs[d] = HW * al.depth * 2 * C; a.kv_ld[d] = al.depth * 2 * C;
                a.k_off[d] = i * 2 * C; a.v_off[d] = i * 2 * C + C;
            } else {
                a.kv[d] = nullptr;
            }
        }
        a.kvpad = m->P(ab.kvpad_off);
        a.biasT = m->P(ab.bias_off);
        a.out = ws.ao;
        a.out_bs = (long)ntile * 16 * C;
        a.D = D; a.C = C; a.heads = c.num_heads; a.H = H; a.W = W; a.Hp = H + ph; a.Wp = W + pw;
        a.pt = pt; a.pl = plft; a.nWw = (W + pw) / 7; a.dilated = dil ? 1 : 0; a.ntile = ntile;
        { ProfScope ps(m, pname("wide_core", l), s); BDE_TRY(attn_tok16_launch(a, B, s)); }
        }
        float* dst = (last && out) ? out : (x == ws.xa ? ws.xb : ws.xa);     // out == nullptr: the caller only wants out_nchw
        // x1 = shortcut + proj(attn)   (uncovered pixels of a dilated block: shortcut only; DTransformer.py:299, 79-82)
        // hidden = GELU(fc1(LN(x1))): with two-term operands both in one launch (projfc1_sb_kernel)
        if (m->wide_fuse_mlp && m->wide_fuse_fc2 && m->sb_terms == 2 && ab.projHF >= 0 && ab.fc1N >= 0 && C == 256 && ab.fc1.Cout == 4 * C &&
            ws.tile_count) {
            // the whole token half of the block in one launch (wide_mlp.h)
            MlpFusedArgs fa;
            memset(&fa, 0, sizeof fa);
            fa.ao = ws.ao; fa.x = x;
            fa.wprojS = reinterpret_cast<const unsigned short*>(m->P(ab.projHF));
            fa.wfc1S = reinterpret_cast<const unsigned short*>(m->P(ab.fc1N));
            fa.wfc2S = reinterpret_cast<const unsigned short*>(m->P(ab.fc2N));
            fa.unscale_proj = m->P(ab.mlpHF_unscale);
            fa.unscale_mlp = m->P(ab.mlpN_unscale);
            fa.bproj = m->P(ab.proj.b_off);
            fa.bfc1 = m->P(ab.fc1.b_off);
            fa.sfc1 = m->P(ab.fc1.s_off);
            fa.bfc2 = m->P(ab.fc2.b_off);
            fa.part = ws.hid;                                // (the hidden activations never leave the workgroups)
            fa.count = ws.tile_count;
            fa.out = dst;
            fa.out_nchw = last ? out_nchw : nullptr;
            fa.addres = last ? addres : nullptr;
            fa.x_bs = (long)ntile * 16 * C;
            fa.nchw_bs = HW * C;
            fa.HW = (int)HW; fa.ntile = ntile; fa.B = B;
            fa.mask_w = dil ? W : 0; fa.mask_pt = pt; fa.mask_pl = plft;
            fa.ovf = m->ovf();
            if (spl) {
                const WideTwin dt = last ? out_twin : (dst == ws.xa ? WideTwin{ws.xaS, ws.stA} : WideTwin{ws.xbS, ws.stB});
                fa.out_spl = reinterpret_cast<unsigned short*>(dt.s);
                fa.out_stats = dt.st;
                fa.spl_bs = (long)ntile * 16 * C * 2;
                fa.st_bs = (long)ntile * 16 * 2;
                xt = dt;
            }
            fa.stamps = m->tok_debug == 21 ? m->tok_stamps : nullptr;
            ProfScope ps(m, pname("wide_mlp", l), s);
            BDE_TRY(mlp_fused_launch(fa, s));
            x = dst;
            continue;
        }
        if (m->wide_fuse_mlp && m->sb_terms == 2 && ab.projHF >= 0 && C == 256 && ab.fc1.Cout == 4 * C) {
            ProjFc1Args pa;
            memset(&pa, 0, sizeof pa);
            pa.ao = ws.ao; pa.x = x; pa.x1 = ws.x1; pa.hid = ws.hid;
            pa.wprojS = reinterpret_cast<const unsigned short*>(m->P(ab.projHF));
            pa.wfc1S = reinterpret_cast<const unsigned short*>(m->P(ab.fc1HF));
            pa.proj_unscale = m->P(ab.mlpHF_unscale);
            pa.fc1_unscale = m->P(ab.mlpHF_unscale + 1);
            pa.bproj = m->P(ab.proj.b_off);
            pa.bfc1 = m->P(ab.fc1.b_off);
            pa.sfc1 = m->P(ab.fc1.s_off);
            pa.x_bs = (long)ntile * 16 * C;
            pa.hid_bs = (long)ntile * 16 * 4 * C;
            pa.C = C; pa.hidden = 4 * C; pa.HW = (int)HW; pa.ntile = ntile;
            pa.mask_w = dil ? W : 0; pa.mask_pt = pt; pa.mask_pl = plft;
            pa.ovf = m->ovf();
            ProfScope ps(m, pname("wide_projfc", l), s);
            BDE_TRY(projfc1_sb_launch(pa, B, s));
        } else {
        BDE_TRY(run_tokgemm(m, "wide_proj", l, ab.projW, ab.proj, C, C, ws.ao, B, HW, nullptr, ws.x1, ACT_NONE, x, nullptr, nullptr,
                            dil ? W : 0, pt, plft, 0, s));
        // x2 = x1 + fc2(GELU(fc1(LN(x1))))  (+ merged[t] after the last block; DTransformer.py:279-283,304, V5.py:166)
        BDE_TRY(run_tokgemm(m, "wide_mlp_in", l, ab.fc1W, ab.fc1, 4 * C, C, ws.x1, B, HW, nullptr, ws.hid, ACT_GELU, nullptr, nullptr,
                            nullptr, 0, 0, 0, 0, s));
        }
        BDE_TRY(run_tokgemm(m, "wide_mlp_out", l, ab.fc2W, ab.fc2, C, 4 * C, ws.hid, B, HW, nullptr, dst, ACT_NONE, ws.x1,
                            last ? addres : nullptr, last ? out_nchw : nullptr, 0, 0, 0, 0, s));
        x = dst;
    }
    return BDE_OK;
}

typedef int (*FrameDoneFn)(bde_model* m, int t, void* ctx);
static int run_attention_level(bde_model* m, int l, int T, int B, int H, int W, hipStream_t s,
                               FrameDoneFn on_frame = nullptr, void* ctx = nullptr) {
    const bde_config& c = m->cfg;
    Workspace& ws = m->W();
    const AttnLevel& al = m->attn[l];
    const int C = al.C, D = c.frame_num;
    const long HW = (long)H * W, fs = (long)B * C * HW;
    const long kvfs = (long)B * al.depth * 2 * C * HW;
    if (winblock_ok(m, l)) {
        // one launch per block; the K|V of the neighbour frames are recomputed inside from the frames
        // themselves (refined in place for f < t, V5.py:166-169), so nothing else is staged per level
        { ProfScope ps(m, pname("to_tok", l), s); BDE_TRY(nchw_to_tok(ws.merged[l], ws.mergedT[l], T * B, C, (int)HW, s)); }
        for (int t = 0; t < T; ++t) {
            const float* frames[BDE_MAX_FRAMES];
            for (int d = 0; d < D; ++d) {
                const int f = t + c.buffer_index[d];
                frames[d] = (f < 0 || f >= T) ? nullptr : ws.mergedT[l] + (long)f * fs;
            }
            float* mt = ws.mergedT[l] + (long)t * fs;
            frames[c.q_idx] = mt;
            BDE_TRY(run_attention_frame_win(m, l, frames, mt, mt, ws.merged[l] + (long)t * fs, B, H, W, 0, al.depth, s));
            if (on_frame) BDE_TRY(on_frame(m, t, ctx));
        }
        return BDE_OK;
    }
    bool need_un = false, need_ref = false;
    for (int d = 0; d < D; ++d) {
        if (d == c.q_idx) continue;
        if (c.buffer_index[d] >= 0) need_un = true; else need_ref = true;
    }
    if (wide_ok(m, l)) {
        const int ntile = (int)cdivl(HW, 16);
        const long ffs = (long)B * ntile * 16 * C;           // FRAG16 frame stride
        const long q0fs = (long)B * HW * 3 * C;
        const bool twins = ws.mergedS[l] != nullptr && m->sb_terms == 2;
        const long sfs = ffs, stfs = (long)B * ntile * 16 * 2;    // SPL16 frame stride (floats of the image) / statistics stride
        {
            ProfScope ps(m, pname("to_frag", l), s);
            if (twins) BDE_TRY(nchw_to_frag_spl(ws.merged[l], ws.mergedT[l], reinterpret_cast<unsigned short*>(ws.mergedS[l]), ws.mstats[l], T * B,
                                                C, (int)HW, m->ovf(), s));
            else BDE_TRY(nchw_to_frag(ws.merged[l], ws.mergedT[l], T * B, C, (int)HW, s));
        }
        if (need_un)
            BDE_TRY(run_tokgemm(m, "wide_kv_all", l, al.kvallW, al.kvall, al.depth * 2 * C, C, ws.mergedT[l], T * B, HW, ws.kvun[l],
                                nullptr, ACT_NONE, nullptr, nullptr, nullptr, 0, 0, 0, 0, s, al.kvallH, al.kvallH_unscale));
        if (!m->wide_fuse_qkv)
            BDE_TRY(run_tokgemm(m, "wide_qkv_all", l, al.blocks[0].qkvW, al.blocks[0].qkv, 3 * C, C, ws.mergedT[l], T * B, HW, ws.qkv0[l],
                                nullptr, ACT_NONE, nullptr, nullptr, nullptr, 0, 0, 0, 0, s));
        // one refined neighbour (one negative buffer offset): its K | V are computed by the attention core of the frame that reads
        // them (wide_core.h), from the neighbour's refined FRAG16 frame -- no K|V GEMM between two frames of the chain
        int nneg = 0, neg_slot = -1;
        for (int d = 0; d < D; ++d)
            if (d != c.q_idx && c.buffer_index[d] < 0) { ++nneg; neg_slot = d; }
        const bool in_core = wide_core2_ok(m, l) && nneg == 1;
        for (int t = 0; t < T; ++t) {
            const float* kvslot[BDE_MAX_FRAMES];
            const float* prev_frag = nullptr;
            WideTwin qt, pt;
            if (twins) qt = WideTwin{ws.mergedS[l] + (long)t * sfs, ws.mstats[l] + (long)t * stfs};
            for (int d = 0; d < D; ++d) {
                const int f = t + c.buffer_index[d];
                if (d == c.q_idx || f < 0 || f >= T) kvslot[d] = nullptr;
                else if (f < t && in_core) {
                    kvslot[d] = nullptr;
                    prev_frag = ws.mergedT[l] + (long)f * ffs;
                    if (twins) pt = WideTwin{ws.mergedS[l] + (long)f * sfs, ws.mstats[l] + (long)f * stfs};
                }
                else if (f < t) kvslot[d] = ws.kvref[l] + (long)f * kvfs;      // refined (V5.py:166-169)
                else kvslot[d] = ws.kvun[l] + (long)f * kvfs;
            }
            float* mtF = ws.mergedT[l] + (long)t * ffs;
            BDE_TRY(run_attention_frame_wide(m, l, mtF, kvslot, mtF, mtF, ws.merged[l] + (long)t * fs, B, H, W, 0, al.depth,
                                             ws.qkv0[l] + (long)t * q0fs, s, prev_frag, neg_slot, qt, qt, pt));
            if (need_ref && !in_core && t + 1 < T)
                BDE_TRY(run_tokgemm(m, "wide_kv", l, al.kvallW, al.kvall, al.depth * 2 * C, C, mtF, B, HW, ws.kvref[l] + (long)t * kvfs,
                                    nullptr, ACT_NONE, nullptr, nullptr, nullptr, 0, 0, 0, 0, s, al.kvallH, al.kvallH_unscale));
            if (on_frame) BDE_TRY(on_frame(m, t, ctx));
        }
        return BDE_OK;
    }
    // K|V of every block for the still-unrefined frames, all T at once
    if (need_un) {
        ProfScope ps(m, pname("chain_kv_all", l), s);
        BDE_TRY(run_pw(m, &al.kvall, ws.merged[l], ws.kvun[l], T * B, HW, ACT_NONE, nullptr, nullptr, 0, 0, 0, s));
    }
    // q|k|v of the first block for every frame at once: its input is the still-unrefined merged[t]
    { ProfScope ps(m, pname("chain_qkv_all", l), s); BDE_TRY(run_pw(m, &al.blocks[0].qkv, ws.merged[l], ws.qkv0[l], T * B, HW, ACT_NONE, nullptr, nullptr, 0, 0, 0, s)); }
    for (int t = 0; t < T; ++t) {
        const float* kvslot[BDE_MAX_FRAMES];
        for (int d = 0; d < D; ++d) {
            const int f = t + c.buffer_index[d];
            if (d == c.q_idx || f < 0 || f >= T) kvslot[d] = nullptr;
            else if (f < t) kvslot[d] = ws.kvref[l] + (long)f * kvfs;      // refined (V5.py:166-169)
            else kvslot[d] = ws.kvun[l] + (long)f * kvfs;
        }
        float* mt = ws.merged[l] + (long)t * fs;
        BDE_TRY(run_attention_frame(m, l, mt, kvslot, mt, mt, B, H, W, 0, al.depth, ws.qkv0[l] + (long)t * B * 3 * C * HW, s));
        if (need_ref && t + 1 < T) {
            ProfScope ps(m, pname("chain_kv", l), s);
            BDE_TRY(run_pw(m, &al.kvall, mt, ws.kvref[l] + (long)t * kvfs, B, HW, ACT_NONE, nullptr, nullptr, 0, 0, 0, s));
        }
        if (on_frame) BDE_TRY(on_frame(m, t, ctx));
    }
    return BDE_OK;
}

// UpsampleConvLayer of decoder j on N frames [Cin][Hs][Ws] (+ skip): upsample kernel, then a plain conv.
// predI + the output activation ride in the conv's epilogue when one workgroup holds every output channel of a pixel
// (Cout <= 64: the 32 channels of the canonical last decoder are one MFMA row tile)
static bool pred_fusable(const bde_model* m) { return m->cfg.basechannels <= 64 && m->fuse_pred && !m->cfg.skip_concat; }

// skip_concat in front of a decoder or of predI (V5.py:285-286, 86-93): y = Conv1x1(cat(first, second)) on N frames of
// [C][hw] each -> ws.fuse
static int run_concat_fuse(bde_model* m, const PackedLayer& pl, const float* first, const float* second, int N, int C, long hw,
                           hipStream_t s) {
    Workspace& ws = m->W();
    BDE_TRY(concat_channels(first, second, ws.cat, N, (long)C * hw, (long)C * hw, s));
    return run_pw(m, &pl, ws.cat, ws.fuse, N, hw, ACT_NONE, nullptr, nullptr, 0, 0, 0, s);
}

// The last level without attention (depths[-1] == 0): Sequential(ParseLayer, ResidualBlockNoBN x n) on the frame buffer
// (V5.py:77-80, 151-169).  ParseLayer takes buffer SLOT 0 (:281-282), i.e. the frame at offset buffer_index[0] -- refined
// already when that offset is negative, still unrefined when it is not, zeros outside the sequence -- and the result is added
// to merged[t] in place: sequential in t like the attention.
static int run_bottleneck_level(bde_model* m, int l, int T, int B, int h, int w, hipStream_t s) {
    const bde_config& c = m->cfg;
    Workspace& ws = m->W();
    const int C = m->cout(l), nb = c.num_res_blocks;
    const long fs = (long)B * C * h * w;
    for (int t = 0; t < T; ++t) {
        ProfScope ps(m, pname("bottleneck", l), s);
        const int f = t + c.buffer_index[0];
        float* mt = ws.merged[l] + (long)t * fs;
        const float* x = (f < 0 || f >= T) ? ws.zero_l : ws.merged[l] + (long)f * fs;
        if (nb == 0) { BDE_TRY(add2(x, mt, mt, fs, s)); continue; }
        for (int k = 0; k < nb; ++k) {
            ConvCall c1;                                     // relu(conv1(x))
            c1.pl = &m->rb1[k]; c1.in = x; c1.out = ws.rbA; c1.N = B; c1.Hs = h; c1.Ws = w; c1.act = ACT_RELU;
            BDE_TRY(run_conv(m, c1, s));
            const bool last = k == nb - 1;
            ConvCall c2;                                     // x + conv2(.)   (+ merged[t] after the last block, V5.py:166)
            c2.pl = &m->rb2[k]; c2.in = ws.rbA; c2.N = B; c2.Hs = h; c2.Ws = w; c2.act = ACT_NONE;
            c2.res1 = x;
            c2.res2 = last ? mt : nullptr;
            c2.out = last ? mt : ws.rbX[k & 1];
            BDE_TRY(run_conv(m, c2, s));
            x = c2.out;
        }
    }
    return BDE_OK;
}

// Parts of a forward (forward_on): PART_PRE = head and the first level's encoder convolution, PART_MAIN = everything between it and
// the last kernel that writes split operands (the captured graph), PART_TAIL = what follows that kernel -- the last decoder
// convolution (+ predI) of the frames decoded last.  The overflow word of the range guard (split.h) is read back in front of the
// tail, so the host learns about an overflow while the tail still runs; PART_ALL = the whole forward in one piece.
enum { PART_ALL = 0, PART_MAIN = 1, PART_TAIL = 2, PART_PRE = 3 };

static int run_decoder(bde_model* m, int j, const float* in, const float* skip, float* out, int N, int Hs, int Ws,
                       hipStream_t s, const float* pred_head = nullptr, float* pred_out = nullptr, int part = PART_ALL, int decide_N = 0) {
    const PackedLayer& pl = m->dec[j];
    Workspace& ws = m->W();
    if (decide_N <= 0) decide_N = N;
    // a split-bf16 convolution reads SB16: the upsampling kernel then writes that image directly (no fp32 map, no conversion)
    const bool to_sb = m->fuse_enc_sb && (!pred_out || pl.Cout <= 32) && conv_takes_sb(m, pl, 1, decide_N, 2 * Hs, 2 * Ws) && ws.sb &&
                       split_bf16_bytes(N, pl.Cin, 4L * Hs * Ws) <= ws.sb_bytes;
    if (part != PART_TAIL) {
        ProfScope ps(m, pname("dec_up", j), s);
        if (to_sb) BDE_TRY(upsample2x_sum_split(in, skip, ws.sb, N, pl.Cin, Hs, Ws, m->sb_terms, m->ovf(), s));
        else BDE_TRY(upsample2x_sum(in, skip, ws.up, Hs, Ws, (long)N * pl.Cin, s));
    }
    if (part == PART_MAIN) return BDE_OK;
    ConvCall d;
    d.pl = &pl;
    d.in = to_sb ? ws.sb : ws.up;
    d.in_sb = to_sb;
    d.out = out;
    d.N = N;
    d.Hs = 2 * Hs;
    d.Ws = 2 * Ws;
    d.act = ACT_RELU6;
    d.pred_head = pred_head;
    d.pred_out = pred_out;
    d.decide_N = decide_N;
    ProfScope ps(m, pname("dec_conv", j), s);
    return run_conv(m, d, s);
}

static int check_dims(const bde_model* m, int T, int B, int H, int W) {
    const bde_config& c = m->cfg;
    BDE_REQUIRE(m->finalized, "weights are not finalized");
    BDE_REQUIRE(T >= 1 && B >= 1, "T=%d B=%d", T, B);
    const int mult = 1 << c.num_encoders;
    BDE_REQUIRE(H > 0 && W > 0 && H % mult == 0 && W % mult == 0, "H=%d W=%d must be multiples of %d", H, W, mult);
    for (int l = 0; l < c.num_encoders; ++l)
        if (c.depths[l] > 0)
            BDE_REQUIRE((H >> (l + 1)) >= 7 && (W >> (l + 1)) >= 7,
                        "feature map %dx%d at attention level %d is smaller than the 7x7 window (the reference "
                        "raises there too)", H >> (l + 1), W >> (l + 1), l);
    return BDE_OK;
}

static int forward_on(bde_model* m, const float* const* events, int T, int B, int H, int W, float* const* images,
                      hipStream_t s);

// The internal streams of the pipelined mode live as long as the process (one set per device, shared by the models
// on it): the caller's allocator may hold them as the last user of a tensor (record_stream) long after a model is
// gone, and recording on a destroyed stream faults.
static int pipeline_stream(int slot, hipStream_t* out) {
    static hipStream_t pool[BDE_MAX_DEVICES][bde_model::MAX_SLOTS] = {};
    int d = 0;
    BDE_HIP(hipGetDevice(&d));
    BDE_REQUIRE(d >= 0 && d < BDE_MAX_DEVICES, "device %d", d);
    if (!pool[d][slot]) BDE_HIP(hipStreamCreateWithFlags(&pool[d][slot], hipStreamNonBlocking));
    *out = pool[d][slot];
    return BDE_OK;
}

// ---- range guard of the two-term operand format: host side (split.h; bde_model::ovf_dev) ---------------------------------------
// Copy the current slot's overflow word to its pinned mirror and mark the point with the slot's event (forward_on).
static int note_overflow_readback(bde_model* m, hipStream_t s) {
    bde_model::Pending& p = m->pend[m->cur];
    if (!p.done) BDE_HIP(hipEventCreateWithFlags(&p.done, hipEventDisableTiming));
    BDE_HIP(hipMemcpyAsync(m->ovf_host + m->cur, m->ovf_dev + m->cur, sizeof(unsigned), hipMemcpyDeviceToHost, s));
    BDE_HIP(hipEventRecord(p.done, s));
    return BDE_OK;
}
// Remember what a forward needs to be recomputed: its stream, shape and output pointers (its events stay in the workspace).
static void note_pending(bde_model* m, int slot, hipStream_t s, int T, int B, int H, int W, float* const* images) {
    bde_model::Pending& p = m->pend[slot];
    p.on = true; p.stream = s; p.T = T; p.B = B; p.H = H; p.W = W;
    p.images.assign(images, images + T);
}
// Look at the overflow words of all forwards issued so far (waits for them).  None set: nothing to do.  Otherwise, "sb_auto" = 1:
// the model switches to three bf16 terms for good and the forwards that overflowed are recomputed from the events their
// workspaces still hold, into the same output buffers, on the streams they ran on; "sb_auto" = 0: BDE_ERR_RANGE.
static int settle_overflow(bde_model* m) {
    bool flagged[bde_model::MAX_SLOTS] = {};
    int nflag = 0;
    for (int i = 0; i < bde_model::MAX_SLOTS; ++i) {
        bde_model::Pending& p = m->pend[i];
        if (!p.on) continue;
        BDE_HIP(hipEventSynchronize(p.done));
        if (m->ovf_host[i] != 0) { flagged[i] = true; ++nflag; }
    }
    if (nflag == 0) {
        for (auto& p : m->pend) p.on = false;
        return BDE_OK;
    }
    m->sb_overflows += nflag;
    for (int i = 0; i < bde_model::MAX_SLOTS; ++i) m->ovf_host[i] = 0;
    if (!m->sb_auto) {
        for (auto& p : m->pend) p.on = false;
        return fail(BDE_ERR_RANGE, "%d forward(s): an activation reached 65520, beyond the two fp16 terms of the default operand format "
                    "(csrc/split.h); the frames of those calls are not valid.  set_tuning(\"sb_terms\", 3) or \"sb_auto\" = 1", nflag);
    }
    // every forward issued so far has to be complete before the workspaces go (the switch of formats releases them)
    for (int i = 0; i < bde_model::MAX_SLOTS; ++i)
        if (m->pend[i].on) BDE_HIP(hipStreamSynchronize(m->pend[i].stream));
    struct Redo { int slot; float* ev; long ev_fs; };
    std::vector<Redo> redo;
    for (int i = 0; i < bde_model::MAX_SLOTS; ++i) {
        if (!flagged[i]) continue;
        const bde_model::Pending& p = m->pend[i];
        const long ev_fs = (long)p.B * m->cfg.num_bins * p.H * p.W;
        float* ev = nullptr;
        BDE_HIP(hipMalloc((void**)&ev, sizeof(float) * ev_fs * p.T));
        BDE_HIP(hipMemcpy(ev, m->wslots[i].ev, sizeof(float) * ev_fs * p.T, hipMemcpyDeviceToDevice));
        redo.push_back({i, ev, ev_fs});
    }
    for (auto& w : m->wslots) w.release();
    m->sb_terms = 3;
    m->sb_latched = 1;
    int st = BDE_OK;
    for (const Redo& r : redo) {
        bde_model::Pending p = m->pend[r.slot];
        std::vector<const float*> evp(p.T);
        for (int t = 0; t < p.T; ++t) evp[t] = r.ev + (long)t * r.ev_fs;
        m->cur = r.slot;
        if (st == BDE_OK) st = forward_on(m, evp.data(), p.T, p.B, p.H, p.W, p.images.data(), p.stream);
        (void)hipStreamSynchronize(p.stream);
        (void)hipFree(r.ev);
    }
    m->cur = 0;
    for (auto& p : m->pend) p.on = false;
    return st;
}

// Pipelined dispatch: call i runs on internal stream i%depth with workspace i%depth.  Inputs are ordered
// after the caller's stream by an event; outputs are ordered back by bde_wait_outputs (or by the
// next call that reuses the slot).
static int forward_impl(bde_model* m, const float* const* events, int T, int B, int H, int W, float* const* images,
                        hipStream_t user) {
    if (m->pipeline < 2) {
        // (a forward still pending here was issued in pipelined mode: look at it before its slot's word is reused)
        for (const auto& p : m->pend) if (p.on) { BDE_TRY(settle_overflow(m)); break; }
        m->cur = 0;
        BDE_TRY(forward_on(m, events, T, B, H, W, images, user));
        if (m->ovf() == nullptr) return BDE_OK;
        // default mode: the frames are final when this call returns, so the overflow word is looked at here -- the host waits
        // for the forward up to its last operand split while the tail (the last convolution) is still running
        note_pending(m, 0, user, T, B, H, W, images);
        return settle_overflow(m);
    }
    const int slot = (int)(m->ncalls++ % m->pipeline);
    if (m->pend[slot].on) BDE_TRY(settle_overflow(m));       // the slot's previous forward must be final before its workspace is reused
    if (!m->pstream[slot]) {
        BDE_TRY(pipeline_stream(slot, &m->pstream[slot]));
        BDE_HIP(hipEventCreateWithFlags(&m->pin[slot], hipEventDisableTiming));
        BDE_HIP(hipEventCreateWithFlags(&m->pout[slot], hipEventDisableTiming));
    }
    m->cur = slot;
    BDE_HIP(hipEventRecord(m->pin[slot], user));
    BDE_HIP(hipStreamWaitEvent(m->pstream[slot], m->pin[slot], 0));
    m->last_stream = m->pstream[slot];
    const int st = forward_on(m, events, T, B, H, W, images, m->pstream[slot]);
    if (st == BDE_OK && m->ovf() != nullptr) note_pending(m, slot, m->pstream[slot], T, B, H, W, images);
    BDE_HIP(hipEventRecord(m->pout[slot], m->pstream[slot]));
    m->pbusy[slot] = true;
    m->cur = 0;
    return st;
}

// Everything between the input copy and the output copy: pointers depend only on the workspace,
// so the launch sequence can be captured once per (slot, T, B, H, W) into a hipGraph and replayed
// (~630 launches per forward at config A; replay removes their host cost).
static int forward_body(bde_model* m, int T, int B, int H, int W, hipStream_t s, int part);

static int forward_on(bde_model* m, const float* const* events, int T, int B, int H, int W, float* const* images,
                      hipStream_t s) {
    BDE_TRY(check_dims(m, T, B, H, W));
    BDE_TRY(ensure_workspace(m, T, B, H, W));
    Workspace& ws = m->W();
    const long ev_fs = (long)B * m->cfg.num_bins * H * W, img_fs = (long)B * H * W;
    BDE_TRY(copy_frames(events, ws.ev, T, ev_fs, 0, s));
    ProfScope whole(m, "forward", s);
    // Range guard (split.h): the slot's overflow word starts at zero and is read back behind the last kernel that writes split
    // operands -- in front of the forward's tail where the tail is a launch of its own (no side-stream decode, the upsampling
    // kernel writes the last convolution's operand image itself), behind it otherwise.
    // In the default mode (one sequence in flight) the host waits for that word before bde_forward returns: the forward is then
    // cut in three -- head + first encoder convolution launched eagerly (PART_PRE: the chip has work while the host replays the
    // graph), the graph (PART_MAIN), the last convolution launched eagerly behind the read-back (PART_TAIL).
    const bool guard = m->ovf() != nullptr;
    const bool cut = guard && m->eager_cut && m->pipeline < 2 && m->fuse_enc_sb && !(m->debug_skip & (8 | 16)) && !m->prof_on;
    const int part_main = cut ? PART_MAIN : PART_ALL;
    if (guard) BDE_HIP(hipMemsetAsync(m->ovf(), 0, sizeof(unsigned), s));
    if (cut) BDE_TRY(forward_body(m, T, B, H, W, s, PART_PRE));
    const bool can_graph = m->use_graph && ws.warm;   // (profiling spans are captured as event-record nodes)
    if (ws.graph_exec && ws.graph_part != part_main) { (void)hipGraphExecDestroy(ws.graph_exec); ws.graph_exec = nullptr; }
    if (can_graph && !ws.graph_exec) {
        // capture on a private stream (the caller's may be the legacy default stream, which cannot
        // capture); the instantiated graph is then launched on the caller's stream
        if (!m->cap_stream) BDE_HIP(hipStreamCreateWithFlags(&m->cap_stream, hipStreamNonBlocking));
        hipGraph_t graph = nullptr;
        BDE_HIP(hipStreamBeginCapture(m->cap_stream, hipStreamCaptureModeRelaxed));
        const int st = forward_body(m, T, B, H, W, m->cap_stream, part_main);
        ws.graph_part = part_main;
        const hipError_t e = hipStreamEndCapture(m->cap_stream, &graph);
        hipError_t ei = hipSuccess;
        if (st == BDE_OK && e == hipSuccess) ei = hipGraphInstantiate(&ws.graph_exec, graph, nullptr, nullptr, 0);
        if (graph) (void)hipGraphDestroy(graph);
        if (st != BDE_OK || e != hipSuccess || ei != hipSuccess) {
            // capture is an optimisation: fall back to eager launches for good
            (void)hipGetLastError();
            ws.graph_exec = nullptr;
            m->use_graph = 0;
        }
    }
    if (m->use_graph && can_graph && ws.graph_exec) {
        BDE_HIP(hipGraphLaunch(ws.graph_exec, s));
    } else {
        BDE_TRY(forward_body(m, T, B, H, W, s, part_main));
        ws.warm = true;                       // first call of a shape runs eagerly (one-time kernel attribute setup)
    }
    if (guard && cut) BDE_TRY(note_overflow_readback(m, s));
    if (cut) BDE_TRY(forward_body(m, T, B, H, W, s, PART_TAIL));     // one or two launches: not worth a graph of their own
    if (guard && !cut) BDE_TRY(note_overflow_readback(m, s));
    BDE_TRY(copy_frames(images, ws.out, T, img_fs, 1, s));
    return BDE_OK;
}

// C. decoder for frames [f0, f0 + nf) of the [T*B] stack (V5.py:183-197)
static int decode_frames(bde_model* mm, int f0, int nf, int T_, int B_, int H_, int W_, hipStream_t st, int part = PART_ALL) {
    // C. decoder (V5.py:183-197): x = L[-1]; x = dec_j(L[-1-j] + x); img = act(predI(x + head))
    Workspace& w = mm->W();
    const int L_ = mm->L;
    const float* x = w.merged[L_ - 1] + (long)f0 * mm->cout(L_ - 1) * (H_ >> L_) * (W_ >> L_);
    for (int j = 0; j < L_; ++j) {
        const int l = L_ - 1 - j;
        const long in_fs = (long)mm->cout(l) * (H_ >> (l + 1)) * (W_ >> (l + 1));
        const long out_fs = (long)mm->cin(l) * (H_ >> l) * (W_ >> l);
        // the tail of a forward = the last decoder's convolution: PART_MAIN stops in front of it, PART_TAIL runs nothing else
        const bool lastj = j == L_ - 1;
        const int jpart = part == PART_ALL ? PART_ALL : (lastj ? part : (part == PART_MAIN ? PART_ALL : -1));
        if (jpart >= 0) {
            ProfScope ps(mm, "decoder", st);
            const bool fuse = lastj && pred_fusable(mm);               // V5.py:195-197 in the last conv's epilogue
            const float* skip = w.merged[l] + (long)f0 * in_fs;
            if (mm->cfg.skip_concat) {                                 // decoder = Sequential(1x1 fusion, UpsampleConvLayer)
                if (jpart != PART_TAIL)
                    BDE_TRY(run_concat_fuse(mm, mm->dec_fuse[j], skip, x, nf, mm->cout(l), (long)(H_ >> (l + 1)) * (W_ >> (l + 1)), st));
                x = w.fuse;
                skip = nullptr;
            }
            BDE_TRY(run_decoder(mm, j, x, skip, w.dec[j] + (long)f0 * out_fs, nf,
                                H_ >> (l + 1), W_ >> (l + 1), st,
                                fuse ? w.head + (long)f0 * mm->cfg.basechannels * H_ * W_ : nullptr,
                                fuse ? w.out + (long)f0 * H_ * W_ : nullptr, jpart, T_ * B_));
        }
        x = w.dec[j] + (long)f0 * out_fs;
    }
    if (part == PART_MAIN) return BDE_OK;
    if (pred_fusable(mm)) return BDE_OK;
    const long total = (long)nf * H_ * W_;
    long blocks = std::min<long>(cdivl(total, 256), 4096);
    ProfScope ps(mm, "pred", st);
    const float* hd = w.head + (long)f0 * mm->cfg.basechannels * H_ * W_;
    if (mm->cfg.skip_concat) {                                     // predI = Sequential(1x1 fusion of cat(x, head), 1x1)
        BDE_TRY(run_concat_fuse(mm, mm->pred_fuse, x, hd, nf, mm->cfg.basechannels, (long)H_ * W_, st));
        x = w.fuse;
        hd = nullptr;
    }
    hipLaunchKernelGGL(pred_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x,
                       hd, mm->P(mm->predw_off),
                       mm->P(mm->predb_off), w.out + (long)f0 * H_ * W_, mm->cfg.basechannels, (long)H_ * W_, total,
                       mm->cfg.activation);
    BDE_HIP(hipGetLastError());
    return BDE_OK;
}

static int forward_body(bde_model* m, int T, int B, int H, int W, hipStream_t s, int part) {
    const bde_config& c = m->cfg;
    Workspace& ws = m->W();
    const int L = c.num_encoders;
    const long TB = (long)T * B;
    // While the attention chain of the LAST level walks the frames one by one (V5.py:154-169), the decoder of the frames already
    // refined -- independent per frame, V5.py:183-202 -- runs on a side stream in chunks of `overlap_chunk` frames: a forked branch
    // of the captured graph.  The chain's launches are latency-bound and leave most of the chip idle; the decoder's are not.  The
    // last chunk stays on the main stream (it is what the forward's tail is cut from).  Results are bit-identical: same launches.
    const bool plain_flags = c.use_rc && c.recurrent_type == 0 && !c.skip_concat && c.depths[L - 1] > 0;
    const bool side_decode = m->overlap != 0 && plain_flags && !(m->debug_skip & (2 | 8)) && !m->prof_on && T > m->overlap_chunk;
    const int last_chunk_t0 = side_decode ? (T - 1) / m->overlap_chunk * m->overlap_chunk : 0;
    const int tail_f0 = last_chunk_t0 * B, tail_nf = (int)TB - tail_f0;
    if (part == PART_TAIL) return (m->debug_skip & 8) ? BDE_OK : decode_frames(m, tail_f0, tail_nf, T, B, H, W, s, PART_TAIL);
    // A. head (V5.py:116) and the first level's encoder convolution: PART_PRE
    if (part != PART_MAIN) {
        ConvCall hc;
        hc.pl = &m->head;
        hc.in = ws.ev;
        hc.out = ws.head;
        hc.N = (int)TB;
        hc.Hs = H;
        hc.Ws = W;
        hc.act = ACT_RELU;
        { ProfScope ps(m, "head", s); BDE_TRY(run_conv(m, hc, s)); }
        if (!(m->debug_skip & 16)) BDE_TRY(run_enc_gx(m, 0, ws.head, 0, (int)TB, T, B, H, W, s));
        if (part == PART_PRE) return BDE_OK;
    }
    // B. levels (V5.py:119-172)
    struct SideCtx { int T, B, H, W, chunk, slot; hipStream_t main, side; };
    static auto decode_fn = decode_frames;     // (plain function pointer for the captureless callback)
    const int slot = (m->cap_stream && s == m->cap_stream) ? bde_model::MAX_SLOTS : m->cur;   // (capturing: the capture-only set)
    if (side_decode && !m->side[slot]) {
        int lo = 0, hi = 0;                               // lowest priority: the chain on the main stream goes first
        BDE_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        BDE_HIP(hipStreamCreateWithPriority(&m->side[slot], hipStreamNonBlocking, lo));
        BDE_HIP(hipEventCreateWithFlags(&m->join_ev[slot], hipEventDisableTiming));
    }
    while (side_decode && (int)m->frame_ev[slot].size() < T) {
        hipEvent_t e;
        BDE_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        m->frame_ev[slot].push_back(e);
    }
    const float* target = ws.head;
    bool decoded = false;
    for (int l = 0; l < L; ++l) {
        const int Hl = H >> l, Wl = W >> l, h = Hl / 2, w = Wl / 2, C = m->cout(l);
        BDE_TRY(run_recurrent_level(m, l, target, T, B, Hl, Wl, s, /*enc_done=*/l == 0));
        const long n = TB * C * h * w;
        { ProfScope ps(m, pname("merge", l), s); BDE_TRY(add2(ws.hseq[l], ws.hseq[l] + n, ws.merged[l], n, s)); }   // V5.py:137-147
        if (c.depths[l] > 0 && !(m->debug_skip & (l == 0 ? 1 : 2))) {
            static const char* names[BDE_MAX_LEVELS] = {"attn0", "attn1", "attn2", "attn3", "attn4", "attn5", "attn6", "attn7"};
            ProfScope ps(m, names[l], s);
            SideCtx sc{T, B, H, W, m->overlap_chunk, slot, s, m->side[slot]};
            FrameDoneFn fn = nullptr;
            const bool fork = side_decode && l == L - 1;
            if (fork) {
                fn = [](bde_model* mm, int t, void* vp) -> int {
                    SideCtx* q = (SideCtx*)vp;
                    const int done = t + 1;
                    if (done % q->chunk != 0 || done > (q->T - 1) / q->chunk * q->chunk) return BDE_OK;   // (the last chunk: main stream)
                    const int t0 = done - q->chunk;
                    BDE_HIP(hipEventRecord(mm->frame_ev[q->slot][t], q->main));
                    BDE_HIP(hipStreamWaitEvent(q->side, mm->frame_ev[q->slot][t], 0));
                    return decode_fn(mm, t0 * q->B, q->chunk * q->B, q->T, q->B, q->H, q->W, q->side, PART_ALL);
                };
            }
            BDE_TRY(run_attention_level(m, l, T, B, h, w, s, fn, &sc));
            if (fork) {
                BDE_HIP(hipEventRecord(m->join_ev[slot], m->side[slot]));
                BDE_HIP(hipStreamWaitEvent(s, m->join_ev[slot], 0));
                decoded = true;
            }
        }
        if (l == L - 1 && c.depths[l] == 0) BDE_TRY(run_bottleneck_level(m, l, T, B, h, w, s));
        target = ws.merged[l];
    }
    if (m->debug_skip & 8) return BDE_OK;
    // C. decoder: every frame, or the last chunk behind a forked decode; PART_MAIN stops in front of its last convolution
    const int dpart = part == PART_MAIN ? PART_MAIN : PART_ALL;
    if (decoded) return decode_frames(m, tail_f0, tail_nf, T, B, H, W, s, dpart);
    return decode_frames(m, 0, (int)TB, T, B, H, W, s, dpart);
}

static int validate_config(const bde_config* c) {
    BDE_REQUIRE(c != nullptr, "null config");
    BDE_REQUIRE(c->num_encoders >= 1 && c->num_encoders <= BDE_MAX_LEVELS, "num_encoders=%d", c->num_encoders);
    BDE_REQUIRE(c->num_bins >= 1 && c->basechannels >= 1, "num_bins=%d basechannels=%d", c->num_bins, c->basechannels);
    BDE_REQUIRE(c->ks == 3 || c->ks == 5, "ks=%d (3 or 5)", c->ks);
    BDE_REQUIRE(c->frame_num >= 1 && c->frame_num <= BDE_MAX_FRAMES, "frame_num=%d", c->frame_num);
    BDE_REQUIRE(c->q_idx >= 0 && c->q_idx < c->frame_num, "q_idx=%d", c->q_idx);
    BDE_REQUIRE(c->buffer_index[c->q_idx] == 0, "buffer_index[q_idx] must be 0 (the query frame is the current frame)");
    BDE_REQUIRE(c->activation == 0 || c->activation == 1, "activation=%d", c->activation);
    BDE_REQUIRE(c->recurrent_type == 0 || c->recurrent_type == 1, "recurrent_type=%d (0 ConvLSTM, 1 ConvGRU)", c->recurrent_type);
    BDE_REQUIRE(c->use_rc == 0 || c->use_rc == 1, "use_rc=%d", c->use_rc);
    BDE_REQUIRE(c->skip_concat == 0 || c->skip_concat == 1, "skip_concat=%d", c->skip_concat);
    BDE_REQUIRE(c->norm >= 0 && c->norm <= 2, "norm=%d (0 none, 1 BN, 2 IN)", c->norm);
    BDE_REQUIRE(c->num_res_blocks >= 0 && c->num_res_blocks <= 64, "num_res_blocks=%d", c->num_res_blocks);
    for (int l = 0; l < c->num_encoders; ++l) {
        BDE_REQUIRE(c->depths[l] >= 0, "depths[%d]=%d", l, c->depths[l]);
        if (c->depths[l] > 0) {
            const int C = c->basechannels << (l + 1);
            BDE_REQUIRE(c->num_heads >= 1 && C % c->num_heads == 0, "C=%d not divisible by %d heads", C, c->num_heads);
            const int hd = C / c->num_heads;
            BDE_REQUIRE(hd == 1 || hd == 2 || hd == 4 || hd == 8 || hd == 16 || hd == 32, "head_dim=%d not built", hd);
        }
    }
    return BDE_OK;
}

}  // namespace bde

// ==========================================================================================
// C ABI
// ==========================================================================================
extern "C" {
#pragma GCC visibility push(default)

const char* bde_last_error(void) { return last_error_ref().c_str(); }
int bde_abi_version(void) { return 4; }   // 4: BDE_ERR_RANGE, "sb_auto" and the range guard of the two-term format; bde_wait_outputs may recompute

int bde_create(const bde_config* cfg, bde_model** out) {
    BDE_REQUIRE(out != nullptr, "null out");
    BDE_TRY(validate_config(cfg));
    bde_model* m = new bde_model();
    // BDE_SB_TERMS=3 in the environment: every model of the process starts in the three-term bf16 operand format (split.h) --
    // fp32's exponent range for feature maps that can pass 65519, without touching the caller's code (= set_tuning("sb_terms", 3))
    if (const char* e = getenv("BDE_SB_TERMS")) {
        if (e[0] == '3' && e[1] == 0) m->sb_terms = 3;
        else if (e[0] == '2' && e[1] == 0) m->sb_terms = 2;
    }
    m->cfg = *cfg;
    m->L = cfg->num_encoders;
    *out = m;
    return BDE_OK;
}

void bde_destroy(bde_model* m) {
    if (!m) return;
    for (auto& w : m->wslots) w.release();
    if (m->cap_stream) (void)hipStreamDestroy(m->cap_stream);
    for (int i = 0; i < bde_model::MAX_SLOTS; ++i) {
        if (m->pin[i]) (void)hipEventDestroy(m->pin[i]);
        if (m->pout[i]) (void)hipEventDestroy(m->pout[i]);
        // (pstream[i] belongs to the per-device pool, pipeline_stream())
    }
    if (m->dev) (void)hipFree(m->dev);
    if (m->ovf_dev) (void)hipFree(m->ovf_dev);
    if (m->ovf_host) (void)hipHostFree(m->ovf_host);
    for (auto& p : m->pend) if (p.done) (void)hipEventDestroy(p.done);
    for (int i = 0; i < 4; ++i) {
        if (m->dir_fork[i]) (void)hipEventDestroy(m->dir_fork[i]);
        if (m->dir_join[i]) (void)hipEventDestroy(m->dir_join[i]);
        if (m->dir_stream[i]) (void)hipStreamDestroy(m->dir_stream[i]);
    }
    for (auto& sp : m->prof) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
    for (auto e : m->prof_pool) (void)hipEventDestroy(e);
    for (int i = 0; i <= bde_model::MAX_SLOTS; ++i) {
        for (auto e : m->frame_ev[i]) (void)hipEventDestroy(e);
        if (m->join_ev[i]) (void)hipEventDestroy(m->join_ev[i]);
        if (m->side[i]) (void)hipStreamDestroy(m->side[i]);
    }
    delete m;
}

int bde_load_weight(bde_model* m, const char* key, const float* data, const int64_t* shape, int32_t ndim) {
    BDE_REQUIRE(m && key && data && shape && ndim >= 1 && ndim <= 8, "bad argument");
    int64_t n = 1;
    std::vector<int64_t> sh(shape, shape + ndim);
    for (auto v : sh) {
        BDE_REQUIRE(v >= 1, "weight '%s' has a non-positive dimension", key);
        n *= v;
    }
    m->raw[key] = {sh, std::vector<float>(data, data + n)};
    m->finalized = false;
    return BDE_OK;
}

int bde_finalize_weights(bde_model* m) {
    BDE_REQUIRE(m != nullptr, "null model");
    BDE_TRY(build_packed(m));           // host only (runs, and is checked, on a box without a GPU as well)
    BDE_HIP(hipGetDevice(&m->device));
    return upload(m);
}

int bde_alloc_packed(bde_model* m) {
    // Same layout as bde_finalize_weights but with zero content (to be overwritten by a broadcast):
    // synthesise zero tensors for every key the packer asks for.
    BDE_REQUIRE(m != nullptr, "null model");
    const bde_config& c = m->cfg;
    auto put = [&](const std::string& k, std::vector<int64_t> sh) {
        int64_t n = 1;
        for (auto v : sh) n *= v;
        m->raw[GP + k] = {sh, std::vector<float>((size_t)n, 0.f)};
    };
    const int L = c.num_encoders, ks = c.ks, bc = c.basechannels;
    auto put_convlayer = [&](const std::string& p, int cout, int cin) {      // ConvLayer / UpsampleConvLayer parameters
        put(p + "conv2d.weight", {cout, cin, ks, ks});
        if (c.norm != 1) put(p + "conv2d.bias", {cout});
        if (c.norm == 1) { put(p + "norm_layer.weight", {cout}); put(p + "norm_layer.bias", {cout}); }
        if (c.norm) { put(p + "norm_layer.running_mean", {cout}); put(p + "norm_layer.running_var", {cout}); }
    };
    put_convlayer("head.", bc, c.num_bins);
    const char* dirs[2] = {"forward_encoder", "backward_encoder"};
    for (int d = 0; d < 2; ++d)
        for (int l = 0; l < L; ++l) {
            std::string p = std::string(dirs[d]) + "." + std::to_string(l) + ".";
            const int ci = bc << l, co = bc << (l + 1);
            if (!c.use_rc) { put_convlayer(p, co, ci); continue; }
            put_convlayer(p + "conv.", co, ci);
            if (c.recurrent_type == 1) {
                for (const char* gate : {"update_gate", "reset_gate", "out_gate"}) {
                    put(p + "recurrent_block." + gate + ".weight", {co, 2 * co, 3, 3});
                    put(p + "recurrent_block." + gate + ".bias", {co});
                }
            } else {
                put(p + "recurrent_block.Gates.weight", {4 * co, 2 * co, 3, 3});
                put(p + "recurrent_block.Gates.bias", {4 * co});
            }
        }
    if (c.depths[L - 1] == 0)
        for (int k = 0; k < c.num_res_blocks; ++k) {
            std::string p = "feat_attns." + std::to_string(L - 1) + "." + std::to_string(1 + k) + ".";
            const int C = bc << L;
            put(p + "conv1.weight", {C, C, 3, 3}); put(p + "conv1.bias", {C});
            put(p + "conv2.weight", {C, C, 3, 3}); put(p + "conv2.bias", {C});
        }
    const int tbl = (2 * c.frame_num - 1) * 169;
    for (int l = 0; l < L; ++l) {
        const int C = bc << (l + 1);
        for (int i = 0; i < c.depths[l]; ++i) {
            std::string p = "feat_attns." + std::to_string(l) + ".blocks." + std::to_string(i) + ".";
            put(p + "attn.relative_position_bias_table", {tbl, c.num_heads});
            put(p + "attn.norm_q.weight", {C}); put(p + "attn.norm_q.bias", {C});
            put(p + "attn.norm_kv.weight", {C}); put(p + "attn.norm_kv.bias", {C});
            put(p + "attn.q.weight", {C, C}); put(p + "attn.q.bias", {C});
            put(p + "attn.kv.weight", {2 * C, C}); put(p + "attn.kv.bias", {2 * C});
            put(p + "attn.proj.weight", {C, C}); put(p + "attn.proj.bias", {C});
            put(p + "norm2.weight", {C}); put(p + "norm2.bias", {C});
            put(p + "mlp.fc1.weight", {4 * C, C}); put(p + "mlp.fc1.bias", {4 * C});
            put(p + "mlp.fc2.weight", {C, 4 * C}); put(p + "mlp.fc2.bias", {C});
        }
    }
    for (int j = 0; j < L; ++j) {
        put_convlayer("decoders." + std::to_string(j) + ".1.", bc << (L - 1 - j), bc << (L - j));
        if (c.skip_concat) {
            put("decoders." + std::to_string(j) + ".0.weight", {bc << (L - j), 2 * (bc << (L - j)), 1, 1});
            put("decoders." + std::to_string(j) + ".0.bias", {bc << (L - j)});
        }
    }
    if (c.skip_concat) { put("predI.0.weight", {bc, 2 * bc, 1, 1}); put("predI.0.bias", {bc}); }
    put("predI.1.weight", {1, bc, 1, 1});
    put("predI.1.bias", {1});
    return bde_finalize_weights(m);
}

int64_t bde_packed_numel(const bde_model* m) { return m ? m->dev_numel : 0; }
float* bde_packed_ptr(bde_model* m) { return m ? m->dev : nullptr; }

int bde_forward(bde_model* m, const float* const* events, int32_t T, int32_t B, int32_t Hp, int32_t Wp,
                float* const* images, void* stream) {
    BDE_REQUIRE(m && events && images, "null argument");
    for (int t = 0; t < T; ++t) BDE_REQUIRE(events[t] && images[t], "null frame pointer at t=%d", t);
    TuningScope ts(&m->tune);
    return forward_impl(m, events, T, B, Hp, Wp, images, (hipStream_t)stream);
}

// ---- one sequence over two GPUs, split by sweep direction (SURVEY.md §8e option 1; V5.py:122-147) --------------------------
// Rank A runs the forward sweeps, the merge, the attention and the decoder; rank B the backward sweeps.  Per level the caller
// moves two tensors between the ranks' workspaces (bde_split_buffer): B's hidden sequence to A before the merge, and A's level
// output to B as the next level's input.  Every launch is the one the joint forward issues for that direction (shapes are
// chosen as for both directions), so rank A's frames equal the single-GPU frames bit for bit.
static int split_dims(bde_model* m, int* T, int* B, int* H, int* W) {
    Workspace& ws = m->wslots[0];
    BDE_REQUIRE(ws.T > 0, "bde_split_begin has not run");
    *T = ws.T; *B = ws.B; *H = ws.H; *W = ws.W;
    return BDE_OK;
}

int bde_split_begin(bde_model* m, const float* const* events, int32_t T, int32_t B, int32_t Hp, int32_t Wp, void* stream) {
    BDE_REQUIRE(m && events, "null argument");
    for (int t = 0; t < T; ++t) BDE_REQUIRE(events[t], "null frame pointer at t=%d", t);
    TuningScope ts(&m->tune);
    hipStream_t s = (hipStream_t)stream;
    m->cur = 0;
    BDE_TRY(check_dims(m, T, B, Hp, Wp));
    BDE_TRY(ensure_workspace(m, T, B, Hp, Wp));
    Workspace& ws = m->W();
    BDE_TRY(copy_frames(events, ws.ev, T, (long)B * m->cfg.num_bins * Hp * Wp, 0, s));
    if (m->ovf()) BDE_HIP(hipMemsetAsync(m->ovf(), 0, sizeof(unsigned), s));      // range guard (split.h): checked by bde_split_decode
    ConvCall hc;
    hc.pl = &m->head; hc.in = ws.ev; hc.out = ws.head; hc.N = T * B; hc.Hs = Hp; hc.Ws = Wp; hc.act = ACT_RELU;
    return run_conv(m, hc, s);
}

int bde_split_sweep(bde_model* m, int32_t level, int32_t direction, void* stream) {
    BDE_REQUIRE(m && level >= 0 && level < m->L && (direction == 0 || direction == 1), "bad argument");
    TuningScope ts(&m->tune);
    int T, B, H, W;
    m->cur = 0;
    BDE_TRY(split_dims(m, &T, &B, &H, &W));
    Workspace& ws = m->W();
    const float* target = level == 0 ? ws.head : ws.merged[level - 1];
    m->dir_mask = 1 << direction;
    const int st = run_recurrent_level(m, level, target, T, B, H >> level, W >> level, (hipStream_t)stream);
    m->dir_mask = 3;
    return st;
}

int bde_split_attend(bde_model* m, int32_t level, void* stream) {
    BDE_REQUIRE(m && level >= 0 && level < m->L, "bad argument");
    TuningScope ts(&m->tune);
    hipStream_t s = (hipStream_t)stream;
    int T, B, H, W;
    m->cur = 0;
    BDE_TRY(split_dims(m, &T, &B, &H, &W));
    Workspace& ws = m->W();
    const int h = H >> (level + 1), w = W >> (level + 1);
    const long n = (long)T * B * m->cout(level) * h * w;
    BDE_TRY(add2(ws.hseq[level], ws.hseq[level] + n, ws.merged[level], n, s));                  // V5.py:137-147
    if (m->cfg.depths[level] > 0) return run_attention_level(m, level, T, B, h, w, s);
    if (level == m->L - 1) return run_bottleneck_level(m, level, T, B, h, w, s);
    return BDE_OK;
}

int bde_split_decode(bde_model* m, float* const* images, void* stream) {
    BDE_REQUIRE(m && images, "null argument");
    TuningScope ts(&m->tune);
    hipStream_t s = (hipStream_t)stream;
    int T, B, H, W;
    m->cur = 0;
    BDE_TRY(split_dims(m, &T, &B, &H, &W));
    for (int t = 0; t < T; ++t) BDE_REQUIRE(images[t], "null frame pointer at t=%d", t);
    BDE_TRY(decode_frames(m, 0, T * B, T, B, H, W, s));
    BDE_TRY(copy_frames(images, m->W().out, T, (long)B * H * W, 1, s));
    if (m->ovf()) {
        // range guard of the two-term operand format (split.h): no automatic recomputation here -- the two ranks of a split
        // forward would have to switch formats together (dist.DirectionSplit does that through "sb_overflow_word")
        unsigned v = 0;
        BDE_HIP(hipMemcpyAsync(m->ovf_host, m->ovf(), sizeof v, hipMemcpyDeviceToHost, s));
        BDE_HIP(hipStreamSynchronize(s));
        v = m->ovf_host[0];
        m->ovf_host[0] = 0;
        if (v) {
            BDE_HIP(hipMemset(m->ovf(), 0, sizeof v));
            ++m->sb_overflows;
            return fail(BDE_ERR_RANGE, "split forward: an activation reached 65520, beyond the two fp16 terms of the default operand "
                        "format (csrc/split.h); set_tuning(\"sb_terms\", 3) on both ranks and run it again");
        }
    }
    return BDE_OK;
}

int bde_split_buffer(bde_model* m, const char* what, int32_t level, int32_t direction, float** ptr, int64_t* numel) {
    BDE_REQUIRE(m && what && ptr && numel && level >= 0 && level < m->L, "bad argument");
    int T, B, H, W;
    m->cur = 0;
    BDE_TRY(split_dims(m, &T, &B, &H, &W));
    Workspace& ws = m->W();
    const long n = (long)T * B * m->cout(level) * (long)(H >> (level + 1)) * (W >> (level + 1));
    const std::string k(what);
    if (k == "hidden") {
        BDE_REQUIRE(direction == 0 || direction == 1, "direction=%d", direction);
        *ptr = ws.hseq[level] + (long)direction * n;
    } else if (k == "level_out") {
        *ptr = ws.merged[level];
    } else {
        return fail(BDE_ERR_ARG, "unknown split buffer '%s' (hidden | level_out)", what);
    }
    *numel = n;
    return BDE_OK;
}

int bde_wait_outputs(bde_model* m, void* stream) {
    BDE_REQUIRE(m != nullptr, "null model");
    TuningScope ts(&m->tune);
    // range guard (split.h): forwards whose operands left fp16's range are recomputed here (or fail the call) -- on their own
    // streams, i.e. still ahead of the events `stream` is about to wait for
    const int st = settle_overflow(m);
    for (int i = 0; i < bde_model::MAX_SLOTS; ++i)
        if (m->pbusy[i]) {
            BDE_HIP(hipStreamWaitEvent((hipStream_t)stream, m->pout[i], 0));
            m->pbusy[i] = false;
        }
    return st;
}

int bde_set_tuning(bde_model* m, const char* key, int64_t value) {
    BDE_REQUIRE(m && key, "null argument");
    if (std::string(key) == "graph") { m->use_graph = (int)value; return BDE_OK; }
    if (std::string(key) == "pipeline") {
        BDE_REQUIRE(value >= 1 && value <= bde_model::MAX_SLOTS, "pipeline depth must be 1..%d", bde_model::MAX_SLOTS);
        m->pipeline = (int)value;
        return BDE_OK;
    }
    if (std::string(key) == "sb_auto") {
        BDE_REQUIRE(value == 0 || value == 1, "sb_auto: 0 (an overflow of the two-term format fails the call) or 1 (recompute with three terms)");
        m->sb_auto = (int)value;
        return BDE_OK;
    }
    {   // forwards still in flight keep their events in the workspaces: look at their overflow words before anything is released
        TuningScope ts(&m->tune);
        for (const auto& p : m->pend) if (p.on) { BDE_TRY(settle_overflow(m)); break; }
    }
    for (auto& w : m->wslots)
        if (w.graph_exec) { (void)hipGraphExecDestroy(w.graph_exec); w.graph_exec = nullptr; }
    if (std::string(key) == "attn_mfma") { m->tune.attn_mfma = (int)value; return BDE_OK; }
    if (std::string(key) == "lstm_hc8") { m->lstm_hc8 = (int)value; return BDE_OK; }
    if (std::string(key) == "winblock") {
        if (m->winblock != (int)value)
            for (auto& w : m->wslots) w.release();           // the two paths stage different buffers
        m->winblock = (int)value;
        return BDE_OK;
    }
    if (std::string(key) == "winblock_sb") { m->winblock_sb = (int)value; return BDE_OK; }
    if (std::string(key) == "wide") {
        if (m->wide != (int)value)
            for (auto& w : m->wslots) w.release();
        m->wide = (int)value;
        return BDE_OK;
    }
    if (std::string(key) == "wide_fuse_qkv") { m->wide_fuse_qkv = (int)value; return BDE_OK; }
    if (std::string(key) == "wide_kv_sb") { m->wide_kv_sb = (int)value; return BDE_OK; }
    if (std::string(key) == "wide_fuse_mlp") { m->wide_fuse_mlp = (int)value; return BDE_OK; }
    if (std::string(key) == "wide_fuse_fc2") { m->wide_fuse_fc2 = (int)value; return BDE_OK; }
    if (std::string(key) == "wide_core2") {
        if (m->wide_core2 != (int)value)
            for (auto& w : m->wslots) w.release();           // the SPL16 twins exist only with it
        m->wide_core2 = (int)value;
        return BDE_OK;
    }
    if (std::string(key) == "wide_spl") { m->wide_spl = (int)value; return BDE_OK; }
    if (std::string(key) == "conv_sb") {
        if (m->conv_sb != (int)value)
            for (auto& w : m->wslots) w.release();           // which recurrent step runs (and its buffers) depends on it
        m->conv_sb = (int)value;
        return BDE_OK;
    }
    if (std::string(key) == "sb_terms") {
        BDE_REQUIRE(value == 2 || value == 3, "sb_terms: 2 (two fp16 terms) or 3 (three bf16 terms)");
        if (m->sb_terms != (int)value)
            for (auto& w : m->wslots) w.release();           // launch shapes and buffer roles depend on the format
        m->sb_terms = (int)value;
        m->sb_latched = 0;
        return BDE_OK;
    }
    if (std::string(key) == "fuse_enc_sb") { m->fuse_enc_sb = (int)value; return BDE_OK; }
    if (std::string(key) == "xcd_remap") { m->xcd_remap = (int)value; return BDE_OK; }
    if (std::string(key) == "lstm_two_streams") { m->lstm_two_streams = (int)value; return BDE_OK; }
    if (std::string(key) == "lstm_fuse_x") {
        if (m->lstm_fuse_x != (int)value)
            for (auto& w : m->wslots) w.release();           // the x-part buffer exists only without it
        m->lstm_fuse_x = (int)value;
        return BDE_OK;
    }
    if (std::string(key) == "lstm_sbk") {
        if (m->use_lstm_sbk != (int)value)
            for (auto& w : m->wslots) w.release();
        m->use_lstm_sbk = (int)value;
        return BDE_OK;
    }
    if (std::string(key) == "lstm_sb") {
        if (m->lstm_sb_mode != (int)value)
            for (auto& w : m->wslots) w.release();
        m->lstm_sb_mode = (int)value;
        return BDE_OK;
    }
    if (std::string(key) == "fuse_pred") { m->fuse_pred = (int)value; return BDE_OK; }
    if (std::string(key) == "fused_min_tiles") { m->fused_min_tiles = value; return BDE_OK; }
    if (std::string(key) == "pw_batched") { m->tune.pw_batched = (int)value; return BDE_OK; }
    if (std::string(key) == "pw_force") { m->tune.pw_force = (int)value; return BDE_OK; }
    if (std::string(key) == "conv_nt") { m->tune.conv_nt = (int)value; return BDE_OK; }
    if (std::string(key) == "conv_vec") { m->tune.conv_vec = (int)value; return BDE_OK; }
    if (std::string(key) == "lstm_shape") { m->tune.lstm_shape = (int)value; return BDE_OK; }
    if (std::string(key) == "tok_npt") { m->tune.tok_npt = (int)value; return BDE_OK; }
    if (std::string(key) == "tok_debug") { m->tok_debug = (int)value; return BDE_OK; }
    if (std::string(key) == "overlap") { m->overlap = (int)value; return BDE_OK; }
    if (std::string(key) == "eager_cut") { m->eager_cut = (int)value; return BDE_OK; }
    if (std::string(key) == "debug_skip") { m->debug_skip = (int)value; return BDE_OK; }
    if (std::string(key) == "overlap_chunk") { m->overlap_chunk = std::max<int>(1, (int)value); return BDE_OK; }
    return fail(BDE_ERR_ARG, "unknown tuning key '%s'", key);
}

int bde_get_info(const bde_model* m, const char* key, int64_t* value) {
    BDE_REQUIRE(m && key && value, "null argument");
    const std::string k(key);
    if (k == "debug_skip") *value = m->debug_skip;                 // != 0: stages are skipped, results are invalid
    else if (k == "graph") *value = m->use_graph;                  // 0 after a failed capture as well
    else if (k == "graphs_live") {                                 // workspaces replaying a captured launch sequence
        int64_t n = 0;
        for (const auto& w : m->wslots) n += w.graph_exec != nullptr;
        *value = n;
    } else if (k == "pipeline") *value = m->pipeline;
    else if (k == "last_stream") *value = (int64_t)(uintptr_t)m->last_stream;   // internal stream of the latest pipelined call
    else if (k == "device") *value = m->device;
    else if (k == "winblock") *value = m->winblock;
    else if (k == "winblock_sb") *value = m->winblock_sb;
    else if (k == "wide") *value = m->wide;
    else if (k == "conv_sb") *value = m->conv_sb;
    else if (k == "lstm_sb") *value = m->lstm_sb_mode;
    else if (k == "lstm_sbk") *value = m->use_lstm_sbk;
    else if (k == "lstm_fuse_x") *value = m->lstm_fuse_x;
    else if (k == "wide_kv_sb") *value = m->wide_kv_sb;
    else if (k == "wide_fuse_mlp") *value = m->wide_fuse_mlp;
    else if (k == "wide_fuse_fc2") *value = m->wide_fuse_fc2;
    else if (k == "wide_core2") *value = m->wide_core2;
    else if (k == "wide_spl") *value = m->wide_spl;
    else if (k == "sb_terms") *value = m->sb_terms;
    else if (k == "sb_auto") *value = m->sb_auto;
    else if (k == "sb_overflows") *value = m->sb_overflows;       // forwards settled so far whose two-term operands left fp16's range
    else if (k == "sb_latched") *value = m->sb_latched;           // 1: such a forward switched the model to three bf16 terms
    else if (k == "sb_overflow_word") {                           // the overflow word of slot 0 as it stands (op-level entry points
        unsigned v = 0;                                           //  and bde_split_sweep raise it without anybody settling it);
        if (m->ovf_dev) {                                         //  reading clears it
            BDE_HIP(hipDeviceSynchronize());
            BDE_HIP(hipMemcpy(&v, m->ovf_dev, sizeof v, hipMemcpyDeviceToHost));
            BDE_HIP(hipMemset(m->ovf_dev, 0, sizeof v));
        }
        *value = v;
    }
    else if (k == "packed_numel") *value = m->dev_numel;
    else if (k == "sb_head") *value = m->head.sb_used;
    else if (k.compare(0, 3, "sb_") == 0 && k.size() >= 5 && k.back() >= '0' && k.back() - '0' < m->L) {
        // did the latest launch of that layer run as split bf16 (conv_sb.h)?  "sb_enc<l>", "sb_gx<l>", "sb_dec<j>"
        const int i = k.back() - '0';
        const std::string what = k.substr(3, k.size() - 4);
        if (what == "enc") *value = m->enc[i].sb_used;
        else if (what == "gx") *value = m->gx[i].sb_used;
        else if (what == "dec") *value = m->dec[i].sb_used;
        else if (what == "lstm") *value = (size_t)i < m->lstm_sbk.size() ? m->lstm_sbk[i].sb_used : 0;
        else return fail(BDE_ERR_ARG, "unknown info key '%s'", key);
    }
    else return fail(BDE_ERR_ARG, "unknown info key '%s'", key);
    return BDE_OK;
}

int bde_debug_token_stamps(bde_model* m, int64_t* host_out, int32_t n) {
    // diagnostic: enable (host_out == NULL) or read back (n values) the token-kernel phase stamps
    BDE_REQUIRE(m != nullptr, "null model");
    if (!m->tok_stamps) {
        BDE_HIP(hipMalloc((void**)&m->tok_stamps, sizeof(unsigned long long) * 64 * 4 * 8));
        BDE_HIP(hipMemset(m->tok_stamps, 0, sizeof(unsigned long long) * 64 * 4 * 8));
        BDE_HIP(hipMemset(m->tok_stamps + 2046, 0xff, sizeof(unsigned long long)));     // (slot 2046: an atomicMin target, lstm_sb.h)
    }
    if (host_out) {
        BDE_HIP(hipDeviceSynchronize());
        BDE_HIP(hipMemcpy(host_out, m->tok_stamps, sizeof(int64_t) * std::min(n, 64 * 4 * 8), hipMemcpyDeviceToHost));
        BDE_HIP(hipMemset(m->tok_stamps + 2046, 0xff, sizeof(unsigned long long)));
        BDE_HIP(hipMemset(m->tok_stamps + 2047, 0, sizeof(unsigned long long)));
    }
    return BDE_OK;
}

int bde_debug_conv_shape(int32_t ks, int32_t stride, int32_t cout, int32_t in_h, int32_t in_w, int32_t* row_tiles) {
    const int pad = ks / 2;
    const int Ho = (in_h + 2 * pad - ks) / stride + 1, Wo = (in_w + 2 * pad - ks) / stride + 1;
    const int terms = BDE_DEFAULT_SB_TERMS;
    const int shape = conv_sb_pick(ks, stride, cout, in_w, Ho, Wo, terms);
    if (row_tiles) {
        if (shape == SB_32x256T) *row_tiles = -conv_sb_tile_cols(256, Ho, Wo);          // 2-D tiles: minus the tile's columns
        else if (shape == SB_64x128T) *row_tiles = -conv_sb_tile_cols(128, Ho, Wo);
        else *row_tiles = shape == SB_NONE ? 0 : conv_sb_tile_mode(ks, stride, shape == SB_128x64 ? 64 : 128, in_w, Ho, Wo, terms);
    }
    return shape;
}

float bde_debug_split(const float* x, int64_t n, int32_t terms, float scale, uint16_t* out) {
    if (!x || !out || n <= 0 || (terms != 2 && terms != 3)) return 0.f;
    for (int64_t i = 0; i < n; ++i) {
        unsigned short t[3];
        split_terms(x[i], terms, scale, t);
        for (int k = 0; k < terms; ++k) out[i * terms + k] = t[k];
    }
    return terms == 2 ? sb_weight_scale(x, (long)n) : 1.f;
}

int bde_debug_occupancy(const char* kernel) {
    int nb = -1;
    std::string k(kernel ? kernel : "");
    if (k == "token_fused") {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, token_fused_kernel<2>, 256, token_lds_bytes(64)) != hipSuccess) return -1;
        return nb;
    }
    return conv_tu_occupancy(kernel);
}

int bde_profile_reset(bde_model* m, int32_t enable) {
    BDE_REQUIRE(m != nullptr, "null model");
    for (auto& sp : m->prof) { m->prof_pool.push_back(sp.a); m->prof_pool.push_back(sp.b); }
    m->prof.clear();
    m->prof_on = enable != 0;
    for (auto& w : m->wslots)          // spans live inside the captured graph: re-capture
        if (w.graph_exec) { (void)hipGraphExecDestroy(w.graph_exec); w.graph_exec = nullptr; }
    return BDE_OK;
}

int bde_profile_names(bde_model* m, char* buf, int64_t buflen) {
    // distinct span names recorded since the last reset, separated by '\n' (truncated to buflen)
    BDE_REQUIRE(m && buf && buflen > 0, "bad argument");
    std::vector<std::string> seen;
    std::string out;
    for (auto& sp : m->prof)
        if (std::find(seen.begin(), seen.end(), sp.name) == seen.end()) {
            seen.push_back(sp.name);
            out += sp.name;
            out += '\n';
        }
    snprintf(buf, (size_t)buflen, "%s", out.c_str());
    return BDE_OK;
}

int bde_profile_get(bde_model* m, const char* name, double* total_ms, int64_t* count) {
    BDE_REQUIRE(m && name && total_ms && count, "null argument");
    double tot = 0.0;
    int64_t n = 0;
    for (auto& sp : m->prof) {
        if (sp.name != name) continue;
        BDE_HIP(hipEventSynchronize(sp.b));
        float ms = 0.f;
        BDE_HIP(hipEventElapsedTime(&ms, sp.a, sp.b));
        tot += ms;
        ++n;
    }
    *total_ms = tot;
    *count = n;
    return BDE_OK;
}

int bde_get_intermediate(bde_model* m, const char* name, float* dst, int64_t numel, void* stream) {
    BDE_REQUIRE(m && name && dst, "null argument");
    Workspace& ws = m->W();
    BDE_REQUIRE(ws.T > 0, "no forward has run");
    const long TB = (long)ws.T * ws.B;
    const float* src = nullptr;
    long n = 0;
    std::string nm(name);
    if (nm == "head") {
        src = ws.head;
        n = TB * m->cfg.basechannels * ws.H * ws.W;
    } else if (nm.rfind("merged", 0) == 0) {
        int l = atoi(nm.c_str() + 6);
        BDE_REQUIRE(l >= 0 && l < m->L, "level %d", l);
        src = ws.merged[l];
        n = TB * m->cout(l) * (long)(ws.H >> (l + 1)) * (ws.W >> (l + 1));
    } else if (nm.rfind("dec", 0) == 0) {
        int j = atoi(nm.c_str() + 3);
        BDE_REQUIRE(j >= 0 && j < m->L, "decoder %d", j);
        BDE_REQUIRE(!(j == m->L - 1 && pred_fusable(m)), "dec%d is not materialised: predI is fused into its conv (set_tuning fuse_pred 0)", j);
        const int l = m->L - 1 - j;
        src = ws.dec[j];
        n = TB * m->cin(l) * (long)(ws.H >> l) * (ws.W >> l);
    } else {
        return fail(BDE_ERR_ARG, "unknown intermediate '%s'", name);
    }
    BDE_REQUIRE(numel == n, "intermediate '%s' has %ld values, caller asked for %ld", name, n, (long)numel);
    BDE_HIP(hipMemcpyAsync(dst, src, sizeof(float) * n, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return BDE_OK;
}

// Binning kernel behind every bde_voxelize* call of the process: 0 = automatic (default: the bucketed path once a call is large
// enough for its extra launch to pay, else the streaming tile kernel), 1 = global float-atomic scatter (the first kernel; A/B),
// 2 = bucketed always, 3 = streaming tile kernel always (last round's default; A/B)
static int& voxel_method_ref() { static int v = 0; return v; }
int bde_voxel_method(int32_t method) {
    BDE_REQUIRE(method >= 0 && method <= 3, "voxel method %d", method);
    voxel_method_ref() = method;
    return BDE_OK;
}
// The bucketed path moves 29 B per event in two launches; the streaming kernel reads 4 B per event and tile (+ 9 B once) in one.
static bool voxel_use_buckets(int nb, int H, int W, long max_win, int nseg) {
    const int method = voxel_method_ref();
    if (method == 2) return true;
    if (method != 0 || max_win <= 0) return false;
    int TH, TW, ntw, nth;
    if (voxel_tile_geometry(nb, H, W, &TH, &TW, &ntw, &nth, 8, VB_TILE_LDS) != BDE_OK) return false;
    const long tiles = (long)ntw * nth;
    if (tiles > VB_MAX_TILES || TH * TW > 8192) return false;
    return tiles >= 4 && max_win * nseg >= 2000000 && max_win >= 16384;
}

int bde_voxelize(const float* xs, const float* ys, const float* ts, const float* ps, int64_t N, int32_t num_bins,
                 int32_t H, int32_t W, float* grid, int32_t* oob_count, void* stream) {
    BDE_REQUIRE(grid && num_bins >= 1 && H >= 1 && W >= 1 && N >= 0, "bad argument");
    BDE_REQUIRE(N == 0 || (xs && ys && ts && ps), "null event array");
    if (voxel_method_ref() == 1)
        return voxel_launch(xs, ys, ts, ps, nullptr, 1, (long)N, num_bins, H, W, grid, oob_count, (hipStream_t)stream);
    if (voxel_use_buckets(num_bins, H, W, (long)N, 1))
        return voxel_bucket_launch<false>(xs, ys, ts, ps, nullptr, nullptr, (long)N, 1, (long)N, num_bins, H, W, grid, oob_count,
                                          (hipStream_t)stream);
    return voxel_tile_launch<false>(xs, ys, ts, ps, nullptr, nullptr, (long)N, 1, num_bins, H, W, grid, oob_count, (hipStream_t)stream);
}

int bde_voxelize_batch(const float* xs, const float* ys, const float* ts, const float* ps, const int64_t* offsets,
                       int32_t nseg, int64_t max_events_per_seg, int32_t num_bins, int32_t H, int32_t W, float* grids,
                       int32_t* oob_count, void* stream) {
    BDE_REQUIRE(grids && offsets && nseg >= 1 && num_bins >= 1 && H >= 1 && W >= 1, "bad argument");
    BDE_REQUIRE(xs && ys && ts && ps, "null event array");
    static_assert(sizeof(long) == sizeof(int64_t), "LP64 expected");
    if (voxel_method_ref() == 1)
        return voxel_launch(xs, ys, ts, ps, (const long*)offsets, nseg, (long)max_events_per_seg, num_bins, H, W, grids,
                            oob_count, (hipStream_t)stream);
    if (voxel_use_buckets(num_bins, H, W, (long)max_events_per_seg, nseg))
        return voxel_bucket_launch<false>(xs, ys, ts, ps, (const long*)offsets, (const long*)offsets + 1, 0, nseg, (long)max_events_per_seg,
                                          num_bins, H, W, grids, oob_count, (hipStream_t)stream);
    return voxel_tile_launch<false>(xs, ys, ts, ps, (const long*)offsets, (const long*)offsets + 1, 0, nseg, num_bins, H, W, grids,
                                    oob_count, (hipStream_t)stream);
}

int bde_voxelize_events(const int16_t* xs, const int16_t* ys, const double* ts, const uint8_t* ps, const int64_t* offsets,
                        int32_t nwin, int64_t max_events_per_window, int32_t num_bins, int32_t H, int32_t W, float* grids,
                        int32_t* oob_count, void* stream) {
    BDE_REQUIRE(grids && offsets && nwin >= 1 && num_bins >= 1 && H >= 1 && W >= 1, "bad argument");
    BDE_REQUIRE(max_events_per_window <= 0 || (xs && ys && ts && ps), "null event column");
    static_assert(sizeof(long) == sizeof(int64_t), "LP64 expected");
    if (voxel_method_ref() == 1)
        return voxel_native_launch(xs, ys, ts, ps, (const long*)offsets, nwin, (long)max_events_per_window, num_bins, H, W, grids,
                                   oob_count, (hipStream_t)stream);
    if (voxel_use_buckets(num_bins, H, W, (long)max_events_per_window, nwin))
        return voxel_bucket_launch<true>(xs, ys, ts, ps, (const long*)offsets, (const long*)offsets + 1, 0, nwin, (long)max_events_per_window,
                                         num_bins, H, W, grids, oob_count, (hipStream_t)stream);
    return voxel_tile_launch<true>(xs, ys, ts, ps, (const long*)offsets, (const long*)offsets + 1, 0, nwin, num_bins, H, W, grids,
                                   oob_count, (hipStream_t)stream);
}

int bde_voxelize_event_ranges(const int16_t* xs, const int16_t* ys, const double* ts, const uint8_t* ps, int64_t n_events,
                              const int64_t* starts, const int64_t* ends, int32_t nwin, int64_t max_events_per_window, int32_t num_bins,
                              int32_t H, int32_t W, float* grids, int32_t* oob_count, void* stream) {
    BDE_REQUIRE(grids && starts && ends && nwin >= 1 && num_bins >= 1 && H >= 1 && W >= 1 && n_events >= 0, "bad argument");
    BDE_REQUIRE(n_events == 0 || (xs && ys && ts && ps), "null event column");
    if (voxel_method_ref() != 1 && max_events_per_window > 0 && voxel_use_buckets(num_bins, H, W, (long)max_events_per_window, nwin))
        return voxel_bucket_launch<true>(xs, ys, ts, ps, (const long*)starts, (const long*)ends, 0, nwin, (long)max_events_per_window,
                                         num_bins, H, W, grids, oob_count, (hipStream_t)stream, (long)n_events);
    return voxel_tile_launch<true>(xs, ys, ts, ps, (const long*)starts, (const long*)ends, 0, nwin, num_bins, H, W, grids, oob_count,
                                   (hipStream_t)stream, (long)n_events);
}

int bde_find_ts_index(const double* ts, int64_t n, const double* timestamps, int32_t nq, int64_t* out, void* stream) {
    BDE_REQUIRE(out && nq >= 0 && n >= 0 && (n == 0 || ts) && (nq == 0 || timestamps), "bad argument");
    if (nq == 0) return BDE_OK;
    hipLaunchKernelGGL(find_ts_index_kernel, dim3(cdiv(nq, 256)), dim3(256), 0, (hipStream_t)stream, ts, (long)n, timestamps, nq,
                       (long*)out);
    BDE_HIP(hipGetLastError());
    return BDE_OK;
}

int bde_metric_mse(const float* a, const float* b, int64_t numel_per_image, int32_t N, double* scratch, double* out, void* stream) {
    BDE_REQUIRE(a && b && scratch && out && numel_per_image >= 1 && N >= 1, "bad argument");
    return metric_mse_launch(a, b, (long)numel_per_image, N, scratch, out, (hipStream_t)stream);
}

int bde_metric_ssim(const float* a, const float* b, int32_t H, int32_t W, int32_t N, double data_range, double* scratch, double* out,
                    void* stream) {
    BDE_REQUIRE(a && b && scratch && out && N >= 1, "bad argument");
    BDE_REQUIRE(H >= 7 && W >= 7, "win_size 7 exceeds the image extent %dx%d (scikit-image raises here as well)", H, W);
    return metric_ssim_launch(a, b, H, W, N, data_range, scratch, out, (hipStream_t)stream);
}
int32_t bde_metric_scratch_doubles(int32_t N) { return N * METRIC_BLOCKS; }

// ---- single sub-modules ----------------------------------------------------------------------
int bde_op_head(bde_model* m, const float* in, int32_t N, int32_t H, int32_t W, float* out, void* stream) {
    BDE_REQUIRE(m && m->finalized && in && out, "bad argument");
    TuningScope ts(&m->tune);
    ConvCall c;
    c.pl = &m->head; c.in = in; c.out = out; c.N = N; c.Hs = H; c.Ws = W; c.act = ACT_RELU;
    return run_conv(m, c, (hipStream_t)stream);
}

int bde_op_encoder_conv(bde_model* m, int32_t level, int32_t dir, const float* in, int32_t N, int32_t H, int32_t W,
                        float* out, void* stream) {
    BDE_REQUIRE(m && m->finalized && in && out && level >= 0 && level < m->L && (dir == 0 || dir == 1), "bad argument");
    TuningScope ts(&m->tune);
    PackedLayer pl = group_view(m->enc[level], dir);   // a single direction
    pl.G_decide = 0;
    ConvCall c;
    c.pl = &pl; c.in = in; c.out = out; c.N = N; c.Hs = H; c.Ws = W; c.stride = 2; c.act = ACT_RELU;
    return run_conv(m, c, (hipStream_t)stream);
}

int bde_op_recurrent_conv(bde_model* m, int32_t level, int32_t dir, const float* in, int32_t T, int32_t B, int32_t H,
                          int32_t W, float* h_out, float* c_out, void* stream) {
    BDE_REQUIRE(m && m->finalized && in && h_out && level >= 0 && level < m->L && (dir == 0 || dir == 1), "bad argument");
    TuningScope ts(&m->tune);
    BDE_REQUIRE(H % 2 == 0 && W % 2 == 0, "H, W must be even");
    hipStream_t s = (hipStream_t)stream;
    // run the level on a private workspace sized for this call (full resolution = H << level)
    BDE_TRY(ensure_workspace(m, T, B, H << level, W << level));
    BDE_TRY(run_recurrent_level(m, level, in, T, B, H, W, s));
    const int C = m->cout(level);
    const long hw = (long)(H / 2) * (W / 2), TB = (long)T * B;
    // dir 0 sweeps t = 0..T-1, dir 1 sweeps t = T-1..0 (V5.py:123); h_out[t] belongs to input frame t.
    BDE_HIP(hipMemcpyAsync(h_out, m->W().hseq[level] + (long)dir * TB * C * hw, sizeof(float) * TB * C * hw,
                           hipMemcpyDeviceToDevice, s));
    if (c_out)
        BDE_HIP(hipMemcpyAsync(c_out, m->W().cst[level] + (long)dir * B * C * hw, sizeof(float) * B * C * hw,
                               hipMemcpyDeviceToDevice, s));
    return BDE_OK;
}

int bde_op_gate_conv(bde_model* m, int32_t level, const float* in, int32_t N, int32_t H, int32_t W, float* out, void* stream) {
    // x-part of the ConvLSTM gates (submodules.py:316-317, the x half of the stacked input) for both directions:
    // in [2][N][C][H][W] (forward / backward encoder outputs), out [2][N][4C][H][W], bias included
    BDE_REQUIRE(m && m->finalized && in && out && level >= 0 && level < m->L && N >= 1, "bad argument");
    TuningScope ts(&m->tune);
    BDE_TRY(ensure_workspace(m, N, 1, H << (level + 1), W << (level + 1)));
    const int C = m->cout(level);
    ConvCall gxc;
    gxc.pl = &m->gx[level];
    gxc.in = in;
    gxc.out = out;
    gxc.N = N;
    gxc.Hs = H;
    gxc.Ws = W;
    gxc.in_gs = (long)N * C * H * W;
    gxc.out_gs = (long)N * 4 * C * H * W;
    return run_conv(m, gxc, (hipStream_t)stream);
}

int bde_op_decoder(bde_model* m, int32_t j, const float* in, const float* skip, int32_t N, int32_t H, int32_t W,
                   float* out, void* stream) {
    BDE_REQUIRE(m && m->finalized && in && out && j >= 0 && j < m->L, "bad argument");
    TuningScope ts(&m->tune);
    const int L = m->L, l = L - 1 - j;
    BDE_TRY(ensure_workspace(m, N, 1, H << (l + 1), W << (l + 1)));
    return run_decoder(m, j, in, skip, out, N, H, W, (hipStream_t)stream);
}

int bde_op_pred(bde_model* m, const float* in, const float* head, int32_t N, int32_t H, int32_t W, float* out,
                void* stream) {
    BDE_REQUIRE(m && m->finalized && in && out, "bad argument");
    const long total = (long)N * H * W;
    long blocks = std::min<long>(cdivl(total, 256), 4096);
    hipLaunchKernelGGL(pred_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, head,
                       m->P(m->predw_off), m->P(m->predb_off), out, m->cfg.basechannels, (long)H * W, total,
                       m->cfg.activation);
    BDE_HIP(hipGetLastError());
    return BDE_OK;
}

int bde_op_dframe_attention(bde_model* m, int32_t level, const float* const* bufs, int32_t B, int32_t H, int32_t W,
                            int32_t first_block, int32_t nblocks, float* out, void* stream) {
    BDE_REQUIRE(m && m->finalized && bufs && out && level >= 0 && level < m->L, "bad argument");
    TuningScope ts(&m->tune);
    const AttnLevel& al = m->attn[level];
    BDE_REQUIRE(al.depth > 0, "level %d has no attention", level);
    BDE_REQUIRE(H >= 7 && W >= 7, "map %dx%d smaller than the window", H, W);
    if (nblocks < 0) { first_block = 0; nblocks = al.depth; }
    BDE_REQUIRE(first_block >= 0 && first_block + nblocks <= al.depth && nblocks >= 1, "block range");
    const bde_config& c = m->cfg;
    BDE_REQUIRE(bufs[c.q_idx] != nullptr, "the query frame must be given");
    hipStream_t s = (hipStream_t)stream;
    // workspace: treat this map as level `level` of a (H<<(level+1)) x (W<<(level+1)) input, T = frame_num
    const int D = c.frame_num;
    BDE_TRY(ensure_workspace(m, D, B, H << (level + 1), W << (level + 1)));
    const long HW = (long)H * W;
    if (winblock_ok(m, level)) {
        const float* frames[BDE_MAX_FRAMES];
        for (int d = 0; d < D; ++d) {
            frames[d] = nullptr;
            if (bufs[d] == nullptr) continue;
            float* tq = m->W().mergedT[level] + (long)d * B * al.C * HW;
            BDE_TRY(nchw_to_tok(bufs[d], tq, B, al.C, (int)HW, s));
            frames[d] = tq;
        }
        return run_attention_frame_win(m, level, frames, nullptr, m->W().qkv, out, B, H, W, first_block, nblocks, s);
    }
    const long kvfs = (long)B * al.depth * 2 * al.C * HW;
    const float* kvslot[BDE_MAX_FRAMES];
    if (wide_ok(m, level)) {
        Workspace& w = m->W();
        const int C = al.C;
        const long ffs = (long)B * cdivl(HW, 16) * 16 * C;
        int nneg = 0;
        for (int d = 0; d < D; ++d) nneg += d != c.q_idx && c.buffer_index[d] < 0;
        const bool in_core = wide_core2_ok(m, level) && nneg == 1;
        const float* prev_frag = nullptr;
        int prev_slot = -1;
        WideTwin q_twin, p_twin;
        for (int d = 0; d < D; ++d) {
            kvslot[d] = nullptr;
            if (bufs[d] == nullptr) continue;
            float* fr = w.mergedT[level] + (long)d * ffs;
            const bool twins = w.mergedS[level] != nullptr && m->sb_terms == 2;
            WideTwin tw;
            if (twins) {
                tw = WideTwin{w.mergedS[level] + (long)d * ffs, w.mstats[level] + (long)d * B * cdivl(HW, 16) * 16 * 2};
                BDE_TRY(nchw_to_frag_spl(bufs[d], fr, reinterpret_cast<unsigned short*>(tw.s), tw.st, B, C, (int)HW, m->ovf(), s));
            } else {
                BDE_TRY(nchw_to_frag(bufs[d], fr, B, C, (int)HW, s));
            }
            if (d == c.q_idx) { q_twin = tw; continue; }
            // as in the forward: the one frame at a negative offset goes to the attention core as it is (wide_core.h)
            if (in_core && c.buffer_index[d] < 0) { prev_frag = fr; prev_slot = d; p_twin = tw; continue; }
            float* dst = w.kvun[level] + (long)d * kvfs;
            BDE_TRY(run_tokgemm(m, "wide_kv", level, al.kvallW, al.kvall, al.depth * 2 * C, C, fr, B, HW, dst, nullptr, ACT_NONE, nullptr,
                                nullptr, nullptr, 0, 0, 0, 0, s, al.kvallH, al.kvallH_unscale));
            kvslot[d] = dst;
        }
        float* qf = w.mergedT[level] + (long)c.q_idx * ffs;
        // (the caller wants the result as NCHW only: the last block's SPL16 twin goes to a scratch pair)
        return run_attention_frame_wide(m, level, qf, kvslot, nullptr, nullptr, out, B, H, W, first_block, nblocks, nullptr, s, prev_frag,
                                        prev_slot, q_twin, q_twin.s ? WideTwin{w.xaS, w.stA} : WideTwin(), p_twin);
    }
    for (int d = 0; d < D; ++d) {
        kvslot[d] = nullptr;
        if (d == c.q_idx || bufs[d] == nullptr) continue;
        float* dst = m->W().kvun[level] + (long)d * kvfs;
        BDE_TRY(run_pw(m, &al.kvall, bufs[d], dst, B, HW, ACT_NONE, nullptr, nullptr, 0, 0, 0, s));
        kvslot[d] = dst;
    }
    return run_attention_frame(m, level, bufs[c.q_idx], kvslot, nullptr, out, B, H, W, first_block, nblocks, nullptr, s);
}

#pragma GCC visibility pop
}  // extern "C"
