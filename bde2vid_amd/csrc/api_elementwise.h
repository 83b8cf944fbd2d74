// Part of libbde2vid's host side, included by bde_api.hip (one translation unit: the kernels of the headers it includes are
// emitted once).  Element-wise kernels of the forward: merge, frame gather / scatter, bilinear x2 (+ skip sum, + split), predI, ConvGRU gates.
#pragma once
namespace bde {


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
// The same image, a thread = the 2 x 2 output pixels of one source pixel (k, j) x half a chunk: their bilinear taps are the 3 x 3
// source neighbourhood (rows k - 1 .. k + 1, columns j - 1 .. j + 1, clamped), nine loads per tensor and channel for four outputs
// instead of sixteen, and a wave's loads are 64 CONSECUTIVE source floats of a plane row (the per-output-pixel kernel reads every
// source float twice over two lanes).  Same expression per output as upsample2x_sum_kernel / upsample2x_sum_split_kernel (bit-identical).
// grid (ceil(Hs Ws / 128), C/16 chunks, N), 256 threads = 128 source pixels x 2 halves.
template <int TERMS>
__global__ __launch_bounds__(256) void upsample2x_sum_split_quad_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                        unsigned short* __restrict__ out, int C, int Hs, int Ws,
                                                                        unsigned* ovf) {
    const int Wo = 2 * Ws;
    const long HWo = 4L * Hs * Ws, HWs = (long)Hs * Ws;
    const int half = threadIdx.x >> 7, sl = threadIdx.x & 127;
    const bool live = (long)blockIdx.x * 128 + sl < HWs;
    const long sp = min((long)blockIdx.x * 128 + sl, HWs - 1);
    const int c16 = blockIdx.y, C16 = gridDim.y;
    const long n = blockIdx.z;
    const int k = (int)(sp / Ws), j = (int)(sp - (long)k * Ws);
    const int km = max(k - 1, 0), kp = min(k + 1, Hs - 1), jm = max(j - 1, 0), jp = min(j + 1, Ws - 1);
    // output (2k + dy, 2j + dx): rows (ya, yb) = dy ? (k, kp) : (km, k), weight of yb = dy ? 0.25 : 0.75 (1 where ya == yb); columns alike
    const float wyb0 = km == k ? 1.f : 0.75f, wyb1 = k == kp ? 1.f : 0.25f;
    const float wxb0 = jm == j ? 1.f : 0.75f, wxb1 = j == jp ? 1.f : 0.25f;
    const float wya0 = 1.f - wyb0, wya1 = 1.f - wyb1, wxa0 = 1.f - wxb0, wxa1 = 1.f - wxb1;
    const int r0 = km * Ws, r1 = k * Ws, r2 = kp * Ws;
    unsigned short t[4][8][TERMS];
    float gm = 0.f;                                        // range guard of the two-term format (split.h)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int c = c16 * 16 + half * 8 + q;
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        if (c < C) {
            const float* pa = a + (n * C + c) * HWs;
            float v[3][3] = {{pa[r0 + jm], pa[r0 + j], pa[r0 + jp]}, {pa[r1 + jm], pa[r1 + j], pa[r1 + jp]}, {pa[r2 + jm], pa[r2 + j], pa[r2 + jp]}};
            if (b) {
                const float* pb = b + (n * C + c) * HWs;
                v[0][0] += pb[r0 + jm]; v[0][1] += pb[r0 + j]; v[0][2] += pb[r0 + jp];
                v[1][0] += pb[r1 + jm]; v[1][1] += pb[r1 + j]; v[1][2] += pb[r1 + jp];
                v[2][0] += pb[r2 + jm]; v[2][1] += pb[r2 + j]; v[2][2] += pb[r2 + jp];
            }
            // o = wya (wxa vaa + wxb vab) + wyb (wxa vba + wxb vbb)
            o[0] = wya0 * (wxa0 * v[0][0] + wxb0 * v[0][1]) + wyb0 * (wxa0 * v[1][0] + wxb0 * v[1][1]);
            o[1] = wya0 * (wxa1 * v[0][1] + wxb1 * v[0][2]) + wyb0 * (wxa1 * v[1][1] + wxb1 * v[1][2]);
            o[2] = wya1 * (wxa0 * v[1][0] + wxb0 * v[1][1]) + wyb1 * (wxa0 * v[2][0] + wxb0 * v[2][1]);
            o[3] = wya1 * (wxa1 * v[1][1] + wxb1 * v[1][2]) + wyb1 * (wxa1 * v[2][1] + wxb1 * v[2][2]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (TERMS == 2) gm = sb_guard_max(gm, o[i]);
            sb_split_dev<TERMS>(o[i], t[i][q]);
        }
    }
    if (TERMS == 2) sb_guard_flush(gm, ovf);
    // The 16-byte pieces go out through LDS: written by their producers (a pixel's pieces of one half sit 32 bytes apart, the
    // pixels of consecutive lanes 128 bytes apart: stored directly, a wave's store instruction is 64 partial writes into 64
    // different lines), read back piece-major, so that a wave stores 1 KiB of consecutive bytes per instruction.
    constexpr int PB = 32 * TERMS, PITCH = PB + 16;       // bytes of a pixel's chunk image; its pitch in LDS (bank spread)
    __shared__ __align__(16) unsigned char stg[2 * 256 * (32 * TERMS + 16)];
    if (live) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned char* d = stg + ((i >> 1) * 256 + 2 * sl + (i & 1)) * PITCH + half * 16;
#pragma unroll
            for (int kk = 0; kk < TERMS; ++kk) {
                uint4 v;
                v.x = t[i][0][kk] | ((unsigned)t[i][1][kk] << 16);
                v.y = t[i][2][kk] | ((unsigned)t[i][3][kk] << 16);
                v.z = t[i][4][kk] | ((unsigned)t[i][5][kk] << 16);
                v.w = t[i][6][kk] | ((unsigned)t[i][7][kk] << 16);
                *reinterpret_cast<uint4*>(d + kk * 32) = v;
            }
        }
    }
    __syncthreads();
    constexpr int PPP = 2 * TERMS;                        // 16-byte pieces per pixel
    const long sp0 = (long)blockIdx.x * 128;
    const int npix = (int)min(128L, HWs - sp0);           // source pixels of this workgroup
    unsigned short* ob = out + ((n * C16 + c16) * HWo) * TERMS * 16;
    for (int it = threadIdx.x; it < 2 * 2 * npix * PPP; it += 256) {
        const int q = it % PPP, px = it / PPP;            // px = dy * (2 npix) + 2 sl + dx
        const int dy = px / (2 * npix), xx = px - dy * 2 * npix, sl2 = xx >> 1, dx = xx & 1;
        const long spx = sp0 + sl2;
        const int k2 = (int)(spx / Ws), j2 = (int)(spx - (long)k2 * Ws);
        const long p = (long)(2 * k2 + dy) * Wo + 2 * j2 + dx;
        const uint4 v = *reinterpret_cast<const uint4*>(stg + (dy * 256 + xx) * PITCH + q * 16);
        *reinterpret_cast<uint4*>(ob + p * TERMS * 16 + q * 8) = v;
    }
}
#ifndef BDE_UPS_QUAD
#define BDE_UPS_QUAD 1
#endif
static int upsample2x_sum_split(const float* a, const float* b, void* out_sb, int N, int C, int Hs, int Ws, int terms, unsigned* ovf,
                                hipStream_t s) {
    if (BDE_UPS_QUAD) {
        const dim3 gq((unsigned)cdivl((long)Hs * Ws, 128), cdiv(C, 16), (unsigned)N);
        if (terms == 2) hipLaunchKernelGGL(upsample2x_sum_split_quad_kernel<2>, gq, dim3(256), 0, s, a, b, (unsigned short*)out_sb, C, Hs, Ws, ovf);
        else hipLaunchKernelGGL(upsample2x_sum_split_quad_kernel<3>, gq, dim3(256), 0, s, a, b, (unsigned short*)out_sb, C, Hs, Ws, ovf);
        BDE_HIP(hipGetLastError());
        return BDE_OK;
    }
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

}  // namespace bde
