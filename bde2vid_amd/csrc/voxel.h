// Event -> voxel-grid binning.
//
// Restates events_to_voxel_torch (events_contrast_maximization/utils/event_utils.py:466-509) and the
// nearest-pixel branch of events_to_image_torch (:360,371-375).  The reference makes B full passes
// over the N events (one index_put_(accumulate) per bin); the temporal-bilinear weight
// max(0, 1-|t_norm-b|) is non-zero for at most two bins, so one pass with two adds per event
// produces the same grid.  Per-event weights are computed with the reference's exact fp32
// expression, so they are bit-identical; only the summation ORDER inside a pixel differs
// (atomic adds), which is why parity for these kernels is 1e-5*max(1,count) instead of bit-exact.
//
// Three kernels, selected by bde_voxel_method:
//   0  voxel_tile_kernel    (default) a workgroup owns a pixel tile of one window's grid in LDS (all bins), streams the
//                           window's events and accumulates with LDS atomics; plain coalesced stores, no zero-fill pass.
//                           Every tile of a window reads the window's events (tiles x 13 B per event, mostly L2 hits).
//   2  voxel_bucket_*       counting + bucketing pass that moves every event ONCE into the segment of its
//                           (window, tile); the tile workgroups then read only their own events.  Independent of the
//                           tile count (480x640: 50 tiles per grid).
//   1  voxel_scatter_*      one global float atomic per tap: sits on the scattered float-atomic roof of MI355X
//                           (~0.08 TB/s when the 64 lanes of a wave hit 64 rows); kept for A/B timing.
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include "common.h"

namespace bde {

__global__ __launch_bounds__(256) void voxel_scatter_kernel(const float* __restrict__ xs,
                                                            const float* __restrict__ ys,
                                                            const float* __restrict__ ts,
                                                            const float* __restrict__ ps,
                                                            const long* __restrict__ offsets, long n_single,
                                                            int nb, int H, int W, float* __restrict__ grids,
                                                            int* __restrict__ oob) {
    const int seg = blockIdx.y;
    const long beg = offsets ? offsets[seg] : 0;
    const long end = offsets ? offsets[seg + 1] : n_single;
    const long n = end - beg;
    if (n <= 0) return;
    float* grid = grids + (long)seg * nb * H * W;
    const float t0 = ts[beg];
    const float dt = ts[end - 1] - t0;                 // event_utils.py:489
    const float bm1 = (float)(nb - 1);
    const long HW = (long)H * W;
    for (long i = beg + blockIdx.x * (long)blockDim.x + threadIdx.x; i < end; i += (long)gridDim.x * blockDim.x) {
        const float tn = (ts[i] - t0) / dt * bm1;      // :490  (division, then multiply, fp32)
        long xi = (long)xs[i], yi = (long)ys[i];       // :371-374  Tensor.long() truncates toward zero
        if (xi < 0) xi += W;                           // index_put_ wraps negative indices
        if (yi < 0) yi += H;
        if (xi < 0 || xi >= W || yi < 0 || yi >= H) {  // the reference raises IndexError here
            if (oob) atomicAdd(oob, 1);
            continue;
        }
        const float p = ps[i];
        float* cell = grid + yi * W + xi;
        if (!(tn == tn)) {                             // dt == 0 -> NaN weights in every bin (:494-495)
            for (int b = 0; b < nb; ++b) atomicAdd(cell + b * HW, p * tn);
            continue;
        }
        const int b0 = (int)floorf(tn);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int b = b0 + k;
            if (b < 0 || b >= nb) continue;
            const float w = fmaxf(0.f, 1.0f - fabsf(tn - (float)b));   // :494
            const float v = p * w;                                      // :495
            if (v != 0.f) atomicAdd(cell + b * HW, v);
        }
    }
}

// The same binning straight from the recording's native event columns (Monash HDF5 schema read by
// DynamicH5Dataset.get_events, data_loader/h5_dataset.py:410-415: xs, ys int16, ts float64 seconds, ps bool),
// one grid per between-frames window [offsets[w], offsets[w+1]) of the event stream -- the slicing of
// BaseVoxelDataset.__getitem__ (:213-226) done on the device: 13 B read per event instead of the 16 B of four
// float32 columns plus the host-side casts.  Restated per window:
//   fewer than 3 events -> all-zero grid                                   (:219-220)
//   ts -> float32(ts - ts[first])  (subtraction in float64, then the cast)  (:224)
//   ps -> +1 / -1                  (ps * 2.0 - 1.0, :414; float32 cast :225)
//   then events_to_voxel_torch on the float32 columns                       (:357)
__global__ __launch_bounds__(256) void voxel_scatter_native_kernel(const short* __restrict__ xs, const short* __restrict__ ys,
                                                                   const double* __restrict__ ts,
                                                                   const unsigned char* __restrict__ ps,
                                                                   const long* __restrict__ offsets, int nb, int H, int W,
                                                                   float* __restrict__ grids, int* __restrict__ oob) {
    const int seg = blockIdx.y;
    const long beg = offsets[seg], end = offsets[seg + 1];
    if (end - beg < 3) return;
    float* grid = grids + (long)seg * nb * H * W;
    const double t0d = ts[beg];
    const float dt = (float)(ts[end - 1] - t0d) - 0.0f;      // event_utils.py:489 on the shifted float32 column
    const float bm1 = (float)(nb - 1);
    const long HW = (long)H * W;
    for (long i = beg + blockIdx.x * (long)blockDim.x + threadIdx.x; i < end; i += (long)gridDim.x * blockDim.x) {
        const float tn = ((float)(ts[i] - t0d) - 0.0f) / dt * bm1;
        long xi = (long)xs[i], yi = (long)ys[i];
        if (xi < 0) xi += W;                               // index_put_ wraps negative indices
        if (yi < 0) yi += H;
        if (xi < 0 || xi >= W || yi < 0 || yi >= H) {      // the reference raises IndexError here
            if (oob) atomicAdd(oob, 1);
            continue;
        }
        const float p = ps[i] ? 1.0f : -1.0f;
        float* cell = grid + yi * W + xi;
        if (!(tn == tn)) {                                 // dt == 0 -> NaN weights in every bin
            for (int b = 0; b < nb; ++b) atomicAdd(cell + b * HW, p * tn);
            continue;
        }
        const int b0 = (int)floorf(tn);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int b = b0 + k;
            if (b < 0 || b >= nb) continue;
            const float w = fmaxf(0.f, 1.0f - fabsf(tn - (float)b));
            const float v = p * w;
            if (v != 0.f) atomicAdd(cell + b * HW, v);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Tile-privatised binning.  Scattered float atomics run at the memory side at ~0.08 TB/s when the 64 lanes of a
// wave hit 64 different rows (MI355X_MICROARCH.md, Global float atomics) -- which is what events do -- and the scatter
// kernels above sit exactly on that roof (24 M events: 81 GB/s of added bytes, 32 B of fabric writes per atomic).
// Here a workgroup owns a TH x TW pixel tile of ONE window's grid (all bins, <= 128 KB of LDS), streams the window's
// events, keeps the ones inside its tile (LDS float atomics) and writes the finished tile with plain coalesced stores:
// no zero-fill pass, no global atomics; the price is that a window's events are read once per tile (13 B x tiles,
// mostly from L2 / Infinity Cache, the tiles of a window run at the same time).
// Windows are [starts[w], ends[w]) and may overlap (the k_events / t_seconds voxel methods with a sliding window,
// data_loader/h5_dataset.py:277-302).  Arithmetic per event exactly as in the kernels above.
template <bool NATIVE>
__global__ __launch_bounds__(1024) void voxel_tile_kernel(const void* __restrict__ xs_, const void* __restrict__ ys_,
                                                          const void* __restrict__ ts_, const void* __restrict__ ps_,
                                                          const long* __restrict__ starts, const long* __restrict__ ends,
                                                          long n_single, long n_cols, int nb, int H, int W, int TH, int TW, int ntw,
                                                          float* __restrict__ grids, int* __restrict__ oob) {
    extern __shared__ float tile[];                    // [nb][TH][TW]
    // The tiles of a window stream the same events: they go to ONE XCD (the dispatcher deals consecutive workgroups round-robin
    // to the 8 XCDs, so in grid order tile k of every window would land on XCD k and each L2 would pull every event itself).
    int tix, seg;
    {
        const unsigned gx = gridDim.x, total = gx * gridDim.y, lin = blockIdx.x + gx * blockIdx.y;
        const unsigned xcd = lin & 7u, q = total >> 3, rem = total & 7u;
        const unsigned L = xcd * q + min(xcd, rem) + (lin >> 3);
        tix = (int)(L % gx);
        seg = (int)(L / gx);
    }
    long beg = starts ? starts[seg] : 0;
    long end = ends ? ends[seg] : n_single;
    if (n_cols >= 0) {                                 // window bounds past the uploaded columns (a file whose attributes
        beg = min(max(beg, 0L), n_cols);               // disagree with its datasets): clamp, as an h5py slice does
        end = min(max(end, beg), n_cols);
    }
    const int ty0 = (tix / ntw) * TH, tx0 = (tix % ntw) * TW;
    const int th = min(TH, H - ty0), tw = min(TW, W - tx0);
    const int tpx = TH * TW;
    for (int i = threadIdx.x; i < nb * tpx; i += blockDim.x) tile[i] = 0.f;
    __syncthreads();
    const bool live = NATIVE ? (end - beg >= 3) : (end - beg > 0);     // h5_dataset.py:219-220 for recordings
    if (live) {
        const float bm1 = (float)(nb - 1);
        double t0d = 0.0;
        float t0f = 0.f, dt;
        if (NATIVE) {
            const double* ts = (const double*)ts_;
            t0d = ts[beg];
            dt = (float)(ts[end - 1] - t0d) - 0.0f;                    // event_utils.py:489 on the shifted float32 column
        } else {
            const float* ts = (const float*)ts_;
            t0f = ts[beg];
            dt = ts[end - 1] - t0f;
        }
        int n_oob = 0;
        // One event = its coordinates tested against the tile, then (one event in ntiles) time stamp, polarity and two LDS
        // atomics.  A thread takes FOUR consecutive events per iteration, their coordinates as one 8-byte (int16 columns) or
        // 16-byte (float columns) load each: with one event per iteration the loop was a chain of dependent round trips
        // (24 M events: 366 iterations x ~1.6 us per workgroup).  The walk starts at beg rounded down to a multiple of four
        // ; the last group of a window falls back to single loads.
        // Phase 1 per group: tile test of the four events; phase 2: time stamps and polarities of the accepted ones, fetched by
        // ALL lanes in one go (a rejected event re-reads the window's first element: a broadcast hit) -- as four branches they
        // were four more serialized round trips per iteration, since some lane of a wave accepts each of the four.
        auto load_xy = [&](long i4, long (&xi)[4], long (&yi)[4], bool vec_ok) {
            if (vec_ok && i4 + 3 < end) {
                if (NATIVE) {
                    const short4 xv = *reinterpret_cast<const short4*>((const short*)xs_ + i4);
                    const short4 yv = *reinterpret_cast<const short4*>((const short*)ys_ + i4);
                    xi[0] = xv.x; xi[1] = xv.y; xi[2] = xv.z; xi[3] = xv.w;
                    yi[0] = yv.x; yi[1] = yv.y; yi[2] = yv.z; yi[3] = yv.w;
                } else {
                    const float4 xv = *reinterpret_cast<const float4*>((const float*)xs_ + i4);
                    const float4 yv = *reinterpret_cast<const float4*>((const float*)ys_ + i4);
                    xi[0] = (long)xv.x; xi[1] = (long)xv.y; xi[2] = (long)xv.z; xi[3] = (long)xv.w;   // Tensor.long() truncates
                    yi[0] = (long)yv.x; yi[1] = (long)yv.y; yi[2] = (long)yv.z; yi[3] = (long)yv.w;
                }
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const long i = max(min(i4 + u, end - 1), beg);
                    if (NATIVE) { xi[u] = ((const short*)xs_)[i]; yi[u] = ((const short*)ys_)[i]; }
                    else { xi[u] = (long)((const float*)xs_)[i]; yi[u] = (long)((const float*)ys_)[i]; }
                }
            }
        };
        // (a column handed over as a view with an odd storage offset takes the single loads throughout)
        const bool vec_ok = ((reinterpret_cast<uintptr_t>(xs_) | reinterpret_cast<uintptr_t>(ys_)) & (NATIVE ? 7u : 15u)) == 0;
        const long stride = 4L * blockDim.x;
        long i4 = (beg & ~3L) + 4L * threadIdx.x;
        long xn[4], yn[4];
        if (i4 < end) load_xy(i4, xn, yn, vec_ok);
        for (; i4 < end; i4 += stride) {
            long xi[4], yi[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { xi[u] = xn[u]; yi[u] = yn[u]; }
            if (i4 + stride < end) load_xy(i4 + stride, xn, yn, vec_ok);        // next group in flight during this one
            bool take[4];
            int cellofs[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                long x = xi[u], y = yi[u];
                const bool mine = i4 + u >= beg && i4 + u < end;
                if (x < 0) x += W;                                     // index_put_ wraps negative indices
                if (y < 0) y += H;
                const bool inside = x >= 0 && x < W && y >= 0 && y < H;
                if (mine && !inside) n_oob += (tix == 0);              // the reference raises IndexError here
                const int lx = (int)x - tx0, ly = (int)y - ty0;
                take[u] = mine && inside && lx >= 0 && lx < tw && ly >= 0 && ly < th;   // else another tile's event
                cellofs[u] = ly * TW + lx;
            }
            float tn[4], pp[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long i = take[u] ? i4 + u : beg;
                if (NATIVE) {
                    tn[u] = ((float)(((const double*)ts_)[i] - t0d) - 0.0f) / dt * bm1;
                    pp[u] = ((const unsigned char*)ps_)[i] ? 1.0f : -1.0f;
                } else {
                    tn[u] = (((const float*)ts_)[i] - t0f) / dt * bm1;  // :490 (division, then multiply, fp32)
                    pp[u] = ((const float*)ps_)[i];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (!take[u]) continue;
                float* cell = tile + cellofs[u];
                const float t = tn[u], p = pp[u];
                if (!(t == t)) {                                       // dt == 0 -> NaN weights in every bin (:494-495)
                    for (int b = 0; b < nb; ++b) atomicAdd(cell + b * tpx, p * t);
                    continue;
                }
                const int b0 = (int)floorf(t);
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int b = b0 + k;
                    if (b < 0 || b >= nb) continue;
                    const float w = fmaxf(0.f, 1.0f - fabsf(t - (float)b));    // :494
                    const float v = p * w;                                      // :495
                    if (v != 0.f) atomicAdd(cell + b * tpx, v);
                }
            }
        }
        if (oob && n_oob) atomicAdd(oob, n_oob);
    }
    __syncthreads();
    float* grid = grids + (long)seg * nb * H * W;
    for (int i = threadIdx.x; i < nb * th * tw; i += blockDim.x) {
        const int b = i / (th * tw), r = i - b * th * tw;
        const int ly = r / tw, lx = r - ly * tw;
        grid[((long)b * H + ty0 + ly) * W + tx0 + lx] = tile[(b * TH + ly) * TW + lx];
    }
}

// find_ts_index of DynamicH5Dataset (data_loader/h5_dataset.py:444-446) = binary_search_h5_dset (event_utils.py:10-28,
// side='left') on the events/ts column: the SAME bisection per query -- on an exact hit it returns the index the
// bisection lands on (not necessarily the first of equal timestamps), otherwise the insertion point.
__global__ __launch_bounds__(256) void find_ts_index_kernel(const double* __restrict__ ts, long n, const double* __restrict__ q,
                                                            int nq, long* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const double x = q[i];
    long l = 0, r = n - 1;
    while (l <= r) {
        const long mid = l + (r - l) / 2;
        const double v = ts[mid];
        if (v == x) { l = mid; break; }
        if (v < x) l = mid + 1;
        else r = mid - 1;
    }
    out[i] = l;
}

template <bool NATIVE>
static inline int voxel_tile_launch(const void* xs, const void* ys, const void* ts, const void* ps, const long* starts,
                                    const long* ends, long n_single, int nseg, int nb, int H, int W, float* grids, int* oob,
                                    hipStream_t stream, long n_cols = -1) {
    if (oob) BDE_HIP(hipMemsetAsync(oob, 0, sizeof(int), stream));
    if (nseg <= 0) return BDE_OK;
    const long cap = (128 * 1024) / (4L * nb);          // pixels of one tile: all bins in <= 128 KB of LDS
    if (cap < 64) return fail(BDE_ERR_UNSUPPORTED, "voxel grid with %d bins does not fit the LDS tile", nb);
    int ntw = cdiv(W, 128);
    int TW = cdiv(W, ntw);
    while ((long)TW > cap) { ++ntw; TW = cdiv(W, ntw); }
    int TH = (int)std::min<long>(H, cap / TW);
    const int nth = cdiv(H, TH);
    TH = cdiv(H, nth);
    const size_t lds = sizeof(float) * (size_t)nb * TH * TW;
    static unsigned char raised[BDE_MAX_DEVICES];
    BDE_HIP(raise_dynamic_lds(raised, (const void*)voxel_tile_kernel<NATIVE>));
    // windows ride in grid.y (<= 65535): long recordings go in batches
    for (int w0 = 0; w0 < nseg; w0 += 65535) {
        const int nw = std::min(65535, nseg - w0);
        hipLaunchKernelGGL(voxel_tile_kernel<NATIVE>, dim3((unsigned)(nth * ntw), (unsigned)nw), dim3(1024), lds, stream, xs, ys, ts, ps,
                           starts ? starts + w0 : nullptr, ends ? ends + w0 : nullptr, n_single, n_cols, nb, H, W, TH, TW, ntw,
                           grids + (long)w0 * nb * H * W, oob);
    }
    BDE_HIP(hipGetLastError());
    return BDE_OK;
}

static inline int voxel_native_launch(const short* xs, const short* ys, const double* ts, const unsigned char* ps,
                                      const long* offsets, int nseg, long n_per_seg_max, int nb, int H, int W, float* grids,
                                      int* oob, hipStream_t stream) {
    BDE_HIP(hipMemsetAsync(grids, 0, sizeof(float) * (size_t)nseg * nb * H * W, stream));
    if (oob) BDE_HIP(hipMemsetAsync(oob, 0, sizeof(int), stream));
    if (n_per_seg_max <= 0 || nseg <= 0) return BDE_OK;
    long blocks = cdivl(n_per_seg_max, 256);
    if (blocks > 2048) blocks = 2048;                  // grid-stride the rest
    hipLaunchKernelGGL(voxel_scatter_native_kernel, dim3((unsigned)blocks, (unsigned)nseg), dim3(256), 0, stream, xs, ys, ts, ps,
                       offsets, nb, H, W, grids, oob);
    BDE_HIP(hipGetLastError());
    return BDE_OK;
}

static inline int voxel_launch(const float* xs, const float* ys, const float* ts, const float* ps,
                               const long* offsets, int nseg, long n_per_seg_max, int nb, int H, int W,
                               float* grids, int* oob, hipStream_t stream) {
    BDE_HIP(hipMemsetAsync(grids, 0, sizeof(float) * (size_t)nseg * nb * H * W, stream));
    if (oob) BDE_HIP(hipMemsetAsync(oob, 0, sizeof(int), stream));
    if (n_per_seg_max <= 0) return BDE_OK;
    long blocks = cdivl(n_per_seg_max, 256);
    if (blocks > 2048) blocks = 2048;                  // grid-stride the rest
    dim3 grid((unsigned)blocks, (unsigned)nseg);
    hipLaunchKernelGGL(voxel_scatter_kernel, grid, dim3(256), 0, stream, xs, ys, ts, ps, offsets, n_per_seg_max,
                       nb, H, W, grids, oob);
    BDE_HIP(hipGetLastError());
    return BDE_OK;
}

}  // namespace bde
