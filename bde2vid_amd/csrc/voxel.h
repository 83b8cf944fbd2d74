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

// ---------------------------------------------------------------------------------------------------------------
// Bucketed binning: independent of the tile count.  The tile kernel above makes every tile's workgroup test every event of
// its window (8 tiles per 180x240 grid, 50 per 480x640, 150 per 720x1280).  Here one pass moves each event ONCE: a workgroup
// takes a chunk of VB_CHUNK consecutive events of one window, computes per event the reference's normalised time stamp and
// its (tile, cell), and writes the chunk back sorted by tile as 8-byte records {cell | tile | polarity, t_norm} together with
// the chunk's run offsets per tile; a tile's workgroup then reads only its own runs (VB_CHUNK / tiles records each,
// contiguous) and accumulates in LDS as before.  29 B of traffic per event (13 read, 8 written, 8 read) whatever the tile
// count.  Per-event arithmetic as in the kernels above (t_norm is computed once, in the bucketing pass).
constexpr int VB_CHUNK = 4096;              // events per bucketing workgroup (256 threads x 16)
constexpr int VB_MAX_TILES = 1024;
#ifndef VB_TILE_LDS_BYTES
#define VB_TILE_LDS_BYTES (128 * 1024)
#endif
constexpr long VB_TILE_LDS = VB_TILE_LDS_BYTES;    // LDS of one tile workgroup on the bucketed path (64 KB = two workgroups per CU measured
                                            // slower at every resolution: the pass pays ~13 us per workgroup, not per byte)
constexpr unsigned VB_NAN_WINDOW = 0x40000000u;     // flag on the last run offset of a window's first chunk
struct VoxelRec { unsigned a; float tn; };  // a = cell (13 bits) | tile << 13 | polarity sign << 31

template <bool NATIVE>
__global__ __launch_bounds__(256) void voxel_bucket_kernel(const void* __restrict__ xs_, const void* __restrict__ ys_,
                                                           const void* __restrict__ ts_, const void* __restrict__ ps_,
                                                           const long* __restrict__ starts, const long* __restrict__ ends,
                                                           long n_single, long n_cols, int nb, int H, int W, int TH, int TW,
                                                           int ntw, int ntiles, int nchunks, VoxelRec* __restrict__ recs,
                                                           float* __restrict__ pvals, int* __restrict__ table,
                                                           int* __restrict__ oob) {
    __shared__ int cnt[VB_MAX_TILES + 1];
    __shared__ VoxelRec rec[VB_CHUNK];
    __shared__ float recp[NATIVE ? 1 : VB_CHUNK];      // float columns carry an arbitrary weight per event (recordings: +-1 = the sign bit)
    const int seg = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    long beg = starts ? starts[seg] : 0;
    long end = ends ? ends[seg] : n_single;
    if (n_cols >= 0) {
        beg = min(max(beg, 0L), n_cols);
        end = min(max(end, beg), n_cols);
    }
    int* tab = table + ((long)seg * nchunks + chunk) * (ntiles + 1);
    const bool live = NATIVE ? (end - beg >= 3) : (end - beg > 0);     // h5_dataset.py:219-220 for recordings
    // chunks start at the window's first event rounded DOWN to a multiple of four: a thread takes four groups of four
    // consecutive events, each group one 8-byte (int16 columns) / 16-byte load per column (single 2-byte loads per event ran
    // the pass at 2.4 TB/s); events in front of `beg` are masked
    const long c0 = (beg & ~3L) + (long)chunk * VB_CHUNK;
    if (!live || c0 >= end) {                                          // empty chunk: all runs empty
        for (int i = tid; i <= ntiles; i += 256) tab[i] = 0;
        return;
    }
    for (int i = tid; i <= ntiles; i += 256) cnt[i] = 0;
    __syncthreads();
    const float bm1 = (float)(nb - 1);
    double t0d = 0.0;
    float t0f = 0.f, dt;
    if (NATIVE) {
        const double* ts = (const double*)ts_;
        t0d = ts[beg];
        dt = (float)(ts[end - 1] - t0d) - 0.0f;                        // event_utils.py:489 on the shifted float32 column
    } else {
        const float* ts = (const float*)ts_;
        t0f = ts[beg];
        dt = ts[end - 1] - t0f;
    }
    // (columns handed over as views with an odd storage offset take single loads)
    const bool vec_ok = NATIVE ? (((reinterpret_cast<uintptr_t>(xs_) | reinterpret_cast<uintptr_t>(ys_)) & 7u) == 0 &&
                                  (reinterpret_cast<uintptr_t>(ts_) & 15u) == 0 && (reinterpret_cast<uintptr_t>(ps_) & 3u) == 0)
                               : (((reinterpret_cast<uintptr_t>(xs_) | reinterpret_cast<uintptr_t>(ys_) | reinterpret_cast<uintptr_t>(ts_) |
                                    reinterpret_cast<uintptr_t>(ps_)) & 15u) == 0);
    unsigned ra[16];
    float rt[16], rp[16];
    int rk[16];
    int n_oob = 0;
    long xq[4][4], yq[4][4];
    float tq[4][4], pq[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long i4 = c0 + (long)(k * 256 + tid) * 4;
        if (vec_ok && i4 + 3 < end) {
            if (NATIVE) {
                const short4 xv = *reinterpret_cast<const short4*>((const short*)xs_ + i4);
                const short4 yv = *reinterpret_cast<const short4*>((const short*)ys_ + i4);
                const double2 ta = *reinterpret_cast<const double2*>((const double*)ts_ + i4);
                const double2 tb = *reinterpret_cast<const double2*>((const double*)ts_ + i4 + 2);
                const uchar4 pv = *reinterpret_cast<const uchar4*>((const unsigned char*)ps_ + i4);
                xq[k][0] = xv.x; xq[k][1] = xv.y; xq[k][2] = xv.z; xq[k][3] = xv.w;
                yq[k][0] = yv.x; yq[k][1] = yv.y; yq[k][2] = yv.z; yq[k][3] = yv.w;
                const double td[4] = {ta.x, ta.y, tb.x, tb.y};
                const unsigned char pb[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    tq[k][u] = ((float)(td[u] - t0d) - 0.0f) / dt * bm1;
                    pq[k][u] = pb[u] ? 1.0f : -1.0f;
                }
            } else {
                const float4 xv = *reinterpret_cast<const float4*>((const float*)xs_ + i4);
                const float4 yv = *reinterpret_cast<const float4*>((const float*)ys_ + i4);
                const float4 tv = *reinterpret_cast<const float4*>((const float*)ts_ + i4);
                const float4 pv = *reinterpret_cast<const float4*>((const float*)ps_ + i4);
                xq[k][0] = (long)xv.x; xq[k][1] = (long)xv.y; xq[k][2] = (long)xv.z; xq[k][3] = (long)xv.w;   // Tensor.long() truncates
                yq[k][0] = (long)yv.x; yq[k][1] = (long)yv.y; yq[k][2] = (long)yv.z; yq[k][3] = (long)yv.w;
                const float tf[4] = {tv.x, tv.y, tv.z, tv.w}, pf[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    tq[k][u] = (tf[u] - t0f) / dt * bm1;               // :490 (division, then multiply, fp32)
                    pq[k][u] = pf[u];
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long i = max(min(i4 + u, end - 1), beg);         // (masked below when outside [beg, end))
                if (NATIVE) {
                    xq[k][u] = ((const short*)xs_)[i]; yq[k][u] = ((const short*)ys_)[i];
                    tq[k][u] = ((float)(((const double*)ts_)[i] - t0d) - 0.0f) / dt * bm1;
                    pq[k][u] = ((const unsigned char*)ps_)[i] ? 1.0f : -1.0f;
                } else {
                    xq[k][u] = (long)((const float*)xs_)[i]; yq[k][u] = (long)((const float*)ys_)[i];
                    tq[k][u] = (((const float*)ts_)[i] - t0f) / dt * bm1;
                    pq[k][u] = ((const float*)ps_)[i];
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = k * 4 + u;
            const long i = c0 + (long)(k * 256 + tid) * 4 + u;
            rk[q] = -1;
            if (i < beg || i >= end) continue;
            long x = xq[k][u], y = yq[k][u];
            if (x < 0) x += W;                                         // index_put_ wraps negative indices
            if (y < 0) y += H;
            if (x < 0 || x >= W || y < 0 || y >= H) { ++n_oob; continue; }   // the reference raises IndexError here
            const int ty = (int)y / TH, tx = (int)x / TW;
            const int tile = ty * ntw + tx;
            const int cell = ((int)y - ty * TH) * TW + ((int)x - tx * TW);
            const float p = pq[k][u];
            ra[q] = (unsigned)cell | ((unsigned)tile << 13) | (p < 0.f ? 0x80000000u : 0u);
            rt[q] = tq[k][u];
            rp[q] = p;
            rk[q] = atomicAdd(&cnt[tile], 1);
        }
    if (oob && n_oob) atomicAdd(oob, n_oob);
    __syncthreads();
    // exclusive scan of the per-tile counts (one wave; ntiles <= 1024 = 16 per lane)
    if (tid < 64) {
        const int per = (ntiles + 63) / 64;
        int sum = 0;
        for (int j = 0; j < per; ++j) { const int t = tid * per + j; if (t < ntiles) sum += cnt[t]; }
        int incl = sum;
#pragma unroll
        for (int sh = 1; sh < 64; sh <<= 1) { const int v = __shfl_up(incl, sh); if (tid >= sh) incl += v; }
        int run = incl - sum;
        for (int j = 0; j < per; ++j) {
            const int t = tid * per + j;
            if (t < ntiles) { const int c = cnt[t]; cnt[t] = run; run += c; }
        }
        if (tid == 63) cnt[ntiles] = incl;
    }
    __syncthreads();
    // (a zero-duration window: dt == 0, or NaN from a NaN time stamp, makes every weight of the window NaN)
    const unsigned wflag = (chunk == 0 && !(dt != 0.f && dt == dt)) ? VB_NAN_WINDOW : 0u;
    for (int i = tid; i <= ntiles; i += 256) tab[i] = cnt[i] | (i == ntiles ? (int)wflag : 0);
#pragma unroll
    for (int k = 0; k < 16; ++k)
        if (rk[k] >= 0) {
            const int pos = cnt[(ra[k] >> 13) & 0x3ffffu] + rk[k];
            rec[pos] = VoxelRec{ra[k], rt[k]};
            if (!NATIVE) recp[pos] = rp[k];
        }
    __syncthreads();
    const int total = cnt[ntiles];
    const long obase = ((long)seg * nchunks + chunk) * VB_CHUNK;
    for (int i = tid; i < total; i += 256) {
        recs[obase + i] = rec[i];
        if (!NATIVE) pvals[obase + i] = recp[i];
    }
}

// FIXED: the tile accumulates in 64-bit fixed point (32.32) with ds_add_u64 instead of ds_add_f32.  Measured on MI355X
// (tools/ubench/lds_atomic_rate.hip): LDS float atomic adds retire 0.37 lanes per cycle and CU (200 G adds/s chip-wide), u32 / u64
// adds >= 3 lanes per cycle (as fast as plain LDS stores) -- the float form was the whole cost of this pass (24 M events, two
// taps each: 270 us).  A recording's weights are +-max(0, 1 - |t_norm - b|) with |v| <= 1: each is converted exactly up to
// 2^-33, the sum of a pixel is exact in the 64-bit accumulator and rounded to fp32 ONCE -- closer to the real-number sum than
// any order of float additions, and the same bits run to run.  A window of zero duration (dt == 0: every weight NaN in the
// reference, event_utils.py:489-495) is flagged by the bucketing pass; its tile counts events per pixel and stores NaN for
// all bins of a pixel that received one.  The float-column entry points keep float accumulation (arbitrary weights).
template <bool FIXED>
__global__ __launch_bounds__(1024) void voxel_tile_from_buckets_kernel(const VoxelRec* __restrict__ recs, const float* __restrict__ pvals,
                                                                        const int* __restrict__ table, int nb, int H, int W, int TH,
                                                                        int TW, int ntw, int ntiles, int nchunks,
                                                                        float* __restrict__ grids) {
    extern __shared__ __align__(8) unsigned char tile_raw[];          // [nb][TH][TW] of float | int64
    float* tile = reinterpret_cast<float*>(tile_raw);
    unsigned long long* tile64 = reinterpret_cast<unsigned long long*>(tile_raw);
    const int tix = blockIdx.x, seg = blockIdx.y;
    const int ty0 = (tix / ntw) * TH, tx0 = (tix % ntw) * TW;
    const int th = min(TH, H - ty0), tw = min(TW, W - tx0);
    const int tpx = TH * TW;
    for (int i = threadIdx.x; i < nb * tpx * (FIXED ? 2 : 1); i += blockDim.x) tile[i] = 0.f;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool nan_window = FIXED && (table[((long)seg * nchunks) * (ntiles + 1) + ntiles] & VB_NAN_WINDOW) != 0;
    auto add_event = [&](unsigned a, float t, float p) {
        const int c0 = a & 0x1fffu;
        if (FIXED) {
            if (nan_window) { atomicAdd(tile64 + c0, 1ull); return; }
            if (!(t == t)) {
                // a NaN time stamp inside an ordinary window: max(0, 1 - |NaN - b|) is NaN in every bin of that pixel (:494-495,
                // torch.max hands the NaN on), as on the streaming and the float paths.  Fixed point has no NaN: bits 62, 61 of
                // the sum are forced to 0, 1 -- a pattern no sum of fewer than 2^28 weights reaches and no later add removes.
                for (int b = 0; b < nb; ++b) {
                    atomicOr(tile64 + c0 + b * tpx, 1ull << 61);
                    atomicAnd(tile64 + c0 + b * tpx, ~(1ull << 62));
                }
                return;
            }
        } else if (!(t == t)) {                                        // dt == 0 -> NaN weights in every bin (:494-495)
            for (int b = 0; b < nb; ++b) atomicAdd(tile + c0 + b * tpx, p * t);
            return;
        }
        const int b0 = (int)floorf(t);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int b = b0 + k;
            if (b < 0 || b >= nb) continue;
            const float w = fmaxf(0.f, 1.0f - fabsf(t - (float)b));    // :494
            const float v = p * w;                                      // :495
            if (v == 0.f) continue;
            if (FIXED) atomicAdd(tile64 + c0 + b * tpx, (unsigned long long)__float2ll_rn(v * 4294967296.0f));
            else atomicAdd(tile + c0 + b * tpx, v);
        }
    };
    // The tile's records are the concatenation of its runs in the window's chunks.  Per batch of 1024 chunks: every thread
    // fetches one chunk's run bounds, a block scan turns the run lengths into positions, then thread t takes records t,
    // t + 1024, ... of the concatenation (its chunk by bisection in LDS): two dependent memory round trips per batch instead of
    // two per chunk -- at 300 tiles per grid a run is a dozen records and walking the chunks one after another was the whole
    // cost of the pass.
    __shared__ int spre[1025], slo[1024], swsum[16];
    for (int cb = 0; cb < nchunks; cb += 1024) {
        const int c = cb + threadIdx.x;
        int lo = 0, len = 0;
        if (c < nchunks) {
            const int* tab = table + ((long)seg * nchunks + c) * (ntiles + 1);
            lo = tab[tix] & 0xffff;
            len = (tab[tix + 1] & 0xffff) - lo;
        }
        int incl = len;
#pragma unroll
        for (int sh = 1; sh < 64; sh <<= 1) { const int v = __shfl_up(incl, sh); if (lane >= sh) incl += v; }
        if (lane == 63) swsum[wave] = incl;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < wave; ++w) base += swsum[w];
        spre[threadIdx.x] = base + incl - len;
        slo[threadIdx.x] = lo;
        if (threadIdx.x == 1023) spre[1024] = base + incl;
        __syncthreads();
        const int total = spre[1024];
        const long rb0 = ((long)seg * nchunks + cb) * VB_CHUNK;
        for (int v0 = threadIdx.x; v0 < total; v0 += 4 * 1024) {
            VoxelRec e[4];
            float pv[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int v = v0 + u * 1024;
                ok[u] = v < total;
                const int vv = ok[u] ? v : 0;
                int k = 0;                                              // largest k with spre[k] <= vv
#pragma unroll
                for (int step = 512; step >= 1; step >>= 1)
                    if (spre[k + step] <= vv) k += step;
                const long j = rb0 + (long)k * VB_CHUNK + slo[k] + (vv - spre[k]);
                e[u] = recs[j];
                pv[u] = (!FIXED && pvals) ? pvals[j] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (ok[u]) add_event(e[u].a, e[u].tn, (!FIXED && pvals) ? pv[u] : ((e[u].a & 0x80000000u) ? -1.0f : 1.0f));
        }
        __syncthreads();
    }
    __syncthreads();
    float* grid = grids + (long)seg * nb * H * W;
    for (int i = threadIdx.x; i < nb * th * tw; i += blockDim.x) {
        const int b = i / (th * tw), r = i - b * th * tw;
        const int ly = r / tw, lx = r - ly * tw;
        float v;
        if (FIXED) {
            if (nan_window) v = tile64[ly * TW + lx] ? __builtin_nanf("") : 0.f;
            else {
                const long long q = (long long)tile64[(b * TH + ly) * TW + lx];
                v = (q >= (1ll << 60) || q <= -(1ll << 60)) ? __builtin_nanf("")                            // marked by a NaN time stamp
                                                            : (float)((double)q * 2.3283064365386963e-10);   // exact product, one rounding
            }
        } else {
            v = tile[(b * TH + ly) * TW + lx];
        }
        grid[((long)b * H + ty0 + ly) * W + tx0 + lx] = v;
    }
}

// Scratch of the bucketed path (records + run tables): STREAM-ORDERED -- allocated on the caller's stream in front of the two
// launches of a call and freed behind them (hipMallocAsync / hipFreeAsync), so calls on different streams or from different host
// threads never share a buffer and nothing is freed under a launch that still reads it.  The device's default pool keeps freed
// blocks (release threshold raised once per device): after the first call of a size an allocation is a pool hit.  A call whose
// scratch would pass VB_SCRATCH_CAP, or whose allocation fails, gets *out = nullptr and takes the streaming kernel instead.
constexpr size_t VB_SCRATCH_CAP = 8ull << 30;
inline int voxel_scratch_acquire(size_t need, hipStream_t stream, void** out) {
    static unsigned char pool_ready[BDE_MAX_DEVICES];
    static std::mutex mu;
    *out = nullptr;
    if (need > VB_SCRATCH_CAP) return BDE_OK;
    int d = 0;
    BDE_HIP(hipGetDevice(&d));
    if (d < 0 || d >= BDE_MAX_DEVICES) return fail(BDE_ERR_ARG, "device %d", d);
    if (!__atomic_load_n(&pool_ready[d], __ATOMIC_ACQUIRE)) {
        std::lock_guard<std::mutex> lock(mu);
        if (!pool_ready[d]) {
            hipMemPool_t pool = nullptr;
            uint64_t keep = ~0ull;
            if (hipDeviceGetDefaultMemPool(&pool, d) == hipSuccess && pool)
                (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
            (void)hipGetLastError();
            __atomic_store_n(&pool_ready[d], (unsigned char)1, __ATOMIC_RELEASE);
        }
    }
    if (hipMallocAsync(out, need, stream) != hipSuccess) {
        (void)hipGetLastError();                        // out of memory: the caller streams instead
        *out = nullptr;
    }
    return BDE_OK;
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

// tile geometry shared by the two tile-privatised paths
static inline int voxel_tile_geometry(int nb, int H, int W, int* TH_, int* TW_, int* ntw_, int* nth_, int cell_bytes = 4,
                                      long lds_budget = 128 * 1024) {
    const long cap = std::min<long>(lds_budget / ((long)cell_bytes * nb), 8192);   // pixels of one tile: all bins within the LDS budget
    if (cap < 64) return fail(BDE_ERR_UNSUPPORTED, "voxel grid with %d bins does not fit the LDS tile", nb);
    int ntw = cdiv(W, 128);
    int TW = cdiv(W, ntw);
    while ((long)TW > cap) { ++ntw; TW = cdiv(W, ntw); }
    int TH = (int)std::min<long>(H, cap / TW);
    const int nth = cdiv(H, TH);
    TH = cdiv(H, nth);
    *TH_ = TH; *TW_ = TW; *ntw_ = ntw; *nth_ = nth;
    return BDE_OK;
}

// Bucketed path.  max_win = an upper bound of the events of one window (sizes the chunk grid and the scratch).
template <bool NATIVE>
static inline int voxel_bucket_launch(const void* xs, const void* ys, const void* ts, const void* ps, const long* starts,
                                      const long* ends, long n_single, int nseg, long max_win, int nb, int H, int W, float* grids,
                                      int* oob, hipStream_t stream, long n_cols = -1) {
    if (oob) BDE_HIP(hipMemsetAsync(oob, 0, sizeof(int), stream));
    if (nseg <= 0) return BDE_OK;
    constexpr int CELL = NATIVE ? 8 : 4;                // recordings accumulate in 64-bit fixed point
    int TH, TW, ntw, nth;
    BDE_TRY(voxel_tile_geometry(nb, H, W, &TH, &TW, &ntw, &nth, CELL, VB_TILE_LDS));
    const int ntiles = nth * ntw;
    if (ntiles > VB_MAX_TILES || TH * TW > 8192) return fail(BDE_ERR_UNSUPPORTED, "bucketed binning: %d tiles of %d pixels", ntiles, TH * TW);
    const int nchunks = (int)std::max<long>(1, cdivl(std::max<long>(max_win, 0) + 3, VB_CHUNK));   // (+ 3: chunks start at a multiple of four)
    const size_t lds = (size_t)CELL * nb * TH * TW;
    static unsigned char raised[BDE_MAX_DEVICES];
    BDE_HIP(raise_dynamic_lds(raised, (const void*)voxel_tile_from_buckets_kernel<NATIVE>, 144 * 1024));   // (+ 8.3 KB of static LDS)
    // one scratch for the call, sized for its largest batch of windows (batches follow each other on the stream)
    auto sizes = [&](int nw, size_t* rec_bytes, size_t* p_bytes, size_t* tab_bytes) {
        const size_t nrec = (size_t)nw * nchunks * VB_CHUNK;
        *rec_bytes = nrec * sizeof(VoxelRec);
        *p_bytes = NATIVE ? 0 : nrec * sizeof(float);
        *tab_bytes = (size_t)nw * nchunks * (ntiles + 1) * sizeof(int);
    };
    size_t rb, pb, tb;
    sizes(std::min(65535, nseg), &rb, &pb, &tb);
    void* scratch = nullptr;
    BDE_TRY(voxel_scratch_acquire(rb + pb + tb + 512, stream, &scratch));
    if (!scratch)                                       // too large or no memory: the streaming kernel needs no scratch
        return voxel_tile_launch<NATIVE>(xs, ys, ts, ps, starts, ends, n_single, nseg, nb, H, W, grids, oob, stream, n_cols);
    for (int w0 = 0; w0 < nseg; w0 += 65535) {
        const int nw = std::min(65535, nseg - w0);
        size_t rec_bytes, p_bytes, tab_bytes;
        sizes(nw, &rec_bytes, &p_bytes, &tab_bytes);
        VoxelRec* recs = (VoxelRec*)scratch;
        float* pvals = NATIVE ? nullptr : (float*)((char*)scratch + rec_bytes);
        int* table = (int*)((char*)scratch + ((rec_bytes + p_bytes + 255) / 256) * 256);
        hipLaunchKernelGGL(voxel_bucket_kernel<NATIVE>, dim3((unsigned)nchunks, (unsigned)nw), dim3(256), 0, stream, xs, ys, ts, ps,
                           starts ? starts + w0 : nullptr, ends ? ends + w0 : nullptr, n_single, n_cols, nb, H, W, TH, TW, ntw, ntiles,
                           nchunks, recs, pvals, table, oob);
        hipLaunchKernelGGL(voxel_tile_from_buckets_kernel<NATIVE>, dim3((unsigned)ntiles, (unsigned)nw), dim3(1024), lds, stream, recs, pvals, table,
                           nb, H, W, TH, TW, ntw, ntiles, nchunks, grids + (long)w0 * nb * H * W);
    }
    const hipError_t le = hipGetLastError();
    (void)hipFreeAsync(scratch, stream);
    BDE_HIP(le);
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
