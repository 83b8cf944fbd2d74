/*
 * libbde2vid -- C ABI of the MI355X-native BDE2VID inference path.
 *
 * This is the drop-in boundary of SURVEY.md §8(b).  Every pointer is a raw pointer (device pointers
 * come from tensor.data_ptr()); no torch types, no C++ types, no exceptions cross it.  Functions
 * return 0 on success and a negative status otherwise; bde_last_error() then describes the failure.
 * `stream` is a hipStream_t passed as void* (NULL = the null stream).
 *
 * Reference interfaces replaced (paths relative to the reference repository):
 *   bde_create / bde_load_weight / bde_finalize_weights
 *       <- MODELS.build(cfg.model) + model.load_state_dict(sd)   eval_models_seq.py:52-60,86
 *          (module tree of model/BDE2VID/bde2vid.py:14-20 and
 *           model/BDE2VID/bde2vid_cross_scale_propogation_V5.py:19-98)
 *   bde_forward
 *       <- BDE2VID.forward(inputs, mode='tensor')               model/BDE2VID/bde2vid.py:30-50
 *          == BDE2VIDCrossscalePropogationV5.forward            ..._V5.py:100-241
 *   bde_voxelize / bde_voxelize_batch / bde_voxelize_events
 *       <- events_to_voxel_torch                                events_contrast_maximization/utils/event_utils.py:466-509
 *          (+ events_to_image_torch nearest branch, :330-376), called from data_loader/h5_dataset.py:357;
 *          bde_voxelize_events also covers the caller's slicing and casts, h5_dataset.py:213-226,410-415
 *   bde_op_*  (one reference sub-module each; used by the per-block parity tests)
 *       <- ConvLayer.forward            model/BDE2VID/submodules.py:105-114
 *          RecurrentConv.forward        model/BDE2VID/submodules.py:191-195 (+ConvLSTM.forward :293-334)
 *          UpsampleConvLayer.forward    model/BDE2VID/submodules.py:137-147
 *          DFrameAttention.forward      model/BDE2VID/DTransformer.py:376-389
 *          predI + activation           ..._V5.py:195-197
 */
#ifndef BDE2VID_H
#define BDE2VID_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BDE_MAX_LEVELS 8
#define BDE_MAX_FRAMES 8

typedef struct bde_model bde_model;

/* Generator hyper-parameters (constructor arguments of the reference generator, V5.py:19-23).
 * Fixed by this build: 7x7 windows, nwindow_size=None, one output channel, ReLU / GELU activations. */
typedef struct bde_config {
    int32_t num_bins;                      /* input channels (5) */
    int32_t basechannels;                  /* 32 */
    int32_t num_encoders;                  /* 3 */
    int32_t ks;                            /* 5 (or 3) */
    int32_t num_heads;                     /* 16 */
    int32_t frame_num;                     /* len(buffer_index) */
    int32_t q_idx;                         /* slot of the query frame; buffer_index[q_idx] must be 0 */
    int32_t activation;                    /* 0 = Identity, 1 = Sigmoid */
    int32_t depths[BDE_MAX_LEVELS];        /* attention blocks per level (0 = none) */
    int32_t buffer_index[BDE_MAX_FRAMES];  /* temporal offsets of the attention buffer */
    /* ---- constructor variants (ABI version 2; all zero / use_rc = 1 = the canonical flags) ---- */
    int32_t recurrent_type;                /* 0 = ConvLSTM (submodules.py:278-334), 1 = ConvGRU (:337-376) */
    int32_t use_rc;                        /* 1 = RecurrentConv encoders, 0 = bare ConvLayer encoders (V5.py:250-258) */
    int32_t skip_concat;                   /* 0 = skip_sum, 1 = skip_concat + 1x1 fusion convs (V5.py:86-93,285-293) */
    int32_t norm;                          /* 0 = none, 1 = BatchNorm2d, 2 = InstanceNorm2d(track_running_stats), eval mode:
                                              folded into the convolutions at bde_finalize_weights (submodules.py:96-109) */
    int32_t num_res_blocks;                /* ResidualBlockNoBN count of the last level when depths[last] == 0 (V5.py:77-80) */
} bde_config;

/* status codes */
#define BDE_OK 0
#define BDE_ERR_ARG (-1)
#define BDE_ERR_STATE (-2)
#define BDE_ERR_HIP (-3)
#define BDE_ERR_UNSUPPORTED (-4)
#define BDE_ERR_RANGE (-5)       /* an activation left the range of the two-term operand format and "sb_auto" is 0 */

/* Version of this header; bde_abi_version() returns the one the library was built from (bde2vid_amd/_lib.py refuses a mismatch).
 * 4: BDE_ERR_RANGE, "sb_auto" and the range guard of the two-term operand format. */
#define BDE_ABI_VERSION 4

const char* bde_last_error(void);
int bde_abi_version(void);

/* ---- model life cycle --------------------------------------------------------------------- */
int bde_create(const bde_config* cfg, bde_model** out);
void bde_destroy(bde_model* m);
/* `data` is a HOST pointer to a contiguous fp32 tensor; `key` is the reference state_dict key
 * (SURVEY.md Appendix B).  Unknown keys that the forward path never reads (fusion_layers.*,
 * relative_position_index) are accepted and ignored. */
int bde_load_weight(bde_model* m, const char* key, const float* data, const int64_t* shape, int32_t ndim);
/* Checks that every tensor of the forward path is present with the right shape, folds the
 * LayerNorms / query scale into the projection weights, packs everything into MFMA fragment
 * order and uploads it to the current HIP device. */
int bde_finalize_weights(bde_model* m);
/* Number of fp32 values of the packed device image, and raw access for the RCCL broadcast
 * (bde2vid_amd/dist.py): rank 0 finalizes, the others allocate with bde_alloc_packed and receive. */
int64_t bde_packed_numel(const bde_model* m);
float* bde_packed_ptr(bde_model* m);
int bde_alloc_packed(bde_model* m);

/* ---- the hot path --------------------------------------------------------------------------- */
/* events[t]: device fp32 [B][num_bins][Hp][Wp] (NCHW);  images[t]: device fp32 [B][1][Hp][Wp].
 * Hp, Wp multiples of 2^num_encoders, feature maps at attention levels >= 7x7.
 * Recurrent state always starts from zero (bde2vid.py:31).  Launches on `stream`.
 * Range guard of the default operand format (two fp16 terms, csrc/split.h): a forward in which an activation reaches 65520 is
 * detected on the device.  With "sb_auto" = 1 (default) the model then switches to three bf16 terms (fp32's exponent range) for good
 * and the forward is recomputed into the same `images`; with "sb_auto" = 0 the call fails with BDE_ERR_RANGE.  In the default
 * mode ("pipeline" = 1) this happens inside bde_forward, which therefore waits for the forward up to its last convolution
 * before it returns (the convolution itself is still in flight; with "sb_terms" = 3 nothing is checked and nothing waited for);
 * in pipelined mode it happens in bde_wait_outputs, or in the bde_forward call that reuses the workspace slot. */
int bde_forward(bde_model* m, const float* const* events, int32_t T, int32_t B, int32_t Hp, int32_t Wp,
                float* const* images, void* stream);
/* Copy a named intermediate of the last bde_forward into `dst` (device): "head", "merged<l>"
 * (level output after attention), "dec<j>".  Layout [T][B][C][H][W]. */
int bde_get_intermediate(bde_model* m, const char* name, float* dst, int64_t numel, void* stream);

/* ---- ONE sequence over TWO GPUs, split by sweep direction (V5.py:122-147: the forward and the backward RecurrentConv sweeps of a
 * level are independent until their outputs are added).  Rank A: bde_split_begin, then per level bde_split_sweep(level, 0),
 * receive "hidden"(level, 1) from rank B, bde_split_attend(level), send "level_out"(level) to B; finally bde_split_decode.
 * Rank B: bde_split_begin, then per level bde_split_sweep(level, 1), send "hidden"(level, 1), receive "level_out"(level)
 * (not after the last level).  bde_split_buffer gives the device address and length of those tensors inside the model's
 * workspace ([T][B][C][h][w] fp32), for a P2P copy / RCCL send-recv by the caller (bde2vid_amd/dist.py::DirectionSplit).
 * Launch shapes are chosen as in the joint forward, so rank A's frames are the single-GPU frames bit for bit. */
int bde_split_begin(bde_model* m, const float* const* events, int32_t T, int32_t B, int32_t Hp, int32_t Wp, void* stream);
int bde_split_sweep(bde_model* m, int32_t level, int32_t direction, void* stream);
int bde_split_attend(bde_model* m, int32_t level, void* stream);
int bde_split_decode(bde_model* m, float* const* images, void* stream);
int bde_split_buffer(bde_model* m, const char* what, int32_t level, int32_t direction, float** ptr, int64_t* numel);

/* Pipelined mode ("pipeline" = 2..4, see bde_set_tuning): consecutive bde_forward calls (independent
 * sequences) rotate over that many internal streams and workspaces; inputs are ordered after the
 * caller's `stream`, but the outputs of a call are only ordered into `stream` by bde_wait_outputs
 * (call it before anything on `stream` reads them; `images` of the calls issued since the last bde_wait_outputs must stay
 * valid until it returns: it is also where a forward that left the range of the two-term operand format is recomputed, see
 * bde_forward, and where BDE_ERR_RANGE is reported).  Default is "pipeline" = 1: no such call needed. */
int bde_wait_outputs(bde_model* m, void* stream);

/* Scheduling knobs (results are unchanged up to fp32 summation order).  Keys:
 *   "pipeline": 1 (default) .. 4 sequences in flight per model object, see bde_wait_outputs.
 *   "graph": 1 (default) replays the launch sequence of a forward from a hipGraph captured at the second
 *              call of a shape; 0 launches eagerly.
 *   "fused_min_tiles": a level with at least this many 32-pixel tiles runs the post-softmax part of an
 *                      attention block as one fused kernel instead of three GEMM launches (default 160).
 *   "sb_terms": operand format of every split kernel (csrc/split.h): 2 (default) = two fp16 terms, three MFMAs per fp32
 *              block, fp32-equivalent accuracy, finite activations must stay below 65520 -- guarded, see "sb_auto"; 3 = three bf16
 *              terms, six MFMAs, fp32's exponent range.
 *   "sb_auto": 1 (default) = a forward in which an activation reaches 65520 under "sb_terms" = 2 is recomputed with three bf16
 *              terms and the model keeps that format (bde_get_info "sb_latched"); 0 = such a forward fails with BDE_ERR_RANGE.
 *   "conv_sb": 1 (default) runs the batched convolutions that have a split-operand shape (csrc/conv_sb.h) on the 16-bit
 *              matrix cores (fp32-equivalent); 0 keeps every convolution -- and the recurrent step -- on the fp32 kernels.
 *   "xcd_remap": 1 (default) orders the workgroups of the batched convolutions so that each XCD's L2 sees one contiguous
 *              range of (frame, pixel tile, channel group); 0 = plain grid order.  Same results.
 *   "fuse_enc_sb": 1 (default) lets an encoder convolution store its result only as the split-bf16 image its gate
 *              convolution reads; 0 writes fp32 planes and converts them in a pass of their own (same frames, bit for bit).
 *   "lstm_sbk": 1 (default) runs the recurrent ConvLSTM step on the 16-bit matrix cores with split operands and the
 *              pointwise tail fused (csrc/lstm_sb.h: fp32-equivalent) wherever a shape fits; 0 = the fp32 matrix-core step (lstm16.h).
 *   "lstm_fuse_x": 1 (default) lets that step contract the stacked input [x | h] itself (submodules.py:316-317): no batched
 *              gate convolution and no buffer for its result; 0 = the x-part of the gates batched over T by conv_sb.
 *   "wide_kv_sb" / "wide_fuse_mlp": 1 (default) run, on head_dim-16 levels under "sb_terms" = 2, the K|V and q|k|v GEMMs on two-term
 *              split operands / the proj and fc1 GEMMs of a block as one launch (csrc/wideblock.h); 0 = the fp32 launches.
 *   "winblock_sb": 1 (default) runs the four GEMM phases of that one-launch block on the 16-bit matrix cores, split
 *              operands (csrc/winblock_sb.h: fp32-equivalent); 0 = fp32 MFMAs throughout (winblock.h).
 *   "winblock": 1 (default) runs an attention block of a 64-channel / 16-head level as ONE launch
 *               (csrc/winblock.h); 0 keeps the split path (attention core + fused token kernel).
 *   "attn_mfma": 1 (default) uses the matrix-core attention core for head_dim 16 (csrc/attn_mfma.h).
 * Diagnostics only (results become wrong): "debug_skip" = bit mask of stages left out of a forward
 * (1 attention level 0, 2 attention levels >= 1, 4 recurrent steps, 8 decoder, 16 encoder + gate convs),
 * used by tools/whatif.sh to read the marginal cost of a stage. */
int bde_set_tuning(bde_model* m, const char* key, int64_t value);
/* Read back the state the measurement has to be honest about: "debug_skip" (non-zero = stages skipped, results
 * invalid), "graph" (0 also after a failed capture), "graphs_live" (workspaces replaying a captured launch
 * sequence), "pipeline", "winblock", "winblock_sb", "wide", "conv_sb", "sb_terms", "sb_auto", "sb_latched" (1 after the range guard
 * switched the model to three bf16 terms), "sb_overflows" (forwards that left the range so far), "sb_overflow_word" (the raw
 * overflow word of slot 0, cleared by the read: set by bde_op_* / bde_split_* calls, which do not recompute), "lstm_sb", "lstm_sbk", "lstm_fuse_x", "wide_kv_sb", "wide_fuse_mlp", "last_stream",
 * "device", "packed_numel"; and which convolutions the latest forward ran on split operands (csrc/conv_sb.h): "sb_head",
 * "sb_enc<l>", "sb_gx<l>" (0 when the step contracts [x | h] itself: no such launch), "sb_dec<j>" (0 / 1), and
 * "sb_lstm<l>": the recurrent steps of level l ran on the fused split-operand step kernel (csrc/lstm_sb.h).
 * Settings are per model object. */
int bde_get_info(const bde_model* m, const char* key, int64_t* value);

/* Diagnostics: resident workgroups per CU the runtime reports for a named kernel (-1 = unknown). */
int bde_debug_occupancy(const char* kernel);
/* Diagnostics (host arithmetic, no GPU): the workgroup shape the split-operand convolution launcher picks, in the default
 * operand format, for a ks x ks conv of `cout` output channels on an in_h x in_w input: 0 = none (fp32 kernels), 1 = 128
 * channels x 128 pixels, 2 = 128 x 64, 3 = 64 x 128, 4 = 32 x 256 on 2-D pixel tiles, 5 = 64 x 128 on 2-D pixel tiles;
 * *row_tiles (may be NULL) = pixel tiles per image row, 0 = tiles run linearly over rows, < 0 = minus the columns of a 2-D tile. */
int bde_debug_conv_shape(int32_t ks, int32_t stride, int32_t cout, int32_t in_h, int32_t in_w, int32_t* row_tiles);
/* Diagnostics (host arithmetic, no GPU): the split operand formats of csrc/split.h as the weight packer applies them.
 * terms = 2: out[2 i], out[2 i + 1] = the two fp16 terms (bit patterns) of x[i] * scale; terms = 3: out[3 i ..] = the three bf16
 * terms of x[i] (scale ignored).  Returns the power-of-two packing scale split.h would choose for x[0..n) (terms = 2; else 1). */
float bde_debug_split(const float* x, int64_t n, int32_t terms, float scale, uint16_t* out);
/* Diagnostics: s_memtime stamps of the fused token kernel's phases ([block<64][wave][8]); first call
 * with host_out == NULL enables them, a later call copies n values back. */
int bde_debug_token_stamps(bde_model* m, int64_t* host_out, int32_t n);

/* HIP-event timing of the tagged launches of subsequent bde_forward calls (bench.py's roofline), one span per
 * launch on the stream the launch goes to: "head", "enc_conv<l>", "gates_x<l>", "lstm<l>" (one per recurrent step),
 * "merge<l>", "to_tok<l>", "winblock<l>", "chain_*<l>" (the kernels of the split attention path), "dec_up<j>",
 * "dec_conv<j>", "pred"; and stage spans around them: "forward", "attn<l>", "decoder".  bde_profile_names lists what
 * was recorded; bde_profile_get synchronises on the recorded events. */
int bde_profile_reset(bde_model* m, int32_t enable);
int bde_profile_names(bde_model* m, char* buf, int64_t buflen);
int bde_profile_get(bde_model* m, const char* name, double* total_ms, int64_t* count);

/* Event -> voxel grid.  xs, ys, ts, ps: device fp32 [N] (ts sorted ascending);
 * grid: device fp32 [num_bins][H][W], overwritten.  oob_count (device int32, may be NULL) receives
 * the number of events whose pixel index fell outside the sensor (the reference raises there). */
int bde_voxelize(const float* xs, const float* ys, const float* ts, const float* ps, int64_t N,
                 int32_t num_bins, int32_t H, int32_t W, float* grid, int32_t* oob_count, void* stream);
/* nseg packets in one launch: packet s owns events [offsets[s], offsets[s+1]) (device int64
 * [nseg+1]) and writes grid[s]. */
int bde_voxelize_batch(const float* xs, const float* ys, const float* ts, const float* ps,
                       const int64_t* offsets, int32_t nseg, int64_t max_events_per_seg, int32_t num_bins,
                       int32_t H, int32_t W, float* grids, int32_t* oob_count, void* stream);

/* The same grids straight from a recording's native event columns (Monash HDF5 schema,
 * data_loader/h5_dataset.py:410-415: xs, ys int16, ts float64 seconds, ps bool as one byte), one grid per
 * between-frames window: window w owns events [offsets[w], offsets[w+1]) (device int64 [nwin+1], e.g. the
 * images' `event_idx` attributes) and writes grids[w].  Restates BaseVoxelDataset.__getitem__
 * (h5_dataset.py:213-226: fewer than 3 events -> zero grid; ts - ts[first] in float64, then float32;
 * ps*2-1) followed by get_voxel_grid (:343-366) with the default all-ones hot-pixel mask.
 * All pointers are device pointers; max_events_per_window sizes the launch. */
int bde_voxelize_events(const int16_t* xs, const int16_t* ys, const double* ts, const uint8_t* ps,
                        const int64_t* offsets, int32_t nwin, int64_t max_events_per_window, int32_t num_bins,
                        int32_t H, int32_t W, float* grids, int32_t* oob_count, void* stream);

/* The same for arbitrary, possibly overlapping windows [starts[w], ends[w]) (device int64 [nwin] each): the k_events and
 * t_seconds voxel methods with a sliding window (h5_dataset.py:277-302).  n_events = length of the four columns: window
 * bounds beyond it are clamped on the device (what an h5py slice does with a file whose attributes overstate its datasets),
 * so no window can read past the columns.  max_events_per_window = an upper bound of ends[w] - starts[w] known to the host
 * (it sizes the bucketed path; a window longer than the bound would lose its tail there), or 0 = unknown: the streaming
 * kernel is used. */
int bde_voxelize_event_ranges(const int16_t* xs, const int16_t* ys, const double* ts, const uint8_t* ps, int64_t n_events,
                              const int64_t* starts, const int64_t* ends, int32_t nwin, int64_t max_events_per_window,
                              int32_t num_bins, int32_t H, int32_t W, float* grids, int32_t* oob_count, void* stream);
/* Binning kernel behind all bde_voxelize* calls of the process: 0 (default) = automatic: large calls are bucketed (one pass moves
 * every event once into the run of its (window, pixel tile), 29 B of traffic per event whatever the tile count; csrc/voxel.h),
 * small ones stream (a workgroup per (window, pixel tile) reads the window's events and accumulates its tile in LDS);
 * 1 = one global float atomic per tap; 2 = bucketed always; 3 = streaming always (1 and 3: kept for A/B timing). */
int bde_voxel_method(int32_t method);
/* DynamicH5Dataset.find_ts_index (h5_dataset.py:444-446 -> event_utils.py:10-28) for nq timestamps at once on the
 * device-resident events/ts column (float64 [n], ascending): out[i] = the index the reference's bisection returns. */
int bde_find_ts_index(const double* ts, int64_t n, const double* timestamps, int32_t nq, int64_t* out, void* stream);

/* ---- quality metrics of eval_model's scoring loop (eval_models_seq.py:229-258), device-resident ----------------- */
/* a, b: device fp32 [N][numel_per_image]; out: device fp64 [N], out[n] = mean((a-b)^2) of image n
 * (F.mse_loss per frame, evaluate/metrics.py:42-43).  scratch: device fp64 [bde_metric_scratch_doubles(N)]. */
int bde_metric_mse(const float* a, const float* b, int64_t numel_per_image, int32_t N, double* scratch, double* out,
                   void* stream);
/* a, b: device fp32 [N][H][W]; out[n] = skimage.metrics.structural_similarity(a[n], b[n]) as called at
 * evaluate/metrics.py:58-63 (7x7 uniform window, K1 .01, K2 .03, sample covariance, float64; data_range = 2 reproduces
 * the call without data_range on float images).  H, W >= 7. */
int bde_metric_ssim(const float* a, const float* b, int32_t H, int32_t W, int32_t N, double data_range, double* scratch,
                    double* out, void* stream);
int32_t bde_metric_scratch_doubles(int32_t N);

/* ---- single reference sub-modules (parity tests) ------------------------------------------- */
int bde_op_head(bde_model* m, const float* in, int32_t N, int32_t H, int32_t W, float* out, void* stream);
/* RecurrentConv of level `level`, direction `dir` (0 forward_encoder, 1 backward_encoder) applied
 * to T consecutive inputs in[t] = [B][Cin][H][W] starting from zero state; h_out [T][B][C][H/2][W/2],
 * c_out [B][C][H/2][W/2] (final cell state).  dir 0 sweeps t = 0..T-1, dir 1 sweeps t = T-1..0
 * (V5.py:123); h_out[t] always belongs to input frame t. */
int bde_op_recurrent_conv(bde_model* m, int32_t level, int32_t dir, const float* in, int32_t T, int32_t B,
                          int32_t H, int32_t W, float* h_out, float* c_out, void* stream);
int bde_op_encoder_conv(bde_model* m, int32_t level, int32_t dir, const float* in, int32_t N, int32_t H,
                        int32_t W, float* out, void* stream);
/* x-part of the ConvLSTM gates of `level` for both directions (submodules.py:316-317 on the x half of the stacked input):
 * in [2][N][C][H][W], out [2][N][4C][H][W], bias included. */
int bde_op_gate_conv(bde_model* m, int32_t level, const float* in, int32_t N, int32_t H, int32_t W, float* out,
                     void* stream);
/* decoder j: out = ReLU6(conv(bilinear_x2(in + skip)))); skip may be NULL */
int bde_op_decoder(bde_model* m, int32_t j, const float* in, const float* skip, int32_t N, int32_t H, int32_t W,
                   float* out, void* stream);
int bde_op_pred(bde_model* m, const float* in, const float* head, int32_t N, int32_t H, int32_t W, float* out,
                void* stream);
/* DFrameAttention of `level` on a buffer of frame_num frames [B][C][H][W]; bufs[d] == NULL is a
 * zero frame.  nblocks < 0 runs all blocks, otherwise only block `first_block` .. +nblocks. */
int bde_op_dframe_attention(bde_model* m, int32_t level, const float* const* bufs, int32_t B, int32_t H,
                            int32_t W, int32_t first_block, int32_t nblocks, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BDE2VID_H */
