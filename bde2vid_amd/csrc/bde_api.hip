// libbde2vid.so -- the C ABI of include/bde2vid.h.
//
// One translation unit (the kernels of the headers it includes are emitted once), in five parts:
//   api_elementwise.h  element-wise kernels of the forward (merge, frame gather / scatter, bilinear x2, predI, ConvGRU gates)
//   api_model.h        the model object: packed-layer descriptors, attention-level tables, workspaces, bde_model, span profiling
//   api_pack.h         weight packing into every consumer's fragment order, LayerNorm / BatchNorm folding, upload
//   api_schedule.h     the forward's launch schedule -- BDE2VIDCrossscalePropogationV5.forward (V5.py:100-241): head conv (all T) ->
//                      per level { encoder conv x2 dirs (all T), T recurrent ConvLSTM steps (both directions per launch), merge,
//                      temporal window attention (sequential in t, V5.py:154-169) } -> decoder (all T) -> predI + sigmoid;
//                      workspaces, hipGraph capture, serving mode, the range guard of the two-term operand format
//   this file          error reporting and the extern "C" entry points
// The batched convolutions and the split-operand kernels are two more translation units (conv_tu.hip, sb_tu.hip).
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
}  // namespace bde

#include "api_elementwise.h"
#include "api_model.h"
#include "api_pack.h"
#include "api_schedule.h"


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
    if (std::string(key) == "wide_prefetch") { m->wide_prefetch = (int)value; return BDE_OK; }
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
    if (std::string(key) == "head3") { m->head3 = (int)value; return BDE_OK; }
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
    else if (k == "head3") *value = m->head3;
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
    return run_head_conv(m, in, out, N, H, W, (hipStream_t)stream);
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
