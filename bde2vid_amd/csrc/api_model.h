// Part of libbde2vid's host side, included by bde_api.hip (one translation unit: the kernels of the headers it includes are
// emitted once).  The model object: packed-layer descriptors, attention-level tables, workspaces, bde_model, span profiling.
#pragma once
namespace bde {

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
    long h3_off[2] = {-1, -1};                  // head3 packing (conv_sb.h, KS_HEAD3): [terms - 2], ten (row, column-group) taps of three columns x <= 5 channels
    long h3_sz[2] = {0, 0};
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
    int head3 = 1;                // head convolution on the three-columns-per-chunk image (conv_sb.h, KS_HEAD3): 10 MFMA taps instead of 25
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
    int wide_prefetch = 0;        // 1: the attention core warms the L2s with the weights of the MLP launch behind it (wide_core.h).  Measured:
                                  // the MLP launch's first phase 13.2 k -> 12.5 k cycles, the bench 2191 -> 2124 / 1790 -> 1779 frames/s: off
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

}  // namespace bde
