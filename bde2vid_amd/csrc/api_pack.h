// Part of libbde2vid's host side, included by bde_api.hip (one translation unit: the kernels of the headers it includes are
// emitted once).  Weight packing: every consumer's fragment order (fp32 MFMA, split operands in both formats, attention tables), LayerNorm / BatchNorm folding, upload.
#pragma once
namespace bde {

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
// The head's 5x5 convolution over <= 5 input channels on the "three columns per chunk" image (conv_sb.h, KS_HEAD3): k = 5 jj + b of
// a 16-channel chunk is channel b of column x - 2 + jj (jj < 3; k = 15 unused), so a kernel row is TWO MFMA taps -- columns kx = 0..2
// at image column x, kx = 3..4 (+ a zero column) at x + 3 -- ten taps instead of twenty-five 16-channel chunks that are 11/16 zeros.
// [co tile 32][tap = 2 ky + g][term][64 lanes][8]: lane l, element j: k = 8 (l >> 5) + j -> W[row][k % 5][ky][3 g + k / 5].
// Same per-layer scale as the plain packing (the same weights).
static void pack_head3(Arena& ar, PackedLayer& pl, const DenseLayer& d) {
    if (d.KS != 5 || d.Cin * 3 > 15 || d.rows % 32 != 0) return;
    const int ncot = d.rows / 32, taps = 10;
    for (int terms = 2; terms <= 3; ++terms) {
        const long sz = (long)ncot * taps * terms * 64 * 8 / 2;      // floats
        const long off = ar.alloc(sz);
        pl.h3_off[terms - 2] = off;
        pl.h3_sz[terms - 2] = sz;
        const float scale = split_scale({&d}, terms);
        unsigned short* dst = reinterpret_cast<unsigned short*>(ar.host.data() + off);
        for (int ct = 0; ct < ncot; ++ct)
            for (int tap = 0; tap < taps; ++tap)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 8; ++j) {
                        const int row = ct * 32 + (l & 31), k = 8 * (l >> 5) + j;
                        const int jj = k / 5, b = k - jj * 5, ky = tap >> 1, kx = 3 * (tap & 1) + jj;
                        const float w = (k < 15 && b < d.Cin && kx < 5) ? d.w[((long)row * d.Cin + b) * 25 + ky * 5 + kx] : 0.f;
                        unsigned short t3[3];
                        split_terms(w, terms, scale, t3);
                        for (int kk = 0; kk < terms; ++kk) dst[((((long)ct * taps + tap) * terms + kk) * 64 + l) * 8 + j] = t3[kk];
                    }
    }
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
        pack_head3(ar, m->head, d);
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


}  // namespace bde
