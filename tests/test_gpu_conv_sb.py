"""Split-operand convolutions (csrc/conv_sb.h, csrc/split.h): fp32-equivalent arithmetic on the 16-bit matrix cores.

Every parity test of the fp32 kernels also runs through them (they are on by default); here their error is measured against
a float64 CPU reference next to the fp32 matrix-core kernel's own, for both operand formats (two fp16 terms / three MFMAs,
the default; three bf16 terms / six MFMAs): split products must stay within 2x of the fp32 kernel's error (plain bf16 or
fp16 operands would be ~1e-3 relative)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.util import dense_like, maxabs
from bde2vid_amd import canonical
from bde2vid_amd.weights import formula_state_dict

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def model_a():
    from bde2vid_amd.model import build_model
    cfg = canonical()
    sd = formula_state_dict(cfg)
    return cfg, sd, build_model(cfg, sd, 'cuda:0')


@pytest.mark.parametrize('level,hw,N', [(0, (92, 120), 4), (1, (46, 60), 8), (2, (23, 30), 16), (0, (33, 47), 6)])
def test_gate_conv_error_against_float64(model_a, level, hw, N):
    from bde2vid_amd import ops
    from oracle import bde2vid_oracle as O
    cfg, sd, m = model_a
    C = cfg.enc_out(level)
    x = torch.from_numpy(dense_like((2, N, C, hw[0], hw[1]), 900 + level))
    ref = []
    for d, name in enumerate(('forward_encoder', 'backward_encoder')):
        w = sd[f'{O.P}{name}.{level}.recurrent_block.Gates.weight'][:, :C].double()      # x half of [x | h] (submodules.py:316)
        b = sd[f'{O.P}{name}.{level}.recurrent_block.Gates.bias'].double()
        ref.append(F.conv2d(x[d].double(), w, b, padding=1))
    ref = torch.stack(ref)
    scale = float(ref.abs().max())
    xd = x.cuda()
    m.set_tuning('conv_sb', 0)
    y32 = ops.gate_conv(m, level, xd).cpu().double()
    m.set_tuning('conv_sb', 1)
    e32 = float((y32 - ref).abs().max()) / scale
    assert e32 <= 5e-6                                       # K up to 2304 products per output
    default_terms = m.get_info('sb_terms')
    assert default_terms == int(os.environ.get('BDE_SB_TERMS', 2))
    try:
        for terms in (2, 3):
            m.set_tuning('sb_terms', terms)
            ysb = ops.gate_conv(m, level, xd).cpu().double()
            assert m.get_info(f'sb_gx{level}') == 1
            esb = float((ysb - ref).abs().max()) / scale
            print(f'level {level} {hw}: fp32 matrix cores {e32:.2e}, {terms}-term split {esb:.2e} of max|ref|')
            assert esb <= max(2 * e32, 1e-6), f'{terms}-term split {esb:.2e} vs fp32 kernel {e32:.2e}'
            assert maxabs(ysb, y32) / scale <= 4e-6
    finally:
        m.set_tuning('sb_terms', default_terms)


def test_five_by_five_split_conv_of_decoder0(model_a):
    """Decoder 0 (5x5 stride 1, 256 -> 128 channels) on the 128 x 128 shape."""
    from bde2vid_amd import ops
    from oracle import bde2vid_oracle as O
    cfg, sd, m = model_a
    x = torch.from_numpy(dense_like((4, 256, 23, 30), 950))
    skip = torch.from_numpy(dense_like((4, 256, 23, 30), 951))
    with torch.no_grad():
        ref = O.upsample_conv_layer(skip + x, sd[O.P + 'decoders.0.1.conv2d.weight'], sd[O.P + 'decoders.0.1.conv2d.bias'])
    y = ops.decoder(m, 0, x.cuda(), skip.cuda())
    assert maxabs(y, ref) <= 1e-4


def test_decoder1_takes_the_64_channel_shape(model_a):
    """Decoder 1 (5x5 stride 1, 128 -> 64 channels) runs on 2 x 2 waves of 32 x 64 by default once a launch has >= 16384
    output pixels; odd map sizes put the tiles across image rows."""
    from bde2vid_amd import ops
    from oracle import bde2vid_oracle as O
    cfg, sd, m = model_a
    for (N, H, W, seed) in ((4, 46, 60, 960), (3, 37, 51, 962)):
        x = torch.from_numpy(dense_like((N, 128, H, W), seed))
        skip = torch.from_numpy(dense_like((N, 128, H, W), seed + 1))
        with torch.no_grad():
            ref = O.upsample_conv_layer(skip + x, sd[O.P + 'decoders.1.1.conv2d.weight'], sd[O.P + 'decoders.1.1.conv2d.bias'])
        y = ops.decoder(m, 1, x.cuda(), skip.cuda())
        m.set_tuning('conv_sb', 0)
        try:
            y32 = ops.decoder(m, 1, x.cuda(), skip.cuda())
        finally:
            m.set_tuning('conv_sb', 1)
        assert maxabs(y, ref) <= 1e-4, (N, H, W)
        assert maxabs(y, y32) <= 2e-5 * max(1.0, float(ref.abs().max()))


def test_stride2_encoder_conv(model_a):
    """The stride-2 encoder convolution of level 1 (64 -> 128 channels) on four waves of 32 x 64."""
    from bde2vid_amd import ops
    from oracle import bde2vid_oracle as O
    cfg, sd, m = model_a
    x = torch.from_numpy(dense_like((8, 64, 92, 120), 970))
    for d, name in enumerate(('forward_encoder', 'backward_encoder')):
        with torch.no_grad():
            ref = O.conv_layer(x, sd[f'{O.P}{name}.1.conv.conv2d.weight'], sd[f'{O.P}{name}.1.conv.conv2d.bias'], 2, 'relu')
        y = ops.encoder_conv(m, 1, d, x.cuda())
        assert maxabs(y, ref) <= 1e-4 * max(1.0, float(ref.abs().max())), name


def test_producers_write_the_split_bf16_inputs(model_a):
    """By default the producer of a split-bf16 convolution's input stores it directly as SB16 (the encoder convolution's
    epilogue for its gate convolution, conv_mfma.h sb_out; the bilinear x2 of decoders 0 and 1, upsample2x_sum_split_kernel):
    no fp32 map, no conversion pass.  Same split arithmetic as split_bf16_kernel, so the frames agree with the unfused
    schedule to rounding of the bilinear expression (its fused-multiply-add contraction may differ between the two kernels)."""
    from tests.util import golden_inputs
    cfg, sd, m = model_a
    xs = golden_inputs(6, 1, 5, 184, 240, 1234)
    inp = [{'events': torch.from_numpy(x).cuda()} for x in xs]
    y1 = torch.stack(m(inp))
    m.set_tuning('fuse_enc_sb', 0)
    try:
        y0 = torch.stack(m(inp))
    finally:
        m.set_tuning('fuse_enc_sb', 1)
    assert torch.isfinite(y1).all()
    assert maxabs(y0, y1) <= 2e-6


def test_workgroup_order_does_not_change_results_and_info_keys(model_a):
    """xcd_remap only permutes which workgroup computes which tile: frames are bit-identical with it on and off; get_info
    reports which convolutions of the latest forward ran on split operands (config A at 184 x 240: all of them, the head and
    the last decoder -- whose epilogue carries predI -- included)."""
    from tests.util import golden_inputs
    cfg, sd, m = model_a
    xs = golden_inputs(16, 1, 5, 184, 240, 4321)
    inp = [{'events': torch.from_numpy(x).cuda()} for x in xs]
    try:                                  # (16 frames: every level is over the launch-size threshold of the split-bf16 kernels)
        y1 = torch.stack(m(inp))
        took = {k: m.get_info(k) for k in ('sb_head', 'sb_enc0', 'sb_enc1', 'sb_enc2', 'sb_gx0', 'sb_gx1', 'sb_gx2', 'sb_dec0', 'sb_dec1', 'sb_dec2')}
        m.set_tuning('xcd_remap', 0)
        y0 = torch.stack(m(inp))
    finally:
        m.set_tuning('xcd_remap', 1)
    # (no batched gate convolution: the recurrent step contracts [x | h] itself, lstm_fuse_x)
    assert took == dict(sb_head=1, sb_enc0=1, sb_enc1=1, sb_enc2=1, sb_gx0=0, sb_gx1=0, sb_gx2=0, sb_dec0=1, sb_dec1=1, sb_dec2=1), took
    assert [m.get_info(f'sb_lstm{l}') for l in range(3)] == [1, 1, 1] and m.get_info('lstm_fuse_x') == 1
    assert torch.equal(y0, y1)
    with pytest.raises(Exception):
        m.get_info('sb_enc9')


def test_gate_conv_random_shapes(model_a):
    """Seeded sweep: odd widths (tiles straddling rows), maps narrower / wider than a pixel tile, few and many frames, on
    both sides of the launch-size threshold under which the fp32 kernels are used."""
    from bde2vid_amd import ops
    from oracle import bde2vid_oracle as O
    cfg, sd, m = model_a
    rng = np.random.default_rng(7)
    for case in range(8):
        level = int(rng.integers(0, 3))
        C = cfg.enc_out(level)
        H, W = int(rng.integers(5, 70)), int(rng.integers(5, 150))
        N = int(rng.integers(1, 6))
        x = torch.from_numpy(dense_like((2, N, C, H, W), 1000 + case))
        ref = []
        for d, name in enumerate(('forward_encoder', 'backward_encoder')):
            w = sd[f'{O.P}{name}.{level}.recurrent_block.Gates.weight'][:, :C]
            b = sd[f'{O.P}{name}.{level}.recurrent_block.Gates.bias']
            ref.append(F.conv2d(x[d], w, b, padding=1))
        ref = torch.stack(ref)
        y = ops.gate_conv(m, level, x.cuda())
        assert maxabs(y, ref) <= 1e-4 * max(1.0, float(ref.abs().max())), f'case {case}: level {level} N={N} {H}x{W}'


def test_recurrent_step_on_split_bf16(model_a):
    """The opt-in form of the recurrent step (h-part of the gates by conv_sb_kernel + an element-wise ConvLSTM tail,
    set_tuning('lstm_sb', 1)) against the default lstm16 kernel: RecurrentConv of level 0, four steps, both directions."""
    from bde2vid_amd import ops
    cfg, sd, m = model_a
    x = torch.from_numpy(dense_like((4, 1, cfg.enc_in(0), 184, 240), 1100)).cuda()
    ref = [ops.recurrent_conv(m, 0, d, x) for d in (0, 1)]
    m.set_tuning('lstm_sb', 1)
    try:
        got = [ops.recurrent_conv(m, 0, d, x) for d in (0, 1)]
    finally:
        m.set_tuning('lstm_sb', 0)
    for (h0, c0), (h1, c1) in zip(ref, got):
        assert maxabs(h1, h0) <= 2e-5 and maxabs(c1, c0) <= 2e-5


def test_non_finite_activations_same_class_on_both_arithmetic_paths(model_a):
    """A NaN activation must come out as NaN at exactly the output pixels its 3x3 footprint reaches, on the fp32 kernels
    (conv_sb 0) and on the split-bf16 ones (conv_sb 1) alike, as torch's convolution gives them; an infinite activation
    rides in the leading bf16 term alone (sb_split3), so it stays non-finite on the same footprint on both paths
    (Inf * w_mid can turn an Inf into a NaN on the split path: INTEGRATION.md §3)."""
    from bde2vid_amd import ops
    from oracle import bde2vid_oracle as O
    cfg, sd, m = model_a
    level, C, N, H, W = 2, cfg.enc_out(2), 16, 23, 30
    x = torch.from_numpy(dense_like((2, N, C, H, W), 990))
    x[0, 3, 17, 5, 7] = float('nan')
    x[1, 9, 100, 20, 29] = float('inf')
    x[1, 2, 5, 0, 0] = float('-inf')
    ref = []
    for d, name in enumerate(('forward_encoder', 'backward_encoder')):
        w = sd[f'{O.P}{name}.{level}.recurrent_block.Gates.weight'][:, :C]
        ref.append(F.conv2d(x[d], w, sd[f'{O.P}{name}.{level}.recurrent_block.Gates.bias'], padding=1))
    ref = torch.stack(ref)
    xd = x.cuda()
    outs = {}
    for sb in (0, 1):
        m.set_tuning('conv_sb', sb)
        outs[sb] = ops.gate_conv(m, level, xd).cpu()
    m.set_tuning('conv_sb', 1)
    assert m.get_info('sb_gx2') == 1
    for sb, y in outs.items():
        assert torch.equal(torch.isfinite(y), torch.isfinite(ref)), f'conv_sb={sb}: non-finite footprint differs from torch'
        assert torch.isnan(y[0, 3, :, 4:7, 6:9]).all()                       # the NaN's 3x3 footprint, every output channel
        fin = torch.isfinite(ref)
        assert maxabs(y[fin], ref[fin]) <= 1e-4 * float(ref[fin].abs().max())
    assert torch.equal(torch.isnan(outs[0][0]), torch.isnan(ref[0]))        # NaN in, NaN out: identical on the fp32 path
    assert torch.equal(torch.isnan(outs[1][0]), torch.isnan(ref[0]))        # ... and on the split path


def test_range_of_the_two_term_format(model_a):
    """fp16 terms keep 5 exponent bits (csrc/split.h): an activation of 65520 or more turns infinite in the default two-term
    format -- the output is non-finite on that activation's footprint and the range guard raises the overflow word (a whole
    forward is then recomputed with three terms: tests/test_gpu_e2e.py::test_range_guard_*) -- while three bf16 terms
    (set_tuning('sb_terms', 3)) and the fp32 kernels carry it; just below the limit all three agree with torch.  Tiny
    activations (fp16 subnormal range) keep an absolute accuracy of 2^-25."""
    from bde2vid_amd import ops
    from oracle import bde2vid_oracle as O
    cfg, sd, m = model_a
    level, C, N, H, W = 2, cfg.enc_out(2), 16, 23, 30
    base = torch.from_numpy(dense_like((2, N, C, H, W), 991))

    def run(x, terms):
        m.set_tuning('sb_terms', terms)
        try:
            return ops.gate_conv(m, level, x.cuda()).cpu()
        finally:
            m.set_tuning('sb_terms', int(os.environ.get('BDE_SB_TERMS', 2)))

    def ref_of(x):
        out = []
        for d, name in enumerate(('forward_encoder', 'backward_encoder')):
            w = sd[f'{O.P}{name}.{level}.recurrent_block.Gates.weight'][:, :C].double()
            out.append(F.conv2d(x[d].double(), w, sd[f'{O.P}{name}.{level}.recurrent_block.Gates.bias'].double(), padding=1))
        return torch.stack(out)

    big = base.clone()
    big[0, 4, 33, 10, 12] = 1.0e5                       # beyond fp16
    big[1, 7, 60, 3, 3] = 65000.0                       # within
    ref = ref_of(big)
    m.get_info('sb_overflow_word')                      # (reading clears it)
    y3 = run(big, 3)
    assert m.get_info('sb_overflow_word') == 0          # three bf16 terms: nothing to guard
    y2 = run(big, 2)
    assert m.get_info('sb_overflow_word') == 1          # the range guard saw the 1e5 (an op-level call does not recompute)
    scale = float(ref.abs().max())
    assert torch.isfinite(y3).all() and maxabs(y3.double(), ref) <= 2e-6 * scale
    bad = ~torch.isfinite(y2)
    assert bad[0, 4, :, 9:12, 11:14].all() and int(bad.sum()) == 9 * 4 * C          # exactly the 3x3 footprint, every gate row
    assert maxabs(y2[~bad].double(), ref[~bad]) <= 2e-6 * scale                       # 65000 is carried
    tiny = base * 1e-6                                   # every low term (and many leading ones) subnormal in fp16
    ref = ref_of(tiny)
    y2 = run(tiny, 2)
    assert m.get_info('sb_overflow_word') == 0
    assert maxabs(y2.double(), ref) <= 1e-5 * float((ref - ref.mean()).abs().max()) + 2e-7


def test_two_dimensional_tiles_on_odd_maps(model_a):
    """The 32- and 64-output-channel 5x5 convolutions (head, encoder 0, decoder 2) run on 2-D pixel tiles of 16 x 16 / 8 x 16
    (8 or 32 columns where that covers the map better): maps whose height and width are not multiples of the tile, narrower than
    a tile, a single tile row -- against torch's convolution."""
    from tests.util import golden_inputs
    from bde2vid_amd import ops
    from oracle import bde2vid_oracle as O
    cfg, sd, m = model_a
    m([{'events': torch.from_numpy(x).cuda()} for x in golden_inputs(16, 1, 5, 184, 240, 77)])     # a workspace with room for the split images
    for (N, H, W, seed) in ((9, 47, 61, 1200), (4, 100, 77, 1201), (2, 16, 600, 1202), (1, 200, 90, 1203), (40, 21, 20, 1204)):
        x = torch.from_numpy(dense_like((N, 5, H, W), seed))
        with torch.no_grad():
            ref = O.conv_layer(x, sd[O.P + 'head.conv2d.weight'], sd[O.P + 'head.conv2d.bias'], 1, 'relu')
        y = ops.head(m, x.cuda())                        # head3: three grid columns per chunk, ten MFMA taps (csrc/conv_sb.h, KS_HEAD3)
        assert m.get_info('sb_head') == 1 and m.get_info('head3') == 1, (N, H, W)
        assert maxabs(y, ref) <= 1e-4 * max(1.0, float(ref.abs().max())), ('head', N, H, W)
        m.set_tuning('head3', 0)                         # ... and the generic split convolution (25 taps on a 16-channel chunk)
        try:
            y0 = ops.head(m, x.cuda())
        finally:
            m.set_tuning('head3', 1)
        assert maxabs(y0, ref) <= 1e-4 * max(1.0, float(ref.abs().max())), ('head, generic', N, H, W)
        assert maxabs(y, y0) <= 2e-5 * max(1.0, float(ref.abs().max())), ('head3 vs generic', N, H, W)
    for (N, H, W, seed) in ((10, 74, 90, 1210), (8, 66, 130, 1211), (3, 34, 490, 1212), (24, 38, 36, 1213)):
        x = torch.from_numpy(dense_like((N, 32, H, W), seed))
        for d, name in enumerate(('forward_encoder', 'backward_encoder')):
            with torch.no_grad():
                ref = O.conv_layer(x, sd[f'{O.P}{name}.0.conv.conv2d.weight'], sd[f'{O.P}{name}.0.conv.conv2d.bias'], 2, 'relu')
            y = ops.encoder_conv(m, 0, d, x.cuda())
            assert m.get_info('sb_enc0') == 1, (N, H, W)
            assert maxabs(y, ref) <= 1e-4 * max(1.0, float(ref.abs().max())), ('enc0', name, N, H, W)
    for (N, H, W, seed) in ((3, 37, 45, 1220), (2, 8, 301, 1221), (9, 23, 22, 1222)):
        x = torch.from_numpy(dense_like((N, 64, H, W), seed))
        skip = torch.from_numpy(dense_like((N, 64, H, W), seed + 50))
        with torch.no_grad():
            ref = O.upsample_conv_layer(skip + x, sd[O.P + 'decoders.2.1.conv2d.weight'], sd[O.P + 'decoders.2.1.conv2d.bias'])
        y = ops.decoder(m, 2, x.cuda(), skip.cuda())
        assert m.get_info('sb_dec2') == 1, (N, H, W)
        assert maxabs(y, ref) <= 1e-4 * max(1.0, float(ref.abs().max())), ('dec2', N, H, W)
