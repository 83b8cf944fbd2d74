"""GPU parity of the one-launch attention block (csrc/winblock.h) at the canonical 64-channel / 16-head
level 0, against the CPU oracle (oracle/bde2vid_oracle.py, itself pinned to the reference by the golden
vectors) and against the split path (attention core + fused token kernel) of the same library.

Shapes pick out the kernel's cases: padding on both sides of the window grid, maps of a single window,
dilated blocks whose uncovered pixels ride in the spare token columns (large maps) or need the extra
workgroups (few windows: more than 15 uncovered pixels per window), zero frames at the sequence ends,
batch > 1.  fp32; tolerance as in test_gpu_blocks.py."""
import numpy as np
import pytest
import torch

from tests.util import maxabs, dense_like
from bde2vid_amd import canonical
from bde2vid_amd.weights import formula_state_dict, relative_position_index

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope='module')
def model_a():
    from bde2vid_amd.model import build_model
    cfg = canonical()
    sd = formula_state_dict(cfg)
    return cfg, sd, build_model(cfg, sd, 'cuda:0')


def oracle_blocks(cfg, sd, bufs, first, n, level=0):
    from oracle import bde2vid_oracle as O
    rel = torch.from_numpy(relative_position_index(cfg.frame_num, 7, 7))
    keys = [b if b is not None else torch.zeros_like(bufs[cfg.q_idx]) for b in bufs]
    x = keys[cfg.q_idx]
    for i in range(first, first + n):
        keys[cfg.q_idx] = x
        x = O.swin_block(torch.stack(keys, 0), sd, f'{O.P}feat_attns.{level}.blocks.{i}.', cfg.num_heads, cfg.q_idx, i % 2 == 1, 7, rel)
    return x


SHAPES = [
    (1, 64, 17, 23),     # pads 4 / 5, 12 windows: 10 uncovered pixels per window in dilated blocks
    (1, 64, 7, 7),       # one window, no padding: 33 uncovered pixels -> extra workgroups
    (2, 64, 14, 14),     # 4 windows, batch 2: 19 per window -> extra workgroups
    (1, 64, 30, 41),     # pads 5 / 1
    (1, 64, 46, 60),     # level-0 map of a 92 x 120 input
]


@pytest.mark.parametrize('shape', SHAPES)
def test_single_blocks_vs_oracle(model_a, shape):
    from bde2vid_amd import ops
    cfg, sd, m = model_a
    bufs = [torch.from_numpy(dense_like(shape, 300 + d)) for d in range(3)]
    dev = [b.cuda() for b in bufs]
    for blk in (0, 1):                                   # plain, dilated
        ref = oracle_blocks(cfg, sd, bufs, blk, 1)
        assert maxabs(ops.dframe_attention(m, 0, dev, blk, 1), ref) <= TOL, f'block {blk}'


@pytest.mark.parametrize('shape', SHAPES[:4])
def test_all_blocks_vs_oracle_and_split_path(model_a, shape):
    from bde2vid_amd import ops
    cfg, sd, m = model_a
    bufs = [torch.from_numpy(dense_like(shape, 320 + d)) for d in range(3)]
    dev = [b.cuda() for b in bufs]
    ref = oracle_blocks(cfg, sd, bufs, 0, cfg.depths[0])
    y = ops.dframe_attention(m, 0, dev)
    assert maxabs(y, ref) <= TOL
    m.set_tuning('winblock', 0)
    try:
        y_split = ops.dframe_attention(m, 0, dev)
    finally:
        m.set_tuning('winblock', 1)
    assert maxabs(y, y_split.cpu()) <= TOL


def test_zero_frames(model_a):
    """Out-of-range temporal slots are all-zero frames (V5.py:152-161): their tokens are LayerNorm(0) = beta."""
    from bde2vid_amd import ops
    cfg, sd, m = model_a
    q = torch.from_numpy(dense_like((1, 64, 15, 16), 341))
    nxt = torch.from_numpy(dense_like((1, 64, 15, 16), 342))
    for bufs in ([None, q, nxt], [nxt, q, None], [None, q, None]):
        ref = oracle_blocks(cfg, sd, bufs, 0, 2)
        y = ops.dframe_attention(m, 0, [None if b is None else b.cuda() for b in bufs], 0, 2)
        assert maxabs(y, ref) <= TOL


def test_inputs_untouched_and_deterministic(model_a):
    from bde2vid_amd import ops
    cfg, sd, m = model_a
    dev = [torch.from_numpy(dense_like((1, 64, 21, 28), 360 + d)).cuda() for d in range(3)]
    keep = [d.clone() for d in dev]
    y0 = ops.dframe_attention(m, 0, dev)
    y1 = ops.dframe_attention(m, 0, dev)
    assert torch.equal(y0, y1)
    for a, b in zip(dev, keep):
        assert torch.equal(a, b)


@pytest.mark.parametrize('shape', [(1, 256, 7, 9), (2, 256, 23, 30), (1, 256, 15, 16)])
def test_level2_matrix_core_attention_vs_oracle(model_a, shape):
    """Head dim 16 (level 2 of config A): attention core on the matrix cores (csrc/attn_mfma.h) against the
    oracle and against the vector-ALU core (csrc/attn.h)."""
    from bde2vid_amd import ops
    cfg, sd, m = model_a
    bufs = [torch.from_numpy(dense_like(shape, 380 + d)) for d in range(3)]
    dev = [b.cuda() for b in bufs]
    ref = oracle_blocks(cfg, sd, bufs, 0, 2, level=2)
    y = ops.dframe_attention(m, 2, dev, 0, 2)
    assert maxabs(y, ref) <= TOL
    m.set_tuning('attn_mfma', 0)
    try:
        y_valu = ops.dframe_attention(m, 2, dev, 0, 2)
    finally:
        m.set_tuning('attn_mfma', 1)
    assert maxabs(y, y_valu.cpu()) <= TOL
    ref0 = oracle_blocks(cfg, sd, [None, bufs[1], None], 0, 2, level=2)
    assert maxabs(ops.dframe_attention(m, 2, [None, dev[1], None], 0, 2), ref0) <= TOL


def test_random_shapes_vs_oracle(model_a):
    """Seeded sweep over map sizes / batch / missing frames: every padding split (0..6 extra rows and columns,
    odd and even), carried-pixel counts on both sides of the 15-per-window limit, both fallbacks."""
    from bde2vid_amd import ops
    cfg, sd, m = model_a
    rng = np.random.default_rng(20251004)
    for case in range(10):
        H, W = int(rng.integers(7, 45)), int(rng.integers(7, 45))
        B = int(rng.integers(1, 4))
        bufs = [torch.from_numpy(dense_like((B, 64, H, W), 500 + 10 * case + d)) for d in range(3)]
        drop = int(rng.integers(0, 4))                       # 0: none, 1: previous, 2: next, 3: both
        if drop in (1, 3):
            bufs[0] = None
        if drop in (2, 3):
            bufs[2] = None
        first = int(rng.integers(0, 3))
        n = int(rng.integers(1, cfg.depths[0] - first + 1))
        ref = oracle_blocks(cfg, sd, bufs, first, n)
        y = ops.dframe_attention(m, 0, [None if b is None else b.cuda() for b in bufs], first, n)
        assert maxabs(y, ref) <= TOL, f'case {case}: {B}x64x{H}x{W}, blocks {first}..{first + n - 1}, drop {drop}'


# ---- the kernels bench.py times, at the map sizes of the larger BASELINE.json resolutions ----------------------------
@pytest.mark.parametrize('hw', [(240, 320), (360, 640)])      # level-0 maps of VGA 480x640 and HD 720x1280
def test_winblock_on_large_level0_maps(model_a, hw):
    from bde2vid_amd import ops
    cfg, sd, m = model_a
    shape = (1, 64, hw[0], hw[1])
    bufs = [torch.from_numpy(dense_like(shape, 700 + d)) for d in range(3)]
    dev = [b.cuda() for b in bufs]
    for blk in (0, 1):                                   # plain, dilated (uncovered pixels ride in the spare columns)
        ref = oracle_blocks(cfg, sd, bufs, blk, 1)
        assert maxabs(ops.dframe_attention(m, 0, dev, blk, 1), ref) <= TOL, f'block {blk}'


@pytest.mark.parametrize('hw', [(33, 44), (60, 80), (90, 160)])   # level-2 maps of 264x352, VGA, HD
def test_level2_chain_on_large_maps(model_a, hw):
    from bde2vid_amd import ops
    cfg, sd, m = model_a
    shape = (1, 256, hw[0], hw[1])
    bufs = [torch.from_numpy(dense_like(shape, 720 + d)) for d in range(3)]
    dev = [b.cuda() for b in bufs]
    ref = oracle_blocks(cfg, sd, bufs, 0, 2, level=2)
    assert maxabs(ops.dframe_attention(m, 2, dev, 0, 2), ref) <= TOL


@pytest.mark.parametrize('hw,level', [((480, 640), 0), ((184, 240), 0), ((240, 320), 1), ((120, 160), 2)])
def test_recurrent_step_at_canonical_widths(model_a, hw, level):
    """RecurrentConv of the canonical config on full-size inputs: lstm16_step_kernel<1,128,2> at level 0 (wide maps),
    the 8-channel-workgroup variant at level 2, the float4-staged stride-2 5x5 and 3x3 gate convs."""
    from bde2vid_amd import ops
    from oracle import bde2vid_oracle as O
    cfg, sd, m = model_a
    Cin = cfg.enc_in(level)
    xs = [torch.from_numpy(dense_like((1, Cin, hw[0], hw[1]), 740 + t)) for t in range(2)]
    for direction, name in ((0, 'forward_encoder'), (1, 'backward_encoder')):
        pre = f'{O.P}{name}.{level}.'
        order = range(2) if direction == 0 else range(1, -1, -1)
        state, ref = None, [None, None]
        with torch.no_grad():
            for t in order:
                x = O.conv_layer(xs[t], sd[pre + 'conv.conv2d.weight'], sd[pre + 'conv.conv2d.bias'], 2, 'relu')
                state = O.convlstm_cell(x, state, sd[pre + 'recurrent_block.Gates.weight'], sd[pre + 'recurrent_block.Gates.bias'])
                ref[t] = state[0]
        h, c = ops.recurrent_conv(m, level, direction, torch.stack(xs).cuda())
        assert maxabs(h, torch.stack(ref)) <= TOL
        assert maxabs(c, state[1]) <= TOL


def test_random_shapes_level2_chain_vs_oracle(model_a):
    """Seeded sweep over map sizes / batch / missing frames for the fragment-layout chain (csrc/wideblock.h): token counts
    that are not multiples of 16 (the last fragment tile is partial), every padding split, dilated blocks with uncovered
    pixels, zero frames at the sequence ends."""
    from bde2vid_amd import ops
    cfg, sd, m = model_a
    rng = np.random.default_rng(20261004)
    for case in range(8):
        H, W = int(rng.integers(7, 40)), int(rng.integers(7, 40))
        B = int(rng.integers(1, 3))
        bufs = [torch.from_numpy(dense_like((B, 256, H, W), 800 + 10 * case + d)) for d in range(3)]
        drop = int(rng.integers(0, 4))
        if drop in (1, 3):
            bufs[0] = None
        if drop in (2, 3):
            bufs[2] = None
        first = int(rng.integers(0, 4))
        n = int(rng.integers(1, min(3, cfg.depths[2] - first) + 1))
        ref = oracle_blocks(cfg, sd, bufs, first, n, level=2)
        y = ops.dframe_attention(m, 2, [None if b is None else b.cuda() for b in bufs], first, n)
        assert maxabs(y, ref) <= TOL, f'case {case}: {B}x256x{H}x{W}, blocks {first}..{first + n - 1}, drop {drop}'
