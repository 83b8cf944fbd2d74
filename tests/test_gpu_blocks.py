"""GPU parity, block by block: HIP kernels (through the C ABI) vs the golden vectors of the
reference and vs the CPU oracle on further seeded shapes.  fp32 everywhere; the tolerance
(1e-4 max-abs on O(1) activations, 10x tighter than north_star's 1e-3) covers only the different
accumulation order of the MFMA contraction and the folded LayerNorm."""
import numpy as np
import pytest
import torch

from tests.util import load_golden, maxabs, dense_like, voxel_like
from bde2vid_amd.config import GeneratorConfig
from bde2vid_amd.weights import formula_state_dict, relative_position_index

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope='module')
def blocks():
    from bde2vid_amd.model import build_model
    z, meta = load_golden('blocks')
    cfg = GeneratorConfig.from_dict(meta['cfg'])
    sd = formula_state_dict(cfg, meta['weight_seed'])
    m = build_model(cfg, sd, 'cuda:0')
    return z, cfg, sd, m


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_head_conv(blocks):
    from bde2vid_amd import ops
    z, cfg, sd, m = blocks
    y = ops.head(m, dev(voxel_like((1, 5, 24, 32), 11)))
    assert maxabs(y, z['head']) <= TOL


def test_encoder_conv_stride2(blocks):
    from bde2vid_amd import ops
    z, cfg, sd, m = blocks
    y = ops.encoder_conv(m, 0, 0, dev(dense_like((1, 16, 24, 32), 20)))
    assert maxabs(y, z['enc_conv']) <= TOL


def test_recurrent_conv_three_steps(blocks):
    from bde2vid_amd import ops
    z, cfg, sd, m = blocks
    x = dev(np.stack([dense_like((1, 16, 24, 32), 20 + t) for t in range(3)]))
    h, c = ops.recurrent_conv(m, 0, 0, x)
    assert maxabs(h, z['rc_h']) <= TOL
    assert maxabs(c, z['rc_c']) <= TOL


def test_recurrent_conv_backward_direction_vs_oracle(blocks):
    from bde2vid_amd import ops
    from oracle import bde2vid_oracle as O
    z, cfg, sd, m = blocks
    xs = [torch.from_numpy(dense_like((2, 16, 16, 20), 70 + t)) for t in range(4)]
    pre = O.P + 'backward_encoder.0.'
    state, ref = None, [None] * 4
    for t in range(3, -1, -1):
        x = O.conv_layer(xs[t], sd[pre + 'conv.conv2d.weight'], sd[pre + 'conv.conv2d.bias'], 2, 'relu')
        state = O.convlstm_cell(x, state, sd[pre + 'recurrent_block.Gates.weight'],
                                sd[pre + 'recurrent_block.Gates.bias'])
        ref[t] = state[0]
    h, c = ops.recurrent_conv(m, 0, 1, torch.stack(xs).cuda())
    assert maxabs(h, torch.stack(ref)) <= TOL
    assert maxabs(c, state[1]) <= TOL


def test_upsample_conv(blocks):
    from bde2vid_amd import ops
    z, cfg, sd, m = blocks
    y = ops.decoder(m, 0, dev(dense_like((1, 128, 9, 11), 30)))
    assert maxabs(y, z['upconv']) <= TOL


def test_upsample_conv_with_skip_vs_oracle(blocks):
    from bde2vid_amd import ops
    from oracle import bde2vid_oracle as O
    z, cfg, sd, m = blocks
    a = torch.from_numpy(dense_like((2, 32, 13, 10), 31))
    b = torch.from_numpy(dense_like((2, 32, 13, 10), 32))
    ref = O.upsample_conv_layer(a + b, sd[O.P + 'decoders.2.1.conv2d.weight'], sd[O.P + 'decoders.2.1.conv2d.bias'])
    y = ops.decoder(m, 2, a.cuda(), b.cuda())
    assert maxabs(y, ref) <= TOL


def test_pred(blocks):
    from bde2vid_amd import ops
    z, cfg, sd, m = blocks
    y = ops.pred(m, dev(dense_like((2, 16, 10, 12), 40)))
    assert maxabs(y, z['predI']) <= TOL


def test_swin_block_plain_and_dilated(blocks):
    from bde2vid_amd import ops
    z, cfg, sd, m = blocks
    buf = [dev(dense_like((1, 32, 17, 23), 50 + d)) for d in range(3)]
    assert maxabs(ops.dframe_attention(m, 0, buf, 0, 1), z['swin_plain']) <= TOL
    assert maxabs(ops.dframe_attention(m, 0, buf, 1, 1), z['swin_dilated']) <= TOL


def test_dframe_attention_level0(blocks):
    from bde2vid_amd import ops
    z, cfg, sd, m = blocks
    buf = [dev(dense_like((1, 32, 17, 23), 50 + d)) for d in range(3)]
    assert maxabs(ops.dframe_attention(m, 0, buf), z['attn_l0']) <= TOL


@pytest.mark.parametrize('level,shape,seed,key', [(0, (1, 32, 17, 23), 50, 'attn_l0'), (2, (2, 128, 7, 9), 60, 'attn_l2')])
def test_dframe_attention_fused_token_kernel(blocks, level, shape, seed, key):
    """Same goldens through the fused proj+MLP+next-qkv kernel (token_fused.h), forced on."""
    from bde2vid_amd import ops
    z, cfg, sd, m = blocks
    buf = [dev(dense_like(shape, seed + d)) for d in range(3)]
    m.set_tuning('fused_min_tiles', 1)
    try:
        y = ops.dframe_attention(m, level, buf)
        if level == 0:
            y1 = ops.dframe_attention(m, 0, buf, 1, 1)
    finally:
        m.set_tuning('fused_min_tiles', 160)
    assert maxabs(y, z[key]) <= TOL
    if level == 0:
        assert maxabs(y1, z['swin_dilated']) <= TOL


def test_dframe_attention_level2_batch2(blocks):
    from bde2vid_amd import ops
    z, cfg, sd, m = blocks
    buf = [dev(dense_like((2, 128, 7, 9), 60 + d)) for d in range(3)]
    assert maxabs(ops.dframe_attention(m, 2, buf), z['attn_l2']) <= TOL


def test_dframe_attention_zero_frames_vs_oracle(blocks):
    """Temporal out-of-range slots are all-zero frames (V5.py:152-161): LayerNorm(0) = beta tokens."""
    from bde2vid_amd import ops
    from oracle import bde2vid_oracle as O
    z, cfg, sd, m = blocks
    q = torch.from_numpy(dense_like((1, 32, 15, 16), 81))
    nxt = torch.from_numpy(dense_like((1, 32, 15, 16), 82))
    rel = torch.from_numpy(relative_position_index(3, 7, 7))
    ref = O.dframe_attention([torch.zeros_like(q), q, nxt], sd, O.P + 'feat_attns.0.', 2, cfg.num_heads, 1, 7, rel)
    y = ops.dframe_attention(m, 0, [None, q.cuda(), nxt.cuda()])
    assert maxabs(y, ref) <= TOL


@pytest.mark.parametrize('name', ['n3', 'n1k_dups', 'n50k', 'edges'])
def test_voxel_scatter(name):
    import os
    from bde2vid_amd.events import events_to_voxel_torch
    from tests.util import voxel_case, GOLDEN
    z = np.load(os.path.join(GOLDEN, 'voxel.npz'))
    xs, ys, ts, ps, size = voxel_case(name)
    v = events_to_voxel_torch(torch.from_numpy(xs).cuda(), torch.from_numpy(ys).cuda(), torch.from_numpy(ts).cuda(),
                              torch.from_numpy(ps).cuda(), 5, sensor_size=size)
    # per-event weights are bit-identical; float atomics reorder the per-pixel sum:
    # <= 300 events/pixel of |w| <= 1  ->  1e-4 absolute
    assert maxabs(v, z[name]) <= 1e-4


def test_voxel_scatter_batch_and_oob():
    from bde2vid_amd.events import events_to_voxel_torch, events_to_voxel_batch
    from oracle import voxel_oracle
    packs = [voxel_oracle.synthetic_events(n, 40, 50, 90 + i) for i, n in enumerate((500, 3, 2000))]
    off = np.cumsum([0] + [len(p[0]) for p in packs])
    cat = [np.concatenate([p[k] for p in packs]) for k in range(4)]
    g = events_to_voxel_batch(*[torch.from_numpy(c) for c in cat], off, 5, sensor_size=(40, 50), device='cuda')
    for i, p in enumerate(packs):
        assert maxabs(g[i], voxel_oracle.events_to_voxel(*p, 5, (40, 50))) <= 1e-4
    xs, ys, ts, ps = packs[0]
    xs = xs.copy()
    xs[10] = 50.0      # one column past the sensor
    with pytest.raises(IndexError):
        events_to_voxel_torch(torch.from_numpy(xs), torch.from_numpy(ys), torch.from_numpy(ts),
                              torch.from_numpy(ps), 5, device='cuda', sensor_size=(40, 50))


@pytest.mark.parametrize('name,case', [('rec_small', (4000, 30, 40, 7, 21)), ('rec_davis', (60000, 180, 240, 9, 22))])
def test_voxel_windows_from_native_columns(name, case):
    """bde_voxelize_events: int16 / float64 / bool columns + window boundaries -> grids, against the
    reference's golden grids (same tolerance as the float32 scatter: only the per-pixel sum order differs)."""
    import os
    from bde2vid_amd.events import events_to_voxel_windows
    from oracle import voxel_oracle
    from tests.util import GOLDEN
    z = np.load(os.path.join(GOLDEN, 'voxel_recording.npz'))
    N, H, W, nwin, seed = case
    xs, ys, ts, ps, idx = voxel_oracle.synthetic_recording(N, H, W, nwin, seed)
    g = events_to_voxel_windows(xs, ys, ts, ps, idx, 5, sensor_size=(H, W))
    assert g.shape == z[name].shape
    assert maxabs(g, z[name]) <= 1e-4
    assert float(g[-1].abs().max()) == 0.0 and float(g[-2].abs().max()) == 0.0
    # device-resident torch columns give the same grids; a coordinate outside the sensor raises
    g2 = events_to_voxel_windows(torch.from_numpy(xs).cuda(), torch.from_numpy(ys).cuda(), torch.from_numpy(ts).cuda(),
                                 torch.from_numpy(ps).cuda(), idx, 5, sensor_size=(H, W))
    assert maxabs(g2, z[name]) <= 1e-4
    bad = xs.copy()
    bad[5] = W
    with pytest.raises(IndexError):
        events_to_voxel_windows(bad, ys, ts, ps, idx, 5, sensor_size=(H, W))


@pytest.mark.parametrize('hc8', [0, 1])
def test_recurrent_step_workgroup_variants(blocks, hc8):
    """The recurrent step with 16-channel workgroups and with 8-channel ones (two gates stacked per MFMA tile,
    csrc/lstm16.h) against the same goldens; the library picks between them per level by launch geometry."""
    from bde2vid_amd import ops
    z, cfg, sd, m = blocks
    x = dev(np.stack([dense_like((1, 16, 24, 32), 20 + t) for t in range(3)]))
    m.set_tuning('lstm_hc8', hc8)
    try:
        h, c = ops.recurrent_conv(m, 0, 0, x)
    finally:
        m.set_tuning('lstm_hc8', -1)
    assert maxabs(h, z['rc_h']) <= TOL
    assert maxabs(c, z['rc_c']) <= TOL
