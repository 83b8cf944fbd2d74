"""Pins the CPU oracle (oracle/) to the golden vectors generated from the real reference.

CPU only.  Tolerance 2e-6 max-abs: the oracle issues the same aten ops as the reference,
differences come only from thread-count dependent reduction order.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import bde2vid_oracle as O
from oracle import voxel_oracle
from tests.util import (load_golden, case_from_meta, maxabs, dense_like, voxel_like, voxel_case,
                        E2E_CASES, GOLDEN, CFGA_SAMPLED, LONGT_CASES, VARIANT_CASES, bench_fixture_inputs, assert_sampled)
from bde2vid_amd.config import GeneratorConfig
from bde2vid_amd.weights import formula_state_dict, relative_position_index

TOL = 2e-6
P = O.P


@pytest.mark.parametrize('name', sorted(E2E_CASES))
def test_e2e_matches_reference(name):
    z, meta = load_golden(name)
    cfg, sd, xs = case_from_meta(meta)
    with torch.no_grad():
        ys = O.forward(sd, cfg, [{'events': torch.from_numpy(x)} for x in xs])
    y = torch.stack(ys).numpy()
    assert y.shape == z['out'].shape
    assert maxabs(y, z['out']) <= TOL


@pytest.mark.parametrize('name', sorted(VARIANT_CASES))
def test_constructor_variants_match_reference(name):
    """ConvGRU, plain (non-recurrent) encoders, skip concat, the residual-block bottleneck on buffer slot 0, BN / IN in eval
    mode, and all of them together: whole forwards of the real reference (oracle/gen_golden.py::gen_variants)."""
    z, meta = load_golden(name)
    cfg, sd, xs = case_from_meta(meta)
    with torch.no_grad():
        ys = O.forward(sd, cfg, [{'events': torch.from_numpy(x)} for x in xs])
    y = torch.stack(ys).numpy()
    assert y.shape == z['out'].shape
    assert maxabs(y, z['out']) <= TOL


def test_e2e_config_a_full_size_sampled():
    z, meta = load_golden('e2e_cfgA_184x240')
    cfg, sd, xs = case_from_meta(meta)
    with torch.no_grad():
        ys = O.forward(sd, cfg, [{'events': torch.from_numpy(x)} for x in xs])
    y = torch.stack(ys).numpy()
    s = meta['stride']
    assert maxabs(y[..., ::s, ::s], z['out']) <= TOL
    assert np.allclose(y.mean(axis=(1, 2, 3, 4)), z['mean'], atol=1e-6)
    assert np.allclose(y.std(axis=(1, 2, 3, 4)), z['std'], atol=1e-6)


@pytest.mark.parametrize('name', sorted(CFGA_SAMPLED))
def test_config_a_at_baseline_resolutions(name):
    """Canonical config at every BASELINE.json resolution and at bench.py's T=16 (reference outputs, sampled)."""
    z, meta = load_golden(name)
    cfg, sd, xs = case_from_meta(meta)
    with torch.no_grad():
        y = torch.stack(O.forward(sd, cfg, [{'events': torch.from_numpy(x)} for x in xs])).numpy()
    assert_sampled(y, z, meta, TOL, 1e-6)


def test_bench_workload_fixture():
    """bench.py's exact workload (events -> voxel grids -> pad -> forward) as the reference computed it."""
    z, meta = load_golden('e2e_bench_T16')
    cfg = GeneratorConfig.from_dict(meta['cfg'])
    sd = formula_state_dict(cfg, meta['weight_seed'])
    with torch.no_grad():
        y = torch.stack(O.forward(sd, cfg, [{'events': v} for v in bench_fixture_inputs(meta)])).numpy()
    assert_sampled(y, z, meta, TOL, 1e-6)


@pytest.mark.parametrize('name', sorted(LONGT_CASES))
def test_longer_than_cpu_cache_length(name):
    """T > cpu_cache_length: the reference parks its feature maps on the host (V5.py:102 ...); same arithmetic."""
    z, meta = load_golden(name)
    cfg, sd, xs = case_from_meta(meta)
    assert meta['T'] > meta['cpu_cache_length']
    with torch.no_grad():
        y = torch.stack(O.forward(sd, cfg, [{'events': torch.from_numpy(x)} for x in xs])).numpy()
    assert_sampled(y, z, meta, TOL, 1e-6)


def test_blocks_match_reference():
    z, meta = load_golden('blocks')
    cfg = GeneratorConfig.from_dict(meta['cfg'])
    sd = formula_state_dict(cfg, meta['weight_seed'])
    rel = torch.from_numpy(relative_position_index(cfg.frame_num, 7, 7))
    t = torch.from_numpy
    with torch.no_grad():
        head = O.conv_layer(t(voxel_like((1, 5, 24, 32), 11)), sd[P + 'head.conv2d.weight'],
                            sd[P + 'head.conv2d.bias'], 1, 'relu')
        assert maxabs(head, z['head']) <= TOL
        pre = P + 'forward_encoder.0.'
        seq = [t(dense_like((1, 16, 24, 32), 20 + i)) for i in range(3)]
        state, hs = None, []
        for s in seq:
            x = O.conv_layer(s, sd[pre + 'conv.conv2d.weight'], sd[pre + 'conv.conv2d.bias'], 2, 'relu')
            state = O.convlstm_cell(x, state, sd[pre + 'recurrent_block.Gates.weight'],
                                    sd[pre + 'recurrent_block.Gates.bias'])
            hs.append(state[0])
        assert maxabs(torch.stack(hs), z['rc_h']) <= TOL
        assert maxabs(state[1], z['rc_c']) <= TOL
        enc = O.conv_layer(seq[0], sd[pre + 'conv.conv2d.weight'], sd[pre + 'conv.conv2d.bias'], 2, 'relu')
        assert maxabs(enc, z['enc_conv']) <= TOL
        up = O.upsample_conv_layer(t(dense_like((1, 128, 9, 11), 30)), sd[P + 'decoders.0.1.conv2d.weight'],
                                   sd[P + 'decoders.0.1.conv2d.bias'])
        assert maxabs(up, z['upconv']) <= TOL
        x = t(dense_like((2, 16, 10, 12), 40))
        pi = torch.sigmoid(torch.nn.functional.conv2d(x, sd[P + 'predI.1.weight'], sd[P + 'predI.1.bias']))
        assert maxabs(pi, z['predI']) <= TOL
        buf = [t(dense_like((1, 32, 17, 23), 50 + d)) for d in range(3)]
        a0 = O.dframe_attention(buf, sd, P + 'feat_attns.0.', 2, cfg.num_heads, cfg.q_idx, 7, rel)
        assert maxabs(a0, z['attn_l0']) <= TOL
        fr = torch.stack(buf)
        assert maxabs(O.swin_block(fr, sd, P + 'feat_attns.0.blocks.0.', 8, 1, False, 7, rel),
                      z['swin_plain']) <= TOL
        assert maxabs(O.swin_block(fr, sd, P + 'feat_attns.0.blocks.1.', 8, 1, True, 7, rel),
                      z['swin_dilated']) <= TOL
        buf = [t(dense_like((2, 128, 7, 9), 60 + d)) for d in range(3)]
        a2 = O.dframe_attention(buf, sd, P + 'feat_attns.2.', 3, cfg.num_heads, cfg.q_idx, 7, rel)
        assert maxabs(a2, z['attn_l2']) <= TOL


def test_dilated_partition_coverage():
    """window_reverse(window_partition(x)) on a dilated block keeps covered pixels and zeroes the
    rest; coverage at 28x35 is 81.6 %."""
    x = torch.rand(1, 1, 2, 28, 35) + 1.0
    w = O.window_partition(x, 7, True)
    back = O.window_reverse(w[0], 1, 28, 35, True)
    covered = back != 0
    assert torch.equal(back[covered], x[0][covered])
    assert abs(float(covered.float().mean()) - 0.816) < 1e-3


@pytest.mark.parametrize('name', ['n3', 'n1k_dups', 'n50k', 'edges'])
def test_voxel_matches_reference(name):
    z = np.load(os.path.join(GOLDEN, 'voxel.npz'))
    xs, ys, ts, ps, size = voxel_case(name)
    v = voxel_oracle.events_to_voxel(xs, ys, ts, ps, 5, size)
    assert v.shape == z[name].shape
    # torch's CPU index_put_(accumulate) may reorder the sum for large N: 1e-6, not bit-exact
    assert maxabs(v, z[name]) <= 1e-6
    # conservation: each event contributes p * (w_lo + w_hi) = p in total
    assert abs(float(v.sum()) - float(ps.sum())) < 1e-2


def test_voxel_degenerate_dt_is_nan_like_reference():
    xs = np.array([1, 2, 3], np.float32)
    v = voxel_oracle.events_to_voxel(xs, xs, np.zeros(3, np.float32), np.ones(3, np.float32), 5, (8, 8))
    assert np.isnan(v).any()          # dt == 0 -> NaN weights (event_utils.py:489-490)


def test_croper_matches_reference():
    with open(os.path.join(GOLDEN, 'croper.json')) as f:
        ref = json.load(f)
    for key, r in ref.items():
        h, w = map(int, key.split('x'))
        p = O.crop_params(w, h, 3)
        assert [p['hc'], p['wc']] == [r['hc'], r['wc']]
        assert list(p['pad']) == r['pad']
        assert list(p['crop']) == r['crop']


@pytest.mark.parametrize('name,case', [('rec_small', (4000, 30, 40, 7, 21)), ('rec_davis', (60000, 180, 240, 9, 22))])
def test_recording_voxels_match_reference(name, case):
    """Native event columns (int16 / float64 / bool) through the dataset's casts and the reference binning:
    the restatement is bit-exact, windows with fewer than 3 events are zero grids."""
    z = np.load(os.path.join(GOLDEN, 'voxel_recording.npz'))
    N, H, W, nwin, seed = case
    xs, ys, ts, ps, idx = voxel_oracle.synthetic_recording(N, H, W, nwin, seed)
    assert xs.dtype == np.int16 and ts.dtype == np.float64 and ps.dtype == bool and len(idx) == nwin + 1
    v = voxel_oracle.between_frames_voxels(xs, ys, ts, ps, idx, 5, (H, W))
    assert v.shape == z[name].shape
    assert np.array_equal(v, z[name])
    assert not v[-1].any() and not v[-2].any()          # the 2-event and the empty window


@pytest.mark.parametrize('name', ['n3', 'n1k_dups', 'n50k', 'edges'])
def test_voxel_indexput_form_matches_reference(name):
    """The operation-for-operation form bench.py times as the host baseline of the voxel path."""
    z = np.load(os.path.join(GOLDEN, 'voxel.npz'))
    xs, ys, ts, ps, size = voxel_case(name)
    v = voxel_oracle.events_to_voxel_indexput(*[torch.from_numpy(a) for a in (xs, ys, ts, ps)], 5, size)
    assert maxabs(v, z[name]) <= 1e-6


def test_recording_index_logic_matches_reference():
    """oracle/recording_oracle.py against what the reference's own DynamicH5Dataset returned (tests/golden/rec_dataset.npz)."""
    from oracle import recording_oracle as R
    from bde2vid_amd.synth import synthetic_recording_with_frames
    z = np.load(os.path.join(GOLDEN, 'rec_dataset.npz'))
    meta = json.loads(str(z['meta']))
    rec = synthetic_recording_with_frames(**meta['recording'])
    ts = rec['ts']
    assert [R.find_ts_index(ts, t) for t in z['probes']] == z['find_ts_index'].tolist()
    assert R.frame_indices_from_attrs(rec['event_idx']) == z['between_frames_indices'].tolist()
    assert R.frame_indices_from_timestamps(ts, rec['frame_ts']) == z['base_frame_indices'].tolist()
    dur = ts[-1] - ts[0]
    for name, vm in meta['methods'].items():
        L = R.dataset_length(vm, rec['num_events'], rec['num_imgs'], dur)
        assert L == int(z[name + '_len'])
    vm = meta['methods']['t_seconds']
    assert R.timeblock_indices(ts, vm, int(z['t_seconds_len'])) == z['t_seconds_indices'].tolist()
    vm = meta['methods']['k_events']
    assert R.k_indices(vm, int(z['k_events_len'])) == z['k_events_indices'].tolist()
    # items: the voxel grids of the reference's __getitem__ against the numpy restatement of the binning
    for name in meta['methods']:
        idx = z[name + '_indices']
        for k in range(z[name + '_events'].shape[0]):
            i0, i1 = int(idx[k][0]), int(idx[k][1])
            g = voxel_oracle.between_frames_voxels(rec['xs'], rec['ys'], ts, rec['ps'], [i0, i1], 5, (36, 48))[0]
            assert maxabs(g, z[name + '_events'][k]) <= 1e-6
