"""CPU-only checks of the host side: C-ABI surface, config/state-dict logic, pad/crop glue,
multi-process sharding + weight broadcast over gloo (world size 2)."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests.conftest import REPO, GOLDEN
from bde2vid_amd.config import GeneratorConfig, canonical
from bde2vid_amd.weights import (state_dict_spec, formula_state_dict, infer_config, num_parameters,
                                 relative_position_index)
from bde2vid_amd.harness import Croper, chunked


def test_library_exports_every_declared_symbol():
    """include/bde2vid.h <-> libbde2vid.so <-> ctypes table must agree (no compute call: no GPU here)."""
    from bde2vid_amd import _lib
    hdr = open(os.path.join(REPO, 'include', 'bde2vid.h')).read()
    declared = set(re.findall(r'\b(bde_[a-z0-9_]+)\s*\(', hdr))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert L.bde_abi_version() == _lib.ABI_VERSION == 4


def test_create_rejects_bad_config_and_reports_error():
    import ctypes as C
    from bde2vid_amd import _lib
    L = _lib.lib()
    cfg = _lib.make_config(GeneratorConfig(basechannels=8, depths=(2, 0, 2), num_heads=4))
    cfg.ks = 7
    h = C.c_void_p()
    assert L.bde_create(C.byref(cfg), C.byref(h)) == -1
    assert b'ks=7' in L.bde_last_error()
    cfg.ks = 5
    cfg.buffer_index[1] = 1           # query slot must be offset 0
    assert L.bde_create(C.byref(cfg), C.byref(h)) == -1
    cfg.buffer_index[1] = 0
    assert L.bde_create(C.byref(cfg), C.byref(h)) == 0
    # loading a weight with a shape the config contradicts is caught at finalize time, before any HIP call
    w = np.zeros((8, 5, 3, 3), np.float32)
    sh = (C.c_int64 * 4)(*w.shape)
    assert L.bde_load_weight(h, b'generator.head.conv2d.weight', w.ctypes.data_as(C.c_void_p), sh, 4) == 0
    L.bde_destroy(h)


def test_product_has_no_cpu_fallback():
    from bde2vid_amd.model import BDE2VID
    m = BDE2VID(generator=GeneratorConfig(basechannels=8, depths=(2, 0, 2), num_heads=4))
    with pytest.raises(RuntimeError):
        m.to('cpu')
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            m.load_state_dict(formula_state_dict(m.cfg))
    for f in os.listdir(os.path.join(REPO, 'bde2vid_amd')):
        if f.endswith('.py'):
            src = open(os.path.join(REPO, 'bde2vid_amd', f)).read()
            assert 'import oracle' not in src and 'from oracle' not in src, f


def test_state_dict_spec_and_param_count():
    assert num_parameters(canonical()) == 20870433          # SURVEY.md §0, measured on the reference
    cfg = GeneratorConfig(basechannels=16, depths=(1, 2, 3), num_heads=8, buffer_index=(-2, -1, 0, 1, 2), q_idx=2)
    sd = formula_state_dict(cfg)
    got = infer_config(sd)
    assert (got.basechannels, got.depths, got.num_heads, got.buffer_index, got.q_idx, got.ks) == \
        (16, (1, 2, 3), 8, (-2, -1, 0, 1, 2), 2, 5)
    a = formula_state_dict(cfg, 7)
    b = formula_state_dict(cfg, 7)
    assert all(torch.equal(a[k], b[k]) for k in a)


def test_unsupported_configs_raise():
    """What is still rejected: the flags the reference itself cannot run ('no_skip' hands a list to the decoder and raises
    there; any recurrent block but convlstm / convgru fails its assert) and the two the kernels are not built for."""
    for kw in (dict(nwindow_size=(3, 3)), dict(window_size=(8, 8)), dict(skip_type='no_skip'), dict(recurrent_block_type='lstm'),
               dict(norm='GN'), dict(num_output_channels=3), dict(act_net='ELU')):
        with pytest.raises(ValueError):
            GeneratorConfig(**kw).validate()


def test_constructor_variants_are_accepted_and_recovered_from_the_state_dict():
    """ConvGRU / bare encoders / skip_concat / the residual bottleneck / BN / IN: accepted, their state-dict layout is the
    reference's (key by key in tests/golden/var_*.npz through load_state_dict on the GPU box), and infer_config reads every
    flag back from the keys alone."""
    base = dict(basechannels=8, depths=(2, 0, 2), num_heads=4)
    for kw in (dict(recurrent_block_type='convgru'), dict(useRC=False), dict(skip_type='concat'), dict(norm='BN'), dict(norm='IN'),
               dict(depths=(2, 0, 0), num_res_blocks=3)):
        cfg = GeneratorConfig(**{**base, **kw})
        cfg.validate()
        got = infer_config(formula_state_dict(cfg))
        assert (got.recurrent_block_type, got.useRC, got.skip_type, got.norm_kind, got.depths) == \
            (cfg.recurrent_block_type, cfg.useRC, cfg.skip_type, cfg.norm_kind, cfg.depths), kw
        if cfg.bottleneck:
            assert got.num_res_blocks == 3


def test_relative_position_index_shape_and_range():
    idx = relative_position_index(3, 7, 7)
    assert idx.shape == (147, 147) and idx.min() == 0 and idx.max() == 5 * 13 * 13 - 1
    assert idx[0, 0] == idx[100, 100]                        # zero offset -> same table row


def test_croper_matches_reference_fixture():
    ref = json.load(open(os.path.join(GOLDEN, 'croper.json')))
    for key, r in ref.items():
        h, w = map(int, key.split('x'))
        c = Croper(3)
        c.update_params(w, h)
        x = torch.arange(h * w, dtype=torch.float32).reshape(1, 1, h, w)
        p = c.pad(x)
        assert list(p.shape[-2:]) == r['padded_shape']
        assert [c.padding_left, c.padding_right, c.padding_top, c.padding_bottom] == r['pad']
        assert [c.iy0, c.iy1, c.ix0, c.ix1] == r['crop']
        assert torch.equal(c.crop(p), x)


def test_chunked():
    assert [list(c) for c in chunked(list(range(5)), 2)] == [[0, 1], [2, 3], [4]]
    assert [list(c) for c in chunked(list(range(3)), None)] == [[0, 1, 2]]


_WORKER = r'''
import os, sys, torch
sys.path.insert(0, %r)
import torch.distributed as dist
from bde2vid_amd.dist import init_from_env, shard_sequences, broadcast_packed, max_over_ranks, barrier
rank, world, local = init_from_env('gloo')
assert world == 2
mine = shard_sequences(7, rank, world)
flat = torch.arange(1000, dtype=torch.float32) if rank == 0 else torch.zeros(1000)
broadcast_packed(flat, 0)
assert torch.equal(flat, torch.arange(1000, dtype=torch.float32))
all_ = [None, None]
dist.all_gather_object(all_, mine)
assert sorted(all_[0] + all_[1]) == list(range(7)) and not set(all_[0]) & set(all_[1])
m = max_over_ranks(1.0 + rank)
assert m == 2.0
barrier()
dist.destroy_process_group()
open(os.path.join(%r, 'ok%%d' %% rank), 'w').write(repr(mine))
'''


def test_two_rank_gloo_sharding_and_weight_broadcast(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER % (REPO, str(tmp_path)))
    import socket
    with socket.socket() as sk:           # a free port: fixed ports collide with sockets in TIME_WAIT
        sk.bind(('127.0.0.1', 0))
        port = str(sk.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=port)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
                        '--master-addr', '127.0.0.1', '--master-port', port, str(script)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    # (ranks report through files: their stdout interleaves)
    assert (tmp_path / 'ok0').read_text() == '[0, 2, 4, 6]' and (tmp_path / 'ok1').read_text() == '[1, 3, 5]'


_CFG_SRC = """
custom_imports = dict(imports=['model.BDE2VID'], allow_failed_imports=False)
num_bins = 5
base = 16
buffer = [-2, -1, 0, 1, 2]
losses = [dict(type='L1Loss', weight=1.0)]
model = dict(
    type='BDE2VID',
    cpu_cache_length=100,
    generator=dict(
        type='BDE2VIDCrossscalePropogationV5',
        num_bins=num_bins, basechannels=base, num_encoders=3, ks=5, num_res_blocks=2,
        norm=None, recurrent_block_type='convlstm', useRC=True, skip_type='sum',
        activation=dict(type='Sigmoid'), buffer_index=buffer, q_idx=len_half,
        window_size=(7, 7), depths=[1, 2, 3], num_heads=2 * 4, drop_path_rate=0.2,
        losses=losses, loss_inds=None))
train_dataloader = dict(batch_size=2, dataset=dict(type='SomeDataset', root=data_root))
"""


def test_checkpoint_config_is_parsed_without_executing_it(tmp_path):
    from bde2vid_amd.checkpoint import generator_config_from_cfg, parse_config_source, load_model
    src = _CFG_SRC.replace('len_half', '2')
    cfg = generator_config_from_cfg(src)
    assert (cfg.basechannels, cfg.depths, cfg.num_heads, cfg.buffer_index, cfg.q_idx, cfg.activation) == \
        (16, (1, 2, 3), 8, (-2, -1, 0, 1, 2), 2, 'Sigmoid')
    env = parse_config_source(src)
    assert env['model']['cpu_cache_length'] == 100 and 'train_dataloader' not in env   # unresolved names are skipped
    with pytest.raises(ValueError):
        generator_config_from_cfg("model = dict(type='E2VIDRecurrent', generator=dict())")
    with pytest.raises(ValueError):
        generator_config_from_cfg("import os\nmodel = __import__('os').system('true')")   # never executed
    # a file in the reference's checkpoint format; loading needs the GPU, parsing does not
    sd = formula_state_dict(cfg)
    path = tmp_path / 'BDE2VID.pth'
    torch.save({'state_dict': sd, 'meta': {'cfg': src}}, str(path))
    if torch.cuda.is_available():
        m = load_model(str(path))
        assert m.cfg.depths == (1, 2, 3)
    else:
        with pytest.raises(RuntimeError):
            load_model(str(path))


class _Payload:
    """A pickle whose loading would run code (os.system) under the full unpickler."""
    def __init__(self, marker):
        self.marker = marker

    def __reduce__(self):
        import os as _os
        return (_os.system, (f'touch {self.marker}',))


def test_checkpoint_with_code_payload_is_rejected(tmp_path):
    """checkpoint.read_checkpoint never falls back to the full unpickler by itself: a file carrying a __reduce__ payload
    is refused (and the payload does not run) unless the caller passes trust_checkpoint=True."""
    from bde2vid_amd.checkpoint import read_checkpoint
    marker = tmp_path / 'pwned'
    path = tmp_path / 'evil.pth'
    torch.save({'state_dict': {}, 'meta': {'cfg': 'model = dict()', 'extra': _Payload(str(marker))}}, str(path))
    with pytest.raises(RuntimeError, match='trust_checkpoint'):
        read_checkpoint(str(path))
    assert not marker.exists()
    with pytest.raises(FileNotFoundError):
        read_checkpoint(str(tmp_path / 'missing.pth'))
    good = tmp_path / 'good.pth'
    torch.save({'state_dict': {'a': torch.ones(2)}, 'meta': {'cfg': 'model = dict()'}}, str(good))
    assert torch.equal(read_checkpoint(str(good))['state_dict']['a'], torch.ones(2))


def test_split_bf16_conv_shapes_at_the_baseline_resolutions():
    """Host arithmetic of csrc/conv_sb.h (conv_sb_pick / conv_sb_tile_mode), no GPU: which workgroup shape and pixel-tile
    layout each batched convolution of the canonical config takes at the BASELINE resolutions in the default operand format
    (two fp16 terms): 0 = fp32 kernels, 1 = 128 channels x 128 pixels, 2 = 128 x 64, 3 = 64 x 128, 4 = 32 x 256 on 2-D
    tiles, 5 = 64 x 128 on 2-D tiles; second number = tiles per image row, 0 = linear / 2-D."""
    import ctypes as C
    from bde2vid_amd import _lib
    L = _lib.lib()

    def shape(ks, stride, cout, h, w):
        rt = C.c_int32(-9)
        return L.bde_debug_conv_shape(ks, stride, cout, h, w, C.byref(rt)), rt.value

    expect = {
        (184, 240): dict(enc=[(5, -8), (2, 1), (1, 0)], gx=[(1, 1), (1, 0), (1, 0)], dec=[(1, 0), (3, 1), (4, -16)]),
        (264, 352): dict(enc=[(5, -16), (1, 1), (2, 0)], gx=[(1, 0), (1, 0), (1, 0)], dec=[(1, 0), (3, 2), (4, -16)]),
        (480, 640): dict(enc=[(5, -16), (2, 3), (1, 1)], gx=[(2, 5), (1, 0), (1, 0)], dec=[(1, 0), (3, 3), (4, -16)]),
        (720, 1280): dict(enc=[(5, -16), (2, 5), (2, 3)], gx=[(1, 5), (2, 5), (1, 0)], dec=[(2, 5), (3, 5), (4, -16)]),
    }
    chans = (64, 128, 256)
    for (H, W), e in expect.items():
        for l, cout in enumerate(chans):
            hin, win = H >> l, W >> l
            assert shape(5, 2, cout, hin, win) == e['enc'][l], ('enc', H, W, l)
            assert shape(3, 1, 4 * cout, hin // 2, win // 2) == e['gx'][l], ('gx', H, W, l)
        for j, cout in enumerate((128, 64, 32)):
            l = 2 - j
            assert shape(5, 1, cout, (H >> (l + 1)) * 2, (W >> (l + 1)) * 2) == e['dec'][j], ('dec', H, W, j)
    # not a convolution the split-bf16 kernels are built for
    assert shape(7, 1, 128, 64, 64)[0] == 0 and shape(3, 2, 128, 64, 64)[0] == 0


def test_split_operand_formats_host_arithmetic():
    """csrc/split.h as the weight packer applies it (bde_debug_split, no GPU): the two-term fp16 split is numpy's float16
    arithmetic bit for bit (round to nearest even, subnormals kept, |x| >= 65520 -> Inf in the leading term alone); it carries
    x to max(2^-23 |x|, 2^-25); the three-term bf16 split is exact for every fp32 value whose low term does not underflow;
    the packing scale puts the largest magnitude into [2^14, 2^15)."""
    import ctypes as C
    from bde2vid_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.standard_normal(4000).astype(np.float32) * np.float32(10.0) ** rng.integers(-6, 4, 4000).astype(np.float32),
                        np.array([0.0, -0.0, 1.0, -1.0, 65504.0, 65519.9, 65520.0, 1e5, -1e6, 6.1e-5, 5.9e-8, 2.9e-8, 1e-9,
                                  np.inf, -np.inf, np.nan, 0.1, 0.3333333], dtype=np.float32)]).astype(np.float32)
    n = x.size

    def split(terms, scale=1.0):
        out = np.zeros(n * terms, dtype=np.uint16)
        sc = L.bde_debug_split(x.ctypes.data_as(C.POINTER(C.c_float)), n, terms, C.c_float(scale), out.ctypes.data_as(C.POINTER(C.c_uint16)))
        return out.reshape(n, terms), sc

    t2, _ = split(2)
    with np.errstate(over='ignore', invalid='ignore'):
        hi = x.astype(np.float16)
        lo = (x - hi.astype(np.float32)).astype(np.float16)
    special = ~np.isfinite(hi.astype(np.float32))
    lo = np.where(special, np.float16(0), lo)
    assert np.array_equal(t2[:, 0], hi.view(np.uint16)), 'leading fp16 term'
    fin = ~special
    assert np.array_equal(t2[fin, 1], lo.view(np.uint16)[fin]), 'second fp16 term'
    assert np.all(t2[special, 1] == 0)                                   # Inf / NaN / out of range ride in the leading term alone
    rec = t2[:, 0].view(np.float16).astype(np.float64) + t2[:, 1].view(np.float16).astype(np.float64)
    err = np.abs(rec[fin] - x[fin].astype(np.float64))
    assert np.all(err <= np.maximum(2.0 ** -23 * np.abs(x[fin]), 2.0 ** -25)), float(err.max())
    assert np.isinf(rec[x == np.float32(65520.0)]).all() and np.isfinite(rec[x == np.float32(65504.0)]).all()

    t3, one = split(3)
    assert one == 1.0
    f3 = (t3.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    ok = np.isfinite(x) & (np.abs(x) >= 2.0 ** -100)
    assert np.array_equal(f3[ok].sum(axis=1), x[ok].astype(np.float64)), 'three bf16 terms are exact'

    # the packing scale: a power of two, max |w| * scale in [2^14, 2^15); small weights keep 22 bits once scaled
    w = (rng.standard_normal(2000) * 0.02).astype(np.float32)
    out = np.zeros(2 * w.size, dtype=np.uint16)
    sc = L.bde_debug_split(w.ctypes.data_as(C.POINTER(C.c_float)), w.size, 2, C.c_float(1.0), out.ctypes.data_as(C.POINTER(C.c_uint16)))
    assert sc > 0 and np.log2(sc) == np.round(np.log2(sc)) and 2.0 ** 14 <= np.abs(w).max() * sc < 2.0 ** 15
    L.bde_debug_split(w.ctypes.data_as(C.POINTER(C.c_float)), w.size, 2, C.c_float(sc), out.ctypes.data_as(C.POINTER(C.c_uint16)))
    o = out.reshape(-1, 2)
    rec = (o[:, 0].view(np.float16).astype(np.float64) + o[:, 1].view(np.float16).astype(np.float64)) / sc
    big = np.abs(w) >= np.abs(w).max() * 2.0 ** -17
    assert np.all(np.abs(rec[big] - w[big]) <= 2.0 ** -22 * np.abs(w[big]))
