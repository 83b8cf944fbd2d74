"""Generate tests/golden/*.npz by running the REAL reference (build container only).

TEST INFRASTRUCTURE ONLY.  Usage:  python oracle/gen_golden.py
Imports /root/reference through oracle/ref_import.py (stub mmengine/mmcv/timm,
SURVEY.md §8c), fills it with the formula weights of bde2vid_amd/weights.py, runs it
on seeded inputs and stores ONLY: the config, the seeds and the reference's outputs.
No weights and no reference source are stored; inputs are regenerated from the seeds
by `golden_inputs` (shared with the tests).
"""
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from bde2vid_amd.config import GeneratorConfig, canonical  # noqa: E402
from oracle import ref_import, voxel_oracle  # noqa: E402

OUT = os.path.join(REPO, 'tests', 'golden')
WEIGHT_SEED = 4   # chosen so every golden case has a well-spread (unsaturated) sigmoid output


# ---------------------------------------------------------------- shared with tests
def voxel_like(shape, seed):
    """Direct-voxel synthetic input: N(0,1)*0.5 with ~70 % zeros."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(shape, dtype=np.float32) * np.float32(0.5)
    keep = rng.random(shape, dtype=np.float32) > np.float32(0.7)
    return (x * keep).astype(np.float32)


def dense_like(shape, seed):
    return np.random.default_rng(seed).standard_normal(shape, dtype=np.float32)


def golden_inputs(T, B, num_bins, H, W, seed):
    return [voxel_like((B, num_bins, H, W), seed + t) for t in range(T)]


E2E_CASES = {
    # name: (config kwargs, H, W, T, B, input seed)
    'e2e_tiny': (dict(basechannels=8, depths=(2, 0, 2), num_heads=4), 56, 64, 5, 1, 100),
    'e2e_buf5': (dict(basechannels=16, depths=(1, 2, 3), num_heads=8,
                      buffer_index=(-2, -1, 0, 1, 2), q_idx=2), 64, 80, 6, 1, 200),
    'e2e_T1': (dict(basechannels=8, depths=(2, 0, 2), num_heads=4), 72, 56, 1, 1, 300),
    'e2e_T2': (dict(basechannels=8, depths=(2, 0, 2), num_heads=4), 72, 56, 2, 1, 310),
    'e2e_B2': (dict(basechannels=8, depths=(2, 0, 2), num_heads=4), 56, 72, 3, 2, 400),
    'e2e_cfgA_small': (dict(), 56, 64, 3, 1, 500),          # canonical channels/heads/depths
    'e2e_ks3': (dict(basechannels=8, ks=3, depths=(1, 0, 2), num_heads=2), 64, 56, 3, 1, 600),
}
# Constructor variants the reference accepts besides the assumed canonical flags (V5.py:19-23): every one a whole forward of the
# REAL reference.  gen_variants() writes them; the name says what differs from the e2e_tiny configuration.
_TINY = dict(basechannels=8, depths=(2, 0, 2), num_heads=4)
VARIANT_CASES = {
    'var_convgru': (dict(_TINY, recurrent_block_type='convgru'), 56, 64, 4, 1, 900),
    'var_norc': (dict(_TINY, useRC=False), 56, 64, 3, 1, 910),
    'var_concat': (dict(_TINY, skip_type='concat'), 56, 64, 3, 2, 920),
    'var_bottleneck': (dict(_TINY, depths=(2, 0, 0), num_res_blocks=2), 56, 64, 4, 1, 930),
    'var_bottleneck_fwd': (dict(_TINY, depths=(1, 0, 0), num_res_blocks=1, buffer_index=(1, 0, -1), q_idx=1), 56, 64, 4, 1, 935),
    'var_bn': (dict(_TINY, norm='BN'), 56, 64, 3, 1, 940),
    'var_in': (dict(_TINY, norm='IN'), 56, 64, 3, 1, 950),
    'var_all': (dict(basechannels=16, depths=(1, 0, 0), num_heads=8, recurrent_block_type='convgru', skip_type='concat', norm='BN',
                     num_res_blocks=1, ks=3), 64, 72, 3, 1, 960),
}
CFGA_FULL = ('e2e_cfgA_184x240', dict(), 184, 240, 4, 1, 700, 4)   # stored at pixel stride 4
# Canonical config at every BASELINE.json resolution and at bench.py's exact workload (T=16): stored as pixels at
# a stride plus per-frame mean/std, like CFGA_FULL.  name: (H, W, T, B, input seed, stride)
CFGA_SAMPLED = {
    'e2e_cfgA_184x240_T16': (184, 240, 16, 1, 7, 4),       # bench.py / cpu_baseline inputs: golden_inputs(16, 1, 5, 184, 240, 7)
    'e2e_cfgA_264x352_B2': (264, 352, 3, 2, 710, 4),       # DAVIS346 260x346 padded (BASELINE config 4)
    'e2e_cfgA_480x640': (480, 640, 2, 1, 720, 8),          # VGA (BASELINE config 3)
    'e2e_cfgA_720x1280': (720, 1280, 2, 1, 730, 8),        # HD (BASELINE config 5)
}
# T > cpu_cache_length: the reference then parks every feature map on the host (V5.py:102,116,126-133,...); same
# arithmetic, different code path.  name: (config kwargs, H, W, T, B, seed, cpu_cache_length, stride)
LONGT_CASES = {
    'e2e_longT_cache3': (dict(basechannels=8, depths=(2, 0, 2), num_heads=4), 56, 64, 7, 1, 800, 3, 1),
    'e2e_longT_104': (dict(basechannels=8, depths=(1, 0, 1), num_heads=4), 56, 64, 104, 1, 810, 100, 2),
}


def _cfg(kw):
    return GeneratorConfig(**kw)


def gen_e2e():
    for name, (kw, H, W, T, B, seed) in E2E_CASES.items():
        cfg = _cfg(kw)
        model = ref_import.build_reference_model(cfg, WEIGHT_SEED)
        xs = golden_inputs(T, B, cfg.num_bins, H, W, seed)
        with torch.no_grad():
            ys = model([{'events': torch.from_numpy(x)} for x in xs])
        y = torch.stack(ys).numpy()
        np.savez_compressed(os.path.join(OUT, name + '.npz'), out=y,
                            meta=json.dumps(dict(cfg=cfg.to_dict(), H=H, W=W, T=T, B=B, seed=seed,
                                                 weight_seed=WEIGHT_SEED)))
        print(name, y.shape, float(y.mean()), float(y.std()))
    name, kw, H, W, T, B, seed, stride = CFGA_FULL
    cfg = _cfg(kw)
    model = ref_import.build_reference_model(cfg, WEIGHT_SEED)
    xs = golden_inputs(T, B, cfg.num_bins, H, W, seed)
    with torch.no_grad():
        ys = model([{'events': torch.from_numpy(x)} for x in xs])
    y = torch.stack(ys).numpy()
    np.savez_compressed(os.path.join(OUT, name + '.npz'), out=y[..., ::stride, ::stride],
                        mean=y.mean(axis=(1, 2, 3, 4)), std=y.std(axis=(1, 2, 3, 4)),
                        meta=json.dumps(dict(cfg=cfg.to_dict(), H=H, W=W, T=T, B=B, seed=seed,
                                             weight_seed=WEIGHT_SEED, stride=stride)))
    print(name, y.shape, float(y.mean()), float(y.std()))


def gen_variants():
    for name, (kw, H, W, T, B, seed) in VARIANT_CASES.items():
        cfg = _cfg(kw)
        model = ref_import.build_reference_model(cfg, WEIGHT_SEED)
        xs = golden_inputs(T, B, cfg.num_bins, H, W, seed)
        with torch.no_grad():
            ys = model([{'events': torch.from_numpy(x)} for x in xs])
        y = torch.stack(ys).numpy()
        np.savez_compressed(os.path.join(OUT, name + '.npz'), out=y,
                            meta=json.dumps(dict(cfg=cfg.to_dict(), H=H, W=W, T=T, B=B, seed=seed, weight_seed=WEIGHT_SEED)))
        print(name, y.shape, float(y.mean()), float(y.std()), flush=True)


def _store_sampled(name, cfg, y, stride, H, W, T, B, seed, **extra):
    np.savez_compressed(os.path.join(OUT, name + '.npz'), out=y[..., ::stride, ::stride],
                        mean=y.mean(axis=(1, 2, 3, 4)), std=y.std(axis=(1, 2, 3, 4)),
                        meta=json.dumps(dict(cfg=cfg.to_dict(), H=H, W=W, T=T, B=B, seed=seed,
                                             weight_seed=WEIGHT_SEED, stride=stride, **extra)))
    print(name, y.shape, float(y.mean()), float(y.std()), flush=True)


def gen_cfgA_sampled(only=None):
    cfg = canonical()
    model = ref_import.build_reference_model(cfg, WEIGHT_SEED)
    for name, (H, W, T, B, seed, stride) in CFGA_SAMPLED.items():
        if only and name not in only:
            continue
        xs = golden_inputs(T, B, cfg.num_bins, H, W, seed)
        with torch.no_grad():
            ys = model([{'events': torch.from_numpy(x)} for x in xs])
        _store_sampled(name, cfg, torch.stack(ys).numpy(), stride, H, W, T, B, seed)


BENCH_FIXTURE = ('e2e_bench_T16', 180, 240, 16, 1000, 4)   # name, sensor H, W, T, event seed of frame 0, stored stride


def gen_bench_fixture():
    """bench.py's exact workload through the reference's own pipeline: synthetic events (bde2vid_amd/synth.py, seeds
    1000+t) -> events_to_voxel_torch (h5_dataset.py:357) -> Croper(3).pad (eval_models_seq.py:195-207) -> model."""
    from bde2vid_amd.synth import synthetic_events
    name, sh, sw, T, seed0, stride = BENCH_FIXTURE
    EU = ref_import.import_event_utils()
    Croper = ref_import.import_croper()
    cfg = canonical()
    model = ref_import.build_reference_model(cfg, WEIGHT_SEED)
    crop = Croper(cfg.num_encoders)
    inputs = []
    for t in range(T):
        xs, ys, ts, ps = synthetic_events(sh * sw // 2, sh, sw, seed0 + t)
        v = EU.events_to_voxel_torch(torch.from_numpy(xs), torch.from_numpy(ys), torch.from_numpy(ts),
                                     torch.from_numpy(ps), cfg.num_bins, sensor_size=(sh, sw))
        inputs.append({'events': crop.pad(v[None])})
    H, W = inputs[0]['events'].shape[-2:]
    with torch.no_grad():
        ys_ = model(inputs)
    _store_sampled(name, cfg, torch.stack(ys_).numpy(), stride, int(H), int(W), T, 1, seed0,
                   sensor=[sh, sw], events_per_frame=sh * sw // 2)


# BASELINE configs 3 and 5 at FULL size (workspaces pass 2^31 elements there): the same pipeline as gen_bench_fixture.
# Batch element b of frame t is the grid of the events with seed seed0 + b*T + t (bde2vid_amd/workload.py::bench_voxels).
# name: (sensor H, W, T, B, event seed of frame 0, stored stride)
BENCH_FULLSIZE = {
    'e2e_bench_480x640_T32_B4': (480, 640, 32, 4, 1000, 16),
    'e2e_bench_720x1280_T64': (720, 1280, 64, 1, 1000, 16),
}


def gen_bench_fullsize(only=None):
    from bde2vid_amd.synth import synthetic_events
    import time
    EU = ref_import.import_event_utils()
    Croper = ref_import.import_croper()
    cfg = canonical()
    for name, (sh, sw, T, B, seed0, stride) in BENCH_FULLSIZE.items():
        if only and name not in only:
            continue
        model = ref_import.build_reference_model(cfg, WEIGHT_SEED)
        crop = Croper(cfg.num_encoders)
        inputs = []
        for t in range(T):
            vs = []
            for b in range(B):
                xs, ys, ts, ps = synthetic_events(sh * sw // 2, sh, sw, seed0 + b * T + t)
                vs.append(EU.events_to_voxel_torch(torch.from_numpy(xs), torch.from_numpy(ys), torch.from_numpy(ts),
                                                   torch.from_numpy(ps), cfg.num_bins, sensor_size=(sh, sw)))
            inputs.append({'events': crop.pad(torch.stack(vs))})
        H, W = inputs[0]['events'].shape[-2:]
        t0 = time.time()
        print(name, 'running the reference ...', flush=True)
        with torch.no_grad():
            ys_ = model(inputs)
        print(name, f'{time.time() - t0:.0f} s', flush=True)
        _store_sampled(name, cfg, torch.stack(ys_).numpy(), stride, int(H), int(W), T, B, seed0,
                       sensor=[sh, sw], events_per_frame=sh * sw // 2)
        del ys_, inputs, model


def gen_longT():
    for name, (kw, H, W, T, B, seed, ccl, stride) in LONGT_CASES.items():
        cfg = _cfg(kw)
        model = ref_import.build_reference_model(cfg, WEIGHT_SEED, cpu_cache_length=ccl)
        xs = golden_inputs(T, B, cfg.num_bins, H, W, seed)
        with torch.no_grad():
            ys = model([{'events': torch.from_numpy(x)} for x in xs])
        _store_sampled(name, cfg, torch.stack(ys).numpy(), stride, H, W, T, B, seed, cpu_cache_length=ccl)


def gen_blocks():
    """Each building block in isolation, called on the reference's own module objects."""
    cfg = _cfg(dict(basechannels=16, depths=(2, 0, 3), num_heads=8))
    model = ref_import.build_reference_model(cfg, WEIGHT_SEED)
    g = model.generator
    out = {}
    with torch.no_grad():
        # ConvLayer stride 1 (head)  -- submodules.py:105-114
        x = torch.from_numpy(voxel_like((1, 5, 24, 32), 11))
        out['head'] = g.head(x).numpy()
        # RecurrentConv: ConvLayer stride 2 + ConvLSTM, 3 consecutive steps from zero state
        enc = g.forward_encoder[0]
        enc.state = None
        seq = [torch.from_numpy(dense_like((1, 16, 24, 32), 20 + t)) for t in range(3)]
        out['rc_h'] = np.stack([enc(s).numpy() for s in seq])
        out['rc_c'] = enc.state[1].numpy()
        enc.state = None
        # bare ConvLayer stride 2
        out['enc_conv'] = enc.conv(seq[0]).numpy()
        # UpsampleConvLayer (decoder 0: 128 -> 64 channels at bc=16)
        x = torch.from_numpy(dense_like((1, 128, 9, 11), 30))
        out['upconv'] = g.decoders[0](x).numpy()
        # predI + sigmoid
        x = torch.from_numpy(dense_like((2, 16, 10, 12), 40))
        out['predI'] = g.activation(g.predI(x)).numpy()
        # DFrameAttention level 0 (C=32, depth 2: plain + dilated), non-multiple-of-7 map
        buf = [torch.from_numpy(dense_like((1, 32, 17, 23), 50 + d)) for d in range(3)]
        out['attn_l0'] = g.feat_attns[0](list(buf)).numpy()
        blk0, blk1 = g.feat_attns[0].blocks[0], g.feat_attns[0].blocks[1]
        out['swin_plain'] = blk0(torch.stack(buf)).numpy()
        out['swin_dilated'] = blk1(torch.stack(buf)).numpy()
        out['swin_plain_part1'] = blk0.forward_part1(torch.stack(buf)).numpy()
        out['swin_dilated_part1'] = blk1.forward_part1(torch.stack(buf)).numpy()
        # level 2 (C=128, depth 3), B=2, map exactly 7 high
        buf = [torch.from_numpy(dense_like((2, 128, 7, 9), 60 + d)) for d in range(3)]
        out['attn_l2'] = g.feat_attns[2](list(buf)).numpy()
    np.savez_compressed(os.path.join(OUT, 'blocks.npz'), **out,
                        meta=json.dumps(dict(cfg=cfg.to_dict(), weight_seed=WEIGHT_SEED)))
    print('blocks', {k: v.shape for k, v in out.items()})


VOXEL_CASES = {
    # name: (N, H, W, seed, special)
    'n3': (3, 20, 30, 1, None),
    'n1k_dups': (1000, 60, 80, 2, 'dups'),
    'n50k': (50000, 90, 120, 3, None),
    'edges': (64, 16, 16, 4, 'edges'),
}


def voxel_case(name):
    N, H, W, seed, special = VOXEL_CASES[name]
    xs, ys, ts, ps = voxel_oracle.synthetic_events(N, H, W, seed)
    if special == 'dups':          # 300 events on one pixel
        xs[100:400] = 7.0
        ys[100:400] = 9.0
    if special == 'edges':         # exact t0 / t_last duplicates, fractional coords, bin-centre times
        ts[:4] = 0.0
        ts[-4:] = ts[-1]
        ts[10:15] = ts[-1] * np.array([0.25, 0.5, 0.75, 0.125, 0.999], dtype=np.float32)
        ts = np.sort(ts)
        xs[20:30] += np.float32(0.75)
        ys[20:30] += np.float32(0.25)
        xs = np.minimum(xs, np.float32(W - 0.01)).astype(np.float32)
        ys = np.minimum(ys, np.float32(H - 0.01)).astype(np.float32)
    return xs, ys, ts, ps, (H, W)


def gen_voxels():
    EU = ref_import.import_event_utils()
    out = {}
    for name in VOXEL_CASES:
        xs, ys, ts, ps, size = voxel_case(name)
        v = EU.events_to_voxel_torch(torch.from_numpy(xs), torch.from_numpy(ys), torch.from_numpy(ts),
                                     torch.from_numpy(ps), 5, sensor_size=size)
        out[name] = v.numpy()
    np.savez_compressed(os.path.join(OUT, 'voxel.npz'), **out)
    print('voxel', {k: v.shape for k, v in out.items()})


RECORDING_CASES = {
    # name: (N, H, W, nwin, seed)
    'rec_small': (4000, 30, 40, 7, 21),
    'rec_davis': (60000, 180, 240, 9, 22),
}


def gen_recording_voxels():
    """Native event columns (int16 / float64 / bool) -> one voxel grid per between-frames window.
    The casts are the reference's own lines (data_loader/h5_dataset.py:219-225 with get_events :410-415:
    h5py, which that module imports, is not installed, so the four lines are applied here verbatim in
    meaning); the binning is the reference's events_to_voxel_torch, imported and called."""
    EU = ref_import.import_event_utils()
    out = {}
    for name, (N, H, W, nwin, seed) in RECORDING_CASES.items():
        xs, ys, ts, ps, idx = voxel_oracle.synthetic_recording(N, H, W, nwin, seed)
        psf = ps * 2.0 - 1.0                                                  # :414
        grids = np.zeros((len(idx) - 1, 5, H, W), dtype=np.float32)
        for w in range(len(idx) - 1):
            i0, i1 = int(idx[w]), int(idx[w + 1])
            if i1 - i0 < 3:                                                   # :219-220
                continue
            x = torch.from_numpy(xs[i0:i1].astype(np.float32))                # :222
            y = torch.from_numpy(ys[i0:i1].astype(np.float32))                # :223
            t = torch.from_numpy((ts[i0:i1] - ts[i0]).astype(np.float32))     # :224
            p = torch.from_numpy(psf[i0:i1].astype(np.float32))               # :225
            grids[w] = EU.events_to_voxel_torch(x, y, t, p, 5, sensor_size=(H, W)).numpy()   # :357
        out[name] = grids
    np.savez_compressed(os.path.join(OUT, 'voxel_recording.npz'), **out)
    print('voxel_recording', {k: v.shape for k, v in out.items()})


REC_DATASET = dict(n=6000, height=36, width=48, nframes=12, seed=31)
REC_METHODS = {
    'between_frames': {'method': 'between_frames'},
    't_seconds': {'method': 't_seconds', 't': 0.06, 'sliding_window_t': 0.02},
    'k_events': {'method': 'k_events', 'k': 700, 'sliding_window_w': 200},
}


def gen_recording_dataset():
    """The reference's own DynamicH5Dataset (data_loader/h5_dataset.py:398-455 on BaseVoxelDataset :45-396) run on a synthetic
    recording held by oracle/fake_h5.File in place of h5py.File (h5py is not installed): event index tables of the three
    voxel methods, find_ts_index on probe timestamps, and complete items (voxel grid, frame, dt, timestamp)."""
    from bde2vid_amd.synth import synthetic_recording_with_frames
    from oracle import fake_h5
    D = ref_import.import_h5_dataset()
    rec = synthetic_recording_with_frames(**REC_DATASET)
    D.h5py.File = lambda path, mode='r': fake_h5.File(rec)
    out = {}
    probes = np.concatenate([rec['ts'][[0, 1, 7, 2999, 5999]], rec['frame_ts'], [rec['ts'][0] - 1.0, rec['ts'][-1] + 1.0],
                             (rec['ts'][100:110] + rec['ts'][101:111]) / 2])
    for name, vm in REC_METHODS.items():
        ds = D.DynamicH5Dataset('synthetic.h5', num_bins=5, voxel_method=dict(vm))
        out[name + '_indices'] = np.asarray(ds.event_indices, dtype=np.int64)
        out[name + '_len'] = np.int64(len(ds))
        if name == 'between_frames':
            out['find_ts_index'] = np.asarray([ds.find_ts_index(float(t)) for t in probes], dtype=np.int64)
            out['base_frame_indices'] = np.asarray(D.BaseVoxelDataset.compute_frame_indices(ds), dtype=np.int64)
        idxs = list(range(min(len(ds), 6)))
        items = [ds[i] for i in idxs]
        out[name + '_events'] = np.stack([it['events'].numpy() for it in items])
        out[name + '_dt'] = np.asarray([float(it['dt']) for it in items])
        out[name + '_timestamp'] = np.asarray([float(it['timestamp']) for it in items])
        if name == 'between_frames':
            out[name + '_frame'] = np.stack([it['frame'].numpy() for it in items])
    out['probes'] = probes
    np.savez_compressed(os.path.join(OUT, 'rec_dataset.npz'), **out,
                        meta=json.dumps(dict(recording=REC_DATASET, methods=REC_METHODS)))
    print('rec_dataset', {k: getattr(v, 'shape', v) for k, v in out.items()})


def gen_croper():
    Croper = ref_import.import_croper()
    res = {}
    for (h, w) in [(180, 240), (260, 346), (480, 640), (720, 1280), (181, 243)]:
        c = Croper(3)
        c.update_params(w, h)
        x = torch.arange(h * w, dtype=torch.float32).reshape(1, 1, h, w)
        p = c.pad(x)
        back = c.crop(p)
        assert torch.equal(back, x)
        res[f'{h}x{w}'] = dict(hc=c.height_crop_size, wc=c.width_crop_size,
                               pad=[c.padding_left, c.padding_right, c.padding_top, c.padding_bottom],
                               crop=[c.iy0, c.iy1, c.ix0, c.ix1],
                               padded_shape=list(p.shape[-2:]),
                               corner=[float(p[0, 0, c.padding_top, c.padding_left])])
    with open(os.path.join(OUT, 'croper.json'), 'w') as f:
        json.dump(res, f, indent=1)
    print('croper', res)


if __name__ == '__main__':
    assert ref_import.available(), 'needs /root/reference (build container only)'
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    if len(sys.argv) > 1:                      # e.g. `python oracle/gen_golden.py gen_cfgA_sampled gen_longT`
        for fn in sys.argv[1:]:
            globals()[fn]()
        sys.exit(0)
    gen_croper()
    gen_voxels()
    gen_recording_voxels()
    gen_recording_dataset()
    gen_blocks()
    gen_e2e()
    gen_cfgA_sampled()
    gen_bench_fixture()
    gen_longT()
    gen_variants()
