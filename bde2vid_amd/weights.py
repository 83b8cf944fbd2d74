"""State-dict layout of the reference model and deterministic "formula" weights.

The pretrained `BDE2VID.pth` is not available, so tests, goldens
and the benchmark use weights that are a pure function of (seed, key, shape).
Golden fixtures therefore never store weights: the generator script loads these
tensors into the reference via `load_state_dict`, and the build regenerates them.

Key map: SURVEY.md Appendix B (measured from the instantiated reference;
modules defined at `bde2vid_cross_scale_propogation_V5.py:43-97`,
`submodules.py:92,186-188,291`, `DTransformer.py:125-158,249-251`).
"""
import math
import re
import zlib
from typing import Dict, List, Tuple

import numpy as np
import torch

from .config import GeneratorConfig

PREFIX = 'generator.'


def state_dict_spec(cfg: GeneratorConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    """(key, shape) of every float parameter, in the reference's registration order."""
    cfg.validate()
    bc, ks, ne = cfg.basechannels, cfg.ks, cfg.num_encoders
    D = cfg.frame_num
    spec: List[Tuple[str, Tuple[int, ...]]] = []

    def add(k, *shape):
        spec.append((PREFIX + k, tuple(shape)))

    def add_conv_layer(prefix, cout, cin):
        """ConvLayer / UpsampleConvLayer parameters (submodules.py:91-103, 123-135): no conv bias under BN; BN has an affine,
        InstanceNorm2d(track_running_stats=True) only the running statistics."""
        add(prefix + 'conv2d.weight', cout, cin, ks, ks)
        if cfg.norm_kind != 1:
            add(prefix + 'conv2d.bias', cout)
        if cfg.norm_kind == 1:
            add(prefix + 'norm_layer.weight', cout)
            add(prefix + 'norm_layer.bias', cout)
        if cfg.norm_kind:
            add(prefix + 'norm_layer.running_mean', cout)
            add(prefix + 'norm_layer.running_var', cout)

    add_conv_layer('head.', bc, cfg.num_bins)
    for d in ('forward_encoder', 'backward_encoder'):
        for l in range(ne):
            ci, co = cfg.enc_in(l), cfg.enc_out(l)
            if not cfg.useRC:                      # a bare ConvLayer (V5.py:256-258)
                add_conv_layer(f'{d}.{l}.', co, ci)
                continue
            add_conv_layer(f'{d}.{l}.conv.', co, ci)
            if cfg.recurrent_block_type == 'convgru':          # submodules.py:348-350
                for gate in ('reset_gate', 'update_gate', 'out_gate'):
                    add(f'{d}.{l}.recurrent_block.{gate}.weight', co, 2 * co, 3, 3)
                    add(f'{d}.{l}.recurrent_block.{gate}.bias', co)
            else:
                add(f'{d}.{l}.recurrent_block.Gates.weight', 4 * co, 2 * co, 3, 3)
                add(f'{d}.{l}.recurrent_block.Gates.bias', 4 * co)
    for l in range(ne):  # dead on the forward path, present in checkpoints (V5.py:54-57)
        co = cfg.enc_out(l)
        add(f'fusion_layers.{l}.weight', co, 2 * co, 1, 1)
        add(f'fusion_layers.{l}.bias', co)
    tbl = (2 * D - 1) * (2 * cfg.window_size[0] - 1) * (2 * cfg.window_size[1] - 1)
    for l in range(ne):
        C = cfg.enc_out(l)
        hid = int(C * cfg.mlp_ratio)
        for i in range(cfg.depths[l]):
            p = f'feat_attns.{l}.blocks.{i}.'
            add(p + 'attn.relative_position_bias_table', tbl, cfg.num_heads)
            add(p + 'attn.norm_q.weight', C)
            add(p + 'attn.norm_q.bias', C)
            add(p + 'attn.norm_kv.weight', C)
            add(p + 'attn.norm_kv.bias', C)
            add(p + 'attn.q.weight', C, C)
            add(p + 'attn.q.bias', C)
            add(p + 'attn.kv.weight', 2 * C, C)
            add(p + 'attn.kv.bias', 2 * C)
            add(p + 'attn.proj.weight', C, C)
            add(p + 'attn.proj.bias', C)
            add(p + 'norm2.weight', C)
            add(p + 'norm2.bias', C)
            add(p + 'mlp.fc1.weight', hid, C)
            add(p + 'mlp.fc1.bias', hid)
            add(p + 'mlp.fc2.weight', C, hid)
            add(p + 'mlp.fc2.bias', C)
    if cfg.bottleneck:                            # Sequential(ParseLayer, ResidualBlockNoBN x num_res_blocks), V5.py:77-80
        C = cfg.enc_out(ne - 1)
        for k in range(cfg.num_res_blocks):
            for conv in ('conv1', 'conv2'):
                add(f'feat_attns.{ne - 1}.{1 + k}.{conv}.weight', C, C, 3, 3)
                add(f'feat_attns.{ne - 1}.{1 + k}.{conv}.bias', C)
    for j in range(ne):
        cin = cfg.enc_out(ne - 1 - j)
        cout = cfg.enc_in(ne - 1 - j)
        if cfg.skip_type == 'concat':              # 1x1 fusion of cat(skip, x), V5.py:86-89
            add(f'decoders.{j}.0.weight', cin, 2 * cin, 1, 1)
            add(f'decoders.{j}.0.bias', cin)
        add_conv_layer(f'decoders.{j}.1.', cout, cin)
    if cfg.skip_type == 'concat':                  # V5.py:92-93
        add('predI.0.weight', bc, 2 * bc, 1, 1)
        add('predI.0.bias', bc)
    add('predI.1.weight', cfg.num_output_channels, bc, 1, 1)
    add('predI.1.bias', cfg.num_output_channels)
    return spec


def num_parameters(cfg: GeneratorConfig) -> int:
    return sum(int(np.prod(s)) for _, s in state_dict_spec(cfg))


def formula_tensor(key: str, shape: Tuple[int, ...], seed: int) -> np.ndarray:
    """Deterministic fp32 tensor for a state-dict entry; pure function of its arguments."""
    rng = np.random.default_rng([int(seed), zlib.crc32(key.encode('utf-8'))])
    x = rng.standard_normal(shape, dtype=np.float32)
    leaf = key.rsplit('.', 1)[-1]
    if leaf == 'relative_position_bias_table':
        return (0.5 * x).astype(np.float32)
    if leaf == 'running_var':
        return (0.6 + 0.4 * np.abs(x)).astype(np.float32)
    if leaf == 'running_mean':
        return (0.2 * x).astype(np.float32)
    if re.search(r'norm\w*\.weight$', key):
        return (1.0 + 0.1 * x).astype(np.float32)
    if leaf == 'bias':
        return (0.1 * x).astype(np.float32)
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else int(shape[0])
    gain = math.sqrt(2.0) if len(shape) == 4 else 1.0
    return (x * np.float32(gain / math.sqrt(fan_in))).astype(np.float32)


def formula_state_dict(cfg: GeneratorConfig, seed: int = 4) -> Dict[str, torch.Tensor]:
    return {k: torch.from_numpy(formula_tensor(k, s, seed)) for k, s in state_dict_spec(cfg)}


def relative_position_index(D: int, wh: int, ww: int) -> np.ndarray:
    """[D*wh*ww, D*wh*ww] int64 index into the relative-position bias table.

    Restates the buffer built at `DTransformer.py:139-153`: token order is
    (d, h, w) row-major; index = ((dd+D-1)*(2wh-1) + (dh+wh-1))*(2ww-1) + (dw+ww-1).
    """
    d, h, w = np.meshgrid(np.arange(D), np.arange(wh), np.arange(ww), indexing='ij')
    d, h, w = d.ravel(), h.ravel(), w.ravel()
    dd = d[:, None] - d[None, :] + (D - 1)
    dh = h[:, None] - h[None, :] + (wh - 1)
    dw = w[:, None] - w[None, :] + (ww - 1)
    return ((dd * (2 * wh - 1) + dh) * (2 * ww - 1) + dw).astype(np.int64)


def infer_config(sd: Dict[str, torch.Tensor], **overrides) -> GeneratorConfig:
    """Recover the generator hyper-parameters from tensor shapes of a state dict.

    Needed because the reference stores them only as a config string inside the
    checkpoint (`eval_models_seq.py:53-60`).  `buffer_index`/`q_idx` beyond the
    frame count D are not recoverable from shapes; D comes from the bias-table
    rows and the symmetric buffer [-D//2..D//2] / centre query is assumed unless
    overridden.
    """
    g = {k[len(PREFIX):]: v for k, v in sd.items() if k.startswith(PREFIX)}
    hw = g['head.conv2d.weight']
    bc, num_bins, ks = int(hw.shape[0]), int(hw.shape[1]), int(hw.shape[2])
    use_rc = 'forward_encoder.0.conv.conv2d.weight' in g
    enc_key = 'forward_encoder.{}.conv.conv2d.weight' if use_rc else 'forward_encoder.{}.conv2d.weight'
    ne = 0
    while enc_key.format(ne) in g:
        ne += 1
    depths, heads, tbl_rows = [], None, None
    for l in range(ne):
        d = 0
        while f'feat_attns.{l}.blocks.{d}.attn.q.weight' in g:
            t = g[f'feat_attns.{l}.blocks.{d}.attn.relative_position_bias_table']
            tbl_rows, heads = int(t.shape[0]), int(t.shape[1])
            d += 1
        depths.append(d)
    kw = dict(num_bins=num_bins, basechannels=bc, num_encoders=ne, ks=ks, depths=tuple(depths), useRC=use_rc)
    if use_rc and 'forward_encoder.0.recurrent_block.reset_gate.weight' in g:
        kw['recurrent_block_type'] = 'convgru'
    if 'head.norm_layer.weight' in g:
        kw['norm'] = 'BN'
    elif 'head.norm_layer.running_mean' in g:
        kw['norm'] = 'IN'
    if 'decoders.0.0.weight' in g:
        kw['skip_type'] = 'concat'
    if ne and depths[-1] == 0:
        k = 0
        while f'feat_attns.{ne - 1}.{1 + k}.conv1.weight' in g:
            k += 1
        kw['num_res_blocks'] = k
    if heads is not None:
        kw['num_heads'] = heads
        D = (tbl_rows // (13 * 13) + 1) // 2
        kw['buffer_index'] = tuple(range(-(D // 2), D - D // 2))
        kw['q_idx'] = D // 2
    kw.update(overrides)
    cfg = GeneratorConfig(**kw)
    cfg.validate()
    return cfg
