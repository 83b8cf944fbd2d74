"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the BDE2VID forward pass.

Plain functional PyTorch (fp32, CPU) over a flat state dict; no mmengine / mmcv /
timm.  It is the parity yard-stick that travels to the GPU box (the reference
itself cannot) and the `cpu_baseline` ("port") leg of bench.py.  The product
(`bde2vid_amd/`) never imports this module.

Parity status: PINNED.  `oracle/gen_golden.py` ran the real reference
(`/root/reference`, imported with the stub packages of `oracle/stubs/`) and wrote
`tests/golden/*.npz`; `tests/test_oracle_golden.py` checks this restatement
against every one of them (<= 2e-6 max abs), on CPU, every round.  The reference
ships no tests or golden vectors of its own.

Every function cites the reference lines it restates (paths relative to
/root/reference).
"""
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

P = 'generator.'


# --------------------------------------------------------------------------
# conv building blocks -- model/BDE2VID/submodules.py
# --------------------------------------------------------------------------
def conv_layer(x, w, b, stride, act: Optional[str]):
    """ConvLayer.forward with norm=None (submodules.py:105-114): conv, bias, activation."""
    y = F.conv2d(x, w, b, stride=stride, padding=w.shape[-1] // 2)
    if act == 'relu':
        y = torch.relu(y)
    elif act == 'relu6':
        y = torch.clamp(y, 0.0, 6.0)
    return y


def conv_block(x, sd, prefix, cfg, stride, act: Optional[str], upsample=False):
    """ConvLayer.forward / UpsampleConvLayer.forward with the norm argument (submodules.py:85-114, 117-147): conv (no bias
    under BN, :91) -> BatchNorm2d | InstanceNorm2d(track_running_stats=True), both in eval mode = the running statistics
    (:96-103, 108-109) -> activation."""
    if upsample:
        x = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False)
    w = sd[prefix + 'conv2d.weight']
    y = F.conv2d(x, w, sd.get(prefix + 'conv2d.bias'), stride=stride, padding=w.shape[-1] // 2)
    kind = cfg.norm_kind
    if kind == 1:
        y = F.batch_norm(y, sd[prefix + 'norm_layer.running_mean'], sd[prefix + 'norm_layer.running_var'],
                         sd[prefix + 'norm_layer.weight'], sd[prefix + 'norm_layer.bias'], training=False, eps=1e-5)
    elif kind == 2:
        y = F.instance_norm(y, sd[prefix + 'norm_layer.running_mean'], sd[prefix + 'norm_layer.running_var'], None, None,
                            use_input_stats=False, eps=1e-5)
    if act == 'relu':
        y = torch.relu(y)
    elif act == 'relu6':
        y = torch.clamp(y, 0.0, 6.0)
    return y


def convgru_cell(x, h_prev, sd, pre):
    """ConvGRU.forward (submodules.py:358-376): update / reset = sigmoid(conv3x3(cat(x, h))); candidate =
    tanh(conv3x3(cat(x, h * reset))); h' = h (1 - update) + candidate * update.  A missing state is zeros (:364-366)."""
    if h_prev is None:
        h_prev = torch.zeros_like(x)
    stacked = torch.cat([x, h_prev], dim=1)
    update = torch.sigmoid(F.conv2d(stacked, sd[pre + 'update_gate.weight'], sd[pre + 'update_gate.bias'], padding=1))
    reset = torch.sigmoid(F.conv2d(stacked, sd[pre + 'reset_gate.weight'], sd[pre + 'reset_gate.bias'], padding=1))
    cand = torch.tanh(F.conv2d(torch.cat([x, h_prev * reset], dim=1), sd[pre + 'out_gate.weight'], sd[pre + 'out_gate.bias'],
                               padding=1))
    return h_prev * (1 - update) + cand * update


def residual_block_no_bn(x, sd, pre):
    """ResidualBlockNoBN.forward (V5.py:271-274): x + conv2(relu(conv1(x)))."""
    y = torch.relu(F.conv2d(x, sd[pre + 'conv1.weight'], sd[pre + 'conv1.bias'], padding=1))
    return x + F.conv2d(y, sd[pre + 'conv2.weight'], sd[pre + 'conv2.bias'], padding=1)


def convlstm_cell(x, state, w, b):
    """ConvLSTM.forward (submodules.py:293-334).

    gates = conv3x3(cat(x, h_prev)); chunk order i, f, o, g (:320);
    c = sigmoid(f)*c_prev + sigmoid(i)*tanh(g); h = sigmoid(o)*tanh(c) (:331-332).
    A missing state is zeros (:300-311).
    """
    if state is None:
        h_prev = torch.zeros_like(x)
        c_prev = torch.zeros_like(x)
    else:
        h_prev, c_prev = state
    gates = F.conv2d(torch.cat([x, h_prev], dim=1), w, b, padding=w.shape[-1] // 2)
    gi, gf, go, gg = torch.chunk(gates, 4, dim=1)
    c = torch.sigmoid(gf) * c_prev + torch.sigmoid(gi) * torch.tanh(gg)
    h = torch.sigmoid(go) * torch.tanh(c)
    return h, c


def upsample_conv_layer(x, w, b):
    """UpsampleConvLayer.forward (submodules.py:137-147) as built at V5.py:84-85:
    bilinear x2 (align_corners=False) -> ks x ks conv -> ReLU6."""
    up = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False)
    return conv_layer(up, w, b, 1, 'relu6')


# --------------------------------------------------------------------------
# temporal window attention -- model/BDE2VID/DTransformer.py
# --------------------------------------------------------------------------
def window_partition(x, ws, dilated: bool):
    """window_partition (DTransformer.py:40-60).  x: [D,B,C,Hp,Wp], Hp,Wp multiples of ws.

    plain:   window (i,j) holds pixels (ws*i + a, ws*j + b).
    dilated: pad ws zeros bottom/right, window (i,j) holds pixels (ws*i + 2a, ws*j + 2b)
             (F.unfold kernel ws, dilation 2, stride ws).
    Returns [D, B*nW, C, ws, ws] with window index = i*nWw + j.
    """
    D, B, C, Hp, Wp = x.shape
    nh, nw = Hp // ws, Wp // ws
    if dilated:
        x = F.pad(x, (0, ws, 0, ws))
    step = 2 if dilated else 1
    ar = torch.arange(ws) * step
    rows = (torch.arange(nh) * ws)[:, None] + ar[None, :]          # [nh, ws]
    cols = (torch.arange(nw) * ws)[:, None] + ar[None, :]          # [nw, ws]
    g = x[:, :, :, rows][:, :, :, :, :, cols]                      # [D,B,C,nh,ws,nw,ws]
    g = g.permute(0, 1, 3, 5, 2, 4, 6).reshape(D, B * nh * nw, C, ws, ws)
    return g


def window_reverse(win, B, Hp, Wp, dilated: bool):
    """window_reverse (DTransformer.py:63-83).  win: [B*nW, C, ws, ws] -> [B,C,Hp,Wp].

    dilated: F.fold with dilation 2 / stride ws onto a (Hp+ws, Wp+ws) canvas, then crop;
    every canvas pixel receives 0 or 1 contributions, uncovered pixels stay 0.
    """
    _, C, ws, _ = win.shape
    nh, nw = Hp // ws, Wp // ws
    w = win.reshape(B, nh, nw, C, ws, ws)
    if not dilated:
        return w.permute(0, 3, 1, 4, 2, 5).reshape(B, C, Hp, Wp)
    canvas = win.new_zeros(B, C, Hp + ws, Wp + ws)
    ar = torch.arange(ws) * 2
    rows = ((torch.arange(nh) * ws)[:, None] + ar[None, :]).reshape(-1)   # [nh*ws]
    cols = ((torch.arange(nw) * ws)[:, None] + ar[None, :]).reshape(-1)
    vals = w.permute(0, 3, 1, 4, 2, 5).reshape(B, C, nh * ws, nw * ws)
    canvas[:, :, rows[:, None], cols[None, :]] = vals   # indices are unique -> plain store == fold sum
    return canvas[:, :, :Hp, :Wp]


def window_attention(xw, sd, pre, heads, q_ind, rel_index):
    """WindowAttention3D.forward (DTransformer.py:165-207), nwin_size=None.

    xw: [D, Bw, C, ws, ws].  Query tokens: frame q_ind's window (ws*ws);
    key/value tokens: all D frames, order (d, h, w) (:177-182).  Separate LayerNorms
    for q and kv (:183-184); q scaled by head_dim**-0.5 before QK^T (:192);
    bias rows q_ind*ws*ws.. of the relative table (:195-199).
    """
    D, Bw, C, wh, ww = xw.shape
    M = wh * ww
    tok = xw.permute(1, 0, 3, 4, 2).reshape(Bw, D * M, C)            # (d,h,w) order
    qtok = tok[:, q_ind * M:(q_ind + 1) * M]
    qn = F.layer_norm(qtok, (C,), sd[pre + 'norm_q.weight'], sd[pre + 'norm_q.bias'], 1e-5)
    kvn = F.layer_norm(tok, (C,), sd[pre + 'norm_kv.weight'], sd[pre + 'norm_kv.bias'], 1e-5)
    hd = C // heads
    q = F.linear(qn, sd[pre + 'q.weight'], sd[pre + 'q.bias']).reshape(Bw, M, heads, hd).transpose(1, 2)
    kv = F.linear(kvn, sd[pre + 'kv.weight'], sd[pre + 'kv.bias']).reshape(Bw, D * M, 2, heads, hd)
    k = kv[:, :, 0].transpose(1, 2)                                   # [Bw, heads, N, hd]
    v = kv[:, :, 1].transpose(1, 2)
    attn = (q * (hd ** -0.5)) @ k.transpose(-1, -2)                   # [Bw, heads, M, N]
    N = D * M
    table = sd[pre + 'relative_position_bias_table']
    bias = table[rel_index[q_ind * M:(q_ind + 1) * M, :N].reshape(-1)].reshape(M, N, heads)
    attn = attn + bias.permute(2, 0, 1).unsqueeze(0)
    attn = torch.softmax(attn, dim=-1)
    out = (attn @ v).transpose(1, 2).reshape(Bw, M, C)
    out = F.linear(out, sd[pre + 'proj.weight'], sd[pre + 'proj.bias'])
    return out.reshape(Bw, wh, ww, C).permute(0, 3, 1, 2)             # [Bw, C, ws, ws]


def swin_block(frames, sd, pre, heads, q_ind, dilated, ws, rel_index):
    """SwinTransformerBlock3D.forward (DTransformer.py:254-306).  frames: [D,B,C,H,W].

    part1 (:254-277): zero-pad H,W to multiples of ws (pad//2 before, rest after);
    partition; attention; reverse; crop.  There is no norm1 (:243,258).
    x = shortcut + attn (:299);  x = x + fc2(GELU(fc1(LN2(x)))) per pixel (:279-283,304).
    """
    D, B, C, H, W = frames.shape
    assert H >= ws and W >= ws, 'maps smaller than the window crash the reference (SURVEY.md §7)'
    ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
    pt, pl = ph // 2, pw // 2
    xp = F.pad(frames, (pl, pw - pl, pt, ph - pt))
    Hp, Wp = H + ph, W + pw
    xw = window_partition(xp, ws, dilated)
    aw = window_attention(xw, sd, pre + 'attn.', heads, q_ind, rel_index)
    a = window_reverse(aw, B, Hp, Wp, dilated)[:, :, pt:pt + H, pl:pl + W]
    x = frames[q_ind] + a
    t = x.permute(0, 2, 3, 1)
    t = F.layer_norm(t, (C,), sd[pre + 'norm2.weight'], sd[pre + 'norm2.bias'], 1e-5)
    t = F.linear(t, sd[pre + 'mlp.fc1.weight'], sd[pre + 'mlp.fc1.bias'])
    t = F.gelu(t)                                                      # exact erf GELU (nn.GELU())
    t = F.linear(t, sd[pre + 'mlp.fc2.weight'], sd[pre + 'mlp.fc2.bias'])
    return x + t.permute(0, 3, 1, 2)


def dframe_attention(buf: Sequence[torch.Tensor], sd, pre, depth, heads, q_ind, ws, rel_index):
    """DFrameAttention.forward (DTransformer.py:376-389): block i is dilated iff i is odd
    (:362); before each block the query slot of the buffer is overwritten with the
    current x (:386); the other slots keep their original frames."""
    keys = list(buf)
    x = keys[q_ind]
    for i in range(depth):
        keys[q_ind] = x
        x = swin_block(torch.stack(keys, 0), sd, f'{pre}blocks.{i}.', heads, q_ind, i % 2 == 1, ws,
                       rel_index)
    return x


# --------------------------------------------------------------------------
# generator -- model/BDE2VID/bde2vid_cross_scale_propogation_V5.py:100-241
# --------------------------------------------------------------------------
def forward(sd: Dict[str, torch.Tensor], cfg, inputs: List[dict], capture: Optional[dict] = None):
    """BDE2VID.forward(inputs, mode='tensor') (bde2vid.py:30-50) == V5.forward (V5.py:100-241).

    `inputs`: list of T dicts {'events': [B, num_bins, Hp, Wp]}.  Returns list of T
    [B,1,Hp,Wp] images.  Recurrent state always starts from zero (bde2vid.py:31).
    `capture`, if given, receives named intermediates for the per-block GPU tests.
    """
    from bde2vid_amd.weights import relative_position_index
    T = len(inputs)
    ne = cfg.num_encoders
    ws = cfg.window_size[0]
    rel_index = torch.from_numpy(relative_position_index(cfg.frame_num, ws, cfg.window_size[1]))

    # A. head conv on every frame (V5.py:116)
    head = [conv_block(d['events'], sd, P + 'head.', cfg, 1, 'relu') for d in inputs]
    if capture is not None:
        capture['head'] = torch.stack(head)
    levels = []
    target = head
    gru = cfg.recurrent_block_type == 'convgru'
    for l in range(ne):
        # B. bidirectional recurrent sweep (V5.py:119-135): both encoders read the same sequence
        f_seq, b_seq = [None] * T, [None] * T
        for name, order, out in (('forward_encoder', range(T), f_seq),
                                 ('backward_encoder', range(T - 1, -1, -1), b_seq)):
            pre = f'{P}{name}.{l}.'
            state = None
            for t in order:
                if not cfg.useRC:                                       # a bare ConvLayer (V5.py:256-258)
                    out[t] = conv_block(target[t], sd, pre, cfg, 2, 'relu')
                    continue
                x = conv_block(target[t], sd, pre + 'conv.', cfg, 2, 'relu')   # submodules.py:192
                if gru:
                    state = convgru_cell(x, state, sd, pre + 'recurrent_block.')
                    out[t] = state                                      # submodules.py:194
                else:
                    state = convlstm_cell(x, state, sd[pre + 'recurrent_block.Gates.weight'],
                                          sd[pre + 'recurrent_block.Gates.bias'])
                    out[t] = state[0]
        merged = [f_seq[t] + b_seq[t] for t in range(T)]                # V5.py:137-147
        if capture is not None:
            capture[f'merged{l}'] = torch.stack(merged)
        # temporal attention with in-place refinement (V5.py:151-169)
        if cfg.depths[l] > 0 or (l == ne - 1 and cfg.bottleneck):
            zero = torch.zeros_like(merged[0])
            for t in range(T):
                buf = [merged[t + o] if 0 <= t + o < T else zero for o in cfg.buffer_index]
                if cfg.depths[l] > 0:
                    x = dframe_attention(buf, sd, f'{P}feat_attns.{l}.', cfg.depths[l], cfg.num_heads,
                                         cfg.q_idx, ws, rel_index)
                else:
                    # Sequential(ParseLayer, ResidualBlockNoBN x n) (V5.py:77-80): ParseLayer takes buffer slot 0 (:281-282),
                    # i.e. the frame at offset buffer_index[0] -- not the query frame
                    x = buf[0]
                    for k in range(cfg.num_res_blocks):
                        x = residual_block_no_bn(x, sd, f'{P}feat_attns.{l}.{1 + k}.')
                merged[t] = x + merged[t]
            if capture is not None:
                capture[f'refined{l}'] = torch.stack(merged)
        levels.append(merged)
        target = merged

    # C. decoder (V5.py:183-197); the last level is appended twice (:149-150,172) so the first
    #    decoder sees L[-1] + L[-1] (skip_sum) or cat(L[-1], L[-1]) (skip_concat, :285-286)
    concat = cfg.skip_type == 'concat'
    out = []
    for t in range(T):
        x = levels[-1][t]
        for j in range(ne):
            skip = levels[ne - 1 - j][t]
            if concat:
                x = F.conv2d(torch.cat([skip, x], dim=1), sd[f'{P}decoders.{j}.0.weight'], sd[f'{P}decoders.{j}.0.bias'])
            else:
                x = skip + x
            x = conv_block(x, sd, f'{P}decoders.{j}.1.', cfg, 1, 'relu6', upsample=True)
            if capture is not None and t == 0:
                capture[f'dec{j}_t0'] = x
        if concat:
            x = F.conv2d(torch.cat([x, head[t]], dim=1), sd[P + 'predI.0.weight'], sd[P + 'predI.0.bias'])
        else:
            x = x + head[t]
        y = F.conv2d(x, sd[P + 'predI.1.weight'], sd[P + 'predI.1.bias'])
        if cfg.activation == 'Sigmoid':
            y = torch.sigmoid(y)
        out.append(y)
    return out


# --------------------------------------------------------------------------
# pad / crop glue -- utils_func/inference_utils.py:26-32,69-114
# --------------------------------------------------------------------------
def crop_params(width, height, num_encoders):
    """Croper.update_params: pad to the next multiple of 2**num_encoders, ceil on top/left."""
    m = 2 ** num_encoders
    wc, hc = -(-width // m) * m, -(-height // m) * m
    pt = -(-(hc - height) // 2)
    pl = -(-(wc - width) // 2)
    pb, pr = (hc - height) // 2, (wc - width) // 2
    cx, cy = wc // 2, hc // 2
    ix0, ix1 = cx - width // 2, cx + -(-width // 2)
    iy0, iy1 = cy - height // 2, cy + -(-height // 2)
    return dict(hc=hc, wc=wc, pad=(pl, pr, pt, pb), crop=(iy0, iy1, ix0, ix1))
