#!/usr/bin/env python3
"""Headline benchmark: reconstructed frames/s of BDE2VID.forward on MI355X.

A step = one `model.forward` over one synthetic sequence (BASELINE.json configs[1]:
single MI355X, 5x240x180 voxels padded to 5x184x240, seq_len 16, fp32, batch 1, config "A").
Inputs are voxel grids produced by the HIP scatter from synthetic events and are resident in HBM
before the timed region.  Multi-GPU: one process per GPU (torchrun), every rank runs its own
replica on its own sequences (weak scaling), one RCCL weight broadcast at start-up, no per-step
collective.

After the timed region the frames the LAST timed step produced (pipelined, replayed from the
captured graphs) are compared with `tests/golden/e2e_bench_T16.npz` -- what the reference itself
computed for these events -- on every rank; the JSON line carries `"verified": true` only then,
and no line is printed when the library reports that stages were skipped (`debug_skip`).
"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md, dense fp32 matrix peak (= fp32 vector peak)
# fp32-equivalent work on the 16-bit matrix cores (csrc/split.h): dense bf16 / fp16 peak (2.5 PFLOP/s, MI355X_MICROARCH.md)
# divided by the MFMAs per fp32 block of the operand format -- three for two fp16 terms (the default), six for three bf16 terms
SPLIT_PEAK_TFLOPS = {2: 2500.0 / 3, 3: 2500.0 / 6}
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md, HBM3E spec
ATOMIC_PEAK_GBPS = 1300.0       # MI355X_MICROARCH.md, global float atomics: ~1.3 TB/s of added bytes chip-wide
PROFILE_TAG = 'r4'              # the committed rocprofv3 artefacts this line cites: profiles/<tag>_*
PMC_FILE = os.path.join(REPO, 'profiles', PROFILE_TAG + '_pmc.json')


def log(msg):
    """Progress line (stderr + gpurun_out/bench_progress.log): long silent runs are killed as hung."""
    line = f'[bench {time.strftime("%H:%M:%S")}] {msg}'
    print(line, file=sys.stderr, flush=True)
    try:
        os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(REPO, 'gpurun_out', 'bench_progress.log'), 'a') as f:
            f.write(line + '\n')
    except OSError:
        pass


def host_cores():
    """Usable host cores: affinity mask capped by the cgroup CPU quota (a GPU box hands out a
    share of its cores; running 256 threads on a 16-core share is pathologically slow)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            q, p = f.read().split()
            if q != 'max':
                n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    env = os.environ.get('BDE_CPU_THREADS')
    if env:
        n = int(env)
    return max(1, min(n, 64))


# ------------------------------------------------------------------------------------------------
# algorithmic work of every timed launch (flops from the layer shapes; bytes = inputs + weights read
# once, outputs written once)
# ------------------------------------------------------------------------------------------------
class Work:
    """flops / bytes per launch of each profiled span of one forward (config, T, B, Hp, Wp)."""

    def __init__(self, cfg, T, B, H, W, fuse_x=(), core2=True):
        self.cfg, self.T, self.B, self.H, self.W = cfg, T, B, H, W
        self.fuse_x = set(fuse_x)          # levels whose recurrent step contracts [x | h] itself (no batched gate convolution)
        self.core2 = core2                 # wide_core.h computes the refined neighbour's K | V inside the attention core

    def _lvl(self, l):
        return self.cfg.enc_in(l), self.cfg.enc_out(l), (self.H >> (l + 1)) * (self.W >> (l + 1))

    def attn_block_flops(self, l):
        """One SwinTransformerBlock3D on one frame (DTransformer.py:286-306): q on the padded query-frame tokens, k|v on
        the padded tokens of all D frames, scores and p*v over D*49 keys, proj, MLP (4C hidden) on the unpadded map."""
        c = self.cfg
        C, D = c.enc_out(l), c.frame_num
        h, w = self.H >> (l + 1), self.W >> (l + 1)
        npad = (-(-h // 7) * 7) * (-(-w // 7) * 7)
        hw = h * w
        return self.B * (2.0 * npad * C * C + 2.0 * npad * D * 2 * C * C + 4.0 * npad * D * 49 * C
                         + 2.0 * npad * C * C + 2.0 * hw * 8 * C * C)

    def flops(self, name):
        """(flops per launch, bound) for a kernel span; None for spans without an entry."""
        c, T, B = self.cfg, self.T, self.B
        TB = T * B
        m = re.fullmatch(r'([a-z_]+?)(\d*)', name)
        if m is None:
            return None
        base, idx = m.group(1), int(m.group(2)) if m.group(2) else None
        if base == 'wide_core':                           # wide_core.h: q|k|v of the query frame's padded tokens, k|v of the refined
            C = c.enc_out(idx)                            # neighbour's (one negative buffer offset), scores and p.v over D*49 keys
            h, w = self.H >> (idx + 1), self.W >> (idx + 1)
            npad = (-(-h // 7) * 7) * (-(-w // 7) * 7)
            nneg = sum(1 for d, o in enumerate(c.buffer_index) if d != c.q_idx and o < 0)
            prev = 2.0 * npad * 2 * C * C if (self.core2 and nneg == 1) else 0.0
            return B * (2.0 * npad * 3 * C * C + prev + 4.0 * npad * c.frame_num * 49 * C), 'mfma'
        if base == 'wide_mlp':                            # wide_mlp.h: proj + fc1 + fc2 on the unpadded map
            C = c.enc_out(idx)
            return B * (self.H >> (idx + 1)) * (self.W >> (idx + 1)) * 2.0 * 9 * C * C, 'mfma'
        if base.startswith('wide_'):                      # the fragment-layout chain runs the same GEMMs as the split path
            base = 'chain_' + base[5:]
        if base == 'head':
            return 2.0 * c.basechannels * c.num_bins * c.ks ** 2 * self.H * self.W * TB, 'mfma'
        if base == 'enc_conv':
            cin, cout, hw = self._lvl(idx)
            return 2 * TB * hw * 2.0 * cout * cin * c.ks ** 2, 'mfma'
        if base == 'gates_x':
            cin, cout, hw = self._lvl(idx)
            return 2 * TB * hw * 2.0 * 4 * cout * cout * 9, 'mfma'
        if base == 'lstm':
            cin, cout, hw = self._lvl(idx)
            k = 2 * cout if idx in self.fuse_x else cout
            return 2 * B * hw * (2.0 * 4 * cout * k * 9 + 20.0 * cout), 'mfma'
        if base in ('winblock', 'wideblock'):
            return self.attn_block_flops(idx), 'mfma'
        if base == 'dec_conv':
            l = c.num_encoders - 1 - idx
            cin, cout = c.enc_out(l), c.enc_in(l)
            return TB * (self.H >> l) * (self.W >> l) * 2.0 * cout * cin * c.ks ** 2, 'mfma'
        C = c.enc_out(idx) if idx is not None and idx < c.num_encoders else 0
        hw = (self.H >> (idx + 1)) * (self.W >> (idx + 1)) if idx is not None and idx < c.num_encoders else 0
        if base == 'chain_qkv':
            return B * hw * 2.0 * 3 * C * C, 'mfma'
        if base == 'chain_qkv_all':
            return TB * hw * 2.0 * 3 * C * C, 'mfma'
        if base == 'chain_kv':
            return B * hw * 2.0 * c.depths[idx] * 2 * C * C, 'mfma'
        if base == 'chain_kv_all':
            return TB * hw * 2.0 * c.depths[idx] * 2 * C * C, 'mfma'
        if base == 'chain_core':
            h, w = self.H >> (idx + 1), self.W >> (idx + 1)
            npad = (-(-h // 7) * 7) * (-(-w // 7) * 7)
            return B * 4.0 * npad * c.frame_num * 49 * C, 'mfma'
        if base == 'chain_projfc':
            return B * hw * 2.0 * (C * C + 4 * C * C), 'mfma'
        if base == 'chain_proj':
            return B * hw * 2.0 * C * C, 'mfma'
        if base in ('chain_mlp_in', 'chain_mlp_out'):
            return B * hw * 2.0 * 4 * C * C, 'mfma'
        if base == 'chain_token':
            return B * hw * 2.0 * (C * C + 8 * C * C + 3 * C * C), 'mfma'
        return None

    def split_share(self, name, info):
        """Fraction of a span's flops that the library issues on the 16-bit matrix cores from split operands (csrc/split.h)
        -- the rest runs as fp32 MFMAs or fp32 vector work, both 157.3 TFLOP/s dense.  `info` = the library's own report of
        what the latest forward launched (bde_get_info)."""
        c = self.cfg
        m = re.fullmatch(r'([a-z_]+?)(\d*)', name)
        base, idx = m.group(1), int(m.group(2)) if m.group(2) else None
        if base in ('head', 'enc_conv', 'gates_x', 'dec_conv', 'lstm'):
            key = {'head': 'sb_head', 'enc_conv': f'sb_enc{idx}', 'gates_x': f'sb_gx{idx}', 'dec_conv': f'sb_dec{idx}', 'lstm': f'sb_lstm{idx}'}[base]
            return 1.0 if info(key) == 1 else 0.0
        if base == 'winblock':
            if info('winblock_sb') != 1:
                return 0.0
            C, D = c.enc_out(idx), c.frame_num
            h, w = self.H >> (idx + 1), self.W >> (idx + 1)
            npad = (-(-h // 7) * 7) * (-(-w // 7) * 7)
            attn = self.B * 4.0 * npad * D * 49 * C                     # scores (fp32 MFMA) + p.v (vector ALU)
            return 1.0 - attn / self.attn_block_flops(idx)
        if base == 'wide_core':                                         # q|k|v (+ the refined neighbour's k|v) split, scores / p.v fp32 MFMA
            if not (info('wide_kv_sb') == 1 and info('sb_terms') == 2):
                return 0.0
            fl, _ = self.flops(name)
            C = c.enc_out(idx)
            h, w = self.H >> (idx + 1), self.W >> (idx + 1)
            npad = (-(-h // 7) * 7) * (-(-w // 7) * 7)
            return 1.0 - self.B * 4.0 * npad * c.frame_num * 49 * C / fl
        if base in ('wide_mlp', 'wide_projfc'):
            return 1.0 if (info('wide_fuse_mlp') == 1 and info('sb_terms') == 2) else 0.0
        if base in ('wide_kv', 'wide_kv_all'):
            return 1.0 if (info('wide_kv_sb') == 1 and info('sb_terms') == 2) else 0.0
        return 0.0

    def first_step_flops(self, l):
        """The first step of a sweep starts from h = 0 and skips the h-part of the contraction."""
        cin, cout, hw = self._lvl(l)
        return 2 * self.B * hw * ((2.0 * 4 * cout * cout * 9 if l in self.fuse_x else 0.0) + 20.0 * cout)

    def stage_flops(self):
        """Per FORWARD: encoder (north star: encoder convs + gate convs + recurrent steps), attention levels, decoder, head."""
        c, T, B = self.cfg, self.T, self.B
        enc = 0.0
        for l in range(c.num_encoders):
            cin, cout, hw = self._lvl(l)
            enc += 2 * B * T * hw * (2.0 * cout * cin * c.ks ** 2 + 2.0 * 4 * cout * 2 * cout * 9 + 20.0 * cout)
            enc -= 2 * B * hw * 2.0 * 4 * cout * cout * 9           # h = 0 at the first step of each sweep
        out = {'encoder_stage': enc, 'head': self.flops('head')[0]}
        for l in range(c.num_encoders):
            if c.depths[l] > 0:
                out[f'attention_l{l}'] = T * c.depths[l] * self.attn_block_flops(l)
        out['decoder'] = sum(self.flops(f'dec_conv{j}')[0] for j in range(c.num_encoders))
        return out


KERNEL_OF_SPAN = [
    # span name pattern -> (regex on the rocprofv3 kernel name, readable description)
    (r'lstm(\d)', r'lstm_sb_step_kernel|lstm16_step_kernel', 'recurrent ConvLSTM step of level {0}, both directions, gates of [x | h] contracted on split operands, pointwise tail fused (csrc/lstm_sb.h; csrc/lstm16.h where no shape fits)'),
    (r'winblock(\d)', r'winblock_sb_kernel|winblock_kernel', 'one temporal window-attention block of level {0} per launch (csrc/winblock_sb.h: GEMM phases on split operands, attention phase fp32)'),
    (r'wideblock(\d)', r'wideblock_', 'temporal window-attention block of level {0} (csrc/wideblock.h)'),
    (r'gates_x(\d)', r'conv_sb_kernel<3, 1|conv_vec_kernel<3, 1', 'x-part of the ConvLSTM gates of level {0}, 3x3 conv batched over T, both directions (csrc/conv_sb.h; conv_vec.h when no split-bf16 shape fits)'),
    (r'enc_conv(\d)', r'conv_sb_kernel<5, 2|conv_vec_kernel<5, 2', 'encoder 5x5 stride-2 conv of level {0}, batched over T, both directions (csrc/conv_sb.h / conv_vec.h)'),
    (r'dec_conv(\d)', r'conv_sb_kernel<5, 1|conv_vec_kernel<5, 1', 'decoder {0}: 5x5 conv on the bilinear x2 of (x + skip), batched over T (csrc/conv_sb.h / conv_vec.h)'),
    (r'head', r'conv_sb_kernel<5, 1|conv_vec_kernel<5, 1', 'head 5x5 conv, batched over T (csrc/conv_sb.h / conv_vec.h)'),
    (r'chain_(\w+?)(\d)', r'pw_gemm_kernel|attn_mfma16_kernel|attn_core_kernel|token_fused_kernel', 'split attention path of level {1}: {0}'),
    (r'wide_mlp(\d)', r'mlp_fused_kernel', 'token half of a level-{0} attention block in one launch: x1 = x + proj(.), GELU(fc1(LN(x1))), x2 = x1 + fc2(.), two-term split operands, fc2 K-split over four workgroups summed in fixed order by the last to arrive (csrc/wide_mlp.h)'),
    (r'wide_projfc(\d)', r'projfc1_sb_kernel', 'x1 = x + proj(.) and GELU(fc1(LN(x1))) of a level-{0} attention block in one launch, two-term split operands (csrc/wideblock.h)'),
    (r'wide_kv(_all)?(\d)', r'tokgemm_sb_kernel|tokgemm_kernel', 'K|V GEMM of the level-{1} attention chain (csrc/wideblock.h: two-term split operands)'),
    (r'wide_core(\d)', r'wide_core_kernel|attn_tok16_kernel', 'window half of a level-{0} attention block, one workgroup per (window, head): q|k|v of the query frame and k|v of the refined neighbour on two-term operands, scores / softmax / p.v fp32 (csrc/wide_core.h)'),
    (r'wide_(\w+?)(\d)', r'tokgemm_kernel', 'token GEMM of the level-{1} attention chain: {0} (csrc/wideblock.h)'),
]


def describe_span(name):
    for pat, kre, desc in KERNEL_OF_SPAN:
        m = re.fullmatch(pat, name)
        if m:
            return kre, desc.format(*m.groups())
    return None, name


def pmc_traffic(span, avg_us, workload):
    """HBM bytes per launch of the kernel behind `span` from the committed rocprofv3 --pmc passes of THIS round
    (tools/gpu.sh <tag> pmc -> profiles/<tag>_pmc.json), corrected as MI355X_MICROARCH.md (HBM) prescribes: on gfx950
    FETCH_SIZE tallies a wide coalesced read at half its bytes -- doubled -- and WRITE_SIZE is exact.  The counters were
    collected on ONE workload (recorded in the file): for any other (T, B, H, W) the answer is None."""
    try:
        with open(PMC_FILE) as f:
            pmc = json.load(f)
    except Exception:
        return None, None
    if pmc.get('workload') != workload:
        return None, {'file': os.path.relpath(PMC_FILE, REPO), 'why': f'counters were collected on {pmc.get("workload")}, this run is {workload}'}
    spans = pmc.get('spans', {})
    ent = spans.get(span)
    if ent is None:                                  # families sharing a kernel template ("gates_x*"): nearest duration
        base = re.sub(r'\d+$', '', span)
        for key, cands in spans.items():
            if isinstance(cands, list) and base in key:
                ent = min(cands, key=lambda e: abs((e.get('avg_us') or 0) - avg_us))
    if isinstance(ent, list):
        ent = min(ent, key=lambda e: abs((e.get('avg_us') or 0) - avg_us))
    if not ent or ent.get('fetch_bytes') is None or ent.get('write_bytes') is None:
        return None, None
    prov = {'file': os.path.relpath(PMC_FILE, REPO), 'kernel': ent.get('kernel'), 'dispatches': ent.get('dispatches'),
            'fetch_bytes_raw': ent.get('fetch_bytes'), 'write_bytes': ent.get('write_bytes'),
            'correction': 'traffic = 2 x FETCH_SIZE (gfx950 tallies 16 B/lane coalesced reads at half) + WRITE_SIZE',
            'avg_us_in_pmc_pass': ent.get('avg_us'), 'tag': pmc.get('tag'), 'workload': pmc.get('workload'), 'note': pmc.get('note')}
    return 2.0 * ent['fetch_bytes'] + ent['write_bytes'], prov


def voxel_report(device, T, sensor_hw, n_events, vox_dt):
    """Event binning: the bench's own grids (incl. H2D), the native-column scatter with columns resident in HBM, one
    large launch against the float-atomic roof, and the reference's algorithm timed on one host core."""
    import numpy as np
    import torch
    from bde2vid_amd import _lib
    from bde2vid_amd.events import events_to_voxel_windows
    from bde2vid_amd.synth import synthetic_recording
    sh, sw = sensor_hw
    out = {'events_per_s': n_events / vox_dt, 'events': n_events,
           'note': 'HIP scatter incl. H2D of the events; outside the timed region'}

    def run(n, nwin, reps, sensor=(sh, sw), method=0):
        h_, w_ = sensor
        xs, ys, ts, ps, _ = synthetic_recording(n, h_, w_, 4, 77)
        idx = np.arange(nwin + 1, dtype=np.int64) * (n // nwin)
        cols = [torch.from_numpy(a).to(device) for a in (xs, ys, ts, ps)]
        idx_d = torch.from_numpy(idx)
        _lib.check(_lib.lib().bde_voxel_method(method))
        try:
            events_to_voxel_windows(*cols, idx_d, 5, sensor_size=(h_, w_), device=device, check_bounds=False)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                events_to_voxel_windows(*cols, idx_d, 5, sensor_size=(h_, w_), device=device, check_bounds=False)
            e1.record()
            torch.cuda.synchronize()
        finally:
            _lib.check(_lib.lib().bde_voxel_method(0))
        dt = e0.elapsed_time(e1) * 1e-3 / reps
        nn = int(idx[-1])
        # pixel tiles per grid as csrc/voxel.h picks them
        cap = (128 * 1024) // (4 * 5)
        ntw = -(-w_ // 128)
        tw = -(-w_ // ntw)
        th = min(h_, cap // tw)
        tiles = -(-h_ // th) * ntw
        alg = nn * 13 + nwin * 5 * h_ * w_ * 4             # every event column byte read once, every grid cell written once
        return dict(events=nn, windows=nwin, sensor=[h_, w_], events_per_s=nn / dt, ms_per_call=dt * 1e3,
                    algorithmic_GBps=alg / dt / 1e9, tiles_per_grid=tiles)
    small = run(T * (sh * sw // 2), T, 20)
    small['note'] = ('int16/int16/float64/bool columns resident in HBM, one grid per window; HIP events on the launch stream '
                     'around the Python call (offsets H2D and allocation included): launch-bound at this size')
    out['native_columns'] = small
    big = run(24_000_000, 64, 3)
    big['roofline'] = {'bound': 'hbm', 'achieved': big['algorithmic_GBps'], 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                       'frac': big['algorithmic_GBps'] / HBM_PEAK_GBPS,
                       'note': 'algorithmic bytes (13 B per event + 4 B per grid cell) / time; bucketed binning (csrc/voxel.h): one pass '
                               'moves every event once into the run of its (window, pixel tile) as an 8-byte record, the tile '
                               'workgroups read only their own runs: 29 B of traffic per event whatever the tile count.  '
                               '`streaming` = last round\'s kernel (every tile\'s workgroup reads its window\'s events), same call'}
    big['streaming'] = {k: v for k, v in run(24_000_000, 64, 3, method=3).items() if k in ('ms_per_call', 'algorithmic_GBps')}
    vga = run(24_000_000, 64, 3, sensor=(480, 640))
    vga['roofline'] = {'bound': 'hbm', 'achieved': vga['algorithmic_GBps'], 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                       'frac': vga['algorithmic_GBps'] / HBM_PEAK_GBPS}
    vga['streaming'] = {k: v for k, v in run(24_000_000, 64, 3, sensor=(480, 640), method=3).items() if k in ('ms_per_call', 'algorithmic_GBps')}
    out['native_columns_24M_vga'] = vga
    out['native_columns_24M'] = big
    return out


def voxel_cpu_baseline(sensor_hw):
    """The reference's binning algorithm (B passes of index_put_(accumulate=True), event_utils.py:466-509) restated in
    oracle/voxel_oracle.py, on one host core: reported baseline, never the product."""
    import torch
    from oracle import voxel_oracle
    sh, sw = sensor_hw
    n = 2_000_000
    xs, ys, ts, ps = voxel_oracle.synthetic_events(n, sh, sw, 5)
    t = [torch.from_numpy(a) for a in (xs, ys, ts, ps)]
    torch.set_num_threads(1)
    voxel_oracle.events_to_voxel_indexput(*[a[:1000] for a in t], 5, (sh, sw))
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < 3.0:
        voxel_oracle.events_to_voxel_indexput(*t, 5, (sh, sw))
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    return dict(value=n / dt, unit='events/s', cores=1, kind='port',
                sample=f'{reps} x {n} events into 5x{sh}x{sw}, torch {torch.__version__} CPU index_put_(accumulate), {dt*1e3:.0f} ms each')


def cpu_baseline(cfg, sd, H, W, T):
    """CPU restatement (oracle) timed on the host cores: reported baseline, never the product."""
    import torch
    from oracle import bde2vid_oracle as O
    from oracle.gen_golden import golden_inputs
    cores = host_cores()
    torch.set_num_threads(cores)
    xs = [{'events': torch.from_numpy(x)} for x in golden_inputs(T, 1, cfg.num_bins, H, W, 7)]
    with torch.no_grad():
        t0 = time.perf_counter()
        O.forward(sd, cfg, xs)
        dt = time.perf_counter() - t0
    return dict(value=T / dt, unit='frames/s', cores=cores, kind='port',
                sample=f'one oracle forward, T={T}, 5x{H}x{W}, torch {torch.__version__} CPU, {dt:.1f} s')


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` without a launcher: start N ranks as CHILD processes (this process has not touched
    the GPU yet) and relay rank 0's JSON line."""
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = str(sk.getsockname()[1])
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    log(f'--gpus {args.gpus} without WORLD_SIZE: launching {args.gpus} ranks as child processes')
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--seq-len', type=int, default=16)
    ap.add_argument('--height', type=int, default=180)
    ap.add_argument('--width', type=int, default=240)
    ap.add_argument('--batch', type=int, default=1)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-strict', action='store_true', help='skip the strict-fp32 leg (profiling runs: only the default kernels in the trace)')
    ap.add_argument('--pipeline', type=int, default=2, help='independent sequences in flight per GPU (1..4); measured this round on one box: 2 / 3 / 4 in flight = 2296 / 2208 / 2098 frames/s')
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(relaunch_under_torchrun(args))
    # stdout carries exactly one JSON line: everything else a library prints there (RCCL's version banner at
    # communicator creation, for one) is sent to stderr
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)

    import ctypes as C
    import torch
    from bde2vid_amd import canonical, _lib
    from bde2vid_amd.dist import init_from_env, build_replicated_model, max_over_ranks, min_over_ranks, barrier
    from bde2vid_amd.weights import formula_state_dict
    from bde2vid_amd.harness import Croper
    from bde2vid_amd import workload

    log('start')
    rank, world, local = init_from_env()
    if world != max(args.gpus, 1):
        raise SystemExit(f'launched with WORLD_SIZE={world} but --gpus {args.gpus}')
    device = torch.device('cuda', local % max(torch.cuda.device_count(), 1))
    torch.cuda.set_device(device)
    cfg = canonical()
    sd_holder = {}

    def get_sd():
        sd_holder['sd'] = formula_state_dict(cfg)
        return sd_holder['sd']
    model = build_replicated_model(cfg, get_sd, device)
    tuning = os.environ.get('BDE_TUNING', '')
    for kv in tuning.split(','):
        if '=' in kv:
            k, v = kv.split('=')
            model.set_tuning(k, int(v))
    if model.get_info('debug_skip') != 0:
        raise SystemExit('debug_skip is set: stages of the forward are skipped, this is not a measurement')
    log('weights packed and resident')

    croper = Croper(cfg.num_encoders)
    croper.update_params(args.width, args.height)
    H, W, T, B = croper.height_crop_size, croper.width_crop_size, args.seq_len, args.batch
    # every rank reconstructs the same synthetic recording (its own replica, its own copy): the reference's frames
    # for it are committed, so every rank -- the broadcast receivers included -- can check what it computed
    fixture = workload.find_fixture(T, B, H, W, [args.height, args.width])
    seed0 = fixture[1]['seed'] if fixture else workload.SEED0
    vox, n_events, vox_dt = workload.bench_voxels(T, (args.height, args.width), device, seed0=seed0, batch=B)
    inputs = [{'events': vox[t]} for t in range(T)]
    log(f'voxel grids ready ({n_events} events in {vox_dt*1e3:.1f} ms)')
    can_verify = fixture is not None

    L = _lib.lib()
    with torch.no_grad():
        # one-time setup (untimed, part of model initialisation): each pipeline slot needs one eager call
        # (workspace allocation, kernel attributes) and one more to capture its hipGraph
        model.set_tuning('pipeline', args.pipeline)
        for i in range(2 * args.pipeline):
            model(inputs)
        model.wait()
        torch.cuda.synchronize(device)
        graphs_live = model.get_info('graphs_live')
        log(f'workspaces allocated, {graphs_live} launch graph(s) captured')
        for i in range(args.warmup):
            model(inputs)
        model.wait()
        torch.cuda.synchronize(device)
        log(f'{args.warmup} warmup steps done')
        barrier()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        last = None
        for _ in range(args.steps):
            last = model(inputs)
        model.wait()
        torch.cuda.synchronize(device)
        barrier()
        elapsed = time.perf_counter() - t0
        if model.get_info('debug_skip') != 0:
            raise SystemExit('debug_skip was set during the timed region')
        # ---- the frames of the last timed step against the reference's ------------------------------
        verified, verr = None, None
        if can_verify:
            ok, verr = workload.verify_against_fixture(torch.stack(last), fixture[0])
            verified = bool(min_over_ranks(1.0 if ok else 0.0, device) > 0.5)
            verr = max_over_ranks(verr, device)
            log(f'last timed step vs reference frames: max abs err {verr:.2e} -> verified={verified}')
        # ---- single stream: the same K steps with ONE sequence in flight (pipeline 1, graph replay): the latency-bound figure
        model.set_tuning('pipeline', 1)
        for _ in range(2):
            model(inputs)
        torch.cuda.synchronize(device)
        single_graph = model.get_info('graphs_live') >= 1
        barrier()
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            model(inputs)
        torch.cuda.synchronize(device)
        barrier()
        single_elapsed = time.perf_counter() - t1
        # ---- strict fp32: the same K steps with every split-operand kernel off (fp32 MFMA / fp32 vector arithmetic throughout:
        # what the path does with exact fp32 products), same number of sequences in flight, frames checked against the same fixture
        STRICT = dict(conv_sb=0, lstm_sbk=0, winblock_sb=0, wide_kv_sb=0, wide_fuse_mlp=0)
        strict_elapsed, strict_ok, strict_err, strict_sb = float('nan'), None, None, None
        if not args.no_strict:
            for k, v in STRICT.items():
                model.set_tuning(k, v)
            model.set_tuning('pipeline', args.pipeline)
            for i in range(2 * args.pipeline):
                model(inputs)
            model.wait()
            torch.cuda.synchronize(device)
            barrier()
            t2 = time.perf_counter()
            strict_last = None
            for _ in range(args.steps):
                strict_last = model(inputs)
            model.wait()
            torch.cuda.synchronize(device)
            barrier()
            strict_elapsed = time.perf_counter() - t2
            if can_verify:
                ok, strict_err = workload.verify_against_fixture(torch.stack(strict_last), fixture[0])
                strict_ok = bool(min_over_ranks(1.0 if ok else 0.0, device) > 0.5)
                strict_err = max_over_ranks(strict_err, device)
            strict_sb = [model.get_info(k) for k in ('sb_head', 'sb_enc0', 'sb_dec0', 'sb_lstm0', 'sb_lstm2')]
            for k in STRICT:
                model.set_tuning(k, 1)
            for kv in tuning.split(','):                              # (BDE_TUNING may have set one of them)
                if '=' in kv:
                    k, v = kv.split('=')
                    model.set_tuning(k, int(v))
            log(f'strict fp32: {args.steps * T * B / strict_elapsed:.1f} frames/s, verified={strict_ok}')
        # ---- per-kernel spans: HIP events around every launch, on the launch stream.  Events recorded inside a
        # replayed hipGraph cannot be read back, so the spans come from a few extra EAGER, un-pipelined steps
        # run right after the timed region (same inputs, same kernels).
        n_eager = min(args.steps, 3)
        model.set_tuning('pipeline', 1)
        model.set_tuning('graph', 0)
        L.bde_profile_reset(model._h, 1)
        for _ in range(n_eager):
            model(inputs)
        torch.cuda.synchronize(device)
    elapsed = max_over_ranks(elapsed, device)
    single_elapsed = max_over_ranks(single_elapsed, device)
    if not args.no_strict:
        strict_elapsed = max_over_ranks(strict_elapsed, device)
    log(f'timed region done: {elapsed:.3f} s for {args.steps} steps')

    if rank == 0:
        buf = C.create_string_buffer(8192)
        L.bde_profile_names(model._h, buf, len(buf))
        names = [n for n in buf.value.decode().split('\n') if n]
        spans = {}
        for nm in names:
            t_, c_ = C.c_double(), C.c_int64()
            L.bde_profile_get(model._h, nm.encode(), C.byref(t_), C.byref(c_))
            spans[nm] = (t_.value, int(c_.value))
        terms = model.get_info('sb_terms')
        fuse_x = [l for l in range(cfg.num_encoders)
                  if model.get_info('lstm_fuse_x') == 1 and model.get_info(f'sb_lstm{l}') == 1 and model.get_info(f'sb_gx{l}') == 0
                  and spans.get(f'gates_x{l}', (0, 0))[1] == 0]
        work = Work(cfg, T, B, H, W, fuse_x, core2=model.get_info('wide_core2') == 1)
        info = model.get_info
        split_peak = SPLIT_PEAK_TFLOPS[terms]

        def blended_peak(pairs):
            """Roof of a span (or a stage) that issues part of its flops on the 16-bit matrix cores from split operands and the rest
            as fp32 MFMAs / fp32 vector work: total flops / (split flops / split peak + fp32 flops / fp32 peak)."""
            fs = sum(f * sh for f, sh in pairs)
            ff = sum(f * (1.0 - sh) for f, sh in pairs)
            t = fs / split_peak + ff / FP32_MFMA_PEAK_TFLOPS
            return (fs + ff) / t if t > 0 else FP32_MFMA_PEAK_TFLOPS

        kernels = {}
        for nm, (ms, cnt) in spans.items():
            fb = work.flops(nm)
            if fb is None or cnt == 0 or ms <= 0:
                continue
            fl, bound = fb
            total_fl = cnt * fl
            if nm.startswith('lstm'):                       # first step of each sweep: no contraction
                l = int(nm[4:])
                total_fl = (cnt - n_eager) * fl + n_eager * work.first_step_flops(l)
            kernels[nm] = dict(ms_per_forward=ms / n_eager, launches_per_forward=cnt / n_eager, avg_us=ms / cnt * 1e3,
                               flops_per_launch=fl, achieved=total_fl / (ms * 1e-3) / 1e12, bound=bound, _total=total_fl)
        fwd_ms = spans.get('forward', (0.0, 0))[0] / max(n_eager, 1)
        for nm, k in kernels.items():
            k['share'] = k['ms_per_forward'] / fwd_ms if fwd_ms > 0 else None
            sh = work.split_share(nm, info)                 # what the library launched (bde_get_info), per arithmetic class
            k['split_share'] = round(sh, 4)
            k['peak'] = blended_peak([(1.0, sh)])
            if sh > 0:
                k['bound'] = (f'mfma: {sh:.0%} of the flops on the {"fp16" if terms == 2 else "bf16"} matrix cores from '
                              f'{"two" if terms == 2 else "three"}-term split operands ({3 if terms == 2 else 6} MFMAs per fp32 block, '
                              f'{split_peak:.1f} TFLOP/s), the rest fp32 (157.3): peak = flops / (split flops / {split_peak:.1f} + fp32 flops / 157.3)')
            k['frac'] = k['achieved'] / k['peak']
        dom = max(kernels, key=lambda n: kernels[n]['ms_per_forward'])
        dk = kernels[dom]
        kre, desc = describe_span(dom)
        wl_key = {'T': T, 'B': B, 'H': H, 'W': W}
        traffic, prov = pmc_traffic(dom, dk['avg_us'], wl_key)
        # stages (same eager steps): encoder = enc convs + gate convs + recurrent steps (the north star's yardstick)
        sf = work.stage_flops()
        stage_pat = {'encoder_stage': r'(enc_conv|gates_x|lstm)\d', 'head': r'head', 'decoder': r'dec_conv\d'}
        stage_ms = {'encoder_stage': sum(ms for nm, (ms, _) in spans.items() if re.fullmatch(r'(enc_conv|gates_x|lstm)\d', nm)),
                    'head': spans.get('head', (0, 0))[0],
                    'decoder': sum(ms for nm, (ms, _) in spans.items() if re.fullmatch(r'dec_(conv|up)\d', nm)) + spans.get('pred', (0, 0))[0]}
        for l in range(cfg.num_encoders):
            if cfg.depths[l] > 0:
                stage_ms[f'attention_l{l}'] = spans.get(f'attn{l}', (0, 0))[0]
                stage_pat[f'attention_l{l}'] = rf'(winblock|wide_\w+?|chain_\w+?){l}'
        stages = {}
        for nm, fl in sf.items():
            ms = stage_ms.get(nm, 0.0)
            if ms > 0:
                tf = fl * n_eager / (ms * 1e-3) / 1e12
                # roof of the stage: blended over the spans that make it up, by what each of them issued
                pairs = [(k['_total'], k['split_share']) for sp, k in kernels.items() if re.fullmatch(stage_pat[nm], sp)]
                pk = blended_peak(pairs) if pairs else FP32_MFMA_PEAK_TFLOPS
                stages[nm] = dict(flops_per_forward=fl, ms_per_forward=ms / n_eager, achieved=tf, unit='TFLOP/s', peak=pk, frac=tf / pk)
        for k in kernels.values():
            k.pop('_total', None)
        frames = args.steps * T * B * world
        out = {
            'metric': 'reconstructed frames/sec at 5x240x180 voxels, seq_len=16',
            'value': frames / elapsed, 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None,
            'dtype': ('f32 (operands split into two fp16 terms on the matrix cores where conv_sb / lstm_sb / winblock_sb apply: '
                      '3 MFMAs per fp32 block, f32 accumulate)' if terms == 2 else
                      'f32 (operands split into three bf16 terms on the matrix cores where conv_sb / lstm_sb / winblock_sb apply: '
                      '6 MFMAs per fp32 block, f32 accumulate)'), 'data': 'synthetic',
            'single_stream_fps': args.steps * T * B * world / single_elapsed,
            'single_stream_ms': 1e3 * single_elapsed / args.steps,
            'single_stream_note': (f'the same {args.steps} steps with ONE sequence in flight per GPU (pipeline 1, graph replay '
                                   f'{bool(single_graph)}): per-sequence latency, each call waits for its frames (range guard of the '
                                   'two-term format); `value` overlaps `config.pipeline` independent sequences'),
            'strict_fp32': None if args.no_strict else {'value': args.steps * T * B * world / strict_elapsed, 'unit': 'frames/s',
                            'ms_per_step': 1e3 * strict_elapsed / args.steps, 'verified': strict_ok, 'max_abs_err': strict_err,
                            'pipeline': args.pipeline,
                            'what': 'the same steps with every split-operand kernel switched off (' +
                                    ', '.join(f'{k}=0' for k in STRICT) + '): fp32 MFMAs and fp32 vector arithmetic throughout',
                            'split_kernels_launched': strict_sb},
            'profile_set': f'profiles/{PROFILE_TAG}_* (rocprofv3 --kernel-trace --stats of this command and of --pipeline 1; PMC passes)',
            'verified': verified,
            'verification': ({'against': f'{os.path.relpath(fixture[0], REPO)} (the reference\'s frames for these events)',
                              'what': 'frames of the last timed step (pipelined, graph replay), every rank', 'max_abs_err': verr,
                              'tolerance': workload.TOLERANCE} if can_verify else
                             {'against': None, 'why': 'reference fixtures exist for BASELINE configs 2, 3 and 5 at full size only: '
                                                      '16 x 5x180x240 B=1, 32 x 5x480x640 B=4, 64 x 5x720x1280 B=1'}),
            'config': {'workload': f'BDE2VID.forward config A (5 bins, 32 ch, depths [4,0,6], 16 heads, D=3), '
                                   f'{T} frames of 5x{args.height}x{args.width} (padded {H}x{W}), batch {B}, '
                                   f'random-init formula weights, one independent sequence per step per GPU',
                       'seq_len': T, 'height': args.height, 'width': args.width, 'batch': B,
                       'tuning': tuning or None, 'pipeline': args.pipeline,
                       'graph_replay': bool(graphs_live == args.pipeline),
                       'graphs_captured': graphs_live,
                       'parallelism': f'{world} replica(s), sequences sharded, 1 RCCL weight broadcast; per GPU '
                                      f'{args.pipeline} independent sequence(s) in flight'},
            'roofline': {'kernel': f'{dom}: {desc}', 'rocprof_kernel': kre, 'bound': dk['bound'],
                         'achieved': dk['achieved'], 'peak': dk['peak'], 'unit': 'TFLOP/s', 'frac': dk['frac'],
                         'traffic': traffic, 'traffic_source': prov,
                         'launches_per_forward': dk['launches_per_forward'], 'avg_us': dk['avg_us'],
                         'flops_per_launch': dk['flops_per_launch'], 'share_of_forward': dk['share'],
                         'chosen': 'the kernel with the largest total time among the HIP-event spans of this run',
                         'stages': stages,
                         'stages_note': 'fp32-equivalent flops / time; every peak is the blended roof of what the span or stage issued: '
                                        f'split-operand flops against {SPLIT_PEAK_TFLOPS[terms]:.1f} TFLOP/s (dense 16-bit matrix peak 2500 / '
                                        f'{3 if terms == 2 else 6} MFMAs per fp32 block, csrc/split.h), fp32 MFMA and vector flops against '
                                        f'157.3; kernels[*].split_share says how much of a span is which; recurrent steps of levels '
                                        f'{fuse_x} contract [x | h] (no gates_x launch)',
                         'kernels': {nm: {k: (round(v, 4) if isinstance(v, float) else v) for k, v in kk.items()}
                                     for nm, kk in sorted(kernels.items(), key=lambda kv: -kv[1]['ms_per_forward'])},
                         'eager_forward_ms': fwd_ms,
                         'measured': f'HIP events on the launch stream around each launch, over {n_eager} eager un-pipelined '
                                     'steps run right after the timed region (events inside hipGraph replays cannot be read); '
                                     'achieved = algorithmic flops of those launches / their total time'},
            'voxelize': voxel_report(device, T, (args.height, args.width), n_events, vox_dt),
        }
        if world == 1 and not args.no_cpu_baseline:
            log(f'cpu baseline on {host_cores()} host cores ...')
            out['cpu_baseline'] = cpu_baseline(cfg, sd_holder['sd'], H, W, T)
            out['voxelize']['cpu_baseline'] = voxel_cpu_baseline((args.height, args.width))
        print(json.dumps(out), file=json_out, flush=True)
    barrier()
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
