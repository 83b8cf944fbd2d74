#!/usr/bin/env python3
"""Headline benchmark: reconstructed frames/s of BDE2VID.forward on MI355X.

A step = one `model.forward` over one synthetic sequence (BASELINE.json configs[1]:
single MI355X, 5x240x180 voxels padded to 5x184x240, seq_len 16, fp32, batch 1, config "A").
Inputs are voxel grids produced by the HIP scatter from synthetic events and are resident in HBM
before the timed region.  Multi-GPU: one process per GPU (torchrun), every rank runs its own
sequences (weak scaling), one RCCL weight broadcast at start-up, no per-step collective.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import numpy as np   # noqa: E402
import torch         # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md, dense fp32 matrix peak


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


def lstm_shape(ho, wo, ch, B):
    """Tile shape csrc/lstm16.h picks for a [ch][ho][wo] hidden map (mirrors lstm16_launch; label only)."""
    cdiv = lambda a, b: -(-a // b)
    fill = lambda r, p: ho * cdiv(wo, 16) / (cdiv(ho, r) * cdiv(wo, p) * (r * p // 16))
    rows, pxw, bf = 1, 64, -1.0
    for r, p in ((1, 64), (2, 32), (4, 16)):
        f = fill(r, p)
        if f > bf + 0.02:
            bf, rows, pxw = f, r, p
    seg = 1
    if cdiv(ho, rows) * cdiv(wo, pxw) * cdiv(ch, 16) * 2 * B > 1024 and fill(1, 128) >= bf - 0.08:
        rows, pxw, seg = 1, 128, 2
    return f'{rows},{pxw},{seg}'


def lstm_flops_per_launch(cfg, B, H, W, level):
    """Algorithmic flops of ONE recurrent ConvLSTM step launch (both directions):
    h-part of the gates conv, 2 * (4C) * C * 9 per pixel, plus ~20 flop/px/channel pointwise."""
    C = cfg.enc_out(level)
    hw = (H >> (level + 1)) * (W >> (level + 1))
    return 2 * B * hw * (2.0 * 4 * C * C * 9 + 20.0 * C)


def synth_voxels(T, H, W, sensor_hw, device, seed0=1000):
    from bde2vid_amd.events import events_to_voxel_batch
    from bde2vid_amd.synth import synthetic_events
    sh, sw = sensor_hw
    packs = [synthetic_events(sh * sw // 2, sh, sw, seed0 + t) for t in range(T)]
    off = np.cumsum([0] + [len(p[0]) for p in packs])
    cat = [torch.from_numpy(np.concatenate([p[k] for p in packs])) for k in range(4)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    grids = events_to_voxel_batch(*cat, off, 5, sensor_size=(sh, sw), device=device)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pt, pl = -(-(H - sh) // 2), -(-(W - sw) // 2)
    out = torch.zeros((T, 1, 5, H, W), device=device)
    out[:, 0, :, pt:pt + sh, pl:pl + sw] = grids
    return out, int(off[-1]), dt


def voxel_native_rate(T, sensor_hw, device, reps=20):
    """Events/s of the native-column scatter (bde_voxelize_events) with the columns resident in HBM."""
    from bde2vid_amd.events import events_to_voxel_windows
    from bde2vid_amd.synth import synthetic_recording
    sh, sw = sensor_hw
    n = T * (sh * sw // 2)
    xs, ys, ts, ps, _ = synthetic_recording(n, sh, sw, 4, 77)
    idx = np.arange(T + 1, dtype=np.int64) * (n // T)
    cols = [torch.from_numpy(a).to(device) for a in (xs, ys, ts, ps)]
    events_to_voxel_windows(*cols, idx, 5, sensor_size=(sh, sw), device=device, check_bounds=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        events_to_voxel_windows(*cols, idx, 5, sensor_size=(sh, sw), device=device, check_bounds=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    bytes_per_call = n * 13 + n * 2 * 4 + T * 5 * sh * sw * 4      # columns + two float atomics per event + zero-fill
    return dict(events_per_s=n / dt, events=n, windows=T, GBps=bytes_per_call / dt / 1e9,
                note='int16/int16/float64/bool columns resident in HBM, one grid per window, zero-fill included; '
                     '13 B read + 2 float atomics per event')


def cpu_baseline(cfg, sd, H, W, T):
    """CPU restatement (oracle) timed on the host cores: reported baseline, never the product."""
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
    ap.add_argument('--pipeline', type=int, default=3, help='independent sequences in flight per GPU (1..4)')
    args = ap.parse_args()
    # stdout carries exactly one JSON line: everything else a library prints there (RCCL's version banner at
    # communicator creation, for one) is sent to stderr
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)

    from bde2vid_amd import canonical, _lib
    from bde2vid_amd.dist import init_from_env, build_replicated_model, max_over_ranks, barrier
    from bde2vid_amd.weights import formula_state_dict
    from bde2vid_amd.harness import Croper

    log('start')
    rank, world, local = init_from_env()
    assert world == max(args.gpus, 1) or world == 1, f'launched with WORLD_SIZE={world} but --gpus {args.gpus}'
    device = torch.device('cuda', local % max(torch.cuda.device_count(), 1))
    torch.cuda.set_device(device)
    cfg = canonical()
    sd_holder = {}

    def get_sd():
        sd_holder['sd'] = formula_state_dict(cfg)
        return sd_holder['sd']
    model = build_replicated_model(cfg, get_sd, device)
    for kv in os.environ.get('BDE_TUNING', '').split(','):
        if '=' in kv:
            k, v = kv.split('=')
            model.set_tuning(k, int(v))
    log('weights packed and resident')

    croper = Croper(cfg.num_encoders)
    croper.update_params(args.width, args.height)
    H, W, T, B = croper.height_crop_size, croper.width_crop_size, args.seq_len, args.batch
    vox, n_events, vox_dt = synth_voxels(T, H, W, (args.height, args.width), device, seed0=1000 + 97 * rank)
    if B > 1:
        vox = vox.repeat(1, B, 1, 1, 1)
    inputs = [{'events': vox[t]} for t in range(T)]
    log(f'voxel grids ready ({n_events} events in {vox_dt*1e3:.1f} ms)')

    with torch.no_grad():
        # one-time setup (untimed, part of model initialisation): each pipeline slot needs one eager call
        # (workspace allocation, kernel attributes) and one more to capture its hipGraph
        model.set_tuning('pipeline', args.pipeline)
        for i in range(2 * args.pipeline):
            model(inputs)
        model.wait()
        torch.cuda.synchronize(device)
        log('workspaces allocated, launch graphs captured')
        for i in range(args.warmup):
            model(inputs)
        model.wait()
        torch.cuda.synchronize(device)
        log(f'{args.warmup} warmup steps done')
        L = _lib.lib()
        barrier()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            model(inputs)
        model.wait()
        torch.cuda.synchronize(device)
        barrier()
        elapsed = time.perf_counter() - t0
        # Kernel roofline: HIP-event spans around every launch of the recurrent step kernel.  Events
        # recorded inside a replayed hipGraph cannot be read back, so the spans come from a few extra
        # EAGER, un-pipelined steps run right after the timed region (same inputs, same kernels).
        model.set_tuning('pipeline', 1)
        model.set_tuning('graph', 0)
        L.bde_profile_reset(model._h, 1)
        for _ in range(min(args.steps, 3)):
            model(inputs)
        torch.cuda.synchronize(device)
    elapsed = max_over_ranks(elapsed, device)
    log(f'timed region done: {elapsed:.3f} s for {args.steps} steps')

    if rank == 0:
        import ctypes as C
        frames = args.steps * T * B * world
        ms = C.c_double()
        cnt = C.c_int64()
        level = 0
        L.bde_profile_get(model._h, b'lstm0', C.byref(ms), C.byref(cnt))
        fl = lstm_flops_per_launch(cfg, B, H, W, level)
        avg_s = (ms.value / max(cnt.value, 1)) * 1e-3
        # the first step of a sweep starts from h = 0 and skips the contraction (one launch in T): its span is
        # in the total, its contraction flops are not
        n_eager = min(args.steps, 3)
        pointwise = 2 * B * (H >> 1) * (W >> 1) * 20.0 * cfg.enc_out(0)
        flops_total = (cnt.value - n_eager) * fl + n_eager * pointwise
        achieved = flops_total / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
        # encoder stage of the north star (encoder convs + gate convs + recurrent steps, SURVEY.md §8d: 45.9 GFLOP per
        # 184x240 frame): HIP-event spans of the same eager steps
        enc_ms = 0.0
        for nm in [b'enc_conv', b'gates_x'] + [b'lstm%d' % l for l in range(cfg.num_encoders)]:
            t_, c_ = C.c_double(), C.c_int64()
            L.bde_profile_get(model._h, nm, C.byref(t_), C.byref(c_))
            enc_ms += t_.value
        enc_flops = 0.0
        for l in range(cfg.num_encoders):
            cin, cout = cfg.enc_in(l), cfg.enc_out(l)
            hw = (H >> (l + 1)) * (W >> (l + 1))
            # both directions: stride-2 k x k conv, 3x3 gate conv on [x | h], ~20 flop per (pixel, channel) pointwise
            enc_flops += 2 * B * T * hw * (2.0 * cout * cin * cfg.ks ** 2 + 2.0 * 4 * cout * 2 * cout * 9 + 20.0 * cout)
        enc_flops -= sum(2 * B * (H >> (l + 1)) * (W >> (l + 1)) * 2.0 * 4 * cfg.enc_out(l) ** 2 * 9 for l in range(cfg.num_encoders))  # h = 0 at the first step
        enc_flops *= n_eager
        enc_tf = enc_flops / (enc_ms * 1e-3) / 1e12 if enc_ms > 0 else 0.0
        traffic = None
        try:   # HBM bytes per launch of the same kernel from the committed rocprofv3 --pmc passes (raw counters)
            with open(os.path.join(REPO, 'profiles', 'r1b_lstm0_pmc.json')) as f:
                traffic = json.load(f)['hbm_bytes_per_launch_raw']
        except Exception:
            pass
        out = {
            'metric': 'reconstructed frames/sec at 5x240x180 voxels, seq_len=16',
            'value': frames / elapsed, 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'BDE2VID.forward config A (5 bins, 32 ch, depths [4,0,6], 16 heads, D=3), '
                                   f'{T} frames of 5x{args.height}x{args.width} (padded {H}x{W}), batch {B}, '
                                   f'random-init formula weights, one independent sequence per step per GPU',
                       'seq_len': T, 'height': args.height, 'width': args.width, 'batch': B,
                       'parallelism': f'{world} replicas, sequences sharded, 1 RCCL weight broadcast; per GPU '
                                      f'{args.pipeline} independent sequence(s) in flight (double-buffered workspaces), '
                                      f'launch sequence replayed from a hipGraph'},
            'roofline': {'kernel': f'lstm16_step_kernel<{lstm_shape(H // 2, W // 2, cfg.basechannels * 2, B)}> (level-0 recurrent ConvLSTM step, both directions)',
                         'bound': 'mfma', 'achieved': achieved, 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': achieved / FP32_MFMA_PEAK_TFLOPS, 'traffic': traffic,
                         'launches': int(cnt.value), 'avg_us': avg_s * 1e6, 'flops_per_launch': fl,
                         'flops_counted': flops_total,
                         'encoder_stage': {'what': 'encoder convs + gate convs + recurrent steps of all levels (the fused '
                                                   'conv+ConvLSTM encoder of the north star), same eager steps',
                                           'flops': enc_flops, 'ms': enc_ms, 'achieved': enc_tf, 'unit': 'TFLOP/s',
                                           'frac': enc_tf / FP32_MFMA_PEAK_TFLOPS},
                         'measured': 'HIP events on the launch stream around each launch, over 3 eager un-pipelined '
                                     'steps run right after the timed region (events inside hipGraph replays cannot be read); '
                                     'achieved = flops of all those launches (the first step of a sweep has no contraction) / their total time'},
            'voxelize': {'events_per_s': n_events / vox_dt, 'events': n_events,
                         'note': 'HIP scatter incl. H2D of the events; outside the timed region',
                         'native_columns': voxel_native_rate(T, (args.height, args.width), device)},
        }
        if world == 1 and not args.no_cpu_baseline:
            log(f'cpu baseline on {host_cores()} host cores ...')
            out['cpu_baseline'] = cpu_baseline(cfg, sd_holder['sd'], H, W, T)
        print(json.dumps(out), file=json_out, flush=True)
    barrier()
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
