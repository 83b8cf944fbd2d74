#!/usr/bin/env python3
"""Merge two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace) of `bench.py --pipeline 1`
into profiles/<tag>_pmc.json: HBM bytes per launch of every kernel, keyed by the span names bench.py uses.

usage: tools/pmc_collect.py <tag> <fetch_dir> <write_dir> [<sq_dir>]
FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of 1 KB = 1024 B?  rocprofv3 defines them as
(TCC_EA0_RDREQ_32B*32 + (RDREQ - RDREQ_32B)*64) / 1024, i.e. KiB: converted with 1024 here.  On gfx950 a wide
(16 B/lane) coalesced read is tallied at half its bytes (MI355X_MICROARCH.md, HBM): `fetch_bytes` below is the RAW
counter; `hbm_bytes_per_launch` = raw fetch + write."""
import collections
import csv
import glob
import json
import os
import re
import sys

# bench.py span -> predicate on (kernel name, grid size): a span can share a kernel template with others; the grid tells them apart
SPAN_KERNELS = [
    ('winblock0', r'winblock_sb_kernel|winblock_kernel'),
    ('wide_*(tokgemm)', r'tokgemm_kernel'),
    ('wide_core2', r'wide_core_kernel|attn_tok16_kernel'),
    ('wide_mlp2', r'mlp_fused_kernel'),
    ('wide_kv_all2', r'tokgemm_sb_kernel'),
    ('lstm0', r'lstm_sb_step_kernel<4, 1, 2|lstm16_step_kernel<1, 128, 2'),
    ('lstm1', r'lstm_sb_step_kernel<2, 2, 2|lstm16_step_kernel<1, 64, 1'),
    ('lstm2', r'lstm_sb_step_kernel<1, 4, 3|lstm16_step_kernel<2, 32, 1'),
    ('gates_x*', r'conv_sb_kernel<3, 1|conv_vec_kernel<3, 1'),
    ('split_bf16', r'split_bf16_kernel'),
    ('enc_conv*', r'conv_sb_kernel<5, 2|conv_vec_kernel<5, 2'),
    ('dec_conv*+head', r'conv_sb_kernel<5, 1|conv_vec_kernel<5, 1'),
    ('gates_x2(dword path)', r'conv_mfma_kernel<3, 1'),
    ('chain(pw_gemm)', r'pw_gemm_kernel'),
    ('chain_core2', r'attn_mfma16_kernel'),
    ('dec_up*', r'upsample2x_sum_kernel'),
    ('pred', r'pred_kernel'),
    ('merge*', r'add2_kernel'),
    ('voxel_native', r'voxel_tile_kernel|voxel_scatter_native_kernel|voxel_bucket_kernel|voxel_tile_from_buckets_kernel'),
]


def load(d):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    if not f:
        return {}
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        name = re.sub(r'\(.*', '', r['Kernel_Name'])
        dur = None
        if r.get('Start_Timestamp') and r.get('End_Timestamp'):
            dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        key = (name, r['Grid_Size'])
        agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
        if dur is not None:
            agg[key]['_us'].append(dur)
    return agg


def main():
    tag, fdir, wdir = sys.argv[1:4]
    sqdir = sys.argv[4] if len(sys.argv) > 4 else None
    fetch, write = load(fdir), load(wdir)
    sq = load(sqdir) if sqdir else {}
    kernels = {}
    for key in sorted(set(fetch) | set(write)):
        name, grid = key
        f = fetch.get(key, {}).get('FETCH_SIZE', [])
        w = write.get(key, {}).get('WRITE_SIZE', [])
        us = fetch.get(key, {}).get('_us', []) or write.get(key, {}).get('_us', [])
        ent = dict(kernel=name, grid=grid, dispatches=max(len(f), len(w)),
                   fetch_bytes=sum(f) / len(f) * 1024 if f else None,
                   write_bytes=sum(w) / len(w) * 1024 if w else None,
                   avg_us=sum(us) / len(us) if us else None)
        if ent['fetch_bytes'] is not None and ent['write_bytes'] is not None:
            ent['hbm_bytes_per_launch'] = ent['fetch_bytes'] + ent['write_bytes']
        if key in sq:
            ent['sq'] = {k: sum(v) / len(v) for k, v in sq[key].items() if k != '_us'}
        kernels[f'{name} grid={grid}'] = ent
    # the busiest (kernel, grid) per span pattern
    spans = {}
    for span, pat in SPAN_KERNELS:
        cands = [e for e in kernels.values() if re.search(pat, e['kernel']) and e.get('hbm_bytes_per_launch') is not None]
        if not cands:
            continue
        if '*' in span or '(' in span:
            spans[span] = sorted(cands, key=lambda e: -(e['avg_us'] or 0) * e['dispatches'])
        else:
            spans[span] = max(cands, key=lambda e: (e['avg_us'] or 0) * e['dispatches'])
    # the workload the passes ran on (bench.py's default command: config 2); bench.py quotes the bytes for this workload only
    wl = json.loads(os.environ.get('BDE_PMC_WORKLOAD', '{"T": 16, "B": 1, "H": 184, "W": 240}'))
    out = dict(tag=tag, workload=wl,
               note='rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (each with --kernel-trace only) over '
                    '`bench.py --pipeline 1 --steps 2 --warmup 1 --no-cpu-baseline`; per-launch averages; FETCH_SIZE raw '
                    '(gfx950 tallies a 16 B/lane coalesced read at half its bytes, MI355X_MICROARCH.md HBM section)',
               spans=spans, kernels=kernels)
    os.makedirs('profiles', exist_ok=True)
    with open(f'profiles/{tag}_pmc.json', 'w') as fo:
        json.dump(out, fo, indent=1)
    for span, e in spans.items():
        for ee in (e if isinstance(e, list) else [e]):
            print(f'{span:22s} {ee["kernel"][:60]:60s} grid {ee["grid"]:>9s} n={ee["dispatches"]:5d} '
                  f'fetch {ee["fetch_bytes"]/1e6:9.2f} MB  write {ee["write_bytes"]/1e6:9.2f} MB  {ee["avg_us"] or 0:8.1f} us')


if __name__ == '__main__':
    main()
