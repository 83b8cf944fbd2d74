#!/usr/bin/env python3
"""Condense a rocprofv3 `*_kernel_stats.csv` into a short per-kernel table (stdout / markdown)."""
import csv
import glob
import re
import sys


def short(name):
    m = re.search(r'conv_mfma_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\w+), (\d+), (\d+)>', name)
    if m:
        ks, st, mt, nt, ck, sk, epi, maxi = m.groups()
        return f'conv_mfma<K{ks} S{st} M{mt} N{nt} CK{ck}{" splitK" if sk == "true" else ""}{" LSTM" if epi == "1" else ""}>'
    m = re.search(r'pw_gemm_kernel<(\d+), (\d+), (\w+)>', name)
    if m:
        return f'pw_gemm<M{m.group(1)} N{m.group(2)}{" splitK" if m.group(3) == "true" else ""}>'
    m = re.search(r'(attn_\w+)<(\d+)', name)
    if m:
        return f'{m.group(1)}<hd{m.group(2)}>'
    return re.sub(r'\(.*', '', name)[:70]


def main(path, steps=None):
    files = glob.glob(path + '/**/*kernel_stats.csv', recursive=True) if not path.endswith('.csv') else [path]
    rows = list(csv.DictReader(open(files[0])))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    print(f'total GPU kernel time {tot / 1e6:.2f} ms' + (f' = {tot / 1e6 / steps:.2f} ms per forward' if steps else ''))
    print(f'{"kernel":58s} {"calls":>7s} {"total ms":>10s} {"avg us":>9s} {"%":>6s}')
    for r in rows:
        if float(r['Percentage']) < 0.01:
            continue
        print(f'{short(r["Name"]):58s} {r["Calls"]:>7s} {float(r["TotalDurationNs"]) / 1e6:10.2f} '
              f'{float(r["AverageNs"]) / 1e3:9.1f} {float(r["Percentage"]):6.2f}')


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else None)
