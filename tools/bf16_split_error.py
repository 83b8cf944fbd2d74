#!/usr/bin/env python3
"""Error of an fp32 GEMM emulated on bf16 matrix cores by splitting each operand into three bf16 terms
(a = a1 + a2 + a3, exact to 24 bits), against the fp32 GEMM itself.  CPU only; evidence for DESIGN.md §7.

    bf16x6: a1b1 + a1b2 + a2b1 + a2b2 + a1b3 + a3b1   (every term >= 2^-16 relative)
    bf16x3: a1b1 + a1b2 + a2b1                         (terms >= 2^-8)
"""
import torch

torch.manual_seed(0)


def split3(x):
    a1 = x.to(torch.bfloat16)
    r = x - a1.float()
    a2 = r.to(torch.bfloat16)
    a3 = (r - a2.float()).to(torch.bfloat16)
    return a1.float(), a2.float(), a3.float()


M, K, N = 256, 576, 4096                     # gate conv of level 0: 4C x 9C x pixels
A, B = torch.randn(M, K) * 0.05, torch.randn(K, N)
ref = A.double() @ B.double()
a, b = split3(A), split3(B)
mm = lambda x, y: x.double() @ y.double()    # bf16 products are exact in fp32; the accumulation is fp32 on the hardware
cands = {
    'fp32 matmul': (A @ B).double(),
    'bf16x6': mm(a[0], b[0]) + mm(a[0], b[1]) + mm(a[1], b[0]) + mm(a[1], b[1]) + mm(a[0], b[2]) + mm(a[2], b[0]),
    'bf16x3': mm(a[0], b[0]) + mm(a[0], b[1]) + mm(a[1], b[0]),
    'bf16x1': mm(a[0], b[0]),
}
scale = ref.abs().max().item()
for name, v in cands.items():
    print(f'{name:12s} max abs err / max|ref| = {((v - ref).abs().max() / scale).item():.2e}')
