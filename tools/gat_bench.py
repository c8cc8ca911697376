#!/usr/bin/env python3
"""Micro-benchmark of the graph-attention core (fused forward kernel, composed backward) at the B=256 step's shape.
usage: gat_bench.py [R N M A H]      (defaults 512 36 36 1024 4)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ekaid_amd import hip, ops  # noqa: E402


def main():
    hip.load()
    R, N, M, A, H = [int(x) for x in (sys.argv[1:6] or (512, 36, 36, 1024, 4))]
    dev = 'cuda'
    g = torch.Generator().manual_seed(0)
    q, k, S = (torch.randn(R, n, A, generator=g).to(dev).requires_grad_(True) for n in (N, M, N))
    cond = (torch.rand(R, N, N, generator=g) > 0.4).float().to(dev)
    lb = (torch.randn(R, N, N, generator=g) * 0.3).to(dev).requires_grad_(True)
    geo = torch.relu(torch.randn(R, N, M, H, generator=g)).to(dev).requires_grad_(True)
    w = torch.randn(R, N, H * A, generator=g).to(dev)

    def run():
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        out, P = ops.gat_attention(q, k, S, cond, lb, geo, H, 1)
        e1.record()
        torch.autograd.grad(out, [q, k, S, lb, geo], w)
        e2.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3, e1.elapsed_time(e2) * 1e3
    for _ in range(3):
        run()
    res = [run() for _ in range(9)]
    med = lambda xs: sorted(xs)[len(xs) // 2]         # noqa: E731
    fwd, bwd = med([x[0] for x in res]), med([x[1] for x in res])
    bytes_fwd = R * (3 * N * A * 4 + N * H * A * 4 + N * H * M * 4)
    bytes_bwd = R * (N * H * A * 4 + 3 * N * A * 4 + N * H * M * 4 + 3 * N * A * 4)
    print('R=%d N=%d M=%d A=%d H=%d' % (R, N, M, A, H))
    print('forward   %8.1f us  %.2f TB/s of %d MB algorithmic' % (fwd, bytes_fwd / fwd / 1e6, bytes_fwd >> 20))
    print('backward  %8.1f us  %.2f TB/s of %d MB algorithmic (composed: 4 batched GEMMs + softmax backward)' % (
        bwd, bytes_bwd / bwd / 1e6, bytes_bwd >> 20))


if __name__ == '__main__':
    main()
