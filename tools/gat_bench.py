#!/usr/bin/env python3
"""Micro-benchmark of the graph-attention core (fused forward kernel, composed backward) at the B=256 step's shape.
usage: gat_bench.py [R N M A H]      (defaults 512 36 36 1024 4)
       gat_bench.py --kernel LIB [R N M A H]   forward kernel only, 30 back-to-back launches through the C ABI of LIB
                                                (any build of libekaid_hip.so: same-box comparisons against older trees)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ekaid_amd import hip, ops  # noqa: E402


def kernel_only(lib_path, R, N, M, A, H):
    import ctypes
    lib = ctypes.CDLL(lib_path)
    P = ctypes.c_void_p
    lib.ekaid_gat_fused_fwd.argtypes = [P] * 6 + [ctypes.c_int] * 6 + [P] * 3
    lib.ekaid_gat_fused_fwd.restype = ctypes.c_int
    g = torch.Generator().manual_seed(0)
    q, k, S = (torch.randn(R, n, A, generator=g).cuda() for n in (N, M, N))
    cond = (torch.rand(R, N, N, generator=g) > 0.4).float().cuda()
    lb = (torch.randn(R, N, N, generator=g) * 0.3).cuda()
    geo = torch.relu(torch.randn(R, N, M, H, generator=g)).cuda()
    Pm, out = torch.empty(R, N, H, M, device='cuda'), torch.empty(R, N, H * A, device='cuda')
    st = torch.cuda.current_stream().cuda_stream

    def launch():
        rc = lib.ekaid_gat_fused_fwd(q.data_ptr(), k.data_ptr(), S.data_ptr(), cond.data_ptr(), lb.data_ptr(), geo.data_ptr(),
                                     R, N, M, A, H, 1, Pm.data_ptr(), out.data_ptr(), st)
        assert rc == 0, rc
    for _ in range(5):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(30):
        launch()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 30
    nbytes = R * (3 * N * A * 4 + N * H * A * 4 + N * H * M * 4)
    print('kernel    %8.1f us  %.2f TB/s of %d MB algorithmic  (R=%d N=%d M=%d, 30 back-to-back launches)' % (
        us, nbytes / us / 1e6, nbytes >> 20, R, N, M))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == '--kernel':
        shape = [int(x) for x in (sys.argv[3:8] or (512, 36, 36, 1024, 4))]
        return kernel_only(sys.argv[2], *shape)
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
    print('backward  %8.1f us  %.2f TB/s of %d MB algorithmic (composed: batched GEMMs + softmax backward + the dq/dk kernel)' % (
        bwd, bytes_bwd / bwd / 1e6, bytes_bwd >> 20))


if __name__ == '__main__':
    main()
