"""How much does the partial last wave of workgroups cost the 128x128-tile GEMM?  Times one layout at a sweep of row counts
(N = K = 1024 by default) whose tile counts are whole and fractional multiples of the 768 resident workgroup slots (256 CUs x 3),
30 back-to-back launches each; prints tiles, waves, microseconds and f32-equivalent TFLOP/s.   usage: gemm_tail.py [nt|nn|tn] [N K]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from ekaid_amd import hip, ops  # noqa: E402

hip.load()
mode = sys.argv[1] if len(sys.argv) > 1 else 'nn'
N, K = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1024, 1024)
print('layout %s, N=%d K=%d' % (mode, N, K))
print('%8s %7s %6s %9s %8s' % ('M', 'tiles', 'waves', 'us', 'TF-eq'))
for M in (6144, 9216, 12288, 15360, 18432, 21504, 24576, 36864):
    A = torch.randn(M * K, device='cuda')
    B = torch.randn(K * N, device='cuda')
    C = torch.empty(M * N, device='cuda')
    ta, tb, lda, ldb = {'nt': (0, 1, K, K), 'nn': (0, 0, K, N), 'tn': (1, 0, M, N)}[mode]
    for _ in range(5):
        ops.gemm(A, B, C, M, N, K, lda, ldb, N, ta, tb)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(30):
        ops.gemm(A, B, C, M, N, K, lda, ldb, N, ta, tb)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 30
    tiles = (M // 128) * (N // 128)
    print('%8d %7d %6.2f %9.1f %8.1f' % (M, tiles, tiles / 768.0, us, 2.0 * M * N * K / us / 1e6))
