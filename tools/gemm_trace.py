"""Run a list of GEMM shapes back to back (tuning aid; use under rocprofv3 --kernel-trace and read per-dispatch
durations with tools/gemm_trace_read.py).  Shapes: argv = mode,M,N,K ..."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ekaid_amd import hip, ops
shapes = [tuple(a.split(',')) for a in sys.argv[1:]]
REP = 12
bufs = []
for mode, M, N, K in shapes:
    M, N, K = int(M), int(N), int(K)
    bufs.append((mode, M, N, K, torch.randn(M * K, device='cuda'), torch.randn(K * N, device='cuda'), torch.empty(M * N, device='cuda'),
                 torch.randn(M * K, device='cuda'), torch.randn(K * N, device='cuda'), torch.empty(M * N, device='cuda')))
torch.cuda.synchronize()
for mode, M, N, K, A, B, C, A2, B2, C2 in bufs:
    pair = mode.endswith('p')                  # 'ntp' / 'nnp' / 'tnp': two independent products of the shape in one launch (ops.gemm_pair)
    ta, tb, lda, ldb = {'nt': (0, 1, K, K), 'nn': (0, 0, K, N), 'tn': (1, 0, M, N)}[mode[:2]]
    for _ in range(REP):
        if pair:
            ops.gemm_pair(((A, B, C, M, N, K, lda, ldb, N, ta, tb), {}), ((A2, B2, C2, M, N, K, lda, ldb, N, ta, tb), {}))
        else:
            ops.gemm(A, B, C, M, N, K, lda, ldb, N, ta, tb)
    torch.cuda.synchronize()
print('done', len(bufs), REP)
