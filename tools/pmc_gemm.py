"""One GEMM shape launched a few times (PMC collection target; tuning aid).
usage: pmc_gemm.py <nt|nn|tn> M N K"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ekaid_amd import hip, ops
hip.load()
mode, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
A = torch.randn(M * K, device='cuda'); B = torch.randn(K * N, device='cuda'); C = torch.empty(M * N, device='cuda')
ta, tb, lda, ldb = {'nt': (0, 1, K, K), 'nn': (0, 0, K, N), 'tn': (1, 0, M, N)}[mode]
for _ in range(6):
    ops.gemm(A, B, C, M, N, K, lda, ldb, N, ta, tb)
torch.cuda.synchronize()
