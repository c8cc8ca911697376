"""One GEMM shape launched a few times (PMC collection target; tuning aid).
usage: pmc_gemm.py <nt|nn|tn> M N K"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ekaid_amd import hip, ops
hip.load()
mode, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
A = torch.randn(M * K, device='cuda'); B = torch.randn(K * N, device='cuda'); C = torch.empty(M * N, device='cuda')
ta, tb, lda, ldb = {'nt': (0, 1, K, K), 'nn': (0, 0, K, N), 'tn': (1, 0, M, N)}[mode]
kw = {}
if ops.get_gemm_mode() == ops.GEMM_F16X2 and os.environ.get('PMC_GEMM_BOUNDS', '1') != '0':
    # the two-term f16 split needs the operand bounds (include/ekaid_hip.h: which maxima go where); taken once, outside the counted launches
    ar, ac = ops.amax_rc(A, *((K, M, M) if ta else (M, K, K)))
    br, bc = ops.amax_rc(B, *((N, K, K) if tb else (K, N, N)))
    kw = dict(a_amax=ac if ta else ar, b_amax=br if tb else bc)
for _ in range(6):
    ops.gemm(A, B, C, M, N, K, lda, ldb, N, ta, tb, **kw)
torch.cuda.synchronize()
