"""Weight-gradient product with the bias gradient riding on it (ekaid_gemm_t.a_rowsum) against product + separate column sums.
usage: wgrad_bias_bench.py            (the step's wgrad shapes at B = 256 and B = 32)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from ekaid_amd import hip, ops  # noqa: E402

hip.load()


def timed(fn, n=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


print('%22s %10s %12s %10s' % ('M(out) N(in) K(rows)', 'plain us', '+colsum us', 'fused us'))
for M, N, K in ((1024, 1024, 18432), (1024, 4096, 18432), (4096, 1024, 18432), (1024, 1024, 2304), (1024, 4096, 2304),
                (4096, 1024, 4096), (2048, 1024, 4096), (512, 2048, 4096), (1024, 1024, 6656)):
    dY = torch.randn(K, M, device='cuda')
    X = torch.randn(K, N, device='cuda')
    dW = torch.empty(M, N, device='cuda')
    db = torch.zeros(M, device='cuda')
    L = hip.lib()
    plain = timed(lambda: ops.gemm(dY, X, dW, M, N, K, M, N, N, 1, 0))

    def sep():
        ops.gemm(dY, X, dW, M, N, K, M, N, N, 1, 0)
        hip.check(L.ekaid_colsum_acc(dY.data_ptr(), K, M, M, db.data_ptr(), ops.stream()), 'colsum')
    both = timed(sep)
    fused = timed(lambda: ops.gemm(dY, X, dW, M, N, K, M, N, N, 1, 0, a_rowsum=db))
    print('%6d %6d %7d  %10.1f %12.1f %10.1f' % (M, N, K, plain, both, fused))
