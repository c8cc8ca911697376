"""The two-term f16 split (EKAID_GEMM_F16X2) against the exact 3-way bf16 split and the f32 matrix pipe: error against float64 and
microseconds per product on the dominant shapes of the 256-pair step, plus what the operand bounds (ekaid_amax_rc) cost.

    python tools/f16x2_bench.py [--quick]            (GPU box; prints a table, see profiles/r06_f16x2_*.txt)

Data: N(0, 1) entries with a per-row (A) / per-column (B) scale exp(s * N(0, 1)), s = 0 (flat), 1.15 (one decade per sigma) and 3 (the
spread of tests/test_hip_ops.py::test_gemm_split_accuracy: ~2^26 between rows).  Error = max and mean over the output of
|C - C64| / (|A| |B|)_ij (the element's own condition scale)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ekaid_amd import hip, ops                             # noqa: E402

QUICK = '--quick' in sys.argv
SHAPES = [('nt', 18432, 1024, 1024), ('nt', 18432, 1024, 4096), ('nn', 18432, 1024, 1024), ('nn', 18432, 4096, 1024),
          ('tn', 1024, 1024, 18432), ('tn', 1024, 4096, 18432), ('nt', 2304, 1024, 1024), ('nn', 2304, 1024, 1024), ('tn', 1024, 1024, 2304)]
if QUICK:
    SHAPES = SHAPES[:1] + SHAPES[2:3] + SHAPES[4:5]
MODES = (('f32', ops.GEMM_F32_MFMA), ('bf16x3', ops.GEMM_BF16_SPLIT), ('f16x2', ops.GEMM_F16X2))


def operands(mode, M, N, K, spread, seed):
    g = torch.Generator(device='cuda').manual_seed(seed)
    A = torch.randn(M, K, device='cuda', generator=g) * torch.exp(spread * torch.randn(M, 1, device='cuda', generator=g))
    B = torch.randn(K, N, device='cuda', generator=g) * torch.exp(spread * torch.randn(1, N, device='cuda', generator=g))
    if mode == 'nt':
        return A, B, A.contiguous(), B.t().contiguous(), 0, 1, K, K
    if mode == 'nn':
        return A, B, A.contiguous(), B.contiguous(), 0, 0, K, N
    return A, B, A.t().contiguous(), B.contiguous(), 1, 0, M, N


def bounds(a, b, ta, tb, M, N, K):
    """a_amax / b_amax of one product from the stored operands (include/ekaid_hip.h: which maxima go where)."""
    ar, ac = ops.amax_rc(a, a.shape[0], a.shape[1], a.shape[1])
    br, bc = ops.amax_rc(b, b.shape[0], b.shape[1], b.shape[1])
    return (ac if ta else ar), (br if tb else bc)


def timed(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            f()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    hip.load()
    old = ops.get_gemm_mode()
    print('%-22s %6s | %s | %s' % ('shape', 'spread', '   '.join('%-8s max      mean    ' % n for n, _ in MODES), 'us: ' + ' '.join('%8s' % n for n, _ in MODES) + '   bounds A+B'))
    try:
        for (mode, M, N, K) in SHAPES:
            for spread in ((0.0,) if QUICK else (0.0, 1.15, 3.0)):
                A, B, a, b, ta, tb, lda, ldb = operands(mode, M, N, K, spread, 7 + M + N + K)
                want = A.double() @ B.double()
                mag = A.double().abs() @ B.double().abs()
                C = torch.empty(M, N, device='cuda')
                am, bm = bounds(a, b, ta, tb, M, N, K)
                errs, us = [], []
                for name, md in MODES:
                    ops.set_gemm_mode(md)
                    kw = dict(a_amax=am, b_amax=bm) if md == ops.GEMM_F16X2 else {}
                    C.fill_(float('nan'))
                    c0 = hip.launch_counts()
                    ops.gemm(a, b, C, M, N, K, lda, ldb, N, ta, tb, **kw)
                    c1 = hip.launch_counts()
                    key = {'f32': 'gemm_f32', 'bf16x3': 'gemm_split3', 'f16x2': 'gemm_split2'}[name]
                    assert c1[key] - c0[key] == 1, (name, {k: c1[k] - c0[k] for k in c0})
                    e = (C.double() - want).abs() / mag
                    errs.append((float(e.max()), float(e.mean())))
                    if spread == 0.0:
                        us.append(timed(lambda: ops.gemm(a, b, C, M, N, K, lda, ldb, N, ta, tb, **kw)))
                tb_us = timed(lambda: bounds(a, b, ta, tb, M, N, K)) if spread == 0.0 else None
                print('%-22s %6.2f | %s | %s' % ('%s %dx%dx%d' % (mode, M, N, K), spread,
                                                 '   '.join('%-8s %.2e %.2e' % ('', e[0], e[1]) for e in errs),
                                                 ('    ' + ' '.join('%8.1f' % u for u in us) + '   %8.1f' % tb_us) if us else ''), flush=True)
                del A, B, a, b, want, mag, C
                torch.cuda.empty_cache()
        # subnormal operands of the matrix pipe: 2^-20 . 2^-4 in f16 x f16 -> f32 must come out exactly (not flushed)
        M = N = 128
        K = 32
        a = torch.full((M, K), 2.0 ** -20, device='cuda')
        a[:, 0] = 1.0                                     # the row maximum: the rest of the row sits 2^20 below it (h subnormal after scaling? no: 2^-5; l = 0)
        b = torch.full((N, K), 0.25, device='cuda')
        C = torch.empty(M, N, device='cuda')
        ops.set_gemm_mode(ops.GEMM_F16X2)
        ar, _ = ops.amax_rc(a, M, K, K)
        br, _ = ops.amax_rc(b, N, K, K)
        ops.gemm(a, b, C, M, N, K, K, K, N, 0, 1, a_amax=ar, b_amax=br)
        want = 0.25 + 31 * 2.0 ** -22
        print('wide-row check (row max 1, rest 2^-20): got %.10e want %.10e' % (float(C[0, 0]), want))
        a = torch.full((M, K), 2.0 ** -30 * 1.000123, device='cuda')
        a[:, 0] = 1.0                                     # the rest of the row sits 2^30 below the maximum: f16 subnormals after scaling
        ar, _ = ops.amax_rc(a, M, K, K)
        ops.gemm(a, b, C, M, N, K, K, K, N, 0, 1, a_amax=ar, b_amax=br)
        want = 0.25 + 31 * 2.0 ** -32 * 1.000123
        print('subnormal check (row max 1, rest 2^-30): got - 0.25 = %.6e want %.6e (zero here = the matrix pipe flushes f16 subnormals)' % (float(C[0, 0].double() - 0.25), want - 0.25))
    finally:
        ops.set_gemm_mode(old)


if __name__ == '__main__':
    main()
