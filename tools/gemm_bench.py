"""GEMM micro-benchmark over library variants (tuning aid, not part of the product)."""
import glob, os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ekaid_amd import hip, ops
BIG = [('nt', 18432, 1024, 1024), ('nt', 18432, 1024, 4096), ('nn', 18432, 1024, 1024), ('nn', 18432, 4096, 1024),
       ('tn', 1024, 1024, 18432), ('tn', 1024, 4096, 18432)]
SMALL = [('nt', 256, 6656, 512), ('nt', 256, 2048, 512), ('nt', 256, 2048, 1024), ('nt', 256, 1024, 2048), ('nt', 256, 3072, 1024),
         ('nn', 256, 512, 6656), ('nn', 256, 512, 2048), ('nn', 256, 1024, 2048), ('nn', 256, 2048, 1024), ('nn', 256, 1024, 3072)]
CUSTOM = os.environ.get('GEMM_BENCH_SHAPES')     # e.g. "nt:9216:128:1024,tn:128:1024:9216"
SHAPES = [(t.split(':')[0],) + tuple(int(x) for x in t.split(':')[1:]) for t in CUSTOM.split(',')] if CUSTOM else SMALL if os.environ.get('GEMM_BENCH_SET') == 'small' else (BIG if os.environ.get('GEMM_BENCH_SET') == 'big' else BIG + SMALL)
US = os.environ.get('GEMM_BENCH_US')          # print microseconds per call instead of TFLOP/s
def run(lib):
    res = []
    for mode, M, N, K in SHAPES:
        A = torch.randn(M * K, device='cuda'); B = torch.randn(K * N, device='cuda'); C = torch.empty(M * N, device='cuda')
        ta, tb, lda, ldb = {'nt': (0, 1, K, K), 'nn': (0, 0, K, N), 'tn': (1, 0, M, N)}[mode]
        f = lambda: ops.gemm(A, B, C, M, N, K, lda, ldb, N, ta, tb)
        for _ in range(10): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        if os.environ.get('GEMM_BENCH_GRAPH'):        # 20 calls replayed from one hipGraph: no host time in the figure
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                f()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(n): f()
            g.replay(); torch.cuda.synchronize()
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        else:
            e0.record()
            for _ in range(n): f()
            e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / n if US else 2.0 * M * N * K * n / (e0.elapsed_time(e1) * 1e-3) / 1e12)
        if os.environ.get('GEMM_BENCH_CHECK'):
            a2 = A.view(K, M).t() if ta else A.view(M, K); b2 = B.view(N, K).t() if tb else B.view(K, N)
            err = float((C.view(M, N) - a2 @ b2).abs().max() / (a2 @ b2).abs().max())
            assert err < 1e-4, (mode, M, N, K, err)
    return res
print('%-22s' % 'variant' + ''.join('%14s' % ('%s %dx%dx%d' % s)[:13] for s in SHAPES))
for path in sorted(glob.glob(os.path.join(os.path.dirname(hip.LIB_PATH), 'libekaid_v*.so'))) or [hip.LIB_PATH]:
    hip._lib = None
    hip._lib = hip.load(path)
    print('%-22s' % os.path.basename(path) + ''.join('%14.1f' % x for x in run(hip._lib)), flush=True)
