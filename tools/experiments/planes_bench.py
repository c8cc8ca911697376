#!/usr/bin/env python3
"""GEMM micro-benchmark: in-kernel split (ekaid_gemm_f32) vs pre-split planes (ekaid_gemm_planes, variant chosen by
EKAID_PLANES_VARIANT) on the dominant shapes of the B=256 training step; interleaved rounds in one process."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ekaid_amd import hip, ops  # noqa: E402

SHAPES = [(0, 1, 18432, 1024, 1024), (0, 0, 18432, 1024, 1024), (1, 0, 1024, 1024, 18432), (0, 1, 18432, 1024, 4096),
          (0, 0, 18432, 4096, 1024), (1, 0, 1024, 4096, 18432), (0, 1, 9216, 1024, 2048)]


def main():
    hip.load()
    dev = 'cuda'
    g = torch.Generator().manual_seed(0)
    print('variant', os.environ.get('EKAID_PLANES_VARIANT', 'default'))
    for tA, tB, M, N, K in SHAPES:
        A = torch.randn(*((M, K) if not tA else (K, M)), generator=g).to(dev)
        B = torch.randn(*((N, K) if tB else (K, N)), generator=g).to(dev)
        lda, ldb = A.shape[1], B.shape[1]
        Ap, Bp = ops.split_planes(A), ops.split_planes(B)
        c0, c1 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
        f = lambda: ops.gemm(A, B, c0, M, N, K, lda, ldb, N, tA, tB)                              # noqa: E731
        p = lambda: ops.gemm(A, B, c1, M, N, K, lda, ldb, N, tA, tB, A_planes=Ap, B_planes=Bp)    # noqa: E731
        for _ in range(3):
            f(); p()                                                                             # noqa: E702
        torch.cuda.synchronize()
        err = float((c0 - c1).abs().max() / c0.abs().max())
        ts = {'f32': [], 'planes': []}
        for _ in range(5):
            for name, fn in (('f32', f), ('planes', p)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                ts[name].append(e0.elapsed_time(e1) / 10)
        fl = 2.0 * M * N * K
        med = {k: sorted(v)[len(v) // 2] for k, v in ts.items()}
        print('%-32s f32-split %7.1f us %6.1f TF | planes %7.1f us %6.1f TF (x6 = %.0f TF executed, %.3f of 2516.6) | maxrel diff %.1e' % (
            (tA, tB, M, N, K), med['f32'] * 1e3, fl / med['f32'] / 1e9, med['planes'] * 1e3, fl / med['planes'] / 1e9,
            6 * fl / med['planes'] / 1e9, 6 * fl / med['planes'] / 1e9 / 2516.6, err), flush=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.split_planes(A)
        e1.record()
        torch.cuda.synchronize()
        print('    split of A (%d elements): %.1f us' % (A.numel(), e0.elapsed_time(e1) * 100), flush=True)


if __name__ == '__main__':
    main()
