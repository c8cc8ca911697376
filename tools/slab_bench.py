#!/usr/bin/env python3
"""The recurrent chains' per-step products (raw K-slabs, ekaid_gemm_slabs: the consumer kernel sums them) at 32 / 64 / 128 rows, 20 calls
replayed from one hipGraph: microseconds per call, slab count, and the time the bytes alone would take at 4 TB/s.
usage: slab_bench.py [rows ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                  # noqa: E402
from ekaid_amd import hip, ops               # noqa: E402

# (name, form, N, K) with M = rows: decoder backward q1 q2 q3 ca cl (sequence.hip: spk_plan), GRU carry (backward) and recurrent (forward)
R, A, G1, NW = 512, 1024, 512, 4 * 512 + 16 + 512 + 4 * 512
SHAPES = [('q1 dgates_l.Wih_l', 'nn', A, 4 * R), ('q2 dz2.Wg2', 'nn', G1, A), ('q3 dg1.Wg1', 'nn', R + A, G1), ('ca dgates_a.Whh_a', 'nn', R, 4 * R),
          ('cl dcat.Wcat', 'nn', R, NW), ('gru carry dgh.Whh', 'nn', 1024, 3072), ('gru fwd h.Whh^T', 'nt', 3072, 1024)]


def main():
    rows = [int(x) for x in sys.argv[1:]] or [32, 64, 128]
    hip.load()
    for M in rows:
        for name, form, N, K in SHAPES:
            a = torch.randn(M, K, device='cuda')
            b = torch.randn(K, N, device='cuda') if form == 'nn' else torch.randn(N, K, device='cuda')
            ldb, tb = (N, 0) if form == 'nn' else (K, 1)
            f = lambda: ops.gemm_slabs(a, b, M, N, K, K, ldb, 0, tb)      # noqa: E731
            S = f()
            want = a.double() @ (b.double() if form == 'nn' else b.double().t())
            err = float((S.total().double() - want).abs().max() / want.abs().max())
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(20):
                    f()
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 100 * 1e3
            byt = 4.0 * (M * K + K * N + S.n * M * N)
            print('rows %4d  %-20s %s N=%5d K=%5d  slabs %2d  %6.1f us   (bytes at 4 TB/s: %4.1f us)   rel err %.1e' %
                  (M, name, form, N, K, S.n, us, byt / 4e12 * 1e6, err), flush=True)


if __name__ == '__main__':
    main()
