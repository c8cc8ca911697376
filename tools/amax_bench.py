"""Microseconds per bounds pass (ekaid_amax_rc) on the operand shapes of the 256- and 32-pair steps: rows + columns, rows only, columns
only, against a plain device-to-device copy of the same matrix as the bandwidth yardstick.  EKAID_AMAX_PANELS=<n> overrides the row-panel
count (the same-address atomic tail grows with it).      python tools/amax_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ekaid_amd import hip, ops  # noqa: E402

hip.load()
dev = torch.device('cuda', 0)
SHAPES = [(18432, 1024), (18432, 4096), (9216, 1024), (4096, 6656), (5120, 3072), (2304, 1024), (2304, 4096), (1152, 1024), (1024, 1024)]
print('panels override: %s' % os.environ.get('EKAID_AMAX_PANELS', '-'))
print('%-14s %8s | %9s %9s %9s | %9s' % ('rows x cols', 'MB', 'rows+cols', 'rows', 'cols', 'copy'))


def timed(fn, n=40):
    for _ in range(3):
        fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for rows, cols in SHAPES:
    xs = [torch.randn(rows, cols, device=dev) for _ in range(6)]          # rotated: 6 x 75 MB does not sit in the 256 MB last-level cache
    out = torch.empty_like(xs[0])

    def run(wr, wc):
        def fn(i):
            if i % 16 == 0:
                ops.bounds_pool_reset(dev)
            ops.amax_rc(xs[i % 6], rows, cols, cols, want_rows=wr, want_cols=wc)
        return timed(fn)
    both, ronly, conly = run(True, True), run(True, False), run(False, True)
    cp = timed(lambda i: out.copy_(xs[i % 6]))
    print('%-14s %8.1f | %9.1f %9.1f %9.1f | %9.1f' % ('%dx%d' % (rows, cols), rows * cols * 4 / 1e6, both, ronly, conly, cp))
