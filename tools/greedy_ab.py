#!/usr/bin/env python3
"""bench.py's extra.greedy_decode_b64 on its own (test_mimic.py:109-122: eval detector forward + greedy _sample, batch 64, 90 steps),
run from the tree given as argv[1] (the repo root, or an export of another commit under _base/): the A/B aid for the decode path.
usage: greedy_ab.py <tree root> [reps]      prints one JSON line"""
import json
import os
import sys
import time

root = os.path.abspath(sys.argv[1])
sys.path.insert(0, root)
os.chdir(root)
import torch                                        # noqa: E402
from ekaid_amd.config import bench_cfg            # noqa: E402
from ekaid_amd.engine import detector_forward      # noqa: E402
from ekaid_amd.formula_init import build_models    # noqa: E402
from ekaid_amd.synthetic import make_batch         # noqa: E402

reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device('cuda:0')
cfg = bench_cfg(D=2048, seq_length=90, graph='all')
cd0, sp0 = build_models(cfg, 147, 148, dev, train=False)
cd0.eval(); sp0.eval()                             # noqa: E702
b64 = {k: v.to(dev) for k, v in make_batch(B=64, seed=77, N=36, D=2048, Lq=20, T=16, seq_length=90, V=148).items()}


def decode():
    outs = detector_forward(cfg, cd0, b64, 'all')
    return sp0._sample(outs[3], outs[4], outs[5], None, cfg, sample_max=1)


def timed(fn, n, w):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


out = []
with torch.no_grad():
    for _ in range(reps):
        msd = timed(decode, 10, 2)
        msf = timed(lambda: detector_forward(cfg, cd0, b64, 'all'), 10, 2)
        outs = detector_forward(cfg, cd0, b64, 'all')
        mss = timed(lambda: sp0._sample(outs[3], outs[4], outs[5], None, cfg, sample_max=1), 10, 2)
        out.append((round(msd, 3), round(msf, 3), round((msd - msf) / 90, 4), round(mss, 3)))
print(json.dumps({'tree': root, 'decode_ms, forward_ms, ms_per_decode_step, sample_only_ms': out}))
