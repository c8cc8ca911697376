#!/usr/bin/env python3
"""rocprofv3 target: the greedy _sample of test_mimic.py:119-122 alone (batch 64, 90 steps, freshly initialised weights), 100 times after one
detector forward -- `rocprofv3 --kernel-trace --stats -- python3 tools/greedy_trace.py`, then tools/stats_read.py <csv> 100
(profiles/r05_greedy_sample_kernels.txt)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import torch
from ekaid_amd.config import bench_cfg
from ekaid_amd.engine import detector_forward
from ekaid_amd.formula_init import build_models
from ekaid_amd.synthetic import make_batch
dev = torch.device('cuda:0')
cfg = bench_cfg(D=2048, seq_length=90, graph='all')
cd0, sp0 = build_models(cfg, 147, 148, dev, train=False)
cd0.eval(); sp0.eval()
b64 = {k: v.to(dev) for k, v in make_batch(B=64, seed=77, N=36, D=2048, Lq=20, T=16, seq_length=90, V=148).items()}
with torch.no_grad():
    outs = detector_forward(cfg, cd0, b64, 'all')
    for _ in range(100):
        sp0._sample(outs[3], outs[4], outs[5], None, cfg, sample_max=1)
    torch.cuda.synchronize()
