#!/usr/bin/env python3
"""Feasibility probe: capture one whole training step (zero_grad -> adjacency -> detector -> decoder -> loss -> backward ->
Adam) into a HIP graph with torch.cuda.graph and compare eager vs replay time per step at several batch sizes.
Timing aid only (dropout seeds / Adam step are baked into the capture here)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import torch  # noqa: E402

import bench  # noqa: E402
from common import build_models  # noqa: E402
from ekaid_amd import hip, ops  # noqa: E402
from ekaid_amd.config import bench_cfg  # noqa: E402
from ekaid_amd.dp import FlatAdam  # noqa: E402
from ekaid_amd.models.dynamic_speaker_change_pos import DynamicSpeaker  # noqa: E402
from ekaid_amd.synthetic import make_batch  # noqa: E402


def run(B, steps=20):
    dev = torch.device('cuda', 0)
    cfg = bench_cfg(D=2048, seq_length=90, graph='all')
    cd, sp = build_models(cfg, 147, 148, dev, train=True)
    opt = FlatAdam([cd, sp], lr=1e-4)
    hb = make_batch(B=B, seed=1238, **bench.SHAPE)
    T = DynamicSpeaker.count_steps(hb['labels'], 90)
    b = {k: v.to(dev) for k, v in hb.items()}
    ops.manual_seed(1238)
    for _ in range(3):
        bench.train_step(cfg, cd, sp, opt, b, T, 'all')
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        bench.train_step(cfg, cd, sp, opt, b, T, 'all')
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / steps * 1e3
    out = {'B': B, 'eager_ms': round(eager, 3)}
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                bench.train_step(cfg, cd, sp, opt, b, T, 'all')
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        tc = time.perf_counter()
        with torch.cuda.graph(g):
            loss = bench.train_step(cfg, cd, sp, opt, b, T, 'all')
        torch.cuda.synchronize()
        out['capture_s'] = round(time.perf_counter() - tc, 3)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            g.replay()
        torch.cuda.synchronize()
        out['graph_ms'] = round((time.perf_counter() - t0) / steps * 1e3, 3)
        out['loss_after'] = float(loss.item())
    except Exception as e:      # noqa: BLE001
        out['graph_error'] = repr(e)[:500]
    print(json.dumps(out), flush=True)
    del cd, sp, opt
    torch.cuda.empty_cache()


if __name__ == '__main__':
    hip.load()
    for B in [int(x) for x in (sys.argv[1:] or ['256', '64', '32'])]:
        run(B)
