#!/usr/bin/env python3
"""Debug aid: capture the training step under several configurations, each in its own process (a crash in one does not
hide the others).  usage: python tools/debug_graph.py            (driver)   |   python tools/debug_graph.py <case>"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

CASES = {
    'full_b32': dict(D=2048, B=32, N=36, T=16, seq=90),
    'small': dict(D=256, B=4, N=12, T=5, seq=12),
    'small_noadam': dict(D=256, B=4, N=12, T=5, seq=12, noadam=1),
    'small_noepoch': dict(D=256, B=4, N=12, T=5, seq=12, noepoch=1),
    'small_fwdonly': dict(D=256, B=4, N=12, T=5, seq=12, fwdonly=1),
    'mid': dict(D=256, B=16, N=36, T=16, seq=90),
}


def run(name):
    import torch
    from common import build_models
    from ekaid_amd import hip, ops
    from ekaid_amd.config import bench_cfg
    from ekaid_amd.dp import FlatAdam
    from ekaid_amd.engine import TrainStep, forward_loss
    from ekaid_amd.synthetic import make_batch
    c = CASES[name]
    cfg = bench_cfg(D=c['D'], seq_length=c['seq'])
    host = make_batch(B=c['B'], N=c['N'], D=c['D'], Lq=8, T=c['T'], seq_length=c['seq'], V=148, seed=5)
    b = {k: v.cuda() for k, v in host.items()}
    cd, sp = build_models(cfg, 147, 148, 'cuda', train=True)
    opt = FlatAdam([cd, sp], lr=1e-3)
    ts = TrainStep(cfg, cd, sp, opt, use_graph=True)
    if c.get('noadam'):
        opt.apply_update = lambda: None
    if c.get('noepoch'):
        real = hip.lib().ekaid_dropout_epoch_advance
        hip.lib().ekaid_dropout_epoch_advance = lambda *a: 0
    if c.get('fwdonly'):
        def body(static, T, denom):
            with torch.no_grad():
                loss, sl, _, _ = forward_loss(cfg, cd, sp, static, T, 'all', 'mode2', denom)
            return loss, sl
        opt.zero_grad(); forward_loss(cfg, cd, sp, b, c['T'])[0].backward(); opt.finish()      # noqa: E702
        ts._body = body
    for i in range(3):
        loss, _ = ts(b, c['T'])
    print(name, 'OK loss', float(loss), 'step', opt.step_count, flush=True)
    ts.close()


if __name__ == '__main__':
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        for name in CASES:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), name], capture_output=True, text=True, timeout=300)
            tail = (r.stdout + r.stderr).strip().splitlines()
            keep = [l for l in tail if 'OK' in l or 'Error' in l or 'error' in l or 'Fatal' in l or 'capture' in l][-4:]
            print('%-16s rc=%d  %s' % (name, r.returncode, ' | '.join(keep)[:600]), flush=True)
