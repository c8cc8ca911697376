"""Helper of tests/test_hip_data.py::test_rccl_path_on_one_gpu (run as a script, one process, world size 1)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import torch                      # noqa: E402
import torch.distributed as dist  # noqa: E402


def main(out_path):
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    from common import build_models
    from ekaid_amd import ops
    from ekaid_amd.config import bench_cfg
    from ekaid_amd.dp import FlatAdam
    from ekaid_amd.engine import TrainStep
    from ekaid_amd.synthetic import make_batch
    cfg = bench_cfg(D=256, seq_length=12)
    host = make_batch(B=4, N=12, D=256, Lq=8, T=5, seq_length=12, V=148, seed=5)
    b = {k: v.cuda() for k, v in host.items()}
    calls = {'n': 0}
    real = dist.all_reduce

    def counted(*a, **k):
        calls['n'] += 1
        return real(*a, **k)
    dist.all_reduce = counted

    order = {}

    def run(force, use_graph, comm):
        cd, sp = build_models(cfg, 147, 148, 'cuda', train=True)
        opt = FlatAdam([cd, sp], lr=1e-3, bucket_mb=8.0)
        opt.force_comm = force
        opt.launch_log = []
        ts = TrainStep(cfg, cd, sp, opt, use_graph=use_graph, comm=comm)
        ops.manual_seed(99)
        try:
            if use_graph:
                for _ in range(3):
                    ts(b, 5)
            else:
                ts._epoch_word(dev)
                s0 = ops.seed_state()
                for _ in range(3):
                    ops.set_seed_state(s0)
                    ts._body(b, 5, None)
        finally:
            ts.close()
        # when was each bucket's all-reduce enqueued, counted in C-ABI calls, vs the end of backward (last step / the capture)
        nb = len(opt.buckets)
        order[(force, use_graph, comm)] = {'launches': opt.launch_log[-nb:] if force else [], 'after_backward': ts.calls_after_backward,
                                           'n_logged': len(opt.launch_log)}
        # flat live parameters and Adam first moments (same layout in every run: the order autograd finishes gradients)
        return opt.pflat.detach().double().clone(), len(opt.buckets), opt.m.detach().double().clone()
    base, _, m = run(False, False, 'captured')
    base2, _, _ = run(False, False, 'captured')         # yardstick: run-to-run float-atomic noise
    eager, nb, _ = run(True, False, 'captured')         # hook-driven async all-reduce per bucket + wait + Adam
    n_eager = calls['n']
    graph, _, _ = run(True, True, 'captured')           # the same inside the captured graph
    graph2, _, _ = run(True, True, 'eager')             # graph for forward/backward, collectives + Adam after the replay
    rms = lambda x: float(x.pow(2).mean().sqrt())       # noqa: E731
    travel = 3 * 1e-3
    sig = torch.ones(1)
    json.dump({'buckets': nb, 'collectives': n_eager, 'noise': rms(base - base2), 'travel': travel, 'signal_fraction': float(sig.double().mean()),
               'delta_eager': rms(base - eager), 'delta_graph': max(rms(base - graph), rms(base - graph2)),
               'order_eager': order[(True, False, 'captured')], 'order_captured': order[(True, True, 'captured')],
               'order_after_replay': order[(True, True, 'eager')]}, open(out_path, 'w'))
    dist.destroy_process_group()


if __name__ == '__main__':
    main(sys.argv[1])
