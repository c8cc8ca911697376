"""Data-parallel step on the GPU box: two processes share cuda:0 and talk over gloo (the box has one GPU, so the
RCCL transport itself is exercised only by the driver's multi-GPU runs; everything above the transport --
live-set discovery, flat views, bucket hooks, fused Adam with grad_scale, the global-batch Q-C gather -- is the
same code).  With global question-attention semantics and the global loss denominator, 2 ranks x B=2 holding answers
of DIFFERENT lengths (different decode-step counts per rank) must reproduce 1 process x B=4."""
import json
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(rank, world, port, out_path, full_width=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    from common import FULL, SMALL, build_models, small_models
    from ekaid_amd import hip
    from ekaid_amd.config import bench_cfg
    from ekaid_amd.dp import FlatAdam, enable_global_question_attention, loss_denominator
    from ekaid_amd.engine import forward_loss
    from ekaid_amd.models.dynamic_speaker_change_pos import DynamicSpeaker
    from ekaid_amd.synthetic import make_batch, slice_batch
    if full_width:
        # the metric's own case: global batch 256 at the yaml widths, 32 pairs per rank at world 8.  Ranks sharing one GPU run the
        # launch-per-product chains (two processes' persistent kernels cannot both be fully resident), and so does the single process
        hip.load()
        hip.check(hip.lib().ekaid_set_seq_mode(0), 'set_seq_mode')
        cfg = bench_cfg()
        cd, sp = build_models(cfg, 147, 148, 'cuda')
        total = 256
        full = make_batch(B=total, **FULL)
    else:
        cfg, cd, sp = small_models('all', device='cuda')
        total = 4
        full = make_batch(**dict(SMALL, B=4))
        # ragged answers: samples 0/1 (rank 0) keep 3 and 1 answer tokens, samples 2/3 (rank 1) keep 2 and 1
        for b, n in enumerate((4, 2, 3, 2)):
            full['labels'][b, n:] = 0
            full['masks'][b] = 0
            full['masks'][b, :n + 1] = 1
    if world > 1:
        enable_global_question_attention(cd)
    per = total // world
    batch = slice_batch(full, rank * per, (rank + 1) * per)
    dev_batch = {k: v.cuda() for k, v in batch.items()}
    T = DynamicSpeaker.count_steps(batch['labels'], cfg.model.speaker.seq_length)       # differs between the ranks
    denom = loss_denominator(batch['masks'][:, 1:].sum().cuda()) if world > 1 else None
    opt = FlatAdam([cd, sp], lr=1e-3, bucket_mb=0.5)
    losses = []
    for _ in range(2 if full_width else 3):          # (full width: 219 MB of gradients through gloo per step; the second step already runs on all-reduced weights)
        opt.zero_grad()
        loss = forward_loss(cfg, cd, sp, dev_batch, T, 'all', 'mode2', denom)[0]
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    if world > 1:
        t = torch.tensor(losses, device='cuda')
        dist.all_reduce(t)
        losses = (t / world).tolist()
    if rank == 0:
        params = {('cd.' + k): p.detach().cpu().double().norm().item() for k, p in cd.named_parameters()}
        params.update({('sp.' + k): p.detach().cpu().double().norm().item() for k, p in sp.named_parameters()})
        json.dump({'losses': losses, 'norms': params, 'buckets': len(opt.buckets), 'live': len(opt.live),
                   'digest': repr((opt.offsets[:4], opt.buckets[:3], opt.total))}, open(out_path, 'w'))
    if world > 1:
        dist.destroy_process_group()


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


@pytest.mark.parametrize('world,full_width', [(2, False), (8, True)])
def test_ranks_match_one_process(world, full_width, tmp_path):
    """2 ranks x 2 pairs (config 1, ragged answers) and 8 RANKS x 32 PAIRS (the metric's global batch of 256 at the yaml widths, the
    8-GPU operating point) against one process holding the whole batch, global question attention (quirk Q-C) and the global loss
    denominator: three (full width: two) optimizer steps, same losses, same weights.  The ranks share the one GPU of the box and talk over gloo, so the
    8-rank code path -- slices, layout digest, 19 buckets reduced in order from the hooks, the logit gather over 8 ranks, the fused
    Adam with 1 / 8 -- has run before the first real 8-GPU job does."""
    ctx = mp.get_context('spawn')
    single = str(tmp_path / 'single.json')
    p = ctx.Process(target=_run, args=(0, 1, _free_port(), single, full_width))
    p.start()                                                 # (alongside the ranks: the start-up of 9 interpreters is most of this test's time)
    multi = str(tmp_path / 'multi.json')
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, world, port, multi, full_width)) for r in range(world)]
    for q in procs:
        q.start()
    for q in [p] + procs:
        q.join(1200)
        assert q.exitcode == 0
    a, b = json.load(open(single)), json.load(open(multi))
    assert b['buckets'] > 1 and a['live'] == b['live'] and a['digest'] == b['digest']
    assert np.allclose(a['losses'], b['losses'], rtol=2e-5), (a['losses'], b['losses'])
    from golden_util import load_golden, negligible_grad_names
    g1 = load_golden('f4_full' if full_width else 'f1_all')
    noisy = set(negligible_grad_names(g1['grad_names'], g1['grad_rows']))
    worst = max((abs(a['norms'][k] - b['norms'][k]) / max(a['norms'][k], 1e-9), k) for k in a['norms'] if k not in noisy)
    assert worst[0] < (1e-4 if full_width else 1e-5), worst


# ---------------------------------------------------------------------------------------------------------------------
# --hip_graph under data parallelism with RAGGED answers: the ranks meet new (shape, decode steps) keys at different
# iterations, so one rank captures (warm-up iterations included) while the other just replays.  The collective sequence must
# stay aligned: capture warm-ups run without collectives (engine.TrainStep._capture).
RAGGED = [  # per iteration: answer tokens kept by the 4 samples of the global batch (rank 0: samples 0/1, rank 1: samples 2/3)
    (4, 2, 3, 2),   # T = 4 | 3
    (3, 2, 4, 2),   # T = 3 | 4   (each rank meets the other's key: capture on both, one iteration late)
    (4, 3, 2, 2),   # T = 4 | 2   (rank 0 replays, rank 1 captures)
    (2, 2, 4, 3),   # T = 2 | 4   (rank 0 captures, rank 1 replays)
    (4, 2, 3, 2),   # all replays
]


def _run_graph(rank, world, port, out_path, use_graph):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    from common import SMALL, small_models
    from ekaid_amd.dp import FlatAdam
    from ekaid_amd.engine import TrainStep
    from ekaid_amd.models.dynamic_speaker_change_pos import DynamicSpeaker
    from ekaid_amd.synthetic import make_batch, slice_batch
    cfg, cd, sp = small_models('all', device='cuda')
    # (question attention at the local batch: the global-batch variant's all-gather sits inside the forward pass and gloo
    # collectives cannot be captured into a hipGraph -- RCCL's can)
    opt = FlatAdam([cd, sp], lr=1e-3, bucket_mb=0.5)
    ts = TrainStep(cfg, cd, sp, opt, use_graph=use_graph, comm='eager')
    per = 4 // world
    losses, keys = [], []
    try:
        for it, keep in enumerate(RAGGED):
            full = make_batch(**dict(SMALL, B=4, seed=100 + it))
            for b, n in enumerate(keep):
                full['labels'][b, n:] = 0
                full['masks'][b] = 0
                full['masks'][b, :n + 1] = 1
            msum = float(full['masks'][:, 1:].sum())
            batch = slice_batch(full, rank * per, (rank + 1) * per)
            dev_batch = {k: v.cuda() for k, v in batch.items()}
            T = DynamicSpeaker.count_steps(batch['labels'], cfg.model.speaker.seq_length)
            keys.append(T)
            denom = torch.tensor([msum / world], dtype=torch.float32, device='cuda') if world > 1 else None
            loss, _ = ts(dev_batch, T, denom)
            losses.append(float(loss.detach()))
    finally:
        n_graphs = len(ts._graphs)
        ts.close()
    if world > 1:
        t = torch.tensor(losses, device='cuda')
        dist.all_reduce(t)
        losses = (t / world).tolist()
    if rank == 0:
        norms = {('cd.' + k): p.detach().cpu().double().norm().item() for k, p in cd.named_parameters()}
        norms.update({('sp.' + k): p.detach().cpu().double().norm().item() for k, p in sp.named_parameters()})
        json.dump({'losses': losses, 'norms': norms, 'keys': keys, 'graphs': n_graphs, 'steps': opt.step_count}, open(out_path, 'w'))
    if world > 1:
        dist.destroy_process_group()


def test_two_ranks_replaying_hip_graphs_with_ragged_answers_match_two_eager_ranks(tmp_path):
    ctx = mp.get_context('spawn')
    outs = {}
    for use_graph, port in ((False, 29741), (True, 29743)):
        path = str(tmp_path / ('graph.json' if use_graph else 'eager.json'))
        procs = [ctx.Process(target=_run_graph, args=(r, 2, port, path, use_graph)) for r in range(2)]
        for q in procs:
            q.start()
        for q in procs:
            q.join(600)                     # a desynchronised collective sequence shows up here as a hang / non-zero exit
            assert q.exitcode == 0
        outs[use_graph] = json.load(open(path))
    a, b = outs[False], outs[True]
    assert len(set(b['keys'])) == 3 and b['graphs'] == 3 and a['graphs'] == 0, (b['keys'], b['graphs'])
    assert b['steps'] == a['steps'] == len(RAGGED)
    assert np.allclose(a['losses'], b['losses'], rtol=1e-6), (a['losses'], b['losses'])
    worst = max(abs(a['norms'][k] - b['norms'][k]) / max(a['norms'][k], 1e-9) for k in a['norms'])
    assert worst < 1e-6, worst


@pytest.mark.parametrize('ranks,extra', [(2, []), (8, ['--hip-graph', 'on', '--graph-comm', 'eager'])])        # (a 2-rank replay case at 8 pairs per
                                         # rank ran here until round 6: 35-130 s of 219 MB gloo all-reduces for what the 8-rank case and
                                         # test_two_ranks_replaying_hip_graphs_with_ragged_answers_match_two_eager_ranks already cover)
def test_bench_runs_on_two_and_on_eight_ranks(ranks, extra, tmp_path):
    """bench.py the way the driver launches it at N > 1 (torch.distributed.run, one rank per GPU) -- here two ranks share the one
    GPU of the box and talk over gloo, so the RCCL transport is not exercised but everything around it is: slicing of the
    global batch, hook-driven overlapped all-reduce (128 pairs per GPU: eager launches) or hipGraph replay + collectives, the
    max-over-ranks clock, the exposed-communication measurement and the one JSON line of rank 0."""
    import subprocess
    port = _free_port()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(ranks), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', str(ranks), '--steps', '2', '--warmup', '1',
           '--backend', 'gloo', '--no-cpu-baseline', '--no-extra'] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=ROOT,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0'))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == ranks and out['scaling'] == 'strong' and out['value'] > 0
    assert out['config']['per_gpu_batch'] * ranks == out['config']['global_batch']
    if ranks == 8:
        assert out['config']['per_gpu_batch'] == 32 and out['config']['global_batch'] == 256      # the metric's 8-GPU operating point
    c = out['comm']
    assert c['allreduce_bytes'] > 200e6 and c['buckets'] > 1 and c['exposed_comm_ms'] >= 0.0 and c['ms_per_step_without_comm'] > 0
    assert ('hipGraph' in out['config']['launch_path']) == bool(extra)
    # the line identifies its own communication set-up (what an 8-GPU run will be judged from): backend as torch.distributed reports
    # it, ranks counted by the collectives themselves, devices all-gathered, raw all-reduce bandwidth outside the step
    assert c['backend'] == 'gloo' and c['backend_requested'] == 'gloo'
    assert c['ranks'] == ranks and c['ranks_counted_by_allreduce'] == ranks and len(c['devices']) == ranks
    assert c['distinct_devices'] == 1, 'the ranks share the one GPU of this box'
    assert {d['rank'] for d in c['devices']} == set(range(ranks)) and all(d['arch'] and d['pci'] for d in c['devices'])
    assert len({d['pid'] for d in c['devices']}) == ranks
    assert c['allreduce_raw']['busbw_GBps'] > 0 and c['allreduce_raw']['bytes'] > 0 and 'rccl_version' in c
    assert out['config']['seq_mode'] == 'stepwise', 'ranks sharing a GPU must run the launch-per-product chains'


def test_bench_starts_its_own_ranks():
    """``python bench.py --gpus 2`` with NO launcher around it (the way the driver runs the N = 1 line): the script spawns its two
    ranks itself -- fresh child processes, the parent never touches the GPU -- and relays rank 0's single JSON line."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR')}
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--steps', '2', '--warmup', '1',
           '--no-cpu-baseline', '--no-extra']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1 and r.stdout.strip() == lines[0], r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['value'] > 0 and out['config']['per_gpu_batch'] == 128
    assert out['comm']['allreduce_bytes'] > 200e6 and out['comm']['exposed_comm_ms'] >= 0.0
