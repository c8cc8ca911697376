#!/usr/bin/env python3
"""Benchmark of the EKAID difference-VQA training step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

One "step" = one full training iteration of train_mimic.py:209-269 on a synthetic batch resident in HBM:
adjacency one-hot expansion, ChangeDetector forward (both images), DynamicSpeaker teacher-forced forward
(T = 16 decode steps), loss, backward, gradient all-reduce (N > 1) and the fused Adam update -- train mode,
every dropout active, float32 arithmetic end to end.

Scaling: BASELINE.json's metric is GLOBAL batch 256 at 1/2/4/8 GPUs, so the default is STRONG scaling (``--batch`` =
global batch, 256/N pairs per GPU; configs[2] at N = 1).  ``--scaling weak`` keeps 256 pairs per GPU (configs[3]:
global 2048 at N = 8).  At N > 1 the bucketed gradient all-reduce (RCCL) always overlaps backward: launched from gradient
hooks in eager mode (>= 128 pairs per GPU), captured inside the replayed hipGraph below that (``--hip-graph auto``;
ekaid_amd/engine.py); the JSON line carries ``comm.exposed_comm_ms`` = step time minus the same step without collectives.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  "roofline":     the dominant kernel (the 128x128-tile f32 GEMM on the bf16 matrix pipe), timed with HIP events around
                  every launch on the launch stream during two extra, untimed steps; the whole GEMM family next to it;
  "cpu_baseline": the CPU oracle (oracle/, test infrastructure: a port of the reference algorithm proven equal to the
                  reference on golden vectors) timed on this box's host cores on a bounded sample: full train step
                  (forward, backward, torch.optim.Adam) and eval forward at batch 64;
  "extra":        (N = 1 only) forward-only at batch 64 (configs[1]), greedy decoding, the step fed through the input pipeline
                  (train_b256_streamed), the T = 90 worst-case train step, the real-data loop (configs[4] on one GPU), and
                  per_gpu_ladder: the 128 / 64 / 32-pair steps of the metric's 2 / 4 / 8-GPU points, replayed from a hipGraph with the
                  all-reduce captured and issued through a world-1 RCCL communicator (in a child process).
"""
import argparse
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch                      # noqa: E402
import torch.distributed as dist  # noqa: E402



ARITH_NAME = {0: 'f32-mfma', 1: 'bf16x3-exact-split', 2: 'f16x2 (large products) + bf16x3-exact-split (rest)'}


def algorithmic_peak(mode):
    """f32-algorithmic TFLOP/s the matrix pipe allows under a GEMM arithmetic: the f32 pipe itself, or the bf16 / f16 pipe divided by the
    products executed per algorithmic product -- 6 for the exact split; 3 under EKAID_GEMM_F16X2 (its large products; the small ones still
    execute 6, so this roof is on the high side and the fractions quoted against it on the low side)."""
    return PEAK_F32_MFMA_TFLOPS if mode == 0 else PEAK_BF16_MFMA_TFLOPS / (6.0 if mode == 1 else 3.0)


def flop_per_pair(N=36, D=2048, Lq=20, T=16, graph='all', train=True, A=1024, Q=1024, H=4):
    """Algorithmic work of one QA pair, SURVEY.md section 8(d) (the minimum that reproduces the reference's outputs: last direction only,
    context1/gate1 once, core.embed hoisted): multiply-adds of img 2NDA, question Lq 3(600 Q + Q^2) + Lq Q^2, per encoder and image
    N(A+Q)A + 2NA^2 + NHA^2 + (1+H)N^2 A (+ 64 H N^2 for the implicit relation), gating 6NA^2, tail 2N 3A 128, speaker
    3A 1024 + 14.6 M per decode step.  Training = 3 x forward minus the input gradient of img that nobody needs.
    (N=36, D=2048, T=16, graph=all: 5.23 GFLOP forward, 15.39 GFLOP training.)"""
    img = 2.0 * N * D * A
    question = Lq * 3.0 * (600 * Q + Q * Q) + Lq * Q * Q
    enc = N * (A + Q) * A + 2.0 * N * A * A + N * H * A * A + (1 + H) * N * N * A
    encoders = {'semantic': ['s'], 'spatial': ['p'], 'implicit': ['i'], 'i+s': ['i', 's'], 'all': ['s', 'p', 'i']}[graph]
    graph_mac = 2 * (len(encoders) * enc + (64.0 * H * N * N if 'i' in encoders else 0.0))
    gating, tail = 6.0 * N * A * A, 2.0 * N * 3 * A * 128
    speaker = 3.0 * A * 1024 + T * 14.6e6
    fwd = 2.0 * (img + question + graph_mac + gating + tail + speaker)
    return 3.0 * fwd - 2.0 * img if train else fwd


FLOP_PER_PAIR_TRAIN = flop_per_pair()     # 15.39e9 at the headline shape
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense f32-input MFMA peak
PEAK_BF16_MFMA_TFLOPS = 2516.6    # MI355X_MICROARCH.md: dense bf16 MFMA peak (16x the f32-input rate, no sparsity)
SHAPE = dict(N=36, D=2048, Lq=20, T=16, seq_length=90, V=148)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=256, help='global batch (strong scaling) or pairs per GPU (weak scaling)')
    ap.add_argument('--scaling', choices=['weak', 'strong'], default='strong')
    ap.add_argument('--graph', default='all')
    ap.add_argument('--hip-graph', choices=['auto', 'on', 'off'], default='auto',
                    help='replay the iteration from a captured hipGraph (auto: at N > 1 with fewer than 128 pairs per GPU, where '
                         'the host cannot issue ~900 launches as fast as the GPU retires them; eager hook-driven launches above)')
    ap.add_argument('--graph-comm', choices=['auto', 'eager', 'captured'], default='auto',
                    help='with --hip-graph: bucketed all-reduce captured INSIDE the graph, overlapped with backward (captured; auto '
                         'picks it, and falls back to eager if the capture fails) or issued after the replay (eager: serialised)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extra', action='store_true', help='skip the forward-only B=64 and T=90 lines')
    ap.add_argument('--no-ladder', action='store_true', help='skip extra.per_gpu_ladder (the 128 / 64 / 32-pair steps of the 2 / 4 / 8-GPU points)')
    ap.add_argument('--ladder-only', action='store_true', help='run only extra.per_gpu_ladder next to the timed step (tuning aid)')
    ap.add_argument('--ladder-child', type=float, default=None, metavar='MS_256',
                    help='(internal) run extra.per_gpu_ladder in THIS process and print it: bench.py starts itself this way so that the ladder has '
                         'the process layout of a real rank (RCCL communicator created before any GPU work) and cannot take the headline line down')
    ap.add_argument('--cpu-batch', type=int, default=64)
    ap.add_argument('--global-qc', action='store_true', help='global-batch semantics for quirk Q-C')
    ap.add_argument('--eval', action='store_true', help='dropout off (parity mode); default is train mode')
    ap.add_argument('--backend', default='nccl', help='collective backend (nccl = RCCL; gloo only for single-GPU plumbing tests)')
    ap.add_argument('--gemm-shapes', action='store_true', help='print the per-shape GEMM time table to stderr')
    ap.add_argument('--real-only', action='store_true', help='run only the real-data loop (extra.real_b64) and print its JSON: profiling target')
    ap.add_argument('--no-real', action='store_true', help='skip the real-data loop of the extra block')
    ap.add_argument('--nodes', type=int, default=None, help='tuning aid: regions per image instead of 36 (the line then names a different workload)')
    ap.add_argument('--feat-dim', type=int, default=None, help='tuning aid: ROI feature width instead of 2048')
    ap.add_argument('--decode-steps', type=int, default=None, help='tuning aid: answer length (decode steps) instead of 16')
    ap.add_argument('--seq-mode', choices=['auto', 'persistent', 'stepwise'], default='auto',
                    help='decoder / GRU time loops: persistent kernels (one launch per sequence; they need every CU of the GPU to themselves) '
                         'or the launch-per-product chains (stepwise: the safe choice if persistent launches time out next to resident '
                         'collective kernels); auto = the library default (EKAID_SEQ_MODE, persistent), stepwise when ranks share a GPU')
    ap.add_argument('--spawn-timeout', type=int, default=1500,
                    help='--gpus N > 1 started without a launcher: seconds after which the spawned ranks are stopped')
    args = ap.parse_args()
    for key, val in (('N', args.nodes), ('D', args.feat_dim), ('T', args.decode_steps)):
        if val is not None:
            SHAPE[key] = val
    return args


def train_step(cfg, cd, sp, opt, b, T, graph, marks=None):
    """One eager iteration (kept for the tools and tests that import it)."""
    from ekaid_amd.engine import TrainStep
    ts = TrainStep(cfg, cd, sp, opt, graph)
    ts.marks = marks
    return ts(b, T)[0]


def pmc_traffic():
    """HBM bytes per launch of the dominant GEMM shape from the newest committed rocprofv3 --pmc summary (PMC counters
    cannot be read inside this process; the summary is produced by tools/pmc_run.sh, FETCH_SIZE doubled as the gfx950
    guide prescribes).  None when no summary is in profiles/."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_gemm_split_nn_18432x1024x1024.json')))
    if not cands:
        return None
    d = json.load(open(cands[-1]))['derived']
    return {'shape': 'nn 18432x1024x1024 (the input-gradient shape of the 1024-wide layers: 8 solo launches + 3 two-product launches per step at B=256)',
            'fetch_bytes_per_launch': d['hbm_fetch_bytes_per_launch_corrected'],
            'write_bytes_per_launch': d['hbm_write_bytes_per_launch'],
            'bytes_per_launch': d['hbm_fetch_bytes_per_launch_corrected'] + d['hbm_write_bytes_per_launch'],
            'algorithmic_bytes_per_launch': d['algorithmic_read_bytes_per_launch'] + d['algorithmic_write_bytes_per_launch'],
            'source': os.path.relpath(cands[-1], ROOT)}


def gat_roofline(R, N, A=1024, H=4):
    """The graph-attention core (north_star's named kernel: neighbour gather + edge softmax + weighted aggregation) as the step runs it,
    against the HBM roof (reported separately from the matrix-pipe line).  R graph slices of N regions, timed alone with
    HIP events on the launch stream: forward = the batched affinity product + gat_fused_fwd_kernel, backward = gat_bwd_core_kernel (from 256
    slices on; below: two batched products + masked_softmax_bwd) + gat_dqk.  Algorithmic bytes per slice: forward reads q, k, S (3 N A) and writes the aggregated values (N H A) and
    the probabilities (N H M); backward reads d(values) (N H A), q, k, S and the probabilities and writes dq, dk, dS.  `traffic`: the
    committed rocprofv3 PMC summary of the same kernels (tools/pmc_gat_all.sh: FETCH_SIZE doubled as the gfx950 guide prescribes)."""
    from ekaid_amd import ops
    g = torch.Generator().manual_seed(0)
    dev = 'cuda'
    q, k, S = (torch.randn(R, N, A, generator=g).to(dev).requires_grad_(True) for _ in range(3))
    cond = (torch.rand(R, N, N, generator=g) > 0.4).float().to(dev)
    lb = (torch.randn(R, N, N, generator=g) * 0.3).to(dev).requires_grad_(True)
    w = torch.randn(R, N, H * A, generator=g).to(dev)

    def run():
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        out, P = ops.gat_attention(q, k, S, cond, lb, None, H, 1)
        e1.record()
        torch.autograd.grad(out, [q, k, S, lb], w)
        e2.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3, e1.elapsed_time(e2) * 1e3
    for _ in range(3):
        run()
    res = [run() for _ in range(9)]
    fwd, bwd = sorted(x[0] for x in res)[4], sorted(x[1] for x in res)[4]
    M = N
    b_fwd = R * (3 * N * A * 4 + N * H * A * 4 + N * H * M * 4)
    b_bwd = R * (N * H * A * 4 + 3 * N * A * 4 + N * H * M * 4 + 3 * N * A * 4)
    traffic = None
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_gat_n%d.json' % N)))
    if cands:
        try:
            ks = json.load(open(cands[-1]))['kernels']
            traffic = {name: {k2: v['derived'][k2] for k2 in ('median_duration_us', 'hbm_fetch_bytes_corrected', 'hbm_write_bytes', 'algorithmic_bytes',
                                                             'hbm_traffic_over_algorithmic', 'frac_of_8TBps') if k2 in v.get('derived', {})}
                       for name, v in ks.items() if 'derived' in v}
            traffic['source'] = os.path.relpath(cands[-1], ROOT)
        except (KeyError, ValueError):
            traffic = None
    from ekaid_amd import hip
    c0 = hip.launch_counts()['gat_bwd_core']
    run()
    one_launch = hip.launch_counts()['gat_bwd_core'] > c0
    return {'bound': 'hbm', 'kernel': 'graph-attention core: batched affinity product + ekaid::gat_fused_fwd_kernel (forward); ' +
                                      ('ekaid::gat_bwd_core_kernel (d aggregation + d edge softmax in one launch, d(outT) read once)' if one_launch else
                                       'batched products + masked_softmax_bwd_kernel') + ' + gat_dqk_kernel (backward)',
            'shape': {'slices': R, 'regions': N, 'width': A, 'heads': H},
            'forward': {'us': round(fwd, 1), 'algorithmic_bytes': b_fwd, 'achieved': round(b_fwd / fwd / 1e3, 1), 'peak': 8000.0, 'unit': 'GB/s',
                        'frac': round(b_fwd / fwd / 1e3 / 8000.0, 4)},
            'backward': {'us': round(bwd, 1), 'algorithmic_bytes': b_bwd, 'achieved': round(b_bwd / bwd / 1e3, 1), 'peak': 8000.0, 'unit': 'GB/s',
                         'frac': round(b_bwd / bwd / 1e3 / 8000.0, 4)},
            'traffic': traffic,
            'note': 'timed alone, launches back to back (9 runs, median); in the 256-pair step the forward kernel takes ~112 us per launch (its store '
                    'drain overlaps the next kernel).  The aggregated values are (R, N, heads x width): 302 MB of the 514 MB at R = 512 -- the reference '
                    'keeps un-projected values per head (graph_att_layer.py:164-172)'}


def physical_cores():
    try:
        import psutil
        return psutil.cpu_count(logical=False) or os.cpu_count() or 1
    except Exception:
        return os.cpu_count() or 1


def cpu_baseline(cfg, cd, sp, batch_size, graph, budget_s=25.0):
    """Oracle on the host cores, bounded sample: full train step (forward + backward + torch.optim.Adam) and eval
    forward at ``batch_size``; then log-prob parity of the very same weights / inputs through the HIP path."""
    from ekaid_amd.engine import forward_loss
    from ekaid_amd.formula_init import state_copy as oracle_params
    from ekaid_amd.synthetic import make_batch
    from oracle import ekaid_oracle as O
    cores = physical_cores()
    torch.set_num_threads(cores)
    batch = make_batch(B=batch_size, **SHAPE)
    Pcd, Psp = oracle_params(cd, True), oracle_params(sp, True)
    leaves = [v for P in (Pcd, Psp) for v in P.values() if v.requires_grad]
    adam = torch.optim.Adam(leaves, lr=1e-4, betas=(0.9, 0.999), eps=1e-8)        # utils/utils.py:96-99

    def one():
        adam.zero_grad(set_to_none=True)
        loss, _ = O.train_loss(Pcd, Psp, cfg, batch, 'mode2', graph)
        loss.backward()
        adam.step()
    # parity first (weights not yet touched by the CPU Adam): HIP path in eval mode vs the oracle, same inputs
    with torch.no_grad():
        t0 = time.perf_counter()
        _, res = O.train_loss(Pcd, Psp, cfg, batch, 'mode2', graph)
        fwd_s = [time.perf_counter() - t0]
        was = (cd.training, sp.training)
        cd.eval()
        sp.eval()
        dev = next(cd.parameters()).device
        bb = {k: v.to(dev) for k, v in batch.items()}
        T = SHAPE['T']
        _, _, _, logp = forward_loss(cfg, cd, sp, bb, T, graph)
        from ekaid_amd import ops
        logp_x2, mode_now = None, ops.get_gemm_mode()
        other = {ops.GEMM_BF16_SPLIT: ops.GEMM_F16X2, ops.GEMM_F16X2: ops.GEMM_BF16_SPLIT}.get(mode_now)
        if other is not None:                       # the same forward on the other split arithmetic
            ops.set_gemm_mode(other)
            try:
                _, _, _, logp_x2 = forward_loss(cfg, cd, sp, bb, T, graph)
            finally:
                ops.set_gemm_mode(mode_now)
        cd.train(was[0])
        sp.train(was[1])
        ref = res['logp'][:, :T]

        def deltas(lp):
            got = lp[:, :T].cpu()
            d = (got - ref).abs()
            diff = got.argmax(-1) != ref.argmax(-1)
            top = ref.topk(2, dim=-1).values
            margin = (top[..., 0] - top[..., 1])
            return {'logit_max_abs_delta': float(d.max()),
                    'logit_max_rel_delta_to_range': float(d.max() / ref.abs().max()),
                    'logit_max_elementwise_rel_delta': float((d / ref.abs().clamp_min(1e-6)).max()),
                    'argmax_identical': bool(torch.equal(got.argmax(-1), ref.argmax(-1))),
                    # (these are the weights the timed steps have just trained, not a fixture: two candidate words may tie within rounding)
                    'argmax_positions': int(diff.numel()), 'argmax_differing': int(diff.sum()),
                    'oracle_top2_margin_at_differing_positions_max': float(margin[diff].max()) if bool(diff.any()) else None}
        parity = deltas(logp)
        parity_x2 = deltas(logp_x2) if logp_x2 is not None else None
        t0 = time.perf_counter()
        O.train_loss(Pcd, Psp, cfg, batch, 'mode2', graph)
        fwd_s.append(time.perf_counter() - t0)
    t_start = time.perf_counter()
    one()                                               # warm-up
    warm = time.perf_counter() - t_start
    times = []
    t_all = time.perf_counter()
    while len(times) < 5 or (len(times) < 8 and time.perf_counter() - t_all + warm < budget_s):
        t0 = time.perf_counter()
        one()
        times.append(time.perf_counter() - t0)
    times.sort()
    dt = times[len(times) // 2]                          # median iteration
    iters = len(times)
    return {'value': round(batch_size / dt, 3), 'unit': 'QA-pairs/s', 'cores': cores, 'kind': 'port',
            'value_best': round(batch_size / times[0], 3), 'value_worst': round(batch_size / times[-1], 3), 'timed_iterations': iters,
            'eval_forward_pairs_per_s': round(batch_size / min(fwd_s), 3), 'parity_vs_hip': parity,
            'parity_vs_hip_other_arithmetic': dict(parity_x2, gemm_arithmetic=ARITH_NAME[other]) if parity_x2 is not None else None,
            'sample': 'CPU oracle (PyTorch %s, %d threads = physical cores of this host, %d logical) full train step '
                      '(forward + backward + torch.optim.Adam) at batch %d: %d timed iterations after 1 warm-up, value = the MEDIAN iteration; eval '
                      'forward at batch %d: best of 2; same shapes and weights as the GPU run, dropout off' % (
                          torch.__version__, cores, os.cpu_count() or 0, batch_size, iters, batch_size)}


def real_data_loop(graph, iters=120, time_from=80, answer_mix='short+90'):
    """BASELINE.json configs[4] on one GPU: train_mimic.py itself (HDF5 reader -> RCCDataLoader -> DevicePrefetcher -> one
    hipGraph per (shape, decode steps rounded up to a multiple of 8)) on a generated dataset of the REAL shapes -- 52 regions per image,
    1024-d ROI features, the yaml's batch 64, ragged answers.  Answer lengths of the generator: 'short+90' = uniform 2..40 tokens,
    every 50th pair 90 tokens (the maximum), so most batches decode 90 steps and the rest 40; 'uniform' = uniform 2..90 tokens.  On real
    data the decoder, not the detector, sets the step.  The dataset is large enough that the run stays inside one epoch (160 iterations:
    the real training set has thousands per epoch; an epoch boundary restarts the loader thread and costs one batch assembly)."""
    import shutil
    import tempfile
    import train_mimic
    from ekaid_amd.datasets.generate import make_dataset
    d = tempfile.mkdtemp(prefix='ekaid_real_')
    try:
        t0 = time.perf_counter()
        make_dataset(d, n_images=512, n_pairs=12800, nodes=52, D=1024, answer_mix=answer_mix)
        gen_s = time.perf_counter() - t0
        with contextlib.redirect_stdout(sys.stderr):        # stdout carries the ONE JSON line of the contract, nothing else
            out = train_mimic.main(['--data_dir', d, '--graph', graph, '--hip_graph', '--max_iter', str(iters), '--time_from', str(time_from),
                                    '--snapshot_interval', '1000000', '--no_final_snapshot', '--quiet', '--exp_dir', os.path.join(d, 'exp')])
    finally:
        shutil.rmtree(d, ignore_errors=True)
    tm = out['timing'] or {}
    step_ms = tm.get('ms_per_iteration', 0.0)
    return {'workload': 'BASELINE.json configs[4] on one GPU: train_mimic.py --hip_graph on a generated dataset in the reference layout '
                        '(52 regions x 2 images, 1024-d ROI features, batch 64, answers %s), reader + prefetcher + one hipGraph per (shape, '
                        'decode steps rounded up to a multiple of %d); %d timed iterations after %d, %d iterations per epoch' % (
                            'uniform 2..40 tokens, every 50th pair 90 tokens' if answer_mix == 'short+90' else 'uniform 2..90 tokens',
                            out.get('t_round', 1), tm.get('iterations', 0), time_from, out.get('iters_per_epoch', 0)),
            'qa_pairs_per_s': round(tm.get('qa_pairs_per_s', 0.0), 1), 'ms_per_iteration': round(step_ms, 3),
            'loader_wait_ms_per_iteration': round(tm.get('loader_wait_ms_per_iteration', 0.0), 3),
            'loader_wait_fraction': round(tm.get('loader_wait_ms_per_iteration', 0.0) / step_ms, 4) if step_ms else None,
            'host_waits_for_device_ms_per_iteration': round(tm.get('host_waits_for_device_ms_per_iteration', 0.0), 3),
            'batch_assembly_ms': round(tm.get('batch_assembly_ms', 0.0), 3), 'staging_ms': round(tm.get('staging_ms', 0.0), 3),
            'h2d_ms_per_batch': round(tm.get('h2d_ms_per_batch', 0.0), 3), 'h2d_bytes_per_batch': tm.get('h2d_bytes_per_batch'),
            'exposed_h2d_ms_per_iteration': round(tm.get('exposed_h2d_ms_per_iteration', 0.0), 3),
            'graphs_captured': out['graphs_captured'], 'graph_capture_s': round(out.get('graph_capture_s', 0.0), 3),
            'graph_pool_bytes': out.get('graph_pool_bytes'), 'decode_steps_histogram': tm.get('decode_steps_histogram'),
            'decode_steps_run_histogram': out.get('decode_steps_run_histogram'), 'dataset_generation_s': round(gen_s, 2)}


def comm_identity(dev, rank, world, backend_arg, grad_bytes):
    """What the N > 1 line must prove by itself: the backend torch.distributed actually runs, the rank count as the COLLECTIVE sees it
    (an all-gather, not argv), that the ranks sit on distinct devices (all-gathered PCI bus ids + uuids), the RCCL version, and the
    raw bus bandwidth of an all-reduce of the step's gradient volume timed outside the step (busbw = 2 (W-1)/W x bytes / time: the
    per-link figure to hold against xGMI's ~153 GB/s per link when the collective runs as a ring)."""
    props = torch.cuda.get_device_properties(dev)
    me = {'rank': rank, 'device_index': dev.index, 'name': props.name, 'arch': getattr(props, 'gcnArchName', None),
          'pci': '%04x:%02x:%02x' % (getattr(props, 'pci_domain_id', 0), getattr(props, 'pci_bus_id', 0), getattr(props, 'pci_device_id', 0)),
          'uuid': str(getattr(props, 'uuid', '')), 'pid': os.getpid()}
    everyone = [None] * world
    dist.all_gather_object(everyone, me)
    ones = torch.ones(1, device=dev)
    dist.all_reduce(ones)
    backend = dist.get_backend()
    # the 223 MB of live gradients over RCCL; a few MB when the backend stages through the host (gloo plumbing tests)
    nbytes = grad_bytes if backend == 'nccl' else min(grad_bytes, 8 << 20)
    buf = torch.zeros(nbytes // 4, device=dev)
    iters = 10 if backend == 'nccl' else 2
    for _ in range(3 if backend == 'nccl' else 1):
        dist.all_reduce(buf)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        dist.all_reduce(buf)
    torch.cuda.synchronize()
    dt = torch.tensor([(time.perf_counter() - t0) / iters], device=dev, dtype=torch.float64)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt.item())
    try:
        ver = '.'.join(str(v) for v in torch.cuda.nccl.version())
    except Exception:            # noqa: BLE001
        ver = None
    return {'backend': backend, 'backend_requested': backend_arg, 'ranks': len(everyone), 'ranks_counted_by_allreduce': int(ones.item()),
            'distinct_devices': len({(e['pci'], e['uuid']) for e in everyone}), 'devices': everyone, 'rccl_version': ver,
            'allreduce_raw': {'bytes': nbytes, 'ms': round(dt * 1e3, 3), 'algbw_GBps': round(nbytes / dt / 1e9, 2),
                              'busbw_GBps': round(2.0 * (world - 1) / world * nbytes / dt / 1e9, 2), 'iterations': iters}}


def per_gpu_ladder(cfg, cd, sp, opt, dev, graph, full_batch, T, ms_256, steps=40, warmup=5):
    """The 2 / 4 / 8-GPU operating points of the metric (global batch 256 cut over N GPUs: 128 / 64 / 32 pairs per GPU), each run on
    THIS GPU exactly the way a rank of the N-GPU job runs it: the whole iteration replayed from one hipGraph, the bucketed gradient
    all-reduce captured inside it and issued through RCCL (a world-1 communicator: the collective kernels run, nothing crosses a link).
    Per rung: ms per step, launches per replay (kernel nodes of the captured graph), the path's algorithmic TFLOP/s against the matrix-pipe
    roof, and the speed-up over the 1-GPU step this rung implies for N GPUs BEFORE link time (ms_256 / ms)."""
    from ekaid_amd import ops
    from ekaid_amd.engine import TrainStep
    from ekaid_amd.synthetic import slice_batch
    assert dist.is_initialized(), 'per_gpu_ladder runs in the --ladder-child process (world-1 RCCL communicator)'
    peak_alg = algorithmic_peak(ops.get_gemm_mode())
    fpp = flop_per_pair(N=SHAPE['N'], D=SHAPE['D'], Lq=SHAPE['Lq'], T=T, graph=graph)
    rungs = []
    was_force = opt.force_comm
    opt.force_comm = os.environ.get('EKAID_LADDER_COMM', '1') != '0'          # (tuning aid: 0 = the rungs without collectives)
    order = ((2, 128), (4, 64), (8, 32))
    if os.environ.get('EKAID_LADDER_ORDER') == 'rev':                         # (tuning aid: smallest rung first)
        order = order[::-1]
    try:
        for n_gpus, per_gpu in order:
            hb = slice_batch(full_batch, 0, per_gpu)              # rank 0's contiguous slice of the global batch
            b = {k: v.to(dev) for k, v in hb.items()}
            step = TrainStep(cfg, cd, sp, opt, graph, use_graph=True, comm='captured',
                             count_nodes=os.environ.get('EKAID_LADDER_NODES', '1') != '0')
            try:
                st_load = []
                ms = timed(lambda: step(b, T), steps, warmup, st_load)
                nodes = step.node_counts
            finally:
                step.close()
            tf = fpp * per_gpu / (ms * 1e-3) / 1e12
            rungs.append({'n_gpus': n_gpus, 'per_gpu_batch': per_gpu, 'ms_per_step': round(ms, 3),
                          'launches_per_step': (nodes or {}).get('kernel'), 'graph_nodes': nodes,
                          'roofline': {'bound': 'mfma', 'algorithmic_f32_tflops': round(tf, 3), 'algorithmic_peak_tflops': round(peak_alg, 1),
                                       'frac': round(tf / peak_alg, 4)},
                          'clocks_under_load': st_load[0] if st_load else None,
                          'implied_speedup_before_link_time': round(ms_256 / ms, 3),
                          'implied_qa_pairs_per_s': round(256.0 / ms * 1e3, 1)})
    finally:
        opt.force_comm = was_force
    return {'workload': 'BASELINE.json metric points at 2 / 4 / 8 GPUs (global batch 256 -> 128 / 64 / 32 pairs per GPU) run on ONE GPU as a '
                        'rank runs them: hipGraph replay, bucketed all-reduce captured in the graph and issued through RCCL (world-1 '
                        'communicator), %d timed replays after %d; speed-ups are over the 1-GPU step of this run and exclude link time' % (steps, warmup),
            'ms_per_step_1gpu': round(ms_256, 3), 'flop_per_pair': fpp, 'rungs': rungs}


def streamed_step(step_fn, host_batch, dev, T, steps, compact, ms_resident):
    """The timed step fed by the input pipeline instead of a batch resident in HBM: a loader hands the SAME host batch over and over
    (pageable memory, the reference's tuple layout), ``DevicePrefetcher`` stages it into pinned buffers -- label matrices compacted to
    int8 N x N (``compact``) or left as the reference's float64 100 x 100 -- and copies it on a side stream while the previous step
    runs; the compute stream waits at an event.  ``exposed_h2d_ms`` = time the compute stream stood at that fence per step.
    ref: train_mimic.py:206-227 (synchronous pageable copies at the top of every iteration)."""
    from ekaid_amd.datasets.rcc_dataset_pos_mimic import DevicePrefetcher

    class Repeat:
        def __iter__(self):
            for _ in range(steps + 6):
                yield host_batch
    pf = DevicePrefetcher(Repeat(), dev, compact=compact, depth=2, timing=True, run_ahead=2)
    t0 = n = None
    for i, b in enumerate(pf):
        if i == 6:                                  # warm-up: pinned buffers allocated, queue primed
            torch.cuda.synchronize()
            pf._fold_events(0)
            pf._h2d_ms = pf._stall_ms = 0.0
            pf.h2d_bytes = 0
            t0, n = time.perf_counter(), 0
        step_fn({k: v for k, v in b.items() if not k.startswith('_')}, T)
        if t0 is not None:
            n += 1
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    st = pf.stats()
    return {'ms_per_step': round(ms, 3), 'pairs_per_s': round(host_batch['input_1'].shape[0] / ms * 1e3, 1),
            'vs_resident_inputs': round(ms / ms_resident, 4), 'exposed_h2d_ms_per_step': round(st['stall_ms'] / n, 4),
            'h2d_ms_per_batch': round(st['h2d_ms'] / n, 3), 'h2d_bytes_per_batch': int(st['h2d_bytes'] / n),
            'loader_wait_ms_per_step': round(st['loader_wait_s'] / n * 1e3, 3),
            'layout': 'int8 N x N label matrices (compact device feed)' if compact else 'float64 100 x 100 label matrices (the reference layout)'}


def timed(fn, steps, warmup, state=None):
    """ms per call; ``state`` (a list) receives the device state read while the last calls are still running on the GPU."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    if state is not None:
        state.append(device_state())
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def device_state(index=0):
    """Shader / memory clock (MHz), package power (W) and junction temperature (C) of the GPU as sysfs shows them right now -- read while
    timed work is still queued on the device, so a line says which power-management state it was taken in (the same build and box run the
    replayed small steps at 2.40 or at 2.1 GHz: DESIGN.md section 8).  Fields the box does not expose are left out; never raises."""
    import glob
    out = {}
    try:
        cards = sorted(glob.glob('/sys/class/drm/card*/device/pp_dpm_sclk'))
        if not cards:
            return {'note': 'no sysfs clock files'}
        # the card whose PCI address is that of HIP device `index` (a box may expose more cards in sysfs than GPUs to the process, in
        # another order: the first card is not necessarily ours)
        d = None
        try:
            props = torch.cuda.get_device_properties(index)
            want = '%04x:%02x:%02x.' % (getattr(props, 'pci_domain_id', 0), getattr(props, 'pci_bus_id', 0), getattr(props, 'pci_device_id', 0))
            for c in cards:
                if os.path.basename(os.path.realpath(os.path.dirname(c))).startswith(want):
                    d = os.path.dirname(c)
                    out['pci'] = want + '0'
                    break
        except Exception:                       # noqa: BLE001
            d = None
        if d is None:
            d = os.path.dirname(cards[min(index, len(cards) - 1)])
            out['note'] = 'PCI address of the HIP device not found in sysfs: card picked by index'
        out['cards_in_sysfs'] = len(cards)
        for key, name in (('sclk_mhz', 'pp_dpm_sclk'), ('mclk_mhz', 'pp_dpm_mclk')):
            try:
                cur = [ln for ln in open(os.path.join(d, name)) if '*' in ln]
                if cur:
                    out[key] = int(''.join(ch for ch in cur[0].split(':')[1] if ch.isdigit()))
            except (OSError, ValueError, IndexError):
                pass
        for key, pat, div in (('power_w', 'hwmon/hwmon*/power1_average', 1e6), ('power_w', 'hwmon/hwmon*/power1_input', 1e6),
                              ('freq1_mhz', 'hwmon/hwmon*/freq1_input', 1e6), ('temp_junction_c', 'hwmon/hwmon*/temp2_input', 1e3),
                              ('temp_edge_c', 'hwmon/hwmon*/temp1_input', 1e3)):
            for f in glob.glob(os.path.join(d, pat)):
                try:
                    out.setdefault(key, round(int(open(f).read()) / div, 1))
                except (OSError, ValueError):
                    pass
    except Exception as e:                      # noqa: BLE001  (a diagnostic must not cost the bench line)
        out['error'] = '%s: %s' % (type(e).__name__, str(e)[:80])
    return out


def spawn_ranks(n, timeout_s):
    """``python bench.py --gpus N`` without a launcher: start N fresh child processes of this script, one rank per GPU, and
    relay rank 0's JSON line.  The parent never touches the GPU (no HIP call before or after the spawn); children are separate
    processes, not exec replacements.  A child that fails, or the time limit, ends the whole process group."""
    import signal
    import socket
    import subprocess
    with socket.socket() as s:                  # a free rendezvous port
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    import tempfile
    procs = []
    out0 = tempfile.TemporaryFile()             # rank 0's stdout (the JSON line); a file, so a long line can never block the child
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL, start_new_session=True))

    def kill_all():
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGKILL)        # the child's own session: exactly the processes started here
                except OSError:
                    pass
    def on_signal(signum, _frame):              # a SIGTERM / SIGHUP to the parent (harness timeout) raises no exception by itself: the
        kill_all()                              # ranks sit in their own sessions and would be orphaned holding the GPUs
        os._exit(128 + signum)
    for sig in (signal.SIGTERM, signal.SIGHUP, signal.SIGINT):
        signal.signal(sig, on_signal)
    deadline = time.monotonic() + timeout_s
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
            if rc != 0:
                print('bench.py: rank %d exited with %d; stopping the other ranks' % (
                    next(i for i, p in enumerate(procs) if p.poll() not in (None, 0)), rc), file=sys.stderr)
                break
            if time.monotonic() > deadline:
                print('bench.py: ranks still running after %d s; stopping them' % timeout_s, file=sys.stderr)
                rc = 124
                break
            if pending:
                time.sleep(0.2)
    finally:
        kill_all()
    for p in procs:
        p.wait()
    out0.seek(0)
    out = out0.read().decode()
    out0.close()
    if rc == 0:
        # stdout carries the ONE JSON line of the contract; anything else a rank's libraries wrote there (gloo announces its
        # connections on stdout) goes to stderr
        for ln in out.splitlines():
            (sys.stdout if ln.startswith('{') else sys.stderr).write(ln + '\n')
        sys.stdout.flush()
    return rc


_REAL_STDOUT = None


def claim_stdout():
    """stdout carries the ONE JSON line of the contract and nothing else: file descriptor 1 is pointed at stderr for the whole run (RCCL
    prints its version banner there, gloo its connection notes, the dataset classes their progress) and the line is written to the
    original descriptor at the end (``emit``)."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    line = (json.dumps(obj) + '\n').encode()
    sys.stdout.flush()
    if _REAL_STDOUT is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, line)


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus, args.spawn_timeout))
    claim_stdout()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('bench.py --gpus %d but WORLD_SIZE=%d: start it as `python bench.py --gpus N` (it spawns its own ranks) or '
                         'with torch.distributed.run, one rank per GPU (--nproc-per-node = --gpus)' % (args.gpus, world))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    local_rank = local_rank % max(torch.cuda.device_count(), 1)      # (several ranks may share a GPU in the gloo plumbing test)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if args.ladder_child is not None:
        # a world-1 RCCL communicator, created before any GPU work exactly as a rank of the N-GPU job creates its own
        import socket
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from ekaid_amd import hip, ops
    from ekaid_amd.formula_init import build_models
    from ekaid_amd.config import bench_cfg
    from ekaid_amd.dp import FlatAdam, enable_global_question_attention
    from ekaid_amd.engine import TrainStep, forward_loss
    from ekaid_amd.models.dynamic_speaker_change_pos import DynamicSpeaker
    from ekaid_amd.synthetic import make_batch, slice_batch
    hip.load()
    hip.check(hip.lib().ekaid_init(local_rank), 'ekaid_init')
    if world > max(torch.cuda.device_count(), 1):
        # several ranks on one GPU (the single-GPU plumbing tests): two processes' persistent sequence kernels cannot both have all
        # their workgroups resident, their seams would time out -- the launch-per-product chains have no such requirement
        hip.check(hip.lib().ekaid_set_seq_mode(0), 'set_seq_mode')
    if args.seq_mode != 'auto':
        hip.check(hip.lib().ekaid_set_seq_mode(1 if args.seq_mode == 'persistent' else 0), 'set_seq_mode')
    seq_mode_name = 'persistent' if hip.lib().ekaid_get_seq_mode() == 1 else 'stepwise'
    if args.ladder_child is not None:
        cfg = bench_cfg(D=SHAPE['D'], seq_length=SHAPE['seq_length'], graph=args.graph)
        cd, sp = build_models(cfg, 147, 148, dev, train=not args.eval)
        opt = FlatAdam([cd, sp], lr=1e-4, betas=(0.9, 0.999), eps=1e-8)
        full = make_batch(B=256, seed=1238, **SHAPE)
        T = DynamicSpeaker.count_steps(full['labels'], SHAPE['seq_length'])
        ops.manual_seed(1238)
        out = per_gpu_ladder(cfg, cd, sp, opt, dev, args.graph, full, T, args.ladder_child)
        hip.seq_status_check()
        emit(out)
        dist.destroy_process_group()
        return
    if args.real_only:
        emit({'real_b64': real_data_loop(args.graph)})
        return

    if args.scaling == 'strong':
        if args.batch % world:
            raise SystemExit('--batch %d (global) is not divisible by %d GPUs' % (args.batch, world))
        per_gpu, global_batch = args.batch // world, args.batch
    else:
        per_gpu, global_batch = args.batch, args.batch * world
    # N > 1: every default path overlaps the gradient all-reduce with backward -- hook-driven eager launches at >= 128
    # pairs per GPU (the GPU step is longer than the host's issue time), below that the whole step incl. the RCCL calls
    # is replayed from one hipGraph (the collectives sit on a side branch of the graph, ekaid_amd/engine.py)
    use_graph = args.hip_graph == 'on' or (args.hip_graph == 'auto' and world > 1 and per_gpu < 128)
    graph_comm = 'captured' if args.graph_comm == 'auto' else args.graph_comm
    cfg = bench_cfg(D=SHAPE['D'], seq_length=SHAPE['seq_length'], graph=args.graph)
    cd, sp = build_models(cfg, 147, 148, dev, train=not args.eval)      # closed-form weights, identical on every rank
    if args.global_qc and world > 1:
        enable_global_question_attention(cd)
    opt = FlatAdam([cd, sp], lr=cfg.train.optim.lr if cfg.train.optim.type == 'adam' else 1e-4,
                   betas=(0.9, 0.999), eps=1e-8)
    # strong scaling: ONE global synthetic batch (seed 1238), every rank keeps its contiguous slice
    full = make_batch(B=global_batch, seed=1238, **SHAPE) if args.scaling == 'strong' else None
    host_batch = slice_batch(full, rank * per_gpu, (rank + 1) * per_gpu) if full is not None else \
        make_batch(B=per_gpu, seed=1238, rank=rank, **SHAPE)
    del full
    T = DynamicSpeaker.count_steps(host_batch['labels'], SHAPE['seq_length'])   # host-side: no device sync per step
    torch.cuda.synchronize()
    th2d = time.perf_counter()
    b = {k: v.to(dev) for k, v in host_batch.items()}                  # inputs resident in HBM before the clock starts
    torch.cuda.synchronize()
    h2d_ms = (time.perf_counter() - th2d) * 1e3                        # pageable host -> device copy of one batch (reported, never in `value`)
    h2d_bytes = sum(v.numel() * v.element_size() for v in host_batch.values())
    ops.manual_seed(1238 + rank)
    step = TrainStep(cfg, cd, sp, opt, args.graph, use_graph=use_graph, comm=graph_comm)
    eager = TrainStep(cfg, cd, sp, opt, args.graph)

    comm_fallback = None
    if use_graph and graph_comm == 'captured' and args.graph_comm == 'auto' and world > 1:
        # First call = capture.  The ranks AGREE on its outcome before anyone goes on: if one rank alone fell back to eager
        # collectives while its peers replayed captured ones, the collective sequences would diverge and the run would hang.
        # (A capture only issues its own collectives on success; TrainStep._capture restores weights / moments / step in a
        # ``finally``, so a failed attempt leaves every rank at the state it started from.)
        err = None
        try:
            step._graphs[step._key(b, T, None)] = step._capture(b, T, None)
        except Exception as e:                 # noqa: BLE001
            err = '%s: %s' % (type(e).__name__, str(e)[:200])
        failed = torch.tensor([1 if err else 0], device=dev, dtype=torch.int32)
        dist.all_reduce(failed, op=dist.ReduceOp.MAX)
        if int(failed.item()):
            comm_fallback = err or 'capture failed on another rank'
            step.close()
            graph_comm = 'eager'
            step = TrainStep(cfg, cd, sp, opt, args.graph, use_graph=True, comm='eager')
    for _ in range(args.warmup):
        step(b, T)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    state_idle = device_state(local_rank)
    # the K timed steps are also cut into (up to) four windows by events recorded on the launch stream -- no synchronisation inside the
    # timed region; their times are read afterwards: median / min / max per step say how steady the device was during the run
    nwin = min(4, args.steps)
    win_at = [args.steps * i // nwin for i in range(nwin + 1)]
    win_ev = [torch.cuda.Event(enable_timing=True) for _ in range(nwin + 1)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i in win_at:
            win_ev[win_at.index(i)].record()
        loss, _ = step(b, T)
    win_ev[nwin].record()
    state_load = device_state(local_rank)                       # (the GPU is still working through the queued steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    win_ms = [win_ev[i].elapsed_time(win_ev[i + 1]) / max(1, win_at[i + 1] - win_at[i]) for i in range(nwin)]
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_val = float(loss.item())
    hip.seq_status_check()
    # host-side cost of issuing one step (GPU idle at the start, no synchronisation inside): the lower bound the
    # launch path puts on the step time -- for the mode that was timed and for the eager launch path
    torch.cuda.synchronize()
    th = time.perf_counter()
    step(b, T)
    host_issue_ms = (time.perf_counter() - th) * 1e3
    torch.cuda.synchronize()
    if use_graph:
        step.close()        # drop the captured graphs (their static loss pins autograd nodes to the capture stream)
    th = time.perf_counter()
    eager(b, T)
    host_issue_eager_ms = (time.perf_counter() - th) * 1e3
    torch.cuda.synchronize()

    # ---- roofline: HIP events around every GEMM launch, two extra eager steps
    roof, phases = None, {}
    if rank == 0:
        ops.gemm_profile_begin()
    marks = []
    eager.marks = marks if rank == 0 else None
    for _ in range(2):          # every rank runs them (they contain the gradient all-reduce); only rank 0 records events
        eager(b, T)
    eager.marks = None
    torch.cuda.synchronize()
    if rank == 0:
        rec = ops.gemm_profile_end()
        ms = sum(r[0] for r in rec)
        fl = sum(r[1] for r in rec)
        # the dominant kernel: every launch the planner ran on the 128x128-tile instantiations (ekaid_gemm_plan_query: tile 2) -- the
        # set rocprofv3 lists as gemm_split_kernel<., ., 2, 2, 32, 1, ., 3, .> plus gemm_split_pair_kernel<., ., true, 3, ., .> (the same
        # tile body, two products per grid: ekaid_gemm_pair_f32)
        big = [r for r in rec if r[2][6] == 2]
        big_ms, big_fl = sum(r[0] for r in big), sum(r[1] for r in big)
        # executed matrix-pipe flops: every launch priced with the products it runs per algorithmic product (6: exact 3-way bf16 split,
        # 3: two-term f16 split, 1: f32 pipe) -- under EKAID_GEMM_F16X2 the large products run the two-term split, the rest the exact one
        big_ex, all_ex = sum(r[1] * r[2][7] for r in big), sum(r[1] * r[2][7] for r in rec)
        n_x2 = sum(1 for r in big if r[2][7] == 3)
        if args.gemm_shapes:
            agg = {}
            for t_ms, f, shape in rec:
                a = agg.setdefault(shape, [0.0, 0.0, 0])
                a[0] += t_ms
                a[1] += f
                a[2] += 1
            print('%-48s %6s %9s %8s' % ('(tA,tB,M,N,K,batch,tile,executed)', 'calls', 'ms/step', 'TFLOP/s'), file=sys.stderr)
            for shape, (t_ms, f, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
                print('%-48s %6d %9.3f %8.2f' % (shape, n // 2, t_ms / 2, f / (t_ms * 1e-3) / 1e12), file=sys.stderr)
        for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
            if n1 != 'start':
                phases[n1] = round(phases.get(n1, 0.0) + e0.elapsed_time(e1) / 2, 3)
        mode = ops.get_gemm_mode()
        split = mode != ops.GEMM_F32_MFMA
        # the split arithmetics run on the bf16 / f16 matrix pipe (same rate): `achieved` counts the EXECUTED matrix flops (6 or 3 per
        # algorithmic flop, launch by launch); the f32-MFMA arithmetic is priced on the f32 pipe
        peak = PEAK_BF16_MFMA_TFLOPS if split else PEAK_F32_MFMA_TFLOPS
        alg_big = big_fl / (big_ms * 1e-3) / 1e12 if big else 0.0
        alg_all = fl / (ms * 1e-3) / 1e12
        ex_big = big_ex / (big_ms * 1e-3) / 1e12 if big else 0.0
        ex_all = all_ex / (ms * 1e-3) / 1e12
        mult_big = big_ex / big_fl if big_fl else 1.0
        mult_all = all_ex / fl if fl else 1.0
        fpp = flop_per_pair(N=SHAPE['N'], D=SHAPE['D'], Lq=SHAPE['Lq'], T=T, graph=args.graph)
        path_alg = fpp * per_gpu * args.steps / elapsed / 1e12
        kernel_name = {ops.GEMM_F16X2: 'ekaid::gemm_split_kernel<.,.,2,2,32,1,.,2> + its two-products-per-grid form gemm_split_pair_kernel<.,.,true,2,.,.> '
                                       '(128x128 tiles; f32 operands scaled per row / column by powers of two and split into TWO f16 terms, 3 x '
                                       'v_mfma_f32_32x32x16_f16 per product block, f32 accumulate) -- %d of the %d launches of the 128x128-tile family per '
                                       'step; the rest (batched or bound-less products) on the exact 3-way bf16 split <.,.,2,2,32,1,.,3>' % (n_x2 // 2, len(big) // 2),
                       ops.GEMM_BF16_SPLIT: 'ekaid::gemm_split_kernel<.,.,2,2,32,1,.,3> + its two-products-per-grid form gemm_split_pair_kernel<.,.,true,3,.,.> '
                                            '(128x128 tiles; f32 operands split exactly into 3 bf16, 6 x v_mfma_f32_32x32x16_bf16 per product block, f32 accumulate)',
                       ops.GEMM_F32_MFMA: 'ekaid::gemm_mfma_kernel<.,.,2,2,32> (v_mfma_f32_32x32x2_f32)'}[mode]
        roof = {'bound': 'mfma', 'kernel': kernel_name,
                'achieved': round(ex_big, 3), 'peak': peak, 'unit': 'TFLOP/s',
                'frac': round(ex_big / peak, 4), 'traffic': pmc_traffic() if split else None,
                'executed_flop_per_algorithmic_flop': round(mult_big, 3),
                'algorithmic_f32_tflops': round(alg_big, 3), 'algorithmic_peak_tflops': round(peak / mult_big, 1),
                'launches_per_step': len(big) // 2, 'launches_on_the_two_term_split': n_x2 // 2,
                'avg_launch_us': round(big_ms / max(len(big), 1) * 1e3, 2),
                'kernel_ms_per_step': round(big_ms / 2, 3), 'kernel_flop_per_step': big_fl / 2,
                'gemm_family': {'launches_per_step': len(rec) // 2, 'ms_per_step': round(ms / 2, 3),
                                'flop_per_step': fl / 2, 'algorithmic_f32_tflops': round(alg_all, 3),
                                'executed_flop_per_algorithmic_flop': round(mult_all, 3),
                                'frac_of_peak': round(ex_all / peak, 4)},
                'whole_step': {'algorithmic_f32_tflops_per_gpu': round(path_alg, 3), 'frac_of_peak': round(path_alg * mult_all / peak, 4),
                               'flop_per_pair': fpp, 'formula': 'bench.flop_per_pair(N=%d, D=%d, Lq=%d, T=%d, graph=%r): SURVEY.md 8(d)' % (
                                   SHAPE['N'], SHAPE['D'], SHAPE['Lq'], T, args.graph)}}

    # ---- the metric's 2 / 4 / 8-GPU operating points on this one GPU (N = 1 only)
    ladder = None
    if (world == 1 and rank == 0 and args.scaling == 'strong' and global_batch == 256 and not args.no_ladder
            and (not args.no_extra or args.ladder_only) and not use_graph):
        # in a child process of its own (this one idles meanwhile): a rank's process layout, and a failure there costs the ladder only
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), '--ladder-child', '%.4f' % (elapsed / args.steps * 1e3), '--graph', args.graph,
               '--seq-mode', args.seq_mode]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0'))
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
            ladder = json.loads(lines[-1]) if r.returncode == 0 and lines else {'error': 'ladder child exited with %d' % r.returncode,
                                                                                'stderr_tail': r.stderr[-600:]}
        except subprocess.TimeoutExpired:
            ladder = {'error': 'ladder child timed out'}
    # ---- extra lines (N = 1): configs[1] forward-only at batch 64, and the T = 90 worst case
    extra = None
    if world == 1 and not args.no_extra and not args.ladder_only and rank == 0:
        extra = {}
        hb = make_batch(B=64, seed=77, **SHAPE)
        b64 = {k: v.to(dev) for k, v in hb.items()}
        cd.eval(); sp.eval()                                         # noqa: E702
        with torch.no_grad():
            ms64 = timed(lambda: forward_loss(cfg, cd, sp, b64, T, args.graph), 20, 3)
        cd.train(not args.eval); sp.train(not args.eval)             # noqa: E702
        extra['forward_only_b64'] = {'ms': round(ms64, 3), 'pairs_per_s': round(64 / ms64 * 1e3, 1),
                                     'workload': 'BASELINE.json configs[1]: eval forward, batch 64, T=16'}
        # setting 'mode0' (modules.py:17-77, 267-275): the SSRE block -- Q / K / V projections of [v | q], 4-head attention with the heads
        # as a strided batch dimension, LayerNorm -- in place of the graph encoders: north_star's "multi-head QKV attention" and "fused
        # LayerNorm" live only here; same shapes, eval forward and a train step at batch 64
        try:
            cfg0 = bench_cfg(D=SHAPE['D'], seq_length=SHAPE['seq_length'], graph=args.graph, setting='mode0')
            cdm, spm = build_models(cfg0, 147, 148, dev, train=False)
            with torch.no_grad():
                ms0 = timed(lambda: forward_loss(cfg0, cdm, spm, b64, T, args.graph, 'mode0'), 20, 3)
            cdm.train(); spm.train()                                     # noqa: E702

            def mode0_step():
                for p_ in list(cdm.parameters()) + list(spm.parameters()):
                    p_.grad = None
                ops.clear_grad_claims()
                forward_loss(cfg0, cdm, spm, b64, T, args.graph, 'mode0')[0].backward()
            ms0t = timed(mode0_step, 10, 3)
            extra['mode0_b64'] = {'forward_ms': round(ms0, 3), 'forward_pairs_per_s': round(64 / ms0 * 1e3, 1),
                                  'fwd_bwd_ms': round(ms0t, 3), 'fwd_bwd_pairs_per_s': round(64 / ms0t * 1e3, 1),
                                  'workload': "setting='mode0' (SSRE: QKV projections + 4-head attention + LayerNorm instead of the graph encoders), "
                                              'batch 64, T=16: eval forward, and forward + backward in train mode (no optimizer step)'}
            del cdm, spm
        except Exception as e:                  # noqa: BLE001  (a side measurement must not cost the line)
            extra['mode0_b64'] = {'error': '%s: %s' % (type(e).__name__, str(e)[:200])}
        # test_mimic.py's path (configs[1] + answer generation): eval forward, then greedy decode.  On freshly initialised weights (a
        # second model pair: the trained one stops after as many steps as its training length happens to teach it) the answers never
        # emit the end token, so every run decodes all seq_length (90) steps: a fixed worst-case workload
        cd0, sp0 = build_models(cfg, 147, 148, dev, train=False)
        cd0.eval(); sp0.eval()                                       # noqa: E702

        from ekaid_amd.engine import detector_forward

        def decode():
            outs = detector_forward(cfg, cd0, b64, args.graph)
            return sp0._sample(outs[3], outs[4], outs[5], None, cfg, sample_max=1)
        with torch.no_grad():
            msd = timed(decode, 10, 2)
            seq, _ = decode()
            ms_f0 = timed(lambda: detector_forward(cfg, cd0, b64, args.graph), 10, 2)
        nsteps = int((seq != 0).any(0).sum().item())
        del cd0, sp0
        extra['greedy_decode_b64'] = {'ms': round(msd, 3), 'pairs_per_s': round(64 / msd * 1e3, 1), 'decode_steps_run': nsteps,
                                      'ms_per_decode_step': round((msd - ms_f0) / max(nsteps, 1), 4),
                                      'workload': 'test_mimic.py:109-122: eval detector forward + '
                                                  'greedy _sample (ekaid_speaker_greedy_steps: one persistent launch for the whole answer under the default sequence mode, '
                                                 'ending by itself once every row has stopped; 8 steps per call on the step-wise chain), batch 64, '
                                                  'freshly initialised weights: the end token never comes, all %d steps of the maximum '
                                                  'answer length are decoded' % nsteps}
        # the headline step through the input pipeline (pinned staging + side-stream copies), compact and reference label layouts
        ms_res = timed(lambda: eager(b, T), 20, 3)
        extra['train_b%d_streamed' % per_gpu] = {
            'workload': 'the timed %d-pair train step fed by DevicePrefetcher (pinned double buffers, copies on a side stream) instead of '
                        'inputs resident in HBM; ms_resident = the same step re-timed here with resident inputs (20 steps)' % per_gpu,
            'ms_resident': round(ms_res, 3),
            'compact': streamed_step(eager, host_batch, dev, T, 30, True, ms_res),
            'reference_layout': streamed_step(eager, host_batch, dev, T, 30, False, ms_res)}
        # the same train step on the OTHER split arithmetic (exact 3-way bf16 split <-> two-term f16 split for the large products), same box,
        # same minute: what the arithmetic of `value` buys; its log-prob deviation is in cpu_baseline.parity_vs_hip_other_arithmetic
        mode_now = ops.get_gemm_mode()
        other = {ops.GEMM_BF16_SPLIT: ops.GEMM_F16X2, ops.GEMM_F16X2: ops.GEMM_BF16_SPLIT}.get(mode_now)
        if other is not None:
            ops.set_gemm_mode(other)
            try:
                msx2 = timed(lambda: eager(b, T), 20, 3)
            finally:
                ops.set_gemm_mode(mode_now)
            extra['train_b%d_other_arithmetic' % per_gpu] = {
                'gemm_arithmetic': ARITH_NAME[other], 'ms_per_step': round(msx2, 3), 'pairs_per_s': round(per_gpu / msx2 * 1e3, 1),
                'workload': 'the timed train step with ekaid_set_gemm_mode(%d) instead of (%d), 20 eager steps' % (other, mode_now)}
        h90 = make_batch(B=per_gpu, seed=78, **dict(SHAPE, T=90))
        b90 = {k: v.to(dev) for k, v in h90.items()}
        T90 = DynamicSpeaker.count_steps(h90['labels'], SHAPE['seq_length'])
        ms90 = timed(lambda: eager(b90, T90), 8, 2)
        peak_alg = algorithmic_peak(ops.get_gemm_mode())                # algorithmic f32 TFLOP/s the matrix pipe allows
        f90 = flop_per_pair(N=SHAPE['N'], D=SHAPE['D'], Lq=SHAPE['Lq'], T=T90, graph=args.graph)
        tf90 = f90 * per_gpu / (ms90 * 1e-3) / 1e12
        extra['train_T90'] = {'ms_per_step': round(ms90, 3), 'pairs_per_s': round(per_gpu / ms90 * 1e3, 1), 'decode_steps': T90,
                              'roofline': {'bound': 'mfma', 'flop_per_pair': f90, 'algorithmic_f32_tflops': round(tf90, 3),
                                           'algorithmic_peak_tflops': round(peak_alg, 1), 'frac': round(tf90 / peak_alg, 4)},
                              'workload': 'train step, batch %d, answers of the maximum length (90 decode steps)' % per_gpu}
        del b64, b90
        if not args.no_real:
            step.close()
            rd = real_data_loop(args.graph)
            hist = rd.get('decode_steps_run_histogram') or rd.get('decode_steps_histogram') or {}
            n_it = sum(hist.values())
            if n_it and rd.get('ms_per_iteration'):
                # work of the timed iterations from their decode-step histogram (52 regions, 1024-d features, batch 64)
                fl = sum(cnt * flop_per_pair(N=52, D=1024, Lq=SHAPE['Lq'], T=int(tk), graph=args.graph) for tk, cnt in hist.items()) / n_it
                tfr = fl * 64 / (rd['ms_per_iteration'] * 1e-3) / 1e12
                rd['roofline'] = {'bound': 'mfma', 'flop_per_pair_mean': fl, 'algorithmic_f32_tflops': round(tfr, 3),
                                  'algorithmic_peak_tflops': round(peak_alg, 1), 'frac': round(tfr / peak_alg, 4),
                                  'note': 'mean over the histogram of decode steps of the iterations run (not only the timed ones)'}
            extra['real_b64'] = rd
            # robustness on a realistic answer-length mix (uniform 2..90): how many graphs, how long their capture, how much memory
            ru = real_data_loop(args.graph, iters=300, time_from=200, answer_mix='uniform')
            extra['real_b64_uniform_lengths'] = {k: ru[k] for k in ('workload', 'qa_pairs_per_s', 'ms_per_iteration', 'loader_wait_fraction',
                                                                    'graphs_captured', 'graph_capture_s', 'graph_pool_bytes',
                                                                    'decode_steps_histogram', 'decode_steps_run_histogram')}
    if world > 1:
        dist.barrier()

    # ---- exposed communication (N > 1): the same step timed again WITHOUT collectives (local gradients, local Adam); the
    # difference to the timed run is what the all-reduce costs after overlap.  Last thing the run does: weights diverge.
    comm = None
    if world > 1:
        opt.comm_suspended = True
        nocomm = TrainStep(cfg, cd, sp, opt, args.graph, use_graph=use_graph, comm=graph_comm)
        k = min(args.steps, 30)
        for _ in range(3):
            nocomm(b, T)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        tc = time.perf_counter()
        for _ in range(k):
            nocomm(b, T)
        torch.cuda.synchronize()
        dist.barrier()
        t_nc = torch.tensor([(time.perf_counter() - tc) / k], device=dev, dtype=torch.float64)
        dist.all_reduce(t_nc, op=dist.ReduceOp.MAX)
        opt.comm_suspended = False
        nocomm.close()
        ms_nc = float(t_nc.item()) * 1e3
        comm = {**comm_identity(dev, rank, world, args.backend, int(opt.total) * 4),
                'exposed_comm_ms': round(max(0.0, elapsed / args.steps * 1e3 - ms_nc), 3), 'ms_per_step_without_comm': round(ms_nc, 3),
                'allreduce_bytes': int(opt.total) * 4, 'buckets': len(opt.buckets),
                'mode': ('captured in the hipGraph, overlapped' if graph_comm == 'captured' else 'after the replay, serialised')
                if use_graph else 'eager, launched from gradient hooks, overlapped', 'capture_fallback': comm_fallback}

    if rank == 0:
        cfg_idx = None
        if args.graph == 'all' and not args.eval:
            if world == 1 and global_batch == 256:
                cfg_idx = 2
            elif args.scaling == 'weak' and global_batch == 2048 and world == 8:
                cfg_idx = 3
        if cfg_idx is not None:
            workload = 'BASELINE.json configs[%d]' % cfg_idx
        elif args.scaling == 'strong' and global_batch == 256 and args.graph == 'all':
            workload = 'BASELINE.json metric point (global batch 256 cut over %d GPUs, %d pairs per GPU)' % (world, per_gpu)
        else:
            workload = 'off-metric run (global batch %d on %d GPU%s, graph=%s)' % (global_batch, world, 's' if world > 1 else '', args.graph)
        out = {
            'metric': 'QA-pairs/sec (train fwd+bwd+Adam), global batch %d on %d GPU%s' % (global_batch, world, 's' if world > 1 else ''),
            'value': round(global_batch * args.steps / elapsed, 2),
            'unit': 'QA-pairs/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3),
            'ms_per_step_windows': {'windows': nwin, 'median': round(sorted(win_ms)[len(win_ms) // 2], 3), 'min': round(min(win_ms), 3),
                                    'max': round(max(win_ms), 3), 'all': [round(x, 3) for x in win_ms],
                                    'note': 'the %d timed steps cut into %d windows by events on the launch stream (GPU time per step inside '
                                            'the one timed region; value is the wall clock over all of them)' % (args.steps, nwin)},
            'clocks': {'before_timed_loop': state_idle, 'end_of_timed_loop_under_load': state_load,
                       'source': 'sysfs pp_dpm_sclk / pp_dpm_mclk / hwmon of the GPU, rank 0'},
            'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None,
            'dtype': 'f32', 'gemm_arithmetic': {ops.GEMM_BF16_SPLIT: 'bf16x3-exact-split (6 MFMA products, f32 accumulate)',
                                                ops.GEMM_F16X2: 'f16x2 two-term split with per-row / per-column power-of-two scales (3 MFMA products, f32 accumulate) for the '
                                                                'products that carry operand bounds -- the large unbatched ones --, bf16x3-exact-split (6 products) for the rest',
                                                ops.GEMM_F32_MFMA: 'f32-mfma'}[ops.get_gemm_mode()],
            'data': 'synthetic',
            'config': {'workload': '%s: synthetic %d-region x 2-image graphs, %d-d ROI features, 20-token questions, %d decode '
                                   'steps, graph=%s, train step (fwd+bwd+all-reduce+Adam), dropout %s' % (
                                       workload if (SHAPE['N'], SHAPE['D'], SHAPE['T']) == (36, 2048, 16) else 'tuning run (NOT a BASELINE.json configuration)',
                                       SHAPE['N'], SHAPE['D'], SHAPE['T'], args.graph, 'off' if args.eval else 'on'),
                       'global_batch': global_batch, 'per_gpu_batch': per_gpu,
                       'parallelism': 'dp%d' % world, 'question_attention': 'global-batch' if args.global_qc else 'local-batch',
                       'launch_path': ('hipGraph replay, all-reduce %s' % graph_comm) if use_graph else 'eager launches',
                       'seq_mode': seq_mode_name,
                       'params_total': sum(p.numel() for m in (cd, sp) for p in m.parameters()), 'loss': round(loss_val, 5)},
            'roofline': roof,
            'phases_ms': phases if rank == 0 else None,
            'host_issue_ms_per_step': round(host_issue_ms, 3),
            'host_issue_ms_per_step_eager': round(host_issue_eager_ms, 3),
            'h2d': {'ms_per_batch': round(h2d_ms, 3), 'bytes_per_batch': h2d_bytes,
                    'note': 'pageable host memory, first transfer of the process, not overlapped; the reference layout '
                            '(f64 100x100 label matrices); not part of value.  What the step costs when its inputs come through the input '
                            'pipeline instead (pinned staging, side-stream copies) is measured in extra.train_b256_streamed'},
        }
        if comm is not None:
            out['comm'] = comm
        if ladder is not None:
            extra = dict(extra or {}, per_gpu_ladder=ladder)
        if extra is not None:
            out['extra'] = extra
        if world == 1 and not args.no_extra:
            try:
                out['roofline_gat'] = gat_roofline(2 * per_gpu, SHAPE['N'])
            except Exception as e:              # noqa: BLE001  (a side measurement must not cost the line)
                out['roofline_gat'] = {'error': '%s: %s' % (type(e).__name__, str(e)[:200])}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(cfg, cd, sp, args.cpu_batch, args.graph)
        emit(out)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
