#!/usr/bin/env python3
"""Benchmark of the EKAID difference-VQA training step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

One "step" = one full training iteration of train_mimic.py:209-269 on a synthetic batch resident in HBM:
adjacency one-hot expansion, ChangeDetector forward (both images), DynamicSpeaker teacher-forced forward
(T = 16 decode steps), loss, backward, gradient all-reduce (N > 1) and the fused Adam update -- train mode,
every dropout active, float32 arithmetic end to end.  Weak scaling: 256 (image-pair, question) samples per
GPU (BASELINE.json configs[2] at N=1, configs[3] at N=8).

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  "roofline":     the fp32-MFMA GEMM kernel family (dominant kernel, ~97 % of the algorithmic FLOPs), timed
                  with HIP events around every launch on the launch stream during two extra, untimed steps;
  "cpu_baseline": the CPU oracle (oracle/, test infrastructure: a port of the reference algorithm proven equal
                  to the reference on golden vectors) timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch                      # noqa: E402
import torch.distributed as dist  # noqa: E402

FLOP_PER_PAIR_TRAIN = 15.39e9     # algorithmic fwd+bwd work per QA pair (N=36, D=2048, T=16)
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense f32-input MFMA peak
PEAK_BF16_MFMA_TFLOPS = 2516.6    # MI355X_MICROARCH.md: dense bf16 MFMA peak (16x the f32-input rate, no sparsity)
SHAPE = dict(N=36, D=2048, Lq=20, T=16, seq_length=90, V=148)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=256, help='samples per GPU (weak) or global (strong)')
    ap.add_argument('--scaling', choices=['weak', 'strong'], default='weak')
    ap.add_argument('--graph', default='all')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-batch', type=int, default=8)
    ap.add_argument('--global-qc', action='store_true', help='global-batch semantics for quirk Q-C')
    ap.add_argument('--eval', action='store_true', help='dropout off (parity mode); default is train mode')
    ap.add_argument('--backend', default='nccl', help='collective backend (nccl = RCCL; gloo only for single-GPU plumbing tests)')
    ap.add_argument('--gemm-shapes', action='store_true', help='print the per-shape GEMM time table to stderr')
    return ap.parse_args()


def train_step(cfg, cd, sp, opt, b, T, graph, marks=None):
    from ekaid_amd.utils.mimic_utils import process_matrix
    from ekaid_amd.utils.utils import LanguageModelCriterion, total_loss
    N = b['input_1'].shape[1]
    dev = b['input_1'].device

    def mark(name):
        if marks is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append((name, e))
    mark('start')
    opt.zero_grad()
    d_adj = process_matrix(b['d_adj'], cfg, N, dev, type='spatial')
    q_adj = process_matrix(b['q_adj'], cfg, N, dev, type='spatial')
    d_sem = process_matrix(b['d_sem'], cfg, N, dev, type='semantic')
    q_sem = process_matrix(b['q_sem'], cfg, N, dev, type='semantic')
    mark('adjacency')
    outs = cd(b['input_1'], b['input_2'], d_adj, q_adj, d_sem, q_sem, b['d_bb'], b['q_bb'], b['question'],
              setting='mode2', graph=graph)
    mark('detector_fwd')
    logp, _ = sp._forward(outs[3], outs[4], outs[5], b['labels'], num_steps=T)
    sp_loss = LanguageModelCriterion()(logp, b['labels'][:, 1:], b['masks'][:, 1:])
    loss = total_loss(sp_loss, outs[1], outs[2])
    mark('speaker_fwd_loss')
    loss.backward()
    mark('backward')
    opt.step()
    mark('allreduce_wait_adam')
    return loss


def pmc_traffic(split):
    """HBM bytes per launch of the dominant GEMM shape from the committed rocprofv3 --pmc summary (PMC counters cannot
    be read inside this process; the summary is produced by tools/pmc_run.sh, FETCH_SIZE doubled as the gfx950 guide
    prescribes).  None when the summary for the active arithmetic is not in profiles/."""
    path = os.path.join(ROOT, 'profiles', 'r01_pmc_gemm_split_nn_18432x1024x1024.json')
    if not split or not os.path.exists(path):
        return None
    d = json.load(open(path))['derived']
    return {'shape': 'nn 18432x1024x1024 (28 launches per step)',
            'fetch_bytes_per_launch': d['hbm_fetch_bytes_per_launch_corrected'],
            'write_bytes_per_launch': d['hbm_write_bytes_per_launch'],
            'algorithmic_bytes_per_launch': d['algorithmic_read_bytes_per_launch'] + d['algorithmic_write_bytes_per_launch'],
            'source': 'profiles/r01_pmc_gemm_split_nn_18432x1024x1024.json'}


def cpu_baseline(cfg, cd, sp, batch_size, graph):
    """Oracle train fwd+bwd on the host cores, bounded sample (1 warm-up + timed iterations, <= ~25 s)."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from common import oracle_params
    from ekaid_amd.synthetic import make_batch
    from oracle import ekaid_oracle as O
    threads = min(os.cpu_count() or 1, 32)      # more threads than this only adds contention at batch 8
    torch.set_num_threads(threads)
    batch = make_batch(B=batch_size, **SHAPE)
    Pcd, Psp = oracle_params(cd, True), oracle_params(sp, True)

    def one():
        for P in (Pcd, Psp):
            for v in P.values():
                v.grad = None
        loss, _ = O.train_loss(Pcd, Psp, cfg, batch, 'mode2', graph)
        loss.backward()
    one()
    t0 = time.perf_counter()
    iters = 0
    while iters < 1 or (time.perf_counter() - t0 < 10.0 and iters < 20):
        one()
        iters += 1
    dt = (time.perf_counter() - t0) / iters
    # parity of the very same weights/inputs through the HIP path (eval mode = dropout off): logit max|delta|
    with torch.no_grad():
        _, res = O.train_loss(Pcd, Psp, cfg, batch, 'mode2', graph)
        was = (cd.training, sp.training)
        cd.eval()
        sp.eval()
        dev = next(cd.parameters()).device
        bb = {k: v.to(dev) for k, v in batch.items()}
        from ekaid_amd.utils.mimic_utils import process_matrix
        N = bb['input_1'].shape[1]
        adj = [process_matrix(bb[k], cfg, N, dev, type=t) for k, t in
               (('d_adj', 'spatial'), ('q_adj', 'spatial'), ('d_sem', 'semantic'), ('q_sem', 'semantic'))]
        outs = cd(bb['input_1'], bb['input_2'], adj[0], adj[1], adj[2], adj[3], bb['d_bb'], bb['q_bb'], bb['question'],
                  setting='mode2', graph=graph)
        logp, _ = sp._forward(outs[3], outs[4], outs[5], bb['labels'])
        cd.train(was[0])
        sp.train(was[1])
        T = SHAPE['T']
        ref = res['logp'][:, :T]
        got = logp[:, :T].cpu()
        parity = {'logit_max_abs_delta': float((got - ref).abs().max()),
                  'logit_max_rel_delta': float((got - ref).abs().max() / ref.abs().max()),
                  'argmax_identical': bool(torch.equal(got.argmax(-1), ref.argmax(-1)))}
    return {'value': round(batch_size / dt, 3), 'parity_vs_hip': parity, 'unit': 'QA-pairs/s', 'cores': threads, 'kind': 'port',
            'sample': 'CPU oracle (PyTorch %s, %d threads) train fwd+bwd, batch %d, %d timed iterations after 1 warm-up; '
                      'same shapes/weights as the GPU run, dropout off, no optimizer step' % (
                          torch.__version__, threads, batch_size, iters)}


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)' % args.gpus)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    local_rank = local_rank % max(torch.cuda.device_count(), 1)      # (several ranks may share a GPU in the gloo plumbing test)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from common import build_models
    from ekaid_amd import hip, ops
    from ekaid_amd.config import bench_cfg
    from ekaid_amd.dp import FlatAdam, enable_global_question_attention
    from ekaid_amd.models.dynamic_speaker_change_pos import DynamicSpeaker
    from ekaid_amd.synthetic import make_batch
    hip.load()

    per_gpu = args.batch if args.scaling == 'weak' else args.batch // world
    cfg = bench_cfg(D=SHAPE['D'], seq_length=SHAPE['seq_length'], graph=args.graph)
    cd, sp = build_models(cfg, 147, 148, dev, train=not args.eval)      # closed-form weights, identical on every rank
    if args.global_qc and world > 1:
        enable_global_question_attention(cd)
    opt = FlatAdam([cd, sp], lr=cfg.train.optim.lr if cfg.train.optim.type == 'adam' else 1e-4,
                   betas=(0.9, 0.999), eps=1e-8)
    host_batch = make_batch(B=per_gpu, seed=1238, rank=rank, **SHAPE)
    T = DynamicSpeaker.count_steps(host_batch['labels'], SHAPE['seq_length'])   # host-side: no device sync per step
    torch.cuda.synchronize()
    th2d = time.perf_counter()
    b = {k: v.to(dev) for k, v in host_batch.items()}                  # inputs resident in HBM before the clock starts
    torch.cuda.synchronize()
    h2d_ms = (time.perf_counter() - th2d) * 1e3                        # pageable host -> device copy of one batch (reported, never in `value`)
    h2d_bytes = sum(v.numel() * v.element_size() for v in host_batch.values())
    ops.manual_seed(1238 + rank)

    for _ in range(args.warmup):
        train_step(cfg, cd, sp, opt, b, T, args.graph)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = train_step(cfg, cd, sp, opt, b, T, args.graph)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_val = float(loss.item())
    # host-side cost of issuing one step (GPU idle at the start, no synchronisation inside): the lower bound the
    # launch path puts on the step time
    torch.cuda.synchronize()
    th = time.perf_counter()
    train_step(cfg, cd, sp, opt, b, T, args.graph)
    host_issue_ms = (time.perf_counter() - th) * 1e3
    torch.cuda.synchronize()

    # ---- roofline of the dominant kernel family: HIP events around every GEMM launch, two extra steps
    roof = None
    if rank == 0:
        ops.gemm_profile_begin()
    marks = []
    for _ in range(2):          # every rank runs them (they contain the gradient all-reduce); only rank 0 records events
        train_step(cfg, cd, sp, opt, b, T, args.graph, marks if rank == 0 else None)
    torch.cuda.synchronize()
    if rank == 0:
        rec = ops.gemm_profile_end()
        ms = sum(r[0] for r in rec)
        fl = sum(r[1] for r in rec)
        big = [r for r in rec if r[1] >= 1e9]
        if args.gemm_shapes:
            agg = {}
            for t_ms, f, shape in rec:
                a = agg.setdefault(shape, [0.0, 0.0, 0])
                a[0] += t_ms
                a[1] += f
                a[2] += 1
            print('%-44s %6s %9s %8s' % ('(tA,tB,M,N,K,batch)', 'calls', 'ms/step', 'TFLOP/s'), file=sys.stderr)
            for shape, (t_ms, f, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
                print('%-44s %6d %9.3f %8.2f' % (shape, n // 2, t_ms / 2, f / (t_ms * 1e-3) / 1e12), file=sys.stderr)
        phases = {}
        for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
            if n1 != 'start':
                phases[n1] = round(phases.get(n1, 0.0) + e0.elapsed_time(e1) / 2, 3)
        alg = fl / (ms * 1e-3) / 1e12             # algorithmic f32 TFLOP/s of the GEMM family (2*M*N*K per call)
        split = ops.get_gemm_mode() == ops.GEMM_BF16_SPLIT
        # bf16-split arithmetic executes SIX bf16 MFMA products per algorithmic f32 product: the roofline is the bf16
        # matrix pipe and `achieved` counts the executed bf16 flops; the f32-MFMA arithmetic is priced on the f32 pipe
        executed, peak = (6.0 * alg, PEAK_BF16_MFMA_TFLOPS) if split else (alg, PEAK_F32_MFMA_TFLOPS)
        path_alg = FLOP_PER_PAIR_TRAIN * per_gpu * args.steps / elapsed / 1e12
        roof = {'bound': 'mfma',
                'kernel': ('ekaid::gemm_split_kernel (f32 operands split exactly into 3 bf16, 6 x v_mfma_f32_32x32x16_bf16 '
                           'per product block, f32 accumulate)') if split else 'ekaid::gemm_mfma_kernel (v_mfma_f32_32x32x2_f32)',
                'achieved': round(executed, 3), 'peak': peak, 'unit': 'TFLOP/s',
                'frac': round(executed / peak, 4), 'traffic': pmc_traffic(split),
                'executed_flop_per_algorithmic_flop': 6 if split else 1,
                'algorithmic_f32_tflops': round(alg, 3), 'algorithmic_peak_tflops': round(peak / (6 if split else 1), 1),
                'f32_mfma_peak_tflops': PEAK_F32_MFMA_TFLOPS,
                'launches_per_step': len(rec) // 2, 'gemm_ms_per_step': round(ms / 2, 3),
                'gemm_flop_per_step': fl / 2,
                'large_gemm_tflops': round(sum(r[1] for r in big) / (sum(r[0] for r in big) * 1e-3) / 1e12, 3) if big else None,
                'path_algorithmic_tflops': round(path_alg * world, 3),
                'path_frac_of_peak': round(path_alg * (6 if split else 1) / peak, 4)}
    if world > 1:
        dist.barrier()

    if rank == 0:
        out = {
            'metric': 'QA-pairs/sec (train fwd+bwd+Adam), batch %d per GPU' % per_gpu,
            'value': round(per_gpu * world * args.steps / elapsed, 2),
            'unit': 'QA-pairs/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None,
            'dtype': 'f32', 'gemm_arithmetic': 'bf16x3-exact-split (6 MFMA products, f32 accumulate)' if ops.get_gemm_mode() == ops.GEMM_BF16_SPLIT else 'f32-mfma',
            'data': 'synthetic',
            'config': {'workload': 'BASELINE.json configs[%d]: synthetic 36-region x 2-image graphs, 2048-d ROI features, '
                                   '20-token questions, 16 decode steps, graph=%s, train step (fwd+bwd+all-reduce+Adam), '
                                   'dropout %s' % (2 if world == 1 else 3, args.graph, 'off' if args.eval else 'on'),
                       'global_batch': per_gpu * world, 'per_gpu_batch': per_gpu,
                       'parallelism': 'dp%d' % world, 'question_attention': 'global-batch' if args.global_qc else 'local-batch',
                       'params_total': 87290418, 'loss': round(loss_val, 5)},
            'roofline': roof,
            'phases_ms': phases if rank == 0 else None,
            'host_issue_ms_per_step': round(host_issue_ms, 3),
            'h2d': {'ms_per_batch': round(h2d_ms, 3), 'bytes_per_batch': h2d_bytes,
                    'note': 'pageable host memory, first transfer of the process, not overlapped; the reference layout '
                            '(f64 100x100 label matrices); not part of value',
                    'value_if_serialised': round(per_gpu * world / (elapsed / args.steps + h2d_ms * 1e-3), 2)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(cfg, cd, sp, args.cpu_batch, args.graph)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
