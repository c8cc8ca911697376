#!/usr/bin/env python3
"""Training entry point -- build-owned counterpart of the reference's ``model/train_mimic.py`` (same command line,
same loop order, same checkpoint dict ``{'change_detector_state', 'speaker_state', 'model_cfg'}``), running the
difference-VQA path on MI355X through libekaid_hip.so.

    python train_mimic.py --graph all                                          # the MIMIC-Diff-VQA HDF5 files of cfg.data.*
    python -m ekaid_amd.datasets.generate --out /tmp/d && python train_mimic.py --data_dir /tmp/d --max_iter 40
    python train_mimic.py --synthetic --max_iter 20                            # no files at all: seeded batches
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train_mimic.py ...

Differences from the reference script (ref: train_mimic.py:1-386), all deliberate:
  * one process per GPU with RCCL gradient all-reduce (the reference pins ``CUDA_VISIBLE_DEVICES=0``, :2); every GLOBAL
    batch (``cfg.data.train.batch_size`` samples, drawn in the reference's order) is cut into contiguous per-rank slices,
    and each rank's language-model loss is divided by (global mask sum / world) so that the averaged gradient is the
    reference's global-batch gradient also for ragged answers (utils/utils.py:213-214);
  * batches come through ``DevicePrefetcher``: int8 adjacency, pinned double buffers, copies on a side stream;
  * ``--hip_graph`` replays the whole iteration from a captured hipGraph (ekaid_amd/engine.py), on several GPUs with the
    bucketed all-reduce captured inside it (overlapped with backward); ragged answers give every rank its own sequence of
    (shape, decode steps) graphs -- the collectives inside them line up because every graph issues the same buckets;
  * no ``.item()`` synchronisation per iteration: losses are read only every ``log_interval`` iterations;
  * the in-loop evaluation greedy-decodes the validation split and scores exact-match accuracy (evaluate_score.py:216-251);
    BLEU/METEOR/ROUGE/CIDEr are delegated to pycocoevalcap when it is importable (ekaid_amd/evaluation.py) and "best" is then
    chosen by BLEU-1 like the reference (:379-383); without the package (this image) the COCO-format json the reference's
    ``evaluation.py`` consumes is still written and "best" falls back to exact-match accuracy;
  * checkpoints additionally carry optimizer moments / step, iteration, epoch position, RNG state at the start of the
    epoch and the dropout seed counter: ``--resume`` continues the run (the reference restarts Adam and the epoch);
  * ``--coef_sem/--coef_spa`` are accepted and, as in the reference (quirk Q-D, :53-54 vs modules.py:117-118),
    do not reach the model.
"""
import argparse
import json
import os
import random
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from ekaid_amd import hip, ops                                            # noqa: E402
from ekaid_amd.config import cfg, default_cfg, merge_cfg_from_file, merge_into, MIMIC_YAML   # noqa: E402
from ekaid_amd.dp import FlatAdam, enable_global_question_attention      # noqa: E402
from ekaid_amd.engine import TrainStep, detector_forward                 # noqa: E402
from ekaid_amd.models.dynamic_speaker_change_pos import DynamicSpeaker   # noqa: E402
from ekaid_amd.models.modules import AddSpatialInfo, ChangeDetector      # noqa: E402
from ekaid_amd.synthetic import make_batch, slice_batch                  # noqa: E402
from ekaid_amd.utils.utils import (AverageMeter, coco_gen_format_save, decode_sequence, load_checkpoint,  # noqa: E402
                                   save_checkpoint, set_mode)


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--cfg', default='')
    p.add_argument('--setting', type=str, default='mode2')
    p.add_argument('--graph', type=str, default='all', choices=['implicit', 'semantic', 'spatial', 'all', 'i+s'])
    p.add_argument('--resume', type=bool, default=False)
    p.add_argument('--checkpoint', type=str, default='')
    p.add_argument('--feature_mode', type=str, default='both', choices=['both', 'location', 'single_ana', 'single_loc'])
    p.add_argument('--eval_target', type=str, default='test', choices=['test', 'val'])
    p.add_argument('--seed', type=int, default=1238)
    p.add_argument('--coef_sem', type=float, default=0.333)
    p.add_argument('--coef_spa', type=float, default=0.333)
    p.add_argument('--lr', type=float, default=0.0001)
    # build-owned extensions
    p.add_argument('--data_dir', default='', help='directory holding the four data files under their reference names')
    p.add_argument('--synthetic', action='store_true', help='seeded synthetic batches, no files')
    p.add_argument('--max_iter', type=int, default=None)
    p.add_argument('--batch_size', type=int, default=None, help='GLOBAL batch size')
    p.add_argument('--snapshot_interval', type=int, default=None)
    p.add_argument('--step_size', type=int, default=None, help='StepLR period in epochs')
    p.add_argument('--iters_per_epoch', type=int, default=50, help='epoch length of the --synthetic stream')
    p.add_argument('--nodes', type=int, default=52, help='regions per image of the synthetic batches')
    p.add_argument('--answer_steps', type=int, default=16)
    p.add_argument('--global_qc', action='store_true', help='global-batch semantics for the question attention (Q-C)')
    p.add_argument('--hip_graph', action='store_true', help='replay each iteration from a captured hipGraph')
    p.add_argument('--graph_comm', choices=['auto', 'captured', 'eager'], default='auto',
                   help='with --hip_graph on several GPUs: gradient all-reduce captured inside the graph and overlapped with '
                        'backward (captured; auto) or issued after each replay (eager)')
    p.add_argument('--eval_batches', type=int, default=None, help='cap on validation batches per in-loop evaluation')
    p.add_argument('--exp_dir', default=None)
    p.add_argument('--seq_mode', choices=['auto', 'persistent', 'stepwise'], default='auto',
                   help='decoder / GRU time loops: one persistent launch per sequence (needs every CU of the GPU) or the launch-per-product '
                        'chains; auto = library default (EKAID_SEQ_MODE), stepwise when ranks share a GPU')
    p.add_argument('--quiet', action='store_true')
    p.add_argument('--no_final_snapshot', action='store_true', help='benchmark aid: no checkpoint / evaluation at max_iter')
    p.add_argument('--t_round', type=int, default=None,
                   help='decode steps rounded up to a multiple of this (extra steps are fully masked); default 8 with --hip_graph (fewer '
                        'captured graphs on ragged answers), 1 otherwise')
    p.add_argument('--time_from', type=int, default=None,
                   help='benchmark aid: time the loop from this iteration on (QA-pairs/s, loader wait, exposed host->device copy)')
    return p.parse_args(argv)


def step_lr(base_lr, gamma, step_size, epoch):
    """``StepLR`` as the reference drives it: ``lr_scheduler.step()`` at the START of every epoch, epoch 0 included
    (train_mimic.py:174-177,190), so the rate in force during epoch e is ``base * gamma ** ((e + 1) // step_size)``."""
    return base_lr * (gamma ** ((epoch + 1) // step_size))


def evaluate(cfg, change_detector, speaker, loader, device, idx_to_word, graph, max_batches=None, hip_graph=False):
    """Greedy decode of a split (train_mimic.py:292-362, test_mimic.py:97-141: detector forward, then ``_sample``): {pair index:
    sentence}, exact-match accuracy.  ``hip_graph``: the detector forward replayed from a hipGraph per batch shape."""
    from ekaid_amd.datasets.rcc_dataset_pos_mimic import DevicePrefetcher
    from ekaid_amd.engine import InferStep
    predictions, references, correct, total = {}, {}, 0, 0
    set_mode('eval', [change_detector, speaker])
    infer = InferStep(cfg, change_detector, graph, cfg.train.setting, use_graph=hip_graph)
    # Evaluation must leave torch's default generator where it found it: creating a loader iterator draws a base seed from it
    # (torch >= 2.x), the training sampler shuffles every epoch from that generator, and only rank 0 evaluates -- without
    # this the ranks' "global" permutations drift apart after the first in-loop evaluation, and a resumed run (which never
    # replays the evaluations that preceded its checkpoint) shuffles differently from the uninterrupted one.
    rng = torch.get_rng_state()
    try:
        with torch.no_grad():
            for bi, b in enumerate(DevicePrefetcher(loader, device)):
                if max_batches is not None and bi >= max_batches:
                    break
                outs = infer(b)
                seq, _ = speaker._sample(outs[3], outs[4], outs[5], b['labels'], cfg, sample_max=1)
                sents = decode_sequence(idx_to_word, seq)
                gts = decode_sequence(idx_to_word, b['_host_labels'][:, 1:])
                for j, s in enumerate(sents):
                    predictions[str(int(b['pair_index'][j]))] = s
                    references[str(int(b['pair_index'][j]))] = gts[j]
                    correct += int(s == gts[j])
                    total += 1
    finally:
        torch.set_rng_state(rng)
        set_mode('train', [change_detector, speaker])
    evaluate.last_references = references          # (for the caption metrics; the return value keeps its three fields)
    return predictions, correct / max(total, 1), total


def main(argv=None):
    args = parse(argv)
    cfg.clear()
    cfg.update(default_cfg())              # the module-level cfg is a process global (as in the reference): start clean
    if args.cfg:
        merge_cfg_from_file(args.cfg)
    else:
        merge_into(json.loads(json.dumps(MIMIC_YAML)), cfg)
    cfg.model.coef_sem, cfg.model.coef_spa = args.coef_sem, args.coef_spa      # Q-D: the model never reads these
    cfg.train.optim.lr = args.lr
    cfg.data.feature_mode = args.feature_mode
    cfg.train.graph, cfg.train.setting = args.graph, args.setting
    if args.max_iter is not None:
        cfg.train.max_iter = args.max_iter
    if args.batch_size is not None:
        cfg.data.train.batch_size = args.batch_size
    if args.snapshot_interval is not None:
        cfg.train.snapshot_interval = args.snapshot_interval
    if args.step_size is not None:
        cfg.train.optim.step_size = args.step_size
    if args.exp_dir:
        cfg.exp_dir = args.exp_dir
    if args.data_dir:
        from ekaid_amd.datasets.generate import point_cfg_at
        point_cfg_at(cfg, args.data_dir)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    own_pg = False
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
        own_pg = True
    hip.load()
    hip.check(hip.lib().ekaid_init(local_rank), 'ekaid_init')
    if world > max(torch.cuda.device_count(), 1):      # ranks sharing a GPU: no persistent sequence kernels (they need every CU at once)
        hip.check(hip.lib().ekaid_set_seq_mode(0), 'set_seq_mode')
    if args.seq_mode != 'auto':
        if args.seq_mode == 'persistent' and world > max(torch.cuda.device_count(), 1):
            #  the plan-time occupancy check only sees this process: two ranks' persistent kernels on one GPU cannot both be
            # fully resident, their seams would time out -- refused rather than left to the time-out
            raise SystemExit('--seq_mode persistent with %d ranks on %d GPU(s): ranks that share a GPU must use the step-wise chains' % (
                world, torch.cuda.device_count()))
        hip.check(hip.lib().ekaid_set_seq_mode(1 if args.seq_mode == 'persistent' else 0), 'set_seq_mode')
    random.seed(args.seed)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)          # identical default initialisation and sample order on every rank
    ops.manual_seed(args.seed + rank)     # different dropout masks per rank
    say = (lambda *a: None) if (args.quiet or rank != 0) else print

    exp_name = '%s_%s_%s_%s_%d' % (args.setting, args.feature_mode, args.graph, args.lr, args.seed)
    output_dir = os.path.join(cfg.exp_dir, 'temp', exp_name)
    snapshot_dir = os.path.join(output_dir, 'snapshots')
    sent_dir = os.path.join(output_dir, 'eval_sents')
    if rank == 0:
        os.makedirs(snapshot_dir, exist_ok=True)
        os.makedirs(sent_dir, exist_ok=True)
        json.dump(cfg, open(os.path.join(output_dir, 'cfg.json'), 'w'))

    global_bs = cfg.data.train.batch_size
    if global_bs % world:
        raise SystemExit('global batch size %d is not divisible by the world size %d' % (global_bs, world))
    per_gpu = global_bs // world
    train_dataset = train_loader = val_loader = None
    if args.synthetic:
        vocab = json.load(open(cfg.data.vocab_json)) if os.path.exists(cfg.data.vocab_json) else \
            {'w%d' % i: i for i in range(1, 148)}
        iters_per_epoch = args.iters_per_epoch
    else:
        from ekaid_amd.datasets import create_dataset
        for k in ('vocab_json', 'splits_json', 'h5_label_file'):
            if not os.path.exists(cfg.data[k]):
                raise SystemExit('data file %s (cfg.data.%s) not found: pass --data_dir, fix the config, or use --synthetic '
                                 '(python -m ekaid_amd.datasets.generate writes a stand-in dataset)' % (cfg.data[k], k))
        train_dataset, train_loader = create_dataset(cfg, 'train', rank=rank, world_size=world, drop_last=world > 1)
        _, val_loader = create_dataset(cfg, 'test' if args.eval_target == 'test' else 'val')
        vocab = train_dataset.get_word_to_idx()
        iters_per_epoch = len(train_loader)
    idx_to_word = {i: w for w, i in vocab.items()}
    V = len(idx_to_word) + 1
    D = cfg.model.change_detector.feat_dim - 2
    change_detector = ChangeDetector(cfg, vocab)
    speaker = DynamicSpeaker(cfg, V)
    ck = None
    if args.resume:
        ck = load_checkpoint(args.checkpoint)
        change_detector.load_state_dict(ck['change_detector_state'])
        speaker.load_state_dict(ck['speaker_state'])
    change_detector.to(device)
    speaker.to(device)
    AddSpatialInfo().to(device)
    if args.global_qc and world > 1:
        enable_global_question_attention(change_detector)

    o = cfg.train.optim
    if o.type != 'adam':
        raise SystemExit('only the configured optimiser (adam, utils.py:96-99) is implemented')
    optimizer = FlatAdam([change_detector, speaker], lr=o.lr, betas=(o.alpha, o.beta), eps=o.epsilon)
    graph_comm = args.graph_comm if args.graph_comm != 'auto' else ('captured' if world > 1 else 'eager')
    t_round = args.t_round if args.t_round is not None else (8 if args.hip_graph else 1)
    step = TrainStep(cfg, change_detector, speaker, optimizer, args.graph, args.setting, use_graph=args.hip_graph, comm=graph_comm,
                     t_round=t_round)
    shape = dict(N=args.nodes, D=D, Lq=20, T=args.answer_steps, seq_length=cfg.model.speaker.seq_length, V=V)

    def named():
        return [('cd.' + n, q) for n, q in change_detector.named_parameters()] + \
               [('sp.' + n, q) for n, q in speaker.named_parameters()]

    t, epoch, i0, best = 0, 0, 0, -1.0
    epoch_rng = None
    if ck is not None and 'optimizer_state' in ck:      # a checkpoint of this framework: continue where it stopped
        optimizer.load_state_dict(ck['optimizer_state'], named())
        t, epoch, i0 = int(ck['iteration']), int(ck['epoch']), int(ck.get('iter_in_epoch', 0))
        best = float(ck.get('best', -1.0))
        ops.set_seed_state((args.seed + rank, int(ck['dropout_seed_counter'])))      # per-rank base, shared counter
        epoch_rng = ck.get('epoch_rng_state')

    def batches(epoch, skip):
        """Device batches of one epoch, starting after ``skip`` iterations: (dict, host labels, loss denominator)."""
        if args.synthetic:
            for i in range(skip, iters_per_epoch):
                host = make_batch(B=global_bs, seed=args.seed + 7919 * (epoch * iters_per_epoch + i), **shape)
                msum = float(host['masks'][:, 1:].sum())
                mine = slice_batch(host, rank * per_gpu, (rank + 1) * per_gpu)
                b = {k: v.to(device, non_blocking=True) for k, v in mine.items()}
                yield b, mine['labels'], msum
            return
        from ekaid_amd.datasets.rcc_dataset_pos_mimic import DevicePrefetcher
        train_loader.rank_slice.skip = skip
        pf = DevicePrefetcher(train_loader, device, timing=args.time_from is not None)
        prefetchers.append(pf)
        for b in pf:
            gidx = b['_meta']['global_indices']
            yield b, b['_host_labels'], (train_dataset.mask_sum(gidx) if world > 1 else None)

    set_mode('train', [change_detector, speaker])
    loss_avg = AverageMeter()
    last = {}
    prefetchers, timing, t_clock, steps_hist, run_hist = [], None, None, {}, {}

    def stop_clock():
        torch.cuda.synchronize()
        dt, n_it = time.perf_counter() - t_clock[0], t - t_clock[1]
        keys = ('h2d_ms', 'stall_ms', 'loader_wait_s', 'batches', 'h2d_bytes', 'produce_s', 'stage_s', 'gpu_wait_s')
        before = {k: sum(d[k] for d in t_clock[2]) for k in keys}
        after = {k: sum(pf.stats()[k] for pf in prefetchers) for k in keys}
        nb = max(after['batches'] - before['batches'], 1)
        return {'iterations': n_it, 'ms_per_iteration': dt / n_it * 1e3, 'qa_pairs_per_s': global_bs * n_it / dt,
                'loader_wait_ms_per_iteration': (after['loader_wait_s'] - before['loader_wait_s']) / nb * 1e3,
                'host_waits_for_device_ms_per_iteration': (after['gpu_wait_s'] - before['gpu_wait_s']) / nb * 1e3,
                'batch_assembly_ms': (after['produce_s'] - before['produce_s']) / nb * 1e3,
                'staging_ms': (after['stage_s'] - before['stage_s']) / nb * 1e3,
                'h2d_ms_per_batch': (after['h2d_ms'] - before['h2d_ms']) / nb,
                'exposed_h2d_ms_per_iteration': (after['stall_ms'] - before['stall_ms']) / nb,
                'h2d_bytes_per_batch': (after['h2d_bytes'] - before['h2d_bytes']) // nb,
                'decode_steps_histogram': dict(sorted(steps_hist.items()))}
    while t < cfg.train.max_iter:
        say('Starting epoch %d' % epoch)
        optimizer.lr = step_lr(o.lr, o.gamma, o.step_size, epoch)
        tr = cfg.train
        if epoch > tr.scheduled_sampling_start and tr.scheduled_sampling_start >= 0:       # train_mimic.py:194-202
            frac = (epoch - tr.scheduled_sampling_start) // tr.scheduled_sampling_increase_every
            ss = min(tr.scheduled_sampling_increase_prob * frac, tr.scheduled_sampling_max_prob)
            if ss != speaker.ss_prob:
                say('Updating scheduled sampling rate: %.4f -> %.4f' % (speaker.ss_prob, ss))
            speaker.ss_prob = ss
        if epoch_rng is not None:
            torch.set_rng_state(epoch_rng)                  # resumed mid-epoch: the sampler redraws the same permutation
        rng_at_epoch_start = torch.get_rng_state()
        epoch_rng = None
        i = i0
        for b, host_labels, msum in batches(epoch, i0):
            if t >= cfg.train.max_iter:
                break
            t0 = time.time()
            if args.time_from is not None and t == args.time_from:
                torch.cuda.synchronize()
                t_clock = (time.perf_counter(), t, [pf.stats() for pf in prefetchers])
            steps = DynamicSpeaker.count_steps(host_labels, shape['seq_length'])
            steps_hist[steps] = steps_hist.get(steps, 0) + 1
            run_hist[step.rounded_steps(steps)] = run_hist.get(step.rounded_steps(steps), 0) + 1
            denom = None
            if world > 1 and msum is not None:
                denom = torch.tensor([msum / world], dtype=torch.float32, device=device)
            loss, speaker_loss = step(b, steps, denom)
            t += 1
            i += 1
            if t_clock is not None and timing is None and t == cfg.train.max_iter:
                timing = stop_clock()           # before the final snapshot / evaluation
            if t % cfg.train.log_interval == 0 or t == cfg.train.max_iter:
                hip.seq_status_check()          # (synchronises, like the loss read below) a persistent kernel never hangs, it reports
                val = float(loss.item())
                loss_avg.update(val, 2 * per_gpu)
                last = {'speaker_loss': float(speaker_loss.item()), 'total_loss': val, 'avg_total_loss': loss_avg.avg}
                say(epoch, i - 1, t, last, time.time() - t0)
            if t % cfg.train.snapshot_interval == 0 or (t == cfg.train.max_iter and not args.no_final_snapshot):
                end_of_epoch = i >= iters_per_epoch
                checkpoint = None
                # never snapshot weights behind an unchecked persistent launch (the snapshot interval need not be a multiple of the log
                # interval); the Adam kernel has already refused the update of a faulted step on the device (ekaid_adam_step_dev)
                hip.seq_status_check()
                if rank == 0:
                    # the reference's three keys (train_mimic.py:283-287) + what a true resume needs; ``epoch`` /
                    # ``iter_in_epoch`` name the NEXT iteration to run
                    checkpoint = {'change_detector_state': change_detector.state_dict(),
                                  'speaker_state': speaker.state_dict(), 'model_cfg': cfg,
                                  'optimizer_state': optimizer.state_dict(named()), 'iteration': t,
                                  'epoch': epoch + 1 if end_of_epoch else epoch, 'iter_in_epoch': 0 if end_of_epoch else i,
                                  # mid-epoch: the state the epoch's sampler drew its permutation from; on an epoch
                                  # boundary: the state NOW, from which the next epoch's sampler will draw (evaluate() restores
                                  # the generator it found, so nothing between here and the next epoch start moves it)
                                  'epoch_rng_state': torch.get_rng_state() if end_of_epoch else rng_at_epoch_start,
                                  'dropout_seed_counter': ops.seed_state()[1], 'best': best}
                    save_checkpoint(checkpoint, os.path.join(snapshot_dir, 'checkpoint_%d.pt' % t))
                say('Running eval at iter %d' % t)
                if val_loader is not None:
                    if rank == 0:
                        preds, acc, n = evaluate(cfg, change_detector, speaker, val_loader, device, idx_to_word, args.graph,
                                                 args.eval_batches)
                        coco_gen_format_save(preds, os.path.join(sent_dir, 'eval_results_%d.json' % t))
                        say('eval at iter %d: exact-match accuracy %.4f over %d pairs' % (t, acc, n))
                        last['eval_accuracy'] = acc
                        # the reference picks its best checkpoint by BLEU-1 from pycocoevalcap (train_mimic.py:364-383): use it
                        # where the package exists, exact-match accuracy otherwise
                        from ekaid_amd.evaluation import caption_metrics
                        metrics = caption_metrics(preds, getattr(evaluate, 'last_references', {}))
                        if metrics:
                            say('eval at iter %d: %s' % (t, ', '.join('%s %.4f' % kv for kv in metrics.items())))
                            last['eval_metrics'] = metrics
                            acc = metrics.get('Bleu_1', acc)
                        if acc > best:
                            best = acc
                            checkpoint['best'] = best
                            save_checkpoint(checkpoint, os.path.join(snapshot_dir, 'checkpoint_best.pt'))
                            say('Best checkpoint saved')
                    if world > 1:
                        dist.barrier()
                else:
                    set_mode('eval', [change_detector, speaker])
                    with torch.no_grad():
                        outs = detector_forward(cfg, change_detector, b, args.graph, args.setting)
                        seq, _ = speaker._sample(outs[3], outs[4], outs[5], b['labels'], cfg, sample_max=1)
                        pred = decode_sequence(idx_to_word, seq)
                        gts = decode_sequence(idx_to_word, host_labels[:, 1:])
                        acc = sum(p == g for p, g in zip(pred, gts)) / max(len(gts), 1)
                    say('eval at iter %d: exact-match accuracy %.4f (greedy decode of the current batch)' % (t, acc))
                    set_mode('train', [change_detector, speaker])
        i0 = 0
        epoch += 1
    torch.cuda.synchronize()
    graphs_captured = len(step._graphs)
    step.close()            # drop the captured graphs and detach the dropout epoch word from the library (it dies with this run)
    if own_pg:
        dist.destroy_process_group()
    return {'iteration': t, 'epoch': epoch, 'lr': optimizer.lr, 'last': last, 'snapshot_dir': snapshot_dir,
            'change_detector': change_detector, 'speaker': speaker, 'optimizer': optimizer,
            'graphs_captured': graphs_captured, 'graph_capture_s': step.capture_s, 'graph_pool_bytes': step.graph_pool_bytes,
            'decode_steps_run_histogram': dict(sorted(run_hist.items())), 't_round': step.t_round,
            'timing': timing, 'iters_per_epoch': iters_per_epoch}


if __name__ == '__main__':
    main()
