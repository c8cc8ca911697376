#!/usr/bin/env python3
"""Training entry point -- build-owned counterpart of the reference's ``model/train_mimic.py`` (same command line,
same loop order, same checkpoint dict ``{'change_detector_state', 'speaker_state', 'model_cfg'}``), running the
difference-VQA path on MI355X through libekaid_hip.so.

    python train_mimic.py --graph all --synthetic --max_iter 20                 # no dataset needed
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train_mimic.py --synthetic ...

Differences from the reference script (ref: train_mimic.py:1-386), all deliberate:
  * one process per GPU with RCCL gradient all-reduce (the reference pins ``CUDA_VISIBLE_DEVICES=0``, :2);
  * ``--synthetic`` feeds seeded synthetic batches of the yaml shapes; the real-data reader (HDF5,
    rcc_dataset_pos_mimic.py:171-313) is the next scope row -- without ``--synthetic``
    the script explains that and exits;
  * no ``.item()`` synchronisation per iteration: losses are read only every ``log_interval`` iterations;
  * the in-loop evaluation computes greedy answers and exact-match accuracy; BLEU/METEOR/CIDEr need pycocoevalcap
    + Java (absent here) and stay delegated to the reference's ``evaluation.py``;
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
from ekaid_amd.config import cfg, merge_cfg_from_file, merge_into, MIMIC_YAML   # noqa: E402
from ekaid_amd.dp import FlatAdam, enable_global_question_attention      # noqa: E402
from ekaid_amd.models.dynamic_speaker_change_pos import DynamicSpeaker   # noqa: E402
from ekaid_amd.models.modules import AddSpatialInfo, ChangeDetector      # noqa: E402
from ekaid_amd.synthetic import make_batch                               # noqa: E402
from ekaid_amd.utils.mimic_utils import process_matrix                   # noqa: E402
from ekaid_amd.utils.utils import (AverageMeter, LanguageModelCriterion, decode_sequence, load_checkpoint,  # noqa: E402
                                   save_checkpoint, set_mode, total_loss)


def parse():
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
    p.add_argument('--synthetic', action='store_true')
    p.add_argument('--max_iter', type=int, default=None)
    p.add_argument('--batch_size', type=int, default=None)
    p.add_argument('--nodes', type=int, default=52, help='regions per image of the synthetic batches')
    p.add_argument('--answer_steps', type=int, default=16)
    p.add_argument('--global_qc', action='store_true', help='global-batch semantics for the question attention (Q-C)')
    p.add_argument('--exp_dir', default=None)
    return p.parse_args()


def main():
    args = parse()
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
    if args.exp_dir:
        cfg.exp_dir = args.exp_dir
    if not args.synthetic:
        raise SystemExit('The MIMIC-Diff-VQA HDF5 reader (SURVEY.md section 8 f-1) is not part of this round; run with '
                         '--synthetic to train on seeded synthetic batches of the same shapes.')

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
    hip.load()
    random.seed(args.seed)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)          # identical default initialisation on every rank
    ops.manual_seed(args.seed + rank)     # different dropout masks per rank

    exp_name = '%s_%s_%s_%s_%d' % (args.setting, args.feature_mode, args.graph, args.lr, args.seed)
    output_dir = os.path.join(cfg.exp_dir, 'temp', exp_name)
    snapshot_dir = os.path.join(output_dir, 'snapshots')
    if rank == 0:
        os.makedirs(snapshot_dir, exist_ok=True)
        json.dump(cfg, open(os.path.join(output_dir, 'cfg.json'), 'w'))

    vocab = json.load(open(cfg.data.vocab_json)) if os.path.exists(cfg.data.vocab_json) else \
        {'w%d' % i: i for i in range(1, 148)}
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

    lang_criterion = LanguageModelCriterion()
    o = cfg.train.optim
    if o.type != 'adam':
        raise SystemExit('only the configured optimiser (adam, utils.py:96-99) is implemented')
    optimizer = FlatAdam([change_detector, speaker], lr=o.lr, betas=(o.alpha, o.beta), eps=o.epsilon)
    per_gpu = cfg.data.train.batch_size // world if world > 1 else cfg.data.train.batch_size
    shape = dict(N=args.nodes, D=D, Lq=20, T=args.answer_steps, seq_length=cfg.model.speaker.seq_length, V=V)
    iters_per_epoch = 50

    def named():
        return [('cd.' + n, q) for n, q in change_detector.named_parameters()] + \
               [('sp.' + n, q) for n, q in speaker.named_parameters()]

    t, epoch = 0, 0
    if ck is not None and 'optimizer_state' in ck:      # a checkpoint of this framework: continue where it stopped
        optimizer.load_state_dict(ck['optimizer_state'], named())
        t, epoch = int(ck['iteration']), int(ck['epoch'])
        ops.set_seed_state(ck['dropout_seed_state'])
    set_mode('train', [change_detector, speaker])
    loss_avg = AverageMeter()
    while t < cfg.train.max_iter:
        if rank == 0:
            print('Starting epoch %d' % epoch)
        if epoch > 0 and epoch % o.step_size == 0:
            optimizer.lr = o.lr * (o.gamma ** (epoch // o.step_size))          # StepLR (train_mimic.py:174-177,190)
        for i in range(t - epoch * iters_per_epoch if 0 <= t - epoch * iters_per_epoch < iters_per_epoch else 0, iters_per_epoch):
            if t >= cfg.train.max_iter:
                break
            t0 = time.time()
            host = make_batch(B=per_gpu, seed=args.seed + 7919 * t, rank=rank, **shape)
            steps = DynamicSpeaker.count_steps(host['labels'], shape['seq_length'])
            b = {k: v.to(device, non_blocking=True) for k, v in host.items()}
            optimizer.zero_grad()
            N = b['input_1'].shape[1]
            adj = [process_matrix(b[k], cfg, N, device, type=ty) for k, ty in
                   (('d_adj', 'spatial'), ('q_adj', 'spatial'), ('d_sem', 'semantic'), ('q_sem', 'semantic'))]
            outs = change_detector(b['input_1'], b['input_2'], adj[0], adj[1], adj[2], adj[3], b['d_bb'], b['q_bb'],
                                   b['question'], setting=cfg.train.setting, graph=args.graph)
            logp, _ = speaker._forward(outs[3], outs[4], outs[5], b['labels'], num_steps=steps)
            speaker_loss = lang_criterion(logp, b['labels'][:, 1:], b['masks'][:, 1:])
            loss = total_loss(speaker_loss, outs[1], outs[2])
            loss.backward()
            optimizer.step()
            t += 1
            if t % cfg.train.log_interval == 0 or t == cfg.train.max_iter:
                val = float(loss.item())
                loss_avg.update(val, 2 * per_gpu)
                if rank == 0:
                    print(epoch, i, t, {'speaker_loss': float(speaker_loss.item()), 'total_loss': val,
                                        'avg_total_loss': loss_avg.avg}, time.time() - t0)
            if t % cfg.train.snapshot_interval == 0 or t == cfg.train.max_iter:
                if rank == 0:
                    # the reference's three keys (train_mimic.py:283-287) + what a true resume needs
                    checkpoint = {'change_detector_state': change_detector.state_dict(),
                                  'speaker_state': speaker.state_dict(), 'model_cfg': cfg,
                                  'optimizer_state': optimizer.state_dict(named()), 'iteration': t, 'epoch': epoch,
                                  'dropout_seed_state': ops.seed_state()}
                    save_checkpoint(checkpoint, os.path.join(snapshot_dir, 'checkpoint_%d.pt' % t))
                set_mode('eval', [change_detector, speaker])
                with torch.no_grad():
                    seq, _ = speaker._sample(outs[3].detach(), outs[4].detach(), outs[5].detach(), b['labels'], cfg,
                                             sample_max=1)
                    pred = decode_sequence(idx_to_word, seq)
                    gts = decode_sequence(idx_to_word, b['labels'][:, 1:])
                    acc = sum(p == g for p, g in zip(pred, gts)) / max(len(gts), 1)
                if rank == 0:
                    print('eval at iter %d: exact-match accuracy %.4f (greedy decode of the current batch)' % (t, acc))
                set_mode('train', [change_detector, speaker])
        epoch += 1
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
