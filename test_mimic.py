#!/usr/bin/env python3
"""Evaluation entry point -- counterpart of the reference's ``model/test_mimic.py`` (:25-152): load a checkpoint in the
reference's layout, greedy-decode a test split on MI355X, write COCO-format captions.

    python test_mimic.py -p experiments/temp/<run>/snapshots/checkpoint_20.pt                  # cfg.data.* HDF5 files
    python test_mimic.py -p .../checkpoint_20.pt --data_dir /tmp/d                             # generated dataset
    python test_mimic.py -p .../checkpoint_20.pt --synthetic                                   # seeded batches, no files

Caption metrics (BLEU/METEOR/ROUGE-L/CIDEr through pycocoevalcap + Java) are outside the hot path and absent from
this image; the script reports exact-match accuracy (reference evaluate_score.py:216-251) and writes the
``test_results_<feature_mode>.json`` file the reference's ``evaluation.py`` consumes.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from ekaid_amd import hip                                                 # noqa: E402
from ekaid_amd.config import cfg, default_cfg, merge_cfg_from_file, merge_into, MIMIC_YAML   # noqa: E402
from ekaid_amd.models.dynamic_speaker_change_pos import DynamicSpeaker   # noqa: E402
from ekaid_amd.models.modules import ChangeDetector                      # noqa: E402
from ekaid_amd.synthetic import make_batch                               # noqa: E402
from ekaid_amd.utils.mimic_utils import process_matrix                   # noqa: E402
from ekaid_amd.utils.utils import coco_gen_format_save, decode_sequence, load_checkpoint, set_mode  # noqa: E402


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--cfg', default='')
    p.add_argument('-p', '--resume_checkpoint', type=str, required=True)
    p.add_argument('--gpu', type=int, default=-1)
    p.add_argument('--feature_mode', type=str, default='location')
    p.add_argument('--graph', type=str, default='all', choices=['implicit', 'semantic', 'spatial', 'all', 'i+s'])
    p.add_argument('--synthetic', action='store_true')
    p.add_argument('--batches', type=int, default=4)
    p.add_argument('--nodes', type=int, default=52)
    p.add_argument('--data_dir', default='')
    p.add_argument('--split', default='test', choices=['test', 'val', 'train', 'all'])
    p.add_argument('--exp_dir', default=None)
    p.add_argument('--hip_graph', action='store_true', help='replay the detector forward from a hipGraph per batch shape')
    args = p.parse_args(argv)
    cfg.clear()
    cfg.update(default_cfg())
    if args.cfg:
        merge_cfg_from_file(args.cfg)
    else:
        merge_into(json.loads(json.dumps(MIMIC_YAML)), cfg)
    cfg.train.graph = args.graph
    cfg.data.feature_mode = args.feature_mode
    if args.exp_dir:
        cfg.exp_dir = args.exp_dir
    if args.data_dir:
        from ekaid_amd.datasets.generate import point_cfg_at
        point_cfg_at(cfg, args.data_dir)
    device = torch.device('cuda', max(args.gpu, 0))
    torch.cuda.set_device(device)
    hip.load()
    ck = load_checkpoint(args.resume_checkpoint)
    dataset = loader = None
    if args.synthetic:
        vocab = json.load(open(cfg.data.vocab_json)) if os.path.exists(cfg.data.vocab_json) else \
            {'w%d' % i: i for i in range(1, 148)}
    else:
        from ekaid_amd.datasets import create_dataset
        dataset, loader = create_dataset(cfg, args.split)
        vocab = dataset.get_word_to_idx()
    idx_to_word = {i: w for w, i in vocab.items()}
    V = len(idx_to_word) + 1
    change_detector = ChangeDetector(cfg, vocab)
    change_detector.load_state_dict(ck['change_detector_state'])
    speaker = DynamicSpeaker(cfg, V)
    speaker.load_state_dict(ck['speaker_state'])
    change_detector.to(device)
    speaker.to(device)
    set_mode('eval', [change_detector, speaker])
    out_dir = os.path.join(cfg.exp_dir, 'mimic-diff', 'test_output', 'captions', 'test')
    os.makedirs(out_dir, exist_ok=True)
    D = cfg.model.change_detector.feat_dim - 2
    predictions, correct, total = {}, 0, 0
    sequences, sequence_logprobs = [], []
    t0 = time.time()
    if loader is not None:
        # the reference's loop (test_mimic.py:97-141) over the real split: device feed, detector, greedy decode
        from train_mimic import evaluate
        predictions, accuracy, total = evaluate(cfg, change_detector, speaker, loader, device, idx_to_word, args.graph,
                                                hip_graph=args.hip_graph)
        correct = round(accuracy * total)
    with torch.no_grad():
        for i in range(args.batches if loader is None else 0):
            host = make_batch(B=cfg.data.test.batch_size, N=args.nodes, D=D, Lq=20, T=16,
                              seq_length=cfg.model.speaker.seq_length, V=V, seed=977 + i)
            b = {k: v.to(device) for k, v in host.items()}
            N = b['input_1'].shape[1]
            adj = [process_matrix(b[k], cfg, N, device, type=ty) for k, ty in
                   (('d_adj', 'spatial'), ('q_adj', 'spatial'), ('d_sem', 'semantic'), ('q_sem', 'semantic'))]
            outs = change_detector(b['input_1'], b['input_2'], adj[0], adj[1], adj[2], adj[3], b['d_bb'], b['q_bb'],
                                   b['question'], setting=cfg.train.setting, graph=args.graph)
            seq, seq_lp = speaker._sample(outs[3], outs[4], outs[5], b['labels'], cfg, sample_max=1)
            sequences.append(seq.cpu())
            sequence_logprobs.append(seq_lp.cpu())
            sents = decode_sequence(idx_to_word, seq)
            gts = decode_sequence(idx_to_word, b['labels'][:, 1:])
            for j, s in enumerate(sents):
                predictions[str(i * len(sents) + j)] = s
                correct += int(s == gts[j])
                total += 1
    print('Test took %.4f seconds' % (time.time() - t0))
    path = os.path.join(out_dir, 'test_results_%s.json' % args.feature_mode)
    coco_gen_format_save(predictions, path)
    print(path)
    print('exact-match accuracy: %.4f (%d/%d)' % (correct / max(total, 1), correct, total))
    metrics = None
    if loader is not None:
        from ekaid_amd.evaluation import caption_metrics
        from train_mimic import evaluate as _ev
        metrics = caption_metrics(predictions, getattr(_ev, 'last_references', {}))        # test_mimic.py:143-152, when pycocoevalcap exists
        if metrics:
            print(', '.join('%s %.4f' % kv for kv in metrics.items()))
    return {'results_file': path, 'accuracy': correct / max(total, 1), 'n': total, 'predictions': predictions, 'metrics': metrics,
            'sequences': sequences, 'sequence_logprobs': sequence_logprobs}


if __name__ == '__main__':
    main()
