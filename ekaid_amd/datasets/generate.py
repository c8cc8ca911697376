"""Write a seeded synthetic dataset in the on-disk layout of the reference (no MIMIC data needed):

    <dir>/cmb_bbox_di_feats.hdf5   image_features (n,52,D) f32, image_bb (n,52,4) f32, image_adj_matrix /
                                   semantic_adj_matrix (n,100,100) i64, bbox_label (n,52) i64     (combine_dicts.py:162-216)
    <dir>/VQA_mimic_dataset.h5     questions (m,20), answers / pos (m,90), label_start_idx / label_end_idx (m,1),
                                   feature_idx (m,2), all i64                                      (dataset_preparation.py:27-88)
    <dir>/vocab_mimic_VQA.json     word -> id, ids 1..147, '<start>' = 1                           (dataset_preparation.py:131-215)
    <dir>/splits_mimic_VQA.json    80 / 10 / 10 % by pair index                                    (dataset_preparation.py:201-205)

Value ranges follow ekaid_amd/synthetic.py (post-ReLU features, boxes on a 1024^2 canvas, spatial labels 0..11 with the
reverse relation on the transposed entry, symmetric semantic labels 0..2) with RAGGED answers (2..40 tokens, a few up to
the 90-token maximum) like the real data.  Files are written by ``ekaid_amd.h5lite.write`` (chunked, resizable, no filters:
what h5py writes for the reference) so ``train_mimic.py`` / ``test_mimic.py`` run on them without ``--synthetic``.

    python -m ekaid_amd.datasets.generate --out /tmp/ekaid_data --images 96 --pairs 400
"""
import argparse
import json
import os

import numpy as np

from .. import h5lite
from ..synthetic import _REV


def make_dataset(out_dir, n_images=96, n_pairs=400, nodes=52, D=1024, V=148, seed=0, chunk_rows=100, max_answer=40, answer_mix='short+90'):
    """``answer_mix``: 'short+90' = answers uniform 2..max_answer tokens, every 50th pair the 90-token maximum (most 64-pair batches then
    decode 90 steps, the rest ~max_answer); 'uniform' = uniform 2..90 tokens, nothing forced."""
    os.makedirs(out_dir, exist_ok=True)
    rng = np.random.default_rng([seed, 77])
    feats = np.maximum(rng.standard_normal((n_images, nodes, D), dtype=np.float32), 0.0)
    feats[:, :, 0] += 0.25
    undetected = rng.random((n_images, nodes)) < 0.03                 # zero-filled regions (ana_bbox_generator.py:591-594)
    feats[undetected] = 0.0
    x1 = rng.uniform(0.0, 700.0, (n_images, nodes))
    y1 = rng.uniform(0.0, 700.0, (n_images, nodes))
    bb = np.stack([x1, y1, x1 + rng.uniform(32.0, 320.0, (n_images, nodes)),
                   y1 + rng.uniform(32.0, 320.0, (n_images, nodes))], -1).astype(np.float32)
    rev = np.zeros(12, dtype=np.int64)
    for a, b in _REV.items():
        rev[a] = b
    iu = np.triu_indices(nodes, 1)
    adj = np.zeros((n_images, 100, 100), dtype=np.int64)
    val = rng.integers(1, 12, (n_images, len(iu[0]))) * (rng.random((n_images, len(iu[0]))) >= 0.3)
    adj[:, iu[0], iu[1]] = val
    adj[:, iu[1], iu[0]] = rev[val]
    sem = np.zeros((n_images, 100, 100), dtype=np.int64)
    sval = rng.choice(3, size=(n_images, len(iu[0])), p=[0.7, 0.2, 0.1])
    sem[:, iu[0], iu[1]] = sval
    sem[:, iu[1], iu[0]] = sval
    lab = rng.integers(0, 40, (n_images, nodes)).astype(np.int64)
    h5lite.write(os.path.join(out_dir, 'cmb_bbox_di_feats.hdf5'), {
        'image_features': (feats, chunk_rows), 'image_bb': (bb, chunk_rows), 'image_adj_matrix': (adj, chunk_rows),
        'semantic_adj_matrix': (sem, chunk_rows), 'bbox_label': (lab, chunk_rows)})

    q = np.zeros((n_pairs, 20), np.int64)
    a = np.zeros((n_pairs, 90), np.int64)
    pos = np.zeros((n_pairs, 90), np.int64)
    for i in range(n_pairs):
        ql = int(rng.integers(3, 13))
        q[i, :ql] = rng.integers(2, V, ql)
        if answer_mix == 'uniform':
            al = int(rng.integers(2, 91))
        else:
            al = int(rng.integers(2, max_answer + 1)) if i % 50 else 90
        a[i, 0] = 1
        a[i, 1:al] = rng.integers(2, V, al - 1)
        pos[i, :al] = rng.integers(1, 16, al)
    start = np.arange(n_pairs, dtype=np.int64).reshape(n_pairs, 1)
    fidx = rng.integers(0, n_images, (n_pairs, 2)).astype(np.int64)
    h5lite.write(os.path.join(out_dir, 'VQA_mimic_dataset.h5'), {
        'questions': (q, chunk_rows), 'answers': (a, chunk_rows), 'pos': (pos, chunk_rows),
        'label_start_idx': (start, chunk_rows), 'label_end_idx': (start + 1, chunk_rows), 'feature_idx': (fidx, chunk_rows)})
    vocab = {'<start>': 1}
    for i in range(2, V):
        vocab['w%d' % i] = i
    json.dump(vocab, open(os.path.join(out_dir, 'vocab_mimic_VQA.json'), 'w'), indent=4)
    idx = list(range(n_pairs))
    c1, c2 = int(np.ceil(0.8 * n_pairs)), int(np.ceil(0.9 * n_pairs))
    json.dump({'train': idx[:c1], 'val': idx[c1:c2], 'test': idx[c2:]},
              open(os.path.join(out_dir, 'splits_mimic_VQA.json'), 'w'), indent=4)
    return out_dir


def point_cfg_at(cfg, data_dir):
    """Set the four data paths of ``cfg`` to the files ``make_dataset`` wrote."""
    cfg.data.vocab_json = os.path.join(data_dir, 'vocab_mimic_VQA.json')
    cfg.data.splits_json = os.path.join(data_dir, 'splits_mimic_VQA.json')
    cfg.data.h5_label_file = os.path.join(data_dir, 'VQA_mimic_dataset.h5')
    cfg.data.h5_feature_file = os.path.join(data_dir, 'cmb_bbox_di_feats.hdf5')
    return cfg


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', required=True)
    ap.add_argument('--images', type=int, default=96)
    ap.add_argument('--pairs', type=int, default=400)
    ap.add_argument('--nodes', type=int, default=52)
    ap.add_argument('--dim', type=int, default=1024)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--answer_mix', choices=['short+90', 'uniform'], default='short+90')
    a = ap.parse_args()
    print(make_dataset(a.out, a.images, a.pairs, a.nodes, a.dim, seed=a.seed, answer_mix=a.answer_mix))
