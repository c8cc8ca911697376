"""Seeded synthetic (image-pair, question, answer) batches.

Shapes and value ranges follow SURVEY.md section 8(d) / BASELINE.md section 3; the
tuple layout is the 13-tuple the reference's ``rcc_collate`` yields (reference
``datasets/rcc_dataset_pos_mimic.py:171-313``) minus the pair index:

  input_1/input_2  f32 (B,N,D)   post-ReLU ROI features, max(N(0,1),0), no all-zero row
  d_bb/q_bb        f64 (B,N,4)   x1,y1,x2,y2 on a 1024^2 canvas  (stay on host)
  d_adj/q_adj      f64 (B,100,100) spatial labels 0..11 (antisymmetric pairs), as on disk
  d_sem/q_sem      f64 (B,100,100) semantic labels 0..2 (symmetric)
  question         i64 (B,Lq)    col 0 = 1, then 3..11 tokens in 2..V-1, 0-padded
  labels           i64 (B,Tmax+1) col 0 = <start>=1, T-1 tokens in 2..V-1, rest 0
  masks            f32 (B,Tmax+1) ones on the first T+1 columns

numpy's PCG64 ``default_rng`` is bit-reproducible across machines, so the build
container (reference import, golden fixtures) and the GPU box see the same data.
"""
import numpy as np
import torch

# reverse relation of spatial label l (1..11): the pair (i,j)=l implies (j,i)=REV[l]
# (reference "feature extraction/ana_bbox_generator.py:266-335": 1/2 inside/cover,
#  3 overlap (symmetric), 4..11 the eight direction octants, opposite = +4 mod 8)
_REV = {1: 2, 2: 1, 3: 3, 4: 8, 5: 9, 6: 10, 7: 11, 8: 4, 9: 5, 10: 6, 11: 7}


def make_batch(B, N=36, D=2048, Lq=20, T=16, seq_length=90, V=148, seed=1238,
               rank=0, pad_nodes=100):
    rng = np.random.default_rng([seed, rank])
    out = {}
    for key in ('input_1', 'input_2'):
        x = np.maximum(rng.standard_normal((B, N, D), dtype=np.float32), 0.0)
        x[:, :, 0] += 0.25                      # guarantees no all-zero ROI row
        out[key] = x
    for key in ('d_bb', 'q_bb'):
        x1 = rng.uniform(0.0, 700.0, (B, N))
        y1 = rng.uniform(0.0, 700.0, (B, N))
        w = rng.uniform(32.0, 320.0, (B, N))
        h = rng.uniform(32.0, 320.0, (B, N))
        out[key] = np.stack([x1, y1, x1 + w, y1 + h], -1).astype(np.float64)
    rev = np.zeros(12, dtype=np.int64)
    for a, b in _REV.items():
        rev[a] = b
    iu = np.triu_indices(N, 1)
    for key in ('d_adj', 'q_adj'):
        lab = np.zeros((B, pad_nodes, pad_nodes), dtype=np.int64)
        has = rng.random((B, len(iu[0]))) >= 0.3
        val = rng.integers(1, 12, (B, len(iu[0]))) * has
        lab[:, iu[0], iu[1]] = val
        lab[:, iu[1], iu[0]] = rev[val]
        out[key] = lab.astype(np.float64)
    for key in ('d_sem', 'q_sem'):
        lab = np.zeros((B, pad_nodes, pad_nodes), dtype=np.int64)
        val = rng.choice(3, size=(B, len(iu[0])), p=[0.7, 0.2, 0.1])
        lab[:, iu[0], iu[1]] = val
        lab[:, iu[1], iu[0]] = val
        out[key] = lab.astype(np.float64)
    q = np.zeros((B, Lq), dtype=np.int64)
    q[:, 0] = 1
    qlen = rng.integers(3, max(4, min(12, Lq)), B)
    for b in range(B):
        n = int(min(qlen[b], Lq - 1))
        q[b, 1:1 + n] = rng.integers(2, V, n)
    out['question'] = q
    lab = np.zeros((B, seq_length + 1), dtype=np.int64)
    lab[:, 0] = 1
    lab[:, 1:T] = rng.integers(2, V, (B, T - 1))
    out['labels'] = lab
    m = np.zeros((B, seq_length + 1), dtype=np.float32)
    m[:, :T + 1] = 1.0
    out['masks'] = m
    return {k: torch.from_numpy(v) for k, v in out.items()}


def slice_batch(batch, lo, hi):
    """Contiguous sample slice [lo, hi) of a batch dict (data-parallel sharding)."""
    return {k: v[lo:hi].contiguous() for k, v in batch.items()}
