"""Shared builders for the parity tests."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from ekaid_amd.config import bench_cfg, small_cfg          # noqa: E402
from ekaid_amd.formula_init import build_models, formula_init_, state_copy as oracle_params   # noqa: E402,F401
from ekaid_amd.synthetic import make_batch                 # noqa: E402

SMALL = dict(B=2, N=8, D=128, Lq=5, T=4, seq_length=6, V=31)
FULL = dict(N=36, D=2048, Lq=20, T=16, seq_length=90, V=148)


def small_models(graph='all', setting='mode2', device='cpu'):
    cfg = small_cfg(graph=graph, setting=setting)
    cd, sp = build_models(cfg, 30, 31, device)
    return cfg, cd, sp


def to_device(batch, device):
    """H2D exactly like train_mimic.py:213-218: boxes stay on the host as float64."""
    out = {}
    for k, v in batch.items():
        out[k] = v if k in ('d_bb', 'q_bb') else v.to(device)
    return out


def real_shape_case():
    """(cfg, batch) of the real-data-shape parity case: Medical-Diff-VQA shapes (52 regions x 1024-d; config.yaml widths shrunk only
    in att_dim to keep the CPU runs fast), ragged answer lengths (per-sample masks), an undetected region (all-zero ROI row ->
    question masked out only if the projected row sums to zero, relation_encoder.py:23-27) and a fully isolated node."""
    cfg = small_cfg(graph='all', D=1024, att_dim=128, heads=4, qdim=64, seq_length=12)
    batch = make_batch(B=3, N=52, D=1024, Lq=9, T=7, seq_length=12, V=31, seed=5)
    batch['input_1'][0, 5] = 0                               # undetected region
    batch['labels'][1, 4:] = 0                               # shorter answers
    batch['labels'][2, 6:] = 0
    batch['masks'] = torch.zeros_like(batch['masks'])
    for b in range(3):
        n = int((batch['labels'][b] != 0).sum()) + 1
        batch['masks'][b, :n] = 1
    batch['d_adj'][0, 7, :] = 0
    batch['d_adj'][0, :, 7] = 0                              # isolated node in the spatial graph
    return cfg, batch


# --------------------------------------------------------------------------- #
# round-6 cases at the production widths (fixtures: tests/golden/make_wide_golden.py -> f10_*.npz)
# --------------------------------------------------------------------------- #
def wide_case(T, B=4):
    """(cfg, batch): BASELINE.json shapes (36 regions x 2048-d, 20-token questions), yaml widths, T decode steps -- 40 and 90 reach the
    persistent decoder backward (taken from 32 steps on; dynamic_speaker_change_pos.py:196-222)."""
    cfg = bench_cfg(D=2048, seq_length=90)
    return cfg, make_batch(B=B, N=36, D=2048, Lq=20, T=T, seq_length=90, V=148, seed=1238 + T)


def real_full_case():
    """(cfg, batch): Medical-Diff-VQA shapes at the yaml widths (52 regions x 1024-d: rcc_dataset_pos_mimic.py:107-108, feat_dim 1026),
    batch 4, ragged answers of 2 ... 40 tokens with per-sample masks, an undetected region (all-zero ROI row), an isolated node."""
    cfg = bench_cfg(D=1024, seq_length=90)
    batch = make_batch(B=4, N=52, D=1024, Lq=20, T=41, seq_length=90, V=148, seed=52)
    batch['input_1'][0, 5] = 0                               # undetected region
    batch['input_2'][3, 51] = 0
    for b, n in enumerate((40, 2, 17, 29)):                  # answer lengths in tokens (labels[:, 0] is <start>)
        batch['labels'][b, 1 + n:] = 0
    batch['masks'] = torch.zeros_like(batch['masks'])
    for b in range(4):
        n = int((batch['labels'][b] != 0).sum()) + 1
        batch['masks'][b, :n] = 1
    batch['d_adj'][0, 7, :] = 0
    batch['d_adj'][0, :, 7] = 0                              # isolated node in the spatial graph
    batch['q_sem'][2, 11, :] = 0
    batch['q_sem'][2, :, 11] = 0                             # ... and one in the semantic graph of the other image
    return cfg, batch


def greedy_case(B=64):
    """(cfg, (bef, aft, diff), logit scale, NULL-bias shift): decoder inputs of the greedy-decoding fixture, from numpy's portable PCG64."""
    import numpy as np
    cfg = bench_cfg(D=2048, seq_length=90)
    A = cfg.model.speaker.input_dim
    rng = np.random.default_rng([64, 1238])
    feats = tuple(torch.from_numpy(rng.standard_normal((B, A), dtype=np.float32)) for _ in range(3))
    return cfg, feats, 12.0, 3.0


def oracle_gradients(cfg, cd, sp, batch, double=False, graph='all', setting='mode2'):
    """(loss, result dict, {name: gradient}) of the CPU oracle on the weights of the modules ``cd`` / ``sp``; ``double``: the whole run in
    float64 (weights, inputs, every intermediate) -- the gradient truth where float32 itself drifts (long answers at full width)."""
    from oracle import ekaid_oracle as O
    Pcd, Psp = oracle_params(cd, not double), oracle_params(sp, not double)
    if double:
        Pcd = {k: (v.double().requires_grad_(True) if v.dtype.is_floating_point else v) for k, v in Pcd.items()}
        Psp = {k: (v.double().requires_grad_(True) if v.dtype.is_floating_point else v) for k, v in Psp.items()}
        batch = {k: (v.double() if v.dtype == torch.float32 else v) for k, v in batch.items()}
    loss, res = O.train_loss(Pcd, Psp, cfg, batch, setting, graph)
    loss.backward()
    grads = {('cd.' + k): Pcd[k].grad for k, _ in cd.named_parameters()}
    grads.update({('sp.' + k): Psp[k].grad for k, _ in sp.named_parameters()})
    return loss.detach(), res, grads


def mode0_case():
    """(cfg, batch): setting 'mode0' at the yaml widths -- the SSRE block (modules.py:17-77) in place of the graph encoders."""
    cfg = bench_cfg(D=2048, seq_length=90, setting='mode0')
    return cfg, make_batch(B=2, **FULL)
