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
