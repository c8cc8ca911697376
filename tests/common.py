"""Shared builders for the parity tests."""
import contextlib
import io
import os
import sys
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from ekaid_amd.config import bench_cfg, small_cfg          # noqa: E402
from ekaid_amd.formula_init import formula_init_           # noqa: E402
from ekaid_amd.synthetic import make_batch                 # noqa: E402

SMALL = dict(B=2, N=8, D=128, Lq=5, T=4, seq_length=6, V=31)
FULL = dict(N=36, D=2048, Lq=20, T=16, seq_length=90, V=148)


def build_models(cfg, n_words, vocab, device='cpu', train=False):
    """Product modules (ekaid_amd.models) with formula weights."""
    from ekaid_amd.models.modules import ChangeDetector
    from ekaid_amd.models.dynamic_speaker_change_pos import DynamicSpeaker
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        with contextlib.redirect_stdout(io.StringIO()):
            cd = ChangeDetector(cfg, {str(i): i for i in range(n_words)})
            sp = DynamicSpeaker(cfg, vocab)
    formula_init_(cd, 'cd.')
    formula_init_(sp, 'sp.')
    cd.train(train)
    sp.train(train)
    return cd.to(device), sp.to(device)


def oracle_params(module, requires_grad=False):
    """state_dict copy on CPU for the functional oracle."""
    P = {}
    for k, v in module.state_dict().items():
        t = v.detach().cpu().clone()
        if requires_grad and t.dtype.is_floating_point:
            t.requires_grad_(True)
        P[k] = t
    return P


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
