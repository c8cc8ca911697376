"""Closed-form ("formula") weight initialisation.

The full-width model has 87 M parameters -- too large to ship as a fixture -- so
both sides of every parity check (the reference imported in the build container,
the CPU oracle, and the HIP path on the GPU box) build *identical* weights from
the tensor's ``state_dict`` key alone (fixture F4):

    w[i] = scale(name, shape) * (fmix32(i * 0x9E3779B1 + crc32(name)) / 2^31 - 1)

(i = flat index; integer hashing only, so the values are exact and portable).

``weight_g`` of a weight-normed layer is set from the Frobenius norm of its
``weight_v`` partner times a name-derived factor in [0.75, 1.25).
All arithmetic is float64 numpy, rounded once to float32.
"""
import zlib

import numpy as np
import torch


def _hash(name):
    return zlib.crc32(name.encode('utf-8')) & 0xFFFFFFFF


def _fmix32(x):
    """murmur3 32-bit finaliser on a uint64 array holding 32-bit values (exact, portable)."""
    m = np.uint64(0xFFFFFFFF)
    x = x & m
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & m
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & m
    x ^= x >> np.uint64(16)
    return x


def _hash_fill(name, shape, scale):
    """w[i] = scale * u_i,  u_i = fmix32(i * 0x9E3779B1 + crc32(name)) / 2^31 - 1  in [-1, 1)."""
    n = int(np.prod(shape)) if len(shape) else 1
    i = np.arange(n, dtype=np.uint64)
    x = _fmix32(i * np.uint64(0x9E3779B1) + np.uint64(_hash(name)))
    u = x.astype(np.float64) / 2147483648.0 - 1.0
    return (scale * u).reshape(shape).astype(np.float32)


EMBEDDING_KEYS = ('cd.w_emb.emb.weight', 'cd.w_emb.emb_.weight', 'sp.embed.0.weight')


def _scale_for(name, shape):
    if name in EMBEDDING_KEYS:
        return 0.9                       # lookup tables
    leaf = name.rsplit('.', 1)[-1]
    if leaf.startswith('bias') or len(shape) <= 1:
        return 0.1
    fan_in = int(np.prod(shape[1:]))
    return 1.7320508075688772 / np.sqrt(max(fan_in, 1))     # unit-gain: std = 1/sqrt(fan_in)


def formula_state(shapes):
    """``shapes``: ordered mapping key -> shape tuple. Returns key -> float32 ndarray."""
    out = {}
    for name, shape in shapes.items():
        if name.endswith('weight_g'):
            continue
        out[name] = _hash_fill(name, tuple(shape), _scale_for(name, tuple(shape)))
    for name, shape in shapes.items():
        if not name.endswith('weight_g'):
            continue
        v = out[name[:-1] + 'v']
        fac = 0.75 + (_hash(name) % 997) / 997.0 * 0.5
        g = np.float32(np.sqrt(np.sum(v.astype(np.float64) ** 2)) * fac)
        out[name] = np.full(tuple(shape), g, dtype=np.float32)
    return {k: out[k] for k in shapes}


@torch.no_grad()
def formula_init_(module, prefix):
    """Fill every entry of ``module.state_dict()`` in place; returns the module.

    ``prefix`` namespaces the keys: ``'cd.'`` for the ChangeDetector, ``'sp.'`` for
    the DynamicSpeaker (both modules own an ``embed.0.weight``)."""
    sd = module.state_dict()
    shapes = {prefix + k: tuple(v.shape) for k, v in sd.items()
              if v.dtype.is_floating_point}
    vals = formula_state(shapes)
    for k, arr in vals.items():
        t = sd[k[len(prefix):]]
        t.copy_(torch.from_numpy(arr).to(t.device))
    return module


def build_models(cfg, n_words, vocab, device='cpu', train=False):
    """The two product modules (``ChangeDetector``, ``DynamicSpeaker``) with the closed-form weights of this file: what the
    benchmark, the smoke test and the parity tests construct (identical on every rank, no checkpoint needed)."""
    import contextlib
    import io
    import warnings
    from .models.dynamic_speaker_change_pos import DynamicSpeaker
    from .models.modules import ChangeDetector
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


def state_copy(module, requires_grad=False):
    """CPU copy of a module's ``state_dict`` (optionally as autograd leaves): the weights handed to a functional
    restatement of the path for comparison."""
    P = {}
    for k, v in module.state_dict().items():
        t = v.detach().cpu().clone()
        if requires_grad and t.dtype.is_floating_point:
            t.requires_grad_(True)
        P[k] = t
    return P
