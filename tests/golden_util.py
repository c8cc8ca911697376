"""Helpers shared by the golden-vector generator and the parity tests."""
import json
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
NSAMP = 8


def sample_index(n):
    return [int((i * 2654435761 + 12345) % n) for i in range(NSAMP)]


def tensor_summary(t):
    """[l2 norm, sum, 8 fixed-position samples] of a tensor, float64."""
    t = t.detach().double().reshape(-1).cpu()
    idx = sample_index(t.numel())
    return np.concatenate([[float(t.norm()), float(t.sum())], t[idx].numpy()])


def summarize(named):
    """named: iterable of (name, tensor|None) -> (json names, (n, 2+NSAMP) array)."""
    names, rows = [], []
    for k, v in named:
        if v is None:
            continue
        names.append(k)
        rows.append(tensor_summary(v))
    return json.dumps(names), np.stack(rows) if rows else np.zeros((0, 2 + NSAMP))


def load_golden(name):
    path = os.path.join(GOLDEN_DIR, name + '.npz')
    z = np.load(path, allow_pickle=False)
    return {k: z[k] for k in z.files}


def assert_summary_close(names_json, ref_rows, named, rtol, what='', atol_frac=1e-6, skip=(), select=None):
    """Compare gradient/parameter summaries: per tensor, |delta| <= rtol * max(|ref row|).
    Tensors whose reference norm is below atol_frac * (median norm) are mathematically-zero gradients
    (e.g. the key bias under a softmax) and only need to be equally negligible."""
    names = json.loads(str(names_json))
    got = dict(named)
    bad = []
    floor = atol_frac * float(np.median(ref_rows[:, 0])) if len(ref_rows) else 0.0
    for i, k in enumerate(names):
        if k in skip or (select is not None and not select(k)):
            continue
        if ref_rows[i][0] <= floor:
            assert k in got, '%s: missing tensor %s' % (what, k)
            if got[k] is not None and float(got[k].detach().double().norm()) > 100 * floor:
                bad.append((k, float(got[k].detach().double().norm()), 0.0))
            continue
        assert k in got and got[k] is not None, '%s: missing tensor %s' % (what, k)
        row = tensor_summary(got[k])
        ref = ref_rows[i]
        scale = max(float(np.abs(ref[2:]).max()), float(ref[0]) / max(np.sqrt(got[k].numel()), 1.0), 1e-12)
        # samples: relative to the tensor's typical magnitude; norm: relative to itself
        if abs(row[0] - ref[0]) > rtol * max(ref[0], 1e-12) or \
                np.abs(row[2:] - ref[2:]).max() > rtol * max(scale, float(np.abs(ref[2:]).max())):
            bad.append((k, float(abs(row[0] - ref[0]) / max(ref[0], 1e-12)),
                        float(np.abs(row[2:] - ref[2:]).max() / scale)))
    assert not bad, '%s: %d tensors off (name, rel-norm-err, rel-sample-err): %s' % (
        what, len(bad), bad[:6])
    extra = [k for k, v in got.items() if v is not None and k not in names]
    return extra


def to_torch(d, keys=None):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in d.items() if keys is None or k in keys}


def negligible_grad_names(names_json, ref_rows, atol_frac=1e-6):
    """Names whose reference gradient is numerical noise around an exact zero (softmax-invariant biases)."""
    names = json.loads(str(names_json))
    floor = atol_frac * float(np.median(ref_rows[:, 0]))
    return [k for i, k in enumerate(names) if ref_rows[i][0] <= floor]
