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


# --------------------------------------------------------------------------- #
# whole-tensor sketches: a compact fixture that still sees EVERY element
# --------------------------------------------------------------------------- #
SK_CHUNKS, SK_PROJ = 16, 4


def _signs(n, salt):
    """n pseudo-random +-1 values from integer hashing only (murmur3 finaliser of i * 0x9E3779B1 + salt): exact and portable."""
    m = np.uint64(0xFFFFFFFF)
    x = (np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B1) + np.uint64(salt & 0xFFFFFFFF)) & m
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & m
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & m
    x ^= x >> np.uint64(16)
    return np.where((x >> np.uint64(31)) & np.uint64(1), 1.0, -1.0)


def tensor_sketch(t):
    """float64 [SK_CHUNKS, 1 + SK_PROJ]: the flat tensor cut into SK_CHUNKS contiguous chunks; per chunk its l2 norm and SK_PROJ
    random-sign projections.  An error vector e moves a chunk's projection by a N(0, |e_chunk|^2) amount, so any wrong slice of any
    size shows up in the chunk that holds it (10-number summaries only see the sampled positions and the global norm)."""
    x = t.detach().double().reshape(-1).cpu().numpy()
    n = x.size
    out = np.zeros((SK_CHUNKS, 1 + SK_PROJ))
    bounds = [(c * n) // SK_CHUNKS for c in range(SK_CHUNKS + 1)]
    for c in range(SK_CHUNKS):
        seg = x[bounds[c]:bounds[c + 1]]
        if seg.size == 0:
            continue
        out[c, 0] = np.sqrt(np.dot(seg, seg))
        for j in range(SK_PROJ):
            out[c, 1 + j] = np.dot(seg, _signs(seg.size, 7919 * (c * SK_PROJ + j) + 104729))
    return out


def sketch_named(named):
    names, rows = [], []
    for k, v in named:
        if v is None:
            continue
        names.append(k)
        rows.append(tensor_sketch(v))
    return json.dumps(names), np.stack(rows)


def assert_sketch_close(names_json, ref_sk, named, tol, what=''):
    """Every tensor of ``named`` against its reference sketch: per chunk, every projection within tol * max(chunk norm, tensor norm /
    sqrt(chunks)) -- a bound on the l2 error of each chunk relative to the chunk itself, floored at the tensor's typical chunk.
    Tensors whose reference norm is below 1e-6 of the median (exact-zero gradients) only need to be equally negligible."""
    names = json.loads(str(names_json))
    got = dict(named)
    norms = np.sqrt((ref_sk[:, :, 0] ** 2).sum(1))
    floor = 1e-6 * float(np.median(norms))
    bad, worst = [], (None, 0.0)
    for i, k in enumerate(names):
        assert k in got and got[k] is not None, '%s: missing tensor %s' % (what, k)
        sk = tensor_sketch(got[k])
        if norms[i] <= floor:
            if np.sqrt((sk[:, 0] ** 2).sum()) > 100 * floor:
                bad.append((k, float(np.sqrt((sk[:, 0] ** 2).sum())), 0.0))
            continue
        scale = np.maximum(ref_sk[i][:, 0], norms[i] / np.sqrt(SK_CHUNKS))[:, None]
        err = float((np.abs(sk[:, 1:] - ref_sk[i][:, 1:]) / scale).max())
        nerr = float((np.abs(sk[:, 0] - ref_sk[i][:, 0]) / scale[:, 0]).max())
        e = max(err, nerr)
        if e > worst[1]:
            worst = (k, e)
        if e > tol:
            bad.append((k, err, nerr))
    assert not bad, '%s: %d tensors off (name, projection err, chunk-norm err): %s' % (what, len(bad), bad[:6])
    extra = [k for k, v in got.items() if v is not None and k not in names]
    return extra, worst


def sketch_errors(names_json, ref_sk, sk_of):
    """name -> worst chunk error of ``sk_of(i, name)`` (a [SK_CHUNKS, 1 + SK_PROJ] sketch, or None to skip) against ``ref_sk[i]``,
    relative to max(chunk norm, tensor norm / sqrt(chunks)); tensors whose reference norm is an exact zero in disguise are left out."""
    names = json.loads(str(names_json))
    norms = np.sqrt((ref_sk[:, :, 0] ** 2).sum(1))
    floor = 1e-6 * float(np.median(norms))
    out = {}
    for i, k in enumerate(names):
        sk = sk_of(i, k)
        if sk is None or norms[i] <= floor:
            continue
        scale = np.maximum(ref_sk[i][:, 0], norms[i] / np.sqrt(SK_CHUNKS))[:, None]
        out[k] = max(float((np.abs(sk[:, 1:] - ref_sk[i][:, 1:]) / scale).max()), float((np.abs(sk[:, 0] - ref_sk[i][:, 0]) / scale[:, 0]).max()))
    return out


def assert_no_farther_from_truth(g, named, what='', tol=1e-3, factor=3.0, tol_weight_g=3e-2):
    """A fixture with the reference's float32 gradient sketches (``grad_sketch``) AND its float64 run's (``grad_sketch_f64``): every
    tensor of ``named`` must sit no farther from the float64 truth than max(tol, factor x the reference's own float32 error), chunk by
    chunk.  The weight_g scalars of the weight-normed layers get ``tol_weight_g`` instead of ``tol``: <dW, v> / ||v|| is a 10^6-term sum
    with heavy cancellation, so ANY float32-grade arithmetic moves it by percents once 40+ decode steps amplify the rounding -- measured
    at 40 steps on the question-attention W2: 1.5e-2 with the two-term f16 split AND with plain v_mfma_f32 (profiles/r06_grad_error_by_gemm_mode.txt),
    3e-3 for the reference's float32 run, 5e-2 for other scalars of that run.  Returns (worst name, its error, the reference's error there)."""
    got = dict(named)
    truth = g['grad_sketch_f64']
    ref_err = sketch_errors(g['grad_names'], truth, lambda i, k: g['grad_sketch'][i])
    mine = sketch_errors(g['grad_names'], truth, lambda i, k: tensor_sketch(got[k]) if got.get(k) is not None else None)
    missing = [k for k in ref_err if k not in mine]
    assert not missing, '%s: no gradient for %s' % (what, missing[:6])
    tol_of = lambda k: tol_weight_g if k.endswith('weight_g') else tol          # noqa: E731
    bad = [(k, e, ref_err[k]) for k, e in mine.items() if e > max(tol_of(k), factor * ref_err[k])]
    assert not bad, '%s: %d gradients farther from the float64 truth than the reference float32 path (name, err, reference err): %s' % (
        what, len(bad), bad[:6])
    k = max(mine, key=lambda n: mine[n] / max(tol_of(n), factor * ref_err[n]))
    return k, mine[k], ref_err[k]
