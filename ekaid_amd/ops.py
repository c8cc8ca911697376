"""Differentiable operators of the difference-VQA path, every one a pair of calls into the
C ABI of libekaid_hip.so (forward kernels + hand-written backward kernels).

PyTorch supplies device memory (``torch.empty``), the stream and the autograd tape only; no
arithmetic on the path is done by ATen.  Every operator raises if its inputs are not on the
HIP device -- there is no CPU fallback (the CPU restatement is ``oracle/``, test-only).
"""
import ctypes
import math
import os

import torch
from torch.autograd import Function

from . import hip
from .hip import (ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH, GemmDesc, check, ptr, require_cuda,
                  stream)

# --------------------------------------------------------------------------- #
# dropout seeds: counter-based masks, forward and backward share the seed
# --------------------------------------------------------------------------- #
_seed_base = 1238
_seed_counter = 0


GEMM_F32_MFMA, GEMM_BF16_SPLIT, GEMM_F16X2 = 0, 1, 2

# Debug mode: with EKAID_CHECK_FINITE=1 every layer output is checked for Inf / NaN (one small reduction + one host read per
# check: a synchronisation each, like the reference's `assert not x.isnan().any()` after every layer, fc.py:42-48).  Off by
# default; never inside a hipGraph capture.
CHECK_FINITE = os.environ.get('EKAID_CHECK_FINITE', '0') not in ('0', 'off', 'false', '')


def assert_finite(t, what):
    """AssertionError naming the producer if ``t`` holds an Inf / NaN (only when CHECK_FINITE is on)."""
    if not CHECK_FINITE or t is None or not t.is_cuda or not t.dtype.is_floating_point or _defer is not None:
        return
    if torch.cuda.is_current_stream_capturing():
        return
    x = t.detach()
    x = x if x.is_contiguous() else x.contiguous()
    cnt = torch.zeros(1, dtype=torch.int32, device=x.device)
    check(hip.lib().ekaid_nonfinite_count(ptr(x), x.numel(), ptr(cnt), stream()), 'nonfinite_count')
    n = int(cnt.item())
    assert n == 0, '%s: %d non-finite values in a tensor of shape %s' % (what, n, tuple(t.shape))


def set_gemm_mode(mode):
    """Arithmetic of every GEMM on the path (include/ekaid_hip.h: ekaid_set_gemm_mode): GEMM_F32_MFMA, GEMM_BF16_SPLIT
    (exact 3-way bf16 split, six bf16 MFMA products, f32 accumulation) or GEMM_F16X2 (the products whose call site supplies operand
    bounds run as the two-term f16 split, three MFMA products; every other product as GEMM_BF16_SPLIT)."""
    hip.check(hip.lib().ekaid_set_gemm_mode(int(mode)), 'set_gemm_mode')


def get_gemm_mode():
    return int(hip.lib().ekaid_get_gemm_mode())


def manual_seed(seed):
    global _seed_base, _seed_counter
    _seed_base = int(seed)
    _seed_counter = 0


def seed_state():
    """(base, counter) of the dropout seed stream -- part of a resumable checkpoint."""
    return (_seed_base, _seed_counter)


def set_seed_state(state):
    global _seed_base, _seed_counter
    _seed_base, _seed_counter = int(state[0]), int(state[1])


def next_seed():
    global _seed_counter
    _seed_counter += 1
    return (_seed_base * 1000003 + _seed_counter * 7919) & 0xFFFFFFFFFFFFFFFF


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _rows2d(t):
    """2-D view with unit inner stride; returns (tensor, leading dim)."""
    if t.dim() != 2:
        t = t.reshape(-1, t.shape[-1])
    if t.stride(1) != 1 or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
        t = t.contiguous()
    ld = t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0))
    return t, ld


# --------------------------------------------------------------------------- #
# raw GEMM
# --------------------------------------------------------------------------- #
# --------------------------------------------------------------------------- #
# gradient slots: backward kernels write weight gradients straight into the optimizer's flat gradient buffer
# --------------------------------------------------------------------------- #
# With ``p.grad`` pre-pointed at a zeroed flat view, autograd ADDS every incoming weight gradient into it: one ATen add per
# parameter and step (~100 launches).  Instead ``FlatAdam.zero_grad`` leaves ``p.grad = None`` and the backward nodes ask
# ``grad_out(W)`` for their output buffer: a fresh alias of the parameter's slice of the flat buffer, which autograd's
# AccumulateGrad then adopts as ``p.grad`` without copying (it steals a gradient nobody else references).  A second
# contribution to the same parameter in the same step (the weight-norm direction term) is accumulated by its kernel into
# the claimed slot and reported to autograd as "no gradient".  Anything that still arrives outside its slot is copied into
# it by the optimizer's post-accumulate hook (ekaid_amd/dp.py), so the flat layout is always complete.
_grad_slots = {}             # parameter data_ptr -> (parameter, flat gradient buffer, offset, numel, shape)
_grad_claimed = set()


def register_grad_slot(param, gflat, offset):
    _grad_slots[param.data_ptr()] = (param, gflat, offset, param.numel(), tuple(param.shape))


def clear_grad_claims():
    _grad_claimed.clear()
    _wn_pending[:] = []          # (a backward pass that died half way must not leave raw-gradient addresses behind for the next one)


def grad_out(W, zero=False):
    """Output buffer for the gradient of weight ``W`` (same shape): its flat-buffer slot when available (already zero: the
    optimizer clears the flat buffer every step), else a fresh tensor."""
    key = W.data_ptr()
    ent = _grad_slots.get(key)
    if ent is not None and ent[0].grad is None and key not in _grad_claimed and tuple(W.shape) == ent[4]:
        _grad_claimed.add(key)
        return ent[1][ent[2]:ent[2] + ent[3]].view(ent[4])
    return torch.zeros_like(W) if zero else torch.empty_like(W)


def in_grad_slot(t, param):
    """Is ``t`` the flat-buffer slot of ``param`` (the view ``grad_out`` hands out when it claims the slot)?  Only such a buffer outlives
    its backward node: autograd adopts it as ``param.grad`` without copying, so a kernel launched LATER in the pass may still write it."""
    ent = _grad_slots.get(param.data_ptr())
    return ent is not None and param.data_ptr() in _grad_claimed and t.data_ptr() == ent[1].data_ptr() + 4 * ent[2] and t.numel() == ent[3]


def bias_grad_slot(b):
    """(buffer the bias gradient is accumulated into -- the bias' zeroed slot of the flat gradient buffer --, what the backward
    node hands to autograd for it).  The buffer goes to ``gemm(..., a_rowsum=)`` of the layer's weight-gradient product, whose
    A operand is the very dY whose column sums the bias gradient is: dY is not read a second time."""
    slot = grad_claimed_slot(b)
    out = slot if slot is not None else grad_out(b, zero=True)
    return out, (None if slot is not None else out)


def zeros_pool(device, *shapes):
    """Several zero-initialised f32 tensors from ONE allocation and ONE fill launch."""
    sizes = [int(math.prod(sh)) for sh in shapes]
    offs, tot = [], 0
    for n in sizes:
        offs.append(tot)
        tot += (n + 3) & ~3                        # keep every tensor 16-byte aligned
    flat = torch.zeros(tot, device=device, dtype=torch.float32)
    return [flat[o:o + n].view(sh) for o, n, sh in zip(offs, sizes, shapes)]


def grad_claimed_slot(W):
    """The slot of ``W`` if an earlier backward node of this step already wrote its gradient there (then a further
    contribution is accumulated in place and returned to autograd as None), else None."""
    key = W.data_ptr()
    ent = _grad_slots.get(key)
    if ent is not None and key in _grad_claimed and tuple(W.shape) == ent[4]:
        return ent[1][ent[2]:ent[2] + ent[3]].view(ent[4])
    return None


_BIAS_IN_WGRAD = os.environ.get('EKAID_BIAS_IN_WGRAD', '1') != '0'
_SEQ_IN_C = os.environ.get('EKAID_SEQ_IN_C', '1') != '0'      # time loops of the GRU / decoder behind ONE C-ABI call each


# Column maxima (and the row maxima of matrices wider than one 256-column slice) are accumulated with atomics into ZEROED outputs.
# Instead of one memset node per output, the outputs of a step are slices of one pool that is cleared once, where the step begins
# (``bounds_pool_reset``: the detector's forward).  A slice is handed out once between two resets; an exhausted pool, or a caller that
# never resets, falls back to fresh memory that the C side clears per pass.
_BOUNDS_POOL_FLOATS = 1 << 21          # 8 MB: a 256-pair step takes ~0.8 M row + 0.1 M column maxima
_bounds_pool = {}            # device -> [zeroed buffer, next free element, generation]


def bounds_pool_reset(device):
    """Clear the pool of column-maxima outputs (one memset node) and start handing out its slices from the top.  Bounds taken before
    a reset are STALE afterwards (their memory is cleared and handed out again): whoever keeps bounds across other work -- a forward
    pass keeps them for its backward pass -- remembers ``bounds_generation`` and takes them afresh if it has moved on (two forward passes
    before the first backward: the second one's reset must not leave the first with the second's maxima)."""
    ent = _bounds_pool.get(device)
    if ent is None:
        _bounds_pool[device] = [torch.zeros(_BOUNDS_POOL_FLOATS, device=device, dtype=torch.float32), 0, 1]
    else:
        ent[0].zero_()
        ent[1] = 0
        ent[2] += 1
    del _xb_cache[:]


def bounds_generation(device):
    ent = _bounds_pool.get(device)
    return ent[2] if ent is not None else 0


_AMAX_SLICE = 256            # columns per workgroup of ekaid_amax_rc (pointwise.hip: AMAX_SLICE)


def _zeroed_cols(device, n):
    """(n zeroed floats, came from the pool)"""
    ent = _bounds_pool.get(device)
    n4 = (n + 3) & ~3
    if ent is not None and ent[1] + n4 <= _BOUNDS_POOL_FLOATS:
        out = ent[0][ent[1]:ent[1] + n]
        ent[1] += n4
        return out, True
    return torch.empty(n, device=device, dtype=torch.float32), False


def amax_rc(x, rows, cols, ld, off=0, want_rows=True, want_cols=True):
    """(rowmax [rows], colmax [cols]) of |x[off + r * ld + c]| on the device (either None when not wanted): the operand bounds of the
    two-term f16 GEMM split (include/ekaid_hip.h: ekaid_amax_rc).  None, None when the matrix is not made of whole 16-byte vectors --
    the product then stays on the exact split."""
    if (cols & 3) or (ld & 3) or ((x.data_ptr() + 4 * off) & 15) or rows < 1 or cols < 1:
        return None, None
    if not want_rows:
        rm, rpooled = None, True
    elif cols > _AMAX_SLICE:               # several column slices meet in a row's maximum: an atomic target like the column maxima
        rm, rpooled = _zeroed_cols(x.device, rows)
    else:
        rm, rpooled = torch.empty(rows, device=x.device, dtype=torch.float32), True
    cm, cpooled = _zeroed_cols(x.device, cols) if want_cols else (None, True)
    check(hip.lib().ekaid_amax_rc(x.data_ptr() + 4 * off, rows, cols, ld, ptr(rm), ptr(cm), int(rpooled and cpooled), stream()), 'amax_rc')
    return rm, cm


# M * N * K below which a layer's products are not worth the bounds passes they need (each pass = a memset node + a kernel: ~10 us even
# for a small operand).  Measured on the replayed 128 / 64 / 32-pair steps (profiles/r06_ladder_f16x2_threshold.txt): with every product
# >= 2^28 on the two-term split the 2304-row layers of a 32-pair step lose more to their ~70 extra graph nodes than their products gain.
F16X2_MIN_WORK = int(float(os.environ.get('EKAID_F16X2_MIN_WORK', 6e9)))
# the same for a block of products that SHARE their passes (the decoder's fifteen time-batched weight gradients read one matrix: one pass)
F16X2_MIN_WORK_SHARED = int(float(os.environ.get('EKAID_F16X2_MIN_WORK_SHARED', 1.5e9)))


def f16x2_wanted(M, N, K, min_work=None):
    """Should the call site of an M x N x K product compute operand bounds for the two-term f16 split?  (Mode 2 only; small products
    stay on the exact split: a bounds pass is a launch of its own.)"""
    return get_gemm_mode() == GEMM_F16X2 and M * N * K >= (F16X2_MIN_WORK if min_work is None else min_work) and min(M, N, K) >= 32


_wb_cache = {}           # weight data_ptr -> (row maxima [N], column maxima [K]) for the current step (weight_bounds_prepare / _release)


def weight_bounds_prepare(weights):
    """Row / column maxima of every 2-D weight in ``weights`` in ONE launch per 32 (the layers pick them up through ``weight_bounds``);
    valid until ``weight_bounds_release`` -- the weights do not change inside a step."""
    _wb_cache.clear()
    if get_gemm_mode() != GEMM_F16X2:
        return
    ws = [w for w in weights if w.dim() == 2 and w.is_contiguous() and not (w.shape[1] & 3) and not (w.data_ptr() & 15) and w.numel() >= (1 << 14)]
    L = hip.lib()
    for i0 in range(0, len(ws), 32):
        chunk = ws[i0:i0 + 32]
        n = len(chunk)
        dev = chunk[0].device
        tot_r, tot_c = sum(w.shape[0] for w in chunk), sum((w.shape[1] + 3) & ~3 for w in chunk)
        buf, rpooled = _zeroed_cols(dev, tot_r)
        cbuf, pooled = _zeroed_cols(dev, tot_c)
        pooled = pooled and rpooled
        rms, cms, o = [], [], 0
        for w in chunk:
            rms.append(buf[o:o + w.shape[0]])
            o += w.shape[0]
        o = 0
        for w in chunk:
            cms.append(cbuf[o:o + w.shape[1]])
            o += (w.shape[1] + 3) & ~3
        xs = (ctypes.c_void_p * n)(*[w.data_ptr() for w in chunk])
        rows = (ctypes.c_longlong * n)(*[w.shape[0] for w in chunk])
        cols = (ctypes.c_int * n)(*[w.shape[1] for w in chunk])
        rp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in rms])
        cp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in cms])
        check(L.ekaid_amax_rc_multi(xs, rows, cols, cols, rp, cp, n, int(pooled), stream()), 'amax_rc_multi')
        for w, r, c in zip(chunk, rms, cms):
            _wb_cache[w.data_ptr()] = (r, c)


def weight_bounds_release():
    _wb_cache.clear()
    del _xb_cache[:]


def weight_bounds(W):
    """(row maxima, column maxima) of the 2-D weight ``W``: the prepared ones, else one launch."""
    ent = _wb_cache.get(W.data_ptr())
    if ent is not None and ent[0].numel() == W.shape[0] and ent[1].numel() == W.shape[1]:
        return ent
    return amax_rc(W, W.shape[0], W.shape[1], W.shape[1])


_xb_cache = []           # [(tensor, (M, K, ld), bounds, pool generation)]: the bounds of the last few layer INPUTS of this forward pass


def input_bounds(x, M, K, ld):
    """``amax_rc(x, M, K, ld)`` for a layer input, taken once when several layers read the same tensor (the encoder output feeds context2,
    gate2 and the region embedding: three passes over 75 MB at 256 pairs, two of them saved).  An entry keeps its tensor alive, so a hit
    on (address, shape) is a hit on the same memory, not on a recycled block; eight entries, dropped with the pool they point into."""
    gen = bounds_generation(x.device)
    for t, shape, b, g in _xb_cache:
        if g == gen and shape == (M, K, ld) and t.data_ptr() == x.data_ptr():
            return b
    b = amax_rc(x, M, K, ld)
    if b[0] is not None:
        _xb_cache.append((x, (M, K, ld), b, gen))
        del _xb_cache[:-8]
    return b


def seq_in_c():
    """The sequence-level entry points are the default; the step-by-step Python loops (the same C-ABI calls, one by one) stay
    for the per-launch GEMM profile of bench.py and as the cross-check of tests (EKAID_SEQ_IN_C=0)."""
    return _SEQ_IN_C and _gemm_profile is None


_PAIR = os.environ.get('EKAID_GEMM_PAIR', '1') != '0'      # A/B knob: 0 = every product its own launch
_defer = None            # while a list: gemm() / deferred_call() record into it instead of launching (LinearPairFn)


def _fill_desc(A, B, C, M, N, K, lda, ldb, ldc, transA, transB, alpha=1.0, alpha_dev=None,
               accumulate=False, bias=None, R=None, ldr=0, rdiv=1, rowmask=None, act=ACT_NONE,
               batch=(1, 1), strides=(0, 0, 0, 0, 0, 0), a_off=0, b_off=0, c_off=0, drop_p=0.0, drop_seed=0, a_rowsum=None,
               post=None, ldpost=0, a_amax=None, b_amax=None):
    d = GemmDesc()
    d.A = A.data_ptr() + 4 * a_off
    d.B = B.data_ptr() + 4 * b_off
    d.C = C.data_ptr() + 4 * c_off
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc = lda, ldb, ldc
    d.transA, d.transB = int(transA), int(transB)
    d.batch1, d.batch2 = batch
    (d.strideA1, d.strideA2, d.strideB1, d.strideB2, d.strideC1, d.strideC2) = strides
    d.alpha = alpha
    d.alpha_dev = ptr(alpha_dev)
    d.accumulate = int(accumulate)
    d.bias = ptr(bias)
    d.R = ptr(R)
    d.ldr = ldr
    d.rdiv = rdiv
    d.rowmask = ptr(rowmask)
    d.act = act
    d.drop_p = drop_p
    d.drop_seed = drop_seed
    d.a_rowsum = ptr(a_rowsum) if _BIAS_IN_WGRAD else None
    d.post = ptr(post)
    d.ldpost = ldpost
    # operand bounds of the two-term f16 split: (vector, element offset) or a vector -- the row maxima of op(A), the column maxima of op(B)
    for name, v in (('a_amax', a_amax), ('b_amax', b_amax)):
        if v is not None:
            t, off = v if isinstance(v, tuple) else (v, 0)
            setattr(d, name, t.data_ptr() + 4 * off)
    return d


def gemm(A, B, C, M, N, K, lda, ldb, ldc, transA, transB, **kw):
    """C = dropout(act(alpha * op(A) op(B) [+C] [+bias] [+R])) [+post] on raw device buffers (element offsets a_off..); ``a_rowsum``
    (transA only): a_rowsum[m] += sum_k op(A)[m][k], the bias gradient riding on a weight-gradient product.  Keyword arguments:
    alpha, alpha_dev, accumulate, bias, R, ldr, rdiv, rowmask, act, batch, strides, a_off, b_off, c_off, drop_p, drop_seed, a_rowsum,
    post, ldpost (``_fill_desc``)."""
    if _defer is not None:
        _defer.append(('gemm', (A, B, C, M, N, K, lda, ldb, ldc, transA, transB), kw))
        return
    d = _fill_desc(A, B, C, M, N, K, lda, ldb, ldc, transA, transB, **kw)
    batch = kw.get('batch', (1, 1))
    a_rowsum = kw.get('a_rowsum')
    L = hip.lib()
    need = L.ekaid_gemm_workspace_bytes(ctypes.byref(d))
    ws = hip.workspace(need, C.device) if need else None
    prof = _gemm_profile
    if prof is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
    check(L.ekaid_gemm_f32(ctypes.byref(d), ptr(ws), need, stream()), 'ekaid_gemm_f32')
    if prof is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        nb = batch[0] * batch[1]
        tile, sk = ctypes.c_int(0), ctypes.c_int(1)
        check(L.ekaid_gemm_plan_query(ctypes.byref(d), ctypes.byref(tile), ctypes.byref(sk)), 'gemm_plan_query')
        prof.append((e0, e1, 2.0 * M * N * K * nb, (int(transA), int(transB), M, N, K, nb, tile.value, _executed_per_product(kw, nb))))
    if a_rowsum is not None and not _BIAS_IN_WGRAD:          # A/B knob: the column sums as their own launches
        check(L.ekaid_colsum_acc(d.A, K, M, lda, ptr(a_rowsum), stream()), 'colsum_acc')


def _executed_per_product(kw, nb):
    """Matrix-pipe products the library executes per algorithmic product for a launch with these keyword arguments: 1 on the f32 pipe,
    6 for the exact 3-way bf16 split, 3 for the two-term f16 split (mode 2, both operand bounds given, unbatched: csrc/gemm.hip use_x2)."""
    mode = get_gemm_mode()
    if mode == GEMM_F32_MFMA:
        return 1
    return 3 if (mode == GEMM_F16X2 and nb == 1 and kw.get('a_amax') is not None and kw.get('b_amax') is not None) else 6


def gemm_pair(g0, g1):
    """Two INDEPENDENT products (``g`` = (positional arguments of ``gemm``, keyword arguments)) in one launch where the library's pair
    kernel takes them (include/ekaid_hip.h: ekaid_gemm_pair_f32; results bit-identical to two ``gemm`` calls), else two launches."""
    if not _PAIR or not _BIAS_IN_WGRAD:
        gemm(*g0[0], **g0[1])
        gemm(*g1[0], **g1[1])
        return
    d0, d1 = _fill_desc(*g0[0], **g0[1]), _fill_desc(*g1[0], **g1[1])
    L = hip.lib()
    need = L.ekaid_gemm_pair_workspace_bytes(ctypes.byref(d0), ctypes.byref(d1))
    ws = hip.workspace(need, g0[0][2].device) if need else None
    e0 = _prof_begin()
    check(L.ekaid_gemm_pair_f32(ctypes.byref(d0), ctypes.byref(d1), ptr(ws), need, stream()), 'ekaid_gemm_pair_f32')
    if e0 is not None:
        tile, sk = ctypes.c_int(0), ctypes.c_int(1)
        check(L.ekaid_gemm_plan_query(ctypes.byref(d0), ctypes.byref(tile), ctypes.byref(sk)), 'gemm_plan_query')
        (_, _, _, M0, N0, K0, _, _, _, tA, tB), (_, _, _, M1, N1, K1) = g0[0], g1[0][:6]
        both = dict(a_amax=g0[1].get('a_amax') if g1[1].get('a_amax') is not None else None,
                    b_amax=g0[1].get('b_amax') if g1[1].get('b_amax') is not None else None)
        _prof_end(e0, 2.0 * (M0 * N0 * K0 + M1 * N1 * K1), (int(tA), int(tB), M0, N0 + N1, K0, 1, tile.value, _executed_per_product(both, 1)))


def deferred_call(fn):
    """Run ``fn()`` now, or -- inside a deferral window -- after the products recorded before it have been launched."""
    if _defer is not None:
        _defer.append(('call', fn))
    else:
        fn()


class _Deferring:
    def __init__(self, queue):
        self.queue = queue

    def __enter__(self):
        global _defer
        assert _defer is None, 'deferral windows do not nest'
        _defer = self.queue

    def __exit__(self, *exc):
        global _defer
        _defer = None


def _span(t):
    """[first byte, past-the-end byte) a tensor may touch (whole remaining allocation for a non-contiguous view)."""
    n = t.numel() if t.is_contiguous() else t.untyped_storage().nbytes() // t.element_size() - t.storage_offset()
    return t.data_ptr(), t.data_ptr() + n * t.element_size()


def _independent(g0, g1):
    """May the two recorded products run in one launch?  Neither reads (operands, bias, residual, post) what the other writes, and
    their outputs do not overlap -- judged on ADDRESS RANGES (gradient slots of different parameters are views of one flat buffer),
    with one interleaved case allowed: disjoint column blocks of one row-major matrix (the weight-gradient blocks of a multi-input
    layer, written with c_off)."""
    def out_span(g):
        C, M, N, ldc = g[0][2], g[0][3], g[0][4], g[0][8]
        lo = C.data_ptr() + 4 * g[1].get('c_off', 0)
        return lo, lo + 4 * ((M - 1) * ldc + N)

    def overlap(x, y):
        return x[0] < y[1] and y[0] < x[1]
    if g0[1].get('batch', (1, 1)) != (1, 1) or g1[1].get('batch', (1, 1)) != (1, 1):
        return False                                  # batched products: out_span() does not follow their strides (never paired anyway)
    s0, s1 = out_span(g0), out_span(g1)
    if overlap(s0, s1):
        ld0, ld1, n0, n1 = g0[0][8], g1[0][8], g0[0][4], g1[0][4]
        d = (s1[0] - s0[0]) // 4                      # column offset of product 1's block relative to product 0's
        if ld0 != ld1 or g0[0][3] != g1[0][3] or not ((0 < d < ld0 and n0 <= d and d + n1 <= ld0) or (-ld0 < d < 0 and n1 <= -d and -d + n0 <= ld0)):
            return False
    r0, r1 = g0[1].get('a_rowsum'), g1[1].get('a_rowsum')
    if r0 is not None and r1 is not None and overlap(_span(r0), _span(r1)):
        return False                                  # both would add into the same bias-gradient slot
    for ga, sb in ((g0, s1), (g1, s0)):
        kw = ga[1]
        for t in (ga[0][0], ga[0][1], kw.get('post'), kw.get('R'), kw.get('bias'), kw.get('rowmask'), kw.get('alpha_dev')):
            if t is not None and overlap(_span(t), sb):
                return False
    return True


def flush_pairs(qa, qb):
    """Launch what two deferral windows recorded: the i-th product of one with the i-th product of the other as a pair where they are
    independent, everything else in its recorded order (calls after the products recorded before them).  Each queue keeps its own
    order and entry i of ``qa`` goes out before entry i of ``qb``; the contract of the caller is that no entry of one queue depends on a
    LATER entry of the other (two layers that do not feed each other: LinearPairFn) -- ``_independent`` only decides whether two
    entries may share a launch."""
    if not _PAIR:
        for q in (qa, qb):
            for ent in q:
                gemm(*ent[1], **ent[2]) if ent[0] == 'gemm' else ent[1]()
        return
    ia = ib = 0
    while ia < len(qa) or ib < len(qb):
        while ia < len(qa) and qa[ia][0] == 'call':
            qa[ia][1]()
            ia += 1
        while ib < len(qb) and qb[ib][0] == 'call':
            qb[ib][1]()
            ib += 1
        ga = qa[ia] if ia < len(qa) else None
        gb = qb[ib] if ib < len(qb) else None
        if ga is not None and gb is not None:
            if _independent(ga[1:], gb[1:]):
                gemm_pair(ga[1:], gb[1:])
            else:
                gemm(*ga[1], **ga[2])
                gemm(*gb[1], **gb[2])
            ia += 1
            ib += 1
        elif ga is not None:
            gemm(*ga[1], **ga[2])
            ia += 1
        elif gb is not None:
            gemm(*gb[1], **gb[2])
            ib += 1


# --------------------------------------------------------------------------- #
# raw partial products ("slabs") for consumers that finish the reduction themselves (include/ekaid_hip.h: ekaid_gemm_slabs)
# --------------------------------------------------------------------------- #
class SlabSet:
    """Slabs [n][M][N] of one product (``buf``), addressed by consumers through ``cols(c0)`` = the ``ekaid_slabs_t`` of the
    column block starting at column c0."""
    __slots__ = ('buf', 'n', 'M', 'N')

    def __init__(self, buf, n, M, N):
        self.buf, self.n, self.M, self.N = buf, n, M, N

    def cols(self, c0=0):
        return hip.Slabs(self.buf.data_ptr() + 4 * c0, self.n, self.M * self.N, self.N)

    def total(self):
        """The reduced product as a tensor (tests)."""
        return self.buf.view(self.n, self.M, self.N).sum(0)


NO_SLABS = hip.Slabs(None, 0, 0, 0)


def _slab_desc(A, B, M, N, K, lda, ldb, transA, transB, a_off, b_off):
    d = GemmDesc()
    d.A = A.data_ptr() + 4 * a_off
    d.B = B.data_ptr() + 4 * b_off
    d.C = None
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc = lda, ldb, N
    d.transA, d.transB = int(transA), int(transB)
    d.batch1 = d.batch2 = 1
    d.alpha = 1.0
    d.rdiv = 1
    return d


def gemm_slabs(A, B, M, N, K, lda, ldb, transA, transB, a_off=0, b_off=0):
    """op(A) . op(B) as raw K-slabs (no epilogue): returns a ``SlabSet``."""
    d = _slab_desc(A, B, M, N, K, lda, ldb, transA, transB, a_off, b_off)
    L = hip.lib()
    n = L.ekaid_gemm_slab_count(ctypes.byref(d))
    if n < 1:
        check(n, 'ekaid_gemm_slab_count')
    buf = torch.empty(n * M * N, device=A.device, dtype=torch.float32)
    e0 = _prof_begin()
    check(L.ekaid_gemm_slabs(ctypes.byref(d), ptr(buf), buf.numel() * 4, stream()), 'ekaid_gemm_slabs')
    _prof_end(e0, 2.0 * M * N * K, (int(transA), int(transB), M, N, K, 1, 1, _executed_per_product({}, 1)))
    return SlabSet(buf, n, M, N)


def gemm_slabs2(g0, g1):
    """Two independent products in one launch; g = (A, B, M, N, K, lda, ldb, transA, transB, a_off, b_off)."""
    L = hip.lib()
    ds, bufs, outs = [], [], []
    for g in (g0, g1):
        d = _slab_desc(*g)
        n = L.ekaid_gemm_slab_count(ctypes.byref(d))
        if n < 1:
            check(n, 'ekaid_gemm_slab_count')
        buf = torch.empty(n * g[2] * g[3], device=g[0].device, dtype=torch.float32)
        ds.append(d)
        bufs.append(buf)
        outs.append(SlabSet(buf, n, g[2], g[3]))
    e0 = _prof_begin()
    check(L.ekaid_gemm_slabs2(ctypes.byref(ds[0]), ptr(bufs[0]), bufs[0].numel() * 4, ctypes.byref(ds[1]), ptr(bufs[1]),
                              bufs[1].numel() * 4, stream()), 'ekaid_gemm_slabs2')
    _prof_end(e0, 2.0 * (g0[2] * g0[3] * g0[4] + g1[2] * g1[3] * g1[4]),
              (int(g0[7]), int(g0[8]), g0[2], g0[3] + g1[3], g0[4], 2, 1, _executed_per_product({}, 1)))
    return outs[0], outs[1]


# --------------------------------------------------------------------------- #
# backward through weight_norm for all layers of a step at once
# --------------------------------------------------------------------------- #
# A weight-normed layer's backward leaves its RAW weight gradient dW = dy^T x in the gradient slot of weight_v and registers the layer
# here; ONE pair of launches (ekaid_wn_grad_multi) turns every registered dW into dv and fills the dg slots (two launches per layer
# before: 30 per step).  When: at the end of the backward pass (an autograd engine callback), or earlier -- right before a gradient
# bucket leaves for its all-reduce (dp.GradBuckets._launch flushes what is registered so far: every layer of a complete bucket is).  So
# the transform always precedes the reduction, in every launch mode, and data-parallel runs stay bit-identical across modes.
_wn_pending = []                 # [(dW pointer, v, g, norm, scale, dg pointer, elements)] -- the two gradients by ADDRESS only: a second
                                 # reference to a gradient tensor makes autograd's AccumulateGrad clone it instead of adopting it
_WN_MULTI = os.environ.get('EKAID_WN_GRAD_MULTI', '1') != '0'      # A/B knob: 0 = two launches per layer inside its backward node


def wn_grad_register(dW, v, g, norm, scale, dg):
    first = not _wn_pending
    _wn_pending.append((dW.data_ptr(), v, g, norm, scale, dg.data_ptr(), dW.numel()))
    if first:
        # (one callback per batch of registrations; a flush in between -- a bucket leaving -- empties the list, and the next layer queues
        # another.  A callback that finds nothing registered launches nothing, so a pass that died half way leaves no state behind.)
        torch.autograd.Variable._execution_engine.queue_callback(wn_grad_flush)


def wn_grad_flush():
    """Transform every registered raw weight gradient now (nothing registered: no launch)."""
    mine = list(_wn_pending)
    _wn_pending[:] = []
    L = hip.lib()
    for i0 in range(0, len(mine), 32):
        chunk = mine[i0:i0 + 32]
        n = len(chunk)
        arr = lambda k: (ctypes.c_void_p * n)(*[e[k] if isinstance(e[k], int) else e[k].data_ptr() for e in chunk])        # noqa: E731
        ns = (ctypes.c_longlong * n)(*[e[6] for e in chunk])
        check(L.ekaid_wn_grad_multi(arr(0), arr(1), arr(2), arr(3), arr(4), ns, arr(5), n, stream()), 'wn_grad_multi')


class pairing_window:
    """``with ops.pairing_window():`` -- the products issued inside are recorded and, on exit, launched in their order with every two
    CONSECUTIVE independent ones sharing a launch (ekaid_gemm_pair_f32 decides whether the pair kernel takes them); ``deferred_call``
    bodies run after all of them.  For blocks of mutually independent products: the time-batched weight gradients of the decoder and of
    the question GRU."""

    def __enter__(self):
        self.q = []
        self.win = _Deferring(self.q)
        self.win.__enter__()
        return self

    def __exit__(self, et, ev, tb):
        self.win.__exit__(et, ev, tb)
        if et is not None:
            return False
        gs = [(e[1], e[2]) for e in self.q if e[0] == 'gemm']
        _launch_group(gs)
        for e in self.q:
            if e[0] == 'call':
                e[1]()
        return False


def _launch_group(gs):
    """The recorded products of one backward step of a layer (one per input): two of them go out as a pair (the column blocks they
    read and write are disjoint), the rest one by one."""
    i = 0
    while i < len(gs):
        if i + 1 < len(gs) and _defer is None and _independent(gs[i], gs[i + 1]):
            gemm_pair(gs[i], gs[i + 1])
            i += 2
        else:
            gemm(*gs[i][0], **gs[i][1])
            i += 1


def slab_array(*srcs):
    """ctypes array of ``ekaid_slabs_t`` + its length."""
    arr = (hip.Slabs * max(len(srcs), 1))(*srcs)
    return arr, len(srcs)


_gemm_profile = None


def _prof_begin():
    if _gemm_profile is None:
        return None
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    return e0


def _prof_end(e0, flops, shape):
    if e0 is not None and _gemm_profile is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        _gemm_profile.append((e0, e1, flops, shape))


def gemm_profile_begin():
    """Start collecting (HIP event pair, FLOPs, shape) for every GEMM launch on the current stream."""
    global _gemm_profile
    _gemm_profile = []


def gemm_profile_end():
    """Stop collecting; returns [(milliseconds, flops, (transA, transB, M, N, K, batch, tile, executed))] (synchronises); tile = the kernel
    instantiation the planner chose (ekaid_gemm_plan_query: 2 = 128x128 tiles, 1 = 64x64, 3 / 4 = 128x64 / 64x128, 0 = tall-skinny);
    executed = matrix-pipe products per algorithmic product (1 f32 pipe, 6 exact bf16 split, 3 two-term f16 split)."""
    global _gemm_profile
    rec, _gemm_profile = _gemm_profile, None
    torch.cuda.synchronize()
    return [(e0.elapsed_time(e1), fl, shape) for e0, e1, fl, shape in rec]


# --------------------------------------------------------------------------- #
# linear layers:  y = act( scale * sum_i x_i W[:, K_i]^T + b )
# --------------------------------------------------------------------------- #
TRANSPOSE_MIN_ROWS = int(os.environ.get('EKAID_TRANSPOSE_MIN_ROWS', 4096))     # f32-MFMA arithmetic only: forward GEMMs with at least this many
                                                                            # rows read a transposed copy of the weight (n-contiguous staging was
                                                                            # 10 % faster there; the bf16-split kernel is indifferent: measured)


def _use_transposed(M):
    # (bf16-split arithmetic: re-measured in round 3 with the plain K loop for forward products -- a transposed copy of the
    # weight costs more than the n-contiguous staging returns: 31.51 vs 31.30 ms per step)
    return M >= TRANSPOSE_MIN_ROWS and get_gemm_mode() == GEMM_F32_MFMA


def weight_transposed(W):
    """W [N][K] -> fresh W^T [K][N] (one small kernel; weights change every step, so no caching)."""
    N, K = W.shape
    Wt = torch.empty(K, N, device=W.device, dtype=torch.float32)
    check(hip.lib().ekaid_transpose(ptr(W), N, K, K, ptr(Wt), N, stream()), 'transpose')
    return Wt


class LinearFn(Function):
    """nn.Linear / weight-normed FCNet layer over one or several inputs whose feature axes are
    concatenated (so ``torch.cat`` of the reference never materialises).
    ref: fc.py:15-49, modules.py:195,278-303, dynamic_speaker_change_pos.py:98-125."""

    @staticmethod
    def forward(ctx, W, b, scale, act, drops, *xs):
        """``drops``: None, or one dropout probability per input (0 = none): the FCNet's input dropout (fc.py:36-37)
        applied here so that its backward is the epilogue of the dgrad GEMM instead of a pass of its own."""
        require_cuda(W, *xs)
        ctx.wn = None
        N, Ktot = W.shape
        W = _c(W)
        xs2, lds = zip(*[_rows2d(x) for x in xs])
        xs2, lds = list(xs2), list(lds)
        seeds = [0] * len(xs2)
        if drops is not None:
            for i, pd in enumerate(drops):
                if pd > 0.0:
                    pre = getattr(xs[i], '_ekaid_predrop', None)
                    if pre is not None and pre[0] == float(pd) and pre[2].numel() == xs2[i].numel():
                        seeds[i], xd = pre[1], pre[2].view(xs2[i].shape)       # the producer's epilogue wrote the dropped input
                    else:
                        x = xs2[i] if lds[i] == xs2[i].shape[1] else xs2[i].contiguous()    # mask index = flat index
                        xd = torch.empty_like(x)
                        seeds[i] = next_seed()
                        check(hip.lib().ekaid_dropout(ptr(x), ptr(xd), x.numel(), float(pd), seeds[i], stream()), 'dropout')
                    xs2[i], lds[i] = xd, xd.shape[1]
        ctx.drops, ctx.seeds = drops, seeds
        ctx.slots = [slot_of(x) for x in xs]
        M = xs2[0].shape[0]
        assert sum(x.shape[1] for x in xs2) == Ktot, 'input widths do not add up to in_features'
        y = torch.empty(M, N, device=W.device, dtype=torch.float32)
        Wt = weight_transposed(W) if _use_transposed(M) else None
        # two-term f16 split (mode 2): row / column maxima of every operand, taken once here and reused by the backward products
        x2 = f16x2_wanted(M, N, min(x.shape[1] for x in xs2))
        ctx.wb = weight_bounds(W) if x2 else (None, None)
        ctx.xb = [input_bounds(x, M, x.shape[1], ld) if x2 else (None, None) for x, ld in zip(xs2, lds)]
        ctx.bgen = bounds_generation(W.device)
        off = 0
        for i, (x, ld) in enumerate(zip(xs2, lds)):
            last = i == len(xs2) - 1
            K = x.shape[1]
            if Wt is not None:      # B = W^T [K][N]: n-contiguous staging (512-byte row segments), the faster variant
                gemm(x, Wt, y, M, N, K, ld, N, N, 0, 0, alpha_dev=scale, accumulate=i > 0,
                     bias=b if last else None, act=act if last else ACT_NONE, b_off=off * N)
            else:
                gemm(x, W, y, M, N, K, ld, Ktot, N, 0, 1, alpha_dev=scale, accumulate=i > 0,
                     bias=b if last else None, act=act if last else ACT_NONE, b_off=off, a_amax=ctx.xb[i][0], b_amax=ctx.wb[0])
            off += K
        ctx.act = act
        ctx.has_b = b is not None
        ctx.bias_ref = b.detach() if b is not None else None       # (an alias: its data_ptr finds the gradient slot)
        ctx.has_scale = scale is not None
        ctx.lead = [x.shape[:-1] for x in xs]
        ctx.save_for_backward(W, scale, y if act != ACT_NONE else None, *xs2)
        ctx.lds = lds
        assert_finite(y, 'linear layer (%d -> %d)' % (Ktot, N))
        return y.view(*xs[0].shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        W, scale, y, *xs2 = ctx.saved_tensors
        N, Ktot = W.shape
        M = xs2[0].shape[0]
        dy = _c(dy).view(M, N)
        L = hip.lib()
        if ctx.act != ACT_NONE:
            dz = torch.empty_like(dy)
            check(L.ekaid_act_bwd(ptr(dy), ptr(y), ptr(dz), dy.numel(), ctx.act, stream()), 'act_bwd')
        else:
            dz = dy
        need = ctx.needs_input_grad
        # (mode 2: the bounds of dz serve both gradient products -- its row maxima the input gradient, its column maxima the weight gradient)
        if ctx.wb[0] is not None and ctx.bgen != bounds_generation(W.device):          # another forward pass has recycled the pool since: afresh
            ctx.wb = amax_rc(W, N, Ktot, Ktot)
            ctx.xb = [amax_rc(x, M, x.shape[1], ld) for x, ld in zip(xs2, ctx.lds)]
        dzb = amax_rc(dz, M, N, N) if ctx.wb[0] is not None else (None, None)
        db = db_buf = None
        if ctx.has_b and need[1]:
            db_buf, db = bias_grad_slot(ctx.bias_ref)          # filled by the first weight-gradient product below
        dxs = []
        off = 0
        dgrads = []                                  # recorded input-gradient products: two independent ones share a launch
        for i, x in enumerate(xs2):
            K = x.shape[1]
            if need[5 + i]:
                dx = torch.empty(M, K, device=W.device, dtype=torch.float32)
                pd = ctx.drops[i] if ctx.drops is not None else 0.0
                slot = ctx.slots[i]
                post = None                          # another reader's gradient of the same tensor joins behind the dropout (GradSlot)
                if slot is not None and slot.lent is not None and slot.lent.numel() == dx.numel():
                    post, slot.lent = slot.lent, None
                dgrads.append(((dz, W, dx, M, K, N, N, Ktot, K, 0, 0), dict(alpha_dev=scale, b_off=off, drop_p=float(pd), drop_seed=ctx.seeds[i],
                                                                            post=post, ldpost=K, a_amax=dzb[0],
                                                                            b_amax=(ctx.wb[1], off) if ctx.wb[1] is not None else None)))          # input-dropout backward in the epilogue
                dxs.append(lend_grad(slot, dx.view(*ctx.lead[i], K)))
            else:
                dxs.append(None)
            off += K
        _launch_group(dgrads)
        dW = dscale = None
        if need[0] or (ctx.has_scale and need[2]):
            dW = grad_out(W)
            off = 0
            wgrads = []
            for (x, ld), xb in zip(zip(xs2, ctx.lds), ctx.xb):
                K = x.shape[1]
                wgrads.append(((dz, x, dW, N, K, M, N, ld, Ktot, 1, 0), dict(c_off=off, a_rowsum=db_buf, a_amax=dzb[1], b_amax=xb[1])))
                db_buf = None
                off += K
            _launch_group(wgrads)
            if ctx.has_scale and ctx.wn is not None:
                # weight-normed layer (WNLinearFn): the unscaled wgrad becomes dv in place and dg comes out of the same two
                # launches (was: dot + scale, then a third pass for the norm's own backward)
                g, norm = ctx.wn
                dg = grad_out(g)
                if _WN_MULTI and dW.is_contiguous() and in_grad_slot(dW, W) and in_grad_slot(dg, g):
                    # all layers of the step together, at the end of backward -- ONLY for gradients that live in their flat-buffer
                    # slots: a temporary (p.grad already set: a second backward without zero_grad, gradient accumulation; a layer used
                    # twice in one step; no optimizer attached) is consumed by AccumulateGrad when this node ends and freed, so its
                    # transform must run inside the node
                    wn_grad_register(dW, W, g, norm, scale, dg)
                else:
                    def fused_now():
                        # an EARLIER use of this layer in the same pass may have left its raw gradient in the slot for the end-of-pass
                        # transform: autograd is about to add this node's (transformed) gradient to that buffer, so it must be transformed
                        # first -- everything registered so far is
                        if any(e[1].data_ptr() == W.data_ptr() for e in _wn_pending):
                            wn_grad_flush()
                        check(L.ekaid_wn_grad_fused(ptr(dW), ptr(W), ptr(g), ptr(norm), ptr(scale), dW.numel(), ptr(dg), stream()), 'wn_grad_fused')
                    deferred_call(fused_now)
                dscale = dg.view(g.shape)
            elif ctx.has_scale:
                dscale = torch.empty((), device=W.device, dtype=torch.float32)
                deferred_call(lambda: check(L.ekaid_scale_dot(ptr(dW), ptr(W), ptr(scale), dW.numel(), ptr(dscale), stream()), 'scale_dot'))
        if db_buf is not None:      # no weight-gradient product ran: the column sums on their own
            deferred_call(lambda: check(L.ekaid_colsum_acc(ptr(dz), M, N, N, ptr(db_buf), stream()), 'colsum_acc'))
        return (dW, db, dscale, None, None) + tuple(dxs)


# --------------------------------------------------------------------------- #
# weight norm (dim=None): all layers of a module tree at once
# --------------------------------------------------------------------------- #
_wn_cache = {}


def wn_prepare(layers):
    """scale = g / ||v|| and ||v|| of every (v, g) in ``layers`` (<= 32 per call) in two launches; the following
    ``wn_linear`` calls on these weights pick their values up (each entry is used once).  Call ``wn_release`` when the
    forward pass that owns the layers is over: the values are only valid until the next optimizer step."""
    _wn_cache.clear()
    L = hip.lib()
    for i0 in range(0, len(layers), 32):
        chunk = layers[i0:i0 + 32]
        n = len(chunk)
        dev = chunk[0][0].device
        scales = torch.empty(n, device=dev, dtype=torch.float32)
        norms = torch.empty(n, device=dev, dtype=torch.float32)
        vs = (ctypes.c_void_p * n)(*[v.data_ptr() for v, _ in chunk])
        gs = (ctypes.c_void_p * n)(*[g.data_ptr() for _, g in chunk])
        ns = (ctypes.c_longlong * n)(*[v.numel() for v, _ in chunk])
        check(L.ekaid_wn_scale_multi(vs, gs, ns, n, ptr(scales), ptr(norms), stream()), 'wn_scale_multi')
        for i, (v, _) in enumerate(chunk):
            _wn_cache[v.data_ptr()] = (scales[i], norms[i])


def wn_release():
    _wn_cache.clear()


def wn_scale(v, g):
    """(scale, ||v||) device scalars of one weight-normed layer: prepared values (``wn_prepare``) or two launches."""
    ent = _wn_cache.pop(v.data_ptr(), None) if v.is_contiguous() else None
    if ent is not None:
        return ent
    v = _c(v)
    scale = torch.empty((), device=v.device, dtype=torch.float32)
    norm = torch.empty((), device=v.device, dtype=torch.float32)
    check(hip.lib().ekaid_wn_scale_fwd(ptr(v), ptr(g), v.numel(), ptr(scale), ptr(norm), stream()), 'wn_scale_fwd')
    return scale, norm


class WNLinearFn(Function):
    """weight_norm(Linear, dim=None) as ONE node: y = act((g / ||v||) x v^T + b); backward through the norm fused into the
    weight-gradient epilogue.  Argument positions match LinearFn (v, b, g in place of W, b, scale).  ref: fc.py:26-33."""

    @staticmethod
    def forward(ctx, v, b, g, act, drops, *xs):
        require_cuda(v, g)
        with torch.no_grad():
            scale, norm = wn_scale(v.detach(), g.detach())
        out = LinearFn.forward(ctx, v, b, scale, act, drops, *xs)
        ctx.wn = (g.detach(), norm)
        return out

    @staticmethod
    def backward(ctx, dy):
        return LinearFn.backward(ctx, dy)


def linear(xs, W, b=None, scale=None, act=ACT_NONE, drops=None):
    if isinstance(xs, torch.Tensor):
        xs = (xs,)
    if drops is not None and not any(d > 0.0 for d in drops):
        drops = None
    return LinearFn.apply(W, b, scale, act, drops, *xs)


class LinearRowStackFn(Function):
    """y = [x_1; x_2; ...] W^T + b with the inputs stacked along ROWS into one output buffer (the two
    images of a pair share ``ChangeDetector.img``).  ref: modules.py:195-196."""

    @staticmethod
    def forward(ctx, W, b, *xs):
        require_cuda(W, *xs)
        W = _c(W)
        N, K = W.shape
        xs2, lds = zip(*[_rows2d(x) for x in xs])
        Ms = [x.shape[0] for x in xs2]
        y = torch.empty(sum(Ms), N, device=W.device, dtype=torch.float32)
        r0 = 0
        Wt = weight_transposed(W) if _use_transposed(sum(Ms)) else None
        x2 = f16x2_wanted(min(Ms), N, K)
        ctx.wb = weight_bounds(W) if x2 else (None, None)
        ctx.xb = [amax_rc(x, M, K, ld) if x2 else (None, None) for x, ld, M in zip(xs2, lds, Ms)]
        ctx.bgen = bounds_generation(W.device)
        for (x, ld, M), xb in zip(zip(xs2, lds, Ms), ctx.xb):
            if Wt is not None:
                gemm(x, Wt, y, M, N, K, ld, N, N, 0, 0, bias=b, c_off=r0 * N)
            else:
                gemm(x, W, y, M, N, K, ld, K, N, 0, 1, bias=b, c_off=r0 * N, a_amax=xb[0], b_amax=ctx.wb[0])
            r0 += M
        ctx.save_for_backward(W, *xs2)
        ctx.bias_ref = b.detach() if b is not None else None
        ctx.lds, ctx.Ms, ctx.has_b = lds, Ms, b is not None
        ctx.lead = [x.shape[:-1] for x in xs]
        return y

    @staticmethod
    def backward(ctx, dy):
        W, *xs2 = ctx.saved_tensors
        N, K = W.shape
        dy = _c(dy)
        Mtot = dy.shape[0]
        need = ctx.needs_input_grad
        db = db_buf = None
        if ctx.has_b and need[1]:
            db_buf, db = bias_grad_slot(ctx.bias_ref)
        dW = None
        if ctx.wb[0] is not None and ctx.bgen != bounds_generation(W.device):          # (stale: see bounds_pool_reset)
            ctx.wb = amax_rc(W, N, K, K)
            ctx.xb = [amax_rc(x, M, K, ld) for x, ld, M in zip(xs2, ctx.lds, ctx.Ms)]
        dyb = amax_rc(dy, Mtot, N, N) if ctx.wb[0] is not None else (None, None)
        if need[0]:
            dW = grad_out(W)
            r0 = 0
            for i, (x, ld, M) in enumerate(zip(xs2, ctx.lds, ctx.Ms)):     # every row block adds its rows' column sums
                gemm(dy, x, dW, N, K, M, N, ld, K, 1, 0, accumulate=i > 0, a_off=r0 * N, a_rowsum=db_buf, a_amax=dyb[1], b_amax=ctx.xb[i][1])
                r0 += M
        elif db_buf is not None:
            check(hip.lib().ekaid_colsum_acc(ptr(dy), Mtot, N, N, ptr(db_buf), stream()), 'colsum_acc')
        dxs = []
        r0 = 0
        for i, (x, M) in enumerate(zip(xs2, ctx.Ms)):
            if need[2 + i]:
                dx = torch.empty(M, K, device=W.device, dtype=torch.float32)
                gemm(dy, W, dx, M, K, N, N, K, K, 0, 0, a_off=r0 * N, a_amax=(dyb[0], r0) if dyb[0] is not None else None, b_amax=ctx.wb[1])
                dxs.append(dx.view(*ctx.lead[i], K))
            else:
                dxs.append(None)
            r0 += M
        return (dW, db) + tuple(dxs)


def linear_rowstack(xs, W, b=None):
    return LinearRowStackFn.apply(W, b, *xs)


class WNScaleFn(Function):
    """scale = g / ||v||_F of weight_norm(..., dim=None).  ref: fc.py:26-33."""

    @staticmethod
    def forward(ctx, v, g):
        require_cuda(v, g)
        v = _c(v)
        scale = torch.empty((), device=v.device, dtype=torch.float32)
        norm = torch.empty((), device=v.device, dtype=torch.float32)
        check(hip.lib().ekaid_wn_scale_fwd(ptr(v), ptr(g), v.numel(), ptr(scale), ptr(norm), stream()),
              'wn_scale_fwd')
        ctx.save_for_backward(v, g, norm)
        return scale

    @staticmethod
    def backward(ctx, dscale):
        v, g, norm = ctx.saved_tensors
        dg = grad_out(g)
        slot = grad_claimed_slot(v)          # the layer's wgrad (LinearFn.backward, run just before) already sits there
        dv = slot if slot is not None else torch.empty_like(v)
        check(hip.lib().ekaid_wn_scale_bwd(ptr(v), ptr(g), ptr(norm), ptr(_c(dscale)), v.numel(), ptr(dv), int(slot is not None),
                                           ptr(dg), stream()), 'wn_scale_bwd')
        return (None if slot is not None else dv), dg.view(g.shape)


def wn_linear_rowbcast(x, q, mask, group, v, g, b=None, act=ACT_NONE):
    """Inference form of ``wn_linear((x, expand_rows(q, mask, group)), v, g, b, act)`` without the expansion:
    y[r] = act(scale * x[r] . Wx^T + b + mask[r] * (scale * q[(r // group) % len(q)] . Wq^T)),  W = [Wx | Wq].
    The question product is taken once per sample (B rows instead of 2*B*N) and rides in the GEMM epilogue as a row-broadcast
    residual.  No autograd (callers use it under ``torch.no_grad()``).  ref: relation_encoder.py:19-29 + graph_att.py:80."""
    require_cuda(x, q, mask, v, g)
    x2, ldx = _rows2d(x)
    q2, ldq = _rows2d(q)
    mask = _c(mask)
    v = _c(v)
    M, Kx = x2.shape
    Kq = q2.shape[1]
    N = v.shape[0]
    assert v.shape[1] == Kx + Kq and M % group == 0 and mask.numel() == M
    scale, _ = wn_scale(v.detach(), g.detach())
    qw = torch.empty(q2.shape[0], N, device=x.device, dtype=torch.float32)
    gemm(q2, v, qw, q2.shape[0], N, Kq, ldq, Kx + Kq, N, 0, 1, alpha_dev=scale, b_off=Kx)
    slices = M // group
    if slices != qw.shape[0]:                       # the main and the reference image share the question rows
        assert slices % qw.shape[0] == 0
        qw = qw.repeat(slices // qw.shape[0], 1)
    y = torch.empty(M, N, device=x.device, dtype=torch.float32)
    gemm(x2, v, y, M, N, Kx, ldx, Kx + Kq, N, 0, 1, alpha_dev=scale, bias=b, R=qw, ldr=N, rdiv=group, rowmask=mask, act=act)
    return y.view(*x.shape[:-1], N)


def wn_linear(xs, v, g, b=None, act=ACT_NONE, drops=None):
    if isinstance(xs, torch.Tensor):
        xs = (xs,)
    if drops is not None and not any(d > 0.0 for d in drops):
        drops = None
    return WNLinearFn.apply(v, b, g, act, drops, *xs)


class _SubCtx:
    """What LinearFn.forward / backward need of an autograd context, for one half of a LinearPairFn."""
    wn = None
    wb = (None, None)
    xb = ()
    bgen = 0
    needs_input_grad = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors


class LinearPairFn(Function):
    """TWO independent linear layers (plain or weight-normed, one input each) as one node whose products go out pairwise: the two
    forward products in one launch, the two weight-gradient products in one launch, the two input-gradient products in one launch
    unless one joins the other through a GradSlot (readers of one tensor hand a running gradient along: then they stay a chain).
    Each half is exactly LinearFn / WNLinearFn -- same epilogues, dropout, slots, gradient buffers -- run inside a deferral window;
    results are bit-identical to two separate nodes (ekaid_gemm_pair_f32).
    Arguments: wn_a, wn_b (weight-normed?), then the five leading LinearFn arguments + the input of layer a, then those of layer b.
    ref: graph_att_layer.py:79,90 (query / key), modules.py:278-286 (context2 / gate2, context1 / gate1)."""

    @staticmethod
    def forward(ctx, wn_a, wn_b, *args):
        a, b = args[:6], args[6:]
        ca, cb = _SubCtx(), _SubCtx()
        qa, qb = [], []
        with _Deferring(qa):
            ya = (WNLinearFn if wn_a else LinearFn).forward(ca, *a)
        with _Deferring(qb):
            yb = (WNLinearFn if wn_b else LinearFn).forward(cb, *b)
        flush_pairs(qa, qb)
        # both halves' saved tensors go through the REAL context (released after backward, version-checked, visible to saved-tensor
        # hooks); the sub-contexts keep only plain Python state
        ctx.n_saved = (len(ca.saved_tensors), len(cb.saved_tensors))
        ctx.save_for_backward(*ca.saved_tensors, *cb.saved_tensors)
        ca.saved_tensors = cb.saved_tensors = None
        ctx.subs = (ca, cb)
        return ya, yb

    @staticmethod
    def backward(ctx, dya, dyb):
        ca, cb = ctx.subs
        saved = ctx.saved_tensors
        ca.saved_tensors, cb.saved_tensors = saved[:ctx.n_saved[0]], saved[ctx.n_saved[0]:]
        need = ctx.needs_input_grad
        ca.needs_input_grad, cb.needs_input_grad = need[2:8], need[8:14]
        assert dya is not None and dyb is not None, 'both halves of a paired layer must be used'
        qa, qb = [], []
        with _Deferring(qa):
            ga = LinearFn.backward(ca, dya)
        with _Deferring(qb):
            gb = LinearFn.backward(cb, dyb)
        flush_pairs(qa, qb)
        ca.saved_tensors = cb.saved_tensors = None
        return (None, None) + tuple(ga) + tuple(gb)


def linear_pair(layer_a, layer_b):
    """``layer`` = (x, W, b, g) with g = None for a plain nn.Linear (W = weight) or the weight-norm magnitude (W = weight_v), optionally
    followed by (act, input dropout p).  Returns (y_a, y_b)."""
    def unpack(layer):
        x, W, b, g = layer[:4]
        act = layer[4] if len(layer) > 4 else ACT_NONE
        p = float(layer[5]) if len(layer) > 5 else 0.0
        return (W, b, g, act, (p,) if p > 0.0 else None, x), g is not None
    a, wn_a = unpack(layer_a)
    b, wn_b = unpack(layer_b)
    return LinearPairFn.apply(wn_a, wn_b, *a, *b)


def predropped(alias, copy):
    """Mark ``alias`` (a tensor about to enter an FCNet with input dropout) as already dropped: ``copy`` = (p, seed, buffer) from
    the kernel that made the dropped copy on its way (``rowmask_dropout``).  The reader uses the buffer as data and (p, seed) for its
    backward."""
    if copy is not None:
        alias._ekaid_predrop = copy
    return alias


# --------------------------------------------------------------------------- #
# element-wise
# --------------------------------------------------------------------------- #
class DropoutFn(Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        require_cuda(x)
        x = _c(x)
        y = torch.empty_like(x)
        check(hip.lib().ekaid_dropout(ptr(x), ptr(y), x.numel(), p, seed, stream()), 'dropout')
        ctx.p, ctx.seed = p, seed
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        dx = torch.empty_like(dy)
        check(hip.lib().ekaid_dropout(ptr(dy), ptr(dx), dy.numel(), ctx.p, ctx.seed, stream()), 'dropout')
        return dx, None, None


def dropout(x, p, training):
    if not training or p <= 0.0:
        return x
    return DropoutFn.apply(x, float(p), next_seed())


class AxpbyFn(Function):
    """out = a*x + b*y."""

    @staticmethod
    def forward(ctx, x, y, a, b):
        require_cuda(x, y)
        x, y = _c(x), _c(y)
        out = torch.empty_like(x)
        check(hip.lib().ekaid_axpby(ptr(x), ptr(y), ptr(out), x.numel(), a, b, stream()), 'axpby')
        ctx.a, ctx.b = a, b
        return out

    @staticmethod
    def backward(ctx, d):
        d = _c(d)
        L = hip.lib()
        dx = dy = None
        if ctx.needs_input_grad[0]:
            if ctx.a == 1.0:
                dx = d                          # the incoming gradient itself (nothing downstream writes into gradients)
            else:
                dx = torch.empty_like(d)
                check(L.ekaid_axpby(ptr(d), None, ptr(dx), d.numel(), ctx.a, 0.0, stream()), 'axpby')
        if ctx.needs_input_grad[1]:
            if ctx.b == 1.0:
                dy = d
            else:
                dy = torch.empty_like(d)
                check(L.ekaid_axpby(ptr(d), None, ptr(dy), d.numel(), ctx.b, 0.0, stream()), 'axpby')
        return dx, dy, None, None


class SelfMix3Fn(Function):
    """a*x + b*x + c*x evaluated in that order (quirk Q-A: the reference's three "parallel" encoder outputs are one tensor,
    modules.py:234-235): one launch forward, one backward, no fan-out adds."""

    @staticmethod
    def forward(ctx, x, a, b, c):
        require_cuda(x)
        x = _c(x)
        L = hip.lib()
        out = torch.empty_like(x)
        check(L.ekaid_self_mix3(ptr(x), ptr(out), x.numel(), a, b, c, stream()), 'self_mix3')
        ctx.abc = (a, b, c)
        return out

    @staticmethod
    def backward(ctx, d):
        a, b, c = ctx.abc
        d = _c(d)
        L = hip.lib()
        dx = torch.empty_like(d)
        check(L.ekaid_self_mix3(ptr(d), ptr(dx), d.numel(), a, b, c, stream()), 'self_mix3')
        return dx, None, None, None


def self_mix3(x, a, b, c):
    return SelfMix3Fn.apply(x, float(a), float(b), float(c))


# --------------------------------------------------------------------------- #
# tensors with several consumers: one n-ary sum instead of autograd's pairwise adds
# --------------------------------------------------------------------------- #
def _sum_into(grads, out):
    """out = sum of the non-None ``grads`` (each the shape of ``out``); zero when there are none."""
    gs = [_c(g) for g in grads if g is not None]
    if not gs:
        out.zero_()
        return
    for i in range(0, len(gs), 7):                  # 8 addends per launch; later launches add onto ``out``
        part = ([out] if i else []) + gs[i:i + 7]
        arr = (ctypes.c_void_p * len(part))(*[g.data_ptr() for g in part])
        check(hip.lib().ekaid_sum_n(arr, len(part), ptr(out), out.numel(), stream()), 'sum_n')


class GradSlot:
    """The gradient of a tensor with several readers, handed from reader to reader during backward instead of being summed by a
    kernel at the end.  Every alias ``fanout`` returns carries the slot of its tensor (``_ekaid_slot``) and goes to ONE reader.  A
    reader whose backward knows slots either LENDS its gradient (``lend_grad``: the slot keeps a reference, the reader returns None
    to autograd) or FOLDS what is lent into its own product (the input-gradient GEMM of a linear layer adds it behind its dropout
    epilogue: ``ekaid_gemm_t.post``) and lends the result.  Lent tensors are only ever read.  ``FanoutFn.backward`` collects what is
    still lent next to the gradients that arrived the ordinary way, so any mix of readers stays correct."""
    __slots__ = ('lent',)

    def __init__(self):
        self.lent = None


def slot_of(x):
    return getattr(x, '_ekaid_slot', None)


def pass_slot(src, dst):
    """``dst`` is a reshaped view of the fan-out alias ``src`` and goes to src's reader in its place."""
    slot = slot_of(src)
    if slot is not None:
        dst._ekaid_slot = slot
    return dst


def lend_grad(slot, g):
    """What a reader's backward returns for an input that came out of ``fanout``: None when the gradient now waits in the slot."""
    if slot is None or g is None or slot.lent is not None or not g.is_contiguous():
        return g
    slot.lent = g
    return None


class FanoutFn(Function):
    """x -> n aliases of x, one per consumer.  Backward: gradients handed over through the GradSlot arrive already summed; the
    others are summed by ONE kernel (autograd would accumulate them pairwise: n-1 ATen adds, each re-reading the running sum).
    ref: every tensor of the path with more than one reader, e.g. the encoder output feeding context2 / gate2 / attention /
    pooling (modules.py:266-303)."""

    @staticmethod
    def forward(ctx, x, n, slot):
        ctx.set_materialize_grads(False)
        ctx.slot, ctx.shape = slot, tuple(x.shape)
        return tuple(x.view(x.shape) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        gs = [g for g in grads if g is not None]
        if ctx.slot.lent is not None:
            gs.append(ctx.slot.lent.view(ctx.shape))
            ctx.slot.lent = None
        if not gs:
            return None, None, None
        if len(gs) == 1:
            return gs[0], None, None
        out = torch.empty(gs[0].shape, device=gs[0].device, dtype=gs[0].dtype)
        _sum_into(gs, out)
        return out, None, None


def fanout(x, n):
    if n <= 1 or not x.requires_grad:
        return (x,) * n
    slot = GradSlot()
    outs = FanoutFn.apply(x, n, slot)
    for o in outs:
        o._ekaid_slot = slot
    return outs


class HalvesFn(Function):
    """x (2B, ...) -> n aliases of x[:B] followed by n aliases of x[B:] (the two images of a pair, stacked along the leading
    axis).  Backward: the gradients of each half are summed straight into their half of one buffer (autograd's slice
    backward would zero-fill a full-size tensor per use, copy the half in and add the results pairwise)."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(False)
        ctx.shape = tuple(x.shape)
        h = x.shape[0] // 2
        return tuple(x[:h] for _ in range(n)) + tuple(x[h:] for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        n = len(grads) // 2
        if all(g is None for g in grads):
            return None, None
        ref = next(g for g in grads if g is not None)
        out = torch.empty(ctx.shape, device=ref.device, dtype=ref.dtype)
        h = ctx.shape[0] // 2
        _sum_into(grads[:n], out[:h])
        _sum_into(grads[n:], out[h:])
        return out, None


def stack2(a, b):
    """cat((a, b), 0) -- as a VIEW when ``b`` directly follows ``a`` in one allocation (engine.forward_loss expands both
    images' adjacency into the halves of one buffer), else a copy."""
    if (a.shape == b.shape and a.dtype == b.dtype and a.is_contiguous() and b.is_contiguous() and a.device == b.device
            and not a.requires_grad and not b.requires_grad
            and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()
            and a.data_ptr() + a.numel() * a.element_size() == b.data_ptr()):
        return torch.as_strided(a, (2 * a.shape[0],) + tuple(a.shape[1:]), a.stride())
    return torch.cat((a, b), 0)


class Repeat2Fn(Function):
    """x (B, ...) -> [x; x] (2B, ...): the term both images of a pair share, laid out next to their row-stacked features so that a
    layer runs ONCE over 2B rows instead of once per image.  Backward: the two halves of the gradient summed by one kernel."""

    @staticmethod
    def forward(ctx, x):
        require_cuda(x)
        return torch.cat([x, x], 0)

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        h = g.shape[0] // 2
        out = torch.empty((h,) + tuple(g.shape[1:]), device=g.device, dtype=g.dtype)
        _sum_into([g[:h], g[h:]], out)
        return out


def repeat2(x):
    return Repeat2Fn.apply(x)


def halves(x, n=1):
    """(first-half aliases..., second-half aliases...) of a tensor stacked along axis 0; see HalvesFn."""
    assert x.shape[0] % 2 == 0 and x.is_contiguous()
    if not x.requires_grad:
        h = x.shape[0] // 2
        return (x[:h],) * n + (x[h:],) * n
    return HalvesFn.apply(x, n)


def sub(x, y):
    return AxpbyFn.apply(x, y, 1.0, -1.0)


def add(x, y):
    return AxpbyFn.apply(x, y, 1.0, 1.0)


class ScaleFn(Function):
    @staticmethod
    def forward(ctx, x, a):
        require_cuda(x)
        x = _c(x)
        out = torch.empty_like(x)
        check(hip.lib().ekaid_axpby(ptr(x), None, ptr(out), x.numel(), a, 0.0, stream()), 'axpby')
        ctx.a = a
        return out

    @staticmethod
    def backward(ctx, d):
        d = _c(d)
        dx = torch.empty_like(d)
        check(hip.lib().ekaid_axpby(ptr(d), None, ptr(dx), d.numel(), ctx.a, 0.0, stream()), 'axpby')
        return dx, None


def scale(x, a):
    return ScaleFn.apply(x, float(a))


class MulFn(Function):
    @staticmethod
    def forward(ctx, x, y):
        require_cuda(x, y)
        x, y = _c(x), _c(y)
        out = torch.empty_like(x)
        check(hip.lib().ekaid_mul(ptr(x), ptr(y), ptr(out), x.numel(), stream()), 'mul')
        ctx.save_for_backward(x, y)
        return out

    @staticmethod
    def backward(ctx, d):
        x, y = ctx.saved_tensors
        d = _c(d)
        L = hip.lib()
        dx = torch.empty_like(d)
        dy = torch.empty_like(d)
        check(L.ekaid_mul(ptr(d), ptr(y), ptr(dx), d.numel(), stream()), 'mul')
        check(L.ekaid_mul(ptr(d), ptr(x), ptr(dy), d.numel(), stream()), 'mul')
        return dx, dy


def mul(x, y):
    return MulFn.apply(x, y)


def rowsum(x, as_mask=False):
    """No-grad row reduction over the last axis (condensed adjacency / zero-row mask)."""
    require_cuda(x)
    x = _c(x.detach())
    rows = x.numel() // x.shape[-1]
    out = torch.empty(x.shape[:-1], device=x.device, dtype=torch.float32)
    check(hip.lib().ekaid_rowsum(ptr(x), rows, x.shape[-1], ptr(out), int(as_mask), stream()), 'rowsum')
    return out


def rowmask_dropout(x, p):
    """(zero-row mask of x [rows], seed, dropout(x)) in one pass over x (no gradient: the dropped copy is DATA for a linear layer
    that takes x itself as its autograd input, see ``predropped``)."""
    require_cuda(x)
    xc = _c(x.detach())
    cols = xc.shape[-1]
    rows = xc.numel() // cols
    mask = torch.empty(rows, device=x.device, dtype=torch.float32)
    xd = torch.empty(rows, cols, device=x.device, dtype=torch.float32)
    seed = next_seed()
    check(hip.lib().ekaid_rowmask_dropout(ptr(xc), rows, cols, ptr(mask), ptr(xd), float(p), seed, stream()), 'rowmask_dropout')
    return mask, seed, xd


class ExpandRowsFn(Function):
    """out[r] = dropout(mask[r] * q[(r // group) % qrows])  (the q half of q_expand_v_cat, with the dropout of the
    following FCNet fused in).  ref: relation_encoder.py:19-29, fc.py:22-24."""

    @staticmethod
    def forward(ctx, q, mask, group, p, seed):
        require_cuda(q, mask)
        q, mask = _c(q), _c(mask)
        rows = mask.numel()
        out = torch.empty(rows, q.shape[1], device=q.device, dtype=torch.float32)
        check(hip.lib().ekaid_expand_rows_fwd(ptr(q), ptr(mask), rows, group, q.shape[0], q.shape[1], ptr(out), p, seed,
                                              stream()), 'expand_rows_fwd')
        ctx.cfg = (group, p, seed)
        ctx.save_for_backward(mask)
        ctx.qshape = q.shape
        return out

    @staticmethod
    def backward(ctx, d):
        (mask,) = ctx.saved_tensors
        group, p, seed = ctx.cfg
        d = _c(d)
        dq = torch.empty(ctx.qshape, device=d.device, dtype=torch.float32)
        check(hip.lib().ekaid_expand_rows_bwd(ptr(d), ptr(mask), mask.numel(), group, ctx.qshape[0], ctx.qshape[1],
                                              ptr(dq), p, seed, stream()), 'expand_rows_bwd')
        return dq, None, None, None, None


def expand_rows(q, mask, group, p=0.0, training=False):
    p = float(p) if training else 0.0
    return ExpandRowsFn.apply(q, mask, group, p, next_seed() if p > 0 else 0)


class ResidualRelu2Fn(Function):
    """v + relu(dropout(2*o)).  ref: graph_att.py:98-104 (Q-B), relation_encoder.py:80-81,128-129 (Q-A)."""

    @staticmethod
    def forward(ctx, v, o, p, seed):
        require_cuda(v, o)
        v, o = _c(v), _c(o)
        out = torch.empty_like(v)
        check(hip.lib().ekaid_residual_relu2_fwd(ptr(v), ptr(o), ptr(out), v.numel(), p, seed, stream()),
              'residual_relu2_fwd')
        ctx.p, ctx.seed = p, seed
        ctx.slot_v = slot_of(v)
        ctx.save_for_backward(o)
        return out

    @staticmethod
    def backward(ctx, d):
        (o,) = ctx.saved_tensors
        d = _c(d)
        do = torch.empty_like(d)
        check(hip.lib().ekaid_residual_relu2_bwd(ptr(o), ptr(d), ptr(do), d.numel(), ctx.p, ctx.seed, stream()),
              'residual_relu2_bwd')
        return lend_grad(ctx.slot_v, d), do, None, None


def residual_relu2(v, o, p, training):
    p = float(p) if training else 0.0
    return ResidualRelu2Fn.apply(v, o, p, next_seed() if p > 0 else 0)


class GateFuseFn(Function):
    """dropout(sigmoid(g2 + g1)) * dropout(tanh(c2 + c1)), c1/g1 broadcast over the two images.
    ref: modules.py:278-288."""

    @staticmethod
    def forward(ctx, c2, g2, c1, g1, p, seed_c, seed_g):
        require_cuda(c2, g2, c1, g1)
        c2, g2, c1, g1 = _c(c2), _c(g2), _c(c1), _c(g1)
        out = torch.empty_like(c2)
        check(hip.lib().ekaid_gate_fuse_fwd(ptr(c2), ptr(g2), ptr(c1), ptr(g1), ptr(out), c2.numel(), c1.numel(), p,
                                            seed_c, seed_g, stream()), 'gate_fuse_fwd')
        ctx.args = (p, seed_c, seed_g)
        ctx.save_for_backward(c2, g2, c1, g1)
        return out

    @staticmethod
    def backward(ctx, d):
        c2, g2, c1, g1 = ctx.saved_tensors
        d = _c(d)
        dc2, dg2 = torch.empty_like(c2), torch.empty_like(c2)
        dc1, dg1 = torch.empty_like(c1), torch.empty_like(c1)
        p, sc, sg = ctx.args
        check(hip.lib().ekaid_gate_fuse_bwd(ptr(c2), ptr(g2), ptr(c1), ptr(g1), ptr(d), ptr(dc2), ptr(dg2), ptr(dc1),
                                            ptr(dg1), c2.numel(), c1.numel(), p, sc, sg, stream()), 'gate_fuse_bwd')
        return dc2, dg2, dc1, dg1, None, None, None


def gate_fuse(c2, g2, c1, g1, p, training):
    p = float(p) if training else 0.0
    if p > 0:
        return GateFuseFn.apply(c2, g2, c1, g1, p, next_seed(), next_seed())
    return GateFuseFn.apply(c2, g2, c1, g1, 0.0, 0, 0)


class WeightedPoolFn(Function):
    """out[g] = sum_i w[g,i] * x[g,i,:].  ref: modules.py:305-308, language_model.py:151."""

    @staticmethod
    def forward(ctx, w, x):
        require_cuda(w, x)
        w, x = _c(w), _c(x)
        G, n, C = x.shape
        out = torch.empty(G, C, device=x.device, dtype=torch.float32)
        check(hip.lib().ekaid_weighted_pool_fwd(ptr(w), ptr(x), G, n, C, ptr(out), stream()), 'weighted_pool_fwd')
        ctx.save_for_backward(w, x)
        return out

    @staticmethod
    def backward(ctx, d):
        w, x = ctx.saved_tensors
        G, n, C = x.shape
        d = _c(d)
        dw = torch.empty_like(w)
        dx = torch.empty_like(x)
        check(hip.lib().ekaid_weighted_pool_bwd(ptr(w), ptr(x), ptr(d), G, n, C, ptr(dw), ptr(dx), stream()),
              'weighted_pool_bwd')
        return dw, dx


def weighted_pool(w, x):
    return WeightedPoolFn.apply(w.reshape(x.shape[0], x.shape[1]), x)


class QAttWeightsFn(Function):
    """softmax over the batch axis of atten^T re-read as (B, L) (quirk Q-C).  ref: language_model.py:149."""

    @staticmethod
    def forward(ctx, atten):
        require_cuda(atten)
        atten = _c(atten)
        B, L = atten.shape
        S = torch.empty(L, B, device=atten.device, dtype=torch.float32)
        w = torch.empty(B, L, device=atten.device, dtype=torch.float32)
        check(hip.lib().ekaid_qatt_weights_fwd(ptr(atten), B, L, ptr(S), ptr(w), stream()), 'qatt_weights_fwd')
        ctx.save_for_backward(S)
        return w

    @staticmethod
    def backward(ctx, dw):
        (S,) = ctx.saved_tensors
        L, B = S.shape
        dw = _c(dw)
        da = torch.empty(B, L, device=S.device, dtype=torch.float32)
        check(hip.lib().ekaid_qatt_weights_bwd(ptr(S), ptr(dw), B, L, ptr(da), stream()), 'qatt_weights_bwd')
        return da


def embedding_grad(tok, table, dout, ld_out, col_off, relu, padding_idx, dtable):
    """dtable (V, E) += gradient of ``act(table[tok])`` given dout (ntok rows, leading dim ld_out, columns col_off..+E):
    OneHot(tok)^T . dout as a split-K GEMM with a fixed-order reduction -- deterministic, and its cost does not depend on how
    many positions share an id (padding and '<start>' ids hit thousands).  ref: autograd of nn.Embedding (+ ReLU),
    language_model.py:48-53, dynamic_speaker_change_pos.py:153-156."""
    V, E = table.shape
    ntok = tok.numel()
    L = hip.lib()
    onehot = torch.empty(ntok, V, device=table.device, dtype=torch.float32)
    check(L.ekaid_onehot(ptr(tok), ntok, V, int(padding_idx), ptr(onehot), stream()), 'onehot')
    raw = torch.empty(V, E, device=table.device, dtype=torch.float32)
    gemm(onehot, dout, raw, V, E, ntok, V, ld_out, E, 1, 0, b_off=col_off)
    check(L.ekaid_masked_add(ptr(raw), ptr(table), int(relu), ptr(dtable), V * E, stream()), 'masked_add')


class EmbeddingFn(Function):
    """One or two lookup tables written side by side (+ optional ReLU).
    ref: language_model.py:48-53, dynamic_speaker_change_pos.py:153-156."""

    @staticmethod
    def forward(ctx, tok, relu, padding_idx, *tables):
        require_cuda(tok, *tables)
        tok = _c(tok)
        ntok = tok.numel()
        tables = [_c(t) for t in tables]
        width = sum(t.shape[1] for t in tables)
        out = torch.empty(*tok.shape, width, device=tok.device, dtype=torch.float32)
        off = 0
        for t in tables:
            check(hip.lib().ekaid_embedding_fwd(ptr(tok), ntok, ptr(t), t.shape[1], ptr(out), width, off, int(relu),
                                                stream()), 'embedding_fwd')
            off += t.shape[1]
        ctx.relu, ctx.pad, ctx.width = relu, padding_idx, width
        ctx.save_for_backward(tok, *tables)
        return out

    @staticmethod
    def backward(ctx, d):
        tok, *tables = ctx.saved_tensors
        d = _c(d)
        grads = []
        off = 0
        for i, t in enumerate(tables):
            if ctx.needs_input_grad[3 + i]:
                dt = grad_out(t, zero=True)
                embedding_grad(tok, t, d, ctx.width, off, ctx.relu, ctx.pad, dt)
                grads.append(dt)
            else:
                grads.append(None)
            off += t.shape[1]
        return (None, None, None) + tuple(grads)


def embedding(tok, tables, relu=False, padding_idx=-1):
    return EmbeddingFn.apply(tok, relu, padding_idx, *tables)


class SoftmaxFn(Function):
    @staticmethod
    def forward(ctx, x, log_mode):
        require_cuda(x)
        x = _c(x)
        y = torch.empty_like(x)
        cols = x.shape[-1]
        check(hip.lib().ekaid_softmax_fwd(ptr(x), ptr(y), x.numel() // cols, cols, int(log_mode), stream()),
              'softmax_fwd')
        ctx.log_mode = log_mode
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = _c(dy)
        dx = torch.empty_like(y)
        cols = y.shape[-1]
        check(hip.lib().ekaid_softmax_bwd(ptr(y), ptr(dy), ptr(dx), y.numel() // cols, cols, int(ctx.log_mode),
                                          stream()), 'softmax_bwd')
        return dx, None


def softmax(x):
    return SoftmaxFn.apply(x, False)


def log_softmax(x):
    return SoftmaxFn.apply(x, True)


class Mix3Fn(Function):
    """w[:,0]*a + w[:,1]*b + w[:,2]*c.  ref: dynamic_speaker_change_pos.py:112-117."""

    @staticmethod
    def forward(ctx, w, a, b, c):
        require_cuda(w, a, b, c)
        w, a, b, c = _c(w), _c(a), _c(b), _c(c)
        B, C = a.shape
        out = torch.empty_like(a)
        check(hip.lib().ekaid_mix3_fwd(ptr(w), ptr(a), ptr(b), ptr(c), B, C, ptr(out), stream()), 'mix3_fwd')
        ctx.save_for_backward(w, a, b, c)
        return out

    @staticmethod
    def backward(ctx, d):
        w, a, b, c = ctx.saved_tensors
        B, C = a.shape
        d = _c(d)
        dw = torch.empty_like(w)
        da, db, dc = torch.empty_like(a), torch.empty_like(a), torch.empty_like(a)
        check(hip.lib().ekaid_mix3_bwd(ptr(w), ptr(a), ptr(b), ptr(c), ptr(d), B, C, ptr(dw), ptr(da), ptr(db),
                                       ptr(dc), 0, stream()), 'mix3_bwd')
        return dw, da, db, dc


class GRUSeqFn(Function):
    """Whole 1-layer unidirectional GRU (h0 = 0) with explicit BPTT: one input GEMM over all time
    steps, then per step one recurrent GEMM + the fused gate kernel writing h_t straight into the
    output sequence.  ref: language_model.py:66-70,106-115 (nn.GRU, gate order r,z,n)."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh):
        require_cuda(x, w_ih, w_hh, b_ih, b_hh)
        x, w_ih, w_hh = _c(x), _c(w_ih), _c(w_hh)
        B, T, E = x.shape
        H = w_hh.shape[1]
        dev = x.device
        L = hip.lib()
        gi = torch.empty(B, T, 3 * H, device=dev, dtype=torch.float32)
        x2 = f16x2_wanted(B * T, 3 * H, E)              # (mode 2: bounds of x and W_ih, reused by the three time-batched products)
        ctx.xb = amax_rc(x, B * T, E, E) if x2 else (None, None)
        ctx.wb = weight_bounds(w_ih) if x2 else (None, None)
        ctx.bgen = bounds_generation(dev)
        gemm(x, w_ih, gi, B * T, 3 * H, E, E, E, 3 * H, 0, 1, bias=b_ih, a_amax=ctx.xb[0], b_amax=ctx.wb[0])
        hall = torch.zeros(T + 1, B, H, device=dev, dtype=torch.float32)     # time-major states, hall[0] = h0 = 0
        save = torch.empty(T, B, 4 * H, device=dev, dtype=torch.float32)
        if seq_in_c():
            need = L.ekaid_gru_seq_workspace_bytes(B, T, H, 0)
            ws = hip.workspace(need, dev)
            check(L.ekaid_gru_seq_fwd(ptr(gi), ptr(w_hh), ptr(b_hh), B, T, H, ptr(hall), ptr(save), ptr(ws), need, stream()),
                  'gru_seq_fwd')
        for t in range(0 if seq_in_c() else T):
            # h_{t-1} . W_hh^T as raw K-slabs; the cell kernel sums them, adds b_hh and applies the gates (one launch fewer
            # per step than GEMM + reduce + cell)
            S = gemm_slabs(hall, w_hh, B, 3 * H, H, H, H, 0, 1, a_off=t * B * H)
            sl = S.cols(0)
            check(L.ekaid_gru_cell_slabs_fwd(gi.data_ptr() + 4 * t * 3 * H, T * 3 * H, ctypes.byref(sl), ptr(b_hh),
                                             hall.data_ptr() + 4 * t * B * H, H, B, H,
                                             hall.data_ptr() + 4 * (t + 1) * B * H, H, save.data_ptr() + 4 * t * B * 4 * H,
                                             stream()), 'gru_cell_slabs_fwd')
        ctx.save_for_backward(x, w_ih, w_hh, hall, save)
        ctx.b_refs = (b_ih.detach(), b_hh.detach())          # (aliases: their data_ptr finds the gradient slots)
        return hall[1:].transpose(0, 1)          # (B, T, H) view of the time-major buffer

    @staticmethod
    def backward(ctx, dout):
        x, w_ih, w_hh, hall, save = ctx.saved_tensors
        b_ih, b_hh = ctx.b_refs
        B, T, E = x.shape
        H = w_hh.shape[1]
        dev = x.device
        L = hip.lib()
        dout = _c(dout)                                   # (B, T, H)
        dgi = torch.empty(B, T, 3 * H, device=dev, dtype=torch.float32)
        dgh = torch.empty(T, B, 3 * H, device=dev, dtype=torch.float32)
        dh_dir = [torch.empty(B, H, device=dev, dtype=torch.float32) for _ in range(2)]    # z * d of step t (ping-pong)
        carry = None                                      # slabs of dgh_{t+1} . W_hh
        if seq_in_c():
            need = L.ekaid_gru_seq_workspace_bytes(B, T, H, 1)
            ws = hip.workspace(need, dev)
            check(L.ekaid_gru_seq_bwd(ptr(dout), ptr(w_hh), ptr(hall), ptr(save), B, T, H, ptr(dgi), ptr(dgh), ptr(ws), need,
                                      stream()), 'gru_seq_bwd')
        for t in (() if seq_in_c() else range(T - 1, -1, -1)):
            csl = carry.cols(0) if carry is not None else NO_SLABS
            check(L.ekaid_gru_cell_slabs_bwd(dout.data_ptr() + 4 * t * H, T * H, ptr(dh_dir[(t + 1) & 1]) if t < T - 1 else None,
                                             ctypes.byref(csl), save.data_ptr() + 4 * t * B * 4 * H,
                                             hall.data_ptr() + 4 * t * B * H, H, B, H, dgi.data_ptr() + 4 * t * 3 * H, T * 3 * H,
                                             dgh.data_ptr() + 4 * t * B * 3 * H, ptr(dh_dir[t & 1]), stream()), 'gru_cell_slabs_bwd')
            if t > 0:       # dh_{t-1} = z*d (dh_dir) + dgh_t . W_hh (slabs, summed by the next cell kernel)
                carry = gemm_slabs(dgh, w_hh, B, H, 3 * H, 3 * H, H, 0, 0, a_off=t * B * 3 * H)
        # weight gradients batched over time: K = T*B rows (previous states are the contiguous slice hall[0:T])
        dw_hh = grad_out(w_hh)
        hh_buf, db_hh = bias_grad_slot(b_hh)
        ih_buf, db_ih = bias_grad_slot(b_ih)
        dw_ih = grad_out(w_ih)
        x2 = ctx.wb[0] is not None
        if x2 and ctx.bgen != bounds_generation(dev):                  # (stale: see bounds_pool_reset)
            ctx.xb, ctx.wb = amax_rc(x, B * T, E, E), amax_rc(w_ih, 3 * H, E, E)
        dgib = amax_rc(dgi, B * T, 3 * H, 3 * H) if x2 else (None, None)
        dghb = amax_rc(dgh, T * B, 3 * H, 3 * H, want_rows=False) if x2 else (None, None)
        hb = amax_rc(hall, T * B, H, H, want_rows=False) if x2 else (None, None)
        with pairing_window():                        # two independent products of one layout: one launch
            gemm(dgh, hall, dw_hh, 3 * H, H, T * B, 3 * H, H, H, 1, 0, a_rowsum=hh_buf, a_amax=dghb[1], b_amax=hb[1])
            gemm(dgi, x, dw_ih, 3 * H, E, B * T, 3 * H, E, E, 1, 0, a_rowsum=ih_buf, a_amax=dgib[1], b_amax=ctx.xb[1])
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            gemm(dgi, w_ih, dx, B * T, E, 3 * H, 3 * H, E, E, 0, 0, a_amax=dgib[0], b_amax=ctx.wb[1])
        return dx, dw_ih, dw_hh, db_ih, db_hh


def gru_sequence(x, w_ih, w_hh, b_ih, b_hh):
    out = GRUSeqFn.apply(x, w_ih, w_hh, b_ih, b_hh)
    assert_finite(out, 'question GRU')
    return out


class LSTMCellFn(Function):
    """Point-wise half of nn.LSTMCell: gates = gx + gh (i,f,g,o)."""

    @staticmethod
    def forward(ctx, gx, gh, c):
        require_cuda(gx, gh, c)
        gx, gh, c = _c(gx), _c(gh), _c(c)
        B, H = c.shape
        h2, c2 = torch.empty_like(c), torch.empty_like(c)
        save = torch.empty(B, 4 * H, device=c.device, dtype=torch.float32)
        check(hip.lib().ekaid_lstm_cell_fwd(ptr(gx), ptr(gh), 4 * H, ptr(c), B, H, ptr(h2), ptr(c2), ptr(save), stream()),
              'lstm_cell_fwd')
        ctx.save_for_backward(save, c, c2)
        return h2, c2

    @staticmethod
    def backward(ctx, dh2, dc2):
        save, c, c2 = ctx.saved_tensors
        B, H = c.shape
        dg = torch.empty(B, 4 * H, device=c.device, dtype=torch.float32)
        dc = torch.empty_like(c)
        check(hip.lib().ekaid_lstm_cell_bwd(ptr(_c(dh2)) if dh2 is not None else None, None,
                                            ptr(_c(dc2)) if dc2 is not None else None, ptr(save), ptr(c), ptr(c2), B, H,
                                            ptr(dg), 4 * H, ptr(dc), stream()), 'lstm_cell_bwd')
        return dg, dg, dc


class LMLossFn(Function):
    """loss = -sum(logp[b,t,target[b,t]] * mask[b,t]) / sum(mask).  ref: utils/utils.py:208-216."""

    @staticmethod
    def forward(ctx, logp, target, mask, denom):
        require_cuda(logp, target, mask, denom)
        logp = _c(logp)
        B, T, V = logp.shape
        assert target.stride(1) == 1 and mask.stride(1) == 1 and target.dtype == torch.int64
        assert target.shape[1] >= T and mask.shape[1] >= T and mask.dtype == torch.float32
        sums = torch.empty(2, device=logp.device, dtype=torch.float32)       # written by the kernel
        loss = torch.empty((), device=logp.device, dtype=torch.float32)
        check(hip.lib().ekaid_nll_fwd(ptr(logp), target.data_ptr(), target.stride(0), mask.data_ptr(), mask.stride(0),
                                      B, T, V, ptr(denom), ptr(sums), ptr(loss), stream()), 'nll_fwd')
        ctx.save_for_backward(target, mask, sums)
        ctx.shape = (B, T, V)
        return loss

    @staticmethod
    def backward(ctx, d):
        target, mask, sums = ctx.saved_tensors
        B, T, V = ctx.shape
        dlogp = torch.empty(B, T, V, device=sums.device, dtype=torch.float32)
        check(hip.lib().ekaid_nll_bwd(target.data_ptr(), target.stride(0), mask.data_ptr(), mask.stride(0), ptr(sums),
                                      ptr(_c(d)), B, T, V, ptr(dlogp), stream()), 'nll_bwd')
        return dlogp, None, None, None


def lm_loss(logp, target, mask, denom=None):
    """``denom``: optional device scalar that replaces sum(mask) (data-parallel loss weighting, include/ekaid_hip.h)."""
    return LMLossFn.apply(logp, target, mask, denom)


class TotalLossFn(Function):
    """speaker_loss + coef * (sum(att_bef) + sum(att_aft)).  ref: train_mimic.py:246-247."""

    @staticmethod
    def forward(ctx, sl, att_b, att_a, coef):
        require_cuda(sl, att_b, att_a)
        att_b, att_a = _c(att_b), _c(att_a)
        out = torch.empty((), device=sl.device, dtype=torch.float32)
        check(hip.lib().ekaid_total_loss_fwd(ptr(sl), ptr(att_b), ptr(att_a), att_b.numel(), coef, ptr(out), stream()),
              'total_loss_fwd')
        ctx.coef = coef
        ctx.shapes = (att_b.shape, att_a.shape)
        return out

    @staticmethod
    def backward(ctx, d):
        d = _c(d)
        L = hip.lib()
        gb = torch.empty(ctx.shapes[0], device=d.device, dtype=torch.float32)
        ga = torch.empty(ctx.shapes[1], device=d.device, dtype=torch.float32)
        check(L.ekaid_fill_scaled(ptr(d), ctx.coef, ptr(gb), gb.numel(), stream()), 'fill_scaled')
        check(L.ekaid_fill_scaled(ptr(d), ctx.coef, ptr(ga), ga.numel(), stream()), 'fill_scaled')
        return d, gb, ga, None


def total_loss(speaker_loss, att_bef, att_aft):
    B = att_bef.shape[0]
    return TotalLossFn.apply(speaker_loss, att_bef, att_aft, 2.5e-03 / (2 * B))


class LayerNormFn(Function):
    """ref: modules.py:33,76 (nn.LayerNorm(att_dim, eps=1e-6))."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        require_cuda(x, gamma, beta)
        x = _c(x)
        cols = x.shape[-1]
        rows = x.numel() // cols
        y = torch.empty_like(x)
        save = torch.empty(rows, 2, device=x.device, dtype=torch.float32)
        check(hip.lib().ekaid_layernorm_fwd(ptr(x), ptr(gamma), ptr(beta), rows, cols, eps, ptr(y), ptr(save), stream()),
              'layernorm_fwd')
        ctx.save_for_backward(x, gamma, save)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, save = ctx.saved_tensors
        cols = x.shape[-1]
        rows = x.numel() // cols
        dy = _c(dy)
        dx = torch.empty_like(x)
        dg, db = torch.empty_like(gamma), torch.empty_like(gamma)
        check(hip.lib().ekaid_layernorm_bwd(ptr(x), ptr(gamma), ptr(save), ptr(dy), rows, cols, ptr(dx), ptr(dg), ptr(db),
                                            stream()), 'layernorm_bwd')
        return dx, dg, db, None


def layer_norm(x, gamma, beta, eps):
    return LayerNormFn.apply(x, gamma, beta, float(eps))


# --------------------------------------------------------------------------- #
# graph attention core
# --------------------------------------------------------------------------- #
def geom_embed(bb_dev, M, feat_dim, dim_mat, p=0.0):
    """float64 boxes (R,N,4) on the device -> float32 pair embedding (R,N,M,feat_dim); no gradient.  ``p`` > 0: the input dropout
    of the ONE layer that reads it, applied as it is written (``out._ekaid_dropped`` = p tells the reader)."""
    require_cuda(bb_dev, dim_mat)
    assert bb_dev.dtype == torch.float64
    bb_dev = _c(bb_dev)
    R, N, _ = bb_dev.shape
    out = torch.empty(R, N, M, feat_dim, device=bb_dev.device, dtype=torch.float32)
    if p > 0.0:
        check(hip.lib().ekaid_geom_embed_dropout(bb_dev.data_ptr(), ptr(dim_mat), R, N, M, feat_dim, ptr(out), float(p), next_seed(),
                                                 stream()), 'geom_embed_dropout')
        out._ekaid_dropped = float(p)
    else:
        check(hip.lib().ekaid_geom_embed(bb_dev.data_ptr(), ptr(dim_mat), R, N, M, feat_dim, ptr(out), stream()), 'geom_embed')
    return out


_GAT_DQK = os.environ.get('EKAID_GAT_DQK', '1') != '0'      # A/B knob: 0 = dq / dk as two batched products
_GAT_BWD_CORE = os.environ.get('EKAID_GAT_BWD_CORE', '1') != '0'      # A/B knob: 0 = dP / dS as two batched products + the softmax-backward kernel
# one workgroup per slice: from this many slices on the one-launch form wins (kernel time of the backward, 36 nodes x 4 heads x 1024:
# 512 slices 435 -> 338 us, 256: 223 -> 207, 128: 145 -> 176, 64: 112 -> 167; profiles/r06_gat_bwd_core_ab.txt)
_GAT_BWD_CORE_MIN_SLICES = int(os.environ.get('EKAID_GAT_BWD_CORE_MIN_SLICES', 256))


class GatAttentionFn(Function):
    """Neighbour gather + edge softmax + weighted aggregation of GraphSelfAttentionLayer.
    ref: graph_att_layer.py:105-172.  Forward = the fused kernel (impl='fused') or the same math
    composed from batched GEMMs + the edge-softmax kernel (impl='gemm'); backward = batched GEMMs
    around the edge-softmax backward kernel."""

    @staticmethod
    def forward(ctx, q, k, S, cond, lbias, geo, H, direction, impl):
        require_cuda(q, k, S, cond, lbias)
        ctx.slot_S = slot_of(S)
        q, k, S, cond, lbias = _c(q), _c(k), _c(S), _c(cond), _c(lbias)
        geo = _c(geo) if geo is not None else None
        R, N, A = q.shape
        M = k.shape[1]
        dh = A // H
        L = hip.lib()
        P = torch.empty(R, N, H, M, device=q.device, dtype=torch.float32)
        outT = torch.empty(R, N, H * A, device=q.device, dtype=torch.float32)
        done = False
        if impl == 'fused':
            rc = L.ekaid_gat_fused_fwd(ptr(q), ptr(k), ptr(S), ptr(cond), ptr(lbias), ptr(geo), R, N, M, A, H,
                                       direction, ptr(P), ptr(outT), stream())
            if rc == hip.ERR_UNSUPPORTED:
                impl = 'gemm'              # shape outside the fused kernel's envelope: still HIP, composed
            else:
                check(rc, 'gat_fused_fwd')
                done = True
        if not done:
            aff = torch.empty(R, N, H, M, device=q.device, dtype=torch.float32)
            gemm(q, k, aff, N, M, dh, A, A, H * M, 0, 1, alpha=1.0 / math.sqrt(float(dh)), batch=(R, H),
                 strides=(N * A, dh, M * A, dh, N * H * M, M))
            check(L.ekaid_masked_softmax_fwd(ptr(aff), ptr(cond), ptr(lbias), ptr(geo), R, N, M, H, direction, ptr(P),
                                             stream()), 'masked_softmax_fwd')
            gemm(P, S, outT, N * H, A, M, M, A, A, 0, 0, batch=(R, 1), strides=(N * H * M, 0, N * A, 0, N * H * A, 0))
        ctx.cfg = (H, direction)
        ctx.set_materialize_grads(False)         # an unused output (the returned softmax) arrives as None, not as zeros
        ctx.save_for_backward(q, k, S, cond, geo, P)
        ctx.has_geo = geo is not None
        return outT, P

    @staticmethod
    def backward(ctx, doutT, dP_ext):
        q, k, S, cond, geo, P = ctx.saved_tensors
        H, direction = ctx.cfg
        R, N, A = q.shape
        M = k.shape[1]
        dh = A // H
        sc = 1.0 / math.sqrt(float(dh))
        L = hip.lib()
        dev = q.device
        if doutT is None:
            doutT = torch.zeros(R, N, H * A, device=dev, dtype=torch.float32)
        doutT = _c(doutT)
        doutT = _c(doutT)
        dS = torch.zeros(R, N, A, device=dev, dtype=torch.float32) if M < N else \
            torch.empty(R, N, A, device=dev, dtype=torch.float32)
        daff = torch.empty(R, N, H, M, device=dev, dtype=torch.float32)
        dlbias = torch.zeros(R, N, N, device=dev, dtype=torch.float32) if M < N else torch.empty(R, N, N, device=dev, dtype=torch.float32)
        dgeo = torch.empty_like(geo) if ctx.has_geo else None
        # d(aggregation) + d(edge softmax) in ONE launch that reads doutT once (csrc/gat.hip: gat_bwd_core_kernel); outside its envelope
        # the same from two batched products around the softmax-backward kernel
        rc = L.ekaid_gat_bwd_core(ptr(doutT), ptr(S), ptr(P), ptr(_c(dP_ext)) if dP_ext is not None else None, ptr(cond), ptr(geo) if ctx.has_geo else None,
                                  R, N, M, A, H, direction, ptr(dS), ptr(daff), ptr(dlbias), ptr(dgeo), stream()) if (_GAT_BWD_CORE and R >= _GAT_BWD_CORE_MIN_SLICES) else hip.ERR_UNSUPPORTED
        if rc == hip.ERR_UNSUPPORTED:
            # dP = doutT . S[:M]^T   (R, N*H, M)
            dP = torch.empty(R, N, H, M, device=dev, dtype=torch.float32)
            gemm(doutT, S, dP, N * H, M, A, A, A, M, 0, 1, batch=(R, 1), strides=(N * H * A, 0, N * A, 0, N * H * M, 0))
            if dP_ext is not None:
                dP = AxpbyFn.apply(dP, _c(dP_ext), 1.0, 1.0)
            # dS[:M] = P^T . doutT
            gemm(P, doutT, dS, M, A, N * H, M, A, A, 1, 0, batch=(R, 1), strides=(N * H * M, 0, N * H * A, 0, N * A, 0))
            check(L.ekaid_masked_softmax_bwd(ptr(P), ptr(dP), ptr(cond), ptr(geo), R, N, M, H, direction, ptr(daff),
                                             ptr(dlbias), ptr(dgeo), stream()), 'masked_softmax_bwd')
        else:
            check(rc, 'gat_bwd_core')
        # dq[r,:,h] = sc * daff[r,:,h,:] . k[r,:,h]      dk[r,:,h] = sc * daff[r,:,h,:]^T . q[r,:,h]: one kernel (both are 36-deep
        # contractions: poor batched products), two batched products outside its envelope
        dq = torch.empty(R, N, A, device=dev, dtype=torch.float32)
        dk = torch.empty(R, M, A, device=dev, dtype=torch.float32)
        rc = L.ekaid_gat_dqk(ptr(daff), ptr(q), ptr(k), R, N, M, A, H, sc, ptr(dq), ptr(dk), stream()) if _GAT_DQK else hip.ERR_UNSUPPORTED
        if rc == hip.ERR_UNSUPPORTED:
            gemm(daff, k, dq, N, dh, M, H * M, A, A, 0, 0, alpha=sc, batch=(R, H),
                 strides=(N * H * M, M, M * A, dh, N * A, dh))
            gemm(daff, q, dk, M, dh, N, H * M, A, A, 1, 0, alpha=sc, batch=(R, H),
                 strides=(N * H * M, M, N * A, dh, M * A, dh))
        else:
            check(rc, 'gat_dqk')
        return dq, dk, lend_grad(ctx.slot_S, dS), None, dlbias, dgeo, None, None, None


def gat_attention(q, k, S, cond, lbias, geo, H, direction, impl='fused'):
    outT, P = GatAttentionFn.apply(q, k, S, cond, lbias, geo, H, direction, impl)
    assert_finite(P, 'graph attention softmax')
    assert_finite(outT, 'graph attention aggregation')
    return outT, P


def bmm(A, B, transA=False, transB=False, alpha=1.0):
    return BmmFn.apply(A, B, transA, transB, alpha)


class BmmFn(Function):
    """Batched matmul C[b] = alpha * op(A[b]) op(B[b]) for 3-D contiguous tensors (transA and transB
    are mutually exclusive).  Used by the SSRE attention (ref: modules.py:61,71)."""

    @staticmethod
    def forward(ctx, A, B, transA, transB, alpha):
        require_cuda(A, B)
        A, B = _c(A), _c(B)
        nb = A.shape[0]
        M = A.shape[2] if transA else A.shape[1]
        K = A.shape[1] if transA else A.shape[2]
        N = B.shape[1] if transB else B.shape[2]
        C = torch.empty(nb, M, N, device=A.device, dtype=torch.float32)
        gemm(A, B, C, M, N, K, A.shape[2], B.shape[2], N, transA, transB, alpha=alpha, batch=(nb, 1),
             strides=(A.shape[1] * A.shape[2], 0, B.shape[1] * B.shape[2], 0, M * N, 0))
        ctx.cfg = (transA, transB, alpha)
        ctx.save_for_backward(A, B)
        return C

    @staticmethod
    def backward(ctx, dC):
        A, B = ctx.saved_tensors
        transA, transB, alpha = ctx.cfg
        dC = _c(dC)
        dA = dB = None
        if not transA and not transB:          # C = A B
            dA = BmmFn.apply(dC, B, False, True, alpha)      # dC B^T
            dB = BmmFn.apply(A, dC, True, False, alpha)      # A^T dC
        elif transB:                            # C = A B^T
            dA = BmmFn.apply(dC, B, False, False, alpha)     # dC B
            dB = BmmFn.apply(dC, A, True, False, alpha)      # dC^T A
        else:                                   # C = A^T B
            dA = BmmFn.apply(B, dC, False, True, alpha)      # B dC^T
            dB = BmmFn.apply(A, dC, False, False, alpha)     # A dC
        return dA, dB, None, None, None


class HeadAttentionFn(Function):
    """softmax(q_h k_h^T / sqrt(dh)) v_h for every (sample, head) of (B, L, heads * dh) tensors, WITHOUT the head-major copies: the
    batched products take the heads as a second batch dimension with a stride of dh inside the rows (ekaid_gemm_t batch2 / strides),
    and the result is written straight into the (B, L, heads * dh) layout the LayerNorm reads.  Dropout on the probabilities.
    ref: modules.py:43-75 (SelfAttention.transpose_for_scores / permute / view: four materialised transposes per call)."""

    @staticmethod
    def forward(ctx, q, k, v, heads, p, training):
        require_cuda(q, k, v)
        q, k, v = _c(q), _c(k), _c(v)
        B, Lq, D = q.shape
        dh = D // heads
        L = hip.lib()
        f32 = dict(device=q.device, dtype=torch.float32)
        scores = torch.empty(B, heads, Lq, Lq, **f32)
        st_in, st_sc = (Lq * D, dh), (heads * Lq * Lq, Lq * Lq)
        gemm(q, k, scores, Lq, Lq, dh, D, D, Lq, 0, 1, alpha=1.0 / math.sqrt(dh), batch=(B, heads),
             strides=(st_in[0], st_in[1], st_in[0], st_in[1], st_sc[0], st_sc[1]))
        probs = torch.empty_like(scores)
        check(L.ekaid_softmax_fwd(ptr(scores), ptr(probs), B * heads * Lq, Lq, 0, stream()), 'softmax_fwd')
        pd, seed = probs, 0
        if training and p > 0.0:
            pd, seed = torch.empty_like(probs), next_seed()
            check(L.ekaid_dropout(ptr(probs), ptr(pd), probs.numel(), float(p), seed, stream()), 'dropout')
        out = torch.empty(B, Lq, D, **f32)
        gemm(pd, v, out, Lq, dh, Lq, Lq, D, D, 0, 0, batch=(B, heads),
             strides=(st_sc[0], st_sc[1], st_in[0], st_in[1], st_in[0], st_in[1]))
        ctx.cfg = (heads, float(p) if training else 0.0, seed)
        ctx.save_for_backward(q, k, v, probs, pd)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, probs, pd = ctx.saved_tensors
        heads, p, seed = ctx.cfg
        B, Lq, D = q.shape
        dh = D // heads
        L = hip.lib()
        dout = _c(dout)
        st_in, st_sc = (Lq * D, dh), (heads * Lq * Lq, Lq * Lq)
        s_in, s_sc = (st_in[0], st_in[1]), (st_sc[0], st_sc[1])
        alpha = 1.0 / math.sqrt(dh)
        dpd = torch.empty_like(probs)                       # d probs_dropped[b, h] = dout_bh v_bh^T
        gemm(dout, v, dpd, Lq, Lq, dh, D, D, Lq, 0, 1, batch=(B, heads), strides=s_in + s_in + s_sc)
        dv = torch.empty_like(v)                            # dv_bh = probs_dropped_bh^T dout_bh
        gemm(pd, dout, dv, Lq, dh, Lq, Lq, D, D, 1, 0, batch=(B, heads), strides=s_sc + s_in + s_in)
        if p > 0.0:
            check(L.ekaid_dropout(ptr(dpd), ptr(dpd), dpd.numel(), p, seed, stream()), 'dropout')
        dsc = torch.empty_like(probs)
        check(L.ekaid_softmax_bwd(ptr(probs), ptr(dpd), ptr(dsc), B * heads * Lq, Lq, 0, stream()), 'softmax_bwd')
        dq, dk = torch.empty_like(q), torch.empty_like(k)
        gemm(dsc, k, dq, Lq, dh, Lq, Lq, D, D, 0, 0, alpha=alpha, batch=(B, heads), strides=s_sc + s_in + s_in)
        gemm(dsc, q, dk, Lq, dh, Lq, Lq, D, D, 1, 0, alpha=alpha, batch=(B, heads), strides=s_sc + s_in + s_in)
        return dq, dk, dv, None, None, None


def head_attention(q, k, v, heads, p=0.0, training=False):
    return HeadAttentionFn.apply(q, k, v, heads, p, training)
