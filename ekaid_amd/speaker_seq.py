"""Teacher-forced DynamicSpeaker sequence as ONE autograd node with explicit back-propagation through time.

ref: dynamic_speaker_change_pos.py:94-131 (DynamicCore.forward), :182-240 (_forward / get_logprobs_state).

Why: taken step by step through autograd the decoder is ~60 tiny launches per step and every weight
receives T partial gradients (16 wgrad GEMMs with K = B, 16 bias reductions and 15 accumulation adds per
weight).  Here everything that does not depend on the recurrence is batched over time
  - input-side gate pre-activations  (dropout_t(embed) . W_ih_att[:, :E]^T,  relu(Emb[tok]) . W_ih_lang[:, :We]^T)
  - the answer-logit GEMM + log-softmax over all T steps at once
  - EVERY weight gradient: one contraction over K = T*B rows per weight, one bias reduction per bias
and only the truly sequential part (two LSTM cells, the POS branch and the gate MLP: 12 small GEMMs per step)
stays in the time loop.  All activations are stored time-major (T, B, .) so that "previous hidden state
for every t" is a contiguous slice of the (T+1, B, R) state buffer.
"""
import ctypes
import os

import torch
from torch.autograd import Function

from . import hip, ops
from .hip import ACT_NONE, ACT_RELU, ACT_SIGMOID, check, ptr, stream
from .ops import gemm

PERSISTENT_BWD_MIN_STEPS = int(os.environ.get('EKAID_SEQ_BWD_MIN_STEPS', '32'))
# ... or any number of steps at small batches, where the step-wise chain's ~10 launches per decode step are issue latency, not work:
# same-box A/B on the replayed rungs (profiles/r06_ab_seq_bwd_min.txt, r06_ab_seq_bwd_batch.txt; ms x GHz): 32 pairs -0.7 ... -2.5 %,
# 64 pairs level (-1.4 ... +0.4 %), 128 level, 256 +2.8 % -> up to 32 pairs
PERSISTENT_BWD_MAX_BATCH_SHORT = int(os.environ.get('EKAID_SEQ_BWD_MAX_BATCH_SHORT', '32'))

PARAM_NAMES = ('embed.0.weight', 'core.embed.0.weight', 'core.embed.0.bias',
               'core.module_att_lstm.weight_ih', 'core.module_att_lstm.weight_hh',
               'core.module_att_lstm.bias_ih', 'core.module_att_lstm.bias_hh',
               'core.weight_fc.0.weight', 'core.weight_fc.0.bias',
               'core.pos1.0.weight', 'core.pos1.0.bias', 'core.weight_pos.weight', 'core.weight_pos.bias',
               'core.pos2.weight', 'core.pos2.bias',
               'core.gate1x.0.weight', 'core.gate1x.0.bias', 'core.gate2x.weight', 'core.gate2x.bias',
               'core.lang_lstm.weight_ih', 'core.lang_lstm.weight_hh', 'core.lang_lstm.bias_ih', 'core.lang_lstm.bias_hh',
               'logit.weight', 'logit.bias')


def _nt(X, xoff, ldx, W, woff, ldw, Y, yoff, ldy, M, N, K, bias=None, acc=False, act=ACT_NONE, ab=None, bb=None):
    """Y[M,N] (+)= X[M,K] . W[N,K]^T (+ bias) with element offsets / leading dims; ab / bb: operand bounds (ops.gemm a_amax / b_amax)."""
    gemm(X, W, Y, M, N, K, ldx, ldw, ldy, 0, 1, accumulate=acc, bias=bias, act=act, a_off=xoff, b_off=woff, c_off=yoff, a_amax=ab, b_amax=bb)


def _nn(D, doff, ldd, W, woff, ldw, Y, yoff, ldy, M, N, K, acc=False, ab=None, bb=None):
    """Y[M,N] (+)= D[M,K] . W[K,N]   (dgrad: W stored [out=K][in], column block at woff)."""
    gemm(D, W, Y, M, N, K, ldd, ldw, ldy, 0, 0, accumulate=acc, a_off=doff, b_off=woff, c_off=yoff, a_amax=ab, b_amax=bb)


def _cols(bounds, off=0):
    """The column maxima of a bounds pair as an ops.gemm a_amax / b_amax argument, from column ``off`` on (None when there are none)."""
    return None if bounds is None or bounds[1] is None else (bounds[1], off)


def _tn(D, ldd, X, ldx, G, goff, ldg, Nout, Kin, rows, doff=0, xoff=0, bias=None, db=None, xb=None):
    """G[Nout,Kin] = D[rows,Nout]^T . X[rows,Kin]   (wgrad over rows = T*B); ``bias``: parameter(s) whose gradient is the column
    sum of D -- it rides on this product (ops.gemm a_rowsum); returns their autograd values.  db / xb: bounds pairs of the matrices D
    and X are column blocks of (ops.amax_rc: their column maxima serve the two-term f16 split; None = exact split)."""
    kw = {}
    if db is not None and xb is not None and db[1] is not None and xb[1] is not None and ops.f16x2_wanted(Nout, Kin, rows, 0):
        kw = dict(a_amax=(db[1], doff % ldd), b_amax=(xb[1], xoff % ldx))       # (offsets: rows * ld + first column)
    if bias is None:
        gemm(D, X, G, Nout, Kin, rows, ldd, ldx, ldg, 1, 0, a_off=doff, b_off=xoff, c_off=goff, **kw)
        return None
    biases = bias if isinstance(bias, (tuple, list)) else (bias,)
    buf, ret = ops.bias_grad_slot(biases[0])
    gemm(D, X, G, Nout, Kin, rows, ldd, ldx, ldg, 1, 0, a_off=doff, b_off=xoff, c_off=goff, a_rowsum=buf, **kw)
    rets = [ret]
    for b2 in biases[1:]:           # a second bias with the same gradient (bias_ih / bias_hh of an LSTM cell)
        buf2, ret2 = ops.bias_grad_slot(b2)
        if ret is not None and ret2 is not None:       # both slots fresh: buf IS the column sum, add it to the zeroed second slot
            ops.deferred_call(lambda buf=buf, buf2=buf2: check(hip.lib().ekaid_axpby(ptr(buf), ptr(buf2), ptr(buf2), buf.numel(), 1.0, 1.0, stream()), 'axpby'))
        else:
            ops.deferred_call(lambda buf2=buf2: check(hip.lib().ekaid_colsum_acc(D.data_ptr() + 4 * doff, rows, Nout, ldd, ptr(buf2), stream()), 'colsum_acc'))
        rets.append(ret2)
    return rets[0] if len(rets) == 1 else rets


def _seq_desc(T, B, R, A, NP, G1, We, p, p5, seeds_t, *tensors):
    """ekaid_speaker_seq_t; ``tensors`` in the order of its pointer fields (None where a direction does not read one)."""
    sq = hip.SpeakerSeq()
    sq.T, sq.B, sq.R, sq.A, sq.P, sq.G1, sq.We = T, B, R, A, NP, G1, We
    sq.p_lm, sq.p_pos = p, p5
    flat = (ctypes.c_uint64 * (3 * T))(*[s for trip in seeds_t for s in trip])
    sq._seeds_keepalive = flat                  # host array read during the call only
    sq.seeds = ctypes.cast(flat, ctypes.c_void_p)
    names = [k for k, _ in hip.SpeakerSeq._fields_[10:]]
    assert len(names) == len(tensors)
    for k, t in zip(names, tensors):
        setattr(sq, k, ptr(t))
    return sq


def _drop(x, y, n, p, seed, xoff=0, yoff=0):
    check(hip.lib().ekaid_dropout(x.data_ptr() + 4 * xoff, y.data_ptr() + 4 * yoff, n, p, seed, stream()), 'dropout')


class SpeakerSeqFn(Function):
    """Layout of the per-step "h_lang projection" block (width NW = 4R + R + G1 + 4R):
        [0, 4R)          h_lang_{t-1} . W_ih_att[:, E:]^T   -> += h_att_{t-1} . W_hh_att^T + b_hh   = att-LSTM hidden-side gates
        [4R, 5R)         h_lang_{t-1} . W_pos1^T                                                   = pos1 pre-activation
        [5R, 5R+G1)      h_lang_{t-1} . W_g1[:, :R]^T       -> += ppos . W_g1[:, R:2R]^T += att . W_g1[:, 2R:]^T + b, relu, dropout
        [5R+G1, NW)      h_lang_{t-1} . W_hh_lang^T         -> += gated . W_ih_lang[:, We:]^T + b_hh = lang-LSTM hidden-side gates
    All four products share the operand h_lang_{t-1}, so they are ONE GEMM against the row-concatenated weight
    ``Wcat`` (NW x R); in the backward pass the four gradient blocks are laid out the same way (``DCAT``), so that
    d h_lang_{t-1} = DCAT_t . Wcat is again one GEMM and the weight gradients read column slices of DCAT."""

    @staticmethod
    def forward(ctx, bef, aft, diff, tok_tb, p_lm, p_pos, training, *params):
        hip.require_cuda(bef, aft, diff, tok_tb, *params)
        (emb, w_e, b_e, wih_a, whh_a, bih_a, bhh_a, w_fc, b_fc, w_p1, b_p1, w_wp, b_wp, w_p2, b_p2,
         w_g1, b_g1, w_g2, b_g2, wih_l, whh_l, bih_l, bhh_l, w_lo, b_lo) = [p.contiguous() for p in params]
        bef, aft, diff = bef.contiguous(), aft.contiguous(), diff.contiguous()
        T, B = tok_tb.shape
        A = bef.shape[1]
        E, R, We, V = w_e.shape[0], whh_a.shape[1], emb.shape[1], w_lo.shape[0]
        G1 = w_g1.shape[0]                        # 2R + A
        NP = w_wp.shape[0]                        # POS classes (16)
        NW = 9 * R + G1
        oP1, oG1, oGL = 4 * R, 5 * R, 5 * R + G1   # column offsets inside the NW-wide block
        dev = bef.device
        L = hip.lib()
        f32 = dict(device=dev, dtype=torch.float32)
        p = float(p_lm) if training else 0.0
        p5 = float(p_pos) if training else 0.0      # core.dropout (nn.Dropout(0.5)) on the POS logits
        seeds = {k: ops.next_seed() for k in ('ed', 'xt', 'out')}
        seeds_t = [(ops.next_seed(), ops.next_seed(), ops.next_seed()) for _ in range(T)]     # vpos, dpos, g1
        TB = T * B

        # ---- time-batched, recurrence-independent part
        embed_pre = torch.empty(B, E, **f32)
        _nt(bef, 0, A, w_e, 0, 3 * A, embed_pre, 0, E, B, E, A)
        _nt(diff, 0, A, w_e, A, 3 * A, embed_pre, 0, E, B, E, A, acc=True)
        _nt(aft, 0, A, w_e, 2 * A, 3 * A, embed_pre, 0, E, B, E, A, bias=b_e, acc=True, act=ACT_RELU)
        ED = torch.empty(T, B, E, **f32)             # dropout mask differs per step (step index folded into the seed)
        check(L.ekaid_dropout_steps(ptr(embed_pre), ptr(ED), B * E, T, 0, p, seeds['ed'], stream()), 'dropout_steps')
        Xr = torch.empty(T, B, We, **f32)
        check(L.ekaid_embedding_fwd(tok_tb.data_ptr(), TB, ptr(emb), We, ptr(Xr), We, 0, 1, stream()), 'embedding_fwd')
        Xd = torch.empty_like(Xr)
        _drop(Xr, Xd, TB * We, p, seeds['xt'])
        # two-term f16 split (mode 2) for the time-batched products: bounds of their operands, taken once and kept for the backward pass
        x2 = ops.f16x2_wanted(TB, 4 * R, We, ops.F16X2_MIN_WORK_SHARED)
        none2 = (None, None)
        if x2:
            ops.weight_bounds_prepare([wih_a, wih_l, w_lo, w_g1])
        wb = {k: (ops.weight_bounds(w) if x2 else none2) for k, w in (('wih_a', wih_a), ('wih_l', wih_l), ('w_lo', w_lo), ('w_g1', w_g1))}
        if x2:
            ops.weight_bounds_release()
        edb = ops.amax_rc(ED, TB, E, E) if x2 else none2
        xdb = ops.amax_rc(Xd, TB, We, We) if x2 else none2
        GXA = torch.empty(T, B, 4 * R, **f32)     # att-LSTM input-side gates: dropout(embed) part + b_ih
        _nt(ED, 0, E, wih_a, 0, E + R, GXA, 0, 4 * R, TB, 4 * R, E, bias=bih_a, ab=edb[0], bb=wb['wih_a'][0])
        GXL = torch.empty(T, B, 4 * R, **f32)     # lang-LSTM input-side gates: word part + b_ih
        _nt(Xd, 0, We, wih_l, 0, We + A, GXL, 0, 4 * R, TB, 4 * R, We, bias=bih_l, ab=xdb[0], bb=wb['wih_l'][0])
        # row-concatenated weights applied to h_lang_{t-1} (a copy, no arithmetic)
        Wcat = torch.cat([wih_a[:, E:], w_p1, w_g1[:, :R], whh_l], 0).contiguous()          # (NW, R)

        # HA / HL: h of the module-attention / language LSTM, CA / CL their cells; row 0 = the zero initial state
        HA, CA, HL, CL = ops.zeros_pool(dev, (T + 1, B, R), (T + 1, B, R), (T + 1, B, R), (T + 1, B, R))
        SA = torch.empty(T, B, 4 * R, **f32)      # gate activations (i,f,g,o)
        SL = torch.empty(T, B, 4 * R, **f32)
        G1D = torch.empty(T, B, G1, **f32)        # dropout(relu(gate1x)): input of gate2x
        MW = torch.empty(T, B, 3, **f32)
        VP = torch.empty(T, B, R, **f32)          # dropout(relu(pos1(h_lang)))
        DP = torch.empty(T, B, NP, **f32)         # dropout(weight_pos(vpos))   (= output_pos of the core)
        PW = torch.empty(T, B, NP, **f32)
        PA = torch.empty(T, B, R + A, **f32)      # [pos_feat | att_feat]: the two trailing inputs of the gate MLP, one GEMM operand
        AT = torch.empty(T, B, A, **f32)
        GT = torch.empty(T, B, A, **f32)          # gate (sigmoid)
        GA = torch.empty(T, B, A, **f32)          # gated att feat

        # Per step: 4 GEMM launches whose K-slabs are summed by their consumers (csrc/steps.hip), 4 consumer kernels -- issued by
        # ONE sequence-level call (csrc/sequence.hip); the loop below is the same chain call by call (GEMM profile, cross-check)
        in_c = ops.seq_in_c()
        if in_c:
            sq = _seq_desc(T, B, R, A, NP, G1, We, p, p5, seeds_t, Wcat, whh_a, bhh_a, w_fc, b_fc, b_p1, w_wp, b_wp, w_p2, b_p2,
                           w_g1, b_g1, w_g2, b_g2, wih_l, bhh_l, bef, diff, aft, GXA, GXL, HA, CA, HL, CL, SA, SL, G1D, MW, VP,
                           DP, PW, PA, AT, GT, GA)
            need = L.ekaid_speaker_seq_workspace_bytes(ctypes.byref(sq), 0)
            ws = hip.workspace(need, dev)
            check(L.ekaid_speaker_seq_fwd(ctypes.byref(sq), ptr(ws), need, stream()), 'speaker_seq_fwd')
        sa = hip.SpkStepAFwd()
        sa.B, sa.R, sa.A, sa.P = B, R, A, NP
        sa.b_hh, sa.w_fc, sa.b_fc = ptr(bhh_a), ptr(w_fc), ptr(b_fc)
        sa.bef, sa.diff, sa.aft = ptr(bef), ptr(diff), ptr(aft)
        sa.b_p1, sa.w_wp, sa.b_wp, sa.w_p2, sa.b_p2 = ptr(b_p1), ptr(w_wp), ptr(b_wp), ptr(w_p2), ptr(b_p2)
        sa.p_v, sa.p_d, sa.ld2, sa.ldpp, sa.n_gates_src = p, p5, R + A, R + A, 2
        for t in range(0 if in_c else T):
            o_prev, o_t = t * B * R, (t + 1) * B * R
            sv, sd, sg = seeds_t[t]
            # 0.+1. everything h_lang_{t-1} feeds (P0, NW columns) and h_att_{t-1} . W_hh_att^T (P1): two products, one launch
            P0, P1 = ops.gemm_slabs2((HL, Wcat, B, NW, R, R, R, 0, 1, o_prev, 0), (HA, whh_a, B, 4 * R, R, R, R, 0, 1, o_prev, 0))
            # att-LSTM cell -> module weights + attended feature -> POS chain: one row-local kernel
            sa.gxa = GXA.data_ptr() + 4 * t * B * 4 * R
            sa.gates_src[0], sa.gates_src[1] = P0.cols(0), P1.cols(0)
            sa.c_prev = CA.data_ptr() + 4 * o_prev
            sa.h_out, sa.c_out = HA.data_ptr() + 4 * o_t, CA.data_ptr() + 4 * o_t
            sa.gate_save = SA.data_ptr() + 4 * t * B * 4 * R
            sa.mw = MW.data_ptr() + 4 * t * B * 3
            sa.att = AT.data_ptr() + 4 * t * B * A
            sa.att2 = PA.data_ptr() + 4 * (t * B * (R + A) + R)
            sa.pos_src = P0.cols(oP1)
            sa.seed_v, sa.seed_d = sv, sd
            sa.vp, sa.dp, sa.pw = VP.data_ptr() + 4 * t * B * R, DP.data_ptr() + 4 * t * B * NP, PW.data_ptr() + 4 * t * B * NP
            sa.pp = PA.data_ptr() + 4 * t * B * (R + A)
            check(L.ekaid_spk_step_a_fwd(ctypes.byref(sa), stream()), 'spk_step_a_fwd')
            # gate MLP, layer 1: relu/dropout(P0 block + [pos_feat | att] W_g1[:, R:]^T + b)
            P2 = ops.gemm_slabs(PA, w_g1, B, G1, R + A, R + A, G1, 0, 1, a_off=t * B * (R + A), b_off=R)
            arr, n = ops.slab_array(P0.cols(oG1), P2.cols(0))
            check(L.ekaid_slab_epilogue(arr, n, ptr(b_g1), B, G1, ACT_RELU, p, sg, G1D.data_ptr() + 4 * t * B * G1, G1, stream()),
                  'slab_epilogue')
            # layer 2 + gating: gate = sigmoid(. W_g2^T + b); gated = gate * att
            P3 = ops.gemm_slabs(G1D, w_g2, B, A, G1, G1, G1, 0, 1, a_off=t * B * G1)
            s3 = P3.cols(0)
            check(L.ekaid_gate_mul_slabs_fwd(ctypes.byref(s3), ptr(b_g2), AT.data_ptr() + 4 * t * B * A, B, A,
                                             GT.data_ptr() + 4 * t * B * A, GA.data_ptr() + 4 * t * B * A, stream()), 'gate_mul_slabs_fwd')
            # language LSTM: gates = GXL[t] + P0 block + gated W_ih[:, We:]^T + b_hh
            P4 = ops.gemm_slabs(GA, wih_l, B, 4 * R, A, A, We + A, 0, 1, a_off=t * B * A, b_off=We)
            arr, n = ops.slab_array(P0.cols(oGL), P4.cols(0))
            check(L.ekaid_lstm_cell_slabs_fwd(GXL.data_ptr() + 4 * t * B * 4 * R, arr, n, ptr(bhh_l), CL.data_ptr() + 4 * o_prev, B, R,
                                              HL.data_ptr() + 4 * o_t, CL.data_ptr() + 4 * o_t, SL.data_ptr() + 4 * t * B * 4 * R,
                                              stream()), 'lstm_cell_slabs_fwd')

        # ---- time-batched head: dropout(h_lang) -> logit -> log-softmax;  log-softmax of the POS logits
        HO = torch.empty(T, B, R, **f32)
        _drop(HL, HO, TB * R, p, seeds['out'], B * R, 0)
        logits = torch.empty(T, B, V, **f32)
        hob = ops.amax_rc(HO, TB, R, R) if x2 else none2
        _nt(HO, 0, R, w_lo, 0, R, logits, 0, V, TB, V, R, bias=b_lo, ab=hob[0], bb=wb['w_lo'][0])
        logp = torch.empty(T, B, V, **f32)
        check(L.ekaid_softmax_fwd(ptr(logits), ptr(logp), TB, V, 1, stream()), 'softmax_fwd')
        logpos = torch.empty(T, B, NP, **f32)
        check(L.ekaid_softmax_fwd(ptr(DP), ptr(logpos), TB, NP, 1, stream()), 'softmax_fwd')

        ctx.dims = (T, B, A, E, R, We, V, G1)
        # aliases of the bias parameters: their data_ptr finds the gradient slots (ops.bias_grad_slot)
        ctx.biases = tuple(t.detach() for t in (b_e, bih_a, bhh_a, b_fc, b_p1, b_wp, b_p2, b_g1, b_g2, bih_l, bhh_l, b_lo))
        ctx.wrefs = (w_p1.detach(), whh_l.detach())   # (read through Wcat in the backward pass; needed for their slots only)
        ctx.pp = (p, p5, seeds, seeds_t)
        ctx.bounds = (wb, edb, xdb, hob, ops.bounds_generation(dev))
        ctx.save_for_backward(bef, aft, diff, tok_tb, embed_pre, ED, Xd, HA, CA, HL, CL, SA, SL, MW, VP, PW, PA, AT,
                              G1D, GT, GA, HO, logp, logpos, emb, w_e, wih_a, whh_a, w_fc, w_wp, w_p2, w_g1, w_g2,
                              wih_l, w_lo, Wcat)
        ctx.mark_non_differentiable(MW)
        return logp, logpos, MW

    @staticmethod
    def backward(ctx, dlogp, dlogpos, _dmw):
        (bef, aft, diff, tok_tb, embed_pre, ED, Xd, HA, CA, HL, CL, SA, SL, MW, VP, PW, PA, AT, G1D, GT, GA, HO,
         logp, logpos, emb, w_e, wih_a, whh_a, w_fc, w_wp, w_p2, w_g1, w_g2, wih_l, w_lo, Wcat) = ctx.saved_tensors
        T, B, A, E, R, We, V, G1 = ctx.dims
        NP = w_wp.shape[0]
        NW = 9 * R + G1
        oP1, oG1, oGL = 4 * R, 5 * R, 5 * R + G1
        p, p5, seeds, seeds_t = ctx.pp
        b_e, bih_a, bhh_a, b_fc, b_p1, b_wp, b_p2, b_g1, b_g2, bih_l, bhh_l, b_lo = ctx.biases
        w_p1, whh_l = ctx.wrefs
        dev = bef.device
        L = hip.lib()
        f32 = dict(device=dev, dtype=torch.float32)
        TB = T * B
        dlogp = dlogp.contiguous()
        # ---- head (time-batched)
        dlogits = torch.empty(T, B, V, **f32)
        check(L.ekaid_softmax_bwd(ptr(logp), ptr(dlogp), ptr(dlogits), TB, V, 1, stream()), 'softmax_bwd')
        wb, edb, xdb, hob, bgen = ctx.bounds
        x2 = edb[0] is not None
        none2 = (None, None)
        if x2 and bgen != ops.bounds_generation(dev):                  # another forward pass has recycled the bounds pool since: afresh
            wb = {k: ops.amax_rc(w, w.shape[0], w.shape[1], w.shape[1]) for k, w in (('wih_a', wih_a), ('wih_l', wih_l), ('w_lo', w_lo), ('w_g1', w_g1))}
            edb, xdb, hob = ops.amax_rc(ED, TB, E, E), ops.amax_rc(Xd, TB, We, We), ops.amax_rc(HO, TB, R, R)
        dlb = ops.amax_rc(dlogits, TB, V, V) if x2 else none2
        d_w_lo = ops.grad_out(w_lo)
        d_b_lo = _tn(dlogits, V, HO, R, d_w_lo, 0, R, V, R, TB, bias=b_lo, db=dlb, xb=hob)
        dHO = torch.empty(T, B, R, **f32)
        _nn(dlogits, 0, V, w_lo, 0, R, dHO, 0, R, TB, R, V, ab=dlb[0], bb=_cols(wb['w_lo']))
        dHLo = torch.empty(T, B, R, **f32)        # gradient reaching h_lang_t through the output dropout
        _drop(dHO, dHLo, TB * R, p, seeds['out'])
        # dDPl: gradient reaching dpos through log_softmax(output_pos); d_bef/d_diff/d_aft and the cell-state carries start at
        # zero: one allocation, one fill
        dDPl, d_bef, d_diff, d_aft, dc_l, dc_a = ops.zeros_pool(dev, (T, B, NP), (B, A), (B, A), (B, A), (B, R), (B, R))
        if dlogpos is not None:
            check(L.ekaid_softmax_bwd(ptr(logpos), ptr(dlogpos.contiguous()), ptr(dDPl), TB, NP, 1, stream()), 'softmax_bwd')

        DCAT = torch.empty(T, B, NW, **f32)       # [d att gates | d pos1 pre-act | d gate1x pre-act | d lang gates] per step
        DZFC = torch.empty(T, B, 3, **f32)
        DWP = torch.empty(T, B, NP, **f32)
        DPP = torch.empty(T, B, R, **f32)         # d pos_feat per step (gradient at the pos2 output)
        DZ2 = torch.empty(T, B, A, **f32)
        dc_tmp = torch.empty(B, R, **f32)
        t_BA = torch.empty(B, A, **f32)           # d gated of the current step

        sb = hip.SpkStepABwd()
        sb.B, sb.R, sb.A, sb.P = B, R, A, NP
        sb.d_gated, sb.bef, sb.diff, sb.aft, sb.w_fc = ptr(t_BA), ptr(bef), ptr(diff), ptr(aft), ptr(w_fc)
        sb.d_bef, sb.d_diff, sb.d_aft = ptr(d_bef), ptr(d_diff), ptr(d_aft)
        sb.w_p2, sb.w_wp = ptr(w_p2), ptr(w_wp)
        sb.p_v, sb.p_d, sb.lddg, sb.lddpp, sb.lddx = p, p5, NW, R, NW
        carry_l = carry_a = None                  # slabs of DCAT_{t+1} . Wcat (-> h_lang_t) and DCAT_{t+1}[att gates] . W_hh_att (-> h_att_t)
        in_c = ops.seq_in_c()
        if in_c:
            sqb = hip.SpeakerSeqBwd()
            fdesc = _seq_desc(T, B, R, A, NP, G1, We, p, p5, seeds_t, Wcat, whh_a, None, w_fc, None, None, w_wp, None, w_p2, None,
                              w_g1, None, w_g2, None, wih_l, None, bef, diff, aft, None, None, HA, CA, HL, CL, SA, SL, G1D, MW, VP,
                              None, PW, PA, AT, GT, GA)      # (kept alive until the call returns: it owns the host seed array)
            sqb.f = fdesc
            sqb.dHLo, sqb.dDPl = ptr(dHLo), ptr(dDPl)
            sqb.DCAT, sqb.DZFC, sqb.DWP, sqb.DPP, sqb.DZ2 = ptr(DCAT), ptr(DZFC), ptr(DWP), ptr(DPP), ptr(DZ2)
            sqb.d_bef, sqb.d_diff, sqb.d_aft = ptr(d_bef), ptr(d_diff), ptr(d_aft)
            # (same-box A/B, profiles/r04_ab_seq_mode.txt: the persistent backward pays 4 weight transposes + 4 small products in front and 4
            # time-batched products behind; it wins from ~32 decode steps on -- the real-data answers -- and, round 6, at <= 32 pairs for any length)
            need_p = L.ekaid_speaker_seq_bwd_persistent_workspace_bytes(ctypes.byref(sqb.f)) if (L.ekaid_get_seq_mode() == 1 and (T >= PERSISTENT_BWD_MIN_STEPS or B <= PERSISTENT_BWD_MAX_BATCH_SHORT)) else 0
            if need_p:
                # ONE persistent launch for the whole BPTT loop (csrc/persist_bwd.hip).  Its contract leaves two time-batched
                # pieces to the caller: DPP (read only by the pos2 weight gradient below) and the gate-MLP part of d feat_m
                dFm = torch.empty(3, B, G1, **f32)
                ws = hip.workspace(need_p, dev)
                check(L.ekaid_speaker_seq_bwd_persistent(ctypes.byref(sqb), ptr(dFm), ptr(ws), need_p, stream()), 'speaker_seq_bwd_persistent')
                _nn(DCAT, oG1, NW, w_g1, R, G1, DPP, 0, R, TB, R, G1)                       # DPP = d gate1x . W_g1[:, R:2R]
                for m, dfeat in enumerate((d_bef, d_diff, d_aft)):                          # d feat_m += dFm[m] . W_g1[:, 2R:]
                    _nn(dFm, m * B * G1, G1, w_g1, 2 * R, G1, dfeat, 0, A, B, A, G1, acc=True)
            else:
                need = L.ekaid_speaker_seq_workspace_bytes(ctypes.byref(sqb.f), 1)
                ws = hip.workspace(need, dev)
                check(L.ekaid_speaker_seq_bwd(ctypes.byref(sqb), ptr(ws), need, stream()), 'speaker_seq_bwd')
        for t in (() if in_c else range(T - 1, -1, -1)):
            o_prev, o_t = t * B * R, (t + 1) * B * R
            ow = t * B * NW
            sv, sd, sg = seeds_t[t]
            # 6'. language LSTM (upstream on h_lang_t = carry slabs + output path)
            csl = carry_l.cols(0) if carry_l is not None else ops.NO_SLABS
            check(L.ekaid_lstm_cell_slabs_bwd(ctypes.byref(csl), dHLo.data_ptr() + 4 * t * B * R, ptr(dc_l),
                                              SL.data_ptr() + 4 * t * B * 4 * R, CL.data_ptr() + 4 * o_prev, CL.data_ptr() + 4 * o_t, B, R,
                                              DCAT.data_ptr() + 4 * (ow + oGL), NW, ptr(dc_tmp), stream()), 'lstm_cell_slabs_bwd')
            dc_l, dc_tmp = dc_tmp, dc_l
            # d gated = d lang gates . W_ih_lang[:, We:]
            Q1 = ops.gemm_slabs(DCAT, wih_l, B, A, 4 * R, NW, We + A, 0, 0, a_off=ow + oGL, b_off=We)
            # 5'. gated = gate * att ; gate = sigmoid(W_g2 g1d + b):  d z2 = d gated * att * gate'
            s1 = Q1.cols(0)
            check(L.ekaid_gate_slabs_bwd(ctypes.byref(s1), AT.data_ptr() + 4 * t * B * A, GT.data_ptr() + 4 * t * B * A, B, A,
                                         DZ2.data_ptr() + 4 * t * B * A, ptr(t_BA), stream()), 'gate_slabs_bwd')
            Q2 = ops.gemm_slabs(DZ2, w_g2, B, G1, A, A, G1, 0, 0, a_off=t * B * A)                  # d g1d
            s2 = Q2.cols(0)
            check(L.ekaid_drop_act_slabs_bwd(ctypes.byref(s2), G1D.data_ptr() + 4 * t * B * G1, G1, B, G1, ACT_RELU, p, sg,
                                             DCAT.data_ptr() + 4 * (ow + oG1), NW, stream()), 'drop_act_slabs_bwd')
            # [d pos_feat | d att (gate-MLP path)] = d gate1x . W_g1[:, R:]   (one GEMM, N = R + A)
            Q3 = ops.gemm_slabs(DCAT, w_g1, B, R + A, G1, NW, G1, 0, 0, a_off=ow + oG1, b_off=R)
            # 4'.+2'.+1'.+3'. module mix, att-LSTM cell, POS chain: one row-local kernel
            sb.dpa_src = Q3.cols(0)
            sb.gate = GT.data_ptr() + 4 * t * B * A
            sb.mw = MW.data_ptr() + 4 * t * B * 3
            sb.dz_fc = DZFC.data_ptr() + 4 * t * B * 3
            sb.dh_carry = carry_a.cols(0) if carry_a is not None else ops.NO_SLABS
            sb.dc_in, sb.dc_out = ptr(dc_a), ptr(dc_tmp)
            sb.gate_save = SA.data_ptr() + 4 * t * B * 4 * R
            sb.c_prev, sb.c_cur = CA.data_ptr() + 4 * o_prev, CA.data_ptr() + 4 * o_t
            sb.dgates = DCAT.data_ptr() + 4 * ow
            sb.pw = PW.data_ptr() + 4 * t * B * NP
            sb.dhead = dDPl.data_ptr() + 4 * t * B * NP
            sb.vp = VP.data_ptr() + 4 * t * B * R
            sb.seed_v, sb.seed_d = sv, sd
            sb.dpp_out = DPP.data_ptr() + 4 * t * B * R
            sb.dwp_out = DWP.data_ptr() + 4 * t * B * NP
            sb.dx_out = DCAT.data_ptr() + 4 * (ow + oP1)
            check(L.ekaid_spk_step_a_bwd(ctypes.byref(sb), stream()), 'spk_step_a_bwd')
            dc_a, dc_tmp = dc_tmp, dc_a
            if t > 0:
                # 0'. what flowed out of h_att_{t-1} (att gates . W_hh) and out of h_lang_{t-1} (everything . Wcat): one launch
                carry_a, carry_l = ops.gemm_slabs2((DCAT, whh_a, B, R, 4 * R, NW, R, 0, 0, ow, 0), (DCAT, Wcat, B, R, NW, NW, R, 0, 0, ow, 0))

        # ---- time-batched weight gradients (K = T*B) and input gradients; DCAT column blocks with ld = NW
        # (fifteen mutually independent products of one layout: recorded and launched two per grid, ops.pairing_window)
        # (mode 2: ONE pass over DCAT gives the column maxima of all its blocks; one pass per right-hand operand -- the state buffers HL / HA
        # over all T + 1 steps, which bounds either of their shifted views)
        cb = (lambda t, rows, cols: ops.amax_rc(t, rows, cols, cols, want_rows=False)) if x2 else (lambda t, rows, cols: None)
        dcb, hlb, hab = cb(DCAT, TB, NW), cb(HL, TB + B, R), cb(HA, TB + B, R)
        vpb, pab, g1b, gab, dz2b = cb(VP, TB, R), cb(PA, TB, R + A), cb(G1D, TB, G1), cb(GA, TB, A), cb(DZ2, TB, A)
        with ops.pairing_window():
            d_wih_a = ops.grad_out(wih_a)
            _tn(DCAT, NW, ED, E, d_wih_a, 0, E + R, 4 * R, E, TB, db=dcb, xb=edb)
            _tn(DCAT, NW, HL, R, d_wih_a, E, E + R, 4 * R, R, TB, db=dcb, xb=hlb)
            d_whh_a = ops.grad_out(whh_a)
            d_bih_a, d_bhh_a = _tn(DCAT, NW, HA, R, d_whh_a, 0, R, 4 * R, R, TB, bias=(bih_a, bhh_a), db=dcb, xb=hab)   # the same gradient for both
            d_w_fc = ops.grad_out(w_fc)
            d_b_fc = _tn(DZFC, 3, HA, R, d_w_fc, 0, R, 3, R, TB, xoff=B * R, bias=b_fc)
            d_w_p1 = ops.grad_out(w_p1)
            d_b_p1 = _tn(DCAT, NW, HL, R, d_w_p1, 0, R, R, R, TB, doff=oP1, bias=b_p1, db=dcb, xb=hlb)
            d_w_wp = ops.grad_out(w_wp)
            d_b_wp = _tn(DWP, NP, VP, R, d_w_wp, 0, R, NP, R, TB, bias=b_wp)
            d_w_p2 = ops.grad_out(w_p2)
            d_b_p2 = _tn(DPP, R, PW, NP, d_w_p2, 0, NP, R, NP, TB, bias=b_p2)
            d_w_g1 = ops.grad_out(w_g1)
            d_b_g1 = _tn(DCAT, NW, HL, R, d_w_g1, 0, G1, G1, R, TB, doff=oG1, bias=b_g1, db=dcb, xb=hlb)
            _tn(DCAT, NW, PA, R + A, d_w_g1, R, G1, G1, R + A, TB, doff=oG1, db=dcb, xb=pab)
            d_w_g2 = ops.grad_out(w_g2)
            d_b_g2 = _tn(DZ2, A, G1D, G1, d_w_g2, 0, G1, A, G1, TB, bias=b_g2, db=dz2b, xb=g1b)                # input = dropout(relu(gate1x))
            d_wih_l = ops.grad_out(wih_l)
            _tn(DCAT, NW, Xd, We, d_wih_l, 0, We + A, 4 * R, We, TB, doff=oGL, db=dcb, xb=xdb)
            _tn(DCAT, NW, GA, A, d_wih_l, We, We + A, 4 * R, A, TB, doff=oGL, db=dcb, xb=gab)
            d_whh_l = ops.grad_out(whh_l)
            d_bih_l, d_bhh_l = _tn(DCAT, NW, HL, R, d_whh_l, 0, R, 4 * R, R, TB, doff=oGL, bias=(bih_l, bhh_l), db=dcb, xb=hlb)
        # embed path: dED = DGA . W_ih_att[:, :E]; per-step dropout backward; sum over t; relu; Linear(3A -> E)
        dED = torch.empty(T, B, E, **f32)
        rows_a = ops.amax_rc(DCAT, TB, 4 * R, NW, want_cols=False)[0] if x2 else None          # row maxima of the att-gate block only
        _nn(DCAT, 0, NW, wih_a, 0, E + R, dED, 0, E, TB, E, 4 * R, ab=rows_a, bb=_cols(wb['wih_a']))
        check(L.ekaid_dropout_steps(ptr(dED), ptr(dED), B * E, T, B * E, p, seeds['ed'], stream()), 'dropout_steps')
        d_epre = torch.empty(B, E, **f32)
        check(L.ekaid_colsum(ptr(dED), T, B * E, B * E, ptr(d_epre), stream()), 'colsum')
        d_ez = torch.empty(B, E, **f32)
        check(L.ekaid_act_bwd(ptr(d_epre), ptr(embed_pre), ptr(d_ez), B * E, ACT_RELU, stream()), 'act_bwd')
        d_w_e = ops.grad_out(w_e)
        d_b_e = _tn(d_ez, E, bef, A, d_w_e, 0, 3 * A, E, A, B, bias=b_e)
        _tn(d_ez, E, diff, A, d_w_e, A, 3 * A, E, A, B)
        _tn(d_ez, E, aft, A, d_w_e, 2 * A, 3 * A, E, A, B)
        _nn(d_ez, 0, E, w_e, 0, 3 * A, d_bef, 0, A, B, A, E, acc=True)
        _nn(d_ez, 0, E, w_e, A, 3 * A, d_diff, 0, A, B, A, E, acc=True)
        _nn(d_ez, 0, E, w_e, 2 * A, 3 * A, d_aft, 0, A, B, A, E, acc=True)
        # word path: dXd = DGL . W_ih_lang[:, :We]; dropout backward; relu + table scatter-add
        dX = torch.empty(T, B, We, **f32)
        rows_l = ops.amax_rc(DCAT, TB, 4 * R, NW, off=oGL, want_cols=False)[0] if x2 else None
        _nn(DCAT, oGL, NW, wih_l, 0, We + A, dX, 0, We, TB, We, 4 * R, ab=rows_l, bb=_cols(wb['wih_l']))
        _drop(dX, dX, TB * We, p, seeds['xt'])
        d_emb = ops.grad_out(emb, zero=True)
        ops.embedding_grad(tok_tb, emb, dX, We, 0, True, -1, d_emb)
        grads = (d_emb, d_w_e, d_b_e, d_wih_a, d_whh_a, d_bih_a, d_bhh_a, d_w_fc, d_b_fc, d_w_p1, d_b_p1, d_w_wp, d_b_wp,
                 d_w_p2, d_b_p2, d_w_g1, d_b_g1, d_w_g2, d_b_g2, d_wih_l, d_whh_l, d_bih_l, d_bhh_l, d_w_lo, d_b_lo)
        return (d_bef, d_aft, d_diff, None, None, None, None) + grads


def speaker_sequence(speaker, feat_bef, feat_aft, feat_diff, tok_tb):
    """tok_tb: (T, B) int64 teacher-forcing tokens, time-major.  Returns time-major (logp, logpos, module_weights)."""
    sd = dict(speaker.named_parameters())
    params = [sd[n] for n in PARAM_NAMES]
    out = SpeakerSeqFn.apply(feat_bef, feat_aft, feat_diff, tok_tb, speaker.drop_prob_lm, speaker.core.dropout.p,
                             speaker.training, *params)
    ops.assert_finite(out[0], 'answer decoder log-probabilities')
    return out


@torch.no_grad()
def greedy_decode(speaker, feat_bef, feat_aft, feat_diff, poll=None, _before_launch=None):
    """Greedy ``_sample`` (eval mode) through ``ekaid_speaker_greedy_steps``: ``poll`` decode steps per C-ABI call, the chosen
    tokens feed the next step on the device, the host reads the "still running" counters once per call.  ``poll=None``: the whole
    sequence in ONE call where the persistent kernel takes it (it ends by itself at the step after which every row has stopped),
    8 steps per call on the step-wise chain.  Returns
    (seq, seq_logprobs, module weights per step run), seq zeroed after the step at which every sample had finished -- what the
    reference's loop leaves behind (dynamic_speaker_change_pos.py:309-357, test_mimic.py:119-122).
    The sticky status word of the persistent kernels is also the TRAINING fault signal (the Adam guard, ``hip.seq_status_check``): a word
    that is already set when decoding begins belongs to an earlier launch and is raised here, not consumed -- only a fault of this
    decode's own launch takes the fall-back below.  ``_before_launch``: test hook, called after that entry check."""
    if hip.lib().ekaid_get_seq_mode() == 1:
        hip.seq_status_check('before greedy decoding (a fault of an earlier persistent launch)')
    if _before_launch is not None:
        _before_launch()
    sd = dict(speaker.named_parameters())
    (emb, w_e, b_e, wih_a, whh_a, bih_a, bhh_a, w_fc, b_fc, w_p1, b_p1, w_wp, b_wp, w_p2, b_p2,
     w_g1, b_g1, w_g2, b_g2, wih_l, whh_l, bih_l, bhh_l, w_lo, b_lo) = [sd[n].detach().contiguous() for n in PARAM_NAMES]
    hip.require_cuda(feat_bef, feat_aft, feat_diff, emb)
    bef, aft, diff = feat_bef.contiguous(), feat_aft.contiguous(), feat_diff.contiguous()
    B, A = bef.shape
    E, R, We, V = w_e.shape[0], whh_a.shape[1], emb.shape[1], w_lo.shape[0]
    G1, NP = w_g1.shape[0], w_wp.shape[0]
    Tm = speaker.seq_length
    dev = bef.device
    L = hip.lib()
    f32 = dict(device=dev, dtype=torch.float32)
    # loop invariants (no dropout in eval mode): relu(embed(bef | diff | aft)) and its gates of the attention LSTM
    embed_pre = torch.empty(B, E, **f32)
    _nt(bef, 0, A, w_e, 0, 3 * A, embed_pre, 0, E, B, E, A)
    _nt(diff, 0, A, w_e, A, 3 * A, embed_pre, 0, E, B, E, A, acc=True)
    _nt(aft, 0, A, w_e, 2 * A, 3 * A, embed_pre, 0, E, B, E, A, bias=b_e, acc=True, act=ACT_RELU)
    GXA = torch.empty(B, 4 * R, **f32)
    _nt(embed_pre, 0, E, wih_a, 0, E + R, GXA, 0, 4 * R, B, 4 * R, E, bias=bih_a)
    Wcat = torch.cat([wih_a[:, E:], w_p1, w_g1[:, :R], whh_l], 0).contiguous()
    HA, CA, HL, CL = ops.zeros_pool(dev, (Tm + 1, B, R), (Tm + 1, B, R), (Tm + 1, B, R), (Tm + 1, B, R))
    MW = torch.zeros(Tm, B, 3, **f32)
    one = [torch.empty(B, n, **f32) for n in (4 * R, 4 * R, 4 * R, G1, R, NP, NP, R + A, A, A, A)]   # GXL SA SL G1D VP DP PW PA AT GT GA
    GXL, SA, SL, G1D, VP, DP, PW, PA, AT, GT, GA = one
    g = hip.SpeakerGreedy()
    fdesc = _seq_desc(Tm, B, R, A, NP, G1, We, 0.0, 0.0, [(0, 0, 0)] * Tm, Wcat, whh_a, bhh_a, w_fc, b_fc, b_p1, w_wp, b_wp, w_p2, b_p2,
                      w_g1, b_g1, w_g2, b_g2, wih_l, bhh_l, bef, diff, aft, GXA, GXL, HA, CA, HL, CL, SA, SL, G1D, MW, VP,
                      DP, PW, PA, AT, GT, GA)
    g.f = fdesc
    g.emb, g.bih_l, g.w_lo, g.b_lo, g.V = ptr(emb), ptr(bih_l), ptr(w_lo), ptr(b_lo), V
    xt, logits, logp = torch.empty(B, We, **f32), torch.empty(B, V, **f32), torch.empty(B, V, **f32)
    it = torch.full((B,), 2, dtype=torch.long, device=dev)               # <bos> hard-coded as id 2 (:311)
    unfinished = torch.zeros(B, dtype=torch.int32, device=dev)
    counts = torch.zeros(Tm, dtype=torch.int32, device=dev)
    seq = torch.zeros(B, Tm, dtype=torch.long, device=dev)
    seq_logp = torch.zeros(B, Tm, **f32)
    g.xt, g.logits, g.logp, g.it, g.unfinished = ptr(xt), ptr(logits), ptr(logp), ptr(it), ptr(unfinished)
    g.seq, g.seq_logp, g.seq_len, g.n_unfinished = ptr(seq), ptr(seq_logp), Tm, ptr(counts)
    need = L.ekaid_speaker_greedy_workspace_bytes(ctypes.byref(g))
    ws = hip.workspace(need, dev)
    def run(poll):
        if poll is None:
            poll = Tm if (L.ekaid_get_seq_mode() == 1 and L.ekaid_speaker_greedy_persistent_workspace_bytes(ctypes.byref(g), Tm)) else 8
        steps_run = Tm
        for t0 in range(0, Tm, poll):
            t1 = min(Tm, t0 + poll)
            check(L.ekaid_speaker_greedy_steps(ctypes.byref(g), t0, t1, ptr(ws), need, stream()), 'speaker_greedy_steps')
            done = (counts[:t1].cpu() == 0).nonzero()                        # ONE device->host read per call
            if done.numel():                                                 # everything had finished at step `first`
                first = int(done[0])
                seq[:, first + 1:] = 0
                seq_logp[:, first + 1:] = 0
                steps_run = first + 1
                break
        return steps_run
    persistent = L.ekaid_get_seq_mode() == 1
    steps_run = run(poll)
    if persistent:
        # The host has just synchronised for `done`.  A persistent launch that abandoned itself (a seam timed out: not all of its
        # workgroups were resident) leaves `counts` at zero, which reads as "every answer ended at step 0" -- one-token garbage.  Read the
        # status word here; on a fault clear it, move this device to the launch-per-product chain for good and decode again.
        st = hip.seq_status_take()
        if st:
            import warnings
            warnings.warn('greedy_decode: a persistent sequence kernel abandoned its launch (status 0x%x); this device now uses the '
                          'step-wise chain (ekaid_set_seq_mode(0)) and the answers are decoded again' % st, RuntimeWarning)
            check(L.ekaid_set_seq_mode(0), 'set_seq_mode')
            for tns in (HA, CA, HL, CL, MW, seq, seq_logp, counts, unfinished):
                tns.zero_()
            it.fill_(2)
            steps_run = run(8)
    return seq, seq_logp, list(MW[:steps_run].unbind(0))
