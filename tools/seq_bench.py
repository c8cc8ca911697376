#!/usr/bin/env python3
"""The recurrent part of the answer decoder (ekaid_speaker_seq_fwd) at the production widths on random data: the step-wise chain
(8 launches per decode step) against the persistent kernel (one launch per sequence), every saved tensor compared, both timed.

    python tools/seq_bench.py [--batch 64] [--steps 16] [--train] [--iters 20]

Also imported by tests/test_hip_persist.py (``problem`` / ``run``)."""
import argparse
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ekaid_amd import hip  # noqa: E402

R, A, P, WE = 512, 1024, 16, 300
G1 = 2 * R + A
NW = 9 * R + G1
OUT_KEYS = ('HA', 'CA', 'HL', 'CL', 'SA', 'SL', 'G1D', 'MW', 'VP', 'DP', 'PW', 'PA', 'AT', 'GT', 'GA')


def problem(B, T, seed=0, train=False, dev='cuda'):
    """Random weights of the decoder's shapes (scaled like the formula init: unit-variance pre-activations) and inputs."""
    g = torch.Generator(device='cpu').manual_seed(seed)

    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g) * scale).to(dev)
    w = {
        'Wcat': rnd(NW, R, scale=R ** -0.5), 'whh_a': rnd(4 * R, R, scale=R ** -0.5), 'bhh_a': rnd(4 * R, scale=0.1),
        'w_fc': rnd(3, R, scale=R ** -0.5), 'b_fc': rnd(3, scale=0.1), 'b_p1': rnd(R, scale=0.1),
        'w_wp': rnd(P, R, scale=R ** -0.5), 'b_wp': rnd(P, scale=0.1), 'w_p2': rnd(R, P, scale=P ** -0.5), 'b_p2': rnd(R, scale=0.1),
        'w_g1': rnd(G1, G1, scale=G1 ** -0.5), 'b_g1': rnd(G1, scale=0.1), 'w_g2': rnd(A, G1, scale=G1 ** -0.5), 'b_g2': rnd(A, scale=0.1),
        'wih_l': rnd(4 * R, WE + A, scale=(WE + A) ** -0.5), 'bhh_l': rnd(4 * R, scale=0.1),
        'bef': rnd(B, A).abs(), 'diff': rnd(B, A), 'aft': rnd(B, A).abs(),
        'GXA': rnd(T, B, 4 * R, scale=0.5), 'GXL': rnd(T, B, 4 * R, scale=0.5),
    }
    init = {k: rnd(B, R, scale=0.3) for k in ('HA', 'CA', 'HL', 'CL')}
    seeds = [(1000 + 3 * t, 1001 + 3 * t, 1002 + 3 * t) for t in range(T)]
    return {'B': B, 'T': T, 'w': w, 'init': init, 'seeds': seeds, 'p': (0.5, 0.5) if train else (0.0, 0.0), 'dev': dev}


def seq_status():
    """The per-device status word of the persistent kernels (0 = fine), cleared after reading."""
    v = ctypes.c_uint(0)
    hip.check(hip.lib().ekaid_seq_status_read(ctypes.byref(v), hip.stream()), 'seq_status_read')
    if v.value:
        hip.lib().ekaid_seq_status_clear(hip.stream())
    return v.value


def buffers(pr):
    B, T, dev = pr['B'], pr['T'], pr['dev']
    f32 = dict(device=dev, dtype=torch.float32)
    out = {}
    for k in ('HA', 'CA', 'HL', 'CL'):
        out[k] = torch.zeros(T + 1, B, R, **f32)
        out[k][0] = pr['init'][k]
    for k, n in (('SA', 4 * R), ('SL', 4 * R), ('G1D', G1), ('MW', 3), ('VP', R), ('DP', P), ('PW', P), ('PA', R + A), ('AT', A), ('GT', A),
                 ('GA', A)):
        out[k] = torch.full((T, B, n), float('nan'), **f32)
    return out


def descriptor(pr, out):
    sq = hip.SpeakerSeq()
    sq.T, sq.B, sq.R, sq.A, sq.P, sq.G1, sq.We = pr['T'], pr['B'], R, A, P, G1, WE
    sq.p_lm, sq.p_pos = pr['p']
    flat = (ctypes.c_uint64 * (3 * pr['T']))(*[s for trip in pr['seeds'] for s in trip])
    sq._keep = flat
    sq.seeds = ctypes.cast(flat, ctypes.c_void_p)
    for k, _ in hip.SpeakerSeq._fields_[10:]:
        setattr(sq, k, (pr['w'][k] if k in pr['w'] else out[k]).data_ptr())
    return sq


def run(pr, mode, iters=0):
    """mode 0: step-wise chain, 1: persistent kernel.  Returns (outputs, ms per call or None, status word)."""
    L = hip.lib()
    old = L.ekaid_get_seq_mode()
    hip.check(L.ekaid_set_seq_mode(mode), 'set_seq_mode')
    try:
        out = buffers(pr)
        sq = descriptor(pr, out)
        # mode 1 calls the persistent entry itself (ekaid_speaker_seq_fwd only routes sequences of >= 8 steps to it)
        fwd = L.ekaid_speaker_seq_fwd_persistent if mode == 1 else L.ekaid_speaker_seq_fwd
        need = L.ekaid_speaker_seq_persistent_workspace_bytes(ctypes.byref(sq)) if mode == 1 else L.ekaid_speaker_seq_workspace_bytes(ctypes.byref(sq), 0)
        assert need > 0, 'unsupported shape'
        ws = torch.empty(max(need, 256), dtype=torch.uint8, device=pr['dev'])
        st = hip.stream()
        hip.check(fwd(ctypes.byref(sq), ws.data_ptr(), need, st), 'speaker_seq_fwd')
        torch.cuda.synchronize()
        status = 0
        if mode == 1:
            status = seq_status()
        ms = None
        if iters:
            for _ in range(3):
                hip.check(fwd(ctypes.byref(sq), ws.data_ptr(), need, st), 'speaker_seq_fwd')
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                hip.check(fwd(ctypes.byref(sq), ws.data_ptr(), need, st), 'speaker_seq_fwd')
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / iters
        if mode == 1 and os.environ.get('EKAID_PERSIST_TRACE') == '1':
            G = 256                                      # one workgroup per CU; the stamps are the last region of the workspace
            nb = (G * pr['T'] * 128 + 255) // 256 * 256
            off = L.ekaid_speaker_seq_persistent_workspace_bytes(ctypes.byref(sq)) - nb
            run.trace = ws[off:off + G * pr['T'] * 128].view(torch.int64).view(G, pr['T'], 16).cpu()
        return out, ms, status
    finally:
        L.ekaid_set_seq_mode(old)


def compare(a, b):
    """max |a - b| per saved tensor, relative to the tensor's scale."""
    res = {}
    for k in OUT_KEYS:
        x, y = a[k].double(), b[k].double()
        if k in ('HA', 'CA', 'HL', 'CL'):
            x, y = x[1:], y[1:]
        res[k] = float((x - y).abs().max() / x.abs().max().clamp_min(1e-30)) if torch.isfinite(y).all() else float('inf')
    return res


BWD_KEYS = ('DCAT', 'DZFC', 'DWP', 'DZ2', 'DPP', 'd_bef', 'd_diff', 'd_aft')


def run_bwd(pr, fwd_out, mode, iters=0, seed=7):
    """Backward over the saved tensors ``fwd_out`` with random upstream gradients.  mode 0: ekaid_speaker_seq_bwd (step-wise chain);
    1: ekaid_speaker_seq_bwd_persistent + what its contract leaves to the caller (DPP as one time-batched product, the gate-MLP
    part of the feature gradients from dFm).  Returns (outputs, ms per call, status)."""
    L = hip.lib()
    B, T, dev = pr['B'], pr['T'], pr['dev']
    f32 = dict(device=dev, dtype=torch.float32)
    g = torch.Generator(device='cpu').manual_seed(seed)
    dHLo = (torch.randn(T, B, R, generator=g) * 0.1).to(dev)
    dDPl = (torch.randn(T, B, P, generator=g) * 0.1).to(dev)
    out = {'DCAT': torch.full((T, B, NW), float('nan'), **f32), 'DZFC': torch.full((T, B, 3), float('nan'), **f32),
           'DWP': torch.full((T, B, P), float('nan'), **f32), 'DPP': torch.full((T, B, R), float('nan'), **f32),
           'DZ2': torch.full((T, B, A), float('nan'), **f32)}
    sqb = hip.SpeakerSeqBwd()
    sqb.f = descriptor(pr, fwd_out)
    sqb.dHLo, sqb.dDPl = dHLo.data_ptr(), dDPl.data_ptr()
    st = hip.stream()
    w_g1 = pr['w']['w_g1']

    def call(post=True):
        for k in ('d_bef', 'd_diff', 'd_aft'):
            out[k] = torch.zeros(B, A, **f32)
        for k in ('DCAT', 'DZFC', 'DWP', 'DPP', 'DZ2', 'd_bef', 'd_diff', 'd_aft'):
            setattr(sqb, k, out[k].data_ptr())
        if mode == 0:
            hip.check(L.ekaid_speaker_seq_bwd(ctypes.byref(sqb), ws.data_ptr(), need, st), 'speaker_seq_bwd')
        else:
            hip.check(L.ekaid_speaker_seq_bwd_persistent(ctypes.byref(sqb), dFm.data_ptr(), ws.data_ptr(), need, st), 'speaker_seq_bwd_persistent')
            if not post:
                return
            # the caller's share: DPP = DCAT[:, gate1x block] . W_g1[:, R:2R]; d feat_m += dFm[m] . W_g1[:, 2R:]
            out['DPP'] = (out['DCAT'][:, :, 5 * R:5 * R + G1].reshape(T * B, G1) @ w_g1[:, R:2 * R]).view(T, B, R)
            for m, k in enumerate(('d_bef', 'd_diff', 'd_aft')):
                out[k] += dFm[m] @ w_g1[:, 2 * R:]
    if mode == 0:
        need = L.ekaid_speaker_seq_workspace_bytes(ctypes.byref(sqb.f), 1)
    else:
        need = L.ekaid_speaker_seq_bwd_persistent_workspace_bytes(ctypes.byref(sqb.f))
        assert need > 0, 'persistent backward does not support this shape'
    ws = torch.empty(max(need, 256), dtype=torch.uint8, device=dev)
    dFm = torch.full((3, B, G1), float('nan'), **f32)
    call()
    torch.cuda.synchronize()
    status = 0
    if mode == 1:
        status = seq_status()
        if os.environ.get('EKAID_PERSIST_TRACE') == '1':
            nb = (256 * T * 128 + 255) // 256 * 256
            run_bwd.trace = ws[need - nb:need - nb + 256 * T * 128].view(torch.int64).view(256, T, 16).cpu()
    ms = None
    if iters:
        for _ in range(3):
            call(False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            call(False)          # (the caller's four time-batched products are not part of the sequence call: ~0.1 ms per backward)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        call()
        torch.cuda.synchronize()
    return out, ms, status


def compare_bwd(a, b):
    res = {}
    for k in BWD_KEYS:
        x, y = a[k].double(), b[k].double()
        res[k] = float((x - y).abs().max() / x.abs().max().clamp_min(1e-30)) if torch.isfinite(y).all() else float('inf')
    return res


def show_trace(tr, B, names):
    d = tr.double()
    spans = torch.zeros(d.shape[0], d.shape[1], 16)
    for k in range(15):
        spans[:, :, k] = (d[:, :, k + 1] - d[:, :, k]) / 100.0
    spans[:, :-1, 15] = (d[:, 1:, 0] - d[:, :-1, 15]) / 100.0
    med = spans[:, 1:-1].median(1).values if d.shape[1] > 2 else spans.mean(1)
    step_us = ((d[:, -1, 15] - d[:, 0, 0]) / 100.0 / d.shape[1]).median()
    print('trace B=%d: %.1f us per step; per span: median over workgroups (max)' % (B, float(step_us)), file=sys.stderr)
    print('  ' + '  '.join('%s %.1f(%.1f)' % (n, float(med[:, k].median()), float(med[:, k].max())) for k, n in enumerate(names)), file=sys.stderr)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, nargs='+', default=[64])
    ap.add_argument('--steps', type=int, default=16)
    ap.add_argument('--train', action='store_true')
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--bwd', action='store_true', help='the backward sequence instead of the forward one')
    args = ap.parse_args()
    hip.load()
    for B in args.batch:
        pr = problem(B, args.steps, train=args.train)
        if args.bwd:
            fwd, _, _ = run(pr, 0)
            ref, ms0, _ = run_bwd(pr, fwd, 0, args.iters)
            got, ms1, status = run_bwd(pr, fwd, 1, args.iters)
            err = compare_bwd(ref, got)
            tr = getattr(run_bwd, 'trace', None)
            if tr is not None:
                show_trace(tr, B, ['P1', 'arr1', 'wait1', 'P2', 'arr2', 'carryL', 'P3(+wait2)', 'P3', 'arr3', 'wait3', 'P4', 'arr4', 'carryG', 'P5(+wait4)', 'P6', 'next'])
            print(json.dumps({'bwd': True, 'batch': B, 'steps': args.steps, 'train': args.train, 'status': status,
                              'stepwise_us_per_step': round(ms0 / args.steps * 1e3, 2), 'persistent_us_per_step': round(ms1 / args.steps * 1e3, 2),
                              'max_rel_err': round(max(err.values()), 9), 'worst': max(err, key=err.get), 'err': {k: round(v, 9) for k, v in err.items()}}))
            continue
        ref, ms0, _ = run(pr, 0, args.iters)
        got, ms1, status = run(pr, 1, args.iters)
        err = compare(ref, got)
        tr = getattr(run, 'trace', None)
        if tr is not None:
            # per-phase time (us) of the last call: stamps are 100 MHz ticks; spans between consecutive stamps, median over steps,
            # for the slowest / median / fastest workgroup
            names = ['waitE', 'A', 'arrA', 'waitA', 'B', 'arrB', 'shadow+waitB', 'C', 'arrC', 'waitC', 'D', 'arrD', 'waitD', 'E', 'arrE', 'next']
            idx = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15]
            d = tr.double()
            spans = torch.zeros(d.shape[0], d.shape[1], 16)
            for k in range(15):
                spans[:, :, k] = (d[:, :, idx[k + 1]] - d[:, :, idx[k]]) / 100.0
            spans[:, :-1, 15] = (d[:, 1:, 0] - d[:, :-1, 15]) / 100.0
            med = spans[:, 1:-1].median(1).values if d.shape[1] > 2 else spans.mean(1)
            step_us = ((d[:, -1, 15] - d[:, 0, 0]) / 100.0 / d.shape[1]).median()
            print('trace B=%d: %.1f us per step; per span: median over workgroups (max)' % (B, float(step_us)), file=sys.stderr)
            print('  ' + '  '.join('%s %.1f(%.1f)' % (n, float(med[:, k].median()), float(med[:, k].max())) for k, n in enumerate(names)), file=sys.stderr)
        print(json.dumps({'batch': B, 'steps': args.steps, 'train': args.train, 'status': status,
                          'stepwise_ms': round(ms0, 4), 'persistent_ms': round(ms1, 4),
                          'stepwise_us_per_step': round(ms0 / args.steps * 1e3, 2), 'persistent_us_per_step': round(ms1 / args.steps * 1e3, 2),
                          'max_rel_err': round(max(err.values()), 9), 'worst': max(err, key=err.get), 'err': {k: round(v, 9) for k, v in err.items()}}))


if __name__ == '__main__':
    main()
