"""The persistent decoder kernel (csrc/persist.hip: the whole teacher-forced time loop in ONE launch) against the step-wise chain
(csrc/sequence.hip: 8 launches per decode step) -- the same C-ABI entry, ekaid_speaker_seq_fwd, under ekaid_set_seq_mode(1 / 0).

The two modes compute the same f32 function in different summation orders (full-K 16x16 MFMA jobs vs split-K slabs on the bf16-split
pipe; gate-MLP layer 1 through its linear maps instead of a K = 1536 product), so they agree to f32 rounding, not bit for bit:
every saved tensor of the contract is compared at 1e-5 of its scale (measured: <= 1.3e-6).  Dropout masks are a pure function of
(seed, element index) and must coincide exactly.   ref: dynamic_speaker_change_pos.py:94-131, 196-222."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _sb():
    import seq_bench
    from ekaid_amd import hip
    hip.load()
    return seq_bench


@pytest.mark.parametrize('B,T', [(5, 4), (16, 5), (37, 4), (64, 6), (130, 4), (256, 4), (64, 90), (1, 96)])
@pytest.mark.parametrize('train', [False, True])
def test_persistent_forward_equals_the_stepwise_chain(B, T, train):
    """Every tensor of the saved-tensor contract (states, gate activations, module / POS weights, gate-MLP activations, attended and
    gated features), ragged batch sizes included (rows past the batch inside the last 16-row block are never stored)."""
    sb = _sb()
    pr = sb.problem(B, T, seed=B * 31 + T, train=train)
    ref, _, _ = sb.run(pr, 0)
    got, _, status = sb.run(pr, 1)
    assert status == 0, 'a seam of the persistent launch did not open (status 0x%x)' % status
    err = sb.compare(ref, got)
    assert max(err.values()) <= TOL, err
    if train:       # the same elements were dropped: the masks depend on (seed, index) only
        for k in ('G1D', 'VP', 'DP'):
            assert torch.equal(ref[k] == 0, got[k] == 0), 'dropout mask of %s differs between the modes' % k


def test_persistent_forward_is_deterministic_and_replayable():
    """Two launches on the same workspace give the same bits (every output element has one owner and a fixed summation order; the seam
    counters are re-zeroed by the memset in front of every launch), also when replayed from a captured hipGraph."""
    import ctypes
    sb = _sb()
    from ekaid_amd import hip
    L = hip.lib()
    pr = sb.problem(48, 6, seed=3, train=True)
    a, _, s1 = sb.run(pr, 1)
    b, _, s2 = sb.run(pr, 1)
    assert s1 == 0 and s2 == 0
    for k in sb.OUT_KEYS:
        assert torch.equal(a[k], b[k]), k
    # hipGraph: memset node + the five small products + the persistent kernel, replayed three times
    old = L.ekaid_get_seq_mode()
    hip.check(L.ekaid_set_seq_mode(1), 'set_seq_mode')
    try:
        out = sb.buffers(pr)
        sq = sb.descriptor(pr, out)
        need = L.ekaid_speaker_seq_persistent_workspace_bytes(ctypes.byref(sq))
        ws = torch.empty(need, dtype=torch.uint8, device='cuda')
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            hip.check(L.ekaid_speaker_seq_fwd_persistent(ctypes.byref(sq), ws.data_ptr(), need, hip.stream()), 'warm-up')
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            hip.check(L.ekaid_speaker_seq_fwd_persistent(ctypes.byref(sq), ws.data_ptr(), need, hip.stream()), 'capture')
        for _ in range(3):
            for k in ('SA', 'SL', 'G1D', 'GA'):
                out[k].fill_(float('nan'))
            g.replay()
            torch.cuda.synchronize()
            for k in sb.OUT_KEYS:
                assert torch.equal(a[k], out[k]), 'replay differs in %s' % k
    finally:
        L.ekaid_set_seq_mode(old)


def test_unsupported_shapes_keep_the_stepwise_chain():
    """Mode 1 is "persistent where supported": other widths (the small test configurations) and single steps take the step-wise chain,
    and the explicit persistent entry refuses them loudly."""
    import ctypes
    sb = _sb()
    from ekaid_amd import hip
    L = hip.lib()
    pr = sb.problem(8, 4)
    out = sb.buffers(pr)
    sq = sb.descriptor(pr, out)
    assert L.ekaid_speaker_seq_persistent_workspace_bytes(ctypes.byref(sq)) > 0
    sq.R = 256
    assert L.ekaid_speaker_seq_persistent_workspace_bytes(ctypes.byref(sq)) == 0
    ws = torch.empty(1 << 20, dtype=torch.uint8, device='cuda')
    assert L.ekaid_speaker_seq_fwd_persistent(ctypes.byref(sq), ws.data_ptr(), ws.numel(), hip.stream()) == hip.ERR_UNSUPPORTED
    assert 'unsupported shape' in hip.last_error()


@pytest.mark.parametrize('train', [False, True])
def test_decoder_through_the_model_agrees_between_the_sequence_modes(train, monkeypatch):
    """DynamicSpeaker._forward + loss + backward at the production widths under both sequence modes: log-probs, module weights and
    every parameter gradient (mode 1 runs the persistent forward AND the persistent backward; both pair with either counterpart: same
    saved tensors)."""
    from common import build_models
    from ekaid_amd import hip, ops, speaker_seq
    from ekaid_amd.config import bench_cfg
    hip.load()
    L = hip.lib()
    monkeypatch.setattr(speaker_seq, 'PERSISTENT_BWD_MIN_STEPS', 4)      # (24 pairs run it by default since round 6; pinned here for any routing)
    cfg = bench_cfg(D=256, seq_length=12)
    cd, sp = build_models(cfg, 147, 148, 'cuda', train=train)
    g = torch.Generator().manual_seed(11)
    B, T, A = 24, 9, cfg.model.speaker.input_dim
    bef, aft, diff = [torch.randn(B, A, generator=g).cuda().requires_grad_(True) for _ in range(3)]
    seq = torch.zeros(B, 13, dtype=torch.long)
    seq[:, 0] = 1
    seq[:, 1:T] = torch.randint(2, 147, (B, T - 1), generator=g)
    seq = seq.cuda()
    res = {}
    old = L.ekaid_get_seq_mode()
    try:
        for mode in (0, 1):
            hip.check(L.ekaid_set_seq_mode(mode), 'set_seq_mode')
            ops.manual_seed(99)
            sp.zero_grad()
            for x in (bef, aft, diff):
                x.grad = None
            logp, logpos = sp._forward(bef, aft, diff, seq, num_steps=T)
            (logp[:, :T].sum() * 1e-2 + logpos[:, :T].sum() * 1e-3).backward()
            res[mode] = (logp.detach().clone(), torch.stack(sp.module_weights, 1).detach().clone(),
                         {k: p.grad.detach().clone() for k, p in sp.named_parameters() if p.grad is not None},
                         [x.grad.detach().clone() for x in (bef, aft, diff)])
    finally:
        L.ekaid_set_seq_mode(old)

    def close(x, y, what, tol=2e-5):
        scale = float(x.abs().max().clamp_min(1e-12))
        err = float((x - y).abs().max()) / scale
        assert err <= tol, '%s: %.3e of its scale' % (what, err)
    close(res[0][0], res[1][0], 'log-probs')
    close(res[0][1], res[1][1], 'module weights')
    assert torch.equal(res[0][0].argmax(-1), res[1][0].argmax(-1))
    assert res[0][2].keys() == res[1][2].keys()
    for k in res[0][2]:
        close(res[0][2][k], res[1][2][k], 'grad ' + k, 1e-4)
    for x, y, n in zip(res[0][3], res[1][3], ('bef', 'aft', 'diff')):
        close(x, y, 'grad feat_' + n, 1e-4)


@pytest.mark.parametrize('B,poll,bias0', [(7, 8, 0.5), (64, 8, 0.5), (64, None, 0.5), (150, 5, 0.5), (256, None, 0.5), (64, None, 60.0), (20, None, 6.0)])
def test_greedy_decoding_agrees_between_the_sequence_modes(B, poll, bias0):
    """speaker_seq.greedy_decode at the production widths: ranges of decode steps as ONE persistent launch (tokens, stop flags and the
    next word's embedding stay on the device between steps) against the step-wise chain: identical token ids and stop positions,
    log-probs and module weights <= 1e-5, the step-wise tail of a range shorter than 4 steps (poll = 5 over 12 steps) in between, and
    the default (poll = None): the whole sequence in one launch that ends by itself once every row has stopped."""
    from common import build_models
    from ekaid_amd import hip, speaker_seq
    from ekaid_amd.config import bench_cfg
    hip.load()
    L = hip.lib()
    cfg = bench_cfg(D=256, seq_length=12)
    cd, sp = build_models(cfg, 147, 148, 'cuda', train=False)
    g = torch.Generator().manual_seed(5 + B)
    A = cfg.model.speaker.input_dim
    bef, aft, diff = [torch.randn(B, A, generator=g).cuda() for _ in range(3)]
    with torch.no_grad():                                                  # random-init logits are nearly flat: spread them so that rows stop at
        sp.logit.weight.mul_(8.0)                                          # different steps and ties within rounding do not decide the ids
        sp.logit.bias[0] += bias0                                          # (60: every row stops at step 1, the launch ends by itself at step 2)
    res = {}
    old = L.ekaid_get_seq_mode()
    try:
        for mode in (0, 1):
            hip.check(L.ekaid_set_seq_mode(mode), 'set_seq_mode')
            seq, lp, mw = speaker_seq.greedy_decode(sp, bef, aft, diff, poll=poll)
            torch.cuda.synchronize()
            hip.seq_status_check('greedy decode')
            res[mode] = (seq.clone(), lp.clone(), torch.stack(mw, 1).clone())
    finally:
        L.ekaid_set_seq_mode(old)
    assert torch.equal(res[0][0], res[1][0])
    assert res[0][2].shape == res[1][2].shape
    if bias0 > 50:
        assert res[1][2].shape[1] == 2 and int((res[1][0] != 0).sum()) == B
    assert float((res[0][1] - res[1][1]).abs().max()) <= 1e-5 * max(1.0, float(res[0][1].abs().max()))
    assert float((res[0][2] - res[1][2]).abs().max()) <= 1e-5


@pytest.mark.parametrize('B,T', [(5, 4), (16, 5), (37, 4), (64, 6), (130, 4), (256, 4), (64, 90), (1, 96)])
@pytest.mark.parametrize('train', [False, True])
def test_persistent_backward_equals_the_stepwise_chain(B, T, train):
    """ekaid_speaker_seq_bwd_persistent (one launch for the whole BPTT loop; the d [pos_feat | att] product replaced by its linear
    maps) + the caller's share of its contract (DPP as one time-batched product, dFm . W_att) against ekaid_speaker_seq_bwd, on the
    saved tensors of one forward and random upstream gradients: every output of the backward contract."""
    sb = _sb()
    pr = sb.problem(B, T, seed=B * 17 + T, train=train)
    fwd, _, _ = sb.run(pr, 0)
    ref, _, _ = sb.run_bwd(pr, fwd, 0)
    got, _, status = sb.run_bwd(pr, fwd, 1)
    assert status == 0, 'a seam of the persistent backward did not open (status 0x%x)' % status
    err = sb.compare_bwd(ref, got)
    assert max(err.values()) <= TOL, err
    again, _, _ = sb.run_bwd(pr, fwd, 1)
    for k in ('DCAT', 'DZFC', 'DWP', 'DZ2'):
        assert torch.equal(got[k], again[k]), 'persistent backward is not deterministic in %s' % k


@pytest.mark.parametrize('B,T', [(5, 4), (32, 20), (64, 20), (200, 6), (256, 20)])
def test_persistent_gru_equals_the_stepwise_chain(B, T):
    """ekaid_gru_seq_fwd_persistent / _bwd_persistent (one launch per direction, one seam per step) against the step-wise chain
    (ekaid_gru_seq_fwd / _bwd under ekaid_set_seq_mode(0)): states, saved gate activations, both gradient outputs; bit-identical
    relaunch.   ref: language_model.py:106-115 (nn.GRU, gate order r, z, n)."""
    from ekaid_amd import hip
    hip.load()
    L = hip.lib()
    H = 1024
    g = torch.Generator().manual_seed(B * 7 + T)
    f32 = dict(device='cuda', dtype=torch.float32)
    gi = (torch.randn(B, T, 3 * H, generator=g) * 0.7).cuda()
    w_hh = (torch.randn(3 * H, H, generator=g) * H ** -0.5).cuda()
    b_hh = (torch.randn(3 * H, generator=g) * 0.1).cuda()
    h0 = (torch.randn(B, H, generator=g) * 0.3).cuda()
    dout = (torch.randn(B, T, H, generator=g) * 0.1).cuda()
    st = hip.stream()

    def run(persistent):
        hall = torch.full((T + 1, B, H), float('nan'), **f32)
        hall[0] = h0
        save = torch.full((T, B, 4 * H), float('nan'), **f32)
        dgi = torch.full((B, T, 3 * H), float('nan'), **f32)
        dgh = torch.full((T, B, 3 * H), float('nan'), **f32)
        if persistent:
            nf, nb = L.ekaid_gru_seq_persistent_workspace_bytes(B, T, H, 0), L.ekaid_gru_seq_persistent_workspace_bytes(B, T, H, 1)
            assert nf > 0 and nb > 0
            ws = torch.empty(max(nf, nb), dtype=torch.uint8, device='cuda')
            hip.check(L.ekaid_gru_seq_fwd_persistent(gi.data_ptr(), w_hh.data_ptr(), b_hh.data_ptr(), B, T, H, hall.data_ptr(), save.data_ptr(),
                                                     ws.data_ptr(), nf, st), 'gru_seq_fwd_persistent')
            hip.check(L.ekaid_gru_seq_bwd_persistent(dout.data_ptr(), w_hh.data_ptr(), hall.data_ptr(), save.data_ptr(), B, T, H, dgi.data_ptr(),
                                                     dgh.data_ptr(), ws.data_ptr(), nb, st), 'gru_seq_bwd_persistent')
            torch.cuda.synchronize()
            hip.seq_status_check('persistent GRU')
        else:
            old = L.ekaid_get_seq_mode()
            hip.check(L.ekaid_set_seq_mode(0), 'set_seq_mode')
            try:
                nf, nb = L.ekaid_gru_seq_workspace_bytes(B, T, H, 0), L.ekaid_gru_seq_workspace_bytes(B, T, H, 1)
                ws = torch.empty(max(nf, nb, 256), dtype=torch.uint8, device='cuda')
                hip.check(L.ekaid_gru_seq_fwd(gi.data_ptr(), w_hh.data_ptr(), b_hh.data_ptr(), B, T, H, hall.data_ptr(), save.data_ptr(),
                                              ws.data_ptr(), nf, st), 'gru_seq_fwd')
                hip.check(L.ekaid_gru_seq_bwd(dout.data_ptr(), w_hh.data_ptr(), hall.data_ptr(), save.data_ptr(), B, T, H, dgi.data_ptr(),
                                              dgh.data_ptr(), ws.data_ptr(), nb, st), 'gru_seq_bwd')
            finally:
                L.ekaid_set_seq_mode(old)
        torch.cuda.synchronize()
        return {'hall': hall, 'save': save, 'dgi': dgi, 'dgh': dgh}
    ref, got, again = run(False), run(True), run(True)
    for k in ref:
        x, y = ref[k].double(), got[k].double()
        assert torch.isfinite(y).all(), k
        err = float((x - y).abs().max() / x.abs().max())
        assert err <= TOL, '%s: %.3e' % (k, err)
        assert torch.equal(got[k], again[k]), 'persistent GRU is not deterministic in %s' % k


def test_greedy_decode_recovers_from_an_abandoned_persistent_launch():
    """an abandoned persistent launch (seam time-out; forced here by raising the sticky status word first, which makes every
    workgroup leave at its first seam) used to leave `counts` at zero -- read as "every answer ended at step 0" -- and a one-token
    garbage answer was returned silently.  Now greedy_decode reads the word where it synchronises anyway, clears it, moves the device to
    the launch-per-product chain and decodes again: same ids as the step-wise chain, a RuntimeWarning, sequence mode 0 afterwards."""
    import ctypes
    import warnings
    from common import build_models
    from ekaid_amd import hip, speaker_seq
    from ekaid_amd.config import bench_cfg
    hip.load()
    L = hip.lib()
    cfg = bench_cfg(D=256, seq_length=12)
    cd, sp = build_models(cfg, 147, 148, 'cuda', train=False)
    g = torch.Generator().manual_seed(11)
    A = cfg.model.speaker.input_dim
    bef, aft, diff = [torch.randn(16, A, generator=g).cuda() for _ in range(3)]
    with torch.no_grad():
        sp.logit.weight.mul_(8.0)
    old = L.ekaid_get_seq_mode()
    rt = ctypes.CDLL('libamdhip64.so')
    try:
        hip.check(L.ekaid_set_seq_mode(0), 'set_seq_mode')
        want = [t.clone() for t in speaker_seq.greedy_decode(sp, bef, aft, diff)[:2]]
        hip.check(L.ekaid_set_seq_mode(1), 'set_seq_mode')
        torch.cuda.synchronize()
        word = ctypes.c_void_p(L.ekaid_speaker_seq_persistent_status_ptr(None))
        val = ctypes.c_uint(0x104)

        def raise_word():
            assert rt.hipMemcpy(word, ctypes.byref(val), 4, 1) == 0
        # a word that is ALREADY set when decoding begins is an earlier launch's fault (training): raised, not consumed
        raise_word()
        with pytest.raises(hip.EkaidError, match='earlier persistent launch'):
            speaker_seq.greedy_decode(sp, bef, aft, diff)
        L.ekaid_seq_status_clear(hip.stream())
        torch.cuda.synchronize()
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter('always')
            seq, lp, _ = speaker_seq.greedy_decode(sp, bef, aft, diff, _before_launch=raise_word)     # ... one raised by this decode's own launch: fall-back
        assert any('abandoned' in str(x.message) for x in w)
        assert L.ekaid_get_seq_mode() == 0
        hip.seq_status_check('after the fallback')                 # cleared
        assert torch.equal(seq, want[0]) and float((lp - want[1]).abs().max()) <= 1e-5
        assert int((seq != 0).sum()) > 16, 'answers longer than one token expected'
    finally:
        L.ekaid_seq_status_clear(hip.stream())
        L.ekaid_set_seq_mode(old)
