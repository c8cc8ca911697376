"""GPU tests of this round's host-side additions: the real-data path end to end (generated HDF5 dataset of the real
shapes -> reader -> compact device feed -> HIP path vs the CPU oracle; train_mimic.py / test_mimic.py without
--synthetic), resume across snapshot / epoch boundaries, hipGraph replay vs eager, the RCCL code path on one GPU."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope='module')
def data_dir(tmp_path_factory):
    from ekaid_amd.datasets.generate import make_dataset
    return make_dataset(str(tmp_path_factory.mktemp('mimic')), n_images=24, n_pairs=90, nodes=52, D=1024, seed=3)


def test_real_shape_batch_reader_to_hip_path_vs_oracle(data_dir):
    """N = 52 regions, D = 1024, ragged answers: the batch the device feed delivers (int8 adjacency, pinned staging, side
    stream) equals the collated reference-layout batch, and the HIP path on it matches the CPU oracle on the
    reference-layout batch (log-probs <= 1e-3 of their range, identical argmax)."""
    from common import build_models, oracle_params
    from ekaid_amd.config import mimic_cfg
    from ekaid_amd.datasets import create_dataset
    from ekaid_amd.datasets.generate import point_cfg_at
    from ekaid_amd.datasets.rcc_dataset_pos_mimic import DevicePrefetcher, batch_dict
    from ekaid_amd.engine import forward_loss
    from ekaid_amd.models.dynamic_speaker_change_pos import DynamicSpeaker
    from oracle import ekaid_oracle as O
    cfg = point_cfg_at(mimic_cfg(), data_dir)
    cfg.data.feature_mode = 'both'
    cfg.data.test.batch_size = 6
    cfg.model.speaker.vocab_size = 148
    ds, loader = create_dataset(cfg, 'test')
    ref = batch_dict(ds[[0, 1, 2, 3, 4, 5]])
    dev_batch = next(iter(DevicePrefetcher(loader, 'cuda')))
    N = 52
    assert dev_batch['input_1'].shape == (6, N, 1024) and dev_batch['d_adj'].dtype == torch.int8
    for k in ('input_1', 'input_2', 'labels', 'masks', 'question', 'd_bb', 'q_bb', 'pos', 'pair_index'):
        assert torch.equal(dev_batch[k].cpu(), ref[k]), k
    for k in ('d_adj', 'q_adj', 'd_sem', 'q_sem'):
        assert torch.equal(dev_batch[k].cpu().double(), ref[k][:, :N, :N]), k
    lens = (ref['labels'] != 0).sum(1)
    assert lens.min() < lens.max(), 'answers must be ragged for this test'
    cd, sp = build_models(cfg, 147, 148, 'cuda')
    T = DynamicSpeaker.count_steps(ref['labels'], cfg.model.speaker.seq_length)
    with torch.no_grad():
        loss, sp_loss, outs, logp = forward_loss(cfg, cd, sp, dev_batch, T, 'all', 'mode2')
        lo, res = O.train_loss(oracle_params(cd), oracle_params(sp), cfg, ref, 'mode2', 'all')
    got, want = logp[:, :T].cpu(), res['logp'][:, :T]
    err = float((got - want).abs().max() / want.abs().max())
    assert err < 1e-3, err
    assert torch.equal(got.argmax(-1), want.argmax(-1))
    assert abs(float(loss) - float(lo)) < 1e-4 * abs(float(lo))


def test_entry_scripts_run_on_a_dataset_without_synthetic(data_dir, tmp_path):
    """train_mimic.py then test_mimic.py on HDF5 files (no --synthetic): checkpoints in the reference layout, in-loop
    evaluation json, greedy captions for every pair of the test split."""
    import test_mimic
    import train_mimic
    out = train_mimic.main(['--data_dir', data_dir, '--max_iter', '5', '--batch_size', '8', '--snapshot_interval', '5',
                            '--exp_dir', str(tmp_path), '--quiet', '--eval_batches', '1'])
    assert out['iteration'] == 5 and np.isfinite(out['last']['total_loss'])
    ck = os.path.join(out['snapshot_dir'], 'checkpoint_5.pt')
    assert os.path.exists(ck) and os.path.exists(os.path.join(out['snapshot_dir'], 'checkpoint_best.pt'))
    res = test_mimic.main(['-p', ck, '--data_dir', data_dir, '--exp_dir', str(tmp_path), '--feature_mode', 'both'])
    n_test = len(json.load(open(os.path.join(data_dir, 'splits_mimic_VQA.json')))['test'])
    rows = json.load(open(res['results_file']))
    assert res['n'] == n_test == len(rows) and set(rows[0]) == {'caption', 'image_id'}


def test_real_data_loop_replayed_from_hip_graphs(data_dir, tmp_path):
    """train_mimic.py --hip_graph on the HDF5 files: ragged answers give batches with different numbers of decode steps,
    each captured once (one hipGraph per shape, one shared memory pool) and replayed afterwards.  The replayed run draws
    its dropout masks from (frozen site seeds, device epoch word) instead of the advancing host seed stream, so it is a
    different -- equally valid -- sample of the same training process: it must travel as far as an eager run and end at a
    comparable loss, not at the same weights (mask-for-mask equality is test_graph_replay_equals_eager)."""
    import train_mimic
    common = ['--data_dir', data_dir, '--max_iter', '8', '--batch_size', '8', '--snapshot_interval', '100', '--quiet',
              '--eval_batches', '1', '--seed', '5']
    w0 = _weights(train_mimic.main(common[:2] + ['--max_iter', '0'] + common[4:] + ['--exp_dir', str(tmp_path / 'w0')]))
    e1 = train_mimic.main(common + ['--exp_dir', str(tmp_path / 'e1')])
    g = train_mimic.main(common + ['--exp_dir', str(tmp_path / 'g'), '--hip_graph'])
    assert g['iteration'] == 8 and np.isfinite(g['last']['total_loss'])
    assert g['graphs_captured'] >= 2 and e1['graphs_captured'] == 0          # several decode lengths among the 8 batches
    we1, wg = _weights(e1), _weights(g)
    rms = lambda x: float(x.pow(2).mean().sqrt())      # noqa: E731
    travelled_e, travelled_g = rms(we1 - w0), rms(wg - w0)
    print('real-data loop: rms weight travel eager %.3e, replayed %.3e; apart %.3e; last loss %.4f vs %.4f; %d graphs'
          % (travelled_e, travelled_g, rms(we1 - wg), e1['last']['total_loss'], g['last']['total_loss'], g['graphs_captured']))
    assert 0.5 * travelled_e <= travelled_g <= 2.0 * travelled_e and rms(we1 - wg) <= 1.5 * travelled_e
    assert abs(g['last']['total_loss'] - e1['last']['total_loss']) <= 0.25 * abs(e1['last']['total_loss'])
    # the replayed run detaches its dropout epoch word when it ends: an eager run afterwards draws the masks of the first one
    e2 = train_mimic.main(common + ['--exp_dir', str(tmp_path / 'e2')])
    assert rms(_weights(e2) - we1) <= 0.2 * travelled_e, 'an eager run after a --hip_graph run does not reproduce the earlier eager run'


def _weights(out):
    return torch.cat([p.detach().reshape(-1).double().cpu() for m in (out['change_detector'], out['speaker'])
                      for p in m.parameters()])


@pytest.mark.parametrize('stop', [3, 4])
def test_resumed_run_continues_the_uninterrupted_run(tmp_path, stop):
    """7 iterations straight == ``stop`` iterations + checkpoint + --resume for the rest, with 3-iteration epochs and a
    learning-rate decay every epoch: the snapshot lands on an epoch boundary (stop = 3) or mid-epoch (stop = 4); epoch
    counter, StepLR phase, Adam moments and the dropout stream all carry over."""
    import train_mimic
    common = ['--synthetic', '--batch_size', '4', '--nodes', '12', '--answer_steps', '5', '--iters_per_epoch', '3',
              '--step_size', '2', '--quiet', '--seed', '77']
    a = train_mimic.main(common + ['--max_iter', '7', '--snapshot_interval', '100', '--exp_dir', str(tmp_path / 'a')])
    a2 = train_mimic.main(common + ['--max_iter', '7', '--snapshot_interval', '100', '--exp_dir', str(tmp_path / 'a2')])
    b1 = train_mimic.main(common + ['--max_iter', str(stop), '--snapshot_interval', str(stop), '--exp_dir', str(tmp_path / 'b')])
    ck = os.path.join(b1['snapshot_dir'], 'checkpoint_%d.pt' % stop)
    b2 = train_mimic.main(common + ['--max_iter', '7', '--snapshot_interval', '100', '--exp_dir', str(tmp_path / 'c'),
                                    '--resume', '1', '--checkpoint', ck])
    assert a['epoch'] == b2['epoch'] == 3 and a['iteration'] == b2['iteration'] == 7
    assert a['lr'] == b2['lr'] == pytest.approx(1e-4 * 0.1 ** 1)          # StepLR stepped at the start of epochs 0, 1, 2
    wa, wa2, wb = _weights(a), _weights(a2), _weights(b2)
    # yardstick: two uninterrupted runs differ by float-atomic noise (Adam turns a noise-level gradient into a +-lr step)
    noise = float((wa - wa2).abs().max())
    drift = float((wa - wb).abs().max())
    moved = float((wa - _weights(b1)).abs().max())
    rms = lambda x: float(x.pow(2).mean().sqrt())      # noqa: E731
    print('resume at %d: max |dw| resumed vs straight %.3e (two straight runs: %.3e; travelled after the snapshot %.3e); '
          'rms %.3e vs %.3e' % (stop, drift, noise, moved, rms(wa - wb), rms(wa - wa2)))
    # the path has no float atomics (test_training_step_is_bit_reproducible): two straight runs are bit-identical, and so is
    # the resumed one -- Adam moments, step count, learning-rate phase, epoch shuffling and dropout stream all carried over
    assert noise == 0.0, 'two uninterrupted runs differ'
    assert drift == 0.0, 'the resumed run left the uninterrupted one'
    assert moved > 0.0
    assert a['last']['total_loss'] == b2['last']['total_loss']


@pytest.mark.parametrize('where', ['epoch_boundary', 'mid_epoch'])
def test_resume_on_real_data_continues_the_shuffle(data_dir, tmp_path, where):
    """The same equality on the HDF5 loader, whose per-epoch permutation comes from torch's global generator: a snapshot on an
    epoch boundary stores the generator state the NEXT epoch will shuffle from (it used to store None, and the resumed run
    replayed epoch 0's order); a mid-epoch snapshot stores the state the running epoch's sampler drew from."""
    import train_mimic
    common = ['--data_dir', data_dir, '--batch_size', '8', '--quiet', '--seed', '11', '--eval_batches', '1']
    probe = train_mimic.main(common + ['--max_iter', '1', '--snapshot_interval', '1000', '--no_final_snapshot',
                                       '--exp_dir', str(tmp_path / 'p')])
    ipe = probe['iters_per_epoch']
    assert ipe >= 3
    stop = ipe if where == 'epoch_boundary' else ipe + 2
    total = 2 * ipe + 1                                    # into the third epoch
    a = train_mimic.main(common + ['--max_iter', str(total), '--snapshot_interval', '1000', '--exp_dir', str(tmp_path / 'a')])
    # the uninterrupted run that ALSO snapshots and evaluates at ``stop`` (and at every multiple of it): the in-loop evaluation
    # builds a loader iterator, which draws from torch's default generator -- evaluate() puts the generator back
    a_ev = train_mimic.main(common + ['--max_iter', str(total), '--snapshot_interval', str(stop), '--exp_dir', str(tmp_path / 'ae')])
    b1 = train_mimic.main(common + ['--max_iter', str(stop), '--snapshot_interval', str(stop), '--exp_dir', str(tmp_path / 'b')])
    ck = os.path.join(b1['snapshot_dir'], 'checkpoint_%d.pt' % stop)
    b2 = train_mimic.main(common + ['--max_iter', str(total), '--snapshot_interval', '1000', '--exp_dir', str(tmp_path / 'c'),
                                    '--resume', '1', '--checkpoint', ck])
    assert a['iteration'] == b2['iteration'] == total and a['epoch'] == b2['epoch']
    drift = float((_weights(a) - _weights(b2)).abs().max())
    assert drift == 0.0, 'resumed real-data run left the uninterrupted one: max |dw| = %.3e' % drift
    assert a['last']['total_loss'] == b2['last']['total_loss']
    drift_ev = float((_weights(a_ev) - _weights(b2)).abs().max())
    assert drift_ev == 0.0, 'a run that evaluates at iteration %d shuffles its later epochs differently: max |dw| = %.3e' % (stop, drift_ev)


def test_graph_replay_equals_eager():
    """Three training steps replayed from the captured hipGraph == the same three steps issued eagerly with the same
    dropout sites and epochs (weights agree to float-atomic noise); consecutive replays draw different masks."""
    from common import build_models
    from ekaid_amd import ops
    from ekaid_amd.config import bench_cfg
    from ekaid_amd.dp import FlatAdam
    from ekaid_amd.engine import TrainStep
    from ekaid_amd.synthetic import make_batch
    cfg = bench_cfg(D=256, seq_length=12)
    host = make_batch(B=4, N=12, D=256, Lq=8, T=5, seq_length=12, V=148, seed=5)
    b = {k: v.cuda() for k, v in host.items()}
    T = 5
    finals, losses = [], []
    for mode in ('eager', 'eager', 'graph'):
        cd, sp = build_models(cfg, 147, 148, 'cuda', train=True)
        opt = FlatAdam([cd, sp], lr=1e-3)
        ts = TrainStep(cfg, cd, sp, opt, use_graph=(mode == 'graph'))
        try:
            ops.manual_seed(4242)
            ls = []
            if mode == 'eager':
                ts._epoch_word(b['input_1'].device)
                s0 = ops.seed_state()
                for _ in range(3):
                    ops.set_seed_state(s0)
                    ls.append(float(ts._body(b, T, None)[0].detach()))
            else:
                for _ in range(3):
                    ls.append(float(ts(b, T)[0].detach()))
                assert opt.step_count == 3
        finally:
            ts.close()
        finals.append(torch.cat([p.detach().reshape(-1).double().cpu() for p in list(cd.parameters()) + list(sp.parameters())]))
        losses.append(ls)
    # yardstick: the two eager runs differ by float-atomic noise, which Adam amplifies (a noise-level gradient becomes a
    # +-lr step); the graph run must sit inside the same band
    rms = lambda x: float(x.pow(2).mean().sqrt())       # noqa: E731
    noise_w = rms(finals[0] - finals[1])
    noise_l = max(abs(a - c) for a, c in zip(losses[0], losses[1]))
    d_w = rms(finals[0] - finals[2])
    d_l = max(abs(a - c) for a, c in zip(losses[0], losses[2]))
    print('graph vs eager: rms dw %.3e (eager vs eager %.3e), max dloss %.3e (eager vs eager %.3e)' % (d_w, noise_w, d_l, noise_l))
    assert losses[0][0] == pytest.approx(losses[2][0], rel=1e-6)
    assert len(set(losses[2])) == 3
    # bit-reproducible path: eager == eager exactly, and the replayed graph issues the same kernels on the same data
    assert noise_w == 0.0 and noise_l == 0.0
    assert d_w == 0.0 and d_l == 0.0, 'hipGraph replay differs from the eager steps'


def test_graphs_captured_at_growing_shapes_keep_replaying_correctly():
    """train_mimic.py --hip_graph on ragged data: a batch with more rows / decode steps arrives AFTER a smaller shape was captured and
    makes the scratch buffer grow.  The graph captured first must go on giving the eager result (hip.workspace retires an outgrown buffer
    instead of freeing it under the captured kernels): small, large, small, large replayed from a scratch buffer that grows in between ==
    the same replays with the buffer sized for the large shape from the start."""
    from common import build_models
    from ekaid_amd import hip as H, ops
    from ekaid_amd.config import bench_cfg
    from ekaid_amd.dp import FlatAdam
    from ekaid_amd.engine import TrainStep
    from ekaid_amd.synthetic import make_batch
    cfg = bench_cfg(D=256, seq_length=12)
    small = {k: v.cuda() for k, v in make_batch(B=2, N=12, D=256, Lq=8, T=3, seq_length=12, V=148, seed=5).items()}
    large = {k: v.cuda() for k, v in make_batch(B=16, N=36, D=256, Lq=8, T=9, seq_length=12, V=148, seed=6).items()}
    seq = [(small, 3), (large, 9), (small, 3), (large, 9)]
    dev = small['input_1'].device
    finals, losses, grown = [], [], []
    for mode in ('presized', 'growing'):
        saved = H._workspaces.pop(dev, None)           # start from no scratch buffer: the first (small) shape sizes it ...
        if mode == 'presized':
            H.reserve_workspace(1 << 30, dev)          # ... or from one no shape of this test outgrows
        retired0 = len(H._retired)
        cd, sp = build_models(cfg, 147, 148, 'cuda', train=True)
        opt = FlatAdam([cd, sp], lr=1e-3)
        ts = TrainStep(cfg, cd, sp, opt, use_graph=True)
        try:
            ops.manual_seed(4242)
            ls = [float(ts(b, T)[0].detach()) for b, T in seq]
            torch.cuda.synchronize()
            grown.append(len(H._retired) - retired0)
        finally:
            ts.close()
            torch.cuda.synchronize()
            H._workspaces.pop(dev, None)
            del H._retired[retired0:]
            if saved is not None:
                H._workspaces[dev] = saved
        finals.append(torch.cat([p.detach().reshape(-1).double().cpu() for p in list(cd.parameters()) + list(sp.parameters())]))
        losses.append(ls)
    assert grown == [0, grown[1]] and grown[1] >= 1, 'the large shape was meant to outgrow the scratch buffer of the small one (%r)' % grown
    assert losses[0] == losses[1], 'losses differ once the scratch buffer grows between the captures: %r vs %r' % (losses[1], losses[0])
    assert torch.equal(finals[0], finals[1]), 'weights after small / large / small / large replays depend on the scratch buffer growing'


def test_decode_steps_rounded_up_leave_loss_and_gradients_unchanged():
    """TrainStep(t_round=8): the decoder runs 8 steps for a batch whose answers need 5 (fewer captured graphs on ragged data).  The three
    extra steps read padding tokens and their log-prob rows meet an all-zero loss mask, so the loss and every gradient are those of the
    exact step count -- up to the summation order of the time-batched products, whose contraction grows by zero rows (hence 1e-4 of
    each tensor's scale -- measured 1.6e-5 -- not bit equality)."""
    from common import build_models
    from ekaid_amd import ops
    from ekaid_amd.config import bench_cfg
    from ekaid_amd.dp import FlatAdam
    from ekaid_amd.engine import TrainStep, forward_loss
    from ekaid_amd.synthetic import make_batch
    cfg = bench_cfg(D=256, seq_length=12)
    host = make_batch(B=6, N=12, D=256, Lq=8, T=5, seq_length=12, V=148, seed=9)
    b = {k: v.cuda() for k, v in host.items()}
    cd, sp = build_models(cfg, 147, 148, 'cuda', train=False)          # eval mode: no dropout, so the two runs are comparable
    ts = TrainStep(cfg, cd, sp, FlatAdam([cd, sp], lr=1e-3), t_round=8)
    assert ts.rounded_steps(5) == 8 and ts.rounded_steps(11) == 12 and ts.rounded_steps(8) == 8
    assert TrainStep(cfg, cd, sp, ts.opt).rounded_steps(5) == 5
    res = []
    for T in (5, ts.rounded_steps(5)):
        ops.manual_seed(5)
        for m in (cd, sp):
            m.zero_grad()
        loss = forward_loss(cfg, cd, sp, b, T)[0]
        loss.backward()
        res.append((float(loss.detach()), {k: p.grad.detach().clone() for m in (cd, sp) for k, p in m.named_parameters() if p.grad is not None}))
    (l5, g5), (l8, g8) = res
    assert l5 == pytest.approx(l8, rel=2e-6), (l5, l8)
    assert g5.keys() == g8.keys() and len(g5) > 40
    gmax = max(float(v.abs().max()) for v in g5.values())
    # (parameters whose true gradient is zero -- key bias under a softmax, W2 bias under the batch softmax -- carry rounding noise only)
    live = [k for k in g5 if float(g5[k].abs().max()) > 1e-6 * gmax]
    assert len(live) > 40
    worst = max(float((g5[k] - g8[k]).abs().max()) / float(g5[k].abs().max()) for k in live)
    print('decode steps 5 vs 8 (3 masked): loss %.7f / %.7f, worst gradient deviation %.2e of the tensor scale' % (l5, l8, worst))
    assert worst <= 1e-4, worst      # measured 1.6e-5: f32 summation order of the longer time-batched contractions


def test_graph_key_follows_the_scheduled_sampling_rate(monkeypatch):
    """``speaker.ss_prob`` is a by-value kernel argument and selects the decoder path (fused teacher-forced node at 0, step-wise
    sampling above): a hipGraph captured at one rate must not be replayed at another.  The rate is part of the graph
    key, a change recaptures through the sampling path, and graphs of the old rate are dropped."""
    from common import build_models
    from ekaid_amd import hip, ops
    from ekaid_amd.config import bench_cfg
    from ekaid_amd.dp import FlatAdam
    from ekaid_amd.engine import TrainStep
    from ekaid_amd.synthetic import make_batch
    cfg = bench_cfg(D=256, seq_length=12)
    host = make_batch(B=4, N=12, D=256, Lq=8, T=5, seq_length=12, V=148, seed=5)
    b = {k: v.cuda() for k, v in host.items()}
    T = 5
    draws = []
    real = hip.lib().ekaid_ss_tokens
    monkeypatch.setattr(hip.lib(), 'ekaid_ss_tokens', lambda *a: (draws.append(a[5]), real(*a))[1], raising=False)
    cd, sp = build_models(cfg, 147, 148, 'cuda', train=True)
    opt = FlatAdam([cd, sp], lr=1e-3)
    ts = TrainStep(cfg, cd, sp, opt, use_graph=True)
    captures = []
    cap = ts._capture
    monkeypatch.setattr(ts, '_capture', lambda *a: (captures.append(sp.ss_prob), cap(*a))[1])
    try:
        ops.manual_seed(777)
        losses = []
        for rate in (0.0, 0.0, 0.6, 0.6, 0.9):
            sp.ss_prob = rate
            n0 = len(draws)
            losses.append(float(ts(b, T)[0].detach()))
            assert len(ts._graphs) == 1 and next(iter(ts._graphs))[4] == rate, list(ts._graphs)
            if rate > 0 and captures.count(rate) == 1 and len(draws) > n0:
                # the capture ran the sampling path: warm-ups + capture, T - 1 draws each, at the NEW rate
                assert len(draws) - n0 == (ts.warmup + 1) * (T - 1) and all(abs(d - rate) < 1e-6 for d in draws[n0:])
        assert captures == [0.0, 0.6, 0.9], captures
        assert opt.step_count == 5
        assert len(set(losses)) == 5 and all(l == l and l > 0 for l in losses), losses
    finally:
        ts.close()


def test_rccl_path_on_one_gpu(tmp_path):
    """backend 'nccl' (= RCCL) with world_size 1 on this box: process-group init with device_id, bucketed async
    all-reduce launched from the gradient hooks, wait, fused Adam -- eager and captured into the hipGraph -- must
    reproduce the run without any collective."""
    import socket
    script = os.path.join(ROOT, 'tests', 'rccl_world1.py')
    for attempt in range(2):                       # (a fixed port can still be in TIME_WAIT from an earlier rendezvous)
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
                   HSA_ENABLE_IPC_MODE_LEGACY='0')
        r = subprocess.run([sys.executable, script, str(tmp_path / 'out.json')], env=env, capture_output=True, text=True,
                           timeout=600)
        if r.returncode == 0:
            break
        print('rccl_world1.py attempt %d failed:\n%s\n%s' % (attempt, r.stdout[-3000:], r.stderr[-3000:]))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    out = json.load(open(str(tmp_path / 'out.json')))
    assert out['buckets'] > 1 and out['collectives'] > 0
    # bit-reproducible path + an all-reduce over one rank (identity) + 1/world = 1: every variant ends at the same bits
    assert out['noise'] == 0.0 and out['delta_eager'] == 0.0 and out['delta_graph'] == 0.0, out
    # overlap, in BOTH default multi-GPU modes (hook-driven eager launches; collectives captured inside the hipGraph): the
    # first gradient bucket's all-reduce is enqueued while backward still has kernels to issue, the buckets go out in order,
    # and only the LAST bucket can land after backward's final kernel.  (comm='eager' after a replay is the serialised
    # form: there every collective comes after backward.)
    for mode in ('order_eager', 'order_captured'):
        o = out[mode]
        calls = [c for _, c in o['launches']]
        assert [b for b, _ in o['launches']] == list(range(out['buckets'])), o
        assert calls == sorted(calls) and calls[0] < o['after_backward'], (mode, o)
        assert sum(c < o['after_backward'] for c in calls) >= out['buckets'] - 1, (mode, o)
    # one eager layout-building pass (all buckets) + the capture itself; the capture's two warm-up iterations add nothing
    assert out['order_captured']['n_logged'] == 2 * out['buckets'], 'warm-up iterations of a capture must not communicate'
    assert all(c >= out['order_after_replay']['after_backward'] for _, c in out['order_after_replay']['launches'])


def test_bench_real_data_line_is_the_only_thing_on_stdout():
    """The driver parses bench.py's stdout as ONE JSON line; the real-data leg runs train_mimic.py and the dataset classes, whose
    progress prints must all go to stderr."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--real-only'], capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-3000:]
    out = json.loads(lines[0])['real_b64']
    assert out['qa_pairs_per_s'] > 0 and out['graphs_captured'] >= 2
    assert out['loader_wait_fraction'] is not None and out['exposed_h2d_ms_per_iteration'] >= 0.0


def test_infer_step_replay_equals_eager_forward():
    """InferStep: the eval-mode detector forward replayed from a hipGraph (captured per input shape) returns exactly what the
    eager forward returns, for two batch shapes in turn, and keeps returning it on later replays with new inputs."""
    from ekaid_amd.engine import InferStep, detector_forward
    from ekaid_amd.synthetic import make_batch
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from common import SMALL, small_models
    cfg, cd, sp = small_models('all', device='cuda')
    cd.eval()
    step = InferStep(cfg, cd, 'all', use_graph=True)
    for rnd in range(2):
        for B in (2, 5):
            shape = dict(SMALL)
            shape['B'] = B
            b = {k: v.cuda() for k, v in make_batch(seed=100 + 10 * rnd + B, **shape).items()}
            with torch.no_grad():
                want = [t.clone() for t in detector_forward(cfg, cd, b, 'all')]
            got = step(b)
            for w, g in zip(want, got):
                assert torch.equal(w, g)
    assert len(step._graphs) == 2


def test_checkpoint_written_by_the_reference_runs_through_test_mimic(tmp_path):
    """tests/golden/ref_checkpoint_small.pt.xz was written by the REFERENCE's own code (its modules, its ``save_checkpoint``, its
    ``utils.attr_dict.AttrDict`` as ``model_cfg``: tests/golden/make_ref_checkpoint.py replays train_mimic.py:281-290).  The product's
    ``test_mimic.py -p`` loads it unchanged and decodes the answers the reference's own evaluation loop (test_mimic.py:97-124,
    greedy ``_sample``) decoded from it: token ids identical, their log-probs within 1e-3.  ref: utils/utils.py:11-23."""
    import lzma
    import yaml
    import test_mimic
    from ekaid_amd.utils.utils import load_checkpoint
    gold = np.load(os.path.join(ROOT, 'tests', 'golden', 'ref_checkpoint_small_golden.npz'))
    ck_path = str(tmp_path / 'checkpoint_0.pt')
    with lzma.open(os.path.join(ROOT, 'tests', 'golden', 'ref_checkpoint_small.pt.xz'), 'rb') as f, open(ck_path, 'wb') as g:
        g.write(f.read())
    raw = open(ck_path, 'rb').read()
    assert b'utils.attr_dict' in raw and b'ekaid_amd' not in raw, 'the file must be the reference\'s own pickle'
    ck = load_checkpoint(ck_path)
    assert set(ck) == {'change_detector_state', 'speaker_state', 'model_cfg'}

    def plain(d):
        return {k: plain(v) for k, v in d.items()} if isinstance(d, dict) else d
    cfg_path = str(tmp_path / 'cfg.yaml')
    with open(cfg_path, 'w') as f:
        yaml.safe_dump(plain(ck['model_cfg']), f)            # the run's yaml: what `--cfg` points at in the reference too
    nb = sum(1 for k in gold.files if k.startswith('seq_') and not k.startswith('seq_logp'))
    res = test_mimic.main(['-p', ck_path, '--cfg', cfg_path, '--synthetic', '--batches', str(nb), '--nodes', '8',
                           '--exp_dir', str(tmp_path), '--graph', 'all'])
    assert len(res['sequences']) == nb
    for i in range(nb):
        want, want_lp = gold['seq_%d' % i], gold['seq_logp_%d' % i]
        got, got_lp = res['sequences'][i].numpy(), res['sequence_logprobs'][i].numpy()
        assert np.array_equal(got, want), 'greedy ids differ from the reference\'s in batch %d' % i
        assert np.abs(got_lp - want_lp).max() <= 1e-3 * max(np.abs(want_lp).max(), 1.0)
