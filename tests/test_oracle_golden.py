"""The CPU oracle (oracle/ekaid_oracle.py) against the golden vectors captured from the reference
(tests/golden/*.npz, written by tests/golden/make_golden.py).  Pins the oracle: every later HIP
parity test compares against it."""
import os

import numpy as np
import pytest
import torch

from common import FULL, SMALL, build_models, oracle_params, real_shape_case, small_models
from ekaid_amd.config import bench_cfg
from ekaid_amd.synthetic import make_batch
from golden_util import assert_no_farther_from_truth, assert_sketch_close, assert_summary_close, load_golden, negligible_grad_names, tensor_sketch
from oracle import ekaid_oracle as O

TOL = 2e-5
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def close(a, b, tol=TOL, what=''):
    a = a.detach().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, '%s: shape %s vs %s' % (what, a.shape, b.shape)
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-6)
    assert err <= tol, '%s: rel-to-max err %.3e > %.1e' % (what, err, tol)


def run_oracle(cfg, cd, sp, batch, setting, graph, grads=True):
    Pcd, Psp = oracle_params(cd, grads), oracle_params(sp, grads)
    loss, res = O.train_loss(Pcd, Psp, cfg, batch, setting, graph)
    named = []
    if grads:
        loss.backward()
        named = [('cd.' + k, Pcd[k].grad) for k, _ in cd.named_parameters()] + \
                [('sp.' + k, Psp[k].grad) for k, _ in sp.named_parameters()]
    return loss, res, named


@pytest.mark.parametrize('graph', ['semantic', 'spatial', 'implicit', 'all', 'i+s'])
def test_f1_forward_backward(graph):
    g = load_golden('f1_' + graph.replace('+', 'p'))
    cfg, cd, sp = small_models(graph)
    batch = make_batch(**SMALL)
    loss, res, named = run_oracle(cfg, cd, sp, batch, 'mode2', graph)
    close(loss, g['loss'], what='loss')
    for i, k in enumerate(('pred', 'att_bef', 'att_aft', 'feat_bef', 'feat_aft', 'feat_diff')):
        close(res['outs'][i], g[k], what=k)
    close(res['logp'], g['logp'], what='logp')
    close(res['logpos'], g['logpos'], what='logpos')
    extra = assert_summary_close(g['grad_names'], g['grad_rows'], named, 2e-4, 'grads[%s]' % graph)
    # parameters the reference never differentiates must get no (or zero) gradient from the oracle either
    got = dict(named)
    for k in extra:
        if k == 'cd.w_emb.emb_.weight':      # frozen in the reference (language_model.py:29)
            continue
        assert float(got[k].abs().max()) == 0.0, 'oracle produced a gradient for dead parameter ' + k


def test_f6_explicit_relation_with_fewer_neighbours_than_nodes():
    g = load_golden('f6_spatial_nongt5')
    cfg, cd, sp = small_models('spatial')
    cfg.model.change_detector.nongt_dim = 5
    loss, res, named = run_oracle(cfg, cd, sp, make_batch(**SMALL), 'mode2', 'spatial')
    close(loss, g['loss'], what='loss')
    close(res['logp'], g['logp'], what='logp')
    close(res['outs'][3], g['feat_bef'], what='feat_bef')
    assert_summary_close(g['grad_names'], g['grad_rows'], named, 2e-4, 'grads[nongt5]')


def test_f7_implicit_relation_with_fewer_neighbours_than_nodes():
    """The reference's row-sliced, re-read pair geometry (mimic_utils.py:186 + graph_att_layer.py:117-126): oracle == reference."""
    g = load_golden('f7_implicit_nongt5')
    cfg, cd, sp = small_models('implicit')
    cfg.model.change_detector.nongt_dim = 5
    loss, res, named = run_oracle(cfg, cd, sp, make_batch(**SMALL), 'mode2', 'implicit')
    close(loss, g['loss'], what='loss')
    close(res['logp'], g['logp'], what='logp')
    close(res['outs'][3], g['feat_bef'], what='feat_bef')
    assert_summary_close(g['grad_names'], g['grad_rows'], named, 2e-4, 'grads[implicit nongt5]')


def test_f1_mode0_ssre_layernorm():
    g = load_golden('f1_mode0')
    cfg, cd, sp = small_models('all', 'mode0')
    batch = make_batch(**SMALL)
    loss, res, named = run_oracle(cfg, cd, sp, batch, 'mode0', 'all')
    close(loss, g['loss'], what='loss')
    close(res['logp'], g['logp'], what='logp')
    close(res['outs'][3], g['feat_bef'], what='feat_bef')
    assert_summary_close(g['grad_names'], g['grad_rows'], named, 2e-4, 'grads[mode0]')


def test_f2_batch_coupled_question_attention():
    g = load_golden('f2_qc')
    cfg, cd, sp = small_models('semantic')
    P = oracle_params(cd)
    b4 = make_batch(**dict(SMALL, B=4))
    qs = {}
    for B in (2, 4):
        seq = O.gru_forward_all(P, O.word_embedding(P, b4['question'][:B]))
        close(seq, g['gru_%d' % B], what='gru B=%d' % B)
        qs[B] = O.question_self_attention(P, seq)
        close(qs[B], g['q_%d' % B], what='q B=%d' % B)
    # quirk Q-C really couples the batch: same two samples, different answer
    assert float((qs[2] - qs[4][:2]).abs().max()) > 1e-3


def test_f3_blocks():
    g = load_golden('f3_blocks')
    cfg, cd, sp = small_models('all')
    Pcd, Psp = oracle_params(cd), oracle_params(sp)
    batch = make_batch(**SMALL)
    N = SMALL['N']
    spa = O.process_matrix(batch['d_adj'], N, 11)
    sem = O.process_matrix(batch['d_sem'], N, 3)
    close(spa, g['onehot_spa'], 0, 'onehot_spa')
    close(sem, g['onehot_sem'], 0, 'onehot_sem')
    pm = O.position_matrix(batch['d_bb'], 52)
    close(pm, g['pos_mat'], 1e-12, 'pos_mat')
    pe = O.position_embedding(pm, 64)
    close(pe, g['pos_emb'], 1e-9, 'pos_emb')
    x = torch.from_numpy(g['layer_x'])
    pre = 'spatial_relation.explicit_relation.neighbor_net.1'
    o, s = O.graph_attention_layer(Pcd, pre, x, torch.from_numpy(g['layer_cond']), None,
                                   torch.from_numpy(g['layer_lb']), 4, 52)
    close(o, g['layer_exp_out'], what='layer_exp_out')
    close(s, g['layer_exp_soft'], what='layer_exp_soft')
    assert np.allclose(s[0, 3].numpy(), 0.125)          # isolated node -> uniform row (the -9e15 mask)
    pre = 'imp_relation.implicit_relation'
    lbi = O.wn_linear(Pcd, pre + '.bias.main.0', torch.ones(2, N, N, 1)).squeeze(-1)
    o, s = O.graph_attention_layer(Pcd, pre + '.neighbor_net.1', x, torch.ones(2, N, N), pe, lbi, 4, 52)
    close(o, g['layer_imp_out'], what='layer_imp_out')
    close(s, g['layer_imp_soft'], what='layer_imp_soft')
    o, affs = O.gat_net(Pcd, 'spatial_relation.explicit_relation', torch.from_numpy(g['gat_in']), spa, None, 2, 4, 52,
                        faithful=True)
    close(o, g['gat_out'], what='gat_out')
    close(affs[0], g['gat_aff0'], what='gat_aff0')
    close(affs[1], g['gat_aff1'], what='gat_aff1')
    o2, _ = O.gat_net(Pcd, 'spatial_relation.explicit_relation', torch.from_numpy(g['gat_in']), spa, None, 2, 4, 52)
    assert torch.equal(o, o2)                            # Q-B: skipping the dead direction changes nothing
    q = O.question_encoder(Pcd, batch['question'])
    close(q, g['enc_q'], what='enc_q')
    v0 = O.linear(Pcd, 'img', batch['input_1'])
    close(v0, g['enc_v0'], what='enc_v0')
    v1, _ = O.explicit_encoder(Pcd, 'semantic_relation', v0, sem, q, cfg)
    close(v1, g['enc_v1'], what='enc_v1')
    v2, _ = O.explicit_encoder(Pcd, 'spatial_relation', v1, spa, q, cfg)
    close(v2, g['enc_v2'], what='enc_v2')
    v3, _ = O.implicit_encoder(Pcd, 'imp_relation', v2, pe, q, cfg)
    close(v3, g['enc_v3'], what='enc_v3')
    h, c = torch.from_numpy(g['core_h']), torch.from_numpy(g['core_c'])
    lp, st, lpp, mw = O.speaker_step(Psp, torch.from_numpy(g['core_it']), torch.from_numpy(g['core_fb']),
                                     torch.from_numpy(g['core_fa']), torch.from_numpy(g['core_fd']),
                                     ((h[0], h[1]), (c[0], c[1])))
    close(lp, g['core_logp'], what='core_logp')
    close(torch.stack(st[0]), g['core_h2'], what='core_h2')
    close(torch.stack(st[1]), g['core_c2'], what='core_c2')
    close(lpp, g['core_logpos'], what='core_logpos')
    close(mw, g['core_modw'], what='core_modw')


@pytest.mark.parametrize('graph', ['semantic', 'spatial', 'implicit', 'all', 'i+s', 'real_shapes'])
def test_f8_every_element_of_every_gradient_is_pinned_to_the_reference(graph):
    """Whole-tensor sketches of the reference's gradients (tests/golden/make_grad_sketch.py: per 1/16 chunk of every gradient
    tensor its norm and 4 random-sign projections): the oracle's gradients agree chunk by chunk, so no slice of any gradient is
    left to the 8 sampled positions of the F1 summaries.  'real_shapes' = 52 regions, ragged answers, zero row, isolated node."""
    g = load_golden('f8_grad_sketch_' + graph.replace('+', 'p'))
    if graph == 'real_shapes':
        cfg, batch = real_shape_case()
        cd, sp = build_models(cfg, 30, 31, 'cpu')
        graph = 'all'
    else:
        cfg, cd, sp = small_models(graph)
        batch = make_batch(**SMALL)
    Pcd, Psp = oracle_params(cd, True), oracle_params(sp, True)
    loss, _ = O.train_loss(Pcd, Psp, cfg, batch, 'mode2', graph)
    loss.backward()
    close(loss, g['loss'], what='loss')
    named = [('cd.' + k, Pcd[k].grad) for k, _ in cd.named_parameters()] + [('sp.' + k, Psp[k].grad) for k, _ in sp.named_parameters()]
    extra, worst = assert_sketch_close(g['grad_names'], g['grad_sketch'], named, 2e-4, 'gradient sketches [%s]' % graph)
    print('worst chunk: %s %.2e' % worst)
    got = dict(named)
    for k in extra:
        if k == 'cd.w_emb.emb_.weight':      # frozen in the reference (language_model.py:29)
            continue
        assert float(got[k].abs().max()) == 0.0, 'oracle produced a gradient for dead parameter ' + k


def test_f9_batch256_forward_at_the_metric_batch_size():
    """BASELINE.json configs[2] shapes at the metric's own batch (256 pairs, full-width model): quirk Q-C makes the question vectors a
    function of the batch size itself (language_model.py:149), so this is a case of its own.  The oracle against the REFERENCE run at
    B = 256 (tests/golden/make_b256_golden.py): loss, question vectors, arg-max answers of all 4096 decode positions, their
    log-probs, the log-prob of every label token, and whole-tensor sketches of the log-probs and of the pooled difference feature."""
    g = load_golden('f9_b256')
    cfg = bench_cfg()
    cd, sp = build_models(cfg, 147, 148, 'cpu')
    batch = make_batch(B=256, **FULL)
    Pcd, Psp = oracle_params(cd), oracle_params(sp)
    with torch.no_grad():
        q = O.question_encoder(Pcd, batch['question'])
        loss, res = O.train_loss(Pcd, Psp, cfg, batch, 'mode2', 'all')
    close(loss, g['loss'], what='loss')
    close(q[:4], g['q_rows'], what='question vectors')
    lp = res['logp'][:, :16]
    rows = [0, 85, 170, 255]
    close(lp[rows], g['logp_rows'], what='log-prob rows')
    close(res['outs'][5][rows], g['feat_diff_rows'], what='feat_diff rows')
    am = lp.argmax(-1)
    assert np.array_equal(am.numpy(), g['argmax'].astype(np.int64)), 'arg-max answers differ at B = 256'
    tgt = batch['labels'][:, 1:17]
    close(lp.gather(-1, am[..., None])[..., 0], g['logp_argmax'], what='logp at arg-max')
    close(lp.gather(-1, tgt[..., None])[..., 0], g['logp_label'], what='logp at label')
    for name, t in (('q', q), ('logp', lp), ('feat_diff', res['outs'][5])):
        sk, ref = tensor_sketch(t), g[name + '_sketch']
        scale = np.maximum(ref[:, :1], np.sqrt((ref[:, 0] ** 2).sum() / len(ref)))
        assert (np.abs(sk - ref) / scale).max() <= 1e-4, name + ' sketch'


def test_f3_greedy_sample_ids():
    g = load_golden('f3_sample')
    cfg, cd, sp = small_models('all')
    batch = make_batch(**SMALL)
    Pcd, Psp = oracle_params(cd), oracle_params(sp)
    with torch.no_grad():
        _, res = O.train_loss(Pcd, Psp, cfg, batch, 'mode2', 'all')
        seq, seq_lp = O.speaker_sample_greedy(Psp, cfg, res['outs'][3], res['outs'][4], res['outs'][5])
    assert np.array_equal(seq.numpy(), g['seq'])
    close(seq_lp, g['seq_logp'], what='seq_logp')


def test_oracle_decodes_the_reference_written_checkpoint_like_the_reference(tmp_path):
    """tests/golden/ref_checkpoint_small.pt.xz: a file written by the reference's own ``save_checkpoint`` from its own modules
    (tests/golden/make_ref_checkpoint.py).  Loaded through the product's ``load_checkpoint``, its two state dicts drive the oracle to
    the token ids the reference's evaluation loop decoded from it."""
    import lzma
    from ekaid_amd.utils.utils import load_checkpoint
    gold = load_golden('ref_checkpoint_small_golden')
    path = str(tmp_path / 'checkpoint_0.pt')
    with lzma.open(os.path.join(ROOT, 'tests', 'golden', 'ref_checkpoint_small.pt.xz'), 'rb') as f, open(path, 'wb') as g:
        g.write(f.read())
    ck = load_checkpoint(path)
    cfg = ck['model_cfg']
    Pcd, Psp = ck['change_detector_state'], ck['speaker_state']
    D = cfg.model.change_detector.feat_dim - 2
    for i in range(2):
        batch = make_batch(B=cfg.data.test.batch_size, N=8, D=D, Lq=20, T=16, seq_length=cfg.model.speaker.seq_length, V=148, seed=977 + i)
        with torch.no_grad():
            _, res = O.train_loss(Pcd, Psp, cfg, batch, 'mode2', 'all')
            close(res['outs'][5], gold['feat_diff_%d' % i], what='feat_diff')
            seq, seq_lp = O.speaker_sample_greedy(Psp, cfg, res['outs'][3], res['outs'][4], res['outs'][5])
        assert np.array_equal(seq.numpy(), gold['seq_%d' % i])
        close(seq_lp, gold['seq_logp_%d' % i], what='seq_logp')


def test_f5_adam_two_steps():
    g = load_golden('f5_adam')
    cfg, cd, sp = small_models('all')
    batch = make_batch(**SMALL)
    Pcd, Psp = oracle_params(cd, True), oracle_params(sp, True)
    params = [(('cd.' + k), Pcd[k]) for k, _ in cd.named_parameters()] + \
             [(('sp.' + k), Psp[k]) for k, _ in sp.named_parameters()]
    frozen = {'cd.w_emb.emb_.weight'}
    state = {k: (torch.zeros_like(p), torch.zeros_like(p)) for k, p in params}
    loss = None
    for step in (1, 2):
        for _, p in params:
            p.grad = None
        loss, _ = O.train_loss(Pcd, Psp, cfg, batch, 'mode2', 'all')
        loss.backward()
        for k, p in params:
            if p.grad is None or k in frozen:
                continue
            O.adam_step(p, p.grad, state[k][0], state[k][1], step)
    close(loss, g['loss_step2'], 1e-5, 'loss at step 2')
    # Adam turns noise-around-zero gradients into +-lr steps of arbitrary sign: those parameters are only
    # required to stay within 2*lr*steps of the reference
    g1 = load_golden('f1_all')
    noisy = negligible_grad_names(g1['grad_names'], g1['grad_rows'])
    assert_summary_close(g['param_names'], g['param_rows'], [(k, p.detach()) for k, p in params], 2e-5, 'adam params',
                         skip=noisy)
    import json
    names = json.loads(str(g['param_names']))
    cur = dict((k, p.detach()) for k, p in params)
    from golden_util import tensor_summary
    for k in noisy:
        ref = g['param_rows'][names.index(k)]
        assert np.abs(tensor_summary(cur[k])[2:] - ref[2:]).max() <= 4.1e-4, k


def test_f4_full_width_spot_check():
    g = load_golden('f4_full')
    cfg = bench_cfg()
    cd, sp = build_models(cfg, 147, 148)
    batch = make_batch(B=2, **FULL)
    loss, res, named = run_oracle(cfg, cd, sp, batch, 'mode2', 'all')
    close(loss, g['loss'], what='loss')
    close(res['logp'][:, :16], g['logp'], 5e-5, 'logp')
    assert np.array_equal(res['logp'][:, :16].argmax(-1).numpy(), g['argmax'])
    close(res['outs'][3], g['feat_bef'], 5e-5, 'feat_bef')
    # weight_g gradients are <dW, v>/||v||: a 10^6-term sum with heavy cancellation, fp32-noisy at full width
    assert_summary_close(g['grad_names'], g['grad_rows'], named, 5e-4, 'grads[full]',
                         select=lambda k: not k.endswith('weight_g'))
    assert_summary_close(g['grad_names'], g['grad_rows'], named, 1e-2, 'grads[full, weight_g]',
                         select=lambda k: k.endswith('weight_g'))


def test_f4_float64_oracle_agrees_with_float64_reference():
    """The gradient truth of the full-width check.  f4_full_ref64 = the REFERENCE modules run in float64
    (tests/golden/make_golden.py: gen_f4_ref64, .float() casts widened), f4_full_f64 = the oracle run in float64
    (tests/golden/make_f64_truth.py).  They agree three decades below the reference's own float32 error (1e-3 .. 5e-2 on
    these gradients; the residue is the position-embedding wave-length table, which the float64 reference run builds in
    float64 and the pipeline -- and the oracle -- in float32), so either bounds the float32 paths; the GPU test uses the
    reference's."""
    import json
    r, o, g32 = load_golden('f4_full_ref64'), load_golden('f4_full_f64'), load_golden('f4_full')
    assert abs(float(r['loss']) - float(o['loss'])) <= 1e-8 * abs(float(r['loss']))
    assert np.abs(r['logp'] - o['logp']).max() <= 1e-6
    names = json.loads(str(o['grad_names']))
    assert names == json.loads(str(r['grad_names'])) == json.loads(str(g32['grad_names']))
    R, O, G = r['grad_rows'], o['grad_rows'], g32['grad_rows']
    floor = 1e-6 * float(np.median(R[:, 0]))
    live = R[:, 0] > floor                      # (gradients that are mathematically zero -- e.g. the key bias under a softmax -- are noise)
    rel = np.abs(R[:, 0] - O[:, 0]) / R[:, 0].clip(1e-300)
    assert live.sum() >= 70 and rel[live].max() <= 5e-5, (np.array(names)[live][np.argsort(-rel[live])[:3]], rel[live].max())
    ref32 = np.abs(G[:, 0] - R[:, 0]) / R[:, 0].clip(1e-300)
    assert np.median(ref32[live]) > 20 * np.median(rel[live])        # the float32 reference is far noisier than the gap between the truths


@pytest.mark.parametrize('case', ['T40', 'T90', 'real_full', 'mode0_full'])
def test_f10_production_width_long_answers_and_real_shapes(case):
    """Round-6 fixtures (tests/golden/make_wide_golden.py): the REFERENCE at the yaml widths with 40 / 90 decode steps and at the real
    52 x 1024 shapes with ragged 2 ... 40-token answers -- the cases whose backward runs in the persistent decoder kernel on the GPU.
    The oracle: loss, every log-prob, every gradient chunk by chunk."""
    from common import mode0_case, real_full_case, wide_case
    g = load_golden('f10_' + case)
    setting = 'mode0' if case == 'mode0_full' else 'mode2'          # (mode0: the SSRE + LayerNorm block at the yaml widths, modules.py:17-77)
    cfg, batch = real_full_case() if case == 'real_full' else mode0_case() if case == 'mode0_full' else wide_case(int(case[1:]))
    cd, sp = build_models(cfg, 147, 148, 'cpu')
    loss, res, named = run_oracle(cfg, cd, sp, batch, setting, 'all')
    T = int(g['steps'])
    assert T == int((batch['labels'][:, 1:] != 0).any(0).sum()) + 1
    close(loss, g['loss'], what='loss')
    close(res['logp'][:, :T], g['logp'], 5e-5, 'logp')
    if res['logp'].shape[1] > T:
        assert float(res['logp'][:, T:].detach().abs().max()) == 0.0          # steps after the break stay zero (:190-191, :213)
    assert np.array_equal(res['logp'][:, :T].argmax(-1).numpy(), g['logp'].argmax(-1))
    close(res['outs'][5], g['feat_diff'], 5e-5, 'feat_diff')
    close(res['outs'][1], g['att_bef'], 5e-5, 'att_bef')
    # gradients: the float64 run of the reference is the truth; the oracle (float32) must be as close to it as the reference's float32 run
    worst = assert_no_farther_from_truth(g, named, 'gradient sketches [%s]' % case)
    print('worst vs float64 truth: %s %.2e (reference float32: %.2e)' % worst)
    # ... and the oracle's own float64 run IS that truth (the GPU test measures the HIP path element by element against it): log-probs
    # and every gradient chunk against the float64 run of the reference
    from common import oracle_gradients
    from golden_util import sketch_errors
    loss64, res64, g64 = oracle_gradients(cfg, cd, sp, batch, double=True, setting=setting)
    assert abs(float(loss64) - float(g['loss_f64'])) <= 1e-7 * abs(float(g['loss_f64']))
    assert float((res64['logp'][:, :T].detach() - torch.from_numpy(g['logp_f64']).double()).abs().max()) <= 2e-6      # (stored as float32)
    e = sketch_errors(g['grad_names'], g['grad_sketch_f64'], lambda i, k: tensor_sketch(g64[k]) if g64.get(k) is not None else None)
    # (what is left between the two float64 runs is the position-embedding wave-length table -- float64 in the float64 reference run,
    #  float32 in the pipeline and in the oracle, as in F4 -- amplified by the recurrence: median 3e-5 over 90 steps, and up to 2e-3 on
    #  the weight_g scalars, whose <dW, v> sums cancel heavily.  The float32 paths sit 1e-2 ... 1e-1 from either.)
    k = max(e, key=e.get)
    print('float64 oracle vs float64 reference, worst chunk: %s %.2e, median %.2e' % (k, e[k], float(np.median(list(e.values())))))
    bad = {n: v for n, v in e.items() if v > (5e-3 if n.endswith('weight_g') else 2e-4)}
    assert not bad, bad


def test_f10_greedy_sample_at_batch_64_full_width():
    """The reference's greedy `_sample` (dynamic_speaker_change_pos.py:287-357) at the production widths, batch 64, spread logits:
    identical token ids and stop positions, seq_logprobs."""
    from common import greedy_case
    g = load_golden('f10_greedy_b64')
    cfg, feats, scale, bias0 = greedy_case()
    cd, sp = build_models(cfg, 147, 148, 'cpu')
    Psp = oracle_params(sp)
    Psp['logit.weight'] *= scale
    Psp['logit.bias'][0] += bias0
    with torch.no_grad():
        seq, seq_lp = O.speaker_sample_greedy(Psp, cfg, feats[0], feats[1], feats[2])
    assert float(g['min_margin']) > 5e-4, 'fixture decided by a rounding tie'
    assert np.array_equal(seq.numpy(), g['seq'].astype(np.int64))
    assert np.array_equal((seq != 0).sum(1).numpy(), g['lengths'].astype(np.int64))
    close(seq_lp, g['seq_logp'], 5e-5, what='seq_logp')
