"""A checkpoint written by the REFERENCE's own code, plus what the reference's own evaluation loop decodes from it.

Runs only in the build container (imports /root/reference through the shims of make_golden.py):

    python tests/golden/make_ref_checkpoint.py

  1. builds the reference ``ChangeDetector`` / ``DynamicSpeaker`` at the config-1 widths (``small_cfg``; vocabulary 147 words as
     test_mimic.py's synthetic mode assumes) with the closed-form weights of ``ekaid_amd.formula_init``;
  2. snaps the five large decoder matrices (> 10^5 elements: 4.7 of the 5.5 M parameters) onto a 7-level grid so that the file
     compresses (the values stay the reference's to choose: they are written and read back by its code, and the golden ids below
     come from exactly these weights);
  3. saves ``{'change_detector_state', 'speaker_state', 'model_cfg'}`` with the reference's ``utils.utils.save_checkpoint`` exactly
     as ``train_mimic.py:281-290`` does -- ``model_cfg`` is the reference's ``utils.attr_dict.AttrDict`` -- and xz-compresses the file
     -> ``ref_checkpoint_small.pt.xz``;
  4. re-loads it with the reference's ``load_checkpoint`` into FRESH reference modules and replays ``test_mimic.py:97-124`` (eval
     mode, greedy ``_sample``) on the seeded batches the product's ``test_mimic.py --synthetic`` draws
     -> ``ref_checkpoint_small_golden.npz`` (token ids, their log-probs, detector outputs).

The GPU test (tests/test_hip_data.py::test_checkpoint_written_by_the_reference_runs_through_test_mimic) feeds the file to the
product's ``test_mimic.py -p`` and must get the same answers.  Fixtures are data: weights + expected outputs, no reference source.
"""
import functools
import lzma
import os
import shutil
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G                                        # noqa: E402  (applies the shims, imports the reference)
from utils.attr_dict import AttrDict as RefAttrDict           # noqa: E402  (reference class: what train_mimic.py pickles)
from utils.utils import load_checkpoint, save_checkpoint     # noqa: E402  (reference: utils/utils.py:11-23)

from ekaid_amd.config import small_cfg                         # noqa: E402
from ekaid_amd.synthetic import make_batch                     # noqa: E402

N_WORDS, VOCAB, NODES, BATCHES, SEQ_LEN = 147, 148, 8, 2, 20
# interpreter shim, never applied to the reference: torch >= 2.6 defaults torch.load to weights_only=True, which cannot rebuild a
# dict subclass such as the reference's config class; the reference's load_cpu (utils.py:15) predates that default
torch.load = functools.partial(torch.load, weights_only=False)


def to_ref_attr(d):
    """our config tree -> the reference's own AttrDict class, recursively (plain values untouched)."""
    if isinstance(d, dict):
        out = RefAttrDict()
        for k, v in d.items():
            out[k] = to_ref_attr(v)
        return out
    return d


def snap(t, levels=3):
    """values onto the grid {-1, ..., 1} * max|t| / levels ... in place (2 * levels + 1 distinct values)."""
    step = float(t.abs().max()) / levels
    t.copy_(torch.round(t / step) * step)


def main():
    cfg = small_cfg(graph='all', seq_length=SEQ_LEN)
    cfg.data.test.batch_size = 4
    ref_cfg = to_ref_attr(cfg)
    cd, sp = G.build(ref_cfg, N_WORDS, VOCAB)
    with torch.no_grad():
        for m in (cd, sp):
            for k, p in m.named_parameters():
                if p.numel() > 100000:
                    snap(p)
    tmp = tempfile.mkdtemp()
    try:
        path = os.path.join(tmp, 'checkpoint_0.pt')
        # train_mimic.py:281-290
        speaker_state = sp.state_dict()
        chg_det_state = cd.state_dict()
        checkpoint = {'change_detector_state': chg_det_state, 'speaker_state': speaker_state, 'model_cfg': ref_cfg}
        save_checkpoint(checkpoint, path)
        out = os.path.join(HERE, 'ref_checkpoint_small.pt.xz')
        with open(path, 'rb') as f, lzma.open(out, 'wb', preset=9) as g:
            shutil.copyfileobj(f, g)
        print('ref_checkpoint_small.pt.xz  %.1f KB (raw %.1f KB)' % (os.path.getsize(out) / 1024, os.path.getsize(path) / 1024))
        # test_mimic.py:63-79: load into fresh modules
        ck = load_checkpoint(path)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    assert type(ck['model_cfg']) is RefAttrDict
    cd2, sp2 = G.build(ck['model_cfg'], N_WORDS, VOCAB)
    cd2.load_state_dict(ck['change_detector_state'])
    sp2.load_state_dict(ck['speaker_state'])
    cd2.eval()
    sp2.eval()
    D = ck['model_cfg'].model.change_detector.feat_dim - 2
    res = {}
    dev = torch.device('cpu')
    with torch.no_grad():
        for i in range(BATCHES):
            b = make_batch(B=ck['model_cfg'].data.test.batch_size, N=NODES, D=D, Lq=20, T=16, seq_length=SEQ_LEN, V=VOCAB, seed=977 + i)
            adj = [G.ref_mu.process_matrix(b[k], ck['model_cfg'], NODES, dev, type=t) for k, t in
                   (('d_adj', 'spatial'), ('q_adj', 'spatial'), ('d_sem', 'semantic'), ('q_sem', 'semantic'))]
            outs = cd2(b['input_1'], b['input_2'], adj[0], adj[1], adj[2], adj[3], b['d_bb'], b['q_bb'], b['question'],
                       setting=ck['model_cfg'].train.setting, graph='all')
            seq, seq_lp = sp2._sample(outs[3], outs[4], outs[5], b['labels'], ck['model_cfg'], sample_max=1)
            res['seq_%d' % i] = seq
            res['seq_logp_%d' % i] = seq_lp
            res['feat_diff_%d' % i] = outs[5]
            print('batch %d greedy ids:' % i, seq.tolist())
    G.save('ref_checkpoint_small_golden', **res)


if __name__ == '__main__':
    main()
