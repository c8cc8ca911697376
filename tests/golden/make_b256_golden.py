"""The REFERENCE at the metric's own batch size: BASELINE.json configs[2] shapes (36 regions x 2 images, 2048-d ROI features, 20-token
questions, 16 decode steps), batch 256, full-width 87 M-parameter model with the closed-form weights, eval forward.

    python tests/golden/make_b256_golden.py          (build container only; ~ 2 TFLOP on the host cores: a few minutes)

Quirk Q-C (language_model.py:149: softmax over the BATCH axis, flat re-read) makes every sample's question vector depend on the
batch size itself, so B = 256 is a case of its own, not B = 64 four times.  Stored (tests/golden/f9_b256.npz, ~ 150 KB): loss, the
question vectors' sketch and first 4 rows, arg-max ids of all 256 x 16 decode positions, the log-prob of the arg-max and of the label at
every position, the full (4, 16, 148) log-prob block of samples 0, 85, 170, 255, detector outputs of those samples."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G                                  # noqa: E402

from ekaid_amd.config import bench_cfg                   # noqa: E402
from ekaid_amd.synthetic import make_batch               # noqa: E402
from golden_util import tensor_sketch                    # noqa: E402

ROWS = [0, 85, 170, 255]


def main():
    cfg = bench_cfg(D=2048, seq_length=90)
    cd, sp = G.build(cfg, 147, 148)
    batch = make_batch(B=256, N=36, D=2048, Lq=20, T=16, seq_length=90, V=148)
    with torch.no_grad():
        q = cd.q_att(cd.q_emb.forward_all(cd.w_emb(batch['question'])))
        loss, sp_loss, outs, logp, logpos, _ = G.ref_step(cfg, cd, sp, batch, 'mode2', 'all')
    lp = logp[:, :16]
    am = lp.argmax(-1)
    tgt = batch['labels'][:, 1:17]
    G.save('f9_b256', loss=loss, speaker_loss=sp_loss, q_sketch=tensor_sketch(q), q_rows=q[:4],
           argmax=am.to(torch.int16), logp_argmax=lp.gather(-1, am[..., None])[..., 0], logp_label=lp.gather(-1, tgt[..., None])[..., 0],
           logp_rows=lp[ROWS], feat_diff_rows=outs[5][ROWS], att_bef_rows=outs[1][ROWS], feat_diff_sketch=tensor_sketch(outs[5]),
           logp_sketch=tensor_sketch(lp))


if __name__ == '__main__':
    main()
