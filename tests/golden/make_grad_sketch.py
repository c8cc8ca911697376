"""Whole-tensor gradient sketches of the REFERENCE at config 1 (all five graph variants) and at the real-data shapes of
tests/test_hip_model.py::test_real_data_shapes_ragged_answers_and_zero_rows -> tests/golden/f8_grad_sketch_*.npz.

    python tests/golden/make_grad_sketch.py          (build container only: imports /root/reference via make_golden.py)

The F1 fixtures hold 10 numbers per gradient tensor (norm, sum, 8 samples); a sketch holds, for each of 16 contiguous chunks of the
flattened gradient, its l2 norm and 4 random-sign projections (golden_util.tensor_sketch): every element of every gradient of the
reference takes part.  tests/test_oracle_golden.py compares the oracle with these (CPU), tests/test_hip_model.py the HIP path (GPU).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G                                  # noqa: E402  (shims + reference import)

from ekaid_amd.config import small_cfg                   # noqa: E402
from ekaid_amd.synthetic import make_batch               # noqa: E402
from golden_util import sketch_named                     # noqa: E402


def real_shape_case():
    """The inputs of test_real_data_shapes_ragged_answers_and_zero_rows (kept in step with tests/common.real_shape_case)."""
    sys.path.insert(0, os.path.join(G.ROOT, 'tests'))
    from common import real_shape_case as case
    return case()


def main():
    for graph in ('semantic', 'spatial', 'implicit', 'all', 'i+s'):
        cfg = small_cfg(graph=graph)
        cd, sp = G.build(cfg, 30, 31)
        loss, sp_loss, outs, logp, logpos, _ = G.ref_step(cfg, cd, sp, make_batch(**G.SMALL), 'mode2', graph)
        loss.backward()
        names, sk = sketch_named(G.named_grads(cd, sp))
        G.save('f8_grad_sketch_' + graph.replace('+', 'p'), loss=loss, grad_names=np.array(names), grad_sketch=sk)
    cfg, batch = real_shape_case()
    cd, sp = G.build(cfg, 30, 31)
    loss, sp_loss, outs, logp, logpos, _ = G.ref_step(cfg, cd, sp, batch, 'mode2', 'all')
    loss.backward()
    names, sk = sketch_named(G.named_grads(cd, sp))
    G.save('f8_grad_sketch_real_shapes', loss=loss, logp=logp, feat_diff=outs[5], grad_names=np.array(names), grad_sketch=sk)


if __name__ == '__main__':
    main()
