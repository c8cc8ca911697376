"""Why does a full-width long-answer case differ element-wise between the HIP path and the CPU oracle?  Prints, per gradient tensor of
one f10 case: HIP vs float64 truth (sketch), oracle-on-this-host vs truth, reference float32 vs truth, HIP vs oracle element-wise
(max / max and l2-relative), under the sequence / GEMM modes given.      python tools/grad_diag.py T40 [stepwise] [f32 | split3 | f16x2] [allx2]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from common import build_models, oracle_params, real_full_case, wide_case      # noqa: E402
from golden_util import load_golden, sketch_errors, tensor_sketch            # noqa: E402
from test_hip_model import hip_step, named_grads                              # noqa: E402
from ekaid_amd import hip, ops                                                # noqa: E402
from oracle import ekaid_oracle as O                                          # noqa: E402


def main():
    case = sys.argv[1] if len(sys.argv) > 1 else 'T40'
    hip.load()
    if 'stepwise' in sys.argv:
        hip.check(hip.lib().ekaid_set_seq_mode(0), 'seq mode')
    if 'f32' in sys.argv:
        ops.set_gemm_mode(ops.GEMM_F32_MFMA)
    if 'split3' in sys.argv:
        ops.set_gemm_mode(ops.GEMM_BF16_SPLIT)
    if 'f16x2' in sys.argv:
        ops.set_gemm_mode(ops.GEMM_F16X2)
    if 'allx2' in sys.argv:
        ops.F16X2_MIN_WORK = 0
    torch.set_num_threads(int(os.environ.get('DIAG_THREADS', '16')))
    g = load_golden('f10_' + case)
    cfg, batch = real_full_case() if case == 'real_full' else wide_case(int(case[1:]))
    cd, sp = build_models(cfg, 147, 148, 'cuda')
    loss, sp_loss, outs, logp, logpos = hip_step(cfg, cd, sp, batch, 'mode2', 'all')
    loss.backward()
    torch.cuda.synchronize()
    got = dict(named_grads(cd, sp))
    Pcd, Psp = oracle_params(cd, True), oracle_params(sp, True)
    loss_o, res = O.train_loss(Pcd, Psp, cfg, batch, 'mode2', 'all')
    loss_o.backward()
    orc = {('cd.' + k): v.grad for k, v in Pcd.items() if getattr(v, 'grad', None) is not None}
    orc.update({('sp.' + k): v.grad for k, v in Psp.items() if getattr(v, 'grad', None) is not None})
    truth = g['grad_sketch_f64']
    e_ref = sketch_errors(g['grad_names'], truth, lambda i, k: g['grad_sketch'][i])
    e_hip = sketch_errors(g['grad_names'], truth, lambda i, k: tensor_sketch(got[k]) if got.get(k) is not None else None)
    e_orc = sketch_errors(g['grad_names'], truth, lambda i, k: tensor_sketch(orc[k]) if orc.get(k) is not None else None)
    T = int(g['steps'])
    print('case %s  loss hip %.8f oracle %.8f ref %.8f | logp max|d| hip-ref %.2e oracle-ref %.2e' % (
        case, float(loss), float(loss_o), float(g['loss']), float((logp[:, :T].cpu() - torch.from_numpy(g['logp'])).abs().max()),
        float((res['logp'][:, :T].detach() - torch.from_numpy(g['logp'])).abs().max())))
    print('%-70s %9s %9s %9s | %9s %9s' % ('tensor', 'ref-f64', 'hip-f64', 'orc-f64', 'hip-orc/mx', 'hip-orc l2'))
    for k in json.loads(str(g['grad_names'])):
        if k not in e_hip or k not in orc:
            continue
        a, b = got[k].detach().double().cpu(), orc[k].detach().double()
        mx = float((a - b).abs().max() / b.abs().max().clamp_min(1e-300))
        l2 = float((a - b).norm() / b.norm().clamp_min(1e-300))
        print('%-70s %9.2e %9.2e %9.2e | %9.2e %9.2e' % (k[-70:], e_ref.get(k, float('nan')), e_hip[k], e_orc.get(k, float('nan')), mx, l2))


if __name__ == '__main__':
    main()
