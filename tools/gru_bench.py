#!/usr/bin/env python3
"""Question GRU (H = 1024, 20 steps): the step-wise chain (2 launches per step and direction) against the persistent kernels
(csrc/persist_gru.hip), forward and backward, microseconds per step.   python tools/gru_bench.py [batch ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ekaid_amd import hip  # noqa: E402


def main():
    hip.load()
    L = hip.lib()
    H, T = 1024, 20
    st = hip.stream()
    for B in [int(x) for x in sys.argv[1:]] or [32, 64, 256]:
        g = torch.Generator().manual_seed(B)
        gi = (torch.randn(B, T, 3 * H, generator=g) * 0.7).cuda()
        w_hh = (torch.randn(3 * H, H, generator=g) * H ** -0.5).cuda()
        b_hh = (torch.randn(3 * H, generator=g) * 0.1).cuda()
        dout = (torch.randn(B, T, H, generator=g) * 0.1).cuda()
        hall = torch.zeros(T + 1, B, H, device='cuda')
        save = torch.empty(T, B, 4 * H, device='cuda')
        dgi, dgh = torch.empty(B, T, 3 * H, device='cuda'), torch.empty(T, B, 3 * H, device='cuda')
        res = {}
        for mode in (0, 1):
            hip.check(L.ekaid_set_seq_mode(mode), 'set_seq_mode')
            nf, nb = L.ekaid_gru_seq_workspace_bytes(B, T, H, 0), L.ekaid_gru_seq_workspace_bytes(B, T, H, 1)
            ws = torch.empty(max(nf, nb, 256), dtype=torch.uint8, device='cuda')

            def fwd():
                hip.check(L.ekaid_gru_seq_fwd(gi.data_ptr(), w_hh.data_ptr(), b_hh.data_ptr(), B, T, H, hall.data_ptr(), save.data_ptr(), ws.data_ptr(), nf, st), 'fwd')

            def bwd():
                hip.check(L.ekaid_gru_seq_bwd(dout.data_ptr(), w_hh.data_ptr(), hall.data_ptr(), save.data_ptr(), B, T, H, dgi.data_ptr(), dgh.data_ptr(), ws.data_ptr(), nb, st), 'bwd')
            for name, fn in (('fwd', fwd), ('bwd', bwd)):
                for _ in range(3):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                res[(mode, name)] = e0.elapsed_time(e1) / 20 / T * 1e3
        print('GRU B=%d: forward %.1f -> %.1f us per step, backward %.1f -> %.1f us per step (step-wise -> persistent)' % (
            B, res[(0, 'fwd')], res[(1, 'fwd')], res[(0, 'bwd')], res[(1, 'bwd')]))


if __name__ == '__main__':
    main()
