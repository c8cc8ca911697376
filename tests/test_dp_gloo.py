"""Data-parallel plumbing on CPU with gloo, world_size 2 (the GPU arithmetic is covered by -m gpu tests):
GradBuckets discovers the live parameter set, lays gradients out flat in completion order, all-reduces
bucket by bucket (overlapped) and leaves every rank with the SUM; dead parameters are neither stored nor
communicated; the global-batch Q-C gather reproduces the single-process result."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Toy(nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(3)
        self.a = nn.Linear(6, 5)
        self.dead = nn.Linear(5, 5)            # never used: stays without gradient, like SSRE.* in the reference
        self.b = nn.Linear(5, 4)
        self.frozen = nn.Parameter(torch.ones(3), requires_grad=False)

    def forward(self, x):
        return self.b(torch.tanh(self.a(x)))


def _worker(rank, world, port, bucket_mb, overlap, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ekaid_amd.dp import GradBuckets, _AllGatherRows
    torch.manual_seed(100)
    xs = torch.randn(8, 6)
    model = Toy()
    gb = GradBuckets(list(model.parameters()), bucket_mb=bucket_mb, overlap=overlap)
    out = []
    for step in range(3):
        gb.zero_grad()
        x = xs[rank * 4:(rank + 1) * 4]
        model(x).pow(2).sum().backward()
        gb.finish()
        out.append(gb.gflat.clone())
    names = gb.live_names(model.named_parameters())
    # Q-C gather: rows gathered in rank order, backward = sum over ranks of the local slice
    a = (torch.arange(6.).view(2, 3) + 10 * rank).requires_grad_(True)
    g = _AllGatherRows.apply(a, None)
    (g * torch.arange(12.).view(4, 3)).sum().backward()
    q.put((rank, [o.tolist() for o in out], names, len(gb.buckets), g.detach().tolist(), a.grad.tolist()))
    dist.destroy_process_group()


@pytest.mark.parametrize('bucket_mb,overlap', [(1e-5, True), (32.0, True), (1e-5, False)])
def test_bucketed_allreduce_matches_single_process(bucket_mb, overlap):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29600 + int(bucket_mb > 1) * 7 + int(overlap) * 3 + os.getpid() % 50
    procs = [ctx.Process(target=_worker, args=(r, 2, port, bucket_mb, overlap, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
    (r0, g0, names0, nb0, gath0, ga0), (r1, g1, names1, nb1, gath1, ga1) = res
    assert g0 == g1, 'ranks disagree after all-reduce'
    assert names0 == names1 and set(names0) == {'a.weight', 'a.bias', 'b.weight', 'b.bias'}
    assert names0[0].startswith('b.'), 'flat order must follow backward completion order (last layer first)'
    assert nb0 == (4 if bucket_mb < 1 else 1)
    # single-process reference: the summed gradient of both shards
    torch.manual_seed(100)
    xs = torch.randn(8, 6)
    m = Toy()
    m(xs[:4]).pow(2).sum().backward()
    m(xs[4:]).pow(2).sum().backward()
    want = {n: p.grad for n, p in m.named_parameters() if p.grad is not None}
    flat = torch.tensor(g0[-1])
    off = 0
    for n in names0:
        k = want[n].numel()
        assert torch.allclose(flat[off:off + k], want[n].reshape(-1), atol=1e-5), n
        off += (k + 7) // 8 * 8          # every tensor starts on an 8-element group (ekaid_amd/dp.py)
    # gather semantics
    assert gath0 == gath1 == [[0., 1., 2.], [3., 4., 5.], [10., 11., 12.], [13., 14., 15.]]
    assert ga0 == [[0., 2., 4.], [6., 8., 10.]] and ga1 == [[12., 14., 16.], [18., 20., 22.]]
