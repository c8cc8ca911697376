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
        per = 8 // world
        x = xs[rank * per:(rank + 1) * per]
        model(x).pow(2).sum().backward()
        gb.finish()
        out.append(gb.gflat.clone())
    names = gb.live_names(model.named_parameters())
    # Q-C gather: rows gathered in rank order, backward = sum over ranks of the local slice
    a = (torch.arange(6.).view(2, 3) + 10 * rank).requires_grad_(True)
    g = _AllGatherRows.apply(a, None)
    (g * torch.arange(6. * world).view(2 * world, 3)).sum().backward()
    q.put((rank, [o.tolist() for o in out], names, len(gb.buckets), g.detach().tolist(), a.grad.tolist()))
    dist.destroy_process_group()


@pytest.mark.parametrize('world,bucket_mb,overlap', [(2, 1e-5, True), (2, 32.0, True), (2, 1e-5, False), (8, 1e-5, True), (8, 32.0, True)])
def test_bucketed_allreduce_matches_single_process(world, bucket_mb, overlap):
    """world 2 and world 8 (the node size the path is built for: the first 8-GPU run must not also be the first 8-rank run)."""
    import socket
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = [ctx.Process(target=_worker, args=(r, world, port, bucket_mb, overlap, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(60)
    (r0, g0, names0, nb0, gath0, ga0), (r1, g1, names1, nb1, gath1, ga1) = res[0], res[-1]
    assert [r[0] for r in res] == list(range(world))
    assert all(r[1] == g0 for r in res), 'ranks disagree after all-reduce'
    assert all(r[2] == names0 and r[3] == nb0 for r in res)
    assert names0 == names1 and set(names0) == {'a.weight', 'a.bias', 'b.weight', 'b.bias'}
    assert names0[0].startswith('b.'), 'flat order must follow backward completion order (last layer first)'
    assert nb0 == (4 if bucket_mb < 1 else 1)
    # single-process reference: the summed gradient of both shards
    torch.manual_seed(100)
    xs = torch.randn(8, 6)
    m = Toy()
    per = 8 // world
    for r in range(world):
        m(xs[r * per:(r + 1) * per]).pow(2).sum().backward()
    want = {n: p.grad for n, p in m.named_parameters() if p.grad is not None}
    flat = torch.tensor(g0[-1])
    off = 0
    for n in names0:
        k = want[n].numel()
        assert torch.allclose(flat[off:off + k], want[n].reshape(-1), atol=1e-5), n
        off += (k + 7) // 8 * 8          # every tensor starts on an 8-element group (ekaid_amd/dp.py)
    # gather semantics: rows in rank order on every rank; backward = the sum over ranks of the gradient of the rank's own rows
    want_g = [[10. * r + 3 * i + j for j in range(3)] for r in range(world) for i in range(2)]
    assert all(r[4] == want_g for r in res)
    for r in res:
        assert r[5] == [[world * (6. * r[0] + 3 * i + j) for j in range(3)] for i in range(2)]


def _ragged_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ekaid_amd.dp import GradBuckets, loss_denominator
    torch.manual_seed(5)
    V, T = 7, 6
    model = nn.Linear(4, V)
    x = torch.randn(4, T, 4)
    target = torch.randint(0, V, (4, T))
    mask = torch.zeros(4, T)
    for b, n in enumerate((6, 2, 3, 1)):         # ragged answers: rank 0 holds 8 target tokens, rank 1 holds 4
        mask[b, :n] = 1
    sl = slice(rank * 2, rank * 2 + 2)
    gb = GradBuckets(list(model.parameters()), bucket_mb=1e-5)
    gb.zero_grad()
    logp = torch.log_softmax(model(x[sl]), -1)
    num = -(logp.gather(2, target[sl].unsqueeze(2)).squeeze(2) * mask[sl]).sum()
    denom = loss_denominator(mask[sl].sum())      # (global mask sum) / world, one tiny all-reduce
    (num / denom).backward()
    gb.finish()                                   # SUM over ranks; the optimizer applies 1/world
    naive = -(logp.detach().gather(2, target[sl].unsqueeze(2)).squeeze(2) * mask[sl]).sum() / mask[sl].sum()
    q.put((rank, (gb.gflat / world).tolist(), gb.live_names(model.named_parameters()), float(denom), float(num.detach() / denom),
           float(naive)))
    dist.destroy_process_group()


def test_ragged_masks_reproduce_the_global_batch_loss():
    """Ranks holding answers of different lengths: dividing each rank's numerator by (global mask sum / world) makes the
    averaged gradient equal to the single-process gradient of the reference loss over the global batch
    (utils/utils.py:213-214); plain per-rank mean losses would not."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29700 + os.getpid() % 50
    procs = [ctx.Process(target=_ragged_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
    (_, g0, names, d0, l0, n0), (_, g1, _, d1, l1, n1) = res
    assert g0 == g1 and d0 == d1 == 6.0           # (8 + 4) / 2
    torch.manual_seed(5)
    V, T = 7, 6
    model = nn.Linear(4, V)
    x = torch.randn(4, T, 4)
    target = torch.randint(0, V, (4, T))
    mask = torch.zeros(4, T)
    for b, n in enumerate((6, 2, 3, 1)):
        mask[b, :n] = 1
    logp = torch.log_softmax(model(x), -1)
    loss = -(logp.gather(2, target.unsqueeze(2)).squeeze(2) * mask).sum() / mask.sum()
    loss.backward()
    want = {n: p.grad for n, p in model.named_parameters()}
    flat = torch.tensor(g0)
    off = 0
    for n in names:
        k = want[n].numel()
        assert torch.allclose(flat[off:off + k], want[n].reshape(-1), atol=1e-6), n
        off += (k + 7) // 8 * 8
    assert (l0 + l1) / 2 == pytest.approx(float(loss), rel=1e-6)
    assert abs((n0 + n1) / 2 - float(loss)) > 1e-3, 'the naive average of per-rank means must differ (else the test is vacuous)'
