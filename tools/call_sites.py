"""Where do the launches of given C-ABI entry points come from in one eager training step?
Wraps the named ``ekaid_*`` functions of the loaded library, runs one step (after warm-up) and prints, per entry point, the
Python call sites (nearest frames inside ekaid_amd/) with counts and the size argument when there is one.
usage: python tools/call_sites.py [B] name [name ...]      e.g.  call_sites.py 256 ekaid_dropout ekaid_sum_n ekaid_axpby"""
import os
import sys
import traceback
from collections import Counter

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ekaid_amd import hip, ops  # noqa: E402
from ekaid_amd.config import bench_cfg  # noqa: E402
from ekaid_amd.dp import FlatAdam  # noqa: E402
from ekaid_amd.engine import TrainStep  # noqa: E402
from ekaid_amd.formula_init import build_models  # noqa: E402
from ekaid_amd.models.dynamic_speaker_change_pos import DynamicSpeaker  # noqa: E402
from ekaid_amd.synthetic import make_batch  # noqa: E402

B = int(sys.argv[1])
names = sys.argv[2:]
SHAPE = dict(N=36, D=1024, seq_length=16)
hip.load()
dev = torch.device('cuda', 0)
cfg = bench_cfg(D=SHAPE['D'], seq_length=SHAPE['seq_length'], graph='all')
cd, sp = build_models(cfg, 147, 148, dev, train=True)
opt = FlatAdam([cd, sp], lr=1e-4)
hb = make_batch(B=B, seed=1238, **SHAPE)
T = DynamicSpeaker.count_steps(hb['labels'], SHAPE['seq_length'])
b = {k: v.to(dev) for k, v in hb.items()}
ops.manual_seed(1)
step = TrainStep(cfg, cd, sp, opt, 'all')
for _ in range(2):
    step(b, T)
torch.cuda.synchronize()

L = hip.lib()
sites = {n: Counter() for n in names}


class _Wrap:
    """Stands in for the ctypes function object (the layers call ``hip.lib().name(...)``)."""

    def __init__(self, name, fn):
        self.name, self.fn = name, fn

    def __call__(self, *a):
        fr = [f for f in traceback.extract_stack()[:-1] if 'ekaid_amd' in f.filename and not f.filename.endswith('hip.py')]
        where = ' < '.join('%s:%d %s' % (os.path.basename(f.filename), f.lineno, f.name) for f in fr[-3:][::-1])
        big = [x for x in a if isinstance(x, int) and 64 < x < (1 << 40)]
        sites[self.name][(where, tuple(big[:3]))] += 1
        return self.fn(*a)


class _Lib:
    def __init__(self, lib):
        self._lib = lib
        for n in names:
            setattr(self, n, _Wrap(n, getattr(lib, n)))

    def __getattr__(self, k):
        return getattr(self._lib, k)


hip._lib = _Lib(L) if hasattr(hip, '_lib') else None
assert hip._lib is not None, 'hip.py keeps the loaded library in hip._lib'
step(b, T)
torch.cuda.synchronize()
for n in names:
    print('== %s: %d calls' % (n, sum(sites[n].values())))
    for (where, args), c in sorted(sites[n].items(), key=lambda kv: -kv[1]):
        print('  %3d x  %s   %s' % (c, where, args))
