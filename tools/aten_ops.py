"""Which ATen kernels does one eager training step still launch, and from where?
torch.profiler over one step (B given on the command line): device kernels whose name is not ekaid::*, grouped by the
CPU op (with input shapes) that launched them.   usage: python tools/aten_ops.py [B]"""
import os
import sys
from collections import Counter

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ekaid_amd.formula_init import build_models  # noqa: E402
from ekaid_amd import hip, ops  # noqa: E402
from ekaid_amd.config import bench_cfg  # noqa: E402
from ekaid_amd.dp import FlatAdam  # noqa: E402
from ekaid_amd.engine import TrainStep  # noqa: E402
from ekaid_amd.models.dynamic_speaker_change_pos import DynamicSpeaker  # noqa: E402
from ekaid_amd.synthetic import make_batch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
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
for _ in range(3):
    step(b, T)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    step(b, T)
    torch.cuda.synchronize()
ev = prof.events()
cnt, dur = Counter(), Counter()
kernels = Counter()
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CUDA:
        kernels[e.name[:60]] += 1
for e in ev:
    if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
        continue
    for k in e.kernels:
        if 'ekaid' in k.name:
            continue
        key = (e.name, str(e.input_shapes)[:90], k.name[:50])
        cnt[key] += 1
        dur[key] += k.duration
print('device kernels in one step: %d launches, %d distinct' % (sum(kernels.values()), len(kernels)))
print('non-ekaid kernels by launching op:')
for key, n in sorted(cnt.items(), key=lambda kv: -dur[kv[0]]):
    print('%4d x %8.1f us  %-28s %-90s %s' % (n, dur[key], key[0], key[1], key[2]))
