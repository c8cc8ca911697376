import sys, os
sys.path.insert(0, '/root/repo'); os.chdir('/root/repo')
import torch
from ekaid_amd.config import bench_cfg
from ekaid_amd.engine import detector_forward
from ekaid_amd.formula_init import build_models
from ekaid_amd.synthetic import make_batch
dev = torch.device('cuda:0')
cfg = bench_cfg(D=2048, seq_length=90, graph='all')
cd0, sp0 = build_models(cfg, 147, 148, dev, train=False)
cd0.eval(); sp0.eval()
b64 = {k: v.to(dev) for k, v in make_batch(B=64, seed=77, N=36, D=2048, Lq=20, T=16, seq_length=90, V=148).items()}
with torch.no_grad():
    outs = detector_forward(cfg, cd0, b64, 'all')
    for _ in range(100):
        sp0._sample(outs[3], outs[4], outs[5], None, cfg, sample_max=1)
    torch.cuda.synchronize()
