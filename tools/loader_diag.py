import time, json, os, sys, tempfile
sys.path.insert(0,'.')
import torch
from ekaid_amd.datasets.generate import make_dataset, point_cfg_at
from ekaid_amd.config import cfg, default_cfg, merge_into, MIMIC_YAML
d=tempfile.mkdtemp()
make_dataset(d, n_images=512, n_pairs=3200, nodes=52, D=1024)
cfg.clear(); cfg.update(default_cfg()); merge_into(json.loads(json.dumps(MIMIC_YAML)), cfg); point_cfg_at(cfg,d)
from ekaid_amd.datasets import create_dataset
from ekaid_amd.datasets.rcc_dataset_pos_mimic import compact_batch, DevicePrefetcher
ds, loader = create_dataset(cfg,'train')
print('threads', torch.get_num_threads(), 'cpus', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
it=iter(loader)
for i in range(6):
    t0=time.perf_counter(); b=next(it); t1=time.perf_counter(); c=compact_batch(b); t2=time.perf_counter()
    print('batch %d: assemble %.1f ms compact %.1f ms' % (i,(t1-t0)*1e3,(t2-t1)*1e3))
pf=DevicePrefetcher(loader,'cuda',timing=True)
t0=time.perf_counter()
for i,bb in enumerate(pf):
    torch.cuda.synchronize(); time.sleep(0.03)
    if i==10: break
print('prefetch loop per batch %.1f ms' % ((time.perf_counter()-t0)/11*1e3), pf.stats())
import numpy as np
x=torch.randn(64,52,1024); pin=torch.empty(64,52,1024).pin_memory()
t0=time.perf_counter()
for _ in range(5): pin.copy_(x)
print('pageable->pinned 13.6MB: %.2f ms' % ((time.perf_counter()-t0)/5*1e3))
