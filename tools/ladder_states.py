#!/usr/bin/env python3
"""Which of its two speeds does a replayed small step run at, and does it change in time?  One process (world-1 RCCL communicator made
first, like bench.py --ladder-child), the 32- and 128-pair steps captured once, then rounds of 40 replays each with the shader / memory
clocks and the package power read from sysfs in between; halfway, 30 eager 256-pair steps (the headline's load) are run to see whether
heavy work moves the state (it does: the package power limit pulls the shader clock down within a second of sustained 256-pair
steps, and the replayed small steps that follow run at that clock until the chip has idled).   usage: ladder_states.py [rounds]"""
import glob
import os
import socket
import sys
import time

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
os.chdir(root)
import torch                                        # noqa: E402
import torch.distributed as dist                    # noqa: E402


def sysfs():
    out = []
    for f in sorted(glob.glob('/sys/class/drm/card*/device/pp_dpm_sclk'))[:1]:
        d = os.path.dirname(f)
        for name in ('pp_dpm_sclk', 'pp_dpm_mclk'):
            try:
                cur = [ln.strip() for ln in open(os.path.join(d, name)) if '*' in ln]
                out.append('%s=%s' % (name[7:], cur[0] if cur else '?'))
            except OSError as e:
                out.append('%s=err(%s)' % (name, e.errno))
        for hw in glob.glob(os.path.join(d, 'hwmon/hwmon*/power1_average')) + glob.glob(os.path.join(d, 'hwmon/hwmon*/power1_input')):
            try:
                out.append('power=%.0fW' % (int(open(hw).read()) / 1e6))
            except (OSError, ValueError):
                pass
        for hw in glob.glob(os.path.join(d, 'hwmon/hwmon*/freq1_input')):
            try:
                out.append('freq1=%.0fMHz' % (int(open(hw).read()) / 1e6))
            except (OSError, ValueError):
                pass
    return ' '.join(out) or 'no sysfs'


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    dev = torch.device('cuda:0')
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    from ekaid_amd import hip, ops
    from ekaid_amd.config import bench_cfg
    from ekaid_amd.dp import FlatAdam
    from ekaid_amd.engine import TrainStep
    from ekaid_amd.formula_init import build_models
    from ekaid_amd.models.dynamic_speaker_change_pos import DynamicSpeaker
    from ekaid_amd.synthetic import make_batch, slice_batch
    hip.load()
    hip.check(hip.lib().ekaid_init(0), 'ekaid_init')
    cfg = bench_cfg(D=2048, seq_length=90, graph='all')
    cd, sp = build_models(cfg, 147, 148, dev, train=True)
    opt = FlatAdam([cd, sp], lr=1e-4, betas=(0.9, 0.999), eps=1e-8)
    full = make_batch(B=256, seed=1238, N=36, D=2048, Lq=20, T=16, seq_length=90, V=148)
    T = DynamicSpeaker.count_steps(full['labels'], 90)
    ops.manual_seed(1238)
    opt.force_comm = True
    b256 = {k: v.to(dev) for k, v in full.items()}
    eager = TrainStep(cfg, cd, sp, opt, 'all')
    steps = {}
    for per in (32, 128):
        b = {k: v.to(dev) for k, v in slice_batch(full, 0, per).items()}
        st = TrainStep(cfg, cd, sp, opt, 'all', use_graph=True, comm='captured')
        st(b, T)
        steps[per] = (st, b)
    torch.cuda.synchronize()

    def timed(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    print('start', sysfs(), flush=True)
    for r in range(rounds):
        if r == rounds // 2:
            for c in range(12):                      # 240 eager 256-pair steps in chunks of 20: the headline's sustained load
                ms = timed(lambda: eager(b256, T), 20)
                print('--- eager 256-pair steps %3d-%3d: %.2f ms each   %s' % (c * 20, c * 20 + 19, ms, sysfs()), flush=True)
        row = []
        for per in (32, 128):
            st, b = steps[per]
            row.append('%d pairs %.3f ms' % (per, timed(lambda: st(b, T), 40)))
        print('round %2d  %s   %s' % (r, '   '.join(row), sysfs()), flush=True)
        if r % 4 == 3:
            time.sleep(2.0)
            print('     (after 2 s idle) %s' % sysfs(), flush=True)
    for st, _ in steps.values():
        st.close()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
