"""Greedy decoding at the production widths (B = 64, 90 steps): step-wise chain vs the persistent launch, by steps per C-ABI call."""
import sys, time, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from ekaid_amd import hip, speaker_seq
from ekaid_amd.config import bench_cfg
from ekaid_amd.formula_init import build_models
hip.load(); L = hip.lib()
cfg = bench_cfg(D=256, seq_length=90)
cd, sp = build_models(cfg, 147, 148, 'cuda', train=False)
B = 64; A = cfg.model.speaker.input_dim
g = torch.Generator().manual_seed(1)
bef, aft, diff = [torch.randn(B, A, generator=g).cuda() for _ in range(3)]
for mode in (0, 1):
    L.ekaid_set_seq_mode(mode)
    for poll in (8, 30, 90):
        for _ in range(2):
            seq, lp, mw = speaker_seq.greedy_decode(sp, bef, aft, diff, poll=poll)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5):
            seq, lp, mw = speaker_seq.greedy_decode(sp, bef, aft, diff, poll=poll)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
        print('mode %d poll %s: %.3f ms per decode, %d steps run, %.1f us/step' % (mode, poll, dt * 1e3, len(mw), dt * 1e6 / len(mw)), flush=True)
    hip.seq_status_check('greedy')
# every row stops at step 1 (the end token gets a large bias): what one call costs beside the steps
with torch.no_grad():
    sp.logit.bias[0] += 60.0
for mode in (0, 1):
    L.ekaid_set_seq_mode(mode)
    for _ in range(2):
        seq, lp, mw = speaker_seq.greedy_decode(sp, bef, aft, diff)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10):
        seq, lp, mw = speaker_seq.greedy_decode(sp, bef, aft, diff)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    print('early stop, mode %d default poll: %.3f ms per decode, %d steps run' % (mode, dt * 1e3, len(mw)), flush=True)
    hip.seq_status_check('greedy')
