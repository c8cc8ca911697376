"""One training iteration of the difference-VQA path (reference ``train_mimic.py:209-269``) as a reusable object, eager or
replayed from a captured hipGraph.

Eager: ~700 kernel launches per step at 256 pairs (round 3; the decoder and GRU time loops are issued from C,
csrc/sequence.hip), ~8 ms of host time -- invisible at 256 pairs per GPU (30 ms of GPU work) but the floor of the
strong-scaling points (32 pairs per GPU on 8 GPUs).  The C ABI is allocation- and synchronisation-free, so the
whole step -- adjacency expansion, detector, decoder, loss, backward, Adam -- is captured ONCE per (batch, nodes, decode steps,
scheduled-sampling rate) key with ``torch.cuda.graph`` (torch's allocator and autograd run at capture time only) and replayed
with a single launch:
  * inputs live in static buffers (one device copy per step);
  * dropout: every site's seed is frozen in the graph, the device-resident epoch word advanced at the head of the graph
    gives every replay fresh masks (include/ekaid_hip.h: ekaid_set_dropout_epoch_ptr);
  * Adam's step counter / learning rate live on the device (ekaid_adam_step_dev);
  * data parallel: ``comm='captured'`` (the default of bench.py / train_mimic.py at world > 1) keeps the hook-driven bucketed
    all-reduce INSIDE the graph, overlapped with the rest of backward exactly as the eager hooks overlap it; ``comm='eager'``
    issues the bucketed RCCL calls + the Adam kernel after the replay (serialised; the fallback for backends that cannot be
    captured, e.g. gloo);
  * scheduled sampling: ``speaker.ss_prob`` is a by-value kernel argument AND selects the decoder path (fused teacher-forced
    sequence node at 0, step-wise sampling path above), so it is part of the graph key: a rate change recaptures.
"""
import torch

from . import hip, ops
from .utils.mimic_utils import process_matrix
from .utils.utils import LanguageModelCriterion, total_loss

ADJ_KEYS = (('d_adj', 'spatial'), ('q_adj', 'spatial'), ('d_sem', 'semantic'), ('q_sem', 'semantic'))
INPUT_KEYS = ('input_1', 'input_2', 'd_adj', 'q_adj', 'd_sem', 'q_sem', 'd_bb', 'q_bb', 'question', 'labels', 'masks')


def detector_forward(cfg, cd, b, graph='all', setting='mode2', mark=None):
    """adjacency expansion -> ChangeDetector: the six detector outputs (reference ``test_mimic.py:109-116``)."""
    N = b['input_1'].shape[1]
    dev = b['input_1'].device
    # the one-hot adjacency of the main and the reference image land in the two halves of one buffer: the detector stacks
    # them along the batch axis (ops.stack2) without a copy
    Bsz = b['input_1'].shape[0]
    adj = []
    for (kd, ty), (kq, _) in ((ADJ_KEYS[0], ADJ_KEYS[1]), (ADJ_KEYS[2], ADJ_KEYS[3])):
        L = cfg.model.change_detector.spa_label_num if ty == 'spatial' else cfg.model.change_detector.sem_label_num
        pair = torch.empty(2 * Bsz, N, N, L, device=dev, dtype=torch.float32)
        adj.append(process_matrix(b[kd], cfg, N, dev, type=ty, out=pair[:Bsz]))
        adj.append(process_matrix(b[kq], cfg, N, dev, type=ty, out=pair[Bsz:]))
    if mark is not None:
        mark('adjacency')
    return cd(b['input_1'], b['input_2'], adj[0], adj[1], adj[2], adj[3], b['d_bb'], b['q_bb'], b['question'],
              setting=setting, graph=graph)


def forward_loss(cfg, cd, sp, b, T, graph='all', setting='mode2', denom=None, marks=None):
    """adjacency expansion -> ChangeDetector -> teacher-forced DynamicSpeaker -> LM loss + attention regulariser.
    ``b``: dict of tensors (ekaid_amd/synthetic.py keys).  Returns (total loss, speaker loss, detector outputs, logp)."""
    def mark(name):
        if marks is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append((name, e))
    outs = detector_forward(cfg, cd, b, graph, setting, mark)
    mark('detector_fwd')
    logp, _ = sp._forward(outs[3], outs[4], outs[5], b['labels'], num_steps=T)
    sp_loss = LanguageModelCriterion()(logp, b['labels'][:, 1:], b['masks'][:, 1:], denom=denom)
    loss = total_loss(sp_loss, outs[1], outs[2])
    mark('speaker_fwd_loss')
    return loss, sp_loss, outs, logp


DETECTOR_KEYS = ('input_1', 'input_2', 'd_adj', 'q_adj', 'd_sem', 'q_sem', 'd_bb', 'q_bb', 'question')


def graph_node_counts(g):
    """{'kernel': n, 'memset': n, 'memcpy': n, 'other': n} of a captured graph created with ``CUDAGraph(keep_graph=True)`` (the launches
    one replay issues), read from the raw hipGraph_t with hipGraphGetNodes / hipGraphNodeGetType; None where that is unavailable."""
    import ctypes
    try:
        raw = g.raw_cuda_graph()
        rt = ctypes.CDLL('libamdhip64.so')
        rt.hipGraphGetNodes.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
        rt.hipGraphGetNodes.restype = ctypes.c_int
        rt.hipGraphNodeGetType.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
        rt.hipGraphNodeGetType.restype = ctypes.c_int
        n = ctypes.c_size_t(0)
        if rt.hipGraphGetNodes(ctypes.c_void_p(raw), None, ctypes.byref(n)) != 0 or n.value == 0:
            return None
        nodes = (ctypes.c_void_p * n.value)()
        if rt.hipGraphGetNodes(ctypes.c_void_p(raw), nodes, ctypes.byref(n)) != 0:
            return None
        out = {'kernel': 0, 'memcpy': 0, 'memset': 0, 'other': 0}
        names = {0: 'kernel', 1: 'memcpy', 2: 'memset'}          # hipGraphNodeTypeKernel / Memcpy / Memset
        for i in range(n.value):
            ty = ctypes.c_int(-1)
            if rt.hipGraphNodeGetType(ctypes.c_void_p(nodes[i]), ctypes.byref(ty)) != 0:
                return None
            out[names.get(ty.value, 'other')] += 1
        return out
    except Exception:           # noqa: BLE001  (statistics only)
        return None


class InferStep:
    """Eval-mode detector forward, eager or replayed from a hipGraph captured once per input shape.  At serving batch sizes
    the ~300 launches of the forward are issued slower than the GPU runs them (16 pairs: 2.9 ms eager, 2.5 ms replayed; at
    64 pairs the GPU is the limiter and both take 3.9 ms).  ``step(batch) -> the six detector outputs``; with ``use_graph`` they
    are static tensors, valid until the next call with the same shapes.  Decoding (``DynamicSpeaker._sample``) stays outside:
    its length is data dependent."""

    def __init__(self, cfg, cd, graph='all', setting='mode2', use_graph=True, warmup=2):
        self.cfg, self.cd, self.graph, self.setting = cfg, cd, graph, setting
        self.use_graph, self.warmup = use_graph, warmup
        self._graphs = {}
        self._pool = None

    def _run(self, b):
        with torch.no_grad():
            return detector_forward(self.cfg, self.cd, b, self.graph, self.setting)

    def __call__(self, b):
        if self.cd.training:
            raise hip.EkaidError('InferStep replays the eval-mode forward: call change_detector.eval() first')
        if not self.use_graph:
            return self._run(b)
        key = tuple((k, tuple(b[k].shape), b[k].dtype, b[k].device.type) for k in DETECTOR_KEYS)
        ent = self._graphs.get(key)
        if ent is None:
            dev = b['input_1'].device
            static = {k: b[k].clone() for k in DETECTOR_KEYS}
            s = torch.cuda.Stream(device=dev)
            s.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s):
                for _ in range(self.warmup):        # sizes the workspaces, raises kernel attributes, warms the allocator
                    self._run(static)
            torch.cuda.current_stream(dev).wait_stream(s)
            torch.cuda.synchronize(dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=self._pool, capture_error_mode='thread_local'):
                outs = self._run(static)
            if self._pool is None:
                self._pool = g.pool()
            ent = self._graphs[key] = {'graph': g, 'static': static, 'outs': outs}
        for k in DETECTOR_KEYS:
            ent['static'][k].copy_(b[k], non_blocking=True)
        ent['graph'].replay()
        return ent['outs']


class TrainStep:
    """``step(batch, T, denom=None) -> (total loss, speaker loss)`` (device scalars; no host synchronisation).
    With ``use_graph`` do not keep loss tensors of EARLIER eager steps of the same modules alive across the first call
    of a new shape (they pin autograd nodes to the eager stream; see ``_capture``)."""

    def __init__(self, cfg, cd, sp, opt, graph='all', setting='mode2', use_graph=False, comm='eager', warmup=2, t_round=1,
                 count_nodes=False):
        if comm not in ('eager', 'captured'):
            raise ValueError("comm must be 'eager' or 'captured'")
        self.cfg, self.cd, self.sp, self.opt = cfg, cd, sp, opt
        self.graph, self.setting = graph, setting
        self.use_graph, self.comm, self.warmup = use_graph, comm, warmup
        # Ragged answers give every batch its own number of decode steps and every (shape, steps) pair its own captured graph -- up to
        # seq_length - 1 of them on real data.  ``t_round`` > 1 runs the decoder for the step count rounded UP to a multiple of it
        # (capped at seq_length): the extra steps read padding tokens, their log-prob rows meet an all-zero loss mask (the criterion
        # multiplies by masks[:, 1:]), so loss and gradients are those of the exact step count up to the summation order of the
        # time-batched products (K = steps x batch rows grows by zero rows), and the number of graphs drops by that factor.
        self.t_round = max(int(t_round), 1)
        self.capture_s = 0.0
        self.graph_pool_bytes = 0
        self.count_nodes = count_nodes        # keep the raw hipGraph of a capture and count its nodes (``node_counts``: last capture)
        self.node_counts = None
        self._graphs = {}
        self._pool = None
        self._epoch = None
        self.marks = None
        self.calls_after_backward = None

    def close(self):
        """Drop the captured graphs and detach the dropout epoch word from the library.  Must run before the object goes
        away: the library would otherwise keep folding a freed device word into every dropout seed of later (eager) runs."""
        self._graphs = {}
        if self._epoch is not None:
            hip.check(hip.lib().ekaid_set_dropout_epoch_ptr(None), 'set_dropout_epoch_ptr')
            self._epoch = None

    def __del__(self):
        try:
            if getattr(self, '_epoch', None) is not None:
                hip.lib().ekaid_set_dropout_epoch_ptr(None)
        except Exception:       # noqa: BLE001  (interpreter shutdown)
            pass

    # ------------------------------------------------------------------ eager
    def _eager(self, b, T, denom):
        marks = self.marks
        if marks is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append(('start', e))
        self.opt.zero_grad()
        loss, sp_loss, _, _ = forward_loss(self.cfg, self.cd, self.sp, b, T, self.graph, self.setting, denom, marks)
        loss.backward()
        if marks is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append(('backward', e))
        self.opt.step()
        if marks is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append(('allreduce_wait_adam', e))
        return loss, sp_loss

    # ------------------------------------------------------------------ captured
    def _epoch_word(self, dev):
        if self._epoch is None:
            self._epoch = torch.zeros(1, dtype=torch.int64, device=dev)
            hip.check(hip.lib().ekaid_set_dropout_epoch_ptr(self._epoch.data_ptr()), 'set_dropout_epoch_ptr')
        return self._epoch

    def _body(self, static, T, denom):
        hip.check(hip.lib().ekaid_dropout_epoch_advance(self._epoch.data_ptr(), 1, hip.stream()), 'dropout_epoch_advance')
        will_step = self.comm == 'captured' or not self.opt.comm_active()
        self.opt.zero_grad()
        loss, sp_loss, _, _ = forward_loss(self.cfg, self.cd, self.sp, static, T, self.graph, self.setting, denom)
        loss.backward()
        self.calls_after_backward = hip.calls_issued()      # (tests: the first bucket's all-reduce was enqueued before this)
        if will_step:
            self.opt.step()
        return loss, sp_loss

    def _capture(self, b, T, denom):
        dev = b['input_1'].device
        self._epoch_word(dev)
        static = {k: b[k].clone() for k in INPUT_KEYS}
        sden = denom.clone() if denom is not None else None
        overlap = self.opt.overlap
        if self.comm == 'eager':
            self.opt.overlap = False            # no collectives inside the capture
        s = torch.cuda.Stream(device=dev)
        if not self.opt.built:
            # one eager forward/backward (no update) fixes the flat parameter/gradient layout and applies a pending
            # optimizer state (resume) BEFORE the snapshot below is taken.  On the side stream, and its loss dropped at
            # once: an autograd graph that outlives this block would pin AccumulateGrad nodes to another stream, and the
            # capture would then try to synchronise with that stream (crash in hipStreamEndCapture).
            seed0 = ops.seed_state()
            s.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s):
                self.opt.zero_grad()
                forward_loss(self.cfg, self.cd, self.sp, static, T, self.graph, self.setting, sden)[0].backward()
                self.opt.finish()
                if getattr(self.opt, '_pending_state', None) is not None:
                    self.opt._apply_pending_state()
            torch.cuda.current_stream(dev).wait_stream(s)
            ops.set_seed_state(seed0)
        # capture is transparent: the warm-up steps below really train, so weights, moments, step counter and the
        # dropout epoch are put back afterwards
        snap_p = self.opt.pflat.clone()
        snap_m = self.opt.m.clone() if self.opt.m is not None else None
        snap_v = self.opt.v.clone() if self.opt.v is not None else None
        snap_step, snap_epoch = self.opt.step_count, self._epoch.clone()
        seed_state = ops.seed_state()
        s.wait_stream(torch.cuda.current_stream(dev))
        # The warm-up iterations run WITHOUT collectives (local Adam on local gradients; everything is restored below): with
        # ragged answers the ranks meet a new (shape, decode steps) key at different iterations, so a rank that captures
        # must not issue all-reduces its peers (which already hold that key and just replay) would not match.
        was_suspended = self.opt.comm_suspended
        self.opt.comm_suspended = True
        try:
            try:
                with torch.cuda.stream(s):
                    for _ in range(self.warmup):    # sizes every workspace, raises kernel attributes, warms the allocator
                        self._body(static, T, sden)
                        if self.comm == 'eager' and self.opt.comm_active():
                            self.opt.step()
            finally:
                self.opt.comm_suspended = was_suspended
            torch.cuda.current_stream(dev).wait_stream(s)
            torch.cuda.synchronize(dev)
            ops.set_seed_state(seed_state)          # the capture draws the same site seeds as the first warm-up step
            g = torch.cuda.CUDAGraph(keep_graph=True) if self.count_nodes else torch.cuda.CUDAGraph()
            # thread_local: the RCCL watchdog thread of torch.distributed queries events while this thread captures
            with torch.cuda.graph(g, pool=self._pool, capture_error_mode='thread_local'):
                loss, sp_loss = self._body(static, T, sden)
            if self.count_nodes:
                self.node_counts = graph_node_counts(g)
                g.instantiate()
            if self._pool is None:
                self._pool = g.pool()
        finally:
            # also when the warm-up or the capture raised: the caller may fall back to another launch path and must continue
            # from the weights, moments, step count and dropout epoch it had (the warm-ups ran a LOCAL Adam)
            torch.cuda.synchronize(dev)
            self.opt.pflat.copy_(snap_p)
            if snap_m is not None:
                self.opt.m.copy_(snap_m)
                self.opt.v.copy_(snap_v)
            elif self.opt.m is not None:
                self.opt.m.zero_()
                self.opt.v.zero_()
            self.opt.step_count = snap_step
            self._epoch.copy_(snap_epoch)
            self.opt.overlap = overlap
        return {'graph': g, 'static': static, 'denom': sden, 'loss': loss, 'sp_loss': sp_loss}

    def _key(self, b, T, denom):
        """What a captured graph froze besides the tensor shapes: the decode-step count, whether the loss denominator is an
        input, the train/eval mode of both modules and the scheduled-sampling rate (a by-value kernel argument that also
        selects the decoder path, ``DynamicSpeaker._forward``)."""
        return (T, denom is not None, bool(self.cd.training), bool(self.sp.training), float(getattr(self.sp, 'ss_prob', 0.0))) + \
            tuple((k, tuple(b[k].shape), b[k].dtype) for k in INPUT_KEYS)

    def rounded_steps(self, T):
        if self.t_round <= 1:
            return T
        return min(int(self.cfg.model.speaker.seq_length), -(-int(T) // self.t_round) * self.t_round)

    def __call__(self, b, T, denom=None):
        T = self.rounded_steps(T)
        if not self.use_graph:
            return self._eager(b, T, denom)
        key = self._key(b, T, denom)
        ent = self._graphs.get(key)
        if ent is None:
            # graphs captured under another mode / scheduled-sampling rate are never replayed again (the rate only rises)
            for k in [k for k in self._graphs if k[2:5] != key[2:5]]:
                del self._graphs[k]
            import time
            # (statistics only.  Round 4 had seen the first replay differ from the eager step in its last bits when a device
            # synchronisation was added here, and suspected an uninitialised read.  Re-tested in round 5 on the final tree: with the extra
            # synchronize() the replay equals the eager steps bit for bit, also with every torch.empty output NaN-filled and the scratch
            # workspace 0xFF-filled before each use (EKAID_POISON=1, hip.enable_poison; profiles/r05_capture_sync_check.txt) -- no kernel
            # on the path reads memory it did not write, and no kernel choice depends on where a block lies)
            t0, m0 = time.perf_counter(), torch.cuda.memory_allocated()
            ent = self._capture(b, T, denom)
            self.capture_s += time.perf_counter() - t0
            self.graph_pool_bytes += max(torch.cuda.memory_allocated() - m0, 0)
            self._graphs[key] = ent
        for k in INPUT_KEYS:
            ent['static'][k].copy_(b[k], non_blocking=True)
        if denom is not None:
            ent['denom'].copy_(denom, non_blocking=True)
        ent['graph'].replay()
        if self.comm == 'eager' and self.opt.comm_active():
            self.opt.overlap = False
            self.opt._reset_pending()
            self.opt.step()                     # bucketed all-reduce (RCCL) + Adam, outside the graph
        return ent['loss'], ent['sp_loss']
