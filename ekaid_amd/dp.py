"""Data-parallel training support: flat parameter/gradient storage, bucketed gradient all-reduce
over RCCL (xGMI) overlapped with backward, and the fused Adam step.

The reference is single-process (train_mimic.py:2 pins one GPU); this is new work.
One process per GPU, weights replicated, the (image-pair, question) batch sharded by contiguous sample
slices.  Only parameters that actually receive a gradient are communicated and updated: the reference
model carries 31.5 M dead parameters (``SSRE.*``, ``fc1.*``, ``*.neighbor_net.0.*``, ``*.linear_out_.*``) and a
frozen table; they are discovered on the first backward (their ``.grad`` stays ``None``), exactly the
set ``torch.optim.Adam`` would skip.

Layout: live gradients sit in ONE flat fp32 buffer ordered by the order in which autograd finished
them during the first backward (speaker -> gating tail -> implicit -> spatial -> semantic encoder ->
question encoder -> ROI projection).  The buffer is cut into buckets of ~``bucket_mb``; a
post-accumulate hook counts finished parameters per bucket and launches an asynchronous
``all_reduce(SUM)`` on the bucket's slice as soon as it is complete, so communication of early
buckets overlaps the rest of backward.  xGMI is point-to-point (7 links x ~153 GB/s per GPU):
223 MB of gradients need ~0.4-2.6 ms per step, far below the backward time, so a few large buckets
(default 32 MB) are preferred over many small ones.  The 1/world_size averaging is folded into the
Adam kernel (``grad_scale``).

Difference from ``torch.optim.Adam`` (documented, not emulated): the live set is fixed at the first backward and updated as
ONE flat range under one global step counter.  A live parameter that receives no gradient in some later step keeps the
zeros of its slot: its moments decay and it still moves by ``lr * m_hat / (sqrt(v_hat) + eps)``, where torch (gradients set
to None) would skip it and keep a per-parameter step count.  On this path every live parameter fires in every step (fixed
graph / setting per run; ragged answers only change the number of decode steps, not which weights are used), which
``FlatAdam.step`` can assert with ``EKAID_DP_ASSERT_ALL_FIRED=1`` (tests run with it).
"""
import hashlib
import os

import torch
import torch.distributed as dist

from . import hip


_comm_suspended = False      # process-wide: no collective of this module runs while set (capture warm-ups, engine.py)


class GradBuckets:
    """Device-agnostic flat-gradient bookkeeping + bucketed all-reduce (tested on CPU with gloo)."""

    @property
    def comm_suspended(self):
        return _comm_suspended

    @comm_suspended.setter
    def comm_suspended(self, value):
        global _comm_suspended
        _comm_suspended = bool(value)

    def __init__(self, params, process_group=None, bucket_mb=32.0, overlap=True):
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.bucket_bytes = int(bucket_mb * (1 << 20))
        self.overlap = overlap
        self.force_comm = False       # issue the collectives even at world size 1 (exercises the RCCL path on one GPU)
        self.launch_log = None        # list -> (bucket, C-ABI calls issued so far) per all-reduce launch (tests, bench)
        self.built = False
        self.order = []             # parameters in first-backward completion order
        self._seen = set()
        self.handles = []
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    # ------------------------------------------------------------------ discovery / build
    def _on_grad(self, p):
        if not self.built:
            if id(p) not in self._seen:
                self._seen.add(id(p))
                self.order.append(p)
            return
        b = self._bucket_of.get(id(p))
        if b is None:
            raise RuntimeError('a parameter outside the live set discovered on the first backward received a '
                               'gradient; rebuild GradBuckets (call reset())')
        o = self._offset_of[id(p)]
        if p.grad.data_ptr() != self._gbase + 4 * o:
            # the gradient was produced outside its flat slot (a node that does not know about slots, a gradient shared
            # by two parameters, a cloned gradient): move it in before the bucket can be reduced
            view = self.gflat[o:o + p.numel()].view(p.shape)
            view.copy_(p.grad)
            p.grad = view
        self._pending[b] -= 1
        if self.overlap and self.comm_active():
            # strictly in bucket order (bucket b only once 0..b-1 are out): every rank issues the same sequence of
            # collectives even if autograd finishes two buckets in a different order on different ranks
            while self._next < len(self.buckets) and self._pending[self._next] == 0:
                self._launch(self._next)

    def build(self):
        """Freeze the live set (parameters whose .grad is set now) and move them into flat storage."""
        live = [p for p in self.order if p.grad is not None]
        if not live:
            raise RuntimeError('GradBuckets.build(): no parameter has a gradient yet (run one backward first)')
        dev, dt = live[0].device, live[0].dtype
        sizes = [p.numel() for p in live]
        offs, total = [], 0
        for n in sizes:
            offs.append(total)
            total += (n + 7) // 8 * 8                    # every tensor starts on an 8-element group (32-byte aligned)
        self.total = total
        self.pflat = torch.zeros(total, device=dev, dtype=dt)
        # one more 8-element group behind the gradients: element `total` is the FAULT FLAG of the step -- each rank writes 1.0 there if
        # one of its persistent sequence kernels abandoned its launch (ekaid_seq_fault_flag, just before the last bucket goes out), the
        # flag travels inside the last bucket's all-reduce, and the Adam kernel of EVERY rank skips the update when the sum is non-zero
        self._gstore = torch.zeros(total + 8, device=dev, dtype=dt)
        self.gflat = self._gstore[:total]
        self.live, self.offsets = live, offs
        self._offset_of = {id(p): o for p, o in zip(live, offs)}
        self._gbase = self.gflat.data_ptr()
        with torch.no_grad():
            for p, o in zip(live, offs):
                n = p.numel()
                self.pflat[o:o + n].copy_(p.detach().reshape(-1))
                self.gflat[o:o + n].copy_(p.grad.reshape(-1))
                p.data = self.pflat[o:o + n].view(p.shape)
                p.grad = self.gflat[o:o + n].view(p.shape)
        # buckets: contiguous ranges of the flat buffer in completion order
        self.buckets, self._bucket_of = [], {}
        start, cur_bytes, members = 0, 0, []
        for i, (p, o) in enumerate(zip(live, offs)):
            members.append(p)
            cur_bytes += p.numel() * 4
            last = i == len(live) - 1
            if cur_bytes >= self.bucket_bytes or last:
                end = total + 8 if last else offs[i + 1]         # (the last bucket carries the fault flag group)
                self.buckets.append((start, end, len(members)))
                for m in members:
                    self._bucket_of[id(m)] = len(self.buckets) - 1
                start, cur_bytes, members = end, 0, []
        self.built = True
        self._reset_pending()
        self._check_layout_agreement()
        if self.gflat.is_cuda:
            from . import ops
            for p, o in zip(self.live, self.offsets):         # backward kernels write straight into the flat buffer (ops.grad_out)
                ops.register_grad_slot(p, self.gflat, o)

    def comm_active(self):
        return (self.world > 1 or self.force_comm) and not self.comm_suspended

    def _check_layout_agreement(self):
        """Buckets are reduced by raw offset: all ranks must have derived the SAME flat layout (same live set, same
        completion order, same cuts).  Compared through a digest of (shape, offset) per tensor + the bucket bounds."""
        if self.world <= 1:
            return
        h = hashlib.sha256(repr(([tuple(p.shape) for p in self.live], self.offsets, self.buckets)).encode()).hexdigest()
        all_h = [None] * self.world
        dist.all_gather_object(all_h, h, group=self.group)
        if any(x != h for x in all_h):
            raise RuntimeError('GradBuckets: ranks disagree on the flat gradient layout (a parameter received a gradient '
                               'on some ranks only, or backward finished gradients in a different order): %r' % (all_h,))

    def _reset_pending(self):
        self._pending = [n for (_, _, n) in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._next = 0
        self.handles = []

    def reset(self):
        self.built = False
        self.order, self._seen = [], set()

    # ------------------------------------------------------------------ per-step API
    def zero_grad(self):
        if self.built:
            self._gstore.zero_()
            if self.gflat.is_cuda:
                # gradients are WRITTEN into their flat slots by the backward kernels and adopted by autograd as p.grad
                # (ops.grad_out); a parameter that receives none this step keeps the zeros of its slot
                from . import ops
                ops.clear_grad_claims()
                for p in self.live:
                    p.grad = None
            else:
                for p, o in zip(self.live, self.offsets):     # autograd accumulates in place into the views
                    p.grad = self.gflat[o:o + p.numel()].view(p.shape)
            self._reset_pending()
        else:
            for p in self.params:
                p.grad = None

    def _launch(self, b):
        if self._launched[b]:
            return
        s, e, _ = self.buckets[b]
        if self._gstore.is_cuda:
            # the raw weight gradients of the weight-normed layers registered so far (every layer of this bucket is) become dv / dg
            # BEFORE the bucket is reduced: ops.wn_grad_register
            from . import ops
            ops.wn_grad_flush()
        self._launched[b] = True
        self._next = max(self._next, b + 1)
        if self.launch_log is not None:
            self.launch_log.append((b, hip.calls_issued()))
        if b == len(self.buckets) - 1 and self._gstore.is_cuda:
            # every persistent launch of this step lies behind us (forward; the decoder's BPTT kernel opens backward, the GRU's closes it
            # before its weights -- members of this last bucket -- have their gradients): this rank's verdict joins the reduction
            hip.check(hip.lib().ekaid_seq_fault_flag(self._gstore.data_ptr() + 4 * self.total, hip.stream()), 'seq_fault_flag')
        self.handles.append(dist.all_reduce(self._gstore[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Make every bucket's reduced SUM visible to the current stream (call after backward)."""
        if not self.built:
            self.build()
        if self.comm_active():
            for b in range(len(self.buckets)):
                self._launch(b)
            for h in self.handles:
                h.wait()
        self.handles = []

    def live_names(self, named_modules):
        ids = {id(p): n for n, p in named_modules}
        return [ids.get(id(p), '?') for p in self.live]


class FlatAdam(GradBuckets):
    """torch.optim.Adam semantics (utils/utils.py:96-99: lr, betas, eps, no weight decay / amsgrad) as one
    fused HIP kernel over the flat live-parameter buffer, with the gradient all-reduce of GradBuckets."""

    def __init__(self, modules, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, process_group=None, bucket_mb=32.0,
                 overlap=True):
        params = []
        for m in modules:
            params += list(m.parameters())
        super().__init__(params, process_group, bucket_mb, overlap)
        self._lr, self.betas, self.eps = lr, betas, eps
        self._step_host = 0
        self.m = self.v = None
        self._hyper = self._step_dev = self._derived = None

    # The step counter and the hyper-parameters live in device memory (ekaid_adam_step_dev): a step captured into a
    # hipGraph keeps counting and follows learning-rate changes made between replays.
    def _device_state(self):
        if self._hyper is None:
            dev = self.pflat.device
            self._hyper = torch.tensor([self._lr, self.betas[0], self.betas[1], self.eps, 1.0 / self.world],
                                       dtype=torch.float32, device=dev)
            self._step_dev = torch.tensor([self._step_host], dtype=torch.int32, device=dev)
            self._derived = torch.zeros(4, dtype=torch.float32, device=dev)
        return self._hyper, self._step_dev, self._derived

    @property
    def lr(self):
        return self._lr

    @lr.setter
    def lr(self, value):
        self._lr = float(value)
        if self._hyper is not None:
            self._hyper[0:1].copy_(torch.tensor([self._lr], dtype=torch.float32), non_blocking=False)

    @property
    def step_count(self):
        """Number of updates applied so far (one device->host read when the counter lives on the device)."""
        return int(self._step_dev.item()) if self._step_dev is not None else self._step_host

    @step_count.setter
    def step_count(self, value):
        self._step_host = int(value)
        if self._step_dev is not None:
            self._step_dev.fill_(int(value))

    def step(self):
        if self.built and os.environ.get('EKAID_DP_ASSERT_ALL_FIRED') == '1':
            fired = [n - q for (_, _, n), q in zip(self.buckets, self._pending)]
            # (after a hipGraph replay no hook has run: nothing to check; after an eager backward every live parameter must
            # have delivered its gradient -- the module docstring explains why that matters for Adam)
            if any(fired) and any(self._pending):
                raise RuntimeError('FlatAdam.step: %d live parameters received no gradient in this step'
                                   % sum(self._pending))
        self.finish()
        self.apply_update()

    def apply_update(self):
        """The Adam kernel alone (gradients already reduced): what a captured step replays."""
        hip.require_cuda(self.pflat)
        if getattr(self, '_pending_state', None) is not None:
            self._apply_pending_state()
        if self.m is None:
            self.m = torch.zeros_like(self.pflat)
            self.v = torch.zeros_like(self.pflat)
        hyper, step, derived = self._device_state()
        hip.check(hip.lib().ekaid_adam_step_dev(hip.ptr(self.pflat), hip.ptr(self.gflat), hip.ptr(self.m), hip.ptr(self.v),
                                                self.total, hip.ptr(hyper), hip.ptr(step), hip.ptr(derived),
                                                self._gstore.data_ptr() + 4 * self.total, hip.stream()),
                  'adam_step_dev')

    # ------------------------------------------------------------------ checkpointing (SURVEY f-4)
    def state_dict(self, named_parameters=None):
        """Optimizer state keyed by parameter NAME (robust against the flat layout, which depends on the order in
        which the first backward finishes gradients).  ``named_parameters``: iterable of (name, parameter) covering the
        optimised modules.  The reference saves no optimizer state at all (train_mimic.py:283-287), so a resumed
        reference run restarts Adam's moments; this one continues bit-exactly."""
        out = {'step': self.step_count, 'lr': self.lr, 'betas': self.betas, 'eps': self.eps,
               'exp_avg': {}, 'exp_avg_sq': {}}
        if self.m is None or named_parameters is None:
            return out
        names = {id(p): n for n, p in named_parameters}
        for p, o in zip(self.live, self.offsets):
            n = names.get(id(p))
            if n is None:
                raise KeyError('FlatAdam.state_dict: a live parameter is missing from named_parameters')
            k = p.numel()
            out['exp_avg'][n] = self.m[o:o + k].view(p.shape).detach().cpu().clone()
            out['exp_avg_sq'][n] = self.v[o:o + k].view(p.shape).detach().cpu().clone()
        return out

    def load_state_dict(self, state, named_parameters):
        """Restore ``state_dict`` output.  The moments are applied when the flat storage exists (after the first
        backward of the resumed run), i.e. before the first ``step()``."""
        self.step_count = int(state['step'])
        self.lr = float(state['lr'])
        self._pending_state = (state, {n: p for n, p in named_parameters})
        if self.built:
            self._apply_pending_state()

    def _apply_pending_state(self):
        state, by_name = self._pending_state
        self._pending_state = None
        if not state['exp_avg']:
            return
        self.m = torch.zeros_like(self.pflat)
        self.v = torch.zeros_like(self.pflat)
        where = {id(p): o for p, o in zip(self.live, self.offsets)}
        for n, t in state['exp_avg'].items():
            p = by_name[n]
            if id(p) not in where:
                raise KeyError('FlatAdam.load_state_dict: %s has optimizer state but received no gradient' % n)
            o, k = where[id(p)], p.numel()
            self.m[o:o + k].copy_(t.reshape(-1).to(self.m.device))
            self.v[o:o + k].copy_(state['exp_avg_sq'][n].reshape(-1).to(self.v.device))


# --------------------------------------------------------------------------- loss weighting for ragged answers
def loss_denominator(local_mask_sum, process_group=None):
    """Denominator that makes data-parallel training reproduce the reference's loss over the GLOBAL batch.

    The reference loss is ``-sum(logp * mask) / sum(mask)`` over the whole batch (utils/utils.py:213-214).  Each rank
    computes its numerator locally and the optimiser averages the ranks' gradients (1/W); dividing every rank's
    numerator by ``sum_over_ranks(mask sum) / W`` therefore yields exactly the gradient of the global-batch loss, also
    when the ranks hold answers of different lengths.  ``local_mask_sum``: float or 1-element
    tensor (host or device); returns a 1-element float32 tensor on the same device (one tiny all-reduce)."""
    t = local_mask_sum if isinstance(local_mask_sum, torch.Tensor) else torch.tensor([float(local_mask_sum)])
    t = t.detach().reshape(1).to(torch.float32).clone()
    world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=process_group)
        t /= world
    return t


# --------------------------------------------------------------------------- Q-C under data parallelism
class _AllGatherRows(torch.autograd.Function):
    """(B_local, L) -> (B_global, L) with the matching backward (sum over ranks of the local slice)."""

    @staticmethod
    def forward(ctx, x, group):
        world = dist.get_world_size(group)
        ctx.group, ctx.rows, ctx.live = group, x.shape[0], not _comm_suspended
        if not ctx.live:
            # warm-up iteration of a hipGraph capture (its results are discarded): same shapes, no collective -- the peers
            # may be replaying a graph they already hold and would not answer
            return torch.cat([x] * world, 0)
        parts = [torch.empty_like(x) for _ in range(world)]
        dist.all_gather(parts, x.contiguous(), group=group)
        return torch.cat(parts, 0)

    @staticmethod
    def backward(ctx, d):
        d = d.contiguous()
        if ctx.live:
            dist.all_reduce(d, op=dist.ReduceOp.SUM, group=ctx.group)
        r = dist.get_rank(ctx.group)
        return d[r * ctx.rows:(r + 1) * ctx.rows].clone(), None


def enable_global_question_attention(change_detector, process_group=None):
    """Opt-in "global-batch semantics" for quirk Q-C (contract ii): the batch-softmax of
    ``QuestionSelfAttention`` runs over the GLOBAL batch, reproducing the single-GPU result at any world
    size.  Default (contract i) is the reference run at the local batch."""
    def gather(atten):
        r = dist.get_rank(process_group)
        return _AllGatherRows.apply(atten, process_group), r * atten.shape[0]
    change_detector.q_att.gather_logits = gather
