"""``FCNet`` and the parameter-holding leaf layers of the path.

Mirror of reference ``models/fc.py:15-49``: a stack of ``[Dropout] -> weight_norm(Linear, dim=None)
-> [activation]``; same constructor, same ``state_dict`` keys (``main.<i>.bias``, ``main.<i>.weight_g``
(scalar), ``main.<i>.weight_v``), same default initialisation (the parameters are drawn by the very
``nn.Linear`` constructor the reference uses, so an identical torch seed gives identical weights).
The arithmetic is ``ekaid_amd.ops`` (HIP); the reference's two NaN-assert host synchronisations per
call (fc.py:42,45) are not reproduced.
"""
import torch
import torch.nn as nn

from .. import ops
from ..hip import ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH

import os
_UNFUSED_INPUT_DROPOUT = bool(os.environ.get('EKAID_UNFUSED_INPUT_DROPOUT'))
_ACT_CODE = {None: ACT_NONE, '': ACT_NONE, 'ReLU': ACT_RELU, 'Tanh': ACT_TANH, 'Sigmoid': ACT_SIGMOID}


class PreDropped:
    """Marks a tensor whose dropout (that of the FCNet it is about to enter) was already applied by its producer
    kernel (``ops.expand_rows`` fuses it), so ``FCNet.forward`` must not drop it again."""

    def __init__(self, tensor):
        self.tensor = tensor


class RowBroadcast:
    """Stands for the rows ``mask[r] * q[(r // group) % len(q)]`` without materialising them (the question half of
    ``q_expand_v_cat`` under ``torch.no_grad()`` with dropout off): the FCNet that consumes it computes ``q . Wq^T`` once per
    sample and adds it as a row-broadcast residual in the epilogue of ``v . Wv^T`` -- ``[v | q] W^T = v Wv^T + q Wq^T``."""

    def __init__(self, q, mask, group):
        self.q, self.mask, self.group = q, mask, group

    def materialise(self):
        return ops.expand_rows(self.q, self.mask, self.group)


class Linear(nn.Linear):
    """``nn.Linear`` parameters, HIP arithmetic. ``xs`` may be a tuple of tensors whose feature axes are
    concatenated (replaces ``torch.cat`` + Linear of the reference)."""

    def forward(self, xs, act=ACT_NONE):
        return ops.linear(xs, self.weight, self.bias, None, act)


class WNLinear(nn.Module):
    """weight_norm(nn.Linear(in, out, bias), dim=None) with the old-style parameter names."""

    def __init__(self, in_dim, out_dim, bias=True):
        super().__init__()
        lin = nn.Linear(in_dim, out_dim, bias=bias)          # same init / RNG draw as the reference
        if bias:
            self.bias = nn.Parameter(lin.bias.detach().clone())
        else:
            self.register_parameter('bias', None)
        self.weight_g = nn.Parameter(torch.linalg.vector_norm(lin.weight.detach()).clone())
        self.weight_v = nn.Parameter(lin.weight.detach().clone())
        self.in_features, self.out_features = in_dim, out_dim

    def forward(self, xs, act=ACT_NONE, drops=None):
        """``drops``: per-input dropout probabilities applied to the inputs inside the linear node (the FCNet's
        Dropout layer; its backward then rides in the dgrad GEMM epilogue)."""
        return ops.wn_linear(xs, self.weight_v, self.weight_g, self.bias, act, drops)

    def extra_repr(self):
        return 'in_features=%d, out_features=%d, bias=%s' % (self.in_features, self.out_features,
                                                             self.bias is not None)


class FCNet(nn.Module):
    """Simple class for non-linear fully connect network (reference docstring)."""

    def __init__(self, dims, act='ReLU', dropout=0, bias=True):
        super(FCNet, self).__init__()
        if act not in _ACT_CODE:
            raise ValueError('FCNet: unsupported activation %r' % (act,))
        layers = []
        for i in range(len(dims) - 1):
            if 0 < dropout:
                layers.append(nn.Dropout(dropout))           # marker: applied inside the following linear node
            layers.append(WNLinear(dims[i], dims[i + 1], bias=bias))
            if '' != act and act is not None:
                layers.append(getattr(nn, act)())            # marker: fused into the GEMM epilogue
        self.main = nn.Sequential(*layers)
        self.act = _ACT_CODE[act]

    def forward(self, x):
        """``x``: tensor or tuple of tensors (feature-axis concatenation)."""
        pending = 0.0                                        # dropout of the layer input, applied inside the linear node
        if isinstance(x, (tuple, list)) and any(isinstance(t, RowBroadcast) for t in x):
            lin = [m for m in self.main if isinstance(m, WNLinear)]
            if (len(x) == 2 and isinstance(x[1], RowBroadcast) and len(lin) == 1 and not torch.is_grad_enabled()
                    and not (self.training and self.input_dropout() > 0)):
                rb = x[1]
                return ops.wn_linear_rowbcast(x[0], rb.q, rb.mask, rb.group, lin[0].weight_v, lin[0].weight_g, lin[0].bias, self.act)
            x = tuple(t.materialise() if isinstance(t, RowBroadcast) else t for t in x)
        for layer in self.main:
            if isinstance(layer, nn.Dropout):
                pending = float(layer.p) if self.training else 0.0
            elif isinstance(layer, WNLinear):
                xs = x if isinstance(x, (tuple, list)) else (x,)
                if _UNFUSED_INPUT_DROPOUT and pending > 0.0:     # A/B aid: dropout as a node of its own
                    xs = tuple(t if isinstance(t, PreDropped) else ops.dropout(t, pending, True) for t in xs)
                    pending = 0.0
                drops = tuple(0.0 if isinstance(t, PreDropped) else pending for t in xs)
                xs = tuple(t.tensor if isinstance(t, PreDropped) else t for t in xs)
                x = layer(xs if len(xs) > 1 else xs[0], self.act, drops)
                pending = 0.0
        return x

    def input_dropout(self):
        """Dropout probability applied to this FCNet's input (0 if none)."""
        first = self.main[0]
        return first.p if isinstance(first, nn.Dropout) else 0.0


def fcnet_pair(fa, xa, fb, xb):
    """``(fa(xa), fb(xb))`` for two single-layer FCNets whose products can share launches (``ops.linear_pair``: the query / key
    projections of a graph-attention layer, graph_att_layer.py:79,90); any other combination runs the two nets one after the other."""
    la = [m for m in fa.main if isinstance(m, WNLinear)]
    lb = [m for m in fb.main if isinstance(m, WNLinear)]
    plain = all(isinstance(t, torch.Tensor) and t.is_cuda for t in (xa, xb))
    if len(la) != 1 or len(lb) != 1 or not plain or _UNFUSED_INPUT_DROPOUT:
        return fa(xa), fb(xb)
    pa = float(fa.input_dropout()) if fa.training else 0.0
    pb = float(fb.input_dropout()) if fb.training else 0.0
    return ops.linear_pair((xa, la[0].weight_v, la[0].bias, la[0].weight_g, fa.act, pa),
                           (xb, lb[0].weight_v, lb[0].bias, lb[0].weight_g, fb.act, pb))
