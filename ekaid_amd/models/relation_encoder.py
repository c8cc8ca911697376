"""Relation encoders -- mirror of reference ``models/relation_encoder.py:19-132``.

Quirk Q-A: with ``v_dim == out_dim`` the reference's ``v_transform`` is None and ``exp_v += exp_v_rel``
mutates the caller's tensor, so ChangeDetector's three "parallel" encoders are really a chain
``x1 = x0 + sem(x0); x2 = x1 + spa(x1); x3 = x2 + imp(x2)``.  Here the encoders are functional
(they return the updated residual stream) and ``ChangeDetector`` chains them explicitly.
"""
import torch
import torch.nn as nn

from .. import ops
from .fc import FCNet, PreDropped, RowBroadcast
from .graph_att import GAttNet as GAT


def q_expand_v_cat(q, v, mask=True, p=0.0, training=False, lazy=False):
    """Returns the PAIR (v [R*N, A], q_expanded [R*N, Q]) that stands for ``cat((v, q_expand), -1)``
    (relation_encoder.py:19-29): the question vector of each sample broadcast to its nodes and zeroed
    on nodes whose feature row sums to exactly 0.  ``q`` may hold fewer rows than R (R = 2B image
    slices share the B question vectors)."""
    R, N, A = v.shape
    v2 = ops.pass_slot(v, v.reshape(R * N, A))          # (v may be a fan-out alias: its reader is the FCNet that takes v2)
    if mask and training and p > 0 and v.is_contiguous():
        # one pass over v: the zero-row mask and the dropped copy the consumer FCNet multiplies (its input dropout)
        keep, seed, vd = ops.rowmask_dropout(v, p)
        ops.predropped(v2, (float(p), seed, vd))
    elif mask:
        keep = ops.rowsum(v, as_mask=True).view(-1)
    else:
        keep = torch.ones(R * N, device=v.device, dtype=torch.float32)
    if training and p > 0:      # the consumer FCNet's input dropout, fused into the expansion kernel
        return v2, PreDropped(ops.expand_rows(q, keep, N, p, True))
    if lazy and not torch.is_grad_enabled():
        # inference: nothing is expanded; the consumer FCNet takes q . Wq^T once per sample (fc.RowBroadcast)
        return v2, RowBroadcast(q, keep, N)
    return v2, ops.expand_rows(q, keep, N)


class ImplicitRelationEncoder(nn.Module):
    def __init__(self, v_dim, q_dim, out_dim, dir_num, pos_emb_dim, nongt_dim, num_heads=16, num_steps=1,
                 residual_connection=True, label_bias=True):
        super(ImplicitRelationEncoder, self).__init__()
        self.v_dim = v_dim
        self.q_dim = q_dim
        self.out_dim = out_dim
        self.residual_connection = residual_connection
        self.num_steps = num_steps
        self.v_transform = FCNet([v_dim, out_dim]) if self.v_dim != self.out_dim else None
        in_dim = out_dim + q_dim
        self.implicit_relation = GAT(dir_num, 1, in_dim, out_dim, nongt_dim=nongt_dim, label_bias=label_bias,
                                     num_heads=num_heads, pos_emb_dim=pos_emb_dim)

    def forward(self, v, position_embedding, q):
        """v: [R, N, v_dim]; position_embedding: [R, N, M, emb]; q: [B, q_dim] (R multiple of B)."""
        R, N = v.shape[0], v.shape[1]
        imp_adj_mat = torch.ones(R, N, N, 1, device=v.device, dtype=torch.float32)
        imp_v = self.v_transform(v) if self.v_transform else v
        affs = []
        for _ in range(self.num_steps):
            fc = self.implicit_relation.self_weights
            v_in, v_res = ops.fanout(imp_v, 2) if self.residual_connection else (imp_v, None)
            v_cat_q = q_expand_v_cat(q, v_in, True, fc.input_dropout(), fc.training, lazy=True)
            imp_v, aff = self.implicit_relation.forward(v_cat_q, imp_adj_mat, position_embedding, residual=v_res)
            affs.append(aff)
        return imp_v, affs[0]


class ExplicitRelationEncoder(nn.Module):
    def __init__(self, v_dim, q_dim, out_dim, dir_num, label_num, nongt_dim=20, num_heads=16, num_steps=1,
                 residual_connection=True, label_bias=True):
        super(ExplicitRelationEncoder, self).__init__()
        self.v_dim = v_dim
        self.q_dim = q_dim
        self.out_dim = out_dim
        self.num_steps = num_steps
        self.residual_connection = residual_connection
        self.v_transform = FCNet([v_dim, out_dim]) if self.v_dim != self.out_dim else None
        in_dim = out_dim + q_dim
        self.explicit_relation = GAT(dir_num, label_num, in_dim, out_dim, nongt_dim=nongt_dim, num_heads=num_heads,
                                     label_bias=label_bias, pos_emb_dim=-1)

    def forward(self, v, exp_adj_matrix, q):
        """v: [R, N, v_dim]; exp_adj_matrix: [R, N, N, L]; q: [B, q_dim]."""
        exp_v = self.v_transform(v) if self.v_transform else v
        aff = None
        for _ in range(self.num_steps):
            fc = self.explicit_relation.self_weights
            v_in, v_res = ops.fanout(exp_v, 2) if self.residual_connection else (exp_v, None)
            v_cat_q = q_expand_v_cat(q, v_in, True, fc.input_dropout(), fc.training, lazy=True)
            exp_v, aff = self.explicit_relation.forward(v_cat_q, exp_adj_matrix, residual=v_res)
        return exp_v, aff
