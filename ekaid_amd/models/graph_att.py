"""``GAttNet`` -- mirror of reference ``models/graph_att.py:17-106``.

Reproduces quirk Q-B by construction: the reference's direction loop overwrites ``output``, so the
result is ``relu(dropout(2 * neighbor_net[dir_num-1](...)))`` -- ``self_feat`` is never added and the
other direction never reaches the output (its parameters never get a gradient).  Only the surviving
direction is computed unless ``compute_dead_directions`` is set (debug / attention-map parity).
The trailing ``dropout -> relu`` (and the in-place residual of the caller) is one fused kernel, so
``forward`` takes the residual stream ``v`` and returns the updated one.
"""
import torch
import torch.nn as nn

from .. import ops
from .fc import FCNet
from .graph_att_layer import GraphSelfAttentionLayer


class GAttNet(nn.Module):
    def __init__(self, dir_num, label_num, in_feat_dim, out_feat_dim, nongt_dim=20, dropout=0.2, label_bias=True,
                 num_heads=16, pos_emb_dim=-1):
        super(GAttNet, self).__init__()
        assert dir_num <= 2, "Got more than two directions in a graph."
        self.dir_num = dir_num
        self.label_num = label_num
        self.in_feat_dim = in_feat_dim
        self.out_feat_dim = out_feat_dim
        self.dropout = nn.Dropout(dropout)
        self.self_weights = FCNet([in_feat_dim, out_feat_dim], '', dropout)
        self.bias = FCNet([label_num, 1], '', 0, label_bias)
        self.nongt_dim = nongt_dim
        self.pos_emb_dim = pos_emb_dim
        neighbor_net = []
        for i in range(dir_num):
            neighbor_net.append(GraphSelfAttentionLayer(pos_emb_dim=pos_emb_dim, num_heads=num_heads,
                                                        feat_dim=out_feat_dim, nongt_dim=nongt_dim))
        self.neighbor_net = nn.ModuleList(neighbor_net)
        self.compute_dead_directions = False

    def edge_terms(self, adj_matrix):
        """condensed adjacency (sum over labels, graph_att.py:89) and label bias (:92), both (R, N, N) in
        the original orientation.  ``adj_matrix``: (R, N, N, L) float one-hot."""
        adj_matrix = adj_matrix.float() if adj_matrix.dtype != torch.float32 else adj_matrix
        R, N, _, L = adj_matrix.shape
        cond = ops.rowsum(adj_matrix)
        lbias = self.bias(adj_matrix.reshape(-1, L)).view(R, N, N)
        return cond, lbias

    def forward(self, v_feat, adj_matrix, pos_emb=None, residual=None):
        """
        v_feat: tensor [R, N, in_feat] or tuple (v [R*N, A], q_expanded [R*N, Q]) standing for their
        concatenation; adj_matrix: [R, N, N, L].  Returns (output, [softmax per direction]); with
        ``residual`` given, output = residual + relu(dropout(2*o)) (the caller's in-place ``+=``).
        """
        if self.pos_emb_dim > 0 and pos_emb is None:
            raise ValueError(f"position embedding is set to None with pos_emb_dim {self.pos_emb_dim}")
        elif self.pos_emb_dim < 0 and pos_emb is not None:
            raise ValueError("position embedding is NOT None with pos_emb_dim < 0")
        R, N = adj_matrix.shape[0], adj_matrix.shape[1]
        self_feat = self.self_weights(v_feat).view(R, N, self.out_feat_dim)
        cond, lbias = self.edge_terms(adj_matrix)
        aff = [None] * self.dir_num
        output = None
        dirs = range(self.dir_num) if self.compute_dead_directions else [self.dir_num - 1]
        for d in dirs:
            output, aff[d] = self.neighbor_net[d].forward(self_feat, cond, pos_emb, lbias, direction=d)
        if residual is None:
            residual = torch.zeros_like(output)
        out = ops.residual_relu2(residual, output, self.dropout.p, self.training)
        return out, aff
