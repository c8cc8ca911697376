"""``GraphSelfAttentionLayer`` -- mirror of reference ``models/graph_att_layer.py:19-178``.

Same constructor / forward signature / ``state_dict`` keys (including the dead, never-called
weight-normed grouped conv ``linear_out_`` of :52-56).  The forward runs on HIP only:

  q = query(drop(x)), k = key(drop(x[:, :M]))                     -> ekaid_gemm_f32 (weight-norm epilogue)
  QK^T/sqrt(dh) (+log geometry) -> mask(-9e15) + label bias -> softmax -> P.x per head
                                                                  -> ekaid_gat_fused_fwd (one kernel)
  linear_out_2                                                    -> ekaid_gemm_f32 (K = heads*feat)
"""
import torch
import torch.nn as nn
from torch.nn.utils import weight_norm

from .. import ops
from ..hip import ACT_RELU
from .fc import FCNet, Linear, fcnet_pair


class GraphSelfAttentionLayer(nn.Module):
    def __init__(self, feat_dim, nongt_dim=20, pos_emb_dim=-1, num_heads=16, dropout=[0.2, 0.5]):
        super(GraphSelfAttentionLayer, self).__init__()
        self.fc_dim = num_heads
        self.feat_dim = feat_dim
        self.dim = (feat_dim, feat_dim, feat_dim)
        self.dim_group = (int(self.dim[0] / num_heads), int(self.dim[1] / num_heads), int(self.dim[2] / num_heads))
        self.num_heads = num_heads
        self.pos_emb_dim = pos_emb_dim
        if self.pos_emb_dim > 0:
            self.pair_pos_fc1 = FCNet([pos_emb_dim, self.fc_dim], None, dropout[0])
        self.query = FCNet([feat_dim, self.dim[0]], None, dropout[0])
        self.nongt_dim = nongt_dim
        self.key = FCNet([feat_dim, self.dim[1]], None, dropout[0])
        # constructed for state_dict compatibility only; the reference never calls it either
        self.linear_out_ = weight_norm(nn.Conv2d(in_channels=self.fc_dim * feat_dim, out_channels=self.dim[2],
                                                 kernel_size=(1, 1), groups=self.fc_dim), dim=None)
        self.linear_out_2 = Linear(self.fc_dim * feat_dim, self.dim[2])
        self.attention_impl = 'fused'        # 'fused' | 'gemm' (same math composed from batched GEMMs)

    def forward(self, roi_feat, adj_matrix, position_embedding, label_biases_att, direction=0):
        """
        roi_feat: [R, N, feat_dim]; adj_matrix / label_biases_att: [R, N, N] in the ORIGINAL orientation
        (``direction=1`` makes the kernel read them transposed, which is what graph_att.py:76,88 feeds
        to neighbor_net[1]); position_embedding: [R, N, M, pos_emb_dim] or None.
        Returns (output [R, N, feat_dim], softmax [R, N, heads, M]).
        """
        R, N, A = roi_feat.shape
        M = self.nongt_dim if self.nongt_dim < N else N
        feat_q, feat_k, feat_s = ops.fanout(roi_feat, 3)          # three readers: query, key, aggregated values
        key_feat = feat_k if M == N else feat_k[:, :M].contiguous()
        # query and key read the same rows through two dropout masks: two independent products of one shape, launched as pairs
        # (forward and weight gradients; their input gradients join one running gradient of roi_feat and stay a chain)
        q_data, k_data = fcnet_pair(self.query, feat_q, self.key, key_feat)
        q_data, k_data = q_data.view(R, N, A), k_data.view(R, M, A)
        geo = None
        if position_embedding is not None and self.pos_emb_dim > 0:
            pe = position_embedding.float() if position_embedding.dtype != torch.float32 else position_embedding
            # relu(pair_pos_fc1(.)): FCNet(act=None) + F.relu of the reference (:121-122) fused in one epilogue
            fc = self.pair_pos_fc1
            pe2 = pe.reshape(-1, self.pos_emb_dim)
            if not (self.training and getattr(pe, '_ekaid_dropped', 0.0) == float(fc.main[0].p)):   # (else: dropped as it was written)
                pe2 = ops.dropout(pe2, fc.main[0].p, self.training)
            geo = fc.main[1](pe2, ACT_RELU).view(R, N, M, self.fc_dim)
        if adj_matrix is None:
            raise ValueError('GraphSelfAttentionLayer: adj_matrix=None is not on the EKAID path')
        out_t, aff_softmax = ops.gat_attention(q_data, k_data, feat_s, adj_matrix, label_biases_att, geo,
                                               self.num_heads, direction, self.attention_impl)
        output = self.linear_out_2(out_t.view(R * N, self.fc_dim * A)).view(R, N, self.dim[2])
        return output, aff_softmax
