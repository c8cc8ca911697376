"""Question encoder -- mirror of reference ``models/language_model.py:17-156``.

``WordEmbedding`` (two 300-d tables, the second frozen), ``QuestionEmbedding`` (1-layer GRU, h0 = 0) and
``QuestionSelfAttention`` with the reference's batch-coupled softmax (quirk Q-C, :149): the logits are
soft-maxed over the BATCH axis of the (Lq, B) transpose and that buffer is re-read as (B, 1, Lq).
``QuestionSelfAttention.gather_logits`` is the hook data-parallel training uses for the optional
"global-batch semantics" (all-gather of the (B_local, Lq) logits).
"""
import torch
import torch.nn as nn

from .. import ops
from ..hip import ACT_TANH
from .fc import FCNet


class WordEmbedding(nn.Module):
    """The ntoken-th dim is used for padding_idx (reference docstring; NB it is a real word id, the
    question padding id is 0 -- reproduced as is)."""

    def __init__(self, ntoken, emb_dim, dropout, op=''):
        super(WordEmbedding, self).__init__()
        self.op = op
        self.emb = nn.Embedding(ntoken + 1, emb_dim, padding_idx=ntoken)
        if 'c' in op:
            self.emb_ = nn.Embedding(ntoken + 1, emb_dim, padding_idx=ntoken)
            self.emb_.weight.requires_grad = False  # fixed
        self.dropout = nn.Dropout(dropout)
        self.ntoken = ntoken
        self.emb_dim = emb_dim

    def forward(self, x):
        tables = [self.emb.weight] + ([self.emb_.weight] if 'c' in self.op else [])
        emb = ops.embedding(x, tables, relu=False, padding_idx=self.ntoken)
        return ops.dropout(emb, self.dropout.p, self.training)


class QuestionEmbedding(nn.Module):
    def __init__(self, in_dim, num_hid, nlayers, bidirect, dropout, rnn_type='GRU'):
        super(QuestionEmbedding, self).__init__()
        if rnn_type != 'GRU' or nlayers != 1 or bidirect:
            raise NotImplementedError('the EKAID path uses a 1-layer unidirectional GRU (modules.py:148-149)')
        self.rnn = nn.GRU(in_dim, num_hid, nlayers, bidirectional=bidirect, dropout=dropout, batch_first=True)
        self.in_dim = in_dim
        self.num_hid = num_hid
        self.nlayers = nlayers
        self.rnn_type = rnn_type
        self.ndirections = 1 + int(bidirect)

    def forward_all(self, x):
        r = self.rnn
        return ops.gru_sequence(x, r.weight_ih_l0, r.weight_hh_l0, r.bias_ih_l0, r.bias_hh_l0)

    def forward(self, x):
        return self.forward_all(x)[:, -1]


class QuestionSelfAttention(nn.Module):
    def __init__(self, num_hid, dropout):
        super(QuestionSelfAttention, self).__init__()
        self.num_hid = num_hid
        self.drop = nn.Dropout(dropout)
        self.W1_self_att_q = FCNet(dims=[num_hid, num_hid], dropout=dropout, act=None)
        self.W2_self_att_q = FCNet(dims=[num_hid, 1], act=None)
        self.gather_logits = None      # optional (atten (B,L)) -> (global atten (Bg,L), row offset of the local slice)

    def forward(self, ques_feat):
        B, L, H = ques_feat.shape
        fc1 = self.W1_self_att_q
        x = ops.dropout(ques_feat.reshape(-1, H), fc1.main[0].p, self.training)
        atten_1 = fc1.main[1](x, ACT_TANH)                      # tanh fused into the GEMM epilogue (:138-140)
        atten = self.W2_self_att_q(atten_1).view(B, L)
        if self.gather_logits is None:
            weight = ops.QAttWeightsFn.apply(atten)             # (B, L), quirk Q-C
        else:
            g_atten, off = self.gather_logits(atten)
            weight = ops.QAttWeightsFn.apply(g_atten)[off:off + B]
        out = ops.weighted_pool(weight, ques_feat)              # bmm((B,1,L), (B,L,H))
        return ops.dropout(out, self.drop.p, self.training)
