"""``ChangeDetector`` (+ ``SelfAttention``, ``AddSpatialInfo``) -- mirror of reference
``models/modules.py:17-332``: same constructor (``cfg``, ``word_to_idx``), same ``forward`` signature and
6-tuple return, same ``state_dict`` keys and shapes (dead parameters ``SSRE.*`` / ``fc1.*`` /
``*.neighbor_net.0.*`` / ``*.linear_out_.*`` included, so reference checkpoints load unchanged).

MI355X mapping: the main and the reference image are stacked into one batch of R = 2B graph slices
so that every dense contraction runs once over 2*B*N rows; all arithmetic is libekaid_hip.so.
Behavioural quirks reproduced on purpose: Q-A chained residual encoders,
Q-B surviving direction only, Q-C batch-coupled question attention, Q-D coefficients read from
``cfg.model.change_detector``.
"""
import math

import torch
import torch.nn as nn

from .. import ops
from ..hip import ACT_RELU, ACT_SIGMOID
from .fc import Linear
from .language_model import QuestionEmbedding, QuestionSelfAttention, WordEmbedding
from .relation_encoder import ExplicitRelationEncoder, ImplicitRelationEncoder, q_expand_v_cat


class SelfAttention(nn.Module):
    """SSRE block (modules.py:17-77): Q/K/V Linear(2A -> A), 4-head softmax attention, dropout .1 on the
    probabilities, LayerNorm(A, eps=1e-6), no residual.  Only reached with ``setting='mode0'``."""

    def __init__(self, cfg):
        super(SelfAttention, self).__init__()
        cd = cfg.model.change_detector
        if cd.att_dim % cd.att_head != 0:
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention "
                             "heads (%d)" % (cd.att_dim, cd.att_head))
        self.num_attention_heads = cd.att_head
        self.attention_head_size = int(cd.att_dim / cd.att_head)
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        self.query = Linear(cd.att_dim * 2, self.all_head_size)
        self.key = Linear(cd.att_dim * 2, self.all_head_size)
        self.value = Linear(cd.att_dim * 2, self.all_head_size)
        self.dropout = nn.Dropout(0.1)
        self.layer_norm = nn.LayerNorm(cd.att_dim, eps=1e-6)

    def forward(self, query_states, key_states, value_states, bl=None):
        """Each state: tensor (B, L, 2A), or tuple (v (B*L, A), q_exp (B*L, A)) standing for the concat
        (then ``bl = (B, L)`` must be given)."""
        if isinstance(query_states, (tuple, list)):
            B, L = bl
        else:
            B, L = query_states.shape[0], query_states.shape[1]
        # heads as a strided batch dimension of the two products: no head-major copies of q / k / v, the context lands in (B, L, A)
        q = self.query(query_states).view(B, L, -1)
        k = self.key(key_states).view(B, L, -1)
        v = self.value(value_states).view(B, L, -1)
        ctx = ops.head_attention(q, k, v, self.num_attention_heads, self.dropout.p, self.training)
        return ops.layer_norm(ctx, self.layer_norm.weight, self.layer_norm.bias, self.layer_norm.eps)


class ChangeDetector(nn.Module):

    def __init__(self, cfg, word_to_idx):
        super().__init__()
        cd = cfg.model.change_detector
        self.input_dim = cd.input_dim
        self.dim = cd.dim
        self.feat_dim = cd.feat_dim - 2
        self.att_head = cd.att_head
        self.att_dim = cd.att_dim
        self.nongt_dim = cd.nongt_dim
        self.pos_emb_dim = cd.pos_emb_dim

        self.img = Linear(self.feat_dim, self.att_dim)
        self.SSRE = SelfAttention(cfg)
        self.context1 = Linear(self.att_dim, self.att_dim, bias=False)
        self.context2 = Linear(self.att_dim, self.att_dim)
        self.gate1 = Linear(self.att_dim, self.att_dim, bias=False)
        self.gate2 = Linear(self.att_dim, self.att_dim)
        self.dropout = nn.Dropout(0.5)
        self.embed = nn.Sequential(Linear(self.att_dim * 3, self.dim), nn.Dropout(0.5), nn.ReLU())
        self.att = Linear(self.dim, 1)
        self.fc1 = Linear(self.att_dim, 6)

        self.coef_sem = cd.coef_sem        # Q-D: read from cfg.model.change_detector, not cfg.model
        self.coef_spa = cd.coef_spa
        assert self.coef_sem + self.coef_spa <= 1

        q_dim = cfg.model.speaker.embed_dim
        if cfg.train.setting == 'mode2':
            g = cfg.train.graph
            if g == 'all' or g == 'semantic':
                self.semantic_relation = ExplicitRelationEncoder(
                    cd.att_dim, q_dim, cd.att_dim, cd.dir_num, cd.sem_label_num, num_heads=cd.att_head,
                    num_steps=1, nongt_dim=cd.nongt_dim, residual_connection=True, label_bias=False)
            if g == 'all' or g == 'spatial' or g == 'i+s':
                self.spatial_relation = ExplicitRelationEncoder(
                    cd.att_dim, q_dim, cd.att_dim, cd.dir_num, cd.spa_label_num, num_heads=cd.att_head,
                    num_steps=1, nongt_dim=cd.nongt_dim, residual_connection=True, label_bias=False)
            if g == 'all' or g == 'implicit' or g == 'i+s':
                self.imp_relation = ImplicitRelationEncoder(
                    cd.att_dim, q_dim, cd.att_dim, cd.dir_num, 64, cd.nongt_dim, num_heads=cd.att_head,
                    num_steps=1, residual_connection=True, label_bias=False)

        self.w_emb = WordEmbedding(len(word_to_idx), 300, .0, 'c')
        self.q_emb = QuestionEmbedding(600, q_dim, 1, False, .0)
        self.q_att = QuestionSelfAttention(q_dim, .2)
        self.cfg = cfg
        if cfg.data.feature_mode == 'mode0':
            raise NotImplementedError("feature_mode='mode0' (raw-image ResNet-101 extractor, modules.py:153-160) "
                                      "is outside the ROI-feature hot path")
        self._dim_mat = None

    # ------------------------------------------------------------------ geometry
    def position_emb(self, bb, reader=None):
        """(R, N, 4) float64 boxes (host or device) -> (R, N, M, pos_emb_dim) float32 on the device.
        ref: modules.py:162-166 -> mimic_utils.py:152-208 (float64 math, cast at graph_att_layer.py:115).
        ``reader``: the implicit relation encoder when it is the embedding's only reader -- in training its pair_pos_fc1 input dropout
        is applied as the embedding is written (one pass over 170 MB at 256 pairs instead of three)."""
        dev = self.img.weight.device
        if self._dim_mat is None or self._dim_mat.device != dev:
            feat_range = torch.arange(0, self.pos_emb_dim / 8)
            self._dim_mat = torch.pow(torch.ones((1,)) * 1000, (8. / self.pos_emb_dim) * feat_range).to(dev)
        bb = bb.to(device=dev, dtype=torch.float64)
        # nongt_dim < regions per image: the reference slices the ROW axis of the pair matrix (mimic_utils.py:186) and then
        # re-reads the (nongt, N) block as (N, nongt) (graph_att_layer.py:117-126); ekaid_geom_embed reproduces that
        # re-indexing (no shipped configuration reaches it: nongt_dim = 52 >= regions per image)
        M = min(self.nongt_dim, bb.shape[1])
        p = 0.0
        if reader is not None and reader.training and reader.num_steps == 1 and not reader.implicit_relation.compute_dead_directions:
            p = float(reader.implicit_relation.neighbor_net[-1].pair_pos_fc1.input_dropout())
        return ops.geom_embed(bb, M, self.pos_emb_dim, self._dim_mat, p)

    def _wn_layers(self, setting, graph):
        """(weight_v, weight_g) of the weight-normed layers a forward call with this setting / graph runs: the encoders
        selected by ``graph``, minus the dead direction (quirk Q-B) unless it is computed for its attention maps."""
        from .fc import WNLinear
        roots = [('q_att', self.q_att)]
        if setting == 'mode2':
            if graph in ('semantic', 'all') and hasattr(self, 'semantic_relation'):
                roots.append(('semantic_relation', self.semantic_relation))
            if graph in ('spatial', 'all', 'i+s') and hasattr(self, 'spatial_relation'):
                roots.append(('spatial_relation', self.spatial_relation))
            if graph in ('implicit', 'all', 'i+s') and hasattr(self, 'imp_relation'):
                roots.append(('imp_relation', self.imp_relation))
        out = []
        for _, root in roots:
            gats = [m for m in root.modules() if hasattr(m, 'compute_dead_directions')]
            dead_ok = any(g.compute_dead_directions for g in gats)
            last = max([g.dir_num for g in gats], default=1) - 1
            for name, m in root.named_modules():
                if not isinstance(m, WNLinear):
                    continue
                if 'neighbor_net.' in name and not dead_ok and ('neighbor_net.%d.' % last) not in name:
                    continue
                if 'pair_pos_fc1' in name and not hasattr(root, 'implicit_relation'):
                    continue
                out.append((m.weight_v, m.weight_g))
        return out

    def encode_question(self, question):
        return self.q_att(self.q_emb.forward_all(self.w_emb(question)))

    # ------------------------------------------------------------------ forward
    def forward(self, input_1, input_2, d_adj_matrix, q_adj_matrix, d_sem_adj_matrix, q_sem_adj_matrix, d_bb, q_bb,
                question, setting='mode2', graph='all'):
        if self.cfg.data.train.empty_image == True:  # noqa: E712  (reference semantics, modules.py:170-178)
            input_1, input_2 = torch.ones_like(input_1), torch.ones_like(input_2)
            d_adj_matrix, q_adj_matrix = torch.ones_like(d_adj_matrix), torch.ones_like(q_adj_matrix)
            d_sem_adj_matrix, q_sem_adj_matrix = torch.ones_like(d_sem_adj_matrix), torch.ones_like(q_sem_adj_matrix)
            d_bb, q_bb = torch.ones_like(d_bb), torch.ones_like(q_bb)
        B, N, C = input_1.size()
        A = self.att_dim
        # ||v|| of every weight-normed layer this call will run, in two launches (the layers pick the values up)
        wn = self._wn_layers(setting, graph)
        if wn and input_1.is_cuda:
            ops.wn_prepare(wn)
        if input_1.is_cuda and ops.get_gemm_mode() == ops.GEMM_F16X2:
            ops.bounds_pool_reset(input_1.device)            # (the step begins here: one clear for the column maxima of all its bounds passes)
            # two-term f16 GEMM split: the row / column maxima of every weight this call multiplies with, one launch per 32 weights
            plain = [self.img.weight, self.context1.weight, self.context2.weight, self.gate1.weight, self.gate2.weight, self.embed[0].weight]
            live_v = {v.data_ptr() for v, _ in wn}
            for _, m in self.named_modules():             # linear_out_2 of the graph-attention layers whose query / key run (same liveness)
                lo = getattr(m, 'linear_out_2', None)
                if lo is not None and getattr(m, 'query', None) is not None and any(p.data_ptr() in live_v for p in m.query.parameters()):
                    plain.append(lo.weight)
            ops.weight_bounds_prepare([v for v, _ in wn] + plain)

        # both images through the shared ROI projection, stacked: rows [0, B*N) = main, [B*N, 2*B*N) = reference
        x = ops.linear_rowstack((input_1, input_2), self.img.weight, self.img.bias).view(2 * B, N, A)
        q = self.encode_question(question)

        if setting == 'mode2':
            n_enc = {'all': 3, 'i+s': 2}.get(graph, 1)
            qs = list(ops.fanout(q, n_enc))                  # the question vector feeds every encoder
            if graph == 'semantic' or graph == 'all':
                x, _ = self.semantic_relation.forward(x, ops.stack2(d_sem_adj_matrix, q_sem_adj_matrix), qs.pop())
            if graph == 'spatial' or graph == 'all' or graph == 'i+s':
                x, _ = self.spatial_relation.forward(x, ops.stack2(d_adj_matrix, q_adj_matrix), qs.pop())
            if graph == 'implicit' or graph == 'all' or graph == 'i+s':
                pos = self.position_emb(ops.stack2(d_bb, q_bb), self.imp_relation)
                x, _ = self.imp_relation.forward(x, pos, qs.pop())
            if graph == 'all':
                # Q-A: input_bef1/2/3 are one tensor -> cs*x + cp*x + (1-cs-cp)*x   (modules.py:234-235)
                cs, cp = self.coef_sem, self.coef_spa
                x = ops.self_mix3(x, cs, cp, 1 - cs - cp)
            elif graph == 'i+s':
                x = ops.scale(ops.add(x, x), 0.5)
        elif setting == 'mode0':
            xq = q_expand_v_cat(q, x)
            x = self.SSRE(xq, xq, xq, bl=(2 * B, N))
        else:
            raise ValueError("setting %r references a non-existent graph_relation in the reference "
                             "(modules.py:214,257,263) and is not supported" % (setting,))

        # every tensor with several readers goes through ops.fanout / ops.halves: its gradient is then ONE n-ary sum
        # instead of autograd's zero-fill + copy per slice and pairwise adds
        x = x.reshape(2 * B, N, A).contiguous()
        x_c, x_g, x_d, x_e, x_p = ops.fanout(x, 5)
        bef_d, aft_d = ops.halves(x_d)                                      # bef = x[:B], aft = x[B:]
        diff = ops.sub(aft_d, bef_d)                                        # (B, N, A)
        diff_c, diff_g, diff_e = ops.fanout(diff, 3)
        # context / gate read the same rows with different weights: pairs of independent products, launched pairwise
        c1d, g1d = ops.linear_pair((diff_c, self.context1.weight, self.context1.bias, None),     # shared by both images: computed once
                                   (diff_g, self.gate1.weight, self.gate1.bias, None))
        c2, g2 = ops.linear_pair((x_c, self.context2.weight, self.context2.bias, None),          # (2B, N, A), bias in the epilogue
                                 (x_g, self.gate2.weight, self.gate2.bias, None))
        gated = ops.gate_fuse(c2, g2, c1d, g1d, self.dropout.p, self.training).view(2 * B, N, A)

        # region attention of BOTH images in one pass over the 2B row-stacked slices (the reference calls it per image,
        # modules.py:286-299: same weights, row-independent layers): embed([feat | diff | gated]) -> att -> sigmoid, then pooling
        emb_lin, emb_drop = self.embed[0], self.embed[1]
        e = emb_lin((x_e, ops.repeat2(diff_e), gated), ACT_RELU)            # relu(dropout(x)) == dropout(relu(x))
        e = ops.dropout(e, emb_drop.p, self.training)
        att = self.att(e, ACT_SIGMOID).view(2 * B, N)
        att_p, att_o = ops.fanout(att, 2)                                   # pooling weights / returned attention maps
        attended = ops.weighted_pool(att_p, x_p)                            # (2B, A)
        att_b, att_a = ops.halves(att_o)
        attended_1, attended_2 = ops.halves(attended)
        input_attended = ops.sub(attended_2, attended_1)
        pred = self.fc1(input_attended)
        ops.wn_release()
        ops.weight_bounds_release()
        return pred, att_b.view(B, 1, N), att_a.view(B, 1, N), attended_1, attended_2, input_attended


class AddSpatialInfo(nn.Module):
    """Instantiated by the entry scripts, never applied on the ROI path (modules.py:316-332)."""

    def _create_coord(self, img_feat):
        batch_size, _, h, w = img_feat.size()
        ys = (torch.arange(h, device=img_feat.device, dtype=img_feat.dtype) * 2.0 / h) - 1
        xs = (torch.arange(w, device=img_feat.device, dtype=img_feat.dtype) * 2.0 / w) - 1
        coord_map = torch.stack([xs.view(1, w).expand(h, w), ys.view(h, 1).expand(h, w)])
        return coord_map.unsqueeze(0).expand(batch_size, -1, -1, -1)

    def forward(self, img_feat):
        return torch.cat([img_feat, self._create_coord(img_feat)], dim=1)
