"""``DynamicCore`` / ``DynamicSpeaker`` -- mirror of reference
``models/dynamic_speaker_change_pos.py:42-357`` (the answer decoder: module-attention LSTMCell, POS
branch, gate MLP, language LSTMCell, ``logit`` + log-softmax per step).

Same constructors, ``_forward`` / ``_sample`` / ``get_logprobs_state`` / ``get_module_weights`` signatures,
``ss_prob`` attribute and ``state_dict`` keys.  Differences that do not change results:
  * the loop-invariant ``core.embed`` Linear+ReLU is computed once per sequence (the reference recomputes
    it every step, :98-99); its Dropout is still drawn per step;
  * ``torch.cat`` inputs of the LSTM cells / gate MLP are fed as split-K GEMMs, never materialised;
  * the teacher-forced loop length is found with ONE host synchronisation instead of one per step (:213).
Scheduled sampling (``ss_prob > 0``, :197-208; inactive with the shipped config, config.py:137) runs step by step through
autograd with the token draw on the device (``ekaid_ss_tokens``).
``_sample`` covers every branch of the reference loop (:287-357): greedy decoding as one C-ABI call per answer batch, multinomial
sampling with ``temperature`` (:336-343) and the decoding constraint (:324-327) step by step through ``get_logprobs_state``.
Not supported (outside the path): beam search (``beam_size > 1``: broken in the reference, :273).
"""
import torch
import torch.nn as nn

from .. import ops
from ..hip import ACT_NONE, ACT_RELU, ACT_SIGMOID
from .CaptionModel import CaptionModel
from .fc import Linear


class TokenEmbedding(nn.Embedding):
    """nn.Embedding parameters; lookup (+ fused ReLU) through the HIP kernel."""

    def forward(self, it, relu=False):
        return ops.embedding(it, [self.weight], relu=relu, padding_idx=-1)


class DynamicCore(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        sp = cfg.model.speaker
        self.rnn_num_layers = 2
        self.drop_prob_lm = sp.drop_prob_lm
        self.input_dim = sp.input_dim
        self.embed_input_dim = sp.embed_input_dim
        self.embed_dim = sp.embed_dim
        self.rnn_size = sp.rnn_size

        self.embed = nn.Sequential(Linear(self.embed_input_dim, self.embed_dim), nn.ReLU(), nn.Dropout(self.drop_prob_lm))
        self.module_att_lstm = nn.LSTMCell(self.embed_dim + sp.rnn_size, sp.rnn_size)
        self.weight_fc = nn.Sequential(Linear(sp.rnn_size, 3), nn.Softmax(dim=1))
        self.pos1 = nn.Sequential(Linear(512, 512), nn.ReLU(), nn.Dropout(self.drop_prob_lm))
        self.softmax = nn.Softmax(dim=1)
        self.dropout = nn.Dropout(0.5)
        self.weight_pos = Linear(sp.rnn_size, 16)
        self.pos2 = Linear(16, sp.rnn_size)
        self.gate1x = nn.Sequential(Linear(sp.rnn_size * 2 + sp.input_dim, sp.rnn_size * 2 + sp.input_dim), nn.ReLU(),
                                    nn.Dropout(self.drop_prob_lm))
        self.gate2x = Linear(sp.rnn_size * 2 + sp.input_dim, sp.input_dim)
        self.lang_lstm = nn.LSTMCell(sp.word_embed_size + sp.input_dim, sp.rnn_size)
        self.module_weights = None

    def embed_pre(self, loc_feat_bef, loc_feat_aft, feat_diff):
        """Linear + ReLU of ``self.embed`` on [bef | diff | aft] (loop invariant; dropout applied per step)."""
        return self.embed[0]((loc_feat_bef, feat_diff, loc_feat_aft), ACT_RELU)

    @staticmethod
    def _lstm(cell, xs, h, c):
        gx = ops.linear(xs, cell.weight_ih, cell.bias_ih)
        gh = ops.linear(h, cell.weight_hh, cell.bias_hh)
        return ops.LSTMCellFn.apply(gx, gh, c)

    def forward(self, xt, loc_feat_bef, loc_feat_aft, feat_diff, state, embed_pre=None):
        prev_h = state[0][-1]                                   # hidden state of the lang_lstm
        if embed_pre is None:
            embed_pre = self.embed_pre(loc_feat_bef, loc_feat_aft, feat_diff)
        embed = ops.dropout(embed_pre, self.drop_prob_lm, self.training)
        h_mod_att, c_mod_att = self._lstm(self.module_att_lstm, (embed, prev_h), state[0][0], state[1][0])
        module_weights = ops.softmax(self.weight_fc[0](h_mod_att))
        self.module_weights = module_weights

        vpos = ops.dropout(self.pos1[0](prev_h, ACT_RELU), self.drop_prob_lm, self.training)
        dpos = ops.dropout(self.weight_pos(vpos), self.dropout.p, self.training)
        ppos = self.pos2(ops.softmax(dpos))

        att_feat = ops.Mix3Fn.apply(module_weights, loc_feat_bef, feat_diff, loc_feat_aft)
        g1 = ops.dropout(self.gate1x[0]((prev_h, ppos, att_feat), ACT_RELU), self.drop_prob_lm, self.training)
        gate = self.gate2x(g1, ACT_SIGMOID)
        gated_att_feat = ops.mul(gate, att_feat)

        h_lang, c_lang = self._lstm(self.lang_lstm, (xt, gated_att_feat), state[0][1], state[1][1])
        output = ops.dropout(h_lang, self.drop_prob_lm, self.training)
        state = ((h_mod_att, h_lang), (c_mod_att, c_lang))
        return output, state, dpos

    def get_module_weights(self):
        return self.module_weights


class DynamicSpeaker(CaptionModel):
    def __init__(self, cfg, vocab_size=0):
        super().__init__()
        sp = cfg.model.speaker
        self.vocab_size = sp.vocab_size if vocab_size == 0 else vocab_size
        self.word_embed_size = sp.word_embed_size
        self.rnn_size = sp.rnn_size
        self.drop_prob_lm = sp.drop_prob_lm
        self.seq_length = sp.seq_length
        self.ss_prob = 0.0  # Scheduled sampling probability

        self.embed = nn.Sequential(TokenEmbedding(self.vocab_size, self.word_embed_size), nn.ReLU(),
                                   nn.Dropout(self.drop_prob_lm))
        self.core = DynamicCore(cfg)
        self.rnn_num_layers = self.core.rnn_num_layers
        self.logit_layers = getattr(sp, 'logit_layers', 1) if not isinstance(sp, dict) else sp.get('logit_layers', 1)
        if self.logit_layers != 1:
            raise NotImplementedError('logit_layers > 1 is not configured anywhere in the reference')
        self.logit = Linear(self.rnn_size, self.vocab_size)
        self.module_weights = []
        self.fused_sequence = True      # _forward as one BPTT node; False = step-by-step through autograd
        self.greedy_poll = 8            # step-by-step greedy _sample: host checks "all finished" every this many steps
        self.greedy_poll_fused = None   # fused greedy _sample: steps per C-ABI call (None: speaker_seq.greedy_decode chooses)

    def init_hidden(self, batch_size):
        """((h_att, h_lang), (c_att, c_lang)); indexable like the reference's stacked (2, B, rnn) tensors."""
        weight = self.logit.weight
        z = weight.new_zeros(batch_size, self.rnn_size)
        return ((z, z), (z, z))

    def _word(self, it):
        return ops.dropout(self.embed[0](it, relu=True), self.drop_prob_lm, self.training)

    @staticmethod
    def count_steps(seq, seq_length):
        """Number of decode steps the reference loop runs: up to the first column i >= 1 whose tokens are
        all 0 (:213).  One host read of <= seq_length flags; call it on the HOST copy of the labels to
        keep the training step free of device->host synchronisation."""
        nz = (seq[:, :seq_length] != 0).any(0).tolist()
        for i in range(1, seq_length):
            if not nz[i]:
                return i
        return seq_length

    def _forward(self, feat_bef, feat_aft, feat_diff, seq, num_steps=None):
        self.module_weights = []
        scheduled = self.training and self.ss_prob > 0.0
        batch_size = feat_bef.size(0)
        state = self.init_hidden(batch_size)
        # "break if all the sequences end" (:213): ``num_steps`` (optional, extension) lets the caller supply
        # the count computed on the host copy of the labels; otherwise it costs one device->host read
        steps = num_steps if num_steps is not None else self.count_steps(seq, self.seq_length)
        if self.fused_sequence and not scheduled:
            # whole teacher-forced sequence as one autograd node with explicit BPTT (ekaid_amd/speaker_seq.py)
            from ..speaker_seq import speaker_sequence
            tok_tb = seq[:, :steps].t().contiguous()
            logp_tb, logpos_tb, mw_tb = speaker_sequence(self, feat_bef, feat_aft, feat_diff, tok_tb)
            self.module_weights = list(mw_tb.unbind(0))
            outputs, outputs_pos = logp_tb.transpose(0, 1), logpos_tb.transpose(0, 1)
            if steps < self.seq_length:
                pad = self.seq_length - steps
                outputs = torch.cat([outputs, outputs.new_zeros(batch_size, pad, self.vocab_size)], 1)
                outputs_pos = torch.cat([outputs_pos, outputs_pos.new_zeros(batch_size, pad, 16)], 1)
            return outputs, outputs_pos
        embed_pre = self.core.embed_pre(feat_bef, feat_aft, feat_diff)
        outs, outs_pos = [], []
        for i in range(steps):
            it = seq[:, i].contiguous()
            if scheduled and i >= 1:
                # scheduled sampling (:197-208): per sample, with probability ss_prob, feed a draw from the previous step's
                # output distribution instead of the teacher token (step by step: the next input depends on the last output)
                from ..hip import check, lib, ptr, stream
                drawn = torch.empty_like(it)
                check(lib().ekaid_ss_tokens(ptr(outs[-1].detach().contiguous()), seq.data_ptr() + 8 * i, seq.stride(0), batch_size,
                                            self.vocab_size, float(self.ss_prob), ops.next_seed(), ptr(drawn), stream()), 'ss_tokens')
                it = drawn
            output, state, output_pos = self.get_logprobs_state(it, feat_bef, feat_aft, feat_diff,
                                                                state, embed_pre)
            outs.append(output)
            outs_pos.append(output_pos)
        outputs = torch.stack(outs, 1)
        outputs_pos = torch.stack(outs_pos, 1)
        if steps < self.seq_length:      # the reference returns zero rows after the break
            pad = self.seq_length - steps
            outputs = torch.cat([outputs, outputs.new_zeros(batch_size, pad, self.vocab_size)], 1)
            outputs_pos = torch.cat([outputs_pos, outputs_pos.new_zeros(batch_size, pad, 16)], 1)
        return outputs, outputs_pos

    def get_logprobs_state(self, it, feat_bef, feat_aft, feat_diff, state, embed_pre=None):
        xt = self._word(it)
        output, state, output_pos = self.core(xt, feat_bef, feat_aft, feat_diff, state, embed_pre)
        self.module_weights.append(self.core.get_module_weights())
        log_probs = ops.log_softmax(self.logit(output))
        log_pos_probs = ops.log_softmax(output_pos)
        return log_probs, state, log_pos_probs

    def get_module_weights(self):
        if len(self.module_weights) == 0:
            print('no module weights accumulated')
            return None
        return torch.stack(self.module_weights, dim=1)

    def _sample_beam(self, *args, **kwargs):
        raise NotImplementedError('beam search is outside the EKAID path (beam_size == 1; the reference '
                                  'implementation is broken, dynamic_speaker_change_pos.py:273)')

    def _sample(self, feat_bef, feat_aft, feat_diff, seq, cfg={}, sample_max=0):
        self.module_weights = []
        sp = cfg.model.speaker if hasattr(cfg, 'model') else {}
        beam_size = sp.get('beam_size', 1)
        decoding_constraint = sp.get('decoding_contraint', 0)
        if beam_size > 1:
            return self._sample_beam(feat_bef, feat_aft, feat_diff, cfg)
        batch_size = feat_bef.size(0)
        state = self.init_hidden(batch_size)
        seq = feat_bef.new_zeros((batch_size, self.seq_length), dtype=torch.long)
        seq_logprobs = feat_bef.new_zeros(batch_size, self.seq_length)
        embed_pre = self.core.embed_pre(feat_bef, feat_aft, feat_diff)
        it = feat_bef.new_zeros(batch_size, dtype=torch.long) + 2          # <bos> hard-coded as id 2 (:311)
        if sample_max and not decoding_constraint and not self.training and self.fused_sequence:
            # greedy path (test_mimic.py:119-122), eval mode: ``greedy_poll_fused`` decode steps per C-ABI call
            # (ekaid_speaker_greedy_steps: the chosen tokens feed the next step on the device), one host read per call
            from ..speaker_seq import greedy_decode
            seq, seq_logprobs, self.module_weights = greedy_decode(self, feat_bef, feat_aft, feat_diff, self.greedy_poll_fused)
            return seq, seq_logprobs
        if sample_max and not decoding_constraint:
            # the same loop step by step (train mode / under autograd / fused_sequence off): argmax / stop-flag bookkeeping in one
            # kernel per step; the "all finished" test (:354) is polled every ``self.greedy_poll`` steps
            from ..hip import check, lib, ptr, stream
            unfinished = torch.zeros(batch_size, dtype=torch.int32, device=feat_bef.device)
            counts = torch.zeros(self.seq_length, dtype=torch.int32, device=feat_bef.device)
            for t in range(self.seq_length):
                logprobs, state, _ = self.get_logprobs_state(it, feat_bef, feat_aft, feat_diff, state, embed_pre)
                it = torch.empty_like(it)
                check(lib().ekaid_greedy_step(ptr(logprobs.detach().contiguous()), batch_size, self.vocab_size, t, ptr(unfinished),
                                              ptr(it), ptr(seq), ptr(seq_logprobs), self.seq_length,
                                              counts.data_ptr() + 4 * t, stream()), 'greedy_step')
                if (t + 1) % self.greedy_poll == 0 or t == self.seq_length - 1:
                    done = (counts[:t + 1] == 0).nonzero()
                    if done.numel():                                       # everything had finished at step `first`
                        first = int(done[0])
                        seq[:, first + 1:] = 0
                        seq_logprobs[:, first + 1:] = 0
                        break
            return seq, seq_logprobs
        # the remaining branches of the reference loop (:320-357), step by step through get_logprobs_state (HIP kernels); the token
        # bookkeeping is a handful of device-side tensor ops per step, the "all finished" test is polled like above:
        #   * multinomial sampling (sample_max == 0) with cfg.model.speaker.temperature (:336-343),
        #   * the decoding constraint (never repeat the word sampled at t - 1, :324-327), for either way of choosing.
        # The random stream is torch's device generator, so the draws differ from the reference's CPU draws by construction.
        temperature = float(sp.get('temperature', 1.0))
        unfinished = None
        for t in range(self.seq_length):
            logprobs, state, _ = self.get_logprobs_state(it, feat_bef, feat_aft, feat_diff, state, embed_pre)
            lp = logprobs.detach().clone()
            if t == 0:
                lp[:, 0] = float('-inf')                                   # never sample NULL first (:318-322)
            if decoding_constraint and t > 0:
                lp.scatter_(1, seq[:, t - 1:t], float('-inf'))
            if sample_max:
                sample_logprobs, it = torch.max(lp, 1)
            else:
                # exp(logprobs / temperature) as in the reference, shifted by the row maximum first: the same distribution once
                # multinomial normalises it, but no row underflows to all zeros at a small temperature
                shifted = lp - lp.max(1, keepdim=True).values
                prob_prev = torch.exp(shifted) if temperature == 1.0 else torch.exp(shifted / temperature)
                it = torch.multinomial(prob_prev, 1)
                sample_logprobs = lp.gather(1, it).view(-1)                # the log-prob of the sampled word, not tempered (:342)
                it = it.view(-1)
            it = it.long()
            # the reference leaves the loop at the first step after which nobody is running (:354-356); polled here, so the steps in
            # between are blanked by a device-side flag: (everything had stopped before this step) -> nothing is recorded
            running = None if t == 0 else (unfinished.sum() > 0)
            unfinished = (it > 0) if t == 0 else unfinished * (it > 0)
            it = it * unfinished.type_as(it)
            seq[:, t] = it
            seq_logprobs[:, t] = sample_logprobs.view(-1) if running is None else sample_logprobs.view(-1) * running.to(sample_logprobs.dtype)
            if ((t + 1) % self.greedy_poll == 0 or t == self.seq_length - 1) and int(unfinished.sum()) == 0:
                break
        return seq, seq_logprobs
