"""Loss / optimiser / checkpoint helpers -- counterparts of reference ``utils/utils.py`` (only the
functions the entry points use: :11-23, :28-43, :70-75, :82-101, :122-136, :171-179, :204-216)."""
import json

import torch
import torch.nn as nn

from .. import ops


def load_cpu(path):
    return torch.load(path, map_location=lambda storage, loc: storage, weights_only=False)


def save_checkpoint(checkpoint, filename):
    print('Saving checkpoint to %s' % filename)
    torch.save(checkpoint, filename)


def load_checkpoint(filename):
    print('Loading checkpoint from %s' % filename)
    return load_cpu(filename)


class AverageMeter(object):
    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def set_mode(mode, models):
    assert mode in ['train', 'eval']
    for m in models:
        if m is None:
            continue
        m.train() if mode == 'train' else m.eval()


class LanguageModelCriterion(nn.Module):
    """-sum(logp[target] * mask) / sum(mask)  (utils.py:204-216), one fused kernel pair."""

    def forward(self, input, target, mask, emphasize_last=False):
        T = input.size(1)
        target = target[:, :T]
        mask = mask[:, :T]
        if mask.dtype != torch.float32:
            mask = mask.float()
        return ops.lm_loss(input, target, mask)


def total_loss(speaker_loss, att_bef, att_aft):
    """speaker_loss + 2.5e-3 * (sum(att_bef) + sum(att_aft)) / (2B)   (train_mimic.py:246-247)."""
    return ops.total_loss(speaker_loss, att_bef, att_aft)


def decode_sequence(ix_to_word, seq):
    N, D = seq.size()
    seq = seq.tolist()
    out = []
    for i in range(N):
        words = []
        for j in range(D):
            ix = seq[i][j]
            if ix > 0:
                words.append(ix_to_word[ix])
            else:
                break
        out.append(' '.join(words))
    return out


def coco_gen_format(gen_dict):
    return [{'caption': v, 'image_id': k} for k, v in gen_dict.items()]


def coco_gen_format_save(gen_dict, save_path):
    json.dump(coco_gen_format(gen_dict), open(save_path, 'w'))
