"""Loss / optimiser / checkpoint helpers -- counterparts of reference ``utils/utils.py`` (only the
functions the entry points use: :11-23, :28-43, :70-75, :82-101, :122-136, :171-179, :204-216).
``load_cpu`` / ``save_checkpoint`` / ``load_checkpoint`` / ``AverageMeter`` / ``set_mode`` keep the reference's names, print
strings and semantics (:11-43, :70-75) because the entry scripts and users' tooling call them by name.

Checkpoint interop: the reference pickles ``model_cfg`` as an instance of ITS ``utils.attr_dict.AttrDict``
(configs/config.py:5, train_mimic.py:283-287).  Loading here maps that class onto ``ekaid_amd.config.AttrDict``;
saving here writes the config under the reference's class path, so a checkpoint written by either side loads on the other.
"""
import contextlib
import json
import pickle
import sys
import types

import torch
import torch.nn as nn

from .. import ops
from ..config import AttrDict

_REF_MODULE, _REF_CLASS = 'utils.attr_dict', 'AttrDict'


class _CompatUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module == _REF_MODULE and name == _REF_CLASS:
            return AttrDict
        return super().find_class(module, name)


_compat_pickle = types.ModuleType('ekaid_amd.utils._compat_pickle')
_compat_pickle.__dict__.update({k: getattr(pickle, k) for k in dir(pickle) if not k.startswith('__')})
_compat_pickle.Unpickler = _CompatUnpickler
_compat_pickle.load = lambda f, **kw: _CompatUnpickler(f, **kw).load()


def load_cpu(path):
    """Loads a torch checkpoint, remapping all tensors to CPU (reference :11-15)."""
    return torch.load(path, map_location=lambda storage, loc: storage, weights_only=False, pickle_module=_compat_pickle)


@contextlib.contextmanager
def _reference_attrdict():
    """Yields a class that pickles as ``utils.attr_dict.AttrDict``: the reference's own when its tree is importable,
    otherwise a stand-in registered under that module path for the duration of the save."""
    try:
        from utils.attr_dict import AttrDict as ref_cls          # running inside the reference's model/ directory
        yield ref_cls
        return
    except Exception:
        pass
    had = {k: sys.modules.get(k) for k in ('utils', _REF_MODULE)}
    pkg = had['utils'] or types.ModuleType('utils')
    mod = types.ModuleType(_REF_MODULE)
    shim = type(_REF_CLASS, (dict,), {'__module__': _REF_MODULE})
    mod.AttrDict = shim
    sys.modules['utils'], sys.modules[_REF_MODULE] = pkg, mod
    had_attr = getattr(pkg, 'attr_dict', None)
    pkg.attr_dict = mod
    try:
        yield shim
    finally:
        for k, v in had.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        if had_attr is None:
            with contextlib.suppress(AttributeError):
                del pkg.attr_dict
        else:
            pkg.attr_dict = had_attr


def _as(cls, tree):
    if isinstance(tree, dict):
        out = cls()
        for k, v in tree.items():
            dict.__setitem__(out, k, _as(cls, v))
        return out
    return tree


def save_checkpoint(checkpoint, filename):
    print('Saving checkpoint to %s' % filename)
    if isinstance(checkpoint.get('model_cfg'), dict):
        with _reference_attrdict() as ref_cls:
            torch.save(dict(checkpoint, model_cfg=_as(ref_cls, checkpoint['model_cfg'])), filename)
        return
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

    def forward(self, input, target, mask, emphasize_last=False, denom=None):
        """``denom`` (extension): device scalar replacing ``sum(mask)`` -- data-parallel loss weighting (ekaid_amd/dp.py)."""
        T = input.size(1)
        target = target[:, :T]
        mask = mask[:, :T]
        if mask.dtype != torch.float32:
            mask = mask.float()
        return ops.lm_loss(input, target, mask, denom)


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
