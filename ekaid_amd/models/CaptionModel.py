"""``CaptionModel`` base class -- the ``forward(mode=...)`` dispatcher of reference
``models/CaptionModel.py:28-32``.  The diverse-beam-search half of that file (:34-182) is outside the
path (``beam_size = 1`` everywhere, config.py:83, and ``_sample_beam`` is broken in the reference:
dynamic_speaker_change_pos.py:273 unpacks two of three return values)."""
import torch.nn as nn


class CaptionModel(nn.Module):
    def __init__(self):
        super(CaptionModel, self).__init__()

    def forward(self, *args, **kwargs):
        mode = kwargs.get('mode', 'forward')
        if 'mode' in kwargs:
            del kwargs['mode']
        return getattr(self, '_' + mode)(*args, **kwargs)

    def beam_search(self, *args, **kwargs):
        raise NotImplementedError('beam search is outside the EKAID hot path (beam_size == 1)')
