#!/usr/bin/env python3
"""Golden vectors for the real-data path: import the REFERENCE's ``RCCDataset_mimic`` /
``rcc_collate`` (``/root/reference/model/datasets/rcc_dataset_pos_mimic.py``) in the build container and record what
they return, for every ``feature_mode`` and for train / test splits.

h5py is not installed here, so the reference class runs over a numpy-backed stand-in for ``h5py.File`` (a dict of
arrays with ``__getitem__`` / ``close``): the arrays are the ones stored in tests/golden/h5/expected.npz, i.e. exactly
the content of the library-written fixture files the product reader is tested on.  No reference source is stored;
the outputs are data.   Re-run:  python tests/golden/make_dataset_golden.py
"""
import json
import os
import random
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/model'
SPLITS = {'train': [3, 0, 7, 12, 21, 9, 16, 5, 1, 19, 22, 14], 'val': [2, 4, 6], 'test': [8, 10, 11, 13, 15, 17, 18, 20]}
VOCAB = {'w%d' % i: i for i in range(1, 148)}


class FakeDataset:
    def __init__(self, a):
        self.a = a
        self.shape = a.shape

    def __getitem__(self, k):
        return self.a[k]

    def __len__(self):
        return len(self.a)


class FakeFile:
    arrays = None

    def __init__(self, path, mode='r'):
        self.path = path

    def __getitem__(self, name):
        return FakeDataset(FakeFile.arrays[name])

    def close(self):
        pass


def main():
    exp = np.load(os.path.join(HERE, 'h5', 'expected.npz'))
    FakeFile.arrays = {k: exp[k] for k in exp.files}
    h5py = types.ModuleType('h5py')
    h5py.File = FakeFile
    sys.modules['h5py'] = h5py
    sys.path.insert(0, REF)
    for m in [k for k in sys.modules if k == 'datasets' or k.startswith('datasets.')]:
        del sys.modules[m]
    from datasets.rcc_dataset_pos_mimic import RCCDataset_mimic, rcc_collate       # the reference
    from utils.attr_dict import AttrDict

    tmp = tempfile.mkdtemp()
    json.dump(VOCAB, open(os.path.join(tmp, 'vocab.json'), 'w'))
    json.dump(SPLITS, open(os.path.join(tmp, 'splits.json'), 'w'))

    def make_cfg(mode, seq_per_img):
        cfg = AttrDict()
        cfg.data = AttrDict()
        cfg.data.vocab_json = os.path.join(tmp, 'vocab.json')
        cfg.data.splits_json = os.path.join(tmp, 'splits.json')
        cfg.data.h5_label_file = 'VQA_mimic_dataset.h5'
        cfg.data.feature_mode = mode
        for s, bs in (('train', 5), ('val', 3), ('test', 4)):
            cfg.data[s] = AttrDict()
            cfg.data[s].batch_size = bs
            cfg.data[s].seq_per_img = seq_per_img
            cfg.data[s].max_samples = None
        return cfg

    out = {}
    for mode in ('both', 'location', 'single_ana', 'single_loc'):
        for split in ('train', 'test'):
            for spi in ((1, 3) if (mode == 'both' and split == 'test') else (1,)):
                ds = RCCDataset_mimic(make_cfg(mode, spi), split)
                idx = list(range(len(ds)))[:6]
                random.seed(999)
                samples = [ds[i] for i in idx]
                tag = '%s_%s_spi%d' % (mode, split, spi)
                out[tag + '_state_after'] = np.array([random.random()])     # the ``random`` stream position is part of the behaviour
                batch = rcc_collate(samples)
                out[tag + '_indices'] = np.array(idx)
                for k, v in enumerate(batch):
                    out['%s_b%d' % (tag, k)] = v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
                s0 = samples[2]
                for k, v in enumerate(s0):
                    a = v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
                    out['%s_s%d' % (tag, k)] = a
                out[tag + '_len'] = np.array([len(ds), ds.get_vocab_size(), ds.get_max_seq_length(), ds.batch_size])
    # loader order: the reference's RCCDataLoader(shuffle=True) under a fixed torch seed
    from datasets.rcc_dataset_pos_mimic import RCCDataLoader
    ds = RCCDataset_mimic(make_cfg('both', 1), 'train')
    torch.manual_seed(4321)
    loader = RCCDataLoader(ds, batch_size=5, shuffle=True, num_workers=0)
    order = [b[5].numpy() for b in loader]
    order2 = [b[5].numpy() for b in loader]                       # second epoch continues the global RNG stream
    out['loader_pair_index_epoch0'] = np.concatenate(order)
    out['loader_pair_index_epoch1'] = np.concatenate(order2)
    out['loader_batch_sizes'] = np.array([len(o) for o in order])
    np.savez_compressed(os.path.join(HERE, 'dataset_golden.npz'), **out)
    json.dump({'splits': SPLITS, 'n_vocab': len(VOCAB)}, open(os.path.join(HERE, 'dataset_golden_meta.json'), 'w'))
    print('wrote', os.path.join(HERE, 'dataset_golden.npz'), len(out), 'arrays')


if __name__ == '__main__':
    main()
