"""Configuration object for the difference-VQA path.

Mirrors the *contract* of the reference's global config (reference
``model/configs/config.py:7-157`` for key names/defaults and ``:165-223`` for the
yaml merge) so that the modules of this package read exactly the keys the
reference modules read (``cfg.model.change_detector.*``, ``cfg.model.speaker.*``,
``cfg.train.{setting,graph}``, ``cfg.data.{feature_mode,train.empty_image}``).

Differences from the reference, on purpose:
  * no process-global mutable singleton is required: ``default_cfg()`` returns a
    fresh tree (a module-level ``cfg`` is still exported for drop-in scripts);
  * yaml is read with ``yaml.safe_load`` (the reference's bare ``yaml.load(f)``
    fails on PyYAML >= 6, reference ``config.py:168``).
"""
import copy
from ast import literal_eval


class AttrDict(dict):
    """dict with attribute access and ``.get`` (reference ``utils/attr_dict.py``)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def __deepcopy__(self, memo):
        return AttrDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _to_attr(d):
    out = AttrDict()
    for k, v in d.items():
        out[k] = _to_attr(v) if isinstance(v, dict) else v
    return out


def default_cfg():
    """Defaults of the reference config tree (only keys the path reads + the
    training-loop keys of ``train_mimic.py``)."""
    return _to_attr({
        'exp_dir': './experiments',
        'exp_name': '',
        'gpu_id': [0],
        'seed': 1111,
        'model': {
            'type': '',
            'change_detector': {
                'type': 'None', 'input_dim': 2048, 'dim': 128, 'feat_dim': 1024,
                'att_dim': 512, 'att_head': 8, 'nongt_dim': 52,
                'spa_label_num': 11, 'sem_label_num': 3, 'dir_num': 2,
                'pos_emb_dim': 64, 'coef_sem': 0.333, 'coef_spa': 0.333,
            },
            'speaker': {
                'type': 'attention', 'img_feat_size': 1024, 'input_dim': 1024,
                'rnn_size': 512, 'embed_input_dim': 4096, 'embed_dim': 512,
                'att_hid_size': 512, 'drop_prob_lm': 0.5, 'word_embed_size': 300,
                'rnn_num_layers': 1, 'vocab_size': 60, 'seq_length': 23,
                'decoding_constraint': 1, 'beam_size': 1, 'sample_max': 0,
                'temperature': 1.0, 'start_from': 'None',
            },
        },
        'data': {
            'dataset': 'rcc_dataset', 'num_workers': 0,
            'vocab_json': './data/vocab_mimic_VQA.json',
            'splits_json': './data/splits.json',
            'h5_label_file': './data/labels.h5',
            'h5_feature_file': '',        # extension: '' = the path hard-coded in the reference (rcc_dataset_pos_mimic.py:95)
            'feature_mode': 'location',
            'train': {'batch_size': 128, 'seq_per_img': 1, 'max_samples': None,
                      'empty_image': False},
            'val': {'batch_size': 64, 'seq_per_img': 5, 'max_samples': None},
            'test': {'batch_size': 64, 'seq_per_img': 5, 'max_samples': None},
        },
        'train': {
            'snapshot_interval': 1000, 'start_from': None, 'max_iter': 10000,
            'max_epoch': 20, 'log_interval': 50,
            'scheduled_sampling_start': 1000000,
            'scheduled_sampling_increase_every': 5,
            'scheduled_sampling_increase_prob': 0.05,
            'scheduled_sampling_max_prob': 0.25,
            'optim': {'type': 'sgdmom', 'lr': 0.01, 'alpha': 0.9, 'beta': 0.999,
                      'weight_decay': 5e-4, 'step_size': 15, 'gamma': 0.9,
                      'epsilon': 1e-08},
            'setting': 'mode2', 'graph': 'all',
        },
    })


# values of reference configs/dynamic/dynamic_change_pos_mimic.yaml
MIMIC_YAML = {
    'exp_name': 'mimic-diff',
    'model': {
        'change_detector': {'input_dim': 2052, 'feat_dim': 1026, 'att_dim': 1024,
                            'att_head': 4, 'nongt_dim': 52, 'spa_label_num': 11,
                            'sem_label_num': 3, 'dir_num': 2, 'pos_emb_dim': 64,
                            'coef_sem': 0.333, 'coef_spa': 0.333},
        'speaker': {'embed_input_dim': 3072, 'embed_dim': 1024, 'input_dim': 1024,
                    'seq_length': 90},
    },
    'data': {'dataset': 'rcc_dataset_mimic',
             'vocab_json': './data/vocab_mimic_VQA.json',
             'h5_label_file': 'data/VQA_mimic_dataset.h5',
             'splits_json': './data/splits_mimic_VQA.json', 'num_workers': 0,
             'train': {'batch_size': 64, 'empty_image': False},
             'val': {'seq_per_img': 1}, 'test': {'seq_per_img': 1}},
    'train': {'max_iter': 40000, 'max_epoch': 20, 'snapshot_interval': 2000,
              'optim': {'type': 'adam', 'lr': 0.0001, 'weight_decay': 0.0,
                        'step_size': 15, 'gamma': 0.1}},
}


def _decode(v):
    if isinstance(v, dict):
        return _to_attr(v)
    if not isinstance(v, str):
        return v
    try:
        return literal_eval(v)
    except (ValueError, SyntaxError):
        return v


def merge_into(a, b, stack=()):
    """Merge mapping ``a`` into config ``b``; unknown keys raise ``KeyError`` and
    incompatible types raise ``ValueError`` (reference ``config.py:197-284``)."""
    for k, v in a.items():
        full = '.'.join(stack + (k,))
        if k not in b:
            raise KeyError('Non-existent config key: {}'.format(full))
        v = _decode(copy.deepcopy(v))
        old = b[k]
        if isinstance(v, dict):
            if not isinstance(old, dict):
                raise ValueError('Type mismatch for config key: {}'.format(full))
            merge_into(v, old, stack + (k,))
            continue
        if old is not None and v is not None and type(v) is not type(old):
            if isinstance(old, str):
                v = str(v)
            elif isinstance(old, float) and isinstance(v, int) and not isinstance(v, bool):
                v = float(v)
            elif isinstance(old, (list, tuple)) and isinstance(v, (list, tuple)):
                v = type(old)(v)
            else:
                raise ValueError('Type mismatch ({} vs. {}) for config key: {}'.format(
                    type(old), type(v), full))
        b[k] = v
    return b


cfg = default_cfg()


def merge_cfg_from_file(path, target=None):
    import yaml
    with open(path, 'r') as f:
        y = yaml.safe_load(f)
    return merge_into(y, cfg if target is None else target)


def mimic_cfg(**overrides):
    """Config tree = defaults + the mimic yaml values (+ dotted overrides)."""
    c = default_cfg()
    merge_into(copy.deepcopy(MIMIC_YAML), c)
    for key, val in overrides.items():
        d = c
        parts = key.split('.')
        for p in parts[:-1]:
            d = d[p]
        d[parts[-1]] = val
    return c


def bench_cfg(D=2048, seq_length=90, graph='all', setting='mode2'):
    """BASELINE.json configs 2-4: yaml widths, ``feat_dim = D + 2``."""
    c = mimic_cfg()
    c.model.change_detector.feat_dim = D + 2
    c.model.change_detector.input_dim = 2 * (D + 2)
    c.model.speaker.seq_length = seq_length
    c.model.speaker.vocab_size = 148
    c.train.graph = graph
    c.train.setting = setting
    return c


def small_cfg(graph='all', setting='mode2', D=128, att_dim=64, heads=4, qdim=64,
              seq_length=6):
    """BASELINE.json config 1 / fixture F1 widths.

    ``rnn_size`` stays 512 because the reference hard-codes ``pos1 = Linear(512, 512)``
    (reference ``dynamic_speaker_change_pos.py:65-69``)."""
    c = default_cfg()
    cd = c.model.change_detector
    cd.feat_dim = D + 2
    cd.input_dim = 2 * (D + 2)
    cd.att_dim = att_dim
    cd.att_head = heads
    cd.dim = 32
    cd.nongt_dim = 52
    sp = c.model.speaker
    sp.embed_dim = qdim
    sp.input_dim = att_dim
    sp.embed_input_dim = 3 * att_dim
    sp.rnn_size = 512
    sp.word_embed_size = 24
    sp.seq_length = seq_length
    sp.vocab_size = 31
    c.train.graph = graph
    c.train.setting = setting
    c.train.optim.type = 'adam'
    c.train.optim.lr = 1e-4
    c.train.optim.weight_decay = 0.0
    return c
