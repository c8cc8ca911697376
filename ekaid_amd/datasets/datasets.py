"""``create_dataset`` -- mirror of reference ``datasets/datasets.py:1-15``."""


def create_dataset(cfg, split='train', **loader_kwargs):
    """Returns ``(dataset, data_loader)`` like the reference.  ``loader_kwargs`` (extension): ``rank`` / ``world_size``
    (data-parallel slicing of every global batch), ``seed``, ``drop_last``."""
    if cfg.data.dataset == 'rcc_dataset_mimic':
        from .rcc_dataset_pos_mimic import RCCDataLoader, RCCDataset_mimic
        dataset = RCCDataset_mimic(cfg, split)
        data_loader = RCCDataLoader(dataset, batch_size=dataset.batch_size, shuffle=True if split == 'train' else False,
                                    num_workers=cfg.data.num_workers, **loader_kwargs)
    else:
        raise Exception('Unknown dataset: %s' % cfg.data.dataset)
    return dataset, data_loader
