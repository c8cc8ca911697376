"""Real-data path of the difference-VQA model: counterparts of the reference's
``model/datasets/{datasets.py,rcc_dataset_pos_mimic.py}`` on top of ``ekaid_amd.h5lite`` (no h5py needed)."""
from .datasets import create_dataset                     # noqa: F401
from .rcc_dataset_pos_mimic import RCCDataLoader, RCCDataset_mimic, rcc_collate   # noqa: F401
