"""Real-data path on CPU: the HDF5 reader/writer against files written by the real library,
and the dataset / collate / loader counterparts against vectors recorded from the imported reference."""
import json
import os
import random

import numpy as np
import pytest
import torch

from ekaid_amd import h5lite
from ekaid_amd.config import mimic_cfg
from ekaid_amd.datasets import RCCDataLoader, RCCDataset_mimic, rcc_collate
from ekaid_amd.datasets.rcc_dataset_pos_mimic import batch_dict, compact_batch

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
H5 = os.path.join(G, 'h5')
EXP = np.load(os.path.join(H5, 'expected.npz'))
GOLD = np.load(os.path.join(G, 'dataset_golden.npz'))
META = json.load(open(os.path.join(G, 'dataset_golden_meta.json')))


# ------------------------------------------------------------------------------------------------ HDF5 reader
@pytest.mark.parametrize('fname', ['cmb_bbox_di_feats.hdf5', 'VQA_mimic_dataset.h5'])
def test_reader_matches_library_written_files(fname):
    """Files created by libhdf5 1.10.6 exactly the way the reference's writers drive h5py (create resizable chunked
    datasets, re-open in append mode, resize + write block by block): every dataset reads back bit-exactly."""
    f = h5lite.File(os.path.join(H5, fname))
    assert len(f.keys()) in (5, 6)
    for k in f.keys():
        d = f[k]
        e = EXP[k]
        assert d.shape == e.shape and d.dtype == e.dtype and d.maxshape[0] is None and d.chunks[1:] == e.shape[1:]
        assert np.array_equal(d[:], e)
        assert np.array_equal(d[3], e[3]) and np.array_equal(d[-1], e[-1])
        assert np.array_equal(d[1:7:2], e[1:7:2]) and np.array_equal(d[2:5], e[2:5])
        rows = [7, 0, 3, 3, 5]
        assert np.array_equal(d.read_rows(rows), e[rows])
        assert len(d) == len(e)
        with pytest.raises(IndexError):
            d[len(e)]
    with pytest.raises(KeyError):
        f['no_such_dataset']


def test_reader_two_level_chunk_btree():
    v = h5lite.File(os.path.join(H5, 'deep_btree.h5'))['v']          # 150 one-row chunks: the index has internal nodes
    assert v.chunks == (1, 6) and len(v._chunk_index()) == 150
    assert np.array_equal(v[:], np.arange(900).reshape(150, 6))
    assert np.array_equal(v.read_rows([149, 0, 64, 65]), np.arange(900).reshape(150, 6)[[149, 0, 64, 65]])


def test_reader_refuses_what_it_does_not_implement(tmp_path):
    p = tmp_path / 'x.h5'
    p.write_bytes(b'not an hdf5 file' * 10)
    with pytest.raises(h5lite.H5Error):
        h5lite.File(str(p))
    with pytest.raises(h5lite.H5Error):
        h5lite.File(os.path.join(H5, 'VQA_mimic_dataset.h5'), 'w')


def test_writer_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    src = {'a': (rng.standard_normal((230, 7, 3)).astype(np.float32), 100), 'b': (rng.integers(-5, 5, (9, 4)), 2),
           'c': (np.zeros((0, 5), np.int64), 4), 'z_many': (np.arange(700, dtype=np.int64).reshape(140, 5), 1),
           'f64': (rng.standard_normal((5, 2)), 100)}
    path = h5lite.write(str(tmp_path / 'w.h5'), src)
    f = h5lite.File(path)
    assert f.keys() == sorted(src)
    for k, (a, c) in src.items():
        assert f[k].shape == a.shape and f[k].dtype == a.dtype and np.array_equal(f[k][:], a), k
        assert f[k].chunks == (c,) + a.shape[1:]


def test_writer_output_is_read_by_the_real_library(tmp_path):
    """h5lite.write -> libhdf5 (when this machine has one; the build container does): bit-exact."""
    import sys
    sys.path.insert(0, G)
    import make_h5_fixtures as M
    if not os.path.exists(M.LIBHDF5):
        pytest.skip('no libhdf5 on this machine')
    lib = M.Lib()
    rng = np.random.default_rng(1)
    src = {'image_features': (rng.standard_normal((11, 52, 8)).astype(np.float32), 4),
           'image_adj_matrix': (rng.integers(0, 12, (11, 100, 100)), 4), 'v': (np.arange(300).reshape(100, 3), 1)}
    path = h5lite.write(str(tmp_path / 'w.h5'), src)
    for k, (a, _) in src.items():
        assert np.array_equal(lib.read_all(path, k, a.dtype), a), k


# ------------------------------------------------------------------------------------------------ dataset
def make_cfg(tmp_path, mode, spi):
    json.dump({'w%d' % i: i for i in range(1, META['n_vocab'] + 1)}, open(tmp_path / 'vocab.json', 'w'))
    json.dump(META['splits'], open(tmp_path / 'splits.json', 'w'))
    c = mimic_cfg()
    c.data.vocab_json, c.data.splits_json = str(tmp_path / 'vocab.json'), str(tmp_path / 'splits.json')
    c.data.h5_label_file = os.path.join(H5, 'VQA_mimic_dataset.h5')
    c.data.h5_feature_file = os.path.join(H5, 'cmb_bbox_di_feats.hdf5')
    c.data.feature_mode = mode
    for s, bs in (('train', 5), ('val', 3), ('test', 4)):
        c.data[s].batch_size, c.data[s].seq_per_img, c.data[s].max_samples = bs, spi, None
    return c


CASES = [(m, s, 1) for m in ('both', 'location', 'single_ana', 'single_loc') for s in ('train', 'test')] + [('both', 'test', 3)]


def same(a, b):
    a = a.numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a, b)


@pytest.mark.parametrize('mode,split,spi', CASES)
def test_dataset_matches_the_reference(tmp_path, mode, split, spi, capsys):
    """Samples, the collated batch, the batched fast path and the position of the global ``random`` stream, all
    against what the imported reference class returned over the same file contents."""
    ds = RCCDataset_mimic(make_cfg(tmp_path, mode, spi), split)
    tag = '%s_%s_spi%d' % (mode, split, spi)
    idx = GOLD[tag + '_indices'].tolist()
    assert [len(ds), ds.get_vocab_size(), ds.get_max_seq_length(), ds.batch_size] == GOLD[tag + '_len'].tolist()
    random.seed(999)
    samples = [ds[i] for i in idx]
    assert random.random() == float(GOLD[tag + '_state_after'][0])
    for k in range(13):
        assert same(samples[2][k], GOLD['%s_s%d' % (tag, k)]), 'sample field %d' % k
    batch = rcc_collate(samples)
    for k in range(13):
        assert same(batch[k], GOLD['%s_b%d' % (tag, k)]), 'collated field %d' % k
    random.seed(999)
    fast = ds[idx]                                     # one call, one gather per HDF5 chunk
    assert random.random() == float(GOLD[tag + '_state_after'][0])
    for k in range(13):
        assert same(fast[k], GOLD['%s_b%d' % (tag, k)]), 'batched field %d' % k
        assert fast[k].is_contiguous()


def test_loader_order_is_the_reference_loaders(tmp_path):
    ds = RCCDataset_mimic(make_cfg(tmp_path, 'both', 1), 'train')
    for per_sample in (False, True):
        torch.manual_seed(4321)
        loader = RCCDataLoader(ds, batch_size=5, shuffle=True, num_workers=0, per_sample=per_sample)
        e0 = [b[5].numpy() for b in loader]
        e1 = [b[5].numpy() for b in loader]
        assert [len(x) for x in e0] == GOLD['loader_batch_sizes'].tolist()
        assert np.array_equal(np.concatenate(e0), GOLD['loader_pair_index_epoch0'])
        assert np.array_equal(np.concatenate(e1), GOLD['loader_pair_index_epoch1'])


def test_loader_rank_slices_partition_the_global_batch(tmp_path):
    ds = RCCDataset_mimic(make_cfg(tmp_path, 'both', 1), 'train')
    torch.manual_seed(7)
    whole = [b for b in RCCDataLoader(ds, batch_size=4, shuffle=True)]
    parts = []
    for r in range(2):
        torch.manual_seed(7)
        parts.append([b for b in RCCDataLoader(ds, batch_size=4, shuffle=True, rank=r, world_size=2)])
    assert len(parts[0]) == len(parts[1]) == len(whole) == 3
    for g, a, b in zip(whole, parts[0], parts[1]):
        for k in range(13):
            assert torch.equal(torch.cat([a[k], b[k]], 0), g[k])
    # the loss denominator of a global batch is known to every rank from the in-memory labels
    torch.manual_seed(7)
    ld = RCCDataLoader(ds, batch_size=4, shuffle=True, rank=1, world_size=2)
    it = iter(ld)
    local = next(it)
    gidx = ld.last_global_indices()
    assert len(gidx) == 4 and ds.mask_sum(gidx) == float(whole[0][4].squeeze(1)[:, 1:].sum())
    assert ds.mask_sum(gidx[2:4]) == float(local[4].squeeze(1)[:, 1:].sum())


def test_compact_batch_is_lossless(tmp_path):
    ds = RCCDataset_mimic(make_cfg(tmp_path, 'single_ana', 1), 'test')
    b = ds[[0, 1, 2, 3]]
    d, c = batch_dict(b), compact_batch(b)
    N = d['input_1'].shape[1]
    assert N == 26 and c['d_adj'].dtype == torch.int8 and c['d_adj'].shape == (4, N, N)
    for k in ('d_adj', 'q_adj', 'd_sem', 'q_sem'):
        assert torch.equal(c[k].double(), d[k][:, :N, :N])
    assert c['masks'].dtype == torch.float32 and c['labels'].shape == (4, 91) and c['d_bb'].dtype == torch.float64
    nbytes = lambda t: sum(v.numel() * v.element_size() for v in t.values())      # noqa: E731
    assert nbytes(c) < 0.5 * nbytes(d)


def test_an_evaluation_pass_on_one_rank_does_not_desynchronise_the_shuffle(tmp_path):
    """Only rank 0 evaluates inside the training loop (train_mimic.py).  Building a loader iterator draws a base seed from torch's
    default generator, the generator every rank shuffles its next epoch from: train_mimic.evaluate() therefore saves and restores
    the generator state around its pass.  Pinned here on the loaders alone: the hazard exists (the state moves), and with the
    guard both ranks still slice the SAME global permutation in the next epoch."""
    ds = RCCDataset_mimic(make_cfg(tmp_path, 'both', 1), 'train')
    val = RCCDataset_mimic(make_cfg(tmp_path, 'both', 1), 'test')
    orders = []
    for rank in range(2):
        torch.manual_seed(99)
        ld = RCCDataLoader(ds, batch_size=4, shuffle=True, rank=rank, world_size=2)
        first = [b.global_indices for b in ld]                     # epoch 0
        if rank == 0:                                              # the in-loop evaluation, rank 0 only
            before = torch.get_rng_state()
            for _ in RCCDataLoader(val, batch_size=4, shuffle=False):
                break
            assert not torch.equal(before, torch.get_rng_state()), 'the hazard this test is about is gone'
            torch.set_rng_state(before)                            # what evaluate() does in its finally
        orders.append((first, [b.global_indices for b in ld]))     # epoch 1
    assert orders[0][0] == orders[1][0] and orders[0][1] == orders[1][1]
    assert orders[0][0] != orders[0][1]
