#!/usr/bin/env python3
"""Write the small HDF5 fixtures under tests/golden/h5/ with the REAL HDF5 library (libhdf5 1.10.6 found in this build
container under /opt/conda/lib, driven through ctypes -- h5py itself is not installed), replaying what the reference's
writers do through h5py:

  * ``feature extraction/combine_dicts.py:162-216``  -> cmb_bbox_di_feats.hdf5
        image_features (n,52,W) f32, image_bb (n,52,4) f32, image_adj_matrix / semantic_adj_matrix (n,100,100) i64,
        bbox_label (n,52) i64 -- created with ``maxshape=(None, ...)``, ``chunks=(C, ...)``, then resized and written
        block by block, the file re-opened in append mode for every block after the first;
  * ``dataset_preparation.py:27-88``                  -> VQA_mimic_dataset.h5
        questions (n,20), answers / pos (n,90), label_start_idx / label_end_idx (n,1), feature_idx (n,2), all i64.

The fixtures use reduced widths (W = 16 instead of 1024, chunk rows 3 instead of 100) so that they stay small; the
object-header / B-tree / chunk structure is the library's own.  The expected contents are stored next to them as .npz.
Also writes ``libhdf5_readback.npz`` = what the real library reads from a file produced by ``ekaid_amd.h5lite.write``
(so the writer is pinned too).  Re-run:  python tests/golden/make_h5_fixtures.py
"""
import ctypes
import os
import sys
from ctypes import POINTER, byref, c_char_p, c_int, c_int64, c_uint, c_uint64, c_void_p

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
OUT = os.path.join(HERE, 'h5')
LIBHDF5 = os.environ.get('LIBHDF5', '/opt/conda/lib/libhdf5.so.103')

hid = c_int64
UNLIMITED = (1 << 64) - 1


class Lib:
    def __init__(self, path=LIBHDF5):
        L = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        assert L.H5open() >= 0
        self.L = L
        sig = {
            'H5Fcreate': (hid, [c_char_p, c_uint, hid, hid]), 'H5Fopen': (hid, [c_char_p, c_uint, hid]),
            'H5Fclose': (c_int, [hid]), 'H5Screate_simple': (hid, [c_int, POINTER(c_uint64), POINTER(c_uint64)]),
            'H5Sclose': (c_int, [hid]), 'H5Pcreate': (hid, [hid]), 'H5Pclose': (c_int, [hid]),
            'H5Pset_chunk': (c_int, [hid, c_int, POINTER(c_uint64)]),
            'H5Pset_obj_track_times': (c_int, [hid, c_int]),
            'H5Dcreate2': (hid, [hid, c_char_p, hid, hid, hid, hid, hid]), 'H5Dopen2': (hid, [hid, c_char_p, hid]),
            'H5Dclose': (c_int, [hid]), 'H5Dset_extent': (c_int, [hid, POINTER(c_uint64)]),
            'H5Dget_space': (hid, [hid]),
            'H5Sselect_hyperslab': (c_int, [hid, c_int, POINTER(c_uint64), c_void_p, POINTER(c_uint64), c_void_p]),
            'H5Dwrite': (c_int, [hid, hid, hid, hid, hid, c_void_p]),
            'H5Dread': (c_int, [hid, hid, hid, hid, hid, c_void_p]),
            'H5Sget_simple_extent_ndims': (c_int, [hid]),
            'H5Sget_simple_extent_dims': (c_int, [hid, POINTER(c_uint64), POINTER(c_uint64)]),
        }
        for n, (res, args) in sig.items():
            f = getattr(L, n)
            f.restype, f.argtypes = res, args
        g = lambda s: c_int64.in_dll(L, s).value      # noqa: E731
        self.DCPL = g('H5P_CLS_DATASET_CREATE_ID_g')
        self.types = {np.dtype('float32'): (g('H5T_IEEE_F32LE_g'), g('H5T_NATIVE_FLOAT_g')),
                      np.dtype('int64'): (g('H5T_STD_I64LE_g'), g('H5T_NATIVE_INT64_g')),
                      np.dtype('float64'): (g('H5T_IEEE_F64LE_g'), g('H5T_NATIVE_DOUBLE_g'))}

    @staticmethod
    def dims(seq):
        return (c_uint64 * len(seq))(*[int(s) for s in seq])

    def create_dataset(self, f, name, shape, maxshape, chunks, dtype):
        """h5py.File.create_dataset(name, shape, maxshape=..., chunks=..., dtype=...) with h5py's defaults."""
        mx = [UNLIMITED if m is None else m for m in maxshape]
        sp = self.L.H5Screate_simple(len(shape), self.dims(shape), self.dims(mx))
        pl = self.L.H5Pcreate(self.DCPL)
        assert self.L.H5Pset_chunk(pl, len(chunks), self.dims(chunks)) >= 0
        self.L.H5Pset_obj_track_times(pl, 0)                       # h5py >= 3 default (track_times=False)
        d = self.L.H5Dcreate2(f, name.encode(), self.types[np.dtype(dtype)][0], sp, 0, pl, 0)
        assert d >= 0
        self.L.H5Pclose(pl)
        self.L.H5Sclose(sp)
        return d

    def write_rows(self, d, r0, arr):
        """dset.resize([r0 + len(arr), ...]); dset[r0:r0 + len(arr)] = arr"""
        arr = np.ascontiguousarray(arr)
        new = (r0 + arr.shape[0],) + arr.shape[1:]
        assert self.L.H5Dset_extent(d, self.dims(new)) >= 0
        fs = self.L.H5Dget_space(d)
        start = (r0,) + (0,) * (arr.ndim - 1)
        assert self.L.H5Sselect_hyperslab(fs, 0, self.dims(start), None, self.dims(arr.shape), None) >= 0
        ms = self.L.H5Screate_simple(arr.ndim, self.dims(arr.shape), None)
        assert self.L.H5Dwrite(d, self.types[arr.dtype][1], ms, fs, 0, arr.ctypes.data_as(c_void_p)) >= 0
        self.L.H5Sclose(ms)
        self.L.H5Sclose(fs)

    def read_all(self, path, name, dtype):
        f = self.L.H5Fopen(path.encode(), 0, 0)
        assert f >= 0, 'H5Fopen failed'
        d = self.L.H5Dopen2(f, name.encode(), 0)
        assert d >= 0, 'H5Dopen2 failed for %s' % name
        sp = self.L.H5Dget_space(d)
        nd = self.L.H5Sget_simple_extent_ndims(sp)
        dims = (c_uint64 * nd)()
        self.L.H5Sget_simple_extent_dims(sp, dims, None)
        out = np.empty(tuple(dims), dtype)
        assert self.L.H5Dread(d, self.types[np.dtype(dtype)][1], 0, 0, 0, out.ctypes.data_as(c_void_p)) >= 0
        self.L.H5Sclose(sp)
        self.L.H5Dclose(d)
        self.L.H5Fclose(f)
        return out


def save_blocks(lib, path, specs, data, block):
    """The reference's ``save_h5(..., times, length)`` protocol: create at the first block ('w'), then re-open in
    append mode for every further block, resize and write it."""
    n = len(next(iter(data.values())))
    for times, r0 in enumerate(range(0, n, block)):
        if times == 0:
            f = lib.L.H5Fcreate(path.encode(), 2, 0, 0)             # H5F_ACC_TRUNC
            ds = {k: lib.create_dataset(f, k, (block,) + tail, (None,) + tail, (chunk,) + tail, dt)
                  for k, (tail, chunk, dt) in specs.items()}
        else:
            f = lib.L.H5Fopen(path.encode(), 1, 0)                  # H5F_ACC_RDWR
            ds = {k: lib.L.H5Dopen2(f, k.encode(), 0) for k in specs}
        for k in specs:
            lib.write_rows(ds[k], r0, data[k][r0:r0 + block])
            lib.L.H5Dclose(ds[k])
        lib.L.H5Fclose(f)


def feature_file_data(n, nodes=52, width=16, seed=7):
    rng = np.random.default_rng(seed)
    feats = np.maximum(rng.standard_normal((n, nodes, width)), 0).astype(np.float32)
    feats[:, 5] = 0.0                                               # undetected region: zero-filled (ana_bbox_generator.py:591-594)
    x1 = rng.uniform(0, 700, (n, nodes)); y1 = rng.uniform(0, 700, (n, nodes))
    bb = np.stack([x1, y1, x1 + rng.uniform(32, 320, (n, nodes)), y1 + rng.uniform(32, 320, (n, nodes))], -1).astype(np.float32)
    adj = np.zeros((n, 100, 100), np.int64)
    adj[:, :nodes, :nodes] = rng.integers(0, 12, (n, nodes, nodes)) * (rng.random((n, nodes, nodes)) < 0.6)
    sem = np.zeros((n, 100, 100), np.int64)
    s = rng.choice(3, size=(n, nodes, nodes), p=[0.7, 0.2, 0.1])
    sem[:, :nodes, :nodes] = np.maximum(s, s.transpose(0, 2, 1))
    lab = rng.integers(0, 40, (n, nodes)).astype(np.int64)
    return {'image_features': feats, 'image_bb': bb, 'image_adj_matrix': adj, 'semantic_adj_matrix': sem,
            'bbox_label': lab}


def label_file_data(n, n_images, V=148, seed=11):
    rng = np.random.default_rng(seed)
    q = np.zeros((n, 20), np.int64)
    a = np.zeros((n, 90), np.int64)
    pos = np.zeros((n, 90), np.int64)
    for i in range(n):
        ql = int(rng.integers(3, 13))
        q[i, :ql] = rng.integers(2, V, ql)
        al = int(rng.integers(2, 30 if i % 5 else 90))
        a[i, 0] = 1                                                 # '<start>' (dataset_preparation.py:170)
        a[i, 1:al] = rng.integers(2, V, al - 1)
        pos[i, :al] = rng.integers(1, 16, al)
    start = np.arange(n, dtype=np.int64).reshape(n, 1)
    fidx = rng.integers(0, n_images, (n, 2)).astype(np.int64)
    return {'questions': q, 'answers': a, 'pos': pos, 'label_start_idx': start, 'label_end_idx': start + 1,
            'feature_idx': fidx}


def main():
    os.makedirs(OUT, exist_ok=True)
    lib = Lib()
    n_img, n_q, W, C = 8, 23, 16, 3
    feats = feature_file_data(n_img, width=W)
    specs = {'image_features': ((52, W), C, 'float32'), 'image_bb': ((52, 4), C, 'float32'),
             'image_adj_matrix': ((100, 100), C, 'int64'), 'semantic_adj_matrix': ((100, 100), C, 'int64'),
             'bbox_label': ((52,), C, 'int64')}
    save_blocks(lib, os.path.join(OUT, 'cmb_bbox_di_feats.hdf5'), specs, feats, block=C)
    labels = label_file_data(n_q, n_img)
    lspecs = {'questions': ((20,), 10, 'int64'), 'answers': ((90,), 10, 'int64'), 'pos': ((90,), 10, 'int64'),
              'label_start_idx': ((1,), 10, 'int64'), 'label_end_idx': ((1,), 10, 'int64'),
              'feature_idx': ((2,), 10, 'int64')}
    save_blocks(lib, os.path.join(OUT, 'VQA_mimic_dataset.h5'), lspecs, labels, block=10)
    np.savez_compressed(os.path.join(OUT, 'expected.npz'), **feats, **labels)
    # a many-chunk dataset: the chunk B-tree grows a second level (> 64 chunks)
    big = {'v': np.arange(150 * 6, dtype=np.int64).reshape(150, 6)}
    save_blocks(lib, os.path.join(OUT, 'deep_btree.h5'), {'v': ((6,), 1, 'int64')}, big, block=50)
    # writer pin: h5lite.write -> read back with the real library
    from ekaid_amd import h5lite
    wpath = os.path.join(OUT, '_h5lite_written.h5')
    src = {'image_features': (feats['image_features'], 3), 'image_adj_matrix': (feats['image_adj_matrix'], 3),
           'questions': (labels['questions'], 10), 'v': (big['v'], 1), 'f64': (np.linspace(0, 1, 35).reshape(7, 5), 2)}
    h5lite.write(wpath, src)
    back = {k: lib.read_all(wpath, k, v[0].dtype) for k, v in src.items()}
    for k, v in src.items():
        assert np.array_equal(back[k], v[0]), 'libhdf5 reads %s differently from what h5lite.write was given' % k
    os.remove(wpath)
    print('fixtures written to', OUT, {k: os.path.getsize(os.path.join(OUT, k)) for k in sorted(os.listdir(OUT))})
    print('h5lite.write output read back bit-exactly by libhdf5 %s' % LIBHDF5)


if __name__ == '__main__':
    main()
