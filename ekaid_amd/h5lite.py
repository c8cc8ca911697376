"""Minimal HDF5 reader / writer for the two data files of the EKAID path (no h5py / libhdf5 needed at run time).

What the reference reads (``datasets/rcc_dataset_pos_mimic.py:76-108``) is what ``h5py`` writes with its default
("earliest") format bounds from ``feature extraction/combine_dicts.py:162-216`` and ``dataset_preparation.py:27-88``:
superblock version 0, old-style groups (symbol table message -> v1 B-tree + local heap + SNOD nodes), version-1
object headers, and numeric datasets that are chunked + resizable along axis 0 (data layout message version 3,
class 2, chunk index = v1 B-tree of node type 1) with no filters.  This module implements exactly that subset of
the HDF5 File Format Specification 1.x/2.0 (plus contiguous and compact layouts and version-2 superblocks/object
headers with compact link messages, which cost a few lines) and refuses everything else loudly.

    f = File(path)                 # h5py-like, read-only
    f['image_features'].shape      # (n, 52, 1024)
    f['answers'][:]                # numpy array
    f['image_features'][17]        # one row
    f['image_adj_matrix'].read_rows(idx)   # gather rows by index (one pass per touched chunk)

    write(path, {'questions': (array, 100), ...})   # same on-disk format (chunk rows per dataset), for generated data

Pinned against files written by the real library (tests/golden/h5/*.h5, produced by tests/golden/make_h5_fixtures.py
through libhdf5 1.10.6) and, for the writer, by reading its output back through that library.
"""
import struct

import numpy as np

SIGNATURE = b'\x89HDF\r\n\x1a\n'
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Error(RuntimeError):
    pass


class _Buf:
    """Random-access little-endian decoder over the memory-mapped file."""

    def __init__(self, mm, so=8, sl=8, base=0):
        self.mm, self.so, self.sl, self.base = mm, so, sl, base

    def u(self, off, n):
        return int.from_bytes(bytes(self.mm[off:off + n]), 'little')

    def addr(self, off):
        v = self.u(off, self.so)
        return None if v == (1 << (8 * self.so)) - 1 else v + self.base

    def length(self, off):
        return self.u(off, self.sl)

    def raw(self, off, n):
        return bytes(self.mm[off:off + n])


# ----------------------------------------------------------------------------------------------- reading
class Dataset:
    def __init__(self, f, name, shape, maxshape, dtype, layout):
        self._f, self.name = f, name
        self.shape, self.maxshape, self.dtype = tuple(shape), maxshape, dtype
        self._layout = layout                    # ('contiguous', addr, size) | ('compact', bytes) | ('chunked', btree, chunk_dims)
        self._index = None

    def __len__(self):
        if not self.shape:
            raise TypeError('len() of a scalar dataset')
        return self.shape[0]

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def chunks(self):
        return tuple(self._layout[2]) if self._layout[0] == 'chunked' else None

    # -- chunk index -------------------------------------------------------------------------------
    def _chunk_index(self):
        if self._index is None:
            self._index = {}
            kind, root, cdims = self._layout
            if root is not None:
                self._walk(root, len(cdims))
        return self._index

    def _walk(self, addr, rank):
        b = self._f._b
        if b.raw(addr, 4) != b'TREE':
            raise H5Error('%s: bad B-tree node signature at %d' % (self.name, addr))
        ntype, level, used = b.u(addr + 4, 1), b.u(addr + 5, 1), b.u(addr + 6, 2)
        if ntype != 1:
            raise H5Error('%s: chunk index is not a raw-data B-tree (node type %d)' % (self.name, ntype))
        p = addr + 8 + 2 * b.so
        ksz = 8 + 8 * (rank + 1)
        cd = self._layout[2]
        for _ in range(used):
            nbytes, mask = b.u(p, 4), b.u(p + 4, 4)
            offs = tuple(b.u(p + 8 + 8 * d, 8) for d in range(rank))
            child = b.addr(p + ksz)
            p += ksz + b.so
            if level > 0:
                self._walk(child, rank)
            else:
                if mask != 0 or nbytes != int(np.prod(cd)) * self.dtype.itemsize:
                    raise H5Error('%s: filtered / partial chunks are not supported (mask %#x, %d bytes)'
                                  % (self.name, mask, nbytes))
                self._index[tuple(o // c for o, c in zip(offs, cd))] = child

    def _chunk_array(self, coord):
        """Read-only view of one stored chunk (always the full chunk shape), or None if never written."""
        a = self._chunk_index().get(coord)
        if a is None:
            return None
        cd = self._layout[2]
        n = int(np.prod(cd))
        return np.frombuffer(self._f._mm, dtype=self.dtype, count=n, offset=a).reshape(cd)

    # -- whole / row reads -------------------------------------------------------------------------
    def _read_rows_range(self, r0, r1):
        """Rows [r0, r1) of axis 0 (all other axes whole) as a fresh array."""
        shape = (r1 - r0,) + self.shape[1:]
        kind = self._layout[0]
        if kind == 'compact':
            full = np.frombuffer(self._layout[1], dtype=self.dtype).reshape(self.shape)
            return np.array(full[r0:r1])
        if kind == 'contiguous':
            addr = self._layout[1]
            if addr is None:
                return np.zeros(shape, self.dtype)
            row = int(np.prod(self.shape[1:], dtype=np.int64)) if len(self.shape) > 1 else 1
            n = (r1 - r0) * row
            v = np.frombuffer(self._f._mm, dtype=self.dtype, count=n, offset=addr + r0 * row * self.dtype.itemsize)
            return np.array(v).reshape(shape)
        cd = self._layout[2]
        out = np.zeros(shape, self.dtype)              # chunks never written read as the fill value (0)
        if r1 <= r0:
            return out
        grid = [range(r0 // cd[0], (r1 - 1) // cd[0] + 1)] + \
               [range((s + c - 1) // c) for s, c in zip(self.shape[1:], cd[1:])]
        for coord in np.ndindex(*[len(g) for g in grid]):
            cc = tuple(g[i] for g, i in zip(grid, coord))
            ch = self._chunk_array(cc)
            if ch is None:
                continue
            src, dst = [], []
            for d, (c, s) in enumerate(zip(cd, self.shape)):
                lo = cc[d] * c
                a0 = max(lo, r0) if d == 0 else lo
                a1 = min(lo + c, r1 if d == 0 else s)
                src.append(slice(a0 - lo, a1 - lo))
                dst.append(slice(a0 - (r0 if d == 0 else 0), a1 - (r0 if d == 0 else 0)))
            out[tuple(dst)] = ch[tuple(src)]
        return out

    def read_rows(self, rows, out=None):
        """Gather ``rows`` (1-D integer array, any order, repeats allowed) along axis 0 -> (len(rows),) + shape[1:]."""
        rows = np.asarray(rows, dtype=np.int64).reshape(-1)
        if rows.size and (rows.min() < 0 or rows.max() >= self.shape[0]):
            raise IndexError('%s: row index out of range' % self.name)
        if out is None:
            out = np.empty((rows.size,) + self.shape[1:], self.dtype)
        whole_rows = self._layout[0] == 'chunked' and tuple(self._layout[2][1:]) == tuple(self.shape[1:])
        if whole_rows:
            c0 = self._layout[2][0]
            ck = rows // c0
            for c in np.unique(ck):
                sel = np.nonzero(ck == c)[0]
                ch = self._chunk_array((int(c),) + (0,) * (len(self.shape) - 1))
                if ch is None:
                    out[sel] = 0
                else:
                    out[sel] = ch[rows[sel] - c * c0]
        else:
            for i, r in enumerate(rows):
                out[i] = self._read_rows_range(int(r), int(r) + 1)[0]
        return out

    def __getitem__(self, key):
        if not isinstance(key, tuple):
            key = (key,)
        if key and key[0] is Ellipsis:
            key = (slice(None),) + key[1:]
        if not self.shape:
            return self._read_rows_range(0, 1).reshape(())[()] if not key else self._read_rows_range(0, 1).reshape(())[key]
        k0, rest = (key[0] if key else slice(None)), key[1:]
        n = self.shape[0]
        if isinstance(k0, (int, np.integer)):
            i = int(k0) + (n if k0 < 0 else 0)
            if not 0 <= i < n:
                raise IndexError('%s: index %d out of range for axis 0 with size %d' % (self.name, k0, n))
            row = self._read_rows_range(i, i + 1)[0]
            return row[rest] if rest else row
        if isinstance(k0, slice):
            start, stop, step = k0.indices(n)
            if step == 1:
                arr = self._read_rows_range(start, max(stop, start))
            else:
                arr = self.read_rows(np.arange(start, stop, step))
            return arr[(slice(None),) + rest] if rest else arr
        arr = self.read_rows(np.asarray(k0))
        return arr[(slice(None),) + rest] if rest else arr

    def __array__(self, dtype=None, copy=None):
        a = self[:]
        return a if dtype is None else a.astype(dtype)


class File:
    """Read-only HDF5 file: ``f[name]`` -> Dataset (nested groups by 'a/b' paths), ``keys()``, ``close()``."""

    def __init__(self, path, mode='r'):
        if mode != 'r':
            raise H5Error('h5lite.File is read-only (use h5lite.write to create files)')
        self.filename = path
        self._mm = np.memmap(path, dtype=np.uint8, mode='r')
        mm = self._mm
        off = 0
        while True:
            if off + 8 > mm.size:
                raise H5Error('%s: not an HDF5 file (no superblock signature)' % path)
            if bytes(mm[off:off + 8]) == SIGNATURE:
                break
            off = 512 if off == 0 else off * 2
        ver = int(mm[off + 8])
        if ver in (0, 1):
            so, sl = int(mm[off + 13]), int(mm[off + 14])
            p = off + 24 + (4 if ver == 1 else 0)
            b = _Buf(mm, so, sl)
            base = b.u(p, so)
            b.base = base
            p += 4 * so                       # base, free-space info, end of file, driver info
            self._b = b
            root_oh = b.addr(p + so)          # symbol table entry: link name offset, object header address
        elif ver in (2, 3):
            so, sl = int(mm[off + 9]), int(mm[off + 10])
            b = _Buf(mm, so, sl)
            b.base = b.u(off + 12, so)
            self._b = b
            root_oh = b.addr(off + 12 + 3 * so)
        else:
            raise H5Error('%s: unsupported superblock version %d' % (path, ver))
        self._root = root_oh
        self._links = {}          # group object-header address -> {name: object header address}
        self._objs = {}

    # -- object headers -----------------------------------------------------------------------------
    def _messages(self, addr):
        """[(type, flags, data_offset, size)] of the object header at ``addr`` (continuations followed)."""
        b = self._b
        out = []
        if b.raw(addr, 4) == b'OHDR':
            if b.u(addr + 4, 1) != 2:
                raise H5Error('unsupported object header version')
            flags = b.u(addr + 5, 1)
            p = addr + 6
            if flags & 0x20:
                p += 16
            if flags & 0x10:
                p += 4
            szb = 1 << (flags & 3)
            csize = b.u(p, szb)
            p += szb
            blocks = [(p, p + csize)]
            while blocks:
                p, end = blocks.pop(0)
                while p + 4 <= end:
                    mtype, msize, mflags = b.u(p, 1), b.u(p + 1, 2), b.u(p + 3, 1)
                    p += 4 + (2 if flags & 0x04 else 0)
                    if mtype == 0x10:
                        ca, cl = b.addr(p), b.length(p + b.so)
                        if b.raw(ca, 4) != b'OCHK':
                            raise H5Error('bad object header continuation block')
                        blocks.append((ca + 4, ca + cl - 4))
                    elif mtype != 0:
                        out.append((mtype, mflags, p, msize))
                    p += msize
            return out
        if b.u(addr, 1) != 1:
            raise H5Error('unsupported object header version %d at %d' % (b.u(addr, 1), addr))
        nmsg, hsize = b.u(addr + 2, 2), b.u(addr + 8, 4)
        blocks = [(addr + 16, addr + 16 + hsize)]
        while blocks and len(out) < nmsg + 64:
            p, end = blocks.pop(0)
            while p + 8 <= end:
                mtype, msize, mflags = b.u(p, 2), b.u(p + 2, 2), b.u(p + 4, 1)
                p += 8
                if mtype == 0x10:
                    blocks.append((b.addr(p), b.addr(p) + b.length(p + b.so)))
                elif mtype != 0:
                    out.append((mtype, mflags, p, msize))
                p += msize
        return out

    def _group_links(self, addr):
        if addr in self._links:
            return self._links[addr]
        b = self._b
        links = {}
        for mtype, mflags, p, size in self._messages(addr):
            if mtype == 0x11:                                       # symbol table: v1 B-tree + local heap
                btree, heap = b.addr(p), b.addr(p + b.so)
                if b.raw(heap, 4) != b'HEAP':
                    raise H5Error('bad local heap signature')
                hdata = b.addr(heap + 8 + 2 * b.sl)
                self._walk_group(btree, hdata, links)
            elif mtype == 0x06:                                     # link message (new-style compact group)
                ver, lf = b.u(p, 1), b.u(p + 1, 1)
                q = p + 2
                ltype = 0
                if lf & 0x08:
                    ltype = b.u(q, 1)
                    q += 1
                if lf & 0x04:
                    q += 8
                if lf & 0x10:
                    q += 1
                ls = 1 << (lf & 3)
                ln = b.u(q, ls)
                q += ls
                name = b.raw(q, ln).decode('utf-8')
                q += ln
                if ltype == 0:
                    links[name] = b.addr(q)
            elif mtype == 0x02:
                if b.addr(p + 2) is not None and (p + 2 + b.so <= p + size) and b.addr(p + 2 + b.so) is not None:
                    raise H5Error('dense (fractal-heap) groups are not supported by h5lite')
        self._links[addr] = links
        return links

    def _walk_group(self, addr, hdata, links):
        b = self._b
        if addr is None:
            return
        sig = b.raw(addr, 4)
        if sig == b'TREE':
            if b.u(addr + 4, 1) != 0:
                raise H5Error('group B-tree of the wrong node type')
            used = b.u(addr + 6, 2)
            p = addr + 8 + 2 * b.so + b.sl
            for _ in range(used):
                self._walk_group(b.addr(p), hdata, links)
                p += b.so + b.sl
        elif sig == b'SNOD':
            n = b.u(addr + 6, 2)
            p = addr + 8
            for _ in range(n):
                noff, oh = b.u(p, b.so), b.addr(p + b.so)
                q = hdata + noff
                e = q
                while self._mm[e] != 0:
                    e += 1
                if oh is not None:                     # soft links carry no object header address
                    links[b.raw(q, e - q).decode('utf-8')] = oh
                p += 2 * b.so + 24
        else:
            raise H5Error('bad group node signature %r' % sig)

    # -- datasets -----------------------------------------------------------------------------------
    def _dataset(self, addr, name):
        b = self._b
        shape = maxshape = dtype = layout = None
        for mtype, mflags, p, size in self._messages(addr):
            if mtype == 0x01:
                ver, rank, fl = b.u(p, 1), b.u(p + 1, 1), b.u(p + 2, 1)
                q = p + (8 if ver == 1 else 4)
                shape = [b.length(q + i * b.sl) for i in range(rank)]
                if fl & 1:
                    q += rank * b.sl
                    mx = [b.length(q + i * b.sl) for i in range(rank)]
                    maxshape = tuple(None if m == (1 << (8 * b.sl)) - 1 else m for m in mx)
                else:
                    maxshape = tuple(shape)
            elif mtype == 0x03:
                cv = b.u(p, 1)
                cls, bits0, tsize = cv & 0x0F, b.u(p + 1, 1), b.u(p + 4, 4)
                order = '>' if bits0 & 1 else '<'
                if cls == 0:
                    dtype = np.dtype('%s%s%d' % (order, 'i' if bits0 & 0x08 else 'u', tsize))
                elif cls == 1:
                    dtype = np.dtype('%sf%d' % (order, tsize))
                else:
                    raise H5Error('%s: datatype class %d is not supported (numeric datasets only)' % (name, cls))
            elif mtype == 0x0B:
                nf = b.u(p + 1, 1)
                if nf:
                    raise H5Error('%s: filtered datasets (%d filters) are not supported' % (name, nf))
            elif mtype == 0x08:
                ver = b.u(p, 1)
                if ver == 3:
                    cls = b.u(p + 1, 1)
                    if cls == 0:
                        n = b.u(p + 2, 2)
                        layout = ('compact', b.raw(p + 4, n))
                    elif cls == 1:
                        layout = ('contiguous', b.addr(p + 2), b.length(p + 2 + b.so))
                    elif cls == 2:
                        nd = b.u(p + 2, 1)
                        bt = b.addr(p + 3)
                        dims = [b.u(p + 3 + b.so + 4 * i, 4) for i in range(nd)]
                        layout = ('chunked', bt, tuple(dims[:-1]))
                    else:
                        raise H5Error('%s: unsupported layout class %d' % (name, cls))
                elif ver in (1, 2):
                    nd, cls = b.u(p + 1, 1), b.u(p + 2, 1)
                    q = p + 8
                    a = None
                    if cls != 0:
                        a = b.addr(q)
                        q += b.so
                    dims = [b.u(q + 4 * i, 4) for i in range(nd)]
                    if cls == 1:
                        layout = ('contiguous', a, None)
                    elif cls == 2:
                        layout = ('chunked', a, tuple(dims[:-1]))
                    else:
                        q += 4 * nd
                        n = b.u(q, 4)
                        layout = ('compact', b.raw(q + 4, n))
                elif ver == 4:
                    cls = b.u(p + 1, 1)
                    if cls == 1:
                        layout = ('contiguous', b.addr(p + 2), b.length(p + 2 + b.so))
                    elif cls == 0:
                        n = b.u(p + 2, 2)
                        layout = ('compact', b.raw(p + 4, n))
                    else:
                        raise H5Error('%s: version-4 chunk indexes (libver="latest") are not supported; the '
                                      'reference files use the default (earliest) format' % name)
                else:
                    raise H5Error('%s: unsupported data layout message version %d' % (name, ver))
        if shape is None or dtype is None or layout is None:
            return None
        return Dataset(self, name, shape, maxshape, dtype, layout)

    def _resolve(self, name):
        addr = self._root
        parts = [s for s in name.split('/') if s]
        for s in parts:
            links = self._group_links(addr)
            if s not in links:
                raise KeyError("Unable to open object (object '%s' doesn't exist)" % s)
            addr = links[s]
        return addr

    def keys(self):
        return sorted(self._group_links(self._root).keys())

    def __contains__(self, name):
        try:
            self._resolve(name)
            return True
        except KeyError:
            return False

    def __getitem__(self, name):
        if name not in self._objs:
            addr = self._resolve(name)
            ds = self._dataset(addr, name)
            if ds is None:
                raise H5Error('%s is a group; open datasets by their full path' % name)
            self._objs[name] = ds
        return self._objs[name]

    def close(self):
        self._objs = {}
        self._mm = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


# ----------------------------------------------------------------------------------------------- writing
def _align8(n):
    return (n + 7) & ~7


def _dtype_message(dt):
    dt = np.dtype(dt)
    if dt.byteorder == '>':
        raise H5Error('big-endian output is not supported')
    if dt.kind in 'iu':
        bits0 = 0x08 if dt.kind == 'i' else 0
        return struct.pack('<BBBBIHH', 0x10, bits0, 0, 0, dt.itemsize, 0, 8 * dt.itemsize)
    if dt.kind == 'f' and dt.itemsize in (4, 8):
        # IEEE: sign position in byte 1 of the class bits, mantissa normalisation = implied (2 << 4), pad 0
        if dt.itemsize == 4:
            return struct.pack('<BBBBIHHBBBBI', 0x11, 0x20, 31, 0, 4, 0, 32, 23, 8, 0, 23, 127)
        return struct.pack('<BBBBIHHBBBBI', 0x11, 0x20, 63, 0, 8, 0, 64, 52, 11, 0, 52, 1023)
    raise H5Error('dtype %s is not supported' % dt)


def _msg(mtype, data, flags=0):
    pad = _align8(len(data)) - len(data)
    return struct.pack('<HHBBBB', mtype, len(data) + pad, flags, 0, 0, 0) + data + b'\0' * pad


def _object_header(msgs):
    body = b''.join(msgs)
    return struct.pack('<BBHII', 1, 0, len(msgs), 1, len(body)) + b'\0' * 4 + body


def write(path, datasets, istore_k=32):
    """Write numeric arrays as chunked, axis-0-resizable datasets in h5py's default on-disk format.

    ``datasets``: {name: array} or {name: (array, chunk_rows)}; chunks are (chunk_rows,) + shape[1:] (the reference
    uses 100 rows, combine_dicts.py:168-185 / dataset_preparation.py:31-53).  Superblock v0, symbol-table root
    group, version-1 object headers, layout v3 class 2 with a v1 B-tree (node type 1) chunk index, no filters.
    """
    names = sorted(datasets)              # SNOD entries and B-tree keys are ordered by name
    for n in names:
        if '/' in n or not n:
            raise H5Error('h5lite.write: flat dataset names only')
    GK_LEAF, GK_INT = 4, 16
    if len(names) > 2 * GK_LEAF * 2 * GK_INT:
        raise H5Error('too many datasets for a single-level group B-tree')
    blob = bytearray()

    def alloc(data, align=8):
        while len(blob) % align:
            blob.append(0)
        a = len(blob)
        blob.extend(data)
        return a

    sb_size = 8 + 8 + 4 + 4 + 4 * 8 + 40            # superblock v0 with 8-byte offsets/lengths = 96 bytes
    blob.extend(b'\0' * sb_size)
    root_oh_addr = alloc(b'\0' * (16 + 24 + 0))       # placeholder: header + one symbol-table message (8 + 16)

    # local heap: "" at offset 0, then the names, each NUL-terminated and 8-aligned
    heap = bytearray(b'\0' * 8)
    name_off = {}
    for n in names:
        name_off[n] = len(heap)
        e = n.encode('utf-8') + b'\0'
        heap.extend(e + b'\0' * (_align8(len(e)) - len(e)))
    free_off = 1                    # H5HL_FREE_NULL: the data segment is exactly full, no free block

    ds_oh = {}
    for n in names:
        spec = datasets[n]
        arr, crow = spec if isinstance(spec, tuple) else (spec, 100)
        arr = np.ascontiguousarray(arr)
        if arr.ndim < 1:
            raise H5Error('scalar datasets are not supported by h5lite.write')
        dt = arr.dtype.newbyteorder('<') if arr.dtype.byteorder == '>' else arr.dtype
        arr = arr.astype(dt, copy=False)
        rank = arr.ndim
        cdims = (int(crow),) + arr.shape[1:]
        csize = int(np.prod(cdims)) * dt.itemsize
        nchunks = (arr.shape[0] + cdims[0] - 1) // cdims[0]
        # raw chunks (always whole chunks; the tail of the last one is fill value 0)
        entries = []
        for c in range(nchunks):
            rows = arr[c * cdims[0]:(c + 1) * cdims[0]]
            data = rows.tobytes()
            data += b'\0' * (csize - len(data))
            entries.append(((c * cdims[0],) + (0,) * (rank - 1), alloc(data)))
        # chunk B-tree, bottom-up
        def key(offs, size=csize):
            return struct.pack('<II', size, 0) + b''.join(struct.pack('<Q', o) for o in offs) + struct.pack('<Q', 0)

        level_nodes = entries            # [(first chunk offsets, address)]
        level = 0
        end_key = None
        if nchunks:
            end_key = ((nchunks * cdims[0],) + (0,) * (rank - 1))
        btree_addr = UNDEF
        while level_nodes:
            nxt = []
            groups = [level_nodes[i:i + 2 * istore_k] for i in range(0, len(level_nodes), 2 * istore_k)]
            addrs = []
            node_size = 8 + 16 + (2 * istore_k) * (8 + 8 * (rank + 1) + 8) + 8 + 8 * (rank + 1)
            base = len(blob) + (-len(blob)) % 8
            for gi in range(len(groups)):
                addrs.append(base + gi * _align8(node_size))
            for gi, grp in enumerate(groups):
                left = addrs[gi - 1] if gi > 0 else UNDEF
                right = addrs[gi + 1] if gi + 1 < len(groups) else UNDEF
                node = bytearray(b'TREE' + struct.pack('<BBHQQ', 1, level, len(grp), left, right))
                for offs, a in grp:
                    node += key(offs) + struct.pack('<Q', a)
                last = groups[gi + 1][0][0] if gi + 1 < len(groups) else end_key
                node += key(last, 0 if gi + 1 == len(groups) else csize)
                node += b'\0' * (_align8(node_size) - len(node))
                got = alloc(bytes(node))
                assert got == addrs[gi]
                nxt.append((grp[0][0], got))
            if len(nxt) == 1:
                btree_addr = nxt[0][1]
                break
            level_nodes = nxt
            level += 1
        # object header
        space = struct.pack('<BBBBI', 1, rank, 1, 0, 0) + b''.join(struct.pack('<Q', s) for s in arr.shape) + \
            struct.pack('<Q', UNDEF) + b''.join(struct.pack('<Q', s) for s in arr.shape[1:])
        fill = struct.pack('<BBBB', 2, 3, 0, 0)          # v2: allocate incrementally, write fill at allocation, undefined value
        layout = struct.pack('<BBB', 3, 2, rank + 1) + struct.pack('<Q', btree_addr) + \
            b''.join(struct.pack('<I', c) for c in cdims) + struct.pack('<I', dt.itemsize)
        oh = _object_header([_msg(0x01, space), _msg(0x03, _dtype_message(dt), 1), _msg(0x05, fill, 1),
                             _msg(0x08, layout)])
        ds_oh[n] = alloc(oh)

    heap_data_addr = alloc(bytes(heap))
    heap_addr = alloc(b'HEAP' + struct.pack('<BBBBQQQ', 0, 0, 0, 0, len(heap), free_off, heap_data_addr))
    # symbol nodes (<= 2*GK_LEAF entries each) and the single B-tree node above them
    snods = []
    for i in range(0, max(len(names), 1), 2 * GK_LEAF):
        grp = names[i:i + 2 * GK_LEAF]
        node = bytearray(b'SNOD' + struct.pack('<BBH', 1, 0, len(grp)))
        for n in grp:
            node += struct.pack('<QQII', name_off[n], ds_oh[n], 0, 0) + b'\0' * 16
        node += b'\0' * (8 + 2 * GK_LEAF * 40 - len(node))
        snods.append((grp, alloc(bytes(node))))
    bt = bytearray(b'TREE' + struct.pack('<BBHQQ', 0, 0, len(snods), UNDEF, UNDEF))
    bt += struct.pack('<Q', 0)
    for grp, a in snods:
        bt += struct.pack('<QQ', a, name_off[grp[-1]] if grp else 0)
    bt += b'\0' * (8 + 16 + 8 + 2 * GK_INT * 16 - len(bt))
    btree_addr = alloc(bytes(bt))
    root = _object_header([_msg(0x11, struct.pack('<QQ', btree_addr, heap_addr))])
    blob[root_oh_addr:root_oh_addr + len(root)] = root
    eof = len(blob)
    sb = SIGNATURE + struct.pack('<BBBBBBBBHHI', 0, 0, 0, 0, 0, 8, 8, 0, GK_LEAF, GK_INT, 0) + \
        struct.pack('<QQQQ', 0, UNDEF, eof, UNDEF) + \
        struct.pack('<QQII', 0, root_oh_addr, 1, 0) + struct.pack('<QQ', btree_addr, heap_addr)
    assert len(sb) == sb_size
    blob[0:sb_size] = sb
    with open(path, 'wb') as f:
        f.write(blob)
    return path
