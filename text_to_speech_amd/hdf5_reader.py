"""Minimal pure-Python reader for the subset of HDF5 that Keras `.weights.h5` / `.h5` checkpoints use.

Why it exists: the reference's shipped models (`pretrained_tacotron2`, `sv2tts_siwis_v2`, `WaveGlow`) are Keras H5 weight
files (`custom_train_objects/checkpoint_manager.py:193-195` -> `model.load_weights`), and neither h5py nor Keras is
available where this engine is built and run.  Written from the published HDF5 File Format Specification (version 3.0);
validated against files produced by the real libhdf5 (h5py 3.3 / HDF5 1.10.6; `tests/golden/make_h5_fixtures.py`).

Supported: superblock versions 0-3; object headers version 1 and 2 (with continuation blocks); old-style groups (symbol
table: v1 B-tree + local heap + SNOD nodes) and new-style groups with compact link messages; datasets of fixed-point and
floating-point type (little or big endian; IEEE half / single / double and bfloat16) and of string type (fixed length, or
variable length through global heap collections -- what h5py's `string_dtype()` writes; returned as object arrays of `str`)
in compact, contiguous or chunked (v1 B-tree; layout version 4: single-chunk and implicit index) layout; deflate / shuffle /
fletcher32 filters.
Not supported (clear `H5Error`): dense link storage (fractal heaps), v2 B-tree / array chunk indexes, compound /
variable-length sequence / reference types, external or virtual storage, soft / external links (skipped while walking).
"""
from __future__ import annotations

import functools
import math
import mmap
import struct
import zlib
from collections import OrderedDict

import numpy as np

SIGNATURE = b'\x89HDF\r\n\x1a\n'


class H5Error(ValueError):
    pass


def _guard(fn):
    """A corrupt file must surface as H5Error, whatever low-level exception its bytes provoke."""
    @functools.wraps(fn)
    def wrapped(*args, **kw):
        try:
            return fn(*args, **kw)
        except H5Error:
            raise
        except (ValueError, ArithmeticError, MemoryError, IndexError, TypeError, struct.error, zlib.error) as e:
            raise H5Error(f'truncated or corrupt HDF5 file: {type(e).__name__}: {e}') from None
    return wrapped


MAX_ELEMENTS = 1 << 34            # sanity bound on one dataset (corrupt dimension fields)


class _Msg:
    __slots__ = ('type', 'flags', 'data')

    def __init__(self, type_, flags, data):
        self.type, self.flags, self.data = type_, flags, data


class H5Dataset:
    """Shape / dtype are parsed eagerly; `read()` materialises the array."""

    def __init__(self, file, path, shape, dtype, bfloat16, layout, filters, string=None):
        self.file, self.path, self.shape, self.dtype = file, path, tuple(shape), dtype
        self._bf16, self._layout, self._filters = bfloat16, layout, filters
        self._string = string                    # None | 'fixed' (dtype S<n>) | 'vlen' (dtype V16: length + global heap id)

    def __repr__(self):
        return f'H5Dataset({self.path!r}, shape={self.shape}, dtype={self.dtype})'

    @_guard
    def read(self) -> np.ndarray:
        f, n = self.file, math.prod(self.shape)
        if n > MAX_ELEMENTS:
            raise H5Error(f'{self.path}: implausible shape {self.shape}')
        item = self.dtype.itemsize
        kind = self._layout[0]
        if kind == 'compact':
            raw = self._layout[1]
        elif kind == 'contiguous':
            addr, size = self._layout[1], self._layout[2]
            if addr is None or n == 0:
                raw = bytes(n * item)                                   # never written: the (zero) fill value
            else:
                if size < n * item:
                    raise H5Error(f'{self.path}: contiguous storage holds {size} bytes, {n * item} needed')
                raw = f._bytes(addr, n * item)
        else:
            return self._read_chunked()
        if len(raw) < n * item:
            raise H5Error(f'{self.path}: {len(raw)} bytes of data, {n * item} needed')
        return self._finish(np.frombuffer(raw, dtype=self.dtype, count=n).reshape(self.shape))

    def _finish(self, a):
        if self._string == 'fixed':
            flat = [bytes(x).split(b'\0', 1)[0].decode('utf-8', 'replace') for x in a.reshape(-1)]
            out = np.empty(len(flat), dtype=object)
            out[:] = flat
            return out.reshape(a.shape)
        if self._string == 'vlen':
            O = self.file._O
            raw = np.ascontiguousarray(a).view(np.uint8).reshape(-1, 8 + O)
            out = np.empty(len(raw), dtype=object)
            for i, e in enumerate(raw):
                e = e.tobytes()
                n = int.from_bytes(e[:4], 'little')
                addr = self.file._addr(e[4:4 + O])
                idx = int.from_bytes(e[4 + O:8 + O], 'little')
                out[i] = '' if addr is None or n == 0 else self.file._global_heap_object(addr, idx)[:n].decode('utf-8', 'replace')
            return out.reshape(a.shape)
        if self._bf16:
            a = (a.astype(np.uint32) << 16).view(np.float32)
        if a.dtype.byteorder == '>':
            a = a.astype(a.dtype.newbyteorder('<'))
        return np.array(a)                                               # own the memory (the file may be closed)

    def _unfilter(self, raw, mask):
        for idx in range(len(self._filters) - 1, -1, -1):                # decode in reverse pipeline order
            fid, cd = self._filters[idx]
            if mask >> idx & 1:
                continue
            if fid == 1:
                raw = zlib.decompress(raw)
            elif fid == 2:
                size = cd[0] if cd else self.dtype.itemsize
                k = len(raw) // size
                body = np.frombuffer(raw, np.uint8, k * size).reshape(size, k).T.tobytes()
                raw = body + bytes(raw[k * size:])
            elif fid == 3:
                raw = raw[:-4]
            else:
                raise H5Error(f'{self.path}: unsupported filter id {fid}')
        return raw

    def _read_chunked(self):
        _, index, chunk = self._layout
        rank, item = len(self.shape), self.dtype.itemsize
        if len(chunk) != rank:
            raise H5Error(f'{self.path}: chunk rank {len(chunk)} != dataset rank {rank}')
        out = np.zeros(self.shape, dtype=self.dtype)
        cbytes = math.prod(chunk) * item
        if cbytes <= 0 or cbytes > 1 << 32:
            raise H5Error(f'{self.path}: implausible chunk shape {chunk}')
        for offset, addr, size, mask in index():
            raw = self.file._bytes(addr, size)
            raw = self._unfilter(raw, mask) if self._filters else raw
            if len(raw) < cbytes:
                raise H5Error(f'{self.path}: chunk at {offset} decodes to {len(raw)} bytes, {cbytes} expected')
            block = np.frombuffer(raw, self.dtype, cbytes // item).reshape(chunk)
            sel = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offset, chunk, self.shape))
            if any(s.start >= s.stop for s in sel):
                continue
            out[sel] = block[tuple(slice(0, s.stop - s.start) for s in sel)]
        return self._finish(out)


class H5File:
    """`H5File(path)`; `.datasets()` -> {'/a/b/c': H5Dataset}; `.read('/a/b/c')` -> ndarray; usable as a context manager."""

    def __init__(self, source):
        self._fh = self._mm = None
        if isinstance(source, (bytes, bytearray, memoryview)):
            self._buf = memoryview(bytes(source))
        else:
            self._fh = open(source, 'rb')
            try:
                self._mm = mmap.mmap(self._fh.fileno(), 0, access=mmap.ACCESS_READ)
            except ValueError as e:                                       # empty file
                self._fh.close()
                raise H5Error(f'{source}: {e}') from None
            self._buf = memoryview(self._mm)
        try:
            self._parse_superblock()
        except (struct.error, IndexError) as e:
            self.close()
            raise H5Error(f'truncated or corrupt HDF5 file: {e}') from None
        except H5Error:
            self.close()
            raise
        self._datasets = None

    # -- plumbing ----------------------------------------------------------------------------------------------------
    def close(self):
        buf, self._buf = getattr(self, '_buf', None), None
        if buf is not None:
            try:
                buf.release()
            except BufferError:
                pass
        if self._mm is not None:
            try:
                self._mm.close()
            except BufferError:               # a view is still alive (e.g. in a traceback): the mapping goes with it
                pass
            self._mm = None
        if self._fh is not None:
            self._fh.close()
            self._fh = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _bytes(self, addr, size):
        a = self._base + addr
        if addr < 0 or size < 0 or a + size > len(self._buf):
            raise H5Error(f'address range [{a}, {a + size}) lies outside the file ({len(self._buf)} bytes)')
        return self._buf[a:a + size]

    def _uint(self, addr, size):
        return int.from_bytes(self._bytes(addr, size), 'little')

    def _addr(self, raw):
        v = int.from_bytes(raw, 'little')
        return None if v == (1 << (8 * len(raw))) - 1 else v

    def _parse_superblock(self):
        buf, pos = self._buf, 0
        while True:                                                       # the superblock may sit at 0, 512, 1024, ...
            if pos + 8 > len(buf):
                raise H5Error('not an HDF5 file (signature not found)')
            if bytes(buf[pos:pos + 8]) == SIGNATURE:
                break
            pos = 512 if pos == 0 else pos * 2
        self._base = 0
        ver = buf[pos + 8]
        if ver in (0, 1):
            self._O, self._L = buf[pos + 13], buf[pos + 14]
            p = pos + 24 + (4 if ver == 1 else 0)
            O = self._O
            base = self._addr(buf[p:p + O])
            p += 4 * O                                                    # base, free-space, end-of-file, driver info
            entry = bytes(buf[p:p + 2 * O + 24])
            if len(entry) < 2 * O + 24:
                raise H5Error('truncated superblock')
            root = self._addr(entry[O:2 * O])
        elif ver in (2, 3):
            self._O, self._L = buf[pos + 9], buf[pos + 10]
            O, p = self._O, pos + 12
            if p + 4 * O > len(buf):
                raise H5Error('truncated superblock')
            base = self._addr(buf[p:p + O])
            root = self._addr(buf[p + 3 * O:p + 4 * O])
        else:
            raise H5Error(f'unsupported superblock version {ver}')
        if self._O not in (2, 4, 8) or self._L not in (2, 4, 8):
            raise H5Error(f'unsupported offset / length sizes {self._O} / {self._L}')
        self._base = base or 0
        if root is None:
            raise H5Error('the file has no root group')
        self._root = root

    # -- object headers ------------------------------------------------------------------------------------------------
    def _messages(self, addr):
        O, L = self._O, self._L
        head = bytes(self._bytes(addr, 16))
        msgs = []
        if head[:4] == b'OHDR':
            if head[4] != 2:
                raise H5Error(f'unsupported object header version {head[4]}')
            flags, p = head[5], addr + 6
            if flags & 0x20:
                p += 16
            if flags & 0x10:
                p += 4
            w = 1 << (flags & 3)
            size = self._uint(p, w)
            p += w
            blocks, order = [(p, size)], 2 if flags & 0x04 else 0
            while blocks:
                start, length = blocks.pop(0)
                q, end = start, start + length
                while q + 4 + order <= end:
                    raw = bytes(self._bytes(q, 4))
                    mtype, msize, mflags = raw[0], raw[1] | raw[2] << 8, raw[3]
                    q += 4 + order
                    if q + msize > end:
                        break
                    data = self._bytes(q, msize)
                    q += msize
                    if mtype == 0x10:
                        caddr, clen = self._addr(data[:O]), int.from_bytes(data[O:O + L], 'little')
                        if bytes(self._bytes(caddr, 4)) != b'OCHK':
                            raise H5Error('bad object header continuation block')
                        blocks.append((caddr + 4, clen - 8))             # signature in front, checksum behind
                    elif mtype != 0:
                        msgs.append(_Msg(mtype, mflags, bytes(data)))
            return msgs
        if head[0] != 1:
            raise H5Error(f'unsupported object header version {head[0]} at {addr}')
        count = head[2] | head[3] << 8
        size = int.from_bytes(head[8:12], 'little')
        blocks, seen = [(addr + 16, size)], 0
        while blocks and seen < count:
            start, length = blocks.pop(0)
            q, end = start, start + length
            while q + 8 <= end and seen < count:
                raw = bytes(self._bytes(q, 8))
                mtype, msize, mflags = raw[0] | raw[1] << 8, raw[2] | raw[3] << 8, raw[4]
                q += 8
                if q + msize > end:
                    raise H5Error('object header message overruns its block')
                data = self._bytes(q, msize)
                q += msize
                seen += 1
                if mtype == 0x10:
                    blocks.append((self._addr(data[:O]), int.from_bytes(data[O:O + L], 'little')))
                elif mtype != 0:
                    msgs.append(_Msg(mtype, mflags, bytes(data)))
        return msgs

    # -- groups --------------------------------------------------------------------------------------------------------
    def _heap_name(self, heap_data_addr, heap_size, offset):
        if offset >= heap_size:
            raise H5Error('link name offset outside the local heap')
        raw = bytes(self._bytes(heap_data_addr + offset, min(heap_size - offset, 4096)))
        end = raw.find(b'\0')
        return raw[:end if end >= 0 else len(raw)].decode('utf-8')

    def _symbol_table_links(self, btree, heap):
        O, L = self._O, self._L
        h = bytes(self._bytes(heap, 8 + 2 * L + O))
        if h[:4] != b'HEAP':
            raise H5Error('bad local heap signature')
        heap_size = int.from_bytes(h[8:8 + L], 'little')
        heap_data = self._addr(h[8 + 2 * L:8 + 2 * L + O])
        links, stack, guard = [], [btree], 0
        while stack:
            node = stack.pop()
            guard += 1
            if guard > 1 << 20:
                raise H5Error('group B-tree does not terminate')
            hd = bytes(self._bytes(node, 8))
            if hd[:4] == b'TREE':
                if hd[4] != 0:
                    raise H5Error('group B-tree node has the wrong type')
                used = hd[6] | hd[7] << 8
                body = self._bytes(node + 8 + 2 * O, used * (L + O) + L)
                children = [self._addr(body[i * (L + O) + L:(i + 1) * (L + O)]) for i in range(used)]
                stack.extend(reversed(children))
            elif hd[:4] == b'SNOD':
                n = hd[6] | hd[7] << 8
                esz = 2 * O + 24
                body = self._bytes(node + 8, n * esz)
                for i in range(n):
                    e = body[i * esz:(i + 1) * esz]
                    name = self._heap_name(heap_data, heap_size, int.from_bytes(e[:O], 'little'))
                    links.append((name, self._addr(e[O:2 * O])))
            else:
                raise H5Error(f'unexpected signature {hd[:4]!r} in a group B-tree')
        return links

    def _link_message(self, data):
        O = self._O
        data = bytes(data)
        if data[0] != 1:
            raise H5Error(f'unsupported link message version {data[0]}')
        flags, p, ltype = data[1], 2, 0
        if flags & 0x08:
            ltype = data[p]
            p += 1
        if flags & 0x04:
            p += 8
        if flags & 0x10:
            p += 1
        w = 1 << (flags & 3)
        n = int.from_bytes(data[p:p + w], 'little')
        p += w
        name = data[p:p + n].decode('utf-8')
        p += n
        if ltype != 0:
            return name, None                                             # soft / external link: not followed
        return name, self._addr(data[p:p + O])

    def _children(self, msgs, path):
        links = []
        for m in msgs:
            if m.type == 0x11:
                O = self._O
                links += self._symbol_table_links(self._addr(m.data[:O]), self._addr(m.data[O:2 * O]))
            elif m.type == 0x06:
                links.append(self._link_message(m.data))
            elif m.type == 0x02:
                d = bytes(m.data)
                p = 2 + (8 if d[1] & 1 else 0)
                if self._addr(d[p:p + self._O]) is not None:
                    raise H5Error(f"group '{path or '/'}' uses dense link storage (fractal heap), which this reader does not "
                                  "parse; rewrite the file with the default (earliest) library version bounds, e.g. "
                                  "`h5repack --low=0 --high=0 in.h5 out.h5`")
        return links

    # -- global heap (variable-length data) -----------------------------------------------------------------------------
    def _global_heap_object(self, addr, index):
        L = self._L
        head = bytes(self._bytes(addr, 8 + L))
        if head[:4] != b'GCOL' or head[4] != 1:
            raise H5Error('bad global heap collection')
        size = int.from_bytes(head[8:8 + L], 'little')
        p, end = addr + 8 + L, addr + size
        guard = 0
        while p + 8 + L <= end:
            guard += 1
            if guard > 1 << 20:
                break
            h = bytes(self._bytes(p, 8 + L))
            idx, osz = h[0] | h[1] << 8, int.from_bytes(h[8:8 + L], 'little')
            if idx == 0:
                break                                                     # free space: end of the used part
            if idx == index:
                return bytes(self._bytes(p + 8 + L, osz))
            p += 8 + L + (osz + 7) // 8 * 8
        raise H5Error(f'global heap object {index} not found in the collection at {addr}')

    # -- datasets ------------------------------------------------------------------------------------------------------
    def _dataspace(self, data):
        d, L = bytes(data), self._L
        ver, rank, flags = d[0], d[1], d[2]
        if ver == 1:
            p = 8
        elif ver == 2:
            p = 4
            if d[3] == 2:                                                 # null dataspace
                return (0,)
        else:
            raise H5Error(f'unsupported dataspace version {ver}')
        return tuple(int.from_bytes(d[p + i * L:p + (i + 1) * L], 'little') for i in range(rank))

    def _datatype(self, data, path):
        d = bytes(data)
        cls, bits0, size = d[0] & 0x0f, d[1], int.from_bytes(d[4:8], 'little')
        order = '>' if bits0 & 1 else '<'
        if cls == 0:
            if size not in (1, 2, 4, 8):
                raise H5Error(f'{path}: unsupported integer size {size}')
            return np.dtype(f"{order}{'i' if bits0 & 0x08 else 'u'}{size}"), False
        if cls == 1:
            exp_size = d[8 + 5]
            if size == 2 and exp_size == 8:
                return np.dtype(f'{order}u2'), True                       # bfloat16
            if (size, exp_size) not in ((2, 5), (4, 8), (8, 11)):
                raise H5Error(f'{path}: unsupported floating-point format (size {size}, exponent bits {exp_size})')
            return np.dtype(f'{order}f{size}'), False
        if cls == 3:
            if size <= 0 or size > 1 << 20:
                raise H5Error(f'{path}: implausible string size {size}')
            return np.dtype(f'S{size}'), 'fixed'
        if cls == 9 and (bits0 & 0x0f) == 1:                              # variable-length STRING (not a sequence)
            return np.dtype(f'V{8 + self._O}'), 'vlen'
        raise H5Error(f'{path}: unsupported datatype class {cls} (numeric and string datasets are read)')

    def _filters(self, data, path):
        d = bytes(data)
        ver, n, out = d[0], d[1], []
        p = 8 if ver == 1 else 2
        for _ in range(n):
            fid = d[p] | d[p + 1] << 8
            p += 2
            name_len = 0
            if ver == 1 or fid >= 256:
                name_len = d[p] | d[p + 1] << 8
                p += 2
            p += 2                                                        # flags
            ncd = d[p] | d[p + 1] << 8
            p += 2
            p += (name_len + 7) // 8 * 8 if ver == 1 else name_len
            cd = [int.from_bytes(d[p + 4 * i:p + 4 * i + 4], 'little') for i in range(ncd)]
            p += 4 * ncd + (4 if ver == 1 and ncd % 2 else 0)
            out.append((fid, cd))
        return out

    def _chunk_btree(self, root, rank, path):
        O = self._O

        def walk():
            if root is None:
                return
            stack, guard = [root], 0
            while stack:
                node = stack.pop()
                guard += 1
                if guard > 1 << 22:
                    raise H5Error(f'{path}: chunk B-tree does not terminate')
                hd = bytes(self._bytes(node, 8))
                if hd[:4] != b'TREE' or hd[4] != 1:
                    raise H5Error(f'{path}: bad chunk B-tree node')
                level, used = hd[5], hd[6] | hd[7] << 8
                ksz = 8 + 8 * (rank + 1)
                body = self._bytes(node + 8 + 2 * O, used * (ksz + O) + ksz)
                for i in range(used):
                    e = body[i * (ksz + O):(i + 1) * (ksz + O)]
                    child = self._addr(e[ksz:ksz + O])
                    if level > 0:
                        stack.append(child)
                    else:
                        size, mask = struct.unpack_from('<II', e, 0)
                        offset = struct.unpack_from(f'<{rank}Q', e, 8)
                        yield offset, child, size, mask
        return walk

    def _layout(self, data, shape, itemsize, has_filters, path):
        d, O, L = bytes(data), self._O, self._L
        ver = d[0]
        if ver not in (3, 4):
            raise H5Error(f'{path}: unsupported data layout message version {ver}')
        cls = d[1]
        if cls == 0:
            n = d[2] | d[3] << 8
            return ('compact', d[4:4 + n])
        if cls == 1:
            return ('contiguous', self._addr(d[2:2 + O]), int.from_bytes(d[2 + O:2 + O + L], 'little'))
        if cls != 2:
            raise H5Error(f'{path}: unsupported storage class {cls} (virtual dataset?)')
        if ver == 3:
            nd = d[2]
            root = self._addr(d[3:3 + O])
            dims = [int.from_bytes(d[3 + O + 4 * i:7 + O + 4 * i], 'little') for i in range(nd)]
            return ('chunked', self._chunk_btree(root, nd - 1, path), tuple(dims[:-1]))
        flags, nd, enc = d[2], d[3], d[4]
        dims = [int.from_bytes(d[5 + enc * i:5 + enc * (i + 1)], 'little') for i in range(nd)]
        p = 5 + enc * nd
        itype = d[p]
        p += 1
        chunk = tuple(dims[:-1])
        if itype == 1:                                                    # single chunk
            size, mask = math.prod(dims), 0
            if flags & 2:
                size = int.from_bytes(d[p:p + L], 'little')
                mask = int.from_bytes(d[p + L:p + L + 4], 'little')
                p += L + 4
            addr = self._addr(d[p:p + O])
            return ('chunked', lambda: iter(() if addr is None else [((0,) * len(chunk), addr, size, mask)]), chunk)
        if itype == 2:                                                    # implicit: all chunks, in order, unfiltered
            addr = self._addr(d[p:p + O])
            cbytes = math.prod(dims)
            counts = [-(-s // c) for s, c in zip(shape, chunk)]

            def implicit():
                if addr is None:
                    return
                for i, idx in enumerate(np.ndindex(*counts)):
                    yield tuple(k * c for k, c in zip(idx, chunk)), addr + i * cbytes, cbytes, 0
            return ('chunked', implicit, chunk)
        raise H5Error(f'{path}: chunk index type {itype} (fixed / extensible array or v2 B-tree) is not supported; '
                      'rewrite the file with `h5repack --low=0 --high=0` or store the dataset contiguously')

    def _dataset(self, msgs, path):
        shape = dtype = layout_msg = None
        filters, bf16, string = [], False, None
        for m in msgs:
            if m.type == 0x01:
                shape = self._dataspace(m.data)
            elif m.type == 0x03:
                dtype, kind = self._datatype(m.data, path)
                bf16, string = kind is True, kind if isinstance(kind, str) else None
            elif m.type == 0x08:
                layout_msg = m.data
            elif m.type == 0x0B:
                filters = self._filters(m.data, path)
        if shape is None or dtype is None or layout_msg is None:
            raise H5Error(f'{path}: incomplete dataset header')
        layout = self._layout(layout_msg, shape, dtype.itemsize, bool(filters), path)
        return H5Dataset(self, path, shape, dtype, bf16, layout, filters, string)

    # -- public --------------------------------------------------------------------------------------------------------
    @_guard
    def datasets(self) -> 'OrderedDict[str, H5Dataset]':
        """Every dataset reachable from the root group through hard links, keyed by absolute path.  Datasets of a type this
        reader cannot decode are skipped (Keras files hold only numeric arrays; attributes are never read)."""
        if self._datasets is not None:
            return self._datasets
        out, visited = OrderedDict(), set()
        stack = [('', self._root)]
        while stack:
            path, addr = stack.pop()
            if addr is None or addr in visited:
                continue
            visited.add(addr)
            msgs = self._messages(addr)
            types = {m.type for m in msgs}
            if 0x08 in types:
                try:
                    out[path or '/'] = self._dataset(msgs, path)
                except H5Error as e:
                    if 'unsupported datatype class' not in str(e):
                        raise
                continue
            for name, child in sorted(self._children(msgs, path), reverse=True):
                stack.append((f'{path}/{name}', child))
        self._datasets = out
        return out

    def read(self, path: str) -> np.ndarray:
        ds = self.datasets().get('/' + path.strip('/'))
        if ds is None:
            raise KeyError(path)
        return ds.read()


def read_all(source) -> 'OrderedDict[str, np.ndarray]':
    """{absolute dataset path: array} of a whole file."""
    with H5File(source) as f:
        return OrderedDict((k, d.read()) for k, d in f.datasets().items())
