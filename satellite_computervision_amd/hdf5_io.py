"""Minimal pure-Python HDF5 READER for Keras weight files (`Model.save_weights('x.h5')`, `Model.save('x.h5')`,
ModelCheckpoint `.hdf5`) -- the weight interchange the reference's callers rely on (`models.load_model` / `m.load_weights(...,
by_name, skip_mismatch)`, utils/model_tools.py:1128-1176, 1178-1269) without h5py or TensorFlow.

Scope: what h5py writes for such files with its default settings (HDF5 1.8/1.10 "earliest" structures) --
superblock v0/v1 (v2/v3 root pointer accepted), version-1 object headers with continuation blocks (version-2 headers with compact
link messages are parsed too), old-style groups (symbol-table message -> v1 B-tree -> SNOD nodes -> local heap), simple / scalar
dataspaces, fixed-point / IEEE float / fixed-length string / variable-length string (global heap) datatypes, contiguous, compact
and chunked (v1 chunk B-tree, optional shuffle + deflate filters) layouts, attribute messages v1-v3.  Anything else (dense
attribute / link storage in fractal heaps, shared messages, compound types, other filters) raises NotImplementedError naming the
feature.  Format reference: "HDF5 File Format Specification Version 2.0" (The HDF Group); pinned against files written by the real
library (tests/golden/make_h5_fixtures.py, tests/test_hdf5_cpu.py).
"""
import struct
import zlib

import numpy as np

SIGNATURE = b'\x89HDF\r\n\x1a\n'
UNDEF = 0xFFFFFFFFFFFFFFFF


def is_hdf5(path):
    try:
        with open(path, 'rb') as f:
            return f.read(8) == SIGNATURE
    except OSError:
        return False


class _Type:
    """Decoded datatype message."""

    def __init__(self, cls, size, dtype=None, vlen_string=False, pad=0):
        self.cls, self.size, self.dtype, self.vlen_string, self.pad = cls, size, dtype, vlen_string, pad


class Dataset:
    def __init__(self, f, name, dtype, shape, layout, filters):
        self._f, self.name, self._type, self.shape, self._layout, self._filters = f, name, dtype, tuple(shape), layout, filters
        self.attrs = {}

    @property
    def dtype(self):
        return self._type.dtype

    def read(self):
        return self._f._read_dataset(self)

    def __array__(self, dtype=None, copy=None):
        a = self.read()
        return a.astype(dtype) if dtype is not None else a


class Group:
    def __init__(self, f, name):
        self._f, self.name = f, name
        self.attrs = {}
        self._links = {}              # child name -> object header address (insertion = B-tree order = name order)

    def keys(self):
        return list(self._links)

    def __contains__(self, key):
        try:
            self[key]
            return True
        except KeyError:
            return False

    def __getitem__(self, path):
        node = self
        for part in [p for p in path.split('/') if p]:
            if not isinstance(node, Group) or part not in node._links:
                raise KeyError(f'{path!r} not found under {self.name!r}')
            node = node._f._object(node._links[part], (node.name.rstrip('/') + '/' + part))
        return node


class File(Group):
    """Read-only view of an HDF5 file: `f['group/dataset'].read()`, `.attrs`, `.keys()`."""

    def __init__(self, path):
        with open(path, 'rb') as fh:
            self._buf = fh.read()
        self._cache = {}
        base = self._buf.find(SIGNATURE)
        if base != 0:
            raise ValueError(f'{path}: not an HDF5 file (no signature at offset 0)')
        ver = self._buf[8]
        if ver in (0, 1):
            self.O, self.L = self._buf[13], self._buf[14]
            pos = 24 + (4 if ver == 1 else 0)
            pos += 4 * self.O                      # base address, free-space info, end of file, driver info
            # root group symbol table entry: link name offset, object header address, cache type, reserved, scratch
            root = self._addr(pos + self.O)
        elif ver in (2, 3):
            self.O, self.L = self._buf[9], self._buf[10]
            root = self._addr(12 + 3 * self.O)
        else:
            raise NotImplementedError(f'HDF5 superblock version {ver}')
        if self.O != 8 or self.L != 8:
            raise NotImplementedError(f'HDF5 files with {self.O}-byte offsets / {self.L}-byte lengths')
        Group.__init__(self, self, '/')
        g = self._object(root, '/')
        self.attrs, self._links = g.attrs, g._links

    def close(self):
        self._buf = b''

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- primitives
    def _addr(self, pos):
        return struct.unpack_from('<Q', self._buf, pos)[0]

    def _u(self, pos, n):
        return int.from_bytes(self._buf[pos:pos + n], 'little')

    # ---- object headers
    def _messages(self, addr):
        """[(type, flags, payload bytes)] of the object header at `addr` (continuation blocks followed)."""
        b = self._buf
        out = []
        if b[addr:addr + 4] == b'OHDR':
            if b[addr + 4] != 2:
                raise NotImplementedError('object header version')
            flags = b[addr + 5]
            pos = addr + 6
            if flags & 0x20:
                pos += 16
            if flags & 0x10:
                pos += 4
            n = 1 << (flags & 3)
            size = self._u(pos, n)
            pos += n
            blocks = [(pos, size)]
            track = bool(flags & 0x04)
            while blocks:
                p, sz = blocks.pop(0)
                end = p + sz
                while p + 4 <= end:
                    mtype, msize, mflags = b[p], self._u(p + 1, 2), b[p + 3]
                    p += 4 + (2 if track else 0)
                    data = b[p:p + msize]
                    p += msize
                    if mtype == 0x10:
                        coff, clen = struct.unpack_from('<QQ', data, 0)
                        if b[coff:coff + 4] != b'OCHK':
                            raise ValueError('bad object header continuation')
                        blocks.append((coff + 4, clen - 8))
                    elif mtype != 0:
                        out.append((mtype, mflags, data))
            return out
        if b[addr] != 1:
            raise NotImplementedError(f'object header version {b[addr]} at {addr}')
        nmsg = self._u(addr + 2, 2)
        size = self._u(addr + 8, 4)
        blocks = [(addr + 16, size)]
        while blocks and len(out) < nmsg + 64:
            p, sz = blocks.pop(0)
            end = p + sz
            while p + 8 <= end:
                mtype, msize, mflags = self._u(p, 2), self._u(p + 2, 2), b[p + 4]
                data = b[p + 8:p + 8 + msize]
                p += 8 + msize
                if mtype == 0x10:
                    coff, clen = struct.unpack_from('<QQ', data, 0)
                    blocks.append((coff, clen))
                elif mtype != 0:
                    out.append((mtype, mflags, data))
        return out

    def _object(self, addr, name):
        if addr in self._cache:
            return self._cache[addr]
        msgs = self._messages(addr)
        types = {t for t, _, _ in msgs}
        for t, fl, _ in msgs:
            if fl & 0x02 and t in (0x01, 0x03, 0x0B):
                raise NotImplementedError('shared object header messages (committed datatypes)')
        if 0x08 in types:                                   # data layout -> dataset
            dt = space = layout = None
            filters = []
            for t, fl, d in msgs:
                if t == 0x03:
                    dt = self._datatype(d)
                elif t == 0x01:
                    space = self._dataspace(d)
                elif t == 0x08:
                    layout = self._layout(d)
                elif t == 0x0B:
                    filters = self._filters(d)
            obj = Dataset(self, name, dt, space, layout, filters)
        else:
            obj = Group(self, name)
            for t, fl, d in msgs:
                if t == 0x11:                               # symbol table message: B-tree + local heap
                    btree, heap = struct.unpack_from('<QQ', d, 0)
                    hdata = self._local_heap(heap)
                    for noff, oaddr in self._group_btree(btree):
                        end = self._buf.index(b'\x00', hdata + noff)
                        obj._links[self._buf[hdata + noff:end].decode('utf8')] = oaddr
                elif t == 0x06:                             # link message (compact new-style group)
                    lname, oaddr = self._link(d)
                    if oaddr is not None:
                        obj._links[lname] = oaddr
                elif t == 0x02:                             # link info: dense storage when a fractal heap is present
                    fl2 = d[1]
                    p = 2 + (8 if fl2 & 1 else 0)
                    if self._addr_in(d, p) != UNDEF:
                        raise NotImplementedError('dense link storage (fractal heap): re-save the file with h5py defaults')
        for t, fl, d in msgs:
            if t == 0x0C:
                k, v = self._attribute(d)
                obj.attrs[k] = v
            elif t == 0x15:
                p = 2 + (2 if d[1] & 1 else 0)
                if self._addr_in(d, p) != UNDEF:
                    raise NotImplementedError('dense attribute storage (fractal heap)')
        self._cache[addr] = obj
        return obj

    @staticmethod
    def _addr_in(d, p):
        return struct.unpack_from('<Q', d, p)[0]

    # ---- groups
    def _local_heap(self, addr):
        b = self._buf
        if b[addr:addr + 4] != b'HEAP':
            raise ValueError('bad local heap')
        return self._addr(addr + 8 + 2 * self.L)

    def _group_btree(self, addr):
        """(name offset in the local heap, object header address) of every entry, in key (= name) order."""
        b = self._buf
        if b[addr:addr + 4] == b'SNOD':
            n = self._u(addr + 6, 2)
            p = addr + 8
            for _ in range(n):
                yield self._addr(p), self._addr(p + 8)
                p += 40
            return
        if b[addr:addr + 4] != b'TREE' or b[addr + 4] != 0:
            raise ValueError('bad group B-tree node')
        used = self._u(addr + 6, 2)
        p = addr + 8 + 2 * self.O
        for _ in range(used):
            p += self.L                                   # key i
            child = self._addr(p)
            p += self.O
            yield from self._group_btree(child)

    def _link(self, d):
        flags = d[1]
        p = 2
        ltype = 0
        if flags & 0x08:
            ltype = d[p]; p += 1
        if flags & 0x04:
            p += 8
        if flags & 0x10:
            p += 1
        n = 1 << (flags & 3)
        ln = int.from_bytes(d[p:p + n], 'little'); p += n
        name = bytes(d[p:p + ln]).decode('utf8'); p += ln
        if ltype != 0:
            return name, None                               # soft / external links are not followed
        return name, struct.unpack_from('<Q', d, p)[0]

    # ---- message decoders
    def _dataspace(self, d):
        ver, rank, flags = d[0], d[1], d[2]
        if ver == 1:
            p = 8
        elif ver == 2:
            p = 4
            if d[3] == 2:
                return None                                 # null dataspace
        else:
            raise NotImplementedError('dataspace message version')
        return tuple(struct.unpack_from('<Q', d, p + 8 * i)[0] for i in range(rank))

    def _datatype(self, d):
        cls, bits = d[0] & 0x0f, int.from_bytes(d[1:4], 'little')
        size = struct.unpack_from('<I', d, 4)[0]
        if cls == 0:
            order = '>' if bits & 1 else '<'
            return _Type(cls, size, np.dtype(f"{order}{'i' if bits & 8 else 'u'}{size}"))
        if cls == 1:
            order = '>' if bits & 1 else '<'
            if size not in (2, 4, 8):
                raise NotImplementedError(f'{size}-byte floating point')
            return _Type(cls, size, np.dtype(f'{order}f{size}'))
        if cls == 3:
            return _Type(cls, size, np.dtype(f'S{size}'), pad=bits & 0x0f)
        if cls == 9:
            if bits & 0x0f != 1:
                raise NotImplementedError('variable-length sequences')
            return _Type(cls, size, np.dtype(object), vlen_string=True)
        raise NotImplementedError(f'HDF5 datatype class {cls}')

    def _layout(self, d):
        ver = d[0]
        if ver in (3, 4):
            cls = d[1]
            if ver == 4 and cls == 2:
                raise NotImplementedError('version-4 chunk indexing (files written with libver="latest"): re-save with h5py defaults')
            if cls == 0:
                n = struct.unpack_from('<H', d, 2)[0]
                return ('compact', bytes(d[4:4 + n]))
            if cls == 1:
                return ('contiguous',) + struct.unpack_from('<QQ', d, 2)
            if cls == 2:
                nd = d[2]
                bt = struct.unpack_from('<Q', d, 3)[0]
                dims = struct.unpack_from(f'<{nd}I', d, 11)
                return ('chunked', bt, dims[:-1])
        elif ver in (1, 2):
            nd, cls = d[1], d[2]
            p = 8
            if cls != 0:
                a = struct.unpack_from('<Q', d, p)[0]; p += 8
            dims = struct.unpack_from(f'<{nd}I', d, p); p += 4 * nd
            if cls == 1:
                return ('contiguous', a, None)
            if cls == 2:
                return ('chunked', a, dims[:-1])
            n = struct.unpack_from('<I', d, p)[0]
            return ('compact', bytes(d[p + 4:p + 4 + n]))
        raise NotImplementedError(f'data layout message version {ver} / class')

    def _filters(self, d):
        ver, n = d[0], d[1]
        p = 8 if ver == 1 else 2
        out = []
        for _ in range(n):
            fid = struct.unpack_from('<H', d, p)[0]
            if ver == 1 or fid >= 256:
                nlen = struct.unpack_from('<H', d, p + 2)[0]
                ncv = struct.unpack_from('<H', d, p + 6)[0]
                p += 8 + nlen + (-nlen % 8 if ver == 1 else 0)
            else:
                ncv = struct.unpack_from('<H', d, p + 4)[0]
                p += 6
            cv = struct.unpack_from(f'<{ncv}I', d, p)
            p += 4 * ncv
            if ver == 1 and ncv % 2:
                p += 4
            out.append((fid, cv))
        return out

    def _attribute(self, d):
        ver = d[0]
        nsz, tsz, ssz = struct.unpack_from('<HHH', d, 2)
        p = 8
        if ver == 3:
            p += 1
        pad = (lambda n: n + (-n % 8)) if ver == 1 else (lambda n: n)
        if ver > 1 and d[1] & 0x03:
            raise NotImplementedError('attributes with shared datatype / dataspace')
        name = bytes(d[p:p + nsz]).split(b'\x00')[0].decode('utf8'); p += pad(nsz)
        t = self._datatype(d[p:p + tsz]); p += pad(tsz)
        shape = self._dataspace(d[p:p + ssz]); p += pad(ssz)
        return name, self._decode(d[p:], t, shape, scalar_unwrap=True)

    # ---- raw data
    def _vlen_string(self, raw):
        length, gaddr, idx = struct.unpack_from('<IQI', raw, 0)
        if length == 0 or gaddr in (0, UNDEF):
            return ''
        b = self._buf
        if b[gaddr:gaddr + 4] != b'GCOL':
            raise ValueError('bad global heap collection')
        size = self._addr(gaddr + 8)
        p, end = gaddr + 16, gaddr + size
        while p + 16 <= end:
            oidx, osz = self._u(p, 2), self._addr(p + 8)
            if oidx == 0:
                break
            if oidx == idx:
                return bytes(b[p + 16:p + 16 + length]).decode('utf8', 'replace')
            p += 16 + osz + (-osz % 8)
        raise ValueError('global heap object not found')

    def _decode(self, raw, t, shape, scalar_unwrap=False):
        if shape is None:
            return None
        count = int(np.prod(shape)) if shape else 1
        if t.vlen_string:
            vals = [self._vlen_string(raw[i * t.size:(i + 1) * t.size]) for i in range(count)]
            arr = np.array(vals, dtype=object).reshape(shape)
            return vals[0] if (scalar_unwrap and not shape) else arr
        arr = np.frombuffer(bytes(raw[:count * t.size]), dtype=t.dtype, count=count).reshape(shape)
        if scalar_unwrap and not shape:
            v = arr[()]
            return bytes(v) if t.cls == 3 else v
        return arr

    def _read_dataset(self, ds):
        t, shape, lay = ds._type, ds.shape, ds._layout
        if shape is None:
            return None
        nbytes = (int(np.prod(shape)) if shape else 1) * t.size
        if lay[0] == 'compact':
            return self._decode(lay[1], t, shape).copy()
        if lay[0] == 'contiguous':
            addr = lay[1]
            if addr == UNDEF:                                # never written: fill value zero
                return np.zeros(shape, dtype=t.dtype)
            return self._decode(self._buf[addr:addr + nbytes], t, shape).copy()
        # chunked: v1 B-tree of raw-data chunks
        bt, cdims = lay[1], lay[2]
        if t.vlen_string:
            raise NotImplementedError('chunked variable-length strings')
        out = np.zeros(shape, dtype=t.dtype)
        if bt == UNDEF:
            return out
        for offs, caddr, csize, fmask in self._chunk_btree(bt, len(cdims)):
            raw = bytes(self._buf[caddr:caddr + csize])
            for k in range(len(ds._filters) - 1, -1, -1):                      # undo the pipeline in reverse order
                if fmask & (1 << k):
                    continue
                fid, cv = ds._filters[k]
                if fid == 1:
                    raw = zlib.decompress(raw)
                elif fid == 2:
                    es = cv[0] if cv else t.size
                    n = len(raw) // es
                    raw = np.frombuffer(raw[:n * es], np.uint8).reshape(es, n).T.tobytes() + raw[n * es:]
                elif fid == 3:
                    raw = raw[:-4]                                              # fletcher32 checksum (not verified)
                else:
                    raise NotImplementedError(f'HDF5 filter id {fid}')
            chunk = np.frombuffer(raw, dtype=t.dtype, count=int(np.prod(cdims))).reshape(cdims)
            sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, shape))
            out[sl] = chunk[tuple(slice(0, s.stop - s.start) for s in sl)]
        return out

    def _chunk_btree(self, addr, nd):
        b = self._buf
        if b[addr:addr + 4] != b'TREE' or b[addr + 4] != 1:
            raise ValueError('bad chunk B-tree node')
        level, used = b[addr + 5], self._u(addr + 6, 2)
        p = addr + 8 + 2 * self.O
        ksz = 8 + 8 * (nd + 1)
        for _ in range(used):
            csize, fmask = struct.unpack_from('<II', b, p)
            offs = struct.unpack_from(f'<{nd}Q', b, p + 8)
            child = self._addr(p + ksz)
            p += ksz + self.O
            if level == 0:
                yield offs, child, csize, fmask
            else:
                yield from self._chunk_btree(child, nd)


# ------------------------------------------------------------------ Keras layouts
def _chunked_attr(attrs, name):
    """Keras splits attributes above 64 KB into name0, name1, ... (save_attributes_to_hdf5_group)."""
    if name in attrs:
        v = attrs[name]
    else:
        parts, i = [], 0
        while f'{name}{i}' in attrs:
            parts.append(np.asarray(attrs[f'{name}{i}'])); i += 1
        if not parts:
            raise KeyError(f'attribute {name} not found: not a Keras weight file / group')
        v = np.concatenate(parts)
    return [x.decode('utf8') if isinstance(x, bytes) else str(x) for x in np.asarray(v).reshape(-1).tolist()]


def read_keras_weights(path):
    """[(layer name, [(weight name, float array)])] in the file's layer order (`layer_names` attribute) -- the order
    tf.keras' load_weights_from_hdf5_group walks.  Accepts weight files (root holds the layers) and full-model files
    (`/model_weights`)."""
    with File(path) as f:
        g = f['model_weights'] if 'model_weights' in f._links and 'layer_names' not in f.attrs else f
        out = []
        for lname in _chunked_attr(g.attrs, 'layer_names'):
            lg = g[lname]
            ws = [(wn, np.asarray(lg[wn].read())) for wn in (_chunked_attr(lg.attrs, 'weight_names') if len(lg.attrs.get('weight_names', [])) or 'weight_names0' in lg.attrs else [])]
            out.append((lname, ws))
        return out


# ------------------------------------------------------------------ writer (Keras weight files)
class _Node:
    """In-memory group: attrs {name: bytes | ndarray | float}, children {name: _Node | ndarray}."""

    def __init__(self):
        self.attrs, self.children = {}, {}

    def group(self, path):
        node = self
        for part in [p for p in path.split('/') if p]:
            node = node.children.setdefault(part, _Node())
        return node

    def dataset(self, path, array):
        parts = [p for p in path.split('/') if p]
        self.group('/'.join(parts[:-1])).children[parts[-1]] = np.asarray(array)


_LEAF_K, _FLOAT32 = 4, bytes.fromhex('11201f000400000000002000170800177f000000')
_DTYPES = {np.dtype('<f4'): _FLOAT32,
           np.dtype('<f8'): bytes.fromhex('11203f000800000000004000340b0034ff030000'),
           np.dtype('<i8'): bytes.fromhex('1008000008000000' + '00004000'), np.dtype('<i4'): bytes.fromhex('1008000004000000' + '00002000')}


def _pad8(b):
    return b + b'\x00' * (-len(b) % 8)


def _space_msg(shape):
    return struct.pack('<BBBB4x', 1, len(shape), 0, 0) + b''.join(struct.pack('<Q', int(s)) for s in shape)


def _type_msg(dt):
    dt = np.dtype(dt)
    if dt.kind == 'S':
        return struct.pack('<BBBBI', 0x13, 0x01, 0, 0, dt.itemsize)          # fixed-length string, null-padded, ASCII
    le = dt.newbyteorder('<') if dt.itemsize > 1 else dt
    if le not in _DTYPES:
        raise NotImplementedError(f'writing dtype {dt}')
    return _DTYPES[le]


class _Writer:
    def __init__(self):
        self.buf = bytearray(96)                           # superblock v0 + root symbol table entry, filled in at the end
        self.max_snods = 1

    def alloc(self, data):
        self.buf += b'\x00' * (-len(self.buf) % 8)
        addr = len(self.buf)
        self.buf += data
        return addr

    def _messages(self, msgs):
        """version-1 object header; continuation blocks are not needed (every message < 64 KiB, total size is a 32-bit field)."""
        body = b''
        for mtype, flags, data in msgs:
            data = _pad8(data)
            if len(data) > 0xFFF8:
                raise ValueError('object header message above 64 KiB (Keras splits such attributes: see write_keras_weights)')
            body += struct.pack('<HHB3x', mtype, len(data), flags) + data
        return struct.pack('<BBHII4x', 1, 0, len(msgs), 1, len(body)) + body

    @staticmethod
    def _attr_msg(name, value):
        if isinstance(value, (bytes, str)):
            raw = value.encode('utf8') if isinstance(value, str) else value
            arr = np.array(raw if raw else b'\x00', dtype=f'S{max(len(raw), 1)}')
        else:
            arr = np.asarray(value)
            if arr.dtype.kind == 'U':
                arr = np.char.encode(arr, 'utf8')
            if arr.dtype.kind == 'S' and arr.dtype.itemsize == 0:
                arr = arr.astype('S1')
            if arr.dtype.kind == 'f' and arr.dtype != np.float32:
                arr = arr.astype('<f8')
        nm = name.encode('utf8') + b'\x00'
        t, s = _type_msg(arr.dtype), _space_msg(arr.shape)
        raw = arr.astype(arr.dtype.newbyteorder('<') if arr.dtype.kind in 'iuf' else arr.dtype).tobytes(order='C')
        return (0x0C, 0, struct.pack('<BBHHH', 1, 0, len(nm), len(t), len(s)) + _pad8(nm) + _pad8(t) + _pad8(s) + raw)

    def dataset(self, arr):
        arr = np.asarray(arr)
        if arr.dtype.kind == 'f' and arr.dtype.itemsize not in (4, 8):
            arr = arr.astype(np.float32)
        arr = arr.astype(arr.dtype.newbyteorder('<'))
        raw = arr.tobytes(order='C')
        daddr = self.alloc(raw) if raw else UNDEF
        msgs = [(0x01, 0, _space_msg(arr.shape)), (0x03, 1, _type_msg(arr.dtype)), (0x05, 1, bytes.fromhex('0202020100000000')),
                (0x08, 0, struct.pack('<BBQQ', 3, 1, daddr, len(raw)))]
        return self.alloc(self._messages(msgs))

    def group(self, node, internal_k):
        """-> (object header address, B-tree address, heap address)."""
        names = sorted(node.children, key=lambda s: s.encode('utf8'))
        addrs = {}
        for n in names:
            c = node.children[n]
            addrs[n] = self.group(c, internal_k)[0] if isinstance(c, _Node) else self.dataset(c)
        # local heap: offset 0 = empty string, then the names, then one free block
        heap, offs = bytearray(8), {}
        for n in names:
            offs[n] = len(heap)
            heap += _pad8(n.encode('utf8') + b'\x00')
        free_at = len(heap)
        heap += struct.pack('<QQ', 1, 32) + b'\x00' * 16          # free block: next = H5HL_FREE_NULL (1), size 32
        haddr_data = self.alloc(bytes(heap))
        haddr = self.alloc(b'HEAP' + struct.pack('<B3xQQQ', 0, len(heap), free_at, haddr_data))
        # symbol table nodes of <= 2 * leaf K entries, one B-tree node above them
        per = 2 * _LEAF_K
        chunks = [names[i:i + per] for i in range(0, len(names), per)]
        self.max_snods = max(self.max_snods, len(chunks))
        keys, kids = [0], []
        for ch in chunks:
            ent = b''.join(struct.pack('<QQII16x', offs[n], addrs[n], 0, 0) for n in ch)
            ent += b'\x00' * (40 * (per - len(ch)))
            kids.append(self.alloc(b'SNOD' + struct.pack('<BBH', 1, 0, len(ch)) + ent))
            keys.append(offs[ch[-1]])
        body = b''
        for i in range(2 * internal_k):
            body += struct.pack('<Q', keys[i] if i < len(keys) else 0)
            body += struct.pack('<Q', kids[i] if i < len(kids) else 0)
        body += struct.pack('<Q', keys[2 * internal_k] if 2 * internal_k < len(keys) else 0)
        baddr = self.alloc(b'TREE' + struct.pack('<BBHQQ', 0, 0, len(kids), UNDEF, UNDEF) + body)
        msgs = [(0x11, 0, struct.pack('<QQ', baddr, haddr))] + [self._attr_msg(k, v) for k, v in node.attrs.items()]
        return self.alloc(self._messages(msgs)), baddr, haddr


def _count_snods(node):
    n = (len(node.children) + 2 * _LEAF_K - 1) // (2 * _LEAF_K)
    return max([n] + [_count_snods(c) for c in node.children.values() if isinstance(c, _Node)])


def write_file(path, root):
    """Serialize a _Node tree as an HDF5 file in the structures h5py writes by default (superblock v0, version-1 object headers,
    symbol-table groups, contiguous datasets, fixed-length string attributes)."""
    internal_k = max(16, (_count_snods(root) + 1) // 2)
    w = _Writer()
    oaddr, baddr, haddr = w.group(root, internal_k)
    w.buf += b'\x00' * (-len(w.buf) % 8)
    sb = SIGNATURE + struct.pack('<BBBBBBBBHHI', 0, 0, 0, 0, 0, 8, 8, 0, _LEAF_K, internal_k, 0)
    sb += struct.pack('<QQQQ', 0, UNDEF, len(w.buf), UNDEF)
    sb += struct.pack('<QQII', 0, oaddr, 1, 0) + struct.pack('<QQ', baddr, haddr)
    w.buf[:96] = sb
    with open(path, 'wb') as f:
        f.write(bytes(w.buf))


_ATTR_LIMIT = 64512            # Keras' HDF5_OBJECT_HEADER_LIMIT: larger string-array attributes are split into name0, name1, ...


def _names_attr(node, name, values):
    arr = np.array([v.encode('utf8') for v in values]) if values else np.array([], dtype='S1')
    if arr.nbytes <= _ATTR_LIMIT:
        node.attrs[name] = arr
        return
    nchunks = 1
    while any(c.nbytes > _ATTR_LIMIT for c in np.array_split(arr, nchunks)):
        nchunks += 1
    for i, c in enumerate(np.array_split(arr, nchunks)):
        node.attrs[f'{name}{i}'] = c


def write_keras_weights(path, layers, root_attrs=None, model_weights_group=False, extra_groups=None):
    """The layout of tf.keras `save_weights('x.h5')` (or, with model_weights_group, of `save('x.h5')`): `layers` =
    [(layer name, [(weight name, array)])] in model order; root_attrs are bytes / str / arrays; extra_groups = {group path:
    [(dataset path, array)]} (e.g. optimizer_weights)."""
    root = _Node()
    for k, v in (root_attrs or {}).items():
        root.attrs[k] = v
    g = root.group('model_weights') if model_weights_group else root
    _names_attr(g, 'layer_names', [ln for ln, _ in layers])
    g.attrs['backend'] = b'tensorflow'
    g.attrs.setdefault('keras_version', b'2.6.0')
    for ln, ws in layers:
        lg = g.group(ln)
        _names_attr(lg, 'weight_names', [wn for wn, _ in ws])
        for wn, arr in ws:
            lg.dataset(wn, arr)
    for gp, items in (extra_groups or {}).items():
        eg = root.group(gp)
        _names_attr(eg, 'weight_names', [wn for wn, _ in items])
        for wn, arr in items:
            eg.dataset(wn, arr)
    write_file(path, root)
