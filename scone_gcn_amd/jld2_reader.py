"""Minimal pure-Python reader for the HDF5 subset that JLD2 0.2 files such as the reference's
`ocean_drifters_data/dataBuoys.jld2` use (no h5py in this image; the reference reads it with h5py, BD:7-33).

Supported: superblock v2, version-2 object headers (+ continuation blocks), link messages in compact groups,
dataspace v1/v2, datatypes fixed-point / float / reference (other classes are returned as raw bytes), compact and
contiguous data layouts.  Object references are 8-byte file addresses (relative to the superblock base).
"""
import struct

import numpy as np


class JLD2File:
    def __init__(self, path):
        with open(path, "rb") as f:
            self.buf = f.read()
        sig = self.buf.find(b"\x89HDF\r\n\x1a\n")
        if sig < 0:
            raise ValueError("not an HDF5 file")
        sb = self.buf[sig:]
        if sb[8] not in (2, 3) or sb[9] != 8 or sb[10] != 8:
            raise ValueError("only superblock v2/v3 with 8-byte offsets is supported")
        self.base, _, _, self.root = struct.unpack("<QQQQ", sb[12:44])
        self.root_links = self.links(self.root)

    # ---- object headers -------------------------------------------------------------------------------
    def _messages(self, addr):
        b, p = self.buf, self.base + addr
        if b[p:p + 4] != b"OHDR" or b[p + 4] != 2:
            raise ValueError("only version-2 object headers are supported (at %d)" % addr)
        flags = b[p + 5]
        q = p + 6
        if flags & 0x20:
            q += 16
        if flags & 0x10:
            q += 4
        nsz = 1 << (flags & 3)
        size = int.from_bytes(b[q:q + nsz], "little")
        q += nsz
        chunks = [(q, q + size)]
        out = []
        while chunks:
            q, end = chunks.pop(0)
            while q + 4 <= end:
                mtype = b[q]
                msize = struct.unpack("<H", b[q + 1:q + 3])[0]
                q += 4
                if flags & 4:
                    q += 2
                data = b[q:q + msize]
                q += msize
                if mtype == 0x10:                                   # continuation -> OCHK block
                    caddr, clen = struct.unpack("<QQ", data[:16])
                    cp = self.base + caddr
                    if b[cp:cp + 4] != b"OCHK":
                        raise ValueError("bad continuation block")
                    chunks.append((cp + 4, cp + clen - 4))
                elif mtype != 0:
                    out.append((mtype, data))
        return out

    def links(self, addr):
        res = {}
        for t, d in self._messages(addr):
            if t != 0x06:
                continue
            flags = d[1]
            q = 2
            ltype = 0
            if flags & 8:
                ltype = d[q]; q += 1
            if flags & 4:
                q += 8
            if flags & 0x10:
                q += 1
            nsz = 1 << (flags & 3)
            nlen = int.from_bytes(d[q:q + nsz], "little"); q += nsz
            name = d[q:q + nlen].decode(); q += nlen
            if ltype == 0:
                res[name] = struct.unpack("<Q", d[q:q + 8])[0]
        return res

    # ---- datasets --------------------------------------------------------------------------------------
    def read(self, addr):
        """Dataset at object-header address -> (numpy array | raw bytes, datatype class)."""
        dims, dt, raw = (), None, None
        for t, d in self._messages(addr):
            if t == 0x01:
                ver, rank = d[0], d[1]
                off = 8 if ver == 1 else 4
                dims = struct.unpack("<%dQ" % rank, d[off:off + 8 * rank]) if rank else ()
            elif t == 0x03:
                cls, size = d[0] & 0x0F, struct.unpack("<I", d[4:8])[0]
                signed = bool(d[1] & 0x08)
                dt = (cls, size, signed)
            elif t == 0x08:
                ver, lcls = d[0], d[1]
                if ver not in (3, 4):
                    raise ValueError("layout version %d" % ver)
                if lcls == 0:
                    n = struct.unpack("<H", d[2:4])[0]
                    raw = d[4:4 + n]
                elif lcls == 1:
                    a, n = struct.unpack("<QQ", d[2:18])
                    raw = b"" if a == 0xFFFFFFFFFFFFFFFF else self.buf[self.base + a:self.base + a + n]
                else:
                    raise ValueError("chunked layout is not supported")
        if dt is None or raw is None:
            raise ValueError("object at %d is not a simple dataset" % addr)
        cls, size, signed = dt
        n = int(np.prod(dims)) if dims else 1
        if cls == 0:
            arr = np.frombuffer(raw[:n * size], dtype=np.dtype("<%s%d" % ("i" if signed else "u", size)))
        elif cls == 1:
            arr = np.frombuffer(raw[:n * size], dtype=np.dtype("<f%d" % size))
        elif cls == 7:
            arr = np.frombuffer(raw[:n * 8], dtype="<u8")
        else:
            return raw, cls
        # HDF5 dims are slowest-first; JLD2 writes Julia (column-major) arrays with reversed dims
        return arr.reshape(dims) if dims else arr.reshape(()), cls

    def __getitem__(self, name):
        return self.read(self.root_links[name])

    def keys(self):
        return list(self.root_links)
