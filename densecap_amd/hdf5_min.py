"""Minimal HDF5 writer/reader for `extract_features.lua`'s output (`/feats` N x M x D, `/boxes` N x M x 4 float32,
extract_features.lua:92-96) -- the image has no h5py.

Writes the classic, universally readable layout of the HDF5 file-format specification (version 1.8 "III. Disk Format"):
superblock v0; root group = v1 object header with a Symbol Table message -> v1 B-tree (one leaf) -> one symbol-table
node (SNOD) + local heap with the link names; every dataset = v1 object header {Dataspace v1, Datatype (IEEE little
endian float / fixed-point), Fill Value v2, Data Layout v3 contiguous} followed by its raw data, C order.  No chunking,
compression, attributes or nested groups.  Files are checked against libhdf5 itself in tests/test_hdf5.py when the
library is present (it is in this image, under /opt/conda/lib), and with the small reader below everywhere.
"""
from __future__ import annotations

import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
SIGNATURE = b"\x89HDF\r\n\x1a\n"
LEAF_K, INTERNAL_K = 4, 16          # group B-tree parameters stored in the superblock


def _pad8(b):
    return b + b"\x00" * (-len(b) % 8)


def _message(mtype, data, flags=0):
    data = _pad8(data)
    return struct.pack("<HHB3x", mtype, len(data), flags) + data


def _datatype_message(dt):
    dt = np.dtype(dt)
    if dt == np.float32:
        # class 1 (floating point), version 1; bit field: little endian, mantissa normalisation = implied msb (2 << 4),
        # sign bit position 31; properties: bit offset 0, precision 32, exponent at 23 (8 bits), mantissa at 0 (23 bits), bias 127
        return struct.pack("<BBBBI", 0x11, 0x20, 31, 0, 4) + struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
    if dt == np.float64:
        return struct.pack("<BBBBI", 0x11, 0x20, 63, 0, 8) + struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
    if dt.kind in "iu" and dt.itemsize in (1, 2, 4, 8):
        # class 0 (fixed point): bit 3 of the first bit-field byte = signed; properties: bit offset, precision
        return struct.pack("<BBBBI", 0x10, 0x08 if dt.kind == "i" else 0x00, 0, 0, dt.itemsize) + struct.pack("<HH", 0, 8 * dt.itemsize)
    raise TypeError("hdf5_min: unsupported dtype %s" % dt)


def _object_header(messages):
    body = b"".join(messages)
    # v1 prefix: version, reserved, #messages, reference count, header size, then 4 bytes of padding to an 8-byte boundary
    return struct.pack("<BBHII4x", 1, 0, len(messages), 1, len(body)) + body


def write_hdf5(path, datasets):
    """datasets: {name: ndarray} (float32/float64/ints), written as contiguous datasets of the root group."""
    names = sorted(datasets)                    # symbol-table entries are ordered by name
    if not names or len(names) > 2 * LEAF_K:
        raise ValueError("hdf5_min: between 1 and %d datasets" % (2 * LEAF_K))
    arrays = {}
    for n in names:
        if "/" in n or not n:
            raise ValueError("hdf5_min: flat dataset names only")
        a = np.asarray(datasets[n])
        arrays[n] = np.ascontiguousarray(a.astype(a.dtype.newbyteorder("<"), copy=False))
    # ---- local heap data: offset 0 = "" (the B-tree's left-most key), then the names, then one free block ----------
    heap = bytearray(b"\x00" * 8)
    name_off = {}
    for n in names:
        name_off[n] = len(heap)
        heap += _pad8(n.encode("ascii") + b"\x00")
    free_off = len(heap)
    heap += struct.pack("<QQ", 1, 32) + b"\x00" * 16      # free block: next = 1 (H5HL_FREE_NULL), size 32
    # ---- addresses -----------------------------------------------------------------------------------------------------
    sb_size = 96
    root_oh = _object_header([_message(0x0011, struct.pack("<QQ", 0, 0))])     # patched below
    addr_root = sb_size
    addr_btree = addr_root + len(root_oh)
    btree_size = 24 + (2 * INTERNAL_K + 1) * 8 + 2 * INTERNAL_K * 8
    addr_heap = addr_btree + btree_size
    addr_heap_data = addr_heap + 32
    addr_snod = addr_heap_data + len(heap)
    snod_size = 8 + 2 * LEAF_K * 40
    cur = addr_snod + snod_size
    ds_headers, ds_addr = {}, {}
    for n in names:
        a = arrays[n]
        dataspace = struct.pack("<BBB5x", 1, a.ndim, 0) + b"".join(struct.pack("<Q", s) for s in a.shape)
        fill = struct.pack("<BBBB", 2, 1, 0, 0)             # v2: allocate early, write fill at allocation, none defined
        def header(data_addr, a=a, dataspace=dataspace, fill=fill):
            layout = struct.pack("<BBQQ", 3, 1, data_addr, a.nbytes)
            return _object_header([_message(0x0001, dataspace), _message(0x0003, _datatype_message(a.dtype), flags=1),
                                   _message(0x0005, fill), _message(0x0008, layout)])
        hsize = len(header(0))
        ds_addr[n] = (cur, cur + hsize)
        ds_headers[n] = header(cur + hsize)
        cur += hsize + a.nbytes + (-a.nbytes % 8)
    eof = cur
    # ---- assemble --------------------------------------------------------------------------------------------------------
    root_entry = struct.pack("<QQII", 0, addr_root, 1, 0) + struct.pack("<QQ", addr_btree, addr_heap)
    superblock = (SIGNATURE + struct.pack("<BBBBBBBB", 0, 0, 0, 0, 0, 8, 8, 0) + struct.pack("<HHI", LEAF_K, INTERNAL_K, 0) +
                  struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF) + root_entry)
    assert len(superblock) == sb_size
    root_oh = _object_header([_message(0x0011, struct.pack("<QQ", addr_btree, addr_heap))])
    # B-tree leaf: node type 0 (group), level 0, one child; key0 = "" (heap offset 0), key1 = the largest name in the child
    btree = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1, UNDEF, UNDEF) + struct.pack("<QQQ", 0, addr_snod, name_off[names[-1]])
    btree += b"\x00" * (btree_size - len(btree))
    heap_hdr = b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), free_off, addr_heap_data)
    snod = b"SNOD" + struct.pack("<BBH", 1, 0, len(names))
    for n in names:
        snod += struct.pack("<QQII16x", name_off[n], ds_addr[n][0], 0, 0)
    snod += b"\x00" * (snod_size - len(snod))
    with open(path, "wb") as f:
        f.write(superblock); f.write(root_oh); f.write(btree); f.write(heap_hdr); f.write(bytes(heap)); f.write(snod)
        for n in names:
            assert f.tell() == ds_addr[n][0]
            f.write(ds_headers[n])
            f.write(arrays[n].tobytes())
            f.write(b"\x00" * (-arrays[n].nbytes % 8))
        assert f.tell() == eof
    return path


def read_hdf5(path):
    """Reads back files of the layout above (superblock v0, one symbol-table node, contiguous datasets) -> {name: ndarray}."""
    buf = open(path, "rb").read()
    if buf[:8] != SIGNATURE or buf[8] != 0:
        raise ValueError("hdf5_min: not a superblock-v0 HDF5 file")
    addr_btree, addr_heap = struct.unpack_from("<QQ", buf, 56 + 24)
    if buf[addr_btree:addr_btree + 4] != b"TREE" or buf[addr_heap:addr_heap + 4] != b"HEAP":
        raise ValueError("hdf5_min: bad root group")
    _, _, heap_data = struct.unpack_from("<QQQ", buf, addr_heap + 8)
    level, nent = struct.unpack_from("<BH", buf, addr_btree + 5)
    if level != 0:
        raise ValueError("hdf5_min: multi-level group B-tree not supported")
    out = {}
    for c in range(nent):
        snod = struct.unpack_from("<Q", buf, addr_btree + 24 + 8 + c * 16)[0]
        if buf[snod:snod + 4] != b"SNOD":
            raise ValueError("hdf5_min: bad symbol table node")
        nsym = struct.unpack_from("<H", buf, snod + 6)[0]
        for i in range(nsym):
            noff, oh = struct.unpack_from("<QQ", buf, snod + 8 + i * 40)
            name = buf[heap_data + noff:buf.index(b"\x00", heap_data + noff)].decode("ascii")
            nmsg, = struct.unpack_from("<H", buf, oh + 2)
            p = oh + 16
            shape = dtype = addr = nbytes = None
            for _ in range(nmsg):
                mtype, msize = struct.unpack_from("<HH", buf, p)
                d = p + 8
                if mtype == 0x0001:
                    rank = buf[d + 1]
                    shape = struct.unpack_from("<%dQ" % rank, buf, d + 8)
                elif mtype == 0x0003:
                    cls, bits0 = buf[d] & 0x0F, buf[d + 1]
                    size = struct.unpack_from("<I", buf, d + 4)[0]
                    dtype = np.dtype("<f%d" % size) if cls == 1 else np.dtype("<%s%d" % ("i" if bits0 & 0x08 else "u", size))
                elif mtype == 0x0008:
                    if buf[d] != 3 or buf[d + 1] != 1:
                        raise ValueError("hdf5_min: only contiguous v3 layouts")
                    addr, nbytes = struct.unpack_from("<QQ", buf, d + 2)
                p = d + msize
            out[name] = np.frombuffer(buf, dtype, count=nbytes // dtype.itemsize, offset=addr).reshape(shape).copy()
    return out
