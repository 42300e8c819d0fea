"""Reader and writer for TensorFlow "bundle" checkpoints (``*.ckpt.index`` + ``*.ckpt.data-*``), without TensorFlow.

The reference saves/restores its weights with ``tf.train.Saver`` (reference ``train.py:280,309-315,611-617``;
``evaluate.py:264-268``).  TensorFlow is not available on MI355X boxes, so this module reads that on-disk
format directly: the ``.index`` file is a LevelDB-style sorted string table whose values are
``BundleEntryProto`` messages (dtype, shape, shard, offset, size, crc32c); the ``.data-00000-of-00001`` file is
the raw little-endian tensor payload.

Only what the EPC-Net checkpoints need is implemented: uncompressed blocks, DT_FLOAT / DT_INT32 / DT_INT64
tensors, a single data shard, no tensor slices.

``write_checkpoint`` is the inverse (``saver.save``, train.py:611-617): one data shard with the tensors in sorted name
order, an index table in the layout the shipped ``.index`` files have (one data block, restart interval 16, empty
metaindex, masked CRC-32C trailers -- the block checksums of those shipped files are this module's known-answer test
for the checksum), so the files restore with ``tf.train.Saver`` as well as with ``load_checkpoint``.

Known-answer data for this reader: ``tests/golden/ckpt_tables.json`` (generated from the reference's shipped
``exp/*/saved_model/*.ckpt.index`` files by ``scripts/make_ckpt_tables.py``).
"""
from __future__ import annotations

import os
import struct
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np

_TABLE_MAGIC = 0xDB4775248B80FB57
_FOOTER_LEN = 48
_BLOCK_TRAILER = 5  # 1 byte compression type + 4 bytes crc

# tensorflow/core/framework/types.proto
_DTYPES = {1: np.dtype("<f4"), 2: np.dtype("<f8"), 3: np.dtype("<i4"), 9: np.dtype("<i8"), 10: np.dtype(bool)}


@dataclass
class BundleEntry:
    name: str
    dtype: np.dtype
    shape: Tuple[int, ...]
    shard_id: int
    offset: int
    size: int
    crc32c: int

    @property
    def numel(self) -> int:
        n = 1
        for d in self.shape:
            n *= d
        return n


def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    result = 0
    shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _read_block(data: bytes, offset: int, size: int) -> bytes:
    block = data[offset:offset + size]
    ctype = data[offset + size]
    if ctype != 0:
        raise NotImplementedError("compressed table blocks (type %d) are not supported" % ctype)
    return block


def _block_entries(block: bytes) -> List[Tuple[bytes, bytes]]:
    num_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    limit = len(block) - 4 - 4 * num_restarts
    pos = 0
    key = b""
    out = []
    while pos < limit:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        out.append((key, block[pos:pos + vlen]))
        pos += vlen
    return out


def _parse_shape(buf: bytes) -> Tuple[int, ...]:
    dims = []
    pos = 0
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        field, wire = tag >> 3, tag & 7
        if wire == 2:
            ln, pos = _varint(buf, pos)
            sub = buf[pos:pos + ln]
            pos += ln
            if field == 2:  # TensorShapeProto.Dim
                size = 0
                sp = 0
                while sp < len(sub):
                    t2, sp = _varint(sub, sp)
                    if t2 & 7 == 0:
                        v, sp = _varint(sub, sp)
                        if t2 >> 3 == 1:
                            size = v
                    elif t2 & 7 == 2:
                        l2, sp = _varint(sub, sp)
                        sp += l2
                    else:
                        raise ValueError("unexpected wire type in Dim")
                dims.append(size)
        elif wire == 0:
            _, pos = _varint(buf, pos)
        else:
            raise ValueError("unexpected wire type in TensorShapeProto")
    return tuple(dims)


def _parse_entry(name: str, buf: bytes) -> BundleEntry:
    dtype = 0
    shape: Tuple[int, ...] = ()
    shard = offset = size = crc = 0
    pos = 0
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        field, wire = tag >> 3, tag & 7
        if wire == 0:
            v, pos = _varint(buf, pos)
            if field == 1:
                dtype = v
            elif field == 3:
                shard = v
            elif field == 4:
                offset = v
            elif field == 5:
                size = v
        elif wire == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
            if field == 6:
                crc = v
        elif wire == 2:
            ln, pos = _varint(buf, pos)
            if field == 2:
                shape = _parse_shape(buf[pos:pos + ln])
            elif field == 7:
                raise NotImplementedError("sliced tensors are not supported (%s)" % name)
            pos += ln
        else:
            raise ValueError("unexpected wire type %d in BundleEntryProto" % wire)
    if dtype not in _DTYPES:
        raise NotImplementedError("dtype enum %d of %s is not supported" % (dtype, name))
    return BundleEntry(name, _DTYPES[dtype], shape, shard, offset, size, crc)


def read_index(index_path: str) -> "OrderedDict[str, BundleEntry]":
    """Return ``{variable name: BundleEntry}`` in the table's (sorted) key order."""
    with open(index_path, "rb") as f:
        data = f.read()
    if len(data) < _FOOTER_LEN:
        raise ValueError("%s: too short for a table footer" % index_path)
    footer = data[-_FOOTER_LEN:]
    magic = struct.unpack_from("<Q", footer, _FOOTER_LEN - 8)[0]
    if magic != _TABLE_MAGIC:
        raise ValueError("%s: bad table magic %#x" % (index_path, magic))
    pos = 0
    _, pos = _varint(footer, pos)  # metaindex offset
    _, pos = _varint(footer, pos)  # metaindex size
    idx_off, pos = _varint(footer, pos)
    idx_size, pos = _varint(footer, pos)
    entries: "OrderedDict[str, BundleEntry]" = OrderedDict()
    for _, handle in _block_entries(_read_block(data, idx_off, idx_size)):
        boff, hp = _varint(handle, 0)
        bsize, hp = _varint(handle, hp)
        for key, value in _block_entries(_read_block(data, boff, bsize)):
            if key == b"":  # BundleHeaderProto
                continue
            name = key.decode("utf-8")
            entries[name] = _parse_entry(name, value)
    return entries


def data_path_for(prefix: str, shard: int = 0, num_shards: int = 1) -> str:
    return "%s.data-%05d-of-%05d" % (prefix, shard, num_shards)


def load_checkpoint(prefix: str, verify_crc: bool = True) -> Dict[str, np.ndarray]:
    """Load every tensor of ``<prefix>.index`` / ``<prefix>.data-00000-of-00001``.

    ``verify_crc``: check every payload against the masked CRC-32C its index entry records, as TensorFlow's BundleReader
    does -- a corrupted payload raises ValueError instead of loading silently.

    Mirrors the reference's refusal to run with an incomplete checkpoint (``evaluate.py:264-266``):
    raises FileNotFoundError when the data shard is missing (it is, for every checkpoint shipped in
    the reference tree -- see ``.MISSING_LARGE_BLOBS``).
    """
    entries = read_index(prefix + ".index")
    dpath = data_path_for(prefix)
    if not os.path.exists(dpath):
        raise FileNotFoundError("checkpoint payload %s is missing" % dpath)
    out: Dict[str, np.ndarray] = {}
    with open(dpath, "rb") as f:
        blob = f.read()
    for name, e in entries.items():
        if e.shard_id != 0:
            raise NotImplementedError("multi-shard checkpoints are not supported")
        raw = blob[e.offset:e.offset + e.size]
        if len(raw) != e.size or e.size != e.numel * e.dtype.itemsize:
            raise ValueError("%s: payload size mismatch" % name)
        if verify_crc and mask_crc(crc32c(raw)) != e.crc32c:
            raise ValueError("%s: payload checksum mismatch (corrupt checkpoint %s)" % (name, dpath))
        out[name] = np.frombuffer(raw, dtype=e.dtype).reshape(e.shape).copy()
    return out


def write_checkpoint_payload(prefix: str, tensors: Dict[str, np.ndarray]) -> None:
    """Write a ``.data-00000-of-00001`` payload laid out per an EXISTING ``<prefix>.index``.

    Used by the tests to round-trip seeded weights through the reference's own index tables.
    """
    entries = read_index(prefix + ".index")
    total = max(e.offset + e.size for e in entries.values())
    blob = bytearray(total)
    for name, e in entries.items():
        arr = np.asarray(tensors[name], dtype=e.dtype)
        if tuple(arr.shape) != e.shape:
            raise ValueError("%s: shape %s != %s" % (name, arr.shape, e.shape))
        blob[e.offset:e.offset + e.size] = arr.tobytes(order="C")
    with open(data_path_for(prefix), "wb") as f:
        f.write(bytes(blob))


# ---- writer ------------------------------------------------------------------------------------------------------------
_DTYPE_ENUM = {np.dtype("<f4"): 1, np.dtype("<f8"): 2, np.dtype("<i4"): 3, np.dtype("<i8"): 9, np.dtype(bool): 10}
_RESTART_INTERVAL = 16
_CRC_MASK_DELTA = 0xA282EAD8


def crc32c(data: bytes, crc: int = 0) -> int:
    """CRC-32C of ``data`` (host routine of libepcnet_hip.so: include/epcnet.h epc_crc32c)."""
    from . import lib as L
    return int(L.lib().epc_crc32c(crc, data, len(data)))


def mask_crc(crc: int) -> int:
    """tensorflow/core/lib/hash/crc32c.h Mask(): rotate right by 15 and add a constant."""
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + _CRC_MASK_DELTA) & 0xFFFFFFFF


def _put_varint(v: int) -> bytes:
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _build_block(items: List[Tuple[bytes, bytes]]) -> bytes:
    """LevelDB block: prefix-compressed entries, restart array, restart count."""
    buf = bytearray()
    restarts = []
    last = b""
    for n, (key, value) in enumerate(items):
        shared = 0
        if n % _RESTART_INTERVAL == 0:
            restarts.append(len(buf))
        else:
            m = min(len(last), len(key))
            while shared < m and last[shared] == key[shared]:
                shared += 1
        buf += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value)) + key[shared:] + value
        last = key
    if not restarts:
        restarts = [0]
    for r in restarts:
        buf += struct.pack("<I", r)
    buf += struct.pack("<I", len(restarts))
    return bytes(buf)


def _with_trailer(block: bytes) -> bytes:
    return block + b"\x00" + struct.pack("<I", mask_crc(crc32c(block + b"\x00")))


def _short_successor(key: bytes) -> bytes:
    """leveldb BytewiseComparator::FindShortSuccessor: the index key of the last data block."""
    for i, b in enumerate(key):
        if b != 0xFF:
            return key[:i] + bytes([b + 1])
    return key


def _entry_proto(dtype: np.dtype, shape: Tuple[int, ...], offset: int, size: int, masked_crc: int) -> bytes:
    """BundleEntryProto bytes as TensorFlow serialises them (proto3: zero-valued fields omitted)."""
    dt = np.dtype(dtype).newbyteorder("<") if np.dtype(dtype).byteorder == ">" else np.dtype(dtype)
    if dt not in _DTYPE_ENUM:
        raise NotImplementedError("dtype %s cannot be written" % dtype)
    out = bytearray()
    out += b"\x08" + _put_varint(_DTYPE_ENUM[dt])
    sh = bytearray()
    for d in shape:
        dim = b"\x08" + _put_varint(int(d))
        sh += b"\x12" + _put_varint(len(dim)) + dim
    out += b"\x12" + _put_varint(len(sh)) + bytes(sh)                 # TensorShapeProto (present even for scalars)
    if offset:
        out += b"\x20" + _put_varint(offset)
    out += b"\x28" + _put_varint(size)
    out += b"\x35" + struct.pack("<I", masked_crc)
    return bytes(out)


def build_index(entries: "OrderedDict[str, BundleEntry]") -> bytes:
    """The ``.index`` file for the given entries (``crc32c`` fields already masked, as read_index returns them)."""
    items: List[Tuple[bytes, bytes]] = [(b"", b"\x08\x01\x1a\x02\x08\x01")]   # BundleHeaderProto: 1 shard, version producer 1
    for name in sorted(entries.keys(), key=lambda s: s.encode("utf-8")):
        e = entries[name]
        items.append((name.encode("utf-8"), _entry_proto(e.dtype, e.shape, e.offset, e.size, e.crc32c)))
    data_block = _with_trailer(_build_block(items))
    meta_off = len(data_block)
    meta_block = _with_trailer(_build_block([]))
    index_off = meta_off + len(meta_block)
    handle = _put_varint(0) + _put_varint(len(data_block) - _BLOCK_TRAILER)
    index_block = _with_trailer(_build_block([(_short_successor(items[-1][0]), handle)]))
    footer = _put_varint(meta_off) + _put_varint(len(meta_block) - _BLOCK_TRAILER) + _put_varint(index_off) + \
        _put_varint(len(index_block) - _BLOCK_TRAILER)
    footer = footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", _TABLE_MAGIC)
    return data_block + meta_block + index_block + footer


def write_checkpoint(prefix: str, tensors: Dict[str, np.ndarray]) -> None:
    """Write ``<prefix>.index`` and ``<prefix>.data-00000-of-00001`` holding ``tensors`` (sorted by name)."""
    names = sorted(tensors.keys(), key=lambda s: s.encode("utf-8"))
    entries: "OrderedDict[str, BundleEntry]" = OrderedDict()
    offset = 0
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    with open(data_path_for(prefix), "wb") as f:
        for name in names:
            arr = np.asarray(tensors[name])
            raw = arr.tobytes(order="C")
            f.write(raw)
            entries[name] = BundleEntry(name, arr.dtype, tuple(arr.shape), 0, offset, len(raw), mask_crc(crc32c(raw)))
            offset += len(raw)
    with open(prefix + ".index", "wb") as f:
        f.write(build_index(entries))
