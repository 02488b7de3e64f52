"""TensorFlow "tensor bundle" (V2 checkpoint) reader / writer without TensorFlow.

The reference restores ``./checkpoints/FlowNetS/flownet-S.ckpt-0`` etc. through ``tf.train.Saver.restore``
(/root/reference src/net.py:566-569, src/flownet_s/test.py:15) and saves through the slim Saver
(net.py:1386-1392).  Those files are ``<prefix>.index`` + ``<prefix>.data-0000i-of-0000N``: the arithmetic-free
container format of TensorFlow 1.x/2.x (third-party, not vendored in the reference tree, not installed here), restated
from its published definition:

* ``.index`` is a LevelDB-format sorted string table (tensorflow/core/lib/io/table*.cc = leveldb/table/format.cc):
  blocks of prefix-compressed entries ``varint32 shared | varint32 non_shared | varint32 value_len | key delta |
  value``, a restart array (uint32 LE offsets + uint32 count), a 5-byte trailer per block (compression type, masked
  CRC-32C of contents + type), and a 48-byte footer (metaindex handle, index handle, padding, magic
  0xdb4775248b80fb57 little endian).  Key "" holds a ``BundleHeaderProto`` (num_shards = 1, endianness = 2,
  version = 3), every other key is a variable name holding a ``BundleEntryProto`` (dtype = 1, shape = 2, shard_id =
  3, offset = 4, size = 5, crc32c = 6 fixed32, slices = 7) -- tensorflow/core/protobuf/tensor_bundle.proto.
* ``.data-*`` shards are the raw little-endian tensor bytes at (offset, size).

No golden checkpoint exists offline (checkpoints/download.sh needs the network), so parity of this reader is
pinned only by the format's own known answers (CRC-32C check value, footer magic, varint vectors) and by
write -> read round trips (tests/test_tf_checkpoint.py).  Partitioned variables (``slices``) are not produced by the
reference's single-device Saver and are rejected loudly.
"""
import os
import struct

import numpy as np

TABLE_MAGIC = 0xDB4775248B80FB57
_MASK_DELTA = 0xA282EAD8

# tensorflow/core/framework/types.proto
_DTYPES = {1: np.dtype("<f4"), 2: np.dtype("<f8"), 3: np.dtype("<i4"), 4: np.dtype("u1"), 5: np.dtype("<i2"),
           6: np.dtype("i1"), 9: np.dtype("<i8"), 10: np.dtype("?"), 17: np.dtype("<u2"), 19: np.dtype("<f2"),
           22: np.dtype("<u4"), 23: np.dtype("<u8")}
_DTYPE_CODES = {np.dtype(v).newbyteorder("="): k for k, v in _DTYPES.items()}
_DT_BFLOAT16 = 14


# ---- CRC-32C (Castagnoli), the checksum of the table format and of every tensor -------------------------------
def _crc_table():
    poly = 0x82F63B78
    t = np.zeros(256, dtype=np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ poly if c & 1 else c >> 1
        t[i] = c
    return t


_CRC_T = _crc_table()
_CRC_L = [int(x) for x in _CRC_T]


def crc32c(data, crc=0):
    """CRC-32C of ``data`` (bytes-like); check value crc32c(b"123456789") == 0xE3069283."""
    c = crc ^ 0xFFFFFFFF
    tab = _CRC_L
    for b in bytes(data):
        c = tab[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _gf2_apply(cols, r):
    """cols[i] = image of bit i under a GF(2)-linear map of 32-bit registers; r: uint32 array."""
    out = np.zeros_like(r)
    for i in range(32):
        out ^= np.where((r >> np.uint32(i)) & np.uint32(1), cols[i], np.uint32(0)).astype(np.uint32)
    return out


def crc32c_lanes(data, lanes=4096):
    """CRC-32C of a large buffer in NumPy, for hosts without the HIP library: the buffer is cut into `lanes`
    contiguous pieces whose register updates run side by side (one table look-up per byte position, vectorised over
    the lanes; the update is GF(2)-linear in the register, so every lane but the first starts from 0), and the lane
    registers are folded pairwise with the 'append m zero bytes' operator, squared per round.  Tens of MB/s against
    ~1 MB/s of the byte loop; same value as crc32c()."""
    buf = np.frombuffer(memoryview(data), dtype=np.uint8)
    n = buf.size
    m = n // lanes
    if m < 16:
        return crc32c(data)
    body = buf[:lanes * m].reshape(lanes, m)
    reg = np.zeros(lanes, np.uint32)
    reg[0] = 0xFFFFFFFF
    tab = _CRC_T.astype(np.uint32)
    for j in range(m):
        reg = tab[(reg ^ body[:, j]) & np.uint32(0xFF)] ^ (reg >> np.uint32(8))
    # operator Z: one zero byte; cols of Z^m by square-and-multiply
    one = (np.uint32(1) << np.arange(32, dtype=np.uint32)).astype(np.uint32)
    z1 = tab[one & np.uint32(0xFF)] ^ (one >> np.uint32(8))
    op, sq, e = one.copy(), z1, m
    while e:
        if e & 1:
            op = _gf2_apply(sq, op)
        sq = _gf2_apply(sq, sq)
        e >>= 1
    while reg.size > 1:  # (left, right) -> Z^len(right) left ^ right; lanes is a power of two
        reg = _gf2_apply(op, reg[0::2]) ^ reg[1::2]
        op = _gf2_apply(op, op)
    c = int(reg[0])
    for b in bytes(buf[lanes * m:]):
        c = _CRC_L[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def mask_crc(crc):
    """leveldb/TF store crcs "masked": rotate right by 15 and add a constant (crc32c.h)."""
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + _MASK_DELTA) & 0xFFFFFFFF


def unmask_crc(masked):
    rot = (masked - _MASK_DELTA) & 0xFFFFFFFF
    return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


# ---- varints / protobuf wire format -----------------------------------------------------------------------------
def _put_varint(v):
    if v < 0:
        v += 1 << 64
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _get_varint(buf, pos):
    shift, v = 0, 0
    while True:
        if pos >= len(buf):
            raise ValueError("truncated varint")
        b = buf[pos]
        pos += 1
        v |= (b & 0x7F) << shift
        if not b & 0x80:
            return v, pos
        shift += 7
        if shift > 63:
            raise ValueError("varint longer than 64 bits")


def _pb_fields(buf):
    """Yield (field number, wire type, value) of one protobuf message; value = int (varint, fixed) or bytes."""
    pos = 0
    while pos < len(buf):
        tag, pos = _get_varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            if len(v) != n:
                raise ValueError("truncated length-delimited field")
            pos += n
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield field, wt, v


def _pb_varint_field(field, v):
    return _put_varint(field << 3) + _put_varint(v)


def _pb_bytes_field(field, b):
    return _put_varint((field << 3) | 2) + _put_varint(len(b)) + b


# ---- snappy (block type 1; TF's BundleWriter writes uncompressed blocks, other writers may not) -----------------
def _snappy_decompress(buf):
    n, pos = _get_varint(buf, 0)
    out = bytearray()
    while pos < len(buf):
        tag = buf[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], "little")
                pos += nb
            ln += 1
            out += buf[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | buf[pos]
            pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = buf[pos] | (buf[pos + 1] << 8)
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], "little")
            pos += 4
        if off == 0 or off > len(out):
            raise ValueError("corrupt snappy block")
        for _ in range(ln):  # copies may overlap their own output
            out.append(out[-off])
    if len(out) != n:
        raise ValueError("snappy: length mismatch")
    return bytes(out)


# ---- table reading ------------------------------------------------------------------------------------------------
def _read_block(buf, offset, size, verify=True):
    raw = buf[offset:offset + size]
    trailer = buf[offset + size:offset + size + 5]
    if len(raw) != size or len(trailer) != 5:
        raise ValueError("table block [%d, +%d) outside the index file" % (offset, size))
    if verify and unmask_crc(struct.unpack("<I", trailer[1:])[0]) != crc32c(raw + trailer[:1]):
        raise ValueError("table block at %d: CRC-32C mismatch (corrupt .index file)" % offset)
    if trailer[0] == 0:
        return raw
    if trailer[0] == 1:
        return _snappy_decompress(raw)
    raise ValueError("table block compression type %d" % trailer[0])


def _block_entries(block):
    if len(block) < 4:
        raise ValueError("table block too short")
    nrestart = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * nrestart
    if end < 0:
        raise ValueError("table block: bad restart count")
    pos, key = 0, b""
    while pos < end:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        if shared > len(key) or pos + non_shared + vlen > end:
            raise ValueError("table block: corrupt entry")
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        yield key, bytes(block[pos:pos + vlen])
        pos += vlen


def read_table(path, verify=True):
    """All (key, value) pairs of a LevelDB-format table file, in key order."""
    buf = open(path, "rb").read()
    if len(buf) < 48 or struct.unpack("<Q", buf[-8:])[0] != TABLE_MAGIC:
        raise ValueError("%s is not a TensorFlow checkpoint index (bad table magic)" % path)
    footer = buf[-48:]
    _, p = _get_varint(footer, 0)      # metaindex handle (unused: TF writes an empty metaindex block)
    _, p = _get_varint(footer, p)
    ioff, p = _get_varint(footer, p)
    isize, p = _get_varint(footer, p)
    out = []
    for _, handle in _block_entries(_read_block(buf, ioff, isize, verify)):
        boff, q = _get_varint(handle, 0)
        bsize, _ = _get_varint(handle, q)
        out.extend(_block_entries(_read_block(buf, boff, bsize, verify)))
    return out


# ---- bundle reading -----------------------------------------------------------------------------------------------
def _parse_shape(buf):
    dims = []
    for f, _, v in _pb_fields(buf):
        if f == 2:  # Dim
            size = 0
            for f2, _, v2 in _pb_fields(v):
                if f2 == 1:
                    size = v2 - (1 << 64) if v2 >= 1 << 63 else v2
            dims.append(size)
        elif f == 3 and v:
            raise ValueError("tensor of unknown rank in a checkpoint")
    return tuple(dims)


def _parse_entry(buf):
    e = {"dtype": 0, "shape": (), "shard_id": 0, "offset": 0, "size": 0, "crc32c": None, "slices": 0}
    for f, _, v in _pb_fields(buf):
        if f == 1:
            e["dtype"] = v
        elif f == 2:
            e["shape"] = _parse_shape(v)
        elif f == 3:
            e["shard_id"] = v
        elif f == 4:
            e["offset"] = v
        elif f == 5:
            e["size"] = v
        elif f == 6:
            e["crc32c"] = v
        elif f == 7:
            e["slices"] += 1
    return e


def _parse_header(buf):
    h = {"num_shards": 0, "endianness": 0}
    for f, _, v in _pb_fields(buf):
        if f == 1:
            h["num_shards"] = v
        elif f == 2:
            h["endianness"] = v
    return h


def is_tf_checkpoint(prefix):
    return os.path.exists(str(prefix) + ".index")


def list_variables(prefix):
    """[(name, shape, numpy dtype or 'bfloat16')] like tf.train.list_variables."""
    out = []
    for key, val in read_table(str(prefix) + ".index"):
        if key == b"":
            continue
        e = _parse_entry(val)
        out.append((key.decode("utf-8"), e["shape"], "bfloat16" if e["dtype"] == _DT_BFLOAT16 else _DTYPES.get(e["dtype"])))
    return out


def load_tf_checkpoint(prefix, verify_crc=False, float_only=True):
    """{variable name: ndarray} of a V2 checkpoint ``prefix`` (``prefix.index`` + data shards).

    float_only (default): float32/64/16/bfloat16 variables only, as float32 -- what the engines consume; the
    Saver's bookkeeping (``global_step`` int64, ...) is dropped.  Adam slot variables (``.../Adam``, ``.../Adam_1``,
    ``beta1_power``) are float and are returned: consumers select by name.  verify_crc checks every tensor's
    CRC-32C (pure Python: ~1 MB/s, meant for tests and small files)."""
    prefix = str(prefix)
    entries = read_table(prefix + ".index")
    if not entries or entries[0][0] != b"":
        raise ValueError("%s.index has no bundle header entry" % prefix)
    header = _parse_header(entries[0][1])
    if header["endianness"] != 0:
        raise ValueError("big-endian checkpoint")
    nshards = max(header["num_shards"], 1)
    shards = {}
    out = {}
    for key, val in entries[1:]:
        e = _parse_entry(val)
        name = key.decode("utf-8")
        if e["slices"]:
            raise ValueError("variable %r is partitioned (slices): not supported" % name)
        if e["dtype"] == _DT_BFLOAT16:
            dt = np.dtype("<u2")
        elif e["dtype"] in _DTYPES:
            dt = _DTYPES[e["dtype"]]
        else:
            if float_only:
                continue  # strings, resources, ...
            raise ValueError("variable %r: unsupported dtype enum %d" % (name, e["dtype"]))
        is_float = e["dtype"] in (1, 2, 19, _DT_BFLOAT16)
        if float_only and not is_float:
            continue
        count = int(np.prod(e["shape"], dtype=np.int64)) if e["shape"] else 1
        if count * dt.itemsize != e["size"]:
            raise ValueError("variable %r: %d bytes stored for shape %s of %s" % (name, e["size"], e["shape"], dt))
        sid = e["shard_id"]
        if sid not in shards:
            path = "%s.data-%05d-of-%05d" % (prefix, sid, nshards)
            if not os.path.exists(path):
                raise FileNotFoundError("checkpoint data shard %s is missing" % path)
            shards[sid] = np.memmap(path, dtype=np.uint8, mode="r")
        raw = shards[sid][e["offset"]:e["offset"] + e["size"]]
        if raw.size != e["size"]:
            raise ValueError("variable %r: [%d, +%d) outside its data shard" % (name, e["offset"], e["size"]))
        if verify_crc and e["crc32c"] is not None and unmask_crc(e["crc32c"]) != _crc_bulk(raw.tobytes()):
            raise ValueError("variable %r: CRC-32C mismatch" % name)
        arr = np.frombuffer(raw.tobytes(), dtype=dt).reshape(e["shape"])
        if e["dtype"] == _DT_BFLOAT16:
            arr = (arr.astype(np.uint32) << 16).view(np.float32)
        out[name] = arr.astype(np.float32) if (float_only and is_float) else arr
    return out


# ---- writing --------------------------------------------------------------------------------------------------------
class _BlockBuilder:
    def __init__(self, restart_interval=16):
        self.buf = bytearray()
        self.restarts = [0]
        self.count = 0
        self.last = b""
        self.interval = restart_interval

    def add(self, key, value):
        shared = 0
        if self.count < self.interval:
            m = min(len(key), len(self.last))
            while shared < m and key[shared] == self.last[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.count = 0
        self.buf += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value))
        self.buf += key[shared:] + value
        self.last = key
        self.count += 1

    def finish(self):
        return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))

    def size(self):
        return len(self.buf) + 4 * len(self.restarts) + 4


def _handle(off, size):
    return _put_varint(off) + _put_varint(size)


def write_table(path, items, block_size=4096):
    """Write sorted (key bytes, value bytes) pairs as an uncompressed LevelDB-format table."""
    out = bytearray()

    def emit(block):
        off = len(out)
        out.extend(block)
        out.extend(b"\x00" + struct.pack("<I", mask_crc(crc32c(block + b"\x00"))))
        return off, len(block)

    index = _BlockBuilder(restart_interval=1)
    bb = _BlockBuilder()
    last_key = None
    prev = None
    for key, val in items:
        if prev is not None and key <= prev:
            raise ValueError("table keys must be strictly increasing")
        prev = key
        bb.add(key, val)
        last_key = key
        if bb.size() >= block_size:
            off, size = emit(bb.finish())
            index.add(last_key, _handle(off, size))
            bb = _BlockBuilder()
            last_key = None
    if last_key is not None:
        off, size = emit(bb.finish())
        index.add(last_key, _handle(off, size))
    moff, msize = emit(_BlockBuilder().finish())   # empty metaindex block
    ioff, isize = emit(index.finish())
    footer = _handle(moff, msize) + _handle(ioff, isize)
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
    out.extend(footer)
    with open(path, "wb") as f:
        f.write(out)


def save_tf_checkpoint(prefix, variables):
    """Write {name: ndarray} as a one-shard V2 checkpoint ``prefix.index`` + ``prefix.data-00000-of-00001`` that
    tf.train.Saver / tf.train.load_checkpoint read (variables in name order, as BundleWriter lays them out)."""
    prefix = str(prefix)
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    items = []
    header = _pb_varint_field(1, 1) + _pb_varint_field(2, 0) + _pb_bytes_field(3, _pb_varint_field(1, 1))
    items.append((b"", header))
    offset = 0
    with open(prefix + ".data-00000-of-00001", "wb") as data:
        for name in sorted(variables, key=lambda s: s.encode("utf-8")):
            arr = np.asarray(variables[name])  # (ascontiguousarray would turn a scalar into shape (1,))
            arr = arr if arr.flags.c_contiguous else arr.copy(order="C")
            dt = arr.dtype.newbyteorder("=")
            if dt not in _DTYPE_CODES:
                raise ValueError("variable %r: dtype %s has no TensorFlow enum here" % (name, arr.dtype))
            raw = arr.astype(arr.dtype.newbyteorder("<"), copy=False).tobytes()
            shape = b"".join(_pb_bytes_field(2, _pb_varint_field(1, int(d))) for d in arr.shape)
            entry = _pb_varint_field(1, _DTYPE_CODES[dt]) + _pb_bytes_field(2, shape)
            if offset:
                entry += _pb_varint_field(4, offset)
            entry += _pb_varint_field(5, len(raw))
            entry += _put_varint((6 << 3) | 5) + struct.pack("<I", mask_crc(_crc_bulk(raw)))
            items.append((name.encode("utf-8"), entry))
            data.write(raw)
            offset += len(raw)
    write_table(prefix + ".index", items)
    return prefix


def _crc_bulk(raw):
    """CRC-32C of a tensor's bytes: the library's slicing-by-8 host routine (fn2_crc32c, ~2 GB/s) when
    libflownet2_hip.so is built; without it (a host that only converts checkpoints / records: nothing here needs a
    GPU) the lane-parallel NumPy form.  The pure-Python loop is ~1 MB/s and only serves the few-KB table blocks."""
    if len(raw) < 65536:
        return crc32c(raw)
    fn = _host_crc()
    if fn is None:
        return crc32c_lanes(raw)
    import ctypes as C
    buf = np.frombuffer(raw, dtype=np.uint8)
    return int(fn(buf.ctypes.data_as(C.c_void_p), buf.size, 0))


_HOST_CRC = []


def _host_crc():
    """fn2_crc32c of the built library, or None (library or torch missing); FN2_NO_HOST_CRC=1 forces None."""
    if not _HOST_CRC:
        fn = None
        if not os.environ.get("FN2_NO_HOST_CRC"):
            try:
                from . import _hip
                fn = _hip.lib().fn2_crc32c
            except Exception:  # HipLibraryMissing, OSError (unloadable .so), ImportError (no torch)
                fn = None
        _HOST_CRC.append(fn)
    return _HOST_CRC[0]
