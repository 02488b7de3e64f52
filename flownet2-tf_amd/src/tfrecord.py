"""TFRecord files of tf.train.Example, read and written without TensorFlow.

The reference trains from ``fc_train_all.tfrecords`` etc. (/root/reference src/dataset_configs.py:56-59): ZLIB-
compressed TFRecord files (src/dataloader.py:442) of tf.train.Example protos whose features ``image_a``, ``image_b``
(float64 H x W x 3, already divided by 255 unless PREPROCESS['scale']), ``flow`` (float32 H x W x 2) and -- for the
interpolation variant -- ``matches_a`` (float64 H x W x 1), ``sparse_flow`` (float32 H x W x 2), ``edges_a`` (float32
H x W x 1) are raw ``ndarray.tostring()`` bytes (scripts/convert_set_to_tfrecords.py:62-86, :509-575), decoded with
``tf.decode_raw`` + reshape to the dataset's PADDED size (dataloader.py:210-270).

Container restated from TensorFlow's published format (third-party, not vendored, not installed here):
record = uint64 LE length | uint32 masked CRC-32C(length bytes) | data | uint32 masked CRC-32C(data);
ZLIB option = one zlib stream over the concatenated records; tf.train.Example = field 1 Features{ map<string,
Feature> feature = 1 }, Feature = oneof { BytesList bytes_list = 1; FloatList float_list = 2; Int64List int64_list =
3 }, each list's ``value`` = field 1 (floats / int64 packed or not).  Parity is pinned by the format's known
answers and write -> read round trips (tests/test_tfrecord.py); no reference .tfrecords file exists offline.
"""
import struct
import zlib

import numpy as np

from .tf_checkpoint import (_crc_bulk, _get_varint, _pb_bytes_field, _pb_fields, _put_varint, mask_crc, unmask_crc)

# feature name -> (dtype, channels) of the reference's records
FEATURE_SPECS = {
    "image_a": (np.float64, 3), "image_b": (np.float64, 3), "matches_a": (np.float64, 1),
    "flow": (np.float32, 2), "sparse_flow": (np.float32, 2), "edges_a": (np.float32, 1),
}


class _Stream:
    """Byte stream over a plain or zlib/gzip-compressed file, decompressed incrementally (records are hundreds of
    MB per thousand samples: never load the file whole)."""

    def __init__(self, f, chunk=1 << 20):
        self.f, self.chunk, self.buf, self.eof = f, chunk, bytearray(), False
        head = f.read(12)
        f.seek(0)
        self.z = None
        # an uncompressed file starts with a record header whose length CRC checks out (a 376-byte first record starts
        # with the bytes 78 01, a valid zlib header: the CRC decides, not the magic); otherwise zlib / gzip by header
        plain = len(head) == 12 and unmask_crc(struct.unpack("<I", head[8:])[0]) == _crc_bulk(head[:8])
        if not plain and len(head) >= 2 and ((head[0] == 0x78 and (head[0] * 256 + head[1]) % 31 == 0) or
                                              head[:2] == b"\x1f\x8b"):
            self.z = zlib.decompressobj(47)  # zlib or gzip header, auto-detected

    def _fill(self, n):
        while len(self.buf) < n and not self.eof:
            raw = self.f.read(self.chunk)
            if not raw:
                self.eof = True
                if self.z is not None:
                    self.buf += self.z.flush()
                break
            self.buf += self.z.decompress(raw) if self.z is not None else raw

    def read(self, n):
        self._fill(n)
        out = bytes(self.buf[:n])
        del self.buf[:n]
        return out


def read_records(path, verify=True):
    """Yield the payload bytes of every record of a TFRecord file (uncompressed, ZLIB or GZIP)."""
    with open(path, "rb") as f:
        s = _Stream(f)
        while True:
            head = s.read(12)
            if not head:
                return
            if len(head) != 12:
                raise ValueError("%s: truncated record header" % path)
            (n,), (lcrc,) = struct.unpack("<Q", head[:8]), struct.unpack("<I", head[8:])
            if verify and unmask_crc(lcrc) != _crc_bulk(head[:8]):
                raise ValueError("%s: record length CRC mismatch (not a TFRecord file, or corrupt)" % path)
            body = s.read(n + 4)
            if len(body) != n + 4:
                raise ValueError("%s: truncated record" % path)
            data = body[:n]
            if verify and unmask_crc(struct.unpack("<I", body[n:])[0]) != _crc_bulk(data):
                raise ValueError("%s: record data CRC mismatch" % path)
            yield data


class TFRecordWriter:
    """with TFRecordWriter(path, compression='ZLIB') as w: w.write(example_bytes)  -- the reference's
    tf.python_io.TFRecordWriter(filename, options=ZLIB) (convert_set_to_tfrecords.py:91-93)."""

    def __init__(self, path, compression="ZLIB"):
        if compression not in ("ZLIB", "GZIP", "", None):
            raise ValueError("compression must be 'ZLIB', 'GZIP' or ''")
        self.f = open(path, "wb")
        self.z = None
        if compression == "ZLIB":
            self.z = zlib.compressobj(-1, zlib.DEFLATED, 15)
        elif compression == "GZIP":
            self.z = zlib.compressobj(-1, zlib.DEFLATED, 31)

    def write(self, data):
        data = bytes(data)
        ln = struct.pack("<Q", len(data))
        rec = ln + struct.pack("<I", mask_crc(_crc_bulk(ln))) + data + struct.pack("<I", mask_crc(_crc_bulk(data)))
        self.f.write(self.z.compress(rec) if self.z is not None else rec)

    def close(self):
        if self.f is not None:
            if self.z is not None:
                self.f.write(self.z.flush())
            self.f.close()
            self.f = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


# ---- tf.train.Example ---------------------------------------------------------------------------------------------
def parse_example(buf):
    """Serialized tf.train.Example -> {name: list of bytes | float32 array | int64 array}."""
    out = {}
    for f, _, features in _pb_fields(buf):
        if f != 1:
            continue
        for f2, _, entry in _pb_fields(features):
            if f2 != 1:
                continue
            key, feat = None, b""
            for f3, _, v in _pb_fields(entry):  # map entry: key = 1, value = 2
                if f3 == 1:
                    key = v.decode("utf-8")
                elif f3 == 2:
                    feat = v
            if key is None:
                continue
            val = None
            for kind, _, lst in _pb_fields(feat):
                if kind == 1:
                    val = [v for f4, _, v in _pb_fields(lst) if f4 == 1]
                elif kind == 2:
                    vals = []
                    for f4, wt, v in _pb_fields(lst):
                        if f4 == 1 and wt == 2:
                            vals.append(np.frombuffer(v, "<f4"))
                        elif f4 == 1:
                            vals.append(np.array([v], "<u4").view("<f4"))
                    val = np.concatenate(vals) if vals else np.zeros(0, np.float32)
                elif kind == 3:
                    vals = []
                    for f4, wt, v in _pb_fields(lst):
                        if f4 != 1:
                            continue
                        if wt == 2:
                            p = 0
                            while p < len(v):
                                x, p = _get_varint(v, p)
                                vals.append(x)
                        else:
                            vals.append(v)
                    val = np.array([x - (1 << 64) if x >= 1 << 63 else x for x in vals], np.int64)
            out[key] = val
    return out


def make_example(features):
    """{name: bytes | list of bytes | float array | int array} -> serialized tf.train.Example (map entries in key
    order, as protobuf's deterministic serialization emits them)."""
    entries = b""
    for key in sorted(features):
        v = features[key]
        if isinstance(v, (bytes, bytearray)):
            v = [bytes(v)]
        if isinstance(v, (list, tuple)) and all(isinstance(x, (bytes, bytearray)) for x in v):
            feat = _pb_bytes_field(1, b"".join(_pb_bytes_field(1, bytes(x)) for x in v))
        else:
            arr = np.asarray(v)
            if arr.dtype.kind == "f":
                feat = _pb_bytes_field(2, _pb_bytes_field(1, arr.astype("<f4").tobytes()))
            elif arr.dtype.kind in "iub":
                feat = _pb_bytes_field(3, _pb_bytes_field(1, b"".join(_put_varint(int(x)) for x in arr.reshape(-1))))
            else:
                raise ValueError("feature %r: unsupported value type" % key)
        entries += _pb_bytes_field(1, _pb_bytes_field(1, key.encode("utf-8")) + _pb_bytes_field(2, feat))
    return _pb_bytes_field(1, entries)


# ---- the reference's samples ----------------------------------------------------------------------------------------
def encode_sample(image_a, image_b=None, flow=None, matches_a=None, sparse_flow=None, edges_a=None):
    """One sample as scripts/convert_set_to_tfrecords.py writes it: every array as raw bytes in the dtype the
    reader decodes (images / matches float64, flows / edges float32).  Arrays left None are stored as b'' like the
    reference's estimation-only records would omit them."""
    given = {"image_a": image_a, "image_b": image_b, "flow": flow, "matches_a": matches_a,
             "sparse_flow": sparse_flow, "edges_a": edges_a}
    feats = {}
    for name, arr in given.items():
        if arr is None:
            continue
        dt, ch = FEATURE_SPECS[name]
        arr = np.asarray(arr)
        if arr.ndim == 2:
            arr = arr[:, :, None]
        if arr.ndim != 3 or arr.shape[2] != ch:
            raise ValueError("%s must be H x W x %d, got %s" % (name, ch, arr.shape))
        feats[name] = np.ascontiguousarray(arr, dtype=np.dtype(dt).newbyteorder("<")).tobytes()
    return make_example(feats)


def decode_sample(example, height, width, names=("image_a", "image_b", "flow")):
    """parse_example output -> {name: float32 array H x W x C}: tf.decode_raw + reshape + cast
    (dataloader.py:171-175, :464).  A byte count that does not match height x width is an error, as in TF."""
    out = {}
    for name in names:
        if name not in example or not example[name]:
            raise ValueError("record has no %r feature" % name)
        dt, ch = FEATURE_SPECS[name]
        raw = example[name][0]
        if len(raw) != height * width * ch * np.dtype(dt).itemsize:
            raise ValueError("feature %r holds %d bytes, expected %d x %d x %d of %s" % (
                name, len(raw), height, width, ch, np.dtype(dt).name))
        out[name] = np.frombuffer(raw, np.dtype(dt).newbyteorder("<")).reshape(height, width, ch).astype(np.float32)
    return out


def read_samples(path, height, width, names=("image_a", "image_b", "flow"), verify=True):
    for rec in read_records(path, verify):
        yield decode_sample(parse_example(rec), height, width, names)


def count_records(path):
    return sum(1 for _ in read_records(path, verify=False))


def convert_list(list_path, out_path, compression="ZLIB", divisor=64):
    """The estimation branch of scripts/convert_set_to_tfrecords.py without TensorFlow: every `image_a image_b
    flow.flo` line becomes one Example (images / 255 as float64, flow float32, all zero-padded bottom/right to a
    multiple of `divisor`, :485-563).  Returns the number of records written."""
    from .dataloader import read_list
    from .flowlib import read_flow
    from .net import imread
    n = 0
    with TFRecordWriter(out_path, compression) as w:
        for pa, pb, pf in read_list(list_path):
            a = imread(pa).astype(np.float64) / 255.0
            b = imread(pb).astype(np.float64) / 255.0
            f = read_flow(pf).astype(np.float32)
            h, wd = a.shape[:2]
            ph, pw = (-h) % divisor, (-wd) % divisor
            pad = [(0, ph), (0, pw), (0, 0)]
            w.write(encode_sample(np.pad(a, pad), np.pad(b, pad), np.pad(f, pad)))
            n += 1
    return n


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description="list of `image_a image_b flow.flo` triples -> .tfrecords (ZLIB)")
    ap.add_argument("--list", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--compression", default="ZLIB", choices=["ZLIB", "GZIP", ""])
    A = ap.parse_args()
    print("%d records -> %s" % (convert_list(A.list, A.out, A.compression), A.out))
