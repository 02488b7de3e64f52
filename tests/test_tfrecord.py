"""TFRecord / tf.train.Example container of the reference's training data (/root/reference src/dataloader.py:210-270,
:442; scripts/convert_set_to_tfrecords.py:62-93) without TensorFlow: known serializations + round trips."""
import os
import struct
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flownet2-tf_amd"))
sys.path.insert(0, ROOT)

from src import tfrecord as R  # noqa: E402
from src.tf_checkpoint import crc32c, mask_crc  # noqa: E402

SAMPLES = os.path.join(ROOT, "tests", "golden", "samples")


def test_example_known_serialization():
    # Example{features{feature{key:"a" value{int64_list{value:[1]}}}}} as protobuf serializes it
    want = bytes.fromhex("0a0c0a0a0a016112051a030a0101")
    assert R.make_example({"a": np.array([1])}) == want
    got = R.parse_example(want)
    assert list(got) == ["a"] and got["a"].dtype == np.int64 and got["a"].tolist() == [1]
    # bytes and float features, unpacked repeated encodings (older writers) included
    ex = R.make_example({"img": b"\x00\x01\x02", "f": np.array([1.5, -2.0], np.float32), "n": [-1, 300]})
    p = R.parse_example(ex)
    assert p["img"] == [b"\x00\x01\x02"] and p["f"].tolist() == [1.5, -2.0] and p["n"].tolist() == [-1, 300]
    unpacked = bytes.fromhex("0a150a130a0166120e120c") + b"\x0d" + struct.pack("<f", 1.5) + b"\x0d" + struct.pack("<f", 2.5) + b"\x0a\x00"
    assert R.parse_example(unpacked)["f"].tolist() == [1.5, 2.5]


def test_record_framing(tmp_path):
    path = str(tmp_path / "r.tfrecords")
    with R.TFRecordWriter(path, "") as w:
        w.write(b"hello")
    raw = open(path, "rb").read()
    assert raw[:8] == struct.pack("<Q", 5) and raw[12:17] == b"hello" and len(raw) == 8 + 4 + 5 + 4
    assert struct.unpack("<I", raw[8:12])[0] == mask_crc(crc32c(raw[:8]))
    assert struct.unpack("<I", raw[17:])[0] == mask_crc(crc32c(b"hello"))
    assert list(R.read_records(path)) == [b"hello"]


@pytest.mark.parametrize("compression", ["ZLIB", "GZIP", ""])
def test_round_trip_and_corruption(tmp_path, compression):
    rng = np.random.default_rng(1)
    payloads = [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in (0, 1, 70000, 3_000_000, 17)]
    path = str(tmp_path / "x.tfrecords")
    with R.TFRecordWriter(path, compression) as w:
        for p in payloads:
            w.write(p)
    head = open(path, "rb").read(2)
    if compression == "ZLIB":
        assert head[0] == 0x78  # a zlib stream, as TFRecordCompressionType.ZLIB writes
        zlib.decompress(open(path, "rb").read())
    assert list(R.read_records(path)) == payloads
    assert R.count_records(path) == len(payloads)
    if compression == "":
        raw = bytearray(open(path, "rb").read())
        raw[12 + 1 + 4 + 12 + 5] ^= 0x10  # a data byte of the third record
        open(path, "wb").write(raw)
        with pytest.raises(ValueError, match="CRC"):
            list(R.read_records(path))
        open(path, "wb").write(raw[:-3])
        with pytest.raises(ValueError, match="truncated|CRC"):
            list(R.read_records(path))


def test_reference_sample_round_trip(tmp_path):
    from src.flowlib import read_flow
    from src.net import imread
    lst = tmp_path / "l.txt"
    # the committed data fixtures: pair 0 and both ground-truth flows of the reference's data/samples
    rows = ["%s %s %s\n" % tuple(os.path.join(SAMPLES, n) for n in ("0img0.ppm", "0img1.ppm", "%dflow.flo" % i))
            for i in (0, 1)]
    if not all(os.path.exists(p) for r in rows for p in r.split()):
        pytest.skip("sample pair fixtures not present")
    lst.write_text("".join(rows) * 3)
    out = str(tmp_path / "fc_train_all.tfrecords")
    assert R.convert_list(str(lst), out) == 6
    smps = list(R.read_samples(out, 384, 512))
    assert len(smps) == 6
    a0 = imread(rows[0].split()[0]).astype(np.float64) / 255.0
    np.testing.assert_array_equal(smps[0]["image_a"], a0.astype(np.float32))
    np.testing.assert_array_equal(smps[1]["flow"], read_flow(rows[1].split()[2]))
    ex = R.parse_example(next(R.read_records(out)))
    assert sorted(ex) == ["flow", "image_a", "image_b"]
    assert len(ex["image_a"][0]) == 384 * 512 * 3 * 8 and len(ex["flow"][0]) == 384 * 512 * 2 * 4  # float64 / float32
    with pytest.raises(ValueError, match="expected"):
        list(R.read_samples(out, 448, 1024))
    # the shuffled epoch source of the loader: every sample exactly once
    from src import dataloader as D
    got = list(D._tfrecord_epoch(out, np.random.default_rng(0), (384, 512), False, shuffle_buffer=4))
    assert len(got) == 6
    sums = sorted(round(float(g[2].sum()), 1) for g in got)
    assert sums == sorted(round(float(s["flow"].sum()), 1) for s in smps)


def test_uncompressed_file_whose_first_length_looks_like_a_zlib_header(tmp_path):
    path = str(tmp_path / "u.tfrecords")
    payload = bytes(range(256)) + bytes(120)        # 376 bytes: the length field starts with 78 01, a valid zlib header
    assert struct.pack("<Q", len(payload))[:2] == b"\x78\x01"
    with R.TFRecordWriter(path, "") as w:
        w.write(payload)
        w.write(b"second")
    assert list(R.read_records(path)) == [payload, b"second"]
