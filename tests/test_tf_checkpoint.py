"""TensorFlow V2 checkpoint (tensor bundle) container: the file the reference restores with tf.train.Saver
(/root/reference src/net.py:566-569; default paths src/flownet_*/test.py:15).  No checkpoint exists offline, so the
reader is pinned by the format's published known answers and by write -> read round trips."""
import os
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flownet2-tf_amd"))

from src import tf_checkpoint as T  # noqa: E402
from src import weights as W  # noqa: E402


def test_crc32c_known_answers():
    # RFC 3720 B.4 test vectors + the "123456789" check value of the Castagnoli polynomial
    assert T.crc32c(b"123456789") == 0xE3069283
    assert T.crc32c(bytes(32)) == 0x8A9136AA
    assert T.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert T.crc32c(bytes(range(32))) == 0x46DD794E
    assert T.crc32c(bytes(range(31, -1, -1))) == 0x113FDB5C
    for v in (0, 1, 0xE3069283, 0xFFFFFFFF):
        assert T.unmask_crc(T.mask_crc(v)) == v
    assert T.mask_crc(T.crc32c(b"foo")) != T.crc32c(b"foo")


def test_library_crc_matches_python():
    import ctypes as C
    from src import _hip
    lib = _hip.lib()
    rng = np.random.default_rng(0)
    buf = rng.integers(0, 256, 100003, dtype=np.uint8)
    for off, n in ((0, 0), (0, 1), (1, 7), (3, 8), (5, 1000), (0, 100003), (7, 65536)):
        part = buf[off:off + n]
        got = lib.fn2_crc32c(part.ctypes.data_as(C.c_void_p), part.size, 0)
        assert got == T.crc32c(part.tobytes()), (off, n)
    # continuation: crc(a + b) == crc(b, crc(a))
    a, b = buf[:1234], buf[1234:5000]
    ca = lib.fn2_crc32c(a.ctypes.data_as(C.c_void_p), a.size, 0)
    assert lib.fn2_crc32c(b.ctypes.data_as(C.c_void_p), b.size, ca) == T.crc32c(buf[:5000].tobytes())


def test_varint_and_snappy_vectors():
    assert T._put_varint(300) == b"\xac\x02"          # protobuf documentation example
    assert T._get_varint(b"\xac\x02", 0) == (300, 2)
    assert T._put_varint(-1) == b"\xff" * 9 + b"\x01"  # int64 -1 (an unknown dimension)
    # snappy: length 10, literal "a", copy(offset 1, length 9)
    assert T._snappy_decompress(b"\x0a\x00a\x15\x01") == b"a" * 10


def _vars(seed=0, n_extra=40):
    rng = np.random.default_rng(seed)
    v = {
        "FlowNetS/conv1/weights": rng.standard_normal((7, 7, 6, 64)).astype(np.float32),
        "FlowNetS/conv1/biases": rng.standard_normal((64,)).astype(np.float32),
        "FlowNetS/deconv5/weights": rng.standard_normal((4, 4, 16, 32)).astype(np.float32),
        "FlowNetS/conv1/weights/Adam": rng.standard_normal((7, 7, 6, 64)).astype(np.float32),
        "beta1_power": np.float32(0.9),
        "global_step": np.int64(1200000),
    }
    for i in range(n_extra):  # enough keys with long shared prefixes for several 4 KB table blocks
        v["FlowNetS/a_rather_long_scope_name_to_fill_index_blocks/layer_%03d/weights" % i] = \
            rng.standard_normal((3, 1, 2, 5)).astype(np.float32)
    return v


def test_round_trip(tmp_path):
    v = _vars(n_extra=600)
    prefix = str(tmp_path / "ckpts" / "flownet-S.ckpt-0")
    T.save_tf_checkpoint(prefix, v)
    idx = open(prefix + ".index", "rb").read()
    assert idx[-8:] == bytes.fromhex("57fb808b247547db")  # kTableMagicNumber, little endian
    assert os.path.getsize(prefix + ".data-00000-of-00001") == sum(np.asarray(a).nbytes for a in v.values())
    table = T.read_table(prefix + ".index")
    assert [k for k, _ in table] == sorted([b""] + [k.encode() for k in v])
    assert len(idx) > 2 * 4096  # really several data blocks + an index block
    got = T.load_tf_checkpoint(prefix, verify_crc=True)
    assert "global_step" not in got  # float_only drops the Saver's bookkeeping
    for k, a in v.items():
        if k == "global_step":
            continue
        assert got[k].dtype == np.float32 and got[k].shape == np.asarray(a).shape
        np.testing.assert_array_equal(got[k], a)
    full = T.load_tf_checkpoint(prefix, float_only=False)
    assert full["global_step"].dtype == np.int64 and int(full["global_step"]) == 1200000
    names = {n: (s, d) for n, s, d in T.list_variables(prefix)}
    assert names["FlowNetS/conv1/weights"] == ((7, 7, 6, 64), np.dtype("<f4"))
    assert names["beta1_power"][0] == ()


def test_large_tensor_crc_through_library(tmp_path):
    rng = np.random.default_rng(3)
    v = {"FlowNetS/conv6/weights": rng.standard_normal((3, 3, 512, 64)).astype(np.float32)}  # 1.2 MB
    prefix = T.save_tf_checkpoint(str(tmp_path / "m.ckpt-7"), v)
    np.testing.assert_array_equal(T.load_tf_checkpoint(prefix, verify_crc=True)["FlowNetS/conv6/weights"],
                                  v["FlowNetS/conv6/weights"])
    with open(prefix + ".data-00000-of-00001", "r+b") as f:  # flip one bit of the tensor
        f.seek(100000)
        b = f.read(1)
        f.seek(100000)
        f.write(bytes([b[0] ^ 1]))
    with pytest.raises(ValueError, match="CRC-32C"):
        T.load_tf_checkpoint(prefix, verify_crc=True)


def test_corruption_is_loud(tmp_path):
    prefix = T.save_tf_checkpoint(str(tmp_path / "x.ckpt-0"), _vars(1, 3))
    raw = bytearray(open(prefix + ".index", "rb").read())
    bad = bytearray(raw)
    bad[10] ^= 0x40
    open(prefix + ".index", "wb").write(bad)
    with pytest.raises(ValueError, match="CRC-32C"):
        T.read_table(prefix + ".index")
    open(prefix + ".index", "wb").write(raw[:-1])
    with pytest.raises(ValueError, match="magic"):
        T.read_table(prefix + ".index")
    open(prefix + ".index", "wb").write(raw)
    os.remove(prefix + ".data-00000-of-00001")
    with pytest.raises(FileNotFoundError):
        T.load_tf_checkpoint(prefix)


def test_bfloat16_and_half_variables(tmp_path):
    # written by hand: dtype enums 14 (bfloat16) and 19 (half) as another writer would emit them
    x = np.array([1.0, -2.5, 3.140625, 0.0], np.float32)
    bf = (x.view(np.uint32) >> 16).astype(np.uint16)
    hf = x.astype(np.float16)
    prefix = str(tmp_path / "h.ckpt-0")
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        f.write(bf.tobytes() + hf.tobytes())

    def entry(dtype, off):
        shape = T._pb_bytes_field(2, T._pb_varint_field(1, 4))
        e = T._pb_varint_field(1, dtype) + T._pb_bytes_field(2, shape)
        if off:
            e += T._pb_varint_field(4, off)
        return e + T._pb_varint_field(5, 8)

    header = T._pb_varint_field(1, 1)
    T.write_table(prefix + ".index", [(b"", header), (b"a/bf", entry(14, 0)), (b"a/half", entry(19, 8))])
    got = T.load_tf_checkpoint(prefix)
    np.testing.assert_array_equal(got["a/bf"], x)      # these values are exact in bfloat16
    np.testing.assert_array_equal(got["a/half"], x)


def test_weights_and_net_load_a_tf_checkpoint(tmp_path):
    wts = W.init_weights("FlowNetS", 5)
    prefix = str(tmp_path / "FlowNetS" / "flownet-S.ckpt-0")
    T.save_tf_checkpoint(prefix, dict(wts, global_step=np.int64(0)))
    for path in (prefix, prefix + ".index", prefix + ".data-00000-of-00001"):
        assert W.checkpoint_exists(path)
        got = W.load_weights(path)
        assert set(got) == set(wts)
        for k in wts:
            np.testing.assert_array_equal(got[k], wts[k])
    assert not W.checkpoint_exists(str(tmp_path / "nope.ckpt-0"))
    from src.flownet_s.flownet_s import FlowNetS
    net = FlowNetS()
    loaded = net.load_weights(prefix)
    np.testing.assert_array_equal(loaded["FlowNetS/conv3_1/weights"], wts["FlowNetS/conv3_1/weights"])


def test_bulk_crc_without_the_hip_library(monkeypatch):
    """Host-only conversion of checkpoints / records must not need libflownet2_hip.so: the lane-parallel NumPy CRC-32C
    equals the byte loop (RFC 3720 check value included) and the library routine on ragged sizes."""
    from src import tf_checkpoint as T
    assert T.crc32c_lanes(b"123456789") == 0xE3069283
    rng = np.random.default_rng(0)
    for n in (65536, 65537, 100003, 4096 * 16 + 5, 300000):
        raw = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert T.crc32c_lanes(raw) == T.crc32c(raw) == T._crc_bulk(raw), n
    monkeypatch.setattr(T, "_HOST_CRC", [None])   # as on a host without the library
    raw = rng.integers(0, 256, 70001, dtype=np.uint8).tobytes()
    assert T._crc_bulk(raw) == T.crc32c(raw)
