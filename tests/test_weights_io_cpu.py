"""Keras HDF5 weight files without h5py (building_detection_amd/h5lite.py, weights_io.py).

Real Keras / h5py output does not exist in this image; the cross-check against the real HDF5 library that does
(libhdf5 1.10.6) is tests/test_h5_libhdf5_cpu.py.  What is pinned HERE: the writer against the reader,
the reader against two byte-level files assembled HERE from the field tables of the HDF5 File Format Specification
(independent of the writer: one in the old format Keras' files use - superblock 0, symbol-table groups, v1 headers - and one
in the new format - superblock 2, OHDR headers, link messages, compact data, variable-length string attribute), and the
layer matching rules of load_weights."""
import os
import struct

import numpy as np
import pytest

from building_detection_amd import h5lite as H
from building_detection_amd import weights_io as WIO

U = 0xFFFFFFFFFFFFFFFF


def test_roundtrip_groups_datasets_attributes(tmp_path):
    rng = np.random.default_rng(0)
    w = H.Writer()
    w.attr("", "backend", b"tensorflow")
    w.attr("", "n", np.int64(7))
    w.attr("", "vec", np.arange(5, dtype=np.float32))
    arrs = {}
    for i in range(700):  # > 2 * 4 * 2 * 16 links: the writer has to raise the SNOD size (file-level K)
        a = rng.standard_normal((i % 5 + 1, 3)).astype(np.float32)
        w.dataset(f"g{i % 7}/sub/d{i}", a)
        arrs[f"g{i % 7}/sub/d{i}"] = a
        w.dataset(f"flat{i}", np.float64(i))
    w.dataset("i32", np.arange(-3, 4, dtype=np.int32))
    w.dataset("u8", np.arange(200, dtype=np.uint8).reshape(10, 20))
    w.dataset("f16", np.array([1.5, -2.25], np.float16))
    w.dataset("empty", np.zeros((0, 4), np.float32))
    w.attr("g0", "weight_names", np.array([b"a/kernel:0", b"a/bias:0"]))
    w.attr("g1", "none", np.array([], dtype="S1"))
    p = str(tmp_path / "t.h5")
    w.save(p)
    assert H.is_hdf5(p)
    f = H.File(p)
    assert f.attrs["backend"] == b"tensorflow" and int(f.attrs["n"]) == 7 and np.array_equal(f.attrs["vec"], np.arange(5, dtype=np.float32))
    assert len(f.keys()) == 7 + 700 + 4
    for k, a in arrs.items():
        got = np.asarray(f[k])
        assert got.dtype == np.float32 and np.array_equal(got, a)
    assert np.asarray(f["flat123"]).shape == () and float(np.asarray(f["flat123"])) == 123.0
    assert np.array_equal(np.asarray(f["i32"]), np.arange(-3, 4)) and np.asarray(f["i32"]).dtype == np.int32
    assert np.array_equal(np.asarray(f["u8"]), np.arange(200, dtype=np.uint8).reshape(10, 20))
    assert np.array_equal(np.asarray(f["f16"]), np.array([1.5, -2.25], np.float16))
    assert np.asarray(f["empty"]).shape == (0, 4)
    assert [x.decode() for x in f["g0"].attrs["weight_names"]] == ["a/kernel:0", "a/bias:0"]
    assert f["g1"].attrs["none"].shape == (0,)
    with pytest.raises(KeyError):
        f["nope/x"]


def _pad8(b):
    return b + b"\0" * (-len(b) % 8)


def _msg1(t, body):
    body = _pad8(body)
    return struct.pack("<HHBBBB", t, len(body), 0, 0, 0, 0) + body


def _ohdr1(msgs):
    body = b"".join(msgs)
    return bytes([1, 0]) + struct.pack("<HII", len(msgs), 1, len(body)) + b"\0" * 4 + body


def test_reader_on_a_hand_assembled_old_format_file(tmp_path):
    """Superblock version 0 (spec II.A), root group = symbol table message -> v1 B-tree node -> SNOD -> local heap
    (III.A.1, III.B, III.D), a contiguous float32 dataset [2,3] (IV.A.2.b/d/i), a v1 attribute holding two fixed-length
    strings (IV.A.2.m); HDF5's default K values (leaf 4, internal 16); junk after the end-of-file address."""
    data = np.arange(6, dtype="<f4").reshape(2, 3)
    buf = bytearray(b"\0" * 96)  # superblock (56) + root symbol table entry (40), filled in last

    def put(b, align=8):
        buf.extend(b"\0" * (-len(buf) % align))
        a = len(buf)
        buf.extend(b)
        return a

    raw = put(data.tobytes())
    f32 = bytes([0x11, 0x20, 31, 0]) + struct.pack("<I", 4) + struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
    space = bytes([1, 2, 0]) + b"\0" * 5 + struct.pack("<QQ", 2, 3)
    layout = bytes([3, 1]) + struct.pack("<QQ", raw, data.nbytes)
    strs = bytes([0x13, 0x00, 0, 0]) + struct.pack("<I", 8)            # 8-byte null-terminated strings
    aspace = bytes([1, 1, 0]) + b"\0" * 5 + struct.pack("<Q", 2)
    aname = b"weight_names\0"
    attr = (bytes([1, 0]) + struct.pack("<HHH", len(aname), len(strs), len(aspace)) + _pad8(aname) + _pad8(strs) + _pad8(aspace) +
            b"k:0\0\0\0\0\0" + b"bias:0\0\0")
    dset = put(_ohdr1([_msg1(0x01, space), _msg1(0x03, f32), _msg1(0x08, layout), _msg1(0x0C, attr)]))
    heap_data = put(b"\0" * 8 + b"kernel:0" + b"\0" * 8 + struct.pack("<QQ", 1, 16))
    heap = put(b"HEAP" + bytes(4) + struct.pack("<QQQ", 40, 24, heap_data))
    snod = put(b"SNOD" + bytes([1, 0]) + struct.pack("<H", 1) + struct.pack("<QQII", 8, dset, 0, 0) + b"\0" * 16 + b"\0" * (7 * 40))
    tree = put(b"TREE" + bytes([0, 0]) + struct.pack("<H", 1) + struct.pack("<QQ", U, U) + struct.pack("<QQQ", 0, snod, 8) +
               b"\0" * (62 * 8))
    root = put(_ohdr1([_msg1(0x11, struct.pack("<QQ", tree, heap))]))
    eof = len(buf)
    buf[0:96] = (H.SIGNATURE + bytes([0, 0, 0, 0, 0, 8, 8, 0]) + struct.pack("<HHI", 4, 16, 0) + struct.pack("<QQQQ", 0, U, eof, U) +
                 struct.pack("<QQII", 0, root, 1, 0) + struct.pack("<QQ", tree, heap))
    buf.extend(b"junk after eof")
    p = tmp_path / "hand0.h5"
    p.write_bytes(bytes(buf))
    f = H.File(str(p))
    assert f.keys() == ["kernel:0"]
    d = f["kernel:0"]
    assert d.shape == (2, 3) and d.dtype == np.float32 and np.array_equal(np.asarray(d), data)
    assert [s.decode() for s in d.attrs["weight_names"]] == ["k:0", "bias:0"]


def test_reader_on_a_hand_assembled_new_format_file(tmp_path):
    """Superblock version 2 (II.B), version-2 object headers "OHDR" (IV.A.1.b) with a link message (IV.A.2.g), a COMPACT
    int16 dataset (layout class 0), a version-3 attribute whose type is a variable-length string stored in a global heap
    collection "GCOL" (III.E)."""
    buf = bytearray(b"\0" * 48)

    def put(b, align=8):
        buf.extend(b"\0" * (-len(buf) % align))
        a = len(buf)
        buf.extend(b)
        return a

    def ohdr2(msgs):
        body = b"".join(struct.pack("<BHB", t, len(m), 0) + m for t, m in msgs)
        return b"OHDR" + bytes([2, 0x00]) + struct.pack("<B", len(body)) + body + b"\0\0\0\0"  # flags 0: 1-byte chunk size; checksum ignored

    text = b"tensorflow"
    gcol_body = struct.pack("<HHIQ", 1, 0, 0, len(text)) + _pad8(text) + struct.pack("<HHIQ", 0, 0, 0, 0)
    gcol = put(b"GCOL" + bytes([1, 0, 0, 0]) + struct.pack("<Q", 16 + len(gcol_body)) + gcol_body)
    vals = np.array([[1, -2, 3], [4, 5, -6]], "<i2")
    i16 = bytes([0x10, 0x08, 0, 0]) + struct.pack("<I", 2) + struct.pack("<HH", 0, 16)
    space2 = bytes([2, 2, 0, 1]) + struct.pack("<QQ", 2, 3)
    compact = bytes([3, 0]) + struct.pack("<H", vals.nbytes) + vals.tobytes()
    dset = put(ohdr2([(0x01, space2), (0x03, i16), (0x08, compact)]))
    vstr = bytes([0x19, 0x01, 0x01, 0]) + struct.pack("<I", 16) + bytes([0x10, 0x00, 0, 0]) + struct.pack("<I", 1) + struct.pack("<HH", 0, 8)
    scalar = bytes([2, 0, 0, 0])
    aname = b"backend\0"
    attr3 = (bytes([3, 0]) + struct.pack("<HHH", len(aname), len(vstr), len(scalar)) + bytes([0]) + aname + vstr + scalar +
             struct.pack("<IQI", len(text), gcol, 1))
    link = bytes([1, 0x00]) + struct.pack("<B", 4) + b"data" + struct.pack("<Q", dset)   # hard link, 1-byte name length
    linfo = bytes([0, 0]) + struct.pack("<QQ", U, U)
    root = put(ohdr2([(0x02, linfo), (0x06, link), (0x0C, attr3)]))
    buf[0:48] = H.SIGNATURE + bytes([2, 8, 8, 0]) + struct.pack("<QQQQ", 0, U, len(buf), root) + b"\0\0\0\0"
    p = tmp_path / "hand2.h5"
    p.write_bytes(bytes(buf))
    f = H.File(str(p))
    assert f.attrs["backend"] == text
    assert np.array_equal(np.asarray(f["data"]), vals) and np.asarray(f["data"]).dtype == np.int16


def test_truncated_and_foreign_files_raise_oserror(tmp_path):
    w = H.Writer()
    w.dataset("a/b", np.ones((4, 4), np.float32))
    p = str(tmp_path / "t.h5")
    w.save(p)
    blob = open(p, "rb").read()
    (tmp_path / "cut.h5").write_bytes(blob[:len(blob) // 2])
    with pytest.raises(OSError):
        np.asarray(H.File(str(tmp_path / "cut.h5"))["a/b"])
    (tmp_path / "no.h5").write_bytes(b"not hdf5 at all" * 10)
    with pytest.raises(OSError):
        H.File(str(tmp_path / "no.h5"))


# ---- weights_io on a stand-in model (the real models need the GPU for get_weights: tests/test_models_gpu.py) --------------
class _P:
    def __init__(self, name, shape):
        self.name, self.shape = name, tuple(shape)


class _N:
    def __init__(self, name, params):
        self.name, self.params = name, params


class _FakeModel:
    name = "fake"

    def __init__(self, spec, seed=0):
        rng = np.random.default_rng(seed)
        self.nodes = [_N(n, [_P(f"{n}/{s}", sh) for s, sh in ws]) for n, ws in spec]
        self.params = [p for n in self.nodes for p in n.params]
        self._w = [rng.standard_normal(p.shape).astype(np.float32) for p in self.params]

    def get_weights(self):
        return [w.copy() for w in self._w]

    def set_weights(self, ws):
        assert len(ws) == len(self.params)
        for w, p in zip(ws, self.params):
            assert tuple(w.shape) == p.shape, (p.name, w.shape)
        self._w = [np.asarray(w, np.float32).copy() for w in ws]


SPEC = [("conv2d", [("kernel", (3, 3, 3, 8)), ("bias", (8,))]),
        ("batch_normalization", [("gamma", (8,)), ("beta", (8,)), ("moving_mean", (8,)), ("moving_variance", (8,))]),
        ("separable_conv2d", [("depthwise_kernel", (3, 3, 8, 1)), ("pointwise_kernel", (1, 1, 8, 16)), ("bias", (16,))]),
        ("conv2d_1", [("kernel", (1, 1, 16, 2)), ("bias", (2,))]),
        ("conv2d_transpose", [("kernel", (3, 3, 4, 2)), ("bias", (4,))]),
        ("dense", [("kernel", (4, 5)), ("bias", (5,))])]


@pytest.mark.parametrize("suffix", [".h5", ".safetensors"])
def test_save_load_roundtrip_and_keras_layout(tmp_path, suffix):
    a, b = _FakeModel(SPEC, 1), _FakeModel(SPEC, 2)
    p = str(tmp_path / ("w" + suffix))
    WIO.save_weights(a, p)
    WIO.load_weights(b, p)
    for x, y in zip(a.get_weights(), b.get_weights()):
        assert np.array_equal(x, y)
    if suffix == ".h5":  # the layout keras.Model.load_weights walks
        f = H.File(p)
        assert [s.decode() for s in f.attrs["layer_names"]] == [n for n, _ in SPEC]
        assert f.attrs["backend"] == b"tensorflow"
        g = f["separable_conv2d"]
        wn = [s.decode() for s in g.attrs["weight_names"]]
        assert wn == ["separable_conv2d/depthwise_kernel:0", "separable_conv2d/pointwise_kernel:0", "separable_conv2d/bias:0"]
        assert np.asarray(g[wn[1]]).shape == (1, 1, 8, 16)
    with pytest.raises(OSError):
        WIO.load_weights(b, str(tmp_path / "missing.h5"))


def test_load_matches_layers_by_class_and_ordinal_when_the_order_differs(tmp_path):
    """A file whose layers come in another order (Keras sorts model.layers by graph depth) and whose uid counters started
    elsewhere (conv2d_40, conv2d_41 instead of conv2d, conv2d_1) still lands on the right layers; a real mismatch raises."""
    a = _FakeModel(SPEC, 3)
    ws = {p.name: w for p, w in zip(a.params, a.get_weights())}
    w = H.Writer()
    renamed = {"conv2d": "conv2d_40", "conv2d_1": "conv2d_41", "batch_normalization": "batch_normalization_7",
               "separable_conv2d": "separable_conv2d_2", "conv2d_transpose": "conv2d_transpose", "dense": "dense_9"}
    order = ["dense", "conv2d_1", "conv2d", "conv2d_transpose", "separable_conv2d", "batch_normalization"]
    w.attr("", "layer_names", np.array([renamed[n].encode() for n in order] + [b"input_1", b"activation_3"]))
    for n in order:
        node = next(x for x in a.nodes if x.name == n)
        w.group(renamed[n])
        names = [f"{renamed[n]}/{p.name.split('/')[1]}:0" for p in node.params]
        w.attr(renamed[n], "weight_names", np.array([s.encode() for s in names]))
        for p, s in zip(node.params, names):
            w.dataset(f"{renamed[n]}/{s}", ws[p.name])
    for n in ("input_1", "activation_3"):
        w.group(n)
        w.attr(n, "weight_names", np.array([], dtype="S1"))
    p = str(tmp_path / "k.h5")
    w.save(p)
    b = _FakeModel(SPEC, 4)
    WIO.load_weights(b, p)
    for x, y in zip(a.get_weights(), b.get_weights()):
        assert np.array_equal(x, y)
    other = _FakeModel(SPEC[:-1] + [("dense", [("kernel", (4, 6)), ("bias", (6,))])], 5)
    with pytest.raises(ValueError):
        WIO.load_weights(other, p)


def test_long_name_lists_are_split_like_keras(tmp_path):
    spec = [(f"conv2d_{i}" if i else "conv2d", [("kernel", (1, 1, 2, 2))]) for i in range(6000)]
    a, b = _FakeModel(spec, 1), _FakeModel(spec, 2)
    p = str(tmp_path / "many.h5")
    WIO.save_weights(a, p)
    f = H.File(p)
    assert "layer_names" not in f.attrs and "layer_names0" in f.attrs and "layer_names1" in f.attrs
    WIO.load_weights(b, p)
    assert all(np.array_equal(x, y) for x, y in zip(a.get_weights(), b.get_weights()))


def _write_keras_file(path, model, order, rename=None):
    """A Keras-layout weight file of `model` with its layers listed in `order` (and optionally renamed)."""
    rename = rename or {}
    ws = {p.name: w for p, w in zip(model.params, model.get_weights())}
    w = H.Writer()
    w.attr("", "layer_names", np.array([rename.get(n, n).encode() for n in order]))
    for n in order:
        node = next(x for x in model.nodes if x.name == n)
        fn = rename.get(n, n)
        w.group(fn)
        names = [f"{fn}/{p.name.split('/')[1]}:0" for p in node.params]
        w.attr(fn, "weight_names", np.array([s.encode() for s in names]))
        for p, s in zip(node.params, names):
            w.dataset(f"{fn}/{s}", ws[p.name])
    w.save(path)


# three equal-shaped parallel convolutions (the ASPP branches at rates 6 / 12 / 18 look like this) between two others
PARALLEL = [("conv2d", [("kernel", (1, 1, 4, 8)), ("bias", (8,))]),
            ("conv2d_1", [("kernel", (3, 3, 8, 8)), ("bias", (8,))]),
            ("conv2d_2", [("kernel", (3, 3, 8, 8)), ("bias", (8,))]),
            ("conv2d_3", [("kernel", (3, 3, 8, 8)), ("bias", (8,))]),
            ("conv2d_4", [("kernel", (1, 1, 24, 2)), ("bias", (2,))])]


def test_equal_shaped_parallel_layers_are_matched_by_name_not_by_a_positional_coincidence(tmp_path):
    """ADVICE r2 (weights_io.py:137): Keras writes layer_names in depth-sorted order.  With equal-shaped parallel layers a
    permuted file still FITS position by position - the old loader accepted that and put the branch weights on the wrong
    layers.  Names (and, with another uid offset, class ordinals) decide now."""
    a = _FakeModel(PARALLEL, 11)
    order = ["conv2d", "conv2d_3", "conv2d_1", "conv2d_2", "conv2d_4"]  # every position fits every shape
    p = str(tmp_path / "perm.h5")
    _write_keras_file(p, a, order)
    b = _FakeModel(PARALLEL, 12)
    WIO.load_weights(b, p)
    for x, y in zip(a.get_weights(), b.get_weights()):
        assert np.array_equal(x, y)
    # the same file written by a session whose Conv2D counter stood at 50: class ordinals pair the layers
    p2 = str(tmp_path / "perm50.h5")
    _write_keras_file(p2, a, order, rename={n: f"conv2d_{50 + i}" for i, (n, _) in enumerate(PARALLEL)})
    c = _FakeModel(PARALLEL, 13)
    WIO.load_weights(c, p2)
    for x, y in zip(a.get_weights(), c.get_weights()):
        assert np.array_equal(x, y)


def test_positional_matching_is_a_last_resort_and_warns(tmp_path):
    a = _FakeModel(PARALLEL, 21)
    p = str(tmp_path / "custom.h5")
    _write_keras_file(p, a, [n for n, _ in PARALLEL], rename={"conv2d": "stem", "conv2d_1": "aspp_r6", "conv2d_2": "aspp_r12",
                                                            "conv2d_3": "aspp_r18", "conv2d_4": "head"})
    b = _FakeModel(PARALLEL, 22)
    with pytest.warns(RuntimeWarning, match="BY POSITION"):
        WIO.load_weights(b, p)
    for x, y in zip(a.get_weights(), b.get_weights()):
        assert np.array_equal(x, y)
    # same names, another architecture: an error, not a positional retry
    other = _FakeModel(PARALLEL[:-1] + [("conv2d_4", [("kernel", (1, 1, 24, 3)), ("bias", (3,))])], 23)
    p3 = str(tmp_path / "named.h5")
    _write_keras_file(p3, a, [n for n, _ in PARALLEL])
    with pytest.raises(ValueError, match="conv2d_4"):
        WIO.load_weights(other, p3)
