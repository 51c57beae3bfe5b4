"""SURVEY row f-3 / VERDICT r4 next #4: building_detection_amd/h5lite.py + weights_io.py against a REAL HDF5 implementation.

The image carries HDF5 1.10.6 (/opt/conda/lib/libhdf5.so.103, /opt/conda/bin/h5dump) - not h5py, not Keras.  tests/_libhdf5.py
binds the C library with ctypes and makes the calls h5py makes for Keras' save_weights / load_weights.  Two directions, at the
scale of the real models (the reference loads resnet34.h5 / hrnet.h5 / v3plus.h5 / scse.h5 / bam.h5, predict.py:21-49, and writes
epoch_N_weights.h5, train_model/DeepLabv3plus.py:778-780):

  (i)  h5lite -> libhdf5: each of the five models' weight files written by weights_io.save_weights is opened by libhdf5 and every
       dataset (class, size, byte order, layout, shape, VALUES), every `weight_names` / `layer_names*` / `backend` attribute is
       compared with get_weights(); h5dump -H walks the file too.
  (ii) libhdf5 -> h5lite: a Keras-2-layout file assembled by libhdf5 (old format = h5py's default: symbol-table groups with
       multi-node B-trees for 400+ layers, name lists split over layer_names0.. like Keras does above 64,512 bytes; also the
       full `model.save()` layout under /model_weights, h5py-3 variable-length scalar strings, and the newest file format) is
       loaded by weights_io.load_weights / h5lite.File.

The models' GRAPHS are the engine's own (zoo builders, host-only); their weights live in a host-side holder here because
get_weights() of the real Model needs the GPU (tests/test_models_gpu.py::test_save_load_weights runs the device round trip)."""
import os
import subprocess

import numpy as np
import pytest

from building_detection_amd import h5lite as H
from building_detection_amd import weights_io as WIO
from building_detection_amd import zoo

import _libhdf5 as L

pytestmark = pytest.mark.skipif(not L.available(), reason="no libhdf5 in this image (the build image has /opt/conda/lib/libhdf5.so.103)")


class _P:
    def __init__(self, name, shape):
        self.name, self.shape = name, tuple(shape)


class _N:
    def __init__(self, name, params):
        self.name, self.params = name, params


class Holder:
    """the (layer, weight) structure of a real engine graph with host-side values"""

    def __init__(self, spec, seed, name="holder"):
        self.name = name
        self.nodes = [_N(n, [_P(pn, sh) for pn, sh in ps]) for n, ps in spec]
        self.params = [p for n in self.nodes for p in n.params]
        rng = np.random.default_rng(seed)
        self._w = [(rng.random(p.shape, dtype=np.float32) - 0.5) for p in self.params]

    def get_weights(self):
        return self._w

    def set_weights(self, ws):
        assert len(ws) == len(self.params)
        for w, p in zip(ws, self.params):
            assert tuple(w.shape) == p.shape and w.dtype == np.float32, (p.name, w.shape, w.dtype)
        self._w = list(ws)


def _spec_of(kind):
    """(all layer names in creation order incl. the weightless ones, [(layer, [(weight name, shape)])] of the real graph)"""
    m = zoo.BUILDERS[kind]((512, 512, 3), 2) if kind in ("v3plus", "bam") else zoo.BUILDERS[kind]((512, 512, 3))
    return [n.name for n in m.nodes], [(n.name, [(p.name, tuple(p.shape)) for p in n.params]) for n in m.nodes if n.params]


def test_library_is_the_real_one():
    assert L.version() >= (1, 8, 0)
    assert L.H5DUMP and os.path.exists(L.H5DUMP)


@pytest.mark.parametrize("kind", ["v3plus", "bam", "scse", "res34", "hrnet"])
def test_files_written_by_h5lite_are_read_by_libhdf5(tmp_path, kind):
    _, spec = _spec_of(kind)
    a = Holder(spec, seed=11)
    p = str(tmp_path / f"{kind}.h5")
    WIO.save_weights(a, p)
    r = L.Reader(p)
    try:
        names = [n for n, _ in spec]
        ra = r.attr_names("/")
        if "layer_names" in ra:
            got = r.attr("/", "layer_names")
        else:   # Keras' split form
            got, i = [], 0
            while f"layer_names{i}" in ra:
                got += r.attr("/", f"layer_names{i}")
                i += 1
        assert [g.decode() for g in got] == names
        assert r.attr("/", "backend") == b"tensorflow" and r.attr("/", "keras_version").startswith(b"2.")
        assert sorted(r.keys("/")) == sorted(names)
        value = {pp.name: w for pp, w in zip(a.params, a.get_weights())}
        n_data = 0
        for name, ps in spec:
            wn = [w.decode() for w in r.attr("/" + name, "weight_names")]
            assert wn == [f"{pn}:0" for pn, _ in ps]                       # '<layer>/<weight>:0'
            assert r.keys("/" + name) == [name]                            # /<layer>/<layer>/<weight>:0 - the nested group
            assert sorted(r.keys(f"/{name}/{name}")) == sorted(w.split("/", 1)[1] for w in wn)
            for (pn, shape), w in zip(ps, wn):
                d = r.dataset(f"/{name}/{w}")
                assert d["class"] == L.H5T_FLOAT and d["size"] == 4 and d["little_endian"] and d["contiguous"], (w, d)
                assert d["shape"] == tuple(shape), (w, d["shape"], shape)
                assert np.array_equal(d["data"], value[pn]), w
                n_data += 1
        assert n_data == len(a.params)
    finally:
        r.close()
    # the library's own tool walks the whole file: every dataset is announced as little-endian IEEE float32
    out = subprocess.run([L.H5DUMP, "-H", p], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert out.returncode == 0, out.stderr.decode()[-400:]
    txt = out.stdout.decode()
    assert txt.count("DATASET \"") == len(a.params) and txt.count("H5T_IEEE_F32LE") >= len(a.params)
    # ... and the values of one tensor as h5dump prints them (text, so a small one: the first bias)
    lname, (pn, shape) = next((n, q) for n, ps in spec for q in ps if len(q[1]) == 1)
    out = subprocess.run([L.H5DUMP, "-d", f"/{lname}/{pn}:0", "-y", "-w", "0", "-m", "%.9g", p], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert out.returncode == 0, out.stderr.decode()[-400:]
    body = out.stdout.decode().split("DATA {", 1)[1].split("}", 1)[0]
    vals = np.array([float(t) for t in body.replace("\n", " ").split(",") if t.strip()], np.float64).astype(np.float32)
    assert np.array_equal(vals, value[pn])


def _keras_file(w, root, layers_all, spec, value, rename=lambda n: n, split_names=False, h5py3=False):
    """save_weights_to_hdf5_group (tf.keras saving/hdf5_format.py) through libhdf5: EVERY layer gets a group and a
    weight_names attribute (the weightless ones an empty float64 array, what `attrs[name] = []` makes); name lists above
    HDF5_OBJECT_HEADER_LIMIT = 64512 bytes are split into name0, name1, ..."""
    by = dict(spec)
    names = [rename(n).encode() for n in layers_all]

    def name_list(path, attr, items):
        arr = np.array(items) if items else None
        if arr is None:
            w.attr_array(path, attr, np.zeros((0,), "<f8"))
        elif arr.nbytes <= 64512 and not split_names:
            w.attr_strings(path, attr, items)
        else:
            k = 2
            while any(c.nbytes > 64512 for c in np.array_split(arr, k)):
                k += 1
            for i, c in enumerate(np.array_split(arr, k)):
                w.attr_strings(path, f"{attr}{i}", list(c))
    if root:
        w.group(root)
    name_list(root or "/", "layer_names", names)
    w.attr_scalar_string(root or "/", "backend", b"tensorflow", variable=h5py3)
    w.attr_scalar_string(root or "/", "keras_version", b"2.4.0", variable=h5py3)
    for n in layers_all:
        g = f"{root}/{rename(n)}"
        w.group(g)
        ps = by.get(n, [])
        wn = [f"{rename(n)}/{pn.split('/', 1)[1]}:0" for pn, _ in ps]
        name_list(g, "weight_names", [s.encode() for s in wn])
        for (pn, _), s in zip(ps, wn):
            w.dataset(f"{g}/{s}", value[pn])


def test_a_keras_layout_file_made_by_libhdf5_at_deeplab_scale_loads(tmp_path):
    """DeepLabv3+: every layer of the engine's graph in the file (339 nodes, 203 of them with weights; 652 datasets, 468 of
    them trainable tensors) - the root group's symbol table is a multi-node B-tree (32 entries per SNOD at the default K), the
    layer groups carry the nested /<layer>/<layer>/ sub-group, the weightless layers an empty weight_names attribute."""
    layers_all, spec = _spec_of("v3plus")
    assert len(layers_all) > 300 and len(spec) == 203 and sum(len(ps) for _, ps in spec) == 652
    a = Holder(spec, seed=5)
    value = {pp.name: w for pp, w in zip(a.params, a.get_weights())}
    p = str(tmp_path / "v3plus_keras.h5")
    w = L.Writer(p)
    _keras_file(w, "", layers_all, spec, value)
    w.close()
    assert H.is_hdf5(p)
    f = H.File(p)
    assert len(f.keys()) == len(layers_all)
    assert f.attrs["backend"] == b"tensorflow"
    b = Holder(spec, seed=6)
    WIO.load_weights(b, p)
    for pp, x, y in zip(a.params, a.get_weights(), b.get_weights()):
        assert np.array_equal(x, y), pp.name


def test_a_file_of_another_session_with_split_name_lists_loads(tmp_path):
    """The same graph saved by a session whose uid counters had moved on (conv2d_173 ... instead of conv2d ...: the class /
    ordinal matching of weights_io._match_layers), the layer list split over layer_names0 / layer_names1 (Keras does that
    above 64,512 bytes; forced here), scalar strings as h5py 3 writes them (variable length, UTF-8: the global heap)."""
    layers_all, spec = _spec_of("hrnet")
    a = Holder(spec, seed=7)
    value = {pp.name: w for pp, w in zip(a.params, a.get_weights())}

    def rename(n):
        cls, idx = WIO._class_key(n)
        return f"{cls}_{idx + 173}"
    p = str(tmp_path / "hrnet_other_session.h5")
    w = L.Writer(p)
    _keras_file(w, "", layers_all, spec, value, rename=rename, split_names=True, h5py3=True)
    w.close()
    f = H.File(p)
    assert "layer_names" not in f.attrs and "layer_names0" in f.attrs and "layer_names1" in f.attrs
    assert f.attrs["backend"] in (b"tensorflow", "tensorflow")
    b = Holder(spec, seed=8)
    WIO.load_weights(b, p)
    for pp, x, y in zip(a.params, a.get_weights(), b.get_weights()):
        assert np.array_equal(x, y), pp.name


def test_a_full_model_save_file_and_the_newest_file_format(tmp_path):
    """`model.save('x.h5')` keeps the same layout under /model_weights (+ model_config / training_config attributes and an
    /optimizer_weights group); and a file written with libver='latest' (superblock 3, version-2 object headers, link
    messages) reads the same as long as its groups are compact - a densely stored group (more than 8 links under the newest
    format; never what Keras' default produces) is refused with the reason, not misread."""
    spec = [("conv2d", [("conv2d/kernel", (3, 3, 3, 8)), ("conv2d/bias", (8,))]),
            ("batch_normalization", [("batch_normalization/gamma", (8,)), ("batch_normalization/beta", (8,)),
                                     ("batch_normalization/moving_mean", (8,)), ("batch_normalization/moving_variance", (8,))]),
            ("dense", [("dense/kernel", (8, 2)), ("dense/bias", (2,))])]
    layers_all = ["input_1", "conv2d", "batch_normalization", "activation", "dense"]
    a = Holder(spec, seed=9)
    value = {pp.name: w for pp, w in zip(a.params, a.get_weights())}
    for latest in (False, True):
        p = str(tmp_path / f"full_{int(latest)}.h5")
        w = L.Writer(p, latest=latest)
        _keras_file(w, "/model_weights", layers_all, spec, value)
        w.attr_scalar_string("/", "model_config", b'{"class_name": "Functional"}')
        w.attr_scalar_string("/", "keras_version", b"2.4.0")
        w.group("/optimizer_weights")
        w.dataset("/optimizer_weights/Adam/iter:0", np.array(7, "<i8"))
        w.close()
        b = Holder(spec, seed=10)
        WIO.load_weights(b, p)
        for x, y in zip(a.get_weights(), b.get_weights()):
            assert np.array_equal(x, y)
        f = H.File(p)
        assert int(np.asarray(f["optimizer_weights/Adam/iter:0"])) == 7
    p = str(tmp_path / "dense_links.h5")
    w = L.Writer(p, latest=True)
    for i in range(40):
        w.dataset(f"/d{i}", np.float32(i))
    w.close()
    with pytest.raises((NotImplementedError, OSError), match="dense|fractal"):
        H.File(p).keys()


def test_dtypes_shapes_and_attributes_round_trip_through_both_implementations(tmp_path):
    """Everything h5lite's writer can emit, read by libhdf5; everything of that which libhdf5 writes, read by h5lite."""
    rng = np.random.default_rng(3)
    arrs = {"f32": rng.standard_normal((3, 1, 5)).astype("<f4"), "f64": rng.standard_normal((7,)), "i32": np.arange(-5, 5, dtype="<i4"),
            "i64": np.array(2 ** 40 + 3, "<i8"), "u8": np.arange(200, dtype=np.uint8).reshape(10, 20), "empty": np.zeros((0, 4), "<f4")}
    p1, p2 = str(tmp_path / "lite.h5"), str(tmp_path / "real.h5")
    w = H.Writer()
    for k, v in arrs.items():
        w.dataset(f"grp/{k}", v)
    w.attr("grp", "names", np.array([b"alpha", b"be", b"gamma_delta"]))
    w.attr("grp", "vec", np.arange(4, dtype=np.float32))
    w.attr("", "n", np.int64(-9))
    w.save(p1)
    r = L.Reader(p1)
    for k, v in arrs.items():
        d = r.dataset(f"/grp/{k}")
        assert d["shape"] == v.shape and d["size"] == v.dtype.itemsize and np.array_equal(d["data"], v), k
    assert r.attr("/grp", "names") == [b"alpha", b"be", b"gamma_delta"]
    assert np.array_equal(r.attr("/grp", "vec"), np.arange(4, dtype=np.float32)) and int(r.attr("/", "n")) == -9
    r.close()
    w = L.Writer(p2)
    for k, v in arrs.items():
        w.dataset(f"/grp/{k}", v)
    w.attr_strings("/grp", "names", [b"alpha", b"be", b"gamma_delta"])
    w.attr_array("/grp", "vec", np.arange(4, dtype="<f4"))
    w.attr_array("/", "n", np.array(-9, "<i8"))
    w.close()
    f = H.File(p2)
    for k, v in arrs.items():
        got = np.asarray(f[f"grp/{k}"])
        assert got.shape == v.shape and got.dtype == v.dtype and np.array_equal(got, v), k
    assert [s for s in f["grp"].attrs["names"]] == [b"alpha", b"be", b"gamma_delta"]
    assert np.array_equal(f["grp"].attrs["vec"], np.arange(4, dtype=np.float32)) and int(f.attrs["n"]) == -9
