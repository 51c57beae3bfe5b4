"""save_weights / load_weights (predict.py:21-49, train_model/DeepLabv3plus.py:778-780).

The reference's files are Keras-2 HDF5 weight files (`model.save_weights('x.h5')`):

    /                       attrs: layer_names [S..], backend, keras_version
    /<layer>/               attrs: weight_names [S..]   e.g. b'conv2d_3/kernel:0', b'conv2d_3/bias:0'
    /<layer>/<layer>/kernel:0   dataset, the array of `layer.get_weights()` (Conv HWIO, depthwise [kh,kw,C,1],
                                Conv2DTranspose [kh,kw,Cout,Cin], Dense [in,out], BN gamma/beta/moving_mean/moving_variance)

h5py is not available in this image, so both directions go through `h5lite` (a from-the-specification HDF5 subset).  A
path ending in `.h5` / `.hdf5` / `.keras.h5` is written in that layout, so a Keras model of the same architecture can
`load_weights` it; any other suffix gets the engine's native container (safetensors, one entry per weight).  Loading
sniffs the file: HDF5 signature -> Keras layout, else safetensors.  A missing file raises OSError, the only error the
reference handles (predict.py:23); an HDF5 file outside the supported subset raises OSError / NotImplementedError with the
reason.  PINNED against a real HDF5 library in both directions (libhdf5 1.10.6 through ctypes, tests/test_h5_libhdf5_cpu.py: the five
models' files written here read back value for value by libhdf5 / h5dump, and Keras-layout files assembled by libhdf5 at
DeepLabv3+ scale loaded here); NOT yet against a file written by Keras itself (none exists in this image).

Layer matching: `_match_layers` - exact layer names, else (layer class, ordinal within the class), position only as a
last resort and with a warning (Keras orders `model.layers` by graph depth, this engine by creation; a positional fit of
equal-shaped parallel layers can be a coincidence).
"""
from __future__ import annotations

import os
import re
from typing import List, Tuple

import numpy as np

from . import h5lite

_H5_SUFFIXES = (".h5", ".hdf5")
_ATTR_LIMIT = 60000  # Keras' HDF5_OBJECT_HEADER_LIMIT is 64512: longer name lists are split into name0, name1, ...


def _layers_with_weights(model) -> List[Tuple[str, list]]:
    return [(n.name, list(n.params)) for n in model.nodes if n.params]


def _save_name_list(w: h5lite.Writer, path: str, name: str, items: List[bytes]):
    arr = np.array(items) if items else np.array([], dtype="S1")
    if arr.nbytes <= _ATTR_LIMIT:
        w.attr(path, name, arr)
        return
    n = 2
    while any(c.nbytes > _ATTR_LIMIT for c in np.array_split(arr, n)):
        n += 1
    for i, c in enumerate(np.array_split(arr, n)):  # save_attributes_to_hdf5_group's chunking
        w.attr(path, f"{name}{i}", c)


def _load_name_list(attrs, name: str) -> List[str]:
    if name in attrs:
        vals = np.atleast_1d(attrs[name])
    else:
        vals, i = [], 0
        while f"{name}{i}" in attrs:
            vals.extend(np.atleast_1d(attrs[f"{name}{i}"]))
            i += 1
    return [v.decode("utf-8") if isinstance(v, (bytes, np.bytes_)) else str(v) for v in vals]


def save_weights(model, path):
    path = os.fspath(path)
    d = os.path.dirname(os.path.abspath(path))
    os.makedirs(d, exist_ok=True)
    ws = model.get_weights()
    if not path.lower().endswith(_H5_SUFFIXES):
        from safetensors.numpy import save_file
        tensors = {f"{i:05d}:{p.name}": np.ascontiguousarray(w) for i, (p, w) in enumerate(zip(model.params, ws))}
        save_file(tensors, path, metadata={"format": "building_detection_amd-v1", "model": model.name})
        return
    value = {id(p): w for p, w in zip(model.params, ws)}
    w = h5lite.Writer()
    layers = _layers_with_weights(model)
    _save_name_list(w, "", "layer_names", [name.encode("utf-8") for name, _ in layers])
    w.attr("", "backend", b"tensorflow")
    w.attr("", "keras_version", b"2.4.0")
    for name, params in layers:
        w.group(name)
        wnames = [f"{p.name}:0" for p in params]  # p.name = '<layer>/<weight>'
        _save_name_list(w, name, "weight_names", [n.encode("utf-8") for n in wnames])
        for p, wn in zip(params, wnames):
            w.dataset(f"{name}/{wn}", np.ascontiguousarray(value[id(p)], dtype=np.float32))
    w.save(path)


def _class_key(name: str) -> Tuple[str, int]:
    m = re.match(r"^(.*?)(?:_(\d+))?$", name)
    return m.group(1), int(m.group(2) or 0)


def _read_keras_h5(path: str) -> List[Tuple[str, List[np.ndarray]]]:
    f = h5lite.File(path)
    root = f
    if "layer_names" not in f.attrs and "layer_names0" not in f.attrs and "model_weights" in f.keys():
        root = f["model_weights"]  # a full `model.save()` file keeps the weights in this sub-group
    names = _load_name_list(root.attrs, "layer_names")
    if not names:
        raise OSError(f"{path}: an HDF5 file, but without the 'layer_names' attribute of a Keras weight file")
    out = []
    for name in names:
        g = root[name]
        wnames = _load_name_list(g.attrs, "weight_names")
        if wnames:
            out.append((name, [np.asarray(g[wn]) for wn in wnames]))
    return out


def load_weights(model, path):
    path = os.fspath(path)
    if not os.path.exists(path):
        raise OSError(f"Unable to open file (unable to open file: name = '{path}', errno = 2, error message = "
                      f"'No such file or directory')")
    if not h5lite.is_hdf5(path):
        from safetensors.numpy import load_file
        try:
            tensors = load_file(path)
        except Exception as e:  # neither container: report it the way the reference's caller handles (predict.py:23)
            raise OSError(f"{path}: neither an HDF5 (Keras) nor a safetensors weight file: {e}") from e
        keys = sorted(tensors)
        if len(keys) != len(model.params):
            raise ValueError(f"{path}: holds {len(keys)} weights, the model expects {len(model.params)}")
        for k, p in zip(keys, model.params):
            if k.split(":", 1)[-1] != p.name:
                raise ValueError(f"{path}: entry {k!r} does not belong to weight {p.name!r} - a file of another model?")
        model.set_weights([tensors[k] for k in keys])
        return
    file_layers = _read_keras_h5(path)
    ours = _layers_with_weights(model)
    pairs = _match_layers(path, file_layers, ours)
    value = {}
    for (_, fw), (_, ps) in pairs:
        for a, p in zip(fw, ps):
            value[id(p)] = np.asarray(a, dtype=np.float32)
    model.set_weights([value[id(p)] for p in model.params])


def _fits(pairs) -> bool:
    return all(len(fw) == len(ps) and all(tuple(a.shape) == tuple(p.shape) for a, p in zip(fw, ps)) for (_, fw), (_, ps) in pairs)


def _first_misfit(pairs):
    return next((a[0], b[0]) for a, b in pairs if not _fits([(a, b)]))


def _match_layers(path, file_layers, ours):
    """Pairs (file layer, model layer).  Keras writes `layer_names` in `model.layers` order - sorted by graph depth - while this
    engine lists layers in creation order, and branched graphs hold equal-shaped parallel layers (ASPP rates 6 / 12 / 18,
    the SK branches, HRNet's stages): a positional pairing can fit every shape by coincidence and still put weights on the
    wrong layers.  So the layer's identity decides, in this order:
      1. exact layer names, when the file's set of names is the model's;
      2. (layer class, ordinal within the class): `conv2d_7` is the 8th Conv2D created, whatever uid offset the saving
         session had reached - file and model are each ranked per class and paired rank by rank;
      3. position, only when the names carry no usable class structure (custom `name=` arguments), and with a warning.
    A pairing chosen by 1 or 2 whose shapes do not fit is an error (another architecture), never a reason to try 3."""
    if len(file_layers) != len(ours):
        raise ValueError(f"{path}: {len(file_layers)} layers with weights, the model has {len(ours)}")
    fnames, onames = [n for n, _ in file_layers], [n for n, _ in ours]
    if len(set(fnames)) == len(fnames) and set(fnames) == set(onames):
        by = dict(ours)
        pairs = [((n, fw), (n, by[n])) for n, fw in file_layers]
        if not _fits(pairs):
            bad = _first_misfit(pairs)
            raise ValueError(f"{path}: weight shapes of layer {bad[0]!r} do not match the model's layer of that name")
        return pairs

    def ranked(layers):
        by = {}
        for item in layers:
            cls, idx = _class_key(item[0])
            by.setdefault(cls, []).append((idx, item))
        return {cls: [it for _, it in sorted(v, key=lambda t: t[0])] for cls, v in by.items()}

    rf, ro = ranked(file_layers), ranked(ours)
    if set(rf) == set(ro) and all(len(rf[c]) == len(ro[c]) for c in rf):
        pairs = [(a, b) for c in rf for a, b in zip(rf[c], ro[c])]
        if not _fits(pairs):
            bad = _first_misfit(pairs)
            raise ValueError(f"{path}: weight shapes of file layer {bad[0]!r} do not match model layer {bad[1]!r}")
        return pairs
    pairs = list(zip(file_layers, ours))
    if not _fits(pairs):
        raise ValueError(f"{path}: layer classes differ from the model's (file {sorted((c, len(v)) for c, v in rf.items())}, "
                         f"model {sorted((c, len(v)) for c, v in ro.items())}) and the layers do not fit by position either: "
                         f"file layer {_first_misfit(pairs)[0]!r}")
    import warnings
    warnings.warn(f"{path}: layer names match neither the model's names nor its (class, ordinal) structure; weights were "
                  "assigned BY POSITION (every shape fits, but equal-shaped parallel layers may be permuted if the file was "
                  "written in another layer order)", RuntimeWarning, stacklevel=3)
    return pairs
