"""The reference's training input pipeline, host side: `decode_img`, `decode_lbel`,
`train_data_gen`, `val_data_gen` of train_model/DeepLabv3plus.py:32-153 (the other four training scripts carry
the same text), restated without OpenCV / TensorFlow - names, arguments, yield shapes and dtypes as there.

What is exact and what is not (OpenCV is absent here, so nothing below could be checked against cv2 itself):
  * file decoding goes through Pillow (`cv.imread` = 8-bit, 3 channels, BGR; then BGR2RGB): identical for the 8-bit
    RGB / gray PNG / TIFF tiles of the WHU set;
  * `cv.resize(img, (512, 512))` is the identity for 512x512 tiles - the case of the data set; other sizes go through
    OpenCV's own 11-bit fixed-point INTER_LINEAR arithmetic (`_resize_bilinear_u8`; on the GPU `sg_resize_linear_u8`),
    restated from the published algorithm and held bit-exact to oracle/input_pipeline.py;
  * `cv.cvtColor(BGR2GRAY)` uses cv2's published 14-bit integer weights (4899 R + 9617 G + 1868 B + 8192) >> 14,
    which is the identity on grey label images;
  * `tf.keras.utils.to_categorical(label, 2)` truncates label / 255 to int, so only pixels equal to 255 are class 1;
  * the edge bands are 5 iterations of a 3x3 erode / dilate (borders do not erode / dilate: cv2's default border value),
    `building_detection_amd.data.edge_weight_channels`; with `engine=` they are built on the GPU instead
    (`sg_edge_labels`, bit-identical, tests/test_ops_gpu.py::test_edge_labels_match_generator);
  * `label_smooth=True` raises: the reference reads `p_label_smooth` / `f_label_smooth`, which no file defines
    (`DeepLabv3plus.py:74`), so that branch cannot run there either.
"""
from __future__ import annotations

import itertools
import queue
import threading

import numpy as np

from .data import edge_weight_channels

SIZE = 512  # the reference resizes every tile and label to 512 x 512 (`:35`, `:45`)


def _imread_bgr_order_free(path) -> np.ndarray:
    """uint8 [H,W,3] in RGB order (= cv.cvtColor(cv.imread(path), cv.COLOR_BGR2RGB))."""
    from PIL import Image
    with Image.open(path) as im:
        if im.mode in ("I;16", "I;16B", "I;16L", "I"):  # cv.imread's default flag reduces 16-bit data to 8 bits
            a = np.asarray(im, np.uint32)
            im = Image.fromarray((a >> 8).astype(np.uint8) if a.max() > 255 else a.astype(np.uint8))
        return np.asarray(im.convert("RGB"), np.uint8)


def _cv_coeffs(n_in: int, n_out: int, clamp: bool):
    """OpenCV's INTER_LINEAR taps for 8-bit images (resize.cpp): first source index and the two 11-bit weights."""
    f = ((np.arange(n_out, dtype=np.float64) + 0.5) * (n_in / n_out) - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    if clamp:  # columns: index clamped and weight reset; rows keep their weights and clamp at the read
        edge = (s < 0) | (s >= n_in - 1)
        f = np.where(edge, np.float32(0), f)
        s = np.clip(s, 0, n_in - 1)
    c0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
    c1 = np.rint(f * np.float32(2048)).astype(np.int64)
    return s, c0, c1


def _resize_bilinear_u8(img: np.ndarray, size=(SIZE, SIZE)) -> np.ndarray:
    """cv.resize(img, size) with the default INTER_LINEAR, in OpenCV's fixed-point arithmetic for 8-bit pixels: 11-bit
    coefficients, int32 horizontal pass, vertical pass (((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2 (the
    vectorised row kernel every SIMD build of OpenCV runs on whole vectors; 512 * channels is a multiple of any vector
    length).  Same size = a copy (coefficients 2048 / 0); an exact 2x downscale is the fast INTER_AREA that resize()
    substitutes.  The GPU twin is sg_resize_linear_u8; both are held bit-exact to oracle/input_pipeline.py (cv2 itself is
    absent from this image: the restatement follows the published algorithm, unpinned against a cv2 build)."""
    h, w = img.shape[:2]
    ow, oh = size
    if (h, w) == (oh, ow):
        return img.copy()
    a = img.reshape(h, w, -1).astype(np.int64)
    if h == 2 * oh and w == 2 * ow:
        out = (a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2
        return out.astype(np.uint8).reshape((oh, ow) + img.shape[2:])
    sx, a0, a1 = _cv_coeffs(w, ow, True)
    sy, b0, b1 = _cv_coeffs(h, oh, False)
    hor = a[:, sx] * a0[None, :, None] + a[:, np.minimum(sx + 1, w - 1)] * a1[None, :, None]   # [h, ow, c], scale 2^11
    d0, d1 = hor[np.clip(sy, 0, h - 1)], hor[np.clip(sy + 1, 0, h - 1)]
    out = (((b0[:, None, None] * (d0 >> 4)) >> 16) + ((b1[:, None, None] * (d1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8).reshape((oh, ow) + img.shape[2:])


def decode_img(img_path) -> np.ndarray:
    """`DeepLabv3plus.py:32-39`: RGB tile, resized to 512x512, float32, / 127.5 - 1  ->  [512,512,3] in [-1, 1]."""
    img = _resize_bilinear_u8(_imread_bgr_order_free(img_path))
    return np.array(img, np.float32) / 127.5 - 1


def decode_lbel(label_path) -> np.ndarray:
    """`DeepLabv3plus.py:42-50` (the reference's spelling): grey label, resized, float32 / 255  ->  [512,512,1]."""
    rgb = _imread_bgr_order_free(label_path).astype(np.int32)
    gray = ((rgb[..., 0] * 4899 + rgb[..., 1] * 9617 + rgb[..., 2] * 1868 + 8192) >> 14).astype(np.uint8)
    gray = _resize_bilinear_u8(gray)
    return (np.array(gray[..., np.newaxis], np.float32)) / 255


def to_categorical(label: np.ndarray, num_classes: int = 2) -> np.ndarray:
    """tf.keras.utils.to_categorical: integer truncation of the values, trailing unit axis dropped, float32 one-hot."""
    y = np.array(label, dtype="int")
    if y.ndim > 1 and y.shape[-1] == 1:
        y = y.reshape(y.shape[:-1])
    out = np.zeros(y.shape + (num_classes,), np.float32)
    np.put_along_axis(out, y[..., None], 1.0, axis=-1)
    return out


def _sample(img, seg, label_smooth, loss, engine):
    image = decode_img(img)
    label = decode_lbel(seg)
    one_hot = to_categorical(label, num_classes=2)
    if label_smooth:
        raise NameError("label_smooth=True reads p_label_smooth / f_label_smooth, which the reference never defines "
                        "(train_model/DeepLabv3plus.py:74): that branch cannot run there either")
    if loss == "edge_focal_loss":
        lab2 = np.squeeze(label)
        if engine is not None:  # the same four channels from the GPU kernel (one-hot of label == 1, f_edge, p_edge)
            import torch
            y = engine.edge_labels(torch.from_numpy(np.ascontiguousarray(lab2[None])).to(engine.device))
            return image, y[0].cpu().numpy().astype(np.float64)
        f_edge, p_edge = edge_weight_channels(lab2)
        one_hot = np.concatenate((one_hot, f_edge[..., np.newaxis], p_edge[..., np.newaxis]), axis=-1)  # float64, as there
    return image, one_hot


def _gen(img_path, lab_path, BATCH_SIZE, label_smooth, loss, engine):
    images, label = img_path, lab_path
    images.sort()  # in place, as the reference does to its caller's lists
    label.sort()
    zipped = itertools.cycle(zip(images, label))
    while True:
        x_train, y_train = [], []
        for _ in range(BATCH_SIZE):
            img, seg = next(zipped)
            x, y = _sample(img, seg, label_smooth, loss, engine)
            x_train.append(x)
            y_train.append(y)
        yield np.array(x_train), np.array(y_train)


def train_data_gen(img_path, lab_path, BATCH_SIZE, label_smooth=False, loss="edge_focal_loss", engine=None):
    """`DeepLabv3plus.py:53-107`: endless generator of (x float32 [N,512,512,3], y [N,512,512,4] float64 with
    loss="edge_focal_loss", else [N,512,512,2] float32) over the sorted, cycled (image, label) pairs.
    `engine` (an `ops.Engine`, not in the reference): build the label channels with the GPU kernel."""
    return _gen(img_path, lab_path, BATCH_SIZE, label_smooth, loss, engine)


def val_data_gen(img_path, lab_path, BATCH_SIZE, label_smooth=False, loss="edge_focal_loss", engine=None):
    """`DeepLabv3plus.py:110-153`: the same generator over the validation lists."""
    return _gen(img_path, lab_path, BATCH_SIZE, label_smooth, loss, engine)


# ---- not in the reference: a non-blocking feed for the GPU -------------------------------------------------------------------
class Prefetcher:
    """Runs any generator (train_data_gen, val_data_gen, ...) in a background thread, `depth` batches ahead, so that file
    decoding overlaps the training step instead of sitting between two steps as in the reference's synchronous loop
    (DeepLabv3plus.py:844: fit_generator pulls `next(generator)` on the training thread).  Iteration order and values are
    the generator's own; an exception in the generator is re-raised by `next()`; `close()` stops the thread."""

    _END = object()

    def __init__(self, generator, depth: int = 2):
        self._gen = generator
        self._q: "queue.Queue" = queue.Queue(maxsize=max(1, int(depth)))
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()

    def _put(self, item) -> bool:
        """Blocking put that gives up when close() was called (a consumer that has gone away must not pin this thread -
        and, on the GPU path, its pinned batches - forever).  True when the item was queued."""
        while not self._stop.is_set():
            try:
                self._q.put(item, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    def _run(self):
        try:
            for item in self._gen:
                if not self._put(item):
                    return
            self._put(self._END)
        except BaseException as e:  # hand the failure to the consumer
            self._put(e)

    def __iter__(self):
        return self

    def __next__(self):
        final = getattr(self, "_final", None)
        if final is None:
            item = self._q.get()
            if not (item is self._END or isinstance(item, BaseException)):
                return item
            final = self._final = item   # remembered, not re-queued (a consumer must never block on a put of its own)
        if final is self._END:
            raise StopIteration
        raise final

    def close(self):
        """Stops the producer and joins it: the queue is drained until the thread has really exited (a put racing with a
        single drain could otherwise refill it and leave the join to time out)."""
        self._stop.set()
        while self._thread.is_alive():
            try:
                self._q.get(timeout=0.05)
            except queue.Empty:
                pass
            self._thread.join(timeout=0.05)
        while True:
            try:
                self._q.get_nowait()
            except queue.Empty:
                break

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _decode_u8(img, seg, resize=True):
    """The file-side half of decode_img / decode_lbel: uint8 RGB [h,w,3] and uint8 grey [h,w]; resized to 512 x 512 on the
    host unless `resize` is False (then the caller resizes on the device)."""
    rgb = _imread_bgr_order_free(img)
    lab = _imread_bgr_order_free(seg).astype(np.int32)
    gray = ((lab[..., 0] * 4899 + lab[..., 1] * 9617 + lab[..., 2] * 1868 + 8192) >> 14).astype(np.uint8)
    if resize:
        return _resize_bilinear_u8(rgb), _resize_bilinear_u8(gray)
    return rgb, gray


def device_data_gen(img_path, lab_path, BATCH_SIZE, engine, depth: int = 2, workers: int = 4):
    """train_data_gen for the GPU (loss="edge_focal_loss"): the same sorted, cycled (image, label) pairs, but
      * files are decoded by `workers` threads, `depth` batches ahead of the consumer (Prefetcher),
      * the batch crosses PCIe as uint8 pixels (a quarter of the fp32 bytes), from pinned memory,
      * tiles that are not 512 x 512 are resized on the device (sg_resize_linear_u8: cv.resize's fixed-point arithmetic)
        when a batch's files share one size, else on the host by the same arithmetic,
      * normalisation (`/ 127.5 - 1`, `/ 255`: sg_u8_to_f32) and the four label channels (sg_edge_labels) run on the device.
    Yields DEVICE tensors (x float32 [N,512,512,3], y float32 [N,512,512,4]) that `fit_generator` / `train_on_batch` take as
    they are; values are bit-identical to train_data_gen's (tests/test_pipeline_gpu.py).  y is float32 where the reference
    yields float64: Keras casts y_true to y_pred's float32 before the loss anyway."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    images, label = img_path, lab_path
    images.sort()
    label.sort()

    def host_batches():
        zipped = itertools.cycle(zip(images, label))
        with ThreadPoolExecutor(max_workers=max(1, int(workers))) as pool:
            while True:
                pairs = [next(zipped) for _ in range(BATCH_SIZE)]
                dec = list(pool.map(lambda p: _decode_u8(*p, resize=False), pairs))
                if len({d[0].shape for d in dec} | {d[1].shape + (3,) for d in dec}) != 1:  # mixed sizes: host resize
                    dec = [(_resize_bilinear_u8(r), _resize_bilinear_u8(g)) for r, g in dec]
                xb = torch.from_numpy(np.stack([d[0] for d in dec]))
                lb = torch.from_numpy(np.stack([d[1] for d in dec]))
                if torch.cuda.is_available():
                    xb, lb = xb.pin_memory(), lb.pin_memory()
                yield xb, lb

    feed = Prefetcher(host_batches(), depth)
    try:
        for xb, lb in feed:
            xd = xb.to(engine.device, non_blocking=True)
            ld = lb.to(engine.device, non_blocking=True)
            if tuple(xd.shape[1:3]) != (SIZE, SIZE):
                xd = engine.resize_linear_u8(xd, SIZE, SIZE)
                ld = engine.resize_linear_u8(ld, SIZE, SIZE)
            x = engine.u8_to_f32(xd, 127.5, 1.0)
            y = engine.edge_labels(engine.u8_to_f32(ld, 255.0, 0.0))
            yield x, y
    finally:
        feed.close()
