"""Drop-in for the reference's inference/detector.py:5-60.

    detector = Detector(model_path)                     # weights .npz (+ config)
    boxes, labels, scores = detector(image_uint8_HW3, score_threshold=0.1)

Same constructor keywords, same return order (boxes, labels, scores), boxes normalised
[ymin, xmin, ymax, xmax].  `model_path` is the reference's frozen graph (`inference/model.pb`), one of the
checkpoints its training leaves behind (a `model.ckpt-N` prefix, the `model_dir`, or the SavedModel directory of
create_pb.py's export/ -- for a user who cannot run create_pb.py without TensorFlow), or this build's own weight
container (a .npz keyed by the reference's TF variable names, see variables.py); the JSON config is the
reference's own file (config_mobilenet.json / config_shufflenet.json).
"""
import json
import os

import numpy as np

from .config import load_config
from .ssd import Engine, _torch
from .pb_import import load_pb_weights
from .ckpt_import import load_ckpt_weights, resolve_checkpoint
from .variables import load_weights


class Detector:
    def __init__(self, model_path, gpu_memory_fraction=0.25, visible_device_list='0',
                 config=None, precision=None, use_ema=False):
        """
        Arguments:
            model_path: path to the reference's frozen graph (.pb, read without TensorFlow), to a
                TensorFlow checkpoint of the reference's training (prefix `model.ckpt-N`, its .index /
                .data file, the model_dir with its `checkpoint` state file, or a SavedModel directory:
                ckpt_import.py), to this build's weight file (.npz), or a dict {variable name: ndarray}.
            gpu_memory_fraction: accepted for compatibility and ignored (the library
                allocates exactly the arena the network needs).
            visible_device_list: a string like the reference's; the first entry is the HIP
                device index.
            config: path to the reference's JSON config or a dict; default: `config.json`
                next to `model_path` (the reference reads 'config.json', create_pb.py:18).
            precision: "f32" (every convolution an exact fp32 chain, bit-identical to the CPU
                oracle), "f16x3" (dense convolutions on split-fp16 operands: fp32-chain accuracy,
                2.3x the throughput; should an activation ever leave the fp16 range the call is
                transparently repeated in f32) or None = the library default (SSD_PRECISION
                environment variable, else "f32").
            use_ema: checkpoints only -- take the variables' exponential moving averages (what the
                evaluation inside train.py restores, model.py:148-161) instead of the raw variables (what
                create_pb.py freezes: export_savedmodel restores the checkpoint as it is).
        """
        if isinstance(model_path, dict):
            weights = model_path
            if config is None:
                raise ValueError("config is required when weights are passed as a dict")
        else:
            ckpt = None if str(model_path).endswith((".pb", ".npz")) else resolve_checkpoint(model_path)
            if ckpt is None and not os.path.isfile(model_path):
                raise FileNotFoundError(model_path)       # tf.gfile.GFile would raise too
            if config is None:
                where = model_path if os.path.isdir(model_path) else os.path.dirname(os.path.abspath(model_path))
                config = os.path.join(where, "config.json")
            if ckpt is not None:                           # train.py's model_dir / create_pb.py's export folder
                weights = load_ckpt_weights(ckpt, load_config(config), use_ema=use_ema)
            elif str(model_path).endswith(".pb"):          # the reference's frozen graph (create_pb.py:57-85)
                weights = load_pb_weights(model_path, load_config(config))
            else:
                weights = load_weights(model_path)
        self.params = load_config(config)
        device = int(str(visible_device_list).split(",")[0])
        self.engine = Engine(self.params, weights, device=device, precision=precision)
        self.device = device

    def close(self):
        """Frees the engine's device memory (weights, every cached layer plan, staging buffers) now instead of at garbage
        collection; the reference's Detector has no counterpart (its tf.Session lives as long as the object)."""
        self.engine.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def _detect_views(self, images):
        """The graph outputs of one host batch as numpy VIEWS of the engine's pinned result block (valid until the next
        call); the caller holds self.engine.lock."""
        out = self.engine.detect_host(images)
        if self.engine.precision == "f16x3" and self.engine.status() & 1:
            # an activation exceeded +-65504 and was clamped (ssd_hip.h ssd_status): these results are not
            # trustworthy -- this detector continues in the exact mode
            import warnings
            warnings.warn("single-shot-detector_amd: activation outside the fp16 range in precision mode "
                          "f16x3; switching this Detector to f32")
            self.engine.set_precision("f32")
            out = self.engine.detect_host(images)
        return out

    def detect_batch(self, images):
        """images: uint8 ndarray [B,H,W,3] (or CUDA tensor), any H, W (the graph's
        resize_keeping_aspect_ratio is fused into the first kernel) -> the graph outputs
        (boxes [B,T,4], labels [B,T], scores [B,T], num_boxes [B]) as numpy arrays
        (model.py:70-73).  Thread-safe, like sess.run on a shared tf.Session (inference/detector.py:34,52)."""
        torch = _torch()
        if isinstance(images, torch.Tensor):
            with self.engine.lock:
                return tuple(t.cpu().numpy() for t in self.engine.forward_cached(images))
        with self.engine.lock:
            return tuple(np.array(v) for v in self._detect_views(images))

    def detect_many(self, images, score_threshold=0.1, max_batch=32):
        """`__call__` for a LIST of images of any sizes, batched: the images are grouped by the size the network sees for them
        (resize_keeping_aspect_ratio, pipeline.py:138-194: 480x640, 375x500, 333x500 all become 640x896), every group runs as
        batches of up to `max_batch` frames of DIFFERENT source sizes (ssd_forward_mixed_host: per-frame geometry in the first
        kernel, per-image box_scaler in the last), and the results come back in the order of `images` as (boxes, labels, scores)
        per image -- each bit for bit what `self(image, score_threshold)` returns.  The reference loops one `sess.run` per image
        (inference/evaluate_on_COCO.ipynb:125-150); this is that loop at batch throughput.  Mode f32.
        Batch sizes are `max_batch` and its halvings only (a group of 56 at max_batch 32 runs as 32 + 16 + 8): the library keeps
        one layer plan per (network shape, batch size), and a stream of arbitrary remainders must not build one for every size.
        Two batches are in flight: batch k + 1 is staged and uploaded while batch k computes."""
        if self.engine.precision != "f32":
            return [self(im, score_threshold) for im in images]
        imgs = [np.asarray(im) for im in images]
        max_batch = max(1, min(int(max_batch), self.engine.MIXED_MAX))
        sizes = []
        b = max_batch
        while b >= 1:
            sizes.append(b)
            b //= 2
        groups = {}
        for i, im in enumerate(imgs):
            if im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] != 3:
                raise ValueError("every image must be a uint8 array of shape [height, width, 3]")
            groups.setdefault(self.engine.network_shape(im.shape[0], im.shape[1]), []).append(i)
        parts = []
        for idx in groups.values():
            k = 0
            for b in sizes:
                while len(idx) - k >= b:
                    parts.append(idx[k:k + b])
                    k += b
        out = [None] * len(imgs)

        def consume(pending):
            slot, part = pending
            slot["done"].synchronize()
            boxes, labels, scores, num = slot["host"]
            for j, i in enumerate(part):
                n = int(num[j])
                keep = scores[j][:n] > score_threshold        # inference/detector.py:54-58
                out[i] = (boxes[j][:n][keep], labels[j][:n][keep], scores[j][:n][keep])

        with self.engine.lock:
            pending = None
            for n, part in enumerate(parts):
                slot = self.engine.detect_host_mixed([imgs[i] for i in part], wait=False, index=5 + (n & 1))
                if pending is not None:
                    consume(pending)
                pending = (slot, part)
            if pending is not None:
                consume(pending)
        return out

    def detect_stream(self, batches):
        """Steady-state serving: an iterable of host uint8 batches [B,H,W,3] -> a generator of the graph outputs per
        batch, in order, with the host-to-device and device-to-host copies of neighbouring batches hidden under the
        compute of the current one (Engine.detect_stream).  Mode f32 (mode f16x3 needs the per-batch status check of
        detect_batch and is served by that method)."""
        if self.engine.precision != "f32":
            for images in batches:
                yield self.detect_batch(images)
            return
        yield from self.engine.detect_stream(batches)

    def __call__(self, image, score_threshold=0.1):
        """
        Arguments:
            image: a numpy uint8 array with shape [height, width, 3] (RGB), or the path of an image file.
            score_threshold: a float number.
        Returns:
            boxes: a float numpy array of shape [N, 4] (ymin, xmin, ymax, xmax!).
            labels: an int numpy array of shape [N].
            scores: a float numpy array of shape [N].
        """
        if isinstance(image, (str, os.PathLike)):
            # BASELINE.json's north star words the API as Detector(image_path): a path is read the way the reference's
            # notebooks read it (inference/just_try_detector.ipynb: PIL, RGB) and then takes the ndarray route
            from PIL import Image
            with Image.open(image) as im:
                image = np.asarray(im.convert("RGB"), dtype=np.uint8)
        image = np.asarray(image)
        if image.dtype != np.uint8 or image.ndim != 3 or image.shape[2] != 3:
            raise ValueError("image must be a uint8 array of shape [height, width, 3]")
        if self.engine.one_call_detect:
            # one library call: upload, forward, wait and the filter below in C (ssd_detect_host).  The engine's lock (an RLock)
            # is held across the call AND the read-and-clear of the status word: with two threads on one Detector in mode
            # f16x3, thread A's status() must not consume the overflow bit that thread B's forward raised.
            with self.engine.lock:
                out = self.engine.detect_one(image, float(score_threshold))
                if self.engine.precision == "f16x3" and self.engine.status() & 1:
                    # an activation exceeded +-65504 and was clamped (ssd_hip.h ssd_status): these results are not
                    # trustworthy -- this detector continues in the exact mode
                    import warnings
                    warnings.warn("single-shot-detector_amd: activation outside the fp16 range in precision mode "
                                  "f16x3; switching this Detector to f32")
                    self.engine.set_precision("f32")
                    out = self.engine.detect_one(image, float(score_threshold))
            return out
        with self.engine.lock:      # the views below live in the engine's pinned result block until the next call
            boxes, labels, scores, n = self._detect_views(image[None])
            n = n[0]  # inference/detector.py:54-58
            to_keep = scores[0][:n] > score_threshold
            boxes = boxes[0][:n][to_keep]         # (boolean indexing copies)
            labels = labels[0][:n][to_keep]
            scores = scores[0][:n][to_keep]
        return boxes, labels, scores
