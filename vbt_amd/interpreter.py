"""`Interpreter`: the detector object of the reference hot loop, backed by libvbt_hip.so.

Mirrors the slice of tflite_runtime.interpreter.Interpreter the reference uses:
  Interpreter(model_path=..., num_threads=...)  reference track.py:93, eval.py:167
  .allocate_tensors()                           reference track.py:94
  .get_input_details()[0]['shape']              reference odt.py:86-87
  .get_signature_runner()(images=uint8[1,H,W,3]) -> {'output_0'..'output_3'}   reference odt.py:58-66
"""
import ctypes
import os

import numpy as np

from . import _lib

MAXDET = 25


class Interpreter:
    def __init__(self, model_path, num_threads=4, device=0, max_batch=1, fuse=True, flags=None):
        # num_threads is accepted for signature compatibility (reference track.py:72); the GPU path ignores it.
        self.model_path = str(model_path)
        self.num_threads = num_threads
        self.device = device
        self.max_batch = int(max_batch)
        self._h = ctypes.c_void_p()
        if flags is None:
            flags = int(os.environ.get('VBT_FUSION_FLAGS', '0')) if fuse else 1
        # --model may name a TFLite flatbuffer like the reference's (track.py:67): it is converted to the container
        # format on the fly (vbt_amd/tflite_import.py); the library itself only parses containers.
        from .tflite_import import as_container_path
        path, temporary = as_container_path(self.model_path)
        try:
            _lib.check(_lib.lib().vbt_model_create_ex(path.encode(), device, self.max_batch, flags, ctypes.byref(self._h)))
        finally:
            if temporary:
                os.unlink(path)
        shp = (ctypes.c_int * 4)()
        _lib.check(_lib.lib().vbt_model_input_shape(self._h, shp))
        self._shape = np.array([1, shp[1], shp[2], shp[3]], dtype=np.int32)

    @classmethod
    def _borrowed(cls, handle, model_path, device, max_batch, owner):
        """A view of a detector instance owned by a vbt_pipeline (never destroyed from here; `owner` is kept alive)."""
        self = cls.__new__(cls)
        self.model_path, self.num_threads, self.device, self.max_batch = str(model_path), 4, device, int(max_batch)
        self._h, self._owner = ctypes.c_void_p(handle), owner
        shp = (ctypes.c_int * 4)()
        _lib.check(_lib.lib().vbt_model_input_shape(self._h, shp))
        self._shape = np.array([1, shp[1], shp[2], shp[3]], dtype=np.int32)
        return self

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and getattr(self, "_owner", None) is None and _lib is not None and _lib._lib is not None:
            _lib._lib.vbt_model_destroy(h)
            self._h = None

    @property
    def handle(self):
        return self._h

    def allocate_tensors(self):
        return None     # buffers are allocated at construction; kept for call-shape parity

    def get_input_details(self):
        return [{"name": "images", "index": 0, "shape": self._shape.copy(), "dtype": np.uint8}]

    def get_signature_runner(self, signature_key=None):
        def run(images):
            images = np.asarray(images)
            if images.dtype != np.uint8 or images.ndim != 4 or tuple(images.shape[1:]) != tuple(self._shape[1:]):
                raise ValueError(f"images must be uint8 [B,{self._shape[1]},{self._shape[2]},3], got {images.dtype} {images.shape}")
            boxes, scores, classes, counts = self.detect(images)
            if images.shape[0] == 1:
                return {"output_0": counts.astype(np.float32), "output_1": scores, "output_2": classes, "output_3": boxes}
            return {"output_0": counts.astype(np.float32), "output_1": scores, "output_2": classes, "output_3": boxes}
        return run

    def detect(self, frames):
        """frames uint8 [B,H,W,3] (host) -> boxes [B,25,4], scores [B,25], classes [B,25], counts [B]."""
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        B = frames.shape[0]
        boxes = np.empty((B, MAXDET, 4), np.float32)
        scores = np.empty((B, MAXDET), np.float32)
        classes = np.empty((B, MAXDET), np.float32)
        counts = np.empty((B,), np.int32)
        _lib.check(_lib.lib().vbt_detect(self._h, frames.ctypes.data, B, 0, None, boxes.ctypes.data, scores.ctypes.data,
                                         classes.ctypes.data, counts.ctypes.data, 0))
        return boxes, scores, classes, counts

    # --- parity/debug helpers -------------------------------------------------
    def num_tensors(self):
        return _lib.lib().vbt_model_num_tensors(self._h)

    def materialized(self, tid):
        return bool(_lib.lib().vbt_model_tensor_materialized(self._h, tid))

    def num_launches(self):
        return _lib.lib().vbt_model_num_launches(self._h)

    def read_tensor(self, tid, B):
        shp = (ctypes.c_int * 3)()
        _lib.check(_lib.lib().vbt_model_tensor_shape(self._h, tid, shp))
        out = np.empty((B, shp[0], shp[1], shp[2]), np.int8)
        _lib.check(_lib.lib().vbt_model_read_tensor(self._h, tid, B, out.ctypes.data))
        return out
