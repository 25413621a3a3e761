"""ctypes wrapper of oracle/libvbt_oracle.so (ORACLE - test infrastructure only; see detector.c)."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        # VBT_ORACLE_LIB: the sanitizer build of the same source (make -C oracle asan; tests/test_oracle_asan.py)
        L = ctypes.CDLL(os.environ.get("VBT_ORACLE_LIB") or os.path.join(_HERE, "libvbt_oracle.so"))
        L.vbto_load.restype = ctypes.c_void_p
        L.vbto_load.argtypes = [ctypes.c_char_p]
        L.vbto_free.argtypes = [ctypes.c_void_p]
        for f in ("vbto_num_tensors", "vbto_num_ops", "vbto_image_size"):
            getattr(L, f).argtypes = [ctypes.c_void_p]
        L.vbto_tensor_shape.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
        L.vbto_tensor_data.restype = ctypes.c_void_p
        L.vbto_tensor_data.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.vbto_run.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 5
        L.vbto_run_batch.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 4
        _LIB = L
    return _LIB


class OracleDetector:
    MAXDET = 25

    def __init__(self, path):
        self.path = path
        self.h = lib().vbto_load(path.encode())
        if not self.h:
            raise RuntimeError(f"oracle: cannot load {path}")
        self.size = lib().vbto_image_size(self.h)
        self.num_tensors = lib().vbto_num_tensors(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().vbto_free(self.h)
            self.h = None

    def run(self, frame):
        """frame uint8 [S,S,3] -> (boxes[25,4] ymin,xmin,ymax,xmax, scores[25], classes[25], count)."""
        frame = np.ascontiguousarray(frame, dtype=np.uint8)
        assert frame.shape == (self.size, self.size, 3)
        boxes = np.zeros((self.MAXDET, 4), np.float32)
        scores = np.zeros(self.MAXDET, np.float32)
        classes = np.zeros(self.MAXDET, np.float32)
        count = np.zeros(1, np.int32)
        rc = lib().vbto_run(self.h, frame.ctypes.data, boxes.ctypes.data, scores.ctypes.data,
                            classes.ctypes.data, count.ctypes.data)
        if rc:
            raise RuntimeError("oracle run failed")
        return boxes, scores, classes, int(count[0])

    def tensor(self, tid):
        shp = (ctypes.c_int * 3)()
        lib().vbto_tensor_shape(self.h, tid, shp)
        n = shp[0] * shp[1] * shp[2]
        p = lib().vbto_tensor_data(self.h, tid)
        return np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_int8)), shape=(n,)).reshape(shp[0], shp[1], shp[2]).copy()


def run_batch(path, frames, threads=1):
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    B = frames.shape[0]
    boxes = np.zeros((B, 25, 4), np.float32)
    scores = np.zeros((B, 25), np.float32)
    classes = np.zeros((B, 25), np.float32)
    counts = np.zeros(B, np.int32)
    rc = lib().vbto_run_batch(path.encode(), frames.ctypes.data, B, threads, boxes.ctypes.data,
                              scores.ctypes.data, classes.ctypes.data, counts.ctypes.data)
    if rc:
        raise RuntimeError("oracle batch run failed")
    return boxes, scores, classes, counts
