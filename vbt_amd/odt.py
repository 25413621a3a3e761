"""Detection helpers with the names and semantics of reference odt.py, running on the GPU.

  preprocess_image   reference odt.py:10-19   (bilinear resize, half-pixel centres, truncating uint8 cast)
  calc_plate_width / calc_plate_height / calc_bounding_box_center    reference odt.py:22-50
  detect_objects     reference odt.py:53-77
  run_odt            reference odt.py:80-99
  results_to_sorttracker_inputs   reference odt.py:102-118
"""
import numpy as np

from . import _lib


def preprocess_image(frame, input_size, device=0):
    """uint8 [H,W,3] -> (uint8 [1,h,w,3], original) on the GPU (vbt_resize_frames): tf.image.resize's
    bilinear with half-pixel centres in float32, then the truncating tf.cast to uint8 (reference
    odt.py:15-18).  A frame already at the network size passes through (scale 1 is the identity)."""
    frame = np.ascontiguousarray(frame)
    if frame.dtype != np.uint8 or frame.ndim != 3 or frame.shape[2] != 3:
        raise ValueError(f"frame must be uint8 [H,W,3], got {frame.dtype} {frame.shape}")
    h, w = int(input_size[0]), int(input_size[1])
    H, W = frame.shape[:2]
    if (H, W) == (h, w):
        return frame[np.newaxis], frame
    out = np.empty((1, h, w, 3), np.uint8)
    _lib.check(_lib.lib().vbt_resize_frames(frame.ctypes.data, 1, H, W, 0, out.ctypes.data, h, w, 0, 0, device, None))
    return out, frame


def calc_plate_width(bounding_box):
    _, xmin, _, xmax = bounding_box
    return abs(xmax - xmin)


def calc_plate_height(bounding_box):
    ymin, _, ymax, _ = bounding_box
    return abs(ymax - ymin)


def calc_bounding_box_center(bounding_box):
    ymin, xmin, ymax, xmax = bounding_box
    return ((xmin + xmax) / 2, (ymin + ymax) / 2)


def detect_objects(interpreter, image, threshold):
    signature_fn = interpreter.get_signature_runner()
    output = signature_fn(images=image)
    count = int(np.squeeze(output["output_0"]))
    scores = np.squeeze(output["output_1"])
    boxes = np.squeeze(output["output_3"])
    results = []
    for i in range(count):
        if scores[i] >= threshold:
            results.append({"bounding_box": boxes[i], "score": scores[i]})
    return results


def run_odt(frame, interpreter, threshold=0.5):
    _, input_height, input_width, _ = interpreter.get_input_details()[0]["shape"]
    preprocessed_image, _ = preprocess_image(frame, (input_height, input_width))
    return detect_objects(interpreter, preprocessed_image, threshold=threshold)


def results_to_sorttracker_inputs(orig_results):
    results = []
    for res in orig_results:
        ymin, xmin, ymax, xmax = res["bounding_box"]
        results.append(np.array([xmin, ymin, xmax, ymax, res["score"], 0]))
    return np.empty((0, 6)) if len(results) == 0 else np.array(results)
