"""Oracle: the host-side image ops of the reference's input pipeline, restated with plain loops.

TEST INFRASTRUCTURE (see oracle/__init__.py) -- PARITY UNPINNED (TensorFlow is not installable here; the
semantics below are TF 1.x's documented kernels at the reference's call sites, data.py:206-223).
Small images only: these are Python loops on purpose (an independent, obviously-correct restatement)."""
import math

import numpy as np


def resize_bilinear_tf1(image, out_h, out_w):
    """tf.image.resize_images(image, [out_h, out_w]) as TF 1.x computes it (data.py:213): ResizeBilinear with
    align_corners=False -> scale = in / out, source coordinate = dst * scale (no half-pixel offset), the four
    neighbours (floor, min(floor + 1, in - 1)) blended with the fractional parts, in float arithmetic."""
    in_h, in_w, ch = image.shape
    if (in_h, in_w) == (out_h, out_w):
        return image.copy()
    out = np.zeros((out_h, out_w, ch), dtype=np.float64)
    # the TF kernel holds scale, source coordinate and lerp weight in float32 (`const float in_y = y * height_scale`):
    # the weights below are those float32 values; the blending itself is done in float64 here
    hs, ws = np.float32(in_h) / np.float32(out_h), np.float32(in_w) / np.float32(out_w)
    for y in range(out_h):
        fy = np.float32(y) * hs
        y0 = int(math.floor(fy))
        y1 = min(y0 + 1, in_h - 1)
        ly = float(np.float32(fy - np.float32(y0)))
        for x in range(out_w):
            fx = np.float32(x) * ws
            x0 = int(math.floor(fx))
            x1 = min(x0 + 1, in_w - 1)
            lx = float(np.float32(fx - np.float32(x0)))
            for c in range(ch):
                top = image[y0, x0, c] + (image[y0, x1, c] - image[y0, x0, c]) * lx
                bot = image[y1, x0, c] + (image[y1, x1, c] - image[y1, x0, c]) * lx
                out[y, x, c] = top + (bot - top) * ly
    return out


def train_example(image01, input_h, input_w, crop, y0x0, flip):
    """data.py:208-221 after decoding: resize -> crop at (y0, x0) -> optional left-right flip -> (x - 0.5) / 0.5."""
    a = resize_bilinear_tf1(np.asarray(image01, np.float64), input_h, input_w)
    if crop is not None:
        a = a[y0x0[0]:y0x0[0] + crop[0], y0x0[1]:y0x0[1] + crop[1], :]
    if flip:
        a = a[:, ::-1, :]
    return (a - 0.5) / 0.5
