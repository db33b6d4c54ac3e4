"""Host-side image augmentation: mirror of the reference's preprocessing.py:22-38 (the only function
train_inputs calls).  CPU image ops, not part of the GPU step; numpy restatements of
tf.image.{random_flip_left_right, adjust_brightness, adjust_hue, adjust_saturation}."""
import colorsys  # noqa: F401  (documented reference for the HSV convention below)

import numpy as np


def _rgb_to_hsv(rgb):
    r, g, b = rgb[..., 0], rgb[..., 1], rgb[..., 2]
    mx, mn = rgb.max(-1), rgb.min(-1)
    d = mx - mn
    s = np.where(mx > 0, d / np.where(mx > 0, mx, 1), 0)
    dz = np.where(d > 0, d, 1)
    h = np.where(mx == r, (g - b) / dz % 6, np.where(mx == g, (b - r) / dz + 2, (r - g) / dz + 4)) / 6.0
    h = np.where(d > 0, h, 0)
    return np.stack([h, s, mx], -1)


def _hsv_to_rgb(hsv):
    h, s, v = hsv[..., 0] * 6.0, hsv[..., 1], hsv[..., 2]
    c = v * s
    x = c * (1 - np.abs(h % 2 - 1))
    z = np.zeros_like(c)
    i = np.floor(h).astype(int) % 6
    r = np.choose(i, [c, x, z, z, x, c])
    g = np.choose(i, [x, c, c, x, z, z])
    b = np.choose(i, [z, z, x, c, c, x])
    m = v - c
    return np.stack([r + m, g + m, b + m], -1)


def adjust_hue(image, delta):
    hsv = _rgb_to_hsv(image)
    hsv[..., 0] = (hsv[..., 0] + delta) % 1.0
    return _hsv_to_rgb(hsv)


def adjust_saturation(image, factor):
    hsv = _rgb_to_hsv(image)
    hsv[..., 1] = np.clip(hsv[..., 1] * factor, 0, 1)
    return _hsv_to_rgb(hsv)


def data_augmentation(image, rng):
    """preprocessing.py:22-38: flip; with prob 1/2 darken by delta in [0,0.1); RGB only: with prob 1/2
    hue shift by -delta (delta in [0,0.2)), with prob 1/2 desaturate by a factor in [0.6,1)."""
    if rng.random() < 0.5:
        image = image[:, ::-1, :]
    delta = rng.uniform(0, 0.2)
    if delta < 0.1:
        image = image - delta
    if image.shape[-1] == 3:
        delta = rng.uniform(0, 0.4)
        if delta < 0.2:
            image = adjust_hue(np.clip(image, 0, 1), -delta)
        delta = rng.uniform(0.6, 1.4)
        if delta < 1.0:
            image = adjust_saturation(np.clip(image, 0, 1), delta)
    return np.ascontiguousarray(image, dtype=np.float32)
