"""Worker-process side of the input pipeline (tf_face_toolbox_amd/data.py): decode + resize + crop + flip / augmentation +
normalise of ONE image, written straight into a shared batch buffer.  Imports numpy and PIL only -- a worker process
(started as `python -m tf_face_toolbox_amd._decode_worker`) never loads torch or touches the GPU.  Mirrors data.py:206-223 of the reference (the tf.data map function)."""
import numpy as np

_RESIZE_TABLES = {}
_SHM = {}


def _resize_table(n_in, n_out):
    """Source rows / lerp weights of TF-1.x ResizeBilinear with align_corners=False (the default of
    tf.image.resize_images, data.py:213): in = out * (n_in / n_out) -- NO half-pixel centres in TF 1.x --
    low = floor(in), high = min(low + 1, n_in - 1), weight of `high` = in - low, all in float32 like the TF kernel."""
    key = (n_in, n_out)
    t = _RESIZE_TABLES.get(key)
    if t is None:
        pos = np.arange(n_out, dtype=np.float32) * (np.float32(n_in) / np.float32(n_out))
        lo = np.floor(pos).astype(np.int64)
        hi = np.minimum(lo + 1, n_in - 1)
        t = _RESIZE_TABLES[key] = (lo, hi, (pos - lo).astype(np.float32))
    return t


def resize_bilinear_tf1(image, height, width):
    """tf.image.resize_images(image, [height, width]) of TF 1.x on a float32 HWC image: bilinear, align_corners=False,
    no antialiasing (PIL's BILINEAR widens its kernel when shrinking and gives different pixels), float32 arithmetic.
    Same size -> returned unchanged (TF skips the op)."""
    h, w = image.shape[:2]
    if (h, w) == (height, width):
        return image
    ylo, yhi, yw = _resize_table(h, height)
    xlo, xhi, xw = _resize_table(w, width)
    top = image[ylo]
    rows = top + (image[yhi] - top) * yw[:, None, None]                 # [height, w, c]: blend of the two source rows
    left = rows[:, xlo]
    return left + (rows[:, xhi] - left) * xw[None, :, None]             # then of the two source columns


def decode(path, num_channels, height, width):
    """tf.read_file + decode_jpeg(channels) + convert_image_dtype(float32) + resize_images (data.py:208-213)."""
    from PIL import Image
    img = Image.open(path)
    img = img.convert('RGB' if num_channels == 3 else 'L')
    a = np.asarray(img, dtype=np.float32) * np.float32(1.0 / 255.0)   # convert_image_dtype(uint8 -> float32): x * (1/255)
    a = a.reshape(a.shape[0], a.shape[1], num_channels)
    return resize_bilinear_tf1(a, height, width)


def train_example(path, num_channels, input_height, input_width, crop_height, crop_width, augmentation, rng):
    image = decode(path, num_channels, input_height, input_width)
    if crop_height != -1 and crop_width != -1:                      # tf.random_crop
        y0 = rng.integers(0, input_height - crop_height + 1)
        x0 = rng.integers(0, input_width - crop_width + 1)
        image = image[y0:y0 + crop_height, x0:x0 + crop_width, :]
    if augmentation:
        from .preprocessing import data_augmentation
        image = data_augmentation(image, rng)
    elif rng.random() < 0.5:                                        # tf.image.random_flip_left_right
        image = image[:, ::-1, :]
    return (np.ascontiguousarray(image, dtype=np.float32) - 0.5) / 0.5


def fill_rows(task):
    """(buffer file, batch shape, [(row, path, seed), ...], num_channels, in_h, in_w, crop_h, crop_w, augmentation): decode the
    listed images into rows of the shared batch buffer (a float32 file under /dev/shm mapped by the parent and every worker).
    Returns the number of rows written (errors propagate)."""
    name, shape, rows, num_channels, in_h, in_w, crop_h, crop_w, augmentation = task
    batch = _SHM.get(name)
    if batch is None or batch.shape != tuple(shape):
        batch = _SHM[name] = np.memmap(name, dtype=np.float32, mode='r+', shape=tuple(shape))
    for row, path, seed in rows:
        batch[row] = train_example(path, num_channels, in_h, in_w, crop_h, crop_w, augmentation, np.random.default_rng(seed))
    return len(rows)


def main():
    """Worker loop: length-prefixed pickled tasks on stdin, one ('ok', rows) / ('err', text) reply each on stdout."""
    import pickle
    import struct
    import sys
    inp, out = sys.stdin.buffer, sys.stdout.buffer
    while True:
        hdr = inp.read(4)
        if len(hdr) < 4:
            return
        task = pickle.loads(inp.read(struct.unpack('<I', hdr)[0]))
        try:
            msg = ('ok', fill_rows(task))
        except BaseException as e:               # noqa: B902 -- reported to the parent, which raises it in the training thread
            msg = ('err', '%s: %s' % (type(e).__name__, e))
        b = pickle.dumps(msg)
        out.write(struct.pack('<I', len(b)) + b)
        out.flush()


if __name__ == '__main__':
    main()
