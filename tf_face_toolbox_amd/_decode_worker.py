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


_S255 = np.float32(1.0 / 255.0)


def resize_window(image, height, width, y0=0, win_h=None, x0=0, win_w=None):
    """Rows [y0, y0 + win_h) x columns [x0, x0 + win_w) of tf.image.resize_images(image, [height, width]) as TF 1.x computes it
    on a float32 HWC image: bilinear, align_corners=False, no antialiasing (PIL's BILINEAR widens its kernel when shrinking and
    gives different pixels), float32 arithmetic in the kernel's own order -- the two neighbours of a row are blended along x
    first, then the two rows along y.  `image` may be uint8: it is scaled by 1/255 (convert_image_dtype) on the way, element
    by element, i.e. to exactly the values a conversion of the whole image would give.  Only the window is computed: a
    random crop of the resized image costs the crop's area, not the image's (the pipeline's second largest cost after the
    JPEG decode).  Same size and a full window -> the (converted) image itself (TF skips the op)."""
    h, w = image.shape[:2]
    win_h = height if win_h is None else win_h
    win_w = width if win_w is None else win_w
    u8 = image.dtype == np.uint8
    if (h, w) == (height, width):
        out = image[y0:y0 + win_h, x0:x0 + win_w]
        return out.astype(np.float32) * _S255 if u8 else out
    ylo, yhi, yw = (t[y0:y0 + win_h] for t in _resize_table(h, height))
    xlo, xhi, xw = (t[x0:x0 + win_w] for t in _resize_table(w, width))

    s = _S255 if u8 else None

    def blend_x(rows):                                            # rows: the gathered source rows, [win_h, w, c]
        left, right = np.take(rows, xlo, axis=1), np.take(rows, xhi, axis=1)
        if u8:
            left = left.astype(np.float32); left *= s
            right = right.astype(np.float32); right *= s
        right -= left
        right *= xw3
        right += left
        return right
    xw3 = xw[None, :, None]
    top = blend_x(image[ylo])
    bot = blend_x(image[yhi])
    bot -= top
    bot *= yw[:, None, None]
    bot += top
    return bot


def resize_bilinear_tf1(image, height, width):
    """tf.image.resize_images(image, [height, width]) of TF 1.x (see resize_window)."""
    return resize_window(image, height, width)


def _load(path, num_channels):
    """tf.read_file + decode_jpeg(channels): uint8 HWC"""
    from PIL import Image
    img = Image.open(path)
    img = img.convert('RGB' if num_channels == 3 else 'L')
    a = np.asarray(img, dtype=np.uint8)
    return a.reshape(a.shape[0], a.shape[1], num_channels)


def decode(path, num_channels, height, width):
    """tf.read_file + decode_jpeg(channels) + convert_image_dtype(float32) + resize_images (data.py:208-213)."""
    return resize_window(_load(path, num_channels), height, width)


def _finish(raw, input_height, input_width, crop_height, crop_width, augmentation, rng):
    if crop_height != -1 and crop_width != -1:                      # tf.random_crop: only the cropped window is resized
        y0 = rng.integers(0, input_height - crop_height + 1)
        x0 = rng.integers(0, input_width - crop_width + 1)
        image = resize_window(raw, input_height, input_width, y0, crop_height, x0, crop_width)
    else:
        image = resize_window(raw, input_height, input_width)
    if augmentation:
        from .preprocessing import data_augmentation
        image = data_augmentation(image, rng)
    elif rng.random() < 0.5:                                        # tf.image.random_flip_left_right
        image = image[:, ::-1, :]
    return (np.ascontiguousarray(image, dtype=np.float32) - 0.5) / 0.5


def train_example(path, num_channels, input_height, input_width, crop_height, crop_width, augmentation, rng):
    return _finish(_load(path, num_channels), input_height, input_width, crop_height, crop_width, augmentation, rng)


HEADER_BYTES = 64            # per raw slot: int32 {mode, h0, w0, y0, x0, flip} + padding (include/fte.h: fte_preprocess_u8)


def raw_example(slot, path, num_channels, input_height, input_width, crop_height, crop_width, rng):
    """The DECODED image and the draws of train_example() (seeded the same way, drawn in the same order) into one slot of a raw
    batch buffer -- resize / crop / flip / normalise then run on the GPU (fte_preprocess_u8) and give the bits train_example()
    gives.  rng None: the evaluation transform (full window, no flip).  An image too large for its slot is transformed here
    and stored finished (mode 1)."""
    raw = _load(path, num_channels)
    h0, w0 = raw.shape[:2]
    cropped = crop_height != -1 and crop_width != -1
    out_h, out_w = (crop_height, crop_width) if cropped else (input_height, input_width)
    hd = slot[:HEADER_BYTES].view(np.int32)
    if raw.size > slot.size - HEADER_BYTES:
        if rng is None:
            image = (resize_window(raw, input_height, input_width) - np.float32(0.5)) / np.float32(0.5)
        else:
            image = _finish(raw, input_height, input_width, crop_height, crop_width, 0, rng)
        slot[HEADER_BYTES:HEADER_BYTES + image.size * 4] = np.ascontiguousarray(image, dtype=np.float32).reshape(-1).view(np.uint8)
        hd[:6] = (1, out_h, out_w, 0, 0, 0)
        return
    y0 = x0 = flip = 0
    if rng is not None:
        if cropped:
            y0 = int(rng.integers(0, input_height - crop_height + 1))
            x0 = int(rng.integers(0, input_width - crop_width + 1))
        flip = int(rng.random() < 0.5)
    slot[HEADER_BYTES:HEADER_BYTES + raw.size] = raw.reshape(-1)
    hd[:6] = (0, h0, w0, y0, x0, flip)


def _buffer(name, shape, dtype):
    """The shared batch buffer a task names: anonymous shared memory behind the file DESCRIPTOR this process inherited from the
    parent (data._WorkerPool: memfd_create + pass_fds), or a file path."""
    batch = _SHM.get(name)
    if batch is None or batch.shape != tuple(shape) or batch.dtype != dtype:
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        if isinstance(name, int):
            import mmap
            mm = mmap.mmap(name, nbytes)
            batch = _SHM[name] = np.frombuffer(mm, dtype=dtype).reshape(tuple(shape))
        else:
            batch = _SHM[name] = np.memmap(name, dtype=dtype, mode='r+', shape=tuple(shape))
    return batch


def fill_rows(task):
    """(buffer, batch shape, [(row, path, seed), ...], num_channels, in_h, in_w, crop_h, crop_w, augmentation[, raw]): decode the
    listed images (seed None: the evaluation transform) into rows of the shared batch buffer -- a float32 array of finished
    examples, or with raw = 1 a uint8 array [rows, slot bytes] of decoded images + draws for the GPU transform.  Returns the
    number of rows written (errors propagate)."""
    name, shape, rows, num_channels, in_h, in_w, crop_h, crop_w, augmentation = task[:9]
    raw = len(task) > 9 and task[9]
    if raw:
        assert not augmentation, 'the colour augmentation runs on the host: no raw slots'
        batch = _buffer(name, shape, np.uint8)
        for row, path, seed in rows:
            raw_example(batch[row], path, num_channels, in_h, in_w, crop_h, crop_w, None if seed is None else np.random.default_rng(seed))
        return len(rows)
    batch = _buffer(name, shape, np.float32)
    for row, path, seed in rows:
        if seed is None:                          # evaluation: decode + resize + normalise, nothing random (data.py:153-191)
            batch[row] = (decode(path, num_channels, in_h, in_w) - np.float32(0.5)) / np.float32(0.5)
        else:
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
