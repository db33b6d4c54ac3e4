"""The loader's GPU transform (fte_preprocess_u8, include/fte.h): decoded uint8 images + the workers' seeded draws in, the
float32 NHWC batch of data.py:206-223 out -- BIT-EQUAL to the host transform (tf_face_toolbox_amd/_decode_worker.py, itself held to
the plain-loop restatement oracle/image_ops.py by tests/test_loader_values.py)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

IMG = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'images')
NAMES = ['a.png', 'b.png', 'c.png', 'd.png', 'e.jpg', 'f.jpg', 'g.jpg', 'h.jpg']


def _slots(ch, in_h, in_w, crop_h, crop_w, seeds, nbytes):
    from tf_face_toolbox_amd import _decode_worker as dw
    buf = np.zeros((len(seeds), nbytes), dtype=np.uint8)
    for i, seed in enumerate(seeds):
        dw.raw_example(buf[i], os.path.join(IMG, NAMES[i % len(NAMES)]), ch, in_h, in_w, crop_h, crop_w,
                       None if seed is None else np.random.default_rng(seed))
    return buf


def _gpu(buf, ch, in_h, in_w, out_h, out_w):
    import torch
    from tf_face_toolbox_amd._lib import call
    raw = torch.from_numpy(buf).cuda()
    out = torch.empty((buf.shape[0], out_h, out_w, ch), dtype=torch.float32, device='cuda')
    call('fte_preprocess_u8', raw.data_ptr(), out.data_ptr(), buf.shape[0], buf.shape[1], ch, in_h, in_w, out_h, out_w,
         torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize('ch', [3, 1])
@pytest.mark.parametrize('geom', [(120, 116, 112, 112), (128, 128, 112, 112), (112, 96, -1, -1), (37, 29, 32, 24), (300, 280, 224, 224)])
def test_train_transform_is_bit_equal_to_the_host(ch, geom):
    """resize (up, down, identity: a.png is 29 x 37) + crop + flip + normalise for 24 seeded examples per geometry; slots large
    enough for every image, and slots that force the larger ones through the finished-crop path."""
    from tf_face_toolbox_amd import _decode_worker as dw
    in_h, in_w, crop_h, crop_w = geom
    out_h, out_w = (crop_h, crop_w) if crop_h != -1 else (in_h, in_w)
    seeds = list(range(7, 31))
    want = np.stack([dw.train_example(os.path.join(IMG, NAMES[i % len(NAMES)]), ch, in_h, in_w, crop_h, crop_w, 0, np.random.default_rng(s))
                     for i, s in enumerate(seeds)])
    flips = 0
    for side in (256, 64):
        nbytes = (dw.HEADER_BYTES + max(side * side * ch, out_h * out_w * ch * 4) + 63) // 64 * 64
        buf = _slots(ch, in_h, in_w, crop_h, crop_w, seeds, nbytes)
        flips += int(buf[:, :24].view(np.int32)[:, 5].sum())
        got = _gpu(buf, ch, in_h, in_w, out_h, out_w)
        assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert flips > 0


def test_eval_transform_is_bit_equal_to_the_host():
    from tf_face_toolbox_amd import _decode_worker as dw
    buf = _slots(3, 112, 96, -1, -1, [None] * 8, dw.HEADER_BYTES + 256 * 256 * 3)
    got = _gpu(buf, 3, 112, 96, 112, 96)
    for i, n in enumerate(NAMES):
        want = (dw.decode(os.path.join(IMG, n), 3, 112, 96) - np.float32(0.5)) / np.float32(0.5)
        assert np.array_equal(got[i].view(np.uint32), want.view(np.uint32))


def test_bad_arguments_are_refused():
    import torch
    from tf_face_toolbox_amd._lib import query
    x = torch.zeros(4096, dtype=torch.uint8, device='cuda')
    o = torch.zeros(4096, dtype=torch.float32, device='cuda')
    for args in [(1, 100, 3, 8, 8, 8, 8), (1, 4096, 2, 8, 8, 8, 8), (1, 4096, 3, 8, 8, 9, 8), (0, 4096, 3, 8, 8, 8, 8)]:
        assert query('fte_preprocess_u8', x.data_ptr(), o.data_ptr(), *args, 0) != 0


def test_train_inputs_with_the_gpu_transform_equal_the_host_pipeline(tmp_path, monkeypatch):
    """train_inputs / eval_inputs end to end on the GPU box: worker processes + raw slots + fte_preprocess_u8 deliver the batches
    the all-host pipeline delivers for the same seed, labels included."""
    from tf_face_toolbox_amd import data
    lst = tmp_path / 'list.txt'
    lst.write_text(''.join('%s %d\n' % (os.path.join(IMG, n), i % 4) for i, n in enumerate(NAMES * 4)))
    monkeypatch.setenv('FTE_LOADER_GPU', '0')
    a = data.train_inputs(str(lst), 120, 116, 112, 112, is_color=1, batch_size=16, device='cuda', seed=5, num_workers=3)
    monkeypatch.setenv('FTE_LOADER_GPU', '1')
    b = data.train_inputs(str(lst), 120, 116, 112, 112, is_color=1, batch_size=16, device='cuda', seed=5, num_workers=3)
    try:
        assert not a['gpu_transform'] and b['gpu_transform']
        for _ in range(6):
            xa, xb = a['images'](), b['images']()
            assert xb.shape == (16, 112, 112, 3) and xb.dtype == xa.dtype
            assert np.array_equal(xa.cpu().numpy().view(np.uint32), xb.cpu().numpy().view(np.uint32))
            assert np.array_equal(a['labels']().cpu().numpy(), b['labels']().cpu().numpy())
    finally:
        a['close'](); b['close']()
    nh, _ = data.eval_inputs(str(lst), 64, True, 48, 40, device='cuda', num_workers=0)
    ng, _ = data.eval_inputs(str(lst), 64, True, 48, 40, device='cuda', num_workers=3)
    try:
        for _ in range(3):
            assert np.array_equal(nh().cpu().numpy().view(np.uint32), ng().cpu().numpy().view(np.uint32))
    finally:
        nh.close(); ng.close()
