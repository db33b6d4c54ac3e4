"""CPU: host logic of train.py / data.py / saver.py (flags, LR schedules, list parsers, samplers,
preprocessing, checkpoint name map) -- the callers either side of the hot path."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import train as train_cli                     # noqa: E402
from oracle import ops                        # noqa: E402
from tf_face_toolbox_amd import data, saver, net_select, preprocessing     # noqa: E402


def test_flag_surface_matches_the_reference():
    F = train_cli.build_parser().parse_args(['--net_name', 'SphereNet', '--model_name', 'm', '--batch_size', '512', '--max_epoches', '3'])
    want = dict(train_dir='train', model_dir='models', pretrained_path='', data_format='NCHW', input_height=384, input_width=128,
                crop_height=-1, crop_width=-1, is_color=1, augmentation=0, num_classes=-1, num_per_class=-1,
                optimizer='Momentum', init_lr=0.1, lr_decay_method='step', lr_decay_rate=0.1, lr_decay_epoch='',
                weight_decay=5e-4, num_gpus=4, display_interval=10, save_interval=1000)
    for k, v in want.items():
        assert getattr(F, k) == v, k
    train_cli.FLAGS_assertion(F)
    F.batch_size = 510
    with pytest.raises(AssertionError):
        train_cli.FLAGS_assertion(F)              # 510 % 4 != 0  (train.py:98)
    F.batch_size, F.optimizer = 512, 'SGD'
    with pytest.raises(AssertionError, match='Unsupported optimizer.'):
        train_cli.FLAGS_assertion(F)


def test_lr_config_equals_oracle_tables():
    F = train_cli.build_parser().parse_args(['--max_epoches', '12', '--lr_decay_epoch', '3,5,9'])
    lr = train_cli.lr_config(F, 'step', 100)
    for s in range(0, 1300, 37):
        assert abs(lr(s) - ops.lr_step(s, 0.1, 0.1, ['3', '5', '9'], 100)) < 1e-15
    F.lr_decay_epoch = '2'
    lr = train_cli.lr_config(F, 'exp', 100)
    for s in range(0, 1300, 37):
        assert abs(lr(s) - ops.lr_exp(s, 0.1, 2, 12, 100)) < 1e-15
    lr = train_cli.lr_config(F, 'cosine', 100)
    for s in range(0, 1300, 37):
        assert abs(lr(s) - ops.lr_cosine(s, 0.1, 12, 100)) < 1e-15
    F.lr_decay_epoch = ''
    with pytest.raises(ValueError, match='Empty learning rate decay epoch boundaries.'):
        train_cli.lr_config(F, 'step', 100)
    with pytest.raises(ValueError, match='Unsupported learning rate decaying method.'):
        train_cli.lr_config(F, 'poly', 100)


def test_log_line_format():
    s = train_cli.format_str(['cross_entropy', 'reg_loss'])
    assert s.count('%s') == 4 and 'Loss #1: reg_loss = %.6f' in s and s.endswith('throughput = %.1fimages/s')


def _write_images(tmp_path, n_per_class=(5, 3, 6)):
    from PIL import Image
    lines = []
    rng = np.random.default_rng(0)
    for c, k in enumerate(n_per_class):
        for i in range(k):
            path = str(tmp_path / ('c%d_%d.png' % (c, i)))
            Image.fromarray(rng.integers(0, 255, (20, 24, 3), dtype=np.uint8)).save(path)
            lines.append('%s %d' % (path, c))
    lst = tmp_path / 'train.txt'
    lst.write_text('\n'.join(lines) + '\n')
    return str(lst), lines


def test_list_parsers(tmp_path):
    lst, lines = _write_images(tmp_path)
    paths, n = data.get_image_paths(lst)
    assert n == 14 and paths[0].endswith('c0_0.png')
    paths, labels, n, ncls = data.get_image_paths_and_labels(lst)
    assert (n, ncls) == (14, 3) and labels[:6] == [0, 0, 0, 0, 0, 1]
    d, n, ncls = data.get_image_paths_and_labels_dict(lst, num_per_class=4)
    assert ncls == 2 and n == 11 and [len(v) for v in d] == [5, 6]     # class 1 (3 images) dropped, classes renumbered


def test_train_inputs_random_and_balanced(tmp_path):
    lst, _ = _write_images(tmp_path)
    inp = data.train_inputs(lst, 16, 16, crop_height=12, crop_width=12, is_color=1, batch_size=4, device='cpu', seed=3)
    x = inp['images'](); y = inp['labels']()
    assert x.shape == (4, 12, 12, 3) and x.dtype == torch.float32 and y.dtype == torch.int32
    assert float(x.min()) >= -1 and float(x.max()) <= 1 and inp['num_classes'] == 3 and inp['num_examples'] == 14
    # PxK: 2 classes x 3 images, images of one identity contiguous (data.py:230-242)
    inp = data.train_inputs(lst, 16, 16, is_color=0, batch_size=-1, num_classes=2, num_per_class=3, device='cpu', seed=4)
    x = inp['images'](); y = inp['labels']().tolist()
    assert x.shape == (6, 16, 16, 1) and y[0] == y[1] == y[2] and y[3] == y[4] == y[5] and inp['batch_size'] == 6
    # two ranks with the same seed see complementary halves of the same global batch
    a = data.train_inputs(lst, 16, 16, batch_size=4, device='cpu', seed=9, rank=0, world_size=2)
    b = data.train_inputs(lst, 16, 16, batch_size=4, device='cpu', seed=9, rank=1, world_size=2)
    g = data.train_inputs(lst, 16, 16, batch_size=4, device='cpu', seed=9)
    xa, xb, xg = a['images'](), b['images'](), g['images']()
    assert torch.equal(torch.cat([xa, xb]), xg) and torch.equal(torch.cat([a['labels'](), b['labels']()]), g['labels']())


def test_eval_inputs_repeat_in_order(tmp_path):
    lst, _ = _write_images(tmp_path)
    nxt, n = data.eval_inputs(lst, 8, 1, 16, 16, device='cpu')
    b1, b2 = nxt(), nxt()
    assert n == 14 and b1.shape == (8, 16, 16, 3)
    nxt2, _ = data.eval_inputs(lst, 14, 1, 16, 16, device='cpu')
    full = nxt2()
    assert torch.equal(torch.cat([b1, b2])[:14], full)              # wraps around after 14 like dataset.repeat()


def test_augmentation_keeps_shape_and_range():
    rng = np.random.default_rng(0)
    img = rng.uniform(0, 1, (10, 12, 3)).astype(np.float32)
    for s in range(20):
        out = preprocessing.data_augmentation(img.copy(), np.random.default_rng(s))
        assert out.shape == img.shape and out.dtype == np.float32 and out.min() >= -0.2 - 1e-6 and out.max() <= 1 + 1e-6
    hsv = preprocessing._rgb_to_hsv(img)
    np.testing.assert_allclose(preprocessing._hsv_to_rgb(hsv), img, atol=1e-6)


def test_checkpoint_round_trip_and_finetune(tmp_path):
    net = net_select('SphereNet', 'NCHW'); net.seed = 5
    net.build(16, 16, 3, 7, 'cpu')
    slots = [torch.randn(net.arena_size)]
    prefix = str(tmp_path / 'SphereNet_m' / 'SphereNet_m.ckpt')
    p1 = saver.save(net, slots, 40, prefix)
    assert p1.endswith('.ckpt-40') and saver.latest_checkpoint(os.path.dirname(prefix)) == p1 and saver.step_of(p1) == 40
    # a checkpoint is layout-independent: restore into an NHWC net
    other = net_select('SphereNet', 'NHWC'); other.seed = 6
    other.build(16, 16, 3, 7, 'cpu')

    class Opt(object):
        slots = None

        def _ensure(self):
            self.slots = [torch.zeros(other.arena_size)]
    opt = Opt()
    assert saver.restore(other, p1, optimizer=opt) == 40
    for k in net.variables:
        assert torch.equal(other.get_variable(k), net.get_variable(k))
        assert torch.equal(other.get_variable(k, opt.slots[0]), net.get_variable(k, slots[0]))
    # finetune: only the backbone (pretrained_param) is restored (train.py:191-193)
    ft = net_select('SphereNet', 'NCHW'); ft.seed = 7
    ft.build(16, 16, 3, 7, 'cpu')
    before = ft.get_variable('classifier/fc_classifier/weights').clone()
    saver.restore(ft, p1, only=ft.pretrained_param())
    assert torch.equal(ft.get_variable('classifier/fc_classifier/weights'), before)
    assert torch.equal(ft.get_variable('SphereNet/conv4/Conv/weights'), net.get_variable('SphereNet/conv4/Conv/weights'))
    for s in range(41, 41 + 25):
        saver.save(net, None, s, prefix)
    kept = [l for l in open(os.path.join(os.path.dirname(prefix), 'checkpoint')).read().split() if l]
    assert len(kept) == saver.MAX_TO_KEEP and not os.path.exists(p1)          # max_to_keep=20 (train.py:188)


def test_checkpoint_keeps_bn_moving_statistics_and_centers(tmp_path):
    """tf.global_variables() is what train.py:188 saves: BatchNorm moving statistics and the center loss's `centers`
    travel with the weights (reference names), are restored on resume, and the moving statistics are part of the
    fine-tune set (param_list(trainable=False), nets/resnet.py:178-193); `centers` is not (it lives outside the scope)."""
    net = net_select('ResNeXt-50-center', 'NCHW'); net.seed = 3
    net.build(32, 32, 3, 11, 'cpu')
    assert len(net.state) == 106 and all(k.endswith(('/moving_mean', '/moving_variance')) for k in net.state)
    g = torch.Generator().manual_seed(1)
    for t in net.state.values():
        t.copy_(torch.rand(t.shape, generator=g) + 0.5)
    net._centers().copy_(torch.randn(net._centers().shape, generator=g))
    prefix = str(tmp_path / 'r' / 'r.ckpt')
    path = saver.save(net, None, 7, prefix)
    saved = torch.load(path, map_location='cpu')['variables']
    assert 'centers' in saved and 'ResNeXt-50/conv1/conv_7x7/BatchNorm/moving_mean' in saved
    other = net_select('ResNeXt-50-center', 'NHWC'); other.seed = 4
    other.build(32, 32, 3, 11, 'cpu')
    assert saver.restore(other, path) == 7
    for k in list(net.variables) + list(net.state):
        assert torch.equal(other.get_variable(k), net.get_variable(k)), k
    assert torch.equal(other._centers(), net._centers())
    # fine-tune: backbone weights AND its moving statistics, neither the classifier nor the centers
    ft = net_select('ResNeXt-50-center', 'NCHW'); ft.seed = 5
    ft.build(32, 32, 3, 11, 'cpu')
    names = [v.name for v in ft.pretrained_param()]
    assert sum('moving_' in n for n in names) == 106 and 'centers' not in names and not any(n.startswith('classifier/') for n in names)
    cls_before = ft.get_variable('classifier/fc_classifier/weights').clone()
    saver.restore(ft, path, only=ft.pretrained_param())
    assert torch.equal(ft.get_variable('classifier/fc_classifier/weights'), cls_before)
    k = 'ResNeXt-50/conv1/conv_7x7/BatchNorm/moving_variance'
    assert torch.equal(ft.get_variable(k), net.get_variable(k))
    assert float(ft._centers().abs().max()) == 0.0
    # the trainable groups are unchanged (data_parallel.py:232 iterates param_list(trainable=True))
    assert [len(g_) for g_ in ft.param_list(True, True)] == [159, 1]


def test_loader_errors_reach_the_training_thread(tmp_path):
    """a corrupt image / bad label must raise from the step (tf.data surfaces it to sess.run), not kill the
    producer thread silently and leave advance() blocked forever (and the other ranks hanging in the all-reduce)."""
    from PIL import Image
    good = str(tmp_path / 'a.png')
    Image.fromarray(np.zeros((8, 8, 3), np.uint8)).save(good)
    bad = str(tmp_path / 'broken.png')
    open(bad, 'wb').write(b'not a png')
    (tmp_path / 'l.txt').write_text('%s 0\n%s 1\n' % (good, bad))
    inp = data.train_inputs(str(tmp_path / 'l.txt'), 8, 8, batch_size=2, device='cpu', seed=0)
    with pytest.raises(RuntimeError, match='input pipeline failed'):
        inp['images']()
    with pytest.raises(RuntimeError, match='input pipeline failed'):       # and it stays failed: no hang on the next call
        inp['images']()
    # out-of-range labels are caught where the batch is built
    pf = data._Prefetcher(lambda: (np.zeros((2, 4, 4, 1), np.float32), np.array([0, 7], np.int32)), torch.device('cpu'), num_classes=7)
    with pytest.raises(RuntimeError, match='label out of range'):
        pf.advance()
    pf = data._Prefetcher(lambda: (np.zeros((2, 4, 4, 1), np.float32), np.array([0, 6], np.int32)), torch.device('cpu'), num_classes=7)
    assert pf.advance()[1].tolist() == [0, 6]


def test_num_gpus_n_launches_its_own_ranks(tmp_path):
    """`python train.py --num_gpus 2 ...` (the reference's invocation, one process driving N towers) starts two ranks through
    torch.distributed.run by itself and exits with their status.  Without a GPU the ranks fail -- loudly, non-zero -- which is
    what this CPU test can observe: the launcher ran, both ranks were started, nothing fell back to a CPU path."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(root, 'train.py'), '--net_name', 'SphereNet', '--model_name', 't', '--synthetic', '1',
                        '--batch_size', '8', '--num_gpus', '2', '--max_epoches', '1', '--lr_decay_epoch', '1', '--max_steps', '1',
                        '--input_height', '112', '--input_width', '112', '--train_dir', str(tmp_path / 'tr'), '--model_dir', str(tmp_path / 'm')],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=300)
    import torch
    if torch.cuda.is_available():
        pytest.skip('CPU-only expectation (on a GPU box two ranks share one device and RCCL refuses)')
    assert r.returncode != 0
    assert 'local_rank: 0' in r.stdout or 'local_rank: 1' in r.stdout, r.stdout[-2000:]
