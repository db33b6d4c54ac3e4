"""CPU: host-side mirror of the reference interface (no kernels run)."""
import numpy as np
import pytest
import torch

from oracle import spherenet as osn
from tf_face_toolbox_amd import net_select, Singular, DataParallel
from tf_face_toolbox_amd.nets.sphere import same_pads


def test_net_select_names_and_errors_follow_the_reference():
    assert net_select('SphereNet').name == 'SphereNet'                  # nets/net_base.py:23-26
    assert net_select('SphereNet-ASoftmax').needs_labels
    with pytest.raises(ValueError, match='Unsupport network architecture.'):   # nets/net_base.py:60-61
        net_select('LeNet')
    with pytest.raises(UnboundLocalError):                             # nets/net_base.py:52-59 `pass` branches
        net_select('MobileNet-v2')
    s = net_select('ShuffleNet-v2-small')                               # nets/net_base.py:37-42: alpha = 2.0
    assert s.name == 'ShuffleNet_v2_small_x2' and s.num_outputs == [122, 244, 488, 2048] and not s.se and not s.residual
    m, l = net_select('ShuffleNet-v2-middle'), net_select('ShuffleNet-v2-large')
    assert m.name == 'ShuffleNet_v2_middle' and m.num_outputs == [244, 488, 976, 1952, 2048]           # shufflenet_v2.py:233-234
    assert l.name == 'ShuffleNet_v2_large_se_res' and l.num_outputs == [340, 680, 1360, 2720, 2048]    # :305-306
    g, spec = s.build_graph(3, 10)
    shapes = dict((n_, s_) for n_, s_, _ in spec)
    assert shapes['ShuffleNet_v2_small_x2/conv3/resBlock_0/separable_conv2_3x3/depthwise_weights'] == (3, 3, 244, 1)
    assert sum(1 for op in g if op[0] in ('shufsplit', 'shufcat')) == 16 and g[-1][0] == 'fc'
    r = net_select('ResNet-50', 'NHWC', 1e-4)                          # nets/net_base.py:32-36
    assert r.name == 'ResNet-50' and r.num_block == [3, 4, 6, 3] and r.weight_decay == 1e-4
    with pytest.raises(AssertionError, match='Unknown data format.'):   # nets/net_base.py:72
        net_select('SphereNet', data_format='NCWH')
    n = net_select('SphereNet', 'NHWC', 1e-3)
    assert (n.data_format, n.weight_decay, n.channel_axis, n.spatial_axis) == ('NHWC', 1e-3, 3, [1, 2])


def test_same_pads_matches_oracle():
    for size in (7, 14, 28, 56, 112, 13, 9):
        for stride in (1, 2):
            assert same_pads(size, 3, stride) == osn.ops.same_pads(size, 3, stride)


@pytest.mark.parametrize('data_format', ['NCHW', 'NHWC'])
def test_arena_layout_variables_and_reference_layout_round_trip(data_format):
    net = net_select('SphereNet', data_format)
    net.build(32, 32, 3, 10, 'cpu')
    p = osn.perturb_params(osn.init_params(3, 3, 10, 32, 32), 4)
    assert sorted(net.variables) == sorted(p)                           # the 47 TF variable names
    for k, v in net.variables.items():
        assert v.ref_shape == p[k].shape
        assert v.offset % 4 == 0                                        # 16-byte aligned views
    # arena = [biases+alphas | conv W | FC W | classifier W (padded to 128 columns)]
    groups = net.arena_groups()
    assert groups[0][:3] == (0, net.small_end, False) and groups[-1][1] == net.arena_size
    assert net.cpad == 128 and net.view('classifier/fc_classifier/weights').numel() == 512 * 128
    # buckets in completion order: head (FC + classifier + the 4 loss slots), stages 4, 3, 2, then [biases/alphas + stage 1];
    # together they tile the arena exactly once and line up with the backward stages
    b = net.grad_buckets()
    assert b[0] == (net.fc_start, net.arena_size + 4) and len(b) == 5 == len(net.backward_stages())
    assert b[-1][0] == 0 and [x[1] for x in b[1:]] == [net.fc_start] + [x[0] for x in b[1:-1]]
    assert sum(e - a for a, e in b) == net.arena_size + 4
    w4 = net.variables['SphereNet/conv4/Conv/weights']
    assert b[1] == (w4.offset, net.fc_start)                              # stage 4 = conv4 + its residual block
    kinds = {v.kind for v in net.variables.values() if v.offset < net.small_end}
    assert kinds == {'bias', 'alpha', 'fc_b'}
    net.load_params(p)
    for k in p:
        np.testing.assert_array_equal(net.get_variable(k).numpy(), p[k].astype(np.float32))
    # the FC weight is stored in H,W,C row order; the reference's NCHW flatten order is C,H,W (nets/sphere.py:72)
    fcw = 'SphereNet/fully_connected/weights'
    internal = net.view(fcw).reshape(net.fin, 512).numpy()
    h, w, c = net.feat_hwc
    if data_format == 'NCHW':
        want = p[fcw].reshape(c, h, w, 512).transpose(1, 2, 0, 3).reshape(net.fin, 512)
    else:
        want = p[fcw]
    np.testing.assert_array_equal(internal, want.astype(np.float32))
    # padded classifier columns are zero and stay out of the export
    wc = net.view('classifier/fc_classifier/weights').reshape(512, net.cpad)
    assert float(wc[:, 10:].abs().max()) == 0.0


def test_initialisers_follow_the_reference():
    net = net_select('SphereNet')
    net.build(112, 112, 3, 10575, 'cpu')
    assert net.arena_size >= 29916352 and net.fin == 25088
    a = net.get_variable('SphereNet/conv2/Repeat/resBlock_1/Conv/alpha')
    assert float(a.min()) == 0.25 == float(a.max())                              # nets/sphere.py:34
    w = net.get_variable('SphereNet/conv3/Repeat/resBlock_2/Conv/weights')
    assert abs(float(w.std()) - 0.01) < 2e-4 and abs(float(w.mean())) < 1e-4     # N(0, 0.01), nets/sphere.py:41
    w = net.get_variable('SphereNet/conv2/Conv/weights')
    lim = (6.0 / (9 * 64 + 9 * 128)) ** 0.5                                       # Xavier-uniform
    assert float(w.abs().max()) <= lim and float(w.abs().max()) > 0.98 * lim
    assert float(net.get_variable('SphereNet/conv2/Conv/biases').abs().max()) == 0
    wc = net.get_variable('classifier/fc_classifier/weights')
    assert wc.shape == (512, 10575) and abs(float(wc.std()) - 1e-3) < 2e-5       # nets/sphere.py:87


def test_param_groups_mult_lr_and_pretrained():
    net = net_select('SphereNet')
    net.build(16, 16, 1, 5, 'cpu')
    groups = net.param_list(is_training=True, trainable=True)
    assert [len(g) for g in groups] == [46, 1]                                   # nets/sphere.py:120-126
    assert len(net.param_list(is_training=False, trainable=True)) == 1
    assert net.mult_lr_list() == [1.0, 1.0]                                      # nets/net_base.py:97-101
    assert all('SphereNet' in v.name for v in net.pretrained_param())            # nets/sphere.py:128-134


def test_product_path_refuses_cpu_tensors_and_bad_wrappers():
    net = net_select('SphereNet')
    net.build(16, 16, 1, 5, 'cpu')
    with pytest.raises(TypeError):                                               # no CPU fallback, ever
        net.forward(torch.zeros(2, 16, 16, 1), num_classes=5, is_training=True)
    with pytest.raises(AssertionError):                                          # data_parallel.py:83
        DataParallel(net, 0.1, 'Momentum', num_gpus=1)
    with pytest.raises(ValueError, match='Unsupported optimizer.'):              # train.py:99
        s = Singular(net, 0.1, 'SGD')
        s._setup({})
    with pytest.raises(AssertionError):
        net.forward(torch.zeros(2, 16, 16, 1), is_training=True)                  # nets/sphere.py:80 num_classes required


def test_asoftmax_lambda_schedule():
    from oracle import ops
    net = net_select('SphereNet-ASoftmax')
    for it in (0, 1, 100, 5000, 10 ** 6):
        net.global_step = it
        assert abs(net.current_lambda() - ops.asoftmax_lambda(it)) < 1e-12


def test_resnet_graph_on_cpu_matches_the_oracle_graph():
    """Same op list and variable table as the oracle's restatement of nets/resnet.py; BN+add+ReLU fusion plan."""
    from oracle import graphnet as og
    from tf_face_toolbox_amd.nets.resnet import ResNet
    net = ResNet(50)
    net.build(112, 112, 3, 1000, 'cpu')
    graph, spec = og.resnet_train_graph(50, 3, 1000)
    assert net.graph == graph and [(n, tuple(s), k) for n, s, k in spec] == [(n, tuple(v[0]), v[1]) for n, v in net.spec.items()]
    assert net.shapes['pool1'] == (28, 28, 64) and net.shapes['s5b2'] == (4, 4, 2048) and net.shapes['features'] == (2048,)
    kinds = [op[0] for op in net.plan]
    assert 'relu' not in kinds and 'add' not in kinds                     # all fused into the BN-apply launches
    fused = [op for op in net.plan if op[0] == 'bn' and op[4] is not None]
    assert len(fused) == 16 and all(op[5] == 1 for op in fused)           # one residual add + ReLU per bottleneck
    assert sum(v.size for v in net.variables.values() if v.kind == 'conv_w') == 23454912 + (160 - 147) * 64
    assert len(net.state) == 2 * 53 and [len(g) for g in net.param_list(True, True)] == [159, 1]
    with pytest.raises(NotImplementedError):
        ResNet(50, pre_act=True)
    with pytest.raises(ValueError, match='Unsupported num_layers.'):
        ResNet(38)


def test_labels_outside_num_classes_fail_at_setup():
    """the loss kernels index by label; the boundary validates resident label tensors once."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from fake_net import FakeOracleNet
    net = FakeOracleNet(3, 16, 16, 1, 6)
    x = torch.zeros(4, 16, 16, 1)
    for bad in ([0, 1, 6, 2], [0, -1, 2, 3]):
        with pytest.raises(ValueError, match=r'labels must lie in \[0, num_classes\)'):
            Singular(net, 0.1, 'Momentum')({'images': x, 'labels': torch.tensor(bad, dtype=torch.int32), 'num_classes': 6, 'num_examples': 4})
    Singular(net, 0.1, 'Momentum')({'images': x, 'labels': torch.tensor([0, 5, 2, 3], dtype=torch.int32), 'num_classes': 6, 'num_examples': 4})


def test_data_parallel_gives_every_rank_its_own_dropout_seed():
    """The reference's towers draw independent dropout masks; a replica-shared seed would repeat one mask per shard."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from fake_net import FakeOracleNet

    class Comm(object):
        def __init__(self, r): self.r = r
        def world_size(self): return 2
        def rank(self): return self.r
        def broadcast(self, t, src=0): pass
        def all_reduce_async(self, t): raise AssertionError('no step is run here')
    seeds = []
    for r in range(2):
        net = FakeOracleNet(3, 16, 16, 1, 6)
        net.dropout_seed = 5
        dp = DataParallel(net, 0.1, 'Momentum', num_gpus=2, comm=Comm(r))
        inputs = {'images': torch.zeros(4, 16, 16, 1), 'labels': torch.tensor([0, 1, 2, 3], dtype=torch.int32), 'num_classes': 6, 'num_examples': 4}
        dp(inputs)
        dp(inputs)                                    # building twice does not compound the seed
        seeds.append(net.dropout_seed)
    assert seeds == [10, 11]


# ------------------------------------------------------------------ round 3: precision modes, exits, loader shutdown
def test_precision_modes_are_three_contracts():
    """f32 | bf16 (bf16 MFMA operands, fp32 tensors) | bf16s (bf16 operands + bf16 storage): the library knows the MFMA dtype, the host
    flag says which entry points the nets call.  (fte_set_mfma_dtype only sets a flag: no GPU needed.)"""
    from tf_face_toolbox_amd import _lib
    try:
        for mode, dt, s16 in (('bf16s', 'bf16', True), ('bf16', 'bf16', False), ('f32', 'f32', False), ('bf16s', 'bf16', True)):
            _lib.set_mfma_dtype(mode)
            assert _lib.get_mfma_dtype() == dt and _lib.bf16_storage() is s16 and _lib.precision_mode() == mode
        with pytest.raises(ValueError):
            _lib.set_mfma_dtype('fp8')
    finally:
        _lib.set_mfma_dtype('f32')
    assert not _lib.bf16_storage()


def test_filter_pack_table_layout():
    """nets/_packs.py: destination offsets are compact and 16-byte aligned, the walk starts are prefix sums / 4, chunks of 64 rows."""
    import torch
    from tf_face_toolbox_amd.nets._packs import FilterPacks
    entries = [('c%d' % i, 100 + 7000 * i, 3 if i % 2 else 1, 32 * (1 + i % 3), 64) for i in range(70)]
    packs = FilterPacks(entries, 'cpu')
    assert [l[1] for l in packs.launches] == [64, 6]
    off = 0
    for li, (tab, n, total, base, _head) in enumerate(packs.launches):
        t = tab.tolist()
        start = 0
        for row, (name, src, k, cin, cout) in zip(t, entries[li * 64:li * 64 + n]):
            assert row[0] == src and row[1] == off - base and row[2:5] == [k * k, cin, cout] and row[5] * 4 == start
            assert (off * 2) % 16 == 0
            assert packs.w16[name].shape == (k, k, cin, cout) and packs.w16t[name].shape == (k, k, cout, cin)
            start += k * k * cin * cout
            off += k * k * cin * cout
        assert start == total
    assert packs.total == off


@pytest.mark.parametrize('body,code', [('return None', 0), ('raise SystemExit(3)', 3), ("raise SystemExit('bad flags')", 1),
                                       ('raise RuntimeError("boom")', 1), ('raise SystemExit(0)', 0)])
def test_cli_leaves_with_the_real_status(tmp_path, body, code):
    """train.py / evaluate.py: _run_and_leave maps what the program did to the exit status (round 2 left with an unconditional 0)."""
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = 'import sys; sys.path.insert(0, %r)\nimport train\ndef f():\n    %s\ntrain._run_and_leave(f)\n' % (root, body)
    r = subprocess.run([sys.executable, '-c', src], capture_output=True, text=True, timeout=120)
    assert r.returncode == code, (r.returncode, r.stderr[-500:])


@pytest.mark.parametrize('name', ['ResNet-50', 'ResNeXt-50-center', 'SENet-50-triplet', 'ShuffleNet-v2-small', 'ShuffleNet-v2-large', 'ResNet-26'])
def test_graph_net_gradient_buckets_follow_the_backward_walk(name):
    """data_parallel.py:88-113 reduces every gradient as it becomes ready; the graph nets hand DataParallel one bucket per backward
    segment (classifier, then the body from its last layers to its first): the buckets tile the arena once, line up with the
    backward stages, and every layer's filters lie in the bucket of the segment that computes their gradient."""
    net = net_select(name)
    net.build(112, 112, 3, 100, 'cpu')
    b, stages, segs = net.grad_buckets(), net.backward_stages(), net._segments()
    assert len(b) == len(stages) == len(segs) + (1 if net.has_classifier else 0)
    assert len(segs) >= 3, 'the body of %s is one all-reduce bucket' % name
    assert sum(e - a for a, e in b) == net.arena_size + 4
    assert sorted(b)[0][0] == 0 and all(x[1] == y[0] for x, y in zip(sorted(b), sorted(b)[1:]))      # disjoint, gap-free
    body = b[1:] if net.has_classifier else b
    assert [x[0] for x in body] == sorted((x[0] for x in body), reverse=True)                          # completion order: from the end
    nops = len(net.plan) - (1 if net.has_classifier else 0)
    assert segs[0][0] == 0 and segs[-1][1] == nops and all(x[1] == y[0] for x, y in zip(segs, segs[1:]))
    for lo, hi, a, e in segs:
        for j in range(lo, hi):
            for w in net._op_weight_names(net.plan[j]):
                v = net.variables[w]
                assert a <= v.offset and v.offset + v.size <= e, (name, w)
    sizes = [e - a for _, _, a, e in segs]
    assert max(sizes) <= 0.6 * sum(sizes)                                # no segment holds most of the body


def test_profile_tables_regenerate_from_the_committed_profiles(capsys):
    """scripts/hbm_table.py and scripts/make_symbol_lists.py read the committed round-5 profiles: the HBM table lists the BN kernels of
    every BN config with a share of 8 TB/s, and the symbol lists the GPU tests check are the ones these profiles produce."""
    import importlib.util
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('hbm_table', os.path.join(root, 'scripts', 'hbm_table.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    import sys
    argv = sys.argv
    sys.argv = ['hbm_table.py', 'r5']
    try:
        mod.main()
    finally:
        sys.argv = argv
    out = capsys.readouterr().out
    assert out == open(os.path.join(root, 'profiles', 'r5_hbm_bound_kernels.md')).read()
    for section in ('config 3: ResNeXt-50', 'config 4: SE-ResNet-50', 'config 5: ShuffleNet-v2', 'SphereNet-20, bf16 storage'):
        assert section in out
    assert out.count('bn_bwd_apply_kernel') >= 3 and 'at::native' not in out.split('### config 3')[1]
    for name, key in (('headline_symbols.json', 'conv_symbols'), ('symbols_resnext50_bf16s_b128.json', None)):
        d = json.load(open(os.path.join(root, 'tests', 'golden', name)))
        assert ('r6_' if name.startswith('headline') else 'r5_') in d['source'], name      # SphereNet: round 6's profiles; the BN nets': round 5's
        assert (d[key] if key else [v for k, v in d.items() if isinstance(v, list)][0])
