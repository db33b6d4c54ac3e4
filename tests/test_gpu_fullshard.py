"""-m gpu: a REAL training step of BASELINE.json configs 3 and 5 at their per-GPU shard (ResNeXt-50 + center loss, 128 x 112 x 112,
bf16 storage; ShuffleNet-v2 x2.0, 256 x 112 x 112, fp32), audited call by call against the float64 oracle (tests/step_audit.py): every
conv / grouped-conv / depthwise / batch-norm entry point the step calls is checked ONCE per distinct shape on the call's own inputs --
training-mode batch statistics over the whole shard included -- and the launch records name the kernel symbol each MFMA launch ran
on.  The symbol lists of the profiled runs (tests/golden/symbols_*.json, generated from profiles/ by scripts/make_symbol_lists.py)
must be covered: every conv-family symbol of a profile ran here, in a call compared with the oracle.
Reference: nets/resnext.py:34-67, nets/resnet.py:47-61,63-92,97-99, nets/shufflenet_v2.py:87-135, loss.py:29-45."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))

if torch.cuda.is_available():
    from step_audit import Audit
    from tf_face_toolbox_amd import net_select, _lib

CONV_FAMILIES = ('igemm_kernel', 'igemm_bn_kernel', 'igemm16', 'wgrad16', 'pw16_kernel')


def _audited_step(name, n, mode, ncls=10575):          # C = 10,575 as in the profiled runs
    _lib.set_mfma_dtype(mode)
    try:
        g = torch.Generator().manual_seed(3)
        x = (torch.rand(n, 112, 112, 3, generator=g) * 2 - 1).cuda()
        y = torch.randint(0, ncls, (n,), generator=g, dtype=torch.int32).cuda()
        net = net_select(name, 'NCHW', 5e-4)
        net.build(112, 112, 3, ncls, 'cuda')
        net.dropout_seed = 5
        with Audit(net, bf16_operands=mode != 'f32') as au:
            out = net.forward(x, num_classes=ncls, is_training=True)
            losses, names, _ = net.loss_function('T', y, **out)
            net.backward()
            torch.cuda.synchronize()
        assert all(np.isfinite(float(v)) for v in losses)
        return au
    finally:
        _lib.set_mfma_dtype('f32')


def _covered(au, golden):
    want = json.load(open(os.path.join(HERE, 'golden', golden)))
    missing = [s for s in want['conv_symbols'] if s.startswith(CONV_FAMILIES) and s not in au.symbols]
    return want, missing


def test_resnext50_center_bf16s_step_at_128_images_call_by_call():
    au = _audited_step('ResNeXt-50-center', 128, 'bf16s')
    print('ResNeXt-50-center bf16s @128: checked %s; worst error / limit %s; symbols %s' % (
        au.checked, {k: round(v, 3) for k, v in au.worst.items()}, sorted(au.symbols)))
    # 36 convs (+ their statistics), 16 grouped 3x3s, every batch norm's backward, data / filter gradients of both kinds
    assert au.checked.get('fte_conv2d_bn_fwd', 0) >= 14 and au.checked.get('fte_gconv3x3_bn_fwd_bf16_s16', 0) >= 6
    assert au.checked.get('fte_conv2d_dgrad_s16', 0) >= 14 and au.checked.get('fte_conv2d_wgrad16', 0) >= 14
    assert au.checked.get('fte_bn_train_bwd_s16', 0) >= 10 and au.checked.get('fte_gconv3x3_wgrad_bf16_s16', 0) >= 6
    assert au.checked.get('fte_bn_apply', 0) >= 4
    want, missing = _covered(au, 'symbols_resnext50_bf16s_b128.json')
    assert not missing, 'conv symbols of %s never compared with the oracle: %s' % (want['source'], missing)


def test_shufflenet_fp32_step_at_256_images_call_by_call():
    au = _audited_step('ShuffleNet-v2-small', 256, 'f32')
    print('ShuffleNet-v2-small fp32 @256: checked %s; worst error / limit %s; symbols %s' % (
        au.checked, {k: round(v, 3) for k, v in au.worst.items()}, sorted(au.symbols)))
    assert au.checked.get('fte_conv2d_bn_fwd', 0) >= 8 and au.checked.get('fte_dwconv3x3_fwd', 0) >= 4
    assert au.checked.get('fte_conv2d_dgrad', 0) >= 8 and au.checked.get('fte_conv2d_wgrad', 0) >= 8
    assert au.checked.get('fte_dwconv3x3_dgrad', 0) >= 4 and au.checked.get('fte_dwconv3x3_wgrad', 0) >= 4
    want, missing = _covered(au, 'symbols_shufflenet_f32_b256.json')
    assert not missing, 'conv symbols of %s never compared with the oracle: %s' % (want['source'], missing)


def test_senet50_triplet_bf16s_step_at_128_images_call_by_call():
    au = _audited_step('SENet-50-triplet', 128, 'bf16s')
    print('SENet-50-triplet bf16s @128: checked %s; worst error / limit %s; symbols %s' % (
        au.checked, {k: round(v, 3) for k, v in au.worst.items()}, sorted(au.symbols)))
    assert au.checked.get('fte_conv2d_bn_fwd', 0) >= 20 and au.checked.get('fte_conv2d_wgrad16', 0) >= 20
    want, missing = _covered(au, 'symbols_senet50_bf16s_b128.json')
    assert not missing, 'conv symbols of %s never compared with the oracle: %s' % (want['source'], missing)
