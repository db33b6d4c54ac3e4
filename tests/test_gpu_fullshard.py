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

OPTIMIZER_FAMILIES = ('momentum_kernel', 'adam_kernel', 'partial_sum_kernel', 'final_sum_kernel')      # run outside forward + backward: checked by _optimizer()


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
        _optimizer(net)
        return au
    finally:
        _lib.set_mfma_dtype('f32')


def _covered(au, golden):
    """(i) every conv-family symbol of the profiled run ran in a call compared with the oracle (launch records); (ii) every kernel FAMILY
    with >= 1 % of the profile's kernel time -- the streaming kernels too -- is covered by such a call: a launch record's family, a
    checked entry point's kernels (step_audit.ENTRY_KERNELS), or the optimizer check below."""
    from step_audit import ENTRY_KERNELS
    want = json.load(open(os.path.join(HERE, 'golden', golden)))
    missing = [s for s in want['conv_symbols'] if s not in au.symbols]
    fam = {s.split('<', 1)[0] for s in au.symbols} | set(OPTIMIZER_FAMILIES)
    for fn, k in au.checked.items():
        if k > 0:
            fam.update(ENTRY_KERNELS.get(fn, []))
    missing += [f for f in want.get('top_families', []) if f not in fam]
    return want, missing


def _optimizer(net):
    """the optimizer's pass over the net's whole arena (its kernels are >= 1 % of a profiled step): fte_momentum_update against the
    float64 update rule (acc <- 0.9 acc + g, w <- w - lr acc; g = gscale * grad + wd * w)"""
    from oracle import ops as oops
    from util_gpu import host, check_maxabs
    nn = net.arena_size
    g = torch.Generator().manual_seed(11)
    w = torch.randn(nn, generator=g).cuda(); acc = (torch.randn(nn, generator=g) * 0.1).cuda(); gr = torch.randn(nn, generator=g).cuda()
    w0, a0 = host(w), host(acc)
    _lib.call('fte_momentum_update', w, acc, gr, nn, 0.1, 0.9, 5e-4, 0.5, torch.cuda.current_stream().cuda_stream)
    w_ref, a_ref = oops.momentum_step(w0, a0, 0.5 * host(gr) + 5e-4 * w0, 0.1)
    check_maxabs(host(w), w_ref, 1e-6, 'momentum w'); check_maxabs(host(acc), a_ref, 1e-6, 'momentum acc')


def test_resnext50_center_bf16s_step_at_128_images_call_by_call():
    au = _audited_step('ResNeXt-50-center', 128, 'bf16s')
    print('ResNeXt-50-center bf16s @128: checked %s; worst error / limit %s; symbols %s' % (
        au.checked, {k: round(v, 3) for k, v in au.worst.items()}, sorted(au.symbols)))
    # 36 convs (+ their statistics), 16 grouped 3x3s, every batch norm's backward, data / filter gradients of both kinds
    assert au.checked.get('fte_conv2d_bn_fwd', 0) >= 14 and au.checked.get('fte_gconv3x3_bn_fwd_bf16_s16', 0) >= 6
    assert au.checked.get('fte_conv2d_dgrad_s16', 0) >= 14 and au.checked.get('fte_conv2d_wgrad16', 0) >= 14
    assert au.checked.get('fte_bn_train_bwd_s16', 0) >= 10 and au.checked.get('fte_gconv3x3_wgrad_bf16_s16', 0) >= 6
    assert au.checked.get('fte_bn_apply', 0) >= 4
    # round 5: the streaming kernels of the step too (stem im2col, max-pool, pooling, filter packs)
    for fn in ('fte_pack_weights_bf16_table', 'fte_im2col_first_s16', 'fte_gap_fwd_s16', 'fte_gap_bwd_s16', 'fte_maxpool3x3s2_fwd_s16', 'fte_maxpool3x3s2_bwd_s16'):
        assert au.checked.get(fn, 0) >= 1, (fn, au.checked)
    want, missing = _covered(au, 'symbols_resnext50_bf16s_b128.json')
    assert not missing, 'kernels of %s never compared with the oracle: %s' % (want['source'], missing)


def test_shufflenet_fp32_step_at_256_images_call_by_call():
    au = _audited_step('ShuffleNet-v2-small', 256, 'f32')
    print('ShuffleNet-v2-small fp32 @256: checked %s; worst error / limit %s; symbols %s' % (
        au.checked, {k: round(v, 3) for k, v in au.worst.items()}, sorted(au.symbols)))
    assert au.checked.get('fte_conv2d_bn_fwd', 0) >= 8 and au.checked.get('fte_dwconv3x3_fwd', 0) >= 4
    assert au.checked.get('fte_conv2d_dgrad', 0) >= 8 and au.checked.get('fte_conv2d_wgrad', 0) >= 8
    assert au.checked.get('fte_dwconv3x3_dgrad', 0) >= 4 and au.checked.get('fte_dwconv3x3_wgrad', 0) >= 4
    assert au.checked.get('fte_channel_gather_affine', 0) >= 6 and au.checked.get('fte_maxpool3x3s2_fwd', 0) >= 1 and au.checked.get('fte_gap_fwd', 0) >= 1
    want, missing = _covered(au, 'symbols_shufflenet_f32_b256.json')
    assert not missing, 'kernels of %s never compared with the oracle: %s' % (want['source'], missing)


def test_senet50_triplet_bf16s_step_at_128_images_call_by_call():
    au = _audited_step('SENet-50-triplet', 128, 'bf16s')
    print('SENet-50-triplet bf16s @128: checked %s; worst error / limit %s; symbols %s' % (
        au.checked, {k: round(v, 3) for k, v in au.worst.items()}, sorted(au.symbols)))
    assert au.checked.get('fte_conv2d_bn_fwd', 0) >= 20 and au.checked.get('fte_conv2d_wgrad16', 0) >= 20
    # the SE residual blocks, fused (round 5): one check per distinct block shape of the four stages, every kernel of the block
    for fn in ('fte_se_squeeze', 'fte_se_apply_fwd', 'fte_se_bwd_gate', 'fte_se_bn_bwd_coef', 'fte_se_bn_bwd_apply'):
        assert au.checked.get(fn, 0) >= 4, (fn, au.checked)
    assert au.checked.get('fte_batch_hard_triplet_fwd_bwd', 0) >= 1 and au.checked.get('fte_maxpool3x3s2_bwd_s16', 0) >= 1
    want, missing = _covered(au, 'symbols_senet50_bf16s_b128.json')
    assert not missing, 'kernels of %s never compared with the oracle: %s' % (want['source'], missing)
