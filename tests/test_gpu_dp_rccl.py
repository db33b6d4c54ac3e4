"""-m gpu: DataParallel's RCCL code path on the one GPU a test box has.  A world_size-1 'nccl' group
still runs the real thing end to end -- async bucketed all_reduce on RCCL's stream, the event
hand-offs between that stream and the kernel stream, the arena broadcast -- and with one replica
the result must equal Singular's bit for bit (the multi-rank algebra is covered on CPU over gloo in
tests/test_data_parallel_gloo.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

from oracle import spherenet as osn

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from util_gpu import dev, host
    from tf_face_toolbox_amd import net_select, Singular, DataParallel_margin


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize('name', ['SphereNet', 'SphereNet-ASoftmax'])
def test_single_rank_rccl_data_parallel_equals_singular(name):
    n, h, w, ch, ncls = 8, 32, 32, 3, 20
    p = osn.perturb_params(osn.init_params(61, ch, ncls, h, w), 62)
    rng = np.random.default_rng(63)
    x = rng.uniform(-1, 1, (n, h, w, ch)); y = rng.integers(0, ncls, n)
    inputs = {'images': dev(x), 'labels': dev(y, torch.int32), 'num_classes': ncls, 'num_examples': n}

    ref = net_select(name, 'NCHW', 5e-4); ref.build(h, w, ch, ncls, 'cuda'); ref.load_params(p)
    step, losses, _, _ = Singular(ref, 0.05, 'Momentum')(inputs)
    for _ in range(3):
        step()
    ref_losses = [float(v) for v in losses]

    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(_free_port())
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', torch.cuda.current_device()))
    try:
        net = net_select(name, 'NCHW', 5e-4); net.build(h, w, ch, ncls, 'cuda'); net.load_params(p)
        model = DataParallel_margin(net, 0.05, 'Momentum', num_gpus=2)
        model.num_gpus = 1                      # a one-replica "multi-GPU" run: shard = whole batch, scale 1/1
        step, losses, names, others = model(inputs)
        for _ in range(3):
            step()
        got_losses = [float(v) for v in losses]
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    assert got_losses == ref_losses
    assert torch.equal(net.params, ref.params)


@pytest.mark.parametrize('name', ['ResNeXt-26', 'ShuffleNet-v2-small'])
def test_single_rank_rccl_bn_nets_equal_singular(name):
    """The graph-engine nets (one all-reduce bucket per backward segment: classifier, then the body from its last layers to its first;
    filter gradients on a second stream, joined at every segment's end) through the
    same world-size-1 RCCL run: async all-reduce per backward stage, per-bucket optimizer after each wait -- bit-identical to
    Singular (same seeds: one replica's dropout seed is base * 1 + 0)."""
    n, h, w, ch, ncls = 8, 64, 64, 3, 10
    rng = np.random.default_rng(64)
    x = rng.uniform(-1, 1, (n, h, w, ch)); y = rng.integers(0, ncls, n)
    inputs = {'images': dev(x), 'labels': dev(y, torch.int32), 'num_classes': ncls, 'num_examples': n}

    def make():
        net = net_select(name, 'NCHW', 5e-4)
        net.seed = 7
        net.build(h, w, ch, ncls, 'cuda')
        return net
    ref = make()
    step, losses, _, _ = Singular(ref, 0.05, 'Momentum')(inputs)
    for _ in range(3):
        step()
    ref_losses = [float(v) for v in losses]
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(_free_port())
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', torch.cuda.current_device()))
    try:
        net = make()
        model = DataParallel_margin(net, 0.05, 'Momentum', num_gpus=2)
        model.num_gpus = 1
        assert len(net.grad_buckets()) == len(net.backward_stages()) >= 4      # classifier + the body's backward segments
        step, losses, names, others = model(inputs)
        for _ in range(3):
            step()
        got_losses = [float(v) for v in losses]
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    assert got_losses == ref_losses
    assert torch.equal(net.params, ref.params)
    for k in ref.state:
        assert torch.equal(net.state[k], ref.state[k]), k
