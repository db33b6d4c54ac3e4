"""CPU, world_size 2 over gloo: the N>1 path of data_parallel.py (product code) driven with an
oracle-backed test double.  Checks against the oracle's tower-split step (data_parallel.py:203-256):
rank shards, 1/num_gpus scaling, bucketed sum-all-reduce incl. the loss slots, replica broadcast at
start (train.py:101-120), identical replicas after every step."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

N, H, W, CH, NCLS, SEED, LR, STEPS = 4, 16, 16, 1, 6, 51, 0.05, 2


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _inputs():
    rng = np.random.default_rng(7)
    x = rng.uniform(-1, 1, (N, H, W, CH)).astype(np.float32)
    y = rng.integers(0, NCLS, N).astype(np.int32)
    return x, y


def _worker(rank, world, port, out_dir, shard_mode):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from fake_net import FakeOracleNet
    from tf_face_toolbox_amd.data_parallel import DataParallel_margin
    x, y = _inputs()
    # replicas start DIFFERENT on purpose: the initial broadcast must make them equal to rank 0
    net = FakeOracleNet(SEED, H, W, CH, NCLS, perturb_rank=rank)
    xt, yt = torch.from_numpy(x), torch.from_numpy(y)
    inputs = {'images': xt, 'labels': yt, 'num_classes': NCLS, 'num_examples': N}
    if shard_mode == 'own_rows':                     # each rank hands over only its own rows
        sh = N // world
        inputs = {'images': xt[rank * sh:(rank + 1) * sh], 'labels': yt[rank * sh:(rank + 1) * sh],
                  'num_classes': NCLS, 'num_examples': N, 'batch_size': N}
    model = DataParallel_margin(net, LR, 'Momentum', num_gpus=world, weight_decay=5e-4)
    train_ops, losses, names, others = model(inputs)
    start = net.params.clone()
    loss_log = []
    for _ in range(STEPS):
        train_ops()
        loss_log.append([float(v) for v in losses])
    assert net.stage_log[-2:] == ['head', 'body'] and model.global_step == STEPS
    torch.save({'start': start, 'params': net.params.clone(), 'losses': loss_log, 'names': names},
               os.path.join(out_dir, 'rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('shard_mode', ['split_global', 'own_rows'])
def test_two_rank_data_parallel_equals_oracle_tower_split(tmp_path, shard_mode):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), shard_mode), nprocs=2, join=True)
    r0 = torch.load(os.path.join(str(tmp_path), 'rank0.pt'))
    r1 = torch.load(os.path.join(str(tmp_path), 'rank1.pt'))
    assert r0['names'] == ['cross_entropy', 'reg_loss']
    # broadcast: both replicas start from rank 0's values, and stay bit-identical afterwards
    assert torch.equal(r0['start'], r1['start'])
    assert torch.equal(r0['params'], r1['params'])
    assert r0['losses'] == r1['losses']
    # oracle: same global batch through data_parallel.py's tower-split algebra
    sys.path.insert(0, HERE)
    from fake_net import FakeOracleNet
    from oracle import spherenet as osn
    ref = FakeOracleNet(SEED, H, W, CH, NCLS, perturb_rank=0)
    p = ref.as_dict()
    slots = osn.zero_slots(p)
    x, y = _inputs()
    for t in range(STEPS):
        p_next, slots, l = osn.train_step(p, slots, x.astype(np.float64), y, LR, num_towers=2)
        assert abs(r0['losses'][t][0] - l[0]) < 1e-12 and abs(r0['losses'][t][1] - l[1]) < 1e-12
        p = p_next
    got = r0['params'].numpy()
    for k in ref.names:
        o = ref.offsets[k]
        np.testing.assert_allclose(got[o:o + p[k].size].reshape(p[k].shape), p[k], rtol=0, atol=1e-13)


def test_shard_rejects_indivisible_batches():
    sys.path.insert(0, HERE)
    from tf_face_toolbox_amd.data_parallel import DataParallel

    class C(object):
        def rank(self):
            return 0
    d = DataParallel(object(), 0.1, 'Momentum', num_gpus=2, comm=C())
    with pytest.raises(AssertionError, match='divisible'):
        d._shard(torch.zeros(5, 3))
    assert d._shard(torch.arange(8).reshape(4, 2)).tolist() == [[0, 1], [2, 3]]
