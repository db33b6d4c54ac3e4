"""-m gpu: a real 2-rank data-parallel run (two processes launched the way the driver launches bench.py).  With two or more
GPUs on the box the ranks take one device each and talk over RCCL ('nccl' backend, tests/dp_worker.py); on the one-GPU test box
they share the device and gloo is the transport.  Checks SURVEY 8c's N-rank pins with the HIP kernels in the loop: replicas identical
after the initial broadcast and after every step, and the result equal to the oracle's 2-tower step
(data_parallel.py:203-256: shard, 1/n pre-scale, SUM, same update everywhere)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import spherenet as osn

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _check_transport(r0, r1):
    """>= 2 GPUs: both ranks ran RCCL on their own device; one GPU: gloo on device 0."""
    if torch.cuda.device_count() >= 2 and os.environ.get('FTE_TEST_FORCE_GLOO') != '1':
        assert str(r0['backend']) == 'nccl' == str(r1['backend']) and int(r0['device']) == 0 and int(r1['device']) == 1
    else:
        assert str(r0['backend']) == 'gloo' == str(r1['backend'])


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize('name,head', [('SphereNet', 'softmax'), ('SphereNet-ASoftmax', 'asoftmax')])
def test_two_ranks_equal_the_oracle_two_tower_step(tmp_path, name, head):
    n, h, w, ch, ncls, steps = 8, 32, 32, 3, 20, 2
    p = osn.perturb_params(osn.init_params(71, ch, ncls, h, w), 72)
    rng = np.random.default_rng(73)
    x = rng.uniform(-1, 1, (n, h, w, ch)); y = rng.integers(0, ncls, n)
    fix = str(tmp_path / 'fix.npz')
    np.savez(fix, x=x, y=y, ncls=ncls, **{'p:' + k: v for k, v in p.items()})
    out = str(tmp_path / 'out')
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
                        os.path.join(ROOT, 'tests', 'dp_worker.py'), fix, out, name, str(steps)],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:]
    r0, r1 = np.load(out + '.rank0.npz'), np.load(out + '.rank1.npz')
    _check_transport(r0, r1)
    for k in r0.files:                                     # replicas bit-identical (weights AND displayed losses)
        if k not in ('backend', 'device'):
            np.testing.assert_array_equal(r0[k], r1[k], err_msg=k)
    # oracle: the same two steps with 2 towers
    ref = dict(p)
    slots = {k: np.zeros_like(v) for k, v in p.items()}
    ref_losses = []
    for t in range(steps):
        lam = None
        if head == 'asoftmax':
            from oracle import ops
            lam = ops.asoftmax_lambda(t)
        ref, slots, ls = osn.train_step(ref, slots, x, y, 0.05, num_towers=2, weight_decay=5e-4, data_format='NCHW', head=head, lam=lam)
        ref_losses.append(ls)
    np.testing.assert_allclose(r0['losses'], np.array(ref_losses), rtol=2e-5)
    for k in p:
        a, b = r0['w:' + k].astype(np.float64), ref[k]
        assert np.sqrt(((a - b) ** 2).sum()) <= 2e-5 * max(np.sqrt((b * b).sum()), 1e-30), k


@pytest.mark.parametrize('world', [4, 8])
def test_n_ranks_equal_the_oracle_n_tower_step(tmp_path, world):
    """The same pins at the world sizes the driver's scaling run uses (BASELINE.json: 1 / 2 / 4 / 8 GPUs; data_parallel.py:203-256,
    train.py:98,101-120): 4 and 8 ranks, two images each (softmax head: the A-softmax margin's angular thresholds are pinned by the 2-rank test), replicas started different, equal after every step and equal to the
    oracle's `world`-tower step.  One GPU per rank over RCCL when the box has them, else all ranks on the GPUs there are over gloo."""
    n, h, w, ch, ncls, steps = 2 * world, 32, 32, 3, 20, 2
    p = osn.perturb_params(osn.init_params(81, ch, ncls, h, w), 82)
    rng = np.random.default_rng(83)
    x = rng.uniform(-1, 1, (n, h, w, ch)); y = rng.integers(0, ncls, n)
    fix = str(tmp_path / 'fix.npz')
    np.savez(fix, x=x, y=y, ncls=ncls, **{'p:' + k: v for k, v in p.items()})
    out = str(tmp_path / 'out')
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS='2')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world),
                        '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
                        os.path.join(ROOT, 'tests', 'dp_worker.py'), fix, out, 'SphereNet', str(steps)],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-4000:]
    res = [np.load(out + '.rank%d.npz' % k) for k in range(world)]
    rccl = torch.cuda.device_count() >= world and os.environ.get('FTE_TEST_FORCE_GLOO') != '1'
    for k, rk in enumerate(res):
        assert str(rk['backend']) == ('nccl' if rccl else 'gloo')
        if rccl:
            assert int(rk['device']) == k
        for key in res[0].files:                               # replicas bit-identical (weights AND displayed losses)
            if key not in ('backend', 'device'):
                np.testing.assert_array_equal(res[0][key], rk[key], err_msg='rank %d %s' % (k, key))
    ref = dict(p)
    slots = {k: np.zeros_like(v) for k, v in p.items()}
    ref_losses = []
    for t in range(steps):
        ref, slots, ls = osn.train_step(ref, slots, x, y, 0.05, num_towers=world, weight_decay=5e-4, data_format='NCHW', head='softmax')
        ref_losses.append(ls)
    np.testing.assert_allclose(res[0]['losses'], np.array(ref_losses), rtol=2e-5)
    worst = max((np.sqrt(((res[0]['w:' + k].astype(np.float64) - ref[k]) ** 2).sum()) / max(np.sqrt((ref[k] * ref[k]).sum()), 1e-30), k) for k in p)
    assert worst[0] <= 2e-5, worst


def test_bench_py_runs_with_eight_ranks(tmp_path):
    """`python bench.py --gpus 8` as the driver's scaling run starts it (its own ranks), at 8 images per rank: ONE JSON line, the five
    gradient buckets summing to the arena, finite losses.  (On the one-GPU test box the eight ranks share it over gloo.)"""
    import json
    shared = torch.cuda.device_count() < 8
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS='2')
    if shared:
        env['FTE_BENCH_SHARED_GPU'] = '1'
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '2', '--warmup', '1', '--global-batch', '64'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500, cwd=str(tmp_path))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 8 and out['steps'] == 2 and out['scaling'] == 'strong'
    assert out['config']['global_batch'] == 64 and out['config']['per_gpu_batch'] == 8 and out['config']['parallelism'] == 'dp8'
    assert out['value'] > 0 and all(np.isfinite(v) for v in out['losses'].values())
    ar = out['allreduce']
    assert ar['rccl_ranks'] == (0 if shared else 8)
    assert len(ar['bucket_bytes']) == 5 and sum(ar['bucket_bytes']) == 4 * (out_arena(out) + 4)
    assert 'other_configs' not in out and out['cpu_baseline'] is None


def test_train_py_launches_eight_ranks(tmp_path):
    """`python train.py --num_gpus 8` as the reference is invoked (train.py:98,178-184): the script starts its eight ranks itself; three
    steps on a resident synthetic batch, the log lines of rank 0 only, exit status 0."""
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS='2')
    if torch.cuda.device_count() < 8:
        env['FTE_BENCH_SHARED_GPU'] = '1'
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'train.py'), '--net_name', 'SphereNet-ASoftmax', '--model_name', 'n8', '--synthetic', '1',
                        '--synthetic_classes', '32', '--input_height', '32', '--input_width', '32', '--batch_size', '16', '--num_gpus', '8',
                        '--init_lr', '0.01', '--lr_decay_epoch', '2', '--max_epoches', '3', '--display_interval', '1', '--save_interval', '1000',
                        '--max_steps', '3'], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-4000:]
    assert r.stdout.count('Loss #0: cross_entropy') == 3 and 'throughput =' in r.stdout


@pytest.mark.parametrize('launcher', ['self', 'torchrun'])
def test_bench_py_runs_with_two_ranks(tmp_path, launcher):
    """bench.py for N = 2, both ways the driver may start it: plain `python bench.py --gpus 2` (the script launches its own
    ranks as child processes before touching the GPU and relays rank 0's line) and under torch.distributed.run.  One JSON
    line from rank 0; the two ranks share the box's GPU over gloo (FTE_BENCH_SHARED_GPU=1)."""
    import json
    shared = torch.cuda.device_count() < 2             # one GPU: both ranks on it, gloo; otherwise the driver's configuration (RCCL)
    env = dict(os.environ, PYTHONPATH=ROOT)
    if shared:
        env['FTE_BENCH_SHARED_GPU'] = '1'
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    tail = [os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--global-batch', '16']
    if launcher == 'self':
        cmd = [sys.executable] + tail
    else:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
               '--master-addr', '127.0.0.1', '--master-port', str(_free_port())] + tail
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout                       # exactly ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 3 and out['warmup'] == 1 and out['scaling'] == 'strong'
    assert out['config']['global_batch'] == 16 and out['config']['per_gpu_batch'] == 8 and out['config']['parallelism'] == 'dp2'
    assert out['value'] > 0 and abs(out['value'] - 16 / (out['ms_per_step'] * 1e-3)) <= 0.01 * out['value']
    assert out['cpu_baseline'] is None and out['roofline']['frac'] > 0
    assert all(np.isfinite(v) for v in out['losses'].values())
    ar = out['allreduce']
    if shared:
        assert ar['rccl_ranks'] == 0 and ar['backend'].startswith('gloo')      # the test transport
    else:
        assert ar['rccl_ranks'] == 2 and ar['backend'].startswith('nccl')      # one GPU per rank: RCCL, as the driver runs it
    assert len(ar['bucket_bytes']) == 5 == len(ar['bucket_alone_ms']) and sum(ar['bucket_bytes']) == 4 * (out_arena(out) + 4)
    assert ar['ms_per_step_without_allreduce'] > 0
    assert {'fwd', 'dgrad', 'wgrad'} <= set(e['op'] for e in out['roofline']['per_shape'])


def out_arena(out):
    """floats of SphereNet-20's parameter arena at C = 10575 (classifier padded to 10624 columns)."""
    from tf_face_toolbox_amd import net_select
    net = net_select('SphereNet-ASoftmax', 'NCHW', 5e-4)
    net.build(112, 112, 3, 10575, 'cpu')
    return net.arena_size


def test_bench_self_launch_propagates_failure(tmp_path):
    """A failing rank must fail the command (the driver reads the exit code), not print a half-made line."""
    env = dict(os.environ, PYTHONPATH=ROOT, FTE_BENCH_SHARED_GPU='1', FTE_LIB=str(tmp_path / 'missing.so'))
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0', '--global-batch', '16'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith('{')]


@pytest.mark.parametrize('name', ['ResNeXt-26', 'ShuffleNet-v2-small'])
def test_two_ranks_bn_nets_stay_identical(tmp_path, name):
    """The graph-engine nets under the same 2-rank run: per-shard BN statistics and per-rank dropout masks differ by design
    (data_parallel.py:242-243), the all-reduced gradients and therefore every trainable variable must not."""
    n, h, w, ch, ncls, steps = 8, 32, 32, 3, 10, 3
    rng = np.random.default_rng(5)
    fix = str(tmp_path / 'fix.npz')
    np.savez(fix, x=rng.uniform(-1, 1, (n, h, w, ch)), y=rng.integers(0, ncls, n), ncls=ncls)
    out = str(tmp_path / 'out')
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
                        os.path.join(ROOT, 'tests', 'dp_worker.py'), fix, out, name, str(steps)],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:]
    r0, r1 = np.load(out + '.rank0.npz'), np.load(out + '.rank1.npz')
    moved = 0
    for k in r0.files:
        if k.startswith('w:'):
            np.testing.assert_array_equal(r0[k], r1[k], err_msg=k)
            moved += int(np.abs(r0[k]).sum() > 0)
    assert moved > 10 and np.isfinite(r0['losses']).all()
    np.testing.assert_array_equal(r0['losses'], r1['losses'])        # displayed losses are all-reduced means


def test_two_ranks_center_loss_state_is_per_replica(tmp_path):
    """Pins what the center loss's `centers` do under data parallelism (loss.py:34-39 + data_parallel.py:216-223): the table
    is created inside each tower's variable scope, every tower scatter_subs ITS shard's rows into ITS copy, and nothing but
    tf.gradients outputs is all-reduced (data_parallel.py:179) -- so after one step rank r's centers equal the single-tower
    update computed from shard r (oracle, from the broadcast weights), the two replicas' tables differ, the trainable
    variables do not.  (Faithful to the reference, documented in DESIGN.md section 6; replica 0's table is what a checkpoint keeps,
    saver.py:36-40.)"""
    from oracle import graphnet as og, ops as oops
    n, h, w, ch, ncls = 8, 32, 32, 3, 10
    rng = np.random.default_rng(15)
    x = rng.uniform(-1, 1, (n, h, w, ch))
    y = np.array([0, 1, 2, 0, 5, 6, 5, 7])                       # duplicates inside each shard, class 0 / 5 only in one shard
    graph, spec = og.resnet_train_graph(26, ch, ncls, 'resnext')
    p, state = og.init_params(spec, 16)
    p = og.perturb(p, 17)
    fix = str(tmp_path / 'fix.npz')
    np.savez(fix, x=x, y=y, ncls=ncls, **{'p:' + k: v for k, v in p.items()})
    out = str(tmp_path / 'out')
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
                        os.path.join(ROOT, 'tests', 'dp_worker.py'), fix, out, 'ResNeXt-26-center', '1'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:]
    r0, r1 = np.load(out + '.rank0.npz'), np.load(out + '.rank1.npz')
    for k in r0.files:
        if k.startswith('w:'):
            np.testing.assert_array_equal(r0[k], r1[k], err_msg=k)            # replicas' trainable variables identical
    c0, c1 = r0['s:centers'], r1['s:centers']
    assert np.abs(c0 - c1).max() > 0                                          # the tables diverge, by construction
    for rank, got in ((0, c0), (1, c1)):
        sl = slice(rank * 4, rank * 4 + 4)
        env_, _, _ = og.forward(graph, p, x[sl], train=True, masks={'features_drop': np.ones((4, 2048))}, state=state)
        _, _, newc = oops.center_loss(env_['features'], y[sl], np.zeros((ncls, 2048)), 0.99)
        err = np.abs(got - newc).max()
        assert err <= 5e-4 * np.abs(newc).max(), (rank, err)      # centers = 0.01 x features of a BN net at 4 images per tower: fp32 noise ~1e-4
        untouched = [c for c in range(ncls) if c not in set(y[sl])]
        assert np.abs(got[untouched]).max() == 0                              # rows of classes outside the shard stay zero


def test_two_ranks_center_loss_reconciled_tables(tmp_path):
    """DataParallel(sync_centers=True) (opt-in; the default above is the reference's per-tower tables): the ranks all-gather
    each step's (labels, f - c_y) rows and every rank applies all of them, so after a step BOTH replicas hold the SAME table,
    bit for bit, and it equals the single-tower scatter_sub of the GLOBAL batch (loss.py:37-39 applied to the concatenated
    shards; features per shard, as each tower's BN sees only its own rows)."""
    from oracle import graphnet as og, ops as oops
    n, h, w, ch, ncls = 8, 32, 32, 3, 10
    rng = np.random.default_rng(15)
    x = rng.uniform(-1, 1, (n, h, w, ch))
    y = np.array([0, 1, 2, 0, 5, 0, 5, 7])                       # class 0 in both shards, duplicates inside each
    graph, spec = og.resnet_train_graph(26, ch, ncls, 'resnext')
    p, state = og.init_params(spec, 16)
    p = og.perturb(p, 17)
    fix = str(tmp_path / 'fix.npz')
    np.savez(fix, x=x, y=y, ncls=ncls, **{'p:' + k: v for k, v in p.items()})
    out = str(tmp_path / 'out')
    env = dict(os.environ, PYTHONPATH=ROOT, FTE_TEST_SYNC_CENTERS='1')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
                        os.path.join(ROOT, 'tests', 'dp_worker.py'), fix, out, 'ResNeXt-26-center', '1'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:]
    r0, r1 = np.load(out + '.rank0.npz'), np.load(out + '.rank1.npz')
    for k in r0.files:
        if k.startswith('w:'):
            np.testing.assert_array_equal(r0[k], r1[k], err_msg=k)
    c0, c1 = r0['s:centers'], r1['s:centers']
    np.testing.assert_array_equal(c0, c1)                                       # ONE table
    feats = []
    for rank in range(2):
        sl = slice(rank * 4, rank * 4 + 4)
        env_, _, _ = og.forward(graph, p, x[sl], train=True, masks={'features_drop': np.ones((4, 2048))}, state=state)
        feats.append(env_['features'])
    _, _, newc = oops.center_loss(np.concatenate(feats), y, np.zeros((ncls, 2048)), 0.99)
    err = np.abs(c0 - newc).max()
    assert err <= 5e-4 * np.abs(newc).max(), err
    assert np.abs(c0[[3, 4, 6, 8, 9]]).max() == 0 and np.abs(c0[0]).max() > 0


@pytest.mark.parametrize('name,b', [('ResNeXt-26-center', 8), ('SENet-50-triplet', 8)])
def test_bench_net_runs_a_bn_net_with_two_ranks(tmp_path, name, b):
    """scripts/bench_net.py --gpus 2 (its own ranks, as bench.py): a graph net as two DataParallel replicas, gradients reduced in the
    buckets of its backward segments (classifier-less nets included) -- the line carries the allreduce block."""
    import re
    shared = torch.cuda.device_count() < 2
    env = dict(os.environ, PYTHONPATH=ROOT, FTE_MFMA_DTYPE='bf16s')
    if shared:
        env['FTE_BENCH_SHARED_GPU'] = '1'
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'scripts', 'bench_net.py'), name, str(b), '3', '--gpus', '2']
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    m = re.search(r'x 2 GPUs: ([0-9.]+) ms/step', r.stdout)
    assert m and float(m.group(1)) > 0, r.stdout
    m = re.search(r'allreduce: backend (\w+), (\d+) buckets', r.stdout)
    assert m and m.group(1) == ('gloo' if shared else 'nccl') and int(m.group(2)) >= 4, r.stdout
