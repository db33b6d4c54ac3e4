"""Training-step throughput of any factory net on synthetic inputs (exploration; bench.py is the contract).

    python scripts/bench_net.py NAME B [steps] [--gpus N]

B = images PER GPU (BASELINE.json configs 3-5 are quoted per GPU: 128 / 128 / 256).  `--gpus N` without a torch.distributed
environment starts N ranks through torch.distributed.run (one process per GPU, RCCL; before anything here touches the GPU), each
rank a DataParallel replica on its own shard of the N * B batch; rank 0 prints the line, which then carries an `allreduce` block:
the gradient buckets of the net's backward segments (nets/graph.py backward_stages / grad_buckets), each bucket's all-reduce timed
alone with its bus bandwidth, and the step time with the collective switched off (what it costs after overlap)."""
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def main():
    argv = sys.argv[1:]
    gpus = 1
    if '--gpus' in argv:
        i = argv.index('--gpus')
        gpus = int(argv[i + 1])
        argv = argv[:i] + argv[i + 2:]
    name, B = argv[0], int(argv[1])
    steps = int(argv[2]) if len(argv) > 2 else 10
    if gpus > 1 and 'RANK' not in os.environ:          # fresh children: this process never touches the GPU
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(gpus), '--master-addr', '127.0.0.1',
               '--master-port', str(_free_port()), os.path.abspath(__file__), name, str(B), str(steps), '--gpus', str(gpus)]
        env = dict(os.environ)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        env.setdefault('OMP_NUM_THREADS', '8')
        raise SystemExit(subprocess.run(cmd, env=env).returncode)

    import torch
    from tf_face_toolbox_amd import net_select, Singular, DataParallel_margin
    ncls = 10575
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    dist = None
    if world > 1:
        import torch.distributed as dist
        shared = os.environ.get('FTE_BENCH_SHARED_GPU') == '1'          # tests: several ranks on one GPU over gloo
        local = 0 if shared else int(os.environ.get('LOCAL_RANK', rank))
        torch.cuda.set_device(local)
        backend = 'gloo' if shared else 'nccl'
        dist.init_process_group(backend, device_id=None if shared else torch.device('cuda', local))
    g = torch.Generator().manual_seed(rank)
    x = (torch.rand(B, 112, 112, 3, generator=g) * 2 - 1).cuda()
    y = torch.randint(0, ncls, (B,), generator=g, dtype=torch.int32).cuda()
    net = net_select(name, 'NCHW', 5e-4)
    inputs = {'images': x, 'labels': y, 'num_classes': ncls, 'num_examples': B * world, 'batch_size': B * world}
    model = DataParallel_margin(net, 1e-3, 'Momentum', num_gpus=world) if world > 1 else Singular(net, 1e-3, 'Momentum')
    step, losses, names, _ = model(inputs)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(k):
        for _ in range(3):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(k):
            step()
        barrier()
        dt = torch.tensor([(time.perf_counter() - t0) / k], dtype=torch.float64, device='cuda')
        if dist is not None:
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        return float(dt.item())
    dt = timed(steps)
    line = '%s B=%d%s: %.2f ms/step, %.1f images/s, losses %s, mem %.1f GB' % (
        name, B, ' x %d GPUs' % world if world > 1 else '', dt * 1e3, B * world / dt, [round(float(l), 4) for l in losses],
        torch.cuda.max_memory_allocated() / 1e9)
    if world > 1:
        buckets = net.grad_buckets()
        alone = []
        for a, b in buckets:
            buf = torch.zeros(b - a, dtype=torch.float32, device='cuda')
            for _ in range(2):
                dist.all_reduce(buf)
            barrier()
            t1 = time.perf_counter()
            for _ in range(5):
                dist.all_reduce(buf)
            barrier()
            alone.append(1000.0 * (time.perf_counter() - t1) / 5)

        class _NoComm(object):                   # same wrapper, collective switched off: the step's compute-only time
            def __init__(self, c): self.c = c
            def world_size(self): return self.c.world_size()
            def rank(self): return self.c.rank()
            def broadcast(self, t, src=0): pass
            def all_gather(self, out, t): return self.c.all_gather(out, t)

            def all_reduce_async(self, t):
                class _W(object):
                    def wait(self): return True
                return _W()
        real = model.comm
        model.comm = _NoComm(real)
        nocomm = timed(max(2, steps // 2))
        model.comm = real
        line += '\n  allreduce: backend %s, %d buckets of %s MB in completion order, alone %s ms (bus %s GB/s), step without the collective %.2f ms' % (
            dist.get_backend(), len(buckets), [round(4 * (b - a) / 1e6, 1) for a, b in buckets], [round(v, 3) for v in alone],
            [round(4 * (b - a) * 2 * (world - 1) / world / (v * 1e-3) / 1e9, 1) for (a, b), v in zip(buckets, alone)], nocomm * 1e3)
    if rank == 0:
        print(line)
    if dist is not None:
        barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
