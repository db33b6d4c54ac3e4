"""Worker of tests/test_gpu_dp_two_ranks.py: one rank of a 2-process DataParallel run.  On a box with a GPU per rank this is
the production configuration: device = LOCAL_RANK, backend 'nccl' (= RCCL over xGMI).  On the one-GPU test box both ranks
share device 0 and talk over gloo (RCCL refuses two ranks on one device); everything else is the production path either way:
real HIP kernels, per-rank shard, bucketed async all-reduce of the gradient arena, broadcast of rank 0's parameters.
FTE_TEST_FORCE_GLOO=1 keeps the gloo transport on a multi-GPU box.  The result file records which backend ran."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tf_face_toolbox_amd import net_select, DataParallel, DataParallel_margin   # noqa: E402


def main():
    fix, out, name, steps = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    local = int(os.environ.get('LOCAL_RANK', rank))
    rccl = torch.cuda.device_count() >= world and os.environ.get('FTE_TEST_FORCE_GLOO') != '1'      # device_count() does not initialise the GPU
    if rccl:
        torch.cuda.set_device(local)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    else:
        torch.cuda.set_device(0)
        dist.init_process_group('gloo')
    d = np.load(fix)
    x, y = d['x'], d['y']
    n, h, w, ch = x.shape
    ncls = int(d['ncls'])
    sh = n // world
    xs = torch.tensor(x[rank * sh:(rank + 1) * sh], dtype=torch.float32, device='cuda')
    ys = torch.tensor(y[rank * sh:(rank + 1) * sh], dtype=torch.int32, device='cuda')
    net = net_select(name, 'NCHW', 5e-4)
    net.seed = 100 + rank                       # replicas start DIFFERENT: the wrapper's broadcast must make them equal
    net.build(h, w, ch, ncls, 'cuda')
    if rank == 0:
        net.load_params({k[2:]: d[k] for k in d.files if k.startswith('p:')})
    wrapper = DataParallel_margin if net.needs_labels else DataParallel
    model = wrapper(net, 0.05, 'Momentum', num_gpus=world, sync_centers=os.environ.get('FTE_TEST_SYNC_CENTERS') == '1')
    step, losses, names, _ = model({'images': xs, 'labels': ys, 'num_classes': ncls, 'num_examples': n, 'batch_size': n})
    hist = []
    for _ in range(steps):
        step()
        hist.append([float(v) for v in losses])
    torch.cuda.synchronize()
    res = {'w:' + k: net.get_variable(k).cpu().numpy() for k in net.variables}
    for k in getattr(net, 'state', None) or {}:          # non-trainable per-replica state: BN moving statistics, centers
        res['s:' + k] = net.get_variable(k).cpu().numpy()
    res['losses'] = np.array(hist)
    res['backend'] = np.array('nccl' if rccl else 'gloo')
    res['device'] = np.array(torch.cuda.current_device())
    np.savez(out + '.rank%d.npz' % rank, **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
    sys.stdout.flush()
    os._exit(0)
