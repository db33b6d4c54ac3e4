"""Host time to ENQUEUE one training step (no synchronisation inside) against the GPU time of the step: is the step launch-bound?
    python scripts/dev/host_time.py ResNeXt-50-center 128"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tf_face_toolbox_amd import net_select, Singular
name, B = sys.argv[1], int(sys.argv[2])
ncls = 10575
g = torch.Generator().manual_seed(0)
x = (torch.rand(B, 112, 112, 3, generator=g) * 2 - 1).cuda()
y = torch.randint(0, ncls, (B,), generator=g, dtype=torch.int32).cuda()
net = net_select(name, 'NCHW', 5e-4)
step, losses, names, _ = Singular(net, 1e-3, 'Momentum')({'images': x, 'labels': y, 'num_classes': ncls, 'num_examples': B})
for _ in range(5):
    step()
torch.cuda.synchronize()
host = []
for _ in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    host.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    step()
torch.cuda.synchronize()
gpu = (time.perf_counter() - t0) / 30
print('%s B=%d: host enqueue %.2f ms/step (min %.2f), back-to-back %.2f ms/step' % (name, B, 1e3 * sum(host) / len(host), 1e3 * min(host), 1e3 * gpu))

# what one enqueue costs: a C-ABI call that returns at its NULL check (ctypes + argument marshalling only), and the smallest real launch
from tf_face_toolbox_amd import _lib
lib = _lib.load()
a = [None] * 11 + [1e-5, 0.9] + [None] * 3 + [128, 56, 56, 64, 64, 1, 1, 1, None, 0, None]
t0 = time.perf_counter()
for _ in range(20000):
    lib.fte_conv2d_bn_fwd(*a)
t_null = (time.perf_counter() - t0) / 20000
d = torch.zeros(1024, device='cuda'); yv = torch.ones(1024, device='cuda'); gq = torch.empty(1024, device='cuda')
st = torch.cuda.current_stream().cuda_stream
for _ in range(100):
    _lib.call('fte_relu_bwd', d, yv, gq, 1024, st)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000):
    _lib.call('fte_relu_bwd', d, yv, gq, 1024, st)
t_launch = (time.perf_counter() - t0) / 2000
torch.cuda.synchronize()
print('per call: ctypes + marshalling of a 27-argument entry point %.2f us; _lib.call of a 5-argument entry point with one kernel launch %.2f us' % (1e6 * t_null, 1e6 * t_launch))
