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
