"""One training step captured into a HIP graph (torch.cuda.CUDAGraph) against the eager step: host time to launch it and GPU time per
step, and whether the replayed step leaves the same weights as the eager one (record once, replay).
    python scripts/dev/graph_capture.py ResNeXt-50-center 128"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tf_face_toolbox_amd import net_select, Singular
name, B = sys.argv[1], int(sys.argv[2])
ncls = 10575
g = torch.Generator().manual_seed(0)
hw = (112, 96) if name.startswith('SphereNet') else (112, 112)
x = (torch.rand(B, hw[0], hw[1], 3, generator=g) * 2 - 1).cuda()
y = torch.randint(0, ncls, (B,), generator=g, dtype=torch.int32).cuda()


def make():
    net = net_select(name, 'NCHW', 5e-4)
    step, losses, names, _ = Singular(net, 1e-3, 'Momentum')({'images': x, 'labels': y, 'num_classes': ncls, 'num_examples': B})
    return net, step


def timed(fn, n=30):
    host = []
    for _ in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        host.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * min(host), 1e3 * (time.perf_counter() - t0) / n


net, step = make()
for _ in range(5):
    step()
torch.cuda.synchronize()
h, gms = timed(step)
print('%s B=%d eager: host enqueue %.2f ms, back-to-back %.2f ms/step' % (name, B, h, gms), flush=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(gr, stream=s):
        step()
except Exception as e:
    print('capture failed: %r' % (e,))
    sys.exit(0)
torch.cuda.synchronize()
h, gms = timed(gr.replay)
print('%s B=%d HIP graph: host launch %.2f ms, back-to-back %.2f ms/step' % (name, B, h, gms), flush=True)
