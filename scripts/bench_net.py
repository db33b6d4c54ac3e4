"""Training-step throughput of any factory net on synthetic inputs (exploration; bench.py is the contract)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf_face_toolbox_amd import net_select, Singular
name = sys.argv[1]; B = int(sys.argv[2]); steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
ncls = 10575
g = torch.Generator().manual_seed(0)
x = (torch.rand(B, 112, 112, 3, generator=g) * 2 - 1).cuda(); y = torch.randint(0, ncls, (B,), generator=g, dtype=torch.int32).cuda()
net = net_select(name, 'NCHW', 5e-4)
step, losses, names, _ = Singular(net, 1e-3, 'Momentum')({'images': x, 'labels': y, 'num_classes': ncls, 'num_examples': B})
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
print('%s B=%d: %.2f ms/step, %.1f images/s, losses %s, mem %.1f GB' % (name, B, dt * 1e3, B / dt, [round(float(l), 4) for l in losses], torch.cuda.max_memory_allocated() / 1e9))
