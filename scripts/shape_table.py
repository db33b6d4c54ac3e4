"""Per-shape MFMA launch table of any factory net's training step (fte_prof_* event pairs around every gathered-GEMM launch):
    python scripts/shape_table.py NAME BATCH
Prints rows x N x K, tile, splits, launches per step, ms, TFLOP/s -- where a net's GEMM time goes (bench.py does the same for
the headline net in `roofline.per_shape`)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf_face_toolbox_amd import net_select, Singular, _lib
name = sys.argv[1]; B = int(sys.argv[2])
ncls = 10575
g = torch.Generator().manual_seed(0)
x = (torch.rand(B, 112, 112, 3, generator=g) * 2 - 1).cuda(); y = torch.randint(0, ncls, (B,), generator=g, dtype=torch.int32).cuda()
net = net_select(name, 'NCHW', 5e-4)
step, losses, names, _ = Singular(net, 1e-3, 'Momentum')({'images': x, 'labels': y, 'num_classes': ncls, 'num_examples': B})
for _ in range(3): step()
torch.cuda.synchronize()
_lib.query('fte_prof_enable', 1)
steps = 3
for _ in range(steps): step()
torch.cuda.synchronize()
_lib.query('fte_prof_enable', 0)
agg = collections.OrderedDict()
for sig, fl, ms, mnk, by, _sym in _lib.prof_records(shapes=True):
    k = (tuple(sig), tuple(mnk))
    a = agg.setdefault(k, [0, 0.0, 0.0]); a[0] += 1; a[1] += fl; a[2] += ms
rows = sorted(agg.items(), key=lambda kv: -kv[1][2])
tot = sum(v[2] for _, v in rows) / steps
print('%s B=%d: %.3f ms of MFMA launches per step' % (name, B, tot))
for (sig, mnk), (n, fl, ms) in rows[:40]:
    print('sig %-28s rows %8d N %5d K %6d | x%5.1f/step %8.3f ms/step %6.1f us each %6.1f TF' % (sig, mnk[0], mnk[1], mnk[2], n / steps, ms / steps, ms / n * 1e3, fl / ms / 1e9))
