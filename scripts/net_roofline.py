"""Step-level HBM roofline of a factory net ("report algorithmic_bytes / time / HBM_peak"): every C-ABI call of one
training step is logged with its ALGORITHMIC bytes -- each tensor argument once (operands, results, per-channel vectors; the
scratch workspace excluded) -- and summed per entry point; the step is then timed without the logging.

    python scripts/net_roofline.py ShuffleNet-v2-small 256 [steps]       (config 5's per-GPU batch: 2048 images on 8 GPUs)

Prints a markdown table (per entry point: calls per step, algorithmic MB per step) and the step-level line
`algorithmic bytes / step time / 8 TB/s`.  MFMA-bound entry points (the igemm family: fte_conv2d_*, fte_gemm_*) are listed
with their bytes but are not HBM-bound; the line is given with and without them."""
import os
import sys
import time
from collections import OrderedDict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                             # noqa: E402
from tf_face_toolbox_amd import net_select, Singular, _lib               # noqa: E402

name = sys.argv[1]
B = int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
ncls = 10575
g = torch.Generator().manual_seed(0)
x = (torch.rand(B, 112, 112, 3, generator=g) * 2 - 1).cuda()
y = torch.randint(0, ncls, (B,), generator=g, dtype=torch.int32).cuda()
net = net_select(name, 'NCHW', 5e-4)
step, losses, names, _ = Singular(net, 1e-3, 'Momentum')({'images': x, 'labels': y, 'num_classes': ncls, 'num_examples': B})
for _ in range(3):
    step()
torch.cuda.synchronize()

MFMA = ('fte_conv2d_', 'fte_conv3x3_fwd', 'fte_conv3x3_dgrad', 'fte_conv3x3_wgrad', 'fte_gemm_')
log = OrderedDict()
real_call = _lib.call
scratch = {t.data_ptr() for t in (getattr(net, 'ws', None), getattr(net, 'ws_side', None)) if t is not None}


def logging_call(fn, *args):
    nbytes = 0
    for a in args:
        if isinstance(a, torch.Tensor) and a.data_ptr() not in scratch:
            nbytes += a.numel() * a.element_size()
    e = log.setdefault(fn, [0, 0])
    e[0] += 1
    e[1] += nbytes
    return real_call(fn, *args)


# every module took `call = _lib.call` at import or per function: patch the attribute they read
_lib.call = logging_call
import tf_face_toolbox_amd.nets.graph as graph_mod                        # noqa: E402
import tf_face_toolbox_amd.nets.sphere as sphere_mod                      # noqa: E402
import tf_face_toolbox_amd.data_parallel as dp_mod                        # noqa: E402
step()
torch.cuda.synchronize()
_lib.call = real_call
if not log:
    raise SystemExit('no calls were logged (the step cached its entry points?)')

torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3

tot = sum(v[1] for v in log.values())
hbm = sum(v[1] for k, v in log.items() if not k.startswith(MFMA))
print('| entry point | bound | calls / step | algorithmic MB / step |')
print('|---|---|---|---|')
for k, v in sorted(log.items(), key=lambda kv: -kv[1][1]):
    print('| `%s` | %s | %d | %.1f |' % (k, 'MFMA' if k.startswith(MFMA) else 'HBM', v[0], v[1] / 1e6))
print()
print('%s, batch %d: %.3f ms per step = %.1f images/s; %d C-ABI calls per step' % (name, B, ms, B / (ms * 1e-3), sum(v[0] for v in log.values())))
print('step-level HBM roofline: algorithmic %.2f GB per step / %.3f ms = %.0f GB/s = %.1f %% of 8 TB/s  (HBM-bound entry points alone: %.2f GB -> %.1f %% if they had the step to themselves)'
      % (tot / 1e9, ms, tot / ms / 1e6, 100 * tot / ms / 1e6 / 8000, hbm / 1e9, 100 * hbm / ms / 1e6 / 8000))
