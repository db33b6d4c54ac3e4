"""Per-call timing of one training step of a factory net (debug aid): every C-ABI call is bracketed by events and synchronised, one
stream (FTE_SIDE_STREAM=0 is forced), minimum over a few steps -> one line per call: entry point, integer arguments, algorithmic
bytes (each tensor argument once), microseconds, GB/s.   python scripts/dev/time_calls.py ResNeXt-50-center 128 [filter]"""
import os
import sys
os.environ['FTE_SIDE_STREAM'] = '0'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch                                                             # noqa: E402
from tf_face_toolbox_amd import net_select, Singular, _lib               # noqa: E402

name, B = sys.argv[1], int(sys.argv[2])
flt = sys.argv[3] if len(sys.argv) > 3 else ''
ncls = 10575
g = torch.Generator().manual_seed(0)
x = (torch.rand(B, 112, 112, 3, generator=g) * 2 - 1).cuda()
y = torch.randint(0, ncls, (B,), generator=g, dtype=torch.int32).cuda()
net = net_select(name, 'NCHW', 5e-4)
step, losses, names, _ = Singular(net, 1e-3, 'Momentum')({'images': x, 'labels': y, 'num_classes': ncls, 'num_examples': B})
for _ in range(3):
    step()
torch.cuda.synchronize()
real_call = _lib.call
scratch = {t.data_ptr() for t in (getattr(net, 'ws', None), getattr(net, 'ws_side', None)) if t is not None}
runs = []
cur = None


def timed_call(fn, *args):
    nbytes = sum(a.numel() * a.element_size() for a in args if isinstance(a, torch.Tensor) and a.data_ptr() not in scratch)
    ints = tuple(a for a in args if isinstance(a, int) and not isinstance(a, bool) and a < (1 << 24))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    r = real_call(fn, *args)
    e1.record()
    torch.cuda.synchronize()
    cur.append((fn, ints, nbytes, e0.elapsed_time(e1) * 1e3))
    return r


_lib.call = timed_call
for _ in range(4):
    cur = []
    step()
    runs.append(cur)
_lib.call = real_call
# one more step with the library's own launch records on: the conv / GEMM symbol(s) each call dispatched
syms = []


def sym_call(fn, *args):
    n0 = _lib.query('fte_prof_count')
    r = real_call(fn, *args)
    syms.append((n0, _lib.query('fte_prof_count')))
    return r


_lib.query('fte_prof_enable', 1)
_lib.call = sym_call
step()
torch.cuda.synchronize()
_lib.call = real_call
_lib.query('fte_prof_enable', 0)
recs = _lib.prof_records(shapes=True)
names = [' + '.join(sorted({recs[k][5].replace('_kernel', '') for k in range(a, b) if recs[k][5]})) for a, b in syms]
base = runs[0]
tot = 0.0
agg = {}
for i, (fn, ints, nb, _) in enumerate(base):
    us = min(r[i][3] for r in runs if len(r) == len(base))
    tot += us
    a = agg.setdefault(fn, [0, 0.0, 0])
    a[0] += 1; a[1] += us; a[2] += nb
    if flt in fn:
        print('%4d %-34s %-44s %8.1f MB %7.1f us %6.0f GB/s  %s' % (i, fn, ','.join(map(str, ints[:9])), nb / 1e6, us, nb / us / 1e3,
                                                                     names[i] if i < len(names) and len(names) == len(base) else ''))
print('--- per entry point (serialised, one stream): total %.2f ms' % (tot / 1e3))
for fn, (cnt, us, nb) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%-36s %4d calls %8.3f ms %9.1f MB %6.0f GB/s' % (fn, cnt, us / 1e3, nb / 1e6, nb / us / 1e3 if us else 0))
