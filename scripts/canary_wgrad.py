"""Out-of-bounds write check for the filter-gradient entry points (debug aid; run on the GPU box)."""
import sys
import numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from util_gpu import ws, stream
from tf_face_toolbox_amd import _lib
_lib.load()
for (n, h, w, cin, cout, k, s) in [(8, 16, 16, 64, 128, 1, 1), (8, 8, 8, 128, 128, 1, 1), (8, 16, 16, 64, 64, 1, 1), (8, 16, 16, 64, 64, 3, 1),
                                    (8, 16, 16, 32, 128, 1, 1), (2, 8, 8, 64, 128, 1, 1), (64, 16, 16, 64, 128, 1, 1)]:
    x = torch.randn(n, h, w, cin, device='cuda'); oh = (h + s - 1) // s
    dy = torch.randn(n, oh, oh, cout, device='cuda')
    size = k * k * cin * cout
    buf = torch.full((size + 65536,), 3.0, device='cuda')
    wsb, nb = ws(_lib.query('fte_conv2d_wgrad_ws_bytes', n, h, w, cin, cout, k, s))
    _lib.call('fte_conv2d_wgrad', x, dy, buf, n, h, w, cin, cout, k, s, wsb, nb, stream())
    torch.cuda.synchronize()
    tail = buf[size:]
    nbad = int((tail != 3.0).sum())
    print((n, h, w, cin, cout, k, s), 'overflow floats:', nbad, 'first', int((tail != 3.0).nonzero()[0]) if nbad else None)
