"""Time of the tile-transform kernel alone (through fte_conv3x3_fwd_keep's launch records is not possible: it is not an MFMA launch) --
timed as the difference whole call - MFMA kernel would be noisy, so: rocprofv3 --kernel-trace --stats over this script."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from tf_face_toolbox_amd import _lib
call, query = _lib.call, _lib.query
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
st = torch.cuda.current_stream().cuda_stream
call('fte_set_conv_algo', 1)
for hw, c in ((28, 128), (14, 256), (7, 512)):
    x = torch.randn(N, hw, hw, c, device='cuda'); w = torch.randn(3, 3, c, c, device='cuda') * 0.05; y = torch.empty_like(x)
    need = query('fte_conv3x3_fwd_ws_bytes', N, hw, hw, c, c, 1)
    ws = torch.empty(need // 4 + 1024, device='cuda')
    for _ in range(10):
        call('fte_conv3x3_fwd', x, w, None, None, None, None, y, N, hw, hw, c, c, 1, ws, ws.numel() * 4, st)
    torch.cuda.synchronize()
