"""Helpers shared by the -m gpu parity tests: numpy <-> device, tolerance checks."""
import numpy as np
import torch

from tf_face_toolbox_amd import _lib

# Stated tolerances (fp32 HIP path vs float64 oracle), SURVEY.md 8c:
#   forward tensors  : max-abs-err <= 1e-4 * max|ref|
#   gradients        : relative L2  <= 1e-4 per tensor
TOL_MAXABS = 1e-4
TOL_RELL2 = 1e-4


def dev(a, dtype=torch.float32):
    return torch.tensor(np.ascontiguousarray(a), dtype=dtype, device='cuda')


def host(t):
    return t.detach().cpu().numpy().astype(np.float64)


def stream():
    return torch.cuda.current_stream().cuda_stream


def check_maxabs(got, ref, tol=TOL_MAXABS, what=''):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert np.isfinite(got).all(), what
    err = np.abs(got - ref).max() if got.size else 0.0
    scale = max(np.abs(ref).max() if ref.size else 0.0, 1e-30)
    assert err <= tol * scale, '%s: max-abs-err %.3e > %.1e * %.3e' % (what, err, tol, scale)
    return err / scale


def check_rell2(got, ref, tol=TOL_RELL2, what=''):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert np.isfinite(got).all(), what
    den = max(np.sqrt((ref * ref).sum()), 1e-30)
    err = np.sqrt(((got - ref) ** 2).sum()) / den
    assert err <= tol, '%s: rel-L2 %.3e > %.1e' % (what, err, tol)
    return err


def ws(nbytes):
    t = torch.empty(max(int(nbytes), 4096) // 4 + 1024, dtype=torch.float32, device='cuda')
    return t, t.numel() * 4


call = _lib.call
query = _lib.query
