"""Helpers shared by the -m gpu parity tests: numpy <-> device, tolerance checks."""
import numpy as np
import torch

from tf_face_toolbox_amd import _lib

# Stated tolerances (fp32 HIP path vs float64 oracle).  SURVEY.md 8c proposed 1e-4; measured on
# MI355X the path sits at fp32's own noise floor (4e-7 .. 2e-6 relative, the same as the oracle
# evaluated in float32), so the tests hold it 5-10x tighter than proposed:
#   forward tensors  : max-abs-err <= 2e-5 * max|ref|
#   gradients        : relative L2  <= 2e-5 per tensor
# PReLU's derivative is discontinuous at z = 0: for |z| below fp32 resolution the float32 and
# float64 evaluations may legitimately sit on different sides; whole-net tests pass the HIP
# path's z to the oracle, which adopts its side inside the kink band only (oracle.spherenet.kink_resolved).
TOL_MAXABS = 2e-5
TOL_RELL2 = 2e-5
# Mixed-precision checks ('bf16' / 'bf16s' modes): the band inside which the oracle adopts the implementation's side of a ReLU /
# PReLU kink is BF16_KINK_MULT x the ORACLE's OWN bf16 noise on that tensor -- rms(oracle with bf16-rounded operands - exact oracle)
# on the same input (oracle.spherenet.bf16_noise, oracle.graphnet.noise_bands16; per operand the unit roundoff is 2^-9, per layer
# ~1.6e-3 * rms(z), compounding with depth) -- never a function of the tensors under test.
BF16_KINK_MULT = 4


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


def kink_of(net):
    """name -> z of every conv layer as the HIP path computed it (for oracle kink resolution)."""
    return {c.name: host(net.z[i]) for i, c in enumerate(net.convs)}


call = _lib.call
query = _lib.query
