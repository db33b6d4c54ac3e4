"""CPU: libfte.so builds, loads, and exports every symbol include/fte.h declares (no compute calls)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib_path():
    path = os.path.join(ROOT, 'tf_face_toolbox_amd', 'libfte.so')
    if not os.path.exists(path):
        subprocess.check_call(['bash', os.path.join(ROOT, 'tf_face_toolbox_amd', 'csrc', 'build.sh')])
    return path


def declared_functions():
    src = open(os.path.join(ROOT, 'include', 'fte.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(fte_[a-z0-9_]+)\s*\(', src)))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ('fte_conv3x3_fwd', 'fte_conv3x3_dgrad', 'fte_conv3x3_wgrad', 'fte_gemm_nn', 'fte_gemm_nt', 'fte_gemm_tn',
                 'fte_softmax_ce_fwd_bwd', 'fte_asoftmax_fwd_bwd', 'fte_center_loss_fwd_bwd_update',
                 'fte_batch_hard_triplet_fwd_bwd', 'fte_momentum_update', 'fte_adam_update'):
        assert must in names


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    for name in declared_functions():
        assert hasattr(lib, name), 'libfte.so does not export %s' % name


def test_ctypes_binding_covers_the_header(lib_path):
    from tf_face_toolbox_amd import _lib
    assert sorted(_lib.exported_names()) == declared_functions()
    _lib.load()
    assert _lib.version().startswith('fte ')
    assert 'gfx950' in _lib.version()


def test_no_cuda_compat_or_oracle_in_the_product():
    """The package must not import the oracle, and the kernels must be plain gfx950 HIP."""
    pkg = os.path.join(ROOT, 'tf_face_toolbox_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.sh')):
                txt = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in txt and 'from oracle' not in txt, f
                assert '__HIP_PLATFORM_AMD__' not in txt and 'cuda_runtime' not in txt and 'triton' not in txt.lower(), f


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from tf_face_toolbox_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(_lib.FteError):
        _lib.load()
