"""tf_face_toolbox_amd -- MI355X-native engine for the data-parallel training step of
medivhna/TF_Face_Toolbox (data_parallel.py + nets/ + loss.py).  Host code mirrors the
reference's Python interface; every FLOP runs in libfte.so (hand-written HIP, gfx950).

The public names are resolved on first use (PEP 562): the input pipeline's decode workers import
`tf_face_toolbox_amd._decode_worker` in fresh interpreters and must not drag torch in with the package."""
__version__ = '0.1'

_LAZY = {'net_select': '.nets.net_base', 'Network': '.nets.net_base',
         'Singular': '.data_parallel', 'DataParallel': '.data_parallel', 'DataParallel_margin': '.data_parallel'}


def __getattr__(name):
    mod = _LAZY.get(name)
    if mod is None:
        raise AttributeError('module %r has no attribute %r' % (__name__, name))
    import importlib
    value = getattr(importlib.import_module(mod, __name__), name)
    globals()[name] = value
    return value


def __dir__():
    return sorted(list(globals()) + list(_LAZY))
