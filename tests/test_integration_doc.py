"""CPU: every ctypes signature shown in INTEGRATION.md agrees with include/fte.h (a maintainer who copies the stub must
not end up passing the stream as the workspace), and the documented stub really binds against libfte.so."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_signatures():
    """name -> (return kind, [parameter kinds]) with kinds 'ptr' | 'int' | 'size_t' | 'long' | 'float' | 'double'."""
    src = open(os.path.join(ROOT, 'include', 'fte.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    src = re.sub(r'//[^\n]*', '', src)
    sigs = {}
    for ret, name, args in re.findall(r'\b(int|size_t|const char\s*\*|void)\s+(fte_[a-z0-9_]+)\s*\(([^)]*)\)\s*;', src):
        kinds = []
        for a in [x.strip() for x in args.split(',') if x.strip() and x.strip() != 'void']:
            if '*' in a:
                kinds.append('ptr')
            else:
                t = a.rsplit(None, 1)[0].replace('const', '').strip()
                kinds.append({'int': 'int', 'size_t': 'size_t', 'long': 'long', 'float': 'float', 'double': 'double',
                              'int32_t': 'int', 'uint32_t': 'int', 'int64_t': 'long', 'uint64_t': 'long', 'unsigned': 'int', 'unsigned long long': 'long'}[t])
        sigs[name] = ('ptr' if '*' in ret else ret.strip(), kinds)
    return sigs


def doc_bindings():
    """The python code blocks of INTEGRATION.md, executed against a recording fake of ctypes.CDLL."""
    md = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    blocks = re.findall(r'```python\n(.*?)```', md, flags=re.S)
    return [b for b in blocks if 'argtypes' in b]


class _Fn(object):
    restype = ctypes.c_int
    argtypes = None


class _FakeLib(object):
    def __init__(self):
        object.__setattr__(self, 'fns', {})

    def __getattr__(self, name):
        return self.fns.setdefault(name, _Fn())


KIND = {ctypes.c_void_p: 'ptr', ctypes.c_char_p: 'ptr', ctypes.c_int: 'int', ctypes.c_size_t: 'size_t', ctypes.c_long: 'long',
        ctypes.c_float: 'float', ctypes.c_double: 'double'}


def test_every_signature_in_the_doc_matches_the_header(monkeypatch):
    sigs = header_signatures()
    assert len(sigs) >= 70 and 'fte_conv3x3_fwd' in sigs
    assert sigs['fte_conv3x3_fwd'][1] == ['ptr'] * 7 + ['int'] * 6 + ['ptr', 'size_t', 'ptr']      # 16 arguments
    blocks = doc_bindings()
    assert blocks, 'INTEGRATION.md shows no ctypes binding'
    seen = 0
    for code in blocks:
        fake = _FakeLib()
        monkeypatch.setattr(ctypes, 'CDLL', lambda path, fake=fake: fake)
        exec(compile(code, 'INTEGRATION.md', 'exec'), {'__name__': 'doc'})
        for name, fn in fake.fns.items():
            assert name in sigs, '%s is not declared in include/fte.h' % name
            if fn.argtypes is None:
                continue
            got = [KIND[t] for t in fn.argtypes]
            assert got == sigs[name][1], (name, got, sigs[name][1])
            want_ret = sigs[name][0]
            assert KIND.get(fn.restype, 'ptr') == want_ret, (name, fn.restype, want_ret)
            seen += 1
    assert seen >= 2


def test_package_binding_matches_the_header_too():
    """tf_face_toolbox_amd/_lib.py is the full table: the same check for every one of its entries."""
    from tf_face_toolbox_amd import _lib
    sigs = header_signatures()
    w64 = lambda k: 'i64' if k in ('size_t', 'long') else k      # c_size_t IS c_uint64 on this ABI: one 64-bit integer class
    assert sorted(_lib._SIGS) == sorted(sigs)
    for name, (ret, args) in _lib._SIGS.items():
        got = [w64(KIND[t]) for t in args]
        assert got == [w64(k) for k in sigs[name][1]], (name, got, sigs[name][1])
        assert w64(KIND[ret]) == w64(sigs[name][0]), (name, ret, sigs[name][0])
