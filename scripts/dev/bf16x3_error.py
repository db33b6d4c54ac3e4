"""Accuracy of fp32 products emulated on the bf16 matrix cores by splitting each fp32 operand into three bf16 terms (x = x1 + x2 + x3):
3 cross terms (x1 y1 + x1 y2 + x2 y1) and 6 cross terms (i + j <= 4) against a plain fp32 product and the float64 reference.
numpy only; each bf16 x bf16 partial GEMM is formed exactly and rounded to fp32 once (the MFMA accumulates in fp32).  python scripts/dev/bf16x3_error.py"""
import numpy as np
rng = np.random.default_rng(0)
def bf16(x):
    # round-to-nearest-even to bf16, returned as float32
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)
def split3(x):
    a1 = bf16(x); r = (x - a1).astype(np.float32); a2 = bf16(r); a3 = bf16((r - a2).astype(np.float32))
    return a1, a2, a3
M, N, K = 256, 256, 2304
A = rng.standard_normal((M, K)).astype(np.float32); B = (rng.standard_normal((K, N)) * 0.05).astype(np.float32)
ref = A.astype(np.float64) @ B.astype(np.float64); sc = np.abs(ref).max()
f32 = A @ B     # fp32 (numpy/BLAS: fp32 accumulate)
a = split3(A); b = split3(B)
def mm(x, y): return (x.astype(np.float64) @ y.astype(np.float64)).astype(np.float32)   # exact bf16 products, fp32 result per term (MFMA accumulates in fp32: emulate per-term rounding only)
terms6 = [(0,0),(0,1),(1,0),(0,2),(1,1),(2,0)]
terms3 = [(0,0),(0,1),(1,0)]
def emu(terms):
    acc = np.zeros((M, N), np.float32)
    for i, j in terms[::-1]:          # small terms first
        acc = (acc + mm(a[i], b[j])).astype(np.float32)
    return acc
print('max-abs / max|ref|:  fp32 %.2e   bf16x3 (3 terms) %.2e   bf16x3 (6 terms) %.2e   plain bf16 %.2e' % (
    np.abs(f32 - ref).max() / sc, np.abs(emu(terms3) - ref).max() / sc, np.abs(emu(terms6) - ref).max() / sc, np.abs(mm(a[0], b[0]) - ref).max() / sc))
