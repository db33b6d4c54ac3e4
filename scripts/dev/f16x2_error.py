"""The same question as bf16x3_error.py for a TWO-term fp16 split (x = h1 + h2, 11 + 11 mantissa bits): 3 cross terms against fp32 and float64,
at three weight magnitudes (h2 of small values falls into fp16 subnormals).  numpy only.  python scripts/dev/f16x2_error.py"""
import numpy as np
rng = np.random.default_rng(0)
def split2(x):
    h1 = x.astype(np.float16); h2 = (x - h1.astype(np.float32)).astype(np.float16)
    return h1, h2
def mm(x, y): return (x.astype(np.float64) @ y.astype(np.float64)).astype(np.float32)
for wscale in (0.05, 0.005, 1.0):
    M, N, K = 256, 256, 2304
    A = rng.standard_normal((M, K)).astype(np.float32); B = (rng.standard_normal((K, N)) * wscale).astype(np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64); sc = np.abs(ref).max()
    f32 = A @ B
    a = split2(A); b = split2(B)
    t3 = ((mm(a[1], b[0]) + mm(a[0], b[1])).astype(np.float32) + mm(a[0], b[0])).astype(np.float32)
    t4 = ((mm(a[1], b[1]) + mm(a[1], b[0])).astype(np.float32) + mm(a[0], b[1]) + mm(a[0], b[0])).astype(np.float32)
    print('weights ~%g: fp32 %.2e   fp16x2 3 terms %.2e   4 terms %.2e   plain fp16 %.2e   (max-abs / max|ref|)' % (
        wscale, np.abs(f32 - ref).max() / sc, np.abs(t3 - ref).max() / sc, np.abs(t4 - ref).max() / sc, np.abs(mm(a[0], b[0]) - ref).max() / sc))
