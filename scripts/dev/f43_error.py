"""fp32 error of Winograd F(4x4,3x3) against F(2x2,3x3) and the direct sum at SphereNet's resBlock shapes (numpy, arithmetic held in float32,
reference = float64 direct convolution): is the 4x-fewer-multiplies variant inside the 2e-5 parity bound?  python scripts/dev/f43_error.py"""
import numpy as np
rng = np.random.default_rng(0)
def direct64(x, w):
    n, h, wd, c = x.shape; co = w.shape[3]
    xp = np.pad(x, ((0,0),(1,1),(1,1),(0,0)))
    y = np.zeros((n, h, wd, co))
    for a in range(3):
        for b in range(3):
            y += np.einsum('nhwc,co->nhwo', xp[:, a:a+h, b:b+wd, :], w[a, b])
    return y
def wino(x, w, m, dtype):
    # F(m x m, 3 x 3) with the standard points; all arithmetic in `dtype`
    if m == 2:
        BT = np.array([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]], dtype)
        G = np.array([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], dtype)
        AT = np.array([[1,1,1,0],[0,1,-1,-1]], dtype)
    else:
        BT = np.array([[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]], dtype)
        G = np.array([[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]], dtype)
        AT = np.array([[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]], dtype)
    a = m + 2
    n, h, wd, c = x.shape; co = w.shape[3]
    th, tw = -(-h // m), -(-wd // m)
    xp = np.zeros((n, th*m+2, tw*m+2, c), dtype); xp[:, 1:h+1, 1:wd+1] = x
    U = np.einsum('ia,abco,jb->ijco', G, w.astype(dtype), G).astype(dtype)
    y = np.zeros((n, th*m, tw*m, co), dtype)
    for ty in range(th):
        for tx in range(tw):
            d = xp[:, ty*m:ty*m+a, tx*m:tx*m+a, :]
            V = np.einsum('ia,nabc,jb->nijc', BT, d, BT).astype(dtype)
            M = np.einsum('nijc,ijco->nijo', V, U).astype(dtype)      # (numpy accumulates in dtype)
            y[:, ty*m:(ty+1)*m, tx*m:(tx+1)*m, :] = np.einsum('ia,nabo,jb->nijo', AT, M, AT).astype(dtype)
    return y[:, :h, :wd]
for (n, h, c) in ((4, 14, 256), (2, 28, 128), (8, 7, 512)):
    x = rng.standard_normal((n, h, h, c)); w = rng.standard_normal((3, 3, c, c)) * 0.05
    ref = direct64(x, w); sc = np.abs(ref).max()
    xp32 = x.astype(np.float32)
    d32 = np.zeros_like(ref, dtype=np.float32)
    xpad = np.pad(xp32, ((0,0),(1,1),(1,1),(0,0)))
    for a in range(3):
        for b in range(3):
            d32 += np.einsum('nhwc,co->nhwo', xpad[:, a:a+h, b:b+h, :], w[a, b].astype(np.float32))
    print('%dx%d c%d: direct fp32 %.2e   F(2,3) fp32 %.2e   F(4,3) fp32 %.2e   (max-abs / max|ref|)' % (h, h, c,
          np.abs(d32 - ref).max() / sc, np.abs(wino(xp32, w, 2, np.float32) - ref).max() / sc, np.abs(wino(xp32, w, 4, np.float32) - ref).max() / sc))
