"""
Index model of the row-local 4096-point workgroup transform (fft_core.hpp, WgFftL): 256 lanes x 16 registers.

forward : input  lane p reg m = x[js(p) + 256 m], js(p) = (p >> 4) + 16 (p & 15)      ("space layout")
          output lane p reg m = X[p + 256 m]                                            ("frequency layout")
inverse : the exact mirror (decimation in frequency with post-twiddles): frequency layout -> space layout,
          same per-lane twiddle registers (conjugated).
The first exchange of the forward transform (last of the inverse) stays inside rows of 16 lanes (no barrier);
the other one is the Stockham exchange of the general code.  LDS position of every access is asserted to
lie in the row region [272 row, 272 row + 272) of the WRITING wave for forward writes / reading wave for
inverse reads (the hazard argument in fft_core.hpp relies on it).
"""
import numpy as np

N, T = 4096, 256
W = lambda e, n=4096, s=-1: np.exp(s * 2j * np.pi * (e % n) / n)


def dft4(v, s):
    a0, a1, a2, d = v[0] + v[2], v[0] - v[2], v[1] + v[3], v[1] - v[3]
    rot = (-1j if s < 0 else 1j) * d
    return [a0 + a2, a1 + rot, a0 - a2, a1 - rot]


def dft16_pre(v, s, tw4=None, tw1=None):
    """existing Dft<16>::run_tw: inputs pre-multiplied by tw4[r2] (r = r1 + 4 r2), inner factor tw1[r1]"""
    v = list(v)
    if tw4 is not None:
        for r2 in range(1, 4):
            for r1 in range(4):
                v[r1 + 4 * r2] *= tw4[r2 - 1]
    t = [[None] * 4 for _ in range(4)]
    for r1 in range(4):
        o = dft4([v[r1], v[r1 + 4], v[r1 + 8], v[r1 + 12]], s)
        for p2 in range(4):
            t[r1][p2] = o[p2] * (tw1[r1 - 1] if (tw1 is not None and r1 > 0) else 1) * np.exp(s * 2j * np.pi * r1 * p2 / 16)
    out = [None] * 16
    for p2 in range(4):
        o = dft4([t[0][p2], t[1][p2], t[2][p2], t[3][p2]], s)
        for p1 in range(4):
            out[4 * p1 + p2] = o[p1]
    return out


def dft16_post(v, s, tw4=None, tw1=None):
    """mirror (DIF): r = 4 r1 + r2, p = p1 + 4 p2; outputs multiplied by tw1[p1] * tw4[p2]"""
    u = [[None] * 4 for _ in range(4)]
    for r2 in range(4):
        o = dft4([v[r2], v[4 + r2], v[8 + r2], v[12 + r2]], s)
        for p1 in range(4):
            u[p1][r2] = o[p1] * np.exp(s * 2j * np.pi * r2 * p1 / 16) * (tw1[p1 - 1] if (tw1 is not None and p1 > 0) else 1)
    out = [None] * 16
    for p1 in range(4):
        o = dft4(u[p1], s)
        for p2 in range(4):
            out[p1 + 4 * p2] = o[p2] * (tw4[p2 - 1] if (tw4 is not None and p2 > 0) else 1)
    return out


def js(p):
    return (p >> 4) + 16 * (p & 15)


def twiddles(p, s):
    # stage 1: k = p & 15, W_256; stage 2: k = p, W_4096   (split form: tw4[q-1] = W^(4 q k), tw1[q-1] = W^(q k))
    k1, k2 = p & 15, p
    t1 = ([W(16 * 4 * q * k1, s=s) for q in (1, 2, 3)], [W(16 * q * k1, s=s) for q in (1, 2, 3)])
    t2 = ([W(4 * q * k2, s=s) for q in (1, 2, 3)], [W(q * k2, s=s) for q in (1, 2, 3)])
    return t1, t2


def forward(x):
    s = -1
    reg = np.array([[x[js(p) + 256 * m] for m in range(16)] for p in range(T)], dtype=np.complex128)
    lds = np.full(16 * 272, np.nan, dtype=np.complex128)
    # stage 0 + local exchange
    for p in range(T):
        reg[p] = dft16_pre(reg[p], s)
        row, c = p >> 4, p & 15
        for i in range(16):
            e = 272 * row + 17 * c + i
            assert 272 * row <= e < 272 * row + 272
            lds[e] = reg[p][i]
    for p in range(T):
        row, c = p >> 4, p & 15
        reg[p] = [lds[272 * row + 17 * m + c] for m in range(16)]
    # stage 1 + global exchange
    for p in range(T):
        t1, _ = twiddles(p, s)
        reg[p] = dft16_pre(reg[p], s, *t1)
        row, c = p >> 4, p & 15
        for r in range(16):
            e = 272 * row + 16 * r + c
            assert 272 * row <= e < 272 * row + 272        # forward global writes stay in the writer's row region
            lds[e] = reg[p][r]
    for p in range(T):
        reg[p] = [lds[272 * m + p] for m in range(16)]
    for p in range(T):
        _, t2 = twiddles(p, s)
        reg[p] = dft16_pre(reg[p], s, *t2)
    X = np.zeros(N, dtype=np.complex128)
    for p in range(T):
        for m in range(16):
            X[p + 256 * m] = reg[p][m]
    return X


def inverse(X):
    s = +1
    reg = np.array([[X[p + 256 * m] for m in range(16)] for p in range(T)], dtype=np.complex128)
    lds = np.full(16 * 272, np.nan, dtype=np.complex128)
    for p in range(T):
        _, t2 = twiddles(p, s)
        reg[p] = dft16_post(reg[p], s, *t2)
        for m in range(16):
            lds[272 * m + p] = reg[p][m]           # global: writes into every row region
    for p in range(T):
        row, c = p >> 4, p & 15
        reg[p] = [lds[272 * row + 16 * r + c] for r in range(16)]     # ... reads from the reader's own region
    for p in range(T):
        t1, _ = twiddles(p, s)
        reg[p] = dft16_post(reg[p], s, *t1)
        row, c = p >> 4, p & 15
        for m in range(16):
            lds[272 * row + 17 * m + c] = reg[p][m]               # local
    for p in range(T):
        row, c = p >> 4, p & 15
        reg[p] = dft16_post([lds[272 * row + 17 * c + i] for i in range(16)], s)
    x = np.zeros(N, dtype=np.complex128)
    for p in range(T):
        for m in range(16):
            x[js(p) + 256 * m] = reg[p][m]
    return x


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    x = rng.normal(size=N) + 1j * rng.normal(size=N)
    X = forward(x)
    print("forward  max err", np.abs(X - np.fft.fft(x)).max())
    xi = inverse(X)
    print("inverse  max err", np.abs(xi - N * x).max())
    # bank check (8-byte elements, 32 element-banks): every 16-lane group hits 16 distinct banks, pairs of rows 32
    for name, f in (("local write", lambda p, i: 272 * (p >> 4) + 17 * (p & 15) + i),
                    ("local read", lambda p, i: 272 * (p >> 4) + 17 * i + (p & 15)),
                    ("global scatter", lambda p, i: 272 * (p >> 4) + 16 * i + (p & 15)),
                    ("global gather", lambda p, i: 272 * i + p)):
        worst = 0
        for i in range(16):
            for g in range(0, 256, 32):
                banks = [f(p, i) % 32 for p in range(g, g + 32)]
                worst = max(worst, 32 - len(set(banks)))
        print(name, "bank collisions per 32-lane group:", worst)
