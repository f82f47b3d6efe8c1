"""
Index model of the 8192-point workgroup transform built on the row-local 4096-point one (fft_core.hpp, WgFftL8k):
512 lanes x 16 registers; lane j = 2 p + h runs, as lane p, the 4096-point transform number h (WgFftL, its own LDS
image), the two halves interleaved lane by lane.

  8192-point DIF split:  y0[n] = x[n] + x[n + 4096],  y1[n] = (x[n] - x[n + 4096]) W_8192^n,  X[2k + h] = FFT_4096(y_h)[k]

forward : input  lane j reg r = x[s(j) + 512 r], s(j) = js(j >> 1) + 256 (j & 1)          ("space layout", T = 512)
          radix-2 in registers (pairs r, r + 8; twiddle W_8192^(s(j)) W_16^r), then ONE wave-local exchange between
          the lanes 2p / 2p + 1 (X1): lane (p, h) collects y_h[js(p) + 256 n2], n2 = 0..15 -- the space layout of
          WgFftL -- and the two 4096-point transforms run side by side.
          output lane j reg m = X[j + 512 m]                                                ("frequency layout")
inverse : the mirror (two mirror flows, X1 backwards, radix-2 with conjugated twiddles).

The two images are interleaved element by element: element e of image h sits at 2 e + h (WgFftL ES = 2), so the lanes
(p, 0), (p, 1) touch neighbouring elements in every access of the two flows.  That matters for the cross-lane ds_write_b64:
the hardware serves them in groups of 16 lanes on 32 banks (16 elements), and with the images a constant offset apart
(round-3 first form: 16 x 272 + 16 elements) both halves of 8 lanes fell on the same 8 elements -- every such write 2-way
conflicted (SQ_LDS_BANK_CONFLICT 40 % of the array cycles, cfg5pad column launch 227 -> 214 us without them).

X1 goes through the wave's own row regions (wave W holds p in [32 W, 32 W + 32): regions 2W, 2W+1 of both images = elements
[1088 W, 1088 W + 1088)): block c in {0: sums, 1: differences} at 513 c + 64 i + 2 (p % 32) + h' for register i of lane
(p, h') -- 64 consecutive elements per write instruction; the odd offset of block 1 puts the readers of the two halves on
odd / even elements.
"""
import numpy as np

import fft_local_model as L4

N, T = 8192, 512
WREG = 2 * 2 * 272          # elements of a wave's own regions (two rows of both images)


def s_of(j):
    return L4.js(j >> 1) + 256 * (j & 1)


def x1_slot(p, hsrc, c, i):
    """LDS element of register i of block c written by lane (p, hsrc)."""
    wave = p >> 5
    return WREG * wave + 513 * c + i * 64 + 2 * (p & 31) + hsrc


def forward(x, nz=16):
    """nz: registers r >= nz of every lane are zero on input (nz <= 8 prunes the radix-2 step to copies)."""
    w8 = lambda e: np.exp(-2j * np.pi * (e % N) / N)
    reg = np.array([[x[s_of(j) + 512 * r] for r in range(16)] for j in range(T)], dtype=np.complex128)
    lds = np.full(2 * 16 * 272, np.nan, dtype=np.complex128)
    for j in range(T):
        p, h = j >> 1, j & 1
        v = reg[j].copy()
        for i in range(8):
            a, b = v[i], v[i + 8]
            v[i] = a + b
            v[i + 8] = (a - b) * w8(s_of(j)) * np.exp(-2j * np.pi * i / 16)
        for c in range(2):
            for i in range(8):
                e = x1_slot(p, h, c, i)
                assert WREG * (p >> 5) <= e < WREG * ((p >> 5) + 1)     # the wave's own regions
                lds[e] = v[i + 8 * c]
    half = np.zeros((2, 256, 16), dtype=np.complex128)
    for j in range(T):
        p, h = j >> 1, j & 1
        for i in range(8):
            for hsrc in range(2):
                half[h, p, 2 * i + hsrc] = lds[x1_slot(p, hsrc, h, i)]     # y_h[js(p) + 256 (2 i + hsrc)]
    X = np.zeros(N, dtype=np.complex128)
    for h in range(2):
        y = np.zeros(4096, dtype=np.complex128)
        for p in range(256):
            for n2 in range(16):
                y[L4.js(p) + 256 * n2] = half[h, p, n2]
        if nz <= 8:
            assert np.all(half[:, :, 2 * nz:] == 0)       # the 4096-point transforms see 2 nz leading non-zero slots
        Y = L4.forward(y)
        for p in range(256):
            for m in range(16):
                X[2 * p + h + 512 * m] = Y[p + 256 * m]      # lane j = 2 p + h, register m: standard frequency layout
    return X


def inverse(X):
    w8c = lambda e: np.exp(+2j * np.pi * (e % N) / N)
    lds = np.full(2 * 16 * 272, np.nan, dtype=np.complex128)
    half = np.zeros((2, 256, 16), dtype=np.complex128)
    for h in range(2):
        Y = np.array([X[2 * k + h] for k in range(4096)])
        y = L4.inverse(Y)                                   # unnormalised: 4096 * y_h
        for p in range(256):
            for n2 in range(16):
                half[h, p, n2] = y[L4.js(p) + 256 * n2]
    for j in range(T):                                       # X1 backwards: lane (p, h) writes y_h[n2 = 2 i + hdst] for lane (p, hdst)
        p, h = j >> 1, j & 1
        for i in range(8):
            for hdst in range(2):
                lds[x1_slot(p, hdst, h, i)] = half[h, p, 2 * i + hdst]
    x = np.zeros(N, dtype=np.complex128)
    for j in range(T):
        p, h = j >> 1, j & 1
        for i in range(8):
            y0 = lds[x1_slot(p, h, 0, i)]
            y1 = lds[x1_slot(p, h, 1, i)] * w8c(s_of(j)) * np.exp(+2j * np.pi * i / 16)
            x[s_of(j) + 512 * i] = y0 + y1
            x[s_of(j) + 512 * (i + 8)] = y0 - y1
    return x


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    x = rng.normal(size=N) + 1j * rng.normal(size=N)
    X = forward(x)
    print("forward  max err", np.abs(X - np.fft.fft(x)).max())
    print("inverse  max err", np.abs(inverse(X) - N * x).max())
    xz = x.copy()
    for j in range(T):
        for r in range(6, 16):
            xz[s_of(j) + 512 * r] = 0
    print("forward, 6 leading slots max err", np.abs(forward(xz, nz=6) - np.fft.fft(xz)).max())
    assert sorted(s_of(j) for j in range(T)) == list(range(512))
    # X1 bank check (8-byte elements, 32 element-banks per 32-lane group): writes (fixed c, i) and reads (fixed hsrc, i)
    worst_w = worst_r = 0
    for i in range(8):
        for g in range(0, T, 32):
            for c in range(2):
                banks = [x1_slot(j >> 1, j & 1, c, i) % 32 for j in range(g, g + 32)]
                worst_w = max(worst_w, 32 - len(set(banks)))
            for hsrc in range(2):
                banks = [x1_slot(j >> 1, hsrc, j & 1, i) % 32 for j in range(g, g + 32)]
                worst_r = max(worst_r, 32 - len(set(banks)))
    print("X1 bank collisions per 32-lane group: writes", worst_w, "reads", worst_r)
    # the two interleaved 4096-point flows (element of image h at 2 e + h; e(p, i) as in fft_local_model): cross-lane writes
    # in 16-lane groups on 16 elements, reads in 32-lane groups on 32
    pats = {"local write": lambda p, i: 272 * (p >> 4) + 17 * (p & 15) + i, "local read": lambda p, i: 272 * (p >> 4) + 17 * i + (p & 15),
            "global scatter": lambda p, i: 272 * (p >> 4) + 16 * i + (p & 15), "global gather": lambda p, i: 272 * i + p}
    for name, f in pats.items():
        for width in (16, 32):
            worst = 0
            for i in range(16):
                for g in range(0, T, width):
                    el = [(2 * f(j >> 1, i) + (j & 1)) % width for j in range(g, g + width)]
                    worst = max(worst, width - len(set(el)))
            print(f"{name:15s} collisions per {width}-lane group on {width} elements: {worst}")
