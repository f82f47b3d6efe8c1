"""Accuracy check of the device sincos (fp64 range reduction + fp32 polynomials), emulated in NumPy."""
import numpy as np

f = np.float32


def fma32(a, b, c):
    return f(np.float64(a) * np.float64(b) + np.float64(c))


def sincos(x):
    xd = np.float64(f(x))
    qd = np.rint(xd * 0.63661977236758134308)
    r = f(xd - qd * 1.57079632679489661923)
    z = f(r * r)
    ps = fma32(fma32(fma32(f(-1.9515295891e-4), z, f(8.3321608736e-3)), z, f(-1.6666654611e-1)), f(z * r), r)
    pc = fma32(fma32(fma32(f(2.443315711809948e-5), z, f(-1.388731625493765e-3)), z, f(4.166664568298827e-2)),
               f(z * z), fma32(f(-0.5), z, f(1.0)))
    n = int(qd)
    ss, cc = (pc, ps) if n & 1 else (ps, pc)
    return (-ss if n & 2 else ss), (-cc if (n + 1) & 2 else cc)


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for lim in (3.2, 100.0, 1e4, 1e6):
        xs = rng.uniform(-lim, lim, 20000).astype(np.float32)
        es = max(abs(float(sincos(x)[0]) - np.sin(np.float64(x))) for x in xs)
        ec = max(abs(float(sincos(x)[1]) - np.cos(np.float64(x))) for x in xs)
        print(f"|x| <= {lim:g}: max abs err sin {es:.2e} cos {ec:.2e}")
