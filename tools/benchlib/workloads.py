"""Workload tables and peaks of bench.py (BASELINE.json configurations; peaks from MI355X_MICROARCH.md)."""
import numpy as np

HBM_PEAK = 8.0e12          # bytes/s, MI355X_MICROARCH.md
MFMA_F32_PEAK = 157.3e12   # flop/s, dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)
MALL_BYTES = 256 * 2 ** 20

SPOT_WORKLOADS = {
    # name: (padded shape, slm shape, spot grid, pitch)
    "cfg2": ((4096, 4096), (1152, 1920), (32, 32), (64, 64)),
    "cfg3": ((4096, 4096), (1152, 1920), (32, 32), (64, 64)),
    "small": ((1024, 1024), (288, 480), (16, 16), (32, 32)),
    "hd": ((2048, 2048), (1080, 1920), (16, 16), (64, 64)),        # a 1920x1080 SLM at padding_order = 1
    "cfg5pad": ((8192, 8192), (1152, 1920), (32, 32), (128, 128)),
}
IMAGE_WORKLOADS = {"cfg1": ((512, 512), (512, 512)), "cfg2dense": ((4096, 4096), (1152, 1920)), "cfg5mraf": ((8192, 8192), (1152, 1920))}
REFBENCH_METHODS = ("GS", "WGS-Leonardo", "WGS-Kim", "WGS-Nogrette")      # test_algorithms.py:121
COMPRESSED_WORKLOADS = {"cfg4": 2, "cfg4d3": 3, "cfg4zern": 5}
# cfg4zern: a basis with a cross term (ANSI 2, 1, 4, 3, 5: tilts, focus, both astigmatisms) does not factor into x and y
# parts, so it runs the direct kernels (exp(i phi) regenerated per pixel and spot, VALU / transcendental bound) -- the
# shape of the CompressedSpotHologram that wavefront_calibrate_zernike re-optimises (cameraslms.py:1840-1930)
ZERN_BASIS = [2, 1, 4, 3, 5]
VALU_F32_PEAK = 157.3e12   # flop/s, packed fp32 vector peak (MI355X_MICROARCH.md)
# cfg 4's DFT-grid companion: the same number of spots at distinct pixels of an 8192^2 grid, inside the
# centred 3360^2 box that |k| <= 0.02 rad spans there (pitch 8 um, 0.78 um), WGS-Kim
VECTOR_WORKLOADS = {"cfg4grid": ((8192, 8192), (1152, 1920), 3360)}
ALL_WORKLOADS = sorted(list(SPOT_WORKLOADS) + list(IMAGE_WORKLOADS) + list(COMPRESSED_WORKLOADS) + list(VECTOR_WORKLOADS) + ["refbench"])


def grid_spots(shape, box, n):
    """``n`` distinct pixels (x, y) inside the centred ``box`` x ``box`` window of ``shape``, from the counter PRNG."""
    from slmsuite_amd import synth
    lin = np.unique((synth.uniform01(4, (4 * n,), stream=7) * box * box).astype(np.int64))
    if lin.size < n:
        raise SystemExit(f"only {lin.size} distinct positions for {n} spots")
    lin = np.sort(lin[np.argsort(synth.uniform01(5, (lin.size,), stream=8), kind="stable")[:n]])
    lo_y, lo_x = (shape[0] - box) // 2, (shape[1] - box) // 2
    return np.stack([lin % box + lo_x, lin // box + lo_y]).astype(np.float64)
