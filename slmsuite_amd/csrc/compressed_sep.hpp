// Separable form of the CompressedSpotHologram transforms (_spots.py:767-914).
//
// When every monomial of the kernel phase polynomial involves x or y alone (tilt, tilt + focus:
// the default 2-D / 3-D bases) and the SLM grid is a product grid, the kernel factorises,
//     exp(-i phi_n(x, y)) = Ex[n][x] * Ey[n][y],   Ex = exp(-i fx_n(x)),  Ey = exp(-i fy_n(y)),
// and both directions become dense complex contractions that run on the matrix cores (cgemm.hpp):
//     n2f:  T[n][y]  = sum_x Ex[n][x] nf[y][x];          ff_n = sum_y Ey[n][y] T[n][y] / sqrt(S)
//     f2n:  conj(nf[y][x]) = sum_n (conj(ff_n) Ey[n][y]) Ex[n][x];   phase = atan2(nf) - kernel
// Ex (in both orientations) and Ey are tabulated once per change of the spot coefficients, with the
// polynomial and its range reduction evaluated in double (the direct kernels, like the reference,
// evaluate phi in fp32).  2 * 8 * N * S flop per iteration: 3.5e11 at N = 1e4, S = 1152 x 1920.
#pragma once
#include "kernels.hpp"

namespace hgs {

constexpr int SEP_MAXDEG = 8;

// tab[n][i] = exp(-i f_n(g[i])), f_n(u) = sum_p c[p][n] u^p; optionally also the transpose tabT[i][n].
// grid = (ceil(len/256), N)
// ld / ldT: leading dimensions of tab / tabT (>= len / N; the matrix-core GEMM wants 128-multiples, zero padded).
// GEMM operands are PLANAR (cgemm_streamk): tab_plane / tabT_plane > 0 = distance in floats from the real to the imaginary
// array of that table; 0 = interleaved float2 (Ey, which only the elementwise helpers read).
__global__ void sep_build_table(const double* c, int deg, int N, const double* g, int len, float2* tab, int ld, size_t tab_plane,
                                float2* tabT, int ldT, size_t tabT_plane) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, n = blockIdx.y;
    if (i >= len) return;
    const double u = g[i];
    double f = 0;
    for (int p = deg; p >= 0; --p) f = f * u + c[(size_t)p * N + n];
    // reduce to [-0.5, 0.5) turns before the trigonometric evaluation
    double tr = f * 0.15915494309189533577;
    tr -= rint(tr);
    double s, co;
    ::sincos(tr * 6.28318530717958647692, &s, &co);
    const float2 v = make_float2((float)co, (float)(-s));
    if (tab_plane > 0) {
        float* q = reinterpret_cast<float*>(tab);
        q[(size_t)n * ld + i] = v.x;
        q[tab_plane + (size_t)n * ld + i] = v.y;
    } else {
        tab[(size_t)n * ld + i] = v;
    }
    if (tabT != nullptr) {
        float* q = reinterpret_cast<float*>(tabT);
        q[(size_t)i * ldT + n] = v.x;
        q[tabT_plane + (size_t)i * ldT + n] = v.y;
    }
}

// nfT[b][x][y] = amp[y][x] * exp(i (phase[b][y][x] + kern[y][x]))   (32 x 32 LDS transpose tiles)
// grid = (ceil(W/32), ceil(H/32), batch), block = (32, 8)
// nfT is planar (real array, then the imaginary one `plane` floats on), leading dimension ldH, batch stride 2 * plane
template <typename R>
__global__ void sep_build_nft(const R* phase, const R* amp, const R* kern, R amp_scalar, int H, int W, float* nfT,
                              int ldH, size_t plane) {
    __shared__ float2 tile[32][33];
    const int b = blockIdx.z;
    const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int y = y0 + r, x = x0 + threadIdx.x;
        float2 v = make_float2(0.f, 0.f);
        if (y < H && x < W) {
            const size_t p = (size_t)y * W + x;
            R ph = phase[(size_t)b * H * W + p];
            if (kern != nullptr) ph += kern[p];
            R s, c;
            Math<R>::sincos(ph, &s, &c);
            const R am = (amp != nullptr) ? amp[p] : amp_scalar;
            v = make_float2((float)(am * c), (float)(am * s));
        }
        tile[r][threadIdx.x] = v;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int x = x0 + r, y = y0 + threadIdx.x;
        if (x < W && y < H) {
            float* q = nfT + (size_t)b * 2 * plane + (size_t)x * ldH + y;
            q[0] = tile[threadIdx.x][r].x;
            q[plane] = tile[threadIdx.x][r].y;
        }
    }
}

// ff_raw[n] = sum of the partial y contractions the n2f GEMM left (cgemm_streamk<1>: one per tile column, wave column and
// partial plane of the tile) / sqrt(S), in double; also the per-block partial of sum |ff_raw|^2 for c_n2f_finish.
// grid = (ceil(N/256), batch), block 256
template <typename R>
__global__ void sep_n2f_sum(const float2* part, int ldP, int planes, const int* nseg, int tiles_m, int tiles_n, int N,
                            double inv_sqrt_s, Cx<R>* ff, double* norm_partial) {
    __shared__ double scratch[16];
    const int b = blockIdx.y;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    double p2 = 0;
    if (n < N) {
        double sr = 0, si = 0;
        const float2* base = part + (size_t)b * tiles_n * 2 * planes * (size_t)ldP + n;
        for (int bn = 0; bn < tiles_n; ++bn) {
            const int ns = nseg[(n >> 7) + bn * tiles_m];
            for (int wn = 0; wn < 2; ++wn)
                for (int s = 0; s < ns; ++s) {
                    const float2 v = base[(size_t)((bn * 2 + wn) * planes + s) * ldP];
                    sr += (double)v.x;
                    si += (double)v.y;
                }
        }
        const Cx<R> f = mk<R>((R)(sr * inv_sqrt_s), (R)(si * inv_sqrt_s));
        ff[(size_t)b * N + n] = f;
        p2 = (double)f.x * f.x + (double)f.y * f.y;
        if (p2 != p2) p2 = 0;
    }
    const double tot = block_sum(p2, scratch);
    if (threadIdx.x == 0) norm_partial[(size_t)b * gridDim.x + blockIdx.x] = tot;
}

// B2[b][n][y] = conj(ff[b][n]) * Ey[n][y], planar like nfT.   grid = (ceil(H/256), N, batch)
template <typename R>
__global__ void sep_build_b2(const Cx<R>* ff, const float2* Ey, int N, int H, float* B2, int ldH, size_t plane) {
    const int y = blockIdx.x * blockDim.x + threadIdx.x, n = blockIdx.y, b = blockIdx.z;
    if (y >= H) return;
    const Cx<R> f = ff[(size_t)b * N + n];
    const float2 e = Ey[(size_t)n * H + y];
    const float fr = (float)f.x, fi = -(float)f.y;
    float* q = B2 + (size_t)b * 2 * plane + (size_t)n * ldH + y;
    q[0] = fr * e.x - fi * e.y;
    q[plane] = fr * e.y + fi * e.x;
}

// nf = conj(sum_s C[s]) / sqrt(S): phase = atan2(nf) - kernel (:1030-1036), or the complex nearfield
// (extract = False).  grid = (ceil(S/256), batch)
template <typename R>
__global__ void sep_f2n_finish(const float2* C, int split, const int* nseg, int tiles_m, int W, size_t S, const R* kern, R* phase,
                               Cx<R>* nf_out) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (p >= S) return;
    float re = 0, im = 0;
    const int y = (int)(p / (size_t)W), x = (int)(p - (size_t)y * W);
    const int ns = nseg[(y >> 7) + (x >> 7) * tiles_m];
    for (int s = 0; s < ns; ++s) {
        const float2 v = C[((size_t)b * split + s) * S + p];
        re += v.x;
        im -= v.y;
    }
    if (nf_out != nullptr) {
        const R sc = Math<R>::rsqrt((R)S);
        nf_out[(size_t)b * S + p] = mk<R>((R)re * sc, (R)im * sc);
    } else {
        R ph = Math<R>::atan2((R)im, (R)re);
        if (kern != nullptr) ph -= kern[p];
        phase[(size_t)b * S + p] = ph;
    }
}

}  // namespace hgs
