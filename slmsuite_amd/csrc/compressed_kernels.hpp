// Non-uniform DFT kernels of the CompressedSpotHologram path (reference: _spots.py:677-914 and the
// opt-in RawKernels of toolbox/cuda.cu:95-288).  N free-floating spots, each with a polynomial
// (Zernike) phase kernel  phi_n(p) = sum_m b[m][n] * x_p^px[m] * y_p^py[m]  over the SLM pixels p:
//     farfield  ff_n = sum_p nf_p * exp(-i phi_n(p)) / sqrt(S)      then ff /= ||ff||   (:787-822)
//     nearfield nf_p = sum_n ff_n * exp(+i phi_n(p)) / sqrt(S)                           (:887-914)
// The reference materialises K[n,p] (N*S complex, cached or cycled in 256-spot batches); here the
// kernel values are regenerated on the fly in registers -- the path is VALU/transcendental bound
// (2*N*S kernel evaluations per iteration), not HBM bound, so no kernel matrix ever touches memory.
// Monomial coefficients of the current spot are wave-uniform (scalar loads); sin/cos use the
// hardware v_sin/v_cos on the fractional number of turns (phi itself is only fp32-accurate, which
// bounds the useful precision exactly as in the reference's complex64 kernel).
#pragma once
#include "kernels.hpp"

namespace hgs {

template <typename R> struct CArgs {
    int S, N, M, batch;
    const R* xg;        // [S] pupil-scaled grid
    const R* yg;
    const int* mono;    // [M][2] (px, py)
    const R* coeff;     // [M][N]
    R* phase;           // [b][S]
    const R* amp;       // [S] or nullptr
    const R* kern;      // [S] or nullptr
    R amp_scalar;
    Cx<R>* ff;          // [b][N]
    R* amp_ff;          // [b][N]
    Cx<R>* partial;     // [b][nblocks][N]
    int nblocks;
    double* fsum;       // [b] sum |ff|^2 after normalisation (= 1) for the constraint kernels
    int degree;         // max(px + py)
    Cx<R>* nf_out;      // c_f2n: store the complex nearfield [b][S] (scaled 1/sqrt(S)) instead of the phase
};

template <typename R> struct Trig;
template <> struct Trig<float> {
    // exp(i*phi) from phi in radians
    static __device__ __forceinline__ void cis(float phi, float* c, float* s) {
        const float t = __builtin_amdgcn_fractf(phi * 0.15915494309189533577f);
        *s = __builtin_amdgcn_sinf(t);
        *c = __builtin_amdgcn_cosf(t);
    }
};
template <> struct Trig<double> {
    static __device__ __forceinline__ void cis(double phi, double* c, double* s) { ::sincos(phi, s, c); }
};

constexpr int C_PT = 4;      // pixels per lane
constexpr int C_WG = 256;
constexpr int C_NC = 64;     // spots per cross-wave reduction round

// Phase polynomial of spot n.  DEG 1/2: the host repacks the monomial weights into the canonical
// order [1, x, y, x^2, xy, y^2] (coeff6[k][n], zero where a monomial is absent) so the evaluation is a
// short Horner form with wave-uniform coefficients; DEG 0: arbitrary monomial list, including the
// non-polynomial pseudo-term (-1, 0) (vortex plate).
template <typename R, int DEG> struct SpotPoly {
    R c[6];
    __device__ __forceinline__ void load(const CArgs<R>& a, int n) {
        if constexpr (DEG >= 1) {
            c[0] = a.coeff[n];
            c[1] = a.coeff[(size_t)a.N + n];
            c[2] = a.coeff[(size_t)2 * a.N + n];
        }
        if constexpr (DEG == 2) {
            c[3] = a.coeff[(size_t)3 * a.N + n];
            c[4] = a.coeff[(size_t)4 * a.N + n];
            c[5] = a.coeff[(size_t)5 * a.N + n];
        }
    }
    __device__ __forceinline__ R eval(const CArgs<R>& a, int n, R x, R y) const {
        if constexpr (DEG == 1) {
            return c[0] + c[1] * x + c[2] * y;
        } else if constexpr (DEG == 2) {
            return c[0] + x * (c[1] + c[3] * x + c[4] * y) + y * (c[2] + c[5] * y);
        } else {
            R phi = 0;
            for (int m = 0; m < a.M; ++m) {
                const R cm = a.coeff[(size_t)m * a.N + n];
                const int px = a.mono[2 * m], py = a.mono[2 * m + 1];
                if (px < 0) {      // pseudo-term (-1, 0): vortex plate, positive charges only (phase.py:1783-1790)
                    if (px == -1 && cm > (R)0) phi += cm * Math<R>::atan2(y, x);
                    continue;
                }
                R v = 1;
                for (int k = 0; k < px; ++k) v *= x;
                for (int k = 0; k < py; ++k) v *= y;
                phi += cm * v;
            }
            return phi;
        }
    }
};

// nearfield -> per-block partial farfield sums.  grid = (nblocks, batch), block = 256.
template <typename R, int DEG> __global__ __launch_bounds__(C_WG) void c_n2f_partial(CArgs<R> a) {
    using M = Math<R>;
    __shared__ R red[C_WG / 64][C_NC][2];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    R x[C_PT], y[C_PT], re[C_PT], im[C_PT];
#pragma unroll
    for (int i = 0; i < C_PT; ++i) {
        const int p = (blockIdx.x * C_PT + i) * C_WG + tid;
        x[i] = y[i] = re[i] = im[i] = 0;
        if (p < a.S) {
            x[i] = a.xg[p];
            y[i] = a.yg[p];
            R ph = a.phase[(size_t)b * a.S + p];      // _build_nearfield :1000-1011
            if (a.kern) ph += a.kern[p];
            R s, c;
            M::sincos(ph, &s, &c);
            const R am = a.amp ? a.amp[p] : a.amp_scalar;
            re[i] = am * c;
            im[i] = am * s;
        }
    }
    Cx<R>* out = a.partial + ((size_t)b * a.nblocks + blockIdx.x) * a.N;
    for (int n0 = 0; n0 < a.N; n0 += C_NC) {
        const int nc = min(C_NC, a.N - n0);
        for (int k = 0; k < nc; ++k) {
            SpotPoly<R, DEG> sp;
            sp.load(a, n0 + k);
            R sr = 0, si = 0;
#pragma unroll
            for (int i = 0; i < C_PT; ++i) {
                R c, s;
                Trig<R>::cis(sp.eval(a, n0 + k, x[i], y[i]), &c, &s);
                sr += re[i] * c + im[i] * s;       // nf * exp(-i phi)
                si += im[i] * c - re[i] * s;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                sr += __shfl_down(sr, o, 64);
                si += __shfl_down(si, o, 64);
            }
            if (lane == 0) { red[wid][k][0] = sr; red[wid][k][1] = si; }
        }
        __syncthreads();
        if (tid < nc) {
            R tr = 0, ti = 0;
#pragma unroll
            for (int w = 0; w < C_WG / 64; ++w) { tr += red[w][tid][0]; ti += red[w][tid][1]; }
            out[n0 + tid] = mk<R>(tr, ti);
        }
        __syncthreads();
    }
}

// ff_raw[n] = sum over blocks of the partials / sqrt(S), plus per-block partial of sum |ff_raw|^2.
// grid = (ceil(N/256), batch)
template <typename R> __global__ void c_n2f_reduce(CArgs<R> a, double* norm_partial) {
    __shared__ double scratch[16];
    const int b = blockIdx.y;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const R inv_sqrt_s = (R)(1.0 / ::sqrt((double)a.S));
    double p2 = 0;
    if (n < a.N) {
        double sr = 0, si = 0;
        for (int k = 0; k < a.nblocks; ++k) {
            const Cx<R> v = a.partial[((size_t)b * a.nblocks + k) * a.N + n];
            sr += (double)v.x;
            si += (double)v.y;
        }
        const Cx<R> f = mk<R>((R)sr * inv_sqrt_s, (R)si * inv_sqrt_s);
        a.ff[(size_t)b * a.N + n] = f;
        p2 = (double)f.x * f.x + (double)f.y * f.y;
        if (p2 != p2) p2 = 0;
    }
    const double s = block_sum(p2, scratch);
    if (threadIdx.x == 0) norm_partial[(size_t)b * gridDim.x + blockIdx.x] = s;
}

// normalise ff to unit L2 (:822), fill amp_ff and sum amp_ff^2.  grid = (batch), block = 256
template <typename R> __global__ void c_n2f_finish(CArgs<R> a, const double* norm_partial, int n_partial) {
    __shared__ double scratch[16];
    __shared__ double nrm;
    const int b = blockIdx.x;
    double acc = 0;
    for (int i = threadIdx.x; i < n_partial; i += blockDim.x) acc += norm_partial[(size_t)b * n_partial + i];
    const double s = block_sum(acc, scratch);
    if (threadIdx.x == 0) nrm = s;
    __syncthreads();
    const R inv = (R)(1.0 / ::sqrt(nrm));
    double acc2 = 0;
    for (int n = threadIdx.x; n < a.N; n += blockDim.x) {
        const Cx<R> f = a.ff[(size_t)b * a.N + n] * inv;
        a.ff[(size_t)b * a.N + n] = f;
        const R am = Math<R>::sqrt(f.x * f.x + f.y * f.y);
        a.amp_ff[(size_t)b * a.N + n] = am;
        acc2 += (double)am * (double)am;
    }
    const double s2 = block_sum(acc2, scratch);
    if (threadIdx.x == 0) a.fsum[b] = s2;
}

// farfield -> nearfield phase.  grid = (nblocks, batch)
template <typename R, int DEG> __global__ __launch_bounds__(C_WG) void c_f2n(CArgs<R> a) {
    using M = Math<R>;
    const int b = blockIdx.y, tid = threadIdx.x;
    R x[C_PT], y[C_PT], re[C_PT], im[C_PT];
#pragma unroll
    for (int i = 0; i < C_PT; ++i) {
        const int p = (blockIdx.x * C_PT + i) * C_WG + tid;
        x[i] = y[i] = 0;
        re[i] = im[i] = 0;
        if (p < a.S) { x[i] = a.xg[p]; y[i] = a.yg[p]; }
    }
    const Cx<R>* ff = a.ff + (size_t)b * a.N;
    for (int n = 0; n < a.N; ++n) {
        const Cx<R> f = ff[n];
        SpotPoly<R, DEG> sp;
        sp.load(a, n);
#pragma unroll
        for (int i = 0; i < C_PT; ++i) {
            R c, s;
            Trig<R>::cis(sp.eval(a, n, x[i], y[i]), &c, &s);
            re[i] += f.x * c - f.y * s;            // ff * exp(+i phi)
            im[i] += f.x * s + f.y * c;
        }
    }
#pragma unroll
    for (int i = 0; i < C_PT; ++i) {
        const int p = (blockIdx.x * C_PT + i) * C_WG + tid;
        if (p < a.S) {
            if (a.nf_out != nullptr) {             // _farfield2nearfield(extract=False) (_spots.py:887-914)
                const R sc = M::rsqrt((R)a.S);
                a.nf_out[(size_t)b * a.S + p] = mk<R>(re[i] * sc, im[i] * sc);
            } else {
                R ph = M::atan2(im[i], re[i]);     // the 1/sqrt(S) scale does not change the phase (:1030)
                if (a.kern) ph -= a.kern[p];
                a.phase[(size_t)b * a.S + p] = ph;
            }
        }
    }
}

template <typename R> __global__ void convert_d2r(const double* in, R* out, int n, int batch) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        for (int b = 0; b < batch; ++b) out[(size_t)b * n + i] = (R)in[i];
}

}  // namespace hgs
