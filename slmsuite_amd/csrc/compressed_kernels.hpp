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
// DEG 3 (round 5): an arbitrary list of at most C_MTAB monomials (Zernike bases up to radial degree 4 -- what
// wavefront_calibrate_zernike re-optimises, cameraslms.py:1840-1930) with the monomial VALUES of a lane's pixels formed once,
// ahead of the spot loop (MonoTab), so that a spot costs one multiply-add per monomial and pixel instead of the exponent
// loops and their loads: 16 spots x 10 terms on a 1152 x 1920 SLM, GS x 3: 2.46 -> see NOTEBOOK round 5.
constexpr int C_MTAB = 16;
template <typename R> struct MonoTab {
    R v[C_PT][C_MTAB];
    // the same products, in the same order, as SpotPoly<R, 0>::eval forms per spot
    __device__ __forceinline__ void build(const CArgs<R>& a, const R (&x)[C_PT], const R (&y)[C_PT]) {
        static_for<0, C_MTAB>([&](auto m_) {
            constexpr int m = m_;
            const int px = m < a.M ? a.mono[2 * m] : 0, py = m < a.M ? a.mono[2 * m + 1] : 0;
#pragma unroll
            for (int i = 0; i < C_PT; ++i) {
                R t = 1;
                if (px < 0) {
                    t = Math<R>::atan2(y[i], x[i]);     // vortex plate pseudo-term (-1, 0)
                } else {
                    for (int k = 0; k < px; ++k) t *= x[i];
                    for (int k = 0; k < py; ++k) t *= y[i];
                }
                v[i][m] = m < a.M ? t : (R)0;
            }
        });
    }
};

template <typename R, int DEG> struct SpotPoly {
    R c[DEG == 3 ? C_MTAB : 6];
    __device__ __forceinline__ void load(const CArgs<R>& a, int n) {
        if constexpr (DEG == 3) {
            static_for<0, C_MTAB>([&](auto m_) {
                constexpr int m = m_;
                R cm = m < a.M ? a.coeff[(size_t)m * a.N + n] : (R)0;
                if (m < a.M && a.mono[2 * m] < 0 && !(cm > (R)0)) cm = 0;      // vortex plate: positive charges only (phase.py:1783-1790)
                c[m] = cm;
            });
            return;
        }
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
    // DEG 3: pixel i of the lane's table
    __device__ __forceinline__ R eval_tab(const MonoTab<R>& tab, int i) const {
        R phi = 0;
        static_for<0, C_MTAB>([&](auto m_) { constexpr int m = m_; phi += c[m] * tab.v[i][m]; });
        return phi;
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
    MonoTab<R> tab;
    if constexpr (DEG == 3) tab.build(a, x, y);
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
                if constexpr (DEG == 3) Trig<R>::cis(sp.eval_tab(tab, i), &c, &s);
                else Trig<R>::cis(sp.eval(a, n0 + k, x[i], y[i]), &c, &s);
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
// grid = (ceil(N / 64), batch), block = 1024: 64 spots x 16 slices of the block list (a thread per spot walking
// thousands of partials one dependent load at a time was a 0.5 ms floor under every direct transform)
constexpr int C_RED_SPOTS = 64, C_RED_SLICES = 16;
template <typename R> __global__ __launch_bounds__(C_RED_SPOTS * C_RED_SLICES) void c_n2f_reduce(CArgs<R> a, double* norm_partial) {
    __shared__ double scratch[16];
    __shared__ double red[C_RED_SLICES][C_RED_SPOTS][2];
    const int b = blockIdx.y;
    const int tx = threadIdx.x % C_RED_SPOTS, ty = threadIdx.x / C_RED_SPOTS;
    const int n = blockIdx.x * C_RED_SPOTS + tx;
    const R inv_sqrt_s = (R)(1.0 / ::sqrt((double)a.S));
    double sr = 0, si = 0;
    if (n < a.N) {
        const Cx<R>* src = a.partial + (size_t)b * a.nblocks * a.N + n;
#pragma unroll 8
        for (int k = ty; k < a.nblocks; k += C_RED_SLICES) {
            const Cx<R> v = src[(size_t)k * a.N];
            sr += (double)v.x;
            si += (double)v.y;
        }
    }
    red[ty][tx][0] = sr;
    red[ty][tx][1] = si;
    __syncthreads();
    double p2 = 0;
    if (ty == 0 && n < a.N) {
        sr = si = 0;
#pragma unroll
        for (int q = 0; q < C_RED_SLICES; ++q) { sr += red[q][tx][0]; si += red[q][tx][1]; }
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
    MonoTab<R> tab;
    if constexpr (DEG == 3) tab.build(a, x, y);
    for (int n = 0; n < a.N; ++n) {
        const Cx<R> f = ff[n];
        SpotPoly<R, DEG> sp;
        sp.load(a, n);
#pragma unroll
        for (int i = 0; i < C_PT; ++i) {
            R c, s;
            if constexpr (DEG == 3) Trig<R>::cis(sp.eval_tab(tab, i), &c, &s);
            else Trig<R>::cis(sp.eval(a, n, x[i], y[i]), &c, &s);
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

// =====================================================================================================
// Run kernels (fp32, regular pixel grid, phase polynomial of degree <= 2 in the canonical form).
// A lane owns CR_RUN consecutive pixels of one SLM row.  Along such a run the kernel value of spot n obeys
//     E(x + h) = E(x) D(x),   D(x + h) = D(x) C_n,   C_n = exp(i 2 c3 h^2)   (degree 1: D = exp(i c1 h) per spot, C = 1)
// so one evaluation costs two packed complex products (and the accumulate) instead of a phase polynomial, a range
// reduction, v_sin and v_cos (quarter rate): 7-8 instead of 22 issue slots per evaluation.  The start values
// E(x0), D(x0) of every (lane, spot) are formed from the polynomial in DOUBLE and reduced to turns there, i.e. they are
// exact to fp32 rounding where the direct kernels (and the reference's complex64 kernel) carry the fp32 error of a phase
// of hundreds of radians; the recurrence adds at most ~16 roundings of 6e-8.
// Work split: pixel runs over grid.x (one wave per workgroup: 2,160 of them for 1152 x 1920, too few to balance 1,024
// SIMDs) times grid.z chunks of the spot list; the inverse direction leaves one partial nearfield per chunk and a small
// kernel adds them in a fixed order (no atomics: results do not depend on the schedule).
// =====================================================================================================
constexpr int CR_RUN = 16;
struct CRunRec {          // one spot, 64 bytes: canonical coefficients in double, C_n (degree 2) / D_n (degree 1)
    double c[6];
    float cr, ci, pad0, pad1;
};
struct CRunArgs {
    CArgs<float> a;
    const CRunRec* rec;   // [N]
    const double* ys;     // [H] row coordinate
    double x0, hx;        // column coordinate = x0 + col * hx
    int H, W, rpr;        // runs per row = ceil(W / CR_RUN)
    int n_per;            // spots per grid.z chunk
    Cx<float>* nf_part;   // [b][gridDim.z][S] partial nearfields of c_f2n_run
};

__device__ __forceinline__ v2f cis_turns(double turns) {
    const float t = (float)__builtin_amdgcn_fract(turns);
    return (v2f){__builtin_amdgcn_cosf(t), __builtin_amdgcn_sinf(t)};
}
// acc + a * conj(b)
__device__ __forceinline__ v2f cmac_conj(v2f acc, v2f a, v2f b) {
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(acc));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "+v"(r) : "v"(a), "v"(b));
    return r;
}
// sum over the 64 lanes, delivered to every lane: prefix sums inside the rows of 16 (row_shr 1, 2, 4, 8), row 0 -> 1 and
// 2 -> 3 (row_bcast:15), rows 0-1 -> 2-3 (row_bcast:31), then lane 63 through a scalar register.  Six DPP adds and a lane
// read instead of six ds_bpermute + add pairs per value.
template <int CTRL, int ROW_MASK = 0xf> __device__ __forceinline__ float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, true));
}
__device__ __forceinline__ float wave_sum_all(float v) {
    v = dpp_add<0x111>(v);
    v = dpp_add<0x112>(v);
    v = dpp_add<0x114>(v);
    v = dpp_add<0x118>(v);
    v = dpp_add<0x142, 0xa>(v);
    v = dpp_add<0x143, 0xc>(v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
template <int DEG> __device__ __forceinline__ void run_start(const CRunRec& r, double xd, double yd, double q, double hx,
                                                             v2f& E, v2f& D) {
    constexpr double INV2PI = 0.15915494309189533577;
    if constexpr (DEG == 2) {
        const double A = fma(yd, fma(r.c[5], yd, r.c[2]), r.c[0]);
        const double Bq = fma(r.c[4], yd, r.c[1]);
        E = cis_turns(fma(xd, fma(r.c[3], xd, Bq), A) * INV2PI);
        D = cis_turns(hx * fma(r.c[3], q, Bq) * INV2PI);            // phi(x + h) - phi(x) = h (B + c3 (2 x + h))
    } else {
        E = cis_turns(fma(xd, r.c[1], fma(yd, r.c[2], r.c[0])) * INV2PI);
        D = (v2f){r.cr, r.ci};
    }
    // v_sin / v_cos results need a wait state before a non-transcendental VALU instruction reads them.  The compiler
    // inserts it for its own instructions but does not look into the inline-asm packed products that consume E and D
    // (seen: the first product of every spot read the stale register).  Pin one here.
    asm volatile("s_nop 1" : "+v"(E), "+v"(D));
}

// nearfield -> partial farfield sums.  grid = (run blocks, batch, spot chunks), block = 64
template <int DEG> __global__ __launch_bounds__(64) void c_n2f_run(CRunArgs ra) {
    const CArgs<float>& a = ra.a;
    const int b = blockIdx.y, lane = threadIdx.x;
    const int idx = blockIdx.x * 64 + lane;
    const int row = idx / ra.rpr, run = idx - row * ra.rpr;
    const bool live = row < ra.H;
    const int col0 = run * CR_RUN;
    const double xd = ra.x0 + (double)col0 * ra.hx, yd = ra.ys[live ? row : 0], q = 2.0 * xd + ra.hx;
    v2f nf[CR_RUN];
#pragma unroll
    for (int k = 0; k < CR_RUN; ++k) {
        nf[k] = (v2f){0.f, 0.f};
        if (live && col0 + k < ra.W) {
            const int p = row * ra.W + col0 + k;
            float ph = a.phase[(size_t)b * a.S + p];          // _build_nearfield :1000-1011
            if (a.kern) ph += a.kern[p];
            float s, c;
            Math<float>::sincos(ph, &s, &c);
            const float am = a.amp ? a.amp[p] : a.amp_scalar;
            nf[k] = (v2f){am * c, am * s};
        }
    }
    Cx<float>* out = a.partial + ((size_t)b * a.nblocks + blockIdx.x) * a.N;
    const int n_lo = blockIdx.z * ra.n_per, n_hi = min(a.N, n_lo + ra.n_per);
    v2f keep = (v2f){0.f, 0.f};
    for (int n = n_lo; n < n_hi; ++n) {
        const CRunRec r = ra.rec[n];
        v2f E, D;
        run_start<DEG>(r, xd, yd, q, ra.hx, E, D);
        const v2f Cn = (v2f){r.cr, r.ci};
        v2f acc = (v2f){0.f, 0.f};
#pragma unroll
        for (int k = 0; k < CR_RUN; ++k) {
            acc = cmac_conj(acc, nf[k], E);                  // nf * exp(-i phi)
            if (k + 1 < CR_RUN) {
                E = cmul(E, D);
                if constexpr (DEG == 2) D = cmul(D, Cn);
            }
        }
        acc = (v2f){wave_sum_all(acc.x), wave_sum_all(acc.y)};
        const int slot = (n - n_lo) & 63;
        if (lane == slot) keep = acc;
        if (slot == 63 || n == n_hi - 1) {                   // one 512-byte store per 64 spots
            const int nn = n - slot + lane;
            if (nn <= n) out[nn] = keep;
        }
    }
}

// farfield -> partial nearfields.  grid = (run blocks, batch, spot chunks), block = 64
template <int DEG> __global__ __launch_bounds__(64) void c_f2n_run(CRunArgs ra) {
    const CArgs<float>& a = ra.a;
    const int b = blockIdx.y, lane = threadIdx.x;
    const int idx = blockIdx.x * 64 + lane;
    const int row = idx / ra.rpr, run = idx - row * ra.rpr;
    const bool live = row < ra.H;
    const int col0 = run * CR_RUN;
    const double xd = ra.x0 + (double)col0 * ra.hx, yd = ra.ys[live ? row : 0], q = 2.0 * xd + ra.hx;
    v2f acc[CR_RUN];
#pragma unroll
    for (int k = 0; k < CR_RUN; ++k) acc[k] = (v2f){0.f, 0.f};
    const Cx<float>* ff = a.ff + (size_t)b * a.N;
    const int n_lo = blockIdx.z * ra.n_per, n_hi = min(a.N, n_lo + ra.n_per);
    for (int n = n_lo; n < n_hi; ++n) {
        const CRunRec r = ra.rec[n];
        v2f E, D;
        run_start<DEG>(r, xd, yd, q, ra.hx, E, D);
        const v2f Cn = (v2f){r.cr, r.ci};
        E = cmul(E, ff[n]);                                  // ff * exp(+i phi): the factor rides the recurrence
#pragma unroll
        for (int k = 0; k < CR_RUN; ++k) {
            acc[k] += E;
            if (k + 1 < CR_RUN) {
                E = cmul(E, D);
                if constexpr (DEG == 2) D = cmul(D, Cn);
            }
        }
    }
    if (live) {
        Cx<float>* dst = ra.nf_part + ((size_t)b * gridDim.z + blockIdx.z) * a.S + (size_t)row * ra.W + col0;
#pragma unroll
        for (int k = 0; k < CR_RUN; ++k)
            if (col0 + k < ra.W) dst[k] = acc[k];
    }
}

// nearfield = sum of the chunk partials (fixed order); phase or complex nearfield out.  grid = (ceil(S / 256), batch)
static __global__ void c_f2n_run_finish(CArgs<float> a, const Cx<float>* nf_part, int chunks) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.S) return;
    v2f s = (v2f){0.f, 0.f};
    for (int z = 0; z < chunks; ++z) s += nf_part[((size_t)b * chunks + z) * a.S + p];
    if (a.nf_out != nullptr) {                               // _farfield2nearfield(extract=False) (_spots.py:887-914)
        const float sc = Math<float>::rsqrt((float)a.S);
        a.nf_out[(size_t)b * a.S + p] = s * sc;
    } else {
        float ph = Math<float>::atan2(s.y, s.x);             // the 1/sqrt(S) scale does not change the phase (:1030)
        if (a.kern) ph -= a.kern[p];
        a.phase[(size_t)b * a.S + p] = ph;
    }
}

template <typename R> __global__ void convert_d2r(const double* in, R* out, int n, int batch) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        for (int b = 0; b < batch; ++b) out[(size_t)b * n + i] = (R)in[i];
}

}  // namespace hgs
