// Padded shapes that are not powers of two (the reference only warns about them, _hologram.py:378-384; e.g. a
// target given at the SLM shape 1152 x 1920): every centred 1-D transform of length N <= 4096 is evaluated
// with Bluestein's identity  n k = (n^2 + k^2 - (k - n)^2) / 2  as a circular convolution of length
// M = 2^p >= 2N - 1 on the workgroup transforms of fft_core.hpp:
//     X[k] = post[k] c[k] * IFFT_M( FFT_M( x[n] pre[n] c[n] ) . FFT_M( conj(c)[(m) wrapped] ) )[k],   c[n] = W^(n^2/2)
// One workgroup owns one line (a row of the SLM block or a column of the padded grid); the two forward and
// two inverse passes of fft2 / ifft2 are four launches of the same kernel with different tables and strides.
// The centring (fftshift . fft . fftshift, and ifftshift . ifft . ifftshift, which differ for odd N) and the
// ortho scale are folded into pre[] / post[]:
//     forward : pre[n] = W^(-n h),        post[k] = W^(h (k - h)) / sqrt(N),    W = exp(-2 pi i / N), h = N / 2 (floor)
//     inverse : pre[m] = conj(W)^(m h),   post[i] = conj(W)^(-h (i + h)) / sqrt(N)
// Tables are computed in double on the host (integer arguments reduced mod 2N before any floating point).
// An axis whose length IS a power of two (256 ... 16384) skips the convolution (BlueArgs::plain): pre[] / post[] alone.
// This path is functional, not tuned: it exists so that every shape the reference accepts runs on the GPU.
#pragma once
#include "kernels.hpp"

namespace hgs {

template <typename R> struct BlueArgs {
    const Cx<R>* in;          // input lines
    Cx<R>* out;               // output lines
    size_t in_line, out_line; // element stride between consecutive lines
    size_t in_batch, out_batch;
    int in_stride, out_stride;   // element stride inside a line
    int in_start, in_len;     // indices [in_start, in_start + in_len) of the length-N line exist in memory (rest is zero);
                              // element n sits at in[(n - in_start) * in_stride]
    int out_start, out_len;   // only these outputs are wanted; element k goes to out[(k - out_start) * out_stride]
    int N;                    // transform length
    const Cx<R>* A;           // [N]  pre[n] * c[n]
    const Cx<R>* Bf;          // [M]  FFT_M of the wrapped conj chirp, divided by M
    const Cx<R>* Cc;          // [N]  post[k] * c[k]
    const Cx<R>* tw;          // W_M table
    int plain;                // N == M, a power of two: no convolution.  1: out[k] = Cc[k] FFT_N(x A)[k] (forward);
                              // 2: the inverse direction through conj(FFT_N(conj(x A))) (A = pre, Cc = post / sqrt(N))
};

// grid = (lines, batch), block = M / 16
template <typename R, int M> __global__ __launch_bounds__(M / 16) void bluestein_lines(BlueArgs<R> a) {
    constexpr int T = M / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Cx<R>* lds = reinterpret_cast<Cx<R>*>(smem);
    using Sel = FftSel<R, M, false>;
    typename Sel::type fft;
    const int j = threadIdx.x;
    fft.init(a.tw, j);
    const int js = Sel::space_lane(j);
    const Cx<R>* in = a.in + (size_t)blockIdx.y * a.in_batch + (size_t)blockIdx.x * a.in_line;
    Cx<R>* out = a.out + (size_t)blockIdx.y * a.out_batch + (size_t)blockIdx.x * a.out_line;
    Cx<R> v[16];
    static_for<0, 16>([&](auto m_) {
        constexpr int m = m_;
        const int n = js + m * T;
        Cx<R> x = mk<R>(0, 0);
        if (n >= a.in_start && n < a.in_start + a.in_len) x = cmul(in[(size_t)(n - a.in_start) * a.in_stride], a.A[n]);
        if (a.plain == 2) x.y = -x.y;
        v[m] = x;
    });
    fft.fwd(v, lds, j);
    if (a.plain) {
        // the transform itself: lane j holds outputs j + m T of the frequency side
        static_for<0, 16>([&](auto m_) {
            constexpr int m = m_;
            const int k = j + m * T;
            if (k >= a.out_start && k < a.out_start + a.out_len) {
                Cx<R> y = v[m];
                if (a.plain == 2) y.y = -y.y;
                out[(size_t)(k - a.out_start) * a.out_stride] = cmul(y, a.Cc[k]);
            }
        });
        return;
    }
    static_for<0, 16>([&](auto m_) { constexpr int m = m_; v[m] = cmul(v[m], a.Bf[j + m * T]); });
    fft.inv_after_fwd(v, lds, j);
    static_for<0, 16>([&](auto m_) {
        constexpr int m = m_;
        const int k = js + m * T;
        if (k >= a.out_start && k < a.out_start + a.out_len)
            out[(size_t)(k - a.out_start) * a.out_stride] = cmul(v[m], a.Cc[k]);
    });
}

// ---- elementwise helpers of the general-size path ------------------------------------------------------
// nf = amp * exp(i (phase + kernel)) over the SLM block (_build_nearfield :1000-1011)
template <typename R>
__global__ void gen_build_nearfield(const R* phase, const R* amp, const R* kern, R amp_scalar, size_t S, Cx<R>* nf) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t b = blockIdx.y;
    if (i >= S) return;
    R p = phase[b * S + i];
    if (kern) p += kern[i];
    R s, c;
    Math<R>::sincos(p, &s, &c);
    const R am = amp ? amp[i] : amp_scalar;
    nf[b * S + i] = mk<R>(am * c, am * s);
}
// phase = atan2(nf) - kernel (_nearfield_extract :1026-1036)
template <typename R> __global__ void gen_extract_phase(const Cx<R>* nf, const R* kern, size_t S, R* phase) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t b = blockIdx.y;
    if (i >= S) return;
    const Cx<R> v = nf[b * S + i];
    R p = Math<R>::atan2(v.y, v.x);
    if (kern) p -= kern[i];
    phase[b * S + i] = p;
}
// amp_ff = |F| (and phase_ff = atan2 F), partial sums of |F|^2 (_midloop_cleaning :953, _populate_results :948)
template <typename R> __global__ void gen_amp_store(const Cx<R>* ff, R* amp_ff, R* pff, size_t P, double* partial) {
    __shared__ double scratch[16];
    const int b = blockIdx.y;
    double acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (size_t)gridDim.x * blockDim.x) {
        const Cx<R> F = ff[(size_t)b * P + i];
        const R p2 = F.x * F.x + F.y * F.y;
        amp_ff[(size_t)b * P + i] = Math<R>::sqrt(p2);
        if (pff) pff[(size_t)b * P + i] = Math<R>::atan2(F.y, F.x);
        acc += (double)p2;
    }
    const double s = block_sum(acc, scratch);
    if (threadIdx.x == 0) partial[(size_t)b * gridDim.x + blockIdx.x] = s;
}

// front-end (instantiations in launch_blue_f32.hip / launch_blue_f64.hip)
template <typename R> int launch_bluestein(int M, dim3 grid, hipStream_t s, const BlueArgs<R>& a);

}  // namespace hgs
