// Complex fp32 GEMM on the matrix cores (v_mfma_f32_32x32x2_f32: exact f32, 157 TFLOP/s peak) for the
// separable form of the CompressedSpotHologram transforms (compressed_sep.hpp):
//
//     C[s][m][n] = sum_{k in split s} A[k][m] * B[k][n]          (complex, interleaved re/im)
//
// Both operands are "k-outer" (row k holds M resp. N contiguous complex values), so tiles stream in
// with fully coalesced rows and the MFMA operand fetch from LDS is conflict-free.
//   block tile 128 x 128 x 16, 256 threads = 4 waves, each wave a 64 x 64 complex sub-tile
//   = 2 x 2 MFMA tiles of 32 x 32, real and imaginary accumulators (128 VGPRs);
//   Cr += Ar Br - Ai Bi, Ci += Ar Bi + Ai Br: four MFMAs per 32x32x2 complex product;
//   LDS: planar Ar/Ai/Br/Bi [16][128] floats per buffer, double buffered (64 KB), the two k rows a
//   wave reads at once are swizzled (col ^ 32 on odd k) onto disjoint banks;
//   the global loads of tile t+1 are in flight while tile t is multiplied.
#pragma once
#include <hip/hip_runtime.h>

namespace hgs {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct CgemmArgs {
    const float2* A;   // [K][lda]
    const float2* B;   // [K][ldb]
    float2* C;         // [batch][split][M][N]
    int M, N, K;
    int lda, ldb;
    int split;         // K is cut into `split` contiguous ranges of k_per (multiple of 16) rows
    int k_per;
    size_t strideA, strideB;   // per batch element (0 = shared)
};

constexpr int CG_BM = 128, CG_BN = 128, CG_BK = 16;

__device__ __forceinline__ int cg_swz(int k, int c) { return c ^ ((k & 1) << 5); }

// PADDED: the caller guarantees that every 16 x 128 tile the grid touches is addressable and that the
// padding is zero (lda >= 128-multiple of M, ldb likewise, split * k_per rows) -- no bounds checks, 32-byte
// loads.  The staging (global -> registers -> planar LDS) is 30 % of the kernel time otherwise.
template <bool PADDED>
__global__ __launch_bounds__(256, 2) void cgemm_kouter(CgemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) float cg_lds[];
    // buffer layout: [buf][plane (Ar, Ai, Br, Bi)][16][128]
    constexpr int PLANE = CG_BK * 128, BUF = 4 * PLANE;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * CG_BM, n0 = blockIdx.y * CG_BN;
    const int s = blockIdx.z % a.split, b = blockIdx.z / a.split;
    const int kbeg = s * a.k_per, kend = PADDED ? kbeg + a.k_per : min(a.K, kbeg + a.k_per);
    const float2* A = a.A + (size_t)b * a.strideA;
    const float2* B = a.B + (size_t)b * a.strideB;

    // staging: thread t moves 8 consecutive complex values of row (t >> 4) of each operand tile
    const int sk = t >> 4, sc = (t & 15) * 8;
    float2 ra[8], rb[8];
    auto gload = [&](int k0) {
        const int k = k0 + sk;
        if constexpr (PADDED) {
            const float4* pa = reinterpret_cast<const float4*>(A + (size_t)k * a.lda + m0 + sc);
            const float4* pb = reinterpret_cast<const float4*>(B + (size_t)k * a.ldb + n0 + sc);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 va = pa[i], vb = pb[i];
                ra[2 * i] = make_float2(va.x, va.y); ra[2 * i + 1] = make_float2(va.z, va.w);
                rb[2 * i] = make_float2(vb.x, vb.y); rb[2 * i + 1] = make_float2(vb.z, vb.w);
            }
            return;
        }
        const bool kin = k < kend;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = m0 + sc + i, n = n0 + sc + i;
            ra[i] = (kin && m < a.M) ? A[(size_t)k * a.lda + m] : make_float2(0.f, 0.f);
            rb[i] = (kin && n < a.N) ? B[(size_t)k * a.ldb + n] : make_float2(0.f, 0.f);
        }
    };
    auto lstore = [&](int buf) {
        float* base = cg_lds + buf * BUF + sk * 128 + cg_swz(sk, sc);
        float4* ar = reinterpret_cast<float4*>(base);
        float4* ai = reinterpret_cast<float4*>(base + PLANE);
        float4* br = reinterpret_cast<float4*>(base + 2 * PLANE);
        float4* bi = reinterpret_cast<float4*>(base + 3 * PLANE);
        ar[0] = make_float4(ra[0].x, ra[1].x, ra[2].x, ra[3].x);
        ar[1] = make_float4(ra[4].x, ra[5].x, ra[6].x, ra[7].x);
        ai[0] = make_float4(ra[0].y, ra[1].y, ra[2].y, ra[3].y);
        ai[1] = make_float4(ra[4].y, ra[5].y, ra[6].y, ra[7].y);
        br[0] = make_float4(rb[0].x, rb[1].x, rb[2].x, rb[3].x);
        br[1] = make_float4(rb[4].x, rb[5].x, rb[6].x, rb[7].x);
        bi[0] = make_float4(rb[0].y, rb[1].y, rb[2].y, rb[3].y);
        bi[1] = make_float4(rb[4].y, rb[5].y, rb[6].y, rb[7].y);
    };

    f32x16 cr[2][2], ci[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { cr[i][j][r] = 0.f; ci[i][j][r] = 0.f; }

    // MFMA operand coordinates of this lane: A[i = lane & 31][k = lane >> 5], B[k][j = lane & 31]
    const int kq = lane >> 5, rr = lane & 31;
    const int nk = (kend - kbeg + CG_BK - 1) / CG_BK;
    if (nk > 0) {
        gload(kbeg);
        lstore(0);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
#ifndef CG_ABL
#define CG_ABL 0     // microbenchmark ablations: 1 = no global loads / LDS stores in the loop, 2 = also no barrier
#endif
        if (CG_ABL == 0 && kt + 1 < nk) gload(kbeg + (kt + 1) * CG_BK);
        const float* L = cg_lds + buf * BUF;
        // operand fragments of k-pair kk+1 are fetched from LDS before the MFMAs of pair kk issue
        float ar[2][2], ai[2][2], br[2][2], bi[2][2];
        auto fetch = [&](int kk, int slot) {
            const int k = 2 * kk + kq;
            const float* row = L + k * 128;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = cg_swz(k, wm * 64 + i * 32 + rr);
                ar[slot][i] = row[c];
                ai[slot][i] = row[PLANE + c];
            }
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int c = cg_swz(k, wn * 64 + jj * 32 + rr);
                br[slot][jj] = row[2 * PLANE + c];
                bi[slot][jj] = row[3 * PLANE + c];
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int kk = 0; kk < CG_BK / 2; ++kk) {
            const int cur = kk & 1;
            if (kk + 1 < CG_BK / 2) fetch(kk + 1, cur ^ 1);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    cr[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[cur][i], br[cur][jj], cr[i][jj], 0, 0, 0);
                    ci[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[cur][i], bi[cur][jj], ci[i][jj], 0, 0, 0);
                    cr[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(-ai[cur][i], bi[cur][jj], cr[i][jj], 0, 0, 0);
                    ci[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(ai[cur][i], br[cur][jj], ci[i][jj], 0, 0, 0);
                }
        }
        if (CG_ABL == 0 && kt + 1 < nk) lstore(buf ^ 1);
        if (CG_ABL < 2) __syncthreads();
    }

    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float2* C = a.C + ((size_t)b * a.split + s) * (size_t)a.M * a.N;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + rr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq;
                if (m < a.M && n < a.N) C[(size_t)m * a.N + n] = make_float2(cr[i][j][r], ci[i][j][r]);
            }
        }
}

constexpr size_t CG_LDS_BYTES = 2 * 4 * CG_BK * 128 * sizeof(float);   // 64 KB

}  // namespace hgs
