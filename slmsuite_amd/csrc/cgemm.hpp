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
// GAUSS: three real products per complex one -- P1 = Ar Br, P2 = Ai Bi, P3 = (Ar + Ai)(Br + Bi), Cr = P1 - P2,
// Ci = P3 - P1 - P2 -- i.e. 12 instead of 16 MFMAs per k pair and wave; the operand sums are two VALU adds per fetched
// fragment (the VALU is idle next to the matrix pipe), the three accumulator sets take 192 registers.
template <bool PADDED, bool GAUSS = false>
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

    // staging: thread t moves 8 consecutive complex values of row (t >> 4) of each operand tile.
    // (GAUSS: the two operands go one after the other through the same eight registers -- the third accumulator set
    //  leaves no room for both)
    const int sk = t >> 4, sc = (t & 15) * 8;
    float2 ra[8], rb[GAUSS ? 1 : 8];
    float2 rq[4];       // GAUSS: a quarter of the tile pair at a time (A low, A high, B low, B high halves of the 8 values)
    auto gload_one = [&](int k0, int which, float2 (&dst)[8]) {
        const int k = k0 + sk;
        const float2* src = which == 0 ? A + (size_t)k * a.lda + m0 + sc : B + (size_t)k * a.ldb + n0 + sc;
        if constexpr (PADDED) {
            const float4* p4 = reinterpret_cast<const float4*>(src);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 v = p4[i];
                dst[2 * i] = make_float2(v.x, v.y); dst[2 * i + 1] = make_float2(v.z, v.w);
            }
            return;
        }
        const bool kin = k < kend;
        const int lim = which == 0 ? a.M - m0 - sc : a.N - n0 - sc;
#pragma unroll
        for (int i = 0; i < 8; ++i) dst[i] = (kin && i < lim) ? src[i] : make_float2(0.f, 0.f);
    };
    auto lstore_one = [&](int buf, int which, const float2 (&v)[8]) {
        float* base = cg_lds + buf * BUF + sk * 128 + cg_swz(sk, sc) + (which == 0 ? 0 : 2 * PLANE);
        float4* re = reinterpret_cast<float4*>(base);
        float4* im = reinterpret_cast<float4*>(base + PLANE);
        re[0] = make_float4(v[0].x, v[1].x, v[2].x, v[3].x);
        re[1] = make_float4(v[4].x, v[5].x, v[6].x, v[7].x);
        im[0] = make_float4(v[0].y, v[1].y, v[2].y, v[3].y);
        im[1] = make_float4(v[4].y, v[5].y, v[6].y, v[7].y);
    };
    auto gload_q = [&](int k0, int q) {
        const int k = k0 + sk, which = q >> 1, off = (q & 1) * 4;
        const float2* src = (which == 0 ? A + (size_t)k * a.lda + m0 + sc : B + (size_t)k * a.ldb + n0 + sc) + off;
        if constexpr (PADDED) {
            const float4* p4 = reinterpret_cast<const float4*>(src);
            const float4 v0 = p4[0], v1 = p4[1];
            rq[0] = make_float2(v0.x, v0.y); rq[1] = make_float2(v0.z, v0.w);
            rq[2] = make_float2(v1.x, v1.y); rq[3] = make_float2(v1.z, v1.w);
            return;
        }
        const bool kin = k < kend;
        const int lim = (which == 0 ? a.M - m0 - sc : a.N - n0 - sc) - off;
#pragma unroll
        for (int i = 0; i < 4; ++i) rq[i] = (kin && i < lim) ? src[i] : make_float2(0.f, 0.f);
    };
    auto lstore_q = [&](int buf, int q) {
        const int which = q >> 1, off = (q & 1) * 4;
        float* base = cg_lds + buf * BUF + sk * 128 + cg_swz(sk, sc) + off + (which == 0 ? 0 : 2 * PLANE);
        *reinterpret_cast<float4*>(base) = make_float4(rq[0].x, rq[1].x, rq[2].x, rq[3].x);
        *reinterpret_cast<float4*>(base + PLANE) = make_float4(rq[0].y, rq[1].y, rq[2].y, rq[3].y);
    };
    auto gload = [&](int k0) {
        if constexpr (!GAUSS) { gload_one(k0, 0, ra); gload_one(k0, 1, rb); }
    };
    auto lstore = [&](int buf) {
        if constexpr (!GAUSS) { lstore_one(buf, 0, ra); lstore_one(buf, 1, rb); }
    };

    f32x16 cr[2][2], ci[2][2], cs[GAUSS ? 2 : 1][GAUSS ? 2 : 1];    // GAUSS: P1, P2, P3
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                cr[i][j][r] = 0.f;
                ci[i][j][r] = 0.f;
                if constexpr (GAUSS) cs[i][j][r] = 0.f;
            }

    // MFMA operand coordinates of this lane: A[i = lane & 31][k = lane >> 5], B[k][j = lane & 31]
    const int kq = lane >> 5, rr = lane & 31;
    const int nk = (kend - kbeg + CG_BK - 1) / CG_BK;
    if (nk > 0) {
        if constexpr (GAUSS) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { gload_q(kbeg, q); lstore_q(0, q); }
        } else {
            gload(kbeg);
            lstore(0);
        }
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
#ifndef CG_ABL
#define CG_ABL 0     // microbenchmark ablations: 1 = no global loads / LDS stores in the loop, 2 = also no barrier
#endif
        const bool more = CG_ABL == 0 && kt + 1 < nk;
        if (more) {
            if constexpr (GAUSS) gload_q(kbeg + (kt + 1) * CG_BK, 0);
            else gload(kbeg + (kt + 1) * CG_BK);
        }
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
            if constexpr (GAUSS) {
                if ((kk == 2 || kk == 4 || kk == 6) && more) {   // a quarter of the next tile pair parked, the next on its way
                    lstore_q(buf ^ 1, kk / 2 - 1);
                    gload_q(kbeg + (kt + 1) * CG_BK, kk / 2);
                }
            }
            if constexpr (GAUSS) {
                float as_[2], bs_[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) { as_[i] = ar[cur][i] + ai[cur][i]; bs_[i] = br[cur][i] + bi[cur][i]; }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        cr[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[cur][i], br[cur][jj], cr[i][jj], 0, 0, 0);
                        ci[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(ai[cur][i], bi[cur][jj], ci[i][jj], 0, 0, 0);
                        cs[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(as_[i], bs_[jj], cs[i][jj], 0, 0, 0);
                    }
            } else
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
        if (more) {
            if constexpr (GAUSS) lstore_q(buf ^ 1, 3);
            else lstore(buf ^ 1);
        }
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
                if (m < a.M && n < a.N) {
                    if constexpr (GAUSS) {
                        const float p1 = cr[i][j][r], p2 = ci[i][j][r];
                        C[(size_t)m * a.N + n] = make_float2(p1 - p2, cs[i][j][r] - p1 - p2);
                    } else {
                        C[(size_t)m * a.N + n] = make_float2(cr[i][j][r], ci[i][j][r]);
                    }
                }
            }
        }
}

// ---- stream-K form ------------------------------------------------------------------------------------------------
// The (output tile, k tile) iteration space is one line of tiles * KT steps; workgroup w of G = 2 * #CU takes the w-th
// G-th of it (to a step), whatever the shape -- 711 tiles x 120 steps or 135 x 625 both fill 512 resident workgroups
// exactly, where whole-tile splits ran 2.78 resp. 1.85 rounds.  A workgroup that ends a tile (or its range) inside it
// stores the partial sums to plane (w - first workgroup of that tile) of C; the consumers add the planes a tile has.
// Operands: PLANAR (a real and an imaginary [KT*16][ld] float array each), k-outer, zero padded to whole 16 x 128
// tiles -- so that a tile row is 512 contiguous bytes that go global -> LDS directly (global_load_lds_dwordx4: 1 KiB =
// two tile rows per wave instruction, wave v fills plane v of the next buffer; the bank swizzle of the odd rows is
// applied to the SOURCE column, the destination being linear by construction).  No staging registers, no ds_write
// pass, and the loads of step t+1 are in flight for the whole of step t (with the three accumulator sets of the Gauss
// form there were registers for a quarter tile at a time only, i.e. two k pairs of latency cover).
struct CgemmSkArgs {
    const float* Ar;   // [KT*16][lda]
    const float* Ai;
    const float* Br;   // [KT*16][ldb]
    const float* Bi;
    float2* C;         // [batch][planes][M][N]
    int M, N, KT;
    int lda, ldb;      // in floats
    int tiles_m, tiles_n, planes;
    const int* first_wg;   // [tiles_m * tiles_n] workgroup that owns step 0 of the tile (sk_owner)
    size_t strideA, strideB;   // per batch element, in floats (0 = shared)
    // EPI 1 (n2f): instead of storing C, contract it with E[m][n] over n on the spot:
    //   part[b][slot][m] = sum over the 64 columns of this wave of C[m][n] * E[m][n],  slot = (tile column * 2 + wave column) * planes + plane
    const float2* E;       // [M][ldE]
    int ldE;
    float2* part;          // [batch][tiles_n * 2 * planes][ldP]
    int ldP;
};
__host__ __device__ inline long long sk_begin(long long total, int G, int w) { return total * w / G; }
// workgroup whose range [sk_begin(w), sk_begin(w + 1)) holds step idx
__host__ __device__ inline int sk_owner(long long idx, long long total, int G) {
    long long w = idx * G / total;
    while (sk_begin(total, G, (int)w + 1) <= idx) ++w;
    while (sk_begin(total, G, (int)w) > idx) --w;
    return (int)w;
}

// sum over the 32 lanes of each half of the wave (row_shr 1, 2, 4, 8 inside the rows of 16, row_bcast:15 from row 0 to 1
// and from row 2 to 3): lanes 31 and 63 end up with the totals of their halves
template <int CTRL, int ROW_MASK = 0xf> __device__ __forceinline__ float cg_dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, true));
}
__device__ __forceinline__ float cg_half_sum(float v) {
    v = cg_dpp_add<0x111>(v);
    v = cg_dpp_add<0x112>(v);
    v = cg_dpp_add<0x114>(v);
    v = cg_dpp_add<0x118>(v);
    return cg_dpp_add<0x142, 0xa>(v);
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void cgemm_streamk(CgemmSkArgs a) {
    extern __shared__ __attribute__((aligned(16))) float cg_lds[];
    constexpr int PLANE = CG_BK * 128, BUF = 4 * PLANE;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int G = gridDim.x, w = blockIdx.x, b = blockIdx.y;
    const long long total = (long long)a.tiles_m * a.tiles_n * a.KT;
    const long long lo = sk_begin(total, G, w), hi = sk_begin(total, G, w + 1);
    if (lo >= hi) return;
    // this wave's plane of the tile pair: 0 = Ar, 1 = Ai, 2 = Br, 3 = Bi
    const float* plane = (wave == 0 ? a.Ar : wave == 1 ? a.Ai : wave == 2 ? a.Br : a.Bi) +
                         (size_t)b * (wave < 2 ? a.strideA : a.strideB);
    const int ld = wave < 2 ? a.lda : a.ldb;
    // lane -> (row parity, 16-byte chunk) of a two-row piece; odd rows take the swizzled source chunk
    const int lrow = lane >> 5, lchunk = (lane & 31) ^ (lrow << 3);
    auto gload = [&](int buf, int m0, int n0, int k0) {
        const float* src = plane + (size_t)(k0 + lrow) * ld + (wave < 2 ? m0 : n0) + 4 * lchunk;
        float* dst = cg_lds + buf * BUF + wave * PLANE;
#pragma unroll
        for (int i = 0; i < CG_BK / 2; ++i)
            __builtin_amdgcn_global_load_lds(src + (size_t)(2 * i) * ld,
                                             (__attribute__((address_space(3))) void*)(dst + 2 * i * 128), 16, 0, 0);
    };
    f32x16 c1[2][2], c2[2][2], c3[2][2];
    auto clear = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) { c1[i][j][r] = 0.f; c2[i][j][r] = 0.f; c3[i][j][r] = 0.f; }
    };
    clear();
    const int kq = lane >> 5, rr = lane & 31;
    int tile = (int)(lo / a.KT), kt = (int)(lo - (long long)tile * a.KT);
    int m0 = (tile % a.tiles_m) * CG_BM, n0 = (tile / a.tiles_m) * CG_BN;
    gload(0, m0, n0, kt * CG_BK);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0;
    for (long long it = lo; it < hi; ++it) {
        const bool more = it + 1 < hi;
        int ntile = tile, nkt = kt + 1;
        if (nkt == a.KT) { nkt = 0; ntile = tile + 1; }
        const int nm0 = (ntile % a.tiles_m) * CG_BM, nn0 = (ntile / a.tiles_m) * CG_BN;
        if (more) gload(buf ^ 1, nm0, nn0, nkt * CG_BK);      // (every wave left buf ^ 1 at the barrier of the step before)
        const float* L = cg_lds + buf * BUF;
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
            float as_[2], bs_[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) { as_[i] = ar[cur][i] + ai[cur][i]; bs_[i] = br[cur][i] + bi[cur][i]; }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    c1[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[cur][i], br[cur][jj], c1[i][jj], 0, 0, 0);
                    c2[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(ai[cur][i], bi[cur][jj], c2[i][jj], 0, 0, 0);
                    c3[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(as_[i], bs_[jj], c3[i][jj], 0, 0, 0);
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the next step's tile pair has landed
        __syncthreads();
        if (nkt == 0 || !more) {     // this workgroup's share of `tile` is complete
            const int seg = w - a.first_wg[tile];
            if constexpr (EPI == 1) {
                // y contraction of n2f on the accumulators: per row m the products with E[m][n] summed over this wave's 64
                // columns (two register tiles, then the 32 lanes of the half wave that shares the row)
                const int slot = ((n0 / CG_BN) * 2 + wn) * a.planes + seg;
                float2* dst = a.part + ((size_t)b * a.tiles_n * 2 * a.planes + slot) * (size_t)a.ldP;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq;
                        float sr = 0.f, si = 0.f;
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int n = n0 + wn * 64 + j * 32 + rr;
                            float2 e = make_float2(0.f, 0.f);
                            if (m < a.M && n < a.N) e = a.E[(size_t)m * a.ldE + n];
                            const float p1 = c1[i][j][r], p2 = c2[i][j][r];
                            const float tr = p1 - p2, ti = c3[i][j][r] - p1 - p2;
                            sr += tr * e.x - ti * e.y;
                            si += tr * e.y + ti * e.x;
                        }
                        sr = cg_half_sum(sr);
                        si = cg_half_sum(si);
                        if (rr == 31 && m < a.M) dst[m] = make_float2(sr, si);
                    }
            } else {
            float2* C = a.C + ((size_t)b * a.planes + seg) * (size_t)a.M * a.N;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int n = n0 + wn * 64 + j * 32 + rr;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq;
                        if (m < a.M && n < a.N) {
                            const float p1 = c1[i][j][r], p2 = c2[i][j][r];
                            C[(size_t)m * a.N + n] = make_float2(p1 - p2, c3[i][j][r] - p1 - p2);
                        }
                    }
                }
            }
            clear();
        }
        tile = ntile; kt = nkt; m0 = nm0; n0 = nn0; buf ^= 1;
    }
}

constexpr size_t CG_LDS_BYTES = 2 * 4 * CG_BK * 128 * sizeof(float);   // 64 KB

}  // namespace hgs
