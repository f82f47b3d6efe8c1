// Memory-only floors of the two hot kernels: same addresses, grids and block shapes as
// row_kernel<MODE 2> and col_tile_kernel at cfg 2, but no transforms.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int Sh = 1152, Sw = 1920, Ph = 4096, Pw = 4096, T = 256, R0 = 1472, C0 = 1088;

// row pattern: read 16 x 8 B per lane from the [ct][row][4] layout (32-B pieces), write them back + phase row
__global__ __launch_bounds__(256) void row_copy(v2f* gh, float* phase, int xcd_map) {
    const int j = threadIdx.x;
    int r = blockIdx.x;
    if (xcd_map) { const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3; r = 4 * ((idx >> 2) * 8 + xcd) + (idx & 3); }
    if (r >= Sh) return;
    const unsigned lane = (unsigned)(j >> 2) * Sh * 4u + (unsigned)(j & 3);
    v2f* ghr = gh + (size_t)r * 4;
    v2f v[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) v[m] = (ghr + (size_t)m * T * Sh)[lane];
    float acc = 0;
#pragma unroll
    for (int m = 0; m < 16; ++m) { acc += v[m].x; v[m] = v[m] * 1.0001f; }
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const int c = j + m * T - C0;
        if (c >= 0 && c < Sw) phase[(size_t)r * Sw + c] = acc + m;
    }
#pragma unroll
    for (int m = 0; m < 16; ++m) (ghr + (size_t)m * T * Sh)[lane] = v[m];
}
// the tile-layout pattern with the tile stride padded to ShP rows (Sh * 32 B = 9 * 4096 B puts the 16 pieces of one
// wave access 36,864 B apart: same channel under a 4 KiB interleave)
__global__ __launch_bounds__(256) void row_copy_pad(v2f* gh, float* phase, int ShP) {
    const int j = threadIdx.x;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int r = 4 * ((idx >> 2) * 8 + xcd) + (idx & 3);
    if (r >= Sh) return;
    const unsigned lane = (unsigned)(j >> 2) * ShP * 4u + (unsigned)(j & 3);
    v2f* ghr = gh + (size_t)r * 4;
    v2f v[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) v[m] = (ghr + (size_t)m * T * ShP)[lane];
    float acc = 0;
#pragma unroll
    for (int m = 0; m < 16; ++m) { acc += v[m].x; v[m] = v[m] * 1.0001f; }
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const int c = j + m * T - C0;
        if (c >= 0 && c < Sw) phase[(size_t)r * Sw + c] = acc + m;
    }
#pragma unroll
    for (int m = 0; m < 16; ++m) (ghr + (size_t)m * T * ShP)[lane] = v[m];
}
// same traffic, but each row is one contiguous 32 KiB run (what a row-major GH would give)
__global__ __launch_bounds__(256) void row_copy_linear(v2f* gh, float* phase) {
    const int j = threadIdx.x, r = blockIdx.x;
    v2f* ghr = gh + (size_t)r * Pw;
    v2f v[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) v[m] = ghr[j + m * T];
    float acc = 0;
#pragma unroll
    for (int m = 0; m < 16; ++m) { acc += v[m].x; v[m] = v[m] * 1.0001f; }
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const int c = j + m * T - C0;
        if (c >= 0 && c < Sw) phase[(size_t)r * Sw + c] = acc + m;
    }
#pragma unroll
    for (int m = 0; m < 16; ++m) ghr[j + m * T] = v[m];
}
// column-tile pattern: 6 x 32 B per lane tile load/store + 4 columns x (16 w + 16 t) loads
__global__ __launch_bounds__(256) void col_copy(v2f* gh, const float* w, const float* t, float* sink, int tiles_per_wg) {
    const int j = threadIdx.x;
    float acc = 0;
    for (int k = 0; k < tiles_per_wg; ++k) {
        const int ct = blockIdx.x + k * gridDim.x;
        v2f* g = gh + (size_t)ct * Sh * 4;
        float4 a[6], b[6];
#pragma unroll
        for (int m = 0; m < 6; ++m) {
            const int r = j + (m + 5) * T - R0;
            a[m] = b[m] = make_float4(0, 0, 0, 0);
            if (r >= 0 && r < Sh) { const float4* q = (const float4*)(g + (unsigned)r * 4u); a[m] = q[0]; b[m] = q[1]; }
        }
        for (int c = 0; c < 4; ++c) {
            const float* wc = w + (size_t)(ct * 4 + c) * Ph;
            const float* tc = t + (size_t)(ct * 4 + c) * Ph;
#pragma unroll
            for (int m = 0; m < 16; ++m) acc += wc[j + m * T] + tc[j + m * T];
        }
#pragma unroll
        for (int m = 0; m < 6; ++m) {
            const int r = j + (m + 5) * T - R0;
            if (r >= 0 && r < Sh) { float4* q = (float4*)(g + (unsigned)r * 4u); a[m].x += acc * 1e-30f; q[0] = a[m]; q[1] = b[m]; }
        }
    }
    if (acc == 12345.f) sink[0] = acc;
}
template <typename F> float timeit(F f, int reps = 20) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipDeviceSynchronize(); hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps * 1e3f;
}
int main() {
    v2f* gh; float *phase, *w, *t, *sink;
    hipMalloc(&gh, (size_t)1280 * Pw * 8); hipMalloc(&phase, (size_t)Sh * Sw * 4);
    hipMalloc(&w, (size_t)Ph * Pw * 4); hipMalloc(&t, (size_t)Ph * Pw * 4); hipMalloc(&sink, 64);
    hipMemset(gh, 0, (size_t)Sh * Pw * 8); hipMemset(w, 0, (size_t)Ph * Pw * 4); hipMemset(t, 0, (size_t)Ph * Pw * 4);
    printf("row_copy (tile layout, xcd map)   %.1f us\n", timeit([&] { hipLaunchKernelGGL(row_copy, dim3(1152), dim3(256), 0, 0, gh, phase, 1); }));
    printf("row_copy (tile layout, no map)    %.1f us\n", timeit([&] { hipLaunchKernelGGL(row_copy, dim3(1152), dim3(256), 0, 0, gh, phase, 0); }));
    for (int pad : {1152, 1156, 1160, 1168, 1184, 1216, 1280})
        printf("row_copy, tile stride %4d rows     %.1f us\n", pad, timeit([&] { hipLaunchKernelGGL(row_copy_pad, dim3(1152), dim3(256), 0, 0, gh, phase, pad); }));
    printf("row_copy_linear (row-major)       %.1f us\n", timeit([&] { hipLaunchKernelGGL(row_copy_linear, dim3(1152), dim3(256), 0, 0, gh, phase); }));
    printf("col_copy 512 WG x 2 tiles         %.1f us\n", timeit([&] { hipLaunchKernelGGL(col_copy, dim3(512), dim3(256), 0, 0, gh, w, t, sink, 2); }));
    printf("col_copy 1024 WG x 1 tile         %.1f us\n", timeit([&] { hipLaunchKernelGGL(col_copy, dim3(1024), dim3(256), 0, 0, gh, w, t, sink, 1); }));
    printf("bytes: row %.1f MB, col %.1f MB\n", (2.0 * Sh * Pw * 8 + Sh * Sw * 4) / 1e6, (2.0 * Sh * Pw * 8 + 2.0 * Ph * Pw * 4) / 1e6);
    return 0;
}
