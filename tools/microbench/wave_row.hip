// One-wave 4096-point transform (csrc/wave_fft.hpp, row_wave.hpp) against the workgroup row kernel at cfg 2:
// same G -> H' for the three modes, launch time of both.
#include "row_wave.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
using namespace hgs;
template <typename F> float timeit(F f, int reps = 50) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) f();
    hipDeviceSynchronize(); hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps * 1e3f;
}
static double rel(const std::vector<v2f>& x, const std::vector<v2f>& y) {
    double n = 0, d = 0;
    for (size_t i = 0; i < x.size(); ++i) {
        const double ex = (double)x[i].x - y[i].x, ey = (double)x[i].y - y[i].y;
        n += ex * ex + ey * ey; d += (double)y[i].x * y[i].x + (double)y[i].y * y[i].y;
    }
    return std::sqrt(n / d);
}
int main(int argc, char** argv) {
    const int SH = argc > 1 ? atoi(argv[1]) : 1152;   // more rows than cfg 2: steady-state throughput per row
    Geo g{4096, 4096, SH, 1920, (4096 - SH) / 2, 1088, 1, 256};
    const size_t S = (size_t)g.Sh * g.Sw, GN = (size_t)g.Sh * g.Pw;
    float *phase, *phase2, *wscale; v2f *gh, *gh2, *gh0, *tw;
    hipMalloc(&phase, S * 4); hipMalloc(&phase2, S * 4); hipMalloc(&gh, GN * 8); hipMalloc(&gh2, GN * 8); hipMalloc(&gh0, GN * 8);
    hipMalloc(&tw, 4096 * 8); hipMalloc(&wscale, 4);
    std::vector<v2f> htw(4096);
    for (int i = 0; i < 4096; ++i) htw[i] = (v2f){(float)cos(-2 * M_PI * i / 4096), (float)sin(-2 * M_PI * i / 4096)};
    hipMemcpy(tw, htw.data(), 4096 * 8, hipMemcpyHostToDevice);
    std::vector<float> hp(S);
    unsigned s = 12345;
    for (size_t i = 0; i < S; ++i) { s = s * 1664525u + 1013904223u; hp[i] = (float)((s >> 8) * (6.283185307 / 16777216.0) - 3.14159265); }
    float one = 1.f;
    hipMemcpy(phase, hp.data(), S * 4, hipMemcpyHostToDevice); hipMemcpy(wscale, &one, 4, hipMemcpyHostToDevice);
    RowArgs<float> ra{}; ra.g = g; ra.phase = phase; ra.amp_scalar = 1e-3f; ra.gh = gh; ra.tw = tw; ra.scale = 1.f / 64; ra.wscale = wscale; ra.xcd_map = 1;
    RowArgs<float> rb = ra; rb.gh = gh2; rb.phase = phase2;
    const size_t lds = lds_elems<4096>() * 8 + 128, wlds = WaveFft4096::LDS_BYTES;
    std::vector<v2f> x(GN), y(GN);
    std::vector<float> px(S), py(S);
    // MODE 0
    hipLaunchKernelGGL((row_kernel<float, 4096, 0>), dim3(SH), dim3(256), lds, 0, ra);
    hipMemcpy(phase2, phase, S * 4, hipMemcpyDeviceToDevice);
    hipLaunchKernelGGL((row_wave_kernel<0, 16, 48, false>), dim3(SH), dim3(64), wlds, 0, rb);
    hipDeviceSynchronize();
    printf("launch status: %s\n", hipGetErrorString(hipGetLastError()));
    hipMemcpy(x.data(), gh2, GN * 8, hipMemcpyDeviceToHost); hipMemcpy(y.data(), gh, GN * 8, hipMemcpyDeviceToHost);
    printf("MODE 0: rel L2 (wave vs workgroup) %.3e\n", rel(x, y));
    for (int k : {0, 1, 2, 3, 4, 5, 64, 65, 256, 1000, 4095}) {
        const size_t i = ((size_t)(k >> 2) * g.Sh + 7) * 4 + (k & 3);
        printf("  row 7 k %4d: wave (% .5e, % .5e)  workgroup (% .5e, % .5e)\n", k, x[i].x, x[i].y, y[i].x, y[i].y);
    }
    hipMemcpy(gh0, gh, GN * 8, hipMemcpyDeviceToDevice);
    // MODE 2 from the same H
    hipMemcpy(gh2, gh0, GN * 8, hipMemcpyDeviceToDevice);
    hipLaunchKernelGGL((row_kernel<float, 4096, 2>), dim3(SH), dim3(256), lds, 0, ra);
    hipLaunchKernelGGL((row_wave_kernel<2, 16, 48, false>), dim3(SH), dim3(64), wlds, 0, rb);
    hipDeviceSynchronize();
    hipMemcpy(x.data(), gh2, GN * 8, hipMemcpyDeviceToHost); hipMemcpy(y.data(), gh, GN * 8, hipMemcpyDeviceToHost);
    printf("MODE 2: rel L2 %.3e\n", rel(x, y));
    // MODE 1
    hipMemcpy(gh, gh0, GN * 8, hipMemcpyDeviceToDevice); hipMemcpy(gh2, gh0, GN * 8, hipMemcpyDeviceToDevice);
    hipMemset(phase2, 0x7f, S * 4);
    hipLaunchKernelGGL((row_kernel<float, 4096, 1>), dim3(SH), dim3(256), lds, 0, ra);
    hipLaunchKernelGGL((row_wave_kernel<1, 16, 48, false>), dim3(SH), dim3(64), wlds, 0, rb);
    hipDeviceSynchronize();
    hipMemcpy(px.data(), phase2, S * 4, hipMemcpyDeviceToHost); hipMemcpy(py.data(), phase, S * 4, hipMemcpyDeviceToHost);
    double md = 0, ref = 0;
    for (size_t i = 0; i < S; ++i) { double d = fabs((double)px[i] - py[i]); if (d > 3.14159) d = fabs(d - 6.283185307); md = fmax(md, d); ref += fabs(hp[i] - py[i]) < 1e-3; }
    printf("MODE 1: max |dphase| %.3e   (phases back at the input within 1e-3: %.1f %%)\n", md, 100.0 * ref / S);
    hipMemcpy(gh, gh0, GN * 8, hipMemcpyDeviceToDevice); hipMemcpy(gh2, gh0, GN * 8, hipMemcpyDeviceToDevice);
    for (int grid : {32, 256, 512, 768, 1024, SH}) {
        printf("rows %4d: workgroup row<2> %.1f us   wave row<2> %.1f us\n", grid,
               timeit([&] { hipLaunchKernelGGL((row_kernel<float, 4096, 2>), dim3(grid), dim3(256), lds, 0, ra); }),
               timeit([&] { hipLaunchKernelGGL((row_wave_kernel<2, 16, 48, false>), dim3(grid), dim3(64), wlds, 0, rb); }));
    }
    printf("workgroup row<0> %.1f  row<1> %.1f  row<2> %.1f us\n",
           timeit([&] { hipLaunchKernelGGL((row_kernel<float, 4096, 0>), dim3(SH), dim3(256), lds, 0, ra); }),
           timeit([&] { hipLaunchKernelGGL((row_kernel<float, 4096, 1>), dim3(SH), dim3(256), lds, 0, ra); }),
           timeit([&] { hipLaunchKernelGGL((row_kernel<float, 4096, 2>), dim3(SH), dim3(256), lds, 0, ra); }));
    printf("wave      row<0> %.1f  row<1> %.1f  row<2> %.1f us\n",
           timeit([&] { hipLaunchKernelGGL((row_wave_kernel<0, 16, 48, false>), dim3(SH), dim3(64), wlds, 0, rb); }),
           timeit([&] { hipLaunchKernelGGL((row_wave_kernel<1, 16, 48, false>), dim3(SH), dim3(64), wlds, 0, rb); }),
           timeit([&] { hipLaunchKernelGGL((row_wave_kernel<2, 16, 48, false>), dim3(SH), dim3(64), wlds, 0, rb); }));
    {   // compute only: every column masked off (loads give zero, stores are dropped)
        unsigned short* zm; hipMalloc(&zm, 256 * 2); hipMemset(zm, 0, 256 * 2);
        RowArgs<float> rc = rb; rc.load_mask = zm; rc.store_mask = zm;
        printf("wave row<2>, all columns masked (no GH traffic): %.1f us\n",
               timeit([&] { hipLaunchKernelGGL((row_wave_kernel<2, 16, 48, true>), dim3(SH), dim3(64), wlds, 0, rc); }));
        RowArgs<float> rd = ra; rd.load_mask = zm; rd.store_mask = zm;
        printf("workgroup row<2>, all columns masked:            %.1f us\n",
               timeit([&] { hipLaunchKernelGGL((row_kernel<float, 4096, 2>), dim3(SH), dim3(256), lds, 0, rd); }));
    }
    rb.xcd_map = 0;
    printf("wave, no XCD map: row<2> %.1f us\n",
           timeit([&] { hipLaunchKernelGGL((row_wave_kernel<2, 16, 48, false>), dim3(SH), dim3(64), wlds, 0, rb); }));
    return 0;
}
