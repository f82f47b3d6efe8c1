// Ablation of the two hot kernels at cfg 2 (compile with -DHGS_ABL_TRANS/-DHGS_ABL_XCHG/-DHGS_ABL_BFLY):
// which part of the time is transcendental math, LDS exchange + barriers, butterflies?
#include "../../slmsuite_amd/csrc/kernels.hpp"
#include <cstdio>
#include <vector>
#include <cmath>
using namespace hgs;
template <typename F> float timeit(F f, int reps = 30) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipDeviceSynchronize(); hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps * 1e3f;
}
int main() {
    Geo g{4096, 4096, 1152, 1920, 1472, 1088, 1, 256};
    const size_t S = (size_t)g.Sh * g.Sw, P = (size_t)g.Ph * g.Pw;
    float *phase, *w, *t, *wscale; v2f *gh, *tw; double* wp;
    hipMalloc(&phase, S * 4); hipMalloc(&w, P * 4); hipMalloc(&t, P * 4); hipMalloc(&gh, (size_t)g.Sh * g.Pw * 8);
    hipMalloc(&tw, 4096 * 8); hipMalloc(&wscale, 4); hipMalloc(&wp, 4096 * 8);
    std::vector<v2f> htw(4096);
    for (int i = 0; i < 4096; ++i) htw[i] = (v2f){(float)cos(-2 * M_PI * i / 4096), (float)sin(-2 * M_PI * i / 4096)};
    hipMemcpy(tw, htw.data(), 4096 * 8, hipMemcpyHostToDevice);
    std::vector<float> hw(P, 1e-3f), ht(P, 0.f), hp(S, 0.3f);
    for (size_t i = 0; i < P; i += 4097) ht[i] = 0.03f;
    float one = 1.f;
    hipMemcpy(w, hw.data(), P * 4, hipMemcpyHostToDevice); hipMemcpy(t, ht.data(), P * 4, hipMemcpyHostToDevice);
    hipMemcpy(phase, hp.data(), S * 4, hipMemcpyHostToDevice); hipMemcpy(wscale, &one, 4, hipMemcpyHostToDevice);
    RowArgs<float> ra{}; ra.g = g; ra.phase = phase; ra.amp_scalar = 1e-3f; ra.gh = gh; ra.tw = tw; ra.scale = 1.f / 64; ra.wscale = wscale; ra.xcd_map = 1;
    ColArgs<float> ca{}; ca.g = g; ca.gh = gh; ca.w = w; ca.t = t; ca.wscale = wscale; ca.wpartial = wp; ca.tw = tw; ca.scale = 1.f / 64;
    ca.cp.method = M_LEONARDO; ca.cp.do_update = 1; ca.cp.p_exp = 0.8f; ca.cp.inv_fnorm = 1.f; ca.cp.log2_inv_fnorm = 0.f;
    const size_t lds = lds_elems<4096>() * 8 + 128, tlds = col_tile_lds_bytes<float, 4096>();
    hipLaunchKernelGGL((row_kernel<float, 4096, 0>), dim3(1152), dim3(256), lds, 0, ra);
    printf("ABL trans=%d xchg=%d bfly=%d : row<2> %.1f us   col_tile %.1f us\n", HGS_ABL_TRANS, HGS_ABL_XCHG, HGS_ABL_BFLY,
           timeit([&] { hipLaunchKernelGGL((row_kernel<float, 4096, 2>), dim3(1152), dim3(256), lds, 0, ra); }),
           timeit([&] { hipLaunchKernelGGL((col_tile_kernel<float, 4096, 0, 6>), dim3(512), dim3(256), tlds, 0, ca, 5 * 256); }));
    return 0;
}
