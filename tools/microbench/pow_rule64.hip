// Accuracy of the float64 rule's power and reciprocal square root on the device (kernels.hpp: pow_lean, rsqrt_full(double))
// against long double on the host.  Prints the largest and the mean error in ulp.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I slmsuite_amd/csrc -o tools/microbench/pow_rule64 tools/microbench/pow_rule64.hip
#include "kernels.hpp"
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void eval(const double* x, double* pw, double* rs, double c, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    pw[i] = hgs::pow_lean(x[i], c);
    rs[i] = hgs::rsqrt_full(x[i]);
}

int main() {
    const int per = 1 << 18;
    const double bands[][2] = {{-900, -600}, {-80, -20}, {-20, -3}, {-3, -0.1}, {-0.1, 0.1}, {0.1, 3}, {3, 20}, {20, 80}, {600, 900}};
    const int nb = sizeof bands / sizeof bands[0];
    std::vector<double> hx((size_t)nb * per);
    unsigned long long s = 88172645463325252ull;
    for (size_t i = 0; i < hx.size(); ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double u = (double)(s >> 11) / 9007199254740992.0;
        const int b = (int)(i / per);
        hx[i] = std::exp2(bands[b][0] + u * (bands[b][1] - bands[b][0]));
    }
    double *dx, *dp, *dr;
    hipMalloc(&dx, hx.size() * 8); hipMalloc(&dp, hx.size() * 8); hipMalloc(&dr, hx.size() * 8);
    hipMemcpy(dx, hx.data(), hx.size() * 8, hipMemcpyHostToDevice);
    std::vector<double> hp(hx.size()), hr(hx.size());
    double worst_all = 0;
    for (double p : {0.8, 0.5, 1.0, 0.3, 4.0}) {
        const double c = -0.5 * p;
        hipLaunchKernelGGL(eval, dim3((unsigned)((hx.size() + 255) / 256)), dim3(256), 0, 0, dx, dp, dr, c, (int)hx.size());
        hipMemcpy(hp.data(), dp, hx.size() * 8, hipMemcpyDeviceToHost);
        hipMemcpy(hr.data(), dr, hx.size() * 8, hipMemcpyDeviceToHost);
        printf("feedback_exponent %.1f (c = %.2f): error in ulp of the exact value, max / mean\n", p, c);
        for (int b = 0; b < nb; ++b) {
            double mp = 0, ap = 0, mr = 0, ar = 0;
            for (int k = 0; k < per; ++k) {
                const size_t i = (size_t)b * per + k;
                const long double ep = powl((long double)hx[i], (long double)c), er = 1.0L / sqrtl((long double)hx[i]);
                if (!(ep < 1.7e308L && ep > 2.3e-308L)) continue;          // outside the double range: inf / 0 is the right answer
                int ex;
                std::frexp((double)ep, &ex);
                const double up = std::ldexp(1.0, ex - 53);
                std::frexp((double)er, &ex);
                const double ur = std::ldexp(1.0, ex - 53);
                const double e1 = (double)fabsl((long double)hp[i] - ep) / up, e2 = (double)fabsl((long double)hr[i] - er) / ur;
                mp = std::fmax(mp, e1); ap += e1; mr = std::fmax(mr, e2); ar += e2;
            }
            printf("  x in 2^[%5.1f, %5.1f]: pow_lean %6.2f / %5.3f   rsqrt %5.2f / %5.3f\n", bands[b][0], bands[b][1], mp, ap / per, mr, ar / per);
            worst_all = std::fmax(worst_all, std::fmax(mp, mr));
        }
    }
    printf("worst %.2f ulp\n", worst_all);
    return worst_all < 8 ? 0 : 1;
}
