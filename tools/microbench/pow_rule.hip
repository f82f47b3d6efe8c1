// Accuracy of the weight rule's power x^c (c = -p/2) on the device: the plain form exp2(c * log2(x)) on v_log_f32 /
// v_exp_f32 against pow_split (kernels.hpp) -- both against double pow rounded to float.  Prints, per decade band of x,
// the largest and the mean error in units of the last place of the exact result.
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench/pow_rule tools/microbench/pow_rule.hip && tools/microbench/pow_rule
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__device__ __forceinline__ float pow_split(float x, float c) {
    const float m = __builtin_amdgcn_frexp_mantf(x);
    const float e = (float)__builtin_amdgcn_frexp_expf(x);
    const float l = __builtin_amdgcn_logf(m);
    const float n = __builtin_rintf(c * e);
    const float f = __builtin_fmaf(c, e, -n) + c * l;
    return __builtin_amdgcn_ldexpf(__builtin_amdgcn_exp2f(f), (int)n);
}
__global__ void eval(const float* x, float* plain, float* split, float c, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    plain[i] = __builtin_amdgcn_exp2f(c * __builtin_amdgcn_logf(x[i]));
    split[i] = pow_split(x[i], c);
}

int main() {
    const int per = 1 << 16;
    const double bands[][2] = {{-40, -30}, {-30, -20}, {-20, -10}, {-10, -3}, {-3, -1}, {-1, 1}, {1, 3}, {3, 10}, {10, 20}, {20, 38}};
    const int nb = sizeof bands / sizeof bands[0];
    std::vector<float> hx((size_t)nb * per);
    unsigned long long s = 88172645463325252ull;
    for (size_t i = 0; i < hx.size(); ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double u = (double)(s >> 11) / 9007199254740992.0;
        const int b = (int)(i / per);
        hx[i] = (float)std::exp2(bands[b][0] + u * (bands[b][1] - bands[b][0]));
    }
    float *dx, *dp, *ds;
    hipMalloc(&dx, hx.size() * 4); hipMalloc(&dp, hx.size() * 4); hipMalloc(&ds, hx.size() * 4);
    hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> hp(hx.size()), hs(hx.size());
    for (float p : {0.8f, 0.5f, 1.0f, 0.3f}) {
        const float c = -0.5f * p;
        hipLaunchKernelGGL(eval, dim3((unsigned)((hx.size() + 255) / 256)), dim3(256), 0, 0, dx, dp, ds, c, (int)hx.size());
        hipMemcpy(hp.data(), dp, hx.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(hs.data(), ds, hx.size() * 4, hipMemcpyDeviceToHost);
        printf("feedback_exponent %.1f (c = %.2f): error in ulp of the exact value, max / mean\n", p, c);
        for (int b = 0; b < nb; ++b) {
            double mp = 0, ms = 0, ap = 0, as = 0;
            for (int k = 0; k < per; ++k) {
                const size_t i = (size_t)b * per + k;
                const double exact = std::pow((double)hx[i], (double)c);
                int ex;
                std::frexp(exact, &ex);
                const double ulp = std::ldexp(1.0, ex - 24);
                const double ep = std::fabs(hp[i] - exact) / ulp, es = std::fabs(hs[i] - exact) / ulp;
                mp = std::fmax(mp, ep); ms = std::fmax(ms, es); ap += ep; as += es;
            }
            printf("  x in 2^[%4.0f, %4.0f]: plain %7.2f / %6.3f   split %5.2f / %5.3f\n", bands[b][0], bands[b][1], mp, ap / per, ms, as / per);
        }
    }
    return 0;
}
