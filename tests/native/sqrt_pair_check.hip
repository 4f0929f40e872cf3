// The product build's square root (pik_math.hpp sqrt_pair: v_rsq_f64 + one coupled Goldschmidt step + one
// residual correction) against the correctly rounded IEEE results, over the arguments the path feeds it: sums of
// squares of lengths and of unit-quaternion components.  Prints, for n inputs spread log-uniformly over
// [lo, hi], how many roots / half-inverses differ from sqrt(x) / (0.5 / sqrt(x)) and by how many ulp at most.
// usage: sqrt_pair_check n lo hi      (tests/test_gpu_product_arithmetic.py builds and runs it on the GPU box)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../pick_ik_amd/csrc/pik_math.hpp"

__global__ void run(const double* x, double* root, double* hinv, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double r, h;
    pik::sqrt_pair(x[i], r, h);
    root[i] = r;
    hinv[i] = h;
}

static long long ulp_distance(double a, double b) {
    long long x, y;
    std::memcpy(&x, &a, 8);
    std::memcpy(&y, &b, 8);
    return x > y ? x - y : y - x;
}

int main(int argc, char** argv) {
    const long long n = argc > 1 ? std::atoll(argv[1]) : 10000000;
    const double lo = argc > 2 ? std::atof(argv[2]) : 1e-12, hi = argc > 3 ? std::atof(argv[3]) : 1e2;
    std::vector<double> x((size_t)n), r((size_t)n), h((size_t)n);
    unsigned long long s = 0x9E3779B97F4A7C15ull;
    for (long long i = 0; i < n; ++i) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        const double u = (double)(s >> 11) * (1.0 / 9007199254740992.0);
        x[(size_t)i] = std::exp(std::log(lo) + (std::log(hi) - std::log(lo)) * u);
    }
    double *dx, *dr, *dh;
    if (hipMalloc(&dx, 8 * n) != hipSuccess || hipMalloc(&dr, 8 * n) != hipSuccess || hipMalloc(&dh, 8 * n) != hipSuccess) return 2;
    hipMemcpy(dx, x.data(), 8 * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(run, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, dx, dr, dh, n);
    if (hipMemcpy(r.data(), dr, 8 * n, hipMemcpyDeviceToHost) != hipSuccess) return 3;
    hipMemcpy(h.data(), dh, 8 * n, hipMemcpyDeviceToHost);
    long long bad_r = 0, bad_h = 0, max_r = 0, max_h = 0;
    for (long long i = 0; i < n; ++i) {
        const double want = std::sqrt(x[(size_t)i]);
        const long long er = ulp_distance(r[(size_t)i], want), eh = ulp_distance(h[(size_t)i], 0.5 / want);
        bad_r += er != 0;
        bad_h += eh != 0;
        max_r = er > max_r ? er : max_r;
        max_h = eh > max_h ? eh : max_h;
    }
    std::printf("{\"n\": %lld, \"lo\": %g, \"hi\": %g, \"root_not_correctly_rounded\": %lld, \"root_max_ulp\": %lld, "
                "\"half_inverse_differs_from_0.5_over_sqrt\": %lld, \"half_inverse_max_ulp\": %lld}\n",
                n, lo, hi, bad_r, max_r, bad_h, max_h);
    return 0;
}
