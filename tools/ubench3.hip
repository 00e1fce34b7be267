// Accuracy of v_rsq_f64 / v_rcp_f64 seeds and of the short refinement sequences (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
__global__ void k(const double* x, double* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    double y = __builtin_amdgcn_rsq(v);
    double r = __builtin_amdgcn_rcp(v);
    // one Goldschmidt iteration + one correction
    double g = v * y, h = 0.5 * y;
    double e = fma(-h, g, 0.5); g = fma(g, e, g); h = fma(h, e, h);
    double g1 = g, h1 = h + h;
    double d = fma(-g, g, v); g = fma(d, h, g);
    double g2 = g;
    // rcp: one and two Newton steps
    double e1 = fma(-v, r, 1.0); double r1 = fma(r, e1, r);
    double e2 = fma(-v, r1, 1.0); double r2 = fma(r1, e2, r1);
    out[i * 8 + 0] = y; out[i * 8 + 1] = r; out[i * 8 + 2] = g1; out[i * 8 + 3] = h1;
    out[i * 8 + 4] = g2; out[i * 8 + 5] = r1; out[i * 8 + 6] = r2; out[i * 8 + 7] = 0;
}
int main() {
    const int n = 1 << 20;
    std::vector<double> x(n);
    std::mt19937_64 rng(1);
    std::uniform_real_distribution<double> u(-300, 300);
    for (auto& v : x) v = std::pow(10.0, u(rng) * (rng() % 8 == 0 ? 1.0 : 0.01)) * (1.0 + (rng() % 1000) / 1000.0);
    double *dx, *dout;
    hipMalloc(&dx, n * 8); hipMalloc(&dout, n * 64);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dout, n);
    std::vector<double> o(n * 8);
    hipMemcpy(o.data(), dout, n * 64, hipMemcpyDeviceToHost);
    double m[7] = {0};
    for (int i = 0; i < n; ++i) {
        long double v = x[i];
        long double t[7] = {1 / sqrtl(v), 1 / v, sqrtl(v), 1 / sqrtl(v), sqrtl(v), 1 / v, 1 / v};
        for (int j = 0; j < 7; ++j) { double rel = (double)fabsl((o[i * 8 + j] - t[j]) / t[j]); if (rel > m[j]) m[j] = rel; }
    }
    const char* names[7] = {"v_rsq_f64 seed", "v_rcp_f64 seed", "sqrt after 1 iter", "rsqrt after 1 iter", "sqrt after 1 iter + 1 corr", "rcp after 1 Newton", "rcp after 2 Newton"};
    for (int j = 0; j < 7; ++j) printf("%-30s max rel err %.3e (%.2f ulp)\n", names[j], m[j], m[j] / 1.11e-16);
    return 0;
}
