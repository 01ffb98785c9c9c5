#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double* x, double* r0, double* r1, double* r2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double d = x[i];
    double r = __builtin_amdgcn_rcp(d);
    r0[i] = r;
    r = fma(fma(-d, r, 1.0), r, r);
    r1[i] = r;
    r = fma(fma(-d, r, 1.0), r, r);
    r2[i] = r;
}
int main() {
    const int n = 1 << 20;
    double *x, *a, *b, *c;
    hipMallocManaged(&x, n * 8); hipMallocManaged(&a, n * 8); hipMallocManaged(&b, n * 8); hipMallocManaged(&c, n * 8);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x[i] = std::ldexp(1.0 + (double)(s >> 11) / 9007199254740992.0, (int)(s % 40) - 20); }
    k<<<n / 256, 256>>>(x, a, b, c, n);
    hipDeviceSynchronize();
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; i++) {
        long double t = 1.0L / (long double)x[i];
        e0 = std::fmax(e0, (double)fabsl((a[i] - t) / t)); e1 = std::fmax(e1, (double)fabsl((b[i] - t) / t)); e2 = std::fmax(e2, (double)fabsl((c[i] - t) / t));
    }
    printf("v_rcp_f64 max rel err: raw %.3e  +1 newton %.3e  +2 newton %.3e\n", e0, e1, e2);
    return 0;
}
