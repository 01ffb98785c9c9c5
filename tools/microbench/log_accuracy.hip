// accuracy of device log / log1p / sqrt / div against host libm for r2 close to 1 (polar method tail)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#pragma clang fp contract(off)
__global__ void k(const double* x, double* a, double* b, double* c, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double r2 = x[i];
    a[i] = log(r2);
    b[i] = log1p(r2 - 1.0);
    double lr = r2 > 0.5 ? log1p(r2 - 1.0) : log(r2);
    c[i] = sqrt(-2.0 * lr / r2);
}
int main() {
    const int n = 1 << 16;
    double *x, *a, *b, *c;
    hipMallocManaged(&x, n * 8); hipMallocManaged(&a, n * 8); hipMallocManaged(&b, n * 8); hipMallocManaged(&c, n * 8);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; double u = (double)(s >> 11) / 9007199254740992.0; x[i] = (i & 1) ? 1.0 - u * 1e-4 : u; if (x[i] <= 0) x[i] = 0.3; }
    k<<<n / 256, 256>>>(x, a, b, c, n);
    hipDeviceSynchronize();
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; i++) {
        double t = std::log(x[i]);
        e0 = std::fmax(e0, std::fabs((a[i] - t) / t)); e1 = std::fmax(e1, std::fabs((b[i] - t) / t));
        double f = std::sqrt(-2.0 * t / x[i]);
        e2 = std::fmax(e2, std::fabs((c[i] - f) / f));
    }
    printf("max rel err vs glibc: log %.3e  log1p(r2-1) %.3e  f %.3e\n", e0, e1, e2);
    return 0;
}
