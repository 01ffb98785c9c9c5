// Dependent-issue latency of the operations on the rollout kernel's critical chains, one lone wavefront:
//   hipcc --offload-arch=gfx950 -O3 -o latency latency.hip && ./latency
// Each loop is a chain of N dependent instructions; cycles come from s_memtime (constant 100 MHz on gfx950? no:
// wall_clock64 is 100 MHz; clock64() counts shader cycles).
#include <hip/hip_runtime.h>
#include <cstdio>

#define N 4096

template <int OP>
__global__ void chain(double* out, double seed, long long* cycles) {
    double x = seed + threadIdx.x * 1e-9;
    float xf = (float)x;
    __shared__ double lds[64 * 8];
    lds[threadIdx.x] = x;
    int idx = threadIdx.x;
    __syncthreads();
    long long t0 = clock64();
#pragma unroll 16
    for (int i = 0; i < N; ++i) {
        if (OP == 0) x = fma(x, 1.0000001, 1e-9);                              // v_fma_f64
        if (OP == 1) x = __builtin_amdgcn_rcp(x) + 0.5;                        // v_rcp_f64 + add
        if (OP == 2) xf = __builtin_amdgcn_rcpf(xf) + 0.5f;                    // v_rcp_f32 + add
        if (OP == 3) xf = fmaf(xf, 1.0000001f, 1e-9f);                          // v_fma_f32
        if (OP == 4) { idx = (int)lds[idx & 63] & 63; }                        // ds_read_b64 -> address
        if (OP == 5) {                                                         // f64 DPP shift (2 movs) + add
            int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x111, 0xF, 0xF, true);
            int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x111, 0xF, 0xF, true);
            x += __hiloint2double(hi, lo) * 1e-9;
        }
        if (OP == 6) x = x * 1.0000001;                                        // v_mul_f64
        if (OP == 7) x = sqrt(x) + 1.0;                                        // sqrt f64 (library sequence)
        if (OP == 8) { double r = __builtin_amdgcn_rcp(x); r = fma(fma(-x, r, 1.0), r, r); x = r + 0.5; }   // rcp_fast
    }
    long long t1 = clock64();
    out[threadIdx.x] = x + xf + idx;
    if (threadIdx.x == 0) *cycles = t1 - t0;
}

template <int OP>
void run(const char* name, double seed) {
    double* out;
    long long* cyc;
    hipMalloc(&out, 64 * 8);
    hipMalloc(&cyc, 8);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(chain<OP>, dim3(1), dim3(64), 0, 0, out, seed, cyc);
    hipDeviceSynchronize();
    long long h;
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-34s %7.1f clock64 ticks per iteration\n", name, (double)h / N);
    hipFree(out);
    hipFree(cyc);
}

int main() {
    run<0>("v_fma_f64 (dependent)", 1.0);
    run<6>("v_mul_f64 (dependent)", 1.0);
    run<3>("v_fma_f32 (dependent)", 1.0);
    run<1>("v_rcp_f64 + v_add_f64", 1.3);
    run<8>("rcp_fast f64 (rcp + 2 fma) + add", 1.3);
    run<2>("v_rcp_f32 + v_add_f32", 1.3);
    run<5>("2 x v_mov_dpp + fma f64", 1.0);
    run<4>("ds_read_b64 -> cvt -> address", 3.0);
    run<7>("sqrt f64 + add", 2.0);
    return 0;
}
