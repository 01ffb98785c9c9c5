// Issue cost of INDEPENDENT instructions for a lone wavefront on gfx950 (the companion of latency.hip, which times
// dependent chains):   hipcc --offload-arch=gfx950 -O3 -o issue_rate issue_rate.hip && ./issue_rate
// Each loop body holds 16 mutually independent instructions of one kind; ticks are shader cycles (clock64()).
#include <hip/hip_runtime.h>
#include <cstdio>

#define N 1024

template <int OP>
__global__ void indep(double* out, double seed, long long* cycles) {
    double x[16];
    int xi[16];
    for (int k = 0; k < 16; ++k) { x[k] = seed + threadIdx.x * 1e-9 + k; xi[k] = threadIdx.x + k; }
    __shared__ double lds[64 * 17];
    for (int k = 0; k < 17; ++k) lds[threadIdx.x + 64 * k] = seed + k;
    __syncthreads();
    long long t0 = clock64();
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (OP == 0) x[k] = fma(x[k], 1.0000001, 1e-9);
            if (OP == 1) xi[k] = __builtin_amdgcn_update_dpp(0, xi[k], 0x111, 0xF, 0xF, true);          // row_shr:1
            if (OP == 2) xi[k] = __builtin_amdgcn_update_dpp(0, xi[k], 0x4E, 0xF, 0xF, true);           // quad_perm
            if (OP == 3) { float f = __int_as_float(xi[k]); f = fmaf(f, 1.0000001f, 1e-9f); xi[k] = __float_as_int(f); }
            if (OP == 4) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(x[k]) : "v"(x[k]));
            if (OP == 5) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x[k]) : "v"(x[(k + 1) & 15]), "v"(seed));
            if (OP == 6) x[k] = lds[(threadIdx.x + 64 * k)] + x[k];                                     // ds_read_b64 (conflict-free) + add
            if (OP == 7) x[k] = x[k] * 1.0000001;
            if (OP == 8) xi[k] = __builtin_amdgcn_ds_bpermute((xi[(k + 1) & 15] & 63) << 2, xi[k]);
            if (OP == 9) { xi[k] = __builtin_amdgcn_update_dpp(0, xi[k], 0x111, 0xF, 0xF, true); x[k] = fma(x[k], 1.0000001, 1e-9); }  // 1 DPP + 1 FMA64
        }
    }
    long long t1 = clock64();
    double s = 0;
    for (int k = 0; k < 16; ++k) s += x[k] + xi[k];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) *cycles = t1 - t0;
}

template <int OP>
void run(const char* name, int per_iter = 16) {
    double* out;
    long long* cyc;
    hipMalloc(&out, 64 * 8);
    hipMalloc(&cyc, 8);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(indep<OP>, dim3(1), dim3(64), 0, 0, out, 1.0, cyc);
    hipDeviceSynchronize();
    long long h;
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-44s %6.2f cycles per instruction\n", name, (double)h / N / per_iter);
    hipFree(out);
    hipFree(cyc);
}

int main() {
    run<0>("v_fma_f64 (independent)");
    run<7>("v_mul_f64 (independent)");
    run<3>("v_fma_f32 (independent)");
    run<1>("v_mov_b32_dpp row_shr:1 (independent)");
    run<2>("v_mov_b32_dpp quad_perm (independent)");
    run<9>("v_mov_b32_dpp + v_fma_f64 pairs", 32);
    run<4>("v_mov_b64_dpp row_newbcast (independent)");
    run<5>("v_fmac_f64_dpp row_newbcast (independent)");
    run<6>("ds_read_b64 + v_add_f64 pairs", 32);
    run<8>("ds_bpermute_b32 (independent)");
    return 0;
}
