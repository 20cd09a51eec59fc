// Micro-benchmark: sustained wave64 VALU issue rate on gfx950 for v_fma_f32, v_mul+v_add (no
// contraction), v_pk_fma_f32, correctly rounded division and sqrt, as a function of waves per SIMD and
// of the number of independent chains per lane.  Prints instructions per cycle per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CHAINS, int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float b, float c) {
    float a[CHAINS];
#pragma unroll
    for (int j = 0; j < CHAINS; ++j) a[j] = threadIdx.x * 0.001f + j;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[CHAINS];
#pragma unroll
    for (int j = 0; j < CHAINS; ++j) p[j] = f2{a[j], a[j] + 1.f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < CHAINS; ++j) {
            if (MODE == 0) a[j] = __builtin_fmaf(a[j], b, c);          // 1 instr
            if (MODE == 1) { a[j] = a[j] * b; a[j] = a[j] + c; }       // 2 instrs (contract off)
            if (MODE == 2) p[j] = __builtin_elementwise_fma(p[j], f2{b, b}, f2{c, c});  // v_pk_fma
            if (MODE == 3) a[j] = c / a[j] + b;                         // IEEE divide (+1 add)
            if (MODE == 4) a[j] = __builtin_sqrtf(a[j]) + c;            // IEEE sqrt (+1 add)
        }
    }
    float s = 0;
#pragma unroll
    for (int j = 0; j < CHAINS; ++j) s += a[j] + p[j].x + p[j].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS, int MODE>
void run(const char* name, int instr_per_step) {
    float* out;
    hipMalloc(&out, 256 * 1024 * 64 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    int iters = MODE >= 3 ? 2000 : 20000;
    for (int waves_per_simd : {1, 2, 4, 8}) {
        int blocks = 256 * waves_per_simd;  // 256 CUs x (waves_per_simd x 4 SIMDs / 4 waves per block)
        hipLaunchKernelGGL((k<CHAINS, MODE>), dim3(blocks), dim3(256), 0, 0, out, 10, 1.0001f, 0.5f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<CHAINS, MODE>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double instr = (double)blocks * 4 * iters * CHAINS * instr_per_step;   // wave-instructions
        double per_simd_per_s = instr / 1024 / (ms * 1e-3);
        printf("%-28s chains=%d waves/SIMD=%d : %.3f ms, %.3f Ginstr/s/SIMD (= %.2f instr/cycle @2.4GHz)\n", name, CHAINS,
               waves_per_simd, ms, per_simd_per_s / 1e9, per_simd_per_s / 2.4e9);
    }
    hipFree(out);
}

int main() {
    run<1, 0>("v_fma_f32", 1);
    run<4, 0>("v_fma_f32", 1);
    run<8, 0>("v_fma_f32", 1);
    run<4, 1>("v_mul_f32+v_add_f32", 2);
    run<8, 1>("v_mul_f32+v_add_f32", 2);
    run<4, 2>("v_pk_fma_f32", 1);
    run<8, 2>("v_pk_fma_f32", 1);
    run<1, 3>("fdiv(IEEE)+add [steps]", 1);
    run<4, 3>("fdiv(IEEE)+add [steps]", 1);
    run<4, 4>("sqrt(IEEE)+add [steps]", 1);
    return 0;
}
