// Micro-benchmark: sustained wave64 VALU issue rate on gfx950, in shader cycles (s_memtime) so that the
// result does not depend on the clock the chip holds.  For each instruction kind, CHAINS independent
// dependency chains per lane and 1..8 waves per SIMD; prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return; } } while (0)

template <int CHAINS, int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters, float b, float c) {
    float a[CHAINS];
#pragma unroll
    for (int j = 0; j < CHAINS; ++j) a[j] = threadIdx.x * 0.001f + j + 1.0f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[CHAINS];
#pragma unroll
    for (int j = 0; j < CHAINS; ++j) p[j] = f2{a[j], a[j] + 1.f};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < CHAINS; ++j) {
            if (MODE == 0) a[j] = __builtin_fmaf(a[j], b, c);
            if (MODE == 1) { a[j] = a[j] * b; a[j] = a[j] + c; }
            if (MODE == 2) p[j] = __builtin_elementwise_fma(p[j], f2{b, b}, f2{c, c});
            if (MODE == 3) a[j] = c / a[j] + b;
            if (MODE == 4) a[j] = __builtin_sqrtf(a[j]) + c;
            if (MODE == 5) a[j] = __builtin_amdgcn_rcpf(a[j]) + c;
            if (MODE == 6) a[j] = (a[j] < b) ? c : a[j] * b;   // v_cmp + v_cndmask + v_mul
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int j = 0; j < CHAINS; ++j) s += a[j] + p[j].x + p[j].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int CHAINS, int MODE>
void run(const char* name, int instr_per_step) {
    float* out; unsigned long long* cyc;
    CHECK(hipMalloc(&out, 256 * 8 * 256 * sizeof(float)));
    CHECK(hipMalloc(&cyc, 256 * 8 * 4 * sizeof(unsigned long long)));
    int iters = MODE >= 3 && MODE <= 4 ? 4000 : 40000;
    for (int waves_per_simd : {1, 2, 4, 8}) {
        int blocks = 256 * waves_per_simd;
        hipLaunchKernelGGL((k<CHAINS, MODE>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 1.0001f, 0.5f);
        CHECK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(blocks * 4);
        CHECK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
        double mean = 0; for (auto v : h) mean += (double)v; mean /= h.size();
        double wave_instr = (double)iters * CHAINS * instr_per_step;
        // all waves of a SIMD run concurrently for ~mean cycles and each issues wave_instr instructions
        printf("%-24s chains=%d waves/SIMD=%d : %8.0f cycles/wave, %.2f cycles per instr per wave, %.2f cycles per instr per SIMD\n",
               name, CHAINS, waves_per_simd, mean, mean / wave_instr, mean / (wave_instr * waves_per_simd));
    }
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    run<1, 0>("v_fma_f32", 1);
    run<4, 0>("v_fma_f32", 1);
    run<8, 0>("v_fma_f32", 1);
    run<8, 1>("v_mul_f32+v_add_f32", 2);
    run<8, 2>("v_pk_fma_f32", 1);
    run<4, 5>("v_rcp_f32+v_add", 2);
    run<4, 6>("v_cmp+v_cndmask+v_mul", 3);
    run<1, 3>("fdiv IEEE (+add) /step", 1);
    run<4, 3>("fdiv IEEE (+add) /step", 1);
    run<4, 4>("sqrt IEEE (+add) /step", 1);
    return 0;
}
