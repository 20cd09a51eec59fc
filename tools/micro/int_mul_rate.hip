// Issue cost of 32-bit integer multiplies next to adds and 24-bit multiplies on gfx950, 4 waves per SIMD, independent chains.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o int_mul_rate int_mul_rate.hip ; run on the GPU box.
// Measured (MI355X): v_mul_lo_u32 4.75 cycles per wave-instruction on a SIMD, v_mad_u32_u24 4.3, a plain fp32 or integer op 2.3 - 2.8:
// the three multiplies of one draw of the counter RNG cost six plain operations, a draw about 14.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed, int iters) {
    uint32_t a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 7u + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) a[i] = a[i] * 0x7feb352dU;                                  // v_mul_lo_u32
            if (OP == 1) a[i] = a[i] + 0x7feb352dU ^ (a[i] >> 3);                    // add + shift + xor (3 full-rate ops)
            if (OP == 2) a[i] = __umul24(a[i], 0x6b352dU) + a[i];    // v_mad_u32_u24
            if (OP == 3) { float f = __uint_as_float(a[i]); f = f * 1.0001f + 0.5f; a[i] = __float_as_uint(f); } // 2 fp32 ops (no fma: contract off)
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP>
float run(uint32_t* d, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<256 * 4 * 4, 256>>>(d, 1, 16);
    hipEventRecord(e0);
    k<OP><<<256 * 4 * 4, 256>>>(d, 1, iters);   // 16 workgroups of 4 waves per CU = 16 waves per SIMD worth of work, 4 resident at a time at most 8
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    uint32_t* d; hipMalloc(&d, 256 * 4 * 4 * 256 * 4);
    const int iters = 20000;
    const double wave_instr = 256.0 * 4 * 4 * 4 * 8.0 * iters; // waves x 8 ops x iters
    const char* names[4] = {"v_mul_lo_u32", "add+shift+xor (3 ops)", "v_mad_u32_u24", "fp32 mul + add (2 ops)"};
    float ms[4] = {run<0>(d, iters), run<1>(d, iters), run<2>(d, iters), run<3>(d, iters)};
    for (int i = 0; i < 4; ++i) {
        double per_simd = wave_instr / 1024.0;            // group-instructions per SIMD
        double cycles = ms[i] * 1e-3 * 2.4e9;
        printf("%-26s %8.3f ms  %.2f cycles per wave-level group on a SIMD\n", names[i], ms[i], cycles / per_simd);
    }
    return 0;
}
