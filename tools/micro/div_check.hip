// Is ONE correction step enough for the short division of rt_device.h?  q1 = fma(fma(-b, q0, a), r, q0) with q0 = a * r and
// r = rcp_refined(b), against the IEEE quotient a / b.  A correctly rounded quotient of normal operands depends on the two
// significands only (scaling by powers of two is exact in every step while nothing under- or overflows), so the check runs
// over significand pairs at fixed exponents.
//   mode 0: b = every float of norm3_unit's window around 1 (8193 values), a = every significand (2^23) in [0.5, 1)   -- exhaustive
//   mode 1: b = significands  [b0, b0 + nb) of [1, 2), a = every significand in [1, 2)                                -- exhaustive per slice
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -o div_check div_check.hip
// run:   ./div_check 0            ./div_check 1 <b0> <nb>          ./div_check 1 0 8388608  = every pair of significands (about a minute)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
__device__ __forceinline__ float rcp_refined(float b) {
    float r = __builtin_amdgcn_rcpf(b);
    return __builtin_fmaf(__builtin_fmaf(-b, r, 1.0f), r, r);
}
__global__ __launch_bounds__(256) void k(int mode, uint32_t b0, uint32_t nb, unsigned long long* bad1, unsigned long long* bad2, unsigned long long* badr, unsigned long long* bad0) {
    // one workgroup row per b, the threads stride over the 2^23 significands of a
    const uint32_t bi = blockIdx.y;
    if (bi >= nb) return;
    float b;
    if (mode == 0) b = __int_as_float(0x3f800000 - 4096 + (int)bi);
    else b = __int_as_float((int)(0x3f800000u + b0 + bi));
    const float r = rcp_refined(b);
    if (threadIdx.x == 0 && blockIdx.x == 0) { // is r the correctly rounded reciprocal?
        const float want = 1.0f / b;
        if (__float_as_int(want) != __float_as_int(r)) atomicAdd(badr, 1ull);
    }
    unsigned long long n0 = 0, n1 = 0, n2 = 0;
    for (uint32_t m = blockIdx.x * 256u + threadIdx.x; m < (1u << 23); m += gridDim.x * 256u) {
        const float a = __int_as_float((int)((mode == 0 ? 0x3f000000u : 0x3f800000u) + m));
        const float want = a / b;
        float q = a * r;
        n0 += __float_as_int(q) != __float_as_int(want); // sanity: the uncorrected product is often off by one
        q = __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
        n1 += __float_as_int(q) != __float_as_int(want);
        q = __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
        n2 += __float_as_int(q) != __float_as_int(want);
    }
    if (n1) atomicAdd(bad1, n1);
    if (n2) atomicAdd(bad2, n2);
    if (n0) atomicAdd(bad0, n0);
}
int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const uint32_t b0 = argc > 2 ? (uint32_t)strtoul(argv[2], 0, 0) : 0, nb = mode == 0 ? 8193u : (argc > 3 ? (uint32_t)strtoul(argv[3], 0, 0) : 4096u);
    unsigned long long *d, h[4] = {0, 0, 0, 0};
    (void)hipMalloc(&d, 32); (void)hipMemset(d, 0, 32);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    // rows of b in chunks so that one launch stays short
    for (uint32_t at = 0; at < nb; at += 1024) {
        const uint32_t n = nb - at < 1024 ? nb - at : 1024;
        k<<<dim3(64, n), 256>>>(mode, b0 + at, n, d, d + 1, d + 2, d + 3);
        if ((at & 0xfffff) == 0) { (void)hipDeviceSynchronize(); fprintf(stderr, "row %u of %u\n", at, nb); } // a sign of life now and then
    }
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
    printf("mode %d, b rows %u (from %u), %.3e quotients in %.1f ms: no step wrong %llu, one step wrong %llu, two steps wrong %llu, reciprocal not correctly rounded %llu rows\n",
           mode, nb, b0, (double)nb * 8388608.0, ms, h[3], h[0], h[1], h[2]);
    return 0;
}
