// Issue rates of the integer instructions a counter-based generator is made of, on gfx950:
// the full 32-bit multiply against the 24-bit forms, shifts, exclusive-or, the conversion
// to float.  8 independent chains, 1 and 8 waves per SIMD; prints cycles per wave64
// instruction per SIMD at the nominal clock.
//   hipcc --offload-arch=gfx950 -O2 -o int_rates int_rates.hip && ./int_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define REP8(INS, TAIL) \
    INS " %0, %0, " TAIL "\n" INS " %1, %1, " TAIL "\n" INS " %2, %2, " TAIL "\n" INS " %3, %3, " TAIL "\n" \
    INS " %4, %4, " TAIL "\n" INS " %5, %5, " TAIL "\n" INS " %6, %6, " TAIL "\n" INS " %7, %7, " TAIL "\n"
#define REGS "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)

template <int FORM>
__global__ void __launch_bounds__(256) k_rate(uint32_t *out, int iters)
{
    const uint32_t t = threadIdx.x;
    uint32_t a0 = t, a1 = t + 2, a2 = t + 4, a3 = t + 6, a4 = t + 8, a5 = t + 10, a6 = t + 12, a7 = t + 14;
    uint32_t c = 747796405u, c24 = 0x9E3779u, sh = 13;
    asm volatile("" : "+v"(c), "+v"(c24), "+v"(sh));
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (FORM == 0) asm volatile(REP8("v_mul_lo_u32", "%8") : REGS : "v"(c));
            else if (FORM == 1) asm volatile(REP8("v_mul_u32_u24", "%8") : REGS : "v"(c24));
            else if (FORM == 2) asm volatile(REP8("v_mad_u32_u24", "%8, %9") : REGS : "v"(c24), "v"(c));
            else if (FORM == 3) asm volatile(REP8("v_xor_b32", "%8") : REGS : "v"(c));
            else if (FORM == 4) asm volatile(REP8("v_lshrrev_b32", "%8") : REGS : "v"(sh));   // a = sh >> a: only the rate matters
            else if (FORM == 5) asm volatile(REP8("v_add_u32", "%8") : REGS : "v"(c));
            else if (FORM == 6) asm volatile(REP8("v_mul_hi_u32", "%8") : REGS : "v"(c));
            else if (FORM == 7) asm volatile(REP8("v_mul_lo_u32", "%8") : REGS : "s"(c24));   // multiplier in a scalar register
            else if (FORM == 8) asm volatile(REP8("v_xad_u32", "%8, %9") : REGS : "v"(sh), "v"(c));   // (a ^ sh) + c
            else if (FORM == 9) asm volatile(REP8("v_alignbit_b32", "%8, %9") : REGS : "v"(c), "v"(sh));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int FORM>
__global__ void __launch_bounds__(256) k_cvt(float *out, int iters)
{
    const uint32_t t = threadIdx.x;
    uint32_t a0 = t, a1 = t + 2, a2 = t + 4, a3 = t + 6, a4 = t + 8, a5 = t + 10, a6 = t + 12, a7 = t + 14;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 8; k++)
            asm volatile("v_cvt_f32_u32 %0, %0\n v_cvt_f32_u32 %1, %1\n v_cvt_f32_u32 %2, %2\n v_cvt_f32_u32 %3, %3\n"
                         "v_cvt_f32_u32 %4, %4\n v_cvt_f32_u32 %5, %5\n v_cvt_f32_u32 %6, %6\n v_cvt_f32_u32 %7, %7\n" : REGS);
    }
    out[blockIdx.x * 256 + threadIdx.x] = __uint_as_float(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
}

template <typename K>
void time_it(const char *what, K launch, int waves_per_simd, double mhz)
{
    const int iters = 20000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch(100);
    CHECK(hipEventRecord(e0));
    launch(iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-40s waves/SIMD=%d: %.3f cycles per instruction per SIMD\n", what, waves_per_simd,
           ms * 1e-3 * mhz * 1e6 / iters / (64.0 * waves_per_simd));
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const double mhz = p.clockRate / 1000.0;
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs, %.0f MHz\n", p.name, cus, mhz);
    uint32_t *out;
    CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * sizeof(uint32_t)));
    for (int w : {1, 8}) {
#define RUN(F, NAME) time_it(NAME, [&](int n) { k_rate<F><<<cus * w, 256>>>(out, n); }, w, mhz)
        RUN(0, "v_mul_lo_u32 v, v, v");
        RUN(7, "v_mul_lo_u32 v, v, s");
        RUN(6, "v_mul_hi_u32");
        RUN(1, "v_mul_u32_u24");
        RUN(2, "v_mad_u32_u24");
        RUN(3, "v_xor_b32");
        RUN(4, "v_lshrrev_b32");
        RUN(5, "v_add_u32");
        RUN(8, "v_xad_u32");
        RUN(9, "v_alignbit_b32");
        time_it("v_cvt_f32_u32", [&](int n) { k_cvt<0><<<cus * w, 256>>>((float *)out, n); }, w, mhz);
    }
    CHECK(hipFree(out));
    return 0;
}
