// Measures how vector and scalar ALU instructions share a SIMD's issue slots on
// this GPU: a loop of V independent v_mul_f32 and S independent s_add_u32 per
// iteration, 8 waves per SIMD, timed for several (V, S).  Prints cycles per
// iteration per SIMD (from the kernel's duration and the shader clock).
//   hipcc --offload-arch=gfx950 -O2 -o issue_rates issue_rates.hip && ./issue_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ long long g_cycles;

// FORM 0: v_mul_f32 with a 32-bit literal (8-byte encoding); 1: v_mul_f32 of two registers
// (4 bytes); 2: v_fma_f32 of three registers (8 bytes).
template <int FORM>
__global__ void __launch_bounds__(256) k_valu(float *out, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float c = 1.0000001f, c2 = 1e-9f, sc = 1.0000001f;
    unsigned long long mask = 0x5555555555555555ull, mask2 = 0;
    asm volatile("" : "+v"(c), "+v"(c2), "+s"(sc), "+s"(mask), "+s"(mask2));
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (FORM == 0)
                asm volatile("v_mul_f32 %0, 0x3f800001, %0\n v_mul_f32 %1, 0x3f800001, %1\n v_mul_f32 %2, 0x3f800001, %2\n v_mul_f32 %3, 0x3f800001, %3\n"
                             "v_mul_f32 %4, 0x3f800001, %4\n v_mul_f32 %5, 0x3f800001, %5\n v_mul_f32 %6, 0x3f800001, %6\n v_mul_f32 %7, 0x3f800001, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            else if (FORM == 1)
                asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n"
                             "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
            else if (FORM == 3)      // distinct multiplier and addend registers
                asm volatile("v_fma_f32 %0, %8, %0, %9\n v_fma_f32 %1, %8, %1, %9\n v_fma_f32 %2, %8, %2, %9\n v_fma_f32 %3, %8, %3, %9\n"
                             "v_fma_f32 %4, %8, %4, %9\n v_fma_f32 %5, %8, %5, %9\n v_fma_f32 %6, %8, %6, %9\n v_fma_f32 %7, %8, %7, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(c2));
            else if (FORM == 4)      // multiplier in a scalar register
                asm volatile("v_fma_f32 %0, %8, %0, %9\n v_fma_f32 %1, %8, %1, %9\n v_fma_f32 %2, %8, %2, %9\n v_fma_f32 %3, %8, %3, %9\n"
                             "v_fma_f32 %4, %8, %4, %9\n v_fma_f32 %5, %8, %5, %9\n v_fma_f32 %6, %8, %6, %9\n v_fma_f32 %7, %8, %7, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(sc), "v"(c2));
            else if (FORM == 5)      // v_fmac_f32: accumulate into the destination (4-byte encoding)
                asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                             "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(c2));
            else if (FORM == 6)      // v_add_f32 of two registers
                asm volatile("v_add_f32 %0, %8, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n"
                             "v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
            else if (FORM == 8)      // v_mul_lo_u32 (32-bit integer multiply)
                asm volatile("v_mul_lo_u32 %0, %8, %0\n v_mul_lo_u32 %1, %8, %1\n v_mul_lo_u32 %2, %8, %2\n v_mul_lo_u32 %3, %8, %3\n"
                             "v_mul_lo_u32 %4, %8, %4\n v_mul_lo_u32 %5, %8, %5\n v_mul_lo_u32 %6, %8, %6\n v_mul_lo_u32 %7, %8, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
            else if (FORM == 9)      // v_rcp_f32
                asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                             "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            else if (FORM == 10)     // v_mad_u32_u24
                asm volatile("v_mad_u32_u24 %0, %8, %0, %8\n v_mad_u32_u24 %1, %8, %1, %8\n v_mad_u32_u24 %2, %8, %2, %8\n v_mad_u32_u24 %3, %8, %3, %8\n"
                             "v_mad_u32_u24 %4, %8, %4, %8\n v_mad_u32_u24 %5, %8, %5, %8\n v_mad_u32_u24 %6, %8, %6, %8\n v_mad_u32_u24 %7, %8, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
            else if (FORM == 11)     // v_cndmask_b32 with a mask in scalar registers
                asm volatile("v_cndmask_b32 %0, %8, %0, %9\n v_cndmask_b32 %1, %8, %1, %9\n v_cndmask_b32 %2, %8, %2, %9\n v_cndmask_b32 %3, %8, %3, %9\n"
                             "v_cndmask_b32 %4, %8, %4, %9\n v_cndmask_b32 %5, %8, %5, %9\n v_cndmask_b32 %6, %8, %6, %9\n v_cndmask_b32 %7, %8, %7, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "s"(mask));
            else if (FORM == 12)     // v_cmp_lt_f32 into a scalar register pair
                asm volatile("v_cmp_lt_f32 %10, %8, %0\n v_cmp_lt_f32 %10, %8, %1\n v_cmp_lt_f32 %10, %8, %2\n v_cmp_lt_f32 %10, %8, %3\n"
                             "v_cmp_lt_f32 %10, %8, %4\n v_cmp_lt_f32 %10, %8, %5\n v_cmp_lt_f32 %10, %8, %6\n v_cmp_lt_f32 %10, %8, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "s"(mask), "s"(mask2));
            else if (FORM == 13)     // v_cmp_lt_f32 into VCC, two vector registers (4-byte encoding)
                asm volatile("v_cmp_lt_f32 vcc, %8, %0\n v_cmp_lt_f32 vcc, %8, %1\n v_cmp_lt_f32 vcc, %8, %2\n v_cmp_lt_f32 vcc, %8, %3\n"
                             "v_cmp_lt_f32 vcc, %8, %4\n v_cmp_lt_f32 vcc, %8, %5\n v_cmp_lt_f32 vcc, %8, %6\n v_cmp_lt_f32 vcc, %8, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc");
            else if (FORM == 14)     // v_cmp_lt_f32 into VCC, a literal and a vector register
                asm volatile("v_cmp_lt_f32 vcc, 0x3f800011, %0\n v_cmp_lt_f32 vcc, 0x3f800011, %1\n v_cmp_lt_f32 vcc, 0x3f800011, %2\n v_cmp_lt_f32 vcc, 0x3f800011, %3\n"
                             "v_cmp_lt_f32 vcc, 0x3f800011, %4\n v_cmp_lt_f32 vcc, 0x3f800011, %5\n v_cmp_lt_f32 vcc, 0x3f800011, %6\n v_cmp_lt_f32 vcc, 0x3f800011, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");
            else if (FORM == 15)     // v_cndmask_b32 with the mask in VCC (4-byte encoding)
                asm volatile("v_cndmask_b32 %0, %8, %0, vcc\n v_cndmask_b32 %1, %8, %1, vcc\n v_cndmask_b32 %2, %8, %2, vcc\n v_cndmask_b32 %3, %8, %3, vcc\n"
                             "v_cndmask_b32 %4, %8, %4, vcc\n v_cndmask_b32 %5, %8, %5, vcc\n v_cndmask_b32 %6, %8, %6, vcc\n v_cndmask_b32 %7, %8, %7, vcc\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc");
            else if (FORM == 16)     // v_sub_f32 with a negated-operand modifier (8-byte encoding), vector registers
                asm volatile("v_mul_f32_e64 %0, %8, -%0\n v_mul_f32_e64 %1, %8, -%1\n v_mul_f32_e64 %2, %8, -%2\n v_mul_f32_e64 %3, %8, -%3\n"
                             "v_mul_f32_e64 %4, %8, -%4\n v_mul_f32_e64 %5, %8, -%5\n v_mul_f32_e64 %6, %8, -%6\n v_mul_f32_e64 %7, %8, -%7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
            else if (FORM == 7)      // v_mul_f32 with a scalar register operand
                asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n"
                             "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(sc));
            else
                asm volatile("v_fma_f32 %0, %8, %0, %8\n v_fma_f32 %1, %8, %1, %8\n v_fma_f32 %2, %8, %2, %8\n v_fma_f32 %3, %8, %3, %8\n"
                             "v_fma_f32 %4, %8, %4, %8\n v_fma_f32 %5, %8, %5, %8\n v_fma_f32 %6, %8, %6, %8\n v_fma_f32 %7, %8, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    // shader-clock cycles this wave spent in the loop (s_memtime), for the first wave of the grid
    if (blockIdx.x == 0 && threadIdx.x == 0) g_cycles = clock64() - t0;
}

template <int FORM>
void run_valu(float *out, int waves_per_simd, double mhz, int cus)
{
    const int iters = 20000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k_valu<FORM><<<cus * waves_per_simd, 256>>>(out, 100);
    CHECK(hipEventRecord(e0));
    k_valu<FORM><<<cus * waves_per_simd, 256>>>(out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    long long cycles = 0;
    CHECK(hipMemcpyFromSymbol(&cycles, HIP_SYMBOL(g_cycles), sizeof cycles));
    printf("form %d, waves/SIMD=%d: %.3f cycles per vector instruction per SIMD at the nominal clock", FORM, waves_per_simd,
           ms * 1e-3 * mhz * 1e6 / iters / (64.0 * waves_per_simd));
    // (with several waves the oldest one is served first and finishes early: its counter says nothing about the SIMD)
    if (waves_per_simd == 1) printf(", %.3f by the wave's own cycle counter", (double)cycles / iters / 64.0);
    printf("\n");
}

template <int V, int S, int B>
__global__ void __launch_bounds__(256) k_mix(float *out, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned s0 = blockIdx.x, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < V / 8; k++)
            asm volatile("v_mul_f32 %0, 0x3f800001, %0\n v_mul_f32 %1, 0x3f800001, %1\n v_mul_f32 %2, 0x3f800001, %2\n v_mul_f32 %3, 0x3f800001, %3\n"
                         "v_mul_f32 %4, 0x3f800001, %4\n v_mul_f32 %5, 0x3f800001, %5\n v_mul_f32 %6, 0x3f800001, %6\n v_mul_f32 %7, 0x3f800001, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
#pragma unroll
        for (int k = 0; k < S / 4; k++)
            asm volatile("s_add_u32 %0, %0, 3\n s_add_u32 %1, %1, 5\n s_add_u32 %2, %2, 7\n s_add_u32 %3, %3, 9\n"
                         : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
#pragma unroll
        for (int k = 0; k < B; k++)          // taken scalar branches
            asm volatile("s_cmp_eq_u32 %0, %0\n s_cbranch_scc1 1f\n s_nop 0\n1:\n" : : "s"(s0) : "scc");
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(s0 + s1 + s2 + s3);
}

template <int V, int S, int B>
void run(float *out, int waves_per_simd, double mhz, int cus)
{
    const int iters = 20000;
    const int blocks = cus * waves_per_simd;            // 256 threads = 4 waves = one per SIMD
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k_mix<V, S, B><<<blocks, 256>>>(out, 100);
    CHECK(hipEventRecord(e0));
    k_mix<V, S, B><<<blocks, 256>>>(out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double cycles = ms * 1e-3 * mhz * 1e6 / iters;     // per iteration, all waves of a SIMD
    printf("V=%3d S=%3d B=%d waves/SIMD=%d: %8.1f cycles/iteration/SIMD  (%.2f per wave-iteration; V alone would be %d, at 2 cycles each)\n",
           V, S, B, waves_per_simd, cycles, cycles / waves_per_simd, 2 * V);
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const double mhz = p.clockRate / 1000.0;
    printf("%s, %d CUs, %.0f MHz\n", p.name, p.multiProcessorCount, mhz);
    float *out;
    CHECK(hipMalloc(&out, (size_t)p.multiProcessorCount * 8 * 256 * sizeof(float)));
    for (int w : {1, 8}) {
        run_valu<0>(out, w, mhz, p.multiProcessorCount);
        run_valu<1>(out, w, mhz, p.multiProcessorCount);
        run_valu<2>(out, w, mhz, p.multiProcessorCount);
        run_valu<3>(out, w, mhz, p.multiProcessorCount);
        run_valu<4>(out, w, mhz, p.multiProcessorCount);
        run_valu<5>(out, w, mhz, p.multiProcessorCount);
        run_valu<6>(out, w, mhz, p.multiProcessorCount);
        run_valu<7>(out, w, mhz, p.multiProcessorCount);
        run_valu<8>(out, w, mhz, p.multiProcessorCount);
        run_valu<9>(out, w, mhz, p.multiProcessorCount);
        run_valu<10>(out, w, mhz, p.multiProcessorCount);
        run_valu<11>(out, w, mhz, p.multiProcessorCount);
        run_valu<12>(out, w, mhz, p.multiProcessorCount);
        run_valu<13>(out, w, mhz, p.multiProcessorCount);
        run_valu<14>(out, w, mhz, p.multiProcessorCount);
        run_valu<15>(out, w, mhz, p.multiProcessorCount);
        run_valu<16>(out, w, mhz, p.multiProcessorCount);
    }
    for (int w : {1, 2, 8}) {
        run<64, 0, 0>(out, w, mhz, p.multiProcessorCount);
        run<0, 64, 0>(out, w, mhz, p.multiProcessorCount);
        run<64, 16, 0>(out, w, mhz, p.multiProcessorCount);
        run<64, 32, 0>(out, w, mhz, p.multiProcessorCount);
        run<64, 64, 0>(out, w, mhz, p.multiProcessorCount);
        run<64, 0, 4>(out, w, mhz, p.multiProcessorCount);
        run<0, 0, 8>(out, w, mhz, p.multiProcessorCount);
    }
    CHECK(hipFree(out));
    return 0;
}
