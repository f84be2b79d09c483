// Issue rates of the packed fp32 instructions of gfx950 (two IEEE operations per lane per
// instruction) against the plain ones, with vector-register, scalar-register and
// half-broadcast operands: 8 independent chains, 1 and 8 waves per SIMD.  Prints cycles
// per wave64 instruction per SIMD at the nominal clock.
//   hipcc --offload-arch=gfx950 -O2 -o packed_rates packed_rates.hip && ./packed_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(INS, TAIL) \
    INS " %0, %0, " TAIL "\n" INS " %1, %1, " TAIL "\n" INS " %2, %2, " TAIL "\n" INS " %3, %3, " TAIL "\n" \
    INS " %4, %4, " TAIL "\n" INS " %5, %5, " TAIL "\n" INS " %6, %6, " TAIL "\n" INS " %7, %7, " TAIL "\n"

template <int FORM>
__global__ void __launch_bounds__(256) k_rate(float *out, int iters)
{
    const float t = threadIdx.x;
    f2 a0 = {t, t + 1}, a1 = {t + 2, t + 3}, a2 = {t + 4, t + 5}, a3 = {t + 6, t + 7};
    f2 a4 = {t + 8, t + 9}, a5 = {t + 10, t + 11}, a6 = {t + 12, t + 13}, a7 = {t + 14, t + 15};
    f2 c = {1.0000001f, 0.9999999f}, sc = {1.0000001f, 0.9999999f};
    f2 c3 = {1e-9f, -1e-9f};
    asm volatile("" : "+v"(c), "+s"(sc), "+v"(c3));
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (FORM == 0)        // v_pk_mul_f32, vector registers
                asm volatile(REP8("v_pk_mul_f32", "%8") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
            else if (FORM == 1)   // v_pk_mul_f32, one operand a scalar register pair
                asm volatile(REP8("v_pk_mul_f32", "%8") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(sc));
            else if (FORM == 2)   // v_pk_add_f32, vector registers
                asm volatile(REP8("v_pk_add_f32", "%8") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c3));
            else if (FORM == 3)   // v_pk_add_f32, scalar register pair
                asm volatile(REP8("v_pk_add_f32", "%8") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(sc));
            else if (FORM == 4)   // v_pk_mul_f32, scalar pair, low half of the vector operand used for both results
                asm volatile(REP8("v_pk_mul_f32", "%8 op_sel:[0,0] op_sel_hi:[0,1]") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(sc));
            else if (FORM == 5)   // v_pk_fma_f32, vector registers
                asm volatile(REP8("v_pk_fma_f32", "%8, %9") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(c3));
            else if (FORM == 6)   // v_pk_fma_f32, multiplier a scalar pair
                asm volatile(REP8("v_pk_fma_f32", "%8, %9") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(sc), "v"(c3));
            else if (FORM == 7)   // plain v_mul_f32, vector registers (on the low halves)
                asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n"
                             "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7\n"
                             : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(c.x));
            else if (FORM == 8)   // plain v_mul_f32, scalar register operand
                asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n"
                             "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7\n"
                             : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "s"(sc.x));
            else if (FORM == 9)   // v_pk_mul_f32 with negated second operand (neg_lo/neg_hi modifiers)
                asm volatile(REP8("v_pk_mul_f32", "%8 neg_lo:[0,1] neg_hi:[0,1]") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
            else if (FORM == 10)  // alternating plain mul with scalar operand and plain mul with vector operand
                asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %9, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %9, %3\n"
                             "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %9, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %9, %7\n"
                             : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "s"(sc.x), "v"(c.x));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0.x + a1.x + a2.x + a3.x + a4.x + a5.x + a6.x + a7.x + a0.y + a1.y + a2.y + a3.y + a4.y + a5.y + a6.y + a7.y;
}

template <int FORM>
void run(const char *what, float *out, int waves_per_simd, double mhz, int cus)
{
    const int iters = 20000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k_rate<FORM><<<cus * waves_per_simd, 256>>>(out, 100);
    CHECK(hipEventRecord(e0));
    k_rate<FORM><<<cus * waves_per_simd, 256>>>(out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-72s waves/SIMD=%d: %.3f cycles per instruction per SIMD\n", what, waves_per_simd,
           ms * 1e-3 * mhz * 1e6 / iters / (64.0 * waves_per_simd));
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
        run<7>("v_mul_f32 v, v, v", out, w, mhz, p.multiProcessorCount);
        run<8>("v_mul_f32 v, s, v", out, w, mhz, p.multiProcessorCount);
        run<10>("v_mul_f32 alternating s and v operand", out, w, mhz, p.multiProcessorCount);
        run<0>("v_pk_mul_f32 v, v, v", out, w, mhz, p.multiProcessorCount);
        run<1>("v_pk_mul_f32 v, v, s", out, w, mhz, p.multiProcessorCount);
        run<4>("v_pk_mul_f32 v, v(lo for both), s", out, w, mhz, p.multiProcessorCount);
        run<9>("v_pk_mul_f32 v, v, -v", out, w, mhz, p.multiProcessorCount);
        run<2>("v_pk_add_f32 v, v, v", out, w, mhz, p.multiProcessorCount);
        run<3>("v_pk_add_f32 v, v, s", out, w, mhz, p.multiProcessorCount);
        run<5>("v_pk_fma_f32 v, v, v, v", out, w, mhz, p.multiProcessorCount);
        run<6>("v_pk_fma_f32 v, v, s, v", out, w, mhz, p.multiProcessorCount);
    }
    CHECK(hipFree(out));
    return 0;
}
