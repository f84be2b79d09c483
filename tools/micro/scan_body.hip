// The closest-hit scan on its own, in steady state: every wave scans the eight surfaces of the
// Cornell box (5 triangles + 3 spheres, src/scene.hpp:129-156) for N rays per lane, ITERS times
// over, with nothing else in the loop (no ray set-up, no stores, no wave start or end).  Prints
// SIMD cycles per 64 rays x 8 surfaces and the fraction of the FP32 vector peak that is, priced
// like the benchmark (63 flop per triangle test, 18 per sphere test).
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -fno-slp-vectorize -o scan_body scan_body.hip
#include "../../path-tracer_amd/csrc/pt_query.hpp"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace ptd;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ unsigned long long g_redone;
__device__ long long g_ticks[2];        // shader cycles and 100 MHz ticks one wave spent in the loop

template <bool FAST, int N>
__global__ void __launch_bounds__(256) k_scan(const float *recs, int n_surfaces, int iters, float *out)
{
    extern __shared__ __align__(16) float lds[];
    query_stage_records(recs, n_surfaces, lds);
    constexpr int M = N > 0 ? N : 1;
    V3 o[M], d[M];
    const int t = blockIdx.x * 256 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < M; k++) {
        // rays from near the camera of the scene into the box, a different one per lane and k
        const float a = (float)((t * M + k) % 9973) * (1.0f / 9973.0f) - 0.5f;
        const float b = (float)((t * M + k) % 7919) * (1.0f / 7919.0f) - 0.5f;
        o[k] = v3(0.01f * a, 0.01f * b, -3.0f);
        d[k] = Num<true>::normalize(v3(a, b, 1.0f));
    }
    float sum = 0.0f;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
        Hit h[N > 0 ? N : 1];
        if constexpr (N > 0) {
            if constexpr (FAST) closest_hits_lean<N>(lds, n_surfaces, o, d, h);
            else closest_hits_exact<N>(lds, as_constant(recs), n_surfaces, o, d, h, &g_redone);
        } else {
            // the one-ray scan with records through the scalar cache (the render kernels' form)
            if constexpr (FAST) h[0] = closest_intersection_lean<false, false>(as_constant(recs), n_surfaces, o[0], d[0]);
            else h[0] = closest_intersection<false, false>(as_constant(recs), n_surfaces, o[0], d[0], &g_redone);
        }
#pragma unroll
        for (int k = 0; k < M; k++) {
            sum += h[k].dist + h[k].t;
            // the next round's ray depends on this round's result: nothing can be hoisted
            o[k].x = __builtin_fmaf(h[k].dist, 1e-9f, o[k].x);
            d[k].y = __builtin_fmaf((float)h[k].idx, 1e-9f, d[k].y);
        }
    }
    out[t] = sum;
    if (t == 0) { g_ticks[0] = clock64() - c0; g_ticks[1] = wall_clock64() - w0; }
}

static void lean_from_geom(const float *g, float *ln)
{
    for (int k = 0; k < 16; k++) ln[k] = 0.0f;
    ln[3] = g[3];
    ln[12] = g[12];
    int kind;
    memcpy(&kind, &g[3], 4);
    if (kind == GEOM_TRIANGLE) {
        const double v0[3] = {g[0], g[1], g[2]}, e1[3] = {g[4], g[5], g[6]}, e2[3] = {g[8], g[9], g[10]};
        const double Nn[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
        const double nn = Nn[0] * Nn[0] + Nn[1] * Nn[1] + Nn[2] * Nn[2], len = sqrt(nn);
        const double ku[3] = {(e2[1] * Nn[2] - e2[2] * Nn[1]) / nn, (e2[2] * Nn[0] - e2[0] * Nn[2]) / nn, (e2[0] * Nn[1] - e2[1] * Nn[0]) / nn};
        const double kv[3] = {(Nn[1] * e1[2] - Nn[2] * e1[1]) / nn, (Nn[2] * e1[0] - Nn[0] * e1[2]) / nn, (Nn[0] * e1[1] - Nn[1] * e1[0]) / nn};
        double d0 = 0, cu = 0, cv = 0;
        for (int k = 0; k < 3; k++) {
            d0 += Nn[k] / len * v0[k]; cu -= ku[k] * v0[k]; cv -= kv[k] * v0[k];
            ln[k] = (float)(Nn[k] / len); ln[6 + 2 * k] = (float)ku[k]; ln[7 + 2 * k] = (float)kv[k];
        }
        ln[4] = (float)d0; ln[5] = (float)(1e-5 / len + 1e-5); ln[14] = (float)cu; ln[15] = (float)cv;
    } else {
        ln[0] = g[0]; ln[1] = g[1]; ln[2] = g[2]; ln[4] = g[4];
    }
}

template <bool FAST, int N>
static void run(const float *d_geom, const float *d_lean, float *out, int waves_per_simd, int cus, double mhz)
{
    const int iters = 400, n_surf = 8;
    const int blocks = cus * waves_per_simd;
    const float *recs = FAST ? d_lean : d_geom;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k_scan<FAST, N><<<blocks, 256, n_surf * 64>>>(recs, n_surf, 20, out);
    CHECK(hipEventRecord(e0));
    k_scan<FAST, N><<<blocks, 256, n_surf * 64>>>(recs, n_surf, iters, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double wave_rays = (double)iters * (N > 0 ? N : 1) * waves_per_simd;          // per SIMD
    const double cycles = ms * 1e-3 * mhz * 1e6 / wave_rays;
    const double flops = 64.0 * (5 * 63 + 3 * 18);
    unsigned long long redone = 0;
    CHECK(hipMemcpyFromSymbol(&redone, HIP_SYMBOL(g_redone), sizeof redone));
    const unsigned long long zero = 0;
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_redone), &zero, sizeof zero));
    hipFuncAttributes attr{};
    CHECK(hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&k_scan<FAST, N>)));
    printf("%s N=%d waves/SIMD=%d (%3d VGPRs): %7.1f cycles per 64 rays x 8 surfaces = %.3f of the FP32 peak (64 flop/cycle/SIMD)",
           FAST ? "FAST " : "EXACT", N, waves_per_simd, attr.numRegs, cycles, flops / cycles / 64.0);
    long long ticks[2] = {0, 1};
    CHECK(hipMemcpyFromSymbol(ticks, HIP_SYMBOL(g_ticks), sizeof ticks));
    printf(" [shader clock in the loop %.2f GHz]", (double)ticks[0] / (double)ticks[1] * 0.1);
    if (!FAST) printf("; %.2e of the rays evaluated directly", (double)redone / (wave_rays * 64.0 * cus * 4 * (20 + iters) / iters));
    printf("\n");
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const double mhz = p.clockRate / 1000.0;
    printf("%s, %d CUs, %.0f MHz\n", p.name, p.multiProcessorCount, mhz);
    // Scenes::cornell_2's surfaces (src/scene.hpp:129-156)
    const float tri[5][9] = {{-2, 1, -1, 0, 1, 10, 2, 1, -1}, {-1, -2, -1, -1, 0, 10, -1, 2, -1}, {1, -2, -1, 1, 2, -1, 1, 0, 10},
                             {-2, -1, -1, 2, -1, -1, 0, -1, 10}, {-2, 1, 1, 0, -10, 1, 2, 1, 1}};
    const float sph[3][4] = {{-.32f, .6f, .26f, .4f}, {.35f, .6f, -.39f, .4f}, {0, -1.8f, 0, 1}};
    std::vector<float> geom(8 * 16, 0.0f), lean(8 * 16, 0.0f);
    for (int i = 0; i < 5; i++) {
        float *g = &geom[i * 16];
        const float *t = tri[i];
        for (int k = 0; k < 3; k++) { g[k] = t[k]; g[4 + k] = t[3 + k] - t[k]; g[8 + k] = t[6 + k] - t[k]; }
        g[7] = g[5] * g[10] - g[9] * g[6];
        const int kind = GEOM_TRIANGLE, run = 5 - i;
        memcpy(&g[3], &kind, 4); memcpy(&g[12], &run, 4);
    }
    for (int i = 0; i < 3; i++) {
        float *g = &geom[(5 + i) * 16];
        for (int k = 0; k < 3; k++) g[k] = sph[i][k];
        g[4] = sph[i][3] * sph[i][3];
        const int kind = GEOM_SPHERE, run = 3 - i;
        memcpy(&g[3], &kind, 4); memcpy(&g[12], &run, 4);
    }
    for (int i = 0; i < 8; i++) lean_from_geom(&geom[i * 16], &lean[i * 16]);
    float *d_geom, *d_lean, *out;
    CHECK(hipMalloc(&d_geom, geom.size() * 4)); CHECK(hipMalloc(&d_lean, lean.size() * 4));
    CHECK(hipMalloc(&out, (size_t)p.multiProcessorCount * 8 * 256 * 4));
    CHECK(hipMemcpy(d_geom, geom.data(), geom.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_lean, lean.data(), lean.size() * 4, hipMemcpyHostToDevice));
    const int cus = p.multiProcessorCount;
    for (int w : {2, 4, 8}) {
        run<true, 0>(d_geom, d_lean, out, w, cus, mhz);
        run<false, 0>(d_geom, d_lean, out, w, cus, mhz);
        run<true, 1>(d_geom, d_lean, out, w, cus, mhz);
        run<true, 2>(d_geom, d_lean, out, w, cus, mhz);
        run<true, 4>(d_geom, d_lean, out, w, cus, mhz);
        if (w <= 4) run<true, 8>(d_geom, d_lean, out, w, cus, mhz);
        run<false, 1>(d_geom, d_lean, out, w, cus, mhz);
        run<false, 2>(d_geom, d_lean, out, w, cus, mhz);
        if (w <= 4) run<false, 4>(d_geom, d_lean, out, w, cus, mhz);
    }
    return 0;
}
