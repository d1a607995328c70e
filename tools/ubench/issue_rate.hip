// Micro-benchmark (GPU box): what does a gfx950 SIMD issue per second from wave64 fp32 VALU streams?
// MI355X_MICROARCH.md: SIMD-32, a wave64 VALU instruction takes 2 cycles on its SIMD (one wave alone: 4).  Round 2's
// tools/ubench/valu_rate.hip reported 0.9-0.95 G instr/s per SIMD -- between 0.6 (4 cycles) and 1.2 (2 cycles at 2.4 GHz) -- from
// short (0.3-1 ms) launches of single-wave workgroups timed by HIP events.  This one removes the unknowns: 256-thread workgroups
// (one wave per SIMD each), W of them per CU, runs of >= 5 ms, and every wave stamps BOTH clocks at its start and end:
//   s_memtime      shader-clock ticks (what the guide calls a cycle)
//   s_memrealtime  100 MHz, constant
// -> ticks per instruction per SIMD (the guide's number), the clock the chip actually ran at under this load (ticks per 10 ns),
//    and wave-instructions per second per SIMD (what the renderer's issue roofline is priced against).
// build: hipcc -O3 --offload-arch=gfx950 -o issue_rate issue_rate.hip ; run: ./issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int KIND>
__global__ void __launch_bounds__(256) k_issue(float *out, int iters, unsigned long long *stamp)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float c = 1.0000001f, d = 0.5f;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++)
    {
#pragma unroll
        for (int u = 0; u < 8; u++)
        {
            if (KIND == 0) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(a0) : "v"(c)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a1) : "v"(c));
                             asm volatile("v_add_f32 %0, %0, %1" : "+v"(a2) : "v"(c)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a3) : "v"(c));
                             asm volatile("v_add_f32 %0, %0, %1" : "+v"(a4) : "v"(c)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a5) : "v"(c));
                             asm volatile("v_add_f32 %0, %0, %1" : "+v"(a6) : "v"(c)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a7) : "v"(c)); }
            if (KIND == 1) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(c), "v"(d)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(c), "v"(d));
                             asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(c), "v"(d)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(c), "v"(d));
                             asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a4) : "v"(c), "v"(d)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a5) : "v"(c), "v"(d));
                             asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a6) : "v"(c), "v"(d)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a7) : "v"(c), "v"(d)); }
            if (KIND == 2) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(a0) : "v"(c)); asm volatile("s_add_u32 s20, s20, 1" ::: "s20", "scc");
                             asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a2) : "v"(c)); asm volatile("s_and_b32 s21, s21, s20" ::: "s21", "scc");
                             asm volatile("v_add_f32 %0, %0, %1" : "+v"(a4) : "v"(c)); asm volatile("s_lshl_b32 s22, s20, 1" ::: "s22", "scc");
                             asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a6) : "v"(c)); asm volatile("s_or_b32 s23, s23, s22" ::: "s23", "scc"); }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if ((threadIdx.x & 63) == 0)
    {
        unsigned long long *o = stamp + (size_t)(blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
        o[0] = t0; o[1] = t1; o[2] = r0; o[3] = r1;
    }
}

template <int KIND>
static void run(const char *name, int wg_per_cu, int iters)
{
    const int blocks = 256 * wg_per_cu;
    float *out; unsigned long long *stamp;
    hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&stamp, (size_t)blocks * 4 * 4 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_issue<KIND><<<blocks, 256>>>(out, 100, stamp); hipDeviceSynchronize();
    hipEventRecord(e0); k_issue<KIND><<<blocks, 256>>>(out, iters, stamp); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)blocks * 16); hipMemcpy(h.data(), stamp, h.size() * 8, hipMemcpyDeviceToHost);
    const int waves = blocks * 4;
    double ticks = 0, real = 0; unsigned long long rmin = ~0ull, rmax = 0;
    for (int w = 0; w < waves; w++)
    {
        ticks += (double)(h[w * 4 + 1] - h[w * 4 + 0]); real += (double)(h[w * 4 + 3] - h[w * 4 + 2]);
        rmin = std::min(rmin, h[w * 4 + 2]); rmax = std::max(rmax, h[w * 4 + 3]);
    }
    ticks /= waves; real /= waves;
    const double n_inst = (double)iters * 64;                  // instructions of the loop body per wave
    const double span_s = (double)(rmax - rmin) * 1e-8;        // first stamp to last stamp of the launch, 100 MHz
    printf("%-26s waves/SIMD %d: wave ticks/instr %.2f -> per SIMD %.2f ticks/instr; shader clock %.0f MHz (ticks per 100 MHz tick); "
           "launch %.3f ms by events, %.3f ms by stamps -> %.3f G wave-instr/s per SIMD\n",
           name, wg_per_cu, ticks / n_inst, ticks / n_inst / wg_per_cu, ticks / real * 100.0, ms, span_s * 1e3,
           n_inst * wg_per_cu / span_s / 1e9);
    fflush(stdout);
    hipFree(out); hipFree(stamp);
}

int main()
{
    for (int w : {1, 2, 3, 4, 8}) run<0>("v_add/v_mul x8 independent", w, 100000 / w);
    for (int w : {1, 2, 4, 8}) run<1>("v_fma_f32 x8 independent", w, 100000 / w);
    for (int w : {1, 2, 4, 8}) run<2>("VALU + SALU interleaved", w, 100000 / w);
    return 0;
}
