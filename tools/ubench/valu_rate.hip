// Micro-benchmark (GPU box): issue cost of plain (non-packed) fp32 VALU instructions on gfx950 at 1..8 waves per SIMD.
// Answers: does a wave64 v_add/v_mul/v_fma_f32 occupy the SIMD for 2 or for 4 cycles?  (sets the VALU roofline of the renderer)
// build: hipcc -O3 --offload-arch=gfx950 -o valu_rate valu_rate.hip ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND>
__global__ void __launch_bounds__(64) k_valu(float *out, int iters, unsigned long long *cyc)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float c = 1.0000001f, d = 0.5f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++)
    {
#pragma unroll
        for (int u = 0; u < 8; u++)
        {
            if (KIND == 0) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(a0) : "v"(c)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a1) : "v"(c));
                             asm volatile("v_add_f32 %0, %0, %1" : "+v"(a2) : "v"(c)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a3) : "v"(c));
                             asm volatile("v_add_f32 %0, %0, %1" : "+v"(a4) : "v"(c)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a5) : "v"(c));
                             asm volatile("v_add_f32 %0, %0, %1" : "+v"(a6) : "v"(c)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a7) : "v"(c)); }
            if (KIND == 1) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(c), "v"(d)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(c), "v"(d));
                             asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(c), "v"(d)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(c), "v"(d));
                             asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a4) : "v"(c), "v"(d)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a5) : "v"(c), "v"(d));
                             asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a6) : "v"(c), "v"(d)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a7) : "v"(c), "v"(d)); }
            if (KIND == 2) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a0) : "v"(c)); asm volatile("v_and_b32 %0, %0, %1" : "+v"(a1) : "v"(c));
                             asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a2), "v"(c) : "vcc"); asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a3) : "v"(c));
                             asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a4) : "v"(c)); asm volatile("v_mov_b32 %0, %1" : "+v"(a5) : "v"(c));
                             asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(a6), "v"(c) : "vcc"); asm volatile("v_or_b32 %0, %0, %1" : "+v"(a7) : "v"(c)); }
            if (KIND == 3) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(a0) : "v"(c)); asm volatile("s_add_u32 s20, s20, 1" ::: "s20", "scc");
                             asm volatile("v_add_f32 %0, %0, %1" : "+v"(a2) : "v"(c)); asm volatile("s_and_b32 s21, s21, s20" ::: "s21", "scc");
                             asm volatile("v_add_f32 %0, %0, %1" : "+v"(a4) : "v"(c)); asm volatile("s_lshl_b32 s22, s20, 1" ::: "s22", "scc");
                             asm volatile("v_add_f32 %0, %0, %1" : "+v"(a6) : "v"(c)); asm volatile("s_or_b32 s23, s23, s22" ::: "s23", "scc"); }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
static void run(const char *name, int waves_per_simd)
{
    const int iters = 2000;
    const int blocks = 256 * 4 * waves_per_simd;
    float *out; unsigned long long *cyc;
    hipMalloc(&out, blocks * 64 * 4); hipMalloc(&cyc, blocks * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_valu<KIND><<<blocks, 64>>>(out, 10, cyc); hipDeviceSynchronize();
    hipEventRecord(e0); k_valu<KIND><<<blocks, 64>>>(out, iters, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks); hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += v; mean /= blocks;
    const double n_inst = (double)iters * 64;       // instructions of the loop body per wave (KIND 3: 32 VALU + 32 SALU)
    printf("%-28s waves/SIMD %d: wave cycles/instr %.2f -> SIMD cycles per instr %.2f ; wall %.3f ms (%.2f cyc/instr/SIMD at 2.4 GHz)\n",
           name, waves_per_simd, mean / n_inst, mean / n_inst / waves_per_simd, ms, ms * 1e-3 * 2.4e9 / (n_inst * waves_per_simd));
    fflush(stdout);
    hipFree(out); hipFree(cyc);
}

int main()
{
    for (int w : {1, 2, 4, 8}) run<0>("v_add_f32 x8 independent", w);
    for (int w : {1, 2, 4, 8}) run<1>("v_fma_f32 x8 independent", w);
    for (int w : {1, 2, 4, 8}) run<2>("cmp/cndmask/logic mix", w);
    for (int w : {1, 2, 4, 8}) run<3>("v_add + salu interleaved", w);
    return 0;
}
