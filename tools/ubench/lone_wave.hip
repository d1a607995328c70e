// Micro-benchmark (GPU box): what a LONE wave64 on a SIMD pays per instruction on gfx950 -- the regime of the slowest
// footprints of a frame, whose serial chain sets the isolated launch time.  One wave per CU (256 blocks of 64).
// Stamps with s_memtime (shader cycles).  build: hipcc -O3 --offload-arch=gfx950 -o lone_wave lone_wave.hip ; run: ./lone_wave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ void __launch_bounds__(64) k(float *out, const unsigned *chain, int iters, unsigned long long *cyc)
{
    float a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3;
    const float m = 1.0000001f, e = 0.5f;
    unsigned p = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++)
    {
        if (KIND == 0) asm volatile(REP64("v_fma_f32 %0, %0, %1, %2\n") : "+v"(a) : "v"(m), "v"(e));                    // dependent VALU chain
        if (KIND == 1) asm volatile(REP8(REP8("v_fma_f32 %0, %0, %4, %5\nv_fma_f32 %1, %1, %4, %5\nv_fma_f32 %2, %2, %4, %5\nv_fma_f32 %3, %3, %4, %5\n"))
                                    : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(e));                             // 4 independent chains (256 instr)
        if (KIND == 2) asm volatile(REP64("s_add_u32 s20, s20, 1\n") ::: "s20", "scc");                                // dependent SALU
        if (KIND == 3) asm volatile(REP64("s_cmp_eq_u32 s20, s20\ns_cbranch_scc1 1f\ns_nop 0\n1:\n") ::: "scc");        // taken forward branch (skips 1)
        if (KIND == 4) asm volatile(REP64("s_cmp_lg_u32 s20, s20\ns_cbranch_scc1 1f\n1:\n") ::: "scc");                // not-taken branch
        if (KIND == 5) asm volatile(REP64("v_cmp_gt_f32 vcc, %0, %1\ns_and_b64 s[20:21], vcc, exec\nv_cndmask_b32 %0, %0, %1, s[20:21]\n")
                                    : "+v"(a) : "v"(m) : "vcc", "s20", "s21", "scc");                                   // VALU -> SALU -> VALU round trip
        if (KIND == 6) { for (int u = 0; u < 64; u++) { p = __builtin_amdgcn_readfirstlane(p); p = *(const __attribute__((address_space(4))) unsigned *)((const __attribute__((address_space(4))) char *)chain + p); } }  // dependent scalar loads, cache hits
        if (KIND == 7) asm volatile(REP64("v_cmp_gt_f32 vcc, %0, %1\ns_cbranch_vccz 1f\nv_add_f32 %0, %0, %1\n1:\n") : "+v"(a) : "v"(m) : "vcc");   // compare + branch on vcc (not taken) + VALU
        if (KIND == 8) asm volatile(REP64("s_and_saveexec_b64 s[20:21], vcc\ns_cbranch_execz 1f\nv_add_f32 %0, %0, %1\n1:\ns_or_b64 exec, exec, s[20:21]\n") : "+v"(a) : "v"(m) : "vcc", "s20", "s21", "scc");  // divergent-if skeleton
        if (KIND == 9) asm volatile(REP64("v_mul_f32 %0, %0, %1\ns_add_u32 s20, s20, 1\n") : "+v"(a) : "v"(m) : "s20", "scc");                       // dependent VALU interleaved with SALU
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d + (float)p;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
static void run(const char *name, double per_iter, const unsigned *chain)
{
    const int iters = 200, blocks = 256;
    float *out; unsigned long long *cyc;
    hipMalloc(&out, blocks * 64 * 4); hipMalloc(&cyc, blocks * 8);
    k<KIND><<<blocks, 64>>>(out, chain, 10, cyc); hipDeviceSynchronize();
    k<KIND><<<blocks, 64>>>(out, chain, iters, cyc); hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks); hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-64s %7.2f cycles per unit (median over %d lone waves; %g units per iteration)\n", name, (double)h[blocks / 2] / (iters * per_iter), blocks, per_iter);
    fflush(stdout);
    hipFree(out); hipFree(cyc);
}

int main()
{
    unsigned *chain; hipMalloc(&chain, 4096);
    std::vector<unsigned> hc(1024);
    for (int i = 0; i < 1024; i++) hc[i] = (unsigned)(((i * 16 + 64) % 1024) * 4);      // a walk through 4 KB in 64-byte steps: stays in the scalar cache
    hipMemcpy(chain, hc.data(), 4096, hipMemcpyHostToDevice);
    run<0>("dependent v_fma_f32 chain", 64, chain);
    run<1>("v_fma_f32, four independent chains", 256, chain);
    run<2>("dependent s_add_u32 chain", 64, chain);
    run<3>("s_cmp + taken s_cbranch_scc1 (forward, over one instruction)", 64, chain);
    run<4>("s_cmp + not-taken s_cbranch_scc1", 64, chain);
    run<5>("v_cmp -> s_and_b64 -> v_cndmask (mask round trip)", 64, chain);
    run<6>("dependent s_load_dword chain, scalar-cache hits (+ readfirstlane)", 64, chain);
    run<7>("v_cmp + s_cbranch_vccz (not taken) + v_add", 64, chain);
    run<8>("s_and_saveexec + s_cbranch_execz (not taken) + v_add + s_or exec", 64, chain);
    run<9>("dependent v_mul_f32 interleaved with s_add_u32 (pairs)", 64, chain);
    return 0;
}
