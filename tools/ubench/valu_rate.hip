// Issue rate of a few VALU instructions on gfx950 (diagnostics): cycles per wave-instruction per SIMD with 4 waves per SIMD resident.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o gpurun_out/valu_rate ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int OP> __global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed)
{
    uint32_t a[8]; float f[8]; double d[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 8 + i; f[i] = 1.0f + (float)a[i] * 1e-9f; d[i] = 1.0 + (double)a[i] * 1e-12; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) { uint64_t p = (uint64_t)a[i] * 0xD2511F53u + (uint64_t)it; a[i] = (uint32_t)(p >> 32) ^ (uint32_t)p; }      // v_mad_u64_u32 + xor
            if (OP == 1) { f[i] = fmaf(f[i], 1.0000001f, 1e-7f); }
            if (OP == 2) { d[i] = fma(d[i], 1.0000001, 1e-7); }
            if (OP == 3) { a[i] = a[i] * 0xCD9E8D57u + 12345u; }                                                                        // v_mul_lo_u32 (+add)
            if (OP == 4) { f[i] = __builtin_amdgcn_exp2f(f[i]) ; }
            if (OP == 5) { a[i] = (a[i] ^ (a[i] >> 7)) + 0x9E3779B9u; }                                                                 // 2-3 full-rate int ops
            if (OP == 6) { d[i] = d[i] * d[i] + 1e-7; }
            if (OP == 7) { a[i] = __umulhi(a[i], 0xD2511F53u) + 1u; }
            if (OP == 8) { d[i] = d[i] + 1e-7; }                                                                                        // v_add_f64
            if (OP == 9) { d[i] = d[i] * 1.0000001; }                                                                                   // v_mul_f64
            if (OP == 10) { d[i] = __builtin_amdgcn_rcp(d[i]); }                                                                        // v_rcp_f64 (transcendental, fp64)
            if (OP == 11) { f[i] = __builtin_amdgcn_rcpf(f[i]); }                                                                       // v_rcp_f32
            if (OP == 12) { d[i] = (double)a[i]; a[i] = (uint32_t)__double2hiint(d[i]) + (uint32_t)it; }                                // v_cvt_f64_u32 (+ add)
            if (OP == 13) { asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]), "v"((uint32_t)it)); }
            if (OP == 14) { uint64_t p = (uint64_t)a[i] * 0xD2511F53u + (uint64_t)a[(i + 1) & 7]; a[i] = (uint32_t)(p >> 32); }          // v_mad_u64_u32 alone
            if (OP == 15) { a[i] = (a[i] & 1u) ? a[(i + 1) & 7] : a[(i + 2) & 7]; a[i] += (uint32_t)it; }                                // v_cndmask (+ and, cmp, add)
            if (OP == 16) { f[i] = f[i] + 1e-7f; }                                                                                      // v_add_f32
            if (OP == 17) { f[i] = __builtin_amdgcn_logf(f[i]) + 2.0f; }                                                                // v_log_f32 (+ add)
        }
    }
    uint32_t r = 0; for (int i = 0; i < 8; ++i) r ^= a[i] ^ __float_as_uint(f[i]) ^ (uint32_t)__double_as_longlong(d[i]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int OP> void run(const char* name, int nper)
{
    uint32_t* out; hipMalloc(&out, 256 * 4 * 256 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<1024, 256>>>(out, 100, 1u);                 // 4 workgroups of 4 waves per CU: 4 waves per SIMD
    hipDeviceSynchronize();
    hipEventRecord(e0); k<OP><<<1024, 256>>>(out, iters, 1u); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD: 4 waves x iters x 8 x nper
    const double insts = 4.0 * iters * 8.0 * nper;
    printf("%-28s %8.3f ms  -> %.2f cycles per wave-instruction at 2.4 GHz (%d instr per element-step)\n", name, ms, ms * 1e-3 * 2.4e9 / insts, nper);
    hipFree(out);
}
int main()
{
    run<1>("v_fma_f32", 1); run<2>("v_fma_f64", 1); run<6>("v_fma_f64 (x*x+c)", 1); run<0>("v_mad_u64_u32 + v_xor", 2); run<3>("v_mul_lo_u32 + add", 2); run<7>("v_mul_hi_u32 + add", 2);
    run<4>("v_exp_f32", 1); run<5>("xor-shift-add (3 int ops)", 3);
    run<8>("v_add_f64", 1); run<9>("v_mul_f64", 1); run<10>("v_rcp_f64", 1); run<11>("v_rcp_f32", 1); run<12>("v_cvt_f64_u32 + hi + add", 3); run<13>("v_bitop3_b32", 1);
    run<14>("v_mad_u64_u32", 1); run<15>("v_and + v_cmp + v_cndmask + v_add", 4); run<16>("v_add_f32", 1); run<17>("v_log_f32 + add", 2);
    return 0;
}
