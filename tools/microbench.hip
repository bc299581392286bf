// microbench.hip -- per-instruction issue cost of the operations the PG loop is made of (gfx950).  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int OP> __global__ void k(float* out, int iters, uint32_t seed)
{
    uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9E3779B9u, c = a + 77u, d = b + 99u;
    float x = 1.0f + (a & 1023) * 1e-3f, y = 0.5f + (b & 1023) * 1e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (OP == 0) { x = fmaf(x, 1.0000001f, y); y = fmaf(y, 0.9999999f, x); }                       // 2 fma
            if (OP == 1) { uint64_t p = (uint64_t)0xD2511F53u * a; uint64_t q = (uint64_t)0xCD9E8D57u * c; a = (uint32_t)(q >> 32) ^ b; b = (uint32_t)q; c = (uint32_t)(p >> 32) ^ d; d = (uint32_t)p; }   // philox round: 2 mad_u64 + 2 xor
            if (OP == 2) { x = __builtin_amdgcn_exp2f(x * 0.5f); y = __builtin_amdgcn_exp2f(y * 0.25f); }   // 2 exp + 2 mul
            if (OP == 3) { x = __builtin_amdgcn_logf(x + 2.0f); y = __builtin_amdgcn_logf(y + 3.0f); }      // 2 log + 2 add
            if (OP == 4) { x = __builtin_amdgcn_rcpf(x + 2.0f); y = __builtin_amdgcn_rcpf(y + 3.0f); }
            if (OP == 5) { x = __builtin_amdgcn_sqrtf(x + 2.0f); y = __builtin_amdgcn_sqrtf(y + 3.0f); }
            if (OP == 6) { a = __umulhi(a, 0xD2511F53u) ^ b; b = b * 0xCD9E8D57u + a; }                     // mul_hi + mul_lo
            if (OP == 7) { a = (a & 0xFFFFFF) * (b & 0xFFFFFF) + c; b = b ^ a; }                            // mad_u32_u24
            if (OP == 8) { a ^= b; b ^= c; c ^= d; d ^= a; }                                                // 4 xor
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x + y + (float)(a ^ b ^ c ^ d);
}
template <int OP> void run(const char* name, float* d)
{
    const int iters = 2000; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<OP>), dim3(1024), dim3(1024), 0, 0, d, 10, 1u); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL((k<OP>), dim3(1024), dim3(1024), 0, 0, d, iters, 1u); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // waves = 1024*1024/64 = 16384; per SIMD (1024 SIMDs) 16 waves sequential-ish at 4 resident; groups of 16 per iter
    double wave_groups = 16384.0 * iters * 16;           // number of (wave, unrolled-op-group) executions
    double ns_per_group_per_simd = ms * 1e6 / (wave_groups / 1024.0);
    printf("%-22s %.3f ms   %.2f ns per op-group per SIMD (~%.1f cycles @2.4GHz)\n", name, ms, ns_per_group_per_simd, ns_per_group_per_simd * 2.4);
}
int main() { float* d; hipMalloc(&d, 1024 * 1024 * 4);
    run<0>("2 fma", d); run<1>("philox round", d); run<2>("2 exp + 2 mul", d); run<3>("2 log + 2 add", d); run<4>("2 rcp + 2 add", d);
    run<5>("2 sqrt + 2 add", d); run<6>("mul_hi + mul_lo + xor", d); run<7>("mad_u24 + and + xor", d); run<8>("4 xor", d); return 0; }
