// Does a kernel launched with hipExtAnyOrderLaunch (no barrier bit) start while its predecessor in the SAME stream still runs?  (diagnostics)
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/anyorder.hip -o gpurun_out/anyorder ; run on the GPU box
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
// every workgroup fills a CU (1024 threads, 100 KB of LDS); workgroup 0 stays `ticks` of the 100 MHz wall clock, the others leave at once
__global__ void __launch_bounds__(1024) spin_kernel(unsigned long long* out, unsigned int ticks)
{
    extern __shared__ unsigned char sm[];
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) sm[0] = 1;
    if (blockIdx.x == 0) { while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8); }
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t0; out[2 * blockIdx.x + 1] = wall_clock64(); }
}
__global__ void __launch_bounds__(1024) probe_kernel(unsigned long long* out)
{
    extern __shared__ unsigned char sm[];
    if (threadIdx.x == 0) { sm[0] = 1; out[blockIdx.x] = wall_clock64(); }
}
static void report(const char* what, const std::vector<unsigned long long>& a, const std::vector<unsigned long long>& b, int n)
{
    unsigned long long end0 = a[1], endmax = 0, first = ~0ull, last = 0; int early = 0;
    for (int i = 0; i < n; ++i) endmax = std::max(endmax, a[2 * i + 1]);
    for (int i = 0; i < n; ++i) { first = std::min(first, b[i]); last = std::max(last, b[i]); if (b[i] < end0) ++early; }
    printf("%-44s predecessor: workgroup 0 ends at +%.1f us; successor's workgroups start at +%.1f .. +%.1f us, %d of %d before the predecessor ended\n", what,
           (end0 - a[0]) / 100.0, ((long long)first - (long long)a[0]) / 100.0, ((long long)last - (long long)a[0]) / 100.0, early, n);
}
int main()
{
    const int n = 256; const size_t lds = 100 * 1024;
    unsigned long long *da, *db; CK(hipMalloc(&da, 2 * n * 8)); CK(hipMalloc(&db, n * 8));
    CK(hipFuncSetAttribute((const void*)spin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void*)probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipStream_t s, s2; CK(hipStreamCreate(&s)); CK(hipStreamCreate(&s2));
    std::vector<unsigned long long> a(2 * n), b(n);
    unsigned int ticks = 5000;   // 50 us
    auto fetch = [&]() { hipDeviceSynchronize(); hipMemcpy(a.data(), da, 2 * n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), db, n * 8, hipMemcpyDeviceToHost); };
    for (int rep = 0; rep < 2; ++rep) {
        // (1) ordinary launches, one stream
        hipLaunchKernelGGL(spin_kernel, dim3(n), dim3(1024), lds, s, da, ticks);
        hipLaunchKernelGGL(probe_kernel, dim3(n), dim3(1024), lds, s, db);
        fetch(); report("same stream, ordinary launch", a, b, n);
        // (2) hipExtAnyOrderLaunch on the successor
        hipLaunchKernelGGL(spin_kernel, dim3(n), dim3(1024), lds, s, da, ticks);
        hipExtLaunchKernelGGL(probe_kernel, dim3(n), dim3(1024), lds, s, nullptr, nullptr, hipExtAnyOrderLaunch, db);
        fetch(); report("same stream, hipExtAnyOrderLaunch", a, b, n);
        // (3) both any-order
        hipExtLaunchKernelGGL(spin_kernel, dim3(n), dim3(1024), lds, s, nullptr, nullptr, hipExtAnyOrderLaunch, da, ticks);
        hipExtLaunchKernelGGL(probe_kernel, dim3(n), dim3(1024), lds, s, nullptr, nullptr, hipExtAnyOrderLaunch, db);
        fetch(); report("same stream, both hipExtAnyOrderLaunch", a, b, n);
        // (4) two streams
        hipLaunchKernelGGL(spin_kernel, dim3(n), dim3(1024), lds, s, da, ticks);
        hipLaunchKernelGGL(probe_kernel, dim3(n), dim3(1024), lds, s2, db);
        fetch(); report("two streams", a, b, n);
        // (5) captured into a graph: does the flag survive?
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(spin_kernel, dim3(n), dim3(1024), lds, s, da, ticks);
        hipExtLaunchKernelGGL(probe_kernel, dim3(n), dim3(1024), lds, s, nullptr, nullptr, hipExtAnyOrderLaunch, db);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s));
        fetch(); report("graph captured from (2)", a, b, n);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
        // (6) a graph whose two kernel nodes have NO edge between them
        {
            hipGraph_t g2; hipGraphExec_t ge2; CK(hipGraphCreate(&g2, 0));
            hipKernelNodeParams p1{}, p2{};
            void* a1[] = {&da, &ticks}; void* a2[] = {&db};
            p1.func = (void*)spin_kernel; p1.gridDim = dim3(n); p1.blockDim = dim3(1024); p1.sharedMemBytes = lds; p1.kernelParams = a1;
            p2.func = (void*)probe_kernel; p2.gridDim = dim3(n); p2.blockDim = dim3(1024); p2.sharedMemBytes = lds; p2.kernelParams = a2;
            hipGraphNode_t n1, n2;
            CK(hipGraphAddKernelNode(&n1, g2, nullptr, 0, &p1));
            CK(hipGraphAddKernelNode(&n2, g2, nullptr, 0, &p2));
            CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge2, s));
            fetch(); report("graph, two kernel nodes without an edge", a, b, n);
            hipGraphExecDestroy(ge2); hipGraphDestroy(g2);
        }
    }
    return 0;
}
