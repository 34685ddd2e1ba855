// ubench_gather.hip -- how fast MI355X serves fully divergent 16-byte per-lane gathers from a table that sits in
// L2 (every lane of a wave-instruction reads a different, random 16-byte record): the access shape of the spatial
// index's walk (one 16-byte node per lane per step, three 16-byte loads per object record; trace_bvh_kernel,
// pathtracer_amd/csrc/pt_kernels.hip).  The peak of the "l2-gather" roofline of bench.py --scene ... lines.
//   ubench_gather [table KB ...]     (defaults: 64 512 4096)
// Each lane keeps four independent gathers in flight; addresses do not depend on loaded data.
// Build: hipcc -O3 --offload-arch=gfx950 -o ubench_gather ubench_gather.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(64) void gather_kernel(const float4 *__restrict__ table, uint32_t mask, int iters, float *sink)
{
    uint32_t x = (blockIdx.x * 64u + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        uint32_t i0 = (x = x * 1664525u + 1013904223u) >> 8, i1 = (x = x * 1664525u + 1013904223u) >> 8;
        uint32_t i2 = (x = x * 1664525u + 1013904223u) >> 8, i3 = (x = x * 1664525u + 1013904223u) >> 8;
        const float4 a = table[i0 & mask], b = table[i1 & mask], c = table[i2 & mask], d = table[i3 & mask];
        acc += a.x + b.y + c.z + d.w;
    }
    if (acc == 123.456f) sink[0] = acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv)
{
    std::vector<int> kbs;
    for (int a = 1; a < argc; ++a) kbs.push_back(atoi(argv[a]));
    if (kbs.empty()) kbs = {64, 512, 4096};
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int simds = p.multiProcessorCount * 4;
    float *sink; CK(hipMalloc(&sink, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("# %s, %d CUs: divergent 16-byte gathers, 4 in flight per lane\n", p.gcnArchName, p.multiProcessorCount);
    for (int kb : kbs) {
        const size_t n = (size_t)kb * 1024 / 16;                 // records (a power of two)
        float4 *table; CK(hipMalloc(&table, n * 16)); CK(hipMemset(table, 0, n * 16));
        for (int w : {4, 5, 6, 8}) {
            const int grid = simds * w, iters = 4000;
            hipLaunchKernelGGL(gather_kernel, dim3(grid), dim3(64), 0, 0, table, (uint32_t)(n - 1), iters / 4, sink);
            CK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(gather_kernel, dim3(grid), dim3(64), 0, 0, table, (uint32_t)(n - 1), iters, sink);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double gathers = (double)grid * 64 * iters * 4;
            printf("table %5d KB  %d waves/SIMD  %7.3f ms  %7.1f G gathers/s  = %6.2f TB/s of 16-byte records\n",
                   kb, w, best, gathers / best / 1e6, gathers * 16 / best / 1e9);
        }
        CK(hipFree(table));
    }
    return 0;
}
