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

// lanes_on: a 64-bit mask of the lanes that gather (the others sit the loop out under exec, as the lanes of a walk that
// have finished or hold a leaf do): does a gather instruction cost by its active lanes or by the instruction?
__global__ __launch_bounds__(64) void gather_kernel(const float4 *__restrict__ table, uint32_t mask, int iters, float *sink,
                                                    unsigned long long lanes_on, int share)
{
    // share: groups of `share` neighbouring lanes read the same record (the lanes of a wave that stand at the same node
    // of the index: all of them at the root, fewer and fewer further down)
    uint32_t x = (blockIdx.x * 64u + threadIdx.x / (unsigned)share) * 2654435761u + 12345u;
    float acc = 0.f;
    if ((lanes_on >> (threadIdx.x & 63)) & 1ull)
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
    // 8 workgroups per wave slot: a grid of exactly one per slot fills the SIMDs unevenly and reads 10-20 % low
    // (profiles/r02_issue_costs.txt, 7)
    const int rounds = 8;
    struct Pat { const char *name; unsigned long long m; int lanes; } pats[] = {
        {"all 64 lanes", ~0ull, 64}, {"lanes 0-31", 0xffffffffull, 32}, {"every 2nd lane", 0x5555555555555555ull, 32},
        {"lanes 0-15", 0xffffull, 16}, {"every 4th lane", 0x1111111111111111ull, 16}};
    for (int kb : kbs) {
        const size_t n = (size_t)kb * 1024 / 16;                 // records (a power of two)
        float4 *table; CK(hipMalloc(&table, n * 16)); CK(hipMemset(table, 0, n * 16));
        for (const Pat &pt : pats)
        for (int w : {6, 8}) {
            const int grid = simds * w * rounds, iters = 4000 / rounds;
            hipLaunchKernelGGL(gather_kernel, dim3(grid), dim3(64), 0, 0, table, (uint32_t)(n - 1), iters / 4, sink, pt.m, 1);
            CK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(gather_kernel, dim3(grid), dim3(64), 0, 0, table, (uint32_t)(n - 1), iters, sink, pt.m, 1);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double gathers = (double)grid * pt.lanes * iters * 4, instrs = (double)grid * iters * 4;
            printf("table %5d KB  %-15s %d waves/SIMD  %7.3f ms  %7.1f G lane-gathers/s = %6.2f TB/s;  %6.2f G wave-instructions/s = %5.1f cycles per instruction per CU at 2.4 GHz\n",
                   kb, pt.name, w, best, gathers / best / 1e6, gathers * 16 / best / 1e9, instrs / best / 1e6,
                   2.4e9 * p.multiProcessorCount / (instrs / (best * 1e-3)));
        }
        // lanes that share a record: does the cache look a line up once per lane or once per distinct line?
        for (int share : {2, 4, 16, 64}) {
            const int w = 8, grid = simds * w * rounds, iters = 4000 / rounds;
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(gather_kernel, dim3(grid), dim3(64), 0, 0, table, (uint32_t)(n - 1), iters, sink, ~0ull, share);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            const double gathers = (double)grid * 64 * iters * 4;
            printf("table %5d KB  all 64 lanes, groups of %2d lanes read one record  %d waves/SIMD  %7.3f ms  %7.1f G lane-gathers/s = %6.2f TB/s (%5.1f G distinct records/s)\n",
                   kb, share, w, best, gathers / best / 1e6, gathers * 16 / best / 1e9, gathers / share / best / 1e6);
        }
        CK(hipFree(table));
    }
    return 0;
}
