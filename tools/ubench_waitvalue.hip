// ubench_waitvalue.hip -- can a stream wait (hipStreamWaitValue64) for a value that a RUNNING kernel on another
// stream writes?  Kernel A writes the value at its start and then spins for ~2 ms; stream B waits for the value and
// runs kernel B, which notes the wall clock.  "gate works" = B started before A ended.
// Build: hipcc --offload-arch=gfx950 -O2 -o ubench_waitvalue ubench_waitvalue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void kernel_a(unsigned long long *sig, unsigned long long value, unsigned long long *t, unsigned long long spin_ticks)
{
    const unsigned long long t0 = wall_clock64();
    __hip_atomic_store(sig, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    while (wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(32);
    t[0] = t0; t[1] = wall_clock64();
}
__global__ void kernel_b(unsigned long long *t) { t[2] = wall_clock64(); }

int main(int argc, char **argv)
{
    const int plain = argc > 1 ? atoi(argv[1]) : 0;          // 1: hipMalloc memory instead of signal memory
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    unsigned long long *sig = nullptr, *t = nullptr;
    if (plain) CK(hipMalloc((void **)&sig, 8));
    else CK(hipExtMallocWithFlags((void **)&sig, 8, hipMallocSignalMemory));
    CK(hipMalloc((void **)&t, 64));
    CK(hipMemset(t, 0, 64));
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    CK(hipStreamWriteValue64(a, sig, 0, 0));
    CK(hipStreamSynchronize(a));
    for (unsigned long long round = 1; round <= 3; ++round) {
        CK(hipStreamWaitValue64(b, sig, round, hipStreamWaitValueGte, ~0ull));
        hipLaunchKernelGGL(kernel_b, dim3(1), dim3(64), 0, b, t);
        hipLaunchKernelGGL(kernel_a, dim3(1), dim3(64), 0, a, sig, round, t, 200000ull);   // 100 MHz ticks: 2 ms
        CK(hipGetLastError());
        CK(hipStreamSynchronize(a));
        CK(hipStreamSynchronize(b));
        unsigned long long h[3];
        CK(hipMemcpy(h, t, sizeof h, hipMemcpyDeviceToHost));
        printf("round %llu (%s memory): A ran %.1f us; B started %.1f us after A's start: gate %s\n", round,
               plain ? "hipMalloc" : "signal", (h[1] - h[0]) / 100.0, ((double)h[2] - (double)h[0]) / 100.0,
               h[2] < h[1] && h[2] >= h[0] ? "works" : "does NOT open before A ends");
    }
    return 0;
}
