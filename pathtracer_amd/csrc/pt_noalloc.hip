// pt_noalloc.hip -- pt_read_rgb8_noalloc: the one entry point of libpt that
// allocates nothing (include/pt.h), for callers that cannot call hipMalloc:
// tight serving loops (the pull-next-frame loop of app/Web.hs:42-65 answers a
// client message with one pt_trace(1) + one RGB8 read), harnesses that forbid
// allocation.  (Not for HIP graph capture: the call ends with a copy to the
// caller's pageable buffer and a stream synchronisation, which capture forbids.)
//
// pt_read_rgb8 stages the quantised image in a device buffer it allocates on
// first use (pt_ctx::d_rgb).  Here that staging buffer is a file-scope device
// array, `scratch`, SCRATCH_BYTES per GPU, held until the process ends.  The
// array's name is only a host-side stand-in for the symbol: its device address
// is looked up once per device with hipGetSymbolAddress (scratch_init, called
// from pt_create with that device current) and kept in g_scratch; everything
// below uses that pointer, never the name.
#include "pt_ctx.hpp"

#include <mutex>

using namespace pt;

namespace {

constexpr size_t SCRATCH_BYTES = (size_t)8 << 20;          // 8 MiB: RGB8 of up to 2,796,202 pixels (1920x1080 = 6.2 MB fits)
static __device__ __align__(256) unsigned char scratch[SCRATCH_BYTES];

constexpr int MAX_DEVICES = 64;
std::mutex g_mu;                                           // guards g_scratch
unsigned char *g_scratch[MAX_DEVICES];                     // per device: the address of ITS copy of `scratch`
// One device's array is shared by every context on that device: a call holds the device's mutex from the kernel
// that fills the array until the copy out of it has completed, so calls from several contexts or threads on one
// GPU queue up instead of overwriting each other's image (the call is synchronous anyway).
std::mutex g_busy[MAX_DEVICES];

} // namespace

namespace pt {

// Set-up, once per device, with `device` current (pt_create calls it after hipSetDevice).
int scratch_init(pt_ctx *c, int device)
{
    if (device < 0 || device >= MAX_DEVICES) return PT_OK;             // no slot: pt_read_rgb8_noalloc reports it
    std::lock_guard<std::mutex> lock(g_mu);
    if (g_scratch[device]) return PT_OK;
    unsigned char *ptr = nullptr;
    const hipError_t e = hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr) {
        (void)hipGetLastError();
        return fail(c, PT_ERR_HIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
    }
    g_scratch[device] = ptr;
    return PT_OK;
}

} // namespace pt

extern "C" int pt_read_rgb8_noalloc(pt_ctx *c, uint8_t *out, int32_t mode)
{
    if (int s = check_ctx(c)) return s;
    if (!out) return fail(c, PT_ERR_INVALID, "pt_read_rgb8_noalloc: null output");
    const size_t bytes = npix(c) * 3;
    if (bytes > SCRATCH_BYTES)
        return fail(c, PT_ERR_LIMIT, "pt_read_rgb8_noalloc: the image (" + std::to_string(bytes) +
                                     " bytes) exceeds the 8388608-byte static scratch; use pt_read_rgb8");
    pt_ctx *on; const float *src;
    if (int s = read_source(c, &on, &src)) return s;                   // (a group: its all-reduce; no allocation either)
    const int device = on->p.device;
    unsigned char *ptr = nullptr;
    if (device >= 0 && device < MAX_DEVICES) {
        std::lock_guard<std::mutex> lock(g_mu);
        ptr = g_scratch[device];
    }
    if (!ptr) return fail(c, PT_ERR_UNSUPPORTED, "pt_read_rgb8_noalloc: no static scratch on this device (ordinal beyond 63)");
    std::lock_guard<std::mutex> busy(g_busy[device]);
    if (int s = rgb8_from(c, src, ptr, mode, mean_samples(c, 0), on->stream)) return s;
    const hipError_t ec = hipMemcpyAsync(out, ptr, bytes, hipMemcpyDeviceToHost, on->stream);
    const hipError_t es = hipStreamSynchronize(on->stream);            // also after a failed copy: the kernel must be done with the array
    PT_HIP(c, ec);
    PT_HIP(c, es);
    return PT_OK;
}
