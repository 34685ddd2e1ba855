// pt_jit.hip -- a trace kernel compiled for ONE scene (pt_scene_specialise,
// include/pt.h).  trace_static_kernel reads the object list as kernel arguments,
// so every object test is the general form `(A - o.n_raw) / (v.n_hat)`: six
// multiply-adds.  Scenes are mostly built from axis-aligned walls (the
// reference's own three planes and its light are, src/Object.hs:82-145), where
// four of those six have a zero coefficient and the other two a coefficient of
// +-1.  Deciding that per object at run time costs more than it skips (scalar
// branch chains), so the decision is taken by the compiler: the
// kernel's source (pt_static_kernel.hpp and the two headers under it, embedded
// in the library as text by the Makefile) is compiled again with hiprtc with
// the scene's geometry as constants, zero terms left out, and the resulting
// code object is launched in place of the general kernel.  The arithmetic that
// remains is the general kernel's, rounding for rounding.
//
// libhiprtc is loaded on first use (dlopen), not linked: a machine without it
// keeps the general kernel and pt_scene_specialise reports why.
#include "pt_internal.hpp"

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "pt_jit_src.inc"          // src_pt_static_kernel_hpp, src_pt_internal_hpp, src_pt_device_hpp (Makefile)

namespace pt {

namespace {

// the few hiprtc entry points used, by their documented signatures
typedef struct _hiprtcProgram *rtc_program;
struct Rtc {
    void *lib = nullptr;
    int (*create)(rtc_program *, const char *, const char *, int, const char **, const char **) = nullptr;
    int (*compile)(rtc_program, int, const char **) = nullptr;
    int (*log_size)(rtc_program, size_t *) = nullptr;
    int (*log)(rtc_program, char *) = nullptr;
    int (*code_size)(rtc_program, size_t *) = nullptr;
    int (*code)(rtc_program, char *) = nullptr;
    int (*destroy)(rtc_program *) = nullptr;
    int (*version)(int *, int *) = nullptr;
    bool ok = false;
    std::string where;             // the file the entry points came from, and its version
};

Rtc &rtc()
{
    static Rtc r;
    static std::once_flag once;
    std::call_once(once, [] {
        // PT_HIPRTC names the library file (experiments: one compiler against another in the same process)
        if (const char *forced = getenv("PT_HIPRTC")) r.lib = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
        for (const char *name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) {
            if (r.lib) break;
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        }
        if (!r.lib) return;
        auto sym = [&](const char *n) { return dlsym(r.lib, n); };
        r.create    = (decltype(r.create))sym("hiprtcCreateProgram");
        r.compile   = (decltype(r.compile))sym("hiprtcCompileProgram");
        r.log_size  = (decltype(r.log_size))sym("hiprtcGetProgramLogSize");
        r.log       = (decltype(r.log))sym("hiprtcGetProgramLog");
        r.code_size = (decltype(r.code_size))sym("hiprtcGetCodeSize");
        r.code      = (decltype(r.code))sym("hiprtcGetCode");
        r.destroy   = (decltype(r.destroy))sym("hiprtcDestroyProgram");
        r.version   = (decltype(r.version))sym("hiprtcVersion");
        r.ok = r.create && r.compile && r.log_size && r.log && r.code_size && r.code && r.destroy;
        if (r.ok) {
            int major = 0, minor = 0;
            if (r.version) (void)r.version(&major, &minor);
            Dl_info di{};
            const bool known = dladdr((void *)r.compile, &di) != 0 && di.dli_fname;
            r.where = "hiprtc " + std::to_string(major) + "." + std::to_string(minor) + " (" + (known ? di.dli_fname : "?") + ")";
        }
    });
    return r;
}

struct Entry {
    hipModule_t mod = nullptr; hipFunction_t fn = nullptr; std::string info;
    int device = 0;
    int users = 0;                 // contexts that hold `fn` (jit_static_kernel hands it out, jit_release takes it back)
    uint64_t stamp = 0;            // last hand-out, for the eviction order
};
std::mutex g_mu;
std::map<std::string, Entry> g_cache;      // (device, NS, SPEC, geometry bits) -> loaded kernel
uint64_t g_clock = 0;
// Kernels nobody holds stay loaded (the next context with that scene skips the 0.7-s compile) up to this many;
// beyond it the least recently handed-out unused ones are unloaded (hipModuleUnload).
constexpr size_t JIT_CACHE_KEEP = 16;

void evict_unused_locked()
{
    while (g_cache.size() > JIT_CACHE_KEEP) {
        auto victim = g_cache.end();
        for (auto it = g_cache.begin(); it != g_cache.end(); ++it)
            if (it->second.users == 0 && (victim == g_cache.end() || it->second.stamp < victim->second.stamp)) victim = it;
        if (victim == g_cache.end()) return;               // every kernel is in use
        int cur = 0;
        (void)hipGetDevice(&cur);
        (void)hipSetDevice(victim->second.device);
        (void)hipModuleUnload(victim->second.mod);
        (void)hipSetDevice(cur);
        (void)hipGetLastError();
        g_cache.erase(victim);
    }
}

} // namespace

hipFunction_t jit_static_kernel(int device, int n_obj, bool spec, const float4 *geom_host, std::string &err)
{
    if (n_obj < 0 || n_obj > MAX_STATIC_OBJECTS) { err = "more objects than the by-value kernel takes"; return nullptr; }
    const int NS = n_obj <= 4 ? 4 : MAX_STATIC_OBJECTS;
    // the geometry exactly as launch_trace_ns pads it
    std::vector<float> g((size_t)NS * 12, 0.0f);
    for (int k = 0; k < NS; ++k) {
        if (k < n_obj) std::memcpy(&g[(size_t)k * 12], &geom_host[3 * k], 12 * sizeof(float));
        else g[(size_t)k * 12 + 7] = __builtin_inff();
    }
    std::string key((const char *)g.data(), g.size() * sizeof(float));
    key += (char)NS; key += (char)(spec ? 1 : 0); key += (char)device;
    if (const char *e = getenv("PT_JIT_FLAGS")) key += e;

    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_cache.find(key);
    if (it != g_cache.end()) { it->second.users += 1; it->second.stamp = ++g_clock; return it->second.fn; }

    Rtc &r = rtc();
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { (void)hipGetLastError(); err = "hipGetDeviceProperties failed"; return nullptr; }
    const std::string arch = std::string("--offload-arch=") + prop.gcnArchName;
    std::vector<char> code;
    // PT_SCENE_KERNEL_CO: a code object built outside (experiments on the kernel's assembly, tools/rebank.py)
    if (const char *co = getenv("PT_SCENE_KERNEL_CO")) {
        if (FILE *f = std::fopen(co, "rb")) {
            char buf[65536];
            size_t got;
            while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) code.insert(code.end(), buf, buf + got);
            std::fclose(f);
        }
        if (code.empty()) { err = std::string("PT_SCENE_KERNEL_CO: cannot read ") + co; return nullptr; }
    }
    if (code.empty()) {
    if (!r.ok) { err = "libhiprtc.so not found or incomplete"; return nullptr; }

    std::string src = "#define PT_JIT 1\n#define PT_JIT_NS " + std::to_string(NS) + "\n#define PT_JIT_SPEC " +
                      (spec ? "1" : "0") + "\n#define PT_JIT_F(x) __builtin_bit_cast(float, x)\n#define PT_JIT_GEOM {";
    for (int k = 0; k < NS; ++k) {
        src += k ? ",{" : "{";
        for (int i = 0; i < 12; ++i) {
            uint32_t bits;
            std::memcpy(&bits, &g[(size_t)k * 12 + i], 4);
            char buf[32];
            std::snprintf(buf, sizeof buf, "%sPT_JIT_F(0x%08xu)", i ? "," : "", bits);
            src += buf;
        }
        src += "}";
    }
    src += "}\n#include \"pt_static_kernel.hpp\"\n"
           // which form lane_of()/pick() took in this compilation (pt_static_kernel.hpp), for pt_jit_info
           "extern \"C\" __device__ __attribute__((used)) int pt_jit_has_inverse_ballot = "
           "__has_builtin(__builtin_amdgcn_inverse_ballot_w64) ? 1 : 0;\n";

    const char *headers[] = {src_pt_static_kernel_hpp, src_pt_internal_hpp, src_pt_device_hpp};
    const char *names[] = {"pt_static_kernel.hpp", "pt_internal.hpp", "pt_device.hpp"};
    rtc_program prog = nullptr;
    if (r.create(&prog, src.c_str(), "pt_scene_kernel.hip", 3, headers, names) != 0) { err = "hiprtcCreateProgram failed"; return nullptr; }
    std::vector<const char *> opts = {arch.c_str(), "-O3", "-std=c++17", "-fno-slp-vectorize"};   // the Makefile's code-generation flags
    // PT_JIT_FLAGS: further compiler options, separated by blanks (scheduling experiments; part of the cache key)
    std::vector<std::string> extra;
    if (const char *e = getenv("PT_JIT_FLAGS")) {
        std::string cur;
        for (const char *q = e;; ++q) {
            if (*q == ' ' || *q == '\0') { if (!cur.empty()) extra.push_back(cur); cur.clear(); if (!*q) break; }
            else cur += *q;
        }
    }
    for (const std::string &x : extra) opts.push_back(x.c_str());
    const int rc = r.compile(prog, (int)opts.size(), opts.data());
    if (rc != 0) {
        size_t n = 0;
        r.log_size(prog, &n);
        std::string log(n, '\0');
        if (n) r.log(prog, &log[0]);
        r.destroy(&prog);
        err = "hiprtcCompileProgram failed: " + log.substr(0, 2000);
        return nullptr;
    }
    size_t n = 0;
    r.code_size(prog, &n);
    code.resize(n);
    r.code(prog, code.data());
    r.destroy(&prog);
    }

    Entry e;
    hipError_t he = hipModuleLoadData(&e.mod, code.data());
    if (he != hipSuccess) { (void)hipGetLastError(); err = std::string("hipModuleLoadData: ") + hipGetErrorString(he); return nullptr; }
    const std::string sym = "_ZN2pt19trace_static_kernelILi" + std::to_string(NS) + "ELb" + (spec ? "1" : "0") +
                            "EEEvNS_9TraceArgsENS_10StaticGeomIXT_EEE";
    he = hipModuleGetFunction(&e.fn, e.mod, sym.c_str());
    if (he != hipSuccess) {
        (void)hipGetLastError(); (void)hipModuleUnload(e.mod);
        err = std::string("hipModuleGetFunction(") + sym + "): " + hipGetErrorString(he);
        return nullptr;
    }
    int has = -1;
    {
        hipDeviceptr_t dp = nullptr; size_t nb = 0;
        if (hipModuleGetGlobal(&dp, &nb, e.mod, "pt_jit_has_inverse_ballot") == hipSuccess && nb == sizeof(int)) {
            if (hipMemcpy(&has, dp, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) has = -1;
        }
        (void)hipGetLastError();
    }
    e.info = r.where + ", " + arch + ", __builtin_amdgcn_inverse_ballot_w64 " +
             (has == 1 ? "available" : has == 0 ? "missing (v_cndmask_b32 form)" : "unknown");
    e.device = device; e.users = 1; e.stamp = ++g_clock;
    g_cache.emplace(std::move(key), e);
    evict_unused_locked();
    return e.fn;
}

void jit_release(hipFunction_t fn)
{
    if (!fn) return;
    std::lock_guard<std::mutex> lock(g_mu);
    for (auto &kv : g_cache)
        if (kv.second.fn == fn) { if (kv.second.users > 0) kv.second.users -= 1; break; }
    evict_unused_locked();
}

int jit_cache_size()
{
    std::lock_guard<std::mutex> lock(g_mu);
    return (int)g_cache.size();
}

std::string jit_kernel_info(hipFunction_t fn)
{
    std::lock_guard<std::mutex> lock(g_mu);
    for (const auto &kv : g_cache)
        if (kv.second.fn == fn) return kv.second.info;
    return "";
}

} // namespace pt
