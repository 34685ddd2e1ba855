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
#include <sys/stat.h>
#include <unistd.h>

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
#ifdef PT_EXPERIMENTS
        // PT_HIPRTC names the library file (experiments: one compiler against another in the same process).  Only in
        // builds made with `make EXP=-DPT_EXPERIMENTS`: the shipped library opens no file the environment names.
        if (const char *forced = getenv("PT_HIPRTC")) r.lib = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
#endif
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

// ---- the on-disk cache of compiled scene kernels ---------------------------------------------------------------
// A scene kernel takes hiprtc 0.4-0.9 s; a host that starts often with the same scene (app/Write.hs renders one image
// per process, src/Object.hs:147-148 is a constant of the program) pays it every time.  Compiled code objects are
// therefore kept as files: <dir>/<128-bit hash of everything the code depends on>.co, each starting with that
// "everything" spelled out -- the kernel source text as embedded in this library, the scene's geometry bits, NS, SPEC,
// the --offload-arch, the compiler options and the hiprtc that compiled it -- which is compared byte for byte
// before the code is used, so a hash collision or a file of another build is a miss, not a wrong kernel.
// <dir> = $PT_KERNEL_CACHE (empty, "0" or "off": no cache), else $XDG_CACHE_HOME/pathtracer_amd, else
// $HOME/.cache/pathtracer_amd.  Best effort throughout: any failure to read or write is a miss / a skipped write.
// The files are code this library will run: the directory must be writable by the user only (it is created 0700).
std::string disk_cache_dir()
{
    if (const char *e = getenv("PT_KERNEL_CACHE")) {
        const std::string v = e;
        return (v.empty() || v == "0" || v == "off") ? std::string() : v;
    }
    if (const char *x = getenv("XDG_CACHE_HOME")) if (*x) return std::string(x) + "/pathtracer_amd";
    if (const char *h = getenv("HOME")) if (*h) return std::string(h) + "/.cache/pathtracer_amd";
    return std::string();
}

std::string disk_cache_name(const std::string &material)
{
    uint64_t a = 0xcbf29ce484222325ull, b = 0x84222325cbf29ce4ull;      // two FNV-1a streams, the second over the reversed text
    for (size_t i = 0; i < material.size(); ++i) {
        a = (a ^ (unsigned char)material[i]) * 0x100000001b3ull;
        b = (b ^ (unsigned char)material[material.size() - 1 - i]) * 0x100000001b3ull;
    }
    char buf[40];
    std::snprintf(buf, sizeof buf, "%016llx%016llx.co", (unsigned long long)a, (unsigned long long)b);
    return buf;
}

constexpr char DISK_MAGIC[8] = {'P', 'T', 'C', 'O', '0', '0', '1', '\n'};

bool disk_cache_load(const std::string &dir, const std::string &material, std::vector<char> &code)
{
    if (dir.empty()) return false;
    FILE *f = std::fopen((dir + "/" + disk_cache_name(material)).c_str(), "rb");
    if (!f) return false;
    bool ok = false;
    char magic[8];
    uint64_t nk = 0, nc = 0;
    if (std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, DISK_MAGIC, 8) == 0 && std::fread(&nk, 8, 1, f) == 1 &&
        nk == material.size()) {
        std::string have(nk, '\0');
        if (std::fread(&have[0], 1, nk, f) == nk && have == material && std::fread(&nc, 8, 1, f) == 1 && nc > 0 &&
            nc < ((uint64_t)1 << 28)) {
            code.resize(nc);
            ok = std::fread(code.data(), 1, nc, f) == nc && std::fgetc(f) == EOF;
        }
    }
    std::fclose(f);
    if (!ok) code.clear();
    return ok;
}

void disk_cache_store(const std::string &dir, const std::string &material, const std::vector<char> &code)
{
    if (dir.empty() || code.empty()) return;
    const size_t slash = dir.find_last_of('/');
    if (slash != std::string::npos && slash > 0) (void)mkdir(dir.substr(0, slash).c_str(), 0700);   // ~/.cache
    (void)mkdir(dir.c_str(), 0700);
    const std::string final_name = dir + "/" + disk_cache_name(material);
    const std::string tmp = final_name + ".tmp" + std::to_string((long)getpid());
    FILE *f = std::fopen(tmp.c_str(), "wb");
    if (!f) return;
    const uint64_t nk = material.size(), nc = code.size();
    const bool ok = std::fwrite(DISK_MAGIC, 1, 8, f) == 8 && std::fwrite(&nk, 8, 1, f) == 1 &&
                    std::fwrite(material.data(), 1, nk, f) == nk && std::fwrite(&nc, 8, 1, f) == 1 &&
                    std::fwrite(code.data(), 1, nc, f) == nc;
    const bool closed = std::fclose(f) == 0;
    if (!ok || !closed || std::rename(tmp.c_str(), final_name.c_str()) != 0) (void)std::remove(tmp.c_str());
}

// ---- scene kernels shipped with the library -----------------------------------------------------------------------
// The reference's scene is a constant of the program (src/Object.hs:147-148), so its kernel can be compiled when the
// library is built, by hipcc, and a host on a machine without libhiprtc -- or one that must not spend 0.7 s in a
// compiler at start-up -- still runs it.  `make scene_kernels` (tools/build_scene_kernels.py) asks the library for the
// kernel source of a scene (pt_scene_kernel_ship below), compiles it offline with the Makefile's code-generation flags
// and hands the code object back to be stored as <directory of libpt.so>/scene_kernels/<hash>.co in the format of the
// on-disk cache: the file starts with everything the code depends on (source text, geometry bits, NS, SPEC, the
// --offload-arch, the embedded headers) and is used only where that matches byte for byte -- a library rebuilt from
// changed sources simply does not find its predecessor's files.  What is NOT part of the key is the compiler: the
// file is this build's own artefact, like the kernels inside libpt.so.  PT_SHIPPED_KERNELS=off ignores the directory.
std::string shipped_dir()
{
    if (const char *e = getenv("PT_SHIPPED_KERNELS")) {
        const std::string v = e;
        if (v.empty() || v == "0" || v == "off") return std::string();
    }
    Dl_info di{};
    if (dladdr((void *)&shipped_dir, &di) == 0 || !di.dli_fname) return std::string();
    const std::string lib = di.dli_fname;
    const size_t slash = lib.find_last_of('/');
    return (slash == std::string::npos ? std::string(".") : lib.substr(0, slash)) + "/scene_kernels";
}

// the geometry exactly as launch_trace_ns pads it, and the kernel's source text for it
int scene_kernel_ns(int n_obj) { return n_obj <= 4 ? 4 : n_obj <= MAX_STATIC_OBJECTS ? MAX_STATIC_OBJECTS : MAX_JIT_OBJECTS; }

std::vector<float> padded_geometry(int NS, int n_obj, const float4 *geom_host)
{
    std::vector<float> g((size_t)NS * 12, 0.0f);
    for (int k = 0; k < NS; ++k) {
        if (k < n_obj) std::memcpy(&g[(size_t)k * 12], &geom_host[3 * k], 12 * sizeof(float));
        else g[(size_t)k * 12 + 7] = __builtin_inff();
    }
    return g;
}

std::string scene_kernel_source(int NS, bool spec, const std::vector<float> &g)
{
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
    return src;
}

std::string shipped_material(const std::string &src, const std::string &arch)
{
    return src + "\n//" + arch + "\n//shipped with the library\n" + src_pt_static_kernel_hpp + src_pt_internal_hpp + src_pt_device_hpp;
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
    if (n_obj < 0 || n_obj > MAX_JIT_OBJECTS) { err = "more objects than a scene kernel takes"; return nullptr; }
    const int NS = scene_kernel_ns(n_obj);
    const std::vector<float> g = padded_geometry(NS, n_obj, geom_host);
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
#ifdef PT_EXPERIMENTS
    // PT_SCENE_KERNEL_CO: a code object built outside (experiments on the kernel's assembly, tools/rebank.py).  It has
    // ONE scene baked in and is loaded for whatever scene is asked for, so it exists only in builds made with
    // `make EXP=-DPT_EXPERIMENTS`; the shipped library never loads code it did not compile from its own source.
    if (const char *co = getenv("PT_SCENE_KERNEL_CO")) {
        if (FILE *f = std::fopen(co, "rb")) {
            char buf[65536];
            size_t got;
            while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) code.insert(code.end(), buf, buf + got);
            std::fclose(f);
        }
        if (code.empty()) { err = std::string("PT_SCENE_KERNEL_CO: cannot read ") + co; return nullptr; }
    }
#endif
    bool from_disk = false, shipped = false;
    std::string material;                                  // everything the code object depends on (disk cache)
    const std::string cache_dir = disk_cache_dir();
    const std::string src = scene_kernel_source(NS, spec, g);
    const std::string ship_dir = shipped_dir();
    if (code.empty() && !getenv("PT_JIT_FLAGS")) {         // a kernel built with this library (make scene_kernels)?
        material = shipped_material(src, arch);
        shipped = disk_cache_load(ship_dir, material, code);
    }
    if (code.empty()) {
    // (PT_NO_HIPRTC in the environment: a host that must not load a compiler, and the tests of the shipped kernels)
    if (!r.ok || getenv("PT_NO_HIPRTC")) { err = "libhiprtc.so not found, incomplete or not wanted (PT_NO_HIPRTC), and no kernel for this scene was shipped with the library"; return nullptr; }

    const char *headers[] = {src_pt_static_kernel_hpp, src_pt_internal_hpp, src_pt_device_hpp};
    const char *names[] = {"pt_static_kernel.hpp", "pt_internal.hpp", "pt_device.hpp"};
    material = src + "\n//" + arch + "\n//" + r.where + "\n//" + (getenv("PT_JIT_FLAGS") ? getenv("PT_JIT_FLAGS") : "") + "\n" +
               src_pt_static_kernel_hpp + src_pt_internal_hpp + src_pt_device_hpp;
    from_disk = disk_cache_load(cache_dir, material, code);
    if (!from_disk) {
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
    disk_cache_store(cache_dir, material, code);
    }
    }

    Entry e;
    hipError_t he = hipModuleLoadData(&e.mod, code.data());
    if (he != hipSuccess) {
        (void)hipGetLastError();
        err = std::string("hipModuleLoadData: ") + hipGetErrorString(he) + (shipped ? " (a kernel shipped in " + ship_dir + ")" : std::string());
        if (from_disk) (void)std::remove((cache_dir + "/" + disk_cache_name(material)).c_str());   // a file that does not load is not kept
        return nullptr;
    }
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
    e.info = (shipped ? "hipcc at build time (make scene_kernels)" : r.where) + ", " + arch + ", __builtin_amdgcn_inverse_ballot_w64 " +
             (has == 1 ? "available" : has == 0 ? "missing (v_cndmask_b32 form)" : "unknown") +
             (shipped ? ", code object shipped with the library in " + ship_dir
                      : from_disk ? ", code object from the on-disk cache " + cache_dir : ", compiled in this process");
    e.device = device; e.users = 1; e.stamp = ++g_clock;
    g_cache.emplace(std::move(key), e);
    evict_unused_locked();
    return e.fn;
}

// Build support (pt_scene_kernel_ship, include/pt.h): the kernel source of a scene to a file, and a code object
// compiled from it into the shipped directory.  No GPU and no HIP call.
int jit_ship(int n_obj, bool spec, const float4 *geom_host, const char *arch, const char *src_out, const char *code_in,
             std::string &err)
{
    if (n_obj < 0 || n_obj > MAX_JIT_OBJECTS) { err = "more objects than a scene kernel takes"; return -1; }
    const int NS = scene_kernel_ns(n_obj);
    const std::string src = scene_kernel_source(NS, spec, padded_geometry(NS, n_obj, geom_host));
    if (src_out) {
        FILE *f = std::fopen(src_out, "wb");
        if (!f || std::fwrite(src.data(), 1, src.size(), f) != src.size() || std::fclose(f) != 0) { err = std::string("cannot write ") + src_out; return -1; }
    }
    if (!src_out && !code_in) {                            // query: is a kernel for this scene and target shipped?
        std::vector<char> code;
        return disk_cache_load(shipped_dir(), shipped_material(src, std::string("--offload-arch=") + arch), code) ? 0 : 1;
    }
    if (code_in) {
        std::vector<char> code;
        if (FILE *f = std::fopen(code_in, "rb")) {
            char buf[65536];
            size_t got;
            while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) code.insert(code.end(), buf, buf + got);
            std::fclose(f);
        }
        if (code.empty()) { err = std::string("cannot read ") + code_in; return -1; }
        const std::string dir = shipped_dir();
        if (dir.empty()) { err = "no directory for shipped kernels (PT_SHIPPED_KERNELS=off?)"; return -1; }
        const std::string material = shipped_material(src, std::string("--offload-arch=") + arch);
        disk_cache_store(dir, material, code);
        std::vector<char> back;
        if (!disk_cache_load(dir, material, back) || back != code) { err = "could not store the kernel in " + dir; return -1; }
    }
    return 0;
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
