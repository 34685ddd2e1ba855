// pt_multi.hip -- multi-GPU behind the C ABI (include/pt.h).
//
// The reference has ONE `traces` (src/Lib.hs:53-56); however many GPUs render
// it, a host sees one.  The work shards by SAMPLE: Philox is keyed by (pixel,
// global sample, bounce), so any partition of a frame's sample indices over
// GPUs renders the same set of paths, and one RCCL all-reduce (sum) of the fp32
// canvases over xGMI combines them.  Two layouts, both through librccl:
//
//   * one process, several devices  -- pt_create_multi: the returned context is
//     the HEAD of a group of ordinary per-device contexts.  pt_trace(k) hands
//     each member a contiguous block of the call's k sample indices; the first
//     read after it (pt_read_sum, pt_read_rgb8, ...) runs ncclAllReduce over
//     the members' canvases (ncclCommInitAll communicators, one
//     ncclGroupStart/End) into a per-device total and reads device 0's.
//   * one process per GPU          -- pt_comm_unique_id / pt_comm_init_rank /
//     pt_allreduce_sum on an ordinary context created with
//     sample_offset = rank, sample_stride = ranks.
//
// A member's own canvas always stays its PARTIAL sum (the all-reduce is out of
// place, into d_total), so a render can go on after a read -- the progressive
// frames of app/Web.hs:42-65 -- without counting anything twice.
//
// librccl is opened on first use (dlopen "librccl.so.1": in a process that has
// PyTorch loaded that is PyTorch's copy, elsewhere /opt/rocm's), not linked, so
// libpt.so loads on a machine without it; the multi-GPU entry points then fail
// with PT_ERR_UNSUPPORTED and say why.
#include "pt_ctx.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>             // types and prototypes only: every call goes through the table below

#include <cstring>
#include <mutex>
#include <new>

using namespace pt;

namespace {

struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetVersion)     get_version = nullptr;
    decltype(&ncclGetUniqueId)    get_unique_id = nullptr;
    decltype(&ncclCommInitRank)   comm_init_rank = nullptr;
    decltype(&ncclCommInitAll)    comm_init_all = nullptr;
    decltype(&ncclCommDestroy)    comm_destroy = nullptr;
    decltype(&ncclAllReduce)      all_reduce = nullptr;
    decltype(&ncclGroupStart)     group_start = nullptr;
    decltype(&ncclGroupEnd)       group_end = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    bool ok = false;
    int version = 0;
    std::string why;
};

Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (!r.lib) { r.why = "librccl.so.1 not found (dlopen)"; return; }
        auto sym = [&](const char *n) { return dlsym(r.lib, n); };
        r.get_version    = (decltype(r.get_version))sym("ncclGetVersion");
        r.get_unique_id  = (decltype(r.get_unique_id))sym("ncclGetUniqueId");
        r.comm_init_rank = (decltype(r.comm_init_rank))sym("ncclCommInitRank");
        r.comm_init_all  = (decltype(r.comm_init_all))sym("ncclCommInitAll");
        r.comm_destroy   = (decltype(r.comm_destroy))sym("ncclCommDestroy");
        r.all_reduce     = (decltype(r.all_reduce))sym("ncclAllReduce");
        r.group_start    = (decltype(r.group_start))sym("ncclGroupStart");
        r.group_end      = (decltype(r.group_end))sym("ncclGroupEnd");
        r.error_string   = (decltype(r.error_string))sym("ncclGetErrorString");
        r.ok = r.get_version && r.get_unique_id && r.comm_init_rank && r.comm_init_all && r.comm_destroy &&
               r.all_reduce && r.group_start && r.group_end && r.error_string;
        if (!r.ok) { r.why = "librccl.so.1 lacks an entry point libpt needs"; return; }
        (void)r.get_version(&r.version);
    });
    return r;
}

#define PT_NCCL(c, call)                                                                                     \
    do {                                                                                                     \
        ncclResult_t r_ = (call);                                                                            \
        if (r_ != ncclSuccess)                                                                               \
            return fail((c), PT_ERR_HIP, std::string(#call) + ": RCCL: " + rccl().error_string(r_));         \
    } while (0)

// total[i] = sum over the listed canvases, in list order (members of a group that share one device)
constexpr int MAX_PER_DEVICE = 8;
struct PlaneList { const float *p[MAX_PER_DEVICE]; int m; };

__global__ __launch_bounds__(256) void planes_sum_kernel(float *__restrict__ total, PlaneList pl, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float s = pl.p[0][i];
        for (int q = 1; q < pl.m; ++q) s += pl.p[q][i];
        total[i] = s;
    }
}

} // namespace

constexpr int MAX_GROUP = 64;

struct pt_group {
    std::vector<pt_ctx *> members;              // in the order of pt_create_multi's device list
    std::vector<int> devices;                   // the distinct devices, in order of first appearance
    std::vector<std::vector<int>> on_dev;       // per distinct device: its members (the first is the device's leader)
    std::vector<ncclComm_t> comms;              // one per distinct device (ncclCommInitAll); empty: one device, no RCCL here
    int64_t samples_done = 0;
    bool total_valid = false;
    bool timing = false;
    bool failed = false;                        // a member's trace failed after others had added their blocks: the
                                                // members' sample counts disagree until pt_reset
};

namespace pt {

pt_ctx *group_first(pt_ctx *head) { return head->group->members[0]; }

static int up(pt_ctx *head, pt_ctx *m, int status)           // a member's failure becomes the head's
{
    if (status != PT_OK) head->err = m->err;
    return status;
}

int group_scene_upload(pt_ctx *head, const pt_object *objs, int32_t n)
{
    head->group->total_valid = false;
    for (pt_ctx *m : head->group->members)
        if (int s = up(head, m, pt_scene_upload(m, objs, n))) return s;
    return PT_OK;
}

int group_scene_specialise(pt_ctx *head)
{
    for (pt_ctx *m : head->group->members)
        if (int s = up(head, m, pt_scene_specialise(m))) return s;
    return PT_OK;
}

// The call's sample indices k in [done, done + iters) in contiguous blocks, one per member (sizes differ by at
// most one): the union is the range whatever the number of members, so G GPUs render what one would.
int group_trace(pt_ctx *head, int32_t iters)
{
    pt_group *g = head->group;
    if (iters < 0) return fail(head, PT_ERR_INVALID, "pt_trace: iters < 0");
    if (g->failed)
        return fail(head, PT_ERR_INVALID, "pt_trace: an earlier pt_trace of this device group failed part-way (some members "
                                          "had added their blocks); pt_reset before tracing again");
    if ((int64_t)head->p.sample_offset + (g->samples_done + iters) * (int64_t)head->p.sample_stride >= MAX_GLOBAL_SAMPLES)
        return fail(head, PT_ERR_LIMIT, "pt_trace: global sample index would exceed 2^26");
    const int n = (int)g->members.size();
    const int32_t base = iters / n, rem = iters % n;
    int64_t k = g->samples_done;
    g->total_valid = false;
    for (int i = 0; i < n; ++i) {
        const int32_t cnt = base + (i < rem ? 1 : 0);
        pt_ctx *m = g->members[i];
        if (cnt == 0) {                              // (a member that sits a call out must still hold a scene)
            if (!m->have_scene) return fail(head, PT_ERR_NO_SCENE, "pt_trace: no scene uploaded");
            continue;
        }
        if (int s = up(head, m, trace_block(m, k, cnt))) {
            // members 0..i-1 hold this call's blocks already: a retry would count those samples twice
            if (k != g->samples_done) g->failed = true;
            return s;
        }
        k += cnt;
    }
    g->samples_done += iters;
    return PT_OK;
}

int group_reset(pt_ctx *head)
{
    for (pt_ctx *m : head->group->members)
        if (int s = up(head, m, pt_reset(m))) return s;
    head->group->samples_done = 0;
    head->group->total_valid = false;
    head->group->failed = false;
    return PT_OK;
}

// The sum of the members' canvases in every device leader's d_total: members that share a device are added by
// a kernel there, the devices by ONE ncclAllReduce each inside one ncclGroupStart/End.
int group_total(pt_ctx *head, pt_ctx **leader)
{
    pt_group *g = head->group;
    *leader = g->members[g->on_dev[0][0]];
    if (g->failed)
        return fail(head, PT_ERR_INVALID, "read: a pt_trace of this device group failed part-way; pt_reset first");
    if (g->total_valid) return PT_OK;
    const size_t count = npix(head) * 3;
    std::vector<const float *> send(g->devices.size());
    for (size_t di = 0; di < g->devices.size(); ++di) {
        pt_ctx *L = g->members[g->on_dev[di][0]];
        PT_HIP(head, hipSetDevice(g->devices[di]));
        PlaneList pl{};
        for (int mi : g->on_dev[di]) {
            pt_ctx *m = g->members[mi];
            if (int s = up(head, m, materialise_canvas(m))) return s;       // (a member without samples since pt_reset)
            pl.p[pl.m++] = m->d_canvas;
            if (m != L) {
                PT_HIP(head, hipEventRecord(m->ev_sync, m->stream));
                PT_HIP(head, hipStreamWaitEvent(L->stream, m->ev_sync, 0));
            }
        }
        send[di] = L->d_canvas;
        if (pl.m > 1) {
            hipLaunchKernelGGL(planes_sum_kernel, dim3(2048), dim3(256), 0, L->stream, L->d_total, pl, count);
            PT_HIP(head, hipGetLastError());
            PT_HIP(head, hipEventRecord(L->ev_sync, L->stream));
            for (int mi : g->on_dev[di])                                     // their next trace must not overtake this read
                if (g->members[mi] != L) PT_HIP(head, hipStreamWaitEvent(g->members[mi]->stream, L->ev_sync, 0));
            send[di] = L->d_total;
        }
    }
    if (!g->comms.empty()) {
        Rccl &r = rccl();
        PT_NCCL(head, r.group_start());
        for (size_t di = 0; di < g->devices.size(); ++di) {
            pt_ctx *L = g->members[g->on_dev[di][0]];
            const ncclResult_t rc = r.all_reduce(send[di], L->d_total, count, ncclFloat, ncclSum, g->comms[di], L->stream);
            if (rc != ncclSuccess) {
                (void)r.group_end();
                return fail(head, PT_ERR_HIP, std::string("ncclAllReduce: RCCL: ") + r.error_string(rc));
            }
        }
        PT_NCCL(head, r.group_end());
    } else if (send[0] != g->members[g->on_dev[0][0]]->d_total) {            // one device, no RCCL on this machine
        pt_ctx *L = g->members[g->on_dev[0][0]];
        PT_HIP(head, hipMemcpyAsync(L->d_total, send[0], count * sizeof(float), hipMemcpyDeviceToDevice, L->stream));
    }
    g->total_valid = true;
    return PT_OK;
}

int group_segments(pt_ctx *head, uint64_t *out)
{
    uint64_t total = 0;
    for (pt_ctx *m : head->group->members) {
        uint64_t v = 0;
        if (int s = up(head, m, pt_segments(m, &v))) return s;
        total += v;
    }
    *out = total;
    return PT_OK;
}

int group_samples_done(pt_ctx *head, int64_t *out)
{
    *out = head->group->samples_done;
    return PT_OK;
}

int group_synchronize(pt_ctx *head)
{
    for (pt_ctx *m : head->group->members)
        if (int s = up(head, m, pt_synchronize(m))) return s;
    return PT_OK;
}

// checkpoint restore: the sums go to member 0, the other members start from zero
int group_write_sum_f32(pt_ctx *head, const float *in, int64_t samples_done)
{
    pt_group *g = head->group;
    for (size_t i = 0; i < g->members.size(); ++i) {
        pt_ctx *m = g->members[i];
        if (i == 0) { if (int s = up(head, m, pt_write_sum_f32(m, in, samples_done))) return s; }
        else { m->canvas_stale = true; m->total_valid = false; m->samples_done = 0; }
    }
    g->samples_done = samples_done;
    g->total_valid = false;
    g->failed = false;
    return PT_OK;
}

int group_set_timing(pt_ctx *head, int32_t enable)
{
    for (pt_ctx *m : head->group->members) (void)pt_set_timing(m, enable);
    return PT_OK;
}

// the slowest member's kernel time (the members run concurrently, one per device)
int group_last_trace_ms(pt_ctx *head, float *ms, int32_t *launches)
{
    float worst = -1.0f;
    int32_t nl = 0;
    for (pt_ctx *m : head->group->members) {
        if (!m->have_timing) continue;
        float t = 0.0f; int32_t l = 0;
        if (int s = up(head, m, pt_last_trace_ms(m, &t, &l))) return s;
        if (t > worst) { worst = t; nl = l; }
    }
    if (worst < 0.0f) return fail(head, PT_ERR_INVALID, "pt_last_trace_ms: no timed pt_trace yet");
    *ms = worst;
    if (launches) *launches = nl;
    return PT_OK;
}

void group_destroy(pt_ctx *head)
{
    pt_group *g = head->group;
    if (!g) return;
    for (pt_ctx *m : g->members) {
        (void)hipSetDevice(m->p.device);
        if (m->stream) (void)hipStreamSynchronize(m->stream);
    }
    for (ncclComm_t cm : g->comms) if (cm) (void)rccl().comm_destroy(cm);
    for (pt_ctx *m : g->members) pt_destroy(m);
    delete g;
    head->group = nullptr;
}

void comm_destroy(pt_ctx *c)
{
    c->comm = nullptr;
    c->comm_ref.reset();               // ncclCommDestroy when this was the last context using it
    if (c->d_total) { (void)hipFree(c->d_total); c->d_total = nullptr; }
    c->total_valid = false;
}

} // namespace pt

extern "C" {

int pt_create_multi(pt_ctx **out, const pt_params *params, const int32_t *devices, int32_t n_devices)
{
    if (!out || !params || !devices) return fail(nullptr, PT_ERR_INVALID, "pt_create_multi: null argument");
    *out = nullptr;
    if (n_devices < 1 || n_devices > MAX_GROUP)
        return fail(nullptr, PT_ERR_LIMIT, "pt_create_multi: need 1 <= n_devices <= 64");
    pt_ctx *head = new (std::nothrow) pt_ctx;
    pt_group *g = new (std::nothrow) pt_group;
    if (!head || !g) { delete head; delete g; return fail(nullptr, PT_ERR_INVALID, "pt_create_multi: out of host memory"); }
    head->p = *params;
    head->p.device = devices[0];
    head->group = g;
    auto bail = [&](int status, const std::string &msg) {
        pt_destroy(head);
        return fail(nullptr, status, msg);
    };
    for (int i = 0; i < n_devices; ++i) {
        pt_params pp = *params;
        pp.device = devices[i];
        pt_ctx *m = nullptr;
        const int s = pt_create(&m, &pp);
        if (s != PT_OK) return bail(s, std::string("pt_create_multi: device ") + std::to_string(devices[i]) + ": " + pt_last_error(nullptr));
        g->members.push_back(m);
        size_t di = 0;
        while (di < g->devices.size() && g->devices[di] != devices[i]) ++di;
        if (di == g->devices.size()) { g->devices.push_back(devices[i]); g->on_dev.emplace_back(); }
        if ((int)g->on_dev[di].size() == MAX_PER_DEVICE)
            return bail(PT_ERR_LIMIT, "pt_create_multi: a device may be listed at most 8 times");
        g->on_dev[di].push_back(i);
    }
    const size_t cbytes = npix(head) * 3 * sizeof(float);
    for (size_t di = 0; di < g->devices.size(); ++di) {
        pt_ctx *L = g->members[g->on_dev[di][0]];
        hipError_t e = hipSetDevice(g->devices[di]);
        if (e == hipSuccess) e = hipMalloc((void **)&L->d_total, cbytes);
        if (e != hipSuccess) return bail(PT_ERR_HIP, std::string("pt_create_multi: hipMalloc(total): ") + hipGetErrorString(e));
    }
    Rccl &r = rccl();
    if (r.ok) {
        g->comms.assign(g->devices.size(), nullptr);
        const ncclResult_t rc = r.comm_init_all(g->comms.data(), (int)g->devices.size(), g->devices.data());
        if (rc != ncclSuccess) {
            g->comms.clear();
            return bail(PT_ERR_HIP, std::string("pt_create_multi: ncclCommInitAll: RCCL: ") + r.error_string(rc));
        }
    } else if (g->devices.size() > 1) {
        return bail(PT_ERR_UNSUPPORTED, "pt_create_multi: " + r.why + "; more than one device needs RCCL");
    }
    *out = head;
    return PT_OK;
}

int pt_comm_unique_id(void *id)
{
    if (!id) return fail(nullptr, PT_ERR_INVALID, "pt_comm_unique_id: null argument");
    static_assert(PT_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "pt.h's id size is RCCL's");
    Rccl &r = rccl();
    if (!r.ok) return fail(nullptr, PT_ERR_UNSUPPORTED, "pt_comm_unique_id: " + r.why);
    ncclUniqueId uid;
    PT_NCCL(nullptr, r.get_unique_id(&uid));
    std::memcpy(id, &uid, sizeof uid);
    return PT_OK;
}

int pt_comm_init_rank(pt_ctx *c, const void *id, int32_t rank, int32_t n_ranks)
{
    if (int s = check_ctx(c)) return s;
    if (c->group) return fail(c, PT_ERR_INVALID, "pt_comm_init_rank: a pt_create_multi context has its communicators already");
    if (!id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(c, PT_ERR_INVALID, "pt_comm_init_rank: need an id and 0 <= rank < n_ranks");
    if (c->comm) return fail(c, PT_ERR_INVALID, "pt_comm_init_rank: the context has a communicator already");
    Rccl &r = rccl();
    if (!r.ok) return fail(c, PT_ERR_UNSUPPORTED, "pt_comm_init_rank: " + r.why);
    if (!c->d_total) PT_HIP(c, hipMalloc((void **)&c->d_total, npix(c) * 3 * sizeof(float)));
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof uid);
    ncclComm_t comm = nullptr;
    PT_NCCL(c, r.comm_init_rank(&comm, n_ranks, uid, rank));
    c->comm = comm; c->comm_rank = rank; c->comm_size = n_ranks;
    c->comm_ref = std::shared_ptr<void>(comm, [](void *cm) { if (cm) (void)rccl().comm_destroy((ncclComm_t)cm); });
    return PT_OK;
}

int pt_comm_share(pt_ctx *c, pt_ctx *owner)
{
    if (int s = check_ctx(c)) return s;
    if (!owner || owner == c) return fail(c, PT_ERR_INVALID, "pt_comm_share: need another context");
    if (c->group || owner->group) return fail(c, PT_ERR_INVALID, "pt_comm_share: not for pt_create_multi contexts");
    if (!owner->comm) return fail(c, PT_ERR_INVALID, "pt_comm_share: the other context has no communicator (pt_comm_init_rank first)");
    if (c->comm) return fail(c, PT_ERR_INVALID, "pt_comm_share: the context has a communicator already");
    if (c->p.device != owner->p.device) return fail(c, PT_ERR_INVALID, "pt_comm_share: the contexts are on different devices");
    if (c->p.height != owner->p.height || c->p.width != owner->p.width)
        return fail(c, PT_ERR_INVALID, "pt_comm_share: the contexts' canvases differ in size");
    if (!c->d_total) PT_HIP(c, hipMalloc((void **)&c->d_total, npix(c) * 3 * sizeof(float)));
    c->comm = owner->comm; c->comm_ref = owner->comm_ref;
    c->comm_rank = owner->comm_rank; c->comm_size = owner->comm_size;
    return PT_OK;
}

int pt_allreduce_sum(pt_ctx *c)
{
    if (int s = check_ctx(c)) return s;
    if (c->group) { pt_ctx *on; return group_total(c, &on); }
    if (!c->comm) return fail(c, PT_ERR_INVALID, "pt_allreduce_sum: no communicator (pt_comm_init_rank first)");
    if (int s = materialise_canvas(c)) return s;
    PT_NCCL(c, rccl().all_reduce(c->d_canvas, c->d_total, npix(c) * 3, ncclFloat, ncclSum, (ncclComm_t)c->comm, c->stream));
    c->total_valid = true;
    return PT_OK;
}

int pt_comm_info(pt_ctx *c, int32_t *n_ranks, int32_t *rank, int32_t *n_local, int32_t *rccl_version)
{
    if (!c) return fail(nullptr, PT_ERR_INVALID, "pt_comm_info: null context");
    const bool via_rccl = c->group ? !c->group->comms.empty() : c->comm != nullptr;
    if (n_ranks) *n_ranks = c->group ? (int32_t)c->group->devices.size() : c->comm_size;
    if (rank) *rank = c->group ? 0 : c->comm_rank;
    if (n_local) *n_local = c->group ? (int32_t)c->group->members.size() : 1;
    if (rccl_version) *rccl_version = via_rccl ? rccl().version : 0;
    return PT_OK;
}

} // extern "C"
