// Several GPUs of one node behind one handle (include/nori_hip.h: nori_hip_group_*).
//
// A group owns one NoriHipContext per device and drives them from the caller's thread: the scene is uploaded to every device
// (replicated: 1.5 GB for ten million triangles against 288 GB of HBM), a render call gives every device its share of the work -
// a slice of the sample range, or every n-th 32-row band of the image - on one host thread per device, and the fp32 films are
// then summed in place over xGMI with one RCCL all-reduce (the accumulate-then-merge of ImageBlock::put(ImageBlock&),
// src/block.cpp:131-140, across devices).  Streams are keyed by the global sample index, so the summed film is the same estimator
// for any number of devices.  Written on top of the public C ABI only; RCCL is loaded on first use (dlopen), so processes
// that never make a group do not load it.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <string>
#include <thread>
#include <vector>
#include <algorithm>
#include <mutex>
#include "../../include/nori_hip.h"

extern "C" void nori_hip_set_error_(const char *msg);       // nori_hip.hip: the calling thread's last-error text

namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl; std::mutex g_rccl_mutex;

int fail(int code, const std::string &m) { nori_hip_set_error_(m.c_str()); return code; }

int load_rccl() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.lib) return NORI_HIP_OK;
    void *h = nullptr;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { h = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (h) break; }
    if (!h) return fail(NORI_HIP_EDEVICE, std::string("cannot load RCCL: ") + dlerror());
    Rccl r; r.lib = h;
    r.CommInitAll = (decltype(r.CommInitAll)) dlsym(h, "ncclCommInitAll"); r.CommDestroy = (decltype(r.CommDestroy)) dlsym(h, "ncclCommDestroy");
    r.AllReduce = (decltype(r.AllReduce)) dlsym(h, "ncclAllReduce"); r.GroupStart = (decltype(r.GroupStart)) dlsym(h, "ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd)) dlsym(h, "ncclGroupEnd"); r.GetErrorString = (decltype(r.GetErrorString)) dlsym(h, "ncclGetErrorString");
    if (!r.CommInitAll || !r.CommDestroy || !r.AllReduce || !r.GroupStart || !r.GroupEnd || !r.GetErrorString) { dlclose(h); return fail(NORI_HIP_EDEVICE, "RCCL library lacks a needed symbol"); }
    g_rccl = r;
    return NORI_HIP_OK;
}

} // namespace

struct NoriHipGroup {
    std::vector<int> devices;
    std::vector<NoriHipContext *> ctx;
    std::vector<ncclComm_t> comms;
    double reduce_ms = 0.0;           // device time of the last all-reduce on device 0's stream
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

namespace {

// runs fn(i) for every member on its own host thread (a render call blocks its caller while it feeds the device), and returns
// the first failure with its message
template <typename F> int for_each_device(NoriHipGroup *g, F fn) {
    const size_t n = g->ctx.size();
    std::vector<int> rc(n, NORI_HIP_OK); std::vector<std::string> msg(n);
    auto body = [&](size_t i) { rc[i] = fn(i); if (rc[i] != NORI_HIP_OK) msg[i] = nori_hip_last_error(); };
    if (n == 1) body(0);
    else {
        std::vector<std::thread> th;
        for (size_t i = 0; i < n; ++i) th.emplace_back(body, i);
        for (auto &t : th) t.join();
    }
    for (size_t i = 0; i < n; ++i) if (rc[i] != NORI_HIP_OK) return fail(rc[i], "device " + std::to_string(g->devices[i]) + ": " + msg[i]);
    return NORI_HIP_OK;
}

} // namespace

extern "C" {

int nori_hip_group_create(const int *devices, int n, NoriHipGroup **out) {
    if (!out) return fail(NORI_HIP_EINVAL, "out is null");
    *out = nullptr;
    if (!devices || n < 1 || n > 64) return fail(NORI_HIP_EINVAL, "device list is null, empty or longer than 64");
    for (int i = 0; i < n; ++i) for (int j = 0; j < i; ++j) if (devices[i] == devices[j]) return fail(NORI_HIP_EINVAL, "device " + std::to_string(devices[i]) + " is listed twice");
    int rc = load_rccl();
    if (rc) return rc;
    NoriHipGroup *g = new NoriHipGroup();
    g->devices.assign(devices, devices + n);
    for (int i = 0; i < n; ++i) {
        NoriHipContext *c = nullptr;
        if ((rc = nori_hip_create(devices[i], &c))) { const std::string why = nori_hip_last_error(); nori_hip_group_destroy(g); return fail(rc, why); }
        g->ctx.push_back(c);
    }
    g->comms.assign(n, nullptr);
    ncclResult_t nr = g_rccl.CommInitAll(g->comms.data(), n, g->devices.data());
    if (nr != ncclSuccess) { g->comms.clear(); nori_hip_group_destroy(g); return fail(NORI_HIP_EDEVICE, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(nr)); }
    if (hipSetDevice(devices[0]) != hipSuccess || hipEventCreate(&g->ev0) != hipSuccess || hipEventCreate(&g->ev1) != hipSuccess) { nori_hip_group_destroy(g); return fail(NORI_HIP_EDEVICE, "event creation failed"); }
    *out = g;
    return NORI_HIP_OK;
}

int nori_hip_group_destroy(NoriHipGroup *g) {
    if (!g) return NORI_HIP_OK;
    for (size_t i = 0; i < g->comms.size(); ++i) if (g->comms[i]) { (void) hipSetDevice(g->devices[i]); g_rccl.CommDestroy(g->comms[i]); }
    if (!g->devices.empty()) (void) hipSetDevice(g->devices[0]);
    if (g->ev0) (void) hipEventDestroy(g->ev0);
    if (g->ev1) (void) hipEventDestroy(g->ev1);
    for (NoriHipContext *c : g->ctx) nori_hip_destroy(c);
    delete g;
    return NORI_HIP_OK;
}

int nori_hip_group_size(NoriHipGroup *g) { return g ? (int) g->ctx.size() : 0; }
NoriHipContext *nori_hip_group_context(NoriHipGroup *g, int i) { return (g && i >= 0 && i < (int) g->ctx.size()) ? g->ctx[i] : nullptr; }

int nori_hip_group_upload_scene(NoriHipGroup *g, const NoriHipSceneDesc *desc, int build_on_device) {
    if (!g) return fail(NORI_HIP_EINVAL, "null group");
    return for_each_device(g, [&](size_t i) { return build_on_device ? nori_hip_upload_scene_build(g->ctx[i], desc) : nori_hip_upload_scene(g->ctx[i], desc); });
}

int nori_hip_group_set_integrator(NoriHipGroup *g, int kind) {
    if (!g) return fail(NORI_HIP_EINVAL, "null group");
    for (NoriHipContext *c : g->ctx) { int rc = nori_hip_set_integrator(c, kind); if (rc) return rc; }
    return NORI_HIP_OK;
}
int nori_hip_group_set_film_filter(NoriHipGroup *g, const float *table33, float radius) {
    if (!g) return fail(NORI_HIP_EINVAL, "null group");
    for (NoriHipContext *c : g->ctx) { int rc = nori_hip_set_film_filter(c, table33, radius); if (rc) return rc; }
    return NORI_HIP_OK;
}
int nori_hip_group_clear_film(NoriHipGroup *g) {
    if (!g) return fail(NORI_HIP_EINVAL, "null group");
    for (NoriHipContext *c : g->ctx) { int rc = nori_hip_clear_film(c); if (rc) return rc; }
    return NORI_HIP_OK;
}

// sums the members' films in place: afterwards every device holds the film of the whole job
int nori_hip_group_reduce_films(NoriHipGroup *g) {
    if (!g) return fail(NORI_HIP_EINVAL, "null group");
    const size_t n = g->ctx.size();
    std::vector<void *> film(n); std::vector<void *> stream(n); size_t bytes = 0;
    for (size_t i = 0; i < n; ++i) {
        size_t b = 0; int rc;
        if ((rc = nori_hip_film_device_ptr(g->ctx[i], &film[i], &b))) return rc;
        if ((rc = nori_hip_stream(g->ctx[i], &stream[i]))) return rc;
        if (i == 0) bytes = b; else if (b != bytes) return fail(NORI_HIP_ESTATE, "the members of the group hold films of different sizes");
    }
    (void) hipSetDevice(g->devices[0]);
    (void) hipEventRecord(g->ev0, (hipStream_t) stream[0]);
    ncclResult_t nr = g_rccl.GroupStart();
    for (size_t i = 0; i < n && nr == ncclSuccess; ++i) {
        (void) hipSetDevice(g->devices[i]);
        nr = g_rccl.AllReduce(film[i], film[i], bytes / sizeof(float), ncclFloat, ncclSum, g->comms[i], (hipStream_t) stream[i]);   // on the context's stream: ordered after its render
    }
    const ncclResult_t ne = g_rccl.GroupEnd();
    if (nr != ncclSuccess || ne != ncclSuccess) return fail(NORI_HIP_EDEVICE, std::string("ncclAllReduce: ") + g_rccl.GetErrorString(nr != ncclSuccess ? nr : ne));
    (void) hipSetDevice(g->devices[0]);
    (void) hipEventRecord(g->ev1, (hipStream_t) stream[0]);
    for (size_t i = 0; i < n; ++i) { int rc = nori_hip_synchronize(g->ctx[i]); if (rc) return rc; }
    float ms = 0.f; if (hipEventElapsedTime(&ms, g->ev0, g->ev1) == hipSuccess) g->reduce_ms = ms; else (void) hipGetLastError();
    return NORI_HIP_OK;
}

// The split of a group render as a pure function of (members, member, sample range, image height): see include/nori_hip.h.
int nori_hip_group_plan(int n, int i, int spp_begin, int spp_end, int height, int partition, int *out_b, int *out_e, int *bands, int max_bands, int *out_n_bands) {
    if (n < 1 || i < 0 || i >= n) return fail(NORI_HIP_EINVAL, "nori_hip_group_plan: no member " + std::to_string(i) + " in a group of " + std::to_string(n));
    if (spp_begin < 0 || spp_end <= spp_begin) return fail(NORI_HIP_EINVAL, "empty sample range");
    if (height < 1) return fail(NORI_HIP_EINVAL, "bad image height");
    if (partition != NORI_HIP_PARTITION_SPP && partition != NORI_HIP_PARTITION_TILES) return fail(NORI_HIP_EINVAL, "unknown partition");
    if (!out_b || !out_e || !out_n_bands) return fail(NORI_HIP_EINVAL, "null argument");
    *out_n_bands = 0;
    if (partition == NORI_HIP_PARTITION_SPP) {
        // contiguous slices, the first (count mod n) one sample longer
        const int total = spp_end - spp_begin, base = total / n, extra = total % n;
        *out_b = spp_begin + i * base + std::min(i, extra);
        *out_e = *out_b + base + (i < extra ? 1 : 0);
        return NORI_HIP_OK;
    }
    // every n-th band of NORI_BLOCK_SIZE (32) rows, all samples
    *out_b = spp_begin; *out_e = spp_end;
    int k = 0;
    for (long long y = 32ll * i; y < height; y += 32ll * n, ++k)
        if (bands && k < max_bands) { bands[2 * k] = (int) y; bands[2 * k + 1] = (int) std::min<long long>(y + 32, height); }
    *out_n_bands = k;
    return NORI_HIP_OK;
}

int nori_hip_group_render(NoriHipGroup *g, int spp_begin, int spp_end, int partition) {
    if (!g) return fail(NORI_HIP_EINVAL, "null group");
    if (spp_begin < 0 || spp_end <= spp_begin) return fail(NORI_HIP_EINVAL, "empty sample range");
    if (partition != NORI_HIP_PARTITION_SPP && partition != NORI_HIP_PARTITION_TILES) return fail(NORI_HIP_EINVAL, "unknown partition");
    const int n = (int) g->ctx.size();
    int rows = 0, cols = 0, border = 0;
    int rc = nori_hip_film_dims(g->ctx[0], &rows, &cols, &border);
    if (rc) return rc;
    const int W = cols - 2 * border, H = rows - 2 * border;
    rc = for_each_device(g, [&](size_t i) -> int {
        // After a group render every member holds the job's film.  What the films held before this call must enter the sum once:
        // it stays in member 0's film, the other members start from zero.
        if (i > 0) { int r0 = nori_hip_clear_film(g->ctx[i]); if (r0) return r0; }
        int b = 0, e = 0, n_bands = 0;
        const int max_bands = (H + 31) / 32;
        std::vector<int> bands((size_t) 2 * max_bands);
        int r1 = nori_hip_group_plan(n, (int) i, spp_begin, spp_end, H, partition, &b, &e, bands.data(), max_bands, &n_bands);
        if (r1) return r1;
        if (partition == NORI_HIP_PARTITION_SPP) {
            if (e <= b) return NORI_HIP_OK;                         // more members than samples: nothing for this one
            int r2 = nori_hip_render(g->ctx[i], 0, 0, W, H, b, e);
            return r2 ? r2 : nori_hip_synchronize(g->ctx[i]);
        }
        if (n_bands == 0) return NORI_HIP_OK;                      // more members than bands
        std::vector<int> tiles;
        for (int k = 0; k < n_bands; ++k) { tiles.push_back(0); tiles.push_back(bands[2 * k]); tiles.push_back(W); tiles.push_back(bands[2 * k + 1]); }
        int r2 = nori_hip_render_tiles(g->ctx[i], tiles.data(), n_bands, b, e);      // the member's bands in ONE pass
        return r2 ? r2 : nori_hip_synchronize(g->ctx[i]);
    });
    if (rc == NORI_HIP_OK) rc = nori_hip_group_reduce_films(g);
    if (rc) {
        // A member failed: some members have dropped what they held, some have rendered a part of their share, and nothing was
        // summed (or the sum itself failed half-way).  No film is the job's film any more; all are cleared so that a caller who
        // goes on sees an empty film and not a plausible wrong one.  (include/nori_hip.h documents this.)
        const std::string why = nori_hip_last_error();
        for (NoriHipContext *c : g->ctx) { (void) nori_hip_clear_film(c); (void) nori_hip_synchronize(c); }
        return fail(rc, why + " (the films of all group members were cleared)");
    }
    return NORI_HIP_OK;
}

int nori_hip_group_reduce_ms(NoriHipGroup *g, double *ms) { if (!g || !ms) return fail(NORI_HIP_EINVAL, "null argument"); *ms = g->reduce_ms; return NORI_HIP_OK; }

} // extern "C"
