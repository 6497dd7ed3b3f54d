// libnori_hip.so — gfx950 kernels and the C ABI of include/nori_hip.h.
//
// Kernels
//   k_trace        one ray per lane, closest- or any-hit (BVH::rayIntersect, src/bvh.cpp:404-463)
//   k_camera_rays  PerspectiveCamera::sampleRay (src/perspective.cpp:90-111)
//   k_li_samples   one camera sample per lane, seeded per sample (parity entry)
//   k_shade, k_trace_pool<any|closest>, k_finish
//                  the default render pipeline: PathMisIntegrator::Li (src/pathMisIntegrator.cpp:18-168) and the
//                  render loop (src/render.cpp:72-101) as stages over a pool of path slots (see "Wavefront pipeline")
//   k_render_wN    NORI_HIP_PIPELINE=mega: persistent lanes with per-lane path regeneration; every lane runs one
//                  bounce per loop trip (extend -> shade -> shadow), a lane whose path ended splats it to the film
//                  and takes the next camera sample; N = waves per SIMD the variant is built for
//   k_bitmap       ImageBlock::toBitmap (src/block.cpp:76-82)
//   k_srgb, k_nlmeans   the filter pass of the CUDA fork (cuda/filter.cc.cu:33-110)
#include <hip/hip_runtime.h>
#include "nori_lbvh.h"
#include <string>
#include <vector>
#include <cstring>
#include <cstdio>
#include <cmath>
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <thread>
#include "nori_device.h"

#pragma clang fp contract(off)

using namespace nd;

namespace {

thread_local std::string g_err;
int fail(int code, const std::string &m) { g_err = m; return code; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(NORI_HIP_EDEVICE, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

// device scratch of one API call: released on every return path, error returns included
struct DevBufs {
    std::vector<void *> owned;
    ~DevBufs() { for (void *q : owned) (void) hipFree(q); }
    template <typename T> hipError_t get(T *&out, size_t bytes) {
        void *q = nullptr; hipError_t e = hipMalloc(&q, bytes ? bytes : 16);
        if (e == hipSuccess) owned.push_back(q);
        out = static_cast<T *>(q);
        return e;
    }
};

// Atomics that hit the same 256-byte line are served one after the other by the memory side, about 11 ns each on
// MI355X (scratch/atombench.hip), whatever their address within the line; distinct lines proceed in parallel.
// Every counter the kernels bump therefore sits on a line of its own, and the hot ones are striped over ND_STRIPES
// lines that the reader sums.
#define ND_STRIPES 32
struct alignas(256) DevStats { unsigned long long samples, closest, shadow, invalid, nodes, prims, ext_rays, ext_nodes, ext_prims, ext_visits, culled, dropped; unsigned long long prof[2][10]; };
__device__ __forceinline__ DevStats *my_stats(DevStats *stats) { return stats + (((blockIdx.x * blockDim.x + threadIdx.x) >> 6) & (ND_STRIPES - 1)); }

// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void k_trace_result(const DevScene &sc, const Ray &ray, const Hit &hit, bool found, NoriHipHit &out, float *full, size_t i) {
    if (found) {
        Surface sf; surface_from_hit<true>(sc, ray, hit, sf, true);
        out.prim = sf.prim; out.t = hit.t; out.u = sf.uv.x; out.v = sf.uv.y;
        if (full) {
            float *f = full + 16 * i;
            f[0] = sf.p.x; f[1] = sf.p.y; f[2] = sf.p.z; f[3] = sf.uv.x; f[4] = sf.uv.y;
            f[5] = sf.sh.n.x; f[6] = sf.sh.n.y; f[7] = sf.sh.n.z; f[8] = sf.geo_n.x; f[9] = sf.geo_n.y; f[10] = sf.geo_n.z;
            f[11] = (float) sf.shape; f[12] = f[13] = f[14] = f[15] = 0.f;
        }
    } else if (full) { float *f = full + 16 * i; for (int k = 0; k < 16; ++k) f[k] = 0.f; }
}
__global__ __launch_bounds__(256) void k_trace(DevScene sc, const NoriHipRay *rays, size_t n, int any_hit, NoriHipHit *hits, float *full) {
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 *rp = reinterpret_cast<const float4 *>(rays + i);
    float4 a = rp[0], b = rp[1];
    Ray ray; ray.o = f3(a.x, a.y, a.z); ray.mint = a.w; ray.d = f3(b.x, b.y, b.z); ray.maxt = b.w;
    ScratchStackT<ND_STACK8> stk; Hit hit; uint32_t nn = 0, np = 0;
    NoriHipHit out; out.prim = NORI_HIP_MISS; out.t = 0.f; out.u = 0.f; out.v = 0.f;
    // (a scene whose trace stage walks quantised four-box nodes is answered by that traversal here too: same node step, same two-record
    // leaf step, stack entries cut like the trace stage's)
    if (sc.nodes4q && !sc.nodes8) {
        // (entry distances are at least mint: the cut entries, which have no sign bit, are for rays with mint >= 0 - every ray a render
        // traces over these nodes; a caller's ray with a negative mint keeps full entries)
        bool fq;
        if (ray.mint >= 0.f) { ScratchStackCutT<ND_STACK> sq; fq = any_hit ? trace4q<true>(sc, ray, hit, sq, nn, np) : trace4q<false>(sc, ray, hit, sq, nn, np); }
        else fq = any_hit ? trace4q<true>(sc, ray, hit, stk, nn, np) : trace4q<false>(sc, ray, hit, stk, nn, np);
        if (any_hit) { if (fq) out.prim = 1u; hits[i] = out; return; }
        k_trace_result(sc, ray, hit, fq, out, full, i); hits[i] = out;
        return;
    }
    if (any_hit) {
        if (sc.nodes8 ? trace8<true>(sc, ray, hit, stk, nn, np) : trace<true>(sc, ray, hit, stk, nn, np)) out.prim = 1u;
        hits[i] = out;
        return;
    }
    bool found = sc.nodes8 ? trace8<false>(sc, ray, hit, stk, nn, np) : trace<false>(sc, ray, hit, stk, nn, np);
    k_trace_result(sc, ray, hit, found, out, full, i);
    hits[i] = out;
}

__global__ __launch_bounds__(256) void k_camera_rays(DevScene sc, const float *ps, size_t n, NoriHipRay *rays) {
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Ray r; camera_ray(sc, f2(ps[2 * i], ps[2 * i + 1]), r);
    NoriHipRay o; o.o[0] = r.o.x; o.o[1] = r.o.y; o.o[2] = r.o.z; o.mint = r.mint; o.d[0] = r.d.x; o.d[1] = r.d.y; o.d[2] = r.d.z; o.maxt = r.maxt;
    rays[i] = o;
}

template <bool EXT, int KIND = NORI_HIP_INTEGRATOR_PATH_MIS>
__global__ __launch_bounds__(256) void k_li_samples(DevScene sc, size_t n, const int32_t *px, const int32_t *py, const uint64_t *states,
                                                    const uint64_t *seqs, const NoriHipRay *rays, float *rgb) {
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Pcg32 rng; rng.seed(states[i], seqs[i]);
    F2 j = rng.next2D(); (void) rng.next2D();
    Ray ray;
    if (rays) { const NoriHipRay &q = rays[i]; ray.o = f3(q.o); ray.d = f3(q.d); ray.mint = q.mint; ray.maxt = q.maxt; }
    else if (px) camera_ray(sc, f2((float) px[i] + j.x, (float) py[i] + j.y), ray);
    else camera_ray(sc, f2(j.x * (float) sc.cam.width, j.y * (float) sc.cam.height), ray);      // src/ttest.cpp:171-172
    PathState ps; path_init(ps, ray);
    ScratchStack stk; Counters c = {0, 0, 0, 0, 0, 0};
    GlobalNodes g; g.nodes = sc.nodes;
    if (KIND == NORI_HIP_INTEGRATOR_PATH_MATS) { while (path_bounce_mats<EXT>(sc, g, ps, rng, stk, c)) { } }
    else if (KIND == NORI_HIP_INTEGRATOR_DIRECT_MIS) path_direct_mis<EXT>(sc, g, ps, rng, stk, c);
    else while (path_bounce<EXT>(sc, g, ps, rng, stk, c)) { }
    rgb[3 * i] = ps.li.x; rgb[3 * i + 1] = ps.li.y; rgb[3 * i + 2] = ps.li.z;
}

// BSDF::sample / eval / pdf alone (the BSDF mode of the reference's test plugins, src/ttest.cpp:107-147, src/chi2test.cpp:93-196)
// out per query: weight rgb, wo xyz, pdf, eval rgb, isDelta, wo_valid
__global__ __launch_bounds__(256) void k_bsdf_vectors(NoriHipShape b, size_t n, const float *wi, const float *sample, const float *sample2, const float *wo_in, float *out) {
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const F3 w_i = f3(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]);
    const F3 alb = f3(b.albedo);
    F3 wo = f3(0.f), weight = f3(0.f); bool is_delta = false, wo_valid = true;
    if (wo_in) wo = f3(wo_in[3 * i], wo_in[3 * i + 1], wo_in[3 * i + 2]);
    else {
        const F2 s1 = f2(sample[2 * i], sample[2 * i + 1]), s2 = sample2 ? f2(sample2[2 * i], sample2[2 * i + 1]) : s1;
        weight = bsdf_sample(b, alb, w_i, s1, s2, wo, is_delta, wo_valid);
    }
    const float pdf = wo_valid ? bsdf_pdf(b, w_i, wo) : 0.f;
    const F3 ev = wo_valid ? bsdf_eval(b, alb, w_i, wo) : f3(0.f);
    float *o = out + 12 * i;
    o[0] = weight.x; o[1] = weight.y; o[2] = weight.z; o[3] = wo.x; o[4] = wo.y; o[5] = wo.z; o[6] = pdf; o[7] = ev.x; o[8] = ev.y; o[9] = ev.z;
    o[10] = is_delta ? 1.f : 0.f; o[11] = wo_valid ? 1.f : 0.f;
}
// environment emitter: the sampling weight of every (u, v) cell, (r + g + b) * sin(theta) at the cell's centre (plus a floor, so that
// no direction with radiance has probability zero)
__global__ __launch_bounds__(256) void k_env_weights(DevScene sc, uint32_t w, uint32_t h, float *weights) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= w * h) return;
    const uint32_t i = k / w, j = k % w;
    const float u = ((float) i + 0.5f) / (float) h, v = ((float) j + 0.5f) / (float) w;
    const F3 c = tex_color(sc, sc.textures[sc.env.texture - 1], f2(u, v));
    weights[k] = (c.x + c.y + c.z + 1e-4f) * sinf(u * ND_PI);
}
// one texture of the uploaded scene at n uv pairs (nori_hip_texture_eval)
__global__ __launch_bounds__(256) void k_texture_eval(DevScene sc, uint32_t k, size_t n, const float *uv, const float *nrm, float *out) {
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const NoriHipTexture &t = sc.textures[k];
    const F2 q = f2(uv[2 * i], uv[2 * i + 1]);
    F3 r;
    if (t.kind == NORI_HIP_TEX_COLOR || t.kind == NORI_HIP_TEX_CHECKER_COLOR) r = tex_color(sc, t, q);
    else { Frame fr; fr.set(f3(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2])); r = tex_normal(sc, t, q, fr); }
    out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
}
// n draws of pcg32.seed(state, seq) by one lane (known-answer test of the device sampler, ext/pcg32/pcg32-demo.out)
__global__ void k_pcg32(uint64_t state, uint64_t seq, unsigned int n, uint32_t *out_u32, float *out_f32) {
    Pcg32 r; r.seed(state, seq);
    for (unsigned int i = 0; i < n; ++i) { if (out_u32) out_u32[i] = r.next_uint(); else out_f32[i] = r.next1D(); }
}

// ---------------------------------------------------------------------------------------------------------
struct RenderJob {
    int x0, y0, tw, th;               // the tile (n_tiles == 1)
    int spp_begin, nspp;
    unsigned long long total;         // camera samples of the call
    unsigned long long padded_total;  // work items: 64 per 8x8 patch and sample, edge patches padded
    int n_tiles;                      // > 1: the call renders a list of tiles (nori_hip_render_tiles)
    const int4 *tiles;                // x0, y0, tw, th per tile (device memory)
    const unsigned int *patch_first;  // n_tiles + 1: patches before each tile
};

// Work item w -> (8x8 pixel patch, sample, lane): the 64 items of a chunk are the 64 pixels of one patch for one
// sample index, so the primary rays of a freshly started wave are coherent.  Patches are numbered tile after tile.
// false: padding of an edge patch (the pixel lies outside its tile).
__device__ __forceinline__ bool work_to_sample(const RenderJob &job, unsigned long long w, int &px, int &py, int &s) {
    unsigned int lane = (unsigned int) (w & 63u);
    unsigned long long chunk = w >> 6;
    unsigned long long patch = chunk / (unsigned long long) job.nspp;
    s = job.spp_begin + (int) (chunk % (unsigned long long) job.nspp);
    int x0 = job.x0, y0 = job.y0, tw = job.tw, th = job.th;
    if (job.n_tiles > 1) {
        int lo = 0, hi = job.n_tiles - 1;                                  // last tile whose first patch is <= patch
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if ((unsigned long long) job.patch_first[mid] <= patch) lo = mid; else hi = mid - 1; }
        const int4 t = job.tiles[lo];
        x0 = t.x; y0 = t.y; tw = t.z; th = t.w; patch -= job.patch_first[lo];
    }
    const int ptx = (tw + 7) >> 3;
    int pyi = (int) (patch / (unsigned long long) ptx), pxi = (int) (patch % (unsigned long long) ptx);
    px = x0 + pxi * 8 + (int) (lane & 7u);
    py = y0 + pyi * 8 + (int) (lane >> 3);
    return px < x0 + tw && py < y0 + th;
}

// ImageBlock::put(sample) as written (src/block.cpp:93-111: single-pixel add, then return) in the coordinates of
// the sample's 32x32 block, moved into the full film as ImageBlock::put(ImageBlock&) does (:131-140).
__device__ __forceinline__ bool film_put(const DevScene &sc, float4 *film, F2 pos, int px, int py, F3 v) {
    if (!color_valid(v)) return false;
    int ox = (px >> 5) << 5, oy = (py >> 5) << 5;
    float fx = pos.x - 0.5f - (float) (ox - sc.border), fy = pos.y - 0.5f - (float) (oy - sc.border);
    if (sc.filt_table) {
        // upstream Nori's put(): every pixel within the filter radius gets value * w(x) * w(y) (src/block.cpp:113-128, the code
        // behind this fork's early return).  With border = ceil(radius - 0.5) the rectangle of a sample never leaves its block's
        // bordered area, so nothing is clipped and the result does not depend on the blocks.
        const int x0 = (int) ceilf(fx - sc.filt_radius), y0 = (int) ceilf(fy - sc.filt_radius);
        const int x1 = (int) floorf(fx + sc.filt_radius), y1 = (int) floorf(fy + sc.filt_radius);
        const int rows = sc.cam.height + 2 * sc.border;
        for (int y = y0; y <= y1; ++y) {
            const float wy = sc.filt_table[(int) (fabsf((float) y - fy) * sc.filt_lookup)];
            for (int x = x0; x <= x1; ++x) {
                const float wx = sc.filt_table[(int) (fabsf((float) x - fx) * sc.filt_lookup)];
                const int ix = x + ox, iy = y + oy;
                if (ix < 0 || iy < 0 || ix >= sc.film_cols || iy >= rows) continue;
                float *p = reinterpret_cast<float *>(film + (size_t) iy * sc.film_cols + ix);
                unsafeAtomicAdd(p + 0, (v.x * wx) * wy); unsafeAtomicAdd(p + 1, (v.y * wx) * wy); unsafeAtomicAdd(p + 2, (v.z * wx) * wy); unsafeAtomicAdd(p + 3, (1.0f * wx) * wy);
            }
        }
        return true;
    }
    int ix = (int) fx + ox, iy = (int) fy + oy;
    float *p = reinterpret_cast<float *>(film + (size_t) iy * sc.film_cols + ix);
    unsafeAtomicAdd(p + 0, v.x); unsafeAtomicAdd(p + 1, v.y); unsafeAtomicAdd(p + 2, v.z); unsafeAtomicAdd(p + 3, 1.0f);
    return true;
}

// Work items are handed out ND_WORK_GRAB at a time per wave (one atomic by one lane) and dealt to the lanes whose
// path has ended with a ballot + prefix count, so the global counter sees one atomic per ND_WORK_GRAB samples.
#define ND_WORK_GRAB 256u
#define ND_MAX_TILES 65536          /* tiles of one nori_hip_render_tiles call (a 4K image has 8 160 blocks of 32 x 32) */
// Workgroups of 256 lanes; LDS holds the traversal stacks (MEGA_DEPTH entries per lane) and, in the rest of the
// workgroup's share, the top `n_cached` nodes of the tree.  The kernel is latency-bound (a dependent node or record
// fetch per trip), so it is built for several register budgets (waves per SIMD) and the host picks one.
#define MEGA_THREADS 256
#define MEGA_DEPTH 8
template <bool EXT, int KIND = NORI_HIP_INTEGRATOR_PATH_MIS, bool SS = EXT>
__device__ __forceinline__ void render_body(const DevScene &sc, const RenderJob &job, float4 *film, unsigned long long *work_counter, DevStats *stats, unsigned int n_cached) {
    extern __shared__ float4 dyn_lds[];
    uint2 stack_spill[ND_STACK - MEGA_DEPTH];
    LdsStackT<MEGA_DEPTH> stk; stk.spill = stack_spill; stk.base = (lds_u64 *) dyn_lds + (threadIdx.x >> 6) * (MEGA_DEPTH * 64) + (threadIdx.x & 63u);
    CachedNodes nodes; nodes.nodes = sc.nodes; nodes.n = n_cached;
    nodes.cache = (lds_f4 *) dyn_lds + (MEGA_THREADS / 64) * (MEGA_DEPTH * 64 * 8 / 16);
    nodes.fill();
    __syncthreads();
    Counters ctr = {0, 0, 0, 0, 0, 0};
    unsigned int n_samples = 0, n_invalid = 0;
    PathState ps; Pcg32 rng; F2 pixel_sample = f2(0.f, 0.f); int px = 0, py = 0;
    bool alive = false;
    const unsigned int lane = threadIdx.x & 63u;
    const unsigned long long padded_total = job.padded_total;
    unsigned long long cursor = 0, end = 0;                               // wave-uniform: work items [cursor, end) are ours
    bool exhausted = false;                                               // wave-uniform
    for (;;) {
        const unsigned long long dead = __ballot(!alive);
        if (dead && !(exhausted && cursor >= end)) {
            if (cursor >= end) {
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(work_counter, (unsigned long long) ND_WORK_GRAB);
                base = ((unsigned long long) __builtin_amdgcn_readfirstlane((unsigned int) (base >> 32)) << 32) | __builtin_amdgcn_readfirstlane((unsigned int) base);
                if (base >= padded_total) exhausted = true;
                else { cursor = base; end = (base + ND_WORK_GRAB < padded_total) ? base + ND_WORK_GRAB : padded_total; }
            }
            if (cursor < end) {
                const unsigned long long w = cursor + __builtin_amdgcn_mbcnt_hi((unsigned int) (dead >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int) dead, 0u));
                if (!alive && w < end) {
                    int s;
                    if (work_to_sample(job, w, px, py, s)) {                              // else: padding of an edge patch
                        rng.seed((uint64_t) py * (uint64_t) sc.cam.width + (uint64_t) px, (uint64_t) s);
                        F2 j = rng.next2D();
                        pixel_sample = f2((float) px + j.x, (float) py + j.y);
                        (void) rng.next2D();                                              // aperture sample, src/render.cpp:86
                        Ray r; camera_ray(sc, pixel_sample, r);
                        path_init(ps, r);
                        alive = true; n_samples++;
                    }
                }
                cursor += (unsigned long long) __popcll(dead);
            }
        }
        if (__ballot(alive) == 0ull) { if (exhausted && cursor >= end) break; else continue; }
        if (alive) {
            const bool go_on = KIND == NORI_HIP_INTEGRATOR_PATH_MATS ? path_bounce_mats<EXT>(sc, nodes, ps, rng, stk, ctr)
                             : KIND == NORI_HIP_INTEGRATOR_DIRECT_MIS ? path_direct_mis<EXT>(sc, nodes, ps, rng, stk, ctr) : path_bounce<EXT, SS>(sc, nodes, ps, rng, stk, ctr);
            if (!go_on) {
                if (!film_put(sc, film, pixel_sample, px, py, ps.li)) n_invalid++;
                alive = false;
            }
        }
    }
    // per-wave reduction of the counters, one set of atomics per wave on the wave's stripe
    unsigned long long v[8] = {n_samples, ctr.closest, ctr.shadow, n_invalid, ctr.nodes, ctr.prims, ctr.culled, ctr.dropped};
    for (int k = 0; k < 8; ++k) {
        unsigned long long x = v[k];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        v[k] = x;
    }
    if (lane == 0) {
        DevStats *st = my_stats(stats);
        atomicAdd(&st->samples, v[0]); atomicAdd(&st->closest, v[1]); atomicAdd(&st->shadow, v[2]);
        atomicAdd(&st->invalid, v[3]); atomicAdd(&st->nodes, v[4]); atomicAdd(&st->prims, v[5]); if (v[6]) atomicAdd(&st->culled, v[6]);
        if (v[7]) atomicAdd(&st->dropped, v[7]);
    }
}
#define ND_RENDER_KERNEL(W) \
__global__ __launch_bounds__(MEGA_THREADS) __attribute__((amdgpu_waves_per_eu(W, W))) \
void k_render_w##W(DevScene sc, RenderJob job, float4 *film, unsigned long long *work_counter, DevStats *stats, unsigned int n_cached) { \
    render_body<false>(sc, job, film, work_counter, stats, n_cached); }
ND_RENDER_KERNEL(3)
ND_RENDER_KERNEL(4)
ND_RENDER_KERNEL(5)
ND_RENDER_KERNEL(6)
ND_RENDER_KERNEL(8)
// scenes with images or subsurface shapes (DevScene::textures / NORI_HIP_BSDF_SUBSURFACE): one register budget
__global__ __launch_bounds__(MEGA_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_render_ext(DevScene sc, RenderJob job, float4 *film, unsigned long long *work_counter, DevStats *stats, unsigned int n_cached) {
    render_body<true>(sc, job, film, work_counter, stats, n_cached); }
// ... scenes with images but no subsurface shape (textureAndBump: 2 394 ms per step through k_render_ext once sampleDx's expf went
// through double arithmetic - 65 spilled vector registers instead of 55 in a kernel that never runs that code - and 2 192 through this one)
__global__ __launch_bounds__(MEGA_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_render_img(DevScene sc, RenderJob job, float4 *film, unsigned long long *work_counter, DevStats *stats, unsigned int n_cached) {
    render_body<true, NORI_HIP_INTEGRATOR_PATH_MIS, false>(sc, job, film, work_counter, stats, n_cached); }
// NORI_HIP_INTEGRATOR_PATH_MATS (src/pathMatsIntegrator.cpp): every scene kind through one variant
__global__ __launch_bounds__(MEGA_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_render_mats(DevScene sc, RenderJob job, float4 *film, unsigned long long *work_counter, DevStats *stats, unsigned int n_cached) {
    render_body<true, NORI_HIP_INTEGRATOR_PATH_MATS>(sc, job, film, work_counter, stats, n_cached); }
// NORI_HIP_INTEGRATOR_DIRECT_MIS (src/directMisIntegrator.cpp)
__global__ __launch_bounds__(MEGA_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_render_direct(DevScene sc, RenderJob job, float4 *film, unsigned long long *work_counter, DevStats *stats, unsigned int n_cached) {
    render_body<true, NORI_HIP_INTEGRATOR_DIRECT_MIS>(sc, job, film, work_counter, stats, n_cached); }

// ---------------------------------------------------------------------------------------------------------
// Wavefront pipeline: the bounce loop of PathMisIntegrator::Li cut into stages that run as separate kernels over
// a pool of path slots (struct-of-arrays in HBM, indexed by slot), so that each kernel keeps few registers live and
// every lane of a wave does the same kind of work.  One iteration = one bounce of every live path:
//   k_shade               adds the NEE term whose shadow ray the previous iteration traced, shades the new hits
//                         (emission+MIS, roulette, BSDF sample, emitter sample), splats finished paths and starts
//                         new ones in their slots from the global work counter; leaves the next ray and the
//                         shadow ray of every slot in the pool (or a marker maxt < mint for "none")
//   k_trace_pool<ANY>     shadow stage: any-hit for the shadow rays, writes one visibility word per slot
//   k_trace_pool<!ANY>    extend stage: closest hit for the next rays, writes the hit record per slot
// All three walk the pool in slot order, so their HBM accesses are coalesced, and nothing but the work counter,
// the live count and the statistics is shared between waves.
enum { PF_LAST_DELTA = 1u, PF_ENDING = 2u, PF_ACTIVE = 4u, PF_SHADOW = 8u,
       PF_PROBE = 16u,                  // the slot's ray is a subsurface probe: its hit ends the bounce that sv_o / sv_d / sv_h began
       PF_TRIES_SHIFT = 8 };            // bits 8..20: probes of this bounce that missed so far
#define ND_PROBE_MARK 0xfffffffeu       /* hit.x of a slot whose ray is a probe (hit.y = the shape it is restricted to), read by the trace stage */

struct Pool {
    float4 *ray_o;     // o.xyz, mint
    float4 *ray_d;     // d.xyz, maxt                (maxt < mint: no ray)
    uint4  *hit;       // record index (0xffffffff = miss), t, u, v
    float4 *thr;       // throughput rgb, pdf of the last BSDF sample
    float4 *li;        // radiance so far rgb, flags (PF_*)
    uint4  *rng;       // pcg32 state, inc
    float4 *pix;       // pixel sample x, y, integer pixel (x | y << 16), unused
    float4 *sh_o;      // shadow ray o.xyz, mint
    float4 *sh_d;      // shadow ray d.xyz, maxt     (maxt < mint: no ray)
    float4 *sh_c;      // NEE contribution rgb if unoccluded; w: set to 1 by the shadow stage when unoccluded
    float4 *sv_o, *sv_d; uint4 *sv_h;   // PF_PROBE: the ray and hit record of the bounce the probe belongs to
};
#define ND_POOL_ARRAYS 13

// Counters, one per 256-byte line (see DevStats).  v[0]: slots that hold a live path after the last shade stage.
#define ND_LINE 64                              /* unsigned ints per 256-byte line */
struct WfCounters { unsigned int v[(2 + 8) * ND_LINE]; };
#define CN_LIVE(c) ((c)->v[0])
#define CN_DYN(c, q) ((c)->v[(2 + (q)) * ND_LINE])        /* trace stage: eight cursors into the dynamically handed-out batches */

__global__ void k_pool_init(Pool pool, unsigned int n, WfCounters *cnt) {
    unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        pool.li[i] = make_float4(0.f, 0.f, 0.f, __uint_as_float(0u));
        pool.ray_d[i] = make_float4(0.f, 0.f, 0.f, -1.f); pool.ray_o[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        pool.sh_d[i] = make_float4(0.f, 0.f, 0.f, -1.f); pool.sh_o[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (i == 0) CN_LIVE(cnt) = 0u;
}
// (`report`: a word of pinned host memory that receives the live count the previous iteration's shade stage left - the host reads it three
// iterations later, after the event recorded behind this kernel.  A store from here instead of a 4-byte hipMemcpyAsync per iteration: the
// runtime's copy kernels of a process share one queue, where the copy of one pipeline waited behind the other pipeline's - 0.67 ms on
// average per iteration with two pipelines)
__global__ void k_iter_begin(WfCounters *cnt, unsigned int *report) {
    if (report) { *report = CN_LIVE(cnt); __threadfence_system(); }
    CN_LIVE(cnt) = 0u; for (int q = 0; q < 8; ++q) CN_DYN(cnt, q) = 0u;
}

// Once the work counter is exhausted the live paths thin out by about a quarter per bounce, while every stage still
// sweeps the whole pool.  When fewer than half of the slots are live the host has them packed into the front of a
// second pool of half the size (the live list of the last shade stage names them) and goes on with that one: slot
// numbers mean nothing outside the pool, so nothing else changes.  Slots [n, n_dst) of the new pool are left inactive.
__global__ __launch_bounds__(256) void k_compact_pool(Pool src, Pool dst, const unsigned int *q_live, const WfCounters *cnt, unsigned int n_dst) {
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x, n = CN_LIVE(cnt);
    if (i >= n_dst) return;
    if (i < n) {
        const unsigned int s = q_live[i];
        dst.ray_o[i] = src.ray_o[s]; dst.ray_d[i] = src.ray_d[s]; dst.hit[i] = src.hit[s]; dst.thr[i] = src.thr[s]; dst.li[i] = src.li[s];
        dst.rng[i] = src.rng[s]; dst.pix[i] = src.pix[s]; dst.sh_o[i] = src.sh_o[s]; dst.sh_d[i] = src.sh_d[s]; dst.sh_c[i] = src.sh_c[s];
        if (__float_as_uint(dst.li[i].w) & PF_PROBE) { dst.sv_o[i] = src.sv_o[s]; dst.sv_d[i] = src.sv_d[s]; dst.sv_h[i] = src.sv_h[s]; }
    } else {
        dst.li[i] = make_float4(0.f, 0.f, 0.f, __uint_as_float(0u));
        dst.ray_d[i] = make_float4(0.f, 0.f, 0.f, -1.f); dst.ray_o[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        dst.sh_d[i] = make_float4(0.f, 0.f, 0.f, -1.f); dst.sh_o[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// What k_shade and k_finish do with a slot before its new hit is looked at: add the NEE term of the previous bounce
// if its shadow ray was unoccluded (src/pathMisIntegrator.cpp:150-163; the sum order of Li is the reference's: the
// NEE term of bounce k enters before the emission term of bounce k+1).
__device__ __forceinline__ F3 add_pending_nee(F3 li, unsigned int flags, float4 c) {
    if ((flags & PF_SHADOW) && c.w != 0.f) li = li + f3(c.x, c.y, c.z);
    return li;
}

// Ray ordering.  After a few bounces the rays of neighbouring slots have nothing to do with each other, and on a scene whose
// tree does not fit the caches every lane of a wave then drags its own lines through L2.  Before the trace stage the slots are
// sorted by the cell of their path vertex (30-bit Morton code of the extend ray's origin in the root box, which is also where
// the slot's shadow ray ends), and the trace stage takes its batches through that order: the rays of a wave start (or end) in the
// same corner of the tree and share the nodes and records there.  Only the order of the work changes, no ray and no result.
__device__ __forceinline__ unsigned int spread10(unsigned int x) {
    x &= 0x3ffu; x = (x | (x << 16)) & 0x30000ffu; x = (x | (x << 8)) & 0x300f00fu; x = (x | (x << 4)) & 0x30c30c3u; x = (x | (x << 2)) & 0x9249249u;
    return x;
}
#define ND_NO_KEY 0x3fffffffu            /* a slot without a ray: sorts last */
__device__ __forceinline__ unsigned int cell_key(const DevScene &sc, float px, float py, float pz) {
    const float ex = sc.root_max[0] - sc.root_min[0], ey = sc.root_max[1] - sc.root_min[1], ez = sc.root_max[2] - sc.root_min[2];
    // (cells per unit length: loop-invariant, so the three divisions leave the shade stage's loop; the key only orders the work)
    const float sx = ex > 0.f ? 1024.f / ex : 0.f, sy = ey > 0.f ? 1024.f / ey : 0.f, sz = ez > 0.f ? 1024.f / ez : 0.f;
    const float qx = (px - sc.root_min[0]) * sx, qy = (py - sc.root_min[1]) * sy, qz = (pz - sc.root_min[2]) * sz;
    const unsigned int ix = (unsigned int) fminf(fmaxf(qx, 0.f), 1023.f), iy = (unsigned int) fminf(fmaxf(qy, 0.f), 1023.f), iz = (unsigned int) fminf(fmaxf(qz, 0.f), 1023.f);
    const unsigned int key = (spread10(ix) << 2) | (spread10(iy) << 1) | spread10(iz);
    return key == ND_NO_KEY ? ND_NO_KEY - 1u : key;
}
__global__ void k_iota(unsigned int *ids, unsigned int n) { const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) ids[i] = i; }
// (the shade stage writes the keys itself, from the rays it has in registers; this kernel is what it replaced and stays for NORI_HIP_KEYS=kernel)
__global__ __launch_bounds__(256) void k_ray_keys(DevScene sc, Pool pool, unsigned int n_slots, unsigned int *keys, unsigned int *ids) {
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_slots) return;
    const float4 o = pool.ray_o[i], d = pool.ray_d[i], sd = pool.sh_d[i], so = pool.sh_o[i];
    const bool ext = !(d.w < o.w), sh = !(sd.w < so.w);
    unsigned int key = 0x3fffffffu;                                       // no ray in the slot: last
    if (ext || sh) {
        // the path vertex: the extend ray's origin, or (the path ends here) the far end of the shadow ray
        const float px = ext ? o.x : so.x + sd.w * sd.x, py = ext ? o.y : so.y + sd.w * sd.y, pz = ext ? o.z : so.z + sd.w * sd.z;
        key = cell_key(sc, px, py, pz);
    }
    keys[i] = key; ids[i] = i;
}

// Shade stage.  Each block owns a contiguous run of slots:
//   pass A  shades (see above); a slot whose path goes on is appended to the block's keep list, a slot that is free
//           afterwards to its regenerate list (both in LDS: one array of 16-bit offsets into the run, the regenerate
//           list growing from the front, the keep list from the back; one LDS atomic pair per wave and 64 slots)
//   then ONE pair of global atomics per block reserves the block's camera samples and its room in the live list
//   pass B  walks the two lists densely: the keep list goes into the live list (used by the host for termination, by
//           k_compact_pool and by k_finish), the regenerate list starts new paths from the reserved work items
//           (src/render.cpp:83-98) with every lane busy, whatever share of the run was freed
// The host sizes the grid so that a run has at most SHADE_RUN_MAX slots.
#define SHADE_RUN_MAX 16384u
enum { SL_IDLE = 0x8000u };
// KEYS: the stage also writes the ray sort's key of every slot (the cell its path stands in after this bounce)
template <bool EXT, bool KEYS = false>
__device__ __forceinline__ void shade_body(const DevScene &sc, const Pool &pool, unsigned int n_slots, const RenderJob &job, float4 *film, unsigned int *q_live,
                                           WfCounters *cnt, unsigned long long *work_counter, DevStats *stats, unsigned int *keys) {
    extern __shared__ unsigned short run_list[];                            // per_block entries
    __shared__ unsigned int l_regen, l_keep;
    __shared__ unsigned long long b_work; __shared__ unsigned int b_live;
    const unsigned int lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const unsigned long long padded_total = job.padded_total;
    const unsigned int per_block = (((n_slots + gridDim.x - 1) / gridDim.x) + 255u) & ~255u;
    const unsigned int lo = blockIdx.x * per_block, hi = (lo + per_block < n_slots) ? lo + per_block : n_slots;
    unsigned int n_invalid = 0, n_degenerate = 0, n_culled = 0, n_dropped = 0;
    const float4 no_ray = make_float4(0.f, 0.f, 0.f, -1.f);
    if (threadIdx.x == 0) { l_regen = 0u; l_keep = 0u; }
    __syncthreads();
    // (the trip count is the block's, not the wave's: a wave whose 64 slots lie past the end of the run makes an empty trip.  Written this way
    // the EXT build of the stage allocates its registers better - cbox_subsurface 92.3 -> 89.5 ms per step on one box, the others unchanged)
    for (unsigned int ib = lo; ib < hi; ib += 256u) {
        const unsigned int slot = ib + threadIdx.x;
        bool regen = false, keep = false, idle = false;
        F3 kp = f3(0.f); bool kvalid = false;            // where the slot's path stands after this bounce (the ray sort's key), if it has a ray
        if (slot < hi) {
            const float4 L = pool.li[slot]; const uint4 h = pool.hit[slot]; const float4 P = pool.pix[slot];
            const float4 ro = pool.ray_o[slot], rd = pool.ray_d[slot], T = pool.thr[slot]; const uint4 R = pool.rng[slot];
            const float4 C = pool.sh_c[slot];
            unsigned int flags = __float_as_uint(L.w);
            regen = true; idle = true;
            if (flags & PF_ACTIVE) {
                PathState ps; ps.li = add_pending_nee(f3(L.x, L.y, L.z), flags, C);
                bool ended = true, new_shadow = false;
                idle = false;
                ShadeOut so; bool shaded = false;
                Pcg32 rng; rng.state = ((uint64_t) R.y << 32) | R.x; rng.inc = ((uint64_t) R.w << 32) | R.z;
                ps.t = f3(T.x, T.y, T.z); ps.last_bsdf_pdf = T.w; ps.last_delta = (flags & PF_LAST_DELTA) != 0;
                if (EXT && (flags & PF_PROBE)) {
                    // second half of a bounce on a subsurface shape: the probe drawn in the first half has been traced
                    const float4 so4 = pool.sv_o[slot], sd4 = pool.sv_d[slot]; const uint4 sh4 = pool.sv_h[slot];
                    Ray ray0; ray0.o = f3(so4.x, so4.y, so4.z); ray0.mint = so4.w; ray0.d = f3(sd4.x, sd4.y, sd4.z); ray0.maxt = sd4.w;
                    Hit hit0; hit0.pos = sh4.x; hit0.t = __uint_as_float(sh4.y); hit0.u = __uint_as_float(sh4.z); hit0.v = __uint_as_float(sh4.w);
                    ShadeCtx cx; shade_ctx<EXT>(sc, ray0, hit0, cx);
                    ps.ray = ray0;
                    if (h.x != 0xffffffffu) {
                        Ray probe; probe.o = f3(ro.x, ro.y, ro.z); probe.mint = ro.w; probe.d = f3(rd.x, rd.y, rd.z); probe.maxt = rd.w;
                        Hit ph; ph.pos = h.x; ph.t = __uint_as_float(h.y); ph.u = __uint_as_float(h.z); ph.v = __uint_as_float(h.w);
                        Surface sp; surface_from_hit<EXT>(sc, probe, ph, sp, false);
                        so.end = false; so.has_shadow = false; so.contrib = f3(0.f); so.culled = false;
                        shade_finish<EXT>(sc, ps, rng, cx, sp.p, sp.sh, so);
                        shaded = true;
                    } else {
                        // the probe left the shape: the reference draws another one (src/pathMisIntegrator.cpp:88-90), up to ND_PROBE_TRIES here
                        const unsigned int tries = ((flags >> PF_TRIES_SHIFT) & 0x1fffu) + 1u;
                        if (tries < ND_PROBE_TRIES) {
                            Ray probe; ss_probe_ray(sc, cx, rng, probe);
                            pool.ray_o[slot] = make_float4(probe.o.x, probe.o.y, probe.o.z, probe.mint);
                            pool.ray_d[slot] = make_float4(probe.d.x, probe.d.y, probe.d.z, probe.maxt);
                            if (KEYS) { kp = probe.o; kvalid = true; }
                            pool.hit[slot] = make_uint4(ND_PROBE_MARK, cx.sf.shape, 0u, 0u);
                            pool.rng[slot] = make_uint4((unsigned int) rng.state, (unsigned int) (rng.state >> 32), (unsigned int) rng.inc, (unsigned int) (rng.inc >> 32));
                            pool.li[slot] = make_float4(ps.li.x, ps.li.y, ps.li.z, __uint_as_float((flags & (PF_ACTIVE | PF_LAST_DELTA | PF_PROBE)) | (tries << PF_TRIES_SHIFT)));
                            ended = false;
                        } else n_dropped++;                                          // every probe left the shape: the path ends with what it has (NoriHipStats::ss_paths_dropped)
                    }
                } else if (EXT && !(flags & PF_ENDING) && h.x == 0xffffffffu) {
                    // the ray left the scene: the environment emitter, if there is one (the path ends either way)
                    ps.li = ps.li + env_miss<EXT>(sc, f3(rd.x, rd.y, rd.z), ps.t, ps.last_bsdf_pdf, ps.last_delta);
                } else if (!(flags & PF_ENDING) && h.x != 0xffffffffu) {
                    ps.ray.o = f3(ro.x, ro.y, ro.z); ps.ray.mint = ro.w; ps.ray.d = f3(rd.x, rd.y, rd.z); ps.ray.maxt = rd.w;
                    Hit hit; hit.pos = h.x; hit.t = __uint_as_float(h.y); hit.u = __uint_as_float(h.z); hit.v = __uint_as_float(h.w);
                    ShadeCtx cx; shade_ctx<EXT>(sc, ps.ray, hit, cx);
                    if (!shade_begin(sc, ps, rng, cx, so)) shaded = true;                       // roulette: so.end is set
                    else if (EXT && sc.shapes[cx.sf.shape].bsdf == NORI_HIP_BSDF_SUBSURFACE) {
                        // first half of a bounce on a subsurface shape (src/pathMisIntegrator.cpp:86-100): the probe goes through the
                        // trace stage like any other ray, restricted to this shape; the bounce is finished when its hit is back
                        Ray probe; ss_probe_ray(sc, cx, rng, probe);
                        pool.sv_o[slot] = ro; pool.sv_d[slot] = rd; pool.sv_h[slot] = h;
                        pool.ray_o[slot] = make_float4(probe.o.x, probe.o.y, probe.o.z, probe.mint);
                        pool.ray_d[slot] = make_float4(probe.d.x, probe.d.y, probe.d.z, probe.maxt);
                        if (KEYS) { kp = probe.o; kvalid = true; }
                        pool.hit[slot] = make_uint4(ND_PROBE_MARK, cx.sf.shape, 0u, 0u);
                        pool.thr[slot] = make_float4(ps.t.x, ps.t.y, ps.t.z, ps.last_bsdf_pdf);
                        pool.rng[slot] = make_uint4((unsigned int) rng.state, (unsigned int) (rng.state >> 32), (unsigned int) rng.inc, (unsigned int) (rng.inc >> 32));
                        pool.li[slot] = make_float4(ps.li.x, ps.li.y, ps.li.z, __uint_as_float(PF_ACTIVE | PF_PROBE | (flags & PF_LAST_DELTA)));
                        ended = false;
                    } else { shade_finish<EXT>(sc, ps, rng, cx, cx.sf.p, cx.sf.sh, so); shaded = true; }
                }
                if (shaded) {
                    if (so.culled) n_culled++;
                    flags = PF_ACTIVE | (ps.last_delta ? PF_LAST_DELTA : 0u);
                    if (so.has_shadow) {
                        flags |= PF_SHADOW;
                        // A light sample closer than 2e-3 gives a shadow ray with maxt < mint.  BVH::rayIntersect returns false for
                        // it without looking at the tree (src/bvh.cpp:414), i.e. unoccluded; in the pool maxt < mint marks "no ray",
                        // so the answer is written here and the trace stage never sees the ray (counted here for the statistics).
                        const bool degenerate = so.shadow.maxt < so.shadow.mint;
                        if (degenerate) n_degenerate++;
                        else {
                            new_shadow = true;
                            pool.sh_o[slot] = make_float4(so.shadow.o.x, so.shadow.o.y, so.shadow.o.z, so.shadow.mint);
                            pool.sh_d[slot] = make_float4(so.shadow.d.x, so.shadow.d.y, so.shadow.d.z, so.shadow.maxt);
                        }
                        pool.sh_c[slot] = make_float4(so.contrib.x, so.contrib.y, so.contrib.z, degenerate ? 1.0f : 0.f);
                    }
                    if (!so.end) {
                        ended = false;
                        pool.ray_o[slot] = make_float4(ps.ray.o.x, ps.ray.o.y, ps.ray.o.z, ps.ray.mint);
                        pool.ray_d[slot] = make_float4(ps.ray.d.x, ps.ray.d.y, ps.ray.d.z, ps.ray.maxt);
                        if (KEYS) { kp = ps.ray.o; kvalid = true; }
                        pool.thr[slot] = make_float4(ps.t.x, ps.t.y, ps.t.z, ps.last_bsdf_pdf);
                        pool.rng[slot] = make_uint4((unsigned int) rng.state, (unsigned int) (rng.state >> 32), (unsigned int) rng.inc, (unsigned int) (rng.inc >> 32));
                        pool.li[slot] = make_float4(ps.li.x, ps.li.y, ps.li.z, __uint_as_float(flags));
                    } else if (so.has_shadow) {
                        // the path ends, but its last NEE term is still pending: it stays one more iteration without a next ray
                        ended = false;
                        pool.ray_d[slot] = no_ray;
                        if (KEYS && new_shadow) { kp = so.shadow.o + so.shadow.d * so.shadow.maxt; kvalid = true; }      // the path ends here: the far end of its last shadow ray
                        pool.li[slot] = make_float4(ps.li.x, ps.li.y, ps.li.z, __uint_as_float(flags | PF_ENDING));
                    }
                }
                if (!new_shadow) pool.sh_d[slot] = no_ray;
                if (ended) {
                    unsigned int pxy = __float_as_uint(P.z);
                    if (!film_put(sc, film, f2(P.x, P.y), (int) (pxy & 0xffffu), (int) (pxy >> 16), ps.li)) n_invalid++;
                } else { regen = false; keep = true; }
            }
            if (KEYS) keys[slot] = (keep && kvalid) ? cell_key(sc, kp.x, kp.y, kp.z) : ND_NO_KEY;      // (a slot that starts a new path gets its key in pass B)
        }
        const unsigned long long m_r = __ballot(regen), m_k = __ballot(keep);
        unsigned int pr = 0, pk = 0;
        if (lane == 0) { if (m_r) pr = atomicAdd(&l_regen, (unsigned int) __popcll(m_r)); if (m_k) pk = atomicAdd(&l_keep, (unsigned int) __popcll(m_k)); }
        pr = __builtin_amdgcn_readfirstlane(pr); pk = __builtin_amdgcn_readfirstlane(pk);
        if (regen) run_list[pr + __builtin_amdgcn_mbcnt_hi((unsigned int) (m_r >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int) m_r, 0u))] = (unsigned short) ((slot - lo) | (idle ? SL_IDLE : 0u));
        if (keep) run_list[per_block - 1u - (pk + __builtin_amdgcn_mbcnt_hi((unsigned int) (m_k >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int) m_k, 0u)))] = (unsigned short) (slot - lo);
    }
    __syncthreads();
    const unsigned int t_regen = l_regen, t_keep = l_keep;
    if (threadIdx.x == 0) {
        // (a plain load first: once the work has run out, the blocks stop bumping the counter)
        unsigned long long bw = padded_total;
        if (t_regen && *(volatile unsigned long long *) work_counter < padded_total) bw = atomicAdd(work_counter, (unsigned long long) t_regen);
        unsigned int n_valid = (bw >= padded_total) ? 0u : (unsigned int) ((padded_total - bw < (unsigned long long) t_regen) ? padded_total - bw : (unsigned long long) t_regen);
        b_work = bw;
        b_live = (t_keep + n_valid) ? atomicAdd(&CN_LIVE(cnt), t_keep + n_valid) : 0u;
    }
    __syncthreads();
    const unsigned long long base_w = b_work; const unsigned int base_n = b_live;
    for (unsigned int k = threadIdx.x; k < t_keep; k += 256u) q_live[base_n + k] = lo + run_list[per_block - 1u - k];
    unsigned int n_samples = 0;
    for (unsigned int r = threadIdx.x; r < t_regen; r += 256u) {
        const unsigned int e = run_list[r], slot = lo + (e & 0x7fffu);
        const unsigned long long w = base_w + r;
        if (w < padded_total) {
            int px, py, sidx;
            if (work_to_sample(job, w, px, py, sidx)) {
                Pcg32 rng; rng.seed((uint64_t) py * (uint64_t) sc.cam.width + (uint64_t) px, (uint64_t) sidx);
                F2 j = rng.next2D();
                F2 pixel_sample = f2((float) px + j.x, (float) py + j.y);
                (void) rng.next2D();
                Ray r0; camera_ray(sc, pixel_sample, r0);
                pool.ray_o[slot] = make_float4(r0.o.x, r0.o.y, r0.o.z, r0.mint);
                pool.ray_d[slot] = make_float4(r0.d.x, r0.d.y, r0.d.z, r0.maxt);
                if (KEYS) keys[slot] = cell_key(sc, r0.o.x, r0.o.y, r0.o.z);
                pool.thr[slot] = make_float4(1.f, 1.f, 1.f, 1.f);
                pool.rng[slot] = make_uint4((unsigned int) rng.state, (unsigned int) (rng.state >> 32), (unsigned int) rng.inc, (unsigned int) (rng.inc >> 32));
                pool.pix[slot] = make_float4(pixel_sample.x, pixel_sample.y, __uint_as_float((unsigned int) px | ((unsigned int) py << 16)), 0.f);
                pool.li[slot] = make_float4(0.f, 0.f, 0.f, __uint_as_float(PF_ACTIVE | PF_LAST_DELTA));
                n_samples++;
            } else {
                // padding of an edge patch: the slot counts as live, idle, and draws again in the next iteration
                pool.ray_d[slot] = no_ray;
                pool.li[slot] = make_float4(0.f, 0.f, 0.f, __uint_as_float(0u));
            }
            q_live[base_n + t_keep + r] = slot;
        } else if (!(e & SL_IDLE)) {
            pool.ray_d[slot] = no_ray;                                            // no work left: the slot retires
            pool.li[slot] = make_float4(0.f, 0.f, 0.f, __uint_as_float(0u));
        }
    }
    unsigned long long v0 = n_samples, v1 = n_invalid, v2 = n_degenerate, v3 = n_culled, v4 = n_dropped;
    for (int off = 32; off > 0; off >>= 1) { v0 += __shfl_down(v0, off, 64); v1 += __shfl_down(v1, off, 64); v2 += __shfl_down(v2, off, 64); v3 += __shfl_down(v3, off, 64); if (EXT) v4 += __shfl_down(v4, off, 64); }
    if (lane == 0 && (v0 | v1 | v2 | v3 | v4)) { DevStats *st = my_stats(stats); atomicAdd(&st->samples, v0); atomicAdd(&st->invalid, v1); if (v2) atomicAdd(&st->shadow, v2); if (v3) atomicAdd(&st->culled, v3); if (v4) atomicAdd(&st->dropped, v4); }
}
// built for several register budgets (waves per SIMD); the stage is bound by the latency of its dependent fetches
#define ND_SHADE_KERNEL(W) \
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(W, W))) \
void k_shade_w##W(DevScene sc, Pool pool, unsigned int n_slots, RenderJob job, float4 *film, unsigned int *q_live, WfCounters *cnt, \
                  unsigned long long *work_counter, DevStats *stats, unsigned int *keys) { shade_body<false>(sc, pool, n_slots, job, film, q_live, cnt, work_counter, stats, keys); }
ND_SHADE_KERNEL(3)
ND_SHADE_KERNEL(4)
ND_SHADE_KERNEL(5)
ND_SHADE_KERNEL(6)
ND_SHADE_KERNEL(8)
// (three waves per SIMD: left to itself the compiler takes 191 registers for two since sampleDx's expf runs in double arithmetic)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_shade_ext(DevScene sc, Pool pool, unsigned int n_slots, RenderJob job, float4 *film, unsigned int *q_live,
                                                   WfCounters *cnt, unsigned long long *work_counter, DevStats *stats, unsigned int *keys) {
    shade_body<true>(sc, pool, n_slots, job, film, q_live, cnt, work_counter, stats, keys); }
// ... writing the ray sort's keys too (the default variants only)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_shade_w4_keys(DevScene sc, Pool pool, unsigned int n_slots, RenderJob job, float4 *film, unsigned int *q_live, WfCounters *cnt,
                     unsigned long long *work_counter, DevStats *stats, unsigned int *keys) { shade_body<false, true>(sc, pool, n_slots, job, film, q_live, cnt, work_counter, stats, keys); }
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_shade_ext_keys(DevScene sc, Pool pool, unsigned int n_slots, RenderJob job, float4 *film, unsigned int *q_live,
                                                        WfCounters *cnt, unsigned long long *work_counter, DevStats *stats, unsigned int *keys) {
    shade_body<true, true>(sc, pool, n_slots, job, film, q_live, cnt, work_counter, stats, keys); }

// Trace stage: persistent waves over the pool.  Wave w of W takes the 64-slot batches w, w+W, w+2W, ... (a static,
// interleaved share: no shared counter) for the first 85 % of the rounds; the rest comes from eight queues, see below.  Rays reach the lanes through a small pipeline, so that the HBM latency of
// the pool is paid once per batch and overlaps with traversal instead of stalling every trip:
//   load     one batch ahead, coalesced, into registers (the wait lands one or more trips later)
//   stage    when the batch has arrived: the start of every ray's traversal (Trav::begin: adaptive epsilon, reciprocal
//            direction, root box) with the whole batch busy; rays that do not enter the tree are answered on the spot,
//            the others are packed into the wave's LDS buffer
//   hand out a lane whose ray has finished takes the next staged ray (ballot + prefix count)
// Every trip of the loop the wave decides, from ballots of the lane states, which traversal step is worth issuing:
//   retire+refill  when at least `thr.z` lanes have finished (or nothing else is left to do)
//   box_step       when at least `thr.x` lanes stand at an inner node (or no lane stands at a leaf)
//   prim_step      (not SPEC) when at least `thr.y` lanes stand at a leaf record (or the box step did not run)
//   pop_step       for the lanes that need a stack entry
//   box_step       a second time when at least `box2_lanes` lanes stand at an inner node again
//   record step    (SPEC) for the parked leaves, see below
// The steps are straight-line code.  At least one step runs per trip, so the loop always makes progress; it ends
// when the wave's batches are used up and every lane has retired its ray.  Results leave through plain stores:
// the hit record (extend) or the visibility word sh_c.w (shadow).
#define TRACE_DEPTH 6
// KIND: TK_EXTEND / TK_SHADOW trace one kind of ray; TK_BOTH traces the shadow rays and then the extend rays of
// the pool in one launch (batches [0, NB) are shadow batches, [NB, 2 NB) extend batches; a lane carries the kind of
// its ray), so an iteration has one kernel tail instead of two.
enum { TK_EXTEND = 0, TK_SHADOW = 1, TK_BOTH = 2 };
// SPEC (speculative traversal): a lane that reaches a leaf parks it (`pend`) and goes on traversing; the record step runs
// for the parked leaves once thr.y lanes have one (default 1: every trip; or no lane can traverse, or the box step did
// not run), so a ray's record tests no longer hold up its box tests.  While a
// leaf is parked the lane's box tests use a maxt that its records may yet shrink, which only lets more boxes pass.  The
// reference tests a leaf's box when it gets to it, after every earlier leaf (src/bvh.cpp:404-463): a lane parks its next
// leaf only when the parked one is finished, and re-tests the entry distance against maxt at that moment - the same
// test at the same point of the depth-first order, so the same records are tested with the same maxt, in the same order.
// WIDE: the tree is walked through its 128-byte four-box nodes (DevScene::nodes4, Trav::box_step4): every fetch uses a whole
// HBM line and a ray's chain of dependent fetches is about half as long; up to three stack entries per visit, so the LDS part of
// the stack is deeper (TRACE_DEPTH_W).
#ifndef TRACE_DEPTH_W
#define TRACE_DEPTH_W 9      /* 20 waves per CU x (9 x 512 B of stack + 3328 B of staged rays) = 155 KB of the 160 KB */
#endif
#define TRACE_DEPTH_Q (TRACE_DEPTH_W * 4 / 3)   /* ... of the kernel over quantised four-box nodes: 6-byte entries (LdsStack6T), same LDS */
#ifndef TRACE_DEPTH_W8
#define TRACE_DEPTH_W8 8     /* ... of the eight-box trace stage (up to seven entries per visit) */
#endif
#ifndef ND_WIDE_WAVES
#define ND_WIDE_WAVES 5         /* waves per SIMD the four-box trace stage is built for */
#endif
// MASKED (scenes with subsurface shapes): an extend ray whose slot holds ND_PROBE_MARK is a probe, a closest-hit query restricted
// to the records of one shape (BVH::rayIntersect's mask, src/bvh.cpp:438).
// PRIM2: a lane at a leaf tests both records of the leaf in one trip (Trav::prim_step2; leaves of at most two records inside one line)
// QUANT (WIDE == 4): the four-box nodes are the 64-byte ones with quantised boxes (DevScene::nodes4q, Trav::box_step4q)
template <int KIND, bool PROF, bool SPEC, int WIDE, bool MASKED = false, bool PRIM2 = false, bool QUANT = false>
__device__ __forceinline__ void trace_pool_body(const DevScene &sc, const Pool &pool, unsigned int n_slots, DevStats *stats, int3 thr, unsigned int box2_lanes, WfCounters *cnt, const unsigned int *order) {
    constexpr int DEPTH = WIDE == 8 ? TRACE_DEPTH_W8 : WIDE ? (QUANT ? TRACE_DEPTH_Q : TRACE_DEPTH_W) : TRACE_DEPTH;
    typedef typename std::conditional<QUANT, LdsStack6T<DEPTH>, LdsStackT<DEPTH> >::type StackLds;
    static_assert(!QUANT || (StackLds::WAVE_BYTES == TRACE_DEPTH_W * 64 * 8), "the host sizes the LDS of the four-box trace kernels alike");
    // dynamic LDS of a workgroup of nw waves: nw stacks | nw x 64 staged ray origins | directions | reciprocal directions | slots
    extern __shared__ float4 dyn_lds[];
    const unsigned int nw = blockDim.x >> 6;
    float4 *st_o_all = dyn_lds + nw * (StackLds::WAVE_BYTES / 16);
    float4 *st_d_all = st_o_all + nw * 64, *st_r_all = st_d_all + nw * 64;
    unsigned int *st_slot_all = reinterpret_cast<unsigned int *>(st_r_all + nw * 64);
    float4 *const st_o = st_o_all + (threadIdx.x >> 6) * 64, *const st_d = st_d_all + (threadIdx.x >> 6) * 64, *const st_r = st_r_all + (threadIdx.x >> 6) * 64;
    unsigned int *const st_slot = st_slot_all + (threadIdx.x >> 6) * 64;
    const unsigned int lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    constexpr bool ANY = (KIND == TK_SHADOW);
    const unsigned int n_waves = (gridDim.x * blockDim.x) >> 6, nb_pool = (n_slots + 63u) >> 6;
    unsigned int n_batches = (KIND == TK_BOTH) ? 2u * nb_pool : nb_pool;
    unsigned int b_next = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;  // wave-uniform: next batch of the static share
    // XCD-aware deal (sorted rays only).  Workgroups b and b + 8 share an XCD and its 4 MiB L2 (observed placement, used for speed
    // only: any placement gives the same results).  dealt round-robin, the eight L2s would each see rays from the whole sorted range
    // and hold eight copies of the same nodes and records.  Instead the sorted order is cut into 8 x_k chunks, chunk c belongs to
    // the workgroups with b % 8 == c % 8, and those walk their chunks one after the other (a chunk's shadow batches, then its extend
    // batches, whose rays start where the shadow rays end): an L2 then serves one region of the tree at a time.  A batch number b is
    // a position in that deal: round b / n_waves of wave b % n_waves (static part), or entry d % per_q of queue d / per_q.
    const unsigned int x_k = (box2_lanes >> 20) & 0xffu;
    const bool xmap = KIND == TK_BOTH && order != nullptr && x_k != 0u && (gridDim.x & 7u) == 0u;
    const unsigned int x_wx = n_waves >> 3;                                                                    // waves per XCD
    // batches of one kind per chunk, a multiple of the granule X_G: inside a chunk X_G shadow batches and X_G extend batches alternate,
    // and which kind comes first alternates from chunk to chunk.  (A wave's positions are x_wx apart.  With a chunk laid out as all its
    // shadow batches and then all its extend batches, a stride that is a multiple of the chunk's length gave every wave one kind only,
    // and the two kinds cost differently: 16 waves per CU took 8.1 ms per launch where 12 took 7.3.)
    constexpr unsigned int X_G = 16u;
    const unsigned int x_cb = xmap ? (((nb_pool + 8u * x_k - 1u) / (8u * x_k) + X_G - 1u) / X_G) * X_G : 0u;
    const unsigned int x_w = (blockIdx.x >> 3) * (blockDim.x >> 6) + (threadIdx.x >> 6);                       // this wave among its XCD's
    if (xmap) n_batches = ((2u * x_k * x_cb + x_wx - 1u) / x_wx) * n_waves;
    // The last dyn_pct percent of the rounds are not dealt statically: their batches [n_static, n_batches) sit in eight
    // queues (one cursor each, on its own cache line) and a wave that has finished its share takes from its queue, then
    // from the others, so that the waves end together.
    const unsigned int dyn_pct = (box2_lanes >> 8) & 0xffu;
    // (a wave tries dyn_tries of the queues, its own first; every queue is some block's own when the grid has 8 blocks)
    const unsigned int dyn_tries = (box2_lanes >> 16) & 0xfu;
    const unsigned int n_static = (dyn_pct && gridDim.x >= 8u) ? ((n_batches / n_waves) * (100u - dyn_pct) / 100u) * n_waves : n_batches;
    const unsigned int per_q = (n_batches - n_static + 7u) >> 3;
    unsigned int dq = (n_static < n_batches) ? 8u - dyn_tries : 8u;   // wave-uniform: 8 - the queues this wave may still try (its own first)
    bool dyn_asked = false; unsigned int dyn_idx = 0u;                   // a cursor has been asked for; lane 0 holds the answer
    bool loading = false;                                                // wave-uniform: lo4/ld4 hold a batch in flight
    float4 lo4 = make_float4(0.f, 0.f, 0.f, 0.f), ld4 = make_float4(0.f, 0.f, 0.f, -1.f); unsigned int l_base = 0;
    unsigned int l_slot = 0;                                             // the slot of this lane's ray of the batch in flight
    unsigned int s_pos = 0, s_n = 0;                                     // wave-uniform: staged rays [s_pos, s_n) wait in LDS
    uint2 stack_spill[(WIDE == 8 ? ND_STACK8 : ND_STACK) - DEPTH];
    StackLds stk; stk.spill = stack_spill; stk.attach(dyn_lds, wv, lane);
    unsigned int n_boxes_lane = 0;
    unsigned int n_visits = 0;                                           // WIDE: node fetches of the wave (statistics; wave-uniform sum of ballots)                                       // WIDE: boxes of the nodes this lane visited
    // (an LDS copy of the top of the tree, CachedNodes, was measured here: no gain, and a branch per node visit)
    GlobalNodes nodes; nodes.nodes = sc.nodes;
    bool any = ANY;                                                      // this lane's ray is a shadow ray
    bool l_any = ANY;                                                    // wave-uniform: kind of the batch in flight
    Trav<ANY> tv; tv.cur = ND_DONE; tv.sp = 0; tv.found = false; tv.special = false; tv.hit.pos = 0xffffffffu; tv.hit.t = tv.hit.u = tv.hit.v = 0.f;
    bool have = false; unsigned int slot = 0;
    uint32_t mask = 0xffffffffu;                                         // MASKED: the shape this lane's ray is restricted to
    uint32_t pend = 0u;                                                  // SPEC: the parked leaf's next record (NORI_HIP_LEAF_BIT | position), 0 = none
    // statistics are wave-uniform sums of ballots (scalar registers): records and boxes tested, rays by kind
    unsigned int n_boxes = 0, n_prims = 0, n_rays = 0, n_rays_any = 0;
    uint32_t dummy = 0;
    unsigned int pf[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};               // PROF only: wave-uniform step statistics
    for (;;) {
        // (wave-uniform by construction; said explicitly so that they live in scalar registers and branch on SCC)
        s_pos = __builtin_amdgcn_readfirstlane(s_pos); s_n = __builtin_amdgcn_readfirstlane(s_n);
        b_next = __builtin_amdgcn_readfirstlane(b_next); l_base = __builtin_amdgcn_readfirstlane(l_base); dq = __builtin_amdgcn_readfirstlane(dq);
        // ---- feed the staging pipeline
        if (s_pos >= s_n && loading) {
            // The batch has arrived.  The start of a traversal (adaptive epsilon, reciprocal direction, root box:
            // Trav::begin) is done here, with all the batch's lanes busy, and not when a ray is handed to a lane, where a
            // dozen lanes would do it; rays that do not enter the tree are answered on the spot and never staged.
            const bool valid = !(ld4.w < lo4.w);
            Trav<ANY> tb; uint32_t root_tested = 0; bool enters = false;
            if (valid) {
                Ray r; r.o = f3(lo4.x, lo4.y, lo4.z); r.mint = lo4.w; r.d = f3(ld4.x, ld4.y, ld4.z); r.maxt = ld4.w;
                enters = tb.begin(sc, r, root_tested);
                if (!enters) {
                    if ((KIND == TK_BOTH) ? l_any : ANY) reinterpret_cast<float *>(pool.sh_c + l_slot)[3] = 1.0f;
                    else pool.hit[l_slot] = make_uint4(0xffffffffu, __float_as_uint(ND_INF), 0u, 0u);
                }
            }
            n_boxes += (unsigned int) __popcll(__ballot(root_tested != 0u));
            if ((KIND == TK_BOTH) ? l_any : false) n_rays_any += (unsigned int) __popcll(__ballot(valid)); else n_rays += (unsigned int) __popcll(__ballot(valid));
            const unsigned long long m = __ballot(enters);
            if (enters) {
                const unsigned int r = __builtin_amdgcn_mbcnt_hi((unsigned int) (m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int) m, 0u));
                st_o[r] = make_float4(lo4.x, lo4.y, lo4.z, tb.mint); st_d[r] = ld4; st_r[r] = make_float4(tb.rcp.x, tb.rcp.y, tb.rcp.z, 0.f);
                st_slot[r] = l_slot | ((KIND == TK_BOTH && l_any) ? 0x80000000u : 0u) | (tb.special ? 0x40000000u : 0u);
            }
            s_n = (unsigned int) __popcll(m); s_pos = 0; loading = false;
        }
        unsigned int b_take = 0xffffffffu;                                   // wave-uniform: the batch to load now
        if (!loading) {
            if (b_next < n_static) { b_take = b_next; b_next += n_waves; }
            else if (dyn_asked) {
                // the cursor asked for in the previous trip (its round trip overlapped with that trip's steps)
                const unsigned int qq = (blockIdx.x + dq + dyn_tries) & 7u, idx = __builtin_amdgcn_readfirstlane(dyn_idx);
                const unsigned int b = n_static + qq * per_q + idx;
                if (idx < per_q && b < n_batches) b_take = b; else dq++;
                dyn_asked = false;
            }
            if (b_take == 0xffffffffu && b_next >= n_static && dq < 8u) {
                if (lane == 0) dyn_idx = atomicAdd(&CN_DYN(cnt, (blockIdx.x + dq + dyn_tries) & 7u), 1u);
                dyn_asked = true;
            }
        }
        if (b_take != 0xffffffffu) {
            if (xmap) {
                unsigned int x, j;
                if (b_take < n_static) { x = blockIdx.x & 7u; j = (b_take / n_waves) * x_wx + x_w; }
                else { const unsigned int d = b_take - n_static; x = d / per_q; j = (n_static / n_waves) * x_wx + (d - x * per_q); }
                const unsigned int k = j / (2u * x_cb), within = j - k * 2u * x_cb, q = within / X_G;
                l_any = ((q + k) & 1u) == 0u;
                const unsigned int pb = (x + 8u * k) * x_cb + (q >> 1) * X_G + (within - q * X_G);
                if (k >= x_k || pb >= nb_pool) { l_any = false; l_base = nb_pool << 6; }      // a position past the end of the deal: an empty batch
                else l_base = pb << 6;
            } else {
            l_any = (KIND == TK_BOTH) ? (b_take < nb_pool) : ANY;
            l_base = ((KIND == TK_BOTH && !l_any) ? b_take - nb_pool : b_take) << 6;
            }
            const unsigned int pos = l_base + lane;
            if (pos < n_slots) {
                // (through the sorted order the slot number is one more dependent load; it is issued a batch ahead like the ray itself)
                const unsigned int sl = order ? order[pos] : pos;
                l_slot = sl;
                lo4 = l_any ? pool.sh_o[sl] : pool.ray_o[sl]; ld4 = l_any ? pool.sh_d[sl] : pool.ray_d[sl];
            } else { l_slot = 0u; lo4 = make_float4(0.f, 0.f, 0.f, 0.f); ld4 = make_float4(0.f, 0.f, 0.f, -1.f); }
            loading = true;
        }
        // ---- retire finished rays, hand out staged ones
        const unsigned long long fin = __ballot(tv.done() && (!SPEC || pend == 0u));   // finished or never started
        if (PROF) { pf[0]++; pf[9] += 64 - __popcll(fin); }
        if (fin && (__popcll(fin) >= thr.z || fin == ~0ull)) {
            if (PROF) { pf[7]++; pf[8] += __popcll(__ballot(have && tv.done() && (!SPEC || pend == 0u))); }
            if (have && tv.done() && (!SPEC || pend == 0u)) {
                if ((KIND == TK_BOTH) ? any : ANY) { if (!tv.found) reinterpret_cast<float *>(pool.sh_c + slot)[3] = 1.0f; }
                else pool.hit[slot] = make_uint4(tv.hit.pos, __float_as_uint(tv.hit.t), __float_as_uint(tv.hit.u), __float_as_uint(tv.hit.v));
                have = false;
            }
            if (s_pos < s_n) {
                const unsigned int idx = s_pos + __builtin_amdgcn_mbcnt_hi((unsigned int) (fin >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int) fin, 0u));
                if (tv.done() && (!SPEC || pend == 0u) && idx < s_n) {
                    const float4 ro = st_o[idx], rd = st_d[idx], rr = st_r[idx]; slot = st_slot[idx];
                    if (KIND == TK_BOTH) any = (slot >> 31) != 0u;
                    tv.special = (slot & 0x40000000u) != 0u; slot &= 0x3fffffffu;
                    if (MASKED) { mask = 0xffffffffu; if (!((KIND == TK_BOTH) ? any : ANY)) { const uint4 hin = pool.hit[slot]; if (hin.x == ND_PROBE_MARK) mask = hin.y; } }
                    tv.o = f3(ro.x, ro.y, ro.z); tv.mint = ro.w; tv.d = f3(rd.x, rd.y, rd.z); tv.maxt = rd.w; tv.rcp = f3(rr.x, rr.y, rr.z);
                    tv.sp = 0; tv.found = false; tv.cur = WIDE ? sc.root4 : sc.root; tv.cur_near = -ND_INF;       // (the root box was tested at staging; the root may be a leaf)
                    tv.hit.pos = 0xffffffffu; tv.hit.t = ND_INF; tv.hit.u = tv.hit.v = 0.f;
                    have = true;
                }
                s_pos += (unsigned int) __popcll(fin);
            }
            if (!loading && s_pos >= s_n && b_next >= n_static && dq >= 8u && !dyn_asked && __ballot(have) == 0ull) break;   // nothing staged, loading or in flight
        }
        // ---- traversal steps
        const unsigned long long m_in = __ballot(tv.at_inner());
        bool ran = false;
        if (m_in && (__popcll(m_in) >= thr.x || __ballot(tv.at_leaf()) == 0ull)) {
            if (PROF) { pf[1]++; pf[2] += __popcll(m_in); }
            // shadow rays take the nearer child first (any child order gives the same any-hit answer), extend rays
            // the reference's order unless sc.ordered
            const bool ord = (((KIND == TK_BOTH) ? any : ANY) && (sc.ordered & 2)) || (sc.ordered & 1);
            if (WIDE == 8) {
                n_visits += (unsigned int) __popcll(m_in);
                if (__ballot(tv.at_inner() && tv.special) == 0ull) { if (tv.at_inner()) n_boxes_lane += tv.box_step8_fast(sc.nodes8, stk); }
                else if (tv.at_inner()) n_boxes_lane += tv.box_step8(sc.nodes8, stk);
            } else if (WIDE) {
                n_visits += (unsigned int) __popcll(m_in);
                const bool sort = __ballot(tv.at_inner() && ord) != 0ull;
                const bool far = ((KIND == TK_BOTH) ? any : ANY) && (sc.ordered & 4);
                if (QUANT) {
                    if (__ballot(tv.at_inner() && tv.special) == 0ull) { if (tv.at_inner()) n_boxes_lane += tv.box_step4q_fast(sc.nodes4q, stk, ord, sort, far); }
                    else if (tv.at_inner()) n_boxes_lane += tv.box_step4q(sc.nodes4q, stk, ord, far);
                } else
                if (__ballot(tv.at_inner() && tv.special) == 0ull) { if (tv.at_inner()) n_boxes_lane += tv.box_step4_fast(sc.nodes4, stk, ord, sort, far); }
                else if (tv.at_inner()) n_boxes_lane += tv.box_step4(sc.nodes4, stk, ord, far);
            } else {
                n_boxes += 2u * (unsigned int) __popcll(m_in);
                if (__ballot(tv.at_inner() && tv.special) == 0ull) { if (tv.at_inner()) tv.box_step_fast(nodes, stk, dummy, ord); }
                else if (tv.at_inner()) tv.box_step(nodes, stk, dummy, ord);
            }
            ran = true;
        }
        if (!SPEC) {
            const unsigned long long m_lf = __ballot(tv.at_leaf());
            if (m_lf && (__popcll(m_lf) >= thr.y || !ran)) {
                if (PROF) { pf[3]++; pf[4] += __popcll(m_lf); }
                if (MASKED) {
                    bool tested = false;
                    if (tv.at_leaf()) tested = tv.prim_step_masked(sc, dummy, (KIND == TK_BOTH) ? any : ANY, mask);
                    n_prims += (unsigned int) __popcll(__ballot(tested));
                } else if (PRIM2) {
                    bool second = false;
                    if (tv.at_leaf()) second = tv.prim_step2(sc, dummy, (KIND == TK_BOTH) ? any : ANY);
                    n_prims += (unsigned int) __popcll(m_lf) + (unsigned int) __popcll(__ballot(second));
                } else {
                    n_prims += (unsigned int) __popcll(m_lf);
                    if (tv.at_leaf()) tv.prim_step(sc, dummy, (KIND == TK_BOTH) ? any : ANY);
                }
            }
        }
        if (PROF) { const unsigned long long m_pp = __ballot(tv.at_pop()); if (m_pp) { pf[5]++; pf[6] += __popcll(m_pp); } }
        if (tv.at_pop()) tv.pop_step(stk);
        // a second box step for the lanes that stand at an inner node again (popped one, or descended into one), when enough do
        if (box2_lanes & 0xffu) {
            const unsigned long long m_in2 = __ballot(tv.at_inner());
            if ((unsigned int) __popcll(m_in2) >= (box2_lanes & 0xffu)) {
                if (PROF) { pf[1]++; pf[2] += __popcll(m_in2); }
                const bool ord = (((KIND == TK_BOTH) ? any : ANY) && (sc.ordered & 2)) || (sc.ordered & 1);
                if (WIDE == 8) {
                    n_visits += (unsigned int) __popcll(m_in2);
                    if (__ballot(tv.at_inner() && tv.special) == 0ull) { if (tv.at_inner()) n_boxes_lane += tv.box_step8_fast(sc.nodes8, stk); }
                    else if (tv.at_inner()) n_boxes_lane += tv.box_step8(sc.nodes8, stk);
                } else if (WIDE) {
                    n_visits += (unsigned int) __popcll(m_in2);
                    const bool sort = __ballot(tv.at_inner() && ord) != 0ull;
                    const bool far = ((KIND == TK_BOTH) ? any : ANY) && (sc.ordered & 4);
                    if (QUANT) {
                        if (__ballot(tv.at_inner() && tv.special) == 0ull) { if (tv.at_inner()) n_boxes_lane += tv.box_step4q_fast(sc.nodes4q, stk, ord, sort, far); }
                        else if (tv.at_inner()) n_boxes_lane += tv.box_step4q(sc.nodes4q, stk, ord, far);
                    } else
                    if (__ballot(tv.at_inner() && tv.special) == 0ull) { if (tv.at_inner()) n_boxes_lane += tv.box_step4_fast(sc.nodes4, stk, ord, sort, far); }
                    else if (tv.at_inner()) n_boxes_lane += tv.box_step4(sc.nodes4, stk, ord, far);
                } else {
                    n_boxes += 2u * (unsigned int) __popcll(m_in2);
                    if (__ballot(tv.at_inner() && tv.special) == 0ull) { if (tv.at_inner()) tv.box_step_fast(nodes, stk, dummy, ord); }
                    else if (tv.at_inner()) tv.box_step(nodes, stk, dummy, ord);
                }
            }
        }
        if (SPEC) {
            // park the leaf the lane stands at, if the previous one is finished and its box still passes
            if (pend == 0u && tv.at_leaf()) { if (tv.cur_near <= tv.maxt) pend = tv.cur; tv.cur = ND_POP; }
            const unsigned long long m_pd = __ballot(pend != 0u);
            if (m_pd && (__popcll(m_pd) >= thr.y || !ran || __ballot(tv.at_inner() || tv.at_pop()) == 0ull)) {     // (!ran: every trip runs a step)
                if (PROF) { pf[3]++; pf[4] += __popcll(m_pd); }
                n_prims += (unsigned int) __popcll(m_pd);
                if (pend != 0u) {
                    const uint32_t where = tv.cur;
                    tv.cur = pend;
                    tv.prim_step(sc, dummy, (KIND == TK_BOTH) ? any : ANY);
                    if (tv.cur == ND_DONE) pend = 0u;                                  // any-hit ray found its occluder
                    else { pend = (tv.cur == ND_POP) ? 0u : tv.cur; tv.cur = where; }
                }
                if (pend == 0u && tv.at_leaf()) { if (tv.cur_near <= tv.maxt) pend = tv.cur; tv.cur = ND_POP; }
            }
        }
    }
    if (WIDE) { unsigned int x = n_boxes_lane; for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64); n_boxes += __builtin_amdgcn_readfirstlane(x); }
    const unsigned long long v[4] = {n_rays, n_boxes, n_prims, n_rays_any};
    if (lane == 0 && (v[0] | v[1] | v[3])) {
        DevStats *st = my_stats(stats);
        atomicAdd(ANY ? &st->shadow : &st->closest, v[0]); atomicAdd(&st->nodes, v[1]); atomicAdd(&st->prims, v[2]);
        if (KIND == TK_BOTH) atomicAdd(&st->shadow, v[3]);
        // the work of the kernel the roofline is quoted on (the extend stage, or the merged trace stage)
        if (!ANY) { atomicAdd(&st->ext_rays, v[0] + v[3]); atomicAdd(&st->ext_nodes, v[1]); atomicAdd(&st->ext_prims, v[2]); if (WIDE) atomicAdd(&st->ext_visits, (unsigned long long) n_visits); }
        if (PROF) for (int k = 0; k < 10; ++k) atomicAdd(&st->prof[ANY ? 1 : 0][k], (unsigned long long) pf[k]);
    }
}

template <int KIND, bool PROF, bool SPEC = false>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(6))) void k_trace_pool(DevScene sc, Pool pool, unsigned int n_slots, DevStats *stats, int3 thr, unsigned int box2_lanes, WfCounters *cnt) {
    trace_pool_body<KIND, PROF, SPEC, 0>(sc, pool, n_slots, stats, thr, box2_lanes, cnt, nullptr);
}
// ... for scenes with subsurface shapes (probe rays; no speculative traversal)
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(ND_WIDE_WAVES))) void k_trace_pool4_masked(DevScene sc, Pool pool, unsigned int n_slots, DevStats *stats, int3 thr, unsigned int box2_lanes, WfCounters *cnt, const unsigned int *order) {
    trace_pool_body<TK_BOTH, false, false, 4, true>(sc, pool, n_slots, stats, thr, box2_lanes, cnt, order);
}
// the merged trace stage over four-box nodes
template <bool PROF, bool SPEC>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(ND_WIDE_WAVES))) void k_trace_pool4(DevScene sc, Pool pool, unsigned int n_slots, DevStats *stats, int3 thr, unsigned int box2_lanes, WfCounters *cnt, const unsigned int *order) {
    trace_pool_body<TK_BOTH, PROF, SPEC, 4>(sc, pool, n_slots, stats, thr, box2_lanes, cnt, order);
}
// ... with both records of a leaf tested in one trip (device-built trees whose leaves are placed inside lines; no speculation)
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(ND_WIDE_WAVES))) void k_trace_pool4_leaf2(DevScene sc, Pool pool, unsigned int n_slots, DevStats *stats, int3 thr, unsigned int box2_lanes, WfCounters *cnt, const unsigned int *order) {
    trace_pool_body<TK_BOTH, false, false, 4, false, true>(sc, pool, n_slots, stats, thr, box2_lanes, cnt, order);
}
// ... and with the 64-byte quantised four-box nodes
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(ND_WIDE_WAVES))) void k_trace_pool4q_leaf2(DevScene sc, Pool pool, unsigned int n_slots, DevStats *stats, int3 thr, unsigned int box2_lanes, WfCounters *cnt, const unsigned int *order) {
    trace_pool_body<TK_BOTH, false, false, 4, false, true, true>(sc, pool, n_slots, stats, thr, box2_lanes, cnt, order);
}
// ... over quantised eight-box nodes (device-built trees that do not fit the caches)
#ifndef ND_WIDE8_WAVES
#define ND_WIDE8_WAVES 4
#endif
template <bool PROF, bool SPEC>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(ND_WIDE8_WAVES))) void k_trace_pool8(DevScene sc, Pool pool, unsigned int n_slots, DevStats *stats, int3 thr, unsigned int box2_lanes, WfCounters *cnt, const unsigned int *order) {
    trace_pool_body<TK_BOTH, PROF, SPEC, 8>(sc, pool, n_slots, stats, thr, box2_lanes, cnt, order);
}

// The two rays of a bounce - next ray and shadow ray - of the up to 64 paths a wave of the tail kernel holds, traced by the wave
// together.  Traced lane by lane (each lane its shadow ray, then its next ray) a bounce lasts as long as the wave's longest shadow
// traversal plus its longest closest-hit traversal, and most lanes idle most of that time (k_finish: 18 % lane utilisation on the
// 10 M-triangle scene).  Here the rays are staged in the wave's LDS, next rays first (they are the longer ones), and a lane that has
// finished a ray takes the next one of the list, whoever it belongs to, as the lanes of the trace stage do; results go back to the
// owners through LDS.  Every ray is traversed by one lane with the same steps in the same order as before, so hits, any-hit answers
// and the counts of box and record tests do not change.  Four-box nodes only.
#define TW_RAYS 128
template <bool QUANT, typename StackT>
__device__ __forceinline__ void trace_wave4(const DevScene &sc, StackT &stk, float4 *w_o, float4 *w_d, unsigned int *w_meta, uint4 *w_hit, unsigned int *w_occ,
                                            const bool want_sh, const Ray &sh, const bool want_ex, const Ray &ex, bool &occluded, bool &found, Hit &hit,
                                            uint32_t &n_nodes, uint32_t &n_prims) {
    const unsigned int lane = threadIdx.x & 63u;
    const unsigned long long m_ex = __ballot(want_ex), m_sh = __ballot(want_sh);
    const unsigned int n_ex = (unsigned int) __popcll(m_ex), n_total = n_ex + (unsigned int) __popcll(m_sh);
    if (want_ex) {
        const unsigned int r = __builtin_amdgcn_mbcnt_hi((unsigned int) (m_ex >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int) m_ex, 0u));
        w_o[r] = make_float4(ex.o.x, ex.o.y, ex.o.z, ex.mint); w_d[r] = make_float4(ex.d.x, ex.d.y, ex.d.z, ex.maxt); w_meta[r] = lane;
    }
    if (want_sh) {
        const unsigned int r = n_ex + __builtin_amdgcn_mbcnt_hi((unsigned int) (m_sh >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int) m_sh, 0u));
        w_o[r] = make_float4(sh.o.x, sh.o.y, sh.o.z, sh.mint); w_d[r] = make_float4(sh.d.x, sh.d.y, sh.d.z, sh.maxt); w_meta[r] = lane | 0x100u;
    }
    w_hit[lane] = make_uint4(0xffffffffu, 0u, 0u, 0u); w_occ[lane] = 0u;
    __builtin_amdgcn_wave_barrier();
    unsigned int next = 0;                                               // wave-uniform: rays [next, n_total) wait in LDS
    Trav<false> tv; tv.cur = ND_DONE; tv.sp = 0; tv.found = false; tv.special = false; tv.hit.pos = 0xffffffffu; tv.hit.t = tv.hit.u = tv.hit.v = 0.f;
    tv.o = tv.d = tv.rcp = f3(0.f); tv.mint = tv.maxt = 0.f; tv.cur_near = 0.f;
    bool have = false, any = false; unsigned int owner = 0;
    for (;;) {
        next = __builtin_amdgcn_readfirstlane(next);
        const unsigned long long fin = __ballot(tv.done());
        if (fin) {
            if (have && tv.done()) {
                if (any) { if (tv.found) w_occ[owner] = 1u; }
                else if (tv.found) w_hit[owner] = make_uint4(tv.hit.pos, __float_as_uint(tv.hit.t), __float_as_uint(tv.hit.u), __float_as_uint(tv.hit.v));
                have = false;
            }
            if (next < n_total) {
                const unsigned int idx = next + __builtin_amdgcn_mbcnt_hi((unsigned int) (fin >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int) fin, 0u));
                if (tv.done() && idx < n_total) {
                    const float4 o4 = w_o[idx], d4 = w_d[idx]; const unsigned int me = w_meta[idx];
                    owner = me & 0xffu; any = (me & 0x100u) != 0u;
                    Ray r; r.o = f3(o4.x, o4.y, o4.z); r.mint = o4.w; r.d = f3(d4.x, d4.y, d4.z); r.maxt = d4.w;
                    uint32_t nb = 0;
                    if (tv.begin(sc, r, nb)) tv.cur = sc.root4;
                    n_nodes += nb; have = true;
                }
                next += (unsigned int) __popcll(fin);
            } else if (fin == ~0ull) break;
        }
        const bool ord = (any && (sc.ordered & 2)) || (sc.ordered & 1), far = any && (sc.ordered & 4);
        const unsigned long long m_in = __ballot(tv.at_inner());
        if (m_in) {
            const bool sort = __ballot(tv.at_inner() && ord) != 0ull;
            if (QUANT) {
                if (__ballot(tv.at_inner() && tv.special) == 0ull) { if (tv.at_inner()) n_nodes += tv.box_step4q_fast(sc.nodes4q, stk, ord, sort, far); }
                else if (tv.at_inner()) { if (tv.special) n_nodes += tv.box_step4q(sc.nodes4q, stk, ord, far); else n_nodes += tv.box_step4q_fast(sc.nodes4q, stk, ord, true, far); }
            } else
            if (__ballot(tv.at_inner() && tv.special) == 0ull) { if (tv.at_inner()) n_nodes += tv.box_step4_fast(sc.nodes4, stk, ord, sort, far); }
            else if (tv.at_inner()) { if (tv.special) n_nodes += tv.box_step4(sc.nodes4, stk, ord, far); else n_nodes += tv.box_step4_fast(sc.nodes4, stk, ord, true, far); }
        }
        const unsigned long long m_lf = __ballot(tv.at_leaf());
        if (m_lf && (__popcll(m_lf) >= sc.prim_lanes || __ballot(tv.at_inner()) == 0ull)) {
            // (QUANT implies leaves of at most two records placed inside a line: both in one trip, Trav::prim_step2)
            if (QUANT) { if (tv.at_leaf()) (void) tv.prim_step2(sc, n_prims, any); }
            else if (tv.at_leaf()) tv.prim_step(sc, n_prims, any);
        }
        if (tv.at_pop()) tv.pop_step(stk);
    }
    __builtin_amdgcn_wave_barrier();
    occluded = w_occ[lane] != 0u;
    const uint4 h = w_hit[lane];
    found = h.x != 0xffffffffu;
    hit.pos = h.x; hit.t = __uint_as_float(h.y); hit.u = __uint_as_float(h.z); hit.v = __uint_as_float(h.w);
}

// Tail of a render call: once the work counter is exhausted the number of live paths decays geometrically, and a
// stage launch per bounce would be mostly launch latency.  Persistent waves run the surviving paths of the live
// list to their end, one bounce per loop trip (shade -> shadow -> extend in registers); wave w of W takes the
// 64-entry batches w, w+W, ... and a lane whose path has ended takes the next entry of the wave's batch.
// WAVES: waves per SIMD the kernel is built for.  With four-box nodes the fused bounce wants ≈ 170 registers: at four waves (128) it
// spills 78 of them, at three one.  Three is 1-2 % faster on the scenes that live in the caches (cbox 25.5 -> 24.9 ms, subsurface 92.0 ->
// 90.4), four on the 10 M-triangle scene (367 against 373 ms: there the fourth wave hides more HBM latency than the spills cost).
template <bool EXT, int WIDE = 0, int WAVES = 4, bool QUANT = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k_finish(DevScene sc, Pool pool_1, float4 *film, const unsigned int *q_live_1, const WfCounters *cnt, DevStats *stats,
                                                                                                        Pool pool_2, const unsigned int *q_live_2, const WfCounters *cnt_2) {
    // stack entries per lane in LDS (the staged rays of trace_wave4 want room too: four workgroups per CU, 40 KB each at most)
    constexpr int FIN_DEPTH = WIDE == 4 ? (QUANT ? 11 : 8) : LDS_DEPTH;
    typedef typename std::conditional<QUANT, LdsStack6T<FIN_DEPTH>, LdsStackT<FIN_DEPTH> >::type StackLds;
    __shared__ __align__(16) unsigned char lds_stack[4 * StackLds::WAVE_BYTES];
    // WIDE == 4: the rays of a bounce staged per wave, and their results (trace_wave4)
    __shared__ float4 tw_o[WIDE == 4 ? 4 * TW_RAYS : 1], tw_d[WIDE == 4 ? 4 * TW_RAYS : 1];
    __shared__ unsigned int tw_meta[WIDE == 4 ? 4 * TW_RAYS : 1], tw_occ[WIDE == 4 ? 4 * 64 : 1];
    __shared__ uint4 tw_hit[WIDE == 4 ? 4 * 64 : 1];
    // Two lists (cnt_2 != null): the tails of a render call's two pipelines (render_entry) run as one launch - the batches of the first
    // pipeline's live list, then those of the second's, each with its own pool.  Which list a batch belongs to is wave-uniform.
    const unsigned int n_1 = CN_LIVE(cnt), n_2 = cnt_2 ? CN_LIVE(cnt_2) : 0u;
    const unsigned int lane = threadIdx.x & 63u;
    const unsigned int n_waves = (gridDim.x * blockDim.x) >> 6, nb_1 = (n_1 + 63u) >> 6, n_batches = nb_1 + ((n_2 + 63u) >> 6);
    unsigned int batch = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;   // wave-uniform
    unsigned int cursor = 0, end = 0;
#define ND_FIN_BATCH() do { const bool s_ = batch >= nb_1; const unsigned int n_ = s_ ? n_2 : n_1; cursor = (s_ ? batch - nb_1 : batch) << 6; end = (cursor + 64u < n_) ? cursor + 64u : n_; } while (0)
    ND_FIN_BATCH();
    uint2 stack_spill[(WIDE == 8 ? ND_STACK8 : ND_STACK) - FIN_DEPTH];
    StackLds stk; stk.spill = stack_spill; stk.attach(lds_stack, threadIdx.x >> 6, lane);
    Counters ctr = {0, 0, 0, 0, 0, 0}; unsigned int n_invalid = 0;
    bool alive = false; unsigned int slot = 0;
    PathState ps; Pcg32 rng; Hit hit; float4 P = make_float4(0.f, 0.f, 0.f, 0.f);
    // EXT: a slot taken over between the two halves of a subsurface bounce (PF_PROBE): its first trip finishes that bounce
    bool resume = false; Ray probe0; Hit hit0; int tries0 = 0;
    probe0.o = probe0.d = f3(0.f); probe0.mint = probe0.maxt = 0.f; hit0.pos = 0u; hit0.t = hit0.u = hit0.v = 0.f;
    if (batch >= n_batches) return;
    for (;;) {
        const unsigned long long dead = __ballot(!alive);
        if (dead && batch < n_batches) {
            const unsigned int idx = cursor + __builtin_amdgcn_mbcnt_hi((unsigned int) (dead >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int) dead, 0u));
            const bool second = batch >= nb_1;                                // wave-uniform
            Pool pool; const unsigned int *q_live = second ? q_live_2 : q_live_1;
            pool.li = second ? pool_2.li : pool_1.li; pool.hit = second ? pool_2.hit : pool_1.hit; pool.pix = second ? pool_2.pix : pool_1.pix; pool.sh_c = second ? pool_2.sh_c : pool_1.sh_c;
            pool.ray_o = second ? pool_2.ray_o : pool_1.ray_o; pool.ray_d = second ? pool_2.ray_d : pool_1.ray_d; pool.thr = second ? pool_2.thr : pool_1.thr; pool.rng = second ? pool_2.rng : pool_1.rng;
            pool.sh_o = second ? pool_2.sh_o : pool_1.sh_o; pool.sh_d = second ? pool_2.sh_d : pool_1.sh_d;
            pool.sv_o = second ? pool_2.sv_o : pool_1.sv_o; pool.sv_d = second ? pool_2.sv_d : pool_1.sv_d; pool.sv_h = second ? pool_2.sv_h : pool_1.sv_h;
            if (!alive && idx < end) {
                slot = q_live[idx];
                const float4 L = pool.li[slot];
                const unsigned int flags = __float_as_uint(L.w);
                if (flags & PF_ACTIVE) {
                    const uint4 h = pool.hit[slot];
                    P = pool.pix[slot];
                    ps.li = add_pending_nee(f3(L.x, L.y, L.z), flags, pool.sh_c[slot]);
                    if (EXT && (flags & PF_PROBE)) {
                        const float4 ro = pool.ray_o[slot], rd = pool.ray_d[slot], T = pool.thr[slot]; const uint4 R = pool.rng[slot];
                        const float4 so4 = pool.sv_o[slot], sd4 = pool.sv_d[slot]; const uint4 sh4 = pool.sv_h[slot];
                        probe0.o = f3(ro.x, ro.y, ro.z); probe0.mint = ro.w; probe0.d = f3(rd.x, rd.y, rd.z); probe0.maxt = rd.w;
                        ps.ray.o = f3(so4.x, so4.y, so4.z); ps.ray.mint = so4.w; ps.ray.d = f3(sd4.x, sd4.y, sd4.z); ps.ray.maxt = sd4.w;
                        hit0.pos = sh4.x; hit0.t = __uint_as_float(sh4.y); hit0.u = __uint_as_float(sh4.z); hit0.v = __uint_as_float(sh4.w);
                        ps.t = f3(T.x, T.y, T.z); ps.last_bsdf_pdf = T.w; ps.last_delta = (flags & PF_LAST_DELTA) != 0;
                        rng.state = ((uint64_t) R.y << 32) | R.x; rng.inc = ((uint64_t) R.w << 32) | R.z;
                        hit.pos = h.x; hit.t = __uint_as_float(h.y); hit.u = __uint_as_float(h.z); hit.v = __uint_as_float(h.w);       // the probe's hit
                        tries0 = (int) ((flags >> PF_TRIES_SHIFT) & 0x1fffu);
                        resume = true; alive = true;
                    } else if (!(flags & PF_ENDING) && h.x != 0xffffffffu) {
                        const float4 ro = pool.ray_o[slot], rd = pool.ray_d[slot], T = pool.thr[slot]; const uint4 R = pool.rng[slot];
                        ps.ray.o = f3(ro.x, ro.y, ro.z); ps.ray.mint = ro.w; ps.ray.d = f3(rd.x, rd.y, rd.z); ps.ray.maxt = rd.w;
                        ps.t = f3(T.x, T.y, T.z); ps.last_bsdf_pdf = T.w; ps.last_delta = (flags & PF_LAST_DELTA) != 0;
                        rng.state = ((uint64_t) R.y << 32) | R.x; rng.inc = ((uint64_t) R.w << 32) | R.z;
                        hit.pos = h.x; hit.t = __uint_as_float(h.y); hit.u = __uint_as_float(h.z); hit.v = __uint_as_float(h.w);
                        resume = false; alive = true;
                    } else {
                        if (EXT && !(flags & PF_ENDING) && sc.env.texture) {       // the slot's last ray left the scene: the environment emitter
                            const float4 rd = pool.ray_d[slot], T = pool.thr[slot];
                            ps.li = ps.li + env_miss<EXT>(sc, f3(rd.x, rd.y, rd.z), f3(T.x, T.y, T.z), T.w, (flags & PF_LAST_DELTA) != 0);
                        }
                        const unsigned int pxy = __float_as_uint(P.z);
                        if (!film_put(sc, film, f2(P.x, P.y), (int) (pxy & 0xffffu), (int) (pxy >> 16), ps.li)) n_invalid++;
                    }
                    pool.li[slot] = make_float4(0.f, 0.f, 0.f, __uint_as_float(0u));
                    pool.ray_d[slot] = make_float4(0.f, 0.f, 0.f, -1.f); pool.sh_d[slot] = make_float4(0.f, 0.f, 0.f, -1.f);
                }
            }
            cursor += (unsigned int) __popcll(dead);
            if (cursor >= end) { batch += n_waves; ND_FIN_BATCH(); }
        }
#undef ND_FIN_BATCH
        if (__ballot(alive) == 0ull) { if (batch >= n_batches) break; else continue; }
        if (WIDE == 4) {
            // one bounce: the hit is in registers; the wave traces its shadow rays and next rays together
            ShadeOut so; so.end = true; so.has_shadow = false; so.culled = false; so.contrib = f3(0.f);
            so.shadow.o = so.shadow.d = f3(0.f); so.shadow.mint = 0.f; so.shadow.maxt = -1.f;
            if (alive) {
                if (EXT && resume) { const Ray ray0 = ps.ray; shade_resume<EXT>(sc, ps, rng, ray0, hit0, probe0, hit, tries0, so, &ctr); resume = false; }
                else shade_hit<EXT>(sc, ps, rng, hit, so, &ctr);
                if (so.culled) ctr.culled++;
                if (so.has_shadow) ctr.shadow++;
                if (!so.end) ctr.closest++;
            }
            const bool want_sh = alive && so.has_shadow, want_ex = alive && !so.end;
            bool occluded = false, found = false; Hit nh;
            const unsigned int wv_ = threadIdx.x >> 6;
            trace_wave4<QUANT>(sc, stk, tw_o + wv_ * TW_RAYS, tw_d + wv_ * TW_RAYS, tw_meta + wv_ * TW_RAYS, tw_hit + wv_ * 64, tw_occ + wv_ * 64,
                        want_sh, so.shadow, want_ex, ps.ray, occluded, found, nh, ctr.nodes, ctr.prims);
            if (alive) {
                if (want_sh && !occluded) ps.li = ps.li + so.contrib;
                bool go_on = want_ex && found;
                if (go_on) hit = nh;
                else if (want_ex) ps.li = ps.li + env_miss<EXT>(sc, ps.ray.d, ps.t, ps.last_bsdf_pdf, ps.last_delta);
                if (!go_on) {
                    const unsigned int pxy = __float_as_uint(P.z);
                    if (!film_put(sc, film, f2(P.x, P.y), (int) (pxy & 0xffffu), (int) (pxy >> 16), ps.li)) n_invalid++;
                    alive = false;
                }
            }
        } else if (alive) {
            // one bounce: the hit is in registers
            ShadeOut so;
            if (EXT && resume) { const Ray ray0 = ps.ray; shade_resume<EXT>(sc, ps, rng, ray0, hit0, probe0, hit, tries0, so, &ctr); resume = false; }
            else shade_hit<EXT>(sc, ps, rng, hit, so, &ctr);
            if (so.culled) ctr.culled++;
            if (so.has_shadow) {
                Hit sh; ctr.shadow++;
                if (!(WIDE == 8 ? trace8<true>(sc, so.shadow, sh, stk, ctr.nodes, ctr.prims) : WIDE ? trace4<true>(sc, so.shadow, sh, stk, ctr.nodes, ctr.prims) : trace<true>(sc, so.shadow, sh, stk, ctr.nodes, ctr.prims))) ps.li = ps.li + so.contrib;
            }
            bool go_on = !so.end;
            if (go_on) {
                ctr.closest++; go_on = WIDE == 8 ? trace8<false>(sc, ps.ray, hit, stk, ctr.nodes, ctr.prims) : WIDE ? trace4<false>(sc, ps.ray, hit, stk, ctr.nodes, ctr.prims) : trace<false>(sc, ps.ray, hit, stk, ctr.nodes, ctr.prims);
                if (!go_on) ps.li = ps.li + env_miss<EXT>(sc, ps.ray.d, ps.t, ps.last_bsdf_pdf, ps.last_delta);
            }
            if (!go_on) {
                const unsigned int pxy = __float_as_uint(P.z);
                if (!film_put(sc, film, f2(P.x, P.y), (int) (pxy & 0xffffu), (int) (pxy >> 16), ps.li)) n_invalid++;
                alive = false;
            }
        }
    }
    unsigned long long v[7] = {ctr.closest, ctr.shadow, n_invalid, ctr.nodes, ctr.prims, ctr.culled, ctr.dropped};
    for (int k = 0; k < 7; ++k) { unsigned long long x = v[k]; for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64); v[k] = x; }
    if (lane == 0 && (v[0] | v[1] | v[2] | v[5] | v[6])) {
        DevStats *st = my_stats(stats);
        atomicAdd(&st->closest, v[0]); atomicAdd(&st->shadow, v[1]); atomicAdd(&st->invalid, v[2]); atomicAdd(&st->nodes, v[3]); atomicAdd(&st->prims, v[4]); if (v[5]) atomicAdd(&st->culled, v[5]);
        if (v[6]) atomicAdd(&st->dropped, v[6]);
    }
}

// ---------------------------------------------------------------------------------------------------------
__global__ void k_bitmap(const float4 *film, int w, int h, int border, int cols, float *rgb) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= h) return;
    float4 c = film[(size_t) (y + border) * cols + (x + border)];
    float *o = rgb + ((size_t) y * w + x) * 3;
    if (c.w != 0) { o[0] = c.x / c.w; o[1] = c.y / c.w; o[2] = c.z / c.w; } else { o[0] = o[1] = o[2] = 0.f; }
}

// Color3f::toSRGB (src/common.cpp:192-206)
__device__ __forceinline__ float to_srgb(float v) { return v <= 0.0031308f ? 12.92f * v : (1.0f + 0.055f) * powf(v, 1.0f / 2.4f) - 0.055f; }
__global__ void k_srgb(const float *rgb, size_t n, float *out) {
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = to_srgb(rgb[i]);
}

// NL-means (cuda/filter.cc.cu:53-110 + the normalisation of copy_weighted_data_to_surface :33-48).
// One output pixel per lane; the (2r+2f+1)^2 neighbourhood of a 16x16 pixel block is staged in LDS.
#define NLM_BW 16
#define NLM_BH 16
__global__ __launch_bounds__(NLM_BW * NLM_BH) void k_nlmeans(const float *img, float *out, int w, int h, int r, int f, float sigma, float hh) {
    extern __shared__ float lds[];          // (NLM_BH + 2R) x (NLM_BW + 2R) x 3, R = r + f; then the (2f+1)^2 patch weights
    const int R = r + f;
    const int tw = NLM_BW + 2 * R, th = NLM_BH + 2 * R;
    // the patch weight exp(-(k^2+l^2)/(2 sigma^2)) (cuda/filter.cc.cu:77) depends on the tap only: one table per block
    float *wtab = lds + 3 * tw * th;
    const int pw = 2 * f + 1;
    for (int i = threadIdx.y * NLM_BW + threadIdx.x; i < pw * pw; i += NLM_BW * NLM_BH) {
        int k = i / pw - f, l = i % pw - f;
        wtab[i] = expf(-(float) (k * k + l * l) / (2.0f * sigma * sigma));
    }
    const int bx = blockIdx.x * NLM_BW, by = blockIdx.y * NLM_BH;
    for (int i = threadIdx.y * NLM_BW + threadIdx.x; i < tw * th; i += NLM_BW * NLM_BH) {
        int ly = i / tw, lx = i % tw;
        int gx = bx + lx - R, gy = by + ly - R;
        float a = 0.f, b = 0.f, c = 0.f;
        if (gx >= 0 && gx < w && gy >= 0 && gy < h) { const float *p = img + ((size_t) gy * w + gx) * 3; a = p[0]; b = p[1]; c = p[2]; }
        lds[3 * i] = a; lds[3 * i + 1] = b; lds[3 * i + 2] = c;
    }
    __syncthreads();
    const int x = bx + threadIdx.x, y = by + threadIdx.y;
    if (x >= w || y >= h) return;
    const float h2 = hh * hh;
    const int minX = max(0, x - r), maxX = min(w - 1, x + r), minY = max(0, y - r), maxY = min(h - 1, y + r);
    float wsum = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int i = minX; i <= maxX; ++i) for (int j = minY; j <= maxY; ++j) {
        float d2 = 0.f;
        int pminx = max(max(-f, -x), -i), pmaxx = min(min(f, w - x - 1), w - i - 1);
        int pminy = max(max(-f, -y), -j), pmaxy = min(min(f, h - y - 1), h - j - 1);
        for (int k = pminx; k <= pmaxx; ++k) for (int l = pminy; l <= pmaxy; ++l) {
            const float *p = &lds[3 * ((y + l - by + R) * tw + (x + k - bx + R))];
            const float *q = &lds[3 * ((j + l - by + R) * tw + (i + k - bx + R))];
            float wk = wtab[(k + f) * pw + (l + f)];
            float dx = q[0] - p[0], dy = q[1] - p[1], dz = q[2] - p[2];
            float patchv = (dx * dx + dy * dy + dz * dz) / h2;
            d2 += wk * patchv;
        }
        d2 /= (float) ((pmaxx - pminx + 1) * (pmaxy - pminy + 1));
        float wgt = expf(-fmaxf(d2, 0.0f));
        wsum += wgt;
        const float *c = &lds[3 * ((j - by + R) * tw + (i - bx + R))];
        a0 += wgt * c[0]; a1 += wgt * c[1]; a2 += wgt * c[2];
    }
    float *o = out + ((size_t) y * w + x) * 3;
    o[0] = a0 / wsum; o[1] = a1 / wsum; o[2] = a2 / wsum;
}

// The same filter with the patch distance evaluated separably.  The tap weight factors, exp(-(k^2+l^2)/(2 sigma^2)) =
// g(k) g(l), and the clipped patch ranges of the reference (both pixels inside the image) are the same as summing a
// squared-difference image that is zero where either pixel is outside.  Per window offset the block builds that
// image for its tile (+f border), filters it horizontally, and every lane finishes its pixel with a vertical pass:
// (2f+1) + (2f+1) taps per pixel and offset instead of (2f+1)^2.  Sums are grouped differently from the direct
// kernel, so results agree to rounding (tests: rtol 2e-4), not bit for bit; NORI_HIP_FILTER=direct selects k_nlmeans.
__global__ __launch_bounds__(NLM_BW * NLM_BH) void k_nlmeans_sep(const float *img, float *out, int w, int h, int r, int f, float sigma, float hh) {
    extern __shared__ float lds[];
    const int R = r + f;
    const int tw = NLM_BW + 2 * R, th = NLM_BH + 2 * R;          // image tile
    const int dw = NLM_BW + 2 * f, dh = NLM_BH + 2 * f;          // squared-difference tile
    float *tile = lds;                                           // th x tw x 3
    float *D = tile + 3 * tw * th;                               // dh x dw
    float *Hb = D + dw * dh;                                     // dh x NLM_BW
    float *g = Hb + dh * NLM_BW;                                 // 2f+1
    const int bx = blockIdx.x * NLM_BW, by = blockIdx.y * NLM_BH;
    const int tid = threadIdx.y * NLM_BW + threadIdx.x, nt = NLM_BW * NLM_BH;
    for (int i = tid; i < tw * th; i += nt) {
        int ly = i / tw, lx = i % tw;
        int gx = bx + lx - R, gy = by + ly - R;
        float a = 0.f, b = 0.f, c = 0.f;
        if (gx >= 0 && gx < w && gy >= 0 && gy < h) { const float *p = img + ((size_t) gy * w + gx) * 3; a = p[0]; b = p[1]; c = p[2]; }
        tile[3 * i] = a; tile[3 * i + 1] = b; tile[3 * i + 2] = c;
    }
    for (int i = tid; i <= 2 * f; i += nt) { int k = i - f; g[i] = expf(-(float) (k * k) / (2.0f * sigma * sigma)); }
    const int x = bx + threadIdx.x, y = by + threadIdx.y;
    const bool inside = x < w && y < h;
    const float h2 = hh * hh;
    float wsum = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int di = -r; di <= r; ++di) for (int dj = -r; dj <= r; ++dj) {
        __syncthreads();                                         // tile/g ready; previous offset's Hb consumed
        for (int i = tid; i < dw * dh; i += nt) {
            int v = i / dw, u = i % dw;
            int px = bx - f + u, py = by - f + v, qx = px + di, qy = py + dj;
            float d = 0.f;
            if (px >= 0 && px < w && py >= 0 && py < h && qx >= 0 && qx < w && qy >= 0 && qy < h) {
                const float *p = &tile[3 * ((py - by + R) * tw + (px - bx + R))];
                const float *q = &tile[3 * ((qy - by + R) * tw + (qx - bx + R))];
                float dx = q[0] - p[0], dy = q[1] - p[1], dz = q[2] - p[2];
                d = (dx * dx + dy * dy + dz * dz) / h2;
            }
            D[i] = d;
        }
        __syncthreads();
        for (int i = tid; i < NLM_BW * dh; i += nt) {
            int v = i / NLM_BW, xx = i % NLM_BW;
            float acc = 0.f;
            for (int k = 0; k <= 2 * f; ++k) acc += g[k] * D[v * dw + xx + k];
            Hb[i] = acc;
        }
        __syncthreads();
        const int qx = x + di, qy = y + dj;
        if (inside && qx >= 0 && qx < w && qy >= 0 && qy < h) {
            float d2 = 0.f;
            for (int l = 0; l <= 2 * f; ++l) d2 += g[l] * Hb[(threadIdx.y + l) * NLM_BW + threadIdx.x];
            int pminx = max(max(-f, -x), -qx), pmaxx = min(min(f, w - x - 1), w - qx - 1);
            int pminy = max(max(-f, -y), -qy), pmaxy = min(min(f, h - y - 1), h - qy - 1);
            d2 /= (float) ((pmaxx - pminx + 1) * (pmaxy - pminy + 1));
            float wgt = expf(-fmaxf(d2, 0.0f));
            wsum += wgt;
            const float *c = &tile[3 * ((qy - by + R) * tw + (qx - bx + R))];
            a0 += wgt * c[0]; a1 += wgt * c[1]; a2 += wgt * c[2];
        }
    }
    if (inside) { float *o = out + ((size_t) y * w + x) * 3; o[0] = a0 / wsum; o[1] = a1 / wsum; o[2] = a2 / wsum; }
}

// ImageBlock::toBitmap followed by Color3f::toSRGB in one pass: what k_bitmap + k_srgb leave in their second buffer
__global__ void k_bitmap_srgb(const float4 *film, int w, int h, int border, int cols, float *srgb) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= h) return;
    float4 c = film[(size_t) (y + border) * cols + (x + border)];
    float *o = srgb + ((size_t) y * w + x) * 3;
    float r = 0.f, g = 0.f, b = 0.f;
    if (c.w != 0) { r = c.x / c.w; g = c.y / c.w; b = c.z / c.w; }
    o[0] = to_srgb(r); o[1] = to_srgb(g); o[2] = to_srgb(b);
}

template <typename T> int upload(T *&dptr, const T *src, size_t n) {
    dptr = nullptr;
    if (n == 0) return NORI_HIP_OK;
    HIPCHK(hipMalloc((void **) &dptr, n * sizeof(T)));
    HIPCHK(hipMemcpy(dptr, src, n * sizeof(T), hipMemcpyHostToDevice));
    return NORI_HIP_OK;
}

} // namespace

// Scratch of nori_hip_filter_noalloc: the sRGB-encoded bitmap the NL-means kernel reads.  One copy per GPU, held until the
// process ends; its device address is taken per device in nori_hip_create (hipGetSymbolAddress) and kept in the context.
static __device__ __align__(256) float scratch[(size_t) NORI_HIP_NOALLOC_MAX_PIXELS * 3];

struct NoriHipContext {
    int device = 0;
    hipStream_t stream = nullptr;
    bool have_scene = false;
    DevScene sc;
    uint32_t n_prims = 0, n_nodes = 0;
    float4 *d_nodes4 = nullptr; uint32_t n_nodes4 = 0;     // the tree as 128-byte four-box nodes (null: one-leaf tree, or too deep for the stack)
    uint4 *d_nodes4q = nullptr;                            // device-built trees beyond the caches: the four-box nodes with quantised boxes, 64 bytes each (n_nodes4 of them)
    int quant4 = -1;                                       // NORI_HIP_QUANT: 1 / 0 = build and walk them or not; -1 = measured default (see upload_scene_impl)
    uint4 *d_nodes8 = nullptr; uint32_t n_nodes8 = 0;      // device-built trees beyond the caches: quantised eight-box nodes (null otherwise)
    bool wide = true;                 // NORI_HIP_WIDE=0: the trace stage walks the 64-byte two-box nodes
    int wide8 = 0;                    // NORI_HIP_WIDE=8: device-built trees are walked through quantised eight-box nodes (measured on the 10 M-triangle workload:
                                      // a third fewer node fetches, but 2.5 x the instructions per visit; slower than four-box nodes, so an option, not a default)
    uint64_t trace_width = 0;         // boxes per node in the trace stage of the last render (NoriHipStats::extend_width)
    uint64_t trace_node_bytes = 0;    // ... and the size of that node (NoriHipStats::extend_node_bytes)
    size_t trace_lds_wide8 = 0;
    size_t trace_lds_wide = 0;
    NoriHipNode *d_nodes = nullptr; NoriHipPrim *d_prims = nullptr; NoriHipShadePrim *d_shade = nullptr, *d_light = nullptr;
    NoriHipShape *d_shapes = nullptr; NoriHipEmitter *d_emitters = nullptr; float *d_cdf = nullptr;
    NoriHipTexture *d_textures = nullptr; uint8_t *d_tex_data = nullptr; uint32_t n_textures = 0;
    float *d_env_cdf = nullptr;       // environment emitter: CDF over its (u, v) cells
    float *d_filt = nullptr; float filt_radius = 0.f;   // nori_hip_set_film_filter (kept across scene uploads)
    bool ext = false;                 // the scene has images or subsurface shapes: the EXT kernel variants run
    bool has_ss = false;              // ... subsurface shapes: the wavefront pipeline traces their probe rays in its trace stage
    int integrator = NORI_HIP_INTEGRATOR_PATH_MIS;   // nori_hip_set_integrator
    lbvh::Result lbvh;                // what nori_hip_upload_scene_build built (zeros after nori_hip_upload_scene)
    uint32_t lbvh_leaf = 0;           // records per leaf of a device-built tree (NORI_HIP_LBVH_LEAF); 0 = 3, or 2 from a million records on (measured)
    bool lbvh_plain = false;          // NORI_HIP_LBVH_MORTON=plain: x, y, z interleaved bit by bit instead of the planned bit order
    float lbvh_big = 0.25f;           // records longer than this fraction of the scene get their own subtree (NORI_HIP_LBVH_BIG, 0 = off)
    float4 *d_film = nullptr; int film_rows = 0, film_cols = 0, border = 0;
    unsigned long long *d_counter = nullptr; DevStats *d_stats = nullptr;
    int *d_tiles = nullptr;           // nori_hip_render_tiles: ND_MAX_TILES x (x0, y0, w, h), then ND_MAX_TILES + 1 patch offsets
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    NoriHipStats stats;
    int n_cus = 256;
    size_t lds_bytes = 65536;         // LDS a workgroup may use (sharedMemPerBlockOptin)
    // wavefront pipeline
    unsigned int pool_slots = 0; Pool pool, pool_b; void *pool_mem = nullptr;     // pool_b: half the slots, for the compaction of the tail
    unsigned int pool_alloc = 0;      // slots the allocated pool holds
    unsigned int *d_sort = nullptr; size_t sort_temp_bytes = 0;   // keys | keys' | slot numbers | slot numbers' (pool_alloc each) | sort workspace
    bool fused_keys = true;           // NORI_HIP_KEYS=kernel: the sort keys come from a kernel of their own (k_ray_keys) instead of from the shade stage
    int sort_bits = 16;               // NORI_HIP_SORT_BITS: most significant bits of the 30-bit cell key the ray sort looks at (8 .. 30)
    int sort_rays = -1;               // NORI_HIP_SORT: 1 / 0 = order the rays before every trace launch or never; -1 = when the tree does not fit the caches
    bool compact = true;              // NORI_HIP_COMPACT=0: never pack the pool (the tail then sweeps dead slots until k_finish takes over)
    unsigned int *d_qlive = nullptr; WfCounters *d_cnt = nullptr;
    unsigned int *h_ring = nullptr; hipEvent_t ring_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<hipEvent_t> ext_ev;   // pairs of events around the extend launches since the last reset_stats
    size_t ext_used = 0; double ext_ms_acc = 0.0; uint64_t ext_launches = 0;
    int pipeline = 1;                 // 1 = wavefront stages, 0 = single persistent kernel (k_render)
    bool pipeline_forced = false;     // NORI_HIP_PIPELINE was given; otherwise EXT scenes take the single kernel (measured faster: the
                                      // probe rays and image lookups of their shading are per-lane work the stages cannot batch)
    int finish_waves = 0;             // NORI_HIP_FINISH_WAVES: 3 or 4 = which build of the tail kernel runs (0: three on scenes that live in the caches, four beyond)
    unsigned int finish_below = 0;    // live paths below which k_finish takes over; 0 = a quarter of the pool, 5/16 for trees beyond the caches (NORI_HIP_FINISH)
    int shade_waves = 4;              // k_shade variant: waves per SIMD (NORI_HIP_SHADE_WAVES = 3, 4, 5, 6 or 8)
    int shade_blocks_per_cu = 12;     // k_shade: blocks per CU (each block owns one contiguous run of the queue)
    int trace_blocks_per_cu = 0;      // persistent trace waves: workgroups per CU (all resident at once); 0 = ask the occupancy query
    int mega_waves = 4;               // k_render variant: waves per SIMD (NORI_HIP_MEGA_WAVES = 3, 4, 5, 6 or 8)
    bool cull_zero = true;            // NORI_HIP_CULL=0: every emitter sample gets its shadow ray, as in the reference, also when its term is exactly zero
    int align_leaves = -1;            // NORI_HIP_LBVH_ALIGN: 1 / 0 = device-built trees with leaves of at most two records place them so that no leaf straddles a
                                      // 128-byte line (two of every eight record positions stay empty) or in plain sorted order; -1 = from a million records on
    int any_far = -1;                 // NORI_HIP_ANYORDER=far / near: any-hit rays enter first the child whose box they reach last / first (four-box nodes; the answer
                                      // is the same for every order).  A shadow ray runs from the light to the shaded point (src/arealight.cpp:77-82), and on a scene
                                      // whose occluders crowd around the shaded points the far end finds one sooner: 10 M-triangle workload 395.7 -> 383.9 ms, cbox
                                      // 26.7 -> 28.1, table / dragon / subsurface within 1 %.  -1: far when the tree does not fit the caches, near otherwise
    bool any_near = true;             // NORI_HIP_ANYORDER=reference: any-hit rays walk the tree in the reference's order too (then every schedule and
                                      // node width tests the same records; by default they enter the nearer child first, which finds occluders sooner)
    bool ordered = false;             // NORI_HIP_ORDER=near: nearer child first in traversal (see DevScene::ordered)
    int prim_lanes = 1;               // see DevScene::prim_lanes (NORI_HIP_PRIM_LANES)
    int trace_threads = 256;          // workgroup size of the trace stages (NORI_HIP_TRACE_THREADS: 256, 512, 768 or 1024)
    float *scratch_ptr = nullptr;     // this device's copy of `scratch` (nori_hip_filter_noalloc)
    size_t trace_lds = 0;             // dynamic LDS of a trace-stage workgroup (set once, with the kernels' attribute, in nori_hip_create)
    bool leaf2_step = true;           // NORI_HIP_LEAF2=0: one record per trip also where the leaves would allow two (Trav::prim_step2)
    int spec = -1;                    // NORI_HIP_SPEC: 1 / 0 = speculative traversal in the merged trace stage or not; -1 = unless the tree does not fit the
                                      // caches (there the stage waits for HBM lines, and the extra boxes that speculation lets pass cost more than the waiting it saves)
    int xcd_chunks = -1;              // NORI_HIP_XCD: chunks of the sorted ray order per XCD in the trace stage's deal (0 = off, -1 = 64)
    int dyn_tries = 3;                // NORI_HIP_DYN_TRIES: queues a finishing wave tries (1..8)
    int dyn_pct = -1;                 // NORI_HIP_DYN: percent of the trace stage's rounds of batches that are handed out dynamically at the end (-1: 15, or 25 for trees beyond the caches)
    int box2 = -1;                    // (-1 = 36) NORI_HIP_BOX2: lanes at an inner node after the pops that make the trace stage run a second box step in the trip
    bool merge_trace = true;          // one trace launch per iteration for shadow and extend rays (NORI_HIP_MERGE=0: two launches)
    bool profile = false;             // NORI_HIP_PROFILE=1: step statistics of the trace stages, printed by nori_hip_stats
    int thr_box = 1, thr_prim = 1, thr_refill = 12;   // k_trace_pool step thresholds (lanes), NORI_HIP_THR
    // Two pipelines (render_entry): a render call over a tree beyond the caches is cut into two halves of its sample range that run side by
    // side, each through its own pool and stream - this context and its `twin`, which borrows the scene, the film and every setting and owns
    // only what a pipeline needs for itself (stream, pool, counters, live list, sort buffers, events).
    int pipelines = -1;               // NORI_HIP_PIPELINES: 1 or 2; -1 = two for trees beyond the caches and for calls of at least twin_min_samples on scenes without
                                      // images or subsurface shapes, one otherwise (measured, DESIGN findings 41 and 42)
    unsigned long long twin_min_samples = 1ull << 20;   // NORI_HIP_TWIN_MIN (measured from 1 M samples per call on: cbox 512 x 512 x 4 spp 4.93 -> 4.81 ms, x 16 spp 10.0 -> 9.0)
    // two pipelines on a scene that lives in the caches: the trace stage leaves a part of every CU to the other pipeline's shade stage, which then
    // runs beside it instead of after it - three workgroups of 256 per CU instead of five, and the shade stage built for five waves per SIMD (it
    // fits beside them twice).  Only where NORI_HIP_TRACE_BPC / NORI_HIP_SHADE_WAVES were not given.  Set per call by render_entry.
    int twin_trace_bpc = 0, twin_shade_waves = 0; bool shade_waves_given = false;
    NoriHipContext *twin = nullptr; bool is_twin = false;
    bool twin_tail_merged = true;     // NORI_HIP_TWIN_TAIL=separate: each pipeline launches its own tail kernel
    hipEvent_t twin_go = nullptr, twin_done = nullptr;
    // adds the elapsed time of the recorded extend launches to ext_ms_acc (waits for them)
    int fold_ext_events() {
        for (size_t k = 0; k + 1 < ext_used; k += 2) {
            float ms = 0.f;
            if (hipEventSynchronize(ext_ev[k + 1]) != hipSuccess || hipEventElapsedTime(&ms, ext_ev[k], ext_ev[k + 1]) != hipSuccess) { (void) hipGetLastError(); continue; }
            ext_ms_acc += ms; ext_launches++;
        }
        ext_used = 0;
        return NORI_HIP_OK;
    }
    void free_scene() {
        if (is_twin) {           // borrowed from the context it belongs to: forget, do not free
            d_nodes4 = nullptr; d_nodes8 = nullptr; d_nodes4q = nullptr; d_env_cdf = nullptr; d_nodes = nullptr; d_prims = nullptr; d_shade = d_light = nullptr; d_shapes = nullptr;
            d_emitters = nullptr; d_cdf = nullptr; d_film = nullptr; d_textures = nullptr; d_tex_data = nullptr; d_filt = nullptr; have_scene = false;
            return;
        }
        hipFree(d_nodes4); d_nodes4 = nullptr; n_nodes4 = 0;
        hipFree(d_nodes8); d_nodes8 = nullptr; n_nodes8 = 0;
        hipFree(d_nodes4q); d_nodes4q = nullptr;
        hipFree(d_env_cdf); d_env_cdf = nullptr;
        hipFree(d_nodes); hipFree(d_prims); hipFree(d_shade); hipFree(d_light); hipFree(d_shapes); hipFree(d_emitters); hipFree(d_cdf); hipFree(d_film);
        hipFree(d_textures); hipFree(d_tex_data); d_textures = nullptr; d_tex_data = nullptr; n_textures = 0; ext = false;
        d_nodes = nullptr; d_prims = nullptr; d_shade = d_light = nullptr; d_shapes = nullptr; d_emitters = nullptr; d_cdf = nullptr; d_film = nullptr;
        have_scene = false;
    }
};

#define CTXCHK(ctx) do { if (!(ctx)) return fail(NORI_HIP_EINVAL, "null context"); HIPCHK(hipSetDevice((ctx)->device)); } while (0)
#define SCENECHK(ctx) do { CTXCHK(ctx); if (!(ctx)->have_scene) return fail(NORI_HIP_ESTATE, "no scene uploaded"); } while (0)

extern "C" {

const char *nori_hip_last_error(void) { return g_err.c_str(); }
void nori_hip_set_error_(const char *msg) { g_err = msg ? msg : ""; }       // for the library's other translation units (nori_group.hip)

static int create_impl(NoriHipContext *c, int device);

int nori_hip_device_count(int *n) {
    if (!n) return fail(NORI_HIP_EINVAL, "null argument");
    HIPCHK(hipGetDeviceCount(n));
    return NORI_HIP_OK;
}

int nori_hip_create(int device, NoriHipContext **out) {
    if (!out) return fail(NORI_HIP_EINVAL, "out is null");
    *out = nullptr;
    int count = 0;
    HIPCHK(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) return fail(NORI_HIP_EDEVICE, "no such HIP device " + std::to_string(device));
    HIPCHK(hipSetDevice(device));
    NoriHipContext *c = new NoriHipContext();
    c->device = device;
    const int rc = create_impl(c, device);
    if (rc != NORI_HIP_OK) { const std::string why = g_err; nori_hip_destroy(c); g_err = why; return rc; }     // nothing half-built is left behind
    *out = c;
    return NORI_HIP_OK;
}

static int create_impl(NoriHipContext *c, int device) {
    hipDeviceProp_t prop; HIPCHK(hipGetDeviceProperties(&prop, device));
    c->n_cus = prop.multiProcessorCount;
    { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess && v > 0) c->lds_bytes = (size_t) v; }
    if (const char *e = getenv("NORI_HIP_LDS")) { long v = atol(e); if (v >= 16384 && v <= 163840) c->lds_bytes = (size_t) v; }
    HIPCHK(hipStreamCreate(&c->stream));
    HIPCHK(hipEventCreate(&c->ev0)); HIPCHK(hipEventCreate(&c->ev1));
    HIPCHK(hipMalloc((void **) &c->d_counter, sizeof(unsigned long long)));
    HIPCHK(hipMalloc((void **) &c->d_stats, ND_STRIPES * sizeof(DevStats)));
    HIPCHK(hipMalloc((void **) &c->d_tiles, (size_t) (5 * ND_MAX_TILES + 1) * sizeof(int)));
    HIPCHK(hipMemset(c->d_stats, 0, ND_STRIPES * sizeof(DevStats)));
    std::memset(&c->stats, 0, sizeof(c->stats));
    if (const char *e = getenv("NORI_HIP_PIPELINE")) { c->pipeline = (std::string(e) == "mega") ? 0 : 1; c->pipeline_forced = true; }
    if (const char *e = getenv("NORI_HIP_LBVH_BIG")) { float v = (float) atof(e); if (v >= 0.f) c->lbvh_big = v; }
    if (const char *e = getenv("NORI_HIP_LBVH_MORTON")) c->lbvh_plain = std::string(e) == "plain";
    if (const char *e = getenv("NORI_HIP_LBVH_LEAF")) { int v = atoi(e); if (v >= 1 && v <= 64) c->lbvh_leaf = (uint32_t) v; }
    if (const char *e = getenv("NORI_HIP_SPEC")) c->spec = atoi(e) != 0 ? 1 : 0;
    if (const char *e = getenv("NORI_HIP_LEAF2")) c->leaf2_step = atoi(e) != 0;
    if (const char *e = getenv("NORI_HIP_QUANT")) c->quant4 = atoi(e) != 0 ? 1 : 0;
    if (const char *e = getenv("NORI_HIP_WIDE")) { const int v = atoi(e); c->wide = v != 0; c->wide8 = v == 8 ? 1 : 0; }
    if (const char *e = getenv("NORI_HIP_SORT")) c->sort_rays = atoi(e);
    if (const char *e = getenv("NORI_HIP_KEYS")) c->fused_keys = std::string(e) != "kernel";
    if (const char *e = getenv("NORI_HIP_SORT_BITS")) { int v = atoi(e); if (v >= 8 && v <= 30) c->sort_bits = v; }
    if (const char *e = getenv("NORI_HIP_DYN_TRIES")) { int v = atoi(e); if (v >= 1 && v <= 8) c->dyn_tries = v; }
    if (const char *e = getenv("NORI_HIP_CULL")) c->cull_zero = atoi(e) != 0;
    if (const char *e = getenv("NORI_HIP_LBVH_ALIGN")) c->align_leaves = atoi(e) != 0 ? 1 : 0;
    if (const char *e = getenv("NORI_HIP_XCD")) { int v = atoi(e); if (v >= 0 && v <= 255) c->xcd_chunks = v; }
    if (const char *e = getenv("NORI_HIP_DYN")) { int v = atoi(e); if (v >= 0 && v <= 90) c->dyn_pct = v; }
    if (const char *e = getenv("NORI_HIP_BOX2")) { int v = atoi(e); if (v >= 0 && v <= 64) c->box2 = v; }
    if (const char *e = getenv("NORI_HIP_COMPACT")) c->compact = atoi(e) != 0;
    if (const char *e = getenv("NORI_HIP_FINISH")) c->finish_below = (unsigned int) atol(e);
    if (const char *e = getenv("NORI_HIP_PIPELINES")) { int v = atoi(e); if (v == 1 || v == 2) c->pipelines = v; }
    if (const char *e = getenv("NORI_HIP_TWIN_TAIL")) c->twin_tail_merged = std::string(e) != "separate";
    if (const char *e = getenv("NORI_HIP_FINISH_WAVES")) { int v = atoi(e); if (v == 3 || v == 4) c->finish_waves = v; }
    if (const char *e = getenv("NORI_HIP_SHADE_WAVES")) { int v = atoi(e); if (v == 3 || v == 4 || v == 5 || v == 6 || v == 8) { c->shade_waves = v; c->shade_waves_given = true; } }
    if (const char *e = getenv("NORI_HIP_TWIN_MIN")) { long long v = atoll(e); if (v >= 0) c->twin_min_samples = (unsigned long long) v; }
    if (const char *e = getenv("NORI_HIP_SHADE_BPC")) { int v = atoi(e); if (v >= 1 && v <= 64) c->shade_blocks_per_cu = v; }
    if (const char *e = getenv("NORI_HIP_TRACE_BPC")) { int v = atoi(e); if (v >= 0 && v <= 64) c->trace_blocks_per_cu = v; }
    if (const char *e = getenv("NORI_HIP_MEGA_WAVES")) { int v = atoi(e); if (v == 3 || v == 4 || v == 5 || v == 6 || v == 8) c->mega_waves = v; }
    if (const char *e = getenv("NORI_HIP_ORDER")) c->ordered = std::string(e) == "near";
    if (const char *e = getenv("NORI_HIP_ANYORDER")) { c->any_near = std::string(e) != "reference"; c->any_far = std::string(e) == "far" ? 1 : 0; }
    if (const char *e = getenv("NORI_HIP_PRIM_LANES")) { int v = atoi(e); if (v >= 1 && v <= 64) c->prim_lanes = v; }
    if (const char *e = getenv("NORI_HIP_TRACE_THREADS")) { int v = atoi(e); if (v == 256 || v == 512 || v == 768 || v == 1024) c->trace_threads = v; }
    if (const char *e = getenv("NORI_HIP_MERGE")) c->merge_trace = atoi(e) != 0;
    if (const char *e = getenv("NORI_HIP_PROFILE")) c->profile = atoi(e) != 0;
    if (const char *e = getenv("NORI_HIP_THR")) {          // "box,prim,refill" lane thresholds, for tuning
        int a, b, d; if (sscanf(e, "%d,%d,%d", &a, &b, &d) == 3 && a >= 1 && b >= 1 && d >= 1 && a <= 64 && b <= 64 && d <= 64) { c->thr_box = a; c->thr_prim = b; c->thr_refill = d; }
    }
    c->pool_slots = 0;                 // 0: sized per render call (three eighths of the call's paths, half for big trees, between 1M and 8M slots)
    if (const char *e = getenv("NORI_HIP_POOL")) { long v = atol(e); if (v >= 64 && v <= (1l << 26)) c->pool_slots = (unsigned int) v; }
    HIPCHK(hipMalloc((void **) &c->d_cnt, sizeof(WfCounters)));
    HIPCHK(hipHostMalloc((void **) &c->h_ring, 4 * sizeof(unsigned int), hipHostMallocDefault));
    for (int i = 0; i < 4; ++i) HIPCHK(hipEventCreateWithFlags(&c->ring_ev[i], hipEventDisableTiming));
    // the device address of this GPU's copy of the no-allocation scratch array (the array's name is a host-side stand-in)
    HIPCHK(hipGetSymbolAddress((void **) &c->scratch_ptr, HIP_SYMBOL(scratch)));
    if (!c->scratch_ptr) return fail(NORI_HIP_EDEVICE, "hipGetSymbolAddress(scratch) returned a null address");
    // dynamic LDS of the trace stages: stacks + staged rays (origin, direction, reciprocal direction, slot) of the workgroup's waves
    c->trace_lds = (size_t) (c->trace_threads / 64) * (TRACE_DEPTH * 64 * 8 + 64 * 52);
    HIPCHK(hipFuncSetAttribute((const void *) k_trace_pool<TK_BOTH, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) c->trace_lds));
    HIPCHK(hipFuncSetAttribute((const void *) k_trace_pool<TK_SHADOW, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) c->trace_lds));
    HIPCHK(hipFuncSetAttribute((const void *) k_trace_pool<TK_EXTEND, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) c->trace_lds));
    HIPCHK(hipFuncSetAttribute((const void *) k_trace_pool<TK_BOTH, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) c->trace_lds));
    HIPCHK(hipFuncSetAttribute((const void *) k_trace_pool<TK_BOTH, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) c->trace_lds));
    HIPCHK(hipFuncSetAttribute((const void *) k_trace_pool<TK_BOTH, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) c->trace_lds));
    HIPCHK(hipFuncSetAttribute((const void *) k_trace_pool<TK_SHADOW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) c->trace_lds));
    HIPCHK(hipFuncSetAttribute((const void *) k_trace_pool<TK_EXTEND, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) c->trace_lds));
    c->trace_lds_wide = (size_t) (c->trace_threads / 64) * (TRACE_DEPTH_W * 64 * 8 + 64 * 52);
    HIPCHK(hipFuncSetAttribute((const void *) k_trace_pool4_masked, hipFuncAttributeMaxDynamicSharedMemorySize, (int) c->trace_lds_wide));
    HIPCHK(hipFuncSetAttribute((const void *) k_trace_pool4_leaf2, hipFuncAttributeMaxDynamicSharedMemorySize, (int) c->trace_lds_wide));
    HIPCHK(hipFuncSetAttribute((const void *) k_trace_pool4q_leaf2, hipFuncAttributeMaxDynamicSharedMemorySize, (int) c->trace_lds_wide));
    HIPCHK(hipFuncSetAttribute((const void *) k_trace_pool4<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) c->trace_lds_wide));
    HIPCHK(hipFuncSetAttribute((const void *) k_trace_pool4<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) c->trace_lds_wide));
    HIPCHK(hipFuncSetAttribute((const void *) k_trace_pool4<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) c->trace_lds_wide));
    HIPCHK(hipFuncSetAttribute((const void *) k_trace_pool4<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) c->trace_lds_wide));
    // the filter kernels may use whatever LDS a workgroup can have (their tile grows with r + f): set once, here, so that the filter
    // entry points - nori_hip_filter_noalloc in particular, which is meant for capture and tight loops - change no driver state
    HIPCHK(hipFuncSetAttribute((const void *) k_nlmeans, hipFuncAttributeMaxDynamicSharedMemorySize, (int) std::min<size_t>(c->lds_bytes, 160 * 1024)));
    HIPCHK(hipFuncSetAttribute((const void *) k_nlmeans_sep, hipFuncAttributeMaxDynamicSharedMemorySize, (int) std::min<size_t>(c->lds_bytes, 160 * 1024)));
    c->trace_lds_wide8 = (size_t) (c->trace_threads / 64) * (TRACE_DEPTH_W8 * 64 * 8 + 64 * 52);
    HIPCHK(hipFuncSetAttribute((const void *) k_trace_pool8<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) c->trace_lds_wide8));
    HIPCHK(hipFuncSetAttribute((const void *) k_trace_pool8<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) c->trace_lds_wide8));
    HIPCHK(hipFuncSetAttribute((const void *) k_trace_pool8<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) c->trace_lds_wide8));
    HIPCHK(hipFuncSetAttribute((const void *) k_trace_pool8<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) c->trace_lds_wide8));
    return NORI_HIP_OK;
}

int nori_hip_destroy(NoriHipContext *ctx) {
    if (!ctx) return NORI_HIP_OK;
    hipSetDevice(ctx->device);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    if (ctx->twin) { nori_hip_destroy(ctx->twin); ctx->twin = nullptr; }
    if (ctx->twin_go) hipEventDestroy(ctx->twin_go);
    if (ctx->twin_done) hipEventDestroy(ctx->twin_done);
    ctx->free_scene();
    hipFree(ctx->d_counter); hipFree(ctx->d_stats); hipFree(ctx->d_tiles); hipFree(ctx->d_cnt); hipFree(ctx->pool_mem);
    hipFree(ctx->d_qlive); hipFree(ctx->d_filt); hipFree(ctx->d_sort);
    if (ctx->h_ring) hipHostFree(ctx->h_ring);
    for (int i = 0; i < 4; ++i) if (ctx->ring_ev[i]) hipEventDestroy(ctx->ring_ev[i]);
    for (hipEvent_t e : ctx->ext_ev) hipEventDestroy(e);
    if (ctx->ev0) hipEventDestroy(ctx->ev0);
    if (ctx->ev1) hipEventDestroy(ctx->ev1);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    (void) hipGetLastError();
    delete ctx;
    return NORI_HIP_OK;
}

// The uploaded binary tree as four-box nodes (Trav::box_step4).  A four-box node starts from a binary inner node's two children
// and, while it has a free slot, opens the inner child with the largest box surface in place, so the slots stay in the binary
// tree's depth-first order (the order the reference visits them in, src/bvh.cpp:425-431); the inner children left become four-box
// nodes themselves (breadth-first numbering: the top of the tree comes first).  Also the check the kernels rely on: every node is
// reached once (no cycles, no shared subtrees).  out: 32 floats per node; depth: levels of inner nodes of the binary tree;
// depth4: levels of four-box nodes.
static int collapse_host(const std::vector<NoriHipNode> &nodes, uint32_t root, std::vector<float> &out, uint32_t &root4, uint32_t &depth, uint32_t &depth4) {
    out.clear(); depth = 0; depth4 = 0; root4 = root;
    if (root & NORI_HIP_LEAF_BIT) return NORI_HIP_OK;
    struct Item { uint32_t bin, level, level4; };
    std::vector<Item> queue; queue.push_back({root, 0, 0});
    std::vector<unsigned char> seen(nodes.size(), 0); seen[root] = 1;
    const float inf = __builtin_huge_valf();
    auto area = [](const float *mn, const float *mx) { const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2]; return dx * dy + dy * dz + dz * dx; };
    for (size_t k = 0; k < queue.size(); ++k) {
        const Item it = queue[k]; const NoriHipNode &w = nodes[it.bin];
        float mn[4][3], mx[4][3]; uint32_t ref[4], lvl[4]; int m = 0;
        for (int j = 0; j < 4; ++j) { ref[j] = NORI_HIP_NO_CHILD; lvl[j] = 0; for (int a = 0; a < 3; ++a) { mn[j][a] = inf; mx[j][a] = -inf; } }
        depth = std::max(depth, it.level + 1); depth4 = std::max(depth4, it.level4 + 1);
        for (int side = 0; side < 2; ++side) {
            const uint32_t r = side ? w.right : w.left;
            if (r == NORI_HIP_NO_CHILD) continue;
            for (int a = 0; a < 3; ++a) { mn[m][a] = side ? w.rmin[a] : w.lmin[a]; mx[m][a] = side ? w.rmax[a] : w.lmax[a]; }
            lvl[m] = it.level + 1; ref[m++] = r;
        }
        while (m < 4) {
            int best = -1; float best_area = -1.f;
            for (int j = 0; j < m; ++j) if (!(ref[j] & NORI_HIP_LEAF_BIT)) { const float ar = area(mn[j], mx[j]); if (ar > best_area) { best_area = ar; best = j; } }
            if (best < 0) break;
            const uint32_t r = ref[best];
            if (seen[r]) return fail(NORI_HIP_EINVAL, "node table is not a tree: node " + std::to_string(r) + " is reached twice");
            seen[r] = 1;
            const NoriHipNode &c = nodes[r]; const uint32_t cl = lvl[best] + 1;
            depth = std::max(depth, cl);
            if (c.right == NORI_HIP_NO_CHILD) {
                for (int a = 0; a < 3; ++a) { mn[best][a] = c.lmin[a]; mx[best][a] = c.lmax[a]; }
                ref[best] = c.left; lvl[best] = cl; continue;
            }
            for (int j = m; j > best + 1; --j) { ref[j] = ref[j - 1]; lvl[j] = lvl[j - 1]; for (int a = 0; a < 3; ++a) { mn[j][a] = mn[j - 1][a]; mx[j][a] = mx[j - 1][a]; } }
            for (int a = 0; a < 3; ++a) { mn[best][a] = c.lmin[a]; mx[best][a] = c.lmax[a]; mn[best + 1][a] = c.rmin[a]; mx[best + 1][a] = c.rmax[a]; }
            ref[best] = c.left; ref[best + 1] = c.right; lvl[best] = lvl[best + 1] = cl; ++m;
        }
        for (int j = 0; j < m; ++j) if (!(ref[j] & NORI_HIP_LEAF_BIT)) {
            if (seen[ref[j]]) return fail(NORI_HIP_EINVAL, "node table is not a tree: node " + std::to_string(ref[j]) + " is reached twice");
            seen[ref[j]] = 1;
            queue.push_back({ref[j], lvl[j], it.level4 + 1}); ref[j] = (uint32_t) queue.size() - 1;
        }
        const size_t base = out.size(); out.resize(base + 32, 0.f);
        for (int j = 0; j < 4; ++j) {
            for (int a = 0; a < 3; ++a) { out[base + 4 * a + j] = mn[j][a]; out[base + 12 + 4 * a + j] = mx[j][a]; }
            std::memcpy(&out[base + 24 + j], &ref[j], 4);
        }
    }
    root4 = 0;
    return NORI_HIP_OK;
}

static int upload_scene_impl(NoriHipContext *ctx, const NoriHipSceneDesc *d, const bool device_build) {
    CTXCHK(ctx);
    if (!d) return fail(NORI_HIP_EINVAL, "scene description is null");
    if (d->n_prims == 0 || !d->prims || !d->shade_prims) return fail(NORI_HIP_EINVAL, "scene has no primitives");
    if (d->n_shapes == 0 || !d->shapes) return fail(NORI_HIP_EINVAL, "scene has no shapes");
    if (!device_build && d->n_nodes > 0 && !d->nodes) return fail(NORI_HIP_EINVAL, "nodes is null");
    if (d->camera.width <= 0 || d->camera.height <= 0) return fail(NORI_HIP_EINVAL, "bad camera size");
    if (d->camera.width > 65535 || d->camera.height > 65535) return fail(NORI_HIP_EINVAL, "camera size above 65535 (the path pool packs a pixel into 2 x 16 bits)");
    if (d->n_textures && (!d->textures || !d->tex_data)) return fail(NORI_HIP_EINVAL, "textures is null");
    // validate every index the kernels will follow: a bad table must be rejected here, not fault on the GPU
    auto ref_ok = [&](uint32_t r) {
        if (r == NORI_HIP_NO_CHILD) return true;
        if (r & NORI_HIP_LEAF_BIT) return (r & 0x7fffffffu) < d->n_prims;
        return r < d->n_nodes;
    };
    if (!device_build) {
        if (d->root == NORI_HIP_NO_CHILD || !ref_ok(d->root)) return fail(NORI_HIP_EINVAL, "bad root reference");
        for (uint32_t i = 0; i < d->n_nodes; ++i)
            if (!ref_ok(d->nodes[i].left) || d->nodes[i].left == NORI_HIP_NO_CHILD || !ref_ok(d->nodes[i].right))
                return fail(NORI_HIP_EINVAL, "node " + std::to_string(i) + " has a child reference out of range");
        if (!(d->prims[d->n_prims - 1].shape_flags & 0x80000000u)) return fail(NORI_HIP_EINVAL, "last primitive record does not end a leaf");
    }
    for (uint32_t i = 0; i < d->n_prims; ++i)
        if ((d->prims[i].shape_flags & 0xffffffu) >= d->n_shapes) return fail(NORI_HIP_EINVAL, "primitive record " + std::to_string(i) + " names a shape out of range");
    bool ext = false;
    for (uint32_t i = 0; i < d->n_shapes; ++i) {
        const NoriHipShape &s = d->shapes[i];
        if (s.emitter >= (int32_t) d->n_emitters) return fail(NORI_HIP_EINVAL, "shape emitter index out of range");
        if (s.bsdf < 0 || s.bsdf > NORI_HIP_BSDF_SUBSURFACE) return fail(NORI_HIP_EINVAL, "unknown BSDF kind");
        if (s.bsdf == NORI_HIP_BSDF_SUBSURFACE) { ext = true; if (!(s.ss_cs > 0.f) || !(s.ss_cd > 0.f)) return fail(NORI_HIP_EINVAL, "subsurface shape without ss_cs / ss_cd"); }
        if (s.albedo_tex > d->n_textures || s.normal_tex > d->n_textures) return fail(NORI_HIP_EINVAL, "shape names a texture out of range");
        if (s.albedo_tex && d->textures[s.albedo_tex - 1].kind != NORI_HIP_TEX_COLOR && d->textures[s.albedo_tex - 1].kind != NORI_HIP_TEX_CHECKER_COLOR) return fail(NORI_HIP_EINVAL, "albedo texture is not a colour texture");
        if (s.normal_tex && (d->textures[s.normal_tex - 1].kind == NORI_HIP_TEX_COLOR || d->textures[s.normal_tex - 1].kind == NORI_HIP_TEX_CHECKER_COLOR)) return fail(NORI_HIP_EINVAL, "normal modifier is a colour texture");
        if (s.albedo_tex || s.normal_tex) ext = true;
    }
    for (uint32_t i = 0; i < d->n_textures; ++i) {
        const NoriHipTexture &t = d->textures[i];
        if (t.kind > NORI_HIP_TEX_IMAGE_FLOAT || t.lookup > NORI_HIP_LOOKUP_BILINEAR) return fail(NORI_HIP_EINVAL, "unknown texture kind or lookup");
        if (t.kind == NORI_HIP_TEX_CHECKER_COLOR && (t.width != 6 || t.height != 1)) return fail(NORI_HIP_EINVAL, "checkerboard colour texture must carry six floats (width 6, height 1)");
        if (t.kind == NORI_HIP_TEX_CHECKER_FLOAT && (t.width != 2 || t.height != 1)) return fail(NORI_HIP_EINVAL, "checkerboard float texture must carry two floats (width 2, height 1)");
        if (t.width == 0 || t.height == 0 || (t.offset & 3u) || t.offset + (uint64_t) t.width * t.height * 4u > d->n_tex_bytes)
            return fail(NORI_HIP_EINVAL, "texture " + std::to_string(i) + " lies outside tex_data");
        if (t.scale[0] == 0.f || t.scale[1] == 0.f) return fail(NORI_HIP_EINVAL, "texture scale is zero");
    }
    if (d->env.texture) {
        if (d->env.texture > d->n_textures || d->textures[d->env.texture - 1].kind != NORI_HIP_TEX_COLOR) return fail(NORI_HIP_EINVAL, "environment emitter names no colour image");
        if (!(d->env.brightness >= 0.f)) return fail(NORI_HIP_EINVAL, "environment emitter: negative brightness");
        ext = true;
    }
    for (uint32_t i = 0; i < d->n_emitters; ++i) {
        const NoriHipEmitter &e = d->emitters[i];
        if (e.shape < 0 || e.shape >= (int32_t) d->n_shapes) return fail(NORI_HIP_EINVAL, "emitter shape index out of range");
        if (d->shapes[e.shape].type == NORI_HIP_SHAPE_MESH) {
            if (e.n_tris == 0 || (uint64_t) e.cdf_offset + e.n_tris + 1 > d->n_cdf) return fail(NORI_HIP_EINVAL, "emitter cdf range out of bounds");
            if ((uint64_t) e.light_prim_offset + e.n_tris > d->n_light_prims) return fail(NORI_HIP_EINVAL, "emitter light_prims range out of bounds");
        }
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->free_scene();
    int rc;
    // Renumber the inner nodes so that the top of the tree comes first, in breadth-first order: kernels keep a prefix
    // of the node array in LDS and test `index < cached`.  Only the numbering changes; boxes, child order and the
    // leaf records are the caller's.
    uint32_t root = d->root, n_top = 0, n_records = d->n_prims;
    float root_min[3] = {d->root_min[0], d->root_min[1], d->root_min[2]}, root_max[3] = {d->root_max[0], d->root_max[1], d->root_max[2]};
    ctx->lbvh = lbvh::Result();
    if (device_build) {
        // the tree is built here, on the device, from the records in whatever order they came
        NoriHipPrim *tmp_prims = nullptr; NoriHipShadePrim *tmp_shade = nullptr;
        if ((rc = upload(tmp_prims, d->prims, d->n_prims))) return rc;
        if ((rc = upload(tmp_shade, d->shade_prims, d->n_prims))) { hipFree(tmp_prims); return rc; }
        hipError_t e1_ = hipMalloc((void **) &ctx->d_prims, (size_t) d->n_prims * sizeof(NoriHipPrim));
        hipError_t e2_ = hipMalloc((void **) &ctx->d_shade, (size_t) d->n_prims * sizeof(NoriHipShadePrim));
        hipError_t e3_ = hipMalloc((void **) &ctx->d_nodes, (size_t) std::max<uint32_t>(d->n_prims - 1, 1u) * sizeof(NoriHipNode));
        std::string lerr;
        rc = (e1_ != hipSuccess || e2_ != hipSuccess || e3_ != hipSuccess) ? NORI_HIP_EDEVICE
             : lbvh::build(ctx->stream, d->n_prims, tmp_prims, tmp_shade, ctx->d_prims, ctx->d_shade, ctx->d_nodes, ctx->lbvh_leaf ? ctx->lbvh_leaf : (d->n_prims >= 1000000u ? 2u : 3u), ctx->lbvh_big, ND_STACK - 2, ctx->lbvh_plain, ctx->wide && ctx->wide8 == 1 && !ext,
                           ctx->align_leaves >= 0 ? ctx->align_leaves != 0 : d->n_prims >= 1000000u, ctx->lbvh, lerr);
        hipFree(tmp_prims); hipFree(tmp_shade);
        if (rc) { hipFree(ctx->lbvh.nodes4); ctx->lbvh.nodes4 = nullptr; hipFree(ctx->lbvh.nodes8); ctx->lbvh.nodes8 = nullptr; hipFree(ctx->lbvh.prims_al); hipFree(ctx->lbvh.shade_al); ctx->lbvh.prims_al = ctx->lbvh.shade_al = nullptr; ctx->free_scene(); return fail(rc, lerr.empty() ? "allocation of the device-built tree failed" : lerr); }
        ctx->d_nodes4 = (float4 *) ctx->lbvh.nodes4; ctx->n_nodes4 = ctx->lbvh.n_nodes4; ctx->lbvh.nodes4 = nullptr;
        n_records = d->n_prims;
        const bool placed = ctx->lbvh.prims_al != nullptr;
        if (placed) {
            // the records placed leaf by leaf (no leaf straddles a line): they replace the arrays in sorted order
            hipFree(ctx->d_prims); hipFree(ctx->d_shade);
            ctx->d_prims = (NoriHipPrim *) ctx->lbvh.prims_al; ctx->d_shade = (NoriHipShadePrim *) ctx->lbvh.shade_al; ctx->lbvh.prims_al = ctx->lbvh.shade_al = nullptr;
            n_records = ctx->lbvh.n_positions;
        }
        // up to three stack entries per four-box level: deeper trees are walked through their two-box nodes
        if (3u * ctx->lbvh.depth4 > (uint32_t) (ND_STACK - 2)) { hipFree(ctx->d_nodes4); ctx->d_nodes4 = nullptr; ctx->n_nodes4 = 0; }
        // quantised eight-box nodes (NORI_HIP_WIDE=8), if their traversal fits the stack; scenes with images or subsurface shapes
        // keep the four-box walk (their kernels trace masked probe rays)
        ctx->d_nodes8 = (uint4 *) ctx->lbvh.nodes8; ctx->n_nodes8 = ctx->lbvh.n_nodes8; ctx->lbvh.nodes8 = nullptr;
        {
            const bool use8 = ctx->d_nodes8 && ctx->d_nodes4 && ctx->lbvh.stack8 + 2u <= (uint32_t) ND_STACK8 && ctx->wide8 == 1;
            if (!use8) { hipFree(ctx->d_nodes8); ctx->d_nodes8 = nullptr; ctx->n_nodes8 = 0; }
        }
        // the 64-byte quantised four-box nodes, for the trace stage of trees beyond the caches (scenes without probe rays)
        if (ctx->d_nodes4 && !ext && placed && (ctx->quant4 >= 0 ? ctx->quant4 != 0 : true)) {
            void *nq = nullptr;
            if (lbvh::quantize4(ctx->stream, ctx->d_nodes4, ctx->n_nodes4, &nq) != hipSuccess) { hipFree(nq); ctx->free_scene(); return fail(NORI_HIP_EDEVICE, "building the quantised four-box nodes failed"); }
            ctx->d_nodes4q = (uint4 *) nq;
        }
        root = ctx->lbvh.root;
        for (int i = 0; i < 3; ++i) { root_min[i] = ctx->lbvh.root_min[i]; root_max[i] = ctx->lbvh.root_max[i]; }
    } else {
        std::vector<NoriHipNode> nodes(d->nodes, d->nodes + d->n_nodes);
        if (d->n_nodes > 0 && !(d->root & NORI_HIP_LEAF_BIT)) {
            const uint32_t top_max = std::min<uint32_t>(d->n_nodes, 2560u);      // 160 KB of LDS / 64 B
            std::vector<uint32_t> bfs; bfs.reserve(top_max); bfs.push_back(d->root);
            std::vector<uint32_t> new_of(d->n_nodes, 0xffffffffu);
            for (size_t k = 0; k < bfs.size() && bfs.size() < top_max; ++k) {
                const NoriHipNode &nd_ = d->nodes[bfs[k]];
                for (uint32_t ch : {nd_.left, nd_.right})
                    if (ch != NORI_HIP_NO_CHILD && !(ch & NORI_HIP_LEAF_BIT) && bfs.size() < top_max) bfs.push_back(ch);
            }
            n_top = (uint32_t) bfs.size();
            for (uint32_t k = 0; k < n_top; ++k) new_of[bfs[k]] = k;
            uint32_t next = n_top;
            for (uint32_t i = 0; i < d->n_nodes; ++i) if (new_of[i] == 0xffffffffu) new_of[i] = next++;
            auto map_ref = [&](uint32_t r) { return (r == NORI_HIP_NO_CHILD || (r & NORI_HIP_LEAF_BIT)) ? r : new_of[r]; };
            for (uint32_t i = 0; i < d->n_nodes; ++i) { NoriHipNode t = d->nodes[i]; t.left = map_ref(t.left); t.right = map_ref(t.right); nodes[new_of[i]] = t; }
            root = new_of[d->root];
        }
        if ((rc = upload(ctx->d_nodes, nodes.data(), d->n_nodes))) return rc;
        std::vector<float> wide; uint32_t root4w = root, depth = 0, depth4 = 0;
        if ((rc = collapse_host(nodes, root, wide, root4w, depth, depth4))) return rc;
        if (depth > (uint32_t) (ND_STACK - 2)) return fail(NORI_HIP_EINVAL, "tree depth " + std::to_string(depth) + " exceeds the traversal stack (" + std::to_string(ND_STACK - 2) + ")");
        if (!wide.empty() && 3u * depth4 <= (uint32_t) (ND_STACK - 2)) {
            float *w = nullptr;
            if ((rc = upload(w, wide.data(), wide.size()))) return rc;
            ctx->d_nodes4 = (float4 *) w; ctx->n_nodes4 = (uint32_t) (wide.size() / 32);
        }
    }
    if (!device_build) {
        if ((rc = upload(ctx->d_prims, d->prims, d->n_prims))) return rc;
        if ((rc = upload(ctx->d_shade, d->shade_prims, d->n_prims))) return rc;
    }
    if ((rc = upload(ctx->d_light, d->light_prims, d->n_light_prims))) return rc;
    if ((rc = upload(ctx->d_shapes, d->shapes, d->n_shapes))) return rc;
    if ((rc = upload(ctx->d_emitters, d->emitters, d->n_emitters))) return rc;
    if ((rc = upload(ctx->d_cdf, d->cdf, d->n_cdf))) return rc;
    if ((rc = upload(ctx->d_textures, d->textures, d->n_textures))) return rc;
    if ((rc = upload(ctx->d_tex_data, d->tex_data, d->n_textures ? (size_t) d->n_tex_bytes : 0))) return rc;
    ctx->ext = ext; ctx->has_ss = false; ctx->n_textures = d->n_textures;
    for (uint32_t i = 0; i < d->n_shapes; ++i) if (d->shapes[i].bsdf == NORI_HIP_BSDF_SUBSURFACE) ctx->has_ss = true;
    ctx->n_prims = n_records; ctx->n_nodes = device_build ? ctx->lbvh.n_nodes : d->n_nodes;
    DevScene &sc = ctx->sc;
    sc.nodes = ctx->d_nodes; sc.prims = reinterpret_cast<const float4 *>(ctx->d_prims); sc.shade = ctx->d_shade;
    sc.shapes = ctx->d_shapes; sc.emitters = ctx->d_emitters; sc.cdf = ctx->d_cdf; sc.light_prims = ctx->d_light;
    sc.textures = ctx->d_textures; sc.tex_data = ctx->d_tex_data;
    // a device-built tree has no reference child order to keep: the nearer child is entered first
    sc.nodes4 = ctx->d_nodes4; sc.root4 = (root & NORI_HIP_LEAF_BIT) ? root : 0u; sc.nodes8 = ctx->d_nodes8; sc.nodes4q = ctx->d_nodes4q;
    sc.leaf2 = (device_build && n_records != d->n_prims) ? 1 : 0;            // (the placed layout: lbvh::Result::prims_al was taken over above)
    sc.root = root; sc.n_emitters = d->n_emitters; sc.n_top = n_top; sc.prim_lanes = ctx->prim_lanes; sc.cull_zero = ctx->cull_zero ? 1 : 0; {
        const size_t tree_b = (size_t) (ctx->n_nodes4 ? (size_t) ctx->n_nodes4 * 128 : (size_t) ctx->n_nodes * 64) + (size_t) ctx->n_prims * sizeof(NoriHipPrim);
        const bool far = ctx->any_far >= 0 ? ctx->any_far != 0 : tree_b > ((size_t) 192 << 20);
        sc.ordered = ((ctx->ordered || device_build) ? 1 : 0) | (ctx->any_near ? 2 : 0) | (far ? 4 : 0);
    }
    for (int i = 0; i < 3; ++i) { sc.root_min[i] = root_min[i]; sc.root_max[i] = root_max[i]; }
    sc.cam = d->camera;
    sc.inv_w = 1.0f / (float) d->camera.width; sc.inv_h = 1.0f / (float) d->camera.height;
    sc.filt_table = ctx->d_filt; sc.filt_radius = ctx->filt_radius; sc.filt_lookup = ctx->d_filt ? 32.0f / ctx->filt_radius : 0.f;
    ctx->border = (int) std::ceil(d->camera.filter_radius - 0.5f);
    if (ctx->border < 0) ctx->border = 0;
    ctx->film_rows = d->camera.height + 2 * ctx->border; ctx->film_cols = d->camera.width + 2 * ctx->border;
    sc.border = ctx->border; sc.film_cols = ctx->film_cols;
    HIPCHK(hipMalloc((void **) &ctx->d_film, (size_t) ctx->film_rows * ctx->film_cols * sizeof(float4)));
    HIPCHK(hipMemset(ctx->d_film, 0, (size_t) ctx->film_rows * ctx->film_cols * sizeof(float4)));
    sc.env = d->env; sc.env_cdf = nullptr; sc.env_w = sc.env_h = 0;
    if (d->env.texture) {
        // the sampling table of the environment emitter: one cell per texel of its image, weights evaluated on the device through
        // the same lookup the integrator uses (scale, delta and filtering included), summed in double on the host
        const NoriHipTexture &t = d->textures[d->env.texture - 1];
        const uint32_t w = t.width, h = t.height; const size_t n = (size_t) w * h;
        // (every failure from here on releases the scene that has just been uploaded: have_scene stays false, nothing stays attached)
        struct Undo { NoriHipContext *c; bool armed = true; ~Undo() { if (armed) { c->free_scene(); c->sc.env.texture = 0u; c->sc.env_cdf = nullptr; c->sc.env_w = c->sc.env_h = 0u; } } } undo{ctx};
        if (n >= (1u << 28)) return fail(NORI_HIP_EINVAL, "environment image too large");
        sc.env_w = w; sc.env_h = h;
        DevBufs bufs; float *d_wt = nullptr;
        HIPCHK(bufs.get(d_wt, n * 4));
        hipLaunchKernelGGL(k_env_weights, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, ctx->stream, sc, w, h, d_wt);
        HIPCHK(hipGetLastError());
        std::vector<float> wt(n), cdf(n + 1);
        HIPCHK(hipMemcpyAsync(wt.data(), d_wt, n * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        double sum = 0; for (size_t k = 0; k < n; ++k) sum += wt[k];
        if (!(sum > 0)) return fail(NORI_HIP_EINVAL, "environment image has no weight");
        double acc = 0; cdf[0] = 0.f;
        for (size_t k = 0; k < n; ++k) { acc += wt[k]; cdf[k + 1] = (float) (acc / sum); }
        cdf[n] = 1.0f;
        if ((rc = upload(ctx->d_env_cdf, cdf.data(), n + 1))) return rc;
        sc.env_cdf = ctx->d_env_cdf;
        undo.armed = false;
    }
    ctx->have_scene = true;
    return NORI_HIP_OK;
}

int nori_hip_upload_scene(NoriHipContext *ctx, const NoriHipSceneDesc *d) { return upload_scene_impl(ctx, d, false); }
int nori_hip_upload_scene_build(NoriHipContext *ctx, const NoriHipSceneDesc *d) { return upload_scene_impl(ctx, d, true); }
int nori_hip_build_info(NoriHipContext *ctx, uint32_t *n_nodes, uint32_t *n_used, uint32_t *depth, double *build_ms) {
    SCENECHK(ctx);
    if (n_nodes) *n_nodes = ctx->lbvh.n_nodes; if (n_used) *n_used = ctx->lbvh.n_used; if (depth) *depth = ctx->lbvh.depth; if (build_ms) *build_ms = ctx->lbvh.build_ms;
    return NORI_HIP_OK;
}

int nori_hip_set_film_filter(NoriHipContext *ctx, const float *table33, float radius) {
    CTXCHK(ctx);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (!table33) { if (ctx->d_filt) hipFree(ctx->d_filt); ctx->d_filt = nullptr; ctx->filt_radius = 0.f; }
    else {
        if (!(radius > 0.f)) return fail(NORI_HIP_EINVAL, "filter radius must be positive");
        if (ctx->have_scene && (int) std::ceil(radius - 0.5f) > ctx->border) return fail(NORI_HIP_EINVAL, "filter radius exceeds the film border of the uploaded scene (NoriHipCamera::filter_radius)");
        if (!ctx->d_filt) HIPCHK(hipMalloc((void **) &ctx->d_filt, 33 * sizeof(float)));
        HIPCHK(hipMemcpy(ctx->d_filt, table33, 33 * sizeof(float), hipMemcpyHostToDevice));
        ctx->filt_radius = radius;
    }
    ctx->sc.filt_table = ctx->d_filt; ctx->sc.filt_radius = ctx->filt_radius; ctx->sc.filt_lookup = ctx->d_filt ? 32.0f / ctx->filt_radius : 0.f;
    return NORI_HIP_OK;
}

int nori_hip_set_integrator(NoriHipContext *ctx, int kind) {
    CTXCHK(ctx);
    if (kind != NORI_HIP_INTEGRATOR_PATH_MIS && kind != NORI_HIP_INTEGRATOR_PATH_MATS && kind != NORI_HIP_INTEGRATOR_DIRECT_MIS) return fail(NORI_HIP_EINVAL, "unknown integrator kind");
    ctx->integrator = kind;
    return NORI_HIP_OK;
}

int nori_hip_trace_device(NoriHipContext *ctx, const NoriHipRay *d_rays, size_t n, int any_hit, NoriHipHit *d_hits) {
    SCENECHK(ctx);
    if (n == 0) return NORI_HIP_OK;
    if (!d_rays || !d_hits) return fail(NORI_HIP_EINVAL, "null ray/hit buffer");
    unsigned blocks = (unsigned) ((n + 255) / 256);
    hipLaunchKernelGGL(k_trace, dim3(blocks), dim3(256), 0, ctx->stream, ctx->sc, d_rays, n, any_hit, d_hits, (float *) nullptr);
    HIPCHK(hipGetLastError());
    return NORI_HIP_OK;
}

int nori_hip_trace(NoriHipContext *ctx, const NoriHipRay *rays, size_t n, int any_hit, NoriHipHit *hits, float *full) {
    SCENECHK(ctx);
    if (n == 0) return NORI_HIP_OK;
    if (!rays || !hits) return fail(NORI_HIP_EINVAL, "null ray/hit buffer");
    DevBufs bufs; NoriHipRay *d_rays = nullptr; NoriHipHit *d_hits = nullptr; float *d_full = nullptr;
    HIPCHK(bufs.get(d_rays, n * sizeof(NoriHipRay)));
    HIPCHK(bufs.get(d_hits, n * sizeof(NoriHipHit)));
    if (full) HIPCHK(bufs.get(d_full, n * 16 * sizeof(float)));
    HIPCHK(hipMemcpyAsync(d_rays, rays, n * sizeof(NoriHipRay), hipMemcpyHostToDevice, ctx->stream));
    unsigned blocks = (unsigned) ((n + 255) / 256);
    hipLaunchKernelGGL(k_trace, dim3(blocks), dim3(256), 0, ctx->stream, ctx->sc, (const NoriHipRay *) d_rays, n, any_hit, d_hits, d_full);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(hits, d_hits, n * sizeof(NoriHipHit), hipMemcpyDeviceToHost, ctx->stream));
    if (full) HIPCHK(hipMemcpyAsync(full, d_full, n * 16 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return NORI_HIP_OK;
}

int nori_hip_camera_rays(NoriHipContext *ctx, const float *ps, size_t n, NoriHipRay *rays) {
    SCENECHK(ctx);
    if (n == 0) return NORI_HIP_OK;
    if (!ps || !rays) return fail(NORI_HIP_EINVAL, "null buffer");
    DevBufs bufs; float *d_ps = nullptr; NoriHipRay *d_rays = nullptr;
    HIPCHK(bufs.get(d_ps, n * 2 * sizeof(float)));
    HIPCHK(bufs.get(d_rays, n * sizeof(NoriHipRay)));
    HIPCHK(hipMemcpyAsync(d_ps, ps, n * 2 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_camera_rays, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->sc, (const float *) d_ps, n, d_rays);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(rays, d_rays, n * sizeof(NoriHipRay), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return NORI_HIP_OK;
}

int nori_hip_li_samples(NoriHipContext *ctx, size_t n, const int32_t *px, const int32_t *py, const uint64_t *states, const uint64_t *seqs,
                        const NoriHipRay *rays, float *rgb) {
    SCENECHK(ctx);
    if (n == 0) return NORI_HIP_OK;
    if (!states || !seqs || !rgb || ((px == nullptr) != (py == nullptr))) return fail(NORI_HIP_EINVAL, "null buffer");
    if (ctx->sc.n_emitters == 0 && !ctx->sc.env.texture && !(ctx->integrator == NORI_HIP_INTEGRATOR_PATH_MATS)) return fail(NORI_HIP_ESTATE, "path_mis needs at least one emitter");
    DevBufs bufs; int32_t *d_px = nullptr, *d_py = nullptr; uint64_t *d_st = nullptr, *d_sq = nullptr; NoriHipRay *d_rays = nullptr; float *d_rgb = nullptr;
    if (px) { HIPCHK(bufs.get(d_px, n * 4)); HIPCHK(hipMemcpyAsync(d_px, px, n * 4, hipMemcpyHostToDevice, ctx->stream)); }
    if (py) { HIPCHK(bufs.get(d_py, n * 4)); HIPCHK(hipMemcpyAsync(d_py, py, n * 4, hipMemcpyHostToDevice, ctx->stream)); }
    HIPCHK(bufs.get(d_st, n * 8)); HIPCHK(hipMemcpyAsync(d_st, states, n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(bufs.get(d_sq, n * 8)); HIPCHK(hipMemcpyAsync(d_sq, seqs, n * 8, hipMemcpyHostToDevice, ctx->stream));
    if (rays) { HIPCHK(bufs.get(d_rays, n * sizeof(NoriHipRay))); HIPCHK(hipMemcpyAsync(d_rays, rays, n * sizeof(NoriHipRay), hipMemcpyHostToDevice, ctx->stream)); }
    HIPCHK(bufs.get(d_rgb, n * 12));
    if (ctx->integrator == NORI_HIP_INTEGRATOR_DIRECT_MIS)
        hipLaunchKernelGGL((k_li_samples<true, NORI_HIP_INTEGRATOR_DIRECT_MIS>), dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->sc, n, (const int32_t *) d_px, (const int32_t *) d_py,
                           (const uint64_t *) d_st, (const uint64_t *) d_sq, (const NoriHipRay *) d_rays, d_rgb);
    else if (ctx->integrator == NORI_HIP_INTEGRATOR_PATH_MATS)
        hipLaunchKernelGGL((k_li_samples<true, NORI_HIP_INTEGRATOR_PATH_MATS>), dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->sc, n, (const int32_t *) d_px, (const int32_t *) d_py,
                           (const uint64_t *) d_st, (const uint64_t *) d_sq, (const NoriHipRay *) d_rays, d_rgb);
    else if (ctx->ext)
        hipLaunchKernelGGL(k_li_samples<true>, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->sc, n, (const int32_t *) d_px, (const int32_t *) d_py,
                           (const uint64_t *) d_st, (const uint64_t *) d_sq, (const NoriHipRay *) d_rays, d_rgb);
    else
        hipLaunchKernelGGL(k_li_samples<false>, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->sc, n, (const int32_t *) d_px, (const int32_t *) d_py,
                           (const uint64_t *) d_st, (const uint64_t *) d_sq, (const NoriHipRay *) d_rays, d_rgb);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(rgb, d_rgb, n * 12, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return NORI_HIP_OK;
}

// What a pipeline hands back instead of launching its tail kernel
struct TailHandoff { bool valid = false; Pool pool; const unsigned int *q_live = nullptr; const WfCounters *cnt = nullptr; unsigned int n_upper = 0; bool wide = false, wide8 = false, big_tree = false; };
// The tail kernel over one live list, or over the lists of two pipelines (cnt_2 != null)
static int launch_finish(NoriHipContext *ctx, hipStream_t stream, bool wide, bool wide8, bool big_tree, Pool p1, const unsigned int *q1, const WfCounters *c1, unsigned int n_upper1,
                         Pool p2, const unsigned int *q2, const WfCounters *c2, unsigned int n_upper2) {
    const bool fin3 = wide && (ctx->finish_waves ? ctx->finish_waves == 3 : !big_tree);            // the tail kernel built for three waves per SIMD (see k_finish)
    const unsigned fin_waves = fin3 ? 3u : 4u;
    const unsigned long long n_upper = (unsigned long long) n_upper1 + n_upper2 + (c2 ? 64u : 0u);
    const dim3 fin_grid(std::max(1u, std::min<unsigned>((unsigned) ((n_upper + 255) / 256), (unsigned) ctx->n_cus * fin_waves)));
#define ND_FIN(K) hipLaunchKernelGGL(K, fin_grid, dim3(256), 0, stream, ctx->sc, p1, ctx->d_film, q1, c1, ctx->d_stats, p2, q2, c2)
    if (ctx->ext && wide && fin3) ND_FIN((k_finish<true, 4, 3>));
    else if (ctx->ext && wide) ND_FIN((k_finish<true, 4>));
    else if (ctx->ext) ND_FIN(k_finish<true>);
    else if (wide8) ND_FIN((k_finish<false, 8>));
    else if (wide && !fin3 && ctx->sc.nodes4q && ctx->sc.leaf2) ND_FIN((k_finish<false, 4, 4, true>));
    else if (wide && fin3) ND_FIN((k_finish<false, 4, 3>));
    else if (wide) ND_FIN((k_finish<false, 4>));
    else ND_FIN(k_finish<false>);
#undef ND_FIN
    HIPCHK(hipGetLastError());
    return NORI_HIP_OK;
}

static int render_impl(NoriHipContext *ctx, const int *tiles_xyxy, int n_tiles, int spp_begin, int spp_end, TailHandoff *tail = nullptr);

int nori_hip_bsdf_vectors(NoriHipContext *ctx, const NoriHipShape *bsdf, size_t n, const float *wi, const float *sample, const float *sample2, const float *wo, float *out) {
    CTXCHK(ctx);
    if (n == 0) return NORI_HIP_OK;
    if (!bsdf || !wi || !out || (!sample && !wo)) return fail(NORI_HIP_EINVAL, "null buffer");
    if (bsdf->bsdf < 0 || bsdf->bsdf > NORI_HIP_BSDF_SUBSURFACE) return fail(NORI_HIP_EINVAL, "unknown BSDF kind");
    if (bsdf->albedo_tex || bsdf->normal_tex) return fail(NORI_HIP_EINVAL, "nori_hip_bsdf_vectors evaluates constant-colour BSDFs");
    DevBufs bufs; float *d_wi = nullptr, *d_s = nullptr, *d_s2 = nullptr, *d_wo = nullptr, *d_out = nullptr;
    HIPCHK(bufs.get(d_wi, n * 12)); HIPCHK(hipMemcpyAsync(d_wi, wi, n * 12, hipMemcpyHostToDevice, ctx->stream));
    if (sample) { HIPCHK(bufs.get(d_s, n * 8)); HIPCHK(hipMemcpyAsync(d_s, sample, n * 8, hipMemcpyHostToDevice, ctx->stream)); }
    if (sample2) { HIPCHK(bufs.get(d_s2, n * 8)); HIPCHK(hipMemcpyAsync(d_s2, sample2, n * 8, hipMemcpyHostToDevice, ctx->stream)); }
    if (wo) { HIPCHK(bufs.get(d_wo, n * 12)); HIPCHK(hipMemcpyAsync(d_wo, wo, n * 12, hipMemcpyHostToDevice, ctx->stream)); }
    HIPCHK(bufs.get(d_out, n * 48));
    hipLaunchKernelGGL(k_bsdf_vectors, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, ctx->stream, *bsdf, n, (const float *) d_wi, (const float *) d_s, (const float *) d_s2, (const float *) d_wo, d_out);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, d_out, n * 48, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return NORI_HIP_OK;
}

int nori_hip_texture_eval(NoriHipContext *ctx, uint32_t texture, size_t n, const float *uv, const float *nrm, float *out) {
    SCENECHK(ctx);
    if (n == 0) return NORI_HIP_OK;
    if (!uv || !out) return fail(NORI_HIP_EINVAL, "null buffer");
    if (!ctx->d_textures || !ctx->sc.textures) return fail(NORI_HIP_ESTATE, "the uploaded scene has no textures");
    NoriHipTexture t;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (texture >= ctx->n_textures) return fail(NORI_HIP_EINVAL, "texture index out of range");
    HIPCHK(hipMemcpy(&t, ctx->d_textures + texture, sizeof(t), hipMemcpyDeviceToHost));
    const bool colour = t.kind == NORI_HIP_TEX_COLOR || t.kind == NORI_HIP_TEX_CHECKER_COLOR;
    if (!colour && !nrm) return fail(NORI_HIP_EINVAL, "a normal modifier needs the shading normals it modifies");
    DevBufs bufs; float *d_uv = nullptr, *d_n = nullptr, *d_out = nullptr;
    HIPCHK(bufs.get(d_uv, n * 8)); HIPCHK(hipMemcpyAsync(d_uv, uv, n * 8, hipMemcpyHostToDevice, ctx->stream));
    if (!colour) { HIPCHK(bufs.get(d_n, n * 12)); HIPCHK(hipMemcpyAsync(d_n, nrm, n * 12, hipMemcpyHostToDevice, ctx->stream)); }
    HIPCHK(bufs.get(d_out, n * 12));
    hipLaunchKernelGGL(k_texture_eval, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->sc, texture, n, (const float *) d_uv, (const float *) d_n, d_out);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, d_out, n * 12, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return NORI_HIP_OK;
}

int nori_hip_pcg32(NoriHipContext *ctx, uint64_t state, uint64_t seq, size_t n, uint32_t *out_u32, float *out_f32) {
    CTXCHK(ctx);
    if (n == 0) return NORI_HIP_OK;
    if ((out_u32 == nullptr) == (out_f32 == nullptr)) return fail(NORI_HIP_EINVAL, "exactly one of out_u32 / out_f32 must be given");
    if (n > (1u << 20)) return fail(NORI_HIP_EINVAL, "nori_hip_pcg32 is a test entry: at most 2^20 draws");
    DevBufs bufs; uint32_t *d = nullptr;
    HIPCHK(bufs.get(d, n * 4));
    hipLaunchKernelGGL(k_pcg32, dim3(1), dim3(1), 0, ctx->stream, state, seq, (unsigned int) n, out_u32 ? d : (uint32_t *) nullptr, out_u32 ? (float *) nullptr : (float *) d);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out_u32 ? (void *) out_u32 : (void *) out_f32, d, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return NORI_HIP_OK;
}

// does a render call of this context go through the wavefront stages (else: the single persistent kernel)
static bool uses_stages(const NoriHipContext *ctx) {
    const bool mats = ctx->integrator == NORI_HIP_INTEGRATOR_PATH_MATS, direct = ctx->integrator == NORI_HIP_INTEGRATOR_DIRECT_MIS;
    // Scenes with images or subsurface shapes: the stages when the scene has subsurface shapes (their probe rays go through the trace
    // stage) and a tree with inner nodes, else the single kernel (image lookups on a few dozen primitives: measured faster).
    const bool ext_stages = ctx->has_ss && ctx->wide && ctx->d_nodes4 != nullptr && ctx->merge_trace;
    return !(ctx->pipeline == 0 || (ctx->ext && !ctx->pipeline_forced && !ext_stages) || (ctx->has_ss && !ext_stages) || mats || direct);
}
// "big tree": nodes and records against the Infinity Cache (256 MiB), see render_impl
static bool tree_beyond_caches(const NoriHipContext *ctx) {
    const size_t tree_bytes = (size_t) (ctx->n_nodes8 ? (size_t) ctx->n_nodes8 * 128 : ctx->n_nodes4 ? (size_t) ctx->n_nodes4 * 128 : (size_t) ctx->n_nodes * 64) + (size_t) ctx->n_prims * sizeof(NoriHipPrim);
    return tree_bytes > ((size_t) 192 << 20);
}
// The twin takes over everything the context knows about the scene and its settings and keeps what is its own.
static void twin_borrow(NoriHipContext *t, const NoriHipContext *c) {
    struct Own {
        hipStream_t stream; unsigned long long *d_counter; DevStats *d_stats; int *d_tiles; hipEvent_t ev0, ev1; NoriHipStats stats;
        Pool pool, pool_b; void *pool_mem; unsigned int pool_alloc; unsigned int *d_sort; size_t sort_temp_bytes; unsigned int *d_qlive; WfCounters *d_cnt;
        unsigned int *h_ring; hipEvent_t ring_ev[4]; std::vector<hipEvent_t> ext_ev; size_t ext_used; double ext_ms_acc; uint64_t ext_launches;
    } own = {t->stream, t->d_counter, t->d_stats, t->d_tiles, t->ev0, t->ev1, t->stats, t->pool, t->pool_b, t->pool_mem, t->pool_alloc, t->d_sort, t->sort_temp_bytes, t->d_qlive, t->d_cnt,
             t->h_ring, {t->ring_ev[0], t->ring_ev[1], t->ring_ev[2], t->ring_ev[3]}, std::move(t->ext_ev), t->ext_used, t->ext_ms_acc, t->ext_launches};
    *t = *c;
    t->stream = own.stream; t->d_counter = own.d_counter; t->d_stats = own.d_stats; t->d_tiles = own.d_tiles; t->ev0 = own.ev0; t->ev1 = own.ev1; t->stats = own.stats;
    t->pool = own.pool; t->pool_b = own.pool_b; t->pool_mem = own.pool_mem; t->pool_alloc = own.pool_alloc; t->d_sort = own.d_sort; t->sort_temp_bytes = own.sort_temp_bytes;
    t->d_qlive = own.d_qlive; t->d_cnt = own.d_cnt; t->h_ring = own.h_ring; for (int i = 0; i < 4; ++i) t->ring_ev[i] = own.ring_ev[i];
    t->ext_ev = std::move(own.ext_ev); t->ext_used = own.ext_used; t->ext_ms_acc = own.ext_ms_acc; t->ext_launches = own.ext_launches;
    t->twin = nullptr; t->is_twin = true; t->twin_go = nullptr; t->twin_done = nullptr; t->pipelines = 1;
}
// One render call.  Over a tree beyond the caches the wavefront stages run as two pipelines side by side (NoriHipContext::pipelines): the
// first half of the sample range here, the second half on the twin's stream, driven by a thread of its own.  A stage of one pipeline then
// runs while the other pipeline is between stages or in a stage that leaves most of the machine idle (the shade stage and the ray sort are
// bound by memory, the trace stage's last waves by their longest rays).  Every sample is the same sample as in one pipeline (its random
// numbers depend on pixel and sample index only); the film is shared and summed by atomics as before.  The context's stream waits for the
// twin's, so for the caller the call completes in stream order as it always did.
static int render_entry(NoriHipContext *ctx, const int *tiles_xyxy, int n_tiles, int spp_begin, int spp_end) {
    SCENECHK(ctx);
    const bool big_tree = tree_beyond_caches(ctx);
    int np = ctx->pipelines > 0 ? ctx->pipelines : (big_tree ? 2 : 1);
    if (ctx->pipelines <= 0 && !big_tree && !ctx->ext && tiles_xyxy && n_tiles >= 1 && n_tiles <= ND_MAX_TILES && spp_end > spp_begin) {
        // a scene that lives in the caches: two pipelines pay from about a million samples per call on (finding 42)
        unsigned long long total = 0;
        for (int k = 0; k < n_tiles; ++k) {
            const long long w_ = (long long) tiles_xyxy[4 * k + 2] - tiles_xyxy[4 * k], h_ = (long long) tiles_xyxy[4 * k + 3] - tiles_xyxy[4 * k + 1];
            if (w_ > 0 && h_ > 0) total += (unsigned long long) (w_ * h_) * (unsigned long long) (spp_end - spp_begin);
        }
        if (total >= ctx->twin_min_samples) np = 2;
    }
    if (np < 2 || ctx->is_twin || !uses_stages(ctx) || spp_end <= spp_begin || spp_begin < 0 || !tiles_xyxy || n_tiles < 1) return render_impl(ctx, tiles_xyxy, n_tiles, spp_begin, spp_end);
    // How the call is halved: by samples; a call of one sample per pixel (a progressive display) by pixels instead - the two halves of its
    // tile list, or the upper and lower half of its one tile, cut at a multiple of the 8-pixel patches the work items are dealt in.
    const bool by_spp = spp_end - spp_begin >= 2;
    int row_tiles[8]; const int *tiles_1 = tiles_xyxy, *tiles_2 = tiles_xyxy; int n_1 = n_tiles, n_2 = n_tiles;
    if (!by_spp) {
        if (n_tiles >= 2) { n_1 = n_tiles / 2; n_2 = n_tiles - n_1; tiles_2 = tiles_xyxy + 4 * (size_t) n_1; }
        else {
            const int y0 = tiles_xyxy[1], y1 = tiles_xyxy[3], ym = y0 + ((((y1 - y0) / 2) + 7) & ~7);
            if (y1 - y0 < 16 || ym >= y1) return render_impl(ctx, tiles_xyxy, n_tiles, spp_begin, spp_end);
            for (int k = 0; k < 4; ++k) row_tiles[k] = row_tiles[4 + k] = tiles_xyxy[k];
            row_tiles[3] = ym; row_tiles[5] = ym;
            tiles_1 = row_tiles; tiles_2 = row_tiles + 4; n_1 = n_2 = 1;
        }
    }
    if (!ctx->twin) {
        NoriHipContext *t = new NoriHipContext();
        t->device = ctx->device; t->is_twin = true;
        const int rc = create_impl(t, ctx->device);
        if (rc != NORI_HIP_OK) { const std::string why = g_err; nori_hip_destroy(t); g_err = why; return rc; }
        (void) hipStreamDestroy(t->stream); t->stream = nullptr;            // (the twin's stream lives for the duration of a call, see below)
        ctx->twin = t;
    }
    if (!ctx->twin_go) { HIPCHK(hipEventCreateWithFlags(&ctx->twin_go, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&ctx->twin_done, hipEventDisableTiming)); }
    NoriHipContext *t = ctx->twin;
    // The second stream is made for the call and given back after it: a stream holds a place on one of the process's hardware queues
    // (GPU_MAX_HW_QUEUES, 4 by default), the runtime puts a new stream on the least used one, and two pipelines on one queue run one
    // after the other - a context that kept its second stream while idle made the next context's land on its partner's queue
    // (scratch/s2_queues.py: cbox 23.4 ms per step alone, 30.3 as the second context of a process).
    if (!t->stream) HIPCHK(hipStreamCreate(&t->stream));
    HIPCHK(hipEventRecord(ctx->twin_go, ctx->stream));                    // what the caller queued before this call (a film clear, say) comes first for both
    HIPCHK(hipStreamWaitEvent(t->stream, ctx->twin_go, 0));
    // (no early return from here to the end: the per-call settings below are taken back and the second stream is given back on every path)
    ctx->twin_trace_bpc = big_tree ? 0 : 3; ctx->twin_shade_waves = (big_tree || ctx->ext) ? 0 : 5;
    twin_borrow(t, ctx);
    const int mid = spp_begin + (spp_end - spp_begin + 1) / 2;             // (unequal halves measured worse: 56 / 62 / 75 % here: +1 / +6 / +7 % per step)
    const int b_1 = spp_begin, e_1 = by_spp ? mid : spp_end, b_2 = by_spp ? mid : spp_begin, e_2 = spp_end;
    // The two pipelines reach their tails within a millisecond of each other.  Two tail kernels side by side get in each other's way (48 and
    // 65 ms where one launch over all the paths takes 48): each pipeline hands its live list back instead, and ONE launch finishes both.
    const bool merged_tail = ctx->twin_tail_merged;
    TailHandoff h1, h2;
    int rc2 = NORI_HIP_OK; std::string err2;
    auto second = [&]() { rc2 = render_impl(t, tiles_2, n_2, b_2, e_2, merged_tail ? &h2 : nullptr); if (rc2 != NORI_HIP_OK) err2 = g_err; };
    std::thread th; bool threaded = true;
    try { th = std::thread(second); } catch (...) { threaded = false; }     // (no thread to be had: the halves are queued one after the other, from this one)
    int rc1 = render_impl(ctx, tiles_1, n_1, b_1, e_1, merged_tail ? &h1 : nullptr);
    std::string err1 = rc1 != NORI_HIP_OK ? g_err : std::string();
    if (threaded) th.join(); else second();
    (void) hipSetDevice(ctx->device);
    if (hipEventRecord(ctx->twin_done, t->stream) == hipSuccess) (void) hipStreamWaitEvent(ctx->stream, ctx->twin_done, 0);
    else { (void) hipGetLastError(); (void) hipStreamSynchronize(t->stream); }
    if (rc1 == NORI_HIP_OK && rc2 == NORI_HIP_OK && (h1.valid || h2.valid)) {
        const TailHandoff &a = h1.valid ? h1 : h2;
        const bool both = h1.valid && h2.valid;
        rc1 = launch_finish(ctx, ctx->stream, a.wide, a.wide8, a.big_tree, a.pool, a.q_live, a.cnt, a.n_upper, both ? h2.pool : a.pool, both ? h2.q_live : nullptr, both ? h2.cnt : nullptr, both ? h2.n_upper : 0u);
        if (rc1 != NORI_HIP_OK) err1 = g_err;
    }
    (void) hipEventRecord(ctx->ev1, ctx->stream);                         // render_ms covers both pipelines
    ctx->stats.pipelines = 2;
    ctx->twin_trace_bpc = 0; ctx->twin_shade_waves = 0;
    (void) hipStreamDestroy(t->stream); t->stream = nullptr;                // (work still in flight on it completes; ctx->stream waits for it through twin_done)
    ctx->trace_width = std::max(ctx->trace_width, t->trace_width); ctx->trace_node_bytes = std::max(ctx->trace_node_bytes, t->trace_node_bytes);
    if (rc1 != NORI_HIP_OK) return fail(rc1, err1);
    if (rc2 != NORI_HIP_OK) return fail(rc2, err2);
    return NORI_HIP_OK;
}
int nori_hip_render(NoriHipContext *ctx, int x0, int y0, int x1, int y1, int spp_begin, int spp_end) {
    const int t[4] = {x0, y0, x1, y1};
    return render_entry(ctx, t, 1, spp_begin, spp_end);
}
int nori_hip_render_tiles(NoriHipContext *ctx, const int *tiles_xyxy, int n_tiles, int spp_begin, int spp_end) {
    return render_entry(ctx, tiles_xyxy, n_tiles, spp_begin, spp_end);
}

static int render_impl(NoriHipContext *ctx, const int *tiles_xyxy, int n_tiles, int spp_begin, int spp_end, TailHandoff *tail) {
    SCENECHK(ctx);
    const int W = ctx->sc.cam.width, H = ctx->sc.cam.height;
    if (!tiles_xyxy || n_tiles < 1 || n_tiles > ND_MAX_TILES) return fail(NORI_HIP_EINVAL, "tile list is null, empty or longer than " + std::to_string(ND_MAX_TILES));
    if (spp_begin < 0 || spp_end <= spp_begin) return fail(NORI_HIP_EINVAL, "empty sample range");
    if (ctx->sc.n_emitters == 0 && !ctx->sc.env.texture && !(ctx->integrator == NORI_HIP_INTEGRATOR_PATH_MATS)) return fail(NORI_HIP_ESTATE, "path_mis needs at least one emitter");
    RenderJob job; job.spp_begin = spp_begin; job.nspp = spp_end - spp_begin; job.total = 0; job.n_tiles = n_tiles;
    job.tiles = nullptr; job.patch_first = nullptr;
    std::vector<int> h_tiles((size_t) 4 * n_tiles); std::vector<unsigned int> h_first((size_t) n_tiles + 1, 0u);
    unsigned long long patches = 0;
    for (int k = 0; k < n_tiles; ++k) {
        const int x0 = tiles_xyxy[4 * k], y0 = tiles_xyxy[4 * k + 1], x1 = tiles_xyxy[4 * k + 2], y1 = tiles_xyxy[4 * k + 3];
        if (x0 < 0 || y0 < 0 || x1 > W || y1 > H || x0 >= x1 || y0 >= y1) return fail(NORI_HIP_EINVAL, "tile " + std::to_string(k) + " lies outside the image or is empty");
        h_tiles[4 * k] = x0; h_tiles[4 * k + 1] = y0; h_tiles[4 * k + 2] = x1 - x0; h_tiles[4 * k + 3] = y1 - y0;
        h_first[k] = (unsigned int) patches;
        patches += (unsigned long long) ((x1 - x0 + 7) >> 3) * (unsigned long long) ((y1 - y0 + 7) >> 3);
        job.total += (unsigned long long) (x1 - x0) * (y1 - y0) * job.nspp;
    }
    if (patches > 0xffffffffull) return fail(NORI_HIP_EINVAL, "tile list covers too many pixels");
    h_first[n_tiles] = (unsigned int) patches;
    job.x0 = h_tiles[0]; job.y0 = h_tiles[1]; job.tw = h_tiles[2]; job.th = h_tiles[3];
    job.padded_total = patches * (unsigned long long) job.nspp * 64ull;
    if (n_tiles > 1) {
        // (tiles must not overlap if every sample is to be rendered once; the library does not check that)
        HIPCHK(hipStreamSynchronize(ctx->stream));                            // the previous call may still read the table
        HIPCHK(hipMemcpyAsync(ctx->d_tiles, h_tiles.data(), h_tiles.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(ctx->d_tiles + 4 * ND_MAX_TILES, h_first.data(), h_first.size() * sizeof(unsigned int), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));                            // h_tiles / h_first are locals
        job.tiles = reinterpret_cast<const int4 *>(ctx->d_tiles); job.patch_first = reinterpret_cast<const unsigned int *>(ctx->d_tiles + 4 * ND_MAX_TILES);
    }
    HIPCHK(hipMemsetAsync(ctx->d_counter, 0, sizeof(unsigned long long), ctx->stream));
    ctx->trace_width = 0; ctx->trace_node_bytes = 0; ctx->stats.pipelines = 1;
    const bool mats = ctx->integrator == NORI_HIP_INTEGRATOR_PATH_MATS, direct = ctx->integrator == NORI_HIP_INTEGRATOR_DIRECT_MIS;
    if (!uses_stages(ctx)) {
        // single persistent kernel with per-lane path regeneration
        unsigned long long chunks = (job.total + MEGA_THREADS - 1) / MEGA_THREADS;
        const int MW = ctx->mega_waves;
        unsigned blocks = (unsigned) ctx->n_cus * (unsigned) MW;                   // MW waves per SIMD = MW workgroups of 4 waves per CU
        if ((unsigned long long) blocks > chunks) blocks = (unsigned) chunks;
        if (blocks == 0) blocks = 1;
        const size_t stack_bytes = (size_t) (MEGA_THREADS / 64) * MEGA_DEPTH * 64 * 8;
        const size_t share = std::min<size_t>(ctx->lds_bytes, (size_t) 163840 / (size_t) MW) & ~(size_t) 1023;
        const unsigned int n_cached = share > stack_bytes ? std::min<unsigned int>(ctx->sc.n_top, (unsigned int) ((share - stack_bytes) / 64)) : 0u;
        const size_t lds = stack_bytes + (size_t) n_cached * 64;
        void (*kern)(DevScene, RenderJob, float4 *, unsigned long long *, DevStats *, unsigned int) =
            direct ? k_render_direct : mats ? k_render_mats : (ctx->ext && !ctx->has_ss) ? k_render_img : ctx->ext ? k_render_ext : MW == 3 ? k_render_w3 : MW == 4 ? k_render_w4 : MW == 5 ? k_render_w5 : MW == 6 ? k_render_w6 : k_render_w8;
        HIPCHK(hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
        HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(MEGA_THREADS), lds, ctx->stream, ctx->sc, job, ctx->d_film, ctx->d_counter, ctx->d_stats, n_cached);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
        ctx->stats.iterations = 1;
        return NORI_HIP_OK;
    }
    // ---- wavefront stages over a pool of path slots
    // The pool is sized for the call (three eighths of its paths - cbox 512 x 512 x 64: 25.0 ms with a quarter, 24.1 with three eighths, 24.5
    // with a half - between 1 M and 8 M slots, or NORI_HIP_POOL) and allocated when a
    // call needs more slots than the context holds; pool, half-size pool and live list are committed to the context together.
    // "big tree": nodes and records against the Infinity Cache (256 MiB).  Beyond it the trace stage waits for HBM lines instead of
    // issuing, and a few defaults change with that (measured on the 10 M-triangle workload): rays are ordered before every trace
    // launch, no speculative traversal, no second box step per trip, and half instead of three eighths of the call's paths in the pool
    // (fewer, larger launches: the stage's ramp-up and drain are paid less often).
    const bool big_tree = tree_beyond_caches(ctx);
    const unsigned int pool_cap = ctx->pool_slots ? ctx->pool_slots : (1u << 23);
    unsigned int P = pool_cap;
    if (!ctx->pool_slots) P = (unsigned int) std::min<unsigned long long>(pool_cap, std::max<unsigned long long>(1ull << 20, ((big_tree ? job.total / 2 : ctx->has_ss ? job.total / 4 : job.total * 3 / 8) + 65535) & ~65535ull));     // (scenes with subsurface shapes: 13 pool arrays instead of 10, a quarter measured better: 89.8 against 92.7 ms)
    if ((unsigned long long) P > ((job.total + 63) / 64) * 64) P = (unsigned int) (((job.total + 63) / 64) * 64);
    if (!ctx->pool_mem || ctx->pool_alloc < P) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        hipFree(ctx->pool_mem); hipFree(ctx->d_qlive); ctx->pool_mem = nullptr; ctx->d_qlive = nullptr; ctx->pool_alloc = 0;
        const size_t PA = ((size_t) P + 63) / 64 * 64, PB = ((PA / 2 + 63) / 64) * 64 + 64;
        DevBufs guard; float4 *mem = nullptr; unsigned int *ql = nullptr;
        HIPCHK(guard.get(mem, (PA + PB) * ND_POOL_ARRAYS * sizeof(float4)));
        HIPCHK(guard.get(ql, PA * 4));
        // ray ordering (k_ray_keys + radix sort): two key and two slot-number arrays and the sort's workspace
        hipFree(ctx->d_sort); ctx->d_sort = nullptr;
        unsigned int *srt = nullptr; const size_t tb = (lbvh::sort_pairs_temp_bytes((uint32_t) PA) + 255) & ~(size_t) 255;
        HIPCHK(guard.get(srt, PA * 16 + tb));
        ctx->d_sort = srt; ctx->sort_temp_bytes = tb;
        hipLaunchKernelGGL(k_iota, dim3((unsigned) ((PA + 255) / 256)), dim3(256), 0, ctx->stream, srt + 2 * PA, (unsigned int) PA);
        guard.owned.clear();                                                   // all succeeded: the context owns them now
        ctx->pool_mem = mem; ctx->d_qlive = ql; ctx->pool_alloc = (unsigned int) PA;
        auto carve = [](Pool &pl, float4 *base, size_t n) {
            pl.ray_o = base; pl.ray_d = base + n; pl.hit = (uint4 *) (base + 2 * n); pl.thr = base + 3 * n; pl.li = base + 4 * n;
            pl.rng = (uint4 *) (base + 5 * n); pl.pix = base + 6 * n; pl.sh_o = base + 7 * n; pl.sh_d = base + 8 * n; pl.sh_c = base + 9 * n;
            pl.sv_o = base + 10 * n; pl.sv_d = base + 11 * n; pl.sv_h = (uint4 *) (base + 12 * n);
        };
        carve(ctx->pool, mem, PA); carve(ctx->pool_b, mem + ND_POOL_ARRAYS * PA, PB);
    }
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    WfCounters *c = ctx->d_cnt;
    hipLaunchKernelGGL(k_pool_init, dim3((P + 255) / 256), dim3(256), 0, ctx->stream, ctx->pool, P, c);
    const int3 thr = make_int3(ctx->thr_box, ctx->thr_prim, ctx->thr_refill);
    // (trees beyond the caches: the tail kernel takes over a little earlier, 5/16 of the pool - C4 286 -> 283 ms)
    // ... and never later than at a million live paths: a stage launch lasts as long as its longest rays however few slots it sweeps, so a small
    // call (or a small rest of a large one) is better off in the tail kernel from the start - C4 at 1 / 2 / 4 spp per call 55.0 / 72.9 / 114.6 ->
    // 39.3 / 60.7 / 103.1 ms, cbox 512 x 512 at 4 / 16 spp 4.80 / 8.97 -> 4.19 / 8.38, dragon at 4 spp 9.14 -> 8.49; the full workloads are
    // unchanged (their fraction of the pool is above a million, or equal within noise: cbox at 64 spp 23.5 / 23.4)
    const unsigned int finish_below = std::min<unsigned int>(ctx->finish_below ? ctx->finish_below : std::max<unsigned int>(big_tree ? P / 16 * 5 : P / 4, 1u << 20), P);
    // the trace stages give every wave a static share of the pool, so all their workgroups must be resident at once.
    // LDS of a workgroup: stacks + staging buffers of its waves, the rest of its share caches the top of the tree.
    const unsigned trace_threads = (unsigned) ctx->trace_threads, trace_nw = trace_threads / 64;
    const bool wide = ctx->wide && ctx->d_nodes4 != nullptr && ctx->merge_trace;          // the merged trace stage over four-box nodes
    // the scene's tree and records against the 256 MiB of Infinity Cache: beyond that, ordering the rays pays for its sort
    const bool sort_rays = ctx->sort_rays < 0 ? big_tree : ctx->sort_rays != 0;
    const bool wide8 = wide && ctx->d_nodes8 != nullptr && !ctx->has_ss;                  // ... over quantised eight-box nodes
    ctx->trace_width = wide8 ? 8 : wide ? 4 : 2;
    ctx->trace_node_bytes = wide8 ? 128 : wide ? 128 : 64;
    const int trace_waves = wide8 ? ND_WIDE8_WAVES : wide ? ND_WIDE_WAVES : 6;            // waves per SIMD the kernel is built for
    const size_t trace_fixed = wide8 ? ctx->trace_lds_wide8 : wide ? ctx->trace_lds_wide : ctx->trace_lds;               // stacks + staged rays (origin, direction, reciprocal direction, slot)
    int trace_bpc = ctx->trace_blocks_per_cu;
    if (trace_bpc == 0 && ctx->twin_trace_bpc > 0 && trace_nw == 4) trace_bpc = ctx->twin_trace_bpc;
    if (trace_bpc == 0) trace_bpc = std::max(1, std::min<int>((int) (trace_waves * 4 / trace_nw), (int) (163840 / trace_fixed)));
    const int trace_resident = std::min<int>(trace_bpc, std::max<int>(1, (int) (trace_waves * 4 / trace_nw)));     // more workgroups than that run in rounds
    const size_t trace_share = (std::min<size_t>(ctx->lds_bytes, (size_t) 163840 / (size_t) trace_resident)) & ~(size_t) 255;
    if (trace_share < trace_fixed) return fail(NORI_HIP_EINVAL, "trace stage: workgroup LDS share too small for its stacks");
    const unsigned int xcd_chunks = (unsigned int) (ctx->xcd_chunks >= 0 ? ctx->xcd_chunks : 64);             // sorted rays: chunks of the sorted order per XCD (0: plain round-robin deal)
    const unsigned int trace_box2 = (unsigned int) (ctx->box2 >= 0 ? ctx->box2 : 36) | ((unsigned int) (ctx->dyn_pct >= 0 ? ctx->dyn_pct : big_tree ? 25 : 15) << 8) | ((unsigned int) ctx->dyn_tries << 16) | (xcd_chunks << 20);
    const bool spec = ctx->spec >= 0 ? ctx->spec != 0 : !big_tree;
    const size_t trace_lds = trace_fixed;
    if (ctx->profile) fprintf(stderr, "[nori_hip profile] trace stage: %u threads x %d workgroups per CU, %zu B LDS, second box step from %u lanes\n", trace_threads, trace_bpc, trace_lds, trace_box2);
    unsigned trace_blocks = 1, shade_blocks = 1; size_t shade_lds = 0;
    auto shade_run = [](unsigned int slots, unsigned int blocks) { return (((slots + blocks - 1) / blocks) + 255u) & ~255u; };   // = per_block in shade_body
    auto size_grids = [&](unsigned int slots) {
        trace_blocks = std::max(1u, std::min<unsigned>((slots + trace_threads - 1) / trace_threads, (unsigned) ctx->n_cus * (unsigned) trace_bpc));
        if (sort_rays && trace_blocks > 8u) trace_blocks &= ~7u;             // the XCD-aware deal of sorted rays wants the same number of workgroups on every XCD
        shade_blocks = std::max(1u, std::min<unsigned>((slots + 255) / 256, (unsigned) ctx->n_cus * (unsigned) ctx->shade_blocks_per_cu));
        if (shade_run(slots, shade_blocks) > SHADE_RUN_MAX) shade_blocks = (slots + (SHADE_RUN_MAX - 256u) - 1) / (SHADE_RUN_MAX - 256u);   // 16-bit offsets in the run lists
        shade_lds = (size_t) shade_run(slots, shade_blocks) * sizeof(unsigned short);
    };
    size_grids(P);
    Pool pools[2] = {ctx->pool, ctx->pool_b}; int pi = 0;                      // the pool in use; P shrinks with every compaction
    const int SW = (!ctx->shade_waves_given && ctx->twin_shade_waves > 0) ? ctx->twin_shade_waves : ctx->shade_waves;
    void (*shade_kern)(DevScene, Pool, unsigned int, RenderJob, float4 *, unsigned int *, WfCounters *, unsigned long long *, DevStats *, unsigned int *) =
        ctx->ext ? k_shade_ext : SW == 3 ? k_shade_w3 : SW == 4 ? k_shade_w4 : SW == 5 ? k_shade_w5 : SW == 6 ? k_shade_w6 : k_shade_w8;
    // the shade stage writes the sort keys itself where a variant for that exists (the default ones); otherwise k_ray_keys does
    const bool fused_keys = wide && sort_rays && ctx->fused_keys && (ctx->ext || SW == 4);
    if (fused_keys) shade_kern = ctx->ext ? k_shade_ext_keys : k_shade_w4_keys;
    unsigned int n_upper = P;              // upper bound of the live-slot count (a ring value that is three iterations old)
    uint64_t it = 0;
    for (;; ++it) {
        // the live count of iteration it - 1 goes to the host ring here (k_iter_begin), its event right behind
        hipLaunchKernelGGL(k_iter_begin, dim3(1), dim3(1), 0, ctx->stream, c, it > 0 ? ctx->h_ring + ((it - 1) & 3) : (unsigned int *) nullptr);
        if (it > 0) HIPCHK(hipEventRecord(ctx->ring_ev[(it - 1) & 3], ctx->stream));
        hipLaunchKernelGGL(shade_kern, dim3(shade_blocks), dim3(256), shade_lds, ctx->stream, ctx->sc, pools[pi], P, job, ctx->d_film, ctx->d_qlive, c, ctx->d_counter, ctx->d_stats,
                           fused_keys ? ctx->d_sort : (unsigned int *) nullptr);
        {
            if (ctx->ext_used + 2 > ctx->ext_ev.size()) {
                if (ctx->ext_ev.size() >= 4096) { int rc2 = ctx->fold_ext_events(); if (rc2) return rc2; }
                else { hipEvent_t a_, b_; HIPCHK(hipEventCreate(&a_)); HIPCHK(hipEventCreate(&b_)); ctx->ext_ev.push_back(a_); ctx->ext_ev.push_back(b_); }
            }
// (the cursors of the dynamically handed-out batches are reset once per iteration, by k_iter_begin: only the merged
            // stage, one launch per iteration, uses them)
#define ND_TRACE(K, PR) hipLaunchKernelGGL((k_trace_pool<K, PR>), dim3(trace_blocks), dim3(trace_threads), trace_lds, ctx->stream, ctx->sc, pools[pi], P, ctx->d_stats, thr, ((K) == TK_BOTH ? trace_box2 : (trace_box2 & 0xffu)), c)
            const unsigned int *order = nullptr;
            if (wide && sort_rays) {
                unsigned int *k0 = ctx->d_sort, *k1 = k0 + ctx->pool_alloc, *v0 = k1 + ctx->pool_alloc, *v1 = v0 + ctx->pool_alloc;
                // (keys: written by the shade stage; slot numbers: 0, 1, 2, ... since the pool was allocated)
                if (!fused_keys) hipLaunchKernelGGL(k_ray_keys, dim3((P + 255) / 256), dim3(256), 0, ctx->stream, ctx->sc, pools[pi], P, k0, v0);
                // (the order only has to gather the rays of a region: the top 16 of the key's 30 bits do as well as all of them, in two radix passes instead of four)
                HIPCHK(lbvh::sort_pairs(ctx->stream, v1 + ctx->pool_alloc, ctx->sort_temp_bytes, k0, k1, v0, v1, P, 30, 30 - ctx->sort_bits));
                order = v1;
            }
            if (ctx->merge_trace) {
                HIPCHK(hipEventRecord(ctx->ext_ev[ctx->ext_used], ctx->stream));
                if (wide && ctx->has_ss) {
                    hipLaunchKernelGGL(k_trace_pool4_masked, dim3(trace_blocks), dim3(trace_threads), trace_lds, ctx->stream, ctx->sc, pools[pi], P, ctx->d_stats, thr, trace_box2, c, order);
                } else if (wide8) {
                    void (*k8)(DevScene, Pool, unsigned int, DevStats *, int3, unsigned int, WfCounters *, const unsigned int *) =
                        ctx->profile ? (spec ? k_trace_pool8<true, true> : k_trace_pool8<true, false>) : (spec ? k_trace_pool8<false, true> : k_trace_pool8<false, false>);
                    hipLaunchKernelGGL(k8, dim3(trace_blocks), dim3(trace_threads), trace_lds, ctx->stream, ctx->sc, pools[pi], P, ctx->d_stats, thr, trace_box2, c, order);
                } else if (wide && !spec && !ctx->profile && ctx->sc.leaf2 && ctx->leaf2_step && ctx->sc.nodes4q) {
                    ctx->trace_node_bytes = 64;
                    hipLaunchKernelGGL(k_trace_pool4q_leaf2, dim3(trace_blocks), dim3(trace_threads), trace_lds, ctx->stream, ctx->sc, pools[pi], P, ctx->d_stats, thr, trace_box2, c, order);
                } else if (wide && !spec && !ctx->profile && ctx->sc.leaf2 && ctx->leaf2_step) {
                    hipLaunchKernelGGL(k_trace_pool4_leaf2, dim3(trace_blocks), dim3(trace_threads), trace_lds, ctx->stream, ctx->sc, pools[pi], P, ctx->d_stats, thr, trace_box2, c, order);
                } else if (wide) {
                    void (*k4)(DevScene, Pool, unsigned int, DevStats *, int3, unsigned int, WfCounters *, const unsigned int *) =
                        ctx->profile ? (spec ? k_trace_pool4<true, true> : k_trace_pool4<true, false>) : (spec ? k_trace_pool4<false, true> : k_trace_pool4<false, false>);
                    hipLaunchKernelGGL(k4, dim3(trace_blocks), dim3(trace_threads), trace_lds, ctx->stream, ctx->sc, pools[pi], P, ctx->d_stats, thr, trace_box2, c, order);
                } else if (spec) {
                    if (ctx->profile) hipLaunchKernelGGL((k_trace_pool<TK_BOTH, true, true>), dim3(trace_blocks), dim3(trace_threads), trace_lds, ctx->stream, ctx->sc, pools[pi], P, ctx->d_stats, thr, trace_box2, c);
                    else hipLaunchKernelGGL((k_trace_pool<TK_BOTH, false, true>), dim3(trace_blocks), dim3(trace_threads), trace_lds, ctx->stream, ctx->sc, pools[pi], P, ctx->d_stats, thr, trace_box2, c);
                } else if (ctx->profile) ND_TRACE(TK_BOTH, true); else ND_TRACE(TK_BOTH, false);
                HIPCHK(hipEventRecord(ctx->ext_ev[ctx->ext_used + 1], ctx->stream));
            } else {
                if (ctx->profile) ND_TRACE(TK_SHADOW, true); else ND_TRACE(TK_SHADOW, false);
                HIPCHK(hipEventRecord(ctx->ext_ev[ctx->ext_used], ctx->stream));
                if (ctx->profile) ND_TRACE(TK_EXTEND, true); else ND_TRACE(TK_EXTEND, false);
                HIPCHK(hipEventRecord(ctx->ext_ev[ctx->ext_used + 1], ctx->stream));
            }
#undef ND_TRACE
            ctx->ext_used += 2;
        }
        HIPCHK(hipGetLastError());
        // the live count comes back through a 4-deep ring of pinned words, so the host runs at most three iterations
        // ahead of the device and never waits on the iteration it has just queued
        if (it >= 3) {
            HIPCHK(hipEventSynchronize(ctx->ring_ev[(it - 3) & 3]));
            n_upper = ctx->h_ring[(it - 3) & 3];
            if (n_upper == 0) break;
            if (n_upper < finish_below) {
                // fewer live paths than the threshold (which also means the work counter is exhausted): finish them in one launch -
                // or leave them to the caller, who finishes both pipelines' paths together (render_entry)
                if (tail) { tail->valid = true; tail->pool = pools[pi]; tail->q_live = ctx->d_qlive; tail->cnt = c; tail->n_upper = n_upper; tail->wide = wide; tail->wide8 = wide8; tail->big_tree = big_tree; }
                else { const int rcf = launch_finish(ctx, ctx->stream, wide, wide8, big_tree, pools[pi], ctx->d_qlive, c, n_upper, pools[pi], nullptr, nullptr, 0u); if (rcf != NORI_HIP_OK) return rcf; }
                HIPCHK(hipGetLastError());
                ++it;
                break;
            }
            if (ctx->compact && n_upper < P / 2 && P >= (1u << 17)) {
                // (live < P means the work counter is exhausted, so the count only falls from here: the value of three
                // iterations ago bounds the one the kernel will read)
                const unsigned int P2 = ((P / 2 + 63u) / 64u) * 64u;
                hipLaunchKernelGGL(k_compact_pool, dim3((P2 + 255) / 256), dim3(256), 0, ctx->stream, pools[pi], pools[pi ^ 1], (const unsigned int *) ctx->d_qlive, (const WfCounters *) c, P2);
                HIPCHK(hipGetLastError());
                pi ^= 1; P = P2; size_grids(P);
            }
        }
        if (it > (1ull << 24)) return fail(NORI_HIP_EDEVICE, "wavefront loop did not terminate");
    }
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    ctx->stats.iterations = it + 1;
    return NORI_HIP_OK;
}

int nori_hip_clear_film(NoriHipContext *ctx) {
    SCENECHK(ctx);
    HIPCHK(hipMemsetAsync(ctx->d_film, 0, (size_t) ctx->film_rows * ctx->film_cols * sizeof(float4), ctx->stream));
    return NORI_HIP_OK;
}
int nori_hip_film_dims(NoriHipContext *ctx, int *rows, int *cols, int *border) {
    SCENECHK(ctx);
    if (rows) *rows = ctx->film_rows; if (cols) *cols = ctx->film_cols; if (border) *border = ctx->border;
    return NORI_HIP_OK;
}
int nori_hip_read_film(NoriHipContext *ctx, float *film) {
    SCENECHK(ctx);
    if (!film) return fail(NORI_HIP_EINVAL, "null buffer");
    HIPCHK(hipMemcpyAsync(film, ctx->d_film, (size_t) ctx->film_rows * ctx->film_cols * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return NORI_HIP_OK;
}
int nori_hip_write_film(NoriHipContext *ctx, const float *film) {
    SCENECHK(ctx);
    if (!film) return fail(NORI_HIP_EINVAL, "null buffer");
    HIPCHK(hipMemcpyAsync(ctx->d_film, film, (size_t) ctx->film_rows * ctx->film_cols * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return NORI_HIP_OK;
}
int nori_hip_film_device_ptr(NoriHipContext *ctx, void **d_film, size_t *bytes) {
    SCENECHK(ctx);
    if (d_film) *d_film = ctx->d_film;
    if (bytes) *bytes = (size_t) ctx->film_rows * ctx->film_cols * sizeof(float4);
    return NORI_HIP_OK;
}
int nori_hip_read_bitmap(NoriHipContext *ctx, float *rgb) {
    SCENECHK(ctx);
    if (!rgb) return fail(NORI_HIP_EINVAL, "null buffer");
    const int W = ctx->sc.cam.width, H = ctx->sc.cam.height;
    DevBufs bufs; float *d = nullptr; HIPCHK(bufs.get(d, (size_t) W * H * 12));
    hipLaunchKernelGGL(k_bitmap, dim3((W + 15) / 16, (H + 15) / 16), dim3(16, 16), 0, ctx->stream, (const float4 *) ctx->d_film, W, H, ctx->border, ctx->film_cols, d);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(rgb, d, (size_t) W * H * 12, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return NORI_HIP_OK;
}

int nori_hip_filter(NoriHipContext *ctx, int r, int f, float sigma, float h, float *out) {
    SCENECHK(ctx);
    if (!out) return fail(NORI_HIP_EINVAL, "null buffer");
    if (r < 0 || f < 0 || !(sigma > 0) || !(h > 0)) return fail(NORI_HIP_EINVAL, "bad filter parameters");
    const int W = ctx->sc.cam.width, H = ctx->sc.cam.height, R = r + f;
    const char *fenv = getenv("NORI_HIP_FILTER");
    const bool direct_early = fenv && std::string(fenv) == "direct";
    size_t lds = ((size_t) (NLM_BW + 2 * R) * (NLM_BH + 2 * R) * 3 + (size_t) (2 * f + 1) * (2 * f + 1)) * sizeof(float);
    if (direct_early && lds > std::min<size_t>(ctx->lds_bytes, 160 * 1024)) return fail(NORI_HIP_EINVAL, "filter window + patch too large for the LDS tile (r + f <= 48)");
    size_t n = (size_t) W * H * 3;
    DevBufs bufs; float *d_lin = nullptr, *d_srgb = nullptr, *d_out = nullptr;
    HIPCHK(bufs.get(d_lin, n * 4)); HIPCHK(bufs.get(d_srgb, n * 4)); HIPCHK(bufs.get(d_out, n * 4));
    hipLaunchKernelGGL(k_bitmap, dim3((W + 15) / 16, (H + 15) / 16), dim3(16, 16), 0, ctx->stream, (const float4 *) ctx->d_film, W, H, ctx->border, ctx->film_cols, d_lin);
    hipLaunchKernelGGL(k_srgb, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, ctx->stream, (const float *) d_lin, n, d_srgb);
    const bool direct = direct_early;
    const size_t lds_sep = ((size_t) (NLM_BW + 2 * R) * (NLM_BH + 2 * R) * 3 + (size_t) (NLM_BW + 2 * f) * (NLM_BH + 2 * f) + (size_t) (NLM_BH + 2 * f) * NLM_BW + (size_t) (2 * f + 1)) * sizeof(float);
    if (!direct && lds_sep > std::min<size_t>(ctx->lds_bytes, 160 * 1024)) return fail(NORI_HIP_EINVAL, "filter window + patch too large for the LDS tile");
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    if (direct)
        hipLaunchKernelGGL(k_nlmeans, dim3((W + NLM_BW - 1) / NLM_BW, (H + NLM_BH - 1) / NLM_BH), dim3(NLM_BW, NLM_BH), lds, ctx->stream,
                           (const float *) d_srgb, d_out, W, H, r, f, sigma, h);
    else
        hipLaunchKernelGGL(k_nlmeans_sep, dim3((W + NLM_BW - 1) / NLM_BW, (H + NLM_BH - 1) / NLM_BH), dim3(NLM_BW, NLM_BH), lds_sep, ctx->stream,
                           (const float *) d_srgb, d_out, W, H, r, f, sigma, h);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    HIPCHK(hipMemcpyAsync(out, d_out, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return NORI_HIP_OK;
}

// nori_hip_filter for callers that cannot allocate: no hipMalloc / hipFree, no copy to the host, no synchronisation.  The
// intermediate image lives in this GPU's `scratch` array (ctx->scratch_ptr), every pixel of which is rewritten before it is
// read, so nothing needs clearing between calls; calls that overlap on one GPU (two contexts, two threads) race on it.
int nori_hip_filter_noalloc(NoriHipContext *ctx, int r, int f, float sigma, float h, float *d_rgb_srgb_out) {
    SCENECHK(ctx);
    if (!d_rgb_srgb_out) return fail(NORI_HIP_EINVAL, "null buffer");
    if (r < 0 || f < 0 || !(sigma > 0) || !(h > 0)) return fail(NORI_HIP_EINVAL, "bad filter parameters");
    const int W = ctx->sc.cam.width, H = ctx->sc.cam.height, R = r + f;
    if ((size_t) W * (size_t) H > (size_t) NORI_HIP_NOALLOC_MAX_PIXELS)
        return fail(NORI_HIP_EINVAL, "nori_hip_filter_noalloc: image of " + std::to_string((size_t) W * H) + " pixels exceeds NORI_HIP_NOALLOC_MAX_PIXELS (" +
                    std::to_string((size_t) NORI_HIP_NOALLOC_MAX_PIXELS) + "); use nori_hip_filter");
    const size_t lds_sep = ((size_t) (NLM_BW + 2 * R) * (NLM_BH + 2 * R) * 3 + (size_t) (NLM_BW + 2 * f) * (NLM_BH + 2 * f) + (size_t) (NLM_BH + 2 * f) * NLM_BW + (size_t) (2 * f + 1)) * sizeof(float);
    if (lds_sep > std::min<size_t>(ctx->lds_bytes, 160 * 1024)) return fail(NORI_HIP_EINVAL, "filter window + patch too large for the LDS tile");
    float *d_srgb = ctx->scratch_ptr;
    hipLaunchKernelGGL(k_bitmap_srgb, dim3((W + 15) / 16, (H + 15) / 16), dim3(16, 16), 0, ctx->stream, (const float4 *) ctx->d_film, W, H, ctx->border, ctx->film_cols, d_srgb);
    hipLaunchKernelGGL(k_nlmeans_sep, dim3((W + NLM_BW - 1) / NLM_BW, (H + NLM_BH - 1) / NLM_BH), dim3(NLM_BW, NLM_BH), lds_sep, ctx->stream,
                       (const float *) d_srgb, d_rgb_srgb_out, W, H, r, f, sigma, h);
    HIPCHK(hipGetLastError());
    return NORI_HIP_OK;
}

int nori_hip_stats(NoriHipContext *ctx, NoriHipStats *out) {
    CTXCHK(ctx);
    if (!out) return fail(NORI_HIP_EINVAL, "null buffer");
    HIPCHK(hipStreamSynchronize(ctx->stream));
    std::vector<DevStats> stripes(ND_STRIPES);
    HIPCHK(hipMemcpy(stripes.data(), ctx->d_stats, ND_STRIPES * sizeof(DevStats), hipMemcpyDeviceToHost));
    if (ctx->twin) {                  // the second pipeline's counters and trace launches count with the first's (its work is behind ctx->stream's)
        stripes.resize(2 * ND_STRIPES);
        HIPCHK(hipMemcpy(stripes.data() + ND_STRIPES, ctx->twin->d_stats, ND_STRIPES * sizeof(DevStats), hipMemcpyDeviceToHost));
    }
    DevStats ds; std::memset(&ds, 0, sizeof(ds));
    for (const DevStats &t : stripes) {
        ds.ext_rays += t.ext_rays; ds.ext_nodes += t.ext_nodes; ds.ext_prims += t.ext_prims; ds.ext_visits += t.ext_visits; ds.culled += t.culled; ds.dropped += t.dropped;
        ds.samples += t.samples; ds.closest += t.closest; ds.shadow += t.shadow; ds.invalid += t.invalid; ds.nodes += t.nodes; ds.prims += t.prims;
        for (int k = 0; k < 2; ++k) for (int j = 0; j < 10; ++j) ds.prof[k][j] += t.prof[k][j];
    }
    ctx->stats.camera_samples = ds.samples; ctx->stats.closest_rays = ds.closest; ctx->stats.shadow_rays = ds.shadow; ctx->stats.invalid_samples = ds.invalid;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) == hipSuccess) ctx->stats.render_ms = ms; else (void) hipGetLastError();
    ctx->stats.boxes_tested = ds.nodes; ctx->stats.prims_tested = ds.prims;
    ctx->fold_ext_events();
    ctx->stats.extend_ms = ctx->ext_ms_acc; ctx->stats.extend_launches = ctx->ext_launches;
    if (ctx->twin) { ctx->twin->fold_ext_events(); ctx->stats.extend_ms += ctx->twin->ext_ms_acc; ctx->stats.extend_launches += ctx->twin->ext_launches; }
    ctx->stats.extend_rays = ds.ext_rays; ctx->stats.extend_boxes = ds.ext_nodes; ctx->stats.extend_prims = ds.ext_prims;
    ctx->stats.extend_visits = ds.ext_visits; ctx->stats.extend_width = ctx->trace_width; ctx->stats.extend_node_bytes = ctx->trace_node_bytes; ctx->stats.shadow_rays_culled = ds.culled; ctx->stats.ss_paths_dropped = ds.dropped;
    if (ctx->profile) {
        static const char *names[10] = {"trips", "box_runs", "box_lanes", "prim_runs", "prim_lanes", "pop_runs", "pop_lanes", "refill_runs", "retired", "busy_lanes"};
        for (int k = 0; k < 2; ++k) {
            fprintf(stderr, "[nori_hip profile] %s:", k ? "shadow" : "extend");
            for (int j = 0; j < 10; ++j) fprintf(stderr, " %s=%llu", names[j], ds.prof[k][j]);
            fprintf(stderr, "\n");
        }
    }
    *out = ctx->stats;
    return NORI_HIP_OK;
}
int nori_hip_reset_stats(NoriHipContext *ctx) {
    CTXCHK(ctx);
    HIPCHK(hipMemsetAsync(ctx->d_stats, 0, ND_STRIPES * sizeof(DevStats), ctx->stream));
    ctx->fold_ext_events(); ctx->ext_ms_acc = 0.0; ctx->ext_launches = 0;
    if (ctx->twin) {
        HIPCHK(hipMemsetAsync(ctx->twin->d_stats, 0, ND_STRIPES * sizeof(DevStats), ctx->stream));
        ctx->twin->fold_ext_events(); ctx->twin->ext_ms_acc = 0.0; ctx->twin->ext_launches = 0;
    }
    return NORI_HIP_OK;
}
int nori_hip_stream(NoriHipContext *ctx, void **stream) { CTXCHK(ctx); if (stream) *stream = ctx->stream; return NORI_HIP_OK; }
int nori_hip_synchronize(NoriHipContext *ctx) { CTXCHK(ctx); HIPCHK(hipStreamSynchronize(ctx->stream)); return NORI_HIP_OK; }

} // extern "C"
