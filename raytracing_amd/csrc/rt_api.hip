// rt_api.hip — implementation of the C ABI declared in include/rt_abi.h:
// context / scene lifetime, scene ingestion (validation, leaf boxes,
// BVH build, upload) and the host side of the wavefront loop.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_abi.h"
#include "rt_bvh.h"
#include "rt_device.h"
#include "rt_kernels.h"

using namespace rt;

namespace {

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(e_ == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                \
    } while (0)

constexpr uint64_t DEFAULT_BATCH_PATHS = 128ull << 20;  // 116 B of queue memory per path
constexpr int N_PHASES = 6;

struct EventPair {
    hipEvent_t a, b;
    int phase;
};

#ifndef RT_MAX_PIPES
#define RT_MAX_PIPES 2
#endif
constexpr int MAX_PIPES = RT_MAX_PIPES;
constexpr int MAX_DEVICES = 64;
constexpr uint64_t BYTES_PER_PATH = 4 * sizeof(float4) + 2 * 3 * sizeof(float) + sizeof(uint2) + sizeof(uint32_t) + sizeof(float4);

// Everything one batch in flight needs: the ray / throughput queues (double-buffered), hit
// records, the list of textured hits, per-path radiance, queue counters, and the extend
// kernel's stack overflow area.
struct Pipe {
    hipStream_t stream = nullptr;  // pipe 0 of a one-pipe context runs on the caller's stream instead
    uint64_t cap = 0;              // paths the queues hold
    float4* ray_o[2] = {nullptr, nullptr};
    float4* ray_d[2] = {nullptr, nullptr};
    float* thr[2] = {nullptr, nullptr};      // throughput, three floats per slot
    uint2* hits = nullptr;
    uint32_t* tex_list = nullptr;
    float4* radiance = nullptr;
    uint32_t* counts = nullptr;
    uint32_t counts_cap = 0;
    unsigned long long* stack_overflow = nullptr;
    hipEvent_t done = nullptr;     // accumulate of the pipe's latest batch
};

// ---- scratch memory of rt_render_accumulate_device_noalloc ------------------------------------------
// One pipeline's queues for NOALLOC_PATHS paths, its queue counters for up to NOALLOC_MAX_DEPTH bounces and a
// stack overflow area for NOALLOC_OVERFLOW_BLOCKS blocks of the extend kernel, carved out of one array that is
// part of the library's device image: every GPU the library is loaded on holds its own copy for the life of the
// process.  Host code never uses the array's name as an address: rt_ctx_create asks the runtime for the array's
// device address on the context's GPU (scratch_base, per device) and everything else goes through that pointer.
constexpr uint64_t NOALLOC_PATHS = 1ull << 18;
constexpr uint32_t NOALLOC_MAX_DEPTH = 1024;
constexpr uint32_t NOALLOC_OVERFLOW_BLOCKS = 256;
constexpr size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }
constexpr size_t NOALLOC_RAY_BYTES = align256(NOALLOC_PATHS * sizeof(float4));
constexpr size_t NOALLOC_THR_BYTES = align256(NOALLOC_PATHS * 3 * sizeof(float));
constexpr size_t NOALLOC_HITS_BYTES = align256(NOALLOC_PATHS * sizeof(uint2));
constexpr size_t NOALLOC_TEX_BYTES = align256(NOALLOC_PATHS * sizeof(uint32_t));
constexpr size_t NOALLOC_COUNTS_BYTES = align256((2 * NOALLOC_MAX_DEPTH + 2) * sizeof(uint32_t));
constexpr size_t NOALLOC_OVERFLOW_BYTES = (size_t)NOALLOC_OVERFLOW_BLOCKS * 4 * (rt::TRAVERSAL_STACK_ENTRIES - 8) * 64 * sizeof(unsigned long long);
constexpr size_t NOALLOC_SCRATCH_BYTES = 5 * NOALLOC_RAY_BYTES /* ray_o x2, ray_d x2, radiance */ + 2 * NOALLOC_THR_BYTES + NOALLOC_HITS_BYTES +
                                         NOALLOC_TEX_BYTES + NOALLOC_COUNTS_BYTES + NOALLOC_OVERFLOW_BYTES;
static __device__ __align__(256) unsigned char scratch[NOALLOC_SCRATCH_BYTES];
unsigned char* scratch_base[MAX_DEVICES] = {};  // device address of `scratch` on each GPU a context was created on

}  // namespace

struct rt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    uint32_t max_blocks = 2048;
    uint32_t shade_blocks = 2048;   // grid limit of the shade kernel
    uint32_t extend_blocks = 1024;  // persistent grid of the extend kernel: what the chip holds at once
    uint32_t extend_blocks_shared = 1024;  // the same for a render call whose batches run on two pipelines (see render)
    uint32_t extend_blocks_counting = 1024;  // the same for its counting build (more registers)
    uint32_t extend_blocks_generic = 1024, extend_blocks_counting_generic = 1024;  // the same for the variants that scenes with
                                                                                  // media of any boundary get (never more blocks)
    uint32_t tail_blocks = 512;     // what the chip holds of the tail kernel
    rt::ExtendTuning tune;
    uint64_t batch_paths = DEFAULT_BATCH_PATHS;
    bool counters_on = false;
    bool phase_timing = false;
    uint32_t tail_rays = 3u << 16;  // queue length at which the tail kernel should take over
    int tail_start = -1;  // first bounce the tail kernel takes over (-1: chosen per batch, see tail_start_for)
    // queue lengths per bounce of the latest batch, copied to pinned host memory when its call ends: what the next
    // batch of the same job uses to place the switch to the tail kernel
    uint32_t* h_counts = nullptr;
    hipEvent_t ev_counts = nullptr;
    struct {
        const rt_scene* scene = nullptr;
        uint32_t width = 0, height = 0, max_depth = 0, paths = 0;
        uint32_t n = 0;           // entries of h_counts in flight or valid
        uint32_t ran_to = 0;      // bounces the per-bounce kernels did in that batch: h_counts[0 .. ran_to] are real
        int chosen = -1;          // the switch derived from them, once they have arrived
    } seen;

    // Batches of one render call are dealt to n_pipes pipelines in turn.  With two, the kernels of
    // consecutive batches overlap on the chip: the extend kernel is bound by instruction issue
    // and latency, ray generation / shade / accumulate by HBM.
    int n_pipes = 2;
    Pipe pipes[MAX_PIPES];
    DCounters* d_counters = nullptr;

    // host-buffer entry point: the device copy of the caller's accumulation buffer, the caller's buffer as it is
    // registered (page-locked) with the runtime, and the stream / event of the upload that runs beside the render
    float* d_acc_cache = nullptr;
    size_t d_acc_cache_bytes = 0;
    void* pinned_ptr = nullptr;
    size_t pinned_bytes = 0;
    bool pin_host_buffers = false;  // rt_ctx_set_host_buffer_pinning
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_upload = nullptr;

    // rt_render_accumulate_multi: this GPU's private buffer (its share of the samples, or its rows), the buffers that
    // the other GPUs' shares arrive in when this is the GPU that holds the image, and the events that order the two
    float* d_partial = nullptr;
    size_t d_partial_bytes = 0;
    std::vector<float*> peer_in;
    size_t peer_in_bytes = 0;
    hipEvent_t ev_multi_ready = nullptr, ev_multi_done = nullptr;
    uint64_t peer_asked = 0;  // GPUs this one has asked the runtime for direct access to (bit per device)

    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    bool have_render_time = false;
    std::vector<EventPair> ev_pool;
    size_t ev_used = 0;
    float phase_ms[N_PHASES] = {};
    uint32_t phase_launches[N_PHASES] = {};
    bool phase_pending = false;
};

struct rt_scene {
    rt_ctx* ctx = nullptr;
    std::vector<void*> allocations;
    DScene d{};
    uint32_t bvh_depth = 0;
    uint32_t stack_need = 0;    // traversal stack entries the tree can need in the worst case
    bool has_textures = false;  // some leaf's material is of the class that k_shade_textured shades
    bool media_are_single_spheres = true;  // every medium's boundary is one unplaced sphere (the short medium test)
    rt_scene_info info{};
};

namespace {

int set_device(const rt_ctx* ctx) {
    HIP_TRY(hipSetDevice(ctx->device));
    return RT_OK;
}

void free_queues(Pipe& p) {
    for (int i = 0; i < 2; ++i) {
        if (p.ray_o[i]) (void)hipFree(p.ray_o[i]);
        if (p.ray_d[i]) (void)hipFree(p.ray_d[i]);
        if (p.thr[i]) (void)hipFree(p.thr[i]);
        p.ray_o[i] = p.ray_d[i] = nullptr;
        p.thr[i] = nullptr;
    }
    if (p.hits) (void)hipFree(p.hits);
    if (p.tex_list) (void)hipFree(p.tex_list);
    if (p.radiance) (void)hipFree(p.radiance);
    p.hits = nullptr;
    p.tex_list = nullptr;
    p.radiance = nullptr;
    p.cap = 0;
}

// need_overflow: the scene's tree can stack more entries than the extend kernel keeps in LDS
int ensure_queues(rt_ctx* c, Pipe& p, uint64_t paths, uint32_t n_counts, bool need_overflow) {
    if (paths > p.cap) {
        free_queues(p);
        for (int i = 0; i < 2; ++i) {
            HIP_TRY(hipMalloc((void**)&p.ray_o[i], paths * sizeof(float4)));
            HIP_TRY(hipMalloc((void**)&p.ray_d[i], paths * sizeof(float4)));
            HIP_TRY(hipMalloc((void**)&p.thr[i], paths * 3 * sizeof(float)));
        }
        HIP_TRY(hipMalloc((void**)&p.hits, paths * sizeof(uint2)));
        HIP_TRY(hipMalloc((void**)&p.tex_list, paths * sizeof(uint32_t)));
        HIP_TRY(hipMalloc((void**)&p.radiance, paths * sizeof(float4)));
        p.cap = paths;
    }
    if (n_counts > p.counts_cap) {
        if (p.counts) (void)hipFree(p.counts);
        p.counts = nullptr;
        HIP_TRY(hipMalloc((void**)&p.counts, n_counts * sizeof(uint32_t)));
        p.counts_cap = n_counts;
    }
    if (need_overflow && !p.stack_overflow) HIP_TRY(hipMalloc((void**)&p.stack_overflow, rt::extend_overflow_bytes(c->extend_blocks)));
    if (!p.done) HIP_TRY(hipEventCreateWithFlags(&p.done, hipEventDisableTiming));
    if (!p.stream && c->n_pipes > 1) HIP_TRY(hipStreamCreateWithFlags(&p.stream, hipStreamNonBlocking));
    return RT_OK;
}

void destroy_pipe(Pipe& p) {
    if (p.stream) (void)hipStreamSynchronize(p.stream);
    free_queues(p);
    if (p.counts) (void)hipFree(p.counts);
    if (p.stack_overflow) (void)hipFree(p.stack_overflow);
    if (p.done) (void)hipEventDestroy(p.done);
    if (p.stream) (void)hipStreamDestroy(p.stream);
    p = Pipe{};
}

// ---- per-launch phase timing ------------------------------------------------------
int phase_begin(rt_ctx* c, bool timing, hipStream_t stream, int phase, EventPair** out) {
    *out = nullptr;
    if (!timing) return RT_OK;
    if (c->ev_used == c->ev_pool.size()) {
        EventPair p{};
        HIP_TRY(hipEventCreate(&p.a));
        HIP_TRY(hipEventCreate(&p.b));
        c->ev_pool.push_back(p);
    }
    EventPair* p = &c->ev_pool[c->ev_used++];
    p->phase = phase;
    HIP_TRY(hipEventRecord(p->a, stream));
    *out = p;
    return RT_OK;
}
int phase_end(hipStream_t stream, EventPair* p) {
    if (p) HIP_TRY(hipEventRecord(p->b, stream));
    return RT_OK;
}
int collect_phases(rt_ctx* c) {
    if (!c->phase_pending) return RT_OK;
    for (int i = 0; i < N_PHASES; ++i) { c->phase_ms[i] = 0; c->phase_launches[i] = 0; }
    for (size_t i = 0; i < c->ev_used; ++i) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev_pool[i].a, c->ev_pool[i].b));
        c->phase_ms[c->ev_pool[i].phase] += ms;
        c->phase_launches[c->ev_pool[i].phase]++;
    }
    c->ev_used = 0;
    c->phase_pending = false;
    return RT_OK;
}

// ---- scene ingestion ---------------------------------------------------------------
struct Xf {
    double s = 0, c = 1, off[3] = {0, 0, 0};
    bool identity = true;
};

// object -> world: p = R^-1 p' + offset, R^-1: x = cos x' + sin z', z = -sin x' + cos z'
void rot_inv(const Xf& x, const double in[3], double out[3]) {
    out[0] = x.c * in[0] + x.s * in[2];
    out[1] = in[1];
    out[2] = -x.s * in[0] + x.c * in[2];
}
void point_to_world(const Xf& x, const float in[3], double out[3]) {
    double v[3] = {in[0], in[1], in[2]};
    if (x.identity) { out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; return; }
    rot_inv(x, v, out);
    for (int a = 0; a < 3; ++a) out[a] += x.off[a];
}
void vec_to_world(const Xf& x, const float in[3], double out[3]) {
    double v[3] = {in[0], in[1], in[2]};
    if (x.identity) { out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; return; }
    rot_inv(x, v, out);
}

// Leaf boxes.  For leaves that sit in world space the box is the reference's own AABB,
// computed in f32 in the reference's order (src/hittable.rs:158-178, src/math/aabb.rs:22-98),
// because the reference's BVH decides with exactly this box whether a leaf is tested at all.
struct Aabb {
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    bool empty() const { return mn[0] == INFINITY && mx[0] == -INFINITY; }
    // AABB::pad_to_minimums (src/math/aabb.rs:79-98)
    void pad_to_minimums() {
        const float delta = 0.0001f, delta_half = delta * 0.5f;
        for (int a = 0; a < 3; ++a)
            if (std::fabs(mx[a] - mn[a]) < delta) { mn[a] = mn[a] - delta_half; mx[a] = mx[a] + delta_half; }
    }
    // AABB::from_points (src/math/aabb.rs:52-58)
    static Aabb from_points(const float a[3], const float b[3]) {
        Aabb r;
        for (int i = 0; i < 3; ++i) { r.mn[i] = a[i] <= b[i] ? a[i] : b[i]; r.mx[i] = a[i] <= b[i] ? b[i] : a[i]; }
        r.pad_to_minimums();
        return r;
    }
    // AABB::from_aabbs (src/math/aabb.rs:61-69)
    static Aabb from_aabbs(const Aabb& x, const Aabb& y) {
        if (y.empty()) return x;
        Aabb r;
        for (int i = 0; i < 3; ++i) { r.mn[i] = std::fmin(x.mn[i], y.mn[i]); r.mx[i] = std::fmax(x.mx[i], y.mx[i]); }
        r.pad_to_minimums();
        return r;
    }
    // conservative growth for placed leaves (world-space bound of an object-space leaf)
    void grow(const double p[3], double r = 0.0) {
        for (int a = 0; a < 3; ++a) {
            mn[a] = std::min(mn[a], std::nextafter((float)(p[a] - r), -INFINITY));
            mx[a] = std::max(mx[a], std::nextafter((float)(p[a] + r), INFINITY));
        }
    }
    void grow(const Aabb& o) {
        for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], o.mn[a]); mx[a] = std::max(mx[a], o.mx[a]); }
    }
};

template <class T>
int upload(rt_scene* sc, const std::vector<T>& host, const T** out) {
    *out = nullptr;
    size_t bytes = std::max<size_t>(host.size() * sizeof(T), 16);
    sc->info.device_bytes += bytes;
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, bytes));
    sc->allocations.push_back(d);
    if (!host.empty()) HIP_TRY(hipMemcpy(d, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = (const T*)d;
    return RT_OK;
}

int validate(const rt_scene_desc* d) {
    if (d->abi_version != RT_ABI_VERSION) return fail(RT_ERR_INVALID_ARGUMENT, "scene desc abi_version %u, library %u", d->abi_version, RT_ABI_VERSION);
    if (d->n_materials == 0) return fail(RT_ERR_INVALID_ARGUMENT, "scene needs at least the default material (MaterialId 0)");
    if (d->n_spheres >= MAX_PRIMS_PER_KIND || d->n_moving_spheres >= MAX_PRIMS_PER_KIND || d->n_quads >= MAX_PRIMS_PER_KIND ||
        d->n_media >= MAX_PRIMS_PER_KIND)
        return fail(RT_ERR_UNSUPPORTED, "more than 2^25 primitives of one kind");
    auto need = [&](const void* p, uint32_t n, const char* what) { return n == 0 || p != nullptr ? 0 : fail(RT_ERR_INVALID_ARGUMENT, "%s is NULL but its count is %u", what, n); };
    int e;
    if ((e = need(d->spheres, d->n_spheres, "spheres"))) return e;
    if ((e = need(d->moving_spheres, d->n_moving_spheres, "moving_spheres"))) return e;
    if ((e = need(d->quads, d->n_quads, "quads"))) return e;
    if ((e = need(d->media, d->n_media, "media"))) return e;
    if ((e = need(d->world, d->n_world, "world"))) return e;
    if ((e = need(d->boundary, d->n_boundary, "boundary"))) return e;
    if ((e = need(d->transforms, d->n_transforms, "transforms"))) return e;
    if ((e = need(d->materials, d->n_materials, "materials"))) return e;
    if ((e = need(d->textures, d->n_textures, "textures"))) return e;
    if ((e = need(d->images, d->n_images, "images"))) return e;
    if ((e = need(d->perlins, d->n_perlins, "perlins"))) return e;
    auto count_of = [&](uint32_t kind) -> int64_t {
        switch (kind) {
            case RT_PRIM_SPHERE: return d->n_spheres;
            case RT_PRIM_MOVING_SPHERE: return d->n_moving_spheres;
            case RT_PRIM_QUAD: return d->n_quads;
            case RT_PRIM_MEDIUM: return d->n_media;
            default: return -1;
        }
    };
    for (uint32_t i = 0; i < d->n_world; ++i)
        if ((int64_t)d->world[i].index >= count_of(d->world[i].kind)) return fail(RT_ERR_INVALID_ARGUMENT, "world[%u] references kind %u index %u out of range", i, d->world[i].kind, d->world[i].index);
    for (uint32_t i = 0; i < d->n_boundary; ++i) {
        if (d->boundary[i].kind == RT_PRIM_MEDIUM) return fail(RT_ERR_UNSUPPORTED, "boundary[%u]: a medium cannot bound another medium", i);
        if ((int64_t)d->boundary[i].index >= count_of(d->boundary[i].kind)) return fail(RT_ERR_INVALID_ARGUMENT, "boundary[%u] out of range", i);
    }
    auto mat_ok = [&](uint32_t m) { return m < d->n_materials; };
    auto xf_ok = [&](uint32_t t) { return t == RT_NO_TRANSFORM || t < d->n_transforms; };
    for (uint32_t i = 0; i < d->n_spheres; ++i)
        if (!mat_ok(d->spheres[i].material) || !xf_ok(d->spheres[i].transform) || !(d->spheres[i].radius == d->spheres[i].radius))
            return fail(RT_ERR_INVALID_ARGUMENT, "spheres[%u]: bad material, transform or radius", i);
    for (uint32_t i = 0; i < d->n_moving_spheres; ++i)
        if (!mat_ok(d->moving_spheres[i].material) || !xf_ok(d->moving_spheres[i].transform)) return fail(RT_ERR_INVALID_ARGUMENT, "moving_spheres[%u]: bad material or transform", i);
    for (uint32_t i = 0; i < d->n_quads; ++i)
        if (!mat_ok(d->quads[i].material) || !xf_ok(d->quads[i].transform)) return fail(RT_ERR_INVALID_ARGUMENT, "quads[%u]: bad material or transform", i);
    for (uint32_t i = 0; i < d->n_media; ++i) {
        const rt_medium& m = d->media[i];
        if (!mat_ok(m.phase_material) || (uint64_t)m.first_boundary + m.n_boundary > d->n_boundary || m.n_boundary == 0)
            return fail(RT_ERR_INVALID_ARGUMENT, "media[%u]: bad phase material or boundary range", i);
    }
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        if (d->materials[i].kind > RT_MAT_ISOTROPIC) return fail(RT_ERR_INVALID_ARGUMENT, "materials[%u]: unknown kind %u", i, d->materials[i].kind);
        if (d->materials[i].texture >= d->n_textures) return fail(RT_ERR_INVALID_ARGUMENT, "materials[%u]: texture %u out of range", i, d->materials[i].texture);
    }
    for (uint32_t i = 0; i < d->n_textures; ++i) {
        const rt_texture& t = d->textures[i];
        switch (t.kind) {
            case RT_TEX_SOLID: break;
            case RT_TEX_CHECKER:
                if (t.even >= d->n_textures || t.odd >= d->n_textures) return fail(RT_ERR_INVALID_ARGUMENT, "textures[%u]: checker children out of range", i);
                break;
            case RT_TEX_IMAGE:
                if (t.image >= d->n_images) return fail(RT_ERR_INVALID_ARGUMENT, "textures[%u]: image %u out of range", i, t.image);
                break;
            case RT_TEX_NOISE:
                if (t.perlin >= d->n_perlins) return fail(RT_ERR_INVALID_ARGUMENT, "textures[%u]: perlin %u out of range", i, t.perlin);
                break;
            default: return fail(RT_ERR_INVALID_ARGUMENT, "textures[%u]: unknown kind %u", i, t.kind);
        }
    }
    for (uint32_t i = 0; i < d->n_images; ++i) {
        const rt_image& im = d->images[i];
        if (im.width == 0 || im.height == 0 || !im.texels || im.format > RT_IMAGE_RGB32F) return fail(RT_ERR_INVALID_ARGUMENT, "images[%u]: empty or unknown format", i);
    }
    for (uint32_t i = 0; i < d->n_perlins; ++i) {
        const rt_perlin& p = d->perlins[i];
        if (p.point_count == 0 || !p.rand_vecs || !p.perm_x || !p.perm_y || !p.perm_z) return fail(RT_ERR_INVALID_ARGUMENT, "perlins[%u]: empty tables", i);
        if (p.point_count > (1u << 20)) return fail(RT_ERR_UNSUPPORTED, "perlins[%u]: %u points (at most %u are supported)", i, p.point_count, 1u << 20);
        for (uint32_t k = 0; k < p.point_count; ++k)
            if (p.perm_x[k] >= p.point_count || p.perm_y[k] >= p.point_count || p.perm_z[k] >= p.point_count)
                return fail(RT_ERR_INVALID_ARGUMENT, "perlins[%u]: permutation entry out of range", i);
    }
    return RT_OK;
}

// First bounce that the tail kernel handles (everything from there to max_depth in one launch); max_depth: no tail.
// The per-bounce kernels are the efficient way to advance a long queue and a poor way to advance a short one (every
// bounce costs its tens of microseconds of launches and dependent round trips whatever the queue holds), while the
// tail kernel - one thread per path, no regrouping - is the reverse: the switch belongs where the queue is down to
// tail_rays rays.  The host does not wait for queue lengths, but a progressive render repeats the same batch with new
// sample numbers, so the lengths of the previous batch (copied to pinned memory when its call ended) place the switch
// of the next one; the first batch of a job uses a model that loses a quarter of the paths per bounce.  Batches above
// 2^24 paths never get that short at the depths in use, and the counting build keeps to the kernels that count.
constexpr uint32_t TAIL_MAX_BATCH = 1u << 24;
constexpr uint32_t COUNTS_SEEN_MAX = 1024;
uint32_t tail_start_for(rt_ctx* c, const rt_scene* sc, const rt_camera_desc* cam, uint64_t paths, bool learn) {
    const uint32_t max_depth = cam->max_depth;
    if (c->counters_on) return max_depth;  // the tail kernel has no counting build: with the work counters on, every bounce runs the kernels that count
    if (c->tail_start >= 0) return std::min<uint32_t>((uint32_t)c->tail_start, max_depth);
    if (paths > TAIL_MAX_BATCH) return max_depth;
    auto& seen = c->seen;
    if (learn && seen.scene == sc && seen.width == cam->width && seen.height == cam->height && seen.max_depth == max_depth && seen.paths == paths) {
        if (seen.n != 0 && hipEventQuery(c->ev_counts) == hipSuccess) {
            // first bounce whose queue was short enough; past the bounces that were counted, the last ratio continues
            uint32_t bounce = 0;
            while (bounce <= seen.ran_to && bounce < seen.n && c->h_counts[bounce] > c->tail_rays) ++bounce;
            if (bounce > seen.ran_to && seen.ran_to >= 1) {
                const double last = (double)c->h_counts[seen.ran_to], before = (double)c->h_counts[seen.ran_to - 1];
                const double ratio = std::min(0.95, std::max(0.05, before > 0.0 ? last / before : 0.5));
                for (double rays = last * ratio; rays > (double)c->tail_rays && bounce < max_depth; rays *= ratio) ++bounce;
            }
            seen.chosen = (int)std::min(bounce, max_depth);
            seen.n = 0;  // used up; the batch about to be enqueued sends new ones
        }
        (void)hipGetLastError();
        if (seen.chosen >= 0) return (uint32_t)seen.chosen;
    }
    uint32_t bounce = 0;
    for (double rays = (double)paths; rays > (double)c->tail_rays && bounce < max_depth; rays *= 0.75) ++bounce;
    return bounce;
}

// persistent grid of the extend kernel variant this scene runs (never more than the overflow area is sized for)
uint32_t extend_grid_of(const rt_ctx* c, const rt_scene* sc, bool counting) {
    if (sc->d.generic_media) return counting ? c->extend_blocks_counting_generic : c->extend_blocks_generic;
    return counting ? c->extend_blocks_counting : c->extend_blocks;
}
bool needs_overflow(const rt_scene* sc) { return sc->stack_need > (uint32_t)rt::extend_lds_stack_levels(sc->d.n_nodes); }

DCamera to_dcamera(const rt_camera_desc* c) {
    DCamera d;
    d.centre = f3(c->centre[0], c->centre[1], c->centre[2]);
    d.pixel00 = f3(c->pixel00_loc[0], c->pixel00_loc[1], c->pixel00_loc[2]);
    d.du = f3(c->pixel_delta_u[0], c->pixel_delta_u[1], c->pixel_delta_u[2]);
    d.dv = f3(c->pixel_delta_v[0], c->pixel_delta_v[1], c->pixel_delta_v[2]);
    d.ddu = f3(c->defocus_disk_u[0], c->defocus_disk_u[1], c->defocus_disk_u[2]);
    d.ddv = f3(c->defocus_disk_v[0], c->defocus_disk_v[1], c->defocus_disk_v[2]);
    d.background = f3(c->background_colour[0], c->background_colour[1], c->background_colour[2]);
    d.defocus_angle = c->defocus_angle;
    d.width = c->width;
    d.height = c->height;
    d.max_depth = c->max_depth;
    return d;
}

int check_camera(const rt_camera_desc* cam) {
    if (!cam) return fail(RT_ERR_INVALID_ARGUMENT, "camera is NULL");
    if (cam->width == 0 || cam->height == 0) return fail(RT_ERR_INVALID_ARGUMENT, "camera has an empty image (%ux%u)", cam->width, cam->height);
    if ((uint64_t)cam->width * cam->height >= (1ull << 31)) return fail(RT_ERR_UNSUPPORTED, "image larger than 2^31 pixels");
    if (cam->max_depth > 4096) return fail(RT_ERR_UNSUPPORTED, "max_depth above 4096");
    return RT_OK;
}

void release_host_buffer(rt_ctx* c) {
    if (c->pinned_ptr) (void)hipHostUnregister(c->pinned_ptr);
    (void)hipGetLastError();
    c->pinned_ptr = nullptr;
    c->pinned_bytes = 0;
}

}  // namespace

extern "C" {

const char* rt_last_error(void) { return g_last_error.c_str(); }
uint32_t rt_abi_version(void) { return RT_ABI_VERSION; }

int rt_ctx_create(int device_id, rt_ctx** out) {
    if (!out) return fail(RT_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(RT_ERR_NO_DEVICE, "no HIP device available");
    if (device_id < 0 || device_id >= n) return fail(RT_ERR_INVALID_ARGUMENT, "device %d out of range (0..%d)", device_id, n - 1);
    HIP_TRY(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    rt_ctx* c = new rt_ctx();
    c->device = device_id;
    c->max_blocks = (uint32_t)std::max(1, prop.multiProcessorCount) * 8u;
    c->shade_blocks = c->max_blocks;
#ifdef RT_DEV_KNOBS
    if (const char* e = getenv("RT_AMD_SHADE_BLOCKS")) c->shade_blocks = (uint32_t)std::max(1, atoi(e));
#endif
    {
        int per_cu = rt::extend_blocks_per_cu(false, false), per_cu_counting = rt::extend_blocks_per_cu(true, false);
        int per_cu_generic = rt::extend_blocks_per_cu(false, true), per_cu_counting_generic = rt::extend_blocks_per_cu(true, true);
#ifdef RT_DEV_KNOBS  // development builds (make DEV=1): environment overrides for tuning experiments, not part of the product
        if (const char* e = getenv("RT_AMD_EXTEND_BLOCKS_PER_CU")) per_cu = atoi(e);
#endif
        if (per_cu < 1) per_cu = 4;
        if (per_cu_counting < 1) per_cu_counting = 3;
        c->extend_blocks = (uint32_t)std::max(1, prop.multiProcessorCount) * (uint32_t)per_cu;
        c->extend_blocks_counting = (uint32_t)std::max(1, prop.multiProcessorCount) * (uint32_t)std::min(per_cu, per_cu_counting);
        if (per_cu_generic < 1) per_cu_generic = 3;
        if (per_cu_counting_generic < 1) per_cu_counting_generic = 3;
        c->extend_blocks_generic = (uint32_t)std::max(1, prop.multiProcessorCount) * (uint32_t)std::min(per_cu, per_cu_generic);
        c->extend_blocks_counting_generic =
            (uint32_t)std::max(1, prop.multiProcessorCount) * (uint32_t)std::min(std::min(per_cu, per_cu_generic), per_cu_counting_generic);
#ifdef RT_DEV_KNOBS
        if (const char* e = getenv("RT_AMD_EXTEND_BLOCKS")) c->extend_blocks = std::min(c->extend_blocks, (uint32_t)std::max(1, atoi(e)));  // whole grid
#endif
        // half a block per CU fewer when two pipelines share the chip
        c->extend_blocks_shared = std::max(c->extend_blocks - (uint32_t)std::max(1, prop.multiProcessorCount) / 2u, (uint32_t)std::max(1, prop.multiProcessorCount));
#ifdef RT_DEV_KNOBS
        if (const char* e = getenv("RT_AMD_EXTEND_BLOCKS_SHARED")) c->extend_blocks_shared = (uint32_t)std::max(1, atoi(e));
#endif
        // the overflow area of the walk's stacks is sized for extend_blocks: no variant's grid may be larger
        c->extend_blocks_counting = std::min(c->extend_blocks_counting, c->extend_blocks);
        c->extend_blocks_generic = std::min(c->extend_blocks_generic, c->extend_blocks);
        c->extend_blocks_counting_generic = std::min(c->extend_blocks_counting_generic, c->extend_blocks);
        int tail_per_cu = std::max(1, rt::tail_blocks_per_cu());
#ifdef RT_DEV_KNOBS
        if (const char* e = getenv("RT_AMD_TAIL_BLOCKS_PER_CU")) tail_per_cu = std::max(1, atoi(e));
#endif
        c->tail_blocks = (uint32_t)std::max(1, prop.multiProcessorCount) * (uint32_t)tail_per_cu;
#ifdef RT_DEV_KNOBS
        if (const char* e = getenv("RT_AMD_EXTEND_TUNING"))  // "inner,refill,medium,quad"
            sscanf(e, "%d,%d,%d,%d", &c->tune.inner_min, &c->tune.refill_min, &c->tune.medium_min, &c->tune.quad_min);
        if (const char* e = getenv("RT_AMD_TAIL_RAYS")) c->tail_rays = (uint32_t)std::max(1, atoi(e));
        if (const char* e = getenv("RT_AMD_PIPES")) c->n_pipes = std::min(MAX_PIPES, std::max(1, atoi(e)));
        if (getenv("RT_AMD_VERBOSE"))
            fprintf(stderr, "[rt_amd] extend kernel: %d blocks per CU, grid %u; %d pipeline(s)\n", per_cu, c->extend_blocks, c->n_pipes);
#endif
    }
    hipError_t e = hipSuccess;
    if (device_id < MAX_DEVICES && scratch_base[device_id] == nullptr) {
        // this GPU's copy of the scratch array (the device is current): its address, once per device
        void* at = nullptr;
        e = hipGetSymbolAddress(&at, HIP_SYMBOL(scratch));
        if (e == hipSuccess) scratch_base[device_id] = (unsigned char*)at;
    }
    if (e == hipSuccess) e = hipMalloc((void**)&c->d_counters, sizeof(DCounters));
    if (e == hipSuccess) e = hipMemset(c->d_counters, 0, sizeof(DCounters));
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_counts, COUNTS_SEEN_MAX * sizeof(uint32_t), hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_counts, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreate(&c->ev_begin);
    if (e == hipSuccess) e = hipEventCreate(&c->ev_end);
    if (e != hipSuccess) {
        (void)rt_ctx_destroy(c);  // frees whatever was created before the step that failed
        return fail(RT_ERR_HIP, "context setup failed: %s", hipGetErrorString(e));
    }
    *out = c;
    return RT_OK;
}

int rt_ctx_destroy(rt_ctx* c) {
    if (!c) return RT_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (Pipe& p : c->pipes) destroy_pipe(p);
    if (c->d_counters) (void)hipFree(c->d_counters);
    release_host_buffer(c);
    if (c->d_partial) (void)hipFree(c->d_partial);
    for (float* p : c->peer_in) if (p) (void)hipFree(p);
    if (c->ev_multi_ready) (void)hipEventDestroy(c->ev_multi_ready);
    if (c->ev_multi_done) (void)hipEventDestroy(c->ev_multi_done);
    if (c->d_acc_cache) (void)hipFree(c->d_acc_cache);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->ev_upload) (void)hipEventDestroy(c->ev_upload);
    if (c->h_counts) (void)hipHostFree(c->h_counts);
    if (c->ev_counts) (void)hipEventDestroy(c->ev_counts);
    if (c->ev_begin) (void)hipEventDestroy(c->ev_begin);
    if (c->ev_end) (void)hipEventDestroy(c->ev_end);
    for (auto& p : c->ev_pool) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    delete c;
    return RT_OK;
}

int rt_ctx_set_stream(rt_ctx* c, void* s) {
    if (!c) return fail(RT_ERR_INVALID_ARGUMENT, "ctx is NULL");
    c->stream = (hipStream_t)s;
    return RT_OK;
}

int rt_ctx_set_batch_paths(rt_ctx* c, uint64_t n) {
    if (!c) return fail(RT_ERR_INVALID_ARGUMENT, "ctx is NULL");
    if (n != 0 && n < 64) return fail(RT_ERR_INVALID_ARGUMENT, "batch of %llu paths is below one wavefront", (unsigned long long)n);
    if (n >= (1ull << 31)) return fail(RT_ERR_UNSUPPORTED, "batch above 2^31 paths");
    c->batch_paths = n ? n : DEFAULT_BATCH_PATHS;
    return RT_OK;
}

int rt_ctx_set_pipelines(rt_ctx* c, int n) {
    if (!c) return fail(RT_ERR_INVALID_ARGUMENT, "ctx is NULL");
    if (n < 1 || n > MAX_PIPES) return fail(RT_ERR_INVALID_ARGUMENT, "%d pipelines: 1 to %d are supported", n, MAX_PIPES);
    c->n_pipes = n;
    return RT_OK;
}

int rt_ctx_set_tail_start(rt_ctx* c, int first_bounce) {
    if (!c) return fail(RT_ERR_INVALID_ARGUMENT, "ctx is NULL");
    c->tail_start = first_bounce < 0 ? -1 : first_bounce;
    return RT_OK;
}

int rt_ctx_set_counters(rt_ctx* c, int on) {
    if (!c) return fail(RT_ERR_INVALID_ARGUMENT, "ctx is NULL");
    c->counters_on = on != 0;
    return RT_OK;
}

int rt_ctx_get_counters(rt_ctx* c, rt_counters* out, int reset) {
    if (!c || !out) return fail(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    int e = set_device(c);
    if (e) return e;
    HIP_TRY(hipStreamSynchronize(c->stream));
    static_assert(sizeof(rt_counters) == sizeof(DCounters), "counter layouts differ");
    HIP_TRY(hipMemcpy(out, c->d_counters, sizeof(DCounters), hipMemcpyDeviceToHost));
    if (reset) HIP_TRY(hipMemset(c->d_counters, 0, sizeof(DCounters)));
    return RT_OK;
}

int rt_ctx_synchronize(rt_ctx* c) {
    if (!c) return fail(RT_ERR_INVALID_ARGUMENT, "ctx is NULL");
    int e = set_device(c);
    if (e) return e;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return collect_phases(c);
}

int rt_ctx_last_render_ms(rt_ctx* c, float* ms) {
    if (!c || !ms) return fail(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!c->have_render_time) return fail(RT_ERR_INVALID_ARGUMENT, "no render has been enqueued on this context");
    HIP_TRY(hipEventSynchronize(c->ev_end));
    HIP_TRY(hipEventElapsedTime(ms, c->ev_begin, c->ev_end));
    return RT_OK;
}

int rt_ctx_set_phase_timing(rt_ctx* c, int on) {
    if (!c) return fail(RT_ERR_INVALID_ARGUMENT, "ctx is NULL");
    c->phase_timing = on != 0;
    return RT_OK;
}

int rt_ctx_phase_ms(rt_ctx* c, int phase, float* ms, uint32_t* launches) {
    if (!c || phase < 0 || phase >= N_PHASES) return fail(RT_ERR_INVALID_ARGUMENT, "bad context or phase");
    if (c->phase_pending) return fail(RT_ERR_INVALID_ARGUMENT, "call rt_ctx_synchronize before reading phase times");
    if (ms) *ms = c->phase_ms[phase];
    if (launches) *launches = c->phase_launches[phase];
    return RT_OK;
}

int rt_buffer_create(rt_ctx* c, size_t bytes, void** d_out) {
    if (!c || !d_out) return fail(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    *d_out = nullptr;
    int e = set_device(c);
    if (e) return e;
    HIP_TRY(hipMalloc(d_out, std::max<size_t>(bytes, 16)));
    return RT_OK;
}

int rt_host_buffer_create(rt_ctx* c, size_t bytes, void** out) {
    if (!c || !out) return fail(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = nullptr;
    int e = set_device(c);
    if (e) return e;
    HIP_TRY(hipHostMalloc(out, std::max<size_t>(bytes, 16), hipHostMallocDefault));
    return RT_OK;
}

int rt_host_buffer_destroy(rt_ctx* c, void* ptr) {
    if (!c) return fail(RT_ERR_INVALID_ARGUMENT, "ctx is NULL");
    if (!ptr) return RT_OK;
    int e = set_device(c);
    if (e) return e;
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipHostFree(ptr));
    return RT_OK;
}

int rt_buffer_destroy(rt_ctx* c, void* d_ptr) {
    if (!c) return fail(RT_ERR_INVALID_ARGUMENT, "ctx is NULL");
    if (!d_ptr) return RT_OK;
    int e = set_device(c);
    if (e) return e;
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipFree(d_ptr));
    return RT_OK;
}

int rt_buffer_zero(rt_ctx* c, void* d_ptr, size_t bytes) {
    if (!c || !d_ptr) return fail(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    int e = set_device(c);
    if (e) return e;
    HIP_TRY(hipMemsetAsync(d_ptr, 0, bytes, c->stream));
    return RT_OK;
}

int rt_buffer_upload(rt_ctx* c, void* d_dst, const void* src, size_t bytes) {
    if (!c || !d_dst || !src) return fail(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    int e = set_device(c);
    if (e) return e;
    HIP_TRY(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_buffer_download(rt_ctx* c, void* dst, const void* d_src, size_t bytes) {
    if (!c || !dst || !d_src) return fail(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    int e = set_device(c);
    if (e) return e;
    HIP_TRY(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return collect_phases(c);
}

int rt_scene_create(rt_ctx* ctx, const rt_scene_desc* desc, rt_scene** out) {
    if (!ctx || !desc || !out) return fail(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = nullptr;
    int e = validate(desc);
    if (e) return e;
    if ((e = set_device(ctx))) return e;
    const auto t_begin = std::chrono::steady_clock::now();

    std::vector<Xf> xfs(desc->n_transforms);
    std::vector<float4> xforms(2 * (size_t)desc->n_transforms);
    for (uint32_t i = 0; i < desc->n_transforms; ++i) {
        const rt_transform& t = desc->transforms[i];
        xfs[i].s = t.sin_theta; xfs[i].c = t.cos_theta; xfs[i].identity = false;
        for (int a = 0; a < 3; ++a) xfs[i].off[a] = t.offset[a];
        xforms[2 * (size_t)i] = make_float4(t.sin_theta, t.cos_theta, 0.0f, 0.0f);
        xforms[2 * (size_t)i + 1] = make_float4(t.offset[0], t.offset[1], t.offset[2], 0.0f);
    }
    const Xf ident;
    auto xf_of = [&](uint32_t t) -> const Xf& { return t == RT_NO_TRANSFORM ? ident : xfs[t]; };

    // bake placements into world-space primitives
    std::vector<float4> spheres(desc->n_spheres), moving(2 * (size_t)desc->n_moving_spheres), quads(4 * (size_t)desc->n_quads);
    std::vector<uint4> sphere_meta(2 * (size_t)desc->n_spheres), moving_meta(2 * (size_t)desc->n_moving_spheres),
        quad_meta(2 * (size_t)desc->n_quads), medium_meta(2 * (size_t)desc->n_media);
    // shading record of a leaf: see the layout notes in rt_device.h
    auto fill_meta = [&](uint4* m, uint32_t material, uint32_t prim_id, uint32_t transform) {
        const rt_material& mat = desc->materials[material];
        const rt_texture& tex = desc->textures[mat.texture];
        uint32_t cls;
        const bool plain_checker = tex.kind == RT_TEX_CHECKER && desc->textures[tex.even].kind == RT_TEX_SOLID &&
                                   desc->textures[tex.odd].kind == RT_TEX_SOLID;
        if (plain_checker && mat.kind == RT_MAT_LAMBERTIAN) cls = CLS_CHECKER;
        else if (tex.kind != RT_TEX_SOLID) cls = CLS_TEXTURED;
        else switch (mat.kind) {
            case RT_MAT_DIFFUSE_LIGHT: cls = CLS_LIGHT; break;
            case RT_MAT_LAMBERTIAN: cls = CLS_LAMBERTIAN; break;
            case RT_MAT_METAL: cls = CLS_METAL; break;
            case RT_MAT_DIELECTRIC: cls = CLS_DIELECTRIC; break;
            default: cls = CLS_ISOTROPIC; break;
        }
        m[0] = make_uint4(material, prim_id, transform, cls | (mat.kind << 8) | (tex.kind << 16));
        m[1] = make_uint4(__builtin_bit_cast(uint32_t, mat.param), __builtin_bit_cast(uint32_t, tex.colour[0]),
                          __builtin_bit_cast(uint32_t, tex.colour[1]), __builtin_bit_cast(uint32_t, tex.colour[2]));
    };
    std::vector<Aabb> sphere_box(desc->n_spheres), moving_box(desc->n_moving_spheres), quad_box(desc->n_quads), medium_box(desc->n_media);
    for (uint32_t i = 0; i < desc->n_spheres; ++i) {
        const rt_sphere& s = desc->spheres[i];
        double c[3];
        point_to_world(xf_of(s.transform), s.centre, c);
        spheres[i] = make_float4(s.centre[0], s.centre[1], s.centre[2], s.radius);
        fill_meta(&sphere_meta[2 * (size_t)i], s.material, s.prim_id, s.transform);
        if (s.transform == RT_NO_TRANSFORM) {
            float lo[3] = {s.centre[0] - s.radius, s.centre[1] - s.radius, s.centre[2] - s.radius};
            float hi[3] = {s.centre[0] + s.radius, s.centre[1] + s.radius, s.centre[2] + s.radius};
            sphere_box[i] = Aabb::from_points(lo, hi);
        } else {
            sphere_box[i].grow(c, std::fabs((double)s.radius));
        }
    }
    for (uint32_t i = 0; i < desc->n_moving_spheres; ++i) {
        const rt_moving_sphere& s = desc->moving_spheres[i];
        double c0[3], dc[3];
        point_to_world(xf_of(s.transform), s.centre0, c0);
        vec_to_world(xf_of(s.transform), s.dcentre, dc);
        moving[2 * (size_t)i] = make_float4(s.centre0[0], s.centre0[1], s.centre0[2], s.radius);
        moving[2 * (size_t)i + 1] = make_float4(s.dcentre[0], s.dcentre[1], s.dcentre[2], 0.0f);
        fill_meta(&moving_meta[2 * (size_t)i], s.material, s.prim_id, s.transform);
        // union of the boxes at time 0 and 1 (src/hittable.rs:166-171)
        if (s.transform == RT_NO_TRANSFORM) {
            float r = s.radius, lo[3], hi[3], lo1[3], hi1[3];
            for (int a = 0; a < 3; ++a) {
                float at0 = s.centre0[a] + 0.0f * s.dcentre[a], at1 = s.centre0[a] + 1.0f * s.dcentre[a];  // centre.at(0), centre.at(1)
                lo[a] = at0 - r; hi[a] = at0 + r; lo1[a] = at1 - r; hi1[a] = at1 + r;
            }
            moving_box[i] = Aabb::from_aabbs(Aabb::from_points(lo, hi), Aabb::from_points(lo1, hi1));
        } else {
            double p1[3] = {c0[0] + dc[0], c0[1] + dc[1], c0[2] + dc[2]};
            moving_box[i].grow(c0, std::fabs((double)s.radius));
            moving_box[i].grow(p1, std::fabs((double)s.radius));
        }
    }
    for (uint32_t i = 0; i < desc->n_quads; ++i) {
        const rt_quad& qd = desc->quads[i];
        double q[3], u[3], v[3];
        point_to_world(xf_of(qd.transform), qd.q, q);
        vec_to_world(xf_of(qd.transform), qd.u, u);
        vec_to_world(xf_of(qd.transform), qd.v, v);
        // Per-quad constants of Quad::hit (src/hittable.rs:373-375) in the reference's f32
        // operation order, from the object-space q, u, v.
        const float* qf = qd.q;
        const float* uf = qd.u;
        const float* vf = qd.v;
        auto crossf = [](const float* l, const float* r, float* o3) {
            o3[0] = l[1] * r[2] - l[2] * r[1]; o3[1] = l[2] * r[0] - l[0] * r[2]; o3[2] = l[0] * r[1] - l[1] * r[0];
        };
        auto dotf = [](const float* l, const float* r) { return ((l[0] * r[0] + l[1] * r[1]) + l[2] * r[2]) + 0.0f; };
        float n[3], N[3], w[3];
        crossf(uf, vf, n);
        float inv_len = 1.0f / std::sqrt(dotf(n, n));
        for (int a = 0; a < 3; ++a) N[a] = inv_len * n[a];
        float D = dotf(N, qf);
        float inv_nn = 1.0f / dotf(n, n);
        for (int a = 0; a < 3; ++a) w[a] = inv_nn * n[a];
        float4* o = &quads[4 * (size_t)i];
        o[0] = make_float4(N[0], N[1], N[2], D);
        o[1] = make_float4(qf[0], qf[1], qf[2], w[0]);
        o[2] = make_float4(uf[0], uf[1], uf[2], w[1]);
        o[3] = make_float4(vf[0], vf[1], vf[2], w[2]);
        fill_meta(&quad_meta[2 * (size_t)i], qd.material, qd.prim_id, qd.transform);
        if (qd.transform == RT_NO_TRANSFORM) {
            float quv[3], qu[3], qv[3];
            for (int a = 0; a < 3; ++a) { qu[a] = qf[a] + uf[a]; qv[a] = qf[a] + vf[a]; quv[a] = qu[a] + vf[a]; }
            quad_box[i] = Aabb::from_aabbs(Aabb::from_points(qf, quv), Aabb::from_points(qu, qv));
        } else {
            double corner[4][3];
            for (int a = 0; a < 3; ++a) { corner[0][a] = q[a]; corner[1][a] = q[a] + u[a]; corner[2][a] = q[a] + v[a]; corner[3][a] = q[a] + u[a] + v[a]; }
            for (auto& cpt : corner) quad_box[i].grow(cpt);
        }
    }
    auto box_of = [&](const rt_prim_ref& r) -> const Aabb& {
        switch (r.kind) {
            case RT_PRIM_SPHERE: return sphere_box[r.index];
            case RT_PRIM_MOVING_SPHERE: return moving_box[r.index];
            case RT_PRIM_QUAD: return quad_box[r.index];
            default: return medium_box[r.index];
        }
    };
    std::vector<DMedium> media(desc->n_media);
    std::vector<uint32_t> boundary_refs(desc->n_boundary);
    auto placed = [&](const rt_prim_ref& r) {
        switch (r.kind) {
            case RT_PRIM_SPHERE: return desc->spheres[r.index].transform != RT_NO_TRANSFORM;
            case RT_PRIM_MOVING_SPHERE: return desc->moving_spheres[r.index].transform != RT_NO_TRANSFORM;
            case RT_PRIM_QUAD: return desc->quads[r.index].transform != RT_NO_TRANSFORM;
            default: return false;
        }
    };
    for (uint32_t i = 0; i < desc->n_boundary; ++i) boundary_refs[i] = make_leaf(desc->boundary[i].kind, desc->boundary[i].index, placed(desc->boundary[i]));
    // The reference's own box of every boundary leaf, in the leaf's space (a placed leaf sits in object space under its
    // Move / RotateY there too): Hittable::calc_aabb for a sphere, a moving sphere and a quad (src/hittable.rs:160-175).
    // A boundary of several leaves is a BVH in the reference, whose nodes test these boxes before the leaves.
    std::vector<float4> boundary_boxes(2 * (size_t)desc->n_boundary);
    for (uint32_t i = 0; i < desc->n_boundary; ++i) {
        const rt_prim_ref& r = desc->boundary[i];
        Aabb own;
        if (r.kind == RT_PRIM_SPHERE) {
            const rt_sphere& sp = desc->spheres[r.index];
            float lo[3] = {sp.centre[0] - sp.radius, sp.centre[1] - sp.radius, sp.centre[2] - sp.radius};
            float hi[3] = {sp.centre[0] + sp.radius, sp.centre[1] + sp.radius, sp.centre[2] + sp.radius};
            own = Aabb::from_points(lo, hi);
        } else if (r.kind == RT_PRIM_MOVING_SPHERE) {
            const rt_moving_sphere& sp = desc->moving_spheres[r.index];
            float rad = sp.radius, lo[3], hi[3], lo1[3], hi1[3];
            for (int a = 0; a < 3; ++a) {
                float at0 = sp.centre0[a] + 0.0f * sp.dcentre[a], at1 = sp.centre0[a] + 1.0f * sp.dcentre[a];
                lo[a] = at0 - rad; hi[a] = at0 + rad; lo1[a] = at1 - rad; hi1[a] = at1 + rad;
            }
            own = Aabb::from_aabbs(Aabb::from_points(lo, hi), Aabb::from_points(lo1, hi1));
        } else {
            const rt_quad& qd = desc->quads[r.index];
            float quv[3], qu[3], qv[3];
            for (int a = 0; a < 3; ++a) { qu[a] = qd.q[a] + qd.u[a]; qv[a] = qd.q[a] + qd.v[a]; quv[a] = qu[a] + qd.v[a]; }
            own = Aabb::from_aabbs(Aabb::from_points(qd.q, quv), Aabb::from_points(qu, qv));
        }
        boundary_boxes[2 * (size_t)i] = make_float4(own.mn[0], own.mn[1], own.mn[2], 0.0f);
        boundary_boxes[2 * (size_t)i + 1] = make_float4(own.mx[0], own.mx[1], own.mx[2], 0.0f);
    }
    bool media_are_single_spheres = true;
    for (uint32_t i = 0; i < desc->n_media; ++i) {
        const rt_medium& m = desc->media[i];
        if (m.n_boundary != 1 || desc->boundary[m.first_boundary].kind != RT_PRIM_SPHERE || placed(desc->boundary[m.first_boundary]))
            media_are_single_spheres = false;
        auto xform_of = [&](const rt_prim_ref& r) {
            return r.kind == RT_PRIM_SPHERE ? desc->spheres[r.index].transform
                 : r.kind == RT_PRIM_MOVING_SPHERE ? desc->moving_spheres[r.index].transform : desc->quads[r.index].transform;
        };
        uint32_t shared = m.n_boundary ? xform_of(desc->boundary[m.first_boundary]) : RT_NO_TRANSFORM;
        for (uint32_t k = 1; k < m.n_boundary; ++k)
            if (xform_of(desc->boundary[m.first_boundary + k]) != shared) shared = rt::XFORM_MIXED;
        media[i] = DMedium{m.first_boundary, m.n_boundary, m.neg_inv_density, m.phase_material, m.prim_id, 0u, shared, 0u, make_float4(0, 0, 0, 0)};
        if (m.n_boundary == 1 && desc->boundary[m.first_boundary].kind == RT_PRIM_SPHERE && !placed(desc->boundary[m.first_boundary])) {
            const rt_sphere& bs = desc->spheres[desc->boundary[m.first_boundary].index];
            media[i].single_sphere = 1u;
            media[i].sphere = make_float4(bs.centre[0], bs.centre[1], bs.centre[2], bs.radius);
        }
        fill_meta(&medium_meta[2 * (size_t)i], m.phase_material, m.prim_id, RT_NO_TRANSFORM);
        for (uint32_t k = 0; k < m.n_boundary; ++k) medium_box[i].grow(box_of(desc->boundary[m.first_boundary + k]));
    }

    // flattened BVH over the world leaves; leaves that span most of the scene (a ground sphere,
    // a scene-wide fog) would sit at the top of any tree, so they are tested up front instead
    std::vector<BuildPrim> prims;
    prims.reserve(desc->n_world);
    Aabb whole;
    for (uint32_t i = 0; i < desc->n_world; ++i) {
        const Aabb& b = box_of(desc->world[i]);
        for (int a = 0; a < 3; ++a)
            if (!(b.mn[a] <= b.mx[a]) || !std::isfinite(b.mn[a]) || !std::isfinite(b.mx[a]))
                return fail(RT_ERR_INVALID_ARGUMENT, "world[%u] has a non-finite or empty bounding box", i);
        whole.grow(b);
    }
    auto half_area = [](const Aabb& b) {
        double dx = (double)b.mx[0] - b.mn[0], dy = (double)b.mx[1] - b.mn[1], dz = (double)b.mx[2] - b.mn[2];
        return dx * dy + dy * dz + dz * dx;
    };
    uint32_t n_global = 0, global_refs[7] = {0, 0, 0, 0, 0, 0, 0};
    for (uint32_t i = 0; i < desc->n_world; ++i) {
        const rt_prim_ref& w = desc->world[i];
        const Aabb& b = box_of(w);
        // the reference carries the leaf's shading class (bits 0..7 of the fourth word of its meta record)
        const uint4* meta_of_kind = w.kind == RT_PRIM_SPHERE ? sphere_meta.data() : w.kind == RT_PRIM_MOVING_SPHERE ? moving_meta.data()
                                  : w.kind == RT_PRIM_QUAD ? quad_meta.data() : medium_meta.data();
        uint32_t ref = make_leaf(w.kind, w.index, placed(w), meta_of_kind[2 * (size_t)w.index].w & 0xFFu);
        bool eligible = (w.kind == RT_PRIM_MEDIUM || (w.kind == RT_PRIM_SPHERE && !placed(w))) && desc->n_world > 4;
        if (eligible && n_global < 7 && half_area(b) >= 0.25 * half_area(whole)) {
            global_refs[n_global++] = ref;
            continue;
        }
        BuildPrim bp;
        for (int a = 0; a < 3; ++a) { bp.bmin[a] = b.mn[a]; bp.bmax[a] = b.mx[a]; }
        bp.ref = ref;
        bp.exact = (w.kind != RT_PRIM_MEDIUM && !placed(w)) ? 1u : 0u;
        prims.push_back(bp);
    }
    // Rays start at the camera or on a leaf.  Where all the leaves of the tree lie inside the box of a sphere that is tested
    // first, nearly every ray starts inside that box and its box test can be decided without a division (sphere_box_test in
    // the kernels tries that first); where they do not (a ground sphere below the scene), trying would be wasted work.
    uint32_t global_inside = 0;
    {
        Aabb rest;
        for (const BuildPrim& bp : prims)
            for (int a = 0; a < 3; ++a) { rest.mn[a] = std::min(rest.mn[a], bp.bmin[a]); rest.mx[a] = std::max(rest.mx[a], bp.bmax[a]); }
        for (uint32_t g = 0; g < n_global; ++g) {
            if (ref_kind(global_refs[g]) != KIND_SPHERE) continue;
            const Aabb& b = sphere_box[ref_index(global_refs[g])];
            bool inside = !prims.empty();
            for (int a = 0; a < 3; ++a) inside = inside && b.mn[a] <= rest.mn[a] && rest.mx[a] <= b.mx[a];
            if (inside) global_inside |= 1u << g;
        }
    }
    FlatBvh bvh = build_bvh(prims, 0);
    if (bvh.stack_need > rt::TRAVERSAL_STACK_ENTRIES)
        return fail(RT_ERR_UNSUPPORTED, "the tree (depth %u) can need %u traversal stack entries, %u are available", bvh.max_depth,
                    bvh.stack_need, rt::TRAVERSAL_STACK_ENTRIES);
    {   // every reference the kernels will follow must point inside its array
        auto ref_ok = [&](uint32_t r) {
            if (!ref_is_leaf(r)) return r < bvh.n_nodes;
            uint32_t idx = ref_index(r);
            switch (ref_kind(r)) {
                case KIND_SPHERE: return idx < desc->n_spheres;
                case KIND_MOVING: return idx < desc->n_moving_spheres;
                case KIND_QUAD: return idx < desc->n_quads;
                default: return idx < desc->n_media && !ref_placed(r);
            }
        };
        bool ok = bvh.root == REF_NONE || ref_ok(bvh.root);
        for (uint32_t i = 0; ok && i < bvh.n_nodes; ++i) {
            for (int k = 0; ok && k < rt::BVH_WIDTH; ++k) {
                uint32_t r;
                std::memcpy(&r, &bvh.nodes[rt::NODE_FLOATS * (size_t)i + rt::NODE_REFS_AT + k], 4);
                const bool empty_slot = !(bvh.nodes[rt::NODE_FLOATS * (size_t)i + rt::NODE_SLACK_AT + k] > 0.0f);
                ok = empty_slot || ref_ok(r);
            }
        }
        for (uint32_t i = 0; ok && i < n_global; ++i) ok = ref_ok(global_refs[i]);
        if (!ok) return fail(RT_ERR_HIP, "internal error: the flattened BVH holds an invalid reference");
    }

    std::vector<DMaterial> materials(desc->n_materials);
    for (uint32_t i = 0; i < desc->n_materials; ++i) materials[i] = DMaterial{desc->materials[i].kind, desc->materials[i].param, desc->materials[i].texture, 0};
    std::vector<DTexture> textures(desc->n_textures);
    for (uint32_t i = 0; i < desc->n_textures; ++i) {
        const rt_texture& t = desc->textures[i];
        textures[i] = DTexture{t.kind, t.colour[0], t.colour[1], t.colour[2], t.scale, t.even, t.odd, t.image, t.perlin, {0, 0, 0}};
    }

    rt_scene* sc = new rt_scene();
    sc->ctx = ctx;
    sc->bvh_depth = bvh.max_depth;
    sc->stack_need = bvh.stack_need;
    sc->media_are_single_spheres = media_are_single_spheres;
    sc->d.generic_media = media_are_single_spheres ? 0u : 1u;
    auto bail = [&](int code) { rt_scene_destroy(sc); return code; };

#if RT_BVH_WIDTH == 4
    // the kernels' form of a node (rt_bvh.h, PackedNode): 64 bytes, the children's boxes on the node's own 8-bit grid
    std::vector<float4> nodes(4 * (size_t)bvh.n_nodes);
    {
        const std::vector<rt::PackedNode> packed = rt::pack_nodes(bvh);
        if (rt::check_packed_nodes(bvh, packed).violations != 0) return bail(fail(RT_ERR_HIP, "internal error: a packed BVH node does not contain its child"));
        static_assert(sizeof(rt::PackedNode) == 4 * sizeof(float4), "a packed node is four 16-byte groups");
        if (!packed.empty()) std::memcpy(nodes.data(), packed.data(), packed.size() * sizeof(rt::PackedNode));
    }
#else
    std::vector<float4> nodes(bvh.nodes.size() / 4);
    if (!nodes.empty()) std::memcpy(nodes.data(), bvh.nodes.data(), bvh.nodes.size() * sizeof(float));
#endif
    if ((e = upload(sc, nodes, &sc->d.nodes))) return bail(e);
    {   // The reference's own box of every leaf that sits in world space (Hittable::calc_aabb, src/hittable.rs:158-178): the
        // kernels test it - as AABBx2::hit would (src/math/aabb.rs:127-167) - before they accept a hit on such a leaf, because
        // that test decides in the reference whether the leaf is looked at at all.
        auto boxes_of = [&](const std::vector<Aabb>& boxes, const float4** out) -> int {
            std::vector<float4> flat(2 * boxes.size());
            for (size_t i = 0; i < boxes.size(); ++i) {
                flat[2 * i] = make_float4(boxes[i].mn[0], boxes[i].mn[1], boxes[i].mn[2], 0.0f);
                flat[2 * i + 1] = make_float4(boxes[i].mx[0], boxes[i].mx[1], boxes[i].mx[2], 0.0f);
            }
            return upload(sc, flat, out);
        };
        // a sphere's box that is just centre -+ |radius| (none of its sides was padded) is computed by the kernels
        bool plain = true;
        for (uint32_t i = 0; i < desc->n_spheres && plain; ++i) {
            const rt_sphere& s = desc->spheres[i];
            if (s.transform != RT_NO_TRANSFORM) continue;  // (a placed sphere's box is not tested by the kernels)
            const float ar = std::fabs(s.radius);
            for (int a = 0; a < 3; ++a) plain = plain && sphere_box[i].mn[a] == s.centre[a] - ar && sphere_box[i].mx[a] == s.centre[a] + ar;
        }
        sc->d.plain_sphere_boxes = plain ? 1u : 0u;
        if (!plain && (e = boxes_of(sphere_box, &sc->d.sphere_boxes))) return bail(e);
        if ((e = boxes_of(moving_box, &sc->d.moving_boxes))) return bail(e);
        if ((e = boxes_of(quad_box, &sc->d.quad_boxes))) return bail(e);
    }
    if ((e = upload(sc, spheres, &sc->d.spheres))) return bail(e);
    if ((e = upload(sc, sphere_meta, &sc->d.sphere_meta))) return bail(e);
    if ((e = upload(sc, moving, &sc->d.moving))) return bail(e);
    if ((e = upload(sc, moving_meta, &sc->d.moving_meta))) return bail(e);
    if ((e = upload(sc, quads, &sc->d.quads))) return bail(e);
    if ((e = upload(sc, quad_meta, &sc->d.quad_meta))) return bail(e);
    if ((e = upload(sc, media, &sc->d.media))) return bail(e);
    if ((e = upload(sc, medium_meta, &sc->d.medium_meta))) return bail(e);
    if ((e = upload(sc, boundary_refs, &sc->d.boundary_refs))) return bail(e);
    if ((e = upload(sc, boundary_boxes, &sc->d.boundary_boxes))) return bail(e);
    if ((e = upload(sc, xforms, &sc->d.xforms))) return bail(e);
    {   // per-leaf copies of the placements, for the kinds that have placed leaves
        auto per_leaf = [&](size_t n, auto transform_of, const float4** out) -> int {
            *out = nullptr;
            bool any = false;
            for (size_t i = 0; i < n && !any; ++i) any = transform_of(i) != RT_NO_TRANSFORM;
            if (!any) return RT_OK;
            std::vector<float4> copy(2 * n, make_float4(0.0f, 1.0f, 0.0f, 0.0f));
            for (size_t i = 0; i < n; ++i) {
                uint32_t t = transform_of(i);
                if (t == RT_NO_TRANSFORM) continue;
                copy[2 * i] = xforms[2 * (size_t)t];
                copy[2 * i + 1] = xforms[2 * (size_t)t + 1];
            }
            return upload(sc, copy, out);
        };
        if ((e = per_leaf(desc->n_spheres, [&](size_t i) { return desc->spheres[i].transform; }, &sc->d.sphere_xf))) return bail(e);
        if ((e = per_leaf(desc->n_moving_spheres, [&](size_t i) { return desc->moving_spheres[i].transform; }, &sc->d.moving_xf))) return bail(e);
        if ((e = per_leaf(desc->n_quads, [&](size_t i) { return desc->quads[i].transform; }, &sc->d.quad_xf))) return bail(e);
    }
    if ((e = upload(sc, materials, &sc->d.materials))) return bail(e);
    if ((e = upload(sc, textures, &sc->d.textures))) return bail(e);
    {   // the two colours and the scale of every material that shades as CLS_CHECKER
        std::vector<DChecker> checkers(desc->n_materials, DChecker{0.0f, {0, 0, 0}, {0, 0, 0}, 0.0f});
        for (uint32_t i = 0; i < desc->n_materials; ++i) {
            const DTexture& t = textures[desc->materials[i].texture];
            if (t.kind != RT_TEX_CHECKER || textures[t.even].kind != RT_TEX_SOLID || textures[t.odd].kind != RT_TEX_SOLID) continue;
            const DTexture &te = textures[t.even], &to = textures[t.odd];
            checkers[i] = DChecker{t.scale, {te.r, te.g, te.b}, {to.r, to.g, to.b}, 0.0f};
        }
        if ((e = upload(sc, checkers, &sc->d.checkers))) return bail(e);
    }

    std::vector<DImage> images(desc->n_images);
    for (uint32_t i = 0; i < desc->n_images; ++i) {
        const rt_image& im = desc->images[i];
        size_t n = (size_t)im.width * im.height;
        images[i].width = im.width; images[i].height = im.height; images[i].format = im.format; images[i]._pad = 0;
        if (im.format == RT_IMAGE_RGB8) {
            std::vector<uchar4> px(n);
            const uint8_t* src = (const uint8_t*)im.texels;
            for (size_t k = 0; k < n; ++k) px[k] = make_uchar4(src[3 * k], src[3 * k + 1], src[3 * k + 2], 255);
            const uchar4* dptr;
            if ((e = upload(sc, px, &dptr))) return bail(e);
            images[i].texels = dptr;
        } else {
            std::vector<float4> px(n);
            const float* src = (const float*)im.texels;
            for (size_t k = 0; k < n; ++k) px[k] = make_float4(src[3 * k], src[3 * k + 1], src[3 * k + 2], 0.0f);
            const float4* dptr;
            if ((e = upload(sc, px, &dptr))) return bail(e);
            images[i].texels = dptr;
        }
    }
    if ((e = upload(sc, images, &sc->d.images))) return bail(e);

    std::vector<DPerlin> perlins(desc->n_perlins);
    for (uint32_t i = 0; i < desc->n_perlins; ++i) {
        const rt_perlin& p = desc->perlins[i];
        // tables expanded so that the kernels index them without the reference's `% point_count` (layout: DPerlin)
        size_t pow2 = 1;
        while (pow2 < p.point_count) pow2 <<= 1;
        std::vector<float4> vecs(pow2);
        std::vector<uint32_t> perm(3 * 256);
        for (size_t v = 0; v < pow2; ++v) {
            const size_t k = v % p.point_count;
            vecs[v] = make_float4(p.rand_vecs[3 * k], p.rand_vecs[3 * k + 1], p.rand_vecs[3 * k + 2], 0.0f);
        }
        for (uint32_t v = 0; v < 256; ++v) {
            const uint32_t k = v % p.point_count;
            perm[v] = p.perm_x[k];
            perm[256 + v] = p.perm_y[k];
            perm[512 + v] = p.perm_z[k];
        }
        if ((e = upload(sc, vecs, &perlins[i].vecs))) return bail(e);
        if ((e = upload(sc, perm, &perlins[i].perm))) return bail(e);
        perlins[i].point_count = p.point_count;
        perlins[i].n_vecs = (uint32_t)pow2;
    }
    sc->d.lds_perlin = (desc->n_perlins == 1 && perlins[0].n_vecs <= LDS_PERLIN_VECS) ? 1u : 0u;
    if ((e = upload(sc, perlins, &sc->d.perlins))) return bail(e);

    sc->d.root = bvh.root;
    sc->d.n_nodes = bvh.n_nodes;
    sc->d.n_media = desc->n_media;
    sc->d.n_materials = desc->n_materials;
    // the second shade kernel is launched only when some leaf's hits are set aside for it
    for (const std::vector<uint4>* metas : {&sphere_meta, &moving_meta, &quad_meta, &medium_meta})
        for (size_t i = 0; i < metas->size(); i += 2)
            if (((*metas)[i].w & 0xFFu) == CLS_TEXTURED) sc->has_textures = true;
    sc->d.textured = sc->has_textures ? 1u : 0u;
    sc->info.bvh_nodes = bvh.n_nodes;
    sc->info.bvh_depth = bvh.max_depth;
    sc->info.leaves_in_tree = (uint32_t)prims.size();
    sc->info.leaves_tested_first = n_global;
    sc->info.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    sc->d.n_global = n_global;
    sc->d.global_inside = global_inside;
    {
        std::vector<uint32_t> g(global_refs, global_refs + 7);
        if ((e = upload(sc, g, &sc->d.global_refs))) return bail(e);
    }
    *out = sc;
    return RT_OK;
}

int rt_scene_get_info(const rt_scene* sc, rt_scene_info* out) {
    if (!sc || !out) return fail(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = sc->info;
    return RT_OK;
}

int rt_bvh_build_check(const float* boxes, uint32_t n, rt_bvh_report* out) {
    if (!out || (n && !boxes)) return fail(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n >= rt::MAX_PRIMS_PER_KIND) return fail(RT_ERR_UNSUPPORTED, "more than 2^25 leaves");
    std::vector<BuildPrim> prims(n);
    for (uint32_t i = 0; i < n; ++i) {
        for (int a = 0; a < 3; ++a) { prims[i].bmin[a] = boxes[6 * (size_t)i + a]; prims[i].bmax[a] = boxes[6 * (size_t)i + 3 + a]; }
        prims[i].ref = make_leaf(KIND_SPHERE, i, false);
        prims[i].exact = 1;
    }
    FlatBvh bvh = build_bvh(prims, 0);
    *out = rt_bvh_report{bvh.n_nodes, bvh.max_depth, (uint32_t)rt::BVH_WIDTH, 0, 0, 0.0f, bvh.stack_need, 1.0f};
    if (n == 0) { out->violations = bvh.root == REF_NONE ? 0 : 1; return RT_OK; }
    if (n == 1) { out->leaves_found = 1; out->violations = bvh.root == make_leaf(KIND_SPHERE, 0, false) ? 0 : 1; return RT_OK; }
    struct Box3 { float mn[3], mx[3]; };
    auto child_box = [&](uint32_t node, int k) {
        const float* q = &bvh.nodes[rt::NODE_FLOATS * (size_t)node];
        Box3 b;
        b.mn[0] = q[4 * k]; b.mn[1] = q[4 * k + 1]; b.mx[0] = q[4 * k + 2]; b.mx[1] = q[4 * k + 3];
        b.mn[2] = q[4 * rt::BVH_WIDTH + 2 * k]; b.mx[2] = q[4 * rt::BVH_WIDTH + 2 * k + 1];
        return b;
    };
    auto area = [](const Box3& b) {
        float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
        return dx * dy + dy * dz + dz * dx;
    };
    std::vector<uint8_t> seen(n, 0);
    uint32_t violations = 0, found = 0;
    double cost = 0.0;
    Box3 root_box{{FLT_MAX, FLT_MAX, FLT_MAX}, {-FLT_MAX, -FLT_MAX, -FLT_MAX}};
    for (uint32_t i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) {
            root_box.mn[a] = std::min(root_box.mn[a], boxes[6 * (size_t)i + a]);
            root_box.mx[a] = std::max(root_box.mx[a], boxes[6 * (size_t)i + 3 + a]);
        }
    const double root_area = std::max(1e-30, (double)area(root_box));
    // walk from the root; `bound` is the box the parent stores for this subtree
    struct Item { uint32_t node; Box3 bound; bool has_bound; };
    std::vector<Item> todo{{bvh.root, root_box, false}};
    if (ref_is_leaf(bvh.root) || bvh.root >= bvh.n_nodes) { out->violations = 1; return RT_OK; }
    std::vector<uint8_t> node_seen(bvh.n_nodes, 0);
    while (!todo.empty()) {
        Item it = todo.back();
        todo.pop_back();
        if (node_seen[it.node]++) { ++violations; continue; }
        for (int k = 0; k < rt::BVH_WIDTH; ++k) {
            const float slack = bvh.nodes[rt::NODE_FLOATS * (size_t)it.node + rt::NODE_SLACK_AT + k];
            if (!(slack > 0.0f)) continue;  // empty slot (NaN)
            uint32_t r;
            std::memcpy(&r, &bvh.nodes[rt::NODE_FLOATS * (size_t)it.node + rt::NODE_REFS_AT + k], 4);
            Box3 cb = child_box(it.node, k);
            cost += area(cb) / root_area;
            if (it.has_bound)
                for (int a = 0; a < 3; ++a)
                    if (cb.mn[a] < it.bound.mn[a] - 1e-3f * (1.0f + std::fabs(it.bound.mn[a])) || cb.mx[a] > it.bound.mx[a] + 1e-3f * (1.0f + std::fabs(it.bound.mx[a]))) ++violations;
            if (ref_is_leaf(r)) {
                uint32_t idx = ref_index(r);
                if (ref_kind(r) != KIND_SPHERE || idx >= n || seen[idx]++) { ++violations; continue; }
                ++found;
                if (slack != 1.0f) ++violations;  // a leaf given as exact keeps its own box, tested without slack
                for (int a = 0; a < 3; ++a)
                    if (cb.mn[a] != boxes[6 * (size_t)idx + a] || cb.mx[a] != boxes[6 * (size_t)idx + 3 + a]) ++violations;
            } else {
                if (r >= bvh.n_nodes || r <= it.node) { ++violations; continue; }  // parents are numbered before their children
                todo.push_back(Item{r, cb, true});
            }
        }
    }
    for (uint32_t i = 0; i < bvh.n_nodes; ++i) if (!node_seen[i]) ++violations;
#if RT_BVH_WIDTH == 4
    {   // the 64-byte form the kernels read: every child's grid box must contain the child's box
        const rt::PackCheck pc = rt::check_packed_nodes(bvh, rt::pack_nodes(bvh));
        violations += pc.violations;
        out->packed_surface_ratio = (float)pc.mean_surface_ratio;
    }
#endif
    out->leaves_found = found;
    out->violations = violations + (found != n ? 1u : 0u);
    out->sah_cost = (float)cost;
    return RT_OK;
}

int rt_scene_destroy(rt_scene* sc) {
    if (!sc) return RT_OK;
    (void)hipSetDevice(sc->ctx->device);
    (void)hipStreamSynchronize(sc->ctx->stream);
    for (void* p : sc->allocations) (void)hipFree(p);
    if (sc->ctx->seen.scene == sc) sc->ctx->seen = {};
    delete sc;
    return RT_OK;
}

int rt_render_accumulate_device(rt_scene* sc, const rt_camera_desc* cam, uint64_t seed, uint32_t first_sample, uint32_t n_samples,
                                float* d_acc) {
    if (cam == nullptr) return fail(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    return rt_render_accumulate_rows_device(sc, cam, seed, first_sample, n_samples, 0, cam->height, d_acc);
}

// The host side of the wavefront loop.  fixed == nullptr: the context's pipelines, their queues (re)allocated as the
// job needs.  fixed != nullptr: that one pipeline, whose queues the caller has put into memory that exists already;
// nothing is allocated on this path (no queue growth, no new stream or event, no per-launch timing events).
// before_accumulate: an event the accumulation kernels wait for (the buffer's contents arriving), or nullptr.
static int render_rows(rt_scene* sc, const rt_camera_desc* cam, uint64_t seed, uint32_t first_sample, uint32_t n_samples,
                       uint32_t first_row, uint32_t n_rows, float* d_acc, Pipe* fixed, hipEvent_t before_accumulate = nullptr) {
    if (!sc || !d_acc) return fail(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    int e = check_camera(cam);
    if (e) return e;
    rt_ctx* c = sc->ctx;
    if ((e = set_device(c))) return e;
    if ((uint64_t)first_sample + n_samples > 0xFFFFFFFFull) return fail(RT_ERR_INVALID_ARGUMENT, "sample numbers overflow 32 bits");
    if ((uint64_t)first_row + n_rows > cam->height) return fail(RT_ERR_INVALID_ARGUMENT, "rows %u..%u outside the image of %u rows", first_row, first_row + n_rows, cam->height);
    if (n_rows == 0) return RT_OK;
    const uint64_t pix0 = (uint64_t)first_row * cam->width;     // the rows are one contiguous range of pixels
    const uint64_t npix = (uint64_t)cam->width * n_rows;
    const uint64_t all_paths = npix * (uint64_t)n_samples;
    // one pipeline for small jobs (and for single-sample calls, which are not split): halves of a call below 16 M paths are batches
    // so small that their own per-bounce latencies cost more than the overlap gives (Book-1 scene 400x225, depth 50, 100 samples per
    // call, 9 M paths: 2.96 ms on one pipeline, 3.32 on two; 18 M paths: 6.2 against 5.3; the Cornell box still gains 4 - 5 % from two
    // at 8 - 13 M paths, the Book-2 scene nothing)
    const int n_pipes = (!fixed && c->n_pipes > 1 && n_samples >= 2 && all_paths >= (1ull << 24)) ? c->n_pipes : 1;
    Pipe* pipes[MAX_PIPES];
    for (int i = 0; i < MAX_PIPES; ++i) pipes[i] = fixed ? fixed : &c->pipes[i];
    const bool timing = c->phase_timing && !fixed;
    // paths per batch: the configured number, but no more than fits the memory that is free now
    uint64_t batch_paths = fixed ? fixed->cap : c->batch_paths;
    if (!fixed) {
        uint64_t have = 0;
        for (int i = 0; i < n_pipes; ++i) have += c->pipes[i].cap;
        const uint64_t wanted = std::min<uint64_t>(batch_paths * n_pipes, all_paths);
        size_t free_b = 0, total_b = 0;
        if (wanted > have && hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            uint64_t usable = (uint64_t)((double)free_b * 0.9) + have * BYTES_PER_PATH;
            batch_paths = std::max<uint64_t>(std::min<uint64_t>(batch_paths, usable / BYTES_PER_PATH / n_pipes), 1u << 16);
        }
        (void)hipGetLastError();
    }
    const uint32_t chunk_pixels = (uint32_t)std::min<uint64_t>(npix, batch_paths);
    uint32_t spb = 1;  // samples per batch
    if (chunk_pixels == npix) {
        spb = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(batch_paths / npix, n_samples));
        if (n_pipes > 1) spb = std::min(spb, (n_samples + (uint32_t)n_pipes - 1) / (uint32_t)n_pipes);  // at least one batch per pipeline
    }
    // counts[0 .. max_depth + 1]: queue lengths per bounce; counts[max_depth + 2 + bounce]: textured hits of that bounce
    const uint32_t n_queue_counts = cam->max_depth + 2;
    const uint32_t n_counts = 2 * cam->max_depth + 2;
    if (fixed) {
        if (n_counts > fixed->counts_cap) return fail(RT_ERR_UNSUPPORTED, "max_depth %u is beyond what the preallocated queue counters hold", cam->max_depth);
    } else {
        for (int i = 0; i < n_pipes; ++i)
            if ((e = ensure_queues(c, c->pipes[i], (uint64_t)chunk_pixels * spb, n_counts, needs_overflow(sc)))) return e;
    }
    // persistent grid of the extend kernel; a preallocated overflow area bounds it for the trees that use that area
    uint32_t extend_grid = extend_grid_of(c, sc, c->counters_on);
    if (fixed && needs_overflow(sc)) extend_grid = std::min(extend_grid, NOALLOC_OVERFLOW_BLOCKS);
    // With two pipelines the extend kernel of one batch runs beside the other batch's kernels.  A grid that fills every
    // wave slot of the chip keeps those out until its waves drain; with half a block per CU fewer (6.5 instead of 7) the
    // other batch's ray generation, first-leaves and shade passes - bound by memory, not by instruction issue - find room
    // to run in its shadow (headline scene: 2680 Msamples/s with 7 blocks per CU, 2715 with 6.5, 2695 with 6, 2675 with 5.5).
    if (n_pipes > 1) extend_grid = std::min(extend_grid, c->extend_blocks_shared);
    const bool overflow_needed = needs_overflow(sc);  // else the extend kernel runs the variant without the global part of its stack
    DCamera dcam = to_dcamera(cam);
    DCounters* counters = c->counters_on ? c->d_counters : nullptr;
    rt::ExtendTuning tune = c->tune;
    if (tune.medium_min <= 0) tune.medium_min = sc->media_are_single_spheres ? 4 : 24;

    HIP_TRY(hipEventRecord(c->ev_begin, c->stream));
    if (n_pipes > 1)
        for (int i = 0; i < n_pipes; ++i) HIP_TRY(hipStreamWaitEvent(pipes[i]->stream, c->ev_begin, 0));
    uint32_t batch_no = 0, last_tail_start = 0, last_total = 0;
    Pipe* prev = nullptr;
    // all launches of the call; whatever it returns, the pipelines' streams are joined to the caller's afterwards
    auto enqueue = [&]() -> int {
        for (uint64_t p0 = 0; p0 < npix; p0 += chunk_pixels) {
            const uint32_t np = (uint32_t)std::min<uint64_t>(chunk_pixels, npix - p0);
            for (uint32_t s0 = 0; s0 < n_samples; s0 += spb, ++batch_no) {
                const uint32_t ns = std::min(spb, n_samples - s0);
                const BatchInfo b = rt::make_batch((uint32_t)(pix0 + p0), np, first_sample + s0, ns);
                const uint32_t total = np * ns;
                Pipe& q = *pipes[batch_no % (uint32_t)n_pipes];
                hipStream_t st = n_pipes > 1 ? q.stream : c->stream;
                EventPair* ep;
                if ((e = phase_begin(c, timing, st, 0, &ep))) return e;
                // leaves kept out of the tree: tested by ray generation for the camera rays (by the counting pass when
                // the work counters are on), by a pass of their own for the rays shade creates
                const bool raygen_first_leaves = sc->d.n_global != 0 && counters == nullptr;
                launch_raygen(st, c->max_blocks, sc->d, dcam, b, seed, q.ray_o[0], q.ray_d[0], q.thr[0], raygen_first_leaves ? q.hits : nullptr,
                              q.counts, n_counts);
                if ((e = phase_end(st, ep))) return e;
                int cur = 0;
                const uint32_t tail_start = tail_start_for(c, sc, cam, total, fixed == nullptr);
                last_tail_start = tail_start;
                last_total = total;
                for (uint32_t bounce = 0; bounce < tail_start; ++bounce) {
                    if (sc->d.n_global != 0 && !(bounce == 0 && raygen_first_leaves)) {
                        if ((e = phase_begin(c, timing, st, 4, &ep))) return e;
                        launch_first_leaves(st, c->max_blocks, total, sc->d, b, seed, bounce, q.counts, q.ray_o[cur], q.ray_d[cur], q.hits, counters);
                        if ((e = phase_end(st, ep))) return e;
                    }
                    if ((e = phase_begin(c, timing, st, 1, &ep))) return e;
                    launch_extend(st, extend_grid, total, sc->d, b, seed, bounce, q.counts, q.ray_o[cur],
                                  q.ray_d[cur], q.hits, overflow_needed ? q.stack_overflow : nullptr, tune, counters);
                    if ((e = phase_end(st, ep))) return e;
                    if ((e = phase_begin(c, timing, st, 2, &ep))) return e;
                    launch_shade(st, c->shade_blocks, total, sc->d, dcam, b, seed, bounce, q.counts, q.counts + n_queue_counts + bounce, q.tex_list,
                                 q.ray_o[cur], q.ray_d[cur], q.thr[cur], q.hits, q.ray_o[cur ^ 1], q.ray_d[cur ^ 1], q.thr[cur ^ 1], q.radiance,
                                 sc->has_textures);
                    if ((e = phase_end(st, ep))) return e;
                    cur ^= 1;
                }
                if (tail_start < cam->max_depth) {
                    if ((e = phase_begin(c, timing, st, 5, &ep))) return e;
                    launch_tail(st, std::min(extend_grid, c->tail_blocks), total, sc->d, dcam, b, seed, tail_start, q.counts, q.ray_o[cur], q.ray_d[cur], q.thr[cur], q.radiance,
                                q.stack_overflow);
                    if ((e = phase_end(st, ep))) return e;
                }
                if (cam->max_depth == 0) HIP_TRY(hipMemsetAsync(q.radiance, 0, (size_t)total * sizeof(float4), st));
                // the accumulation buffer takes the batches in their order: sample sums stay bit-identical
                // to a one-pipeline render
                if (n_pipes > 1 && prev && prev != &q) HIP_TRY(hipStreamWaitEvent(st, prev->done, 0));
                if (before_accumulate) HIP_TRY(hipStreamWaitEvent(st, before_accumulate, 0));
                if ((e = phase_begin(c, timing, st, 3, &ep))) return e;
                launch_accumulate(st, c->max_blocks, b, q.radiance, (float4*)d_acc);
                if ((e = phase_end(st, ep))) return e;
                if (n_pipes > 1) HIP_TRY(hipEventRecord(q.done, st));
                prev = &q;
            }
        }
        return RT_OK;
    };
    const int enqueued = enqueue();
    if (n_pipes > 1) {
        if (enqueued != RT_OK)  // a batch was cut short: its pipeline's stream still has to be waited for
            for (int i = 0; i < n_pipes; ++i) (void)hipEventRecord(pipes[i]->done, pipes[i]->stream);
        for (int i = 0; i < n_pipes && (enqueued != RT_OK || i < (int)batch_no); ++i) HIP_TRY(hipStreamWaitEvent(c->stream, pipes[i]->done, 0));
    }
    if (enqueued != RT_OK) return enqueued;
    if (!fixed && prev && c->h_counts && c->tail_start < 0 && !c->counters_on && last_total <= TAIL_MAX_BATCH && cam->max_depth >= 1) {
        // the last batch's queue lengths, for the schedule of the next call (nobody waits for them)
        auto& seen = c->seen;
        const bool same = seen.scene == sc && seen.width == cam->width && seen.height == cam->height && seen.max_depth == cam->max_depth && seen.paths == last_total;
        if (!same) seen.chosen = -1;
        seen.scene = sc; seen.width = cam->width; seen.height = cam->height; seen.max_depth = cam->max_depth; seen.paths = last_total;
        seen.n = std::min<uint32_t>(cam->max_depth + 1, COUNTS_SEEN_MAX);
        seen.ran_to = std::min<uint32_t>(last_tail_start, seen.n - 1);
        HIP_TRY(hipMemcpyAsync(c->h_counts, prev->counts, seen.n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipEventRecord(c->ev_counts, c->stream));
    }
    HIP_TRY(hipEventRecord(c->ev_end, c->stream));
    HIP_TRY(hipGetLastError());
    c->have_render_time = true;
    c->phase_pending = c->phase_pending || timing;
    if (c->counters_on) {
        // paths started: one per (pixel, sample)
        unsigned long long add = npix * n_samples;
        DCounters h;
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipMemcpy(&h, c->d_counters, sizeof(h), hipMemcpyDeviceToHost));
        h.samples += add;
        HIP_TRY(hipMemcpy(c->d_counters, &h, sizeof(h), hipMemcpyHostToDevice));
    }
    return RT_OK;
}

int rt_render_accumulate_rows_device(rt_scene* sc, const rt_camera_desc* cam, uint64_t seed, uint32_t first_sample, uint32_t n_samples,
                                     uint32_t first_row, uint32_t n_rows, float* d_acc) {
    return render_rows(sc, cam, seed, first_sample, n_samples, first_row, n_rows, d_acc, nullptr);
}

int rt_render_accumulate_device_noalloc(rt_scene* sc, const rt_camera_desc* cam, uint64_t seed, uint32_t first_sample, uint32_t n_samples,
                                        float* d_acc) {
    if (!sc || !cam || !d_acc) return fail(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    rt_ctx* c = sc->ctx;
    if (cam->max_depth > NOALLOC_MAX_DEPTH)
        return fail(RT_ERR_UNSUPPORTED, "max_depth %u: this entry point's preallocated queue counters hold at most %u bounces", cam->max_depth,
                    NOALLOC_MAX_DEPTH);
    unsigned char* base = c->device < MAX_DEVICES ? scratch_base[c->device] : nullptr;
    if (!base) return fail(RT_ERR_UNSUPPORTED, "no preallocated scratch memory on device %d", c->device);
    if (rt::extend_overflow_bytes(NOALLOC_OVERFLOW_BLOCKS) > NOALLOC_OVERFLOW_BYTES) return fail(RT_ERR_HIP, "internal error: overflow area too small");
    // one pipeline on the caller's stream, its queues laid out in the scratch array of this GPU
    Pipe q;
    unsigned char* at = base;
    auto take = [&](size_t bytes) { unsigned char* p = at; at += bytes; return p; };
    for (int i = 0; i < 2; ++i) {
        q.ray_o[i] = (float4*)take(NOALLOC_RAY_BYTES);
        q.ray_d[i] = (float4*)take(NOALLOC_RAY_BYTES);
        q.thr[i] = (float*)take(NOALLOC_THR_BYTES);
    }
    q.radiance = (float4*)take(NOALLOC_RAY_BYTES);
    q.hits = (uint2*)take(NOALLOC_HITS_BYTES);
    q.tex_list = (uint32_t*)take(NOALLOC_TEX_BYTES);
    q.counts = (uint32_t*)take(NOALLOC_COUNTS_BYTES);
    q.counts_cap = 2 * NOALLOC_MAX_DEPTH + 2;
    q.stack_overflow = (unsigned long long*)take(NOALLOC_OVERFLOW_BYTES);
    q.cap = NOALLOC_PATHS;
    if ((size_t)(at - base) != NOALLOC_SCRATCH_BYTES) return fail(RT_ERR_HIP, "internal error: scratch layout");
    return render_rows(sc, cam, seed, first_sample, n_samples, 0, cam->height, d_acc, &q);
}

int rt_ctx_set_host_buffer_pinning(rt_ctx* c, int on) {
    if (!c) return fail(RT_ERR_INVALID_ARGUMENT, "ctx is NULL");
    int e = set_device(c);
    if (e) return e;
    c->pin_host_buffers = on != 0;
    if (!c->pin_host_buffers) release_host_buffer(c);
    return RT_OK;
}

int rt_ctx_memory_info(rt_ctx* c, size_t* free_bytes, size_t* total_bytes) {
    if (!c || !free_bytes || !total_bytes) return fail(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    int e = set_device(c);
    if (e) return e;
    HIP_TRY(hipMemGetInfo(free_bytes, total_bytes));
    return RT_OK;
}

int rt_ctx_release_host_buffer(rt_ctx* c) {
    if (!c) return fail(RT_ERR_INVALID_ARGUMENT, "ctx is NULL");
    int e = set_device(c);
    if (e) return e;
    release_host_buffer(c);
    return RT_OK;
}

int rt_render_accumulate(rt_scene* sc, const rt_camera_desc* cam, uint64_t seed, uint32_t first_sample, uint32_t n_samples,
                         float* acc_host) {
    if (!sc || !acc_host) return fail(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    int e = check_camera(cam);
    if (e) return e;
    rt_ctx* c = sc->ctx;
    if ((e = set_device(c))) return e;
    const size_t bytes = (size_t)cam->width * cam->height * 4 * sizeof(float);
    // the device copy of the buffer is kept between calls (its contents are not: the caller owns the buffer and
    // may have changed it, so every call uploads it again)
    if (bytes > c->d_acc_cache_bytes) {
        if (c->d_acc_cache) (void)hipFree(c->d_acc_cache);
        c->d_acc_cache = nullptr;
        c->d_acc_cache_bytes = 0;
        HIP_TRY(hipMalloc((void**)&c->d_acc_cache, bytes));
        c->d_acc_cache_bytes = bytes;
    }
    if (!c->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    if (!c->ev_upload) HIP_TRY(hipEventCreateWithFlags(&c->ev_upload, hipEventDisableTiming));
    // When the caller has asked for it, the buffer is page-locked once, for as long as the same buffer keeps coming (copies from and to pageable
    // memory run at a fraction of the link's rate, and only copies from page-locked memory run beside kernels).  Not
    // every system lets the runtime lock memory it did not allocate; rt_host_buffer_create hands out memory that is
    // page-locked from the start, and pageable memory still works, only slower.
    if (!c->pin_host_buffers) {
        release_host_buffer(c);
    } else if (c->pinned_ptr != (void*)acc_host || c->pinned_bytes != bytes) {
        release_host_buffer(c);
        hipPointerAttribute_t attr;
        const bool locked_already = hipPointerGetAttributes(&attr, acc_host) == hipSuccess && attr.type == hipMemoryTypeHost;
        (void)hipGetLastError();
        if (!locked_already && hipHostRegister(acc_host, bytes, hipHostRegisterDefault) == hipSuccess) {
            c->pinned_ptr = acc_host;
            c->pinned_bytes = bytes;
        }
        (void)hipGetLastError();
    }
    // the upload runs beside ray generation, extend and shade; only the accumulation kernels wait for it
    float* d_acc = c->d_acc_cache;
    hipError_t he = hipMemcpyAsync(d_acc, acc_host, bytes, hipMemcpyHostToDevice, c->copy_stream);
    if (he == hipSuccess) he = hipEventRecord(c->ev_upload, c->copy_stream);
    if (he == hipSuccess) {
        e = render_rows(sc, cam, seed, first_sample, n_samples, 0, cam->height, d_acc, nullptr, c->ev_upload);
        if (e == RT_OK) he = hipMemcpyAsync(acc_host, d_acc, bytes, hipMemcpyDeviceToHost, c->stream);
    }
    hipError_t hc = hipStreamSynchronize(c->copy_stream);
    hipError_t hs = hipStreamSynchronize(c->stream);
    if (e) return e;
    if (he != hipSuccess) return fail(RT_ERR_HIP, "accumulation buffer copy failed: %s", hipGetErrorString(he));
    if (hs != hipSuccess || hc != hipSuccess) return fail(RT_ERR_HIP, "render failed: %s", hipGetErrorString(hs != hipSuccess ? hs : hc));
    return collect_phases(c);
}

int rt_render_accumulate_multi(rt_scene* const* scenes, int n_gpus, int split, const rt_camera_desc* cam, uint64_t seed, uint32_t first_sample,
                               uint32_t n_samples, float* d_acc) {
    if (!scenes || !d_acc) return fail(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n_gpus < 1 || n_gpus > 64) return fail(RT_ERR_INVALID_ARGUMENT, "%d GPUs: 1 to 64 are supported", n_gpus);
    if (split != RT_SPLIT_SAMPLES && split != RT_SPLIT_ROWS) return fail(RT_ERR_INVALID_ARGUMENT, "unknown split %d", split);
    int e = check_camera(cam);
    if (e) return e;
    for (int i = 0; i < n_gpus; ++i) {
        if (!scenes[i]) return fail(RT_ERR_INVALID_ARGUMENT, "scenes[%d] is NULL", i);
        for (int j = 0; j < i; ++j)
            if (scenes[j]->ctx == scenes[i]->ctx) return fail(RT_ERR_INVALID_ARGUMENT, "scenes[%d] and scenes[%d] share a context: one context per share of the work", j, i);
    }
    if ((uint64_t)first_sample + n_samples > 0xFFFFFFFFull) return fail(RT_ERR_INVALID_ARGUMENT, "sample numbers overflow 32 bits");
    rt_ctx* c0 = scenes[0]->ctx;
    if (n_gpus == 1) return render_rows(scenes[0], cam, seed, first_sample, n_samples, 0, cam->height, d_acc, nullptr);
    const size_t row_bytes = (size_t)cam->width * 4 * sizeof(float);
    const size_t bytes = row_bytes * cam->height;
    const unsigned long long n_pixels = (unsigned long long)cam->width * cam->height;
    // share i of `total` units: the first (total % n) shares get one unit more, shares are contiguous and in GPU order
    auto share = [&](uint32_t total, int i, uint32_t& first, uint32_t& count) {
        const uint32_t base = total / (uint32_t)n_gpus, rem = total % (uint32_t)n_gpus;
        count = base + ((uint32_t)i < rem ? 1u : 0u);
        first = (uint32_t)i * base + std::min<uint32_t>((uint32_t)i, rem);
    };
    auto event_of = [](hipEvent_t& ev) -> hipError_t { return ev ? hipSuccess : hipEventCreateWithFlags(&ev, hipEventDisableTiming); };

    // GPU 0 holds the image: whatever is already enqueued for d_acc on its stream comes first
    if ((e = set_device(c0))) return e;
    HIP_TRY(event_of(c0->ev_multi_ready));
    HIP_TRY(hipEventRecord(c0->ev_multi_ready, c0->stream));
    if (split == RT_SPLIT_SAMPLES && (c0->peer_in.size() < (size_t)n_gpus || c0->peer_in_bytes < bytes)) {
        HIP_TRY(hipStreamSynchronize(c0->stream));
        for (float* p : c0->peer_in) if (p) (void)hipFree(p);
        c0->peer_in.assign((size_t)n_gpus, nullptr);
        c0->peer_in_bytes = 0;
        for (int i = 1; i < n_gpus; ++i) HIP_TRY(hipMalloc((void**)&c0->peer_in[(size_t)i], bytes));
        c0->peer_in_bytes = bytes;
    }
    // every other GPU: its share into its private buffer, then the buffer (or its rows) over to GPU 0
    for (int i = 1; i < n_gpus; ++i) {
        rt_ctx* c = scenes[i]->ctx;
        if ((e = set_device(c))) return e;
        if (c->device != c0->device && c0->device < 64 && !(c->peer_asked >> c0->device & 1u)) {
            // copies between the two GPUs go over their direct link when the runtime grants access (through host memory
            // otherwise: slower, same result); asked once per pair
            (void)hipDeviceEnablePeerAccess(c0->device, 0);
            (void)hipGetLastError();
            c->peer_asked |= 1ull << c0->device;
        }
        if (c->d_partial_bytes < bytes) {
            HIP_TRY(hipStreamSynchronize(c->stream));
            if (c->d_partial) (void)hipFree(c->d_partial);
            c->d_partial = nullptr;
            c->d_partial_bytes = 0;
            HIP_TRY(hipMalloc((void**)&c->d_partial, bytes));
            c->d_partial_bytes = bytes;
        }
        HIP_TRY(event_of(c->ev_multi_done));
        uint32_t first = 0, count = 0;
        if (split == RT_SPLIT_SAMPLES) {
            share(n_samples, i, first, count);
            HIP_TRY(hipMemsetAsync(c->d_partial, 0, bytes, c->stream));
            if (count && (e = render_rows(scenes[i], cam, seed, first_sample + first, count, 0, cam->height, c->d_partial, nullptr))) return e;
            if ((e = set_device(c))) return e;
            // peer_in[i] is kept from call to call and the call returns without waiting: the additions of the previous
            // call (on GPU 0's stream, which may lag behind this GPU's) must have read it before the new share lands in it.
            // ev_multi_ready was recorded on that stream after them.
            HIP_TRY(hipStreamWaitEvent(c->stream, c0->ev_multi_ready, 0));
            HIP_TRY(hipMemcpyPeerAsync(c0->peer_in[(size_t)i], c0->device, c->d_partial, c->device, bytes, c->stream));
        } else {
            // its rows of the image as they are now, the new samples on top, and back: exactly what one GPU would have computed
            share(cam->height, i, first, count);
            if (count) {
                float* mine = c->d_partial + (size_t)first * cam->width * 4;
                float* theirs = d_acc + (size_t)first * cam->width * 4;
                HIP_TRY(hipStreamWaitEvent(c->stream, c0->ev_multi_ready, 0));
                HIP_TRY(hipMemcpyPeerAsync(mine, c->device, theirs, c0->device, row_bytes * count, c->stream));
                if ((e = render_rows(scenes[i], cam, seed, first_sample, n_samples, first, count, c->d_partial, nullptr))) return e;
                if ((e = set_device(c))) return e;
                HIP_TRY(hipMemcpyPeerAsync(theirs, c0->device, mine, c->device, row_bytes * count, c->stream));
            }
        }
        HIP_TRY(hipEventRecord(c->ev_multi_done, c->stream));
    }
    // GPU 0: its own share straight into the image, then the others' shares in GPU order
    if ((e = set_device(c0))) return e;
    uint32_t first = 0, count = 0;
    if (split == RT_SPLIT_SAMPLES) {
        share(n_samples, 0, first, count);
        if (count && (e = render_rows(scenes[0], cam, seed, first_sample + first, count, 0, cam->height, d_acc, nullptr))) return e;
    } else {
        share(cam->height, 0, first, count);
        if (count && (e = render_rows(scenes[0], cam, seed, first_sample, n_samples, first, count, d_acc, nullptr))) return e;
    }
    if ((e = set_device(c0))) return e;
    for (int i = 1; i < n_gpus; ++i) {
        HIP_TRY(hipStreamWaitEvent(c0->stream, scenes[i]->ctx->ev_multi_done, 0));
        if (split == RT_SPLIT_SAMPLES) launch_add_buffers(c0->stream, c0->max_blocks, (float4*)d_acc, (const float4*)c0->peer_in[(size_t)i], n_pixels);
    }
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

int rt_finalize_device(rt_ctx* c, const float* d_acc, uint64_t n_pixels, uint32_t total_samples, int mode, float exposure, void* d_out) {
    if (!c || !d_acc || !d_out) return fail(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    if (mode < RT_FINALIZE_MEAN || mode > RT_FINALIZE_TONEMAP_RGBA) return fail(RT_ERR_INVALID_ARGUMENT, "unknown finalize mode %d", mode);
    if (total_samples == 0) return fail(RT_ERR_INVALID_ARGUMENT, "total_samples is 0");
    int e = set_device(c);
    if (e) return e;
    float inv = 1.0f / (float)total_samples;  // src/camera.rs:279
    launch_finalize(c->stream, c->max_blocks, (const float4*)d_acc, n_pixels, inv, mode, exposure, d_out);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

int rt_finalize(rt_ctx* c, const float* acc_host, uint64_t n_pixels, uint32_t total_samples, int mode, float exposure, void* out_host) {
    if (!c || !acc_host || !out_host) return fail(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    int e = set_device(c);
    if (e) return e;
    size_t in_bytes = n_pixels * 4 * sizeof(float);
    size_t out_bytes = mode == RT_FINALIZE_TONEMAP_RGBA ? n_pixels * sizeof(uint32_t) : in_bytes;
    float* d_acc = nullptr;
    void* d_out = nullptr;
    HIP_TRY(hipMalloc((void**)&d_acc, std::max<size_t>(in_bytes, 16)));
    hipError_t he = hipMalloc(&d_out, std::max<size_t>(out_bytes, 16));
    if (he != hipSuccess) { (void)hipFree(d_acc); return fail(RT_ERR_OUT_OF_MEMORY, "hipMalloc failed: %s", hipGetErrorString(he)); }
    he = hipMemcpyAsync(d_acc, acc_host, in_bytes, hipMemcpyHostToDevice, c->stream);
    if (he == hipSuccess) {
        e = rt_finalize_device(c, d_acc, n_pixels, total_samples, mode, exposure, d_out);
        if (e == RT_OK) he = hipMemcpyAsync(out_host, d_out, out_bytes, hipMemcpyDeviceToHost, c->stream);
    }
    hipError_t hs = hipStreamSynchronize(c->stream);
    (void)hipFree(d_acc);
    (void)hipFree(d_out);
    if (e) return e;
    if (he != hipSuccess || hs != hipSuccess) return fail(RT_ERR_HIP, "finalize failed: %s", hipGetErrorString(he != hipSuccess ? he : hs));
    return RT_OK;
}

int rt_intersect_batch(rt_scene* sc, const rt_ray* rays, size_t n, uint64_t seed, rt_hit* out) {
    if (!sc || (n && (!rays || !out))) return fail(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n == 0) return RT_OK;
    rt_ctx* c = sc->ctx;
    int e = set_device(c);
    if (e) return e;
    rt_ray* d_rays = nullptr;
    rt_hit* d_hits = nullptr;
    HIP_TRY(hipMalloc((void**)&d_rays, n * sizeof(rt_ray)));
    hipError_t he = hipMalloc((void**)&d_hits, n * sizeof(rt_hit));
    if (he != hipSuccess) { (void)hipFree(d_rays); return fail(RT_ERR_OUT_OF_MEMORY, "hipMalloc failed: %s", hipGetErrorString(he)); }
    he = hipMemcpyAsync(d_rays, rays, n * sizeof(rt_ray), hipMemcpyHostToDevice, c->stream);
    if (he == hipSuccess) {
        launch_intersect(c->stream, c->max_blocks, sc->d, seed, d_rays, n, d_hits);
        he = hipGetLastError();
    }
    if (he == hipSuccess) he = hipMemcpyAsync(out, d_hits, n * sizeof(rt_hit), hipMemcpyDeviceToHost, c->stream);
    hipError_t hs = hipStreamSynchronize(c->stream);
    (void)hipFree(d_rays);
    (void)hipFree(d_hits);
    if (he != hipSuccess || hs != hipSuccess) return fail(RT_ERR_HIP, "intersect failed: %s", hipGetErrorString(he != hipSuccess ? he : hs));
    return RT_OK;
}

int rt_extend_batch(rt_scene* sc, const rt_ray* rays, size_t n, uint64_t seed, rt_hit* out) {
    if (!sc || (n && (!rays || !out))) return fail(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n == 0) return RT_OK;
    if (n >= (1ull << 31)) return fail(RT_ERR_UNSUPPORTED, "more than 2^31 rays");
    rt_ctx* c = sc->ctx;
    int e = set_device(c);
    if (e) return e;
    // the render path's interval, and path ids that hold (rng_pixel, rng_sample) in 32 bits
    uint32_t max_pixel = 0, max_sample = 0, max_bounce = 0;
    for (size_t i = 0; i < n; ++i) {
        if (rays[i].tmin != 0.001f || rays[i].tmax != INFINITY)
            return fail(RT_ERR_INVALID_ARGUMENT, "rays[%zu]: the render path casts every ray over (0.001, +inf)", i);
        max_pixel = std::max(max_pixel, rays[i].rng_pixel);
        max_sample = std::max(max_sample, rays[i].rng_sample);
        max_bounce = std::max(max_bounce, rays[i].rng_bounce);
    }
    uint32_t pixel_bits = 0;
    while (pixel_bits < 32 && (1ull << pixel_bits) <= max_pixel) ++pixel_bits;
    if (pixel_bits >= 32 || ((uint64_t)max_sample << pixel_bits) > 0xFFFFFFFFull)
        return fail(RT_ERR_UNSUPPORTED, "rng_pixel (max %u) and rng_sample (max %u) do not fit one 32-bit path id", max_pixel, max_sample);
    if (max_bounce > 4096) return fail(RT_ERR_UNSUPPORTED, "rng_bounce above 4096");
    // one launch per bounce number: rays grouped by rng_bounce (stable), results scattered back afterwards
    std::vector<uint32_t> group_size(max_bounce + 1, 0), group_at(max_bounce + 2, 0);
    for (size_t i = 0; i < n; ++i) group_size[rays[i].rng_bounce]++;
    for (uint32_t b = 0; b <= max_bounce; ++b) group_at[b + 1] = group_at[b] + group_size[b];
    std::vector<uint32_t> slot_of(n), fill(group_at.begin(), group_at.end() - 1);
    std::vector<rt_ray> grouped(n);
    for (size_t i = 0; i < n; ++i) {
        slot_of[i] = fill[rays[i].rng_bounce]++;
        grouped[slot_of[i]] = rays[i];
    }
    Pipe& q = c->pipes[0];
    if ((e = ensure_queues(c, q, n, max_bounce + 1, needs_overflow(sc)))) return e;
    rt_ray* d_rays = nullptr;
    rt_hit* d_hits = nullptr;
    std::vector<rt_hit> grouped_out(n);
    HIP_TRY(hipMalloc((void**)&d_rays, n * sizeof(rt_ray)));
    hipError_t he = hipMalloc((void**)&d_hits, n * sizeof(rt_hit));
    if (he != hipSuccess) { (void)hipFree(d_rays); return fail(RT_ERR_OUT_OF_MEMORY, "hipMalloc failed: %s", hipGetErrorString(he)); }
    hipStream_t st = c->stream;
    he = hipMemcpyAsync(d_rays, grouped.data(), n * sizeof(rt_ray), hipMemcpyHostToDevice, st);
    if (he == hipSuccess) he = hipMemcpyAsync(q.counts, group_size.data(), (max_bounce + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, st);
    if (he == hipSuccess) {
        DCounters* counters = c->counters_on ? c->d_counters : nullptr;
        rt::ExtendTuning tune = c->tune;
        if (tune.medium_min <= 0) tune.medium_min = sc->media_are_single_spheres ? 4 : 24;
        const BatchInfo b{0u, pixel_bits == 0 ? 1u : (pixel_bits >= 31 ? 0x7FFFFFFFu : (1u << pixel_bits)), 0u, max_sample + 1u, pixel_bits};
        launch_pack_rays(st, c->max_blocks, d_rays, n, pixel_bits, q.ray_o[0], q.ray_d[0]);
        for (uint32_t bounce = 0; bounce <= max_bounce; ++bounce) {
            if (group_size[bounce] == 0) continue;
            const size_t at = group_at[bounce];
            // the kernels read their queue length from counts[bounce]: the queue of this group starts at `at`
            if (sc->d.n_global != 0)
                launch_first_leaves(st, c->max_blocks, group_size[bounce], sc->d, b, seed, bounce, q.counts, q.ray_o[0] + at, q.ray_d[0] + at,
                                    q.hits + at, counters);
            launch_extend(st, extend_grid_of(c, sc, counters != nullptr), group_size[bounce], sc->d, b, seed, bounce, q.counts,
                          q.ray_o[0] + at, q.ray_d[0] + at, q.hits + at, needs_overflow(sc) ? q.stack_overflow : nullptr, tune, counters);
        }
        launch_unpack_hits(st, c->max_blocks, sc->d, d_rays, q.hits, n, d_hits);
        he = hipGetLastError();
    }
    if (he == hipSuccess) he = hipMemcpyAsync(grouped_out.data(), d_hits, n * sizeof(rt_hit), hipMemcpyDeviceToHost, st);
    hipError_t hs = hipStreamSynchronize(st);
    (void)hipFree(d_rays);
    (void)hipFree(d_hits);
    if (he != hipSuccess || hs != hipSuccess) return fail(RT_ERR_HIP, "extend failed: %s", hipGetErrorString(he != hipSuccess ? he : hs));
    for (size_t i = 0; i < n; ++i) out[i] = grouped_out[slot_of[i]];
    return RT_OK;
}

int rt_shade_batch(rt_scene* sc, const rt_ray* rays, const rt_hit* hits, size_t n, uint64_t seed, rt_ray* out_scattered,
                   float* out_attenuation3, float* out_emitted3) {
    if (!sc || (n && (!rays || !hits || !out_scattered || !out_attenuation3 || !out_emitted3))) return fail(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n == 0) return RT_OK;
    for (size_t i = 0; i < n; ++i)
        if (hits[i].material >= sc->d.n_materials) return fail(RT_ERR_INVALID_ARGUMENT, "hits[%zu].material %u out of range", i, hits[i].material);
    rt_ctx* c = sc->ctx;
    int e = set_device(c);
    if (e) return e;
    rt_ray *d_rays = nullptr, *d_out = nullptr;
    rt_hit* d_hits = nullptr;
    float *d_att = nullptr, *d_emit = nullptr;
    hipError_t he = hipMalloc((void**)&d_rays, n * sizeof(rt_ray));
    if (he == hipSuccess) he = hipMalloc((void**)&d_out, n * sizeof(rt_ray));
    if (he == hipSuccess) he = hipMalloc((void**)&d_hits, n * sizeof(rt_hit));
    if (he == hipSuccess) he = hipMalloc((void**)&d_att, n * 3 * sizeof(float));
    if (he == hipSuccess) he = hipMalloc((void**)&d_emit, n * 3 * sizeof(float));
    if (he == hipSuccess) he = hipMemcpyAsync(d_rays, rays, n * sizeof(rt_ray), hipMemcpyHostToDevice, c->stream);
    if (he == hipSuccess) he = hipMemcpyAsync(d_hits, hits, n * sizeof(rt_hit), hipMemcpyHostToDevice, c->stream);
    if (he == hipSuccess) {
        launch_shade_records(c->stream, c->max_blocks, sc->d, seed, d_rays, d_hits, n, d_out, d_att, d_emit);
        he = hipGetLastError();
    }
    if (he == hipSuccess) he = hipMemcpyAsync(out_scattered, d_out, n * sizeof(rt_ray), hipMemcpyDeviceToHost, c->stream);
    if (he == hipSuccess) he = hipMemcpyAsync(out_attenuation3, d_att, n * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (he == hipSuccess) he = hipMemcpyAsync(out_emitted3, d_emit, n * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    hipError_t hs = hipStreamSynchronize(c->stream);
    (void)hipFree(d_rays); (void)hipFree(d_out); (void)hipFree(d_hits); (void)hipFree(d_att); (void)hipFree(d_emit);
    if (he != hipSuccess || hs != hipSuccess) return fail(RT_ERR_HIP, "shade failed: %s", hipGetErrorString(he != hipSuccess ? he : hs));
    return RT_OK;
}

int rt_primary_rays(rt_ctx* c, const rt_camera_desc* cam, uint64_t seed, uint32_t sample, const uint32_t* pixels, size_t n, rt_ray* out) {
    if (!c || (n && (!pixels || !out))) return fail(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    int e = check_camera(cam);
    if (e) return e;
    if (n == 0) return RT_OK;
    const uint64_t npix = (uint64_t)cam->width * cam->height;
    for (size_t i = 0; i < n; ++i)
        if (pixels[i] >= npix) return fail(RT_ERR_INVALID_ARGUMENT, "pixels[%zu] = %u outside the %ux%u image", i, pixels[i], cam->width, cam->height);
    if ((e = set_device(c))) return e;
    uint32_t* d_pix = nullptr;
    rt_ray* d_out = nullptr;
    hipError_t he = hipMalloc((void**)&d_pix, n * sizeof(uint32_t));
    if (he == hipSuccess) he = hipMalloc((void**)&d_out, n * sizeof(rt_ray));
    if (he == hipSuccess) he = hipMemcpyAsync(d_pix, pixels, n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream);
    if (he == hipSuccess) {
        launch_primary_rays(c->stream, c->max_blocks, to_dcamera(cam), seed, sample, d_pix, n, d_out);
        he = hipGetLastError();
    }
    if (he == hipSuccess) he = hipMemcpyAsync(out, d_out, n * sizeof(rt_ray), hipMemcpyDeviceToHost, c->stream);
    hipError_t hs = hipStreamSynchronize(c->stream);
    (void)hipFree(d_pix); (void)hipFree(d_out);
    if (he != hipSuccess || hs != hipSuccess) return fail(RT_ERR_HIP, "primary rays failed: %s", hipGetErrorString(he != hipSuccess ? he : hs));
    return RT_OK;
}

}  // extern "C"
