/*
 * rt_abi.h — C ABI of the MI355X (gfx950) path-tracing backend.
 *
 * This is the drop-in boundary for the per-pixel render loop of
 * rookieCookies/raytracing.  The reference has no FFI of its own; the seam
 * that this ABI replaces is the private method
 *
 *     RaytracingCamera::render(&self, world, n_samples, acc_colours, renderer)
 *                                                   (src/camera.rs:269-306)
 *
 * together with everything it reaches (get_ray, ray_colour, Ray::hit_anything,
 * the leaf hit routines, Material::scatter/emitted, Texture::value,
 * PerlinNoise).  Scene construction stays on the caller's side: the caller
 * flattens its Hittable tree / MaterialMap / textures into the plain arrays
 * below and hands them over once with rt_scene_create.
 *
 * Conventions
 *   - every call returns 0 (RT_OK) or a negative rt_status; rt_last_error()
 *     returns a thread-local, NUL-terminated description of the last failure;
 *   - all structs are plain C data, no padding surprises (explicit pads);
 *   - no exception or panic crosses the boundary;
 *   - a context is bound to one GPU and is not thread-safe (the reference's
 *     callers take `&mut Camera`, src/camera.rs:80-122);
 *   - pointers named d_* are device pointers on the context's GPU, all other
 *     pointers are host pointers.  Everything passed to rt_scene_create is
 *     copied; the caller may free it when the call returns.
 */
#ifndef RT_ABI_H
#define RT_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 1u

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID_ARGUMENT = -1,
    RT_ERR_NO_DEVICE = -2,
    RT_ERR_HIP = -3,
    RT_ERR_OUT_OF_MEMORY = -4,
    RT_ERR_UNSUPPORTED = -5
} rt_status;

#define RT_NO_TRANSFORM 0xFFFFFFFFu

/* ---- geometry ----------------------------------------------------------- */

/* Object-to-world placement of a leaf: the composite of the reference's
 * `Move { RotateY { obj } }` wrappers (src/hittable.rs:47-56,227-243).
 * A world-space point p maps to object space as
 *     p' = R * (p - offset),   R: x' = cos*x - sin*z,  z' = sin*x + cos*z
 * (src/math/ray.rs:207-229).  Any chain of Move/RotateY composes to this. */
typedef struct rt_transform {
    float sin_theta;
    float cos_theta;
    float offset[3];
    float _pad[3];
} rt_transform;

/* Sphere::new(centre, radius, material)          src/hittable.rs:248-259 */
typedef struct rt_sphere {
    float centre[3];
    float radius;
    uint32_t material;
    uint32_t prim_id;    /* caller's tag, reported back in rt_hit.prim_id */
    uint32_t transform;  /* index into transforms or RT_NO_TRANSFORM       */
    uint32_t _pad;
} rt_sphere;

/* MovingSphere::new(centre_1, centre_2, radius, material); the centre at ray
 * time t is centre0 + t*dcentre                  src/hittable.rs:293-310 */
typedef struct rt_moving_sphere {
    float centre0[3];
    float radius;
    float dcentre[3];
    uint32_t material;
    uint32_t prim_id;
    uint32_t transform;
    uint32_t _pad[2];
} rt_moving_sphere;

/* Quad::new(q, u, v, material)                   src/hittable.rs:355-368 */
typedef struct rt_quad {
    float q[3];
    uint32_t material;
    float u[3];
    uint32_t prim_id;
    float v[3];
    uint32_t transform;
} rt_quad;

typedef enum rt_prim_kind {
    RT_PRIM_SPHERE = 0,
    RT_PRIM_MOVING_SPHERE = 1,
    RT_PRIM_QUAD = 2,
    RT_PRIM_MEDIUM = 3
} rt_prim_kind;

/* Reference to one leaf: (kind, index into the array of that kind). */
typedef struct rt_prim_ref {
    uint32_t kind;
    uint32_t index;
} rt_prim_ref;

/* ConstantMedium { boundary, neg_inv_density, phase_function }
 *                                                 src/hittable.rs:412-425
 * The boundary object is the list of leaves
 * scene.boundary[first_boundary .. first_boundary + n_boundary).          */
typedef struct rt_medium {
    uint32_t first_boundary;
    uint32_t n_boundary;
    float neg_inv_density;
    uint32_t phase_material;  /* the Isotropic material pushed by ::new     */
    uint32_t prim_id;
    uint32_t _pad[3];
} rt_medium;

/* ---- appearance ---------------------------------------------------------- */

typedef enum rt_material_kind {     /* MaterialKind, src/material.rs:14-28 */
    RT_MAT_LAMBERTIAN = 0,
    RT_MAT_METAL = 1,               /* param = fuzz_radius                  */
    RT_MAT_DIELECTRIC = 2,          /* param = refraction_index             */
    RT_MAT_DIFFUSE_LIGHT = 3,
    RT_MAT_ISOTROPIC = 4
} rt_material_kind;

typedef struct rt_material {
    uint32_t kind;
    float param;
    uint32_t texture;
    uint32_t _pad;
} rt_material;

typedef enum rt_texture_kind {      /* TextureKind, src/texture.rs:11-36   */
    RT_TEX_SOLID = 0,               /* colour                               */
    RT_TEX_CHECKER = 1,             /* scale = inv_scale; even, odd = ids   */
    RT_TEX_IMAGE = 2,               /* image = id                           */
    RT_TEX_NOISE = 3                /* perlin = id; scale                   */
} rt_texture_kind;

typedef struct rt_texture {
    uint32_t kind;
    float colour[3];
    float scale;
    uint32_t even;
    uint32_t odd;
    uint32_t image;
    uint32_t perlin;
    uint32_t _pad[3];
} rt_texture;

typedef enum rt_image_format {
    RT_IMAGE_RGB8 = 0,    /* 3 bytes per texel; texel value = byte / 255   */
    RT_IMAGE_RGB32F = 1   /* 3 floats per texel (the reference's Rgb32FImage) */
} rt_image_format;

/* Row-major, top row first, like image::Rgb32FImage (src/texture.rs:54-64). */
typedef struct rt_image {
    uint32_t width;
    uint32_t height;
    uint32_t format;
    uint32_t _pad;
    const void* texels;
} rt_image;

/* PerlinNoise tables (src/perlin_noise.rs:8-37). */
typedef struct rt_perlin {
    uint32_t point_count;
    uint32_t _pad;
    const float* rand_vecs;   /* 3 * point_count                            */
    const uint32_t* perm_x;   /* point_count each                           */
    const uint32_t* perm_y;
    const uint32_t* perm_z;
} rt_perlin;

typedef struct rt_scene_desc {
    uint32_t abi_version;     /* RT_ABI_VERSION */
    uint32_t n_spheres;
    uint32_t n_moving_spheres;
    uint32_t n_quads;
    uint32_t n_media;
    uint32_t n_world;         /* leaves of World::entry (src/lib.rs:17-26)  */
    uint32_t n_boundary;      /* leaves that only bound media               */
    uint32_t n_transforms;
    uint32_t n_materials;
    uint32_t n_textures;
    uint32_t n_images;
    uint32_t n_perlins;
    const rt_sphere* spheres;
    const rt_moving_sphere* moving_spheres;
    const rt_quad* quads;
    const rt_medium* media;
    const rt_prim_ref* world;
    const rt_prim_ref* boundary;
    const rt_transform* transforms;
    const rt_material* materials;   /* index = MaterialId (src/material.rs:31-49) */
    const rt_texture* textures;
    const rt_image* images;
    const rt_perlin* perlins;
} rt_scene_desc;

/* The fields of RaytracingCamera (src/camera.rs:206-218) as computed by
 * RaytracingCamera::new (src/camera.rs:222-266). */
typedef struct rt_camera_desc {
    uint32_t width;
    uint32_t height;
    uint32_t max_depth;
    float defocus_angle;
    float centre[3];
    float pixel00_loc[3];
    float pixel_delta_u[3];
    float pixel_delta_v[3];
    float defocus_disk_u[3];
    float defocus_disk_v[3];
    float background_colour[3];
    float _pad;
} rt_camera_desc;

/* ---- per-ray interface (parity checks) ------------------------------------ */

/* Ray { origin, direction, time } (src/math/ray.rs:8-13); the interval is the
 * one Ray::hit_anything starts from, normally (0.001, +inf) (:64-65).       */
typedef struct rt_ray {
    float origin[3];
    float time;
    float direction[3];
    float tmin;
    float tmax;
    uint32_t rng_pixel;    /* key of the medium draw (see rt_render_*)       */
    uint32_t rng_sample;
    uint32_t rng_bounce;
} rt_ray;

/* HitRecord (src/hittable.rs:7-16) plus the leaf tag. hit == 0 means miss and
 * leaves the other fields zero. */
typedef struct rt_hit {
    float t;
    float point[3];
    float normal[3];
    float u;
    float v;
    uint32_t prim_id;
    uint32_t material;
    uint32_t front_face;
    uint32_t hit;
    uint32_t _pad[3];
} rt_hit;

typedef struct rt_counters {
    uint64_t samples;        /* camera paths started                          */
    uint64_t segments;       /* rays cast (calls of Ray::hit_anything)        */
    uint64_t node_visits;    /* BVH nodes popped                              */
    uint64_t sphere_tests;
    uint64_t quad_tests;
    uint64_t medium_tests;
    uint64_t hits;           /* segments that hit something                   */
    uint64_t max_stack;      /* deepest traversal stack of any ray (entries): a
                                maximum, not a sum; beyond the extend kernel's
                                LDS levels (8 when the whole tree is copied to
                                LDS, else 12) the global overflow area was used */
} rt_counters;

typedef enum rt_finalize_mode {
    RT_FINALIZE_MEAN = 0,     /* acc / n            -> 4 floats per pixel     */
    RT_FINALIZE_SQRT = 1,     /* Camera::hdr_render, src/camera.rs:107-122    */
    RT_FINALIZE_TONEMAP_RGBA = 2 /* Camera::realtime_render, :89-104 -> u32  */
} rt_finalize_mode;

typedef struct rt_ctx rt_ctx;
typedef struct rt_scene rt_scene;

const char* rt_last_error(void);
uint32_t rt_abi_version(void);

/* One context per GPU.  device_id is the HIP device ordinal. */
int rt_ctx_create(int device_id, rt_ctx** out);
int rt_ctx_destroy(rt_ctx* ctx);
/* Stream (hipStream_t) on which all work of this context is enqueued;
 * NULL selects the default stream. */
int rt_ctx_set_stream(rt_ctx* ctx, void* hip_stream);
/* Paths kept in flight per batch (queue capacity); 0 restores the default. */
int rt_ctx_set_batch_paths(rt_ctx* ctx, uint64_t max_paths);
/* Batches of one render call that are in flight at a time (1 or 2, default 2;
 * only calls with >= 2 samples and >= 16 Mi paths are split).  With 2 the
 * batches run on two internal streams that start after the work already on the
 * context's stream and that the context's stream waits for; the accumulation
 * buffer takes the batches in sample order either way, so results are
 * bit-identical.  Queue memory is held per pipeline. */
int rt_ctx_set_pipelines(rt_ctx* ctx, int n);
/* How the bounces of a batch are scheduled (no effect on the image, which is
 * bit-identical either way).  Bounces below first_bounce run as one pass of
 * the per-bounce kernels each (first leaves, extend, shade over the compacted
 * queue); from first_bounce on, one launch of the tail kernel takes every
 * surviving path to its end (one thread per path, the loop of ray_colour,
 * src/camera.rs:339-364).  -1 (default): chosen per batch from its size -
 * small batches (interactive frames, low-resolution renders) switch after a
 * few bounces, batches above 2^24 paths never.  0: whole paths in the tail
 * kernel.  Any value >= max_depth: per-bounce kernels only. */
int rt_ctx_set_tail_start(rt_ctx* ctx, int first_bounce);
/* Turn the work counters on or off (off by default: they cost atomics). */
int rt_ctx_set_counters(rt_ctx* ctx, int enabled);
int rt_ctx_get_counters(rt_ctx* ctx, rt_counters* out, int reset);

/* Device memory for callers that have no HIP binding of their own (the
 * accumulation buffer `acc_colours` and the output buffers of a Camera,
 * src/camera.rs:25-27): allocate, zero, copy in, copy out (blocking), free. */
int rt_buffer_create(rt_ctx* ctx, size_t bytes, void** d_out);
int rt_buffer_destroy(rt_ctx* ctx, void* d_ptr);
int rt_buffer_zero(rt_ctx* ctx, void* d_ptr, size_t bytes);
int rt_buffer_upload(rt_ctx* ctx, void* d_dst, const void* src, size_t bytes);
int rt_buffer_download(rt_ctx* ctx, void* dst, const void* d_src, size_t bytes);

/* Page-locked host memory for the buffers that cross the bus on every call of
 * the host-buffer entry points (a Camera's `acc_colours`, src/camera.rs:25):
 * copies from and to it run at the link's full rate and beside kernels, which
 * copies from ordinary (pageable) memory do not.  Ordinary memory is accepted
 * everywhere; this is the faster choice for callers that can place the buffer. */
int rt_host_buffer_create(rt_ctx* ctx, size_t bytes, void** out);
int rt_host_buffer_destroy(rt_ctx* ctx, void* ptr);

/* Replaces World::new + the walk a Rust caller would do over its Hittable
 * tree (src/lib.rs:17-26, src/hittable.rs:35-57): copies the description,
 * bakes transforms, builds the flattened BVH and uploads everything.
 * Limits (RT_ERR_UNSUPPORTED beyond them): fewer than 2^25 spheres, moving
 * spheres, quads and media each; noise tables of at most 2^20 points; a tree
 * whose traversal can need more than 64 stack entries. */
int rt_scene_create(rt_ctx* ctx, const rt_scene_desc* desc, rt_scene** out);
int rt_scene_destroy(rt_scene* scene);

/* What rt_scene_create built: size of the flattened BVH, how many leaves are
 * tested ahead of it, bytes resident on the GPU, host seconds spent building. */
typedef struct rt_scene_info {
    uint32_t bvh_nodes;
    uint32_t bvh_depth;
    uint32_t leaves_in_tree;
    uint32_t leaves_tested_first;
    uint64_t device_bytes;
    double build_seconds;
} rt_scene_info;
int rt_scene_get_info(const rt_scene* scene, rt_scene_info* out);

/* The host-side tree builder on its own (no GPU involved): builds the tree that
 * rt_scene_create would build over `n` leaf boxes (min.xyz, max.xyz per leaf,
 * 6 floats each) and checks it: every leaf referenced exactly once, every box
 * of an inner child covering the boxes beneath it, references in range, nodes
 * numbered parents first.  Replaces nothing in the reference (its
 * Hittable::bvh, src/hittable.rs:103-155, is checked only by rendering); it is
 * here so that the builder can be tested where there is no GPU. */
typedef struct rt_bvh_report {
    uint32_t nodes;
    uint32_t depth;
    uint32_t width;            /* children per node */
    uint32_t leaves_found;     /* distinct leaves reachable from the root */
    uint32_t violations;       /* 0 for a correct tree */
    float sah_cost;            /* sum over inner children and leaves of area / root area */
    uint32_t stack_need;       /* traversal stack entries needed in the worst case (rt_scene_create accepts <= 64) */
    float packed_surface_ratio; /* mean surface of a child's box in the kernels' 64-byte node (8-bit grid coordinates) over
                                 * the surface of the box itself: >= 1, the price of the grid in extra box hits */
} rt_bvh_report;
int rt_bvh_build_check(const float* boxes, uint32_t n, rt_bvh_report* out);

/* Replaces `n_samples` consecutive calls of RaytracingCamera::render
 * (src/camera.rs:269-306) with sample numbers first_sample,
 * first_sample+1, ... (the reference counts from 1, src/camera.rs:82-83):
 * for every pixel, acc[pixel] += sum over those samples of the path radiance.
 * acc is width*height*4 floats, row-major, row 0 on top, lane 3 untouched
 * (the reference's Vec<f32x4> acc_colours, src/camera.rs:25).  Random numbers
 * are a pure function of (seed, pixel, sample, bounce), so images do not
 * depend on the batch size, the scheduling or the split over GPUs:
 * bit-identical across batch sizes, pipelines, row ranges and consecutive
 * calls on one buffer, and equal up to the order of the f32 additions when
 * per-GPU buffers of different sample ranges are summed.
 * Blocking; acc_host is read and written on the host.  The buffer is uploaded
 * (beside the render), accumulated into on the GPU and downloaded on every
 * call; the context keeps the device copy's memory.  Nothing about acc_host
 * outlives the call unless rt_ctx_set_host_buffer_pinning asked for it. */
int rt_render_accumulate(rt_scene* scene, const rt_camera_desc* cam,
                         uint64_t seed, uint32_t first_sample,
                         uint32_t n_samples, float* acc_host);
/* on != 0: rt_render_accumulate registers (page-locks) the caller's buffer
 * with the HIP runtime and keeps it registered for as long as the same
 * pointer and size keep being passed (copies then run at the link's rate and
 * beside kernels).  The caller takes on a lifetime obligation with it: call
 * rt_ctx_release_host_buffer (or destroy the context) BEFORE freeing or
 * reallocating such a buffer - the library recognises the buffer by address
 * and size only.  Default 0: no registration, nothing to release; callers
 * that want the fast copies without the obligation allocate the buffer with
 * rt_host_buffer_create. */
int rt_ctx_set_host_buffer_pinning(rt_ctx* ctx, int on);
int rt_ctx_release_host_buffer(rt_ctx* ctx);
/* Free and total device memory of the context's GPU in bytes, as the HIP
 * runtime reports them now (what the allocating entry points size their
 * batches by; also the way to check that a call has allocated nothing).
 * No counterpart in the reference. */
int rt_ctx_memory_info(rt_ctx* ctx, size_t* free_bytes, size_t* total_bytes);
/* Same with the accumulation buffer resident on the GPU; enqueues on the
 * context's stream and returns without waiting. */
int rt_render_accumulate_device(rt_scene* scene, const rt_camera_desc* cam,
                                uint64_t seed, uint32_t first_sample,
                                uint32_t n_samples, float* d_acc);
/* The same for rows first_row .. first_row + n_rows - 1 of the image only (the
 * other rows of d_acc, which is still the whole width*height*4 buffer, are not
 * touched).  Every pixel is rendered wholly by one call, so GPUs that split an
 * image by rows and sum their buffers get bit for bit the image of one GPU;
 * the reference's own parallel unit is the row as well (src/camera.rs:282-304). */
int rt_render_accumulate_rows_device(rt_scene* scene, const rt_camera_desc* cam,
                                     uint64_t seed, uint32_t first_sample,
                                     uint32_t n_samples, uint32_t first_row,
                                     uint32_t n_rows, float* d_acc);

/* One render call spread over several GPUs of this process: the counterpart of
 * the reference's rows handed to rayon workers (src/camera.rs:282-304), with
 * GPUs in place of threads.  scenes[i] is the same scene created on GPU i's
 * context (rt_ctx_create + rt_scene_create per GPU; scenes[0]'s GPU holds
 * d_acc).  RT_SPLIT_SAMPLES: GPU i renders its contiguous share of the sample
 * numbers for all pixels - GPU 0 straight into d_acc, the others into private
 * zeroed buffers that are copied to GPU 0 (peer copies over xGMI, all at once)
 * and added to d_acc in GPU order, so the result is reproducible, and equal to
 * the one-GPU image up to the order of the f32 additions.  RT_SPLIT_ROWS: GPU
 * i renders all samples for its contiguous share of the rows, on top of those
 * rows of d_acc, and the rows are copied back: bit for bit the one-GPU image
 * (the choice for calls with few samples, such as interactive frames).
 * The same context must not appear twice; two contexts on one GPU are
 * accepted.  Enqueues on the contexts' streams and returns without waiting;
 * everything has finished when scenes[0]'s context has (rt_ctx_synchronize). */
typedef enum rt_split { RT_SPLIT_SAMPLES = 0, RT_SPLIT_ROWS = 1 } rt_split;
int rt_render_accumulate_multi(rt_scene* const* scenes, int n_gpus, int split,
                               const rt_camera_desc* cam, uint64_t seed,
                               uint32_t first_sample, uint32_t n_samples,
                               float* d_acc);

/* rt_render_accumulate_device for callers that cannot allocate.  It has no
 * counterpart in the reference; it is for callers that must not reach
 * hipMalloc / hipFree: code under HIP graph capture, tight serving loops
 * (one sample per frame of an interactive view), harnesses that forbid
 * allocation.  Same arguments, same result bit for bit; the difference is
 * where the ray queues live: in a fixed array that is part of the library's
 * device image (about 57 MiB per GPU the library runs on, held until the
 * process ends) instead of memory the context allocates on demand.  The array
 * holds one batch of 262144 paths (larger jobs are cut into batches of that
 * size: more launches, same image), queue counters for max_depth <= 1024
 * (RT_ERR_UNSUPPORTED beyond it) and a traversal-stack overflow area for 256
 * blocks of the extend kernel (trees deep enough to need that area run on that
 * smaller grid).  The call allocates nothing: no device or host memory, no
 * stream, no event; per-launch phase timing is ignored by it, and the work
 * counters - a blocking read-back - should be off.  All calls on one GPU share
 * that GPU's array: two of them that overlap in time (other contexts, other
 * streams) race; calls on one stream are ordered and safe.  The context must
 * have been created before graph capture starts (rt_ctx_create looks up the
 * array's address).  Enqueues on the context's stream. */
int rt_render_accumulate_device_noalloc(rt_scene* scene,
                                        const rt_camera_desc* cam,
                                        uint64_t seed, uint32_t first_sample,
                                        uint32_t n_samples, float* d_acc);

/* Replaces the per-pixel closures of Camera::hdr_render / realtime_render
 * (src/camera.rs:95-100,115-118) and Colour::to_rgba (src/math/vec3.rs:252-259).
 * d_acc: n_pixels*4 floats.  d_out: n_pixels*4 floats (modes 0, 1) or
 * n_pixels uint32 (mode 2).  Enqueues on the context's stream. */
int rt_finalize_device(rt_ctx* ctx, const float* d_acc, uint64_t n_pixels,
                       uint32_t total_samples, int mode, float exposure,
                       void* d_out);
/* Host-buffer convenience wrapper of rt_finalize_device (blocking). */
int rt_finalize(rt_ctx* ctx, const float* acc_host, uint64_t n_pixels,
                uint32_t total_samples, int mode, float exposure,
                void* out_host);

/* Replaces Ray::hit_anything (src/math/ray.rs:59-275) on a batch of rays:
 * closest hit over World::entry including constant media.  Host buffers,
 * blocking.  seed keys the medium free-path draw as in rt_render_*. */
int rt_intersect_batch(rt_scene* scene, const rt_ray* rays, size_t n,
                       uint64_t seed, rt_hit* out);

/* Ray::hit_anything (src/math/ray.rs:59-275) once more, computed by the kernels
 * of the render path itself: the rays are put into a ray queue and go through
 * the pass over the leaves kept out of the tree and the extend kernel exactly
 * as the rays of rt_render_accumulate* do (persistent waves, traversal stack in
 * LDS with its overflow area in global memory, the tree in LDS when it is small),
 * whereas rt_intersect_batch walks the same tree with one thread per ray and a
 * private stack.  Results must be identical; the call exists so that the
 * render path's traversal can be checked ray by ray.  Every ray must carry the
 * interval the render path uses, tmin = 0.001 and tmax = +inf, and
 * (rng_sample << bits(rng_pixel)) must fit 32 bits, rng_bounce <= 4096
 * (RT_ERR_INVALID_ARGUMENT / RT_ERR_UNSUPPORTED otherwise).  With the work
 * counters on, rt_counters.max_stack tells how deep the stacks got.  Host
 * buffers, blocking. */
int rt_extend_batch(rt_scene* scene, const rt_ray* rays, size_t n,
                    uint64_t seed, rt_hit* out);

/* Replaces Material::emitted + Material::scatter (src/material.rs:65-122, with
 * Texture::value, src/texture.rs:40-73) on caller-supplied (ray, hit record)
 * pairs: for pair i writes the scattered ray (tmin = 1 when scatter returned
 * Some, all zero for None), the attenuation and the emitted colour (3 floats
 * each).  The scatter draws are keyed by the ray's rng_* fields.  Host
 * buffers, blocking. */
int rt_shade_batch(rt_scene* scene, const rt_ray* rays, const rt_hit* hits,
                   size_t n, uint64_t seed, rt_ray* out_scattered,
                   float* out_attenuation3, float* out_emitted3);

/* Replaces get_ray (src/camera.rs:311-320) for the listed pixels of one sample
 * number: the primary rays the render would start from.  Host buffers. */
int rt_primary_rays(rt_ctx* ctx, const rt_camera_desc* cam, uint64_t seed,
                    uint32_t sample, const uint32_t* pixels, size_t n,
                    rt_ray* out);

/* Block until all work enqueued by this context has finished. */
int rt_ctx_synchronize(rt_ctx* ctx);

/* Timing with HIP events on the context's stream.  rt_ctx_last_render_ms: the
 * most recent rt_render_accumulate* call, first to last launch.  With phase
 * timing on, every kernel launch is bracketed by an event pair;
 * rt_ctx_synchronize sums them per phase over all renders enqueued since the
 * previous rt_ctx_synchronize, and rt_ctx_phase_ms reads the sums (milliseconds,
 * number of launches).  phase: 0 raygen, 1 extend, 2 shade, 3 accumulate,
 * 4 the pass over the leaves kept out of the tree (part of extend), 5 the tail
 * kernel. */
int rt_ctx_last_render_ms(rt_ctx* ctx, float* total_ms);
int rt_ctx_set_phase_timing(rt_ctx* ctx, int enabled);
int rt_ctx_phase_ms(rt_ctx* ctx, int phase, float* ms, uint32_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* RT_ABI_H */
