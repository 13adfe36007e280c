// rt_kernels.h — host-callable launchers of the kernels in rt_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rt_abi.h"
#include "rt_device.h"

namespace rt {

// One batch of paths: pixels [first_pixel, first_pixel + n_pixels) of the image,
// sample numbers [first_sample, first_sample + n_samples).
// A path of the batch is known by its id (s_local << pixel_bits) | pixel_local, carried in the w lane of its ray's
// direction: sample and pixel come out of it with a shift and a mask, its slot in the radiance array is
// s_local * n_pixels + pixel_local.
struct BatchInfo {
    uint32_t first_pixel, n_pixels, first_sample, n_samples;
    uint32_t pixel_bits;  // smallest number of bits that holds n_pixels - 1
};
inline BatchInfo make_batch(uint32_t first_pixel, uint32_t n_pixels, uint32_t first_sample, uint32_t n_samples) {
    uint32_t bits = 0;
    while (bits < 32 && (1ull << bits) < n_pixels) ++bits;
    return BatchInfo{first_pixel, n_pixels, first_sample, n_samples, bits};
}

// max_blocks caps the grid (grid-stride kernels); upper_bound is a host-side
// bound on the queue length, the exact length is read from counts[bounce].
// first_hits (may be NULL): where ray generation leaves, for every ray, its closest hit among the leaves
// kept out of the tree (scenes with sc.n_global != 0); the extend kernel starts from that record.  With
// NULL the caller runs launch_first_leaves before the first launch_extend, as it does before all later ones.
void launch_raygen(hipStream_t st, uint32_t max_blocks, const DScene& sc, const DCamera& cam, const BatchInfo& b, uint64_t seed, float4* ray_o,
                   float4* ray_d, float* thr, uint2* first_hits, uint32_t* counts, uint32_t n_counts);
void launch_primary_rays(hipStream_t st, uint32_t max_blocks, const DCamera& cam, uint64_t seed, uint32_t sample, const uint32_t* pixels,
                         unsigned long long n, rt_ray* out);
// Lane-count thresholds of the extend kernel's loop (see k_extend).
struct ExtendTuning {
    int inner_min = 20;   // keep stepping through inner nodes while this many lanes are at one
    int refill_min = 16;  // fetch new rays once this many lanes are free
    int medium_min = 0;   // run the medium test once this many lanes wait at a medium; 0: chosen per scene (4 when every
                          // medium in the tree is bounded by one sphere - a short test -, 24 when some boundary is a list
                          // of leaves, whose test runs to several hundred instructions)
    int quad_min = 6;     // run the quad test once this many lanes wait at a quad
};
int extend_blocks_per_cu(bool counting, bool generic_media);  // generic_media: the variant for scenes with DScene::generic_media set
// stack levels per lane the extend kernel keeps in LDS for a tree of n_nodes nodes; deeper levels go to the overflow area
int extend_lds_stack_levels(uint32_t n_nodes);
size_t extend_overflow_bytes(uint32_t blocks);
// The same computation as a pass of its own over a queue.
void launch_first_leaves(hipStream_t st, uint32_t max_blocks, uint32_t upper_bound, const DScene& sc, const BatchInfo& b, uint64_t seed,
                         uint32_t bounce, const uint32_t* counts, const float4* ray_o, const float4* ray_d, uint2* hits, DCounters* counters);
// resident_blocks: CUs x extend_blocks_per_cu(counters != nullptr, sc.generic_media); stack_overflow: extend_overflow_bytes(resident_blocks) bytes
void launch_extend(hipStream_t st, uint32_t resident_blocks, uint32_t upper_bound, const DScene& sc, const BatchInfo& b, uint64_t seed,
                   uint32_t bounce, const uint32_t* counts, const float4* ray_o, const float4* ray_d, uint2* hits,
                   unsigned long long* stack_overflow, const ExtendTuning& tune, DCounters* counters);
// counts[bounce] in, counts[bounce + 1] out; tex_count / tex_list: this bounce's list of hits on
// textured materials (filled by the first kernel, consumed by the second).
void launch_shade(hipStream_t st, uint32_t max_blocks, uint32_t upper_bound, const DScene& sc, const DCamera& cam, const BatchInfo& b,
                  uint64_t seed, uint32_t bounce, uint32_t* counts, uint32_t* tex_count, uint32_t* tex_list, const float4* ray_o,
                  const float4* ray_d, const float* thr, const uint2* hits, float4* out_o, float4* out_d, float* out_thr, float4* radiance,
                  bool scene_has_textures);
// Every remaining bounce (first_bounce .. max_depth - 1) of the paths in the queue of counts[first_bounce] rays, one thread
// per path.  resident_blocks: at most CUs x tail_blocks_per_cu() and at most the extend kernel's grid (the two kernels
// index the stack overflow area alike).
int tail_blocks_per_cu();
void launch_tail(hipStream_t st, uint32_t resident_blocks, uint32_t upper_bound, const DScene& sc, const DCamera& cam, const BatchInfo& b,
                 uint64_t seed, uint32_t first_bounce, const uint32_t* counts, const float4* ray_o, const float4* ray_d, const float* thr,
                 float4* radiance, unsigned long long* stack_overflow);
void launch_accumulate(hipStream_t st, uint32_t max_blocks, const BatchInfo& b, const float4* radiance, float4* acc);
void launch_add_buffers(hipStream_t st, uint32_t max_blocks, float4* acc, const float4* part, unsigned long long n_pixels);
void launch_finalize(hipStream_t st, uint32_t max_blocks, const float4* acc, unsigned long long n_pixels, float inv_samples, int mode,
                     float exposure, void* out);
void launch_intersect(hipStream_t st, uint32_t max_blocks, const DScene& sc, uint64_t seed, const rt_ray* rays, unsigned long long n,
                      rt_hit* out);
// caller-supplied rays into / hit records out of the queue layout (the extend kernel on its own, for parity checks)
void launch_pack_rays(hipStream_t st, uint32_t max_blocks, const rt_ray* rays, unsigned long long n, uint32_t pixel_bits, float4* ray_o,
                      float4* ray_d);
void launch_unpack_hits(hipStream_t st, uint32_t max_blocks, const DScene& sc, const rt_ray* rays, const uint2* hits, unsigned long long n,
                        rt_hit* out);
void launch_shade_records(hipStream_t st, uint32_t max_blocks, const DScene& sc, uint64_t seed, const rt_ray* rays, const rt_hit* hits,
                          unsigned long long n, rt_ray* out_rays, float* out_att, float* out_emit);

}  // namespace rt
