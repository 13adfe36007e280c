// Every scene through the C ABI on the GPU with the host code of librt_amd / librt_host instrumented
// (AddressSanitizer + UBSan on the host side only): scene creation, a few render calls, the output
// mappings, counters, per-ray entry points, checkpoint and image files, destruction.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#include <unistd.h>
#include "../../include/rt_abi.h"
extern "C" {
void* rth_scene_builtin(int id, uint32_t width, uint32_t height, uint32_t max_depth, const uint8_t* earth_rgb8, uint32_t earth_w, uint32_t earth_h, uint64_t stress_count, int device);
void rth_scene_free(void* p);
int rth_render_samples(void* p, uint32_t n);
int rth_hdr_render(void* p, float* out_rgb);
int rth_realtime_render(void* p, uint32_t* out);
int rth_realtime_render_view(void* p, const uint32_t** words, uint64_t* n_words);
int rth_download_accumulator(void* p, float* out);
int rth_render_to_image(void* p, uint64_t sample_count, const char* path, float* out_rgba);
int rth_save_checkpoint(void* p, const char* path);
int rth_load_checkpoint(void* p, const char* path);
int rth_camera_desc(void* p, rt_camera_desc* out);
rt_ctx* rth_ctx(void* p);
rt_scene* rth_scene(void* p);
const char* rth_last_error();
}
#define CHECK(x) do { int rc_ = (x); if (rc_) { printf("FAILED %s: %d %s / %s\n", #x, rc_, rth_last_error(), rt_last_error()); return 1; } } while (0)
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);  // what was printed survives an abort
    const uint32_t W = 96, H = 64;
    std::vector<uint8_t> earth(1024 * 512 * 3, 128);
    for (int id : {0, 1, 2, 3, 4, 5, 6, 7, 8, 100, 101, 102}) {
        void* h = rth_scene_builtin(id, W, H, 12, earth.data(), 1024, 512, 20000, 0);
        if (!h) { printf("scene %d: %s\n", id, rth_last_error()); return 1; }
        std::vector<float> rgb(W * H * 3), acc(W * H * 4), rgba(W * H * 4);
        std::vector<uint32_t> packed(W * H);
        CHECK(rth_render_samples(h, 3));
        CHECK(rth_hdr_render(h, rgb.data()));
        CHECK(rth_realtime_render(h, packed.data()));
        const uint32_t* frame = nullptr;
        uint64_t frame_words = 0;
        CHECK(rth_realtime_render_view(h, &frame, &frame_words));
        if (frame_words != (uint64_t)W * H || (frame[0] >> 24) != 0) { printf("scene %d: realtime view %llu words\n", id, (unsigned long long)frame_words); return 1; }
        CHECK(rth_download_accumulator(h, acc.data()));
        CHECK(rth_save_checkpoint(h, "/tmp/rt_sanitize.ckpt"));
        CHECK(rth_load_checkpoint(h, "/tmp/rt_sanitize.ckpt"));
        CHECK(rth_render_to_image(h, 8, "/tmp/rt_sanitize.exr", rgba.data()));
        rt_ctx* ctx = rth_ctx(h);
        rt_scene* sc = rth_scene(h);
        rt_camera_desc cam;
        CHECK(rth_camera_desc(h, &cam));
        CHECK(rt_ctx_set_counters(ctx, 1));
        std::vector<float> host_acc(W * H * 4, 0.0f);
        CHECK(rt_render_accumulate(sc, &cam, 5, 1, 2, host_acc.data()));
        rt_counters cnt;
        CHECK(rt_ctx_get_counters(ctx, &cnt, 1));
        CHECK(rt_ctx_set_counters(ctx, 0));
        const size_t n = 1000;
        std::vector<uint32_t> pixels(n);
        for (size_t i = 0; i < n; ++i) pixels[i] = (uint32_t)((i * 7919u) % (W * H));
        std::vector<rt_ray> rays(n), out_rays(n);
        std::vector<rt_hit> hits(n);
        std::vector<float> att(3 * n), emit(3 * n);
        CHECK(rt_primary_rays(ctx, &cam, 9, 1, pixels.data(), n, rays.data()));
        CHECK(rt_intersect_batch(sc, rays.data(), n, 9, hits.data()));
        CHECK(rt_shade_batch(sc, rays.data(), hits.data(), n, 9, out_rays.data(), att.data(), emit.data()));
        // the render path's traversal kernels on the same rays; the entry point that allocates nothing; the tail kernel
        // from bounce 0 and never; a second context on the same GPU for the multi-GPU entry point (samples, then rows)
        std::vector<rt_hit> hits2(n);
        CHECK(rt_extend_batch(sc, rays.data(), n, 9, hits2.data()));
        for (size_t i = 0; i < n; ++i)
            if (hits2[i].hit != hits[i].hit || hits2[i].t != hits[i].t) { printf("scene %d: extend and intersect differ at ray %zu\n", id, i); return 1; }
        void* d_acc = nullptr;
        CHECK(rt_buffer_create(ctx, W * H * 16, &d_acc));
        CHECK(rt_buffer_zero(ctx, d_acc, W * H * 16));
        CHECK(rt_render_accumulate_device_noalloc(sc, &cam, 5, 1, 2, (float*)d_acc));
        for (int start : {0, 1 << 20, -1}) {
            CHECK(rt_ctx_set_tail_start(ctx, start));
            CHECK(rt_render_accumulate_device(sc, &cam, 5, 3, 2, (float*)d_acc));
        }
        void* h2 = rth_scene_builtin(id, W, H, 12, earth.data(), 1024, 512, 20000, 0);
        if (!h2) { printf("scene %d (second context): %s\n", id, rth_last_error()); return 1; }
        rt_scene* both[2] = {sc, rth_scene(h2)};
        CHECK(rt_render_accumulate_multi(both, 2, RT_SPLIT_SAMPLES, &cam, 5, 7, 3, (float*)d_acc));
        CHECK(rt_render_accumulate_multi(both, 2, RT_SPLIT_ROWS, &cam, 5, 10, 1, (float*)d_acc));
        CHECK(rt_ctx_synchronize(ctx));
        CHECK(rt_buffer_download(ctx, host_acc.data(), d_acc, W * H * 16));
        CHECK(rt_buffer_destroy(ctx, d_acc));
        rth_scene_free(h2);
        void* locked = nullptr;
        CHECK(rt_host_buffer_create(ctx, W * H * 16, &locked));
        std::memset(locked, 0, W * H * 16);
        CHECK(rt_render_accumulate(sc, &cam, 5, 1, 2, (float*)locked));
        CHECK(rt_render_accumulate(sc, &cam, 5, 3, 1, host_acc.data()));
        CHECK(rt_ctx_release_host_buffer(ctx));
        // the same with the caller's pageable buffer registered by the library (opt-in), and released again
        CHECK(rt_ctx_set_host_buffer_pinning(ctx, 1));
        CHECK(rt_render_accumulate(sc, &cam, 5, 4, 1, host_acc.data()));
        CHECK(rt_render_accumulate(sc, &cam, 5, 5, 1, host_acc.data()));
        CHECK(rt_ctx_set_host_buffer_pinning(ctx, 0));
        CHECK(rt_render_accumulate(sc, &cam, 5, 6, 1, host_acc.data()));
        size_t mem_free = 0, mem_total = 0;
        CHECK(rt_ctx_memory_info(ctx, &mem_free, &mem_total));
        if (mem_free == 0 || mem_free > mem_total) { printf("scene %d: memory info %zu of %zu\n", id, mem_free, mem_total); return 1; }
        CHECK(rt_host_buffer_destroy(ctx, locked));
        rt_scene_info info;
        CHECK(rt_scene_get_info(sc, &info));
        double sum = 0;
        for (float v : host_acc) sum += v;
        printf("scene %d: segments %llu, %u nodes, acc sum %.3f\n", id, (unsigned long long)cnt.segments, info.bvh_nodes, sum);
        rth_scene_free(h);
    }
    printf("all scenes done\n");
    // Every scene, context and buffer has been destroyed above.  The process ends here without running the exit handlers
    // of the libraries it loaded: with the sanitizer runtime loaded, the HSA runtime's own teardown ends in an internal
    // CHECK of its own (dev_runtime_unloaded_), after everything of this library is gone.
    fflush(stdout);
    _exit(0);
}
