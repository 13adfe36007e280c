#include <cstdint>
#include <cstdio>
#include <vector>
#include "../../include/rt_abi.h"
extern "C" {
void* rth_scene_builtin(int id, uint32_t width, uint32_t height, uint32_t max_depth, const uint8_t* earth_rgb8, uint32_t earth_w, uint32_t earth_h, uint64_t stress_count, int device);
void rth_scene_free(void* p);
int rth_flat_counts(void* p, uint32_t* out);
uint32_t rth_export_leaves(void* p, void* out, uint32_t cap);
int rth_scene_desc(void* p, rt_scene_desc* out);
int rth_write_exr(const char* path, uint32_t w, uint32_t h, const float* rgba);
int rth_write_pfm(const char* path, uint32_t w, uint32_t h, const float* rgba);
const char* rth_last_error();
uint32_t rth_material_count(void* p);
}
int main() {
    std::vector<uint8_t> earth(1024 * 512 * 3, 128);
    for (int id : {0, 1, 2, 3, 4, 5, 6, 7, 8, 100, 101, 102}) {
        void* h = rth_scene_builtin(id, 64, 48, 10, earth.data(), 1024, 512, 5000, 0);
        if (!h) { printf("scene %d: %s\n", id, rth_last_error()); continue; }
        uint32_t c[11];
        rth_flat_counts(h, c);
        uint32_t n = rth_export_leaves(h, nullptr, 0);
        std::vector<char> buf((size_t)n * 256);
        rt_scene_desc d;
        rth_scene_desc(h, &d);
        printf("scene %d: spheres %u moving %u quads %u media %u world %u leaves %u materials %u\n", id, c[0], c[1], c[2], c[3], c[4], n, rth_material_count(h));
        rth_scene_free(h);
    }
    std::vector<float> img(7 * 5 * 4, 0.25f);
    printf("exr %d pfm %d\n", rth_write_exr("/tmp/rt_sanitize_t.exr", 7, 5, img.data()), rth_write_pfm("/tmp/rt_sanitize_t.pfm", 7, 5, img.data()));
    return 0;
}
