#include <cstdint>
#include <cstdio>
#include <vector>
#include "../../include/rt_abi.h"
extern "C" {
void* orc_scene_builtin(int id, uint32_t w, uint32_t h, uint32_t depth, const void* earth, uint32_t ew, uint32_t eh, uint64_t stress);
void orc_scene_free(void*);
void orc_render(void* p, int mode, uint64_t seed, uint32_t first_sample, uint32_t n_samples, float* acc, int threads, rt_counters* counters);
}
int main() {
    std::vector<uint8_t> earth(1024 * 512 * 3, 128);
    for (int id : {0, 1, 2, 3, 4, 5, 6, 7, 8, 100, 101}) {
        void* h = orc_scene_builtin(id, 32, 24, 12, earth.data(), 1024, 512, 1000);
        if (!h) { printf("scene %d failed\n", id); continue; }
        std::vector<float> acc(32 * 24 * 4, 0.0f);
        rt_counters c{};
        for (int mode : {0, 1}) orc_render(h, mode, 7, 1, 2, acc.data(), mode ? 4 : 1, &c);
        double sum = 0; for (float v : acc) sum += v;
        printf("scene %d: segments %llu sum %.4f\n", id, (unsigned long long)c.segments, sum);
        orc_scene_free(h);
    }
    return 0;
}
