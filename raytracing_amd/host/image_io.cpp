// image_io.cpp — the on-disk side of the render path: what `--image` leaves behind
// (src/main.rs:157-186) and a checkpoint of a running accumulation.
//
// The reference saves `out.exr` through the `image` crate: an Rgba32F image of
// hdr_render's sqrt(mean) values with alpha 1.  write_exr produces the same
// content as an uncompressed scanline OpenEXR file (channels A, B, G, R of type
// FLOAT), write_pfm the RGB part as a little-endian PFM.
#include "raytracing.hpp"

#include <cstdio>
#include <cstring>
#include <memory>
#include <string>

namespace raytracing {
namespace {

struct File {
    FILE* f = nullptr;
    explicit File(const std::string& path, const char* mode) : f(std::fopen(path.c_str(), mode)) {
        if (!f) throw RenderError("cannot open " + path);
    }
    ~File() { if (f) std::fclose(f); }
    // flushes and closes, reporting what the destructor could not (a full disk shows up here)
    void close() {
        FILE* g = f;
        f = nullptr;
        const bool flushed = g && std::fflush(g) == 0;
        const bool closed = g && std::fclose(g) == 0;
        if (g && !(flushed && closed)) throw RenderError("cannot finish writing the file");
    }
    void put(const void* p, size_t n) { if (n && std::fwrite(p, 1, n, f) != n) throw RenderError("short write"); }
    void get(void* p, size_t n) { if (n && std::fread(p, 1, n, f) != n) throw RenderError("short read"); }
    template <class T> void put(T v) { put(&v, sizeof v); }
    template <class T> T get() { T v; get(&v, sizeof v); return v; }
    void str(const char* s) { put(s, std::strlen(s) + 1); }
};

void attribute(File& f, const char* name, const char* type, const void* value, int32_t size) {
    f.str(name);
    f.str(type);
    f.put<int32_t>(size);
    f.put(value, (size_t)size);
}

}  // namespace

void write_exr(const std::string& path, size_t width, size_t height, const float* rgba) {
    if (width == 0 || height == 0 || width > 0x7FFFFFFF / 16 || height > 0x7FFFFFFF) throw RenderError("write_exr: bad image size");
    File f(path, "wb");
    const unsigned char magic[8] = {0x76, 0x2f, 0x31, 0x01, 2, 0, 0, 0};  // OpenEXR, version 2, single-part scanline
    f.put(magic, 8);
    {   // channel list, alphabetical: name, type FLOAT (2), pLinear, reserved, x/y sampling
        std::string ch;
        for (const char* name : {"A", "B", "G", "R"}) {
            ch += name;
            ch.push_back('\0');
            const int32_t type = 2, sampling = 1;
            const unsigned char linear[4] = {0, 0, 0, 0};
            ch.append((const char*)&type, 4);
            ch.append((const char*)linear, 4);
            ch.append((const char*)&sampling, 4);
            ch.append((const char*)&sampling, 4);
        }
        ch.push_back('\0');
        attribute(f, "channels", "chlist", ch.data(), (int32_t)ch.size());
    }
    const unsigned char none = 0;
    attribute(f, "compression", "compression", &none, 1);
    const int32_t window[4] = {0, 0, (int32_t)width - 1, (int32_t)height - 1};
    attribute(f, "dataWindow", "box2i", window, 16);
    attribute(f, "displayWindow", "box2i", window, 16);
    attribute(f, "lineOrder", "lineOrder", &none, 1);  // increasing y
    const float one = 1.0f, centre[2] = {0.0f, 0.0f};
    attribute(f, "pixelAspectRatio", "float", &one, 4);
    attribute(f, "screenWindowCenter", "v2f", centre, 8);
    attribute(f, "screenWindowWidth", "float", &one, 4);
    f.put<unsigned char>(0);  // end of header

    const uint64_t line_bytes = (uint64_t)width * 16;
    uint64_t offset = (uint64_t)std::ftell(f.f) + 8ull * height;
    for (size_t y = 0; y < height; ++y, offset += 8 + line_bytes) f.put<uint64_t>(offset);
    std::vector<float> line(4 * width);
    for (size_t y = 0; y < height; ++y) {
        const float* src = rgba + 4 * width * y;
        for (size_t x = 0; x < width; ++x) {
            line[0 * width + x] = src[4 * x + 3];  // A
            line[1 * width + x] = src[4 * x + 2];  // B
            line[2 * width + x] = src[4 * x + 1];  // G
            line[3 * width + x] = src[4 * x + 0];  // R
        }
        f.put<int32_t>((int32_t)y);
        f.put<int32_t>((int32_t)line_bytes);
        f.put(line.data(), (size_t)line_bytes);
    }
}

void write_pfm(const std::string& path, size_t width, size_t height, const float* rgba) {
    File f(path, "wb");
    std::string head = "PF\n" + std::to_string(width) + " " + std::to_string(height) + "\n-1.0\n";
    f.put(head.data(), head.size());
    std::vector<float> line(3 * width);
    for (size_t y = height; y-- > 0;) {  // PFM rows run bottom to top
        const float* src = rgba + 4 * width * y;
        for (size_t x = 0; x < width; ++x) {
            line[3 * x] = src[4 * x];
            line[3 * x + 1] = src[4 * x + 1];
            line[3 * x + 2] = src[4 * x + 2];
        }
        f.put(line.data(), line.size() * sizeof(float));
    }
}

// `--image` (src/main.rs:157-186): sample_count - 1 empty_render calls, one hdr_render, alpha 1.
// The empty_render calls go to the GPU as one submission.
std::vector<float> Camera::render_to_image(size_t sample_count, const std::string& path) {
    if (sample_count == 0) throw RenderError("render_to_image: sample_count is 0");
    if (sample_count > 1) render_samples((uint32_t)(sample_count - 1));
    std::vector<Vec3> buffer = hdr_render();
    const size_t w = render_resolution_.first, h = render_resolution_.second;
    std::vector<float> rgba(4 * w * h);
    for (size_t i = 0; i < w * h; ++i) {
        rgba[4 * i] = buffer[i][0];
        rgba[4 * i + 1] = buffer[i][1];
        rgba[4 * i + 2] = buffer[i][2];
        rgba[4 * i + 3] = 1.0f;
    }
    if (!path.empty()) {
        if (path.size() >= 4 && path.compare(path.size() - 4, 4, ".pfm") == 0) write_pfm(path, w, h, rgba.data());
        else write_exr(path, w, h, rgba.data());
    }
    return rgba;
}

// ---- checkpoint: accumulation buffer + how many samples it holds -------------------------
namespace {
struct CheckpointHeader {
    char magic[8];
    uint32_t version, width, height, max_depth;
    uint64_t samples, seed;
    uint64_t camera_hash;   // FNV-1a of the camera description the accumulation was rendered with
    uint32_t n_leaves, _pad;
};
const char CHECKPOINT_MAGIC[8] = {'R', 'T', 'A', 'M', 'D', 'C', 'K', '2'};
uint64_t fnv1a(const void* p, size_t n) {
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; ++i) h = (h ^ ((const unsigned char*)p)[i]) * 0x100000001b3ull;
    return h;
}
}  // namespace

void Camera::save_checkpoint(const std::string& path) {
    std::vector<float> acc = download_accumulator();
    CheckpointHeader h{};
    std::memcpy(h.magic, CHECKPOINT_MAGIC, 8);
    h.version = 2;
    h.width = (uint32_t)render_resolution_.first;
    h.height = (uint32_t)render_resolution_.second;
    h.max_depth = (uint32_t)max_depth_;
    h.samples = samples_;
    h.seed = seed_;
    const rt_camera_desc cam = raytracing_camera_desc();
    h.camera_hash = fnv1a(&cam, sizeof cam);
    h.n_leaves = (uint32_t)flat_.world.size();
    // written beside the target and moved over it when complete: a crash or a full disk never leaves a
    // truncated file where the previous checkpoint was
    const std::string tmp = path + ".tmp";
    try {
        File f(tmp, "wb");
        f.put(h);
        f.put(acc.data(), acc.size() * sizeof(float));
        f.close();
    } catch (...) {
        std::remove(tmp.c_str());
        throw;
    }
    if (std::rename(tmp.c_str(), path.c_str()) != 0) {
        std::remove(tmp.c_str());
        throw RenderError("cannot move the checkpoint into place: " + path);
    }
}

void Camera::load_checkpoint(const std::string& path) {
    File f(path, "rb");
    CheckpointHeader h = f.get<CheckpointHeader>();
    if (std::memcmp(h.magic, CHECKPOINT_MAGIC, 8) != 0 || h.version != 2) throw RenderError("not a checkpoint of this library: " + path);
    if (h.width != render_resolution_.first || h.height != render_resolution_.second || h.max_depth != max_depth_)
        throw RenderError("checkpoint was written for another resolution or depth");
    // resuming adds samples to the sums in the file: they must be sums of the same image
    const rt_camera_desc cam = raytracing_camera_desc();
    if (h.camera_hash != fnv1a(&cam, sizeof cam) || h.n_leaves != (uint32_t)flat_.world.size())
        throw RenderError("checkpoint was written for another camera pose or another scene");
    std::vector<float> acc(4 * (size_t)h.width * h.height);
    f.get(acc.data(), acc.size() * sizeof(float));
    upload_accumulator(acc.data(), (size_t)h.samples);
    seed_ = h.seed;
}

}  // namespace raytracing
