// prt_render.cpp -- headless equivalent of the reference's main() (src/main.cpp:323-485) on top of
// the prt C ABI: same argv flags (-scene -width -height -hdr -alpha), same start-up order
// (scene -> model -> BVH -> buffers -> camera -> kernel args), same per-frame protocol
// (frame counter from 1, two rand() values per frame after two consumed at start-up), with the
// GLFW loop replaced by "-frames N" or "-spp N" and PrtSc replaced by "-out file.{png,hdr,pfm}" (default render.png / render.hdr by -encoder, as saveImage()).
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "bvh.h"
#include "camera.h"
#include "host_capi.h"
#include "model_loader.h"
#include "prt.h"
#include "scene.h"

static bool write_pfm(const std::string& path, const std::vector<float>& rgba, int w, int h) {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    std::fprintf(f, "PF\n%d %d\n-1.0\n", w, h);
    std::vector<float> row((size_t)w * 3);
    for (int y = 0; y < h; ++y) {                    // PFM is bottom-up, and so is the framebuffer (row 0 = lowest scan line)
        for (int x = 0; x < w; ++x)
            for (int c = 0; c < 3; ++c) row[(size_t)x * 3 + c] = rgba[((size_t)y * w + x) * 4 + c];
        std::fwrite(row.data(), sizeof(float), row.size(), f);
    }
    std::fclose(f);
    return true;
}

// PNG with stored (uncompressed) deflate blocks: no zlib needed.  `rgba` bottom-up, PNG rows are top-down
// (the reference flips on write too, include/GL/cl_gl_interop.h:139).
static bool write_png(const std::string& path, const std::vector<uint8_t>& rgba, int w, int h) {
    static uint32_t crc_table[256];
    static bool have_table = false;
    if (!have_table) {
        for (uint32_t n = 0; n < 256; ++n) { uint32_t c = n; for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; crc_table[n] = c; }
        have_table = true;
    }
    auto crc = [&](const std::vector<uint8_t>& d) { uint32_t c = 0xFFFFFFFFu; for (uint8_t b : d) c = crc_table[(c ^ b) & 0xFF] ^ (c >> 8); return c ^ 0xFFFFFFFFu; };
    auto be32 = [](std::vector<uint8_t>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); };
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    auto chunk = [&](const char* type, const std::vector<uint8_t>& data) {
        std::vector<uint8_t> len; be32(len, (uint32_t)data.size());
        std::vector<uint8_t> td(type, type + 4); td.insert(td.end(), data.begin(), data.end());
        std::vector<uint8_t> c; be32(c, crc(td));
        std::fwrite(len.data(), 1, 4, f); std::fwrite(td.data(), 1, td.size(), f); std::fwrite(c.data(), 1, 4, f);
    };
    const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::fwrite(sig, 1, 8, f);
    std::vector<uint8_t> ihdr; be32(ihdr, (uint32_t)w); be32(ihdr, (uint32_t)h);
    ihdr.insert(ihdr.end(), {8, 6, 0, 0, 0});                     // 8 bit, RGBA
    chunk("IHDR", ihdr);
    std::vector<uint8_t> raw;
    raw.reserve((size_t)h * ((size_t)w * 4 + 1));
    for (int y = h - 1; y >= 0; --y) { raw.push_back(0); raw.insert(raw.end(), rgba.begin() + (size_t)y * w * 4, rgba.begin() + (size_t)(y + 1) * w * 4); }
    std::vector<uint8_t> z = {0x78, 0x01};
    uint32_t a = 1, b = 0;
    for (size_t off = 0; off < raw.size();) {
        const size_t n = std::min<size_t>(65535, raw.size() - off);
        z.push_back(off + n == raw.size() ? 1 : 0);
        z.push_back(n & 0xFF); z.push_back(n >> 8); z.push_back(~n & 0xFF); z.push_back((~n >> 8) & 0xFF);
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
        for (size_t i = off; i < off + n; ++i) { a = (a + raw[i]) % 65521u; b = (b + a) % 65521u; }
        off += n;
    }
    be32(z, (b << 16) | a);
    chunk("IDAT", z);
    chunk("IEND", {});
    std::fclose(f);
    return true;
}

#define CHECK(call) do { int rc_ = (call); if (rc_ != PRT_OK) { \
    std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, ctx ? prt_last_error(ctx) : prt_last_global_error()); return 1; } } while (0)

int main(int argc, char** argv) {
    int window_width = 1280, window_height = 720;                 // src/main.cpp:55-57
    std::string scene_filepath = "../scenes/cornell.json";        // :90
    std::string models_directory = "../resources/models/";        // :35
    std::string env_map_filepath, out_path;
    int encoder = 0;                                               // { 0: ".png", 1: ".hdr" }, src/main.cpp:365-367
    bool alpha = false;
    uint32_t view = PRT_VIEW_RESULTS;                              // kernels/main.cl:15: a source edit in the reference, a flag here
    unsigned frames = 0, spp = 16;
    int device = 0;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "-scene") scene_filepath = next();
        else if (a == "-width") window_width = std::atoi(next());
        else if (a == "-height") window_height = std::atoi(next());
        else if (a == "-hdr") env_map_filepath = next();
        else if (a == "-alpha") alpha = true;
        else if (a == "-view") { const std::string v = next(); view = v == "normal" ? PRT_VIEW_NORMAL : v == "bvh_hit" ? PRT_VIEW_BVH_HIT : PRT_VIEW_RESULTS; }
        else if (a == "-encoder") encoder = std::atoi(next());
        else if (a == "-models") models_directory = next();
        else if (a == "-frames") frames = (unsigned)std::atoi(next());
        else if (a == "-spp") spp = (unsigned)std::atoi(next());
        else if (a == "-out") out_path = next();
        else if (a == "-device") device = std::atoi(next());
    }
    prt_ctx* ctx = nullptr;
    try {
        prt::host_scene scene;
        scene.load(scene_filepath);                                // :375-376
        prt_config cfg = scene.make_config(alpha);
        cfg.view_option = view;
        CHECK(prt_create(device, &cfg, &ctx));                     // initOpenCL(), :380

        std::vector<float> vertices4, normals4;
        std::unique_ptr<std::vector<prt::cl_BVHnode>> nodes;
        std::unique_ptr<std::vector<uint64_t>> indices;
        if (scene.BUILD_BVH) {                                     // :401-415
            auto ml = std::make_shared<prt::IO::ModelLoader>();
            if (!ml->ImportFromFile(models_directory + scene.obj_path)) { std::fprintf(stderr, "%s\n", ml->last_error().c_str()); return 1; }
            prt::BVH bvh(ml);
            nodes = bvh.PrepareData();
            indices = bvh.GetPrimitiveIndices();
            ml->flatten(vertices4, normals4);
        }
        prt_scene_desc d;
        std::memset(&d, 0, sizeof(d));
        d.meshes = scene.cpu_meshes.data();                        // :418
        for (int i = 0; i < 8; ++i) d.object_count[i] = scene.object_count[i];
        d.obj_material = &scene.obj_mat;
        if (indices) {
            d.vertices = vertices4.data(); d.normals = normals4.data(); d.primitive_indices = indices->data();
            d.triangle_count = (uint32_t)indices->size(); d.bvh_nodes = nodes->data(); d.bvh_node_count = (uint32_t)nodes->size();
        }
        CHECK(prt_upload_scene(ctx, &d));

        prt::InteractiveCamera ic;                                 // initCamera(), :312-319
        ic.setResolution((float)window_width, (float)window_height);
        ic.setFOVX(45.0f);
        prt::Camera cam;
        ic.buildRenderCamera(&cam);                                // render(), :294-297
        CHECK(prt_set_camera(ctx, &cam));
        if (env_map_filepath == "sky") {                           // no .hdr ships with the reference
            std::vector<float> sky((size_t)1024 * 512 * 3);
            prth_make_sky(1024, 512, sky.data());
            CHECK(prt_upload_envmap(ctx, sky.data(), 1024, 512));
        } else if (!env_map_filepath.empty()) {                    // loadHDR, include/GL/cl_gl_interop.h:77-80
            char herr[256] = "";
            int ew = 0, eh = 0;
            const float* erg = nullptr;
            void* hdr = prth_hdr_load(env_map_filepath.c_str(), &ew, &eh, &erg, herr, sizeof(herr));
            if (!hdr) { std::fprintf(stderr, "-hdr: %s\n", herr); prt_destroy(ctx); return 1; }
            const int rc_env = prt_upload_envmap(ctx, erg, ew, eh);
            prth_hdr_free(hdr);
            CHECK(rc_env);
        }
        CHECK(prt_resize(ctx, window_width, window_height));       // cl_flattenI, :451

        const unsigned max_frames = frames ? frames : spp * (unsigned)(cfg.max_bounces > 8 ? cfg.max_bounces : 8) + 64;
        std::vector<int32_t> seeds((size_t)max_frames * 2);
        prth_seed_pairs(1, max_frames, seeds.data());              // rand() protocol, :226-227,301-302
        if (frames) CHECK(prt_render_frames(ctx, 1, frames, seeds.data()));
        else CHECK(prt_render_spp(ctx, spp, max_frames, seeds.data(), nullptr));

        std::vector<float> rgba((size_t)window_width * window_height * 4);
        CHECK(prt_read_framebuffer(ctx, rgba.data()));             // saveImage(), include/GL/cl_gl_interop.h:144-160
        prt_stats st;
        CHECK(prt_query_counts(ctx, frames ? 0 : spp, &st));
        std::printf("%dx%d: %llu samples, %llu segments, %.1f ms on the device (%u launches)\n", window_width, window_height,
                    (unsigned long long)st.samples, (unsigned long long)st.segments, st.kernel_ms, st.launches);
        // saveImage(), include/GL/cl_gl_interop.h:144-160: -encoder 0 -> render.png (the displayed, tonemapped picture), -encoder 1 ->
        // render.hdr (the linear one); -out <file> picks the name, and the format by its extension (.png / .hdr / .pfm)
        if (out_path.empty()) out_path = encoder == 1 ? "render.hdr" : "render.png";
        auto ends_with = [&](const char* ext) { return out_path.size() > 4 && out_path.compare(out_path.size() - 4, 4, ext) == 0; };
        bool ok;
        if (ends_with(".png")) {                                   // encoder 0 of the reference: the tonemapped picture
            std::vector<uint8_t> ldr((size_t)window_width * window_height * 4);
            CHECK(prt_tonemap_rgba8(ctx, ldr.data()));
            ok = write_png(out_path, ldr, window_width, window_height);
        } else if (ends_with(".hdr")) {                            // encoder 1: the linear picture as Radiance RGBE
            char herr[256] = "";
            ok = prth_hdr_write(out_path.c_str(), rgba.data(), window_width, window_height, 4, 1, herr, sizeof(herr)) == 0;
            if (!ok) std::fprintf(stderr, "%s\n", herr);
        } else {                                                   // the linear picture, lossless
            ok = write_pfm(out_path, rgba, window_width, window_height);
        }
        if (!ok) { std::fprintf(stderr, "cannot write %s\n", out_path.c_str()); return 1; }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        if (ctx) prt_destroy(ctx);
        return 1;
    }
    prt_destroy(ctx);
    return 0;
}
