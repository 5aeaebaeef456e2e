// prt_render.cpp -- headless equivalent of the reference's main() (src/main.cpp:323-485) on top of
// the prt C ABI: same argv flags (-scene -width -height -hdr -alpha), same start-up order
// (scene -> model -> BVH -> buffers -> camera -> kernel args), same per-frame protocol
// (frame counter from 1, two rand() values per frame after two consumed at start-up), with the
// GLFW loop replaced by "-frames N" or "-spp N" and PrtSc replaced by "-out file.pfm".
#include <cstdio>
#include <cstdlib>
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
    for (int y = h - 1; y >= 0; --y) {               // PFM is bottom-up; framebuffer row 0 is the top
        for (int x = 0; x < w; ++x)
            for (int c = 0; c < 3; ++c) row[(size_t)x * 3 + c] = rgba[((size_t)y * w + x) * 4 + c];
        std::fwrite(row.data(), sizeof(float), row.size(), f);
    }
    std::fclose(f);
    return true;
}

#define CHECK(call) do { int rc_ = (call); if (rc_ != PRT_OK) { \
    std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, ctx ? prt_last_error(ctx) : prt_last_global_error()); return 1; } } while (0)

int main(int argc, char** argv) {
    int window_width = 1280, window_height = 720;                 // src/main.cpp:55-57
    std::string scene_filepath = "../scenes/cornell.json";        // :90
    std::string models_directory = "../resources/models/";        // :35
    std::string env_map_filepath, out_path = "render.pfm";
    bool alpha = false;
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
        else if (a == "-encoder") next();                          // PNG/HDR encoder choice of the GL path: ignored
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
        if (!write_pfm(out_path, rgba, window_width, window_height)) { std::fprintf(stderr, "cannot write %s\n", out_path.c_str()); return 1; }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        if (ctx) prt_destroy(ctx);
        return 1;
    }
    prt_destroy(ctx);
    return 0;
}
