#include "host_capi.h"
#include "hdr_loader.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "bvh.h"
#include "camera.h"
#include "model_loader.h"
#include "prt_detmath.h"
#include "scene.h"

struct prth_scene {
    prt::host_scene scene;
    std::shared_ptr<prt::IO::ModelLoader> ml;
    std::vector<float> vertices4, normals4;
    std::vector<uint64_t> indices;
    std::vector<prt_bvh_node> nodes;
    unsigned bvh_depth = 0;
};

static void set_err(char* err, int n, const std::string& msg) {
    if (err && n > 0) { std::snprintf(err, (size_t)n, "%s", msg.c_str()); }
}

static prth_scene* finish_load(std::unique_ptr<prth_scene> h, const char* models_dir, char* err, int err_len) {
    if (h->scene.BUILD_BVH) {                                   // src/main.cpp:401-415
        h->ml = std::make_shared<prt::IO::ModelLoader>();
        std::string dir = models_dir ? models_dir : "";
        if (!dir.empty() && dir.back() != '/') dir += '/';
        if (!h->ml->ImportFromFile(dir + h->scene.obj_path)) { set_err(err, err_len, h->ml->last_error()); return nullptr; }
        // build parameters are a host-side choice (the reference's builder is an absent dependency);
        // PRT_BVH_LEAF / PRT_BVH_COST override them for experiments
        const char* e_leaf = std::getenv("PRT_BVH_LEAF");
        const char* e_cost = std::getenv("PRT_BVH_COST");
        prt::BVH bvh(h->ml, e_leaf ? (unsigned)std::atoi(e_leaf) : 16u, e_cost ? (float)std::atof(e_cost) : 0.0f);
        h->nodes = *bvh.PrepareData();
        h->indices = *bvh.GetPrimitiveIndices();
        h->bvh_depth = bvh.max_depth();
        h->ml->flatten(h->vertices4, h->normals4);
    }
    return h.release();
}

extern "C" prth_scene* prth_scene_load(const char* path, const char* models_dir, char* err, int err_len) {
    try {
        std::unique_ptr<prth_scene> h(new prth_scene());
        h->scene.load(path);
        return finish_load(std::move(h), models_dir, err, err_len);
    } catch (const std::exception& e) { set_err(err, err_len, e.what()); return nullptr; }
}
extern "C" prth_scene* prth_scene_load_text(const char* text, const char* models_dir, char* err, int err_len) {
    try {
        std::unique_ptr<prth_scene> h(new prth_scene());
        h->scene.load_text(text);
        return finish_load(std::move(h), models_dir, err, err_len);
    } catch (const std::exception& e) { set_err(err, err_len, e.what()); return nullptr; }
}
extern "C" void prth_scene_free(prth_scene* s) { delete s; }

extern "C" int prth_scene_get_desc(const prth_scene* s, prt_scene_desc* out) {
    if (!s || !out) return PRT_ERR_INVALID_ARGUMENT;
    std::memset(out, 0, sizeof(*out));
    out->meshes = s->scene.cpu_meshes.data();
    for (int i = 0; i < 8; ++i) out->object_count[i] = s->scene.object_count[i];
    out->obj_material = &s->scene.obj_mat;
    out->vertices = s->vertices4.data();
    out->normals = s->normals4.data();
    out->primitive_indices = s->indices.data();
    out->triangle_count = (uint32_t)s->indices.size();
    out->bvh_nodes = s->nodes.data();
    out->bvh_node_count = (uint32_t)s->nodes.size();
    return PRT_OK;
}
extern "C" int prth_scene_get_config(const prth_scene* s, int alpha, prt_config* out) {
    if (!s || !out) return PRT_ERR_INVALID_ARGUMENT;
    *out = s->scene.make_config(alpha != 0);
    return PRT_OK;
}
extern "C" int prth_scene_bvh_depth(const prth_scene* s) { return s ? (int)s->bvh_depth : -1; }
extern "C" const char* prth_scene_obj_path(const prth_scene* s) { return s ? s->scene.obj_path.c_str() : ""; }

extern "C" int prth_default_camera(int w, int h, float fovx, prt_camera* out) {
    return prth_orbit_camera(w, h, fovx, 0.f, 0.f, 0.f, 0.f, 0.f, out);
}
extern "C" int prth_orbit_camera(int w, int h, float fovx, float dyaw, float dpitch, float dradius,
                                 float daperture, float dfocal, prt_camera* out) {
    if (!out || w <= 0 || h <= 0) return PRT_ERR_INVALID_ARGUMENT;
    prt::InteractiveCamera ic;
    ic.setResolution((float)w, (float)h);
    ic.setFOVX(fovx);
    if (dyaw != 0.f) ic.changeYaw(dyaw);
    if (dpitch != 0.f) ic.changePitch(dpitch);
    if (dradius != 0.f) ic.changeRadius(dradius);
    if (daperture != 0.f) ic.changeApertureDiameter(daperture);
    if (dfocal != 0.f) ic.changeFocalDistance(dfocal);
    ic.buildRenderCamera(out);
    return PRT_OK;
}

extern "C" int prth_seed_pairs(uint32_t first_frame, uint32_t n_frames, int32_t* out) {
    if (!out || first_frame == 0) return PRT_ERR_INVALID_ARGUMENT;
    // private copy of glibc's default rand() stream: rand() == random() seeded with 1
    char statebuf[128];
    struct random_data rd;
    std::memset(&rd, 0, sizeof(rd));
    std::memset(statebuf, 0, sizeof(statebuf));
    initstate_r(1u, statebuf, sizeof(statebuf), &rd);
    int32_t v;
    const uint64_t skip = 2 + 2ull * (first_frame - 1);
    for (uint64_t i = 0; i < skip; ++i) random_r(&rd, &v);
    for (uint32_t i = 0; i < 2 * n_frames; ++i) { random_r(&rd, &v); out[i] = v; }
    return PRT_OK;
}

extern "C" int prth_convert_model(const char* in_path, const char* out_path, char* err, int err_len) {
    prt::IO::ModelLoader ml;
    if (!ml.ImportFromFile(in_path)) { set_err(err, err_len, ml.last_error()); return PRT_ERR_INVALID_ARGUMENT; }
    if (!ml.SaveSoup(out_path)) { set_err(err, err_len, "cannot write soup"); return PRT_ERR_INVALID_ARGUMENT; }
    return PRT_OK;
}

// meshes of a model file and, per mesh, its triangles and the vertices left after welding (prt::IO::ModelLoader::weld): counts[3 * m + {0, 1, 2}]
extern "C" int prth_model_meshes(const char* path, uint32_t* counts, int max_meshes, char* err, int err_len) {
    prt::IO::ModelLoader ml;
    if (!path || !ml.ImportFromFile(path)) { set_err(err, err_len, path ? ml.last_error() : std::string("null argument")); return PRT_ERR_INVALID_ARGUMENT; }
    const auto& meshes = ml.getFaces().meshes;
    for (size_t m = 0; m < meshes.size() && (int)m < max_meshes && counts; ++m) {
        std::vector<prt::IO::Vertex> v;
        std::vector<uint32_t> idx;
        ml.weld(m, v, idx);
        bool same = idx.size() == 3 * meshes[m].faces.size();          // de-indexing must give the soup back
        for (size_t k = 0; same && k < idx.size(); ++k) {
            const prt::IO::Vertex& a = v[idx[k]];
            const prt::IO::Vertex& b = meshes[m].faces[k / 3].points[k % 3];
            same = a.pos.x == b.pos.x && a.pos.y == b.pos.y && a.pos.z == b.pos.z && a.nor.x == b.nor.x && a.nor.y == b.nor.y && a.nor.z == b.nor.z;
        }
        counts[3 * m] = (uint32_t)meshes[m].faces.size();
        counts[3 * m + 1] = (uint32_t)v.size();
        counts[3 * m + 2] = same ? 1u : 0u;
    }
    return (int)meshes.size();
}

extern "C" void* prth_hdr_load(const char* path, int* width, int* height, const float** rgb, char* err, int err_len) {
    auto* v = new std::vector<float>();
    std::string e;
    int w = 0, h = 0;
    if (!path || !width || !height || !rgb || !prt::IO::load_hdr(path, *v, w, h, e)) {
        if (err && err_len > 0) { std::strncpy(err, (path ? e : std::string("null argument")).c_str(), (size_t)err_len - 1); err[err_len - 1] = 0; }
        delete v;
        return nullptr;
    }
    *width = w; *height = h; *rgb = v->data();
    return v;
}
extern "C" void prth_hdr_free(void* handle) { delete static_cast<std::vector<float>*>(handle); }

extern "C" int prth_hdr_write(const char* path, const float* pixels, int width, int height, int channels, int bottom_up, char* err, int err_len) {
    std::string e;
    if (!path || !prt::IO::write_hdr(path, pixels, width, height, channels, bottom_up != 0, e)) {
        if (err && err_len > 0) { std::strncpy(err, path ? e.c_str() : "null path", (size_t)err_len - 1); err[err_len - 1] = 0; }
        return 1;
    }
    return 0;
}

extern "C" int prth_make_sky(int w, int h, float* rgb) {
    if (!rgb || w <= 0 || h <= 0) return PRT_ERR_INVALID_ARGUMENT;
    // equirect: u -> azimuth, v -> polar angle from +y (kernels/utils.cl:46).  Gradient sky,
    // warm horizon band, a sun lobe at azimuth 0.8*pi, elevation ~40 degrees; values up to ~40.
    const float PI = 3.14159274f;
    const float sun_phi = 0.8f * PI, sun_theta = 0.28f * PI;
    const float sx = prt_sin(sun_theta) * prt_cos(sun_phi), sy = prt_cos(sun_theta), sz = prt_sin(sun_theta) * prt_sin(sun_phi);
    for (int j = 0; j < h; ++j) {
        float theta = ((float)j + 0.5f) / (float)h * PI;
        float cy = prt_cos(theta), sy_ = prt_sin(theta);
        for (int i = 0; i < w; ++i) {
            float phi = (((float)i + 0.5f) / (float)w - 0.5f) * 2.0f * PI;
            float dx = sy_ * prt_cos(phi), dz = sy_ * prt_sin(phi);
            float up = prt_fmax(cy, 0.0f);
            float horizon = prt_exp(-8.0f * prt_fabs(cy));
            float d = prt_fmax(dx * sx + cy * sy + dz * sz, 0.0f);
            float d2 = d * d, d4 = d2 * d2, d8 = d4 * d4, d16 = d8 * d8, d64 = d16 * d16 * d16 * d16;
            float sun = 40.0f * d64 * d64 + 0.6f * d8;
            float ground = cy < 0.0f ? 0.15f : 0.0f;
            float* p = rgb + ((size_t)j * w + i) * 3;
            p[0] = 0.25f + 0.35f * (1.0f - up) + 0.5f * horizon + sun + ground;
            p[1] = 0.40f + 0.35f * (1.0f - up) + 0.4f * horizon + 0.9f * sun + ground;
            p[2] = 0.85f - 0.25f * (1.0f - up) + 0.3f * horizon + 0.7f * sun + ground * 0.8f;
        }
    }
    return PRT_OK;
}
