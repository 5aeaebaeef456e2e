// scene.cpp -- JSON scene loader producing the reference's host buffers bit for bit
// (checked against the reference's own loader output in tests/test_host_scene.py).
#include "scene.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

#include "json.h"

namespace prt {

Material default_material() {
    Material m;
    std::memset(&m, 0, sizeof(m));
    m.color[0] = m.color[1] = m.color[2] = 1.0f;                 // material.h:105
    m.eta[0] = 0.17229f; m.eta[1] = 0.36901f; m.eta[2] = 1.5478f; // Au_eta, material.h:86
    m.k[0] = 4.2223f; m.k[1] = 2.4628f; m.k[2] = 1.8063f;         // Au_k,   material.h:87
    m.roughness = 0.0f;
    m.t = PRT_MAT_DIFF;
    m.lobes = PRT_LOBE_DIFFUSE_R | PRT_LOBE_DIFFUSE_T;           // DiffuseLobe, material.h:110
    m.dist = PRT_DIST_BECKMANN;
    return m;
}

Mesh default_mesh() {
    Mesh m;
    std::memset(&m, 0, sizeof(m));
    m.mat = default_material();
    m.t = PRT_GEOM_SPHERE;
    return m;
}

// `1 << n` of the reference's parser (include/Scene/scene.h:90-93,101) narrowed to the field it lands in; a shift count
// a scene file should never hold (negative, or past the field) gives 0 instead of undefined behaviour
template <typename T>
static T shifted_bit(int n) { return (n >= 0 && n < (int)(8 * sizeof(T))) ? (T)(1u << n) : (T)0; }

static uint8_t lobes_for_type(uint16_t t, uint8_t current) {
    // scene.h:99-116 / :223-240 -- first matching branch wins, no branch leaves lobes untouched
    if (t & PRT_MAT_LIGHT) return PRT_LOBE_NULL;
    if (t & PRT_MAT_DIFF) return PRT_LOBE_DIFFUSE_R;
    if (t & PRT_MAT_COND) return PRT_LOBE_SPECULAR_R;
    if (t & (PRT_MAT_ROUGH_COND | PRT_MAT_COAT)) return PRT_LOBE_GLOSSY_R;
    if (t & PRT_MAT_DIEL) return PRT_LOBE_SPECULAR_R | PRT_LOBE_SPECULAR_T;
    if (t & PRT_MAT_ROUGH_DIEL) return PRT_LOBE_GLOSSY_R | PRT_LOBE_GLOSSY_T;
    return current;
}

void host_scene::parse_material(const void* jv, Material& m) {
    const json::Value& d = *static_cast<const json::Value*>(jv);
    if (d.HasMember("color") && d["color"].IsArray()) {
        const json::Value& c = d["color"];
        for (size_t p = 0; p < c.Size() && p < 4; ++p) m.color[p] = c[p].GetFloat();
    }
    if (d.HasMember("roughness") && d["roughness"].IsNumber()) m.roughness = d["roughness"].GetFloat();
    if (d.HasMember("dist") && d["dist"].IsInt()) m.dist = shifted_bit<uint8_t>(d["dist"].GetInt());
    if (d.HasMember("type") && d["type"].IsInt()) {
        m.t = shifted_bit<uint16_t>(d["type"].GetInt());
        m.lobes = lobes_for_type(m.t, m.lobes);
        if (m.t & (PRT_MAT_DIEL | PRT_MAT_ROUGH_DIEL)) {
            m.eta[0] = 1.5121f; m.eta[1] = 1.5180f; m.eta[2] = 1.5337f; m.eta[3] = 0.0f;  // BK7_eta, material.h:79
            if (d.HasMember("absorptive") && d["absorptive"].IsNumber()) {
                int cc = d["absorptive"].GetInt();
                if (cc) m.t |= (cc == 1) ? PRT_MAT_ABS_REFR : PRT_MAT_ABS_REFR2;
            }
        }
    }
    ACTIVE_MATS |= m.t;
}

void host_scene::get_lights() {
    for (uint32_t i = 0; i < object_count[7]; ++i)
        if (cpu_meshes[i].mat.t & PRT_MAT_LIGHT) { ++LIGHT_COUNT; LIGHT_INDICES.push_back(i); }
}

void host_scene::load(const std::string& path) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error("cannot open scene file '" + path + "'");
    std::stringstream ss;
    ss << f.rdbuf();
    load_text(ss.str());
}

static int member_int(const json::Value& o, const char* k, int dflt) { return o.HasMember(k) ? o[k].GetInt() : dflt; }
static float member_float(const json::Value& o, const char* k, float dflt) { return o.HasMember(k) ? o[k].GetFloat() : dflt; }

void host_scene::load_text(const std::string& text) {
    json::Value doc = json::parse(text);
    if (!doc.IsObject() || !doc.HasMember("scene")) throw std::runtime_error("scene file has no \"scene\" object");  // scene.h:146-147

    HAS_GLOBAL_MEDIUM = doc.HasMember("global_medium");                           // scene.h:150-158
    if (HAS_GLOBAL_MEDIUM) {
        const json::Value& gm = doc["global_medium"];
        GLOBAL_MEDIUM.density = member_float(gm, "density", 0.1f);
        GLOBAL_MEDIUM.sigmaA = GLOBAL_MEDIUM.density * member_float(gm, "sigmaA", 0.2f);
        GLOBAL_MEDIUM.sigmaS = GLOBAL_MEDIUM.density * member_float(gm, "sigmaS", 1.0f);
        GLOBAL_MEDIUM.sigmaT = GLOBAL_MEDIUM.sigmaA + GLOBAL_MEDIUM.sigmaS;
        GLOBAL_MEDIUM.absorptionOnly = (GLOBAL_MEDIUM.sigmaS == 0.0f);
    }
    if (doc.HasMember("settings")) {                                              // scene.h:161-171
        const json::Value& st = doc["settings"];
        MAX_BOUNCES = member_int(st, "MAX_BOUNCES", 12);
        MAX_DIFF_BOUNCES = member_int(st, "MAX_DIFF_BOUNCES", 4);
        MAX_SPEC_BOUNCES = member_int(st, "MAX_SPEC_BOUNCES", 4);
        MAX_TRANS_BOUNCES = member_int(st, "MAX_TRANS_BOUNCES", 12);
        MAX_SCATTERING_EVENTS = member_int(st, "MAX_SCATTERING_EVENTS", 12);
        MARCHING_STEPS = member_int(st, "MARCHING_STEPS", 128);
        SHADOW_MARCHING_STEPS = member_int(st, "SHADOW_MARCHING_STEPS", 64);
    }
    const json::Value& sc = doc["scene"];

    // ---- obj (scene.h:178-260)
    BUILD_BVH = sc.HasMember("obj") && sc["obj"].IsObject() && sc["obj"].HasMember("path") && sc["obj"]["path"].IsString();
    if (BUILD_BVH) {
        obj_path = sc["obj"]["path"].GetString();
        if (sc["obj"].HasMember("material") && sc["obj"]["material"].IsObject()) {
            // same fields as parse_material; the reference duplicates the code with a 3-entry colour loop
            const json::Value& m = sc["obj"]["material"];
            if (m.HasMember("color") && m["color"].IsArray())
                for (int p = 0; p < 3; ++p) obj_mat.color[p] = m["color"][(size_t)p].GetFloat();
            json::Value rest = m;
            for (auto it = rest.obj.begin(); it != rest.obj.end();) it = (it->first == "color") ? rest.obj.erase(it) : it + 1;
            parse_material(&rest, obj_mat);
        }
    }

    auto grow = [&](uint32_t slot, size_t n) {
        object_count[slot] = (uint32_t)n;
        object_count[7] += (uint32_t)n;
        cpu_meshes.resize(object_count[7], default_mesh());
    };
    auto read3 = [](const json::Value& a, float* dst) { for (size_t p = 0; p < 3; ++p) dst[p] = a[p].GetFloat(); };

    // ---- spheres (scene.h:263-304)
    if (sc.HasMember("spheres") && sc["spheres"].IsArray()) {
        const json::Value& arr = sc["spheres"];
        H_SPHERE = arr.Size() > 0;
        grow(0, arr.Size());
        for (size_t i = 0; i < arr.Size(); ++i) {
            Mesh& m = cpu_meshes[i];
            m.t = PRT_GEOM_SPHERE;
            if (arr[i].HasMember("pos") && arr[i]["pos"].IsArray()) read3(arr[i]["pos"], m.pos);
            if (arr[i].HasMember("radius") && arr[i]["radius"].IsNumber()) m.joker[0] = arr[i]["radius"].GetFloat();
            if (arr[i].HasMember("material") && arr[i]["material"].IsObject()) parse_material(&arr[i]["material"], m.mat);
        }
    }
    // ---- sdfs (scene.h:307-364): raymarched primitives, kernels/geometry/sdf.cl
    if (sc.HasMember("sdfs") && sc["sdfs"].IsArray()) {
        const json::Value& arr = sc["sdfs"];
        H_SDF = arr.Size() > 0;
        size_t at = object_count[0];
        grow(1, arr.Size());
        for (size_t i = 0; i < arr.Size(); ++i, ++at) {
            Mesh& m = cpu_meshes[at];
            m.t = PRT_GEOM_SDF;
            if (arr[i].HasMember("pos") && arr[i]["pos"].IsArray()) read3(arr[i]["pos"], m.pos);
            if (arr[i].HasMember("type") && arr[i]["type"].IsInt()) m.t |= shifted_bit<uint8_t>(arr[i]["type"].GetInt());
            if (arr[i].HasMember("params") && arr[i]["params"].IsArray())
                for (size_t p = 0; p < arr[i]["params"].Size() && p < 16; ++p) m.joker[p] = arr[i]["params"][p].GetFloat();
            if (arr[i].HasMember("material") && arr[i]["material"].IsObject()) parse_material(&arr[i]["material"], m.mat);
        }
    }
    // ---- boxes (scene.h:367-414)
    if (sc.HasMember("boxes") && sc["boxes"].IsArray()) {
        const json::Value& arr = sc["boxes"];
        H_BOX = arr.Size() > 0;
        size_t at = object_count[0] + object_count[1];
        grow(2, arr.Size());
        for (size_t i = 0; i < arr.Size(); ++i, ++at) {
            Mesh& m = cpu_meshes[at];
            m.t = PRT_GEOM_BOX;
            if (arr[i].HasMember("pos") && arr[i]["pos"].IsArray()) read3(arr[i]["pos"], m.pos);
            if (arr[i].HasMember("scale") && arr[i]["scale"].IsArray()) read3(arr[i]["scale"], m.joker);
            if (arr[i].HasMember("material") && arr[i]["material"].IsObject()) parse_material(&arr[i]["material"], m.mat);
        }
    }
    // ---- quads (scene.h:417-469)
    if (sc.HasMember("quads") && sc["quads"].IsArray()) {
        const json::Value& arr = sc["quads"];
        H_QUAD = arr.Size() > 0;
        size_t at = object_count[0] + object_count[1] + object_count[2];
        grow(3, arr.Size());
        for (size_t i = 0; i < arr.Size(); ++i, ++at) {
            Mesh& m = cpu_meshes[at];
            m.t = PRT_GEOM_QUAD;
            if (arr[i].HasMember("vertices") && arr[i]["vertices"].IsArray()) {
                const json::Value& v = arr[i]["vertices"];
                for (size_t p = 0; p < v.Size() && p < 16; ++p) m.joker[p] = v[p].GetFloat();
                const float* e0 = m.joker + 3; const float* e1 = m.joker + 6;
                // cross(), lengthsq3() (which returns the LENGTH, linear_algebra.h:79) and normalize()
                float nx = e0[1] * e1[2] - e0[2] * e1[1];
                float ny = e0[2] * e1[0] - e0[0] * e1[2];
                float nz = e0[0] * e1[1] - e0[1] * e1[0];
                float len = sqrtf(nx * nx + ny * ny + nz * nz);
                m.joker[12] = len;                           // parallelogram area
                m.joker[9] = nx / len; m.joker[10] = ny / len; m.joker[11] = nz / len;
            }
            if (arr[i].HasMember("material") && arr[i]["material"].IsObject()) parse_material(&arr[i]["material"], m.mat);
        }
    }
    if (ACTIVE_MATS & PRT_MAT_LIGHT) get_lights();            // scene.h:472-473
}

static float to_string_roundtrip(float v) {
    // std::to_string(float) prints "%f" of the promoted double; the kernel text then reads it
    // back as a float literal (cl_kernel.h:72-108, kernels/header.cl:46-50)
    char buf[64];
    std::snprintf(buf, sizeof(buf), "%f", (double)v);
    return std::strtof(buf, nullptr);
}

prt_config host_scene::make_config(bool alpha_testing) const {
    prt_config c;
    std::memset(&c, 0, sizeof(c));
    c.abi_version = PRT_ABI_VERSION;
    c.max_bounces = MAX_BOUNCES; c.max_diff_bounces = MAX_DIFF_BOUNCES; c.max_spec_bounces = MAX_SPEC_BOUNCES;
    c.max_trans_bounces = MAX_TRANS_BOUNCES; c.max_scattering_events = MAX_SCATTERING_EVENTS;
    c.marching_steps = MARCHING_STEPS; c.shadow_marching_steps = SHADOW_MARCHING_STEPS;
    c.active_mats = (uint32_t)ACTIVE_MATS;
    c.geom_flags = (H_SPHERE ? PRT_GEOM_SPHERE : 0) | (H_BOX ? PRT_GEOM_BOX : 0) | (H_SDF ? PRT_GEOM_SDF : 0) | (H_QUAD ? PRT_GEOM_QUAD : 0);
    c.light_count = LIGHT_COUNT;
    for (size_t i = 0; i < LIGHT_INDICES.size() && i < PRT_MAX_LIGHTS; ++i) c.light_indices[i] = LIGHT_INDICES[i];
    c.has_global_medium = HAS_GLOBAL_MEDIUM ? 1 : 0;
    if (HAS_GLOBAL_MEDIUM) {
        c.fog_density = to_string_roundtrip(GLOBAL_MEDIUM.density);
        c.fog_sigma_a = to_string_roundtrip(GLOBAL_MEDIUM.sigmaA);
        c.fog_sigma_s = to_string_roundtrip(GLOBAL_MEDIUM.sigmaS);
        c.fog_sigma_t = to_string_roundtrip(GLOBAL_MEDIUM.sigmaT);
        c.fog_abs_only = GLOBAL_MEDIUM.absorptionOnly ? 1 : 0;
    }
    c.alpha_testing = alpha_testing ? 1 : 0;
    c.phase_function = PRT_PHASE_ISOTROPIC;       // kernels/media.cl:61
    c.phase_g = 0.6f;                             // kernels/phasefunctions/HenyeyGreenstein.cl:4
    c.view_option = PRT_VIEW_RESULTS;             // kernels/main.cl:15
    c.pick_random_light = 0;                      // kernels/integrators/base.cl:9
    c.env_importance_sampling = 0;                // (not in the reference)
    return c;
}

}  // namespace prt
