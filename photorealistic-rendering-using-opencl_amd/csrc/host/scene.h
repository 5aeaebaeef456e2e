// scene.h -- host scene model: same public surface as the reference's `host_scene`
// (include/Scene/scene.h:18-475) so host code written against it keeps working, filled by an
// independent JSON loader.  Mesh/Material are the binary layouts of include/prt_types.h.
#pragma once
#include <string>
#include <vector>
#include "prt.h"

namespace prt {

using Material = prt_material;
using Mesh = prt_mesh;

// defaults of the reference's Material() constructor (include/Types/material.h:105-111):
// white, gold eta/k, roughness 0, DIFF, DiffuseLobe (both diffuse bits), Beckmann.
Material default_material();
// defaults of Mesh() (include/Scene/geometry.h:27): t = SPHERE, everything else zero here.
Mesh default_mesh();

struct cl_medium {           // include/Types/media.h:5-11
    float density = 0.f, sigmaA = 0.f, sigmaS = 0.f, sigmaT = 0.f;
    bool absorptionOnly = false;
};

struct host_scene {
    // n_sphere, n_sdf, n_box, n_quad, _, _, _, total   (scene.h:20-22)
    uint32_t object_count[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<Mesh> cpu_meshes;

    int MAX_BOUNCES = 12, MAX_DIFF_BOUNCES = 4, MAX_SPEC_BOUNCES = 4, MAX_TRANS_BOUNCES = 12,
        MAX_SCATTERING_EVENTS = 12;                         // scene.h:26-30
    bool H_SPHERE = false, H_SDF = false, H_BOX = false, H_QUAD = false;
    int ACTIVE_MATS = 0;
    uint32_t LIGHT_COUNT = 0;
    std::vector<uint32_t> LIGHT_INDICES;
    bool HAS_GLOBAL_MEDIUM = false;
    cl_medium GLOBAL_MEDIUM;
    int MARCHING_STEPS = 128, SHADOW_MARCHING_STEPS = 64;
    bool BUILD_BVH = false;
    std::string obj_path;
    Material obj_mat = default_material();

    // scene.h:134-474.  Throws std::runtime_error on malformed input (the reference asserts).
    void load(const std::string& scene_filepath);
    void load_text(const std::string& json_text);

    // The specialisation parameters the reference bakes into the kernel text
    // (include/CL/cl_kernel.h:13-446) as a prt_config, including the "%f" round trip of the
    // medium coefficients (std::to_string(float) + "f", cl_kernel.h:72-108).
    prt_config make_config(bool alpha_testing = false) const;

private:
    void parse_material(const void* json_value, Material& m);
    void get_lights();
};

}  // namespace prt
