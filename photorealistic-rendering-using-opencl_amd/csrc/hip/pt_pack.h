// pt_pack.h -- reference-layout scene buffers -> the device records of pt_layout.h (host code, see pt_pack.cpp)
#pragma once
#include <string>
#include <vector>

#include <hip/hip_runtime.h>      // vector types of pt_layout.h

#include "prt.h"
#include "pt_layout.h"

namespace prt {

struct PackedScene {
    std::vector<NodePair> pairs;
    std::vector<TriGeom> tg;
    std::vector<TriNrm> tn;
    std::vector<DevSphere> spheres;
    std::vector<DevQuad> quads;
    std::vector<DevSdf> sdfs;
    std::vector<DevMaterial> mats;
    std::vector<uint32_t> light_tab;     // DevScene::light_tab
    DevScene sc{};             // every scalar field; the pointers (and env) are filled in by whoever owns the memory
};

// prt_config::env_importance_sampling: the sampling density of an environment map as cumulative sums (DevScene::env_cdf_rows / _cols)
void build_env_cdf(const float* rgb, int w, int h, std::vector<float>& rows, std::vector<float>& cols);

// PRT_OK, or the prt error code with its message in `err`.  Nothing outside `out` is touched.
int pack_scene(const prt_config& cfg, const prt_scene_desc* s, PackedScene& out, std::string& err);

}  // namespace prt
