// pt_pack.cpp -- host side of prt_upload_scene: validates the reference-layout scene buffers (include/prt.h,
// src/main.cpp:93-122,401-418) and re-packs them into the records of pt_layout.h.  Pure host code (no HIP call), so
// a rejected scene leaves the context untouched and tests/emu can run the same packing without a device.
#include "pt_pack.h"

#include <cmath>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <utility>

namespace prt {

static DevMaterial pack_material(const prt_material& m) {
    DevMaterial d;
    std::memset(&d, 0, sizeof(d));
    for (int i = 0; i < 3; ++i) { d.color[i] = m.color[i]; d.eta[i] = m.eta[i]; d.k[i] = m.k[i]; }
    d.roughness = m.roughness;
    d.bits = (uint32_t)m.t | ((uint32_t)m.lobes << 16) | ((uint32_t)m.dist << 24);
    return d;
}

// Sampling density of an environment map for prt_config::env_importance_sampling: per texel the largest luminance of its 3 x 3
// neighbourhood (the bilinear lookup of a direction inside the texel can reach that far; columns wrap, rows clamp) x sin(theta) of its
// row, plus 5 % of the mean as a floor -- positive wherever the lookup can be.  Normalised cumulative sums in double precision.
void build_env_cdf(const float* rgb, int w, int h, std::vector<float>& rows, std::vector<float>& cols) {
    std::vector<double> f((size_t)w * h);
    auto lum = [&](int i, int j) {
        i = (i % w + w) % w; j = j < 0 ? 0 : (j >= h ? h - 1 : j);
        const float* p = rgb + ((size_t)j * w + i) * 3;
        const double l = 0.2126 * p[0] + 0.7152 * p[1] + 0.0722 * p[2];
        return l > 0.0 && l < 1e30 ? l : 0.0;
    };
    double total = 0.0;
    for (int j = 0; j < h; ++j) {
        const double st = std::sin(3.14159265358979323846 * (j + 0.5) / h);
        for (int i = 0; i < w; ++i) {
            double m = 0.0;
            for (int dj = -1; dj <= 1; ++dj) for (int di = -1; di <= 1; ++di) m = std::fmax(m, lum(i + di, j + dj));
            f[(size_t)j * w + i] = m * st;
            total += m * st;
        }
    }
    const double floor_ = total > 0.0 ? 0.05 * total / ((double)w * h) : 1.0;
    rows.assign((size_t)h + 1, 0.0f);
    cols.assign((size_t)h * (w + 1), 0.0f);
    std::vector<double> row_sum(h);
    double all = 0.0;
    for (int j = 0; j < h; ++j) {
        const double st = std::sin(3.14159265358979323846 * (j + 0.5) / h);
        double s = 0.0;
        for (int i = 0; i < w; ++i) { f[(size_t)j * w + i] += floor_ * st / 0.6366197723675814; s += f[(size_t)j * w + i]; }   // (floor in proportion to the solid angle)
        row_sum[j] = s; all += s;
    }
    double acc = 0.0;
    for (int j = 0; j < h; ++j) {
        rows[j] = (float)(acc / all);
        acc += row_sum[j];
        double c = 0.0;
        float* cj = cols.data() + (size_t)j * (w + 1);
        for (int i = 0; i < w; ++i) { cj[i] = (float)(c / row_sum[j]); c += f[(size_t)j * w + i]; }
        cj[w] = 1.0f;
    }
    rows[h] = 1.0f;
}

int pack_scene(const prt_config& cfg, const prt_scene_desc* s, PackedScene& out, std::string& err) {
    auto fail = [&err](int, int code, const char* msg) { err = msg; return code; };
    const int c = 0;
    if (!s) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_upload_scene: null scene");
    const uint32_t n_sph = s->object_count[0], n_sdf = s->object_count[1], n_box = s->object_count[2], n_quad = s->object_count[3];
    const uint32_t n_mesh = s->object_count[7];
    if (n_box) return fail(c, PRT_ERR_UNSUPPORTED, "prt_upload_scene: box primitives never render in the reference (box.cl is dead code)");
    if (n_sph + n_sdf + n_quad != n_mesh) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_upload_scene: object_count does not add up");
    if (n_sdf && !(cfg.geom_flags & PRT_GEOM_SDF)) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_upload_scene: SDF meshes but the config has no H_SDF");
    if (n_mesh && !s->meshes) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_upload_scene: meshes is null");
    if (n_mesh >= (1u << 23)) return fail(c, PRT_ERR_UNSUPPORTED, "prt_upload_scene: 2^23 primitives or more (a hit record keeps the mesh index in 24 bits)");
    const uint32_t T = s->triangle_count, N = s->bvh_node_count;
    if (T >= (1u << 29)) return fail(c, PRT_ERR_UNSUPPORTED, "prt_upload_scene: 2^29 triangles or more");
    if (T && (!s->vertices || !s->normals || !s->primitive_indices || !s->bvh_nodes || !N))
        return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_upload_scene: triangle buffers incomplete");

    // ---- primitives + materials
    std::vector<DevSphere>& spheres = out.spheres;
    std::vector<DevQuad>& quads = out.quads;
    std::vector<DevSdf>& sdfs = out.sdfs;
    std::vector<DevMaterial>& mats = out.mats;
    spheres.assign(n_sph, DevSphere{}); quads.assign(n_quad, DevQuad{}); sdfs.assign(n_sdf, DevSdf{}); mats.assign(n_mesh + 2, DevMaterial{});
    std::memset(mats.data(), 0, mats.size() * sizeof(DevMaterial));
    for (uint32_t i = 0; i < n_mesh; ++i) {
        const prt_mesh& m = s->meshes[i];
        mats[1 + i] = pack_material(m.mat);
        if (i < n_sph) {
            if (!(m.t & PRT_GEOM_SPHERE)) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_upload_scene: mesh order/type mismatch (sphere expected)");
            DevSphere& d = spheres[i];
            d.pos[0] = m.pos[0]; d.pos[1] = m.pos[1]; d.pos[2] = m.pos[2]; d.radius = m.joker[0];
        } else if (i < n_sph + n_sdf) {
            if (!(m.t & PRT_GEOM_SDF)) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_upload_scene: mesh order/type mismatch (sdf expected)");
            DevSdf& d = sdfs[i - n_sph];
            d.pos[0] = m.pos[0]; d.pos[1] = m.pos[1]; d.pos[2] = m.pos[2]; d.type = m.t;
            for (int k = 0; k < 4; ++k) d.params[k] = m.joker[k];
        } else {
            if (!(m.t & PRT_GEOM_QUAD)) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_upload_scene: mesh order/type mismatch (quad expected)");
            DevQuad& d = quads[i - n_sph - n_sdf];
            std::memset(&d, 0, sizeof(d));
            pack_quad(m.joker, d);
        }
    }
    if (s->obj_material) mats[n_mesh + 1] = pack_material(*s->obj_material);

    // ---- BVH: reference layout -> NodePair records (inner nodes only), DFS order
    std::vector<NodePair>& pairs = out.pairs;
    pairs.clear();
    DevScene sc{};
    sc.root_is_leaf = 1;
    sc.stack_levels = 1;
    // Triangle SLOTS: the triangles are gathered leaf by leaf in the order of the NodePair records, a pair's left leaf child
    // first, so that the leaves a walk step finds are ONE run of slots (pt_device.h walk_box / walk_tri).  slot_src[slot] =
    // position in primitive_indices.  A tree whose leaves share triangles gets one slot per reference (bounded: 4 x T).
    std::vector<uint32_t> slot_src;
    const uint64_t max_slots = 4ull * T < (1ull << 29) ? 4ull * T : (1ull << 29) - 1;
    auto add_leaf = [&](const prt_bvh_node& nd, uint32_t& first_slot) {
        if ((uint64_t)slot_src.size() + nd.primitive_count > max_slots) return false;
        first_slot = (uint32_t)slot_src.size();
        for (uint32_t j = 0; j < nd.primitive_count; ++j) slot_src.push_back(nd.first_child_or_primitive + j);
        return true;
    };
    if (T) {
        const prt_bvh_node* nodes = s->bvh_nodes;
        auto leaf_ok = [&](const prt_bvh_node& nd) { return (uint64_t)nd.first_child_or_primitive + nd.primitive_count <= T; };
        if (nodes[0].is_leaf) {
            if (!leaf_ok(nodes[0])) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_upload_scene: root leaf range out of bounds");
            if (!add_leaf(nodes[0], sc.root_leaf_first)) return fail(c, PRT_ERR_UNSUPPORTED, "prt_upload_scene: the leaves reference more than 4 x the triangles");
            sc.root_leaf_count = nodes[0].primitive_count;
        } else {
            sc.root_is_leaf = 0;
            std::vector<uint32_t> pair_of(N, 0xFFFFFFFFu);
            // Order of the NodePair records in memory.  Breadth-first keeps the top of the tree (which every ray
            // walks) contiguous; depth-first (pre-order, left child right behind its parent) gives deep walks through
            // big trees better line / page locality.  PRT_PAIR_ORDER=bfs|dfs overrides the choice.
            std::vector<uint32_t> order;
            const char* e_order = std::getenv("PRT_PAIR_ORDER");
            const bool dfs = e_order ? (std::strcmp(e_order, "dfs") == 0) : (N > 65536u);
            // "treelet[:K]": subtrees of at most K pairs (K x 64 B contiguous) grown from their root by surface area -- the child a ray is
            // most likely to visit next joins first --, the subtrees hanging off a treelet's frontier laid out behind it, depth-first
            unsigned treelet_k = 0;
            if (e_order && std::strncmp(e_order, "treelet", 7) == 0) { treelet_k = e_order[7] == ':' ? (unsigned)std::atoi(e_order + 8) : 512u; if (treelet_k < 2u) treelet_k = 2u; }
            order.reserve(N / 2 + 1);
            if (treelet_k) {
                auto area = [&](uint32_t n) {
                    const float* b = nodes[n].bounds;
                    const double dx = (double)b[1] - b[0], dy = (double)b[3] - b[2], dz = (double)b[5] - b[4];
                    return dx * dy + dy * dz + dz * dx;
                };
                std::vector<uint32_t> roots{0};
                std::vector<std::pair<double, uint32_t>> frontier;        // max-heap on area
                while (!roots.empty()) {
                    const uint32_t r = roots.back();
                    roots.pop_back();
                    frontier.clear();
                    frontier.push_back({area(r), r});
                    unsigned count = 0;
                    while (!frontier.empty() && count < treelet_k) {
                        std::pop_heap(frontier.begin(), frontier.end());
                        const uint32_t n = frontier.back().second;
                        frontier.pop_back();
                        if (n >= N || pair_of[n] != 0xFFFFFFFFu || order.size() >= N) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_upload_scene: BVH is not a tree");
                        pair_of[n] = (uint32_t)order.size();
                        order.push_back(n);
                        ++count;
                        const uint32_t fc = nodes[n].first_child_or_primitive;
                        if ((uint64_t)fc + 1 >= N) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_upload_scene: BVH child index out of range");
                        for (uint32_t ch = fc; ch <= fc + 1; ++ch)
                            if (!nodes[ch].is_leaf) { frontier.push_back({area(ch), ch}); std::push_heap(frontier.begin(), frontier.end()); }
                    }
                    // what is left on the frontier are the roots of the treelets below this one: the biggest first (popped last from `roots`... pushed smallest first)
                    std::sort(frontier.begin(), frontier.end());
                    for (const auto& f : frontier) roots.push_back(f.second);
                }
            } else if (!dfs) {
                order.push_back(0);
                pair_of[0] = 0;
                for (size_t head = 0; head < order.size(); ++head) {
                    const uint32_t n = order[head];
                    const uint32_t fc = nodes[n].first_child_or_primitive;
                    if ((uint64_t)fc + 1 >= N) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_upload_scene: BVH child index out of range");
                    for (uint32_t ch = fc; ch <= fc + 1; ++ch) {
                        if (nodes[ch].is_leaf) continue;
                        if (pair_of[ch] != 0xFFFFFFFFu || order.size() >= N) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_upload_scene: BVH is not a tree");
                        pair_of[ch] = (uint32_t)order.size();
                        order.push_back(ch);
                    }
                }
            } else {
                std::vector<uint32_t> st{0};
                while (!st.empty()) {
                    const uint32_t n = st.back();
                    st.pop_back();
                    if (n >= N || pair_of[n] != 0xFFFFFFFFu || order.size() >= N) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_upload_scene: BVH is not a tree");
                    pair_of[n] = (uint32_t)order.size();
                    order.push_back(n);
                    const uint32_t fc = nodes[n].first_child_or_primitive;
                    if ((uint64_t)fc + 1 >= N) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_upload_scene: BVH child index out of range");
                    if (!nodes[fc + 1].is_leaf) st.push_back(fc + 1);
                    if (!nodes[fc].is_leaf) st.push_back(fc);
                }
            }
            pairs.resize(order.size());
            for (size_t k = 0; k < order.size(); ++k) {
                const prt_bvh_node& nd = nodes[order[k]];
                NodePair& p = pairs[k];
                for (int ch = 0; ch < 2; ++ch) {
                    const prt_bvh_node& cn = nodes[nd.first_child_or_primitive + ch];
                    for (int j = 0; j < 6; ++j) p.b[6 * ch + j] = cn.bounds[j];
                    if (cn.is_leaf) {
                        if (!leaf_ok(cn)) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_upload_scene: leaf range out of bounds");
                        if (cn.primitive_count == 0xFFFFFFFFu) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_upload_scene: bad leaf");
                        // the walk keeps the triangles a step found pending in a 25-bit count (WalkState::pend_count = both leaf children of a pair)
                        if (cn.primitive_count >= (1u << 24)) return fail(c, PRT_ERR_UNSUPPORTED, "prt_upload_scene: a leaf of 2^24 triangles or more");
                        if (!add_leaf(cn, p.meta[2 * ch])) return fail(c, PRT_ERR_UNSUPPORTED, "prt_upload_scene: the leaves reference more than 4 x the triangles");
                        p.meta[2 * ch + 1] = cn.primitive_count;
                    } else {
                        p.meta[2 * ch] = pair_of[nd.first_child_or_primitive + ch];
                        p.meta[2 * ch + 1] = 0xFFFFFFFFu;
                    }
                }
            }
            // Most entries a walk can hold: one push per pair with two inner children on the way down.
            // The reference's stack has 64 entries (bvh.cl:131) and overflows silently beyond that.
            uint32_t max_sp = 0;
            std::vector<std::pair<uint32_t, uint32_t>> todo{{0u, 0u}};
            while (!todo.empty()) {
                const std::pair<uint32_t, uint32_t> it = todo.back();
                todo.pop_back();
                const NodePair& p = pairs[it.first];
                const bool in0 = p.meta[1] == 0xFFFFFFFFu, in1 = p.meta[3] == 0xFFFFFFFFu;
                const uint32_t sp = it.second + ((in0 && in1) ? 1u : 0u);
                if (sp > max_sp) max_sp = sp;
                if (in0) todo.push_back({p.meta[0], sp});
                if (in1) todo.push_back({p.meta[2], sp});
            }
            if (max_sp > 64u) return fail(c, PRT_ERR_UNSUPPORTED, "prt_upload_scene: BVH needs more than the 64 traversal-stack entries of the reference (bvh.cl:131)");
            sc.stack_levels = max_sp + 1;
        }
    } else {
        sc.root_leaf_first = 0; sc.root_leaf_count = 0;        // "no OBJ" = empty leaf root
    }
    // ---- triangles in slot order
    std::vector<TriGeom>& tg = out.tg;
    std::vector<TriNrm>& tn = out.tn;
    const size_t S = slot_src.size();
    for (uint32_t i = 0; i < T; ++i)
        if ((uint64_t)((uint32_t)s->primitive_indices[i] * 3u) + 2 >= (uint64_t)T * 3) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_upload_scene: primitive index out of range");
    tg.assign(S, TriGeom{}); tn.assign(S, TriNrm{});
    for (size_t i = 0; i < S; ++i) {
        const uint32_t fv = (uint32_t)s->primitive_indices[slot_src[i]] * 3u;        // triangle.cl:7 (uint arithmetic)
        const float* p0 = s->vertices + 4 * (size_t)fv;
        const float* p1 = p0 + 4; const float* p2 = p0 + 8;
        TriGeom& g = tg[i];
        float e1[3], e2[3];
        for (int k = 0; k < 3; ++k) { g.p0[k] = p0[k]; e1[k] = p0[k] - p1[k]; e2[k] = p2[k] - p0[k]; }   // triangle.cl:12-13
        for (int k = 0; k < 3; ++k) { g.e1[k] = e1[k]; g.e2[k] = e2[k]; }
        g.n[0] = e1[1] * e2[2] - e1[2] * e2[1];                                                       // triangle.cl:15
        g.n[1] = e1[2] * e2[0] - e1[0] * e2[2];
        g.n[2] = e1[0] * e2[1] - e1[1] * e2[0];
        const float* n0 = s->normals + 4 * (size_t)fv;
        for (int k = 0; k < 3; ++k) { tn[i].n0[k] = n0[k]; tn[i].n1[k] = n0[4 + k]; tn[i].n2[k] = n0[8 + k]; }
        tn[i].n0[3] = tn[i].n1[3] = tn[i].n2[3] = 0.0f;
    }

    sc.n_pairs = (uint32_t)pairs.size();
    sc.n_spheres = n_sph; sc.n_quads = n_quad; sc.quad_mesh_base = n_sph + n_sdf; sc.n_meshes = n_mesh; sc.n_sdfs = n_sdf;
    sc.marching_steps = cfg.marching_steps; sc.shadow_marching_steps = cfg.shadow_marching_steps;
    sc.light_sphere = sc.light_quad = 0xFFFFFFFFu; sc.light_mesh = 0;
    sc.light_count = cfg.light_count; sc.pick_random_light = cfg.pick_random_light ? 1u : 0u;
    out.light_tab.assign(cfg.light_count + 1u, 0u);             // [light_count]: the entry behind LIGHT_INDICES (include/prt.h)
    for (uint32_t k = 0; k < cfg.light_count && k < PRT_MAX_LIGHTS; ++k) {
        if (cfg.light_indices[k] >= n_mesh) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_upload_scene: light index out of range");
        out.light_tab[k] = cfg.light_indices[k];
    }
    if (cfg.pick_random_light) {
        if (!n_mesh) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_upload_scene: pick_random_light without a mesh");
        if (cfg.light_count >= PRT_MAX_LIGHTS) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_upload_scene: pick_random_light needs light_count < PRT_MAX_LIGHTS");
        if (n_sdf || cfg.view_option != PRT_VIEW_RESULTS)
            return fail(c, PRT_ERR_UNSUPPORTED, "prt_upload_scene: pick_random_light is not built together with SDF primitives or a debug view");
    }
    if (cfg.env_importance_sampling && (cfg.has_global_medium || n_sdf || cfg.view_option != PRT_VIEW_RESULTS || cfg.pick_random_light))
        return fail(c, PRT_ERR_UNSUPPORTED, "prt_upload_scene: env_importance_sampling is built for surfaces only (no global medium, SDF primitives, debug view, pick_random_light)");
    sc.env_is = cfg.env_importance_sampling ? 1u : 0u;
    if (cfg.light_count) {
        const uint32_t li = cfg.light_indices[0];
        if (li >= n_mesh) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_upload_scene: light index out of range");
        sc.light_mesh = li;
        if (li < n_sph) sc.light_sphere = li;
        else if (li >= n_sph + n_sdf) sc.light_quad = li - n_sph - n_sdf;
        // an SDF light cannot be sampled (kernels/geometry/geometry.cl:11-32 returns false): both stay unset
    }
    sc.active_mats = cfg.active_mats; sc.geom_flags = cfg.geom_flags;
    sc.dist_mask = 0;
    for (const DevMaterial& dm : mats)
        if ((dm.bits & 0xffffu) & (PRT_MAT_COAT | PRT_MAT_ROUGH_COND | PRT_MAT_ROUGH_DIEL)) sc.dist_mask |= (dm.bits >> 24) & 7u;
    sc.max_bounces = cfg.max_bounces; sc.max_diff_bounces = cfg.max_diff_bounces; sc.max_spec_bounces = cfg.max_spec_bounces;
    sc.max_trans_bounces = cfg.max_trans_bounces; sc.max_scattering_events = cfg.max_scattering_events;
    sc.has_medium = cfg.has_global_medium; sc.fog_abs_only = cfg.fog_abs_only; sc.alpha_testing = cfg.alpha_testing;
    sc.phase_function = cfg.phase_function; sc.fog_sigma_s = cfg.fog_sigma_s; sc.fog_sigma_t = cfg.fog_sigma_t; sc.phase_g = cfg.phase_g;
    sc.ntrans_mask = cfg.active_mats & (PRT_MAT_DIEL | PRT_MAT_ROUGH_DIEL);
    if (cfg.view_option != PRT_VIEW_RESULTS && cfg.view_option != PRT_VIEW_NORMAL && cfg.view_option != PRT_VIEW_BVH_HIT)
        return fail(c, PRT_ERR_UNSUPPORTED, "prt_upload_scene: view_option: only PRT_VIEW_RESULTS, PRT_VIEW_NORMAL and PRT_VIEW_BVH_HIT render "
                                            "anything in the reference (VIEW_STACK_INDEX does not compile, VIEW_ALBEDO / VIEW_SPECULAR never call radiance())");
    sc.view = cfg.view_option != PRT_VIEW_RESULTS;
    out.sc = sc;
    return PRT_OK;
}

}  // namespace prt
