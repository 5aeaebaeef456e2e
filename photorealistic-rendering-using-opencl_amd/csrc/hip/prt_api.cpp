// prt_api.cpp -- implementation of the C ABI in include/prt.h on top of HIP.
// Replaces the OpenCL host sequence of the reference's src/main.cpp (see prt.h for the
// call-by-call mapping).  No CPU fallback: every entry point needs a HIP device.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "prt.h"
#include "pt_launch.h"
#include "pt_layout.h"

using namespace prt;

namespace {
thread_local std::string g_global_error;
}

struct prt_ctx {
    int device = 0;
    prt_config cfg{};
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timing_pending = false;
    // scene
    DevScene sc{};
    void* d_pairs = nullptr; void* d_tri_geom = nullptr; void* d_tri_nrm = nullptr;
    void* d_spheres = nullptr; void* d_quads = nullptr; void* d_sdfs = nullptr; void* d_mats = nullptr; void* d_env = nullptr;
    bool have_scene = false, have_cam = false, have_size = false;
    DevCamera cam{};
    // frame
    int width = 0, full_height = 0, row0 = 0, rows = 0;
    int block_rows = 1, n_parts = 1, part = 0;
    size_t npix = 0;
    DevState S{};
    float4* fb = nullptr;
    int32_t* d_seeds = nullptr; size_t seeds_cap = 0;
    unsigned long long* d_counters = nullptr;     // [0] unfinished, [1..3] count_kernel, [4 + 2j, 5 + 2j] {unfinished, waves done} of sub-part j
    // The megakernel renders the frame as n_sub interleaved sets of tiles, each on its own stream: the sets are
    // independent (pixels are), so while one set's launch drains on its slowest tiles the other fills the CUs.
    static constexpr int MAX_SUB = 4;
    int n_sub = 2;
    hipStream_t sub_stream[MAX_SUB] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t sub_ev[MAX_SUB][2] = {};           // end of a sub-part's launch (two in flight per sub-part)
    hipEvent_t sub_ev0[MAX_SUB][2] = {};          // its start (prt_render_spp times every launch)
    hipEvent_t fork_ev = nullptr;
    unsigned long long* h_unfinished = nullptr;   // pinned, [MAX_SUB][2]: written by the last wave of a launch
    // wavefront pipeline (optional)
    int pipeline = 0;                              // 0 = megakernel, 1 = wavefront
    DevWave wv{};
    bool wv_allocated = false;
    prt_stats stats{};
    std::string err;
    const char* variant = "";
};

#define CTX_CHECK(ctx) do { if (!(ctx)) return PRT_ERR_INVALID_ARGUMENT; } while (0)
#define HIPCHK(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
        (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_); return PRT_ERR_HIP; } } while (0)

static int fail(prt_ctx* ctx, int code, const std::string& msg) { ctx->err = msg; return code; }
static FrameArgs frame_args(prt_ctx* c, uint32_t first_frame, uint32_t n, const int32_t* d_seeds, uint32_t spp, bool count);

static void free_dev(void*& p) { if (p) { (void)hipFree(p); p = nullptr; } }

extern "C" const char* prt_last_global_error(void) { return g_global_error.c_str(); }
extern "C" const char* prt_last_error(prt_ctx* ctx) { return ctx ? ctx->err.c_str() : g_global_error.c_str(); }

extern "C" int prt_create(int device, const prt_config* cfg, prt_ctx** out) {
    if (!cfg || !out) { g_global_error = "prt_create: null argument"; return PRT_ERR_INVALID_ARGUMENT; }
    if (cfg->abi_version != PRT_ABI_VERSION) { g_global_error = "prt_create: abi_version mismatch"; return PRT_ERR_INVALID_ARGUMENT; }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_global_error = std::string("prt_create: no HIP device (") + (e != hipSuccess ? hipGetErrorString(e) : "count 0") + "); libprt has no CPU fallback";
        return PRT_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) { g_global_error = "prt_create: device ordinal out of range"; return PRT_ERR_INVALID_ARGUMENT; }
    if (cfg->light_count > PRT_MAX_LIGHTS) { g_global_error = "prt_create: too many lights"; return PRT_ERR_INVALID_ARGUMENT; }
    if (cfg->marching_steps < 0 || cfg->marching_steps > 65536 || cfg->shadow_marching_steps < 0 || cfg->shadow_marching_steps > 65536) {
        g_global_error = "prt_create: MARCHING_STEPS / SHADOW_MARCHING_STEPS outside 0..65536 (they bound a loop every lane runs)";
        return PRT_ERR_INVALID_ARGUMENT;
    }
    if (cfg->geom_flags & PRT_GEOM_BOX) {
        g_global_error = "prt_create: box primitives never render in the reference (geometry/box.cl is not included, SURVEY.md s9-Q10)";
        return PRT_ERR_UNSUPPORTED;
    }
    prt_ctx* c = new prt_ctx();
    c->device = device;
    c->cfg = *cfg;
    if ((e = hipSetDevice(device)) != hipSuccess || (e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreate(&c->ev0)) != hipSuccess || (e = hipEventCreate(&c->ev1)) != hipSuccess ||
        (e = hipMalloc(reinterpret_cast<void**>(&c->d_counters), (4 + 2 * prt_ctx::MAX_SUB) * sizeof(unsigned long long))) != hipSuccess ||
        (e = hipMemset(c->d_counters, 0, (4 + 2 * prt_ctx::MAX_SUB) * sizeof(unsigned long long))) != hipSuccess ||
        (e = hipHostMalloc(reinterpret_cast<void**>(&c->h_unfinished), 2 * prt_ctx::MAX_SUB * sizeof(unsigned long long))) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming)) != hipSuccess) {
        g_global_error = std::string("prt_create: ") + hipGetErrorString(e);
        prt_destroy(c);
        return PRT_ERR_HIP;
    }
    if (const char* ev = std::getenv("PRT_STREAMS")) { const int k = std::atoi(ev); if (k >= 1 && k <= prt_ctx::MAX_SUB) c->n_sub = k; }
    for (int j = 0; j < c->n_sub && c->n_sub > 1; ++j)
        if ((e = hipStreamCreateWithFlags(&c->sub_stream[j], hipStreamNonBlocking)) != hipSuccess ||
            (e = hipEventCreate(&c->sub_ev[j][0])) != hipSuccess || (e = hipEventCreate(&c->sub_ev0[j][0])) != hipSuccess ||
            (e = hipEventCreate(&c->sub_ev[j][1])) != hipSuccess || (e = hipEventCreate(&c->sub_ev0[j][1])) != hipSuccess) {
            g_global_error = std::string("prt_create: ") + hipGetErrorString(e);
            prt_destroy(c);
            return PRT_ERR_HIP;
        }
    c->stream = c->own_stream;
    if (const char* e = std::getenv("PRT_PIPELINE")) c->pipeline = (std::strcmp(e, "wavefront") == 0 || std::strcmp(e, "1") == 0) ? 1 : 0;
    *out = c;
    return PRT_OK;
}

static void free_frame(prt_ctx* c) {
    void* p;
    p = c->S.q0; free_dev(p); c->S.q0 = nullptr;
    p = c->S.q1; free_dev(p); c->S.q1 = nullptr;
    p = c->S.q2; free_dev(p); c->S.q2 = nullptr;
    p = c->S.q3; free_dev(p); c->S.q3 = nullptr;
    p = c->S.q4; free_dev(p); c->S.q4 = nullptr;
    p = c->fb; free_dev(p); c->fb = nullptr;
}
static void free_wave(prt_ctx* c) {
    void* p;
    p = c->wv.hc0; free_dev(p); p = c->wv.hc1; free_dev(p); p = c->wv.prog; free_dev(p); p = c->wv.ctx; free_dev(p);
    p = c->wv.ray_o; free_dev(p); p = c->wv.ray_d; free_dev(p); p = c->wv.res0; free_dev(p); p = c->wv.res1; free_dev(p);
    p = c->wv.qcount; free_dev(p);
    c->wv = DevWave{};
    c->wv_allocated = false;
}
static int alloc_wave(prt_ctx* c) {
    if (c->wv_allocated) return PRT_OK;
    const size_t n = c->npix;
    DevWave& w = c->wv;
    w.npix = n;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&w.hc0), n * 16));
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&w.hc1), n * 16));
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&w.prog), n * 16));
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&w.ctx), n * 16 * PRT_WF_CTX_PLANES));
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&w.ray_o), n * 16));
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&w.ray_d), n * 16));
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&w.res0), n * 16));
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&w.res1), n * 4));
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&w.qcount), 2 * sizeof(unsigned)));
    c->wv_allocated = true;
    return PRT_OK;
}

static void free_scene(prt_ctx* c) {
    free_dev(c->d_pairs); free_dev(c->d_tri_geom); free_dev(c->d_tri_nrm);
    free_dev(c->d_spheres); free_dev(c->d_quads); free_dev(c->d_sdfs); free_dev(c->d_mats);
}

extern "C" void prt_destroy(prt_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (int j = 0; j < prt_ctx::MAX_SUB; ++j) {
        if (c->sub_stream[j]) { (void)hipStreamSynchronize(c->sub_stream[j]); (void)hipStreamDestroy(c->sub_stream[j]); }
        for (int k = 0; k < 2; ++k) {
            if (c->sub_ev[j][k]) (void)hipEventDestroy(c->sub_ev[j][k]);
            if (c->sub_ev0[j][k]) (void)hipEventDestroy(c->sub_ev0[j][k]);
        }
    }
    if (c->fork_ev) (void)hipEventDestroy(c->fork_ev);
    if (c->h_unfinished) (void)hipHostFree(c->h_unfinished);
    free_frame(c);
    free_wave(c);
    free_scene(c);
    free_dev(c->d_env);
    void* p = c->d_seeds; free_dev(p);
    p = c->d_counters; free_dev(p);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

template <typename T>
static int upload(prt_ctx* c, void*& dst, const std::vector<T>& src) {
    free_dev(dst);
    size_t bytes = (src.empty() ? 1 : src.size()) * sizeof(T);
    HIPCHK(c, hipMalloc(&dst, bytes));
    if (!src.empty()) HIPCHK(c, hipMemcpy(dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return PRT_OK;
}

static DevMaterial pack_material(const prt_material& m) {
    DevMaterial d;
    std::memset(&d, 0, sizeof(d));
    for (int i = 0; i < 3; ++i) { d.color[i] = m.color[i]; d.eta[i] = m.eta[i]; d.k[i] = m.k[i]; }
    d.roughness = m.roughness;
    d.bits = (uint32_t)m.t | ((uint32_t)m.lobes << 16) | ((uint32_t)m.dist << 24);
    return d;
}

extern "C" int prt_upload_scene(prt_ctx* c, const prt_scene_desc* s) {
    CTX_CHECK(c);
    if (!s) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_upload_scene: null scene");
    HIPCHK(c, hipSetDevice(c->device));
    const uint32_t n_sph = s->object_count[0], n_sdf = s->object_count[1], n_box = s->object_count[2], n_quad = s->object_count[3];
    const uint32_t n_mesh = s->object_count[7];
    if (n_box) return fail(c, PRT_ERR_UNSUPPORTED, "prt_upload_scene: box primitives never render in the reference (box.cl is dead code)");
    if (n_sph + n_sdf + n_quad != n_mesh) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_upload_scene: object_count does not add up");
    if (n_sdf && !(c->cfg.geom_flags & PRT_GEOM_SDF)) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_upload_scene: SDF meshes but the config has no H_SDF");
    if (n_mesh && !s->meshes) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_upload_scene: meshes is null");
    const uint32_t T = s->triangle_count, N = s->bvh_node_count;
    if (T && (!s->vertices || !s->normals || !s->primitive_indices || !s->bvh_nodes || !N))
        return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_upload_scene: triangle buffers incomplete");

    // ---- primitives + materials
    std::vector<DevSphere> spheres(n_sph);
    std::vector<DevQuad> quads(n_quad);
    std::vector<DevSdf> sdfs(n_sdf);
    std::vector<DevMaterial> mats(n_mesh + 2);
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
            for (int k = 0; k < 3; ++k) { d.base[k] = m.joker[k]; d.edge0[k] = m.joker[3 + k]; d.edge1[k] = m.joker[6 + k]; d.normal[k] = m.joker[9 + k]; }
            d.area = m.joker[12];
            // kernels/geometry/quad.cl:16 anchor = base - (edge0 + edge1) * 0.5f ; :26-27 dot(edge, edge)
            for (int k = 0; k < 3; ++k) d.anchor[k] = d.base[k] - (d.edge0[k] + d.edge1[k]) * 0.5f;
            d.e0e0 = d.edge0[0] * d.edge0[0] + d.edge0[1] * d.edge0[1] + d.edge0[2] * d.edge0[2];
            d.e1e1 = d.edge1[0] * d.edge1[0] + d.edge1[1] * d.edge1[1] + d.edge1[2] * d.edge1[2];
            // pt_device.h out_of_unit_range: half an ulp of 1.0 scaled by the divisor, NaN = "divide instead"
            auto half_ulp = [](float c) { return (c >= 9.094947017729282e-13f && c <= 1099511627776.0f) ? c * 5.9604644775390625e-08f : std::nanf(""); };
            d.u0 = half_ulp(d.e0e0); d.u1 = half_ulp(d.e1e1);
        }
    }
    if (s->obj_material) mats[n_mesh + 1] = pack_material(*s->obj_material);

    // ---- BVH: reference layout -> NodePair records (inner nodes only), DFS order
    std::vector<NodePair> pairs;
    DevScene sc{};
    sc.root_is_leaf = 1;
    sc.stack_levels = 1;
    if (T) {
        const prt_bvh_node* nodes = s->bvh_nodes;
        auto leaf_ok = [&](const prt_bvh_node& nd) { return (uint64_t)nd.first_child_or_primitive + nd.primitive_count <= T; };
        if (nodes[0].is_leaf) {
            if (!leaf_ok(nodes[0])) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_upload_scene: root leaf range out of bounds");
            sc.root_leaf_first = nodes[0].first_child_or_primitive;
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
            order.reserve(N / 2 + 1);
            if (!dfs) {
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
                        p.meta[2 * ch] = cn.first_child_or_primitive;
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
    // ---- triangles in leaf-slot order
    std::vector<TriGeom> tg(T);
    std::vector<TriNrm> tn(T);
    for (uint32_t i = 0; i < T; ++i) {
        const uint32_t fv = (uint32_t)s->primitive_indices[i] * 3u;        // triangle.cl:7 (uint arithmetic)
        if ((uint64_t)fv + 2 >= (uint64_t)T * 3) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_upload_scene: primitive index out of range");
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

    (void)hipStreamSynchronize(c->stream);
    int rc;
    if ((rc = upload(c, c->d_pairs, pairs)) || (rc = upload(c, c->d_tri_geom, tg)) || (rc = upload(c, c->d_tri_nrm, tn)) ||
        (rc = upload(c, c->d_spheres, spheres)) || (rc = upload(c, c->d_quads, quads)) || (rc = upload(c, c->d_sdfs, sdfs)) ||
        (rc = upload(c, c->d_mats, mats)))
        return rc;

    sc.pairs = static_cast<const NodePair*>(c->d_pairs);
    sc.n_pairs = (uint32_t)pairs.size();
    sc.tri_geom = static_cast<const TriGeom*>(c->d_tri_geom);
    sc.tri_nrm = static_cast<const TriNrm*>(c->d_tri_nrm);
    sc.spheres = static_cast<const DevSphere*>(c->d_spheres);
    sc.quads = static_cast<const DevQuad*>(c->d_quads);
    sc.sdfs = static_cast<const DevSdf*>(c->d_sdfs);
    sc.mats = static_cast<const DevMaterial*>(c->d_mats);
    sc.n_spheres = n_sph; sc.n_quads = n_quad; sc.quad_mesh_base = n_sph + n_sdf; sc.n_meshes = n_mesh; sc.n_sdfs = n_sdf;
    sc.marching_steps = c->cfg.marching_steps; sc.shadow_marching_steps = c->cfg.shadow_marching_steps;
    sc.light_sphere = sc.light_quad = 0xFFFFFFFFu; sc.light_mesh = 0;
    const prt_config& cfg = c->cfg;
    if (cfg.light_count) {
        const uint32_t li = cfg.light_indices[0];
        if (li >= n_mesh) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_upload_scene: light index out of range");
        sc.light_mesh = li;
        if (li < n_sph) sc.light_sphere = li;
        else if (li >= n_sph + n_sdf) sc.light_quad = li - n_sph - n_sdf;
        // an SDF light cannot be sampled (kernels/geometry/geometry.cl:11-32 returns false): both stay unset
    }
    sc.active_mats = cfg.active_mats; sc.geom_flags = cfg.geom_flags;
    sc.max_bounces = cfg.max_bounces; sc.max_diff_bounces = cfg.max_diff_bounces; sc.max_spec_bounces = cfg.max_spec_bounces;
    sc.max_trans_bounces = cfg.max_trans_bounces; sc.max_scattering_events = cfg.max_scattering_events;
    sc.has_medium = cfg.has_global_medium; sc.fog_abs_only = cfg.fog_abs_only; sc.alpha_testing = cfg.alpha_testing;
    sc.phase_function = cfg.phase_function; sc.fog_sigma_s = cfg.fog_sigma_s; sc.fog_sigma_t = cfg.fog_sigma_t; sc.phase_g = cfg.phase_g;
    sc.ntrans_mask = cfg.active_mats & (PRT_MAT_DIEL | PRT_MAT_ROUGH_DIEL);
    // keep the environment map across scene uploads
    sc.env = c->sc.env; sc.env_w = c->sc.env_w; sc.env_h = c->sc.env_h;
    c->sc = sc;
    c->have_scene = true;
    if (!c->sc.env) {
        const float black[3] = {0.f, 0.f, 0.f};
        rc = prt_upload_envmap(c, black, 1, 1);
        if (rc) return rc;
    }
    return PRT_OK;
}

extern "C" int prt_set_camera(prt_ctx* c, const prt_camera* cam) {
    CTX_CHECK(c);
    if (!cam) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_set_camera: null camera");
    make_dev_camera(*cam, c->cam);
    c->have_cam = true;
    return PRT_OK;
}

extern "C" int prt_upload_envmap(prt_ctx* c, const float* rgb, int w, int h) {
    CTX_CHECK(c);
    if (!rgb || w <= 0 || h <= 0) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_upload_envmap: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    free_dev(c->d_env);
    const size_t bytes = (size_t)w * h * 3 * sizeof(float);
    HIPCHK(c, hipMalloc(&c->d_env, bytes));
    HIPCHK(c, hipMemcpy(c->d_env, rgb, bytes, hipMemcpyHostToDevice));
    c->sc.env = static_cast<const float*>(c->d_env);
    c->sc.env_w = w; c->sc.env_h = h;
    return PRT_OK;
}

static int alloc_frame(prt_ctx* c, int width, int full_height, int row0, int rows);

extern "C" int prt_set_tile(prt_ctx* c, int width, int full_height, int row0, int rows) {
    CTX_CHECK(c);
    if (width <= 0 || full_height <= 0 || rows <= 0 || row0 < 0 || row0 + rows > full_height)
        return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_set_tile: bad tile");
    c->block_rows = 1; c->n_parts = 1; c->part = 0;
    return alloc_frame(c, width, full_height, row0, rows);
}

extern "C" int prt_set_row_blocks(prt_ctx* c, int width, int full_height, int block_rows, int n_parts, int part) {
    CTX_CHECK(c);
    if (width <= 0 || full_height <= 0 || block_rows <= 0 || n_parts <= 0 || part < 0 || part >= n_parts)
        return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_set_row_blocks: bad arguments");
    int rows = 0;
    for (int r = 0; r < full_height; ++r) rows += ((r / block_rows) % n_parts == part);
    if (rows == 0) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_set_row_blocks: this part owns no rows");
    c->block_rows = block_rows; c->n_parts = n_parts; c->part = part;
    return alloc_frame(c, width, full_height, 0, rows);
}

static int alloc_frame(prt_ctx* c, int width, int full_height, int row0, int rows) {
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    free_frame(c);
    free_wave(c);
    c->width = width; c->full_height = full_height; c->row0 = row0; c->rows = rows;
    c->npix = (size_t)width * (size_t)rows;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->S.q0), c->npix * 16));
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->S.q1), c->npix * 16));
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->S.q2), c->npix * 16));
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->S.q3), c->npix * 16));
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->S.q4), c->npix * 16));
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->fb), c->npix * 16));
    c->have_size = true;
    return prt_reset(c);
}

extern "C" int prt_resize(prt_ctx* c, int width, int height) { return prt_set_tile(c, width, height, 0, height); }

extern "C" int prt_reset(prt_ctx* c) {
    CTX_CHECK(c);
    if (!c->have_size) return fail(c, PRT_ERR_NOT_READY, "prt_reset: no frame size set");
    HIPCHK(c, hipSetDevice(c->device));
    // enqueueFillBuffer(cl_flattenI, 0, ...), src/main.cpp:288
    HIPCHK(c, hipMemsetAsync(c->S.q0, 0, c->npix * 16, c->stream));
    HIPCHK(c, hipMemsetAsync(c->S.q1, 0, c->npix * 16, c->stream));
    HIPCHK(c, hipMemsetAsync(c->S.q2, 0, c->npix * 16, c->stream));
    HIPCHK(c, hipMemsetAsync(c->S.q3, 0, c->npix * 16, c->stream));
    HIPCHK(c, hipMemsetAsync(c->S.q4, 0, c->npix * 16, c->stream));
    HIPCHK(c, hipMemsetAsync(c->fb, 0, c->npix * 16, c->stream));
    return PRT_OK;
}

static int ensure_seeds(prt_ctx* c, const int32_t* seed_pairs, size_t n_frames) {
    if (c->seeds_cap < n_frames) {
        void* p = c->d_seeds; free_dev(p); c->d_seeds = nullptr; c->seeds_cap = 0;
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->d_seeds), n_frames * 2 * sizeof(int32_t)));
        c->seeds_cap = n_frames;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_seeds, seed_pairs, n_frames * 2 * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));   // caller may free seed_pairs on return
    return PRT_OK;
}

static unsigned frames_per_launch() {
    static unsigned v = 0;
    if (!v) {
        const char* e = std::getenv("PRT_FRAMES_PER_LAUNCH");
        v = e ? (unsigned)std::atoi(e) : 128u;   // measured on MI355X: 32 -> 3.82, 64 -> 3.95, 128 -> 4.01 G segments/s
        if (v == 0) v = 128u;
    }
    return v;
}

static int ready(prt_ctx* c, const char* who) {
    if (!c->have_scene) return fail(c, PRT_ERR_NOT_READY, std::string(who) + ": no scene uploaded");
    if (!c->have_cam) return fail(c, PRT_ERR_NOT_READY, std::string(who) + ": no camera set");
    if (!c->have_size) return fail(c, PRT_ERR_NOT_READY, std::string(who) + ": no frame size set");
    return PRT_OK;
}

static FrameArgs frame_args(prt_ctx* c, uint32_t first_frame, uint32_t n, const int32_t* d_seeds, uint32_t spp, bool count) {
    FrameArgs fa;
    fa.width = c->width; fa.full_height = c->full_height; fa.row0 = c->row0; fa.rows = c->rows;
    fa.block_rows = c->block_rows; fa.n_parts = c->n_parts; fa.part = c->part;
    fa.first_frame = first_frame; fa.n_frames = n; fa.seed_pairs = d_seeds; fa.spp_limit = spp;
    fa.unfinished = count ? c->d_counters : nullptr;
    fa.unfinished_host = nullptr;
    fa.tile_first = 0; fa.tile_stride = 1;
    return fa;
}

// sub-parts the megakernel renders this frame part in (1 = one launch covers every tile)
static int sub_parts(const prt_ctx* c) { return (c->n_sub > 1 && c->sub_stream[0]) ? c->n_sub : 1; }
// the internal streams start behind everything already queued on the caller's stream ...
static int fork_streams(prt_ctx* c, int K) {
    HIPCHK(c, hipEventRecord(c->fork_ev, c->stream));
    for (int j = 0; j < K; ++j) HIPCHK(c, hipStreamWaitEvent(c->sub_stream[j], c->fork_ev, 0));
    return PRT_OK;
}
// ... and the caller's stream continues behind them
static int join_streams(prt_ctx* c, int K) {
    for (int j = 0; j < K; ++j) {
        HIPCHK(c, hipEventRecord(c->sub_ev[j][0], c->sub_stream[j]));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->sub_ev[j][0], 0));
    }
    return PRT_OK;
}

// Wavefront pipeline: passes of (shade, traverse) until every pixel has done its n_frames segments (or froze).
static int render_wavefront(prt_ctx* c, uint32_t first_frame, uint32_t n_frames, uint32_t spp, uint32_t* frames_used) {
    int rc = alloc_wave(c);
    if (rc) return rc;
    const DevWave& w = c->wv;
    // every pixel starts the call at a segment boundary with an empty hit cache
    HIPCHK(c, hipMemsetAsync(w.prog, 0, c->npix * 16, c->stream));
    HIPCHK(c, hipMemsetAsync(w.hc1, 0, c->npix * 16, c->stream));
    HIPCHK(c, hipMemsetAsync(w.qcount, 0, 2 * sizeof(unsigned), c->stream));
    const unsigned trav_blocks = 2048;                      // 256 CUs x 8 workgroups of 4 waves = 8 waves/SIMD, grid-stride
    const unsigned batch = 16;
    const unsigned long long max_passes = 4ull * n_frames + 8;
    unsigned long long unfinished = 1, pass = 0;
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    while (unfinished && pass < max_passes) {
        for (unsigned b = 0; b < batch; ++b, ++pass) {
            const bool last = (b + 1 == batch);
            if (last) HIPCHK(c, hipMemsetAsync(c->d_counters, 0, sizeof(unsigned long long), c->stream));
            FrameArgs fa = frame_args(c, first_frame, n_frames, c->d_seeds, spp, last);
            launch_wf_pass(c->sc, c->cam, c->S, w, fa, c->fb, (unsigned)pass, trav_blocks, c->stream);
            c->stats.launches += 2;
        }
        HIPCHK(c, hipMemcpyAsync(&unfinished, c->d_counters, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    c->timing_pending = true;
    c->variant = "wavefront";
    if (frames_used) *frames_used = (uint32_t)pass;
    if (unfinished) return fail(c, PRT_ERR_NOT_READY, "wavefront: pass limit reached before every pixel finished");
    return PRT_OK;
}

extern "C" int prt_render_frames(prt_ctx* c, uint32_t first_frame, uint32_t n_frames, const int32_t* seed_pairs) {
    CTX_CHECK(c);
    int rc = ready(c, "prt_render_frames");
    if (rc) return rc;
    if (first_frame == 0 || (n_frames && !seed_pairs)) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_render_frames: frames start at 1 and need seed pairs");
    HIPCHK(c, hipSetDevice(c->device));
    c->stats.launches = 0; c->stats.frames = 0; c->stats.kernel_ms = 0.0; c->stats.kernel_sum_ms = 0.0; c->stats.concurrent = 1;
    if (!n_frames) return PRT_OK;
    if ((rc = ensure_seeds(c, seed_pairs, n_frames))) return rc;
    if (c->pipeline == 1) {
        rc = render_wavefront(c, first_frame, n_frames, 0, nullptr);
        c->stats.frames = n_frames;
        return rc;
    }
    const unsigned step = frames_per_launch();
    const int K = sub_parts(c);
    c->stats.concurrent = (uint32_t)K;
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    if (K > 1 && (rc = fork_streams(c, K))) return rc;
    for (uint32_t f = 0; f < n_frames; f += step) {
        const uint32_t n = (n_frames - f < step) ? n_frames - f : step;
        for (int j = 0; j < K; ++j) {
            FrameArgs fa = frame_args(c, first_frame + f, n, c->d_seeds + 2 * (size_t)f, 0, false);
            fa.tile_first = (uint32_t)j; fa.tile_stride = (uint32_t)K;
            c->variant = launch_render(c->sc, c->cam, c->S, fa, c->fb, K > 1 ? c->sub_stream[j] : c->stream);
            ++c->stats.launches;
        }
    }
    HIPCHK(c, hipGetLastError());
    if (K > 1 && (rc = join_streams(c, K))) return rc;
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    c->timing_pending = true;
    c->stats.frames = n_frames;
    return PRT_OK;
}

extern "C" int prt_render_spp(prt_ctx* c, uint32_t spp, uint32_t max_frames, const int32_t* seed_pairs, uint32_t* frames_used) {
    CTX_CHECK(c);
    int rc = ready(c, "prt_render_spp");
    if (rc) return rc;
    if (!spp || !max_frames || !seed_pairs) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_render_spp: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    c->stats.launches = 0; c->stats.frames = 0; c->stats.kernel_ms = 0.0; c->stats.kernel_sum_ms = 0.0; c->stats.concurrent = 1;
    if ((rc = ensure_seeds(c, seed_pairs, max_frames))) return rc;
    if (c->pipeline == 1) {
        rc = render_wavefront(c, 1, max_frames, spp, frames_used);
        c->stats.frames = max_frames;
        return rc;
    }
    const unsigned step = frames_per_launch();
    const int K = sub_parts(c);
    c->stats.concurrent = (uint32_t)K;
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    uint32_t f = 0;
    unsigned long long unfinished = 1;
    if (K == 1) {
        while (f < max_frames && unfinished) {
            const uint32_t n = (max_frames - f < step) ? max_frames - f : step;
            HIPCHK(c, hipMemsetAsync(c->d_counters, 0, sizeof(unsigned long long), c->stream));
            c->variant = launch_render(c->sc, c->cam, c->S, frame_args(c, 1 + f, n, c->d_seeds + 2 * (size_t)f, spp, true), c->fb, c->stream);
            ++c->stats.launches;
            f += n;
            HIPCHK(c, hipMemcpyAsync(&unfinished, c->d_counters, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
    } else {
        // Every sub-part advances on its own stream until its own pixels are frozen.  The last wave of a launch writes
        // the number of pixels still running to pinned host memory and clears the device counters, so there is no copy
        // or fill kernel between launches (with the other stream's kernel filling the chip those waited ~1.7 ms for a
        // wave slot) and the HIP events around a launch time the kernel alone.  A launch queued behind one that
        // reported 0 (PRT_QUEUE_DEPTH=2) finds every pixel frozen and returns at once.
        if ((rc = fork_streams(c, K))) return rc;
        uint32_t fj[prt_ctx::MAX_SUB] = {0, 0, 0, 0};            // frames queued so far
        uint32_t f_end[prt_ctx::MAX_SUB][2] = {};                 // ... up to the end of the launch in each slot
        unsigned issued[prt_ctx::MAX_SUB] = {0, 0, 0, 0}, retired[prt_ctx::MAX_SUB] = {0, 0, 0, 0};
        uint32_t f_done[prt_ctx::MAX_SUB] = {0, 0, 0, 0};        // frames after which the sub-part reported 0
        bool stop[prt_ctx::MAX_SUB] = {false, false, false, false};
        bool exhausted = false;
        // launches queued per sub-part: 1 (the other sub-part's kernel covers the host round trip; 2 measured the same)
        static const unsigned depth = [] { const char* e = std::getenv("PRT_QUEUE_DEPTH"); return (e && std::atoi(e) == 2) ? 2u : 1u; }();
        const unsigned n_tiles = render_tile_count(c->width, c->rows);
        for (int j = 0; j < K; ++j) stop[j] = (unsigned)j >= n_tiles;      // a sub-part without tiles has nothing to do
        for (;;) {
            bool progressed = false, busy = false;
            for (int j = 0; j < K; ++j) {
                while (!stop[j] && issued[j] - retired[j] < depth && fj[j] < max_frames) {
                    const unsigned slot = issued[j] & 1u;
                    const uint32_t n = (max_frames - fj[j] < step) ? max_frames - fj[j] : step;
                    FrameArgs fa = frame_args(c, 1 + fj[j], n, c->d_seeds + 2 * (size_t)fj[j], spp, true);
                    fa.unfinished = c->d_counters + 4 + 2 * j;
                    fa.unfinished_host = c->h_unfinished + 2 * j + slot;
                    fa.tile_first = (uint32_t)j; fa.tile_stride = (uint32_t)K;
                    c->h_unfinished[2 * j + slot] = ~0ull;
                    HIPCHK(c, hipEventRecord(c->sub_ev0[j][slot], c->sub_stream[j]));
                    c->variant = launch_render(c->sc, c->cam, c->S, fa, c->fb, c->sub_stream[j]);
                    HIPCHK(c, hipEventRecord(c->sub_ev[j][slot], c->sub_stream[j]));
                    ++c->stats.launches;
                    fj[j] += n;
                    f_end[j][slot] = fj[j];
                    ++issued[j];
                    progressed = true;
                }
                if (issued[j] != retired[j]) {
                    busy = true;
                    const unsigned slot = retired[j] & 1u;
                    const hipError_t q = hipEventQuery(c->sub_ev[j][slot]);
                    if (q == hipErrorNotReady) continue;
                    HIPCHK(c, q);
                    float ms = 0.f;
                    if (hipEventElapsedTime(&ms, c->sub_ev0[j][slot], c->sub_ev[j][slot]) == hipSuccess) c->stats.kernel_sum_ms += ms;
                    const unsigned long long left = __atomic_load_n(c->h_unfinished + 2 * j + slot, __ATOMIC_ACQUIRE);
                    if (left == ~0ull) return fail(c, PRT_ERR_HIP, "prt_render_spp: a launch ended without reporting its unfinished pixels");
                    ++retired[j];
                    progressed = true;
                    if (!stop[j]) {
                        if (left == 0) { stop[j] = true; f_done[j] = f_end[j][slot]; }
                        else if (f_end[j][slot] >= max_frames) { stop[j] = true; f_done[j] = max_frames; exhausted = true; }
                    }
                }
            }
            if (!busy && !progressed) break;
            if (!progressed) std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
        for (int j = 0; j < K; ++j) fj[j] = f_done[j];
        for (int j = 0; j < K; ++j) f = fj[j] > f ? fj[j] : f;
        unfinished = exhausted ? 1 : 0;
        if ((rc = join_streams(c, K))) return rc;
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    c->timing_pending = true;
    c->stats.frames = f;
    if (frames_used) *frames_used = f;
    if (unfinished) return fail(c, PRT_ERR_NOT_READY, "prt_render_spp: max_frames reached before every pixel finished");
    return PRT_OK;
}

extern "C" int prt_set_pipeline(prt_ctx* c, int pipeline) {
    CTX_CHECK(c);
    if (pipeline != 0 && pipeline != 1) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_set_pipeline: 0 = megakernel, 1 = wavefront");
    c->pipeline = pipeline;
    return PRT_OK;
}

extern "C" int prt_synchronize(prt_ctx* c) {
    CTX_CHECK(c);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->timing_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->ev0, c->ev1) == hipSuccess) c->stats.kernel_ms = ms;
        if (c->stats.kernel_sum_ms == 0.0) c->stats.kernel_sum_ms = c->stats.kernel_ms;      // launches were not timed one by one
        c->timing_pending = false;
    }
    return PRT_OK;
}

extern "C" int prt_read_framebuffer(prt_ctx* c, float* rgba) {
    CTX_CHECK(c);
    if (!rgba || !c->have_size) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_read_framebuffer: bad arguments");
    int rc = prt_synchronize(c);
    if (rc) return rc;
    HIPCHK(c, hipMemcpy(rgba, c->fb, c->npix * 16, hipMemcpyDeviceToHost));
    return PRT_OK;
}

extern "C" int prt_tonemap_rgba8(prt_ctx* c, uint8_t* rgba) {
    CTX_CHECK(c);
    if (!rgba || !c->have_size) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_tonemap_rgba8: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    unsigned char* d = nullptr;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&d), c->npix * 4));
    launch_tonemap(c->fb, d, frame_args(c, 1, 0, nullptr, 0, false), c->stream);
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(rgba, d, c->npix * 4, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    HIPCHK(c, e);
    return PRT_OK;
}

extern "C" int prt_copy_framebuffer_to_device(prt_ctx* c, void* device_rgba) {
    CTX_CHECK(c);
    if (!device_rgba || !c->have_size) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_copy_framebuffer_to_device: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(device_rgba, c->fb, c->npix * 16, hipMemcpyDeviceToDevice, c->stream));
    // on a caller's stream (prt_set_stream) the copy is ordered like any other work of the caller; the context's own
    // stream is private and non-blocking, nothing of the caller's could wait for it: finish the copy before returning
    if (c->stream == c->own_stream) HIPCHK(c, hipStreamSynchronize(c->stream));
    return PRT_OK;
}

extern "C" int prt_read_state(prt_ctx* c, prt_path_state* state) {
    CTX_CHECK(c);
    if (!state || !c->have_size) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_read_state: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    prt_path_state* d = nullptr;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&d), c->npix * sizeof(prt_path_state)));
    launch_state_to_rtd(c->S, d, c->npix, c->stream);
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(state, d, c->npix * sizeof(prt_path_state), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    HIPCHK(c, e);
    return PRT_OK;
}

extern "C" int prt_write_state(prt_ctx* c, const prt_path_state* state) {
    CTX_CHECK(c);
    if (!state || !c->have_size) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_write_state: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    prt_path_state* d = nullptr;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&d), c->npix * sizeof(prt_path_state)));
    hipError_t e = hipMemcpy(d, state, c->npix * sizeof(prt_path_state), hipMemcpyHostToDevice);
    if (e == hipSuccess) { launch_rtd_to_state(d, c->S, c->fb, c->npix, c->stream); e = hipStreamSynchronize(c->stream); }
    (void)hipFree(d);
    HIPCHK(c, e);
    return PRT_OK;
}

extern "C" int prt_set_stream(prt_ctx* c, void* hip_stream) {
    CTX_CHECK(c);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
    return PRT_OK;
}

extern "C" int prt_get_stats(prt_ctx* c, prt_stats* out) {
    CTX_CHECK(c);
    if (!out) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_get_stats: null");
    int rc = prt_synchronize(c);
    if (rc) return rc;
    *out = c->stats;
    return PRT_OK;
}

extern "C" int prt_query_counts(prt_ctx* c, uint32_t spp, prt_stats* out) {
    CTX_CHECK(c);
    if (!out || !c->have_size) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_query_counts: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemsetAsync(c->d_counters + 1, 0, 3 * sizeof(unsigned long long), c->stream));
    launch_count(c->S, c->npix, spp, c->d_counters + 1, c->stream);
    unsigned long long h[3] = {0, 0, 0};
    HIPCHK(c, hipMemcpyAsync(h, c->d_counters + 1, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    int rc = prt_synchronize(c);
    if (rc) return rc;
    c->stats.samples = h[0]; c->stats.segments = h[1]; c->stats.finished_pixels = h[2];
    *out = c->stats;
    return PRT_OK;
}

extern "C" int prt_selftest_math(prt_ctx* c, int fn, const float* a, const float* b, float* out, int n) {
    CTX_CHECK(c);
    if (!a || !b || !out || n <= 0) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_selftest_math: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    float *da = nullptr, *db = nullptr, *dout = nullptr;
    const size_t bytes = (size_t)n * sizeof(float);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&da), bytes);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&db), bytes);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&dout), bytes);
    if (e == hipSuccess) e = hipMemcpy(da, a, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(db, b, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) { launch_selftest_math(fn, da, db, dout, n, c->stream); e = hipStreamSynchronize(c->stream); }
    if (e == hipSuccess) e = hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost);
    if (da) (void)hipFree(da);
    if (db) (void)hipFree(db);
    if (dout) (void)hipFree(dout);
    HIPCHK(c, e);
    return PRT_OK;
}
