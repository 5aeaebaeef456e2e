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
#include <algorithm>

#include "prt.h"
#include "pt_launch.h"
#include "pt_layout.h"
#include "pt_pack.h"

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
    void* d_spheres = nullptr; void* d_quads = nullptr; void* d_sdfs = nullptr; void* d_mats = nullptr; void* d_light_tab = nullptr; void* d_env = nullptr; void* d_env_rows = nullptr; void* d_env_cols = nullptr;
    bool have_scene = false, have_cam = false, have_size = false;
    bool state_undefined = false;   // a render call failed half-way (prt_render_spp's abort path): pixels may be ahead of the launch windows
                                    // (run-ahead leads in the state) -- the state is unusable until prt_reset / prt_write_state
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
    bool n_sub_forced = false;      // PRT_STREAMS given: sub_parts does not fall back to one launch for small frames
    hipStream_t sub_stream[MAX_SUB] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t sub_ev[MAX_SUB][2] = {};           // end of a sub-part's launch (two in flight per sub-part)
    hipEvent_t sub_ev0[MAX_SUB][2] = {};          // its start (prt_render_spp times every launch)
    hipEvent_t fork_ev = nullptr;
    unsigned long long* h_unfinished = nullptr;   // pinned, [MAX_SUB][2]: written by the last wave of a launch
    // frames a launch of render_kernel covers (PRT_FRAMES_PER_LAUNCH).  Lanes drift apart in frame number inside a launch
    // and the wave waits for its last lane at the end of each: long launches amortise that (cornell 1080p, MI355X:
    // 128 -> 7.85, 256 -> 8.18, 512 -> 8.33, 1024 -> 8.38, 2048 -> 8.08 G segments/s; 2 x 100 ms launches in flight at 512)
    // Through a tree beyond one XCD's L2 the lanes of a wave drift much further apart (deep walks), and every launch boundary makes a wave
    // wait for its slowest lane: 871 k triangles at 3840x2160, 512 spp: 512 -> 3.44, 1024 -> 3.57, 2048 -> 3.75, 4096 -> 3.84, 8192 -> 3.82 G
    // segments/s (a launch of 4096 frames of that scene runs 7.5 s; cornell: 256 -> 12.74, 512 -> 12.85, 1024 -> 12.66).
    // 0 = by tree size: 512, or 4096 beyond 64 k node pairs.
    unsigned frames_per_launch = 0;
    // walk phases end below this many walking lanes (prt_set_walk_min_lanes, PRT_WALK_MIN_LANES; 0 = chosen per launch, pt_kernels.hip
    // launch_variant_w).  Round-2 kernel at 4 waves: 1 -> 7.78, 4 -> 8.33, 6 -> 8.37, 8 -> 8.28, 12 -> 7.98 G segments/s; at 5 waves:
    // 4 -> 10.65, 6 -> 11.36, 8 -> 11.45, 12 -> 11.4
    uint32_t walk_min_lanes = 0;
    uint32_t shadow_min_lanes = 0;                 // the same for the shadow rays' walk phases (PRT_SHADOW_MIN_LANES; 0 = by tree size)
    // the pending triangle tests of a walk phase run once this many sixteenths of its walking lanes have one (PRT_TRI_Q; render_kernel)
    uint32_t tri_sixteenths = 4;
    uint32_t run_ahead = 1;                        // FrameArgs::run_ahead of prt_render_spp's launches (PRT_RUN_AHEAD=0: off)
    // prt_render_spp: pixels whose paths are longer than the frame's average owe every launch proportionally more frames (FrameArgs::pace_inv_ref;
    // PRT_PACE=0 / option "pace": off).  The frame's mean path length is taken once per render, after the first launch that retires.
    int pace = 1;
    // Expensive tiles first (prt_render_spp through trees of more than 64 k node pairs, FrameArgs::tile_order): the waves of launch 0 of
    // each sub-part leave what their tile cost (iterations),
    // the host sorts, and from launch 1 on -- and in later renders, until scene, camera or frame change -- the sub-part's workgroups take
    // their tiles in that order.  A launch ends with its last tile; started last, an expensive one keeps the launch open alone.
    int tile_sort = 1;                             // PRT_TILE_ORDER=0 / option "tile_order": off
    uint32_t* d_tile_order[MAX_SUB] = {nullptr, nullptr, nullptr, nullptr};
    uint32_t* d_tile_cost[MAX_SUB] = {nullptr, nullptr, nullptr, nullptr};
    bool have_order[MAX_SUB] = {false, false, false, false};
    std::vector<uint32_t> h_tile_cost, h_tile_order;
    bool launch_log = false;                       // PRT_LAUNCH_LOG=1: one line per retired launch on stderr
    bool test_drop_report = false;                 // option "test_drop_report" (tests only): the launches of prt_render_spp report into a spare word
    prt_stats stats{};
    std::string err;
    LaunchOpts lo{};                               // forced wave-count build / pixel mapping / generic material set (prt_set_option)
    RenderLaunch last{};                           // what the last launch ran
    RenderLaunch last_sub[MAX_SUB] = {};           // ... per sub-part (their grids differ by up to one tile: so can the pixel mapping)
    std::string variant;                           // ... as text (prt_kernel_variant)
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
        (e = hipHostMalloc(reinterpret_cast<void**>(&c->h_unfinished), (2 * prt_ctx::MAX_SUB + 1) * sizeof(unsigned long long))) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming)) != hipSuccess) {
        g_global_error = std::string("prt_create: ") + hipGetErrorString(e);
        prt_destroy(c);
        return PRT_ERR_HIP;
    }
    if (const char* ev = std::getenv("PRT_STREAMS")) { const int k = std::atoi(ev); if (k >= 1 && k <= prt_ctx::MAX_SUB) { c->n_sub = k; c->n_sub_forced = true; } }
    for (int j = 0; j < c->n_sub && c->n_sub > 1; ++j)
        if ((e = hipStreamCreateWithFlags(&c->sub_stream[j], hipStreamNonBlocking)) != hipSuccess ||
            (e = hipEventCreate(&c->sub_ev[j][0])) != hipSuccess || (e = hipEventCreate(&c->sub_ev0[j][0])) != hipSuccess ||
            (e = hipEventCreate(&c->sub_ev[j][1])) != hipSuccess || (e = hipEventCreate(&c->sub_ev0[j][1])) != hipSuccess) {
            g_global_error = std::string("prt_create: ") + hipGetErrorString(e);
            prt_destroy(c);
            return PRT_ERR_HIP;
        }
    c->stream = c->own_stream;
    if (const char* ev = std::getenv("PRT_FRAMES_PER_LAUNCH")) { const int k = std::atoi(ev); if (k >= 1) c->frames_per_launch = (unsigned)k; }
    if (const char* ev = std::getenv("PRT_RUN_AHEAD")) c->run_ahead = std::atoi(ev) != 0 ? 1u : 0u;
    if (const char* ev = std::getenv("PRT_PACE")) c->pace = std::atoi(ev) != 0 ? 1 : 0;
    if (const char* ev = std::getenv("PRT_WALK_MIN_LANES")) { const int k = std::atoi(ev); if (k >= 1 && k <= 64) c->walk_min_lanes = (uint32_t)k; }
    if (const char* ev = std::getenv("PRT_WAVES")) { const int k = std::atoi(ev); if (k == 5 || k == 6) c->lo.waves = k; }
    if (const char* ev = std::getenv("PRT_SCATTER")) { const int k = std::atoi(ev); if (k == 0 || k == 1) c->lo.scatter = k; }
    if (const char* ev = std::getenv("PRT_GENERIC")) c->lo.generic = std::atoi(ev) != 0 ? 1 : 0;
    if (const char* ev = std::getenv("PRT_ANY_DIST")) c->lo.any_dist = std::atoi(ev) != 0 ? 1 : 0;
    if (const char* ev = std::getenv("PRT_PIX_PER_WAVE")) { const int k = std::atoi(ev); if (k == 64 || k == 32 || k == 16) c->lo.pix_per_wave = k; }
    if (const char* ev = std::getenv("PRT_POOL")) c->lo.pool = std::atoi(ev) != 0 ? 1 : 0;
    if (const char* ev = std::getenv("PRT_TRI_Q")) { const int k = std::atoi(ev); if (k >= 0 && k <= 16) c->tri_sixteenths = (uint32_t)k; }
    if (const char* ev = std::getenv("PRT_TILE_ORDER")) c->tile_sort = std::atoi(ev) != 0 ? 1 : 0;
    if (const char* ev = std::getenv("PRT_LAUNCH_LOG")) c->launch_log = std::atoi(ev) != 0;
    if (const char* ev = std::getenv("PRT_SHADOW_MIN_LANES")) { const int k = std::atoi(ev); if (k >= 1 && k <= 64) c->shadow_min_lanes = (uint32_t)k; }
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
    for (int j = 0; j < prt_ctx::MAX_SUB; ++j) {
        p = c->d_tile_order[j]; free_dev(p); c->d_tile_order[j] = nullptr;
        p = c->d_tile_cost[j]; free_dev(p); c->d_tile_cost[j] = nullptr;
        c->have_order[j] = false;
    }
}
static void free_scene(prt_ctx* c) {
    free_dev(c->d_pairs); free_dev(c->d_tri_geom); free_dev(c->d_tri_nrm);
    free_dev(c->d_spheres); free_dev(c->d_quads); free_dev(c->d_sdfs); free_dev(c->d_mats); free_dev(c->d_light_tab);
}

extern "C" void prt_destroy(prt_ctx* c) {
#ifdef PT_PHASE_CLOCKS
    if (c) { (void)hipDeviceSynchronize(); prt::dump_phase_clocks(); }
#endif
#ifdef PT_POOL_STATS
    if (c) { (void)hipDeviceSynchronize(); prt::dump_pool_stats(); }
#endif
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
    free_scene(c);
    free_dev(c->d_env); free_dev(c->d_env_rows); free_dev(c->d_env_cols);
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

extern "C" int prt_upload_scene(prt_ctx* c, const prt_scene_desc* s) {
    CTX_CHECK(c);
    // everything that can be refused is refused before the old scene is touched (pt_pack.cpp: pure host code)
    PackedScene ps;
    std::string perr;
    int rc = pack_scene(c->cfg, s, ps, perr);
    if (rc) return fail(c, rc, perr);
    HIPCHK(c, hipSetDevice(c->device));
    (void)hipStreamSynchronize(c->stream);
    // from here on the old buffers are being replaced: the context has no scene until every upload has succeeded
    c->have_scene = false;
    const float* env = c->sc.env; const int env_w = c->sc.env_w, env_h = c->sc.env_h;    // the environment map survives scene uploads
    const float* env_rows = c->sc.env_cdf_rows; const float* env_cols = c->sc.env_cdf_cols;
    c->sc = DevScene{};
    c->sc.env = env; c->sc.env_w = env_w; c->sc.env_h = env_h; c->sc.env_cdf_rows = env_rows; c->sc.env_cdf_cols = env_cols;
    if ((rc = upload(c, c->d_pairs, ps.pairs)) || (rc = upload(c, c->d_tri_geom, ps.tg)) || (rc = upload(c, c->d_tri_nrm, ps.tn)) ||
        (rc = upload(c, c->d_spheres, ps.spheres)) || (rc = upload(c, c->d_quads, ps.quads)) || (rc = upload(c, c->d_sdfs, ps.sdfs)) ||
        (rc = upload(c, c->d_mats, ps.mats)) || (rc = upload(c, c->d_light_tab, ps.light_tab)))
        return rc;
    DevScene sc = ps.sc;
    sc.pairs = static_cast<const NodePair*>(c->d_pairs);
    sc.tri_geom = static_cast<const TriGeom*>(c->d_tri_geom);
    sc.tri_nrm = static_cast<const TriNrm*>(c->d_tri_nrm);
    sc.spheres = static_cast<const DevSphere*>(c->d_spheres);
    sc.quads = static_cast<const DevQuad*>(c->d_quads);
    sc.sdfs = static_cast<const DevSdf*>(c->d_sdfs);
    sc.mats = static_cast<const DevMaterial*>(c->d_mats);
    sc.light_tab = static_cast<const uint32_t*>(c->d_light_tab);
    sc.env = env; sc.env_w = env_w; sc.env_h = env_h; sc.env_cdf_rows = env_rows; sc.env_cdf_cols = env_cols;
    c->sc = sc;
    if (!c->sc.env) {
        const float black[3] = {0.f, 0.f, 0.f};
        rc = prt_upload_envmap(c, black, 1, 1);
        if (rc) return rc;
    }
    c->have_scene = true;
    for (int j = 0; j < prt_ctx::MAX_SUB; ++j) c->have_order[j] = false;
    return PRT_OK;
}

extern "C" int prt_set_camera(prt_ctx* c, const prt_camera* cam) {
    CTX_CHECK(c);
    if (!cam) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_set_camera: null camera");
    make_dev_camera(*cam, c->cam);
    c->have_cam = true;
    for (int j = 0; j < prt_ctx::MAX_SUB; ++j) c->have_order[j] = false;      // (tile costs are the view's)
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
    free_dev(c->d_env_rows); free_dev(c->d_env_cols);
    c->sc.env_cdf_rows = c->sc.env_cdf_cols = nullptr;
    if (c->cfg.env_importance_sampling) {                       // the map's sampling density (pt_kernels.hip build_env_cdf)
        std::vector<float> rows, cols;
        build_env_cdf(rgb, w, h, rows, cols);
        int rc = upload(c, c->d_env_rows, rows);
        if (!rc) rc = upload(c, c->d_env_cols, cols);
        if (rc) return rc;
        c->sc.env_cdf_rows = static_cast<const float*>(c->d_env_rows);
        c->sc.env_cdf_cols = static_cast<const float*>(c->d_env_cols);
    }
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
    // the context has no frame until every plane of the new one exists: a failed (re)allocation must leave it NOT READY,
    // never "ready" with null planes
    c->have_size = false;
    free_frame(c);
    c->npix = 0;
    const size_t npix = (size_t)width * (size_t)rows;
    hipError_t e = hipSuccess;
    void** planes[6] = {reinterpret_cast<void**>(&c->S.q0), reinterpret_cast<void**>(&c->S.q1), reinterpret_cast<void**>(&c->S.q2),
                        reinterpret_cast<void**>(&c->S.q3), reinterpret_cast<void**>(&c->S.q4), reinterpret_cast<void**>(&c->fb)};
    for (int k = 0; k < 6 && e == hipSuccess; ++k) e = hipMalloc(planes[k], npix * 16);
    const size_t n_tiles_alloc = (size_t)render_tile_count(width, rows);
    for (int j = 0; j < c->n_sub && c->n_sub > 1 && e == hipSuccess; ++j) {
        e = hipMalloc(reinterpret_cast<void**>(&c->d_tile_order[j]), n_tiles_alloc * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->d_tile_cost[j]), n_tiles_alloc * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemset(c->d_tile_cost[j], 0, n_tiles_alloc * sizeof(uint32_t));    // (a forced 5-wave build reports no costs: all equal then)
    }
    if (e != hipSuccess) {
        free_frame(c);
        (void)hipGetLastError();
        c->err = std::string("prt frame allocation (") + std::to_string(width) + " x " + std::to_string(rows) + "): " + hipGetErrorString(e);
        return PRT_ERR_HIP;
    }
    c->width = width; c->full_height = full_height; c->row0 = row0; c->rows = rows;
    c->npix = npix;
    c->have_size = true;
    return prt_reset(c);
}

extern "C" int prt_resize(prt_ctx* c, int width, int height) { return prt_set_tile(c, width, height, 0, height); }

extern "C" int prt_reset(prt_ctx* c) {
    CTX_CHECK(c);
    if (!c->have_size) return fail(c, PRT_ERR_NOT_READY, "prt_reset: no frame size set");
    HIPCHK(c, hipSetDevice(c->device));
    c->state_undefined = false;
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

static int ready(prt_ctx* c, const char* who) {
    if (c->state_undefined) return fail(c, PRT_ERR_NOT_READY, std::string(who) + ": an earlier render call failed half-way; prt_reset or prt_write_state first");
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
    fa.seed_frames = n; fa.run_ahead = 0; fa.pace_inv_ref = 0.0f;
    fa.unfinished = count ? c->d_counters : nullptr;
    fa.unfinished_host = nullptr;
    fa.tile_first = 0; fa.tile_stride = 1; fa.scatter = 0; fa.sub_shift = 0;
    fa.tile_order = nullptr; fa.tile_cost = nullptr;
    // 0 = by launch (pt_kernels.hip launch_variant_w: the scattered-pixel launches and the medium variants 6, the others 8; shadow
    // phases in lock step in small trees, bounded like the closest-hit phases in big ones)
    fa.walk_min_lanes = c->walk_min_lanes;
    fa.shadow_min_lanes = c->shadow_min_lanes;
    fa.tri_sixteenths = c->tri_sixteenths;
    return fa;
}

// 1 / (mean path length of the frame so far) = paths started / segments executed over all pixels (one small reduction over the state planes on
// `stream`, which must be idle); 0 if nothing has been rendered yet
static float inverse_mean_path_length(prt_ctx* c, uint32_t spp, hipStream_t stream) {
    unsigned long long h[3] = {0, 0, 0};
    if (hipMemsetAsync(c->d_counters + 1, 0, 3 * sizeof(unsigned long long), stream) != hipSuccess) return 0.0f;
    launch_count(c->S, c->npix, spp, c->d_counters + 1, stream);
    if (hipMemcpyAsync(h, c->d_counters + 1, sizeof(h), hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) return 0.0f;
    // (the reference IS the mean: 0.9 / 1.15 / 1.3 / 1.5 / 2 x the mean were all slower)
    return (h[0] && h[1]) ? (float)((double)h[0] / (double)h[1]) : 0.0f;
}

// sub-parts the megakernel renders this frame part in (1 = one launch covers every tile)
// Sets of tiles rendered side by side on internal streams: 2 (PRT_STREAMS) -- but ONE launch for a SHORT render of a frame whose waves all
// fit on the chip at once (up to 5 120 tiles = one round at 5 waves per SIMD; `frames` = the frames the render is expected to take, at
// most two launches' worth): two half-size launches share the chip unevenly there and the render lasts as long as the slower one (512x512 x
// 64 spp: 19 ... 22 ms beside 21 ... 25 ms; one launch +5 ... 8 %, also at 128 spp; 640x640 and up: two are better or equal).  A LONG render
// of such a frame -- one rank's share of a 1080p frame at 1 024 spp, 16 launches -- wants the two sets: each covers the ends of the
// other's launches (0.375 s against 0.426 s).
static int sub_parts(const prt_ctx* c, unsigned long long frames) {
    if (!(c->n_sub > 1 && c->sub_stream[0])) return 1;
    const unsigned step = c->frames_per_launch ? c->frames_per_launch : (c->sc.n_pairs > 65536u ? 4096u : 512u);
    if (!c->n_sub_forced && render_tile_count(c->width, c->rows) <= 5120u && frames <= 2ull * step) return 1;
    return c->n_sub;
}
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

extern "C" int prt_render_frames(prt_ctx* c, uint32_t first_frame, uint32_t n_frames, const int32_t* seed_pairs) {
    CTX_CHECK(c);
    int rc = ready(c, "prt_render_frames");
    if (rc) return rc;
    if (first_frame == 0 || (n_frames && !seed_pairs)) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_render_frames: frames start at 1 and need seed pairs");
    HIPCHK(c, hipSetDevice(c->device));
    c->stats.launches = 0; c->stats.frames = 0; c->stats.kernel_ms = 0.0; c->stats.kernel_sum_ms = 0.0; c->stats.concurrent = 1;
    if (!n_frames) return PRT_OK;
    if ((rc = ensure_seeds(c, seed_pairs, n_frames))) return rc;
    const unsigned step = c->frames_per_launch ? c->frames_per_launch : (c->sc.n_pairs > 65536u ? 4096u : 512u);
    const int K = sub_parts(c, n_frames);
    c->stats.concurrent = (uint32_t)K;
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    if (K > 1 && (rc = fork_streams(c, K))) return rc;
    for (uint32_t f = 0; f < n_frames; f += step) {
        const uint32_t n = (n_frames - f < step) ? n_frames - f : step;
        for (int j = 0; j < K; ++j) {
            FrameArgs fa = frame_args(c, first_frame + f, n, c->d_seeds + 2 * (size_t)f, 0, false);
            fa.tile_first = (uint32_t)j; fa.tile_stride = (uint32_t)K;
            c->last = launch_render(c->sc, c->cam, c->S, fa, c->fb, K > 1 ? c->sub_stream[j] : c->stream, c->lo);
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
    const unsigned step = c->frames_per_launch ? c->frames_per_launch : (c->sc.n_pairs > 65536u ? 4096u : 512u);
    const int K = sub_parts(c, 8ull * spp);            // (a path takes 4.4 ... 7.2 segments on the BASELINE scenes)
    c->stats.concurrent = (uint32_t)K;
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    uint32_t f = 0;
    unsigned long long unfinished = 1;
    float pace_inv_ref = 0.0f;                     // 1 / (the frame's mean path length): known once the first launch has retired
    // (through a big tree the launches are 4 096 frames and a render is a handful of them: 871 k triangles, 3840x2160 x 2 048 spp 4.35 without, 4.31 with)
    const bool pacing = c->pace && c->run_ahead && c->sc.n_pairs <= 65536u;
    if (K == 1) {
        while (f < max_frames && unfinished) {
            const uint32_t n = (max_frames - f < step) ? max_frames - f : step;
            HIPCHK(c, hipMemsetAsync(c->d_counters, 0, sizeof(unsigned long long), c->stream));
            FrameArgs fa = frame_args(c, 1 + f, n, c->d_seeds + 2 * (size_t)f, spp, true);
            fa.seed_frames = max_frames - f; fa.run_ahead = c->run_ahead;
            fa.pace_inv_ref = pacing ? pace_inv_ref : 0.0f;
            c->last = launch_render(c->sc, c->cam, c->S, fa, c->fb, c->stream, c->lo);
            ++c->stats.launches;
            f += n;
            HIPCHK(c, hipMemcpyAsync(&unfinished, c->d_counters, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (pacing && pace_inv_ref == 0.0f && unfinished) pace_inv_ref = inverse_mean_path_length(c, spp, c->stream);
        }
    } else {
        // Every sub-part advances on its own stream until its own pixels are frozen.  The last wave of a launch writes
        // the number of pixels still running to pinned host memory and clears the device counters, so there is no copy
        // or fill kernel between launches (with the other stream's kernel filling the chip those waited ~1.7 ms for a
        // wave slot) and the HIP events around a launch time the kernel alone.  A launch queued behind one that
        // reported 0 (PRT_QUEUE_DEPTH=2) finds every pixel frozen and returns at once.
        // the sub-part counters are {unfinished, waves done}; a launch that ended abnormally in an earlier call would
        // have left them non-zero and every later ticket wrong: start from zero, ordered before the fork
        HIPCHK(c, hipMemsetAsync(c->d_counters + 4, 0, 2 * prt_ctx::MAX_SUB * sizeof(unsigned long long), c->stream));
        if ((rc = fork_streams(c, K))) return rc;
        // any failure below leaves through abort_streams: kernels may still be running on the internal streams, the
        // caller's stream must not run ahead of them and the context must stay usable
        auto abort_streams = [&](int code, const std::string& msg) {
            for (int j = 0; j < K; ++j) (void)hipStreamSynchronize(c->sub_stream[j]);
            (void)hipMemsetAsync(c->d_counters + 4, 0, 2 * prt_ctx::MAX_SUB * sizeof(unsigned long long), c->stream);
            (void)hipStreamSynchronize(c->stream);
            (void)hipGetLastError();
            c->state_undefined = true;
            return fail(c, code, msg);
        };
#define SUBCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return abort_streams(PRT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)
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
                    fa.seed_frames = max_frames - fj[j]; fa.run_ahead = c->run_ahead;
                    fa.pace_inv_ref = pacing ? pace_inv_ref : 0.0f;
                    fa.unfinished = c->d_counters + 4 + 2 * j;
                    fa.unfinished_host = c->h_unfinished + 2 * j + slot;
                    if (c->test_drop_report) fa.unfinished_host = c->h_unfinished + 2 * prt_ctx::MAX_SUB;     // (tests: the report goes astray)
                    fa.tile_first = (uint32_t)j; fa.tile_stride = (uint32_t)K;
                    if (c->tile_sort && c->d_tile_cost[j] && c->sc.n_pairs > 65536u) {
                        fa.tile_order = c->have_order[j] ? c->d_tile_order[j] : nullptr;
                        fa.tile_cost = (!c->have_order[j] && issued[j] == 0u) ? c->d_tile_cost[j] : nullptr;
                    }
                    c->h_unfinished[2 * j + slot] = ~0ull;
                    // (a tile's cost is the maximum over its waves -- atomicMax in the kernel: the measuring launch starts from zero)
                    if (fa.tile_cost) SUBCHK(hipMemsetAsync(c->d_tile_cost[j], 0, (size_t)render_tile_count(c->width, c->rows) * sizeof(uint32_t), c->sub_stream[j]));
                    SUBCHK(hipEventRecord(c->sub_ev0[j][slot], c->sub_stream[j]));
                    c->last = c->last_sub[j] = launch_render(c->sc, c->cam, c->S, fa, c->fb, c->sub_stream[j], c->lo);
                    SUBCHK(hipEventRecord(c->sub_ev[j][slot], c->sub_stream[j]));
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
                    SUBCHK(q);
                    float ms = 0.f;
                    if (hipEventElapsedTime(&ms, c->sub_ev0[j][slot], c->sub_ev[j][slot]) == hipSuccess) c->stats.kernel_sum_ms += ms;
                    const unsigned long long left = __atomic_load_n(c->h_unfinished + 2 * j + slot, __ATOMIC_ACQUIRE);
                    if (c->launch_log) std::fprintf(stderr, "prt launch: part %d #%u %.3f ms, %llu pixels unfinished\n", j, retired[j], ms, left);
                    // the host armed the slot with ~0 before the launch; the last wave of the launch overwrites it (render_kernel).  A launch whose
                    // report never arrived must not be read as 1.8e19 unfinished pixels
                    if (left == ~0ull) return abort_streams(PRT_ERR_HIP, "prt_render_spp: a launch ended without reporting its unfinished pixels");
                    if (c->tile_sort && c->d_tile_cost[j] && c->sc.n_pairs > 65536u && !c->have_order[j] && retired[j] == 0u && !c->last_sub[j].scatter && n_tiles > (unsigned)j) {
                        // launch 0 of this sub-part is over and its stream idle: its tiles by run time, longest first, for every launch from here on
                        const unsigned grid = (n_tiles - (unsigned)j + (unsigned)K - 1u) / (unsigned)K;
                        c->h_tile_cost.resize(grid); c->h_tile_order.resize(grid);
                        SUBCHK(hipMemcpyAsync(c->h_tile_cost.data(), c->d_tile_cost[j], grid * sizeof(uint32_t), hipMemcpyDeviceToHost, c->sub_stream[j]));
                        SUBCHK(hipStreamSynchronize(c->sub_stream[j]));
                        for (unsigned k = 0; k < grid; ++k) c->h_tile_order[k] = k;
                        const uint32_t* cost = c->h_tile_cost.data();
                        std::stable_sort(c->h_tile_order.begin(), c->h_tile_order.end(), [cost](uint32_t a, uint32_t b) { return cost[a] > cost[b]; });
                        SUBCHK(hipMemcpyAsync(c->d_tile_order[j], c->h_tile_order.data(), grid * sizeof(uint32_t), hipMemcpyHostToDevice, c->sub_stream[j]));
                        SUBCHK(hipStreamSynchronize(c->sub_stream[j]));      // (the host vector is reused by the other sub-part)
                        c->have_order[j] = true;
                    }
                    // the frame's mean path length, once per render (this sub-part's stream is idle: its launch has just retired)
                    if (pacing && pace_inv_ref == 0.0f && left) pace_inv_ref = inverse_mean_path_length(c, spp, c->sub_stream[j]);
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
#undef SUBCHK
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

extern "C" int prt_set_walk_min_lanes(prt_ctx* c, uint32_t lanes) {
    CTX_CHECK(c);
    if (lanes < 1 || lanes > 64) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_set_walk_min_lanes: 1..64");
    c->walk_min_lanes = lanes;
    c->shadow_min_lanes = lanes;                    // an explicit setting applies to both kinds of walk phase
    return PRT_OK;
}

// Schedule and build choices of a context.  None of them changes a bit of any result (the tests render the goldens under each);
// they exist for tests, experiments and tuning.  The same names in upper case with the prefix PRT_ are read from the environment
// by prt_create.
extern "C" int prt_set_option(prt_ctx* c, const char* name, int value) {
    CTX_CHECK(c);
    if (!name) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_set_option: null name");
    const std::string n(name);
    auto bad = [&]() { return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_set_option: value out of range for " + n); };
    if (n == "waves") { if (value != 0 && value != 5 && value != 6) return bad(); c->lo.waves = value; }
    else if (n == "scatter") { if (value < -1 || value > 1) return bad(); c->lo.scatter = value; }
    else if (n == "generic") { if (value < 0 || value > 1) return bad(); c->lo.generic = value; }
    else if (n == "any_dist") { if (value < 0 || value > 1) return bad(); c->lo.any_dist = value; }
    else if (n == "pix_per_wave") { if (value != 0 && value != 64 && value != 32 && value != 16) return bad(); c->lo.pix_per_wave = value; }
    else if (n == "pool") { if (value < 0 || value > 1) return bad(); c->lo.pool = value; }
    else if (n == "walk_min_lanes") { if (value < 0 || value > 64) return bad(); c->walk_min_lanes = (uint32_t)value; }
    else if (n == "shadow_min_lanes") { if (value < 0 || value > 64) return bad(); c->shadow_min_lanes = (uint32_t)value; }
    else if (n == "tri_q") { if (value < 0 || value > 16) return bad(); c->tri_sixteenths = (uint32_t)value; }
    else if (n == "frames_per_launch") { if (value < 0) return bad(); c->frames_per_launch = (unsigned)value; }
    else if (n == "run_ahead") { if (value < 0 || value > 1) return bad(); c->run_ahead = (uint32_t)value; }
    else if (n == "pace") { if (value < 0 || value > 1) return bad(); c->pace = value; }
    else if (n == "tile_order") {
        if (value < 0 || value > 1) return bad();
        if (value != c->tile_sort) for (int j = 0; j < prt_ctx::MAX_SUB; ++j) c->have_order[j] = false;      // (setting it again keeps a measured order)
        c->tile_sort = value;
    }
    else if (n == "test_drop_report") { if (value < 0 || value > 1) return bad(); c->test_drop_report = value != 0; }
    else return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_set_option: unknown option " + n);
    return PRT_OK;
}

extern "C" const char* prt_kernel_variant(prt_ctx* c) {
    if (!c) return "";
    c->variant = std::string(c->last.name) + (c->last.waves ? " waves=" + std::to_string(c->last.waves) + (c->last.scatter ? " pixels=scattered" : (c->last.ordered ? " pixels=tiles, expensive first" : " pixels=tiles")) + (c->last.pix_per_wave != 64 ? ", " + std::to_string(c->last.pix_per_wave) + " per wave" : "") + (c->last.pool ? ", pool" : "") : "");
    return c->variant.c_str();
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
    c->state_undefined = false;
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

extern "C" int prt_selftest_fn(prt_ctx* c, int fn, const float* params, const float* in, float* out, int n) {
    CTX_CHECK(c);
    if (!params || !in || !out || n <= 0 || fn < 1 || fn > 11) return fail(c, PRT_ERR_INVALID_ARGUMENT, "prt_selftest_fn: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    float *dp = nullptr, *di = nullptr, *dout = nullptr;
    const size_t bytes = (size_t)n * 32 * sizeof(float);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&dp), 80 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&di), bytes);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&dout), bytes);
    if (e == hipSuccess) e = hipMemcpy(dp, params, 80 * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(di, in, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) { launch_selftest_fn(fn, dp, di, dout, n, c->stream); e = hipStreamSynchronize(c->stream); }
    if (e == hipSuccess) e = hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost);
    if (dp) (void)hipFree(dp);
    if (di) (void)hipFree(di);
    if (dout) (void)hipFree(dout);
    HIPCHK(c, e);
    return PRT_OK;
}
