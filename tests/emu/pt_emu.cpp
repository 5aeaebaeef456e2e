// tests/emu/pt_emu.cpp -- TEST INFRASTRUCTURE, never part of libprt.
//
// The device functions of the product (csrc/hip/pt_device.h: the lane machine, BVH steps, BSDFs, ...) compiled for the
// HOST (-DPT_EMU turns __device__ into __host__ __device__; hipcc --cuda-host-only) and driven pixel by pixel through
// the passes of render_kernel (pt_kernels.hip), on the CPU (-m "not gpu"): the lane machine, the parking of a pixel's
// context between passes and the hand-over of deep rays to the batch walk == the reference goldens and
// oracle/pt_oracle.c, bit for bit, without a GPU.
// What this does NOT check is the GPU code generation: that is what the -m gpu parity tests are for.  On the host,
// hw_recip is the IEEE divide it is exhaustively equal to on the device (prt_selftest_math fn 17).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "pt_device.h"
#include "pt_pack.h"

using namespace prt;
using namespace prt::dev;

namespace {

// one pixel through render_kernel's passes (pt_kernels.hip), exactly in the kernel's order: unpark into a POISONED Lane
// (so a field lane_pack forgets shows), answer the walk, lane_back + shadow walk + lane_finish, lane_front, the step at
// the root, park; a ray that goes deeper is walked the way the kernel's batch walk does it (from {node, far child},
// result squeezed through {t, u, v, slot | found}).
template <unsigned MATS, bool MEDIUM>
void run_pixel(const DevScene& sc, const DevCamera& cam, const FrameArgs& fa, int lx, int ly, prt_path_state* state, float* out_rgba,
               std::vector<unsigned>& stack_mem) {
    const size_t id = (size_t)ly * (size_t)fa.width + (size_t)lx;
    const int gx = lx;
    const int gy = fa.row0 + (ly / fa.block_rows * fa.n_parts + fa.part) * fa.block_rows + ly % fa.block_rows;
    stack_mem.assign((size_t)sc.stack_levels, 0u);
    TravStack stk;
    stk.lds = stack_mem.data(); stk.stride = 1;
    Parked<MEDIUM> P;
    {
        Lane l;
        lane_init(l);
        const prt_path_state& r = state[id];
        l.origin = F3(r.origin[0], r.origin[1], r.origin[2]); l.t = r.time;
        l.dir = F3(r.dir[0], r.dir[1], r.dir[2]); l.time = r.dist;
        l.mask = F3(r.mask[0], r.mask[1], r.mask[2]); l.total = r.total;
        for (int k = 0; k < 4; ++k) l.acc[k] = r.acc[k];
        l.samples = r.samples;
        l.diff = r.diff; l.spec = r.spec; l.trans = r.trans; l.scatters = r.scatters;
        l.wasSpecular = r.was_specular != 0; l.reset = r.reset != 0;
        lane_pack<MEDIUM>(l, false, false, P);
    }
    uint32_t result[4] = {0, 0, 0, 0};
    unsigned frames_done = 0;
    for (;;) {
        Lane L;
        std::memset(&L, 0xA5, sizeof(L));                          // poison: everything a pass needs must come out of P
        bool off, pooled;
        lane_unpack<MEDIUM>(P, L, off, pooled);
        if (pooled) {
            L.w.t = prt_u2f(result[0]); L.w.th.u = prt_u2f(result[1]); L.w.th.v = prt_u2f(result[2]); L.w.th.w = 1.0f - L.w.th.u - L.w.th.v;
            L.w.th.slot = result[3] & 0x7fffffffu; L.w.found = (result[3] >> 31) != 0; L.w.done = true;
            pooled = false;
        }
        if (L.stage == ST_WALKC && L.w.done) lane_closest_done<MATS, MEDIUM>(sc, L);
        for (;;) {
            if (L.stage == ST_BACK) lane_back<MATS, MEDIUM>(sc, L);
            if (L.stage == ST_WALKS) {
                const Ray wr = lane_shadow_ray<MEDIUM>(L);
                const RayPre p = ray_pre(wr);
                walk_begin(sc, true, wr, wr.t, p, L.w, stk);
                while (!L.w.done) walk_step(sc, true, wr, p, L.w, stk);
                L.occluded = L.w.found; L.stage = ST_FINISH;
            }
            if (L.stage == ST_FINISH) lane_finish<MEDIUM>(sc, L);
            if (lane_runnable(fa, L)) lane_front<MATS, MEDIUM>(sc, cam, fa, L, gx, gy);
            if (L.stage != ST_BACK) break;
        }
        if (L.stage == ST_WALKC && L.fresh) {
            const Ray wr = lane_closest_ray<MEDIUM>(L);
            const RayPre p = ray_pre(wr);
            walk_begin(sc, false, wr, PT_INF, p, L.w, stk);
            L.fresh = false;
            if (!L.w.done) {                                       // the batch walk of the kernel
                const unsigned node = L.w.node, far = L.w.sp ? stk.lds[0] : 0xFFFFFFFFu;
                WalkState w;
                w.found = false; w.done = false; w.t = PT_INF;
                w.th.u = w.th.v = w.th.w = 0.0f; w.th.slot = 0;
                w.node = node; w.sp = 0;
                if (far != 0xFFFFFFFFu) { stk.lds[0] = far; w.sp = 1; }
                while (!w.done) walk_step(sc, false, wr, p, w, stk);
                result[0] = prt_f2u(w.t); result[1] = prt_f2u(w.th.u); result[2] = prt_f2u(w.th.v); result[3] = w.th.slot | (w.found ? 0x80000000u : 0u);
                pooled = true;
            }
        }
        frames_done = L.f;
        const bool live = L.stage != ST_READY || lane_runnable(fa, L);
        lane_pack<MEDIUM>(L, false, pooled, P);
        if (!live) break;
    }
    Lane l;
    bool off, pooled;
    lane_unpack<MEDIUM>(P, l, off, pooled);
    if (!frames_done) return;
    prt_path_state& r = state[id];
    r.origin[0] = l.origin.x; r.origin[1] = l.origin.y; r.origin[2] = l.origin.z; r.time = l.t;
    r.dir[0] = l.dir.x; r.dir[1] = l.dir.y; r.dir[2] = l.dir.z; r.dist = l.time;
    r.mask[0] = l.mask.x; r.mask[1] = l.mask.y; r.mask[2] = l.mask.z; r.total = l.total;
    for (int k = 0; k < 4; ++k) r.acc[k] = l.acc[k];
    r.samples = l.samples;
    r.diff = (uint16_t)l.diff; r.spec = (uint16_t)l.spec; r.trans = (uint16_t)l.trans; r.scatters = (uint16_t)l.scatters;
    r.was_specular = l.wasSpecular ? 1 : 0; r.reset = l.reset ? 1 : 0;
    const float ns = (float)l.samples;
    float* px = out_rgba + 4 * id;
    for (int k = 0; k < 4; ++k) px[k] = l.acc[k] / ns;
}

}  // namespace

// same arguments as oracle/pt_oracle.h's pto_job.  Returns 0, or the prt error code of pack_scene.
extern "C" int emu_render(const prt_config* cfg, const prt_scene_desc* desc, const prt_camera* camera, const float* env_rgb, int env_w, int env_h,
                          int width, int full_height, int row0, int rows, int block_rows, int n_parts, int part,
                          uint32_t first_frame, uint32_t n_frames, const int32_t* seed_pairs, prt_path_state* state, float* out_rgba,
                          uint32_t spp_limit, char* err, int err_len) {
    PackedScene ps;
    std::string perr;
    const int rc = pack_scene(*cfg, desc, ps, perr);
    if (rc) {
        if (err && err_len > 0) { std::strncpy(err, perr.c_str(), (size_t)err_len - 1); err[err_len - 1] = 0; }
        return rc;
    }
    DevScene sc = ps.sc;
    sc.pairs = ps.pairs.data(); sc.tri_geom = ps.tg.data(); sc.tri_nrm = ps.tn.data();
    sc.spheres = ps.spheres.data(); sc.quads = ps.quads.data(); sc.sdfs = ps.sdfs.data(); sc.mats = ps.mats.data();
    static const float black[3] = {0.f, 0.f, 0.f};
    sc.env = env_rgb ? env_rgb : black; sc.env_w = env_rgb ? env_w : 1; sc.env_h = env_rgb ? env_h : 1;
    DevCamera cam{};
    camera_basis(*camera, cam);
    FrameArgs fa{};
    fa.width = width; fa.full_height = full_height; fa.row0 = row0; fa.rows = rows;
    fa.block_rows = block_rows > 0 ? block_rows : 1; fa.n_parts = n_parts > 0 ? n_parts : 1; fa.part = part;
    fa.first_frame = first_frame; fa.n_frames = n_frames; fa.seed_pairs = seed_pairs; fa.spp_limit = spp_limit;
    fa.unfinished = nullptr; fa.unfinished_host = nullptr; fa.tile_first = 0; fa.tile_stride = 1;
    constexpr unsigned LD = PRT_MAT_LIGHT | PRT_MAT_DIFF;
    std::vector<unsigned> stack_mem;
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < width; ++x) {
            // the variant launch_render (pt_kernels.hip) picks
            if (sc.n_sdfs) {
                if (!sc.has_medium) run_pixel<PT_MATS_SDF, false>(sc, cam, fa, x, y, state, out_rgba, stack_mem);
                else run_pixel<PT_MATS_SDF, true>(sc, cam, fa, x, y, state, out_rgba, stack_mem);
            } else if (!sc.has_medium) {
                if (sc.active_mats == LD) run_pixel<LD, false>(sc, cam, fa, x, y, state, out_rgba, stack_mem);
                else run_pixel<0u, false>(sc, cam, fa, x, y, state, out_rgba, stack_mem);
            } else {
                if (sc.active_mats == LD) run_pixel<LD, true>(sc, cam, fa, x, y, state, out_rgba, stack_mem);
                else run_pixel<0u, true>(sc, cam, fa, x, y, state, out_rgba, stack_mem);
            }
        }
    return 0;
}
