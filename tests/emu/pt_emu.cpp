// tests/emu/pt_emu.cpp -- TEST INFRASTRUCTURE, never part of libprt.
//
// The device functions of the product (csrc/hip/pt_device.h: the lane machine, BVH steps, BSDFs, ...) compiled for the
// HOST (-DPT_EMU turns __device__ into __host__ __device__; hipcc --cuda-host-only) and driven by a wave emulator that
// mirrors render_kernel's schedule (pt_kernels.hip) over 64 Lane records.  Two uses, both on the CPU (-m "not gpu"):
//   * the refactored per-lane phases == oracle/pt_oracle.c, bit for bit, without a GPU;
//   * schedule independence: with `sched_seed` != 0 the length of every walk phase is drawn at random, which must not
//     change a bit of any pixel (lanes drift in frame number; pixels are independent).
// What this does NOT check is the GPU code generation: that is what the -m gpu parity tests are for.  On the host,
// hw_recip is the IEEE divide it is exhaustively equal to on the device (prt_selftest_math fn 17).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "pt_device.h"
#include "pt_pack.h"
#include "pt_selftest.h"

using namespace prt;
using namespace prt::dev;

namespace {

struct XorShift {
    uint32_t s;
    uint32_t next() { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }
};

template <unsigned MATS, bool MEDIUM>
void run_tile(const DevScene& sc, const DevCamera& cam, const FrameArgs& fa, int tile_x, int tile_y, prt_path_state* state, float* out_rgba,
              uint32_t sched_seed, std::vector<unsigned>& stack_mem, uint32_t* ahead) {
    Lane L[64];
    bool in_frame[64];
    int gxs[64] = {0}, gys[64] = {0};
    size_t ids[64];
    for (int lane = 0; lane < 64; ++lane) {
        const int lx = tile_x * 8 + (lane & 7), ly = tile_y * 8 + (lane >> 3);
        in_frame[lane] = lx < fa.width && ly < fa.rows;
        if (!in_frame[lane]) continue;
        ids[lane] = (size_t)ly * (size_t)fa.width + (size_t)lx;
        gxs[lane] = lx;
        gys[lane] = fa.row0 + (ly / fa.block_rows * fa.n_parts + fa.part) * fa.block_rows + ly % fa.block_rows;
        Lane& l = L[lane];
        lane_init(l);
        const prt_path_state& r = state[ids[lane]];
        l.origin = F3(r.origin[0], r.origin[1], r.origin[2]); l.t = r.time;
        l.dir = F3(r.dir[0], r.dir[1], r.dir[2]); l.time = r.dist;
        l.mask = F3(r.mask[0], r.mask[1], r.mask[2]); l.total = r.total;
        for (int k = 0; k < 4; ++k) l.acc[k] = r.acc[k];
        l.samples = r.samples;
        l.diff = r.diff; l.spec = r.spec; l.trans = r.trans; l.scatters = r.scatters;
        l.wasSpecular = r.was_specular != 0; l.reset = r.reset != 0;
        if (ahead) l.f = ahead[ids[lane]];
    }
    stack_mem.assign((size_t)sc.stack_levels * 64, 0u);
    XorShift xs{sched_seed ? sched_seed * 2654435761u + (uint32_t)(tile_x * 7919 + tile_y) : 0u};
    if (sched_seed && xs.s == 0) xs.s = 1;
    auto phase_T = [&]() -> unsigned { return sched_seed ? 1u + xs.next() % 64u : fa.walk_min_lanes; };
    TravStack stk[64];
    for (int lane = 0; lane < 64; ++lane) { stk[lane].lds = stack_mem.data() + lane; stk[lane].stride = 64; }

    // one walk phase of the wave (render_kernel B / D)
    auto walk_phase = [&](const bool any_hit, const int walk_stage, const int waiting_stage) {
        Ray wr[64];
        RayPre p[64];
        bool go[64];
        unsigned n_start = 0;
        unsigned n_other = 0;
        for (int lane = 0; lane < 64; ++lane) {
            go[lane] = false;
            if (!in_frame[lane]) continue;
            Lane& l = L[lane];
            const bool walking = l.stage == walk_stage;
            if (walking) {
                wr[lane] = any_hit ? lane_shadow_ray<MEDIUM, (MATS & PT_MATS_ENVIS) != 0>(l) : lane_closest_ray<MEDIUM>(l);
                p[lane] = ray_pre(wr[lane]);
                if (l.fresh) { walk_begin(sc, any_hit, wr[lane], any_hit ? wr[lane].t : PT_INF, p[lane], l.w, stk[lane]); l.fresh = false; }
            }
            go[lane] = walking && !l.w.done;
            n_start += go[lane] ? 1u : 0u;
            n_other += ((walking && l.w.done) || l.stage == waiting_stage) ? 1u : 0u;
        }
        const unsigned T = phase_T();
        const unsigned TQ = sched_seed ? xs.next() % 17u : fa.tri_sixteenths;        // a random share of pending lanes starts the triangle tests
        for (;;) {
            // render_kernel's loop: box steps of the lanes without a pending triangle, then -- once enough of the lanes in the loop
            // have one, or the phase is about to end -- one triangle of every pending lane
            unsigned n_in = 0, n_pend = 0, n_act = 0;
            for (int lane = 0; lane < 64; ++lane) {
                if (!go[lane]) continue;
                ++n_in;
                if (!L[lane].w.pend_count) walk_box(sc, any_hit, wr[lane], p[lane], L[lane].w, stk[lane]);
                if (L[lane].w.pend_count) ++n_pend;
                if (!L[lane].w.done) ++n_act;
            }
            if (!n_in) break;
            const bool cut = n_act < T && n_other + (n_start - n_act) > n_act;      // render_kernel's rule (PT_WAIT_RATIO 1)
            const bool tri = n_pend * 16u >= n_in * TQ || cut;
            for (int lane = 0; lane < 64; ++lane) {
                if (!go[lane]) continue;
                if (tri && L[lane].w.pend_count) walk_tri(sc, any_hit, wr[lane], L[lane].w);
                if (L[lane].w.done) go[lane] = false;
            }
            if (cut) break;
        }
    };

    for (;;) {
        bool any = false, laggards = false;
        bool runnable[64];
        for (int lane = 0; lane < 64; ++lane) laggards |= in_frame[lane] && lane_owes_frames(fa, L[lane]);
        for (int lane = 0; lane < 64; ++lane) {
            runnable[lane] = in_frame[lane] && lane_runnable(fa, L[lane], laggards);
            any |= in_frame[lane] && (runnable[lane] || L[lane].stage != ST_READY);
        }
        if (!any) break;
        for (int lane = 0; lane < 64; ++lane)
            if (runnable[lane]) lane_front<MATS, MEDIUM>(sc, cam, fa, L[lane], gxs[lane], gys[lane]);
        walk_phase(false, ST_WALKC, ST_BACK);
        for (int lane = 0; lane < 64; ++lane)
            if (in_frame[lane] && L[lane].stage == ST_WALKC && L[lane].w.done) lane_closest_done<MATS, MEDIUM>(sc, L[lane]);
        for (int lane = 0; lane < 64; ++lane)
            if (in_frame[lane] && L[lane].stage == ST_BACK) lane_back<MATS, MEDIUM>(sc, L[lane]);
        walk_phase(true, ST_WALKS, ST_FINISH);
        for (int lane = 0; lane < 64; ++lane)
            if (in_frame[lane] && L[lane].stage == ST_WALKS && L[lane].w.done) { L[lane].occluded = L[lane].w.found; L[lane].stage = ST_FINISH; }
        for (int lane = 0; lane < 64; ++lane)
            if (in_frame[lane] && L[lane].stage == ST_FINISH) lane_finish<MATS, MEDIUM>(sc, L[lane]);
    }
    for (int lane = 0; lane < 64; ++lane) {
        if (!in_frame[lane] || !L[lane].f) continue;
        const Lane& l = L[lane];
        prt_path_state& r = state[ids[lane]];
        r.origin[0] = l.origin.x; r.origin[1] = l.origin.y; r.origin[2] = l.origin.z; r.time = l.t;
        r.dir[0] = l.dir.x; r.dir[1] = l.dir.y; r.dir[2] = l.dir.z; r.dist = l.time;
        r.mask[0] = l.mask.x; r.mask[1] = l.mask.y; r.mask[2] = l.mask.z; r.total = l.total;
        for (int k = 0; k < 4; ++k) r.acc[k] = l.acc[k];
        r.samples = l.samples;
        r.diff = (uint16_t)l.diff; r.spec = (uint16_t)l.spec; r.trans = (uint16_t)l.trans; r.scatters = (uint16_t)l.scatters;
        r.was_specular = l.wasSpecular ? 1 : 0; r.reset = l.reset ? 1 : 0;
        if (ahead) {                                             // render_kernel keeps this in DevState::q4.w >> 2
            const bool frozen = fa.spp_limit && l.reset && l.samples >= fa.spp_limit;
            ahead[ids[lane]] = (!frozen && l.f > fa.n_frames) ? l.f - fa.n_frames : 0u;
        }
        const float ns = (MATS & PT_MATS_VIEW) ? 1.0f : (float)l.samples;
        float* px = out_rgba + 4 * ids[lane];
        for (int k = 0; k < 4; ++k) px[k] = l.acc[k] / ns;
    }
}

}  // namespace

// same arguments as oracle/pt_oracle.h's pto_job, plus the schedule.  Returns 0, or the prt error code of pack_scene.
extern "C" int emu_render(const prt_config* cfg, const prt_scene_desc* desc, const prt_camera* camera, const float* env_rgb, int env_w, int env_h,
                          int width, int full_height, int row0, int rows, int block_rows, int n_parts, int part,
                          uint32_t first_frame, uint32_t n_frames, const int32_t* seed_pairs, prt_path_state* state, float* out_rgba,
                          uint32_t spp_limit, uint32_t walk_min_lanes, uint32_t sched_seed, char* err, int err_len,
                          uint32_t seed_frames, uint32_t* ahead) {   // ahead != null: FrameArgs::run_ahead, per-pixel frames ahead in / out
    PackedScene ps;
    std::string perr;
    const int rc = pack_scene(*cfg, desc, ps, perr);
    if (rc) {
        if (err && err_len > 0) { std::strncpy(err, perr.c_str(), (size_t)err_len - 1); err[err_len - 1] = 0; }
        return rc;
    }
    DevScene sc = ps.sc;
    sc.pairs = ps.pairs.data(); sc.tri_geom = ps.tg.data(); sc.tri_nrm = ps.tn.data();
    sc.spheres = ps.spheres.data(); sc.quads = ps.quads.data(); sc.sdfs = ps.sdfs.data(); sc.mats = ps.mats.data(); sc.light_tab = ps.light_tab.data();
    static const float black[3] = {0.f, 0.f, 0.f};
    sc.env = env_rgb ? env_rgb : black; sc.env_w = env_rgb ? env_w : 1; sc.env_h = env_rgb ? env_h : 1;
    std::vector<float> cdf_rows, cdf_cols;
    if (cfg->env_importance_sampling) {
        build_env_cdf(sc.env, sc.env_w, sc.env_h, cdf_rows, cdf_cols);
        sc.env_cdf_rows = cdf_rows.data(); sc.env_cdf_cols = cdf_cols.data();
    }
    DevCamera cam{};
    camera_basis(*camera, cam);
    FrameArgs fa{};
    fa.width = width; fa.full_height = full_height; fa.row0 = row0; fa.rows = rows;
    fa.block_rows = block_rows > 0 ? block_rows : 1; fa.n_parts = n_parts > 0 ? n_parts : 1; fa.part = part;
    fa.first_frame = first_frame; fa.n_frames = n_frames; fa.seed_pairs = seed_pairs; fa.spp_limit = spp_limit;
    fa.seed_frames = (ahead && seed_frames > n_frames) ? seed_frames : n_frames; fa.run_ahead = ahead ? 1u : 0u;
    fa.unfinished = nullptr; fa.unfinished_host = nullptr; fa.tile_first = 0; fa.tile_stride = 1;
    fa.walk_min_lanes = walk_min_lanes ? walk_min_lanes : 8;
    fa.tri_sixteenths = 4;
    constexpr unsigned LD = PRT_MAT_LIGHT | PRT_MAT_DIFF;
    const char* e_generic = std::getenv("PT_EMU_GENERIC");
    const bool generic = e_generic && e_generic[0] == '1';
    std::vector<unsigned> stack_mem;
    const int tiles_x = (width + 7) / 8, tiles_y = (rows + 7) / 8;
    for (int ty = 0; ty < tiles_y; ++ty)
        for (int tx = 0; tx < tiles_x; ++tx) {
            // the variant launch_render (pt_kernels.hip) picks
            if (sc.env_is) {
                run_tile<PT_MATS_ENVIS, false>(sc, cam, fa, tx, ty, state, out_rgba, sched_seed, stack_mem, ahead);
            } else if (sc.pick_random_light) {
                if (!sc.has_medium) run_tile<PT_MATS_PICK, false>(sc, cam, fa, tx, ty, state, out_rgba, sched_seed, stack_mem, ahead);
                else run_tile<PT_MATS_PICK, true>(sc, cam, fa, tx, ty, state, out_rgba, sched_seed, stack_mem, ahead);
            } else if (sc.view) {
                constexpr unsigned V = PT_MATS_VIEW, VS = PT_MATS_VIEW | PT_MATS_SDF;
                if (sc.n_sdfs) {
                    if (!sc.has_medium) run_tile<VS, false>(sc, cam, fa, tx, ty, state, out_rgba, sched_seed, stack_mem, ahead);
                    else run_tile<VS, true>(sc, cam, fa, tx, ty, state, out_rgba, sched_seed, stack_mem, ahead);
                } else {
                    if (!sc.has_medium) run_tile<V, false>(sc, cam, fa, tx, ty, state, out_rgba, sched_seed, stack_mem, ahead);
                    else run_tile<V, true>(sc, cam, fa, tx, ty, state, out_rgba, sched_seed, stack_mem, ahead);
                }
            } else if (sc.n_sdfs) {
                if (!sc.has_medium) run_tile<PT_MATS_SDF, false>(sc, cam, fa, tx, ty, state, out_rgba, sched_seed, stack_mem, ahead);
                else run_tile<PT_MATS_SDF, true>(sc, cam, fa, tx, ty, state, out_rgba, sched_seed, stack_mem, ahead);
            } else if (!sc.has_medium) {
                // the compiled material sets of launch_render (pt_kernels.hip); PT_EMU_GENERIC=1: the run-time dispatch instead
                constexpr unsigned CO = LD | PRT_MAT_COAT, RC = LD | PRT_MAT_ROUGH_COND, RD = LD | PRT_MAT_DIEL | PRT_MAT_ROUGH_DIEL;
                const unsigned am = generic ? 0u : sc.active_mats;
                if (am == LD) run_tile<LD, false>(sc, cam, fa, tx, ty, state, out_rgba, sched_seed, stack_mem, ahead);
                else if (am == CO) run_tile<CO, false>(sc, cam, fa, tx, ty, state, out_rgba, sched_seed, stack_mem, ahead);
                else if (am == RC) run_tile<RC, false>(sc, cam, fa, tx, ty, state, out_rgba, sched_seed, stack_mem, ahead);
                else if (am == RD) run_tile<RD, false>(sc, cam, fa, tx, ty, state, out_rgba, sched_seed, stack_mem, ahead);
                else run_tile<0u, false>(sc, cam, fa, tx, ty, state, out_rgba, sched_seed, stack_mem, ahead);
            } else {
                if (!generic && sc.active_mats == LD) run_tile<LD, true>(sc, cam, fa, tx, ty, state, out_rgba, sched_seed, stack_mem, ahead);
                else run_tile<0u, true>(sc, cam, fa, tx, ty, state, out_rgba, sched_seed, stack_mem, ahead);
            }
        }
    return 0;
}

// prt_selftest_fn on the host: the same dispatcher (csrc/hip/pt_selftest.h) compiled for x86-64
extern "C" void emu_selftest_fn(int fn, const float* params, const float* in, float* out, int n) {
    for (int i = 0; i < n; ++i) selftest_fn(fn, params, in + 32 * (size_t)i, out + 32 * (size_t)i);
}
