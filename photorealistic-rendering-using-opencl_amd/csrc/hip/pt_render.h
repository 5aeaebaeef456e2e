// pt_render.h -- render_kernel (the hot kernel of libprt) and its launcher templates.  Included by the instance files
// pt_inst_*.hip, one per compile-time material set, so that the sets compile in parallel (build.py) -- the AOT analogue of the
// reference's per-scene program build (include/CL/cl_kernel.h:226-345 compiles exactly the scene's ACTIVE_MATS).  See pt_device.h
// for the arithmetic contract and the lane machine.
//
// Launch geometry: one wave per workgroup, owning an 8x8 pixel tile (primary rays of one wave walk the same BVH nodes), or 64
// pixels of 64 tiles in launches with few rounds of waves; dynamic LDS (DevScene::stack_levels x 256 B per workgroup) holds the
// traversal stacks.  A 1920x1080 frame is 32 400 workgroups >> 256 CUs x 24 resident waves, rendered as two interleaved sets of
// tiles on two streams (prt_api.cpp).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "pt_device.h"
#include "pt_launch.h"
#include "pt_pool.h"

namespace prt {

using namespace dev;

#ifndef PT_WAIT_RATIO
#define PT_WAIT_RATIO 1u    // a walk phase is cut short only while more than this many lanes wait per lane still walking (0: while
                            // any lane waits -- the first form of the rule: 1.3 % slower on cornell, the same on the big mesh)
#endif
// waves per SIMD the register allocator must leave room for: two builds per set, 5 (96 VGPRs) and 6 (80).  The history of these numbers
// is the history of the lane's registers: 4 (128 VGPRs) while the SLP vectorizer paired floats into 64-bit registers; 5
// without it and with the lane's flags as bit-fields of one word; 6 once fields of different phases shared registers, the cached hit
// dropped its position and the walk state its spare words.  Which of the two a launch takes: launch_variant.  7 waves the same as 6
// or worse, 8: -10 ... 20 %.
// LaunchOpts::waves = 5 / 6 forces one build (PT_WAVES / PT_BIG_WAVES; prt_set_option "waves", PRT_WAVES).
#ifndef PT_BIG_WAVES
#define PT_BIG_WAVES 6
#endif
#ifndef PT_WAVES
#define PT_WAVES 5
#endif
#ifndef PT_SUBWAVE_SLOTS
#define PT_SUBWAVE_SLOTS 6144u    // launch_sub_shift: waves the launches in flight may add up to with fewer pixels per wave
#endif
#define PT_BLOCK 64         // threads per workgroup: ONE wave.  A workgroup's slot (LDS, dispatch) frees only when its last wave
                            // ends, and waves over the mesh run ~3x longer than waves over a wall: one-wave groups +4 % over 256

// One wave = one 8x8 tile; every lane runs the lane machine of pt_device.h on its pixel until it has done its n_frames
// segments (or froze).  What is wave-level here is only the SCHEDULE: when the two walk phases of an iteration end.
//   walk phase rule: go on while at least fa.walk_min_lanes (closest-hit phase) / fa.shadow_min_lanes (any-hit phase) lanes
//   are still walking; below that, stop as soon as more lanes wait for the phase to end (they finished their walk in it, or
//   sit in the stage behind it) than walk.  A lane cut off keeps its WalkState and LDS stack and resumes in the same phase
//   of the next iteration.
//   triangle rule: inside a walk phase a lane alternates between box steps (walk_box) and the triangles those found (walk_tri, one
//   per call: pt_device.h "deferred leaves"); the wave runs the triangle unit once fa.tri_sixteenths / 16 of the lanes in the loop
//   have one pending, or when the phase is about to end -- it ran for 1.7 lanes of 64 when every box step carried its own loop.
//   run-ahead ("N spp" launches, fa.run_ahead): a lane that has done its n_frames starts on the next launch's frames for as
//   long as another lane of the wave still owes frames of this one; its lead goes into the state (q4.w >> 2).
// WAVES = waves per SIMD the register allocator leaves room for.
#ifdef PT_PHASE_CLOCKS                    // development builds: cycles of a wave per phase of the iteration (tools/phase_clocks.sh)
__device__ unsigned long long g_phase_clocks[12];    // 0..5, 7 cycles per phase, 6 iterations; 8 most iterations of one wave, 9 longest wave (cycles), 10 waves, 11 idle lane-iterations
#define PT_CLK(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); clk_[k] += now_ - last_; last_ = now_; } while (0)
#else
#define PT_CLK(k) do { } while (0)
#endif

// ORDER: the build that takes its tiles in the launcher's order and reports what they cost (FrameArgs::tile_order / tile_cost).  A build
// of its own because the three lines it adds are not free elsewhere: compiled into every kernel they cost the sets that spill most
// 1.6 ... 3.7 % (rough conductor 10.83 -> 10.46 G segments/s, rough dielectric 7.06 -> 6.95, medium 8.74 -> 8.60; LIGHT|DIFF +-0) -- one more
// scalar register alive through the frame loop -- while the order pays through big trees only (+3 ... 6 %).
template <unsigned MATS, bool MEDIUM, int WAVES, bool ORDER = false>
__global__ __launch_bounds__(PT_BLOCK, WAVES) void render_kernel(const DevScene sc, const DevCamera cam, const DevState S,
                                                                 const FrameArgs fa, float4* __restrict__ fb) {
    const int tiles_x = (fa.width + 7) / 8;
    const int lane = threadIdx.x & 63;
    // fa.scatter: the wave's 64 pixels come from 64 tiles spread over the launch's share of the frame instead of one 8x8 tile.
    // Every wave then gets its share of the expensive regions: a launch with few rounds of waves no longer waits for the
    // tiles over the mesh (512x512: +39 %); a big frame loses the coherence of neighbouring pixels' first segments (-17 %).
    // (not scattered: one tile per wave, and which one is the launcher's choice -- FrameArgs::tile_order)
    // fa.sub_shift: the wave renders P = 64 >> sub_shift pixels, 2^sub_shift waves share a tile (or, scattered, the launch's pixels)
    const unsigned P = 64u >> fa.sub_shift;
    const unsigned tile_g = blockIdx.x >> fa.sub_shift;
    const unsigned tile_k = (ORDER && !fa.scatter && fa.tile_order) ? fa.tile_order[tile_g] : tile_g;
    // What a tile costs is reported as the wave's ITERATIONS, not its clock ticks: a time stamp taken here (`s_memtime`, as intrinsic or as
    // inline assembly) counts for the compiler as something every later load may depend on, and such a load cannot go through the scalar
    // cache -- the quads, spheres, materials and the root of the tree all came through the vector-memory path in the first build of this
    // (2.2 x its instructions, 6 % of the scalar loads left; found with the instruction counters, the clock said +-0).
    unsigned iterations = 0u;
    const unsigned vpix = fa.scatter ? (unsigned)lane * gridDim.x + blockIdx.x : tile_k * 64u + (blockIdx.x & ((1u << fa.sub_shift) - 1u)) * P + (unsigned)lane;
    const unsigned tile = (vpix >> 6) * fa.tile_stride + fa.tile_first;
    const int tl = (int)(vpix & 63u);
    const int tile_x = (int)(tile % (unsigned)tiles_x), tile_y = (int)(tile / (unsigned)tiles_x);
    const int lx = tile_x * 8 + (tl & 7);
    const int ly = tile_y * 8 + (tl >> 3);
    // a lane outside the frame (edge tiles) idles through the kernel: every wave reaches the end, where the last one reports
    const bool in_frame = (unsigned)lane < P && lx < fa.width && ly < fa.rows;
    const size_t id = in_frame ? (size_t)ly * (size_t)fa.width + (size_t)lx : 0;
    const int gx = lx;
    const int gy = fa.row0 + (ly / fa.block_rows * fa.n_parts + fa.part) * fa.block_rows + ly % fa.block_rows;

    Lane L;
    lane_init(L);
    unsigned target = fa.n_frames;                              // frames this lane owes the launch
    if (!in_frame) { L.f = 0xffffffffu; L.reset = true; L.samples = 0xffffffffu; L.wasSpecular = false; }   // owes no frame, starts none
    else {
        const float4 a = S.q0[id], b = S.q1[id], c = S.q2[id], d = S.q3[id];
        const uint4 e = S.q4[id];
        L.origin = F3(a.x, a.y, a.z); L.t = a.w;                // TempRay.time = ray.t of the last segment (main.cl:28)
        L.dir = F3(b.x, b.y, b.z); L.time = b.w;                // TempRay.dist = ray.time
        L.mask = F3(c.x, c.y, c.z); L.total = prt_f2u(c.w);
        L.acc[0] = d.x; L.acc[1] = d.y; L.acc[2] = d.z; L.acc[3] = d.w;
        L.samples = e.x;
        L.diff = e.y & 0xffffu; L.spec = e.y >> 16;
        L.trans = e.z & 0xffffu; L.scatters = e.z >> 16;
        L.wasSpecular = (e.w & 1u) != 0; L.reset = (e.w & 2u) != 0;
        L.f = fa.run_ahead ? e.w >> 2 : 0u;                     // frames of this launch done in an earlier one ("N spp" launches only)
        // the pace of this pixel: its own mean path length so far (segments / paths started) over the frame's (FrameArgs::pace_inv_ref)
        if (fa.pace_inv_ref > 0.0f && e.x >= 8u) {
            // (cumulative: after this launch the pixel should have done pace x the frames of the launches so far; its lead L.f counts towards that)
            const float pace = fminf(fmaxf(d.w / (float)e.x * fa.pace_inv_ref, 1.0f), 3.0f);
            target = min(fa.n_frames + (unsigned)((pace - 1.0f) * (float)(fa.first_frame - 1u + fa.n_frames)), fa.seed_frames);
        }
    }
    extern __shared__ unsigned lds_stack[];                     // sc.stack_levels x PT_BLOCK, sized by the launch
    TravStack stk;
    stk.lds = lds_stack + threadIdx.x; stk.stride = PT_BLOCK;
    const unsigned T = fa.walk_min_lanes, TD = fa.shadow_min_lanes, TQ = fa.tri_sixteenths;
#ifdef PT_TEST_CLOBBER
    // tests/test_codegen.py: what a time stamp, an `asm volatile` or an LDS atomic in front of the frame loop is to the compiler -- a
    // write that every later load may depend on.  The uniform loads of the kernel must stay scalar behind it (pt_device.h, PT_CONST).
    asm volatile("" ::: "memory");
    atomicAdd(&lds_stack[0], 1u);
#endif
#ifdef PT_PHASE_CLOCKS
    unsigned long long clk_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_readcyclecounter();
    const unsigned long long start_ = last_;
    unsigned long long done_lanes_ = 0;
#endif
    for (;;) {
        const bool runnable = lane_runnable(fa, L, __any(lane_owes_frames(fa, L, target)), target);
        if (!__any(runnable || L.stage != ST_READY)) break;     // every lane has done its frames (or is frozen)
        if (ORDER) ++iterations;
        PT_CLK(7);
#ifdef PT_PHASE_CLOCKS
        done_lanes_ += (unsigned long long)__popcll(__ballot(!runnable && L.stage == ST_READY));
#endif
        if (runnable) { PT_WSTAT(4); lane_front<MATS, MEDIUM>(sc, cam, fa, L, gx, gy); }                // A
        PT_CLK(0);
        {                                                                                                 // B
            const bool walking = L.stage == ST_WALKC;
            const Ray wr = lane_closest_ray<MEDIUM>(L);
            const RayPre p = ray_pre(wr);
            if (walking && L.fresh) { walk_begin(sc, false, wr, PT_INF, p, L.w, stk); L.fresh = false; }
            const bool go = walking && !L.w.done;
            const unsigned n_start = (unsigned)__popcll(__ballot(go));
            const unsigned n_other = (unsigned)__popcll(__ballot((walking && L.w.done) || L.stage == ST_BACK));
            if (go) {
                for (;;) {
                    if (!L.w.pend_count) walk_box(sc, false, wr, p, L.w, stk);
                    // the triangles that the box steps found are tested once enough of the walking lanes have one pending (or the
                    // phase is about to end: a pending lane tests at least one per iteration of the wave)
                    const bool pending = L.w.pend_count != 0u;
                    const unsigned n_in = (unsigned)__popcll(__ballot(1)), n_pend = (unsigned)__popcll(__ballot(pending));
                    const unsigned n_act = (unsigned)__popcll(__ballot(!L.w.done));
                    const bool cut = n_act < T && n_other + (n_start - n_act) > PT_WAIT_RATIO * n_act;   // the lanes that wait outnumber the walkers
                    if (pending && (n_pend * 16u >= n_in * TQ || cut)) walk_tri(sc, false, wr, L.w);
                    if (L.w.done || cut) break;
                }
            }
            PT_CLK(1);
            if (walking && L.w.done) { PT_WSTAT(5); lane_closest_done<MATS, MEDIUM>(sc, L); }
            PT_CLK(2);
        }
        if (L.stage == ST_BACK) { PT_WSTAT(6); lane_back<MATS, MEDIUM>(sc, L); }                         // C
        PT_CLK(3);
        {                                                                                                 // D
            const bool walking = L.stage == ST_WALKS;
            const Ray wr = lane_shadow_ray<MEDIUM, (MATS & PT_MATS_ENVIS) != 0>(L);
            const RayPre p = ray_pre(wr);
            if (walking && L.fresh) { walk_begin(sc, true, wr, wr.t, p, L.w, stk); L.fresh = false; }
            const bool go = walking && !L.w.done;
            const unsigned n_start = (unsigned)__popcll(__ballot(go));
            const unsigned n_other = (unsigned)__popcll(__ballot((walking && L.w.done) || L.stage == ST_FINISH));
            if (go) {
                for (;;) {
                    if (!L.w.pend_count) walk_box(sc, true, wr, p, L.w, stk);
                    const bool pending = L.w.pend_count != 0u;
                    const unsigned n_in = (unsigned)__popcll(__ballot(1)), n_pend = (unsigned)__popcll(__ballot(pending));
                    const unsigned n_act = (unsigned)__popcll(__ballot(!L.w.done));
                    const bool cut = n_act < TD && n_other + (n_start - n_act) > PT_WAIT_RATIO * n_act;
                    if (pending && (n_pend * 16u >= n_in * TQ || cut)) walk_tri(sc, true, wr, L.w);
                    if (L.w.done || cut) break;
                }
            }
            if (walking && L.w.done) { L.occluded = L.w.found; L.stage = ST_FINISH; }
        }
        PT_CLK(4);
        if (L.stage == ST_FINISH) { PT_WSTAT(7); lane_finish<MATS, MEDIUM>(sc, L); }                           // E
        PT_CLK(5);
#ifdef PT_PHASE_CLOCKS
        ++clk_[6];
#endif
    }
#ifdef PT_PHASE_CLOCKS
    if (lane == (int)__builtin_ctzll(__ballot(1))) {
        for (int k = 0; k < 8; ++k) atomicAdd(&g_phase_clocks[k], clk_[k]);
        atomicMax(&g_phase_clocks[8], clk_[6]);
        atomicMax(&g_phase_clocks[9], last_ - start_);
        atomicAdd(&g_phase_clocks[10], 1ull);
        atomicAdd(&g_phase_clocks[11], done_lanes_);
    }
#endif
    if (in_frame && L.f) {
        // frames of the NEXT launch already done (run_ahead); a frozen pixel owes nothing and is ahead of nothing
        const bool frozen = fa.spp_limit && L.reset && L.samples >= fa.spp_limit;
        const unsigned frames_ahead = (!frozen && L.f > fa.n_frames) ? L.f - fa.n_frames : 0u;
        S.q0[id] = make_float4(L.origin.x, L.origin.y, L.origin.z, L.t);
        S.q1[id] = make_float4(L.dir.x, L.dir.y, L.dir.z, L.time);
        S.q2[id] = make_float4(L.mask.x, L.mask.y, L.mask.z, prt_u2f(L.total));
        S.q3[id] = make_float4(L.acc[0], L.acc[1], L.acc[2], L.acc[3]);
        S.q4[id] = make_uint4(L.samples, (L.diff & 0xffffu) | (L.spec << 16), (L.trans & 0xffffu) | (L.scatters << 16),
                              (L.wasSpecular ? 1u : 0u) | (L.reset ? 2u : 0u) | (frames_ahead << 2));
        const float ns = (MATS & PT_MATS_VIEW) ? 1.0f : (float)L.samples;      // write_imagef, main.cl:159 (a debug view: :161)
        fb[id] = make_float4(L.acc[0] / ns, L.acc[1] / ns, L.acc[2] / ns, L.acc[3] / ns);
    }
    if (ORDER && !fa.scatter && fa.tile_cost && lane == 0) atomicMax(&fa.tile_cost[tile_k], iterations);     // (the longest of the tile's waves)
    if (fa.unfinished) {
        const bool unfinished = in_frame && !(fa.spp_limit && L.reset && L.samples >= fa.spp_limit);
        const unsigned long long m = __ballot(unfinished);
        if (lane == (int)__builtin_ctzll(__ballot(1))) {
            // returning atomic: its value is back only once the add has been performed at the device's coherence point
            const unsigned long long before = m ? atomicAdd(fa.unfinished, (unsigned long long)__popcll(m)) : 0ull;
            if (fa.unfinished_host && before != ~0ull) {       // (never equal: the test orders the ticket behind the add without a
                // fence -- a device-scope fence writes back and invalidates this XCD's L2, 2.5 % when every wave does it)
                // The last wave of the launch hands the total to the host and leaves the counters clean for the next launch
                // (no wave returns early, so every wave of the grid gets here).
                if (atomicAdd(fa.unfinished + 1, 1ull) == (unsigned long long)gridDim.x - 1ull) {
                    const unsigned long long total = atomicExch(fa.unfinished, 0ull);
                    atomicExch(fa.unfinished + 1, 0ull);
                    *reinterpret_cast<volatile unsigned long long*>(fa.unfinished_host) = total;   // visible to the host at kernel end
                }
            }
        }
    }
}

// ---- host-side launchers -------------------------------------------------------------------------------
// tiles (= waves) of this launch, and whether its pixels are scattered over them (render_kernel): on when the launch has few rounds
// of waves -- up to 6 144 tiles, or 24 576 through a tree beyond one XCD's L2, whose expensive tiles are more expensive (1080p: +14 %,
// 3840x2160: -4 %)
static inline unsigned launch_grid(const DevScene& sc, const FrameArgs& fa, const LaunchOpts& lo, bool& scatter) {
    const unsigned tiles_x = ((unsigned)fa.width + 7u) / 8u, tiles_y = ((unsigned)fa.rows + 7u) / 8u;
    const unsigned n_tiles = tiles_x * tiles_y;
    const unsigned grid = fa.tile_first >= n_tiles ? 0u : (n_tiles - fa.tile_first + fa.tile_stride - 1) / fa.tile_stride;   // tiles of this sub-part
    scatter = lo.scatter >= 0 ? lo.scatter != 0 : grid <= (sc.n_pairs > 65536u ? 24576u : 6144u);
    return grid;
}
// Pixels per wave (FrameArgs::sub_shift).  A launch whose tiles leave wave slots of the chip empty -- one rank's share of a frame
// split N ways, a small frame -- lasts as long as its slowest wave, and a wave as long as its slowest lane's chain of segments (every
// frame of a pixel follows its previous one: DESIGN.md s5); with 32 or 16 pixels per wave the maximum runs over fewer lanes, an
// iteration walks for fewer of them, and the waves that would have idled hold the other halves.  `in_flight`: launches that share the
// chip (the sub-parts of prt_api.cpp).  6 144 slots = 256 CUs x 4 SIMDs x 6 waves.
static inline unsigned launch_sub_shift(const LaunchOpts& lo, unsigned grid, unsigned in_flight) {
    if (lo.pix_per_wave == 64) return 0u;
    if (lo.pix_per_wave == 32) return 1u;
    if (lo.pix_per_wave == 16) return 2u;
    // Measured (tools/shard_time.py, one rank's share of the 1080p frame split 8 / 4 / 2 ways, 256 spp): 64 pixels per wave 0.091 / 0.117 /
    // 0.175 s, 32: 0.111 / 0.174 / 0.256 s, 16: 0.158 / 0.267 / 0.418 s -- the chain of a wave's slowest pixel does not get shorter with
    // fewer neighbours (an iteration issues the same instructions whoever takes part), and the extra waves cost issue slots the chip
    // does not have to spare even at an eighth of the frame.  So: never by itself.
    (void)grid; (void)in_flight;
    return 0u;
}
template <unsigned MATS, bool MEDIUM, int WAVES>
static void launch_variant_w(const DevScene& sc, const DevCamera& cam, const DevState& S, const FrameArgs& fa, float4* fb,
                             hipStream_t stream, unsigned grid, bool scatter) {
    const size_t lds = (size_t)sc.stack_levels * PT_BLOCK * sizeof(unsigned);
    static size_t lds_attr = 0;                                  // per template instance
    if (lds > 65536u && lds > lds_attr) {   // only a tree that fills the reference's 64-entry stack to the brim
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&render_kernel<MATS, MEDIUM, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (WAVES == PT_BIG_WAVES) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&render_kernel<MATS, MEDIUM, WAVES, WAVES == PT_BIG_WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        lds_attr = lds;
    }
    if (!grid) return;
    FrameArgs fb_args = fa;
    fb_args.scatter = scatter ? 1u : 0u;
    grid <<= fb_args.sub_shift;                                  // workgroups = waves: 2^sub_shift per tile
    // walk phases end below this many walking lanes (0 = not set by the caller): 8 (10 / 12: +-0.7 % in either wave-count build; the SDF
    // sets: 10 -3 %); 6 with a medium (8: -1 %) and for scattered pixels, whose waves hold more deep walks (512x512 coat: 6 +4 %).  Shadow rays: in a small tree 99 % end at the root and the
    // rest is shallow -- cutting one off costs its pixel a whole iteration, letting the wave finish them costs a few steps (cornell
    // +3 %); through a big mesh they are as deep as any ray and the bound pays as it does for the closest-hit walks (+14 %).
    // Through a tree beyond one XCD's L2: 20 / 12 (3840x2160 x 512 spp, 4096 frames per launch: 12 / 12 -> 3.84, 16 / 16 -> 3.93, 16 / 12 -> 3.96,
    // 20 / 12 -> 4.02 G segments/s; round 2, 512 frames per launch: 8 -> 2.46, 12 -> 2.52, 16 -> 2.51).
    // A launch that leaves wave slots empty (up to 1.5 rounds of waves in flight: one rank's share of a 1080p frame split 4 or 8 ways) lasts as
    // long as its slowest pixel's chain of segments, and a lane that is cut off needs another iteration of its wave for the same segment: there
    // every walk runs to its end (lock step) -- the share of 8 ranks 0.091 -> 0.077 s, of 4 ranks 0.117 -> 0.108 s, of 2 ranks 0.175 -> 0.209 s, the
    // whole frame 0.286 -> 0.340 s (tools/shard_time.py, 256 spp).  Through a big tree the bound stays (the same sweep: +-2 %, or worse).
    // (with fewer pixels per wave the thresholds shrink with the lanes that can walk at all)
    const bool few_waves = sc.n_pairs <= 65536u && (unsigned long long)(grid >> fb_args.sub_shift) * (fa.tile_stride ? fa.tile_stride : 1u) <= 9216ull;
    if (!fb_args.walk_min_lanes) fb_args.walk_min_lanes = few_waves ? 1u : max(2u, (sc.n_pairs > 65536u ? 20u : ((MEDIUM || scatter) ? 6u : 8u)) >> fb_args.sub_shift);
    if (!fb_args.shadow_min_lanes) fb_args.shadow_min_lanes = sc.n_pairs > 65536u ? max(2u, 12u >> fb_args.sub_shift) : 1u;
    if (WAVES == PT_BIG_WAVES && (fb_args.tile_order || fb_args.tile_cost))     // (prt_render_spp asks for it through big trees only)
        hipLaunchKernelGGL((render_kernel<MATS, MEDIUM, WAVES, WAVES == PT_BIG_WAVES>), dim3(grid), dim3(PT_BLOCK), lds, stream, sc, cam, S, fb_args, fb);
    else
        hipLaunchKernelGGL((render_kernel<MATS, MEDIUM, WAVES>), dim3(grid), dim3(PT_BLOCK), lds, stream, sc, cam, S, fb_args, fb);
}
// one material set x medium: picks the wave-count build and the pixel-to-wave mapping; reports both (RenderLaunch)
template <unsigned MATS, bool MEDIUM>
static RenderLaunch launch_variant(const char* name, const DevScene& sc, const DevCamera& cam, const DevState& S, const FrameArgs& fa, float4* fb,
                                   hipStream_t stream, const LaunchOpts& lo) {
    bool scatter;
    const unsigned grid = launch_grid(sc, fa, lo, scatter);
    // Waves per SIMD = which register budget the set runs best at.  6 (80 registers) for the light sets, with a medium, for the raymarched
    // SDF sets and through big trees; 5 (96 registers) for the sets whose 80-register build spills most -- the coat set (68 instead of 152 B
    // of scratch) and the generic dispatch (112 instead of 208 B) -- and for scattered pixels of a small tree (one or two rounds of waves,
    // every wave at its own latency: 512x512 coat +5 %).  Measured at 1080p, 5 against 6 waves: LIGHT|DIFF -6 %, rough conductor -3.6 %,
    // rough dielectric -2.3 %, coat +-0 ... +1.3 %, generic +-0; with a medium -7 %, SDF -15 %.
    // Round 4, the sets outside the BASELINE configs (same call, twice each, 1080p): generic dispatch 5 against 6 waves +8 % (cornell_mixed;
    // 84 instead of 180 B of scratch), generic with a medium +3 % (132 / 200 B), coat +1 %, the raymarched SDF set -8 % (116 / 172 ... 256 B:
    // the raymarcher's latency wants the sixth wave more than its spills cost) -- so the run-time dispatch runs its 96-register build with or
    // without a medium (and with it the debug-view, light-pick and environment-sampling sets, which are that dispatch plus a flag), SDF its 80.
    constexpr bool five = !(MATS & PT_MATS_SDF) && ((MATS & ~PT_MATS_FLAGS) == 0u || (!MEDIUM && (MATS & PRT_MAT_COAT) != 0u));
    const int waves = lo.waves ? lo.waves : (sc.n_pairs > 65536u ? PT_BIG_WAVES : ((scatter || five) ? PT_WAVES : PT_BIG_WAVES));
    RenderLaunch r;
    r.name = name; r.scatter = scatter ? 1 : 0;
    FrameArgs fs = fa;
    fs.sub_shift = lo.pool ? 0u : launch_sub_shift(lo, grid, fa.tile_stride);
    r.pix_per_wave = 64 >> fs.sub_shift;
    r.ordered = (!scatter && fa.tile_order && waves >= PT_BIG_WAVES) ? 1 : 0;
    // render_kernel_rp (pt_pool.h): the same launch -- same tiles or scattered pixels, same order -- with the deep walks on walker waves
    if (lo.pool) {
        FrameArgs fp = fa;
        fp.scatter = scatter ? 1u : 0u;
        const bool ordered = !scatter && (fa.tile_order || fa.tile_cost);
        const bool ok = ordered ? launch_rp<MATS, MEDIUM, PT_BIG_WAVES, true>(sc, cam, S, fp, fb, stream, grid)
                                : launch_rp<MATS, MEDIUM, PT_BIG_WAVES, false>(sc, cam, S, fp, fb, stream, grid);
        if (ok) { r.waves = PT_BIG_WAVES; r.pool = 1; return r; }
    }
#if defined(PT_DEV_ONE_VARIANT) && !defined(PT_DEV_BOTH_WAVES)
    r.waves = PT_BIG_WAVES;
    launch_variant_w<MATS, MEDIUM, PT_BIG_WAVES>(sc, cam, S, fs, fb, stream, grid, scatter);
#else
    if (waves >= PT_BIG_WAVES) { r.waves = PT_BIG_WAVES; launch_variant_w<MATS, MEDIUM, PT_BIG_WAVES>(sc, cam, S, fs, fb, stream, grid, scatter); }
    else { r.waves = PT_WAVES; launch_variant_w<MATS, MEDIUM, PT_WAVES>(sc, cam, S, fs, fb, stream, grid, scatter); }
#endif
    return r;
}

}  // namespace prt
