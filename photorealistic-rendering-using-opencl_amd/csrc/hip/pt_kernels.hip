// pt_kernels.hip -- gfx950 kernels of libprt (see pt_device.h for the arithmetic contract and the lane machine).
//
//   render_kernel<MATS, MEDIUM>   the hot path: one lane = one pixel, n_frames segments per launch
//   state_to_rtd / rtd_to_state   80 B/px SoA planes <-> the reference's 112 B RTD records
//   count_kernel                  sum of samples / segments / frozen pixels (Msamples/s accounting)
//
// Launch geometry: one wave per workgroup, owning an 8x8 pixel tile (primary rays of one wave walk the same
// BVH nodes); dynamic LDS (DevScene::stack_levels x 256 B per workgroup) holds the traversal stacks.  A
// 1920x1080 frame is 32 400 workgroups >> 256 CUs x 16 resident waves, rendered as two interleaved sets of
// tiles on two streams (prt_api.cpp).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "pt_device.h"
#include "pt_launch.h"
#include "pt_selftest.h"

namespace prt {

using namespace dev;

#ifndef PT_WAIT_RATIO
#define PT_WAIT_RATIO 1u    // a walk phase is cut short only while more than this many lanes wait per lane still walking (0: while
                            // any lane waits -- the first form of the rule: 1.3 % slower on cornell, the same on the big mesh)
#endif
// waves per SIMD the register allocator must leave room for: 6 (80 VGPRs).  The history of this number is the history of the lane's
// registers: 4 (128 VGPRs) while the SLP vectorizer paired floats into 64-bit registers; 5 (96) without it and with the
// lane's flags as bit-fields of one word; 6 once fields of different phases shared registers, the cached hit dropped its position
// and the walk state its spare words -- +1 ... 3 % over 5 on every variant (LIGHT|DIFF: 56 B of scratch at 80 VGPRs, none at 96;
// generic: 88 B; with a medium 76 ... 120 B), +4 ... 7 % through a tree beyond one XCD's L2.  7 waves the same, 8: -10 %.
// PRT_WAVES=4 / 5 / 6 forces one build (PT_MIN_WAVES / PT_WAVES / PT_BIG_WAVES).
#ifndef PT_BIG_WAVES
#define PT_BIG_WAVES 6
#endif
#ifndef PT_WAVES
#define PT_WAVES 5
#endif
#ifndef PT_MIN_WAVES
#define PT_MIN_WAVES 4
#endif
#ifndef PT_UNIFIED
#define PT_UNIFIED 0        // a lane's box step and its pending triangle test share one load per iteration (0: box steps, then triangle tests)
#endif
#define PT_BLOCK 64         // threads per workgroup: ONE wave.  A workgroup's slot (LDS, dispatch) frees only when its last wave
                            // ends, and waves over the mesh run ~3x longer than waves over a wall: one-wave groups +4 % over 256

// One wave = one 8x8 tile; every lane runs the lane machine of pt_device.h on its pixel until it has done its n_frames
// segments (or froze).  What is wave-level here is only the SCHEDULE: when the two walk phases of an iteration end.
//   walk phase rule: go on while at least fa.walk_min_lanes (closest-hit phase) / fa.shadow_min_lanes (any-hit phase) lanes
//   are still walking; below that, stop as soon as more lanes wait for the phase to end (they finished their walk in it, or
//   sit in the stage behind it) than walk.  A lane cut off keeps its WalkState and LDS stack and resumes in the same phase
//   of the next iteration.
//   run-ahead ("N spp" launches, fa.run_ahead): a lane that has done its n_frames starts on the next launch's frames for as
//   long as another lane of the wave still owes frames of this one; its lead goes into the state (q4.w >> 2).
// WAVES = waves per SIMD the register allocator leaves room for.
#ifdef PT_PHASE_CLOCKS                    // development builds: cycles of a wave per phase of the iteration (tools/phase_clocks.sh)
__device__ unsigned long long g_phase_clocks[12];    // 0..5, 7 cycles per phase, 6 iterations; 8 most iterations of one wave, 9 longest wave (cycles), 10 waves, 11 idle lane-iterations
#define PT_CLK(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); clk_[k] += now_ - last_; last_ = now_; } while (0)
#else
#define PT_CLK(k) do { } while (0)
#endif

template <unsigned MATS, bool MEDIUM, int WAVES>
__global__ __launch_bounds__(PT_BLOCK, WAVES) void render_kernel(const DevScene sc, const DevCamera cam, const DevState S,
                                                                 const FrameArgs fa, float4* __restrict__ fb) {
    const int tiles_x = (fa.width + 7) / 8;
    const int lane = threadIdx.x & 63;
    // fa.scatter: the wave's 64 pixels come from 64 tiles spread over the launch's share of the frame instead of one 8x8 tile.
    // Every wave then gets its share of the expensive regions: a launch with few rounds of waves no longer waits for the
    // tiles over the mesh (512x512: +39 %); a big frame loses the coherence of neighbouring pixels' first segments (-17 %).
    const unsigned vpix = fa.scatter ? (unsigned)lane * gridDim.x + blockIdx.x : blockIdx.x * 64u + (unsigned)lane;
    const unsigned tile = (vpix >> 6) * fa.tile_stride + fa.tile_first;
    const int tl = (int)(vpix & 63u);
    const int tile_x = (int)(tile % (unsigned)tiles_x), tile_y = (int)(tile / (unsigned)tiles_x);
    const int lx = tile_x * 8 + (tl & 7);
    const int ly = tile_y * 8 + (tl >> 3);
    // a lane outside the frame (edge tiles) idles through the kernel: every wave reaches the end, where the last one reports
    const bool in_frame = lx < fa.width && ly < fa.rows;
    const size_t id = in_frame ? (size_t)ly * (size_t)fa.width + (size_t)lx : 0;
    const int gx = lx;
    const int gy = fa.row0 + (ly / fa.block_rows * fa.n_parts + fa.part) * fa.block_rows + ly % fa.block_rows;

    Lane L;
    lane_init(L);
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
        L.f = e.w >> 2;                                         // frames of this launch done in an earlier one (run_ahead)
    }
    extern __shared__ unsigned lds_stack[];                     // sc.stack_levels x PT_BLOCK, sized by the launch
    TravStack stk;
    stk.lds = lds_stack + threadIdx.x; stk.stride = PT_BLOCK;
    const unsigned T = fa.walk_min_lanes, TD = fa.shadow_min_lanes, TQ = fa.tri_sixteenths;
#ifdef PT_PHASE_CLOCKS
    unsigned long long clk_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_readcyclecounter();
    const unsigned long long start_ = last_;
    unsigned long long done_lanes_ = 0;
#endif
    for (;;) {
        const bool runnable = lane_runnable(fa, L, __any(lane_owes_frames(fa, L)));
        if (!__any(runnable || L.stage != ST_READY)) break;     // every lane has done its frames (or is frozen)
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
#if PT_UNIFIED
                // every iteration a lane takes ONE unit of its walk (the box half of a step, or one pending triangle) behind one
                // load; the lanes with a triangle pending sit iterations out until enough of the phase's lanes have one (the first
                // iteration of a phase takes them anyway: a pending lane tests at least one triangle per iteration of the wave)
                for (bool first = true;; first = false) {
                    const bool pending = L.w.pend_count != 0u;
                    const unsigned n_in = (unsigned)__popcll(__ballot(1)), n_pend = (unsigned)__popcll(__ballot(pending));
                    if (!pending || first || n_pend * 16u >= n_in * TQ) walk_unit(sc, false, wr, p, L.w, stk);
                    if (L.w.done) break;
                    const unsigned n_act = (unsigned)__popcll(__ballot(1));
                    if (n_act < T && n_other + (n_start - n_act) > PT_WAIT_RATIO * n_act) break;   // the lanes that wait outnumber the walkers
                }
#else
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
#endif
            }
            PT_CLK(1);
            if (walking && L.w.done) { PT_WSTAT(5); lane_closest_done<MATS, MEDIUM>(sc, L); }
            PT_CLK(2);
        }
        if (L.stage == ST_BACK) { PT_WSTAT(6); lane_back<MATS, MEDIUM>(sc, L); }                         // C
        PT_CLK(3);
        {                                                                                                 // D
            const bool walking = L.stage == ST_WALKS;
            const Ray wr = lane_shadow_ray<MEDIUM>(L);
            const RayPre p = ray_pre(wr);
            if (walking && L.fresh) { walk_begin(sc, true, wr, wr.t, p, L.w, stk); L.fresh = false; }
            const bool go = walking && !L.w.done;
            const unsigned n_start = (unsigned)__popcll(__ballot(go));
            const unsigned n_other = (unsigned)__popcll(__ballot((walking && L.w.done) || L.stage == ST_FINISH));
            if (go) {
#if PT_UNIFIED
                // every iteration a lane takes ONE unit of its walk (the box half of a step, or one pending triangle) behind one
                // load; the lanes with a triangle pending sit iterations out until enough of the phase's lanes have one (the first
                // iteration of a phase takes them anyway: a pending lane tests at least one triangle per iteration of the wave)
                for (bool first = true;; first = false) {
                    const bool pending = L.w.pend_count != 0u;
                    const unsigned n_in = (unsigned)__popcll(__ballot(1)), n_pend = (unsigned)__popcll(__ballot(pending));
                    if (!pending || first || n_pend * 16u >= n_in * TQ) walk_unit(sc, true, wr, p, L.w, stk);
                    if (L.w.done) break;
                    const unsigned n_act = (unsigned)__popcll(__ballot(1));
                    if (n_act < TD && n_other + (n_start - n_act) > PT_WAIT_RATIO * n_act) break;   // the lanes that wait outnumber the walkers
                }
#else
                for (;;) {
                    if (!L.w.pend_count) walk_box(sc, true, wr, p, L.w, stk);
                    const bool pending = L.w.pend_count != 0u;
                    const unsigned n_in = (unsigned)__popcll(__ballot(1)), n_pend = (unsigned)__popcll(__ballot(pending));
                    const unsigned n_act = (unsigned)__popcll(__ballot(!L.w.done));
                    const bool cut = n_act < TD && n_other + (n_start - n_act) > PT_WAIT_RATIO * n_act;
                    if (pending && (n_pend * 16u >= n_in * TQ || cut)) walk_tri(sc, true, wr, L.w);
                    if (L.w.done || cut) break;
                }
#endif
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

__global__ void state_to_rtd(const DevState S, prt_path_state* __restrict__ out, size_t n) {
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n) return;
    const float4 a = S.q0[id], b = S.q1[id], c = S.q2[id], d = S.q3[id];
    const uint4 e = S.q4[id];
    prt_path_state r;
    __builtin_memset(&r, 0, sizeof(r));
    r.origin[0] = a.x; r.origin[1] = a.y; r.origin[2] = a.z; r.time = a.w;
    r.dir[0] = b.x; r.dir[1] = b.y; r.dir[2] = b.z; r.dist = b.w;
    r.mask[0] = c.x; r.mask[1] = c.y; r.mask[2] = c.z; r.total = prt_f2u(c.w);
    r.acc[0] = d.x; r.acc[1] = d.y; r.acc[2] = d.z; r.acc[3] = d.w;
    r.samples = e.x;
    r.diff = (uint16_t)(e.y & 0xffffu); r.spec = (uint16_t)(e.y >> 16);
    r.trans = (uint16_t)(e.z & 0xffffu); r.scatters = (uint16_t)(e.z >> 16);
    r.was_specular = (uint8_t)(e.w & 1u); r.reset = (uint8_t)((e.w >> 1) & 1u);
    out[id] = r;
}

__global__ void rtd_to_state(const prt_path_state* __restrict__ in, const DevState S, float4* __restrict__ fb, size_t n) {
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n) return;
    const prt_path_state r = in[id];
    S.q0[id] = make_float4(r.origin[0], r.origin[1], r.origin[2], r.time);
    S.q1[id] = make_float4(r.dir[0], r.dir[1], r.dir[2], r.dist);
    S.q2[id] = make_float4(r.mask[0], r.mask[1], r.mask[2], prt_u2f(r.total));
    S.q3[id] = make_float4(r.acc[0], r.acc[1], r.acc[2], r.acc[3]);
    S.q4[id] = make_uint4(r.samples, (uint32_t)r.diff | ((uint32_t)r.spec << 16), (uint32_t)r.trans | ((uint32_t)r.scatters << 16),
                          (r.was_specular ? 1u : 0u) | (r.reset ? 2u : 0u));
    if (r.samples) {
        const float ns = (float)r.samples;
        fb[id] = make_float4(r.acc[0] / ns, r.acc[1] / ns, r.acc[2] / ns, r.acc[3] / ns);
    } else {
        fb[id] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// out[0] = sum samples, out[1] = sum acc.w (exact: acc.w is an integer-valued float < 2^24), out[2] = frozen pixels
__global__ void count_kernel(const DevState S, size_t n, unsigned spp, unsigned long long* __restrict__ out) {
    unsigned long long s = 0, g = 0, z = 0;
    for (size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (size_t)gridDim.x * blockDim.x) {
        const uint4 e = S.q4[id];
        s += e.x;
        g += (unsigned long long)S.q3[id].w;
        z += (spp && (e.w & 2u) && e.x >= spp) ? 1u : 0u;
    }
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_down(s, off); g += __shfl_down(g, off); z += __shfl_down(z, off);
    }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&out[0], s); atomicAdd(&out[1], g); atomicAdd(&out[2], z); }
}

// shaders/tonemapper.glsl:12-23,47-64 -- display transform of the linear framebuffer (not part of the radiance
// loop; float accuracy of an 8-bit output is uncritical, the stated math library is used anyway)
PT_DEV float filmic_reinhard_curve(float x) {
    const float T2 = 7.5f;
    float q = (T2 * T2 + 1.0f) * x * x;
    return q / (q + x + T2 * T2);
}
PT_DEV float smoothstep_f(float e0, float e1, float x) {
    float t = prt_fmin(prt_fmax((x - e0) / (e1 - e0), 0.0f), 1.0f);
    return t * t * (3.0f - 2.0f * t);
}
__global__ void tonemap_kernel(const float4* __restrict__ fb, uchar4* __restrict__ out, const FrameArgs fa) {
    const size_t npix = (size_t)fa.width * (size_t)fa.rows;
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= npix) return;
    const int lx = (int)(id % (size_t)fa.width), ly = (int)(id / (size_t)fa.width);
    const int gy = fa.row0 + (ly / fa.block_rows * fa.n_parts + fa.part) * fa.block_rows + ly % fa.block_rows;
    // gl_FragCoord = pixel centre; p = 1 - 2 * fragCoord / resolution
    const float px = 1.0f - 2.0f * ((float)lx + 0.5f) / (float)fa.width;
    const float py = 1.0f - 2.0f * ((float)gy + 0.5f) / (float)fa.full_height;
    float vignette = 1.25f / (1.1f + 1.1f * (px * px + py * py));
    vignette *= vignette;
    vignette = 1.0f * (1.0f - 0.25f) + smoothstep_f(0.1f, 1.1f, vignette) * 0.25f;             // mix(1, smoothstep(..), 0.25)
    const float4 c = fb[id];
    const float w = filmic_reinhard_curve(1.2f);
    float rgb[3] = {c.x, c.y, c.z};
    unsigned char o[3];
    for (int k = 0; k < 3; ++k) {
        float v = rgb[k] * vignette;
        v = filmic_reinhard_curve(1.0f * v) / w;
        v = smoothstep_f(-0.025f, 1.0f, v);
        v = prt_pow(v, 1.0f / 2.2f);
        v = prt_fmin(prt_fmax(v, 0.0f), 1.0f);                    // NaN -> 0 (prt_fmax ignores a NaN operand)
        o[k] = (unsigned char)prt_rint(v * 255.0f);
    }
    out[id] = make_uchar4(o[0], o[1], o[2], 255);
}

// the same switch is evaluated on the host by oracle/detmath_probe.c
__global__ void selftest_math_kernel(int fn, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int n) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    if (fn == 17) {         // exhaustive: lane i compares hw_recip with the IEEE divide on the 65536 bit patterns (i << 16) + k
        unsigned bad = 0;
        for (unsigned k = 0; k < 65536u; ++k) {
            const float v = prt_u2f(((unsigned)i << 16) + k);
            const float q = 1.0f / v, h = hw_recip(v);
            if (prt_f2u(q) != prt_f2u(h) && !(q != q && h != h)) ++bad;
        }
        out[i] = (float)bad;
        return;
    }
    if (fn == 18) {         // exhaustive: out_of_unit_range(x, c, u) against the divide, c = b[0], x = the 65536 patterns (i << 16) + k
        const float c = b[0];
        const float u = (c >= 9.094947017729282e-13f && c <= 1099511627776.0f) ? c * 5.9604644775390625e-08f : prt_u2f(0x7fc00000u);
        unsigned bad = 0;
        for (unsigned k = 0; k < 65536u; ++k) {
            const float v = prt_u2f(((unsigned)i << 16) + k);
            const float l = v / c;
            if ((l < 0.0f || l > 1.0f) != out_of_unit_range(v, c, u)) ++bad;
        }
        out[i] = (float)bad;
        return;
    }
    const float x = a[i], y = b[i];
    float r;
    switch (fn) {
        case 0: r = prt_sin(x); break;
        case 1: r = prt_cos(x); break;
        case 2: r = prt_tan(x); break;
        case 3: r = prt_exp(x); break;
        case 4: r = prt_log(x); break;
        case 5: r = prt_acos(x); break;
        case 6: r = prt_atan2(x, y); break;
        case 7: r = prt_pow(x, y); break;
        case 8: r = prt_sqrt(x); break;
        case 9: r = x / y; break;
        case 10: r = prt_fma(x, y, x); break;
        case 11: r = prt_fmin(x, y); break;
        case 12: r = prt_fmax(x, y); break;
        case 13: r = prt_round(x); break;
        case 14: r = prt_floor(x); break;
        case 16: r = prt_cbrt(x); break;
        default: r = prt_recip(x); break;
    }
    out[i] = r;
}

// per-function known-answer entry (pt_selftest.h): one case per lane
__global__ void selftest_fn_kernel(int fn, const float* __restrict__ params, const float* __restrict__ in, float* __restrict__ out, int n) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    float p[80], x[32], y[32];
    for (int k = 0; k < 80; ++k) p[k] = params[k];
    for (int k = 0; k < 32; ++k) x[k] = in[32 * (size_t)i + k];
    selftest_fn(fn, p, x, y);
    for (int k = 0; k < 32; ++k) out[32 * (size_t)i + k] = y[k];
}

// ---- host-side launchers -------------------------------------------------------------------------------
// tiles (= waves) of this launch, and whether its pixels are scattered over them (render_kernel): on when the launch has few rounds
// of waves -- up to 6 144 tiles, or 24 576 through a tree beyond one XCD's L2, whose expensive tiles are more expensive (1080p: +14 %,
// 3840x2160: -4 %)
static unsigned launch_grid(const DevScene& sc, const FrameArgs& fa, bool& scatter) {
    const unsigned tiles_x = ((unsigned)fa.width + 7u) / 8u, tiles_y = ((unsigned)fa.rows + 7u) / 8u;
    const unsigned n_tiles = tiles_x * tiles_y;
    const unsigned grid = fa.tile_first >= n_tiles ? 0u : (n_tiles - fa.tile_first + fa.tile_stride - 1) / fa.tile_stride;   // tiles of this sub-part
    static const int forced_scatter = [] { const char* e = std::getenv("PRT_SCATTER"); return e ? std::atoi(e) : -1; }();
    scatter = forced_scatter >= 0 ? forced_scatter != 0 : grid <= (sc.n_pairs > 65536u ? 24576u : 6144u);
    return grid;
}
template <unsigned MATS, bool MEDIUM, int WAVES>
static void launch_variant_w(const DevScene& sc, const DevCamera& cam, const DevState& S, const FrameArgs& fa, float4* fb,
                             hipStream_t stream, unsigned grid, bool scatter) {
    const size_t lds = (size_t)sc.stack_levels * PT_BLOCK * sizeof(unsigned);
    static size_t lds_attr = 0;                                  // per template instance
    if (lds > 65536u && lds > lds_attr) {   // only a tree that fills the reference's 64-entry stack to the brim
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&render_kernel<MATS, MEDIUM, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        lds_attr = lds;
    }
    if (!grid) return;
    FrameArgs fb_args = fa;
    fb_args.scatter = scatter ? 1u : 0u;
    // walk phases end below this many walking lanes (0 = not set by the caller): 8; 6 with a medium (8: -1 %) and for scattered
    // pixels, whose waves hold more deep walks (512x512 coat: 6 +4 %).  Shadow rays: in a small tree 99 % end at the root and the
    // rest is shallow -- cutting one off costs its pixel a whole iteration, letting the wave finish them costs a few steps (cornell
    // +3 %); through a big mesh they are as deep as any ray and the bound pays as it does for the closest-hit walks (+14 %).
    // Through a tree beyond one XCD's L2: 12 (3840x2160: 8 -> 2.46, 12 -> 2.52, 16 -> 2.51 G segments/s).
    if (!fb_args.walk_min_lanes) fb_args.walk_min_lanes = sc.n_pairs > 65536u ? 12u : ((MEDIUM || scatter) ? 6u : 8u);
    if (!fb_args.shadow_min_lanes) fb_args.shadow_min_lanes = sc.n_pairs > 65536u ? fb_args.walk_min_lanes : 1u;
    hipLaunchKernelGGL((render_kernel<MATS, MEDIUM, WAVES>), dim3(grid), dim3(PT_BLOCK), lds, stream, sc, cam, S, fb_args, fb);
}
template <unsigned MATS, bool MEDIUM>
static void launch_variant(const DevScene& sc, const DevCamera& cam, const DevState& S, const FrameArgs& fa, float4* fb,
                           hipStream_t stream) {
    static const int forced = [] { const char* e = std::getenv("PRT_WAVES"); return e ? std::atoi(e) : 0; }();   // 4 / 5 / 6: override (tests, experiments)
    bool scatter;
    const unsigned grid = launch_grid(sc, fa, scatter);
    // 6 waves per SIMD; 5 where a launch of a small tree is one or two rounds of waves (scattered pixels): every wave then runs at its
    // own latency and the spills of the 80-register build cost more than the sixth wave hides (512x512 coat: 5 +5 %)
    const int waves = forced ? forced : ((scatter && sc.n_pairs <= 65536u) ? PT_WAVES : PT_BIG_WAVES);
#ifdef PT_DEV_ONE_VARIANT
    (void)waves;
    launch_variant_w<MATS, MEDIUM, PT_BIG_WAVES>(sc, cam, S, fa, fb, stream, grid, scatter);
#else
    if (waves >= PT_BIG_WAVES) launch_variant_w<MATS, MEDIUM, PT_BIG_WAVES>(sc, cam, S, fa, fb, stream, grid, scatter);
    else if (waves <= PT_MIN_WAVES) launch_variant_w<MATS, MEDIUM, PT_MIN_WAVES>(sc, cam, S, fa, fb, stream, grid, scatter);
    else launch_variant_w<MATS, MEDIUM, PT_WAVES>(sc, cam, S, fa, fb, stream, grid, scatter);
#endif
}

// Variant choice = the AOT analogue of the reference's per-scene program build (include/CL/cl_kernel.h):
// material set (LIGHT|DIFF only, or generic) x global medium.
#ifdef PT_PHASE_CLOCKS
void dump_phase_clocks() {
    unsigned long long h[12] = {0};
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase_clocks), sizeof(h)) != hipSuccess) return;
    const char* names[8] = {"A front", "B walk", "B closest_done", "C back", "D shadow walk", "E finish", "iterations", "loop head"};
    double tot = 0;
    for (int k = 0; k < 8; ++k) if (k != 6) tot += (double)h[k];
    for (int k = 0; k < 8; ++k)
        if (k == 6) fprintf(stderr, "phase clocks: %-16s %llu (%.0f cycles each)\n", names[k], h[k], h[k] ? tot / (double)h[k] : 0.0);
        else fprintf(stderr, "phase clocks: %-16s %5.1f %%\n", names[k], 100.0 * (double)h[k] / tot);
    if (h[10]) fprintf(stderr, "phase clocks: waves %llu, iterations per wave mean %.0f max %llu, cycles per wave mean %.0f max %llu\n", h[10],
                       (double)h[6] / (double)h[10], h[8], tot / (double)h[10], h[9]);
#ifdef PT_WALK_STATS
    unsigned long long ws[16] = {0};
    if (hipMemcpyFromSymbol(ws, HIP_SYMBOL(dev::g_walk_stats), sizeof(ws)) == hipSuccess) {
        const char* wn[8] = {"closest box steps", "closest triangle iterations", "any-hit box steps", "any-hit triangle iterations",
                             "A lane_front", "B lane_closest_done", "C lane_back", "E lane_finish"};
        for (int k = 0; k < 8; ++k)
            fprintf(stderr, "phase clocks: %-28s %llu wave-level, %.1f lanes each\n", wn[k], ws[2 * k], ws[2 * k] ? (double)ws[2 * k + 1] / (double)ws[2 * k] : 0.0);
    }
#endif
    if (h[6]) fprintf(stderr, "phase clocks: lanes that have done their frames (or their samples) and wait for the wave: %.1f %% of the lane-iterations\n",
                      100.0 * (double)h[11] / (64.0 * (double)h[6]));
}
#endif

const char* launch_render(const DevScene& sc, const DevCamera& cam, const DevState& S, const FrameArgs& fa, float4* fb,
                          hipStream_t stream) {
    constexpr unsigned LD = PRT_MAT_LIGHT | PRT_MAT_DIFF;
    const unsigned am = sc.active_mats;
#ifdef PT_DEV_ONE_VARIANT                 // development builds (tools/): only the headline variant, compiles in seconds
    if (sc.n_sdfs || sc.has_medium || am != LD) return nullptr;
    launch_variant<LD, false>(sc, cam, S, fa, fb, stream);
    return "render_kernel<LIGHT|DIFF>";
#else
    if (sc.view) {                        // the debug views: generic material set, the default wave count only
        constexpr unsigned V = PT_MATS_VIEW, VS = PT_MATS_VIEW | PT_MATS_SDF;
        bool vscatter;
        const unsigned vgrid = launch_grid(sc, fa, vscatter);
        if (sc.n_sdfs) {
            if (!sc.has_medium) launch_variant_w<VS, false, PT_BIG_WAVES>(sc, cam, S, fa, fb, stream, vgrid, vscatter);
            else launch_variant_w<VS, true, PT_BIG_WAVES>(sc, cam, S, fa, fb, stream, vgrid, vscatter);
        } else {
            if (!sc.has_medium) launch_variant_w<V, false, PT_BIG_WAVES>(sc, cam, S, fa, fb, stream, vgrid, vscatter);
            else launch_variant_w<V, true, PT_BIG_WAVES>(sc, cam, S, fa, fb, stream, vgrid, vscatter);
        }
        return "render_kernel<generic,view>";
    }
    if (sc.n_sdfs) {                      // H_SDF scenes: the generic variants that carry the raymarcher
        if (!sc.has_medium) { launch_variant<PT_MATS_SDF, false>(sc, cam, S, fa, fb, stream); return "render_kernel<generic,sdf>"; }
        launch_variant<PT_MATS_SDF, true>(sc, cam, S, fa, fb, stream);
        return "render_kernel<generic,sdf,medium>";
    }
    if (!sc.has_medium) {
        if (am == LD) { launch_variant<LD, false>(sc, cam, S, fa, fb, stream); return "render_kernel<LIGHT|DIFF>"; }
        launch_variant<0u, false>(sc, cam, S, fa, fb, stream);
        return "render_kernel<generic>";
    }
    if (am == LD) { launch_variant<LD, true>(sc, cam, S, fa, fb, stream); return "render_kernel<LIGHT|DIFF,medium>"; }
    launch_variant<0u, true>(sc, cam, S, fa, fb, stream);
    return "render_kernel<generic,medium>";
#endif
}

unsigned render_tile_count(int width, int rows) {
    return (((unsigned)width + 7u) / 8u) * (((unsigned)rows + 7u) / 8u);
}

void make_dev_camera(const prt_camera& in, DevCamera& out) {
    out = DevCamera{};
    camera_basis(in, out);
}

void launch_state_to_rtd(const DevState& S, prt_path_state* out, size_t n, hipStream_t stream) {
    hipLaunchKernelGGL(state_to_rtd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, S, out, n);
}
void launch_rtd_to_state(const prt_path_state* in, const DevState& S, float4* fb, size_t n, hipStream_t stream) {
    hipLaunchKernelGGL(rtd_to_state, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, in, S, fb, n);
}
void launch_selftest_math(int fn, const float* a, const float* b, float* out, int n, hipStream_t stream) {
    hipLaunchKernelGGL(selftest_math_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, fn, a, b, out, n);
}
void launch_selftest_fn(int fn, const float* params, const float* in, float* out, int n, hipStream_t stream) {
    hipLaunchKernelGGL(selftest_fn_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, fn, params, in, out, n);
}
void launch_tonemap(const float4* fb, unsigned char* out, const FrameArgs& fa, hipStream_t stream) {
    const size_t npix = (size_t)fa.width * (size_t)fa.rows;
    hipLaunchKernelGGL(tonemap_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, stream, fb, reinterpret_cast<uchar4*>(out), fa);
}
void launch_count(const DevState& S, size_t n, unsigned spp, unsigned long long* out3, hipStream_t stream) {
    unsigned blocks = (unsigned)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(count_kernel, dim3(blocks), dim3(256), 0, stream, S, n, spp, out3);
}

}  // namespace prt
