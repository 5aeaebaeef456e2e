// pt_pool.h -- render_kernel_rp: the lane machine with the DEEP WALKS HANDED TO WALKER WAVES through a ray pool in LDS (round 4).
//
// What binds render_kernel: every wave runs the shading phases A / C / E and the two walk loops B / D for all its
// lanes, and the lanes disagree -- a box step of a walk loop runs at 11 lanes of 64 on cornell, because only the rays that go deep
// into the tree are in it (14 % of the rays, 63 % of the steps), and the lanes that own them are held for several iterations of the
// wave, a few steps per iteration.  Here a workgroup is PT_RP_SHADERS shading waves + PT_RP_WALKERS walker waves:
//
//   * a shading wave is render_kernel as it was -- same pixel mapping, same frame loop, same lane record in registers -- except that of
//     a walk it takes only the step at the root (scalar loads, every lane of the phase).  A lane whose ray has to go deeper POSTS the
//     ray {origin, dir, limit, closest / any hit} into the slot of the pool that belongs to it (one per shading lane: nothing to
//     allocate) and waits for the answer {t, u, v, slot, found} with its context where it is, as a cut-off lane does in render_kernel;
//   * a walker wave does nothing but walk: its idle lanes take posted rays from the pool (whenever a quarter of them are idle: the
//     persistent while-while scheme), run bvh.cl's traversal with the same walk_begin / walk_box / walk_tri (same boxes, same order,
//     same triangle rule), write the answer into the slot and mark it done.  Rays of several waves fill the walker's lanes (24 - 42 of
//     64, measured) where one wave's own deep rays filled a sixth of them.
//
// MEASURED SLOWER than render_kernel on every workload (DESIGN.md s4 "Round 4", profiles/r04_regrouping_experiments.txt: 9.85 against 13.1 G
// segments/s on the headline config with 3 + 1 waves, 2.06 against 3.75 through the 871 k-triangle mesh): a posted ray is away for the latency
// of its chain of dependent loads while the shading iteration, rid of its walk loops, got three times shorter -- a third of the shading lanes
// wait.  Off by default (prt_set_option "pool"); kept, tested, as the measured answer to "why not walker waves".
//
// Results cannot depend on any of this: the walk of a ray is the same function whoever runs it (tests/test_gpu_parity.py renders the
// goldens through this kernel: prt_set_option "pool").
//
// Protocol (LDS of one workgroup): slot s = shading thread s.  The owner writes the ray, then sets its bit in the wave's `pend` mask
// (one atomic OR per wave); the walker that serves the wave (walker k serves the shading waves k, k + W, ...: one consumer per mask)
// snapshots the mask, matches its idle lanes to set bits and clears the bits it takes with one atomic AND; the answer goes into the ray's own words, then
// done[s] = 1 (release); the owner polls done[s] once per iteration of its wave.  Nobody waits inside an iteration.
// The walkers leave when every shading wave of the workgroup has left (no ray can be outstanding then).  A watchdog bounds the
// iterations of every wave: a scheduling bug must end in wrong pixels, never in a hung GPU.
#pragma once
#include "pt_device.h"
#include "pt_launch.h"

namespace prt {

using namespace dev;

#ifndef PT_RP_SHADERS
#define PT_RP_SHADERS 3
#endif
#ifndef PT_RP_WALKERS
#define PT_RP_WALKERS 1
#endif
#ifndef PT_RP_REFILL
#define PT_RP_REFILL 16                   // a walker looks for posted rays once this many of its lanes are idle
#endif
#ifndef PT_RP_STEPS
#define PT_RP_STEPS 4                     // walk steps of a walker between two looks at the pool
#endif
#define PT_RP_WAVES (PT_RP_SHADERS + PT_RP_WALKERS)
#define PT_RP_BLOCK (64 * PT_RP_WAVES)
#define PT_RP_SLOTS (64 * PT_RP_SHADERS)

#ifdef PT_POOL_STATS           // development builds (tools/build_variant.sh ... -DPT_POOL_STATS, unity build): what the waves of the kernel do
__device__ unsigned long long g_pool_stats[32];
#define PT_PSTAT(k, v) (pstat_[k] += (unsigned long long)(v))
#else
#define PT_PSTAT(k, v) do { } while (0)
#endif

// LDS layout of a workgroup, in dwords (dynamic shared memory)
struct RpLayout { unsigned done, pend, ctl, rows, dummy, wstack, ray, total; };
__host__ __device__ inline RpLayout rp_layout(unsigned stack_levels) {
    RpLayout l;
    l.done = 0;                                             // [slot]: the answer is in the slot
    l.pend = l.done + PT_RP_SLOTS;                          // [shading wave] 64-bit masks of posted rays
    l.ctl = l.pend + 2 * PT_RP_SHADERS;                     // [0] shading waves that have left, [1] abort
    l.rows = l.ctl + 4;                                     // [walker] 64 words for the matching
    l.rows += l.rows & 1u;
    l.dummy = l.rows + PT_RP_WALKERS * 64;                  // [shading wave] the one stack level the step at the root may write
    l.wstack = l.dummy + PT_RP_SHADERS * 64;                // [walker][level][lane]
    l.ray = l.wstack + PT_RP_WALKERS * stack_levels * 64;   // [word 0..7][slot]
    l.total = l.ray + 8 * PT_RP_SLOTS;
    return l;
}

PT_DEV unsigned rp_rank(unsigned long long m) {            // set bits of m below this lane
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}
PT_DEV unsigned rp_ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
PT_DEV void rp_st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

// ---- a walker wave ------------------------------------------------------------------------------------------------------------------
PT_DEV void rp_walker(const DevScene& sc, const FrameArgs& fa, unsigned* lds, const RpLayout& lay, const unsigned wk, const unsigned lane) {
    unsigned* const done = lds + lay.done;
    unsigned long long* const pend = reinterpret_cast<unsigned long long*>(lds + lay.pend);
    unsigned* const ctl = lds + lay.ctl;
    unsigned* const row = lds + lay.rows + wk * 64u;
    unsigned* const rays = lds + lay.ray;
    TravStack stk;
    stk.lds = lds + lay.wstack + wk * sc.stack_levels * 64u + lane; stk.stride = 64u;
    const unsigned TQ = fa.tri_sixteenths;
    bool active = false, any_hit = false;
    unsigned slot = 0u;
    Ray wr;
    wr.origin = wr.dir = wr.normal = wr.pos = splat(0.0f); wr.t = 0.0f; wr.backside = false; wr.time = 0.0f;
    RayPre p = ray_pre(wr);
    WalkState w;
    w.node = 0u; w.sp = 0u; w.pend_count = 0u; w.t = 0.0f; w.u = w.v = 0.0f; w.slot = 0u; w.found = false; w.done = true; w.last = false; w.pend_first = 0u;
    unsigned idle_spins = 0u;
#ifdef PT_POOL_STATS
    unsigned long long pstat_[32];
    for (int k = 0; k < 32; ++k) pstat_[k] = 0ull;
#endif
#ifdef PT_RP_PRIO
    __builtin_amdgcn_s_setprio(PT_RP_PRIO);                    // the walkers are what the shading lanes wait for
#endif
    for (;;) {
        const unsigned long long actm = __ballot(active);
        if (!actm && (__builtin_amdgcn_readfirstlane(rp_ld(ctl)) >= PT_RP_SHADERS || rp_ld(ctl + 1))) break;
        PT_PSTAT(16, 1);
        if (64u - (unsigned)__popcll(actm) >= PT_RP_REFILL) {
            PT_PSTAT(17, 1);
            unsigned taken = ~0u;
#pragma unroll
            for (unsigned sw = 0; sw < PT_RP_SHADERS; ++sw) {      // (straight-line rounds, one per shading wave's mask)
                const unsigned long long idlem = __ballot(!active && taken == ~0u);
                // (every shading wave's mask has ONE consumer: walker wk serves the shading waves sw = wk, wk + W, ...)
                const unsigned long long m = (sw % PT_RP_WALKERS == wk) ? __hip_atomic_load(pend + sw, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) : 0ull;
                if (m && idlem) {
                    const unsigned n_take = min((unsigned)__popcll(m), (unsigned)__popcll(idlem));
                    const bool offered = ((m >> lane) & 1ull) && rp_rank(m) < n_take;       // the lowest n_take posted rays of this mask
                    const unsigned long long want = __ballot(offered);
                    if (offered) row[rp_rank(m)] = lane;
                    unsigned long long old = 0ull;
                    if (lane == 0u) old = __hip_atomic_fetch_and(pend + sw, ~want, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const unsigned long long got = (((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(old >> 32)) << 32) |
                                                    (unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)old)) & want;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    const unsigned r = rp_rank(idlem);
                    if (!active && taken == ~0u && r < n_take) {
                        const unsigned j = row[r];
                        if ((got >> j) & 1ull) taken = sw * 64u + j;               // (else another walker was quicker)
                    }
                    __builtin_amdgcn_wave_barrier();                               // (the row is rewritten in the next round)
                }
            }
            PT_PSTAT(18, __popcll(__ballot(taken != ~0u)));
            if (taken != ~0u) {
                const unsigned* rr = rays + taken;
                wr.origin = F3(prt_u2f(rr[0]), prt_u2f(rr[PT_RP_SLOTS]), prt_u2f(rr[2 * PT_RP_SLOTS]));
                wr.dir = F3(prt_u2f(rr[3 * PT_RP_SLOTS]), prt_u2f(rr[4 * PT_RP_SLOTS]), prt_u2f(rr[5 * PT_RP_SLOTS]));
                wr.t = prt_u2f(rr[6 * PT_RP_SLOTS]);
                any_hit = rr[7 * PT_RP_SLOTS] != 0u;
                p = ray_pre(wr);
                walk_begin(sc, any_hit, wr, wr.t, p, w, stk);
                active = true; slot = taken;
            }
        }
        if (!__ballot(active)) {                               // nothing to walk: leave the issue slots to the shading waves
            __builtin_amdgcn_s_sleep(16);
            PT_PSTAT(19, 1);
            if (++idle_spins > (1u << 24)) { if (lane == 0u) rp_st(ctl + 1, 1u); break; }
        } else {
            idle_spins = 0u;
            for (int k = 0; k < PT_RP_STEPS; ++k) {
                const bool go = active && !w.done;
                if (!__ballot(go)) break;
                PT_PSTAT(20, 1); PT_PSTAT(21, __popcll(__ballot(go))); PT_PSTAT(22, __popcll(__ballot(go && !w.pend_count)));
                if (go) {
                    if (!w.pend_count) walk_box(sc, any_hit, wr, p, w, stk);
                    const bool pending = w.pend_count != 0u;
                    const unsigned n_in = (unsigned)__popcll(__ballot(1)), n_pend = (unsigned)__popcll(__ballot(pending));
                    if (pending && n_pend * 16u >= n_in * TQ) { PT_PSTAT(23, 1); walk_tri(sc, any_hit, wr, w); }
                }
            }
            if (active && w.done) {                            // the answer goes into the ray's own words
                unsigned* rr = rays + slot;
                rr[0] = prt_f2u(w.t); rr[PT_RP_SLOTS] = prt_f2u(w.u); rr[2 * PT_RP_SLOTS] = prt_f2u(w.v);
                rr[3 * PT_RP_SLOTS] = (unsigned)w.slot | ((unsigned)w.found << 29);
                rp_st(lds + lay.done + slot, 1u);
                active = false;
            }
        }
    }
    (void)done;
#ifdef PT_POOL_STATS
    if (lane == 0u) for (int k = 16; k < 32; ++k) if (pstat_[k]) atomicAdd(&g_pool_stats[k], pstat_[k]);
#endif
}

// ---- the kernel: shading waves = render_kernel (pt_render.h) with the walk loops replaced by post / poll -------------------------------
// n_waves: tiles (= shading waves with pixels) of the launch; the grid is ceil(n_waves / PT_RP_SHADERS) workgroups
template <unsigned MATS, bool MEDIUM, int WAVES, bool ORDER = false>
__global__ __launch_bounds__(PT_RP_BLOCK, WAVES) void render_kernel_rp(const DevScene sc, const DevCamera cam, const DevState S, const FrameArgs fa,
                                                                       float4* __restrict__ fb, const unsigned n_waves) {
    extern __shared__ unsigned rp_lds[];
    const RpLayout lay = rp_layout(sc.stack_levels);
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (unsigned i = threadIdx.x; i < lay.rows; i += PT_RP_BLOCK) rp_lds[i] = 0u;          // done, pend, ctl
    __syncthreads();
    if (wv >= PT_RP_SHADERS) { rp_walker(sc, fa, rp_lds, lay, wv - PT_RP_SHADERS, lane); return; }

    unsigned* const ctl = rp_lds + lay.ctl;
    const unsigned slot = wv * 64u + lane;                      // this lane's slot of the pool
    unsigned* const my_done = rp_lds + lay.done + slot;
    unsigned* const my_ray = rp_lds + lay.ray + slot;
    unsigned long long* const my_pend = reinterpret_cast<unsigned long long*>(rp_lds + lay.pend) + wv;
    const unsigned g = blockIdx.x * PT_RP_SHADERS + wv;         // this wave among the launch's shading waves
    const bool wave_ok = g < n_waves;
    const int tiles_x = (fa.width + 7) / 8;
    const unsigned tile_k = (ORDER && !fa.scatter && fa.tile_order && wave_ok) ? fa.tile_order[g] : g;
    unsigned iterations = 0u;
    const unsigned vpix = fa.scatter ? lane * n_waves + g : tile_k * 64u + lane;
    const unsigned tile = (vpix >> 6) * fa.tile_stride + fa.tile_first;
    const int tl = (int)(vpix & 63u);
    const int tile_x = (int)(tile % (unsigned)tiles_x), tile_y = (int)(tile / (unsigned)tiles_x);
    const int lx = tile_x * 8 + (tl & 7);
    const int ly = tile_y * 8 + (tl >> 3);
    const bool in_frame = wave_ok && lx < fa.width && ly < fa.rows;
    const size_t id = in_frame ? (size_t)ly * (size_t)fa.width + (size_t)lx : 0;
    const int gx = lx;
    const int gy = fa.row0 + (ly / fa.block_rows * fa.n_parts + fa.part) * fa.block_rows + ly % fa.block_rows;

    Lane L;
    lane_init(L);
    if (!in_frame) { L.f = 0xffffffffu; L.reset = true; L.samples = 0xffffffffu; L.wasSpecular = false; }
    else {
        const float4 a = S.q0[id], b = S.q1[id], c = S.q2[id], d = S.q3[id];
        const uint4 e = S.q4[id];
        L.origin = F3(a.x, a.y, a.z); L.t = a.w;
        L.dir = F3(b.x, b.y, b.z); L.time = b.w;
        L.mask = F3(c.x, c.y, c.z); L.total = prt_f2u(c.w);
        L.acc[0] = d.x; L.acc[1] = d.y; L.acc[2] = d.z; L.acc[3] = d.w;
        L.samples = e.x;
        L.diff = e.y & 0xffffu; L.spec = e.y >> 16;
        L.trans = e.z & 0xffffu; L.scatters = e.z >> 16;
        L.wasSpecular = (e.w & 1u) != 0; L.reset = (e.w & 2u) != 0;
        L.f = fa.run_ahead ? e.w >> 2 : 0u;
    }
    TravStack stk;
    stk.lds = rp_lds + lay.dummy + slot; stk.stride = 64u;      // the step at the root pushes at most one entry
    const unsigned long long max_iters = 1024ull * 256ull * ((unsigned long long)fa.seed_frames + 1024ull);
    unsigned long long iters = 0ull;
#ifdef PT_POOL_STATS
    unsigned long long pstat_[32];
    for (int k = 0; k < 32; ++k) pstat_[k] = 0ull;
#endif
    for (;;) {
        const bool runnable = lane_runnable(fa, L, __any(lane_owes_frames(fa, L)));
        // ONE exit: every lane has done its frames (or is frozen) -- no ray is outstanding then -- or the watchdog says so.  (A second
        // `break` behind the first one cost 200 B of scratch per lane: two exit edges, two copies of the lane record to merge.)
        const bool dog = iters > max_iters || rp_ld(ctl + 1) != 0u;
        if (!__any(runnable || L.stage != ST_READY) || dog) break;
        if (ORDER) ++iterations;
        PT_PSTAT(0, 1); PT_PSTAT(3, __popcll(__ballot(runnable))); PT_PSTAT(8, __popcll(__ballot(L.posted)));
        // answers of the walkers
        bool progress = false;
        if (__ballot(L.posted)) {
            if (L.posted && rp_ld(my_done)) {
                const unsigned sf = my_ray[3 * PT_RP_SLOTS];
                if (L.stage == ST_WALKC) { L.w.t = prt_u2f(my_ray[0]); L.w.u = prt_u2f(my_ray[PT_RP_SLOTS]); L.w.v = prt_u2f(my_ray[2 * PT_RP_SLOTS]); L.w.slot = sf & 0x1fffffffu; }
                L.w.found = (sf >> 29) & 1u;                    // (an any-hit walk leaves (t, u, v) alone: Lane::a lives there)
                L.w.done = true; L.w.pend_count = 0u;
                L.posted = false;
                *my_done = 0u;
                progress = true;
            }
        }
        PT_PSTAT(7, __popcll(__ballot(progress)));
        // (the watchdog counts the iterations that did something and, 1024 to one, those in which every lane that is left waited for a walker)
        if (__any(runnable || progress)) iters += 1024ull;
        else { __builtin_amdgcn_s_sleep(8); iters += 1ull; }
        if (runnable) lane_front<MATS, MEDIUM>(sc, cam, fa, L, gx, gy);                                      // A
        {                                                                                                     // B: the step at the root only
            const bool walking = L.stage == ST_WALKC;
            if (walking && L.fresh) {
                const Ray wr = lane_closest_ray<MEDIUM>(L);
                const RayPre p = ray_pre(wr);
                walk_begin(sc, false, wr, PT_INF, p, L.w, stk);
                L.fresh = false;
            }
            if (walking && L.w.done) lane_closest_done<MATS, MEDIUM>(sc, L);
        }
        if (L.stage == ST_BACK) lane_back<MATS, MEDIUM>(sc, L);                                               // C
        {                                                                                                     // D
            const bool walking = L.stage == ST_WALKS;
            if (walking && L.fresh) {
                const Ray wr = lane_shadow_ray<MEDIUM, (MATS & PT_MATS_ENVIS) != 0>(L);
                const RayPre p = ray_pre(wr);
                walk_begin(sc, true, wr, wr.t, p, L.w, stk);
                L.fresh = false;
            }
            if (walking && L.w.done) { L.occluded = L.w.found; L.stage = ST_FINISH; }
        }
        if (L.stage == ST_FINISH) lane_finish<MATS, MEDIUM>(sc, L);                                           // E
        // rays that have to go deeper than the root: to the walkers
        {
            const bool closest = L.stage == ST_WALKC;
            const bool post = (closest || L.stage == ST_WALKS) && !L.fresh && !L.w.done && !L.posted;
            const unsigned long long pm = __ballot(post);
            if (pm) {
                PT_PSTAT(4, __popcll(pm));
                if (post) {
                    const Ray wr = closest ? lane_closest_ray<MEDIUM>(L) : lane_shadow_ray<MEDIUM, (MATS & PT_MATS_ENVIS) != 0>(L);
                    my_ray[0] = prt_f2u(wr.origin.x); my_ray[PT_RP_SLOTS] = prt_f2u(wr.origin.y); my_ray[2 * PT_RP_SLOTS] = prt_f2u(wr.origin.z);
                    my_ray[3 * PT_RP_SLOTS] = prt_f2u(wr.dir.x); my_ray[4 * PT_RP_SLOTS] = prt_f2u(wr.dir.y); my_ray[5 * PT_RP_SLOTS] = prt_f2u(wr.dir.z);
                    my_ray[6 * PT_RP_SLOTS] = prt_f2u(closest ? PT_INF : wr.t);
                    my_ray[7 * PT_RP_SLOTS] = closest ? 0u : 1u;
                    L.posted = true;
                }
                if (lane == (unsigned)__builtin_ctzll(pm)) __hip_atomic_fetch_or(my_pend, pm, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    if (iters > max_iters && lane == 0u) rp_st(ctl + 1, 1u);                                             // (the watchdog: everybody out)
    if (lane == 0u) __hip_atomic_fetch_add(ctl, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);      // this shading wave has left
#ifdef PT_POOL_STATS
    if (lane == 0u) for (int k = 0; k < 16; ++k) if (pstat_[k]) atomicAdd(&g_pool_stats[k], pstat_[k]);
#endif
    if (in_frame && L.f) {
        const bool frozen = fa.spp_limit && L.reset && L.samples >= fa.spp_limit;
        const unsigned frames_ahead = (!frozen && L.f > fa.n_frames) ? L.f - fa.n_frames : 0u;
        S.q0[id] = make_float4(L.origin.x, L.origin.y, L.origin.z, L.t);
        S.q1[id] = make_float4(L.dir.x, L.dir.y, L.dir.z, L.time);
        S.q2[id] = make_float4(L.mask.x, L.mask.y, L.mask.z, prt_u2f(L.total));
        S.q3[id] = make_float4(L.acc[0], L.acc[1], L.acc[2], L.acc[3]);
        S.q4[id] = make_uint4(L.samples, (L.diff & 0xffffu) | (L.spec << 16), (L.trans & 0xffffu) | (L.scatters << 16),
                              (L.wasSpecular ? 1u : 0u) | (L.reset ? 2u : 0u) | (frames_ahead << 2));
        const float ns = (MATS & PT_MATS_VIEW) ? 1.0f : (float)L.samples;
        fb[id] = make_float4(L.acc[0] / ns, L.acc[1] / ns, L.acc[2] / ns, L.acc[3] / ns);
    }
    if (ORDER && !fa.scatter && fa.tile_cost && lane == 0u && wave_ok) fa.tile_cost[tile_k] = iterations;
    if (fa.unfinished) {                                         // as render_kernel: the last shading wave of the launch reports
        const bool unfinished = in_frame && !(fa.spp_limit && L.reset && L.samples >= fa.spp_limit);
        const unsigned long long m = __ballot(unfinished);
        if (lane == 0u) {
            const unsigned long long before = m ? atomicAdd(fa.unfinished, (unsigned long long)__popcll(m)) : 0ull;
            if (fa.unfinished_host && before != ~0ull) {
                if (atomicAdd(fa.unfinished + 1, 1ull) == (unsigned long long)gridDim.x * PT_RP_SHADERS - 1ull) {
                    const unsigned long long tot = atomicExch(fa.unfinished, 0ull);
                    atomicExch(fa.unfinished + 1, 0ull);
                    *reinterpret_cast<volatile unsigned long long*>(fa.unfinished_host) = tot;
                }
            }
        }
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------------------------
template <unsigned MATS, bool MEDIUM, int WAVES, bool ORDER>
static bool launch_rp(const DevScene& sc, const DevCamera& cam, const DevState& S, const FrameArgs& fa, float4* fb, hipStream_t stream, unsigned n_waves) {
    const size_t lds = (size_t)rp_layout(sc.stack_levels).total * sizeof(unsigned);
    if (lds > 160u * 1024u) return false;
    static size_t lds_attr = 0;
    if (lds > 65536u && lds > lds_attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&render_kernel_rp<MATS, MEDIUM, WAVES, ORDER>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        lds_attr = lds;
    }
    const unsigned grid = (n_waves + PT_RP_SHADERS - 1u) / PT_RP_SHADERS;
    if (!grid) return true;
    hipLaunchKernelGGL((render_kernel_rp<MATS, MEDIUM, WAVES, ORDER>), dim3(grid), dim3(PT_RP_BLOCK), lds, stream, sc, cam, S, fa, fb, n_waves);
    return true;
}

}  // namespace prt
