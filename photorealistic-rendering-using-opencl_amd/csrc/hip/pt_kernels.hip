// pt_kernels.hip -- gfx950 kernels of libprt (see pt_device.h for the arithmetic contract).
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

#include <cstdlib>

#include "pt_device.h"
#include "pt_launch.h"

namespace prt {

using namespace dev;

#ifndef PT_MIN_WAVES
#define PT_MIN_WAVES 4      // waves per SIMD the register allocator must leave room for (128 VGPRs)
#endif
#ifndef PT_BLOCK
#define PT_BLOCK 64         // threads per workgroup: ONE wave.  A workgroup's slot (LDS, dispatch) frees only when its last wave
                            // ends, and waves over the mesh run ~3x longer than waves over a wall: one-wave groups +4 % over 256
#endif

// WAVES = waves per SIMD the register allocator leaves room for: 4 (128 VGPRs) is best while the tree sits in L2;
// on a tree of tens of MB every node step is a trip to the Infinity Cache or HBM and 5 waves (96 VGPRs, more
// spills, more latency hidden) win +13 % (871 k triangles); 6 and 8 lose again.
template <unsigned MATS, bool MEDIUM, int WAVES>
__global__ __launch_bounds__(PT_BLOCK, WAVES) void render_kernel(const DevScene sc, const DevCamera cam, const DevState S,
                                                                 const FrameArgs fa, float4* __restrict__ fb) {
    // workgroup tile: 8x8 pixels per wave; 1 wave (PT_BLOCK 64), 1x2 (128) or 2x2 (256) waves per workgroup
    constexpr int TILE_H = PT_BLOCK >= 128 ? 16 : 8, TILE_W = PT_BLOCK / TILE_H, WAVES_X = TILE_W / 8;
    const int tiles_x = (fa.width + TILE_W - 1) / TILE_W;
    const unsigned tile = blockIdx.x * fa.tile_stride + fa.tile_first;
    const int tile_x = (int)(tile % (unsigned)tiles_x), tile_y = (int)(tile / (unsigned)tiles_x);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lx = tile_x * TILE_W + (wave % WAVES_X) * 8 + (lane & 7);
    const int ly = tile_y * TILE_H + (wave / WAVES_X) * 8 + (lane >> 3);
    static_assert(PT_BLOCK == 64, "the end-of-launch ticket counts one wave per workgroup, each with a pixel inside the frame");
    if (lx >= fa.width || ly >= fa.rows) return;                // no barriers in this kernel
    const size_t id = (size_t)ly * (size_t)fa.width + (size_t)lx;
    const int gx = lx;
    const int gy = fa.row0 + (ly / fa.block_rows * fa.n_parts + fa.part) * fa.block_rows + ly % fa.block_rows;

    Path st;
    st.hc.valid = false; st.hc.didHit = false; st.hc.backside = false; st.hc.t = 0.0f; st.hc.mesh_id = -1;
    st.hc.normal = splat(0.0f); st.hc.pos = splat(0.0f);
    {
        const float4 a = S.q0[id], b = S.q1[id], c = S.q2[id], d = S.q3[id];
        const uint4 e = S.q4[id];
        st.origin = F3(a.x, a.y, a.z); st.time = a.w;
        st.dir = F3(b.x, b.y, b.z); st.dist = b.w;
        st.mask = F3(c.x, c.y, c.z); st.total = prt_f2u(c.w);
        st.acc[0] = d.x; st.acc[1] = d.y; st.acc[2] = d.z; st.acc[3] = d.w;
        st.samples = e.x;
        st.diff = e.y & 0xffffu; st.spec = e.y >> 16;
        st.trans = e.z & 0xffffu; st.scatters = e.z >> 16;
        st.wasSpecular = (e.w & 1u) != 0; st.reset = (e.w & 2u) != 0;
    }
    extern __shared__ unsigned lds_stack[];                     // sc.stack_levels x PT_BLOCK, sized by the launch
    TravStack stk;
    stk.lds = lds_stack + threadIdx.x; stk.stride = PT_BLOCK;
    bool ran = false;
    for (unsigned f = 0; f < fa.n_frames; ++f) {
        if (fa.spp_limit && st.reset && st.samples >= fa.spp_limit) break;     // frozen (the "N spp" rule)
        SegCtx c;
        TravRes none;
        none.found = false; none.t = PT_INF; none.th.u = none.th.v = none.th.w = 0.0f; none.th.slot = 0;
        const TravReq rq1 = seg_begin(cam, c, st, gx, gy, fa.width, fa.full_height, fa.first_frame + f,
                                      fa.seed_pairs[2 * f], fa.seed_pairs[2 * f + 1]);
        const TravRes r1 = rq1.want ? walk(sc, false, rq1, stk) : none;
        const TravReq rq2 = seg_after_w1<MATS, MEDIUM>(sc, c, st, r1);
        const TravRes r2 = rq2.want ? walk(sc, false, rq2, stk) : none;
        const TravReq rq3 = seg_after_w2<MATS, MEDIUM>(sc, c, st, rq2.want, r2);
        const bool occluded = rq3.want ? walk(sc, true, rq3, stk).found : false;
        seg_finish(sc, c, st, occluded);
        ran = true;
    }
    if (ran) {
        S.q0[id] = make_float4(st.origin.x, st.origin.y, st.origin.z, st.time);
        S.q1[id] = make_float4(st.dir.x, st.dir.y, st.dir.z, st.dist);
        S.q2[id] = make_float4(st.mask.x, st.mask.y, st.mask.z, prt_u2f(st.total));
        S.q3[id] = make_float4(st.acc[0], st.acc[1], st.acc[2], st.acc[3]);
        S.q4[id] = make_uint4(st.samples, (st.diff & 0xffffu) | (st.spec << 16), (st.trans & 0xffffu) | (st.scatters << 16),
                              (st.wasSpecular ? 1u : 0u) | (st.reset ? 2u : 0u));
        const float ns = (float)st.samples;                                    // write_imagef, main.cl:159
        fb[id] = make_float4(st.acc[0] / ns, st.acc[1] / ns, st.acc[2] / ns, st.acc[3] / ns);
    }
    if (fa.unfinished) {
        const bool unfinished = !(fa.spp_limit && st.reset && st.samples >= fa.spp_limit);
        const unsigned long long m = __ballot(unfinished);
        if (lane == (int)__builtin_ctzll(__ballot(1))) {
            // returning atomic: its value is back only once the add has been performed at the device's coherence point
            const unsigned long long before = m ? atomicAdd(fa.unfinished, (unsigned long long)__popcll(m)) : 0ull;
            if (fa.unfinished_host && before != ~0ull) {       // (never equal: the test orders the ticket behind the add without a
                // fence -- a device-scope fence writes back and invalidates this XCD's L2, 2.5 % when every wave does it)
                // The last wave of the launch hands the total to the host and leaves the counters clean for the next launch
                // (every 8x8 tile holds at least one pixel of the frame, so every wave of the grid gets here).
                if (atomicAdd(fa.unfinished + 1, 1ull) == (unsigned long long)gridDim.x * (PT_BLOCK / 64) - 1ull) {
                    const unsigned long long total = atomicExch(fa.unfinished, 0ull);
                    atomicExch(fa.unfinished + 1, 0ull);
                    *reinterpret_cast<volatile unsigned long long*>(fa.unfinished_host) = total;   // visible to the host at kernel end
                }
            }
        }
    }
}

// ======================================================================================================
// Wavefront pipeline: wf_shade_kernel (all pixels: resume or start a segment, run phases until a walk that
// really enters the tree, then suspend) + wf_trav_kernel (dense waves walk the compacted ray queue, 64 VGPRs
// -> 8 waves/SIMD to hide the walk's latency chain).  One pass = one launch of each.  Pixels are independent
// (seeds depend on the pixel's own frame number), so they are allowed to drift apart: a segment with d deep
// walks simply takes d + 1 passes.
// ======================================================================================================
// What a suspended pixel needs when its walk has been answered depends on where it stopped, so each of
// the three suspension points has its own (small) record: 3 float4 while waiting for W1 (a fresh ray),
// 6 (surface) / 11 (medium scatter) for W2, 5 / 8 for W3.  Everything else of SegCtx is at its initial
// value or dead at that point; the tangent frame is rebuilt from its normal (header.cl:179-192 is pure).
PT_DEV float4 pk(f3 v, float w) { return make_float4(v.x, v.y, v.z, w); }
PT_DEV float4 pku(f3 v, unsigned w) { return make_float4(v.x, v.y, v.z, prt_u2f(w)); }
PT_DEV f3 xyz(float4 v) { return F3(v.x, v.y, v.z); }
#define CTXP(j) wv.ctx[(size_t)(j) * wv.npix + id]
PT_DEV unsigned ctx_bits(const SegCtx& c, bool w2_ran) {
    return (c.e.sampledLobe & 0xffu) | ((unsigned)c.kind << 8) | (c.terminate ? 1u << 16 : 0u) | (c.surface ? 1u << 17 : 0u) |
           (c.done ? 1u << 18 : 0u) | (c.sh ? 1u << 19 : 0u) | (w2_ran ? 1u << 20 : 0u);
}
PT_DEV void ctx_unbits(unsigned b, SegCtx& c, bool& w2_ran) {
    c.e.sampledLobe = b & 0xffu; c.kind = (int)((b >> 8) & 0xffu);
    c.terminate = (b >> 16) & 1u; c.surface = (b >> 17) & 1u; c.done = (b >> 18) & 1u; c.sh = (b >> 19) & 1u; w2_ran = (b >> 20) & 1u;
}
PT_DEV void ctx_store1(const DevWave& wv, size_t id, const SegCtx& c) {
    CTXP(0) = pk(c.ray.origin, c.ray.time);
    CTXP(1) = pku(c.ray.dir, c.rng.s0);
    CTXP(2) = make_float4(prt_u2f(c.rng.s1), 0.f, 0.f, 0.f);
}
PT_DEV void ctx_load1(const DevWave& wv, size_t id, SegCtx& c) {
    seg_ctx_init(c);
    const float4 p0 = CTXP(0), p1 = CTXP(1), p2 = CTXP(2);
    c.ray.origin = xyz(p0); c.ray.time = p0.w;
    c.ray.dir = xyz(p1); c.rng.s0 = prt_f2u(p1.w); c.rng.s1 = prt_f2u(p2.x);
    c.ray.normal = splat(0.0f); c.ray.pos = splat(0.0f); c.ray.t = 0.0f; c.ray.backside = false;
}
template <bool MEDIUM>
PT_DEV void ctx_store2(const DevWave& wv, size_t id, const SegCtx& c, bool w2_ran) {
    CTXP(0) = pk(c.ray.origin, c.ray.time);
    CTXP(1) = pku(c.ray.dir, c.rng.s0);
    CTXP(2) = pku(c.ray.normal, c.rng.s1);
    CTXP(3) = pk(c.e.wi, c.e.pdf);
    CTXP(4) = pku(c.e.weight, ctx_bits(c, w2_ran));
    CTXP(5) = pku(c.e.frame.normal, (unsigned)(c.mesh_id + 1));
    if (MEDIUM && c.kind == K_SCATTER) {
        CTXP(6) = pk(c.ms_p, c.ps.pdf);
        CTXP(7) = pk(c.ps.w, c.sh_tmax);
        CTXP(8) = pk(c.ps.weight, c.ray.t);          // a scatter keeps the path ray's t (it becomes RTD.time)
        CTXP(9) = pk(c.a_vis, 0.f);
        CTXP(10) = pk(c.sh_d, 0.f);
    }
}
template <bool MEDIUM>
PT_DEV void ctx_load2(const DevWave& wv, size_t id, SegCtx& c, bool& w2_ran) {
    seg_ctx_init(c);
    const float4 p0 = CTXP(0), p1 = CTXP(1), p2 = CTXP(2), p3 = CTXP(3), p4 = CTXP(4), p5 = CTXP(5);
    c.ray.origin = xyz(p0); c.ray.time = p0.w;
    c.ray.dir = xyz(p1); c.rng.s0 = prt_f2u(p1.w);
    c.ray.normal = xyz(p2); c.rng.s1 = prt_f2u(p2.w);
    c.ray.pos = splat(0.0f); c.ray.t = 0.0f; c.ray.backside = false;
    c.e.wi = xyz(p3); c.e.pdf = p3.w;
    c.e.weight = xyz(p4); ctx_unbits(prt_f2u(p4.w), c, w2_ran);
    c.e.frame = make_frame(xyz(p5)); c.mesh_id = (int)prt_f2u(p5.w) - 1;
    if (MEDIUM && c.kind == K_SCATTER) {
        const float4 p6 = CTXP(6), p7 = CTXP(7), p8 = CTXP(8), p9 = CTXP(9), p10 = CTXP(10);
        c.ms_p = xyz(p6); c.ps.pdf = p6.w;
        c.ps.w = xyz(p7); c.sh_tmax = p7.w;
        c.ps.weight = xyz(p8); c.ray.t = p8.w;
        c.a_vis = xyz(p9);
        c.sh_d = xyz(p10); c.sh_o = c.ms_p;
    }
}
template <bool MEDIUM>
PT_DEV void ctx_store3(const DevWave& wv, size_t id, const SegCtx& c) {
    CTXP(0) = pk(c.ray.origin, c.ray.time);
    CTXP(1) = pk(c.ray.dir, c.ray.t);
    CTXP(2) = pku(c.e.weight, ctx_bits(c, false));
    CTXP(3) = pku(c.a, c.rng.s0);
    CTXP(4) = pku((MEDIUM && c.kind == K_SCATTER) ? c.a_vis : c.b_vis, c.rng.s1);
    if (MEDIUM && c.kind == K_SCATTER) {
        CTXP(5) = pk(c.ms_p, c.alpha);
        CTXP(6) = pk(c.ps.w, 0.f);
        CTXP(7) = pk(c.ps.weight, 0.f);
    }
}
template <bool MEDIUM>
PT_DEV void ctx_load3(const DevWave& wv, size_t id, SegCtx& c) {
    seg_ctx_init(c);
    bool unused;
    const float4 p0 = CTXP(0), p1 = CTXP(1), p2 = CTXP(2), p3 = CTXP(3), p4 = CTXP(4);
    c.ray.origin = xyz(p0); c.ray.time = p0.w;
    c.ray.dir = xyz(p1); c.ray.t = p1.w;
    c.e.weight = xyz(p2); ctx_unbits(prt_f2u(p2.w), c, unused);
    c.a = xyz(p3); c.rng.s0 = prt_f2u(p3.w);
    c.rng.s1 = prt_f2u(p4.w);
    if (MEDIUM && c.kind == K_SCATTER) {
        c.a_vis = xyz(p4);
        const float4 p5 = CTXP(5), p6 = CTXP(6), p7 = CTXP(7);
        c.ms_p = xyz(p5);
        c.ps.w = xyz(p6);
        c.ps.weight = xyz(p7);
    } else {
        c.b_vis = xyz(p4);
    }
}
#undef CTXP

// does the walk of `rq` get past its first step?  (bvh.cl:144-157 at node 0 -- the same test the walk starts with)
PT_DEV bool walk_is_deep(const DevScene& sc, const TravReq& rq) {
    if (sc.root_is_leaf) return sc.root_leaf_count != 0;
    Ray ray;
    ray.origin = rq.o; ray.dir = rq.d;
    const PairTest pt = test_pair(load_pair(sc.pairs, 0u), ray_pre(ray), rq.tmax);
    return pt.go0 || pt.go1;
}

template <unsigned MATS, bool MEDIUM>
#ifndef PT_SHADE_WAVES
#define PT_SHADE_WAVES 4
#endif
__global__ __launch_bounds__(256, PT_SHADE_WAVES) void wf_shade_kernel(const DevScene sc, const DevCamera cam, const DevState S, const DevWave wv,
                                                       const FrameArgs fa, float4* __restrict__ fb, const unsigned pass) {
    constexpr int TILE_W = 16, WAVES_X = 2;
    const int tiles_x = (fa.width + TILE_W - 1) / TILE_W;
    const int tile_x = (int)(blockIdx.x % (unsigned)tiles_x), tile_y = (int)(blockIdx.x / (unsigned)tiles_x);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lx = tile_x * TILE_W + (wave % WAVES_X) * 8 + (lane & 7);
    const int ly = tile_y * 16 + (wave / WAVES_X) * 8 + (lane >> 3);
    if (lx >= fa.width || ly >= fa.rows) return;
    const size_t id = (size_t)ly * (size_t)fa.width + (size_t)lx;
    const int gx = lx;
    const int gy = fa.row0 + (ly / fa.block_rows * fa.n_parts + fa.part) * fa.block_rows + ly % fa.block_rows;

    uint4 prog = wv.prog[id];
    const uint4 e4 = S.q4[id];
    bool active = prog.x < fa.n_frames;
    if (active && prog.y == 0 && fa.spp_limit && (e4.w & 2u) && e4.x >= fa.spp_limit) active = false;     // frozen
    bool enqueue = false;
    TravReq rq;
    rq.want = false; rq.o = rq.d = splat(0.0f); rq.tmax = PT_INF;
    bool rq_any = false;
    if (active) {
        Path st;
        {
            const float4 a = S.q0[id], b = S.q1[id], c = S.q2[id], d = S.q3[id];
            st.origin = F3(a.x, a.y, a.z); st.time = a.w;
            st.dir = F3(b.x, b.y, b.z); st.dist = b.w;
            st.mask = F3(c.x, c.y, c.z); st.total = prt_f2u(c.w);
            st.acc[0] = d.x; st.acc[1] = d.y; st.acc[2] = d.z; st.acc[3] = d.w;
            st.samples = e4.x;
            st.diff = e4.y & 0xffffu; st.spec = e4.y >> 16;
            st.trans = e4.z & 0xffffu; st.scatters = e4.z >> 16;
            st.wasSpecular = (e4.w & 1u) != 0; st.reset = (e4.w & 2u) != 0;
            const float4 h0 = wv.hc0[id], h1 = wv.hc1[id];
            const unsigned hb = prt_f2u(h1.w);
            st.hc.valid = (hb & 1u) != 0; st.hc.didHit = (hb & 2u) != 0; st.hc.backside = (hb & 4u) != 0;
            st.hc.mesh_id = (int)(hb >> 8) - 1;
            st.hc.t = h0.x; st.hc.normal = F3(h0.y, h0.z, h0.w); st.hc.pos = F3(h1.x, h1.y, h1.z);
        }
        SegCtx c;
        TravRes res;
        res.found = false; res.t = PT_INF; res.th.u = res.th.v = res.th.w = 0.0f; res.th.slot = 0;
        bool w2_ran = false;
        unsigned stage = prog.y;                                // 0 = segment boundary, k = the answer of W_k is in res0/res1
        if (stage != 0) {
            const float4 r = wv.res0[id];
            const unsigned r1 = wv.res1[id];
            res.found = (r1 >> 31) != 0; res.t = r.x; res.th.u = r.y; res.th.v = r.z; res.th.w = r.w; res.th.slot = r1 & 0x7fffffffu;
            if (stage == 1) ctx_load1(wv, id, c);
            else if (stage == 2) ctx_load2<MEDIUM>(wv, id, c, w2_ran);
            else ctx_load3<MEDIUM>(wv, id, c);
        }
        // straight-line phases; `enqueue` = suspended at a walk that really enters the tree
        auto trivial = [&](const TravReq& q) {                  // answer of a walk that is not needed / ends at its first step
            res.found = false; res.t = q.tmax; res.th.u = res.th.v = res.th.w = 0.0f; res.th.slot = 0;
        };
        if (stage == 0) {
            const unsigned f = prog.x;
            rq = seg_begin(cam, c, st, gx, gy, fa.width, fa.full_height, fa.first_frame + f, fa.seed_pairs[2 * f], fa.seed_pairs[2 * f + 1]);
            rq_any = false; stage = 1;
            if (rq.want && walk_is_deep(sc, rq)) { enqueue = true; ctx_store1(wv, id, c); } else trivial(rq);
        }
        if (stage == 1 && !enqueue) {
            rq = seg_after_w1<MATS, MEDIUM>(sc, c, st, res);
            w2_ran = rq.want; rq_any = false; stage = 2;
            if (rq.want && walk_is_deep(sc, rq)) { enqueue = true; ctx_store2<MEDIUM>(wv, id, c, w2_ran); } else trivial(rq);
        }
        if (stage == 2 && !enqueue) {
            rq = seg_after_w2<MATS, MEDIUM>(sc, c, st, w2_ran, res);
            rq_any = true; stage = 3;
            if (rq.want && walk_is_deep(sc, rq)) { enqueue = true; ctx_store3<MEDIUM>(wv, id, c); } else trivial(rq);
        }
        bool finished_segment = false;
        if (stage == 3 && !enqueue) {
            seg_finish(sc, c, st, res.found);
            finished_segment = true;
        }
        if (finished_segment) { prog.x += 1; prog.y = 0; }
        else prog.y = stage;
        // state back to HBM
        S.q0[id] = make_float4(st.origin.x, st.origin.y, st.origin.z, st.time);
        S.q1[id] = make_float4(st.dir.x, st.dir.y, st.dir.z, st.dist);
        S.q2[id] = make_float4(st.mask.x, st.mask.y, st.mask.z, prt_u2f(st.total));
        S.q3[id] = make_float4(st.acc[0], st.acc[1], st.acc[2], st.acc[3]);
        const unsigned flags = (st.wasSpecular ? 1u : 0u) | (st.reset ? 2u : 0u);
        S.q4[id] = make_uint4(st.samples, (st.diff & 0xffffu) | (st.spec << 16), (st.trans & 0xffffu) | (st.scatters << 16), flags);
        wv.hc0[id] = make_float4(st.hc.t, st.hc.normal.x, st.hc.normal.y, st.hc.normal.z);
        wv.hc1[id] = make_float4(st.hc.pos.x, st.hc.pos.y, st.hc.pos.z,
                                 prt_u2f((st.hc.valid ? 1u : 0u) | (st.hc.didHit ? 2u : 0u) | (st.hc.backside ? 4u : 0u) | ((unsigned)(st.hc.mesh_id + 1) << 8)));
        wv.prog[id] = prog;
        if (finished_segment) {
            const float ns = (float)st.samples;                    // write_imagef, main.cl:159
            fb[id] = make_float4(st.acc[0] / ns, st.acc[1] / ns, st.acc[2] / ns, st.acc[3] / ns);
            active = prog.x < fa.n_frames && !(fa.spp_limit && st.reset && st.samples >= fa.spp_limit);
        }
    }
    // compact the suspended rays of this wave into the queue: ballot + one atomic per wave
    const unsigned long long m = __ballot(enqueue);
    if (m) {
        unsigned base = 0;
        const int leader = (int)__builtin_ctzll(m);
        if (lane == leader) base = atomicAdd(&wv.qcount[pass & 1u], (unsigned)__popcll(m));
        base = (unsigned)__builtin_amdgcn_readlane((int)base, leader);
        if (enqueue) {
            const unsigned idx = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
            wv.ray_o[idx] = make_float4(rq.o.x, rq.o.y, rq.o.z, rq.tmax);
            wv.ray_d[idx] = make_float4(rq.d.x, rq.d.y, rq.d.z, prt_u2f((unsigned)id | (rq_any ? 0x80000000u : 0u)));
        }
    }
    if (fa.unfinished) {
        const unsigned long long u = __ballot(active);
        if (u && lane == (int)__builtin_ctzll(u)) atomicAdd(fa.unfinished, (unsigned long long)__popcll(u));
    }
}

#ifndef PT_TRAV_WAVES
#define PT_TRAV_WAVES 8     // 64 VGPRs: the walk is a chain of dependent fetches, occupancy is what hides it
#endif
__global__ __launch_bounds__(256, PT_TRAV_WAVES) void wf_trav_kernel(const DevScene sc, const DevWave wv, const unsigned pass) {
    extern __shared__ unsigned lds_stack[];                     // sc.stack_levels x 256
    TravStack stk;
    stk.lds = lds_stack + threadIdx.x; stk.stride = 256;
    const unsigned n = wv.qcount[pass & 1u];
    if (blockIdx.x == 0 && threadIdx.x == 0) wv.qcount[(pass + 1u) & 1u] = 0;       // next pass appends to the other counter
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const float4 a = wv.ray_o[i], b = wv.ray_d[i];
        const unsigned ob = prt_f2u(b.w);
        TravReq rq;
        rq.want = true; rq.o = F3(a.x, a.y, a.z); rq.d = F3(b.x, b.y, b.z); rq.tmax = a.w;
        const TravRes r = walk(sc, (ob >> 31) != 0, rq, stk);
        const unsigned pix = ob & 0x7fffffffu;
        wv.res0[pix] = make_float4(r.t, r.th.u, r.th.v, r.th.w);
        wv.res1[pix] = r.th.slot | (r.found ? 0x80000000u : 0u);
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

// ---- host-side launchers -------------------------------------------------------------------------------
template <unsigned MATS, bool MEDIUM, int WAVES>
static void launch_variant_w(const DevScene& sc, const DevCamera& cam, const DevState& S, const FrameArgs& fa, float4* fb,
                             hipStream_t stream) {
    constexpr unsigned TILE_H = PT_BLOCK >= 128 ? 16 : 8, TILE_W = PT_BLOCK / TILE_H;
    const unsigned tiles_x = ((unsigned)fa.width + TILE_W - 1) / TILE_W, tiles_y = ((unsigned)fa.rows + TILE_H - 1) / TILE_H;
    const size_t lds = (size_t)sc.stack_levels * PT_BLOCK * sizeof(unsigned);
    if (lds > 65536u)        // only a 4-wave build with a tree that fills the reference's 64-entry stack to the brim
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&render_kernel<MATS, MEDIUM, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const unsigned n_tiles = tiles_x * tiles_y;
    if (fa.tile_first >= n_tiles) return;
    const unsigned grid = (n_tiles - fa.tile_first + fa.tile_stride - 1) / fa.tile_stride;      // tiles of this sub-part
    hipLaunchKernelGGL((render_kernel<MATS, MEDIUM, WAVES>), dim3(grid), dim3(PT_BLOCK), lds, stream, sc, cam, S, fa, fb);
}
template <unsigned MATS, bool MEDIUM>
static void launch_variant(const DevScene& sc, const DevCamera& cam, const DevState& S, const FrameArgs& fa, float4* fb,
                           hipStream_t stream) {
    const char* e_waves = std::getenv("PRT_WAVES");                       // 4 / 5: override (tests, experiments)
    const int forced = e_waves ? std::atoi(e_waves) : 0;
    // 5 waves where latency rules: the node records alone exceed one XCD's L2, or the scene raymarches SDFs (+11 %)
    const bool big = forced ? forced >= 5 : (sc.n_pairs > 65536u || sc.n_sdfs != 0u);
    if (big) launch_variant_w<MATS, MEDIUM, 5>(sc, cam, S, fa, fb, stream);
    else launch_variant_w<MATS, MEDIUM, PT_MIN_WAVES>(sc, cam, S, fa, fb, stream);
}

// Variant choice = the AOT analogue of the reference's per-scene program build (include/CL/cl_kernel.h):
// material set (LIGHT|DIFF only, or generic) x global medium.
const char* launch_render(const DevScene& sc, const DevCamera& cam, const DevState& S, const FrameArgs& fa, float4* fb,
                          hipStream_t stream) {
    constexpr unsigned LD = PRT_MAT_LIGHT | PRT_MAT_DIFF;
    const unsigned am = sc.active_mats;
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
}

template <unsigned MATS, bool MEDIUM>
static void launch_wf_variant(const DevScene& sc, const DevCamera& cam, const DevState& S, const DevWave& wv, const FrameArgs& fa,
                              float4* fb, unsigned pass, hipStream_t stream) {
    const unsigned tiles_x = ((unsigned)fa.width + 15u) / 16u, tiles_y = (unsigned)((fa.rows + 15) >> 4);
    hipLaunchKernelGGL((wf_shade_kernel<MATS, MEDIUM>), dim3(tiles_x * tiles_y), dim3(256), 0, stream, sc, cam, S, wv, fa, fb, pass);
}
void launch_wf_pass(const DevScene& sc, const DevCamera& cam, const DevState& S, const DevWave& wv, const FrameArgs& fa, float4* fb,
                    unsigned pass, unsigned trav_blocks, hipStream_t stream) {
    constexpr unsigned LD = PRT_MAT_LIGHT | PRT_MAT_DIFF;
    const unsigned am = sc.active_mats;
    if (sc.n_sdfs) {
        if (!sc.has_medium) launch_wf_variant<PT_MATS_SDF, false>(sc, cam, S, wv, fa, fb, pass, stream);
        else launch_wf_variant<PT_MATS_SDF, true>(sc, cam, S, wv, fa, fb, pass, stream);
    } else if (!sc.has_medium) {
        if (am == LD) launch_wf_variant<LD, false>(sc, cam, S, wv, fa, fb, pass, stream);
        else launch_wf_variant<0u, false>(sc, cam, S, wv, fa, fb, pass, stream);
    } else {
        if (am == LD) launch_wf_variant<LD, true>(sc, cam, S, wv, fa, fb, pass, stream);
        else launch_wf_variant<0u, true>(sc, cam, S, wv, fa, fb, pass, stream);
    }
    const size_t lds = (size_t)sc.stack_levels * 256 * sizeof(unsigned);
    if (lds > 65536u) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wf_trav_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(wf_trav_kernel, dim3(trav_blocks), dim3(256), lds, stream, sc, wv, pass);
}

unsigned render_tile_count(int width, int rows) {
    constexpr unsigned TILE_H = PT_BLOCK >= 128 ? 16 : 8, TILE_W = PT_BLOCK / TILE_H;
    return (((unsigned)width + TILE_W - 1) / TILE_W) * (((unsigned)rows + TILE_H - 1) / TILE_H);
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
