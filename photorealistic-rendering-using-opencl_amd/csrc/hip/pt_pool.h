// pt_pool.h -- render_kernel_pool: the lane machine with the DEEP WALKS TAKEN OUT OF THE SHADING WAVES (round 4).
//
// What binds render_kernel: every wave runs the shading phases A / C / E and the two walk loops B / D for all
// its lanes, and the lanes disagree -- a box step of a walk loop runs at 11 lanes of 64 on cornell, the shading phases enter with
// 40 - 47, because the lanes whose rays go deep into the tree (14 % of the rays, 63 % of the steps) hold their lanes for several
// iterations of the wave.  A lane is bound to its pixel only through its registers, so here it is not bound at all:
//
//   * a workgroup is PT_POOL_SHADERS shading waves + PT_POOL_WALKERS walker waves around a POOL of parked pixel contexts in LDS;
//   * a shading wave runs the lane machine's phases as before, but of a walk only the step at the root (scalar loads, every lane
//     of the phase).  A lane whose ray has to go deeper PARKS its whole context -- the Lane record, the pixel it belongs to, the
//     ray -- in a pool slot and is EMPTY; an empty lane takes over a context whose walk has come back (any pixel of the
//     workgroup: the context carries its pixel index) and goes on with it in the same iteration;
//   * a walker wave does nothing but walk: its lanes claim waiting rays from the pool (refilled whenever a quarter of them are idle:
//     the persistent while-while scheme), run bvh.cl's traversal exactly as render_kernel does (walk_begin / walk_box / walk_tri of
//     pt_device.h: same boxes, same order, same triangle rule), write {t, u, v, slot, found} into the slot and mark it DONE.
//
// Results cannot depend on any of this: a pixel's path is a function of its own (x, y, frame, seed pair) only (main.cl:108-109) and the
// walk of a ray is the same function whoever runs it -- the goldens must come out bit for bit (tests/test_gpu_parity.py runs them
// through this kernel with prt_set_option "pool").
//
// Pool protocol (all in LDS, one workgroup): slot state FREE -> FILLING -> WAIT -> WALKING -> DONE -> TAKING -> FREE; every transition
// out of a shared state is an atomic compare-and-swap, the payload is written before the state that publishes it (release / acquire
// at workgroup scope).  Matching k claimants to k candidate slots inside a wave: the candidates' owners write their slot numbers into
// a row of LDS at their rank, claimant r reads entry r.  No wave ever waits for another inside an iteration: a lane that finds the pool
// full keeps its context and tries again in the next one; the walkers drain it whatever the shading waves do.
// Termination: every context is stored exactly once (ctl[STORED] counts), and every wave leaves when that count is the workgroup's total.
// A watchdog bounds the iterations of every wave (a scheduling bug must end in wrong pixels, never in a hung GPU).
#pragma once
#include "pt_device.h"
#include "pt_launch.h"

namespace prt {

using namespace dev;

#ifndef PT_POOL_SHADERS
#define PT_POOL_SHADERS 5
#endif
#ifndef PT_POOL_WALKERS
#define PT_POOL_WALKERS 1
#endif
#ifndef PT_POOL_ROUNDS
#define PT_POOL_ROUNDS 2                  // pool slots per workgroup = 64 x this
#endif
#ifndef PT_POOL_REFILL
#define PT_POOL_REFILL 16                 // a walker looks for waiting rays once this many of its lanes are idle
#endif
#ifndef PT_POOL_OCC
#define PT_POOL_OCC 6                     // waves per SIMD the register allocator leaves room for
#endif
#define PT_POOL_SLOTS (64 * PT_POOL_ROUNDS)
#define PT_POOL_WAVES (PT_POOL_SHADERS + PT_POOL_WALKERS)
#define PT_POOL_BLOCK (64 * PT_POOL_WAVES)

enum { ST_EMPTY = 5 };                     // a lane without a context (beside the ST_* of pt_device.h)
enum { PS_FREE = 0, PS_FILLING = 1, PS_WAIT = 2, PS_WALKING = 3, PS_DONE = 4, PS_TAKING = 5 };
enum { PC_STORED = 0, PC_OWING = 1, PC_TOTAL = 2, PC_ABORT = 3, PC_WORDS = 16 };

constexpr int PT_LANE_DW = 32 + 12 + 1 + 3;          // words of a parked context (ctx_store: 32 floats, 10 integers, the pixel (2), ps_pdf, view_n)
// a slot: the Lane record, the pixel (index into the state planes; global x | y << 16), and eight words that are the ray on the way to
// the walker {origin, dir, tmax, any-hit} and its answer {t, u, v, slot | found << 29} on the way back
constexpr int PT_CTX_DW = PT_LANE_DW + 8;
constexpr int PT_CTX_RAY = PT_LANE_DW;

// LDS layout of a workgroup, in dwords (dynamic shared memory, sized by pool_lds_bytes)
struct PoolLayout {
    unsigned st, ctl, rows, dummy, wstack, ctx, total;
};
__host__ __device__ inline PoolLayout pool_layout(unsigned stack_levels) {
    PoolLayout l;
    l.st = 0;
    l.ctl = l.st + PT_POOL_SLOTS;
    l.rows = l.ctl + PC_WORDS;
    l.dummy = l.rows + PT_POOL_WAVES * 64;
    l.wstack = l.dummy + PT_POOL_SHADERS * 64;
    l.ctx = l.wstack + PT_POOL_WALKERS * stack_levels * 64;
    l.total = l.ctx + PT_CTX_DW * PT_POOL_SLOTS;
    return l;
}

#ifdef PT_POOL_STATS           // development builds (tools/build_variant.sh ... -DPT_POOL_STATS, unity build): what the waves of the pool kernel do
__device__ unsigned long long g_pool_stats[32];
#define PT_PSTAT(k, v) (pstat_[k] += (unsigned long long)(v))
#else
#define PT_PSTAT(k, v) do { } while (0)
#endif
PT_DEV unsigned pool_rank(unsigned long long m) {      // set bits of m below this lane
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}
PT_DEV unsigned pool_ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
PT_DEV void pool_st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

// Every lane with `need` tries to move ONE slot from state `from` to state `to` and gets its index (or ~0u: none left, or another
// wave was quicker -- the caller tries again later).  Wave-uniform control flow; `row` = this wave's 64 words of LDS.
PT_DEV unsigned pool_claim(unsigned* st, unsigned* row, const bool need, const unsigned from, const unsigned to) {
    unsigned got = ~0u;
    const unsigned lane = threadIdx.x & 63u;
#pragma unroll
    for (int r = 0; r < PT_POOL_ROUNDS; ++r) {                   // (straight-line rounds: no break / continue, so that the loop unrolls into plain code)
        const bool want = need && got == ~0u;
        const unsigned long long needm = __ballot(want);
        const unsigned s = (unsigned)r * 64u + lane;
        const bool avail = needm != 0ull && pool_ld(st + s) == from;
        const unsigned long long am = __ballot(avail);
        if (am) {
            if (avail) row[pool_rank(am)] = s;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const unsigned rank = pool_rank(needm);
            if (want && rank < (unsigned)__popcll(am)) {
                const unsigned cand = row[rank];
                unsigned expected = from;
                if (__hip_atomic_compare_exchange_strong(st + cand, &expected, to, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) got = cand;
            }
            __builtin_amdgcn_wave_barrier();                    // (the row is rewritten in the next round)
        }
    }
    return got;
}

// The context word by word (the Lane record has bit-fields and unions: copied as a block of memory it goes through scratch).  `W` is
// called once per word with a reference the visitor reads (store) or writes (load); the two callers below share this one list.
#define PT_CTX_WORDS(X) \
    X(L.mask.x) X(L.mask.y) X(L.mask.z) X(L.acc[0]) X(L.acc[1]) X(L.acc[2]) X(L.acc[3]) \
    X(L.origin.x) X(L.origin.y) X(L.origin.z) X(L.dir.x) X(L.dir.y) X(L.dir.z) X(L.t) X(L.time) \
    X(L.h.t) X(L.h.normal.x) X(L.h.normal.y) X(L.h.normal.z) \
    X(L.weight.x) X(L.weight.y) X(L.weight.z) \
    X(L.wi.x) X(L.wi.y) X(L.wi.z) X(L.n_shade.x) X(L.n_shade.y) X(L.n_shade.z) X(L.pdf) \
    X(L.w.t) X(L.w.u) X(L.w.v)
// 32 float words above; then the integer words: total, samples, rng (2), mesh_id, f, counters (2), hit bits, flags = 10; optional: ps_pdf (medium), view_n (3)
constexpr int PT_CTX_FLOATS = 32;
template <unsigned MATS, bool MEDIUM>
PT_DEV void ctx_store(unsigned* ctx, unsigned slot, const Lane& Lc, unsigned pix, unsigned gxy) {
    Lane& L = const_cast<Lane&>(Lc);
    unsigned* q = ctx + slot;
    int d = 0;
#define X(f) q[(d++) * PT_POOL_SLOTS] = prt_f2u(f);
    PT_CTX_WORDS(X)
#undef X
    q[(d++) * PT_POOL_SLOTS] = L.total; q[(d++) * PT_POOL_SLOTS] = L.samples;
    q[(d++) * PT_POOL_SLOTS] = L.rng.s0; q[(d++) * PT_POOL_SLOTS] = L.rng.s1;
    q[(d++) * PT_POOL_SLOTS] = (unsigned)L.mesh_id; q[(d++) * PT_POOL_SLOTS] = L.f;
    q[(d++) * PT_POOL_SLOTS] = (unsigned)L.diff | ((unsigned)L.spec << 16);
    q[(d++) * PT_POOL_SLOTS] = (unsigned)L.trans | ((unsigned)L.scatters << 16);
    q[(d++) * PT_POOL_SLOTS] = ((unsigned)L.h.mesh_id & 0xffffffu) | ((unsigned)L.h.didHit << 24) | ((unsigned)L.h.backside << 25);
    q[(d++) * PT_POOL_SLOTS] = (unsigned)L.stage | ((unsigned)L.kind << 3) | ((unsigned)L.sampledLobe << 5) | ((unsigned)L.wasSpecular << 13) |
                               ((unsigned)L.reset << 14) | ((unsigned)L.h_valid << 15) | ((unsigned)L.terminate << 16) | ((unsigned)L.w2_ran << 17) |
                               ((unsigned)L.sh << 18) | ((unsigned)L.begun << 19) | ((unsigned)L.w2 << 20) | ((unsigned)L.occluded << 21) |
                               ((unsigned)L.sh_vertex << 22);
    q[(d++) * PT_POOL_SLOTS] = pix; q[(d++) * PT_POOL_SLOTS] = gxy;
    if (MEDIUM) q[44 * PT_POOL_SLOTS] = prt_f2u(L.ps_pdf);
    if (MATS & PT_MATS_VIEW) { q[45 * PT_POOL_SLOTS] = prt_f2u(L.view_n.x); q[46 * PT_POOL_SLOTS] = prt_f2u(L.view_n.y); q[47 * PT_POOL_SLOTS] = prt_f2u(L.view_n.z); }
}
// Taking a context over is written WITHOUT control flow around the lane record: every lane of the wave reads eight words at a time (a lane
// that takes nothing reads slot 0 and throws the words away) and each field becomes `take ? word : field` -- one v_cndmask into the
// register the field lives in.  As assignments inside a divergent branch the compiler kept two copies of the whole record alive across
// the branch (before / after, merged register by register behind it: 300 B of scratch per lane more at 80 registers).
// Eight words per statement, the wait for them INSIDE the statement: to the compiler an asm's outputs are there when the statement ends.
#define PT_LDS_RD8(o)                                                                                                              \
    asm volatile("ds_read_b32 %0, %8 offset:%9\n\tds_read_b32 %1, %8 offset:%10\n\tds_read_b32 %2, %8 offset:%11\n\t"                \
                 "ds_read_b32 %3, %8 offset:%12\n\tds_read_b32 %4, %8 offset:%13\n\tds_read_b32 %5, %8 offset:%14\n\t"               \
                 "ds_read_b32 %6, %8 offset:%15\n\tds_read_b32 %7, %8 offset:%16\n\ts_waitcnt lgkmcnt(0)"                              \
                 : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)                          \
                 : "v"(base), "n"(((o) + 0) * PT_POOL_SLOTS * 4), "n"(((o) + 1) * PT_POOL_SLOTS * 4), "n"(((o) + 2) * PT_POOL_SLOTS * 4), \
                   "n"(((o) + 3) * PT_POOL_SLOTS * 4), "n"(((o) + 4) * PT_POOL_SLOTS * 4), "n"(((o) + 5) * PT_POOL_SLOTS * 4),     \
                   "n"(((o) + 6) * PT_POOL_SLOTS * 4), "n"(((o) + 7) * PT_POOL_SLOTS * 4)                                         \
                 : "memory")
// (the selects of a batch are done before the next batch is read: a volatile statement that takes their results keeps the scheduler
// from piling up all 48 loaded words first)
#define PT_PIN8(f0, f1, f2, f3, f4, f5, f6, f7) asm volatile("" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7))
#define PT_SELF(f, t) f = take ? prt_u2f(t) : f
#define PT_SELU(f, t) f = take ? (t) : f
// wave-uniform call; `take`: this lane takes the context in `slot` over
template <unsigned MATS, bool MEDIUM>
PT_DEV void ctx_load(const unsigned* ctx, const bool take, unsigned slot, Lane& L, unsigned& pix, unsigned& gxy) {
    // byte address of the slot's first word in LDS; the words in the order of ctx_store (PT_CTX_WORDS, then the integers)
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned*)(ctx + (take ? slot : 0u));
    unsigned t0, t1, t2, t3, t4, t5, t6, t7;
    PT_LDS_RD8(0);
    PT_SELF(L.mask.x, t0); PT_SELF(L.mask.y, t1); PT_SELF(L.mask.z, t2); PT_SELF(L.acc[0], t3); PT_SELF(L.acc[1], t4); PT_SELF(L.acc[2], t5); PT_SELF(L.acc[3], t6); PT_SELF(L.origin.x, t7);
    PT_PIN8(L.mask.x, L.mask.y, L.mask.z, L.acc[0], L.acc[1], L.acc[2], L.acc[3], L.origin.x);
    PT_LDS_RD8(8);
    PT_SELF(L.origin.y, t0); PT_SELF(L.origin.z, t1); PT_SELF(L.dir.x, t2); PT_SELF(L.dir.y, t3); PT_SELF(L.dir.z, t4); PT_SELF(L.t, t5); PT_SELF(L.time, t6); PT_SELF(L.h.t, t7);
    PT_PIN8(L.origin.y, L.origin.z, L.dir.x, L.dir.y, L.dir.z, L.t, L.time, L.h.t);
    PT_LDS_RD8(16);
    PT_SELF(L.h.normal.x, t0); PT_SELF(L.h.normal.y, t1); PT_SELF(L.h.normal.z, t2); PT_SELF(L.weight.x, t3); PT_SELF(L.weight.y, t4); PT_SELF(L.weight.z, t5); PT_SELF(L.wi.x, t6); PT_SELF(L.wi.y, t7);
    PT_PIN8(L.h.normal.x, L.h.normal.y, L.h.normal.z, L.weight.x, L.weight.y, L.weight.z, L.wi.x, L.wi.y);
    PT_LDS_RD8(24);
    PT_SELF(L.wi.z, t0); PT_SELF(L.n_shade.x, t1); PT_SELF(L.n_shade.y, t2); PT_SELF(L.n_shade.z, t3); PT_SELF(L.pdf, t4); PT_SELF(L.w.t, t5); PT_SELF(L.w.u, t6); PT_SELF(L.w.v, t7);
    PT_PIN8(L.wi.z, L.n_shade.x, L.n_shade.y, L.n_shade.z, L.pdf, L.w.t, L.w.u, L.w.v);
    PT_LDS_RD8(32);
    PT_SELU(L.total, t0); PT_SELU(L.samples, t1); PT_SELU(L.rng.s0, t2); PT_SELU(L.rng.s1, t3); PT_SELU(L.mesh_id, (int)t4); PT_SELU(L.f, t5);
    if (take) { L.diff = t6 & 0xffffu; L.spec = t6 >> 16; L.trans = t7 & 0xffffu; L.scatters = t7 >> 16; }
    asm volatile("" : "+v"(L.total), "+v"(L.samples), "+v"(L.rng.s0), "+v"(L.rng.s1), "+v"(L.mesh_id), "+v"(L.f));
    PT_LDS_RD8(40);
    PT_SELU(pix, t2); PT_SELU(gxy, t3);
    if (MEDIUM) PT_SELF(L.ps_pdf, t4);
    if (MATS & PT_MATS_VIEW) { PT_SELF(L.view_n.x, t5); PT_SELF(L.view_n.y, t6); PT_SELF(L.view_n.z, t7); }
    if (take) {
        const unsigned hbits = t0, flags = t1;
        L.h.mesh_id = (int)(hbits << 8) >> 8; L.h.didHit = (hbits >> 24) & 1u; L.h.backside = (hbits >> 25) & 1u;
        L.stage = flags & 7u; L.kind = (flags >> 3) & 3u; L.sampledLobe = (flags >> 5) & 0xffu; L.wasSpecular = (flags >> 13) & 1u; L.reset = (flags >> 14) & 1u;
        L.h_valid = (flags >> 15) & 1u; L.terminate = (flags >> 16) & 1u; L.w2_ran = (flags >> 17) & 1u; L.sh = (flags >> 18) & 1u; L.begun = (flags >> 19) & 1u;
        L.w2 = (flags >> 20) & 1u; L.occluded = (flags >> 21) & 1u; L.sh_vertex = (flags >> 22) & 1u;
        L.fresh = false;                               // (a parked context's walk has begun; the walk state itself is the walker's)
        L.w.node = 0u; L.w.sp = 0u; L.w.pend_count = 0u; L.w.pend_first = 0u; L.w.slot = 0u; L.w.found = false; L.w.done = true; L.w.last = false;
    }
}

PT_DEV bool ctx_frozen(const FrameArgs& fa, const Lane& L) { return fa.spp_limit && L.reset && L.samples >= fa.spp_limit; }

template <unsigned MATS, bool MEDIUM, int WAVES>
__global__ __launch_bounds__(PT_POOL_BLOCK, WAVES) void render_kernel_pool(const DevScene sc, const DevCamera cam, const DevState S,
                                                                           const FrameArgs fa, float4* __restrict__ fb) {
    extern __shared__ unsigned pool_lds[];
    const PoolLayout lay = pool_layout(sc.stack_levels);
    unsigned* const st = pool_lds + lay.st;
    unsigned* const ctl = pool_lds + lay.ctl;
    unsigned* const ctx = pool_lds + lay.ctx;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned* const row = pool_lds + lay.rows + wv * 64u;

    for (unsigned i = threadIdx.x; i < PT_POOL_SLOTS + PC_WORDS; i += PT_POOL_BLOCK) pool_lds[i] = 0u;      // st (FREE) and ctl
    __syncthreads();

    const bool shader = wv < PT_POOL_SHADERS;
    Lane L;
    lane_init(L);
    L.stage = ST_EMPTY; L.f = 0xffffffffu;
    unsigned pix = 0u, gxy = 0u;
    if (shader) {                                                // this wave's tile: its 64 contexts to begin with
        const unsigned tiles_x = ((unsigned)fa.width + 7u) / 8u, tiles_y = ((unsigned)fa.rows + 7u) / 8u;
        const unsigned tile_k = blockIdx.x * PT_POOL_SHADERS + wv;
        const unsigned long long tile64 = (unsigned long long)tile_k * fa.tile_stride + fa.tile_first;
        const bool tile_ok = tile64 < (unsigned long long)tiles_x * tiles_y;
        const unsigned tile = tile_ok ? (unsigned)tile64 : 0u;
        const int lx = (int)(tile % tiles_x) * 8 + (int)(lane & 7u);
        const int ly = (int)(tile / tiles_x) * 8 + (int)(lane >> 3);
        const bool in_frame = tile_ok && lx < fa.width && ly < fa.rows;
        if (in_frame) {
            const size_t id = (size_t)ly * (size_t)fa.width + (size_t)lx;
            const int gy = fa.row0 + (ly / fa.block_rows * fa.n_parts + fa.part) * fa.block_rows + ly % fa.block_rows;
            pix = (unsigned)id; gxy = (unsigned)lx | ((unsigned)gy << 16);
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
            L.stage = ST_READY;
        }
        const unsigned n_in = (unsigned)__popcll(__ballot(in_frame));
        const unsigned n_owe = (unsigned)__popcll(__ballot(in_frame && lane_owes_frames(fa, L)));
        if (lane == 0u) {
            if (n_in) __hip_atomic_fetch_add(ctl + PC_TOTAL, n_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (n_owe) __hip_atomic_fetch_add(ctl + PC_OWING, n_owe, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    const unsigned total = __builtin_amdgcn_readfirstlane(pool_ld(ctl + PC_TOTAL));
    const unsigned TQ = fa.tri_sixteenths;
    // watchdog: iterations a wave may take (a context needs a few per segment, plus the waits for a slot)
    const unsigned long long max_iters = 64ull * ((unsigned long long)fa.seed_frames + 1024ull);
    unsigned long long iters = 0ull;
    unsigned n_unfinished = 0u;                                   // (shading waves) pixels stored that are not frozen
#ifdef PT_POOL_STATS
    unsigned long long pstat_[32];
    for (int k = 0; k < 32; ++k) pstat_[k] = 0ull;
#endif

#ifdef PT_POOL_NO_WALKER
    if (!shader) return;
#endif
#ifdef PT_POOL_NO_SHADER
    if (shader) return;
#endif
    if (!shader) {
        // ---- walker wave ------------------------------------------------------------------------------------------------------------
        TravStack stk;
        stk.lds = pool_lds + lay.wstack + (wv - PT_POOL_SHADERS) * sc.stack_levels * 64u + lane; stk.stride = 64u;
        bool active = false, any_hit = false;
        unsigned slot = 0u;
        Ray wr;
        wr.origin = wr.dir = wr.normal = wr.pos = splat(0.0f); wr.t = 0.0f; wr.backside = false; wr.time = 0.0f;
        RayPre p = ray_pre(wr);
        WalkState w;
        w.node = 0u; w.sp = 0u; w.pend_count = 0u; w.t = 0.0f; w.u = w.v = 0.0f; w.slot = 0u; w.found = false; w.done = true; w.last = false; w.pend_first = 0u;
        unsigned idle_spins = 0u;
        for (;;) {
            if (__builtin_amdgcn_readfirstlane(pool_ld(ctl + PC_STORED)) >= total || pool_ld(ctl + PC_ABORT)) break;
            if (++iters > max_iters * 64ull) { if (lane == 0u) pool_st(ctl + PC_ABORT, 1u); break; }
            const unsigned n_idle = (unsigned)__popcll(__ballot(!active));
            PT_PSTAT(16, 1);
            if (n_idle >= PT_POOL_REFILL) {
                PT_PSTAT(17, 1);
                const unsigned s = pool_claim(st, row, !active, PS_WAIT, PS_WALKING);
                PT_PSTAT(18, __popcll(__ballot(s != ~0u)));
                if (s != ~0u) {
                    const unsigned* rr = ctx + PT_CTX_RAY * PT_POOL_SLOTS + s;
                    wr.origin = F3(prt_u2f(rr[0]), prt_u2f(rr[PT_POOL_SLOTS]), prt_u2f(rr[2 * PT_POOL_SLOTS]));
                    wr.dir = F3(prt_u2f(rr[3 * PT_POOL_SLOTS]), prt_u2f(rr[4 * PT_POOL_SLOTS]), prt_u2f(rr[5 * PT_POOL_SLOTS]));
                    wr.t = prt_u2f(rr[6 * PT_POOL_SLOTS]);
                    any_hit = rr[7 * PT_POOL_SLOTS] != 0u;
                    p = ray_pre(wr);
                    walk_begin(sc, any_hit, wr, wr.t, p, w, stk);
                    active = true; slot = s;
                }
            }
            if (!__ballot(active)) {                              // nothing to walk: let the shading waves have the issue slots
                __builtin_amdgcn_s_sleep(32);
                PT_PSTAT(19, 1);
                if (++idle_spins > (1u << 23)) { if (lane == 0u) pool_st(ctl + PC_ABORT, 1u); break; }
                continue;
            }
            idle_spins = 0u;
            for (int k = 0; k < 4; ++k) {                          // a few steps between two looks at the pool
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
            if (active && w.done) {                                // the answer goes back into the slot; whoever has an empty lane takes it
                unsigned* rr = ctx + PT_CTX_RAY * PT_POOL_SLOTS + slot;
                rr[0] = prt_f2u(w.t); rr[PT_POOL_SLOTS] = prt_f2u(w.u); rr[2 * PT_POOL_SLOTS] = prt_f2u(w.v);
                rr[3 * PT_POOL_SLOTS] = (unsigned)w.slot | ((unsigned)w.found << 29);
                pool_st(st + slot, PS_DONE);
                active = false;
            }
        }
#ifdef PT_POOL_STATS
        if (lane == 0u) for (int k = 16; k < 32; ++k) if (pstat_[k]) atomicAdd(&g_pool_stats[k], pstat_[k]);
#endif
        return;
    }

    // ---- shading wave ---------------------------------------------------------------------------------------------------------------
    TravStack stk;
    stk.lds = pool_lds + lay.dummy + wv * 64u + lane; stk.stride = 64u;       // the step at the root pushes at most one entry
    unsigned idle_spins = 0u;
    for (;;) {
#ifdef PT_POOL_X6
        if (!__any(lane_runnable(fa, L, false) || (L.stage != ST_READY && L.stage != ST_EMPTY))) break;
#else
        const unsigned stored = __builtin_amdgcn_readfirstlane(pool_ld(ctl + PC_STORED));
        if (stored >= total || pool_ld(ctl + PC_ABORT)) break;
#endif
#ifndef PT_POOL_X4
        if (++iters > max_iters) { if (lane == 0u) pool_st(ctl + PC_ABORT, 1u); break; }
#endif
        // 1. empty lanes take over contexts whose walk is done
        bool done_waiting = false;
        {
            const bool empty = L.stage == ST_EMPTY;
#ifdef PT_POOL_NO_UNPARK
            if (false) {
#else
            if (__ballot(empty)) {
#endif
                const unsigned s = pool_claim(st, row, empty, PS_DONE, PS_TAKING);
                PT_PSTAT(6, __popcll(__ballot(empty))); PT_PSTAT(7, __popcll(__ballot(s != ~0u)));
#ifndef PT_POOL_NO_LOAD
                if (__ballot(s != ~0u)) ctx_load<MATS, MEDIUM>(ctx, s != ~0u, s, L, pix, gxy);
#endif
                if (s != ~0u) {
                    const unsigned* rr = ctx + PT_CTX_RAY * PT_POOL_SLOTS + s;
                    const unsigned sf = rr[3 * PT_POOL_SLOTS];
                    if (L.stage == ST_WALKC) { L.w.t = prt_u2f(rr[0]); L.w.u = prt_u2f(rr[PT_POOL_SLOTS]); L.w.v = prt_u2f(rr[2 * PT_POOL_SLOTS]); L.w.slot = sf & 0x1fffffffu; }
                    L.w.found = (sf >> 29) & 1u;                  // (an any-hit walk leaves (t, u, v) alone: Lane::a lives there)
                    L.w.done = true; L.w.pend_count = 0u;
                    pool_st(st + s, PS_FREE);
                }
            } else {
                // no lane to spare: is a finished walk waiting for one?  (then nobody of this wave runs ahead: step 5)
                bool dw = false;
#pragma unroll
                for (int r = 0; r < PT_POOL_ROUNDS; ++r) dw = dw || pool_ld(st + r * 64 + lane) == PS_DONE;
                done_waiting = __ballot(dw) != 0ull;
            }
        }
        const bool occupied = L.stage != ST_EMPTY;
        // A wave without a context waits for one (or for the end) off the issue slots -- and then runs the body all the same, with every
        // phase masked out: a path AROUND the body (`continue`, or the body as the else branch) carries the whole lane record past it,
        // and the register allocator answered that with 130 - 150 B of scratch per lane in the bare loop (0 without).
        if (!__ballot(occupied)) {
            __builtin_amdgcn_s_sleep(64);
            PT_PSTAT(2, 1);
            --iters;
            if (++idle_spins > (1u << 22) && lane == 0u) pool_st(ctl + PC_ABORT, 1u);
        } else {
            idle_spins = 0u;
        }
        {
        PT_PSTAT(0, 1); PT_PSTAT(1, __popcll(__ballot(occupied)));
#ifdef PT_POOL_X7
        const unsigned owing = 0u;
#else
        const unsigned owing = __builtin_amdgcn_readfirstlane(pool_ld(ctl + PC_OWING));
#endif
        const bool owes_before = occupied && lane_owes_frames(fa, L);
        const bool runnable = occupied && lane_runnable(fa, L, owing != 0u && !done_waiting);
#ifdef PT_POOL_X2
        const int gx = (int)lane, gy = (int)wv;
#else
        const int gx = (int)(gxy & 0xffffu), gy = (int)(gxy >> 16);
#endif
        PT_PSTAT(3, __popcll(__ballot(runnable)));
        PT_PSTAT(8, __popcll(__ballot(L.stage == ST_WALKC || L.stage == ST_WALKS)));      // lanes that wait for a slot since an earlier iteration
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
#ifndef PT_POOL_X3
            if (walking && L.fresh) {
                const Ray wr = lane_shadow_ray<MEDIUM, (MATS & PT_MATS_ENVIS) != 0>(L);
                const RayPre p = ray_pre(wr);
                walk_begin(sc, true, wr, wr.t, p, L.w, stk);
                L.fresh = false;
            }
#endif
            if (walking && L.w.done) { L.occluded = L.w.found; L.stage = ST_FINISH; }
        }
        if (L.stage == ST_FINISH) lane_finish<MATS, MEDIUM>(sc, L);                                           // E
        // 3. contexts that owe the launch no frame any more
        {
#ifdef PT_POOL_X1
            const bool owes_after = owes_before;
#else
            const bool owes_after = (L.stage != ST_EMPTY) && lane_owes_frames(fa, L);
#endif
            const unsigned n = (unsigned)__popcll(__ballot(owes_before && !owes_after));
            if (n && lane == 0u) __hip_atomic_fetch_sub(ctl + PC_OWING, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        // 4. a ray that has to go deeper than the root: the context goes into the pool, the walkers take it from there
        {
            const bool closest = L.stage == ST_WALKC;
            const bool need = (closest || L.stage == ST_WALKS) && !L.fresh && !L.w.done;
#ifdef PT_POOL_NO_PARK
            if (false) {
#else
            if (__ballot(need)) {
#endif
                PT_PSTAT(4, __popcll(__ballot(need)));
                const unsigned s = pool_claim(st, row, need, PS_FREE, PS_FILLING);
                PT_PSTAT(5, __popcll(__ballot(s != ~0u)));
                if (s != ~0u) {
                    const Ray wr = closest ? lane_closest_ray<MEDIUM>(L) : lane_shadow_ray<MEDIUM, (MATS & PT_MATS_ENVIS) != 0>(L);
#ifndef PT_POOL_NO_STORE
                    ctx_store<MATS, MEDIUM>(ctx, s, L, pix, gxy);
#endif
                    unsigned* rr = ctx + PT_CTX_RAY * PT_POOL_SLOTS + s;
                    rr[0] = prt_f2u(wr.origin.x); rr[PT_POOL_SLOTS] = prt_f2u(wr.origin.y); rr[2 * PT_POOL_SLOTS] = prt_f2u(wr.origin.z);
                    rr[3 * PT_POOL_SLOTS] = prt_f2u(wr.dir.x); rr[4 * PT_POOL_SLOTS] = prt_f2u(wr.dir.y); rr[5 * PT_POOL_SLOTS] = prt_f2u(wr.dir.z);
                    rr[6 * PT_POOL_SLOTS] = prt_f2u(closest ? PT_INF : wr.t);
                    rr[7 * PT_POOL_SLOTS] = closest ? 0u : 1u;
                    pool_st(st + s, PS_WAIT);
                    L.stage = ST_EMPTY; L.begun = false; L.f = 0xffffffffu;
                }
            }
        }
        // 5. a context between two segments that may not start another one is stored: it has done the frames of this launch (and the
        //    workgroup has nobody left to wait for, or a finished walk waits for a lane), or it has its samples
        {
#ifdef PT_POOL_NO_RETIRE
            const bool retire = false;
#else
            const bool retire = L.stage == ST_READY && !L.begun && !lane_runnable(fa, L, owing != 0u && !done_waiting);
#endif
            const unsigned long long rm = __ballot(retire);
            if (rm) {
                const bool frozen = ctx_frozen(fa, L);
                n_unfinished += (unsigned)__popcll(__ballot(retire && !frozen));      // (wave-uniform: counted outside the branch)
                if (retire) {
                    const size_t id = (size_t)pix;
                    if (L.f) {
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
                    L.stage = ST_EMPTY; L.f = 0xffffffffu;
                }
                if (lane == (unsigned)__builtin_ctzll(rm)) __hip_atomic_fetch_add(ctl + PC_STORED, (unsigned)__popcll(rm), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        }   // (a wave with contexts)
    }
#ifdef PT_POOL_STATS
    if (lane == 0u) for (int k = 0; k < 16; ++k) if (pstat_[k]) atomicAdd(&g_pool_stats[k], pstat_[k]);
#endif
    if (fa.unfinished) {                                         // as render_kernel: the last shading wave of the launch reports
        const unsigned nu = __builtin_amdgcn_readfirstlane(n_unfinished);
        if (lane == 0u) {
            const unsigned long long before = nu ? atomicAdd(fa.unfinished, (unsigned long long)nu) : 0ull;
            if (fa.unfinished_host && before != ~0ull) {
                if (atomicAdd(fa.unfinished + 1, 1ull) == (unsigned long long)gridDim.x * PT_POOL_SHADERS - 1ull) {
                    const unsigned long long tot = atomicExch(fa.unfinished, 0ull);
                    atomicExch(fa.unfinished + 1, 0ull);
                    *reinterpret_cast<volatile unsigned long long*>(fa.unfinished_host) = tot;
                }
            }
        }
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------------------------
static inline size_t pool_lds_bytes(const DevScene& sc) { return (size_t)pool_layout(sc.stack_levels).total * sizeof(unsigned); }

template <unsigned MATS, bool MEDIUM, int WAVES>
static bool launch_pool(const DevScene& sc, const DevCamera& cam, const DevState& S, const FrameArgs& fa, float4* fb, hipStream_t stream, unsigned n_tiles_launch) {
    const size_t lds = pool_lds_bytes(sc);
    if (lds > 160u * 1024u) return false;
    static size_t lds_attr = 0;
    if (lds > 65536u && lds > lds_attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&render_kernel_pool<MATS, MEDIUM, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        lds_attr = lds;
    }
    const unsigned grid = (n_tiles_launch + PT_POOL_SHADERS - 1u) / PT_POOL_SHADERS;
    if (!grid) return true;
    hipLaunchKernelGGL((render_kernel_pool<MATS, MEDIUM, WAVES>), dim3(grid), dim3(PT_POOL_BLOCK), lds, stream, sc, cam, S, fa, fb);
    return true;
}

}  // namespace prt
