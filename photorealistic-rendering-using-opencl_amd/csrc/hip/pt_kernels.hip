// pt_kernels.hip -- gfx950 kernels of libprt (see pt_device.h for the arithmetic contract).
//
//   render_kernel<MATS, MEDIUM>   the hot path: one lane = one pixel, n_frames segments per launch
//   state_to_rtd / rtd_to_state   80 B/px SoA planes <-> the reference's 112 B RTD records
//   count_kernel                  sum of samples / segments / frozen pixels (Msamples/s accounting)
//
// Launch geometry: 256-thread workgroups = 4 waves; each wave owns an 8x8 pixel tile (primary rays of
// one wave walk the same BVH nodes), a workgroup a 16x16 tile; 32 KiB of LDS per workgroup hold the
// traversal stacks.  A 1920x1080 frame is 8 160 workgroups >> 256 CUs.
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
#define PT_BLOCK 256        // threads per workgroup
#endif
#ifndef PT_LDS_STACK
#define PT_LDS_STACK 32     // traversal-stack levels kept in LDS ([level][thread]: conflict-free, 32 KiB per workgroup)
#endif

template <unsigned MATS, bool MEDIUM>
__global__ __launch_bounds__(PT_BLOCK, PT_MIN_WAVES) void render_kernel(const DevScene sc, const DevCamera cam, const DevState S,
                                                                        const FrameArgs fa, float4* __restrict__ fb) {
    // workgroup tile: (PT_BLOCK / 128) x 2 waves of 8x8 pixels
    constexpr int TILE_W = PT_BLOCK / 16, WAVES_X = TILE_W / 8;
    const int tiles_x = (fa.width + TILE_W - 1) / TILE_W;
    const int tile_x = (int)(blockIdx.x % (unsigned)tiles_x), tile_y = (int)(blockIdx.x / (unsigned)tiles_x);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lx = tile_x * TILE_W + (wave % WAVES_X) * 8 + (lane & 7);
    const int ly = tile_y * 16 + (wave / WAVES_X) * 8 + (lane >> 3);
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
    __shared__ unsigned lds_stack[PT_LDS_STACK * PT_BLOCK];
    unsigned deep_stack[PT_STACK_DEPTH - PT_LDS_STACK];
    TravStack stk;
    stk.lds = lds_stack + threadIdx.x; stk.stride = PT_BLOCK; stk.deep = deep_stack; stk.lds_levels = PT_LDS_STACK;
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
        if (m && lane == (int)__builtin_ctzll(__ballot(1))) atomicAdd(fa.unfinished, (unsigned long long)__popcll(m));
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

// the same switch is evaluated on the host by oracle/detmath_probe.c
__global__ void selftest_math_kernel(int fn, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int n) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
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
        default: r = prt_recip(x); break;
    }
    out[i] = r;
}

// ---- host-side launchers -------------------------------------------------------------------------------
template <unsigned MATS, bool MEDIUM>
static void launch_variant(const DevScene& sc, const DevCamera& cam, const DevState& S, const FrameArgs& fa, float4* fb,
                           hipStream_t stream) {
    constexpr unsigned TILE_W = PT_BLOCK / 16;
    const unsigned tiles_x = ((unsigned)fa.width + TILE_W - 1) / TILE_W, tiles_y = (unsigned)((fa.rows + 15) >> 4);
    hipLaunchKernelGGL((render_kernel<MATS, MEDIUM>), dim3(tiles_x * tiles_y), dim3(PT_BLOCK), 0, stream, sc, cam, S, fa, fb);
}

// Variant choice = the AOT analogue of the reference's per-scene program build (include/CL/cl_kernel.h):
// material set (LIGHT|DIFF only, or generic) x global medium.
const char* launch_render(const DevScene& sc, const DevCamera& cam, const DevState& S, const FrameArgs& fa, float4* fb,
                          hipStream_t stream) {
    constexpr unsigned LD = PRT_MAT_LIGHT | PRT_MAT_DIFF;
    const unsigned am = sc.active_mats;
    if (!sc.has_medium) {
        if (am == LD) { launch_variant<LD, false>(sc, cam, S, fa, fb, stream); return "render_kernel<LIGHT|DIFF>"; }
        launch_variant<0u, false>(sc, cam, S, fa, fb, stream);
        return "render_kernel<generic>";
    }
    if (am == LD) { launch_variant<LD, true>(sc, cam, S, fa, fb, stream); return "render_kernel<LIGHT|DIFF,medium>"; }
    launch_variant<0u, true>(sc, cam, S, fa, fb, stream);
    return "render_kernel<generic,medium>";
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
void launch_count(const DevState& S, size_t n, unsigned spp, unsigned long long* out3, hipStream_t stream) {
    unsigned blocks = (unsigned)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(count_kernel, dim3(blocks), dim3(256), 0, stream, S, n, spp, out3);
}

}  // namespace prt
