// pt_kernels.hip -- the small kernels of libprt and the dispatch to the render kernel's compiled material sets.
//
//   render_kernel<MATS, MEDIUM, WAVES>   pt_render.h, instantiated by pt_inst_*.hip (one file per material set: they compile in parallel)
//   state_to_rtd / rtd_to_state          80 B/px SoA planes <-> the reference's 112 B RTD records
//   count_kernel                         sum of samples / segments / frozen pixels (Msamples/s accounting)
//   tonemap_kernel                       shaders/tonemapper.glsl
//   selftest_*                           per-function known-answer entries
#include "pt_render.h"
#include "pt_selftest.h"

#ifdef PT_UNITY            // development builds (tools/build_variant.sh, tools/isa.sh): one translation unit
#ifdef PT_DEV_ONE_VARIANT
#include "pt_inst_light_diff.hip"
#else
#include "pt_inst_light_diff.hip"
#include "pt_inst_coat.hip"
#include "pt_inst_rough_cond.hip"
#include "pt_inst_rough_diel.hip"
#include "pt_inst_generic.hip"
#include "pt_inst_sdf.hip"
#include "pt_inst_view.hip"
#include "pt_inst_view_sdf.hip"
#include "pt_inst_pick.hip"
#include "pt_inst_envis.hip"
#endif
#elif defined(PT_PHASE_CLOCKS) || defined(PT_DEV_ONE_VARIANT)
#error "PT_PHASE_CLOCKS / PT_DEV_ONE_VARIANT builds are unity builds: add -DPT_UNITY"
#endif

namespace prt {

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
    if (fn == 19) {         // exhaustive: hw_sqrt against the IEEE square root on the 65536 bit patterns (i << 16) + k
        unsigned bad = 0;
        for (unsigned k = 0; k < 65536u; ++k) {
            const float v = prt_u2f(((unsigned)i << 16) + k);
            const float q = __builtin_sqrtf(v), h = hw_sqrt(v);
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

#ifdef PT_POOL_STATS
void dump_pool_stats() {
    unsigned long long h[32] = {0};
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_pool_stats), sizeof(h)) != hipSuccess) return;
    const double it = h[0] ? (double)h[0] : 1.0;
    fprintf(stderr, "pool stats: shading-wave iterations %llu, runnable lanes %.1f, lanes waiting for a walker %.1f; rays posted %llu (%.2f per iteration), answers taken %llu\n",
            h[0], h[3] / it, h[8] / it, h[4], h[4] / it, h[7]);
    const double st = h[20] ? (double)h[20] : 1.0;
    fprintf(stderr, "pool stats: walker loops %llu, refill tries %llu got %llu rays, sleeps %llu; steps %llu at %.1f lanes (box %.1f), triangle units %llu\n",
            h[16], h[17], h[18], h[19], h[20], h[21] / st, h[22] / st, h[23]);
}
#endif
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

// Variant choice = the AOT analogue of the reference's per-scene program build (include/CL/cl_kernel.h:226-345 compiles exactly the
// scene's ACTIVE_MATS): the material sets of the BASELINE configs are compiled (LIGHT|DIFF, +COAT, +ROUGH_COND, +DIEL|ROUGH_DIEL),
// any other set runs the generic variant, which dispatches on the material's type bits at run time -- the same code, the same bits
// (LaunchOpts::generic forces it: the tests run every golden through both).
RenderLaunch launch_render(const DevScene& sc, const DevCamera& cam, const DevState& S, const FrameArgs& fa, float4* fb,
                           hipStream_t stream, const LaunchOpts& lo) {
    const unsigned am = sc.active_mats;
    const bool medium = sc.has_medium != 0;
#ifdef PT_DEV_ONE_VARIANT                 // development builds (tools/): only the headline variant, compiles in seconds
    if (sc.n_sdfs || medium || sc.view || am != (PRT_MAT_LIGHT | PRT_MAT_DIFF)) return RenderLaunch{};
    return launch_set_light_diff(false, sc, cam, S, fa, fb, stream, lo);
#else
    if (sc.env_is) return launch_set_envis(medium, sc, cam, S, fa, fb, stream, lo);              // (pack_scene refuses it with a medium, SDFs, views, the light pick)
    if (sc.pick_random_light) return launch_set_pick(medium, sc, cam, S, fa, fb, stream, lo);     // PICK_RANDOM_LIGHT: the generic set (pack_scene refuses it with SDFs / views)
    if (sc.view) return sc.n_sdfs ? launch_set_view_sdf(medium, sc, cam, S, fa, fb, stream, lo) : launch_set_view(medium, sc, cam, S, fa, fb, stream, lo);
    if (sc.n_sdfs) return launch_set_sdf(medium, sc, cam, S, fa, fb, stream, lo);      // H_SDF scenes: the generic set with the raymarcher
    if (!lo.generic) {
        if (am == (PRT_MAT_LIGHT | PRT_MAT_DIFF)) return launch_set_light_diff(medium, sc, cam, S, fa, fb, stream, lo);
        if (am == (PRT_MAT_LIGHT | PRT_MAT_DIFF | PRT_MAT_COAT)) return launch_set_coat(medium, sc, cam, S, fa, fb, stream, lo);
        if (am == (PRT_MAT_LIGHT | PRT_MAT_DIFF | PRT_MAT_ROUGH_COND)) return launch_set_rough_cond(medium, sc, cam, S, fa, fb, stream, lo);
        if (am == (PRT_MAT_LIGHT | PRT_MAT_DIFF | PRT_MAT_DIEL | PRT_MAT_ROUGH_DIEL)) return launch_set_rough_diel(medium, sc, cam, S, fa, fb, stream, lo);
    }
    return launch_set_generic(medium, sc, cam, S, fa, fb, stream, lo);
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
