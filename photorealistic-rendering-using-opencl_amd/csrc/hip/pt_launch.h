// pt_launch.h -- host-callable launchers of the kernels in pt_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "prt_types.h"
#include "pt_layout.h"

namespace prt {

// launcher-level choices of a context (prt_set_option / PRT_WAVES, PRT_SCATTER, PRT_GENERIC): what is forced, for tests and experiments
struct LaunchOpts {
    int waves = 0;          // 0: chosen per launch; 5 / 6: that build of the kernel (waves per SIMD the register allocator leaves room for)
    int scatter = -1;       // -1: chosen per launch; 0: one 8x8 tile per wave; 1: a wave's pixels scattered over the launch's tiles
    int generic = 0;        // 1: the run-time-dispatched material set even where the scene's own set is compiled
    int any_dist = 0;       // 1: the set's instance that carries every microfacet distribution even where the scene uses one (PT_MATS_DISTS)
    int pix_per_wave = 0;   // 0: chosen per launch; 64 / 32 / 16: pixels a wave renders (FrameArgs::sub_shift)
    int pool = 0;           // 1: render_kernel_pool (pt_pool.h: shading waves + walker waves around a pool of parked contexts) where a launch can take it
};
// what a launch ran: kernel variant, wave-count build, pixel-to-wave mapping (prt_kernel_variant)
struct RenderLaunch { const char* name = ""; int waves = 0; int scatter = 0; int ordered = 0; int pool = 0; int pix_per_wave = 64; };   // ordered: the tiles were taken in the launcher's order

// launches the scene-specialised variant (the AOT analogue of the reference's per-scene program
// build, include/CL/cl_kernel.h:226-345)
RenderLaunch launch_render(const DevScene& sc, const DevCamera& cam, const DevState& S, const FrameArgs& fa, float4* fb,
                           hipStream_t stream, const LaunchOpts& lo);
// one per compile-time material set (pt_inst_*.hip; each covers medium off / on)
#define PT_DECLARE_SET(fn) RenderLaunch fn(bool medium, const DevScene& sc, const DevCamera& cam, const DevState& S, const FrameArgs& fa, float4* fb, \
                                           hipStream_t stream, const LaunchOpts& lo)
PT_DECLARE_SET(launch_set_light_diff);
PT_DECLARE_SET(launch_set_coat);
PT_DECLARE_SET(launch_set_rough_cond);
PT_DECLARE_SET(launch_set_rough_diel);
PT_DECLARE_SET(launch_set_generic);
PT_DECLARE_SET(launch_set_sdf);
PT_DECLARE_SET(launch_set_view);
PT_DECLARE_SET(launch_set_view_sdf);
PT_DECLARE_SET(launch_set_pick);
PT_DECLARE_SET(launch_set_envis);

// workgroups (tiles) launch_render uses for a width x rows frame part
unsigned render_tile_count(int width, int rows);
#ifdef PT_POOL_STATS
void dump_pool_stats();                   // development builds: what the waves of the pool kernel did (pt_pool.h)
#endif
#ifdef PT_PHASE_CLOCKS
void dump_phase_clocks();                 // development builds: prints the per-phase cycle shares of all launches so far
#endif
// per-camera part of createCamRay (camera.cl:19-28), on the host with the arithmetic of pt_device.h
void make_dev_camera(const prt_camera& in, DevCamera& out);
void launch_state_to_rtd(const DevState& S, prt_path_state* out, size_t n, hipStream_t stream);
void launch_rtd_to_state(const prt_path_state* in, const DevState& S, float4* fb, size_t n, hipStream_t stream);
void launch_selftest_math(int fn, const float* a, const float* b, float* out, int n, hipStream_t stream);
void launch_selftest_fn(int fn, const float* params, const float* in, float* out, int n, hipStream_t stream);
void launch_tonemap(const float4* fb, unsigned char* out, const FrameArgs& fa, hipStream_t stream);
void launch_count(const DevState& S, size_t n, unsigned spp, unsigned long long* out3, hipStream_t stream);

}  // namespace prt
