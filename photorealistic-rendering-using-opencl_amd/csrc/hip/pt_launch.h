// pt_launch.h -- host-callable launchers of the kernels in pt_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "prt_types.h"
#include "pt_layout.h"

namespace prt {

// launches the scene-specialised variant (the AOT analogue of the reference's per-scene program
// build, include/CL/cl_kernel.h); returns the variant's name for profiles/stats
const char* launch_render(const DevScene& sc, const DevCamera& cam, const DevState& S, const FrameArgs& fa, float4* fb,
                          hipStream_t stream);
// workgroups (tiles) launch_render uses for a width x rows frame part
unsigned render_tile_count(int width, int rows);
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
