/*
 * prt.h -- C ABI of libprt: the MI355X-native drop-in for the OpenCL enqueue sequence of the
 * reference renderer (Mourtz/Photorealistic-Rendering-using-OpenCL, src/main.cpp).
 *
 * The reference has no plugin interface for its radiance loop; the boundary is the sequence of
 * OpenCL host calls in src/main.cpp.  Each entry point below replaces one group of those calls
 * (file:line given per function) with the same ownership rule the reference uses
 * (CL_MEM_COPY_HOST_PTR, include/CL/cl_help.h:196-202): the caller keeps ownership of every host
 * array, the library copies on upload.  Plain pointers and sizes only; no C++/torch types.
 *
 * Threading: one context per device; calls on one context are not re-entrant (the reference is
 * single-threaded with an in-order queue and a finish() after every enqueue).  Contexts on
 * different devices may be driven from different threads/processes.
 *
 * Errors: every function returns PRT_OK (0) or a negative prt_status; prt_last_error() gives a
 * message.  The library never aborts or exits (the reference prints and exit(1)s,
 * include/CL/cl_kernel.h:22-28).  There is NO CPU fallback: without a HIP device prt_create fails.
 */
#ifndef PRT_H
#define PRT_H

#include <stdint.h>
#include "prt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

#define PRT_ABI_VERSION 3
#define PRT_MAX_LIGHTS 16

typedef enum prt_status {
    PRT_OK = 0,
    PRT_ERR_INVALID_ARGUMENT = -1,
    PRT_ERR_NO_DEVICE = -2,
    PRT_ERR_HIP = -3,
    PRT_ERR_NOT_READY = -4,     /* scene / camera / size not set before rendering */
    PRT_ERR_UNSUPPORTED = -5    /* box primitives: they never render in the reference either (box.cl is not included) */
} prt_status;

/* phase function of the global medium.  The reference wires Isotropic at source level
 * (kernels/media.cl:61); HG (g fixed 0.6, kernels/phasefunctions/HenyeyGreenstein.cl:4) and
 * Rayleigh (kernels/phasefunctions/Rayleigh.cl) exist as files.  Here it is a run-time choice. */
typedef enum prt_phase { PRT_PHASE_ISOTROPIC = 0, PRT_PHASE_HG = 1, PRT_PHASE_RAYLEIGH = 2 } prt_phase;

/*
 * The scene-specialisation parameters of the reference's kernel builder
 * (include/CL/cl_kernel.h:13-446 substitutes them into kernels/header.cl:39-122 as #defines).
 */
#define PRT_VIEW_RESULTS 0u
#define PRT_VIEW_NORMAL 1u
#define PRT_VIEW_BVH_HIT 16u

typedef struct prt_config {
    uint32_t abi_version;            /* PRT_ABI_VERSION */
    int32_t max_bounces;             /* cl_kernel.h:115-122   MAX_BOUNCES            (default 12) */
    int32_t max_diff_bounces;        /* :124-131              MAX_DIFF_BOUNCES       (4)  */
    int32_t max_spec_bounces;        /* :133-140              MAX_SPEC_BOUNCES       (4)  */
    int32_t max_trans_bounces;       /* :142-149              MAX_TRANS_BOUNCES      (12) */
    int32_t max_scattering_events;   /* :151-158              MAX_SCATTERING_EVENTS  (12) */
    int32_t marching_steps;          /* :160-167  MARCHING_STEPS of the raymarched SDF primitives (128) */
    int32_t shadow_marching_steps;   /* :169-176  SHADOW_MARCHING_STEPS (64) */
    uint32_t active_mats;            /* :226-345  ACTIVE_MATS: OR of all material type bits in the scene */
    uint32_t geom_flags;             /* :180-222  PRT_GEOM_* bits for H_SPHERE/H_BOX/H_SDF/H_QUAD */
    uint32_t light_count;            /* :367-400  LIGHT_COUNT */
    uint32_t light_indices[PRT_MAX_LIGHTS]; /*   LIGHT_INDICES (only [0] is ever sampled, base.cl:92 -- unless pick_random_light) */
    int32_t has_global_medium;       /* :47-54    GLOBAL_MEDIUM */
    float fog_density;               /* :66-75    values AFTER the "%f" round trip of std::to_string */
    float fog_sigma_a;               /* :77-84 */
    float fog_sigma_s;               /* :86-93 */
    float fog_sigma_t;               /* :95-102 */
    int32_t fog_abs_only;            /* :104-111 */
    int32_t alpha_testing;           /* :56-63    -alpha */
    int32_t phase_function;          /* prt_phase */
    float phase_g;                   /* HG asymmetry; the reference fixes 0.6 */
    uint32_t view_option;            /* kernels/main.cl:6-15 VIEW_OPTION: PRT_VIEW_RESULTS (0), or PRT_VIEW_NORMAL / PRT_VIEW_BVH_HIT:
                                      * every frame overwrites the accumulator with (ray.normal after radiance(), 1) and the image is the
                                      * accumulator itself (main.cl:143-145,150-152,161).  The other values of the reference's list are
                                      * refused: VIEW_STACK_INDEX does not compile there (Ray has no bvh_stackIndex), VIEW_ALBEDO and
                                      * VIEW_SPECULAR have no branch at all -- radiance() is never called and the image stays black */
    uint32_t pick_random_light;      /* kernels/integrators/base.cl:9 PICK_RANDOM_LIGHT (a source-level `#define ... 0` in the reference): 1 = the
                                      * light of lightSample / volumeLightSample is LIGHT_INDICES[(int)(next1D() * (LIGHT_COUNT + 1))]
                                      * (base.cl:88-90,202-204) -- one draw more per light sample, and an index one PAST the array with
                                      * probability 1 / (LIGHT_COUNT + 1).  That entry is defined here as 0 (mesh 0 is sampled as if it were
                                      * a light), which is what the reference build of the fixtures reads there (the array is declared one
                                      * element longer in the temporary text: zero-initialised).  Needs light_count < PRT_MAX_LIGHTS. */
    uint32_t env_importance_sampling;/* NOT in the reference (it only looks the map up where a ray escapes, pathtracing.cl:72): 1 = at every
                                      * vertex that samples the light (handleSurface, base.cl:168-172) the light-sample strategy is a coin flip
                                      * between LIGHT_INDICES[0] and the ENVIRONMENT MAP, sampled in proportion to its luminance x sin(theta)
                                      * and combined with the BSDF sample by the power heuristic; a BSDF-sampled ray that escapes adds the map
                                      * with the complementary weight and ends its path in that segment.  The expectation of every pixel is
                                      * the one of the default mode (a GPU test checks the converged pictures against each other); the random
                                      * number sequence is not, so this mode has no bit-exact counterpart.  Surfaces only: refused together
                                      * with a global medium, SDF primitives, a debug view or pick_random_light. */
} prt_config;

/* Host buffers of one scene, in the reference's layouts (src/main.cpp:93-122,401-418). */
typedef struct prt_scene_desc {
    const prt_mesh* meshes;          /* cl_meshes  src/main.cpp:418: order spheres, sdfs, boxes, quads */
    uint32_t object_count[8];        /* kernel arg 3 (cl_uint8): n_sphere,n_sdf,n_box,n_quad,_,_,_,total  include/Scene/scene.h:20-22 */
    const prt_material* obj_material;/* mBufMaterial src/main.cpp:403-404: ONE material for the whole OBJ (may be NULL if no OBJ) */
    const float* vertices;           /* mBufVertices :117  float4[3*T], de-indexed, xyz used */
    const float* normals;            /* mBufNormals  :118  float4[3*T] */
    const uint64_t* primitive_indices;/* mNewBufIndices :119 cl_ulong[T] */
    uint32_t triangle_count;         /* T */
    const prt_bvh_node* bvh_nodes;   /* mNewBufBVH :412   node 0 = root */
    uint32_t bvh_node_count;
} prt_scene_desc;

typedef struct prt_stats {
    double kernel_ms;        /* device time of the render kernels of the last prt_render_* call (HIP events on the context's
                                stream, around everything the call queued: wall time of the GPU work) */
    uint32_t launches;       /* kernel launches in that call */
    uint32_t frames;         /* frames (= segments per live pixel) executed in that call */
    uint64_t samples;        /* sum over pixels of RLH.samples (paths started)   -- filled by prt_query_counts */
    uint64_t segments;       /* sum over pixels of acc.w (segments executed)     -- filled by prt_query_counts */
    uint64_t finished_pixels;/* pixels frozen by the spp rule                     -- filled by prt_query_counts */
    double kernel_sum_ms;    /* sum of the durations of the individual launches (HIP events around each launch on the
                                stream it ran on; prt_render_spp only, else = kernel_ms).  The megakernel keeps
                                `concurrent` launches in flight (interleaved sets of tiles on internal streams), so
                                kernel_sum_ms ~ concurrent x kernel_ms; kernel_sum_ms / launches is what a profiler
                                reports as the kernel's average duration */
    uint32_t concurrent;     /* launches in flight at a time (internal streams; PRT_STREAMS, default 2) */
    uint32_t _pad;
} prt_stats;

typedef struct prt_ctx prt_ctx;

/* initOpenCL(), src/main.cpp:124-209 + cl_help::kernel::parse: pick the device and specialise
 * the integrator for one scene.  `device` is a HIP device ordinal. */
int prt_create(int device, const prt_config* cfg, prt_ctx** out);

/* process exit in the reference; explicit here. */
void prt_destroy(prt_ctx* ctx);

/* clw::buffer::create x5 + mBufMaterial, src/main.cpp:401-418,93-122.  Copies and re-packs. */
int prt_upload_scene(prt_ctx* ctx, const prt_scene_desc* scene);

/* enqueueWriteBuffer(cl_camera), src/main.cpp:294-297 (every frame in the reference). */
int prt_set_camera(prt_ctx* ctx, const prt_camera* cam);

/* cl_env_map, src/main.cpp:433-437 + include/GL/cl_gl_interop.h:71-86: RGB float, row 0 first.
 * Without a call the map is a 1x1 black texel. */
int prt_upload_envmap(prt_ctx* ctx, const float* rgb, int width, int height);

/* cl_flattenI = W*H*112 bytes, src/main.cpp:451; output texture tex0.  Implies prt_reset. */
int prt_resize(prt_ctx* ctx, int width, int height);

/* Multi-GPU row tile: this context renders rows [row0, row0+rows) of the width x full_height
 * image; seeds and camera use GLOBAL pixel coordinates so the union of tiles is bit-identical to
 * a single-context render.  State/framebuffer calls then address the tile only.  Implies reset. */
int prt_set_tile(prt_ctx* ctx, int width, int full_height, int row0, int rows);

/* Interleaved variant for load balance: the frame is cut into blocks of `block_rows` rows and
 * this context owns blocks part, part + n_parts, part + 2 n_parts, ... (local row order = global
 * row order of the owned rows).  State/framebuffer calls address the owned rows only. */
int prt_set_row_blocks(prt_ctx* ctx, int width, int full_height, int block_rows, int n_parts, int part);

/* buffer_reset branch of render(), src/main.cpp:283-291: zero the path state. */
int prt_reset(prt_ctx* ctx);

/* setArg(4,++framenumber); setArg(6,rand()); setArg(7,rand()); enqueueNDRangeKernel; finish --
 * src/main.cpp:299-304,260-261 -- batched: frames first_frame .. first_frame+n-1 (frame numbers
 * start at 1), seed_pairs = {random0, random1} per frame.  Every pixel advances one path segment
 * per frame.  Asynchronous on the context's stream; prt_synchronize / any read waits. */
int prt_render_frames(prt_ctx* ctx, uint32_t first_frame, uint32_t n_frames, const int32_t* seed_pairs);

/* "N spp": frames 1,2,... with a pixel frozen once its N-th path has terminated
 * (reset && samples == spp).  Runs until every pixel is frozen or max_frames frames were used;
 * seed_pairs must hold max_frames pairs.  *frames_used (optional) receives the frames launched: the frame count
 * of the slowest pixel rounded up to the launch size, at most max_frames.  A pixel's result depends on its own frames
 * only, so the launches let a pixel whose wave waits for slower neighbours start on the frames of the next launch
 * (the per-pixel lead lives in the device state for the duration of the call; PRT_RUN_AHEAD=0 turns it off); if
 * max_frames is reached (PRT_ERR_NOT_READY) every unfrozen pixel has done exactly max_frames.
 * Requires a freshly reset context. */
int prt_render_spp(prt_ctx* ctx, uint32_t spp, uint32_t max_frames, const int32_t* seed_pairs,
                   uint32_t* frames_used);

/* Scheduling knob of the render kernel (no counterpart in the reference; results do not depend on it, tests check
 * that): a wave ends a BVH-walk phase once fewer than `lanes` of its 64 lanes are still walking (and fewer than wait for the
 * phase to end); the lanes cut off resume in the wave's next phase.  1 = every walk runs to its end (lock step).
 * Default: 8 (6 with a global medium and for launches with scattered pixels, 20 through big trees) for the closest-hit phases
 * (PRT_WALK_MIN_LANES); for the shadow rays' any-hit
 * phases 1 in small trees and 12 in big ones (PRT_SHADOW_MIN_LANES).  A call of this function sets both. */
int prt_set_walk_min_lanes(prt_ctx* ctx, uint32_t lanes);

/* Build and schedule choices of a context (no counterpart in the reference; NONE changes a bit of any result -- the tests render
 * the goldens under each).  For tests, experiments and tuning.  prt_create reads the same names from the environment, in upper
 * case with the prefix PRT_.
 *   "waves"             0 (default: chosen per launch) | 5 | 6   build of the render kernel: waves per SIMD its registers leave room for
 *   "scatter"           -1 (default: chosen per launch) | 0 | 1  a wave renders one 8x8 tile | 64 pixels scattered over the launch's tiles
 *   "generic"           0 | 1   1: the material set dispatched at run time even where the scene's own ACTIVE_MATS is compiled (the
 *                               reference compiles exactly the scene's set, include/CL/cl_kernel.h:226-345; compiled here: LIGHT|DIFF,
 *                               +COAT, +ROUGH_COND, +DIEL|ROUGH_DIEL)
 *   "any_dist"          0 | 1   1: the compiled set's instance that carries all three microfacet distributions even where every microfacet
 *                               lobe of the scene uses the same one (`mat->dist` is a run-time field in the reference,
 *                               kernels/bxdf/microfacet.cl:6-9; compiled here: GGX for the rough sets, Beckmann for the coat set)
 *   "walk_min_lanes", "shadow_min_lanes"   0 (by launch) .. 64   see prt_set_walk_min_lanes
 *   "tri_q"             0 .. 16  the triangle tests that a walk phase's box steps found run once this many sixteenths of its walking
 *                               lanes have one pending (default 4)
 *   "frames_per_launch" >= 0    frames one launch of the render kernel covers (0 = default: 512, or 4096 through a tree of more than 64 k node pairs)
 *   "run_ahead"         0 | 1   prt_render_spp: see there
 *   "pace"              1 | 0   prt_render_spp: a pixel whose paths are longer than the frame's average owes every launch proportionally more frames
 *                               (it needs proportionally more for its samples; what it does not do while the chip is full it does in the tail of the
 *                               render, alone); 0: every pixel owes a launch the same number of frames
 *   "tile_order"        1 | 0   prt_render_spp starts the tiles whose waves ran longest in a sub-part's first launch first in its later
 *                               launches (and renders, until scene, camera or frame change; setting the option to the value it has
 *                               keeps a measured order); 0: in index order
 *   "pix_per_wave"      0 (default: chosen per launch) | 64 | 32 | 16   pixels a wave of the render kernel renders (its other lanes idle).  Launches
 *                               that leave wave slots of the chip empty -- one rank's share of a frame split N ways, a small frame -- finish sooner
 *                               with more waves of fewer pixels: a wave lasts as long as the slowest of its pixels' chains of segments
 *   "pool"              0 | 1   1: render_kernel_rp (csrc/hip/pt_pool.h): workgroups of shading waves that post the rays that go deeper than the root
 *                               of the tree to walker waves through LDS.  Bit-exact, measured slower than the default kernel: off
 *   "test_drop_report"  0 | 1   tests only: the launches of prt_render_spp report their unfinished pixels into a spare word, so that the
 *                               call sees a launch end without a report (PRT_ERR_HIP, state unusable until prt_reset) */
int prt_set_option(prt_ctx* ctx, const char* name, int value);

/* What this library was built from (no counterpart in the reference): a hash of the content of every source, header and compiler flag
 * of libprt.so, compiled in by the build (photorealistic-rendering-using-opencl_amd/build.py source_build_id(); a development variant of
 * tools/build_variant.sh reads "variant-<name>-<hash>").  bench.py prints it as `build_id` and refuses to time a library whose id is
 * not the working tree's; tests/conftest.py asserts the same.  Needs no device and no context. */
const char* prt_build_id(void);
/* what the last launch ran, as text: "render_kernel<LIGHT|DIFF> waves=6 pixels=tiles" ("" before the first launch; "pixels=scattered",
 * "pixels=tiles, expensive first": see prt_set_option) */
const char* prt_kernel_variant(prt_ctx* ctx);

int prt_synchronize(prt_ctx* ctx);

/* output texture: linear float4 acc/samples per pixel (kernels/main.cl:159).  Row 0 is the BOTTOM of the
 * picture, as in the reference's GL texture (the camera maps coord.y = 0 to the lowest scan line,
 * kernels/camera.cl:29-35; its PNG writer flips on write, include/GL/cl_gl_interop.h:139). */
int prt_read_framebuffer(prt_ctx* ctx, float* rgba);
/* display side ("next" row N3): the reference's fragment shader (shaders/tonemapper.glsl:47-64: vignette,
 * filmic Reinhard with white point 1.2, smoothstep, gamma 2.2) applied on the device; 8-bit RGBA out, rows in
 * framebuffer order (what glReadPixels returns, include/GL/cl_gl_interop.h:147-150). */
int prt_tonemap_rgba8(prt_ctx* ctx, uint8_t* rgba);
/* same, device to device, into caller-owned device memory (e.g. a torch tensor).  Asynchronous on a stream given
 * with prt_set_stream (ordered with the caller's other work there); complete on return otherwise (the context's own
 * stream is private and non-blocking: nothing of the caller's is ordered against it). */
int prt_copy_framebuffer_to_device(prt_ctx* ctx, void* device_rgba);

/* r_flat, in the reference's 112-byte RTD layout (checkpoint / resume / parity checks). */
int prt_read_state(prt_ctx* ctx, prt_path_state* state);
int prt_write_state(prt_ctx* ctx, const prt_path_state* state);

/* run on a caller-provided hipStream_t (NULL = the context's own stream) */
int prt_set_stream(prt_ctx* ctx, void* hip_stream);

int prt_get_stats(prt_ctx* ctx, prt_stats* stats);
/* device-side reduction of samples / segments / frozen pixels (fills those prt_stats fields) */
int prt_query_counts(prt_ctx* ctx, uint32_t spp, prt_stats* stats);

/* Diagnostics: evaluates one function of include/prt_detmath.h ON THE DEVICE for n inputs
 * (fn: 0 sin, 1 cos, 2 tan, 3 exp, 4 log, 5 acos, 6 atan2(a,b), 7 pow(a,b), 8 sqrt, 9 a/b,
 * 10 fma(a,b,a), 11 fmin(a,b), 12 fmax(a,b), 13 round, 14 floor, 15 1/a, 16 cbrt).  Host arrays in and out.
 * The numerics contract says the result must equal the host evaluation bit for bit. */
int prt_selftest_math(prt_ctx* ctx, int fn, const float* a, const float* b, float* out, int n);
/* test hook: one device FUNCTION of the radiance loop on `n` cases (BSDF sampling / evaluation, microfacet terms,
 * Fresnel, light sampling, medium and phase sampling, camera ray, primitive tests, environment lookup -- fn 1..11, layouts in
 * csrc/hip/pt_selftest.h): 80 floats of shared parameters, 32 floats in and 32 floats out per case.  The known-answer
 * fixtures tests/golden/kat_*.npz hold what the REFERENCE's own functions return on the same cases. */
int prt_selftest_fn(prt_ctx* ctx, int fn, const float* params, const float* in, float* out, int n);

const char* prt_last_error(prt_ctx* ctx);
/* message for a failed prt_create (ctx == NULL) */
const char* prt_last_global_error(void);

#ifdef __cplusplus
}
#endif
#endif /* PRT_H */
