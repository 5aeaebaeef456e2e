/* host_capi.h -- C exports of the host-side model (scene loader, OBJ reader, BVH builder,
 * camera, seed protocol) so that non-C++ hosts (the Python test/bench plumbing) can build the
 * exact buffers the prt C-ABI consumes.  Not part of the drop-in boundary itself (that is
 * include/prt.h); it mirrors what src/main.cpp:372-427 does before the first enqueue. */
#ifndef PRT_HOST_CAPI_H
#define PRT_HOST_CAPI_H
#include "prt.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct prth_scene prth_scene;

/* host_scene::load + ModelLoader::ImportFromFile(models_dir + obj_path) + BVH(ml)
 * (src/main.cpp:375-415).  Returns NULL and fills err on failure. */
prth_scene* prth_scene_load(const char* scene_json_path, const char* models_dir, char* err, int err_len);
/* same from JSON text */
prth_scene* prth_scene_load_text(const char* scene_json_text, const char* models_dir, char* err, int err_len);
void prth_scene_free(prth_scene* s);
/* pointers inside *out stay valid until prth_scene_free */
int prth_scene_get_desc(const prth_scene* s, prt_scene_desc* out);
int prth_scene_get_config(const prth_scene* s, int alpha_testing, prt_config* out);
int prth_scene_bvh_depth(const prth_scene* s);
const char* prth_scene_obj_path(const prth_scene* s);

/* initCamera() + buildRenderCamera(): default InteractiveCamera, setResolution(w,h), setFOVX(fovx)
 * (src/main.cpp:312-319, src/Camera/camera.cpp:4-12,88-104) */
int prth_default_camera(int width, int height, float fovx_degrees, prt_camera* out);
/* orbit-camera variant: yaw/pitch offsets and radius factor applied through the InteractiveCamera API */
int prth_orbit_camera(int width, int height, float fovx_degrees, float d_yaw, float d_pitch, float d_radius,
                      float d_aperture, float d_focal, prt_camera* out);

/* The reference never seeds rand(): glibc's default stream (seed 1).  initCLKernel consumes two
 * values (src/main.cpp:226-227); frame f (1-based) then uses values 2f+1, 2f+2 (src/main.cpp:
 * 301-302).  Fills 2*n_frames ints for frames first_frame.. */
int prth_seed_pairs(uint32_t first_frame, uint32_t n_frames, int32_t* out_pairs);

/* OBJ -> ".prtmesh" soup conversion (tools / fixtures) */
int prth_convert_model(const char* in_path, const char* out_soup_path, char* err, int err_len);

/* the meshes of a model file (an OBJ file's objects / groups / materials, in file order: src/Models/model_loader.cpp:58-74 walks them all):
 * returns their number (negative: error) and fills, per mesh m < max_meshes, counts[3m] = triangles, counts[3m + 1] = vertices left after
 * welding identical (position, normal) pairs (aiProcess_JoinIdenticalVertices of the reference's import preset, model_loader.cpp:38),
 * counts[3m + 2] = 1 if de-indexing the welded mesh gives the soup back float for float */
int prth_model_meshes(const char* path, uint32_t* counts, int max_meshes, char* err, int err_len);

/* deterministic procedural sky used as the HDR environment stand-in (no .hdr ships with the
 * reference): width x height RGB float, row 0 = top (v = 0) */
int prth_make_sky(int width, int height, float* rgb);
/* loadHDR (include/Texture/texture.h:31-39): a Radiance .hdr file as width x height x 3 floats, rows in file order.
 * Returns a handle (prth_hdr_free) or NULL with the reason in err. */
void* prth_hdr_load(const char* path, int* width, int* height, const float** rgb, char* err, int err_len);
void prth_hdr_free(void* handle);
/* the reference's `-encoder 1` (saveImage -> stbi_write_hdr, include/GL/cl_gl_interop.h:151-156): `channels` (>= 3) floats per pixel,
 * rows bottom-up (bottom_up != 0: the framebuffer's order) or top-down; writes a run-length encoded Radiance picture.  0 on success. */
int prth_hdr_write(const char* path, const float* pixels, int width, int height, int channels, int bottom_up, char* err, int err_len);

#ifdef __cplusplus
}
#endif
#endif
