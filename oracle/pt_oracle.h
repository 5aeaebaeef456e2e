/* oracle/pt_oracle.h -- TEST INFRASTRUCTURE: interface of the plain-C restatement (pt_oracle.c). */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H
#include "prt.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct pto_job {
    const prt_config* cfg;
    const prt_scene_desc* scene;
    const prt_camera* camera;
    const float* env_rgb; int env_w, env_h;     /* NULL = 1x1 black */
    int width, full_height, row0, rows;         /* tile = `rows` local rows of width x full_height */
    int block_rows, n_parts, part;              /* local row ly -> global row0 + (ly / B * n_parts + part) * B + ly % B (B = 0: contiguous) */
    uint32_t first_frame, n_frames;             /* frame numbers start at 1 */
    const int32_t* seed_pairs;                  /* 2*n_frames */
    prt_path_state* state;                      /* width*rows, in/out */
    float* out_rgba;                            /* width*rows*4, out */
    uint32_t spp_limit;                         /* 0 = progressive; N = freeze after the N-th path */
    int n_threads;
} pto_job;

typedef struct pto_diag { int max_stack, max_shadow_stack; } pto_diag;

int pto_render(const pto_job* job, pto_diag* diag);
float pto_medium_lane3(const prt_config* cfg);

#ifdef __cplusplus
}
#endif
#endif
