/*
 * prt_types.h -- binary layouts shared between the host of the reference renderer and the prt
 * C-ABI.  Every struct here is byte-compatible with the struct the reference uploads to / keeps
 * on the device; offsets are the ones both the reference host headers and its OpenCL structs
 * agree on (checked in this container):
 *
 *   prt_material  64 B   include/Types/material.h:95-120      kernels/header.cl:219-234
 *   prt_mesh     256 B   include/Scene/geometry.h:21-28       kernels/header.cl:238-247
 *   prt_bvh_node  36 B   include/BVH/bvh.h:24-30              kernels/header.cl:256-261
 *   prt_camera    80 B   include/Camera/camera.h:7-15         kernels/camera.cl:7-15
 *   prt_path_state 112 B src/main.cpp:39 (RayI_size)          kernels/main.cl:30-47,58-61 (RTD)
 */
#ifndef PRT_TYPES_H
#define PRT_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* material type bits (include/Types/material.h:9-32) */
enum {
    PRT_MAT_LIGHT = 1 << 0, PRT_MAT_DIFF = 1 << 1, PRT_MAT_COND = 1 << 2, PRT_MAT_DIEL = 1 << 3,
    PRT_MAT_COAT = 1 << 4, PRT_MAT_VOL = 1 << 5, PRT_MAT_TRANS = 1 << 6, PRT_MAT_SPECSUB = 1 << 7,
    PRT_MAT_ABS_REFR = 1 << 8, PRT_MAT_ABS_REFR2 = 1 << 9, PRT_MAT_ROUGH_COND = 1 << 10,
    PRT_MAT_ROUGH_DIEL = 1 << 11
};
/* lobes (include/Types/material.h:36-54, kernels/header.cl:92-111) */
enum {
    PRT_LOBE_NULL = 0, PRT_LOBE_GLOSSY_R = 1 << 0, PRT_LOBE_GLOSSY_T = 1 << 1, PRT_LOBE_DIFFUSE_R = 1 << 2,
    PRT_LOBE_DIFFUSE_T = 1 << 3, PRT_LOBE_SPECULAR_R = 1 << 4, PRT_LOBE_SPECULAR_T = 1 << 5,
    PRT_LOBE_ANISO = 1 << 6, PRT_LOBE_FORWARD = 1 << 7,
    PRT_LOBE_SPECULAR = (1 << 4) | (1 << 5),
    PRT_LOBE_TRANSMISSIVE = (1 << 1) | (1 << 3) | (1 << 5)
};
/* microfacet distributions (kernels/bxdf/microfacet.cl:6-9); JSON "dist": n -> 1 << n */
enum { PRT_DIST_BECKMANN = 1 << 0, PRT_DIST_PHONG = 1 << 1, PRT_DIST_GGX = 1 << 2 };
/* geometry type bits of prt_mesh.t as the HOST writes them (include/Scene/geometry.h:9-13).
 * NB the host's BOX/SDF values are what the kernel is specialised with (cl_kernel.h:180-222). */
enum { PRT_GEOM_SPHERE = 1 << 0, PRT_GEOM_BOX = 1 << 1, PRT_GEOM_SDF = 1 << 2, PRT_GEOM_QUAD = 1 << 3 };

typedef struct prt_material {
    float color[4];      /* @0  color / emission / albedo (xyz) */
    float eta[4];        /* @16 */
    float k[4];          /* @32 */
    float roughness;     /* @48 */
    uint16_t t;          /* @52 material type bits */
    uint8_t lobes;       /* @54 */
    uint8_t dist;        /* @55 */
    uint8_t _pad[8];     /* @56 */
} prt_material;

typedef struct prt_mesh {
    prt_material mat;    /* @0   */
    float pos[4];        /* @64  */
    uint8_t _pad0[48];   /* @80  */
    float joker[16];     /* @128 sphere: [0]=radius; quad: base[0-2] edge0[3-5] edge1[6-8] normal[9-11] area[12] */
    uint8_t t;           /* @192 geometry type bits */
    uint8_t _pad1[63];   /* @193 */
} prt_mesh;

typedef struct prt_bvh_node {
    float bounds[6];                    /* min_x max_x min_y max_y min_z max_z */
    uint32_t first_child_or_primitive;  /* inner: index of the left child (right = +1); leaf: first slot in the index array */
    uint32_t primitive_count;
    uint8_t is_leaf;
    uint8_t _pad[3];
} prt_bvh_node;

typedef struct prt_camera {
    float position[4];   /* @0  */
    float view[4];       /* @16 */
    float up[4];         /* @32 */
    float resolution[2]; /* @48 */
    float fov[2];        /* @56 degrees */
    float apertureRadius;/* @64 */
    float focalDistance; /* @68 */
    uint8_t _pad[8];     /* @72 */
} prt_camera;

/* RTD = { TempRay ray; RLH data; } -- the per-pixel path state that lives between launches */
typedef struct prt_path_state {
    float origin[4];     /* @0   TempRay.origin */
    float dir[4];        /* @16  TempRay.dir    */
    float time;          /* @32  */
    float dist;          /* @36  */
    uint8_t _pad0[8];    /* @40  */
    float mask[4];       /* @48  RLH.mask (xyz) */
    float acc[4];        /* @64  RLH.acc  */
    uint32_t total;      /* @80  bounce.total */
    uint16_t diff, spec, trans, scatters; /* @84..@91 */
    uint8_t was_specular;/* @92  */
    uint8_t _pad1[3];
    uint8_t reset;       /* @96  */
    uint8_t _pad2[3];
    uint32_t samples;    /* @100 */
    uint8_t _pad3[8];    /* @104 */
} prt_path_state;

#ifdef __cplusplus
}
static_assert(sizeof(prt_material) == 64, "Material ABI");
static_assert(sizeof(prt_mesh) == 256, "Mesh ABI");
static_assert(sizeof(prt_bvh_node) == 36, "cl_BVHnode ABI");
static_assert(sizeof(prt_camera) == 80, "Camera ABI");
static_assert(sizeof(prt_path_state) == 112, "RTD ABI");
static_assert(__builtin_offsetof(prt_mesh, joker) == 128 && __builtin_offsetof(prt_mesh, t) == 192, "Mesh offsets");
static_assert(__builtin_offsetof(prt_path_state, reset) == 96 && __builtin_offsetof(prt_path_state, samples) == 100, "RTD offsets");
#endif

#endif /* PRT_TYPES_H */
