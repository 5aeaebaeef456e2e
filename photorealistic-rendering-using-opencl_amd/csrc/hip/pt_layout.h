// pt_layout.h -- how a scene and the per-pixel path state live in HBM on the MI355X side of the
// prt C-ABI.  prt_upload_scene() re-packs the reference-layout host buffers (36-byte BVH nodes,
// float4 vertex/normal soups behind a u64 index array, 256-byte Mesh records) into these
// 16-byte-aligned records so that every device access is a dwordx4 load:
//
//   NodePair   64 B  one per INNER node: bounds of both children + what each child is.
//              The reference walks `node -> children[first_child + {0,1}]` and tests both children
//              per step (kernels/geometry/bvh.cl:144-196); storing the pair together turns two
//              36-byte unaligned gathers into four aligned 16-byte loads.
//   TriGeom    48 B  per leaf SLOT (triangles pre-gathered in primitive_indices order, so the
//              u64 index indirection of kernels/geometry/triangle.cl:7 is gone): p0, e1 = p0-p1,
//              e2 = p2-p0, n = cross(e1,e2) -- the exact float operations of triangle.cl:8-15 done
//              once on the host (no contraction), hence identical bits.
//   TriNrm     48 B  per slot: the three vertex normals, touched once per closest hit.
//   prims      spheres 16 B, quads 80 B (with the ray-independent anchor and edge dot products
//              of kernels/geometry/quad.cl:16,26-27 pre-computed by the same arithmetic).
//   DevMaterial 48 B per mesh (+ guard at [0], OBJ material last).
//
// Path state: 5 float4 planes (SoA of 16-byte lanes), 80 B per pixel instead of the reference's
// 112-byte AoS RTD; prt_read_state()/prt_write_state() convert to and from the RTD layout.
#pragma once
#include <stdint.h>

namespace prt {

struct alignas(16) NodePair {
    // b0 = {c0.minx, c0.maxx, c0.miny, c0.maxy}, b1 = {c0.minz, c0.maxz, c1.minx, c1.maxx},
    // b2 = {c1.miny, c1.maxy, c1.minz, c1.maxz}
    float b[12];
    // child k: meta[2k]   = inner: index of the child's NodePair; leaf: first slot
    //          meta[2k+1] = inner: 0xFFFFFFFF;                    leaf: primitive count
    uint32_t meta[4];
};
static_assert(sizeof(NodePair) == 64, "NodePair");

struct alignas(16) TriGeom { float p0[3]; float e1[3]; float e2[3]; float n[3]; };
struct alignas(16) TriNrm { float n0[4]; float n1[4]; float n2[4]; };
static_assert(sizeof(TriGeom) == 48 && sizeof(TriNrm) == 48, "Tri records");

struct alignas(16) DevSphere { float pos[3]; float radius; };
struct alignas(16) DevQuad {
    float base[3]; float area;
    float edge0[3]; float e0e0;
    float edge1[3]; float e1e1;
    float normal[3]; float u0;     // u0 = e0e0 * 2^-24, u1 = e1e1 * 2^-24: half an ulp of 1.0 in units of the divisor, or a
    float anchor[3]; float u1;     // NaN when the divisor is outside [2^-40, 2^40] (out_of_unit_range then divides)
};
// Mesh.joker {base, edge0, edge1, normal, area} (include/Scene/scene.h:434-452) -> DevQuad, with the ray-independent
// values of kernels/geometry/quad.cl:16,26-27 (anchor = base - (edge0 + edge1) * 0.5f, dot(edge, edge)) computed once
// by the reference's own float operations, and u = half an ulp of 1.0 in units of the divisor for out_of_unit_range
#if defined(__HIPCC__)
#define PT_LAYOUT_HD __host__ __device__ inline
#else
#define PT_LAYOUT_HD inline
#endif
PT_LAYOUT_HD float quad_half_ulp(float c) { return (c >= 9.094947017729282e-13f && c <= 1099511627776.0f) ? c * 5.9604644775390625e-08f : __builtin_nanf(""); }
PT_LAYOUT_HD void pack_quad(const float* j, DevQuad& d) {
    for (int k = 0; k < 3; ++k) { d.base[k] = j[k]; d.edge0[k] = j[3 + k]; d.edge1[k] = j[6 + k]; d.normal[k] = j[9 + k]; }
    d.area = j[12];
    for (int k = 0; k < 3; ++k) d.anchor[k] = d.base[k] - (d.edge0[k] + d.edge1[k]) * 0.5f;
    d.e0e0 = d.edge0[0] * d.edge0[0] + d.edge0[1] * d.edge0[1] + d.edge0[2] * d.edge0[2];
    d.e1e1 = d.edge1[0] * d.edge1[0] + d.edge1[1] * d.edge1[1] + d.edge1[2] * d.edge1[2];
    d.u0 = quad_half_ulp(d.e0e0); d.u1 = quad_half_ulp(d.e1e1);
}

struct alignas(16) DevSdf { float pos[3]; uint32_t type; float params[4]; };     // Mesh.pos, Mesh.t, joker.s0123
struct alignas(16) DevMaterial {
    float color[3]; float roughness;
    float eta[3]; uint32_t bits;      // t | lobes << 16 | dist << 24
    float k[3]; float _p;
};
static_assert(sizeof(DevSphere) == 16 && sizeof(DevQuad) == 80 && sizeof(DevMaterial) == 48, "prim records");

// kernel argument block (passed by value: lives in the kernarg segment, read with scalar loads)
struct DevScene {
    const NodePair* pairs;          // breadth-first order
    uint32_t n_pairs;
    const TriGeom* tri_geom;
    const TriNrm* tri_nrm;
    const DevSphere* spheres;
    const DevQuad* quads;
    const DevSdf* sdfs;             // raymarched primitives (mesh indices n_spheres .. n_spheres + n_sdfs - 1)
    const DevMaterial* mats;        // [0] guard (zero), [1+i] mesh i, [1+n_meshes] OBJ material
    const float* env;               // RGB float
    int env_w, env_h;
    // prt_config::env_importance_sampling (the PT_MATS_ENVIS kernel variants): the map's sampling density as cumulative sums --
    // env_cdf_rows[j], j = 0 .. env_h: rows 0 .. j - 1; env_cdf_cols[j * (env_w + 1) + i]: texels 0 .. i - 1 of row j (each ends at 1)
    const float* env_cdf_rows;
    const float* env_cdf_cols;
    uint32_t n_spheres, n_quads, quad_mesh_base, n_meshes, n_sdfs;
    int marching_steps, shadow_marching_steps;
    unsigned stack_levels;          // LDS traversal-stack levels per lane (most entries any walk can hold, + 1)
    uint32_t root_leaf_first, root_leaf_count;   // used when root_is_leaf
    int root_is_leaf;
    uint32_t light_sphere;          // LIGHT_INDICES[0] as index into spheres, or 0xFFFFFFFF
    uint32_t light_quad;            // ... as index into quads, or 0xFFFFFFFF
    uint32_t light_mesh;            // LIGHT_INDICES[0]
    // prt_config::pick_random_light (base.cl:9 PICK_RANDOM_LIGHT; the PT_MATS_PICK kernel variants): mesh index per entry of LIGHT_INDICES
    // and, at [light_count], of the entry behind it (0)
    const uint32_t* light_tab;
    uint32_t light_count, pick_random_light;
    uint32_t env_is;                // prt_config::env_importance_sampling
    // prt_config
    uint32_t active_mats, geom_flags;
    uint32_t dist_mask;             // PRT_DIST_* bits of the materials that have a microfacet lobe (coat, rough conductor, rough dielectric)
    int max_bounces, max_diff_bounces, max_spec_bounces, max_trans_bounces, max_scattering_events;
    int has_medium, fog_abs_only, alpha_testing, phase_function;
    float fog_sigma_s, fog_sigma_t, phase_g;
    uint32_t ntrans_mask;
    uint32_t view;                  // prt_config::view_option is PRT_VIEW_NORMAL or PRT_VIEW_BVH_HIT (the PT_MATS_VIEW kernel variants)
};

// camera.cl:19-28 evaluated once per prt_set_camera (pt_device.h camera_basis), not per path start
struct DevCamera {
    float position[3], apertureRadius;
    float hAxis[3], focalDistance;
    float vAxis[3], pad0;
    float middle[3], pad1;
    float horizontal[3], pad2;
    float vertical[3], pad3;
};

// path state planes
struct DevState {
    float4* q0;   // origin.xyz, time
    float4* q1;   // dir.xyz, dist
    float4* q2;   // mask.xyz, bits(total)
    float4* q3;   // acc
    uint4* q4;    // samples, diff | spec << 16, trans | scatters << 16, was_specular | reset << 1 | frames ahead << 2
};

struct FrameArgs {
    int width, full_height, row0, rows;     // tile of the image: `rows` local rows
    int block_rows, n_parts, part;          // local row ly -> global row row0 + (ly / B * n_parts + part) * B + ly % B
    uint32_t first_frame, n_frames;
    const int32_t* seed_pairs;              // device, 2 * seed_frames
    uint32_t seed_frames;                   // >= n_frames: frames from first_frame on that have seeds
    uint32_t run_ahead;                     // "N spp" launches: a lane that has done its n_frames goes on (up to seed_frames) for as long
                                            // as its wave waits for other lanes; how far it got is kept per pixel (DevState::q4.w >> 2)
    float pace_inv_ref;                     // "N spp" launches with run_ahead: 0, or 1 / (the frame's mean path length so far).  A pixel whose own mean
                                            // path length is longer owes the launches proportionally more than n_frames frames each (cumulatively; up to 2 x, within the seed
                                            // table): it needs proportionally more frames for its samples, and what it does not do while the chip is full
                                            // it does in the tail of the render, alone (DESIGN.md s4 "Round 4: the tail").  Schedule only: no bit depends on it
    uint32_t spp_limit;
    unsigned long long* unfinished;         // device counter: pixels not yet frozen (spp mode), or null
    unsigned long long* unfinished_host;    // null: the host reads `unfinished` itself.  Else `unfinished` is a pair {count, waves
                                            // done}; the last wave of the launch writes the count here (pinned host memory) and
                                            // zeroes the pair for the next launch: no copy / fill kernels between launches
    uint32_t tile_first, tile_stride;       // this launch covers the tiles tile_first + k * tile_stride (sub-part of the frame)
    const uint32_t* tile_order;             // null, or a permutation of the launch's k: workgroup g renders tile k = tile_order[g] (the expensive
                                            // tiles first: the hardware starts workgroups in index order, and a launch ends with its last tile)
    uint32_t* tile_cost;                    // null, or where the wave of tile k leaves the iterations it took (tile_cost[k]): what the tile costs
    uint32_t sub_shift;                     // set by the launcher: a wave renders 64 >> sub_shift pixels (its other lanes idle): 2^sub_shift waves per tile.
                                            // For launches that leave wave slots of the chip empty (one rank's share of a frame): pixels are independent, so
                                            // more waves with fewer pixels each finish sooner -- a wave lasts as long as its slowest lane
    uint32_t scatter;                       // set by the launcher: lane l of wave g renders pixel l of tile (l * waves + g) / 64 ...
    uint32_t walk_min_lanes;                // lane machine: a closest-hit walk phase of a wave ends once fewer lanes than this are still walking
    uint32_t shadow_min_lanes;              // ... and an any-hit (shadow ray) phase below this many
    uint32_t tri_sixteenths;                // the pending triangle tests of a walk phase run once this many sixteenths of its walking lanes have one
};

}  // namespace prt
