// pt_device.h -- device functions of the MI355X path-tracing integrator (gfx950, wave64).
//
// One work-item owns one pixel and advances it by one path SEGMENT per frame, exactly like the
// reference's render_kernel (kernels/main.cl:66-163) -- but the frame loop runs INSIDE the kernel
// with the path state in registers, so the 112 B/px state round trip and the 16 B/px image write
// of the reference happen once per launch instead of once per frame.  Seeds are re-derived per
// (x, y, frame, seed pair) as in main.cl:108-109, which is what makes pixels independent and the
// batching legal.
//
// Arithmetic contract: every floating-point value is produced by the same sequence of IEEE
// binary32 operations as the reference expression it cites (component-wise vectors, left to
// right, no contraction: the file is compiled with -ffp-contract=off) with include/prt_detmath.h
// as the OpenCL built-in library.  That is what makes the output bit-identical to the CPU
// oracle; do not "simplify" an expression here without re-reading the cited reference line.
#pragma once
#include <hip/hip_runtime.h>

#include "prt_detmath.h"
#include "prt_types.h"
#include "pt_layout.h"

namespace prt {
namespace dev {

#define PT_EPS 1e-5f
#define PT_INF 2e1f
#define PT_PI 3.1415926535897932384626433832795f
#define PT_TWO_PI 6.283185307179586476925286766559f
#define PT_INV_PI 0.3183098861837906715377675267450f
#define PT_INV_TWO_PI 0.1591549430918953357688837633725f
#define PT_INV_FOUR_PI 0.0795774715459476678844418816863f

#ifdef PT_EMU                                            // tests/emu: the same functions compiled for the host (checker of the
#define PT_DEV __host__ __device__ __forceinline__      // lane machine's schedule independence), never part of libprt
#else
#define PT_DEV __device__ __forceinline__
#endif
#define PT_HD __host__ __device__ __forceinline__      // the few helpers the host shares (camera basis)

// 1/x, correctly rounded (= prt_recip, the IEEE divide the CPU side does), in 6 vector instructions instead of the
// 11 of the compiler's divide expansion: hardware estimate + one fma Newton step is exact whenever neither x nor
// 1/x is subnormal (checked over all 2^32 inputs by prt_selftest_math fn 17, tests/test_gpu_parity.py); the rest
// (zero, subnormal, huge, inf, nan) takes the divide.
#ifndef PT_RECIP_STEPS
#define PT_RECIP_STEPS 1
#endif
PT_HD float hw_recip(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned e = (prt_f2u(x) >> 23) & 0xffu;
    if (e - 2u < 251u) {                                        // 2^-125 <= |x| < 2^126
        float r = __builtin_amdgcn_rcpf(x);
        for (int k = 0; k < PT_RECIP_STEPS; ++k) r = __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
        return r;
    }
#endif
    return 1.0f / x;
}
// sqrt(x), correctly rounded (= prt_sqrt, the IEEE square root the CPU side computes), without the 17 vector instructions of the
// compiler's expansion (scaling for subnormals, two neighbour tests, class fix-up): hardware estimate of 1/sqrt(x), s0 = x * r and one
// fma correction step s0 + (x - s0 * s0) * r / 2 in the range where none of it needs care (5 instructions + the range test); everything else takes the IEEE one.  Checked over all 2^32 inputs by prt_selftest_math
// fn 19 (tests/test_gpu_parity.py).
PT_HD float hw_sqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned e = (prt_f2u(x) >> 23) & 0x1ffu;              // sign and exponent
    if (e - 27u < 200u) {                                        // positive, 2^-100 <= x < 2^100
        const float r = __builtin_amdgcn_rsqf(x);
        const float s0 = x * r, hh = 0.5f * r;
        return __builtin_fmaf(__builtin_fmaf(-s0, s0, x), hh, s0);
    }
#endif
    return __builtin_sqrtf(x);
}
struct f3 { float x, y, z; };
PT_HD f3 F3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_HD f3 splat(float s) { return F3(s, s, s); }
PT_HD f3 operator+(f3 a, f3 b) { return F3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_HD f3 operator-(f3 a, f3 b) { return F3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_HD f3 operator*(f3 a, f3 b) { return F3(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_HD f3 operator*(f3 a, float s) { return F3(a.x * s, a.y * s, a.z * s); }
PT_HD f3 operator/(f3 a, float s) { return F3(a.x / s, a.y / s, a.z / s); }
PT_HD f3 operator-(f3 a) { return F3(-a.x, -a.y, -a.z); }
PT_HD float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PT_HD f3 cross(f3 a, f3 b) { return F3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
PT_HD float length(f3 a) { return hw_sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
PT_HD f3 normalize(f3 a) { float inv = hw_recip(hw_sqrt(a.x * a.x + a.y * a.y + a.z * a.z)); return F3(a.x * inv, a.y * inv, a.z * inv); }
PT_HD f3 ld3(const float* p) { return F3(p[0], p[1], p[2]); }

// ---- uniform scene tables: scalar loads BY CONSTRUCTION ------------------------------------------------------------------------
// The quads, spheres, SDF primitives, the materials, the light table, the seed table and the root of the tree are read at addresses
// that are (mostly) the same in every lane.  Through an ordinary global pointer such a load is given to the scalar unit only while the
// compiler can prove that nothing in the kernel wrote before it -- one LDS atomic, one `asm volatile`, one time stamp in front of the
// frame loop and every one of them silently became a vector load (11 ... 19 % slower, bit-exact; round 3).  The kernel never writes these
// tables, so they are read through the CONSTANT address space (AMDGPU address space 4: memory that does not change while the kernel
// runs): a load from it at a uniform address is an s_load whatever else the kernel does, at a divergent address the same global load as
// before.  tests/test_codegen.py compiles the headline set with a deliberate clobber in front of the frame loop and counts.
#if defined(__HIP_DEVICE_COMPILE__)
#define PT_CONST __attribute__((address_space(4)))
#else
#define PT_CONST                                       // host compilations (tests/emu, the camera basis): plain memory
#endif
typedef float pt_vf4 __attribute__((ext_vector_type(4)));
typedef unsigned pt_vu4 __attribute__((ext_vector_type(4)));
template <typename T> PT_HD const PT_CONST pt_vf4* const_vf4(const T* p) { return (const PT_CONST pt_vf4*)p; }
PT_HD unsigned const_u32(const uint32_t* p, unsigned i) { return ((const PT_CONST uint32_t*)p)[i]; }
PT_HD int const_i32(const int32_t* p, unsigned i) { return ((const PT_CONST int32_t*)p)[i]; }
PT_HD float const_f32(const float* p, size_t i) { return ((const PT_CONST float*)p)[i]; }
PT_HD DevSphere const_sphere(const DevSphere* p, unsigned i) {
    const pt_vf4 a = const_vf4(p + i)[0];
    DevSphere s;
    s.pos[0] = a.x; s.pos[1] = a.y; s.pos[2] = a.z; s.radius = a.w;
    return s;
}
PT_HD DevQuad const_quad(const DevQuad* p, unsigned i) {
    const PT_CONST pt_vf4* q = const_vf4(p + i);
    const pt_vf4 a = q[0], b = q[1], c = q[2], d = q[3], e = q[4];
    DevQuad r;
    r.base[0] = a.x; r.base[1] = a.y; r.base[2] = a.z; r.area = a.w;
    r.edge0[0] = b.x; r.edge0[1] = b.y; r.edge0[2] = b.z; r.e0e0 = b.w;
    r.edge1[0] = c.x; r.edge1[1] = c.y; r.edge1[2] = c.z; r.e1e1 = c.w;
    r.normal[0] = d.x; r.normal[1] = d.y; r.normal[2] = d.z; r.u0 = d.w;
    r.anchor[0] = e.x; r.anchor[1] = e.y; r.anchor[2] = e.z; r.u1 = e.w;
    return r;
}
PT_HD DevSdf const_sdf(const DevSdf* p, unsigned i) {
    const PT_CONST pt_vf4* q = const_vf4(p + i);
    const pt_vf4 a = q[0], b = q[1];
    DevSdf r;
    r.pos[0] = a.x; r.pos[1] = a.y; r.pos[2] = a.z; r.type = prt_f2u(a.w);
    r.params[0] = b.x; r.params[1] = b.y; r.params[2] = b.z; r.params[3] = b.w;
    return r;
}
PT_DEV f3 vexp(f3 a) { return F3(prt_exp(a.x), prt_exp(a.y), prt_exp(a.z)); }
PT_DEV float fmax3(f3 v) { return prt_fmax(prt_fmax(v.x, v.y), v.z); }
PT_DEV float avg3(f3 v) { return (v.x * 1.0f + v.y * 1.0f + v.z * 1.0f) * 0.3333333333333333333333333333333333333333333333f; }

// min/max for the slab test only.  The operands there are compared with <=, > afterwards, so the
// sign of a zero result is irrelevant, and a NaN operand (0 * inf) must be ignored exactly like
// OpenCL fmin/fmax do -- which is what v_min_f32 / v_max_f32 implement in IEEE mode.
PT_DEV float hw_min(float a, float b) { return __builtin_fminf(a, b); }
PT_DEV float hw_max(float a, float b) { return __builtin_fmaxf(a, b); }

struct Rng { unsigned s0, s1; };
PT_DEV float next1D(Rng& r) {                                   // kernels/prng/prng.cl:5-16
    r.s0 = 36969u * (r.s0 & 65535u) + (r.s0 >> 16);
    r.s1 = 18000u * (r.s1 & 65535u) + (r.s1 >> 16);
    unsigned ires = (r.s0 << 16) + r.s1;
    return (prt_u2f((ires & 0x007fffffu) | 0x40000000u) - 2.0f) * 0.5f;
}

struct Frame { f3 normal, tangent, bitangent; };
PT_DEV Frame make_frame(f3 n) {                                 // kernels/header.cl:179-192
    Frame f;
    float sn = prt_copysign(1.0f, n.z);
    float a = -hw_recip(sn + n.z);                       // -1.0f / (sn + n.z): rounding is symmetric
    float b = n.x * n.y * a;
    f.normal = n;
    f.tangent = F3(1.0f + sn * n.x * n.x * a, sn * b, -sn * n.x);
    f.bitangent = F3(b, sn + n.y * n.y * a, -n.y);
    return f;
}
PT_DEV f3 to_local(const Frame& f, f3 p) { return F3(dot(f.tangent, p), dot(f.bitangent, p), dot(f.normal, p)); }
PT_DEV f3 to_global(const Frame& f, f3 p) { return f.tangent * p.x + f.bitangent * p.y + f.normal * p.z; }

struct Ray {
    f3 origin, dir, normal, pos;
    float t;
    bool backside;
    float time;
};

struct Event {                                                   // SurfaceScatterEvent, header.cl:208-215
    f3 wi, wo, weight;
    float pdf;
    unsigned sampledLobe;
    Frame frame;
};

struct Mat {                                                     // Material, header.cl:219-234
    f3 color, eta, k;
    float roughness;
    unsigned t, lobes, dist;
};
// DM: the microfacet distributions the scene's materials use (PRT_DIST_* bits; 7 = any): masking the field with a compile-time constant
// lets the compiler drop the lobes' code for the others (render_kernel's PT_MATS_DISTS bits)
template <unsigned DM = 7u>
PT_DEV Mat load_mat(const DevMaterial* m) {
    Mat r;
    const PT_CONST pt_vf4* q = const_vf4(m);                 // (a uniform `m` -- the light's material -- is three scalar loads)
    const pt_vf4 a = q[0], e = q[1], kk = q[2];
    r.color = F3(a.x, a.y, a.z); r.roughness = a.w;
    r.eta = F3(e.x, e.y, e.z); r.k = F3(kk.x, kk.y, kk.z);
    unsigned b = prt_f2u(e.w);
    r.t = b & 0xffffu; r.lobes = (b >> 16) & 0xffu; r.dist = (b >> 24) & DM;
    return r;
}

PT_DEV unsigned mat_bits(const DevMaterial* m) { return const_u32(reinterpret_cast<const uint32_t*>(m), 7u); }   // DevMaterial::bits

// ---- sampling warps, kernels/utils.cl:92-152 ------------------------------------------------
PT_DEV f3 uniform_sphere(float xi_x, float xi_y) {
    float phi = xi_x * PT_TWO_PI;
    float z = xi_y * 2.0f - 1.0f;
    float r = hw_sqrt(prt_fmax(1.0f - z * z, 0.0f));
    return F3(prt_cos(phi) * r, prt_sin(phi) * r, z);
}
PT_DEV f3 cosine_hemisphere(float xi_x, float xi_y) {
    float phi = xi_x * PT_TWO_PI;
    float r = hw_sqrt(xi_y);
    return F3(prt_cos(phi) * r, prt_sin(phi) * r, hw_sqrt(prt_fmax(1.0f - xi_y, 0.0f)));
}
PT_DEV bool check_reflection(f3 wi, f3 wo) {                     // utils.cl:50-52
    return prt_fabs(wi.z * wo.z - wi.x * wo.x - wi.y * wo.y - 1.0f) < 1e-3f;
}
PT_DEV bool check_refraction(f3 wi, f3 wo, float eta, float cosThetaT) {   // utils.cl:54-58
    float dotP = -wi.x * wo.x * eta - wi.y * wo.y * eta - prt_copysign(cosThetaT, wi.z) * wo.z;
    return prt_fabs(dotP - 1.0f) < 1e-3f;
}

// ---- camera, kernels/camera.cl:17-66 ---------------------------------------------------------
// The part of createCamRay (camera.cl:19-28) that is the same for every pixel: basis vectors and the two
// tan() of the field of view.  Evaluated ONCE per prt_set_camera on the host -- same header, same IEEE
// operations, so the bits are the ones every lane used to recompute at each path start (at ~15 % lane
// occupancy, 10 lanes of a wave restart per frame).
PT_HD void camera_basis(const prt_camera& in, DevCamera& cam) {
    f3 view = normalize(ld3(in.view));
    f3 up = normalize(ld3(in.up));
    f3 hAxis = normalize(cross(view, up));
    f3 vAxis = normalize(cross(hAxis, view));
    f3 position = ld3(in.position);
    f3 middle = position + view;
    f3 horizontal = hAxis * prt_tan(in.fov[0] * 0.5f * (PT_PI / 180));
    f3 vertical = vAxis * prt_tan(in.fov[1] * -0.5f * (PT_PI / 180));
    cam.position[0] = position.x; cam.position[1] = position.y; cam.position[2] = position.z;
    cam.hAxis[0] = hAxis.x; cam.hAxis[1] = hAxis.y; cam.hAxis[2] = hAxis.z;
    cam.vAxis[0] = vAxis.x; cam.vAxis[1] = vAxis.y; cam.vAxis[2] = vAxis.z;
    cam.middle[0] = middle.x; cam.middle[1] = middle.y; cam.middle[2] = middle.z;
    cam.horizontal[0] = horizontal.x; cam.horizontal[1] = horizontal.y; cam.horizontal[2] = horizontal.z;
    cam.vertical[0] = vertical.x; cam.vertical[1] = vertical.y; cam.vertical[2] = vertical.z;
    cam.apertureRadius = in.apertureRadius; cam.focalDistance = in.focalDistance;
}

PT_DEV Ray create_cam_ray(int cx, int cy, int width, int height, const DevCamera& cam, Rng& rng) {
    const f3 hAxis = ld3(cam.hAxis), vAxis = ld3(cam.vAxis), position = ld3(cam.position);
    const f3 middle = ld3(cam.middle), horizontal = ld3(cam.horizontal), vertical = ld3(cam.vertical);
    int pixelx = cx;
    int pixely = height - cy - 1;
    float sx = (float)pixelx / (width - 1.0f);
    float sy = (float)pixely / (height - 1.0f);
    f3 onPlane = middle + (horizontal * ((2 * sx) - 1)) + (vertical * ((2 * sy) - 1));
    f3 onImagePlane = position + ((onPlane - position) * cam.focalDistance);
    f3 aperturePoint;
    if (cam.apertureRadius > 0.00001f) {
        float random1 = next1D(rng);
        float random2 = next1D(rng);
        float angle = 2 * PT_PI * random1;
        float distance = cam.apertureRadius * hw_sqrt(random2);
        float apertureX = prt_cos(angle) * distance;
        float apertureY = prt_sin(angle) * distance;
        aperturePoint = position + (hAxis * apertureX) + (vAxis * apertureY);
    } else {
        aperturePoint = position;
    }
    Ray ray;
    ray.backside = false;
    ray.origin = aperturePoint;
    ray.dir = normalize(onImagePlane - aperturePoint);
    ray.time = next1D(rng);
    ray.normal = splat(0.0f);
    ray.pos = splat(0.0f);
    ray.t = 0.0f;
    return ray;
}

// ---- BVH, kernels/geometry/bvh.cl + triangle.cl ------------------------------------------------
struct RayPre { float ix, iy, iz, sx, sy, sz; bool nx, ny, nz; };
PT_DEV RayPre ray_pre(const Ray& ray) {                          // bvh.cl:4-13 hoisted out of the node loop
    RayPre p;
    p.ix = hw_recip(ray.dir.x); p.iy = hw_recip(ray.dir.y); p.iz = hw_recip(ray.dir.z);
    p.sx = -ray.origin.x * p.ix; p.sy = -ray.origin.y * p.iy; p.sz = -ray.origin.z * p.iz;
    p.nx = ray.dir.x < 0.0f; p.ny = ray.dir.y < 0.0f; p.nz = ray.dir.z < 0.0f;
    return p;
}

struct TriHit { float u, v; unsigned slot; };      // w = 1.0f - u - v is formed again where it is needed (the same expression, the same bits)

// triangle.cl:4-43; `best_t` is ray->t.  The smooth normal (triangle.cl:30-34) is deferred to the
// end of the traversal: it only depends on (u, v, w, slot) of the last accepted hit.
PT_DEV bool hit_triangle_data(const float4 a, const float4 b, const float4 c4, const unsigned slot, const Ray& ray, float& best_t, TriHit& th) {
    const f3 p0 = F3(a.x, a.y, a.z), e1 = F3(a.w, b.x, b.y), e2 = F3(b.z, b.w, c4.x), n = F3(c4.y, c4.z, c4.w);
    f3 c = p0 - ray.origin;
    f3 r = cross(ray.dir, c);
    float inv_det = hw_recip(dot(n, ray.dir));
    float u = dot(r, e2) * inv_det;
    float v = dot(r, e1) * inv_det;
    float w = 1.0f - u - v;
    if (u >= 0 && v >= 0 && w >= 0) {
        float t = dot(n, c) * inv_det;
        if (t > PT_EPS && t < best_t) {
            best_t = t;
            th.u = u; th.v = v; th.slot = slot;
            return true;
        }
    }
    return false;
}
PT_DEV bool hit_triangle(const TriGeom* __restrict__ tg, unsigned slot, const Ray& ray, float& best_t, TriHit& th) {
    const PT_CONST pt_vf4* q = const_vf4(tg + slot);                // (only the root-is-a-leaf loop comes here: a uniform slot)
    const pt_vf4 a = q[0], b = q[1], c4 = q[2];
    return hit_triangle_data(make_float4(a.x, a.y, a.z, a.w), make_float4(b.x, b.y, b.z, b.w), make_float4(c4.x, c4.y, c4.z, c4.w), slot, ray, best_t, th);
}

// Traversal stack in LDS, [level][thread] (conflict-free), `levels` chosen per scene from the tree
// (prt_upload_scene: DevScene::stack_levels, never more than the reference's 64): a pop sits on the
// critical path of the walk (pop -> node index -> node fetch), LDS is the closest memory there is.
struct TravStack {
    unsigned* lds;         // this lane's column: level l is lds[l * stride]
    unsigned stride;       // threads per workgroup
};

struct PairData { float4 b0, b1, b2; uint4 meta; };
PT_DEV PairData load_pair(const NodePair* __restrict__ pairs, unsigned node) {
    const float4* q = reinterpret_cast<const float4*>(pairs + node);
    PairData d;
    d.b0 = q[0]; d.b1 = q[1]; d.b2 = q[2];
    d.meta = *reinterpret_cast<const uint4*>(q + 3);
    return d;
}
// Slab test of both children of one NodePair against the ray (bvh.cl:11-26), `best_t` = ray->t.
struct PairTest { float entry0, entry1; bool go0, go1; };
PT_DEV PairTest test_pair(const PairData& d, const RayPre& p, float best_t) {
    const float4 b0 = d.b0, b1 = d.b1, b2 = d.b2;
    PairTest r;
    // child 0: x = b0.xy, y = b0.zw, z = b1.xy ; child 1: x = b1.zw, y = b2.xy, z = b2.zw
    float e0x = prt_fma(p.nx ? b0.y : b0.x, p.ix, p.sx), x0x = prt_fma(p.nx ? b0.x : b0.y, p.ix, p.sx);
    float e0y = prt_fma(p.ny ? b0.w : b0.z, p.iy, p.sy), x0y = prt_fma(p.ny ? b0.z : b0.w, p.iy, p.sy);
    float e0z = prt_fma(p.nz ? b1.y : b1.x, p.iz, p.sz), x0z = prt_fma(p.nz ? b1.x : b1.y, p.iz, p.sz);
    float e1x = prt_fma(p.nx ? b1.w : b1.z, p.ix, p.sx), x1x = prt_fma(p.nx ? b1.z : b1.w, p.ix, p.sx);
    float e1y = prt_fma(p.ny ? b2.y : b2.x, p.iy, p.sy), x1y = prt_fma(p.ny ? b2.x : b2.y, p.iy, p.sy);
    float e1z = prt_fma(p.nz ? b2.w : b2.z, p.iz, p.sz), x1z = prt_fma(p.nz ? b2.z : b2.w, p.iz, p.sz);
    r.entry0 = hw_max(e0x, hw_max(e0y, hw_max(e0z, PT_EPS)));
    const float exit0 = hw_min(x0x, hw_min(x0y, hw_min(x0z, best_t)));
    r.entry1 = hw_max(e1x, hw_max(e1y, hw_max(e1z, PT_EPS)));
    const float exit1 = hw_min(x1x, hw_min(x1y, hw_min(x1z, best_t)));
    r.go0 = r.entry0 <= exit0;
    r.go1 = r.entry1 <= exit1;
    return r;
}

// the pair at a wave-uniform index (the root): four scalar loads, by construction (PT_CONST above)
PT_DEV PairData load_pair_uniform(const NodePair* __restrict__ pairs, unsigned node) {
    const PT_CONST pt_vf4* q = const_vf4(pairs + node);
    const pt_vf4 a = q[0], b = q[1], c = q[2], m = q[3];
    PairData d;
    d.b0 = make_float4(a.x, a.y, a.z, a.w); d.b1 = make_float4(b.x, b.y, b.z, b.w); d.b2 = make_float4(c.x, c.y, c.z, c.w);
    d.meta = make_uint4(prt_f2u(m.x), prt_f2u(m.y), prt_f2u(m.z), prt_f2u(m.w));
    return d;
}
struct TravRes { bool found; float t; TriHit th; };

#if defined(PT_WALK_STATS) && !defined(PT_EMU)    // development builds (with -DPT_PHASE_CLOCKS): wave-level counts of the walk (tools/phase_clocks.sh)
__device__ unsigned long long g_walk_stats[16];
#define PT_WSTAT(k) do { const unsigned long long m_ = __ballot(1); if ((threadIdx.x & 63u) == (unsigned)__builtin_ctzll(m_)) { atomicAdd(&g_walk_stats[2 * (k)], 1ull); atomicAdd(&g_walk_stats[2 * (k) + 1], (unsigned long long)__popcll(m_)); } } while (0)
#else
#define PT_WSTAT(k) do { } while (0)
#endif
// bvh.cl:132-206 (closest hit) / :43-114 (any hit), as a RESUMABLE walk with DEFERRED LEAVES: walk_begin = set-up + the step at
// the root, walk_box = the box tests and the descent of one step of bvh.cl:144-196 (closest hit) / :54-104 (any hit), walk_tri =
// ONE triangle of the leaves that step found.  Same visiting order as the reference: both children's boxes are tested against
// the CURRENT best t before either leaf is tested; leaf children are tested at once, left first -- a lane whose step hit a leaf
// takes no further box test before every triangle of that leaf (of both, when both children are leaves that were hit) has been
// tested; of two inner children the nearer (by entry distance, ties -> left) is followed and the other pushed.  What the split
// buys: the triangle test is the code a wave executes worst -- in the one-call step every walk step of a wave ran the triangle
// loop for the one or two lanes that had found a leaf (1.7 lanes of 64 per iteration through an 871 k-triangle mesh, two
// iterations per box step) -- and as a call of its own the kernel runs it when enough lanes have a triangle to test
// (render_kernel).  The pending triangles are ONE run of slots: pack_scene lays the triangles of a pair's two leaf children out
// next to each other.  Real branches (scalar mask work, free next to the vector pipe) rather than selects.  The lane machine runs
// a wave's walks for a bounded number of steps; a lane whose ray needs more keeps {node, sp, t, hit, pending run} for the wave's
// next walk phase, the LDS stack column stays the lane's own.  found: (closest) a triangle was accepted, (any) a triangle closer
// than tmax exists.  (u, v, slot: the last accepted triangle, as in TriHit; slot and three flags share a word -- pack_scene
// refuses 2^29 triangle slots and more; `last`: the stack was empty when the step ended: the walk is over once the run is tested)
struct WalkState {
    unsigned node;
    unsigned sp : 7, pend_count : 25;       // entries on the stack (at most 64); triangles of the step's leaves still to test
    float t, u, v;
    unsigned slot : 29, found : 1, done : 1, last : 1;
    unsigned pend_first;                    // first slot of the pending run
};

PT_DEV void walk_begin(const DevScene& sc, const bool ANY_HIT, const Ray& ray, const float tmax, const RayPre& p, WalkState& w,
                       const TravStack& stack) {
    w.found = false; w.done = false; w.last = false;
    w.slot = 0;
    w.pend_count = 0; w.pend_first = 0;
    if (!ANY_HIT) { w.t = tmax; w.u = w.v = 0.0f; }              // an any-hit walk leaves (t, u, v) alone: Lane::a lives there (see Lane)
    float t_any = tmax;
    TriHit th;
    th.u = th.v = 0.0f; th.slot = 0;
    w.node = 0; w.sp = 0;
    if (sc.root_is_leaf) {                                       // tiny meshes: no tree to walk
        for (unsigned i = sc.root_leaf_first; i < sc.root_leaf_first + sc.root_leaf_count; ++i)
            if (hit_triangle(sc.tri_geom, i, ray, ANY_HIT ? t_any : w.t, th)) { w.found = true; if (ANY_HIT) break; w.u = th.u; w.v = th.v; w.slot = th.slot; }
        w.done = true;
        return;
    }
    // The step at the root, which every lane of the wave takes, reads its node through the scalar cache (the address is
    // uniform): no vector-memory instruction, and the 6 rays in 10 that miss both of the root's children never issue one
    // in this walk.  A root with a leaf child that is hit takes the general step from node 0.
    const PairData d = load_pair_uniform(sc.pairs, 0u);
    const PairTest pt = test_pair(d, p, ANY_HIT ? tmax : w.t);
    const uint4 meta = d.meta;
    if (!((pt.go0 & (meta.y != 0xFFFFFFFFu)) | (pt.go1 & (meta.w != 0xFFFFFFFFu)))) {
        if (pt.go0 != pt.go1) {
            w.node = pt.go0 ? meta.x : meta.z;
        } else if (pt.go0) {
            unsigned nearc = meta.x, farc = meta.z;
            if (pt.entry0 > pt.entry1) { nearc = meta.z; farc = meta.x; }
            stack.lds[0] = farc;
            w.sp = 1;
            w.node = nearc;
        } else {
            w.done = true;                                       // missed both: the stack is empty, the walk is over
        }
    }
}

// the box half of a step (the lane has no triangle pending), on the pair `d` = pairs[w.node]
PT_DEV void walk_box_data(const PairData& d, const bool ANY_HIT, const Ray& ray, const RayPre& p, WalkState& w, const TravStack& stack) {
    PT_WSTAT(ANY_HIT ? 2 : 0);
    // an any-hit walk never shrinks its limit (it ends at the first hit): ray.t
    const PairTest pt = test_pair(d, p, ANY_HIT ? ray.t : w.t);
    const uint4 meta = d.meta;
    bool go0 = pt.go0, go1 = pt.go1;
    const bool hit_leaf0 = go0 & (meta.y != 0xFFFFFFFFu), hit_leaf1 = go1 & (meta.w != 0xFFFFFFFFu);
    unsigned pend = 0u;
    if (hit_leaf0 | hit_leaf1) {
        // the triangles of both leaf children are one run of slots, the left child's first (the order the reference tests them in)
        w.pend_first = hit_leaf0 ? meta.x : meta.z;
        pend = (hit_leaf0 ? meta.y : 0u) + (hit_leaf1 ? meta.w : 0u);
        if (hit_leaf0) go0 = false;
        if (hit_leaf1) go1 = false;
    }
    w.pend_count = pend;
    if (go0 != go1) {
        w.node = go0 ? meta.x : meta.z;
    } else if (go0) {
        unsigned nearc = meta.x, farc = meta.z;
        if (pt.entry0 > pt.entry1) { nearc = meta.z; farc = meta.x; }
        stack.lds[w.sp * stack.stride] = farc;
        ++w.sp;
        w.node = nearc;
    } else if (w.sp == 0u) {
        if (pend) w.last = true; else w.done = true;             // nothing left but (maybe) this step's triangles
    } else {
        --w.sp;                                                  // (the triangle tests touch neither the stack nor the node)
        w.node = stack.lds[w.sp * stack.stride];
    }
}

// one triangle of the pending run (triangle.cl:4-43 through hit_triangle_data), on the record (a, b, c4) = tri_geom[w.pend_first]
PT_DEV void walk_tri_data(const float4 a, const float4 b, const float4 c4, const bool ANY_HIT, const Ray& ray, WalkState& w) {
    PT_WSTAT(ANY_HIT ? 3 : 1);
    float t_any = ray.t;
    TriHit th;
    th.u = th.v = 0.0f; th.slot = 0;
    const unsigned i = w.pend_first;
    w.pend_first = i + 1u;
    const unsigned left = w.pend_count - 1u;
    w.pend_count = left;
    if (hit_triangle_data(a, b, c4, i, ray, ANY_HIT ? t_any : w.t, th)) {
        w.found = true;
        if (ANY_HIT) { w.pend_count = 0; w.done = true; return; }
        w.u = th.u; w.v = th.v; w.slot = th.slot;
    }
    if (left == 0u && w.last) w.done = true;
}

PT_DEV void walk_box(const DevScene& sc, const bool ANY_HIT, const Ray& ray, const RayPre& p, WalkState& w, const TravStack& stack) {
    const PairData d = load_pair(sc.pairs, w.node);
    walk_box_data(d, ANY_HIT, ray, p, w, stack);
}
PT_DEV void walk_tri(const DevScene& sc, const bool ANY_HIT, const Ray& ray, WalkState& w) {
    const float4* q = reinterpret_cast<const float4*>(sc.tri_geom + w.pend_first);
    const float4 a = q[0], b = q[1], c4 = q[2];
    walk_tri_data(a, b, c4, ANY_HIT, ray, w);
}
// ---- sphere / quad, kernels/geometry/sphere.cl:5-41, quad.cl:11-38 ------------------------------
PT_DEV bool hit_sphere(const DevSphere& s, const Ray& ray, float& best_t) {
    f3 p = ray.origin - ld3(s.pos);
    float B = dot(p, ray.dir);
    float C = dot(p, p) - s.radius * s.radius;
    float detSq = B * B - C;
    if (detSq >= 0.0f) {
        float det = hw_sqrt(detSq);
        float t = -B - det;
        if (t < best_t && t > PT_EPS) { best_t = t; return true; }
        t = -B + det;
        if (t < best_t && t > PT_EPS) { best_t = t; return true; }
    }
    return false;
}
// quad.cl:27-31: `l = x / c; if (l < 0.0f || l > 1.0f) miss` -- the quotient is only compared, so the 11-instruction
// IEEE divide is replaced by the comparisons it is equivalent to (c = dot(edge, edge) > 0, u = c * 2^-24 from the host):
//   RN(x/c) > 1  <=>  x/c > 1 + 2^-24 (the midpoint rounds to the even 1.0)  <=>  x - c > u, and x - c is exact for
//                     c < x < 2c (Sterbenz), <= 0 for x <= c, >= c for x >= 2c;
//   RN(x/c) < 0  <=>  x < 0, unless the quotient underflows to -0: with c <= 2^40 that needs |x| < 2^-100, which takes
//                     the divide, as does a divisor outside [2^-40, 2^40] (u is NaN then).  NaN / inf in x behave as in
//                     the divide.  Checked against the divide on all 2^32 values of x by prt_selftest_math fn 18.
PT_DEV bool out_of_unit_range(float x, float c, float u) {
    if (!(prt_fabs(x) >= 7.888609052210118e-31f) || u != u) {     // |x| < 2^-100 (or NaN), or no valid u: the reference expression
        const float l = x / c;
        return l < 0.0f || l > 1.0f;
    }
    return x < 0.0f || (x - c) > u;
}
// The reference's four exits (quad.cl:15, 19, 29) taken as two: a wave skips the code behind an exit only when ALL its lanes
// leave, so the values behind the first and the third exit are computed by the wave anyway; evaluating them ahead of the
// test (a quotient or a dot product that is thrown away has no side effect) halves the branches of the wave: +2.7 % on the
// whole kernel; all four merged into one: +0.4 % (the first pair does let whole waves skip the second).
PT_DEV bool hit_quad(const DevQuad& qd, const Ray& ray, float& best_t, f3& q_out) {
    const f3 normal = ld3(qd.normal), anchor = ld3(qd.anchor);
    const float nDotW = dot(normal, ray.dir);
    const float rt = dot(normal, anchor - ray.origin) / nDotW;
    if (nDotW <= 1e-5f || rt <= PT_EPS || rt >= best_t) return false;   // reference: (double)nDotW < 1e-5  <=>  nDotW <= 1e-5f in binary32
    f3 q = ray.origin + ray.dir * rt;            // origin + rt * dir
    f3 v = q - anchor;
    const float x0 = dot(v, ld3(qd.edge0)), x1 = dot(v, ld3(qd.edge1));
    const bool out0 = out_of_unit_range(x0, qd.e0e0, qd.u0), out1 = out_of_unit_range(x1, qd.e1e1, qd.u1);
    if (out0 || out1) return false;
    best_t = rt;
    q_out = q;
    return true;
}
// ---- raymarched SDF primitives, kernels/geometry/sdf.cl:5-118 ("next" row N4) ---------------------------
PT_DEV f3 vabs3(f3 a) { return F3(prt_fabs(a.x), prt_fabs(a.y), prt_fabs(a.z)); }
PT_DEV f3 vmax0(f3 a) { return F3(prt_fmax(a.x, 0.0f), prt_fmax(a.y, 0.0f), prt_fmax(a.z, 0.0f)); }
PT_DEV float s_map(const DevSdf& sdf, f3 pos) {
    const f3 c = pos - ld3(sdf.pos);
    const f3 b = ld3(sdf.params);
    if (sdf.type & (1u << 4)) return length(c) - sdf.params[0];                                 // SDF_SPHERE
    else if (sdf.type & (1u << 5)) {                                                              // SDF_BOX
        f3 d = vabs3(c) - b;
        return prt_fmin(prt_fmax(d.x, prt_fmax(d.y, d.z)), 0.0f) + length(vmax0(d));
    } else if (sdf.type & (1u << 6)) return length(vmax0(vabs3(c) - b)) - sdf.params[3];         // SDF_ROUND_BOX
    else if (sdf.type & (1u << 7)) return dot(c, b) + sdf.params[3];                             // SDF_PLANE
    return PT_INF;
}
PT_DEV float sdf_map(const DevScene& sc, float tmin, f3 pos, int& id) {
    float dist = tmin;
    for (unsigned i = 0; i < sc.n_sdfs; ++i) {
        const float temp_dist = s_map(const_sdf(sc.sdfs, i), pos);
        if (temp_dist < dist) { dist = temp_dist; id = (int)(sc.n_spheres + i); }
    }
    return dist;
}
PT_DEV f3 sdf_normal(const DevSdf& m, f3 pos) {
    const float e = PT_EPS * 2.0f;
    return normalize(F3(s_map(m, pos + F3(e, 0, 0)) - s_map(m, pos - F3(e, 0, 0)),
                        s_map(m, pos + F3(0, e, 0)) - s_map(m, pos - F3(0, e, 0)),
                        s_map(m, pos + F3(0, 0, e)) - s_map(m, pos - F3(0, 0, e))));
}
PT_DEV bool shadow_sdf(const DevScene& sc, const f3 o, const f3 d, const float tmax) {
    float t = PT_EPS * 100.0f;
    int id = -1;
    for (int i = 0; i < sc.shadow_marching_steps; ++i) {
        const float h = prt_fabs(sdf_map(sc, tmax, o + d * t, id));
        t += h;
        if (h < PT_EPS || t > tmax) break;
    }
    return t <= tmax;
}
PT_DEV bool intersect_sdf(const DevScene& sc, const Ray& ray, float& best_t, int& mesh_id) {
    float t = PT_EPS * 10.0f;
    int id = -1;
    for (int i = 0; i < sc.marching_steps; ++i) {
        const float h = prt_fabs(sdf_map(sc, best_t, ray.origin + ray.dir * t, id));
        if (h < PT_EPS || t > best_t) break;
        t += h;
    }
    if (t > best_t) return false;
    best_t = t;
    mesh_id = id;
    return true;
}

// ---- the rest of intersect_scene, kernels/intersect.cl:167-236, given the BVH result ----------------
// in: ray.origin/dir/normal + traversal result.  out: ray.t/normal/pos/backside, mesh_id (-1 = OBJ or nothing).
template <bool SDF>
PT_DEV bool finish_closest(const DevScene& sc, Ray& ray, const TravRes& tr, int& mesh_id) {
    float t = tr.t;
    mesh_id = -1;
    if (tr.found) {
        const float4* nq = reinterpret_cast<const float4*>(sc.tri_nrm + tr.th.slot);
        const float4 a = nq[0], b = nq[1], c = nq[2];
        ray.normal = F3(a.x, a.y, a.z) * (1.0f - tr.th.u - tr.th.v) + F3(b.x, b.y, b.z) * tr.th.u + F3(c.x, c.y, c.z) * tr.th.v;   // triangle.cl:22,30-34
    }
    ray.normal = normalize(ray.normal);
    ray.pos = ray.origin + ray.dir * t;
    if (sc.geom_flags & PRT_GEOM_SPHERE) {
        for (unsigned i = 0; i < sc.n_spheres; ++i) {
            const DevSphere s = const_sphere(sc.spheres, i);
            if (hit_sphere(s, ray, t)) {
                ray.pos = ray.origin + ray.dir * t;
                ray.normal = normalize(ray.pos - ld3(s.pos));
                mesh_id = (int)i;
            }
        }
    }
    if (SDF && sc.n_sdfs) {                                                                  // intersect.cl:185-194
        if (intersect_sdf(sc, ray, t, mesh_id)) {
            ray.pos = ray.origin + ray.dir * t;
            ray.normal = sdf_normal(const_sdf(sc.sdfs, (unsigned)(mesh_id - (int)sc.n_spheres)), ray.pos);
        }
    }
    if (sc.geom_flags & PRT_GEOM_QUAD) {
        for (unsigned i = 0; i < sc.n_quads; ++i) {
            f3 q;
            const DevQuad qd = const_quad(sc.quads, i);
            if (hit_quad(qd, ray, t, q)) {
                ray.backside = false;
                ray.normal = ld3(qd.normal);
                ray.pos = q;
                mesh_id = (int)(sc.quad_mesh_base + i);
            }
        }
    }
    bool nTrans = true;
    if (sc.ntrans_mask) nTrans = ((mat_bits(sc.mats + (mesh_id + 1)) & 0xffffu) & ~sc.ntrans_mask) != 0;
    ray.t = t;
    ray.backside = dot(ray.normal, ray.dir) > 0.0f;
    if (nTrans && ray.backside) ray.normal = -ray.normal;
    return t < PT_INF;
}

// ---- the rest of shadow(), kernels/intersect.cl:108-151, given that the BVH did not occlude: true = unoccluded
template <bool SDF>
PT_DEV bool finish_shadow(const DevScene& sc, const f3 origin, const f3 dir, const float maxDist) {
    Ray ray;
    ray.origin = origin; ray.dir = dir;
    float t = maxDist;
    if (sc.geom_flags & PRT_GEOM_SPHERE) {
        for (unsigned i = 0; i < sc.n_spheres; ++i)
            if (hit_sphere(const_sphere(sc.spheres, i), ray, t)) return false;     // an accepted hit always has t < maxDist
    }
    if (SDF && sc.n_sdfs && shadow_sdf(sc, origin, dir, maxDist)) return false;                     // intersect.cl:119-126
    if (sc.geom_flags & PRT_GEOM_QUAD) {
        for (unsigned i = 0; i < sc.n_quads; ++i) {
            f3 q;
            if (hit_quad(const_quad(sc.quads, i), ray, t, q)) return false;
        }
    }
    return true;
}

// ---- light sampling, kernels/geometry/{sphere.cl:59-88, quad.cl:40-62, geometry.cl:11-52} -----------
struct LightSample { f3 d; float dist, pdf; };

PT_DEV float sphere_direct_pdf(const DevSphere& s, f3 p) {
    float dist = length(ld3(s.pos) - p);
    float cosTheta = hw_sqrt(prt_fmax(dist * dist - s.radius * s.radius, 0.0f)) / dist;
    return PT_INV_TWO_PI / (1.0f - cosTheta);
}
PT_DEV bool sphere_sample_direct(const DevSphere& s, f3 p, LightSample& ls, Rng& rng) {
    f3 L = ld3(s.pos) - p;
    float d = length(L);
    float C = d * d - s.radius * s.radius;
    if (C <= 0.0f) return false;
    L = normalize(L);
    float cosTheta = hw_sqrt(C) / d;
    float xi_x = next1D(rng), xi_y = next1D(rng);
    (void)xi_x;                                  // the cap's azimuth is sampled and then discarded (sphere.cl:76-84)
    float z = xi_y * (1.0f - cosTheta) + cosTheta;
    float B = d * z;
    float det = hw_sqrt(prt_fmax(B * B - C, 0.0f));
    ls.dist = B - det;
    Frame frame = make_frame(L);
    ls.d = to_global(frame, splat(cosTheta));
    ls.pdf = PT_INV_TWO_PI / (1.0f - cosTheta);
    return true;
}
PT_DEV bool quad_sample_direct(const DevQuad& qd, f3 p, LightSample& ls, Rng& rng) {
    const f3 base = ld3(qd.base), normal = ld3(qd.normal);
    if (dot(normal, p - base) <= 0.0f) return false;
    float xi_x = next1D(rng), xi_y = next1D(rng);
    f3 q = base + ld3(qd.edge0) * xi_x + ld3(qd.edge1) * xi_y;
    ls.d = q - p;
    float rSq = dot(ls.d, ls.d);
    ls.dist = hw_sqrt(rSq);
    ls.d = ls.d / ls.dist;
    float cosTheta = -dot(normal, ls.d);
    ls.pdf = rSq / (cosTheta * qd.area);
    return true;
}
PT_DEV float quad_direct_pdf(const DevQuad& qd, f3 dir, f3 p) {
    const f3 normal = ld3(qd.normal);
    float cosTheta = prt_fabs(dot(normal, dir));
    float t = dot(normal, ld3(qd.base) - p) / dot(normal, dir);
    return t * t / (cosTheta * qd.area);
}
// directPdf of the mesh that a probe ray hit (mesh_id >= 0)
PT_DEV float direct_pdf_mesh(const DevScene& sc, int mesh_id, f3 dir, f3 p) {
    if ((sc.geom_flags & PRT_GEOM_SPHERE) && (unsigned)mesh_id < sc.n_spheres) return sphere_direct_pdf(const_sphere(sc.spheres, (unsigned)mesh_id), p);
    if ((sc.geom_flags & PRT_GEOM_QUAD) && (unsigned)mesh_id >= sc.quad_mesh_base) return quad_direct_pdf(const_quad(sc.quads, (unsigned)mesh_id - sc.quad_mesh_base), dir, p);
    return 0.0f;
}
PT_DEV bool sample_light0(const DevScene& sc, f3 p, LightSample& ls, Rng& rng) {
    if (sc.light_sphere != 0xFFFFFFFFu) return sphere_sample_direct(const_sphere(sc.spheres, sc.light_sphere), p, ls, rng);
    if (sc.light_quad != 0xFFFFFFFFu) return quad_sample_direct(const_quad(sc.quads, sc.light_quad), p, ls, rng);
    return false;
}
// the light of lightSample / volumeLightSample (base.cl:88-93,202-207) and its sample: LIGHT_INDICES[0], or -- PICK: the reference's
// PICK_RANDOM_LIGHT, prt_config::pick_random_light -- LIGHT_INDICES[(int)(next1D() * (LIGHT_COUNT + 1))], the draw BEFORE the sampler's own
// (the entry behind the array is in the table: DevScene::light_tab).  `mesh`: whose colour the contribution carries.
PT_DEV bool sample_light(const DevScene& sc, const bool PICK, f3 p, LightSample& ls, Rng& rng, unsigned& mesh) {
    if (!PICK) { mesh = sc.light_mesh; return sample_light0(sc, p, ls, rng); }
    const int k = (int)(next1D(rng) * (float)(sc.light_count + 1u));
    mesh = const_u32(sc.light_tab, (unsigned)k);
    if ((sc.geom_flags & PRT_GEOM_SPHERE) && mesh < sc.n_spheres) return sphere_sample_direct(const_sphere(sc.spheres, mesh), p, ls, rng);
    if ((sc.geom_flags & PRT_GEOM_QUAD) && mesh >= sc.quad_mesh_base) return quad_sample_direct(const_quad(sc.quads, mesh - sc.quad_mesh_base), p, ls, rng);
    return false;                                    // geometry.cl:11-32: only spheres and quads can be sampled
}

// ---- Fresnel, kernels/bxdf/Fresnel.cl:6-67 ----------------------------------------------------------
PT_DEV float conductor_reflectance(float eta, float k, float cosThetaI) {
    float cosThetaISq = cosThetaI * cosThetaI;
    float sinThetaISq = prt_fmax(1.0f - cosThetaISq, 0.0f);
    float sinThetaIQu = sinThetaISq * sinThetaISq;
    float innerTerm = eta * eta - k * k - sinThetaISq;
    float aSqPlusBSq = hw_sqrt(prt_fmax(innerTerm * innerTerm + 4.0f * eta * eta * k * k, 0.0f));
    float a = hw_sqrt(prt_fmax((aSqPlusBSq + innerTerm) * 0.5f, 0.0f));
    float Rs = ((aSqPlusBSq + cosThetaISq) - (2.0f * a * cosThetaI)) / ((aSqPlusBSq + cosThetaISq) + (2.0f * a * cosThetaI));
    float Rp = ((cosThetaISq * aSqPlusBSq + sinThetaIQu) - (2.0f * a * cosThetaI * sinThetaISq)) /
               ((cosThetaISq * aSqPlusBSq + sinThetaIQu) + (2.0f * a * cosThetaI * sinThetaISq));
    return 0.5f * (Rs + Rs * Rp);
}
PT_DEV f3 conductor_reflectance3(f3 eta, f3 k, float c) {
    return F3(conductor_reflectance(eta.x, k.x, c), conductor_reflectance(eta.y, k.y, c), conductor_reflectance(eta.z, k.z, c));
}
PT_DEV float dielectric_reflectance(float eta, float cosThetaI, float& cosThetaT) {
    if (cosThetaI < 0.0f) { eta = hw_recip(eta); cosThetaI = -cosThetaI; }
    float sinThetaTSq = eta * eta * (1.0f - cosThetaI * cosThetaI);
    if (sinThetaTSq > 1.0f) { cosThetaT = 0.0f; return 1.0f; }
    cosThetaT = hw_sqrt(prt_fmax(1.0f - sinThetaTSq, 0.0f));
    float Rs = (eta * cosThetaI - cosThetaT) / (eta * cosThetaI + cosThetaT);
    float Rp = (eta * cosThetaT - cosThetaI) / (eta * cosThetaT + cosThetaI);
    return (Rs * Rs + Rp * Rp) * 0.5f;
}

// ---- microfacet, kernels/bxdf/microfacet.cl:11-108 ----------------------------------------------------
PT_DEV float roughness_to_alpha(unsigned dist, float roughness) {
    roughness = prt_fmax(roughness, 1e-3f);
    if (dist & PRT_DIST_PHONG) return 2.0f / (roughness * roughness) - 2.0f;
    return roughness;
}
PT_DEV float mf_D(unsigned dist, float alpha, f3 m) {
    if (m.z <= 0.0f) return 0.0f;
    if (dist & PRT_DIST_BECKMANN) {
        float alphaSq = alpha * alpha, cosThetaSq = m.z * m.z;
        float tanThetaSq = prt_fmax(1.0f - cosThetaSq, 0.0f) / cosThetaSq;
        float cosThetaQu = cosThetaSq * cosThetaSq;
        return PT_INV_PI * prt_exp(-tanThetaSq / alphaSq) / (alphaSq * cosThetaQu);
    } else if (dist & PRT_DIST_PHONG) {
        return (alpha + 2.0f) * PT_INV_TWO_PI * prt_pow(m.z, alpha);
    } else if (dist & PRT_DIST_GGX) {
        float alphaSq = alpha * alpha, cosThetaSq = m.z * m.z;
        float tanThetaSq = prt_fmax(1.0f - cosThetaSq, 0.0f) / cosThetaSq;
        float cosThetaQu = cosThetaSq * cosThetaSq;
        float s = alphaSq + tanThetaSq;
        return alphaSq * PT_INV_PI / (cosThetaQu * (s * s));   // pow(x, 2.0f) == x*x in prt_detmath.h
    }
    return 0.0f;
}
PT_DEV float mf_G1(unsigned dist, float alpha, f3 v, f3 m) {
    if (dot(v, m) * v.z <= 0.0f) return 0.0f;
    if (dist & PRT_DIST_BECKMANN) {
        float cosThetaSq = v.z * v.z;
        float tanTheta = prt_fabs(hw_sqrt(prt_fmax(1.0f - cosThetaSq, 0.0f)) / v.z);
        float a = hw_recip(alpha * tanTheta);
        if (a < 1.6f) return (3.535f * a + 2.181f * a * a) / (1.0f + 2.276f * a + 2.577f * a * a);
        return 1.0f;
    } else if (dist & PRT_DIST_PHONG) {
        float cosThetaSq = v.z * v.z;
        float tanTheta = prt_fabs(hw_sqrt(prt_fmax(1.0f - cosThetaSq, 0.0f)) / v.z);
        float a = hw_sqrt(0.5f * alpha + 1.0f) / tanTheta;
        if (a < 1.6f) return (3.535f * a + 2.181f * a * a) / (1.0f + 2.276f * a + 2.577f * a * a);
        return 1.0f;
    } else if (dist & PRT_DIST_GGX) {
        float alphaSq = alpha * alpha, cosThetaSq = v.z * v.z;
        float tanThetaSq = prt_fmax(1.0f - cosThetaSq, 0.0f) / cosThetaSq;
        return 2.0f / (1.0f + hw_sqrt(1.0f + alphaSq * tanThetaSq));
    }
    return 0.0f;
}
PT_DEV float mf_G(unsigned dist, float alpha, f3 i, f3 o, f3 m) { return mf_G1(dist, alpha, i, m) * mf_G1(dist, alpha, o, m); }
PT_DEV float mf_pdf(unsigned dist, float alpha, f3 m) { return mf_D(dist, alpha, m) * m.z; }
PT_DEV f3 mf_sample(unsigned dist, float alpha, float xi_x, float xi_y) {
    float phi = xi_y * PT_TWO_PI;
    float cosTheta = 0.0f;
    if (dist & PRT_DIST_BECKMANN) {
        float tanThetaSq = -alpha * alpha * prt_log(1.0f - xi_x);
        cosTheta = hw_recip(hw_sqrt(1.0f + tanThetaSq));
    } else if (dist & PRT_DIST_PHONG) {
        cosTheta = prt_pow(xi_x, hw_recip(alpha + 2.0f));
    } else if (dist & PRT_DIST_GGX) {
        float tanThetaSq = alpha * alpha * xi_x / (1.0f - xi_x);
        cosTheta = hw_recip(hw_sqrt(1.0f + tanThetaSq));
    }
    float r = hw_sqrt(prt_fmax(1.0f - cosTheta * cosTheta, 0.0f));
    return F3(prt_cos(phi) * r, prt_sin(phi) * r, cosTheta);
}

// ---- materials ------------------------------------------------------------------------------------------
// Lambert.cl:4-31 (pdf: see oracle/pt_oracle.c LambertBSDF_pdf -- the reference build returns 0)
PT_DEV bool lambert_sample(Event& e, const Mat& mat, Rng& rng) {
    if (e.wi.z <= 0.0f) return false;
    float xi_x = next1D(rng), xi_y = next1D(rng);
    e.wo = cosine_hemisphere(xi_x, xi_y);
    e.pdf = prt_fabs(e.wo.z) * PT_INV_PI;
    e.weight = mat.color;
    e.sampledLobe = PRT_LOBE_DIFFUSE_R;
    return true;
}
PT_DEV f3 lambert_eval(const Event& e, const Mat& mat) {
    if (e.wi.z <= 0.0f || e.wo.z <= 0.0f) return splat(0.0f);
    return mat.color * PT_INV_PI * e.wo.z;
}
// Conductor.cl:4-29
PT_DEV bool conductor_sample(Event& e, const Mat& mat) {
    f3 F = conductor_reflectance3(mat.eta, mat.k, e.wi.z);
    e.wo = F3(-e.wi.x, -e.wi.y, e.wi.z);
    e.pdf = 1.0f;
    e.weight = mat.color * F;
    e.sampledLobe = PRT_LOBE_SPECULAR_R;
    return true;
}
PT_DEV f3 conductor_eval(const Event& e, const Mat& mat) {
    f3 F = conductor_reflectance3(mat.eta, mat.k, e.wi.z);
    if (check_reflection(e.wi, e.wo)) return mat.color * F;
    return splat(0.0f);
}
// RoughConductor.cl:4-62
PT_DEV bool rough_conductor_sample(Event& e, const Mat& mat, Rng& rng) {
    if (e.wi.z <= 0.0f) return false;
    float alpha = roughness_to_alpha(mat.dist, mat.roughness);
    float xi_x = next1D(rng), xi_y = next1D(rng);
    f3 m = mf_sample(mat.dist, alpha, xi_x, xi_y);
    float wiDotM = dot(e.wi, m);
    e.wo = m * (2.0f * wiDotM) - e.wi;
    if (wiDotM <= 0.0f || e.wo.z <= 0.0f) return false;
    float G = mf_G(mat.dist, alpha, e.wi, e.wo, m);
    float D = mf_D(mat.dist, alpha, m);
    float mPdf = mf_pdf(mat.dist, alpha, m);
    float pdf = mPdf * 0.25f / wiDotM;
    float weight = wiDotM * G * D / (e.wi.z * mPdf);
    f3 F = conductor_reflectance3(mat.eta, mat.k, wiDotM);
    e.pdf = pdf;
    e.weight = mat.color * F * weight;
    e.sampledLobe = PRT_LOBE_GLOSSY_R;
    return true;
}
PT_DEV f3 rough_conductor_eval(f3 wi, f3 wo, const Mat& mat) {
    if (wi.z <= 0.0f || wo.z <= 0.0f) return splat(0.0f);
    float alpha = roughness_to_alpha(mat.dist, mat.roughness);
    f3 hr = normalize(wi + wo);
    float cosThetaM = dot(wi, hr);
    f3 F = conductor_reflectance3(mat.eta, mat.k, cosThetaM);
    float G = mf_G(mat.dist, alpha, wi, wo, hr);
    float D = mf_D(mat.dist, alpha, hr);
    float fr = (G * D * 0.25f) / wi.z;
    return mat.color * (F * fr);
}
PT_DEV float rough_conductor_pdf(f3 wi, f3 wo, const Mat& mat) {
    if (wi.z <= 0.0f || wo.z <= 0.0f) return 0.0f;
    float sampleAlpha = roughness_to_alpha(mat.dist, mat.roughness);
    f3 hr = normalize(wi + wo);
    return mf_pdf(mat.dist, sampleAlpha, hr) * 0.25f / dot(wi, hr);
}
// Dielectric.cl:30-37 == RoughDielectric.cl:55-62
PT_DEV f3 absorb_weight(f3 weight, const Mat& mat, bool backside, float ray_t) {
    const bool ABS1 = (mat.t & PRT_MAT_ABS_REFR) != 0, ABS2 = (mat.t & PRT_MAT_ABS_REFR2) != 0;
    if (ABS1 | ABS2) {
        weight = weight * (ABS2 ? mat.color : splat(1.0f));
        if (backside) {
            f3 c = ABS1 ? mat.color : splat(1.0f);
            weight = weight * vexp(c * (-ray_t) * 10.0f);
        } else {
            weight = weight * splat(1.0f);
        }
    } else {
        weight = weight * mat.color;
    }
    return weight;
}
// Dielectric.cl:4-87
PT_DEV bool dielectric_sample(Event& e, const Mat& mat, bool backside, float ray_t, Rng& rng) {
    const float eta = e.wi.z < 0.0f ? mat.eta.x : hw_recip(mat.eta.x);
    float cosThetaT = 0.0f;
    float F = dielectric_reflectance(eta, prt_fabs(e.wi.z), cosThetaT);
    if (next1D(rng) < F) {
        e.wo = F3(-e.wi.x, -e.wi.y, e.wi.z);
        e.pdf = F;
        e.sampledLobe = PRT_LOBE_SPECULAR_R;
        e.weight = splat(F);
    } else {
        if (F == 1.0f) return false;
        e.wo = F3(-e.wi.x * eta, -e.wi.y * eta, -prt_copysign(cosThetaT, e.wi.z));
        e.pdf = 1.0f - F;
        e.sampledLobe = PRT_LOBE_SPECULAR_T;
        e.weight = splat(1.0f - F);
    }
    e.weight = absorb_weight(e.weight, mat, backside, ray_t);
    return true;
}
PT_DEV f3 dielectric_eval(const Event& e, const Mat& mat) {
    const float eta = e.wi.z < 0.0f ? mat.eta.x : hw_recip(mat.eta.x);
    float cosThetaT = 0.0f;
    float F = dielectric_reflectance(eta, prt_fabs(e.wi.z), cosThetaT);
    if (e.wi.z * e.wo.z >= 0.0f) {
        if (check_reflection(e.wi, e.wo)) return mat.color * F;
        return splat(0.0f);
    }
    if (check_refraction(e.wi, e.wo, eta, cosThetaT)) return mat.color * (1.0f - F);
    return splat(0.0f);
}
PT_DEV float dielectric_pdf(const Event& e, const Mat& mat) {
    const float eta = e.wi.z < 0.0f ? mat.eta.x : hw_recip(mat.eta.x);
    float cosThetaT = 0.0f;
    float F = dielectric_reflectance(eta, prt_fabs(e.wi.z), cosThetaT);
    if (e.wi.z * e.wo.z >= 0.0f) return check_reflection(e.wi, e.wo) ? F : 0.0f;
    return check_refraction(e.wi, e.wo, eta, cosThetaT) ? 1.0f - F : 0.0f;
}
PT_DEV float dielectric_eta(const Event& e, const Mat& mat) {   // Dielectric.cl:82-87 == RoughDielectric.cl:132-137
    if (e.wi.z * e.wo.z >= 0.0f) return 1.0f;
    return e.wi.z < 0.0f ? mat.eta.x : hw_recip(mat.eta.x);
}
// RoughDielectric.cl:4-137
PT_DEV float sgnE(float t) { return t < 0.0f ? -1.0f : 1.0f; }
PT_DEV bool rough_dielectric_sample(Event& e, const Mat& mat, bool backside, float ray_t, Rng& rng) {
    const float wiDotN = e.wi.z;
    const float eta = e.wi.z < 0.0f ? mat.eta.x : hw_recip(mat.eta.x);
    float sampleRoughness = (1.2f - 0.2f * hw_sqrt(prt_fabs(wiDotN))) * mat.roughness;
    float alpha = roughness_to_alpha(mat.dist, mat.roughness);
    float sampleAlpha = roughness_to_alpha(mat.dist, sampleRoughness);
    float xi_x = next1D(rng), xi_y = next1D(rng);
    f3 m = mf_sample(mat.dist, sampleAlpha, xi_x, xi_y);
    float pm = mf_pdf(mat.dist, sampleAlpha, m);
    if (pm < 1e-10f) return false;
    float wiDotM = dot(e.wi, m);
    float cosThetaT = 0.0f;
    float F = dielectric_reflectance(hw_recip(mat.eta.x), wiDotM, cosThetaT);
    float etaM = wiDotM < 0.0f ? mat.eta.x : hw_recip(mat.eta.x);
    bool reflect = next1D(rng) < F;
    if (reflect) e.wo = m * (2.0f * wiDotM) - e.wi;
    else e.wo = m * (etaM * wiDotM - sgnE(wiDotM) * cosThetaT) - e.wi * etaM;
    float woDotN = e.wo.z;
    bool reflected = wiDotN * woDotN > 0.0f;
    if (reflected != reflect) return false;
    float woDotM = dot(e.wo, m);
    float G = mf_G(mat.dist, alpha, e.wi, e.wo, m);
    float D = mf_D(mat.dist, alpha, m);
    e.weight = splat(prt_fabs(wiDotM) * G * D / (prt_fabs(wiDotN) * pm));
    if (reflect) {
        e.pdf = F * pm * 0.25f / prt_fabs(wiDotM);
        e.sampledLobe = PRT_LOBE_GLOSSY_R;
    } else {
        float s = eta * wiDotM + woDotM;
        e.pdf = (1.0f - F) * pm * prt_fabs(woDotM) / (s * s);
        e.sampledLobe = PRT_LOBE_GLOSSY_T;
    }
    e.weight = absorb_weight(e.weight, mat, backside, ray_t);
    return true;
}
PT_DEV f3 rough_diel_half(const Event& e, float eta, bool reflect) {
    if (reflect) return normalize(e.wi + e.wo) * sgnE(e.wi.z);
    return -normalize(e.wi * eta + e.wo);
}
PT_DEV f3 rough_dielectric_eval(const Event& e, const Mat& mat) {
    float wiDotN = e.wi.z, woDotN = e.wo.z;
    bool reflect = wiDotN * woDotN >= 0.0f;
    float alpha = roughness_to_alpha(mat.dist, mat.roughness);
    const float eta = wiDotN < 0.0f ? mat.eta.x : hw_recip(mat.eta.x);
    f3 m = rough_diel_half(e, eta, reflect);
    float wiDotM = dot(e.wi, m), woDotM = dot(e.wo, m);
    float cosThetaT = 0.0f;
    float F = dielectric_reflectance(hw_recip(mat.eta.x), wiDotM, cosThetaT);
    float G = mf_G(mat.dist, alpha, e.wi, e.wo, m);
    float D = mf_D(mat.dist, alpha, m);
    float fx;
    if (reflect) {
        fx = (F * G * D * 0.25f) / prt_fabs(wiDotN);
    } else {
        float s = eta * wiDotM + woDotM;
        fx = prt_fabs(wiDotM * woDotM) * (1.0f - F) * G * D / ((s * s) * prt_fabs(wiDotN));
    }
    return mat.color * fx;
}
PT_DEV float rough_dielectric_pdf(const Event& e, const Mat& mat) {
    float wiDotN = e.wi.z, woDotN = e.wo.z;
    bool reflect = wiDotN * woDotN >= 0.0f;
    float sampleRoughness = (1.2f - 0.2f * hw_sqrt(prt_fabs(wiDotN))) * mat.roughness;
    float sampleAlpha = roughness_to_alpha(mat.dist, sampleRoughness);
    float eta = wiDotN < 0.0f ? mat.eta.x : hw_recip(mat.eta.x);
    f3 m = rough_diel_half(e, eta, reflect);
    float wiDotM = dot(e.wi, m), woDotM = dot(e.wo, m);
    float cosThetaT = 0.0f;
    float F = dielectric_reflectance(hw_recip(mat.eta.x), wiDotM, cosThetaT);
    float pm = mf_pdf(mat.dist, sampleAlpha, m);
    if (reflect) return F * pm * 0.25f / prt_fabs(wiDotM);
    float s = eta * wiDotM + woDotM;
    return (1.0f - F) * pm * prt_fabs(woDotM) / (s * s);
}
// Coat.cl:4-112 (ior 1.3, thickness 1, sigmaA 0 -> avgTransmittance = exp(-0) = 1)
#define PT_COAT_IOR 1.3f
PT_DEV bool coat_sample(Event& e, const Mat& mat, Rng& rng) {
    if (e.wi.z <= 0.0f) return false;
    const float eta = 1.0f / PT_COAT_IOR;
    const float avgTransmittance = 1.0f;
    float cosThetaTi;
    float Fi = dielectric_reflectance(eta, e.wi.z, cosThetaTi);
    float specularProbability = Fi / (Fi + avgTransmittance * (1.0f - Fi));
    if (next1D(rng) < specularProbability) {
        e.wo = F3(-e.wi.x, -e.wi.y, e.wi.z);
        e.pdf = specularProbability;
        e.weight = splat(Fi / specularProbability);
        e.sampledLobe = PRT_LOBE_SPECULAR_R;
    } else {
        f3 originalWi = e.wi;
        e.wi = F3(originalWi.x * eta, originalWi.y * eta, cosThetaTi);
        if (!rough_conductor_sample(e, mat, rng)) return false;
        e.wi = originalWi;
        float cosThetaTo;
        float Fo = dielectric_reflectance(PT_COAT_IOR, e.wo.z, cosThetaTo);
        if (Fo == 1.0f) return false;
        float cosThetaSubstrate = e.wo.z;
        e.wo = F3(e.wo.x * PT_COAT_IOR, e.wo.y * PT_COAT_IOR, cosThetaTo);
        e.weight = e.weight * ((1.0f - Fi) * (1.0f - Fo));
        e.weight = e.weight / (1.0f - specularProbability);
        e.pdf *= 1.0f - specularProbability;
        e.pdf *= eta * eta * cosThetaTo / cosThetaSubstrate;
    }
    return true;
}
PT_DEV f3 coat_eval(const Event& e, const Mat& mat) {
    if (e.wi.z <= 0.0f || e.wo.z <= 0.0f) return splat(0.0f);
    const float eta = 1.0f / PT_COAT_IOR;
    float cosThetaTi;
    float Fi = dielectric_reflectance(eta, e.wi.z, cosThetaTi);
    if (check_reflection(e.wi, e.wo)) return splat(Fi);
    float cosThetaTo;
    float Fo = dielectric_reflectance(eta, e.wo.z, cosThetaTo);
    f3 nwi = F3(e.wi.x * eta, e.wi.y * eta, prt_copysign(cosThetaTi, e.wi.z));
    f3 nwo = F3(e.wo.x * eta, e.wo.y * eta, prt_copysign(cosThetaTo, e.wo.z));
    f3 substrateF = rough_conductor_eval(nwi, nwo, mat);
    float laplacian = eta * eta * e.wo.z / cosThetaTo;
    return substrateF * (laplacian * (1.0f - Fi) * (1.0f - Fo));
}
PT_DEV float coat_pdf(const Event& e, const Mat& mat) {
    if (e.wi.z <= 0.0f || e.wo.z <= 0.0f) return 0.0f;
    const float eta = 1.0f / PT_COAT_IOR;
    const float avgTransmittance = 1.0f;
    float cosThetaTi;
    float Fi = dielectric_reflectance(eta, e.wi.z, cosThetaTi);
    float specularProbability = Fi / (Fi + avgTransmittance * (1.0f - Fi));
    if (check_reflection(e.wi, e.wo)) return specularProbability;
    float cosThetaTo;
    dielectric_reflectance(eta, e.wo.z, cosThetaTo);
    f3 nwi = F3(e.wi.x * eta, e.wi.y * eta, prt_copysign(cosThetaTi, e.wi.z));
    f3 nwo = F3(e.wo.x * eta, e.wo.y * eta, prt_copysign(cosThetaTo, e.wo.z));
    return rough_conductor_pdf(nwi, nwo, mat) * (1.0f - specularProbability) * eta * eta * prt_fabs(e.wo.z / cosThetaTo);
}

// ---- dispatch, kernels/bxdf/bxdf.cl:57-273.  MATS = compile-time ACTIVE_MATS (0 = run-time); the
// PT_MATS_SDF bit marks the variants that carry the raymarched primitives (H_SDF scenes only) ---------
#define PT_MATS_SDF 0x80000000u
// PT_MATS_VIEW bit: the debug views VIEW_NORMAL / VIEW_BVH_HIT of kernels/main.cl:6-15,143-152 (prt_config::view_option)
#define PT_MATS_VIEW 0x40000000u
// PT_MATS_PICK bit: PICK_RANDOM_LIGHT of kernels/integrators/base.cl:9 (prt_config::pick_random_light)
#define PT_MATS_PICK 0x20000000u
// PT_MATS_ENVIS bit: prt_config::env_importance_sampling (not in the reference)
#define PT_MATS_ENVIS 0x10000000u
// PT_MATS_DISTS bits (3, at PT_MATS_DIST_SHIFT): the only microfacet distributions the scene's materials use, 0 = any -- the analogue for
// `mat->dist` (kernels/bxdf/microfacet.cl:6-9, a run-time field) of compiling the scene's ACTIVE_MATS: a GGX scene carries no Beckmann
// exponential and no Phong power (whose binary64 polynomials alone are 40 - 80 B of scratch per lane)
#define PT_MATS_DIST_SHIFT 24
#define PT_MATS_DISTS (7u << PT_MATS_DIST_SHIFT)
#define PT_MATS_FLAGS (PT_MATS_SDF | PT_MATS_VIEW | PT_MATS_PICK | PT_MATS_ENVIS | PT_MATS_DISTS)
template <unsigned MATS>
PT_HD constexpr unsigned dist_mask() { return ((MATS >> PT_MATS_DIST_SHIFT) & 7u) ? ((MATS >> PT_MATS_DIST_SHIFT) & 7u) : 7u; }
template <unsigned MATS>
PT_DEV unsigned active_mats(const DevScene& sc) { return (MATS & ~PT_MATS_FLAGS) ? (MATS & ~PT_MATS_FLAGS) : sc.active_mats; }

template <unsigned MATS>
PT_DEV bool bsdf_sample2(const DevScene& sc, Event& e, const Ray& ray, const Mat& mat, Rng& rng) {
    const unsigned am = active_mats<MATS>(sc);
    const unsigned t = mat.t & am;
    bool ok;
    if (t & PRT_MAT_DIFF) ok = lambert_sample(e, mat, rng);
    else if (t & PRT_MAT_COND) ok = conductor_sample(e, mat);
    else if (t & PRT_MAT_ROUGH_COND) ok = rough_conductor_sample(e, mat, rng);
    else if (t & PRT_MAT_DIEL) ok = dielectric_sample(e, mat, ray.backside, ray.t, rng);
    else if (t & PRT_MAT_ROUGH_DIEL) ok = rough_dielectric_sample(e, mat, ray.backside, ray.t, rng);
    else if (t & PRT_MAT_COAT) ok = coat_sample(e, mat, rng);
    else ok = false;
    if (!ok) return false;
    if (am & (PRT_MAT_DIEL | PRT_MAT_ROUGH_DIEL)) {              // bxdf.cl:121-140
        float eta = 1.0f;
        if (t & (PRT_MAT_DIEL | PRT_MAT_ROUGH_DIEL)) eta = dielectric_eta(e, mat);
        e.weight = e.weight * (eta * eta);
    }
    return true;
}
template <unsigned MATS>
PT_DEV f3 bsdf_eval2(const DevScene& sc, const Event& e, const Mat& mat) {
    const unsigned am = active_mats<MATS>(sc);
    const unsigned t = mat.t & am;
    f3 f;
    if (t & PRT_MAT_DIFF) f = lambert_eval(e, mat);
    else if (t & PRT_MAT_COND) f = conductor_eval(e, mat);
    else if (t & PRT_MAT_ROUGH_COND) f = rough_conductor_eval(e.wi, e.wo, mat);
    else if (t & PRT_MAT_DIEL) f = dielectric_eval(e, mat);
    else if (t & PRT_MAT_ROUGH_DIEL) f = rough_dielectric_eval(e, mat);
    else if (t & PRT_MAT_COAT) f = coat_eval(e, mat);
    else f = splat(0.0f);
    if (am & (PRT_MAT_DIEL | PRT_MAT_ROUGH_DIEL)) {              // bxdf.cl:204-223
        float eta = 1.0f;
        if (t & (PRT_MAT_DIEL | PRT_MAT_ROUGH_DIEL)) eta = dielectric_eta(e, mat);
        f = f * (eta * eta);
    }
    return f;
}
template <unsigned MATS>
PT_DEV float bsdf_pdf(const DevScene& sc, const Event& e, const Mat& mat) {
    const unsigned t = mat.t & active_mats<MATS>(sc);
    if (t & PRT_MAT_DIFF) return 0.0f;                           // LambertBSDF_pdf as compiled
    else if (t & PRT_MAT_COND) return check_reflection(e.wi, e.wo) ? 1.0f : 0.0f;
    else if (t & PRT_MAT_ROUGH_COND) return rough_conductor_pdf(e.wi, e.wo, mat);
    else if (t & PRT_MAT_DIEL) return dielectric_pdf(e, mat);
    else if (t & PRT_MAT_ROUGH_DIEL) return rough_dielectric_pdf(e, mat);
    else if (t & PRT_MAT_COAT) return coat_pdf(e, mat);
    return 0.0f;
}

// ---- phase functions + medium, kernels/phasefunctions/*.cl, kernels/media/homogeneous.cl:11-51 -------
PT_DEV float hg(float g, float cosTheta) {
    float term = 1.0f + g * g - 2.0f * g * cosTheta;
    return PT_INV_FOUR_PI * (1.0f - g * g) / (term * hw_sqrt(term));
}
PT_DEV float rayleigh(float cosTheta) { return (3.0f / (16.0f * PT_PI)) * (1.0f + cosTheta * cosTheta); }   // Rayleigh.cl:4-6
PT_DEV float phase_value(const DevScene& sc, f3 wi, f3 wo) {    // phase_eval (splat) == phase_pdf
    if (sc.phase_function == 2) return rayleigh(dot(wi, wo));
    if (sc.phase_function == 1) return hg(sc.phase_g, dot(wi, wo));
    return PT_INV_FOUR_PI;
}
struct PhaseSample { f3 w, weight; float pdf; };
PT_DEV void phase_sample(const DevScene& sc, f3 wi, PhaseSample& ps, Rng& rng) {
    float xi_x = next1D(rng), xi_y = next1D(rng);
    ps.weight = splat(1.0f);
    if (sc.phase_function == 2) {                                   // Rayleigh.cl:16-39
        float phi = xi_x * PT_TWO_PI;
        float z = xi_y * 4.0f - 2.0f;
        float invZ = hw_sqrt(z * z + 1.0f);
        float u = prt_cbrt(z + invZ);
        float cosTheta = u - hw_recip(u);
        float sinTheta = hw_sqrt(prt_fmax(1.0f - cosTheta * cosTheta, 0.0f));
        Frame tf = make_frame(wi);
        ps.w = to_global(tf, F3(prt_cos(phi) * sinTheta, prt_sin(phi) * sinTheta, cosTheta));
        ps.pdf = rayleigh(cosTheta);
    } else if (sc.phase_function == 1 && sc.phase_g != 0.0f) {
        const float g = sc.phase_g;
        float phi = xi_x * PT_TWO_PI;
        float q = (1.0f - g * g) / (1.0f + g * (xi_y * 2.0f - 1.0f));
        float cosTheta = (1.0f + g * g - q * q) / (2.0f * g);
        float sinTheta = hw_sqrt(prt_fmax(1.0f - cosTheta * cosTheta, 0.0f));
        Frame tf = make_frame(wi);
        ps.w = to_global(tf, F3(prt_cos(phi) * sinTheta, prt_sin(phi) * sinTheta, cosTheta));
        ps.pdf = hg(g, cosTheta);
    } else {
        ps.w = uniform_sphere(xi_x, xi_y);
        ps.pdf = PT_INV_FOUR_PI;
    }
}
struct MediumSample { f3 p, weight; bool exited; };
PT_DEV void medium_sample_distance(const DevScene& sc, MediumSample& ms, const Ray& ray, Rng& rng) {
    const f3 sigmaT = splat(sc.fog_sigma_t), sigmaS = splat(sc.fog_sigma_s);
    const float maxT = ray.t;
    float mt;
    if (sc.fog_abs_only) {
        mt = maxT;
        ms.weight = vexp(sigmaT * (-mt));
        ms.exited = true;
    } else {
        // lane 3 of the float3 is the zero padding lane in the reference build
        int lane = (int)prt_round(next1D(rng) * 3.0f);
        float sigmaTc = (lane == 3) ? 0.0f : sc.fog_sigma_t;
        float t = -prt_log(1.0f - next1D(rng)) / sigmaTc;
        mt = prt_fmin(t, maxT);
        ms.exited = (t >= maxT);
        f3 tau = sigmaT * mt;
        ms.weight = vexp(-tau);
        float pdf;
        if (ms.exited) {
            pdf = avg3(vexp(-tau));
        } else {
            pdf = avg3(sigmaT * vexp(-tau));
            ms.weight = ms.weight * sigmaS;
        }
        ms.weight = ms.weight / pdf;
    }
    ms.p = ray.origin + ray.dir * mt;
}

// ---- env map: kernels/utils.cl:46 + read_imagef(samplerA), OpenCL 1.2 s8.2, CLK_ADDRESS_CLAMP ----------
PT_DEV f3 env_texel(const DevScene& sc, int i, int j) {
    if (i < 0 || j < 0 || i >= sc.env_w || j >= sc.env_h) return splat(0.0f);
    const float* p = sc.env + ((size_t)j * sc.env_w + i) * 3;
    return F3(p[0], p[1], p[2]);
}
PT_DEV f3 env_lookup(const DevScene& sc, f3 dir) {
    float cx = (prt_atan2(dir.z, dir.x) * PT_INV_TWO_PI) + 0.5f;
    float cy = prt_acos(dir.y) * PT_INV_PI;
    float u = cx * (float)sc.env_w, v = cy * (float)sc.env_h;
    float fu = prt_floor(u - 0.5f), fv = prt_floor(v - 0.5f);
    float a = (u - 0.5f) - fu, b = (v - 0.5f) - fv;
    int i0 = (int)fu, j0 = (int)fv;
    f3 t00 = env_texel(sc, i0, j0), t10 = env_texel(sc, i0 + 1, j0), t01 = env_texel(sc, i0, j0 + 1), t11 = env_texel(sc, i0 + 1, j0 + 1);
    return t00 * ((1.f - a) * (1.f - b)) + t10 * (a * (1.f - b)) + t01 * ((1.f - a) * b) + t11 * (a * b);
}

// ---- environment-map importance sampling (prt_config::env_importance_sampling; not in the reference) -----------------------------
// The density lives on the unit square of envMapEquirect's (u, v) (utils.cl:46): piecewise constant per texel, proportional to
// what prt_upload_envmap built from the texel's neighbourhood maximum of the luminance x sin(theta) (+ a uniform floor, so that it
// is positive wherever the bilinear lookup can be).  pdf over directions = p(u, v) / (2 pi^2 sin(theta)).
struct EnvSample { f3 d; float pdf; };
PT_DEV unsigned cdf_find(const float* __restrict__ cdf, unsigned n, float xi) {        // largest k < n with cdf[k] <= xi (cdf[0] = 0, cdf[n] = 1)
    unsigned lo = 0u, hi = n;
    while (hi - lo > 1u) { const unsigned mid = (lo + hi) >> 1; if (cdf[mid] <= xi) lo = mid; else hi = mid; }
    return lo;
}
PT_DEV float env_texel_pdf(const DevScene& sc, unsigned i, unsigned j, float v) {     // density over directions at texel (i, j), polar coordinate v in [0, 1]
    const float* cols = sc.env_cdf_cols + (size_t)j * (unsigned)(sc.env_w + 1);
    const float prow = sc.env_cdf_rows[j + 1u] - sc.env_cdf_rows[j], pcol = cols[i + 1u] - cols[i];
    const float sin_theta = prt_sin(v * PT_PI);
    if (!(sin_theta > 0.0f)) return 0.0f;
    return prow * pcol * (float)sc.env_w * (float)sc.env_h / (2.0f * PT_PI * PT_PI * sin_theta);
}
PT_DEV EnvSample env_sample(const DevScene& sc, float xi0, float xi1) {
    const unsigned j = cdf_find(sc.env_cdf_rows, (unsigned)sc.env_h, xi0);
    const float r0 = sc.env_cdf_rows[j], r1 = sc.env_cdf_rows[j + 1u];
    const float* cols = sc.env_cdf_cols + (size_t)j * (unsigned)(sc.env_w + 1);
    const unsigned i = cdf_find(cols, (unsigned)sc.env_w, xi1);
    const float c0 = cols[i], c1 = cols[i + 1u];
    const float fv = r1 > r0 ? (xi0 - r0) / (r1 - r0) : 0.5f, fu = c1 > c0 ? (xi1 - c0) / (c1 - c0) : 0.5f;
    const float u = ((float)i + fu) / (float)sc.env_w, v = ((float)j + fv) / (float)sc.env_h;
    const float phi = (u - 0.5f) * PT_TWO_PI, theta = v * PT_PI;
    const float st = prt_sin(theta);
    EnvSample es;
    es.d = F3(st * prt_cos(phi), prt_cos(theta), st * prt_sin(phi));                   // the inverse of envMapEquirect
    es.pdf = env_texel_pdf(sc, i, j, v);
    return es;
}
PT_DEV float env_pdf(const DevScene& sc, f3 dir) {
    const float cx = (prt_atan2(dir.z, dir.x) * PT_INV_TWO_PI) + 0.5f, cy = prt_acos(dir.y) * PT_INV_PI;
    if (!(cx == cx) || !(cy == cy)) return 0.0f;
    int i = (int)(cx * (float)sc.env_w), j = (int)(cy * (float)sc.env_h);
    i = i < 0 ? 0 : (i >= sc.env_w ? sc.env_w - 1 : i);
    j = j < 0 ? 0 : (j >= sc.env_h ? sc.env_h - 1 : j);
    return env_texel_pdf(sc, (unsigned)i, (unsigned)j, cy);
}
// the density a BSDF sample of direction e.wo was drawn with (LambertBSDF_pdf as the reference compiles it is 0: SURVEY s9-Q7)
template <unsigned MATS>
PT_DEV float bsdf_pdf_sampling(const DevScene& sc, const Event& e, const Mat& mat) {
    if (mat.t & active_mats<MATS>(sc) & PRT_MAT_DIFF) return (e.wi.z <= 0.0f || e.wo.z <= 0.0f) ? 0.0f : prt_fabs(e.wo.z) * PT_INV_PI;
    return bsdf_pdf<MATS>(sc, e, mat);
}

PT_DEV float power_heuristic(float pdf0, float pdf1) { return (pdf0 * pdf0) / (pdf0 * pdf0 + pdf1 * pdf1); }

// ---- per-pixel state in registers: the lane machine ------------------------------------------------------------
// One lane owns one pixel and advances it segment by segment (one segment = one launch of the reference's
// render_kernel for that pixel, kernels/main.cl:66-163 -> integrators/pathtracing.cl:4-120 + base.cl:31-260), as a
// small state machine, so that a wave never waits for its deepest BVH walk.  One ITERATION of the wave runs, in order,
//   A  lane_front          READY lanes: seeds + path (re)start (main.cl:108-136) once per segment; then either ask for
//                          W1 = intersect_scene of the path ray (pathtracing.cl:27) when no cached hit exists (fresh
//                          camera ray, continuation of a specular bounce), or -- with the hit in hand -- the medium
//                          event / miss / emitter / BSDF or phase sampling of the segment; lanes that need no probe
//                          finish their segment right here, the others ask for W2 = intersect_scene of the BSDF-sampled
//                          probe ray (bsdfSample, base.cl:54-57) or of the phase-sampled one (base.cl:247)
//   B  closest-hit walk    every lane with a W1 / W2 in flight steps through the tree; the phase ends when fewer than
//                          FrameArgs::walk_min_lanes lanes are still walking and somebody has finished.  Unfinished
//                          lanes keep their WalkState and go on in the next iteration's B; a finished lane runs
//                          lane_closest_done (the rest of intersect_scene; result into Lane::h)
//   C  lane_back           lanes whose probe is answered: MIS term of the probe, light sampling (lightSample base.cl:
//                          79-134 / the probe part of volumePhaseSample), may ask for W3 = shadow ray (any hit)
//   D  any-hit walk        as B, for the shadow rays
//   E  lane_finish         radiance into acc, Russian roulette, bounce caps (pathtracing.cl:93-118, main.cl:142)
// A lane with shallow rays completes one segment per iteration; a lane with a deep ray sits out A/C/E until its walk
// is done and falls behind in frame number -- legal because pixels are independent: the seeds are a function of the
// lane's own frame number (main.cl:108-109).  Which lanes walk together changes nothing a lane computes: results are
// schedule-independent (tests/test_emu.py runs these same functions on the host with random phase lengths).
//
// Exact shortcuts relative to the reference's segment (each removes work, none changes a bit):
//   * hit cache: the reference intersects the ray a segment leaves behind twice, as the probe of segment f and as the
//     path ray of segment f+1 (same ray, same scene => same result).  Lane::h keeps the probe's hit: W1 only runs for
//     rays that were never probed.
//   * the shadow ray tests the primitives before the tree (shadow(), intersect.cl:94-152, is an OR of independent tests).
//   * W3 is asked for last also at a medium scatter (the reference walks it before the probe): it only selects the
//     radiance added to acc; no RNG draw or path state depends on it.
// The RNG draws happen exactly in the reference's order.
// (the hit POSITION is not kept: every branch of intersect_scene forms it as origin + dir * t with the final t -- finish_closest --
// and the ray that was walked stays in the lane's (origin, dir) until the next segment re-aims it: lane_hit_pos forms it again)
struct Hit { float t; f3 normal; int mesh_id : 24; unsigned didHit : 1, backside : 1; };   // (pack_scene refuses 2^23 meshes and more)

enum { K_NONE = 0, K_SURFACE_MIS = 1, K_SCATTER = 2 };
enum { ST_READY = 0, ST_WALKC = 1, ST_BACK = 2, ST_WALKS = 3, ST_FINISH = 4 };

struct Lane {
    // the pixel's RTD record (kernels/main.cl:117-119; what prt_read_state shows between launches)
    f3 mask;
    float acc[4];
    unsigned total, samples;
    unsigned diff : 16, spec : 16;          // (16 bits each in the RTD record)
    unsigned trans : 16, scatters : 16;
    // the segment's ray.  Inside a segment: Ray {origin, dir, t, time}; between segments the same four values ARE the
    // TempRay {origin, dir, time = ray.t, dist = ray.time} of rayToTemp (main.cl:28) -- lane_front swaps (t, time) where
    // tempToRay (main.cl:27) reads them back crosswise.
    f3 origin, dir;
    float t, time;
    Hit h;                   // closest hit of a ray: of (origin, dir) when h_valid; of the probe after W2
    Rng rng;
    // the scatter event at the segment's vertex (SurfaceScatterEvent, header.cl:208-215) that later phases need
    f3 weight;
    int mesh_id;
    // Two sets of seven words that are never alive together in one lane share their registers: {wi, n_shade, pdf} goes from
    // lane_front to lane_back, which has read it into an Event before it writes {vis, sh_d, sh_tmax} for phases D and E.
    union {
        struct { f3 wi, n_shade; float pdf; };
        struct {
            f3 vis;          // light-sample term if the shadow ray is unoccluded
            f3 sh_d;         // shadow ray (origin: h.pos, or the scatter position)
            float sh_tmax;
        };
    };
    // a segment that scatters in the medium has no surface event: its position and phase-sampled direction ARE (origin, dir) from
    // lane_front on (pathtracing.cl:58-59 assigns them at the end of the segment; nothing reads the old ray in between), the phase
    // weight is `weight`; only the pdf of the phase sample needs a word of its own
    float ps_pdf;
    // the walk in flight -- and, over its (t, u, v), the MIS term of the probe ("a" of base.cl:170 / "b" of base.cl:259): lane_back
    // writes `a` after lane_closest_done has read the closest-hit walk's result, lane_finish reads it, and the any-hit walk in
    // between keeps its limit and hit record in locals (walk_begin / walk_step)
    union {
        WalkState w;
        struct { unsigned w_node_, w_sp_; f3 a; };
    };
    f3 view_n;               // PT_MATS_VIEW variants only: ray.normal as render_kernel finds it after radiance() (main.cl:143-145)
    unsigned f;              // segments completed in this launch
    // Flags and small integers share ONE register (bit-fields of one word): as members of their own each of them costs a
    // VGPR for the whole life of the lane -- fourteen registers of the 96 a wave has at 5 waves per SIMD.
    unsigned stage : 3;      // ST_*
    unsigned kind : 2;       // K_*
    unsigned sampledLobe : 8;
    unsigned wasSpecular : 1, reset : 1;     // RTD
    unsigned h_valid : 1;
    unsigned terminate : 1, w2_ran : 1;
    unsigned sh : 1;
    unsigned begun : 1;      // seeds / restart of segment f done
    unsigned fresh : 1;      // the walk asked for has not started
    unsigned w2 : 1;         // the closest-hit walk in flight is the probe (W2), else the path ray (W1)
    unsigned occluded : 1;
    unsigned sh_vertex : 1;  // PT_MATS_ENVIS: the shadow ray starts at the vertex (= origin of the probe), not where the probe ended
    unsigned posted : 1;     // render_kernel_rp (pt_pool.h): the lane's ray is with the workgroup's walker waves
};

// position of the hit in L.h: intersect_scene's `ray.pos = ray.origin + ray.dir * t` (same operations, same bits) on the ray that was walked
PT_DEV f3 lane_hit_pos(const Lane& L) { return L.origin + L.dir * L.h.t; }

PT_DEV void lane_init(Lane& L) {
    L.kind = K_NONE; L.mesh_id = -1; L.terminate = L.w2_ran = L.sh = false;
    L.weight = L.vis = L.sh_d = splat(0.0f);
    L.sampledLobe = 0u; L.sh_tmax = 0.0f; L.ps_pdf = 1.0f;
    L.rng.s0 = L.rng.s1 = 0u;
    L.h.t = 0.0f; L.h.normal = splat(0.0f); L.h.mesh_id = -1; L.h.didHit = L.h.backside = false; L.h_valid = false;
    L.w.node = 0u; L.w.sp = 0u; L.w.pend_count = 0u; L.w.pend_first = 0u; L.w.t = 0.0f; L.w.u = L.w.v = 0.0f; L.w.slot = 0u; L.w.found = false; L.w.done = true; L.w.last = false;
    L.f = 0u; L.stage = ST_READY;
    L.begun = L.fresh = L.w2 = L.occluded = false;
    L.sh_vertex = false;
    L.posted = false;
}

// may this lane start (or go on with) a segment?  The "N spp" rule freezes a pixel at a segment boundary.
// `laggards`: some lane of the wave still owes frames of this launch.  Then, in a launch that allows it (FrameArgs::run_ahead:
// "N spp" mode, where a pixel's result does not depend on how many frames the others have done), a lane that has done its
// n_frames starts further segments instead of idling -- frame numbers and seeds are its own, so its path is the same path.
// `target`: the frames this lane owes the launch -- fa.n_frames, or more for a pixel whose paths are longer than the frame's average ("N spp"
// launches, FrameArgs::pace_inv_ref: render_kernel)
PT_DEV bool lane_owes_frames(const FrameArgs& fa, const Lane& L, const unsigned target) {
    if (L.stage != ST_READY || L.begun) return L.f < target;
    if (fa.spp_limit && L.reset && L.samples >= fa.spp_limit) return false;
    return L.f < target;
}
PT_DEV bool lane_owes_frames(const FrameArgs& fa, const Lane& L) { return lane_owes_frames(fa, L, fa.n_frames); }
PT_DEV bool lane_runnable(const FrameArgs& fa, const Lane& L, const bool laggards, const unsigned target) {
    if (L.stage != ST_READY) return false;
    if (L.begun) return true;
    if (fa.spp_limit && L.reset && L.samples >= fa.spp_limit) return false;
    return L.f < target || (fa.run_ahead && laggards && L.f < fa.seed_frames);
}
PT_DEV bool lane_runnable(const FrameArgs& fa, const Lane& L, const bool laggards) { return lane_runnable(fa, L, laggards, fa.n_frames); }

// E (also reached straight from A by lanes that need no walk): the end of radiance() and of render_kernel.
// `surface`: handleSurface sampled a direction (base.cl:183-191 still to do); `lit`: the shadow ray was unoccluded.
template <unsigned MATS, bool MEDIUM>
PT_DEV void lane_finish_segment(const DevScene& sc, Lane& L, f3 emission, const float alpha, const bool surface, bool done, const bool lit) {
    if (L.kind == K_SURFACE_MIS) {
        const f3 b = lit ? L.vis : splat(0.0f);
        emission = emission + (L.a + b) * L.mask;                        // base.cl:170-171
    } else if (MEDIUM && L.kind == K_SCATTER) {
        const f3 a = lit ? L.vis : splat(0.0f);
        emission = emission + (a + L.a) * L.mask;                        // pathtracing.cl:52-56
        L.mask = L.mask * L.weight;                                      // pathtracing.cl:58-61: origin, dir are in place (lane_front)
    }
    if (surface && !done) {                                              // handleSurface tail, base.cl:183-191
        L.wasSpecular = (L.sampledLobe & PRT_LOBE_SPECULAR) != 0;
        L.mask = L.mask * L.weight;
        L.diff = (L.diff + ((L.sampledLobe & (PRT_LOBE_DIFFUSE_R | PRT_LOBE_GLOSSY_R)) != 0)) & 0xffffu;
        L.spec = (L.spec + ((L.sampledLobe & PRT_LOBE_SPECULAR_R) != 0)) & 0xffffu;
        L.trans = (L.trans + ((L.sampledLobe & PRT_LOBE_TRANSMISSIVE) != 0)) & 0xffffu;
        if (L.terminate) {
            L.reset = true;
            done = true;
        } else {
            L.scatters = 0;                                              // pathtracing.cl:93-94
            ++L.total;
        }
    }
    if (!done) {
        const float roulettePdf = fmax3(L.mask);                         // pathtracing.cl:97-106
        if (L.total > 2 && roulettePdf < 0.1f) {
            if (next1D(L.rng) < roulettePdf) L.mask = L.mask / roulettePdf;
            else { L.reset = true; done = true; }
        }
    }
    if (!done) {
        if (L.total >= (unsigned)sc.max_bounces || (int)L.diff >= sc.max_diff_bounces ||
            (int)L.spec >= sc.max_spec_bounces || (int)L.trans >= sc.max_trans_bounces)
            L.reset = true;                                              // pathtracing.cl:109-115
    }
    if (MATS & PT_MATS_VIEW) { L.acc[0] = L.view_n.x; L.acc[1] = L.view_n.y; L.acc[2] = L.view_n.z; L.acc[3] = 1.0f; }   // main.cl:143-145,150-152
    else { L.acc[0] += emission.x; L.acc[1] += emission.y; L.acc[2] += emission.z; L.acc[3] += alpha; }   // main.cl:142
    // rayToTemp (main.cl:28): {origin, dir, ray.t, ray.time} are already where the next segment finds them
    ++L.f;
    L.begun = false;
    L.stage = ST_READY;
}

// A
template <unsigned MATS, bool MEDIUM>
PT_DEV void lane_front(const DevScene& sc, const DevCamera& cam, const FrameArgs& fa, Lane& L, int gx, int gy) {
    if (!L.begun) {                                                      // main.cl:108-136
        const unsigned frame = fa.first_frame + L.f;
        const int random0 = const_i32(fa.seed_pairs, 2u * L.f), random1 = const_i32(fa.seed_pairs, 2u * L.f + 1u);
        L.rng.s0 = (unsigned)gx * frame % 1000u + ((unsigned)random0 * 100u);      // main.cl:108-109
        L.rng.s1 = (unsigned)gy * frame % 1000u + ((unsigned)random1 * 100u);
        { const float tt = L.t; L.t = L.time; L.time = tt; }                        // tempToRay after rayToTemp, main.cl:27-28
        if (L.reset || L.samples == 0) {                                            // main.cl:122-136
            ++L.samples;
            L.total = 0; L.diff = 0; L.spec = 0; L.trans = 0; L.scatters = 0;
            L.wasSpecular = true;
            L.reset = false;
            L.mask = splat(1.0f);
            L.h_valid = false;
            const Ray cr = create_cam_ray(gx, gy, fa.width, fa.full_height, cam, L.rng);
            L.origin = cr.origin; L.dir = cr.dir; L.t = cr.t; L.time = cr.time;
        }
        L.begun = true;
    }
    if (!L.h_valid) {                                                    // W1: intersect_scene of the path ray, pathtracing.cl:27
        L.stage = ST_WALKC; L.fresh = true; L.w2 = false;
        return;
    }
    const unsigned am = active_mats<MATS>(sc);
    Ray ray;                                                             // the path ray with its hit
    ray.origin = L.origin; ray.dir = L.dir; ray.time = L.time;
    ray.t = L.h.t; ray.normal = L.h.normal; ray.pos = lane_hit_pos(L); ray.backside = L.h.backside;
    // what a segment without a probe of `ray` itself leaves in ray.normal.  The hit may come from the previous segment's probe,
    // which entered intersect_scene with that segment's normal in `ray`; the reference walks the ray again from tempToRay's zero
    // normal, and where nothing is hit that input is all intersect_scene has: normalize(0)
    if (MATS & PT_MATS_VIEW) L.view_n = L.h.didHit ? ray.normal : normalize(splat(0.0f));
    const bool didHit = L.h.didHit;
    L.mesh_id = L.h.mesh_id;
    L.t = L.h.t;
    L.h_valid = false;
    L.kind = K_NONE; L.terminate = false; L.w2_ran = false; L.sh = false;
    L.vis = splat(0.0f);
    const Mat mat = load_mat<dist_mask<MATS>()>((L.mesh_id + 1) ? &sc.mats[L.mesh_id + 1] : &sc.mats[sc.n_meshes + 1]);
    f3 emission = splat(0.0f);
    float alpha = 1.0f;
    if (MEDIUM) {
        MediumSample ms;
        medium_sample_distance(sc, ms, ray, L.rng);
        L.mask = L.mask * ms.weight;
        if (!ms.exited && (int)L.scatters < sc.max_scattering_events) {
            L.kind = K_SCATTER;
            L.scatters = (L.scatters + 1u) & 0xffffu;
            L.wasSpecular = false;
            // volumeLightSample, base.cl:194-230 (samples the light from ray.pos, not ms.p: SURVEY s9-Q4)
            LightSample rec;
            unsigned lmesh;
            if (sample_light(sc, (MATS & PT_MATS_PICK) != 0, ray.pos, rec, L.rng, lmesh)) {
                const float fv = phase_value(sc, ray.dir, rec.d);
                const f3 f = splat(fv);
                if (!(dot(f, f) == 0.0f)) {
                    L.sh = true; L.sh_d = rec.d; L.sh_tmax = rec.dist;
                    const Mat lm = load_mat(&sc.mats[lmesh + 1]);
                    const f3 tr = vexp(splat(sc.fog_sigma_t) * (-1.0f * rec.dist));
                    const f3 contribution = tr * lm.color * f * power_heuristic(rec.pdf, fv);
                    L.vis = contribution / rec.pdf;
                }
            }
            PhaseSample ps;
            phase_sample(sc, ray.dir, ps, L.rng);                        // volumePhaseSample, base.cl:232-260
            L.origin = ms.p; L.dir = ps.w; L.weight = ps.weight; L.ps_pdf = ps.pdf;
            L.w2_ran = true;
            L.stage = ST_WALKC; L.fresh = true; L.w2 = true;
            return;
        }
    }
    if (!didHit) {                                                       // pathtracing.cl:66-75
        L.reset = true;
        if (sc.alpha_testing) { emission = splat(0.0f); alpha = 0.0f; }
        else emission = L.mask * env_lookup(sc, ray.dir);
        lane_finish_segment<MATS, MEDIUM>(sc, L, emission, alpha, false, true, false);
        return;
    }
    if ((am & PRT_MAT_LIGHT) && (mat.t & PRT_MAT_LIGHT)) {               // pathtracing.cl:77-84
        if (L.wasSpecular) emission = emission + mat.color * L.mask;
        L.reset = true;
        lane_finish_segment<MATS, MEDIUM>(sc, L, emission, alpha, false, true, false);
        return;
    }
    Event e;                                                             // makeLocalScatterEvent, base.cl:11-14
    e.wo = splat(0.0f); e.weight = splat(1.0f); e.pdf = 1.0f; e.sampledLobe = 0;
    e.frame = make_frame(ray.normal);
    e.wi = to_local(e.frame, -ray.dir);
    const bool mis = (am & PRT_MAT_LIGHT) && (mat.lobes & ~(PRT_LOBE_SPECULAR | PRT_LOBE_FORWARD) & 0xffu);
    const bool ok = bsdf_sample2<MATS>(sc, e, ray, mat, L.rng);          // bsdfSample base.cl:31-77 / handleSurface base.cl:175-181
    L.wi = e.wi; L.weight = e.weight; L.pdf = e.pdf; L.sampledLobe = e.sampledLobe; L.n_shade = ray.normal;
    if (ok) {
        L.origin = ray.pos;
        L.dir = to_global(e.frame, e.wo);
    }
    if (mis) {                                                           // handleSurface, base.cl:168-172
        L.kind = K_SURFACE_MIS;
        if (ok) { L.w2_ran = true; L.stage = ST_WALKC; L.fresh = true; L.w2 = true; }
        else { L.terminate = true; L.stage = ST_BACK; }
    } else if (ok) {
        lane_finish_segment<MATS, MEDIUM>(sc, L, emission, alpha, true, false, false);
    } else {
        L.reset = true;
        lane_finish_segment<MATS, MEDIUM>(sc, L, emission, alpha, false, true, false);
    }
}

// the ray of the closest-hit walk in flight (`normal` = what intersect_scene finds in ray->normal on entry)
template <bool MEDIUM>
PT_DEV Ray lane_closest_ray(const Lane& L) {
    Ray r;
    r.pos = splat(0.0f); r.backside = false; r.t = PT_INF; r.time = 0.0f;
    if (MEDIUM && L.w2 && L.kind == K_SCATTER) {                          // volumePhaseSample's probe, base.cl:243-247
        r.origin = L.origin; r.dir = L.dir; r.normal = splat(0.0f);
    } else {
        r.origin = L.origin; r.dir = L.dir;
        r.normal = L.w2 ? L.h.normal : splat(0.0f);                       // bsdfSample re-aims `ray` itself (base.cl:54-57); tempToRay zeroes it
    }
    return r;
}

// B, a lane whose closest-hit walk is over: the rest of intersect_scene (intersect.cl:167-236)
template <unsigned MATS, bool MEDIUM>
PT_DEV void lane_closest_done(const DevScene& sc, Lane& L) {
    TravRes r;
    r.found = L.w.found; r.t = L.w.t; r.th.u = L.w.u; r.th.v = L.w.v; r.th.slot = L.w.slot;
    Ray wr = lane_closest_ray<MEDIUM>(L);
    int mid;
    const bool hit = finish_closest<(MATS & PT_MATS_SDF) != 0>(sc, wr, r, mid);
    L.h.t = wr.t; L.h.normal = wr.normal; L.h.mesh_id = mid; L.h.didHit = hit; L.h.backside = wr.backside;
    L.h_valid = true;
    if ((MATS & PT_MATS_VIEW) && L.w2 && !(MEDIUM && L.kind == K_SCATTER)) L.view_n = wr.normal;   // bsdfSample's intersect_scene works on `ray` itself
    if (!L.w2) {
        L.stage = ST_READY;                                              // lane_front goes on with the hit in the next iteration
    } else {
        if (!(MEDIUM && L.kind == K_SCATTER)) L.t = wr.t;                // bsdfSample's intersect_scene works on `ray` itself
        L.stage = ST_BACK;
    }
}

// C
template <unsigned MATS, bool MEDIUM>
PT_DEV void lane_back(const DevScene& sc, Lane& L) {
    const f3 hit_pos = lane_hit_pos(L);
    L.a = splat(0.0f);                                                   // (from here on the words of the finished walk's (t, u, v) are `a`)
    f3 sh_o = hit_pos;
    if (MATS & PT_MATS_ENVIS) L.sh_vertex = false;
    if (L.kind == K_SURFACE_MIS) {
        constexpr bool ENVIS = (MATS & PT_MATS_ENVIS) != 0;
        const Mat mat = load_mat<dist_mask<MATS>()>((L.mesh_id + 1) ? &sc.mats[L.mesh_id + 1] : &sc.mats[sc.n_meshes + 1]);
        if (L.w2_ran && L.h.didHit) {                                    // the probe ray, base.cl:58-75
            const int mid = L.h.mesh_id;
            const unsigned lbits = mat_bits(sc.mats + (mid + 1));
            if (lbits & PRT_MAT_LIGHT) {
                const Mat lm = load_mat(&sc.mats[mid + 1]);
                L.a = lm.color * L.weight * power_heuristic(L.pdf, direct_pdf_mesh(sc, mid, L.dir, hit_pos));
                if (MEDIUM) L.a = L.a * vexp(splat(sc.fog_sigma_t) * (-1.0f * L.t));
            }
        } else if (ENVIS && L.w2_ran) {
            // the BSDF-sampled ray escapes: the map along it, weighted against the map's own sampling strategy (a mirror direction
            // cannot be produced by it), and the path ends here instead of in a segment of its own
            const float w = (L.sampledLobe & PRT_LOBE_SPECULAR) ? 1.0f : power_heuristic(L.pdf, 0.5f * env_pdf(sc, L.dir));
            L.a = env_lookup(sc, L.dir) * L.weight * w;
            L.terminate = true;
        }
        // lightSample, base.cl:79-134 (from ray.pos = the probe ray's hit point: SURVEY s9-Q4)
        if (ENVIS && next1D(L.rng) < 0.5f) {                             // the light-sample strategy of this vertex is the environment map
            const float xi0 = next1D(L.rng), xi1 = next1D(L.rng);
            const EnvSample es = env_sample(sc, xi0, xi1);
            if (es.pdf > 0.0f) {
                Event e;
                e.frame = make_frame(L.n_shade);
                e.wi = L.wi; e.weight = L.weight; e.pdf = L.pdf; e.sampledLobe = L.sampledLobe;
                e.wo = to_local(e.frame, es.d);
                const f3 fr = bsdf_eval2<MATS>(sc, e, mat);
                if (!(dot(fr, fr) == 0.0f)) {
                    // (the reference samples its light from where the PROBE ended; the map is sampled from the vertex, which
                    // is what its BSDF-sampled counterpart sees: the probe's origin, or -- no probe -- the hit point itself)
                    if (L.w2_ran) { sh_o = L.origin; L.sh_vertex = true; }
                    L.sh = true; L.sh_d = es.d; L.sh_tmax = PT_INF;
                    const float pe = 0.5f * es.pdf;
                    L.vis = env_lookup(sc, es.d) * fr * (power_heuristic(pe, bsdf_pdf_sampling<MATS>(sc, e, mat)) / pe);
                }
            }
        } else {
            LightSample rec;
            unsigned lmesh;
            if (sample_light(sc, (MATS & PT_MATS_PICK) != 0, hit_pos, rec, L.rng, lmesh)) {
                Event e;
                e.frame = make_frame(L.n_shade);
                e.wi = L.wi; e.weight = L.weight; e.pdf = L.pdf; e.sampledLobe = L.sampledLobe;
                e.wo = to_local(e.frame, rec.d);
                const f3 fr = bsdf_eval2<MATS>(sc, e, mat);
                if (!(dot(fr, fr) == 0.0f)) {
                    L.sh = true; L.sh_d = rec.d; L.sh_tmax = rec.dist;
                    const Mat lm = load_mat(&sc.mats[lmesh + 1]);
                    f3 contribution = lm.color * fr;
                    if (MEDIUM) contribution = contribution * vexp(splat(sc.fog_sigma_t) * (-1.0f * rec.dist));
                    contribution = contribution * power_heuristic(rec.pdf, bsdf_pdf<MATS>(sc, e, mat));
                    L.vis = contribution / rec.pdf;
                    if (ENVIS) L.vis = L.vis * 2.0f;                     // (the light was the coin's other side: probability 1 / 2)
                }
            }
        }
    } else if (MEDIUM && L.kind == K_SCATTER) {
        sh_o = L.origin;
        if (L.h.didHit) {
            const int mid = L.h.mesh_id;
            const unsigned lbits = mat_bits(sc.mats + (mid + 1));
            if (lbits & PRT_MAT_LIGHT) {
                const Mat lm = load_mat(&sc.mats[mid + 1]);
                const f3 tr = vexp(splat(sc.fog_sigma_t) * (-1.0f * L.h.t));
                L.a = tr * lm.color * L.weight * power_heuristic(L.ps_pdf, direct_pdf_mesh(sc, mid, L.dir, L.origin));   // "b" of base.cl:259
            }
        }
    }
    // shadow(), intersect.cl:94-152: BVH, sphere and quad tests are independent and the any-hit walk never shrinks
    // ray.t, so the boolean does not depend on their order: the primitives first, the tree only if they do not occlude
    if (L.sh && !finish_shadow<(MATS & PT_MATS_SDF) != 0>(sc, sh_o, L.sh_d, L.sh_tmax)) L.sh = false;
    if (L.sh) { L.stage = ST_WALKS; L.fresh = true; }
    else { L.stage = ST_FINISH; L.occluded = false; }
}

template <bool MEDIUM, bool ENVIS = false>
PT_DEV Ray lane_shadow_ray(const Lane& L) {
    Ray r;
    r.origin = ((MEDIUM && L.kind == K_SCATTER) || (ENVIS && L.sh_vertex)) ? L.origin : lane_hit_pos(L);
    r.dir = L.sh_d; r.normal = splat(0.0f); r.pos = splat(0.0f); r.t = L.sh_tmax; r.backside = false; r.time = 0.0f;
    return r;
}

// E
template <unsigned MATS, bool MEDIUM>
PT_DEV void lane_finish(const DevScene& sc, Lane& L) {
    lane_finish_segment<MATS, MEDIUM>(sc, L, splat(0.0f), 1.0f, L.kind == K_SURFACE_MIS, false, L.sh && !L.occluded);
}

}  // namespace dev
}  // namespace prt
