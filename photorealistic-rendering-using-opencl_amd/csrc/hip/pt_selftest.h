// pt_selftest.h -- per-function known-answer entry of the device code (prt_selftest_fn, include/prt.h).
// Same case layout as oracle/ref/kat_harness.cl, which drives the REFERENCE's own functions to make the fixtures
// (tests/golden/kat_*.npz): `params` up to 80 floats shared by a call, 32 floats in / 32 floats out per case, uint
// values as float bit patterns, the RNG state in in[30..31] / out[30..31].  Compiled into libprt (selftest_fn_kernel)
// and, for the CPU tests, into tests/emu.
#pragma once
#include "pt_device.h"

namespace prt {
namespace dev {

PT_DEV Mat selftest_material(const float* p) {
    Mat m;
    m.color = F3(p[0], p[1], p[2]); m.eta = F3(p[3], p[4], p[5]); m.k = F3(p[6], p[7], p[8]);
    m.roughness = p[9];
    m.t = prt_f2u(p[10]) & 0xffffu; m.lobes = prt_f2u(p[11]) & 0xffu; m.dist = prt_f2u(p[12]) & 0xffu;
    return m;
}

PT_DEV void selftest_fn(const int fn, const float* params, const float* x, float* y) {
    for (int k = 0; k < 32; ++k) y[k] = 0.0f;
    Rng rng;
    rng.s0 = prt_f2u(x[30]); rng.s1 = prt_f2u(x[31]);
    DevScene sc = DevScene();
    sc.active_mats = PRT_MAT_LIGHT | PRT_MAT_DIFF | PRT_MAT_COND | PRT_MAT_DIEL | PRT_MAT_COAT | PRT_MAT_ROUGH_COND | PRT_MAT_ROUGH_DIEL;
    switch (fn) {
    case 1: {                                   // BSDF2, kernels/bxdf/bxdf.cl:105-143
        const Mat mat = selftest_material(params);
        Event e;
        e.frame = make_frame(F3(x[3], x[4], x[5]));
        e.wi = F3(x[0], x[1], x[2]);
        e.wo = splat(0.0f); e.weight = splat(1.0f); e.pdf = 1.0f; e.sampledLobe = 0;
        Ray ray;
        ray.origin = splat(0.0f); ray.dir = F3(0.0f, 0.0f, 1.0f); ray.normal = F3(x[3], x[4], x[5]); ray.pos = splat(0.0f);
        ray.t = x[6]; ray.backside = x[7] != 0.0f; ray.time = 0.0f;
        const bool ok = bsdf_sample2<0u>(sc, e, ray, mat, rng);
        y[0] = ok ? 1.0f : 0.0f;
        y[1] = e.wo.x; y[2] = e.wo.y; y[3] = e.wo.z;
        y[4] = e.weight.x; y[5] = e.weight.y; y[6] = e.weight.z;
        y[7] = e.pdf; y[8] = prt_u2f(e.sampledLobe & 0xffu);
    } break;
    case 2: {                                   // BSDF_eval2 / BSDF_pdf, bxdf.cl:192-273
        const Mat mat = selftest_material(params);
        Event e;
        e.frame = make_frame(F3(0.0f, 0.0f, 1.0f));
        e.wi = F3(x[0], x[1], x[2]); e.wo = F3(x[3], x[4], x[5]);
        e.weight = splat(1.0f); e.pdf = 1.0f; e.sampledLobe = 0;
        const f3 f = bsdf_eval2<0u>(sc, e, mat);
        y[0] = f.x; y[1] = f.y; y[2] = f.z;
        y[3] = (x[6] != 0.0f) ? bsdf_pdf<0u>(sc, e, mat) : 0.0f;
    } break;
    case 3: {                                   // microfacet.cl:11-108
        const unsigned dist = prt_f2u(params[0]);
        const float alpha = roughness_to_alpha(dist, params[1]);
        const f3 v = F3(x[0], x[1], x[2]), m = F3(x[3], x[4], x[5]);
        y[0] = alpha;
        y[1] = mf_D(dist, alpha, m);
        y[2] = mf_G1(dist, alpha, v, m);
        y[3] = mf_pdf(dist, alpha, m);
        const f3 s = mf_sample(dist, alpha, x[6], x[7]);
        y[4] = s.x; y[5] = s.y; y[6] = s.z;
    } break;
    case 4: {                                   // Fresnel.cl:6-57
        float ct = 0.0f;
        y[0] = conductor_reflectance(x[0], x[1], x[2]);
        y[1] = dielectric_reflectance(x[0], x[2], ct);
        y[2] = ct;
    } break;
    case 5: {                                   // geometry/sphere.cl:59-88
        DevSphere s;
        s.pos[0] = params[0]; s.pos[1] = params[1]; s.pos[2] = params[2]; s.radius = params[3];
        const f3 p = F3(x[0], x[1], x[2]);
        LightSample ls;
        ls.d = splat(0.0f); ls.dist = 0.0f; ls.pdf = 0.0f;
        const bool ok = sphere_sample_direct(s, p, ls, rng);
        y[0] = ok ? 1.0f : 0.0f; y[1] = ls.d.x; y[2] = ls.d.y; y[3] = ls.d.z; y[4] = ls.dist; y[5] = ls.pdf;
        y[6] = sphere_direct_pdf(s, p);
    } break;
    case 6: {                                   // geometry/quad.cl:40-62
        DevQuad q;
        pack_quad(params + 3, q);
        const f3 p = F3(x[0], x[1], x[2]), dir = F3(x[3], x[4], x[5]);
        LightSample ls;
        ls.d = splat(0.0f); ls.dist = 0.0f; ls.pdf = 0.0f;
        const bool ok = quad_sample_direct(q, p, ls, rng);
        y[0] = ok ? 1.0f : 0.0f; y[1] = ls.d.x; y[2] = ls.d.y; y[3] = ls.d.z; y[4] = ls.dist; y[5] = ls.pdf;
        y[6] = quad_direct_pdf(q, dir, p);
    } break;
    case 7: {                                   // media/homogeneous.cl:11-51 (out[7], out[8] -- t and pdf -- are not kept by the product)
        sc.fog_sigma_s = params[1]; sc.fog_sigma_t = params[2]; sc.fog_abs_only = params[3] != 0.0f;
        Ray ray;
        ray.origin = F3(x[0], x[1], x[2]); ray.dir = F3(x[3], x[4], x[5]); ray.normal = splat(0.0f); ray.pos = splat(0.0f);
        ray.t = x[6]; ray.backside = false; ray.time = 0.0f;
        MediumSample ms;
        medium_sample_distance(sc, ms, ray, rng);
        y[0] = ms.p.x; y[1] = ms.p.y; y[2] = ms.p.z; y[3] = ms.weight.x; y[4] = ms.weight.y; y[5] = ms.weight.z;
        y[6] = ms.exited ? 1.0f : 0.0f;
    } break;
    case 8: {                                   // phasefunctions/HenyeyGreenstein.cl:4-48 (g = 0.6)
        sc.phase_function = 1; sc.phase_g = 0.6f;
        const f3 wi = F3(x[0], x[1], x[2]), wo = F3(x[3], x[4], x[5]);
        PhaseSample ps;
        phase_sample(sc, wi, ps, rng);
        y[0] = 1.0f; y[1] = ps.w.x; y[2] = ps.w.y; y[3] = ps.w.z; y[4] = ps.weight.x; y[5] = ps.weight.y; y[6] = ps.weight.z;
        y[7] = ps.pdf; y[8] = phase_value(sc, wi, wo); y[9] = y[8];
    } break;
    case 9: {                                   // camera.cl:17-66
        prt_camera pc;
        for (int k = 0; k < 4; ++k) { pc.position[k] = params[k]; pc.view[k] = params[4 + k]; pc.up[k] = params[8 + k]; }
        pc.resolution[0] = params[12]; pc.resolution[1] = params[13]; pc.fov[0] = params[14]; pc.fov[1] = params[15];
        pc.apertureRadius = params[16]; pc.focalDistance = params[17];
        DevCamera cam = DevCamera();
        camera_basis(pc, cam);
        const Ray r = create_cam_ray((int)x[0], (int)x[1], (int)x[2], (int)x[3], cam, rng);
        y[0] = r.origin.x; y[1] = r.origin.y; y[2] = r.origin.z; y[3] = r.dir.x; y[4] = r.dir.y; y[5] = r.dir.z; y[6] = r.time;
    } break;
    case 10: {                                  // geometry/sphere.cl:5-41, quad.cl:11-38 (the sphere's normal / position are intersect_scene's job)
        Ray ray;
        ray.origin = F3(x[0], x[1], x[2]); ray.dir = F3(x[3], x[4], x[5]); ray.normal = splat(0.0f); ray.pos = splat(0.0f);
        ray.t = x[6]; ray.backside = false; ray.time = 0.0f;
        float t = x[6];
        if (prt_f2u(params[19]) & PRT_GEOM_SPHERE) {
            DevSphere s;
            s.pos[0] = params[0]; s.pos[1] = params[1]; s.pos[2] = params[2]; s.radius = params[3];
            const bool hit = hit_sphere(s, ray, t);
            y[0] = hit ? 1.0f : 0.0f; y[1] = t;
        } else {
            DevQuad q;
            pack_quad(params + 3, q);
            f3 pos = splat(0.0f);
            const bool hit = hit_quad(q, ray, t, pos);
            y[0] = hit ? 1.0f : 0.0f; y[1] = t;
            if (hit) { y[2] = q.normal[0]; y[3] = q.normal[1]; y[4] = q.normal[2]; y[5] = pos.x; y[6] = pos.y; y[7] = pos.z; }
        }
    } break;
    case 11: {                                  // env-map lookup: utils.cl:46 + read_imagef(CLK_NORMALIZED_COORDS_TRUE | CLK_ADDRESS_CLAMP |
        sc.env_w = (int)prt_f2u(params[0]); sc.env_h = (int)prt_f2u(params[1]);   // CLK_FILTER_LINEAR), main.cl:25; params: w, h, then w*h RGB texels
        sc.env = params + 2;
        const f3 c = env_lookup(sc, F3(x[0], x[1], x[2]));
        y[0] = c.x; y[1] = c.y; y[2] = c.z;
    } break;
    default: break;
    }
    y[30] = prt_u2f(rng.s0); y[31] = prt_u2f(rng.s1);
}

}  // namespace dev
}  // namespace prt
