/* oracle/ref/kat_harness.cl -- TEST INFRASTRUCTURE (own code, no reference text).
 *
 * Appended by oracle/ref/build_ref.py --kat to the reference's assembled kernel text, so that it sits in the same
 * translation unit as the reference's functions and can call them directly: per-function known-answer vectors of the
 * REFERENCE'S OWN code (tests/golden/kat_*.npz, generator tests/golden/make_kat.py).  One entry point, plain arrays:
 *   params  up to 80 floats shared by all cases of a call (a Material / Mesh / Camera record, medium coefficients)
 *   in/out  32 floats per case
 * uint values (seeds, type bits) travel as float bit patterns (as_uint / as_float).
 * The layouts are mirrored by prt_selftest_fn (csrc/hip/pt_kernels.hip) and documented in tests/golden/make_kat.py. */

static Material kat_material(__global const float* p) {
    Material m;
    m.color = (float3)(p[0], p[1], p[2]);
    m.eta = (float3)(p[3], p[4], p[5]);
    m.k = (float3)(p[6], p[7], p[8]);
    m.roughness = p[9];
    m.t = (ushort)as_uint(p[10]);
    m.lobes = (uchar)as_uint(p[11]);
    m.dist = (uchar)as_uint(p[12]);
    return m;
}

static Mesh kat_mesh(__global const float* p) {          /* pos, joker[16], type */
    Mesh m;
    m.pos = (float3)(p[0], p[1], p[2]);
    m.joker = (float16)(p[3], p[4], p[5], p[6], p[7], p[8], p[9], p[10], p[11], p[12], p[13], p[14], p[15], p[16], p[17], p[18]);
    m.t = (uchar)as_uint(p[19]);
    return m;
}

void kat_run(int fn, __global const float* params, __constant float* cparams, __global const float* in, __global float* out, int n) {
    for (int i = 0; i < n; ++i) {
        __global const float* x = in + 32 * i;
        __global float* y = out + 32 * i;
        for (int k = 0; k < 32; ++k) y[k] = 0.0f;
        uint s0 = as_uint(x[30]), s1 = as_uint(x[31]);
        switch (fn) {
        case 1: {   /* BSDF2: in wi.xyz, shading normal.xyz, ray.t, ray.backside -> ok, wo, weight, pdf, sampledLobe */
            Material mat = kat_material(params);
            SurfaceScatterEvent e;
            float3 nrm = (float3)(x[3], x[4], x[5]);
            e.frame = createTangentFrame(&nrm);
            e.wi = (float3)(x[0], x[1], x[2]);
            e.wo = (float3)(0.0f); e.weight = (float3)(1.0f); e.pdf = 1.0f; e.requestedLobe = 0; e.sampledLobe = 0;
            Ray ray;
            ray.origin = (float3)(0.0f); ray.dir = (float3)(0.0f, 0.0f, 1.0f); ray.normal = nrm; ray.pos = (float3)(0.0f);
            ray.t = x[6]; ray.backside = x[7] != 0.0f; ray.time = 0.0f;
            bool ok = BSDF2(&e, &ray, (const Scene*)0, &mat, &s0, &s1, false);
            y[0] = ok ? 1.0f : 0.0f;
            y[1] = e.wo.x; y[2] = e.wo.y; y[3] = e.wo.z;
            y[4] = e.weight.x; y[5] = e.weight.y; y[6] = e.weight.z;
            y[7] = e.pdf; y[8] = as_float((uint)e.sampledLobe);
        } break;
        case 2: {   /* BSDF_eval2 / BSDF_pdf: in wi.xyz, wo.xyz -> f.xyz, pdf (pdf only where in[6] != 0: LambertBSDF_pdf is UB) */
            Material mat = kat_material(params);
            SurfaceScatterEvent e;
            float3 nrm = (float3)(0.0f, 0.0f, 1.0f);
            e.frame = createTangentFrame(&nrm);
            e.wi = (float3)(x[0], x[1], x[2]);
            e.wo = (float3)(x[3], x[4], x[5]);
            e.weight = (float3)(1.0f); e.pdf = 1.0f; e.requestedLobe = 0; e.sampledLobe = 0;
            float3 f = BSDF_eval2(&e, &mat, false);
            y[0] = f.x; y[1] = f.y; y[2] = f.z;
            y[3] = (x[6] != 0.0f) ? BSDF_pdf(&e, &mat) : 0.0f;
        } break;
        case 3: {   /* microfacet: params dist, roughness; in v.xyz, m.xyz, xi.xy -> alpha, D, G1, pdf, sample.xyz */
            int dist = (int)as_uint(params[0]);
            float alpha = roughnessToAlpha(dist, params[1]);
            float3 v = (float3)(x[0], x[1], x[2]), m = (float3)(x[3], x[4], x[5]);
            y[0] = alpha;
            y[1] = Microfacet_D(dist, alpha, m);
            y[2] = Microfacet_G1(dist, alpha, v, m);
            y[3] = Microfacet_pdf(dist, alpha, m);
            float3 s = Microfacet_sample(dist, alpha, (float2)(x[6], x[7]));
            y[4] = s.x; y[5] = s.y; y[6] = s.z;
        } break;
        case 4: {   /* Fresnel: in eta, k, cosThetaI -> conductorReflectance, dielectricReflectance, cosThetaT */
            float ct = 0.0f;
            y[0] = conductorReflectance(x[0], x[1], x[2]);
            y[1] = dielectricReflectance(x[0], x[2], &ct);
            y[2] = ct;
        } break;
        case 5: {   /* sphere light: params Mesh; in p.xyz -> ok, d.xyz, dist, pdf, directPdf */
            Mesh m = kat_mesh(params);
            float3 p = (float3)(x[0], x[1], x[2]);
            LightSample ls; ls.d = (float3)(0.0f); ls.dist = 0.0f; ls.pdf = 0.0f;
            bool ok = sphere_sampleDirect(&m, &p, &ls, &s0, &s1);
            y[0] = ok ? 1.0f : 0.0f; y[1] = ls.d.x; y[2] = ls.d.y; y[3] = ls.d.z; y[4] = ls.dist; y[5] = ls.pdf;
            y[6] = sphere_directPdf(&m, &p);
        } break;
        case 6: {   /* quad light: params Mesh; in p.xyz, dir.xyz -> ok, d.xyz, dist, pdf, directPdf(dir, p) */
            Mesh m = kat_mesh(params);
            float3 p = (float3)(x[0], x[1], x[2]), dir = (float3)(x[3], x[4], x[5]);
            LightSample ls; ls.d = (float3)(0.0f); ls.dist = 0.0f; ls.pdf = 0.0f;
            bool ok = quad_sampleDirect(&m, &p, &ls, &s0, &s1);
            y[0] = ok ? 1.0f : 0.0f; y[1] = ls.d.x; y[2] = ls.d.y; y[3] = ls.d.z; y[4] = ls.dist; y[5] = ls.pdf;
            y[6] = quad_directPdf(&dir, &m, &p);
        } break;
#ifdef GLOBAL_MEDIUM
        case 7: {   /* HomogeneousMedium_sampleDistance: params sigmaA, sigmaS, sigmaT, absOnly; in origin, dir, maxT -> p, weight, exited, t, pdf */
            Medium med;
            med.density = (float3)(1.0f); med.sigmaA = (float3)(params[0]); med.sigmaS = (float3)(params[1]); med.sigmaT = (float3)(params[2]);
            med.absorptionOnly = params[3] != 0.0f;
            ((float*)&med.sigmaT)[3] = 0.0f;          /* the padding lane of the reference's constant Medium */
            Ray ray;
            ray.origin = (float3)(x[0], x[1], x[2]); ray.dir = (float3)(x[3], x[4], x[5]); ray.normal = (float3)(0.0f); ray.pos = (float3)(0.0f);
            ray.t = x[6]; ray.backside = false; ray.time = 0.0f;
            MediumSample ms;
            ms.continuedWeight = (float3)(1.0f);
            HomogeneousMedium_sampleDistance(&ms, &med, &ray, &s0, &s1);
            y[0] = ms.p.x; y[1] = ms.p.y; y[2] = ms.p.z; y[3] = ms.weight.x; y[4] = ms.weight.y; y[5] = ms.weight.z;
            y[6] = ms.exited ? 1.0f : 0.0f; y[7] = ms.t; y[8] = ms.pdf;
        } break;
        case 8: {   /* phase function: in wi.xyz, wo.xyz -> ok, w.xyz, weight.xyz, pdf, eval.x, phase_pdf */
            float3 wi = (float3)(x[0], x[1], x[2]), wo = (float3)(x[3], x[4], x[5]);
            PhaseSample ps; ps.w = (float3)(0.0f); ps.weight = (float3)(0.0f); ps.pdf = 0.0f;
            bool ok = phase_sample(wi, &ps, &s0, &s1);
            y[0] = ok ? 1.0f : 0.0f; y[1] = ps.w.x; y[2] = ps.w.y; y[3] = ps.w.z; y[4] = ps.weight.x; y[5] = ps.weight.y; y[6] = ps.weight.z;
            y[7] = ps.pdf; y[8] = phase_eval(wi, wo).x; y[9] = phase_pdf(wi, wo);
        } break;
#endif
        case 9: {   /* createCamRay: cparams Camera (80 bytes); in x, y, width, height -> origin, dir, time */
            Ray r = createCamRay((int2)((int)x[0], (int)x[1]), (int)x[2], (int)x[3], (__constant Camera*)cparams, &s0, &s1);
            y[0] = r.origin.x; y[1] = r.origin.y; y[2] = r.origin.z; y[3] = r.dir.x; y[4] = r.dir.y; y[5] = r.dir.z; y[6] = r.time;
        } break;
        case 10: {  /* intersect_sphere / intersect_quad: params Mesh (type bit decides); in origin, dir, t -> hit, t, normal, pos */
            Mesh m = kat_mesh(params);
            Ray ray;
            ray.origin = (float3)(x[0], x[1], x[2]); ray.dir = (float3)(x[3], x[4], x[5]); ray.normal = (float3)(0.0f); ray.pos = (float3)(0.0f);
            ray.t = x[6]; ray.backside = false; ray.time = 0.0f;
            bool hit = (m.t & SPHERE) ? intersect_sphere(&ray, &m) : intersect_quad(&m, &ray);
            y[0] = hit ? 1.0f : 0.0f; y[1] = ray.t; y[2] = ray.normal.x; y[3] = ray.normal.y; y[4] = ray.normal.z;
            y[5] = ray.pos.x; y[6] = ray.pos.y; y[7] = ray.pos.z;
        } break;
        default: break;
        }
        y[30] = as_float(s0); y[31] = as_float(s1);
    }
}
