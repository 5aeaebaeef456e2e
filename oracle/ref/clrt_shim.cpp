// oracle/ref/clrt_shim.cpp -- TEST INFRASTRUCTURE (oracle side), never linked into the product.
//
// A minimal OpenCL C *runtime* for running the reference's own kernel text on the host.
// The reference kernel (/root/reference/kernels/main.cl, specialised by the reference's own
// include/CL/cl_kernel.h) is compiled, unmodified, by ROCm clang as OpenCL C for x86-64
// (oracle/ref/build_ref.py).  That object leaves the OpenCL built-in library undefined; this
// file defines it:
//   * work-item functions      get_global_id                      (harness sets the id)
//   * image functions          read_imagef / write_imagef / sampler init   (OpenCL 1.2 §8.2
//                              CLK_NORMALIZED_COORDS_TRUE | CLK_ADDRESS_CLAMP | CLK_FILTER_LINEAR)
//   * math / geometric         every one in terms of include/prt_detmath.h, the numerics
//                              contract of the C-ABI (see that header for why).
//                              -DSHIM_LIBM (build_ref.py --math libm): the second flavour -- the scalar
//                              transcendental built-ins (sin cos tan acos atan2 exp log pow cbrt and their
//                              native_ forms) forward to the GNU C library's libm instead: math this repository
//                              did not write.  Pictures of the two flavours differ in individual decisions
//                              (a path tracer is chaotic) but must agree as estimates: tools/independent_math.py,
//                              tests/test_oracle.py.  The exactly rounded operations (sqrt, divide, fma, fmin /
//                              fmax, rounding) and the geometric / image functions are the same in both.
// Vector forms are component-wise; dot/cross/length/normalize evaluate left to right with one
// rounding per operation (no fma), which is also what oracle/pt_oracle.c and the HIP kernels do.
//
// Build: clang++ -O2 -ffp-contract=off (same compiler and -march as the kernel object, so the
// ext_vector_type calling convention matches).
#include <cstring>
#include <cstdint>
#include "prt_detmath.h"

#ifdef SHIM_LIBM
// (declared here rather than through <math.h>, whose C++ overloads would collide with the built-ins this file defines)
extern "C" { float sinf(float); float cosf(float); float tanf(float); float acosf(float); float atan2f(float, float);
             float expf(float); float logf(float); float powf(float, float); float cbrtf(float); }
#define T_SIN(x) ::sinf(x)
#define T_COS(x) ::cosf(x)
#define T_TAN(x) ::tanf(x)
#define T_ACOS(x) ::acosf(x)
#define T_ATAN2(y, x) ::atan2f(y, x)
#define T_EXP(x) ::expf(x)
#define T_LOG(x) ::logf(x)
#define T_POW(x, y) ::powf(x, y)
#define T_CBRT(x) ::cbrtf(x)
#else
#define T_SIN(x) prt_sin(x)
#define T_COS(x) prt_cos(x)
#define T_TAN(x) prt_tan(x)
#define T_ACOS(x) prt_acos(x)
#define T_ATAN2(y, x) prt_atan2(y, x)
#define T_EXP(x) prt_exp(x)
#define T_LOG(x) prt_log(x)
#define T_POW(x, y) prt_pow(x, y)
#define T_CBRT(x) prt_cbrt(x)
#endif

typedef float float2 __attribute__((ext_vector_type(2)));
typedef float float3 __attribute__((ext_vector_type(3)));
typedef float float4 __attribute__((ext_vector_type(4)));
typedef int int2 __attribute__((ext_vector_type(2)));

// ---- harness-visible state ---------------------------------------------------------------
struct ShimImage {
    float* data;     // row-major, `channels` floats per texel
    int width, height, channels;
};
static thread_local size_t g_global_id = 0;
extern "C" void shim_set_global_id(size_t id) { g_global_id = id; }

// ---- work-item / image functions ------------------------------------------------------------
size_t shim_get_global_id(unsigned dim) __asm__("_Z13get_global_idj");
size_t shim_get_global_id(unsigned dim) { return dim == 0 ? g_global_id : 0; }

extern "C" void* __translate_sampler_initializer(int v) { return (void*)(intptr_t)(v | 0x40000000); }

static inline float4 texel(const ShimImage* im, int i, int j) {
    // CLK_ADDRESS_CLAMP: out-of-range coordinates return the border colour; for an image
    // without an alpha channel (the reference shares a GL_RGB32F texture,
    // include/GL/cl_gl_interop.h:71-86) that is (0,0,0,1).
    if (i < 0 || j < 0 || i >= im->width || j >= im->height) return float4{0.f, 0.f, 0.f, 1.f};
    const float* p = im->data + ((size_t)j * im->width + i) * im->channels;
    return float4{p[0], p[1], p[2], im->channels > 3 ? p[3] : 1.f};
}
float4 shim_read_imagef(const ShimImage* im, void* sampler, float2 c)
    __asm__("_Z11read_imagef14ocl_image2d_ro11ocl_samplerDv2_f");
float4 shim_read_imagef(const ShimImage* im, void* sampler, float2 c) {
    // OpenCL 1.2 §8.2 linear filter with normalised coordinates
    float u = c.x * (float)im->width, v = c.y * (float)im->height;
    float fu = prt_floor(u - 0.5f), fv = prt_floor(v - 0.5f);
    float a = (u - 0.5f) - fu, b = (v - 0.5f) - fv;
    int i0 = (int)fu, j0 = (int)fv, i1 = i0 + 1, j1 = j0 + 1;
    float4 t00 = texel(im, i0, j0), t10 = texel(im, i1, j0), t01 = texel(im, i0, j1), t11 = texel(im, i1, j1);
    return (1.f - a) * (1.f - b) * t00 + a * (1.f - b) * t10 + (1.f - a) * b * t01 + a * b * t11;
}
void shim_write_imagef(ShimImage* im, int2 c, float4 v) __asm__("_Z12write_imagef14ocl_image2d_woDv2_iDv4_f");
void shim_write_imagef(ShimImage* im, int2 c, float4 v) {
    if (c.x < 0 || c.y < 0 || c.x >= im->width || c.y >= im->height) return;
    float* p = im->data + ((size_t)c.y * im->width + c.x) * 4;
    p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
}

// ---- scalar math ---------------------------------------------------------------------------
float sin(float x) { return T_SIN(x); }
float cos(float x) { return T_COS(x); }
float tan(float x) { return T_TAN(x); }
float acos(float x) { return T_ACOS(x); }
float atan2(float y, float x) { return T_ATAN2(y, x); }
float exp(float x) { return T_EXP(x); }
float log(float x) { return T_LOG(x); }
float pow(float x, float y) { return T_POW(x, y); }
float sqrt(float x) { return prt_sqrt(x); }
float fabs(float x) { return prt_fabs(x); }
float fmin(float a, float b) { return prt_fmin(a, b); }
float fmax(float a, float b) { return prt_fmax(a, b); }
float fma(float a, float b, float c) { return prt_fma(a, b, c); }
float mix(float a, float b, float t) { return prt_mix(a, b, t); }
float round(float x) { return prt_round(x); }
float copysign(float x, float s) { return prt_copysign(x, s); }
float native_sin(float x) { return T_SIN(x); }
float native_cos(float x) { return T_COS(x); }
float native_exp(float x) { return T_EXP(x); }
float native_log(float x) { return T_LOG(x); }
float native_sqrt(float x) { return prt_sqrt(x); }
float cbrt(float x) { return T_CBRT(x); }
float native_recip(float x) { return prt_recip(x); }
float shim_fract1(float x, float* ip) __asm__("_Z5fractfPU9CLprivatef");
float shim_fract1(float x, float* ip) { *ip = prt_floor(x); return prt_fract(x); }

// ---- float3 forms --------------------------------------------------------------------------
#define V3(fn) float3{fn(v.x), fn(v.y), fn(v.z)}
float3 exp(float3 v) { return float3{T_EXP(v.x), T_EXP(v.y), T_EXP(v.z)}; }
float3 native_exp(float3 v) { return float3{T_EXP(v.x), T_EXP(v.y), T_EXP(v.z)}; }
float3 native_recip(float3 v) { return V3(prt_recip); }
float3 fmin(float3 a, float3 b) { return float3{prt_fmin(a.x, b.x), prt_fmin(a.y, b.y), prt_fmin(a.z, b.z)}; }
float3 fmax(float3 a, float3 b) { return float3{prt_fmax(a.x, b.x), prt_fmax(a.y, b.y), prt_fmax(a.z, b.z)}; }
float3 shim_fract3(float3 v, float3* ip) __asm__("_Z5fractDv3_fPU9CLprivateS_");
float3 shim_fract3(float3 v, float3* ip) {
    *ip = float3{prt_floor(v.x), prt_floor(v.y), prt_floor(v.z)};
    return V3(prt_fract);
}
float3 fabs(float3 v) { return V3(prt_fabs); }
float3 fmax(float3 a, float b) { return float3{prt_fmax(a.x, b), prt_fmax(a.y, b), prt_fmax(a.z, b)}; }
float fast_length(float2 v) { return prt_sqrt(v.x * v.x + v.y * v.y); }
float dot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
float dot(float2 a, float2 b) { return a.x * b.x + a.y * b.y; }
float3 cross(float3 a, float3 b) {
    return float3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
float length(float3 v) { return prt_sqrt(v.x * v.x + v.y * v.y + v.z * v.z); }
float fast_length(float3 v) { return prt_sqrt(v.x * v.x + v.y * v.y + v.z * v.z); }
float3 normalize(float3 v) {
    float inv = 1.0f / prt_sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
    return float3{v.x * inv, v.y * inv, v.z * inv};
}
float3 fast_normalize(float3 v) {
    float inv = 1.0f / prt_sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
    return float3{v.x * inv, v.y * inv, v.z * inv};
}
