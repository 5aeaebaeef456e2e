// development microbenchmark: how many scalar-ALU instructions per cycle a CU issues, by waves per SIMD, alone and beside a
// vector stream (the render kernel issues 0.46 scalar per vector instruction; is the scalar unit shared by the CU's four SIMDs?)
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/salu_rate.hip -o gpurun_out/salu_rate ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int MODE>   // 0: scalar only, 1: vector only, 2: both interleaved 1:2 (S:V), 3: branch-heavy
__global__ void __launch_bounds__(256) k(unsigned* out, int iters) {
    unsigned s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3;
    float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        } else if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                asm volatile("v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %3, %3, %3" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
        } else if (MODE == 2) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                asm volatile("v_add_f32 %0, %0, %0\n s_add_u32 %4, %4, 1\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n s_add_u32 %5, %5, 1\n v_add_f32 %3, %3, %3"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+s"(s0), "+s"(s1) : : "scc");
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                asm volatile("v_add_f32 %0, %0, %0\n s_cmp_lg_u32 %4, 0\n s_cbranch_scc1 1f\n v_add_f32 %1, %1, %1\n1:\n v_add_f32 %2, %2, %2\n v_add_f32 %3, %3, %3"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+s"(s1) : : "scc");
        }
    }
    if (s0 + s1 + s2 + s3 == 0xffffffffu || v0 + v1 + v2 + v3 == 12345.f) out[0] = 1;
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    unsigned* d; hipMalloc(&d, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2000;
    for (int mode = 0; mode < 4; ++mode)
        for (int wps = 1; wps <= 8; ++wps) {                 // waves per SIMD: wps blocks of 256 threads per CU
            const int grid = 256 * wps;
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(a);
                if (mode == 0) k<0><<<grid, 256>>>(d, iters); else if (mode == 1) k<1><<<grid, 256>>>(d, iters);
                else if (mode == 2) k<2><<<grid, 256>>>(d, iters); else k<3><<<grid, 256>>>(d, iters);
                hipEventRecord(b); hipEventSynchronize(b);
                if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
                float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
            }
            const double per_wave = mode == 0 || mode == 1 ? 64.0 * iters : (mode == 2 ? 96.0 * iters : 64.0 * 1.5 * iters);
            const double inst = per_wave * wps * 4;          // per CU
            printf("mode %d waves/SIMD %d  %.3f ms  %.3f instr/ns/CU  (at 2.4 GHz: %.3f per cycle per CU)\n", mode, wps, best, inst / (best * 1e6), inst / (best * 1e6) / 2.4);
        }
    return 0;
}
