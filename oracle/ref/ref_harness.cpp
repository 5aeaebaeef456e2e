// oracle/ref/ref_harness.cpp -- TEST INFRASTRUCTURE (oracle side), never linked into the product.
//
// Replays the launch protocol of the reference host (src/main.cpp:280-307 render(),
// :247-276 runKernel(), :213-239 initCLKernel()) around the reference's own `render_kernel`
// (kernels/main.cl:66-163) compiled for x86-64: one call of the kernel function per
// work-item, global id 0 .. W*H-1 (no padding work-items: SURVEY §9-Q12).
//
// Decisions that had to be made where the reference is undefined:
//   Q6   scene->meshes[-1] is read for OBJ hits: the harness places a zeroed 256-byte guard
//        Mesh in front of meshes[0].
//   Q18  no -hdr: a 1x1 black environment image.
//   spp  "N spp" = a pixel is frozen once its N-th path has terminated (reset set and
//        samples == N); frozen pixels are skipped.  spp_limit = 0 disables freezing
//        (= the reference's progressive loop).
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <thread>
#include <vector>

typedef unsigned uint8v __attribute__((ext_vector_type(8)));

struct ShimImage { float* data; int width, height, channels; };
extern "C" void shim_set_global_id(size_t id);

extern "C" void render_kernel(const void* meshes, int width, int height, uint8v mesh_count,
                              unsigned framenumber, const void* cam, int random0, int random1,
                              void* output_tex, const void* primitive_indices, const void* vertices,
                              const void* normals, const void* mat, void* env_map, void* r_flat,
                              const void* new_bvh_node);

static const size_t RTD_SIZE = 112;      // src/main.cpp:39
static const size_t OFF_RESET = 48 + 48; // RTD.data.reset   (kernels/main.cl:30-47,58-61)
static const size_t OFF_SAMPLES = 48 + 52;

extern "C" int ref_rtd_size(void) { return (int)RTD_SIZE; }

// Renders frames first_frame .. first_frame+n_frames-1 (frame numbers start at 1).
// seed_pairs holds 2*n_frames ints: (random0, random1) per frame.
extern "C" int ref_render_frames(const void* meshes, int n_meshes, const unsigned* counts8,
                                 int W, int H, unsigned first_frame, int n_frames,
                                 const int* seed_pairs, const void* cam,
                                 const void* indices_u64, const void* vertices, const void* normals,
                                 const void* obj_mat, const void* bvh_nodes,
                                 const float* env_rgb, int env_w, int env_h,
                                 void* r_flat, float* out_rgba, unsigned spp_limit, int n_threads) {
    // Q6 guard mesh
    std::vector<unsigned char> mbuf((size_t)(n_meshes + 1) * 256 + 256, 0);
    unsigned char* mbase = mbuf.data();
    mbase += (256 - ((uintptr_t)mbase & 255)) & 255;
    memcpy(mbase + 256, meshes, (size_t)n_meshes * 256);
    const void* meshes0 = mbase + 256;

    float black[3] = {0.f, 0.f, 0.f};
    ShimImage env = { const_cast<float*>(env_rgb ? env_rgb : black), env_rgb ? env_w : 1, env_rgb ? env_h : 1, 3 };
    ShimImage out = { out_rgba, W, H, 4 };
    uint8v counts;
    for (int i = 0; i < 8; ++i) counts[i] = counts8[i];

    if (n_threads < 1) n_threads = 1;
    const size_t npix = (size_t)W * H;
    for (int f = 0; f < n_frames; ++f) {
        const unsigned frame = first_frame + (unsigned)f;
        const int r0 = seed_pairs[2 * f], r1 = seed_pairs[2 * f + 1];
        auto work = [&](size_t lo, size_t hi) {
            unsigned char* st = (unsigned char*)r_flat;
            for (size_t id = lo; id < hi; ++id) {
                if (spp_limit) {
                    unsigned samples; memcpy(&samples, st + id * RTD_SIZE + OFF_SAMPLES, 4);
                    if (st[id * RTD_SIZE + OFF_RESET] && samples >= spp_limit) continue;
                }
                shim_set_global_id(id);
                render_kernel(meshes0, W, H, counts, frame, cam, r0, r1, &out, indices_u64, vertices,
                              normals, obj_mat, &env, r_flat, bvh_nodes);
            }
        };
        if (n_threads == 1) { work(0, npix); continue; }
        std::vector<std::thread> th;
        size_t chunk = (npix + n_threads - 1) / n_threads;
        for (int t = 0; t < n_threads; ++t) {
            size_t lo = (size_t)t * chunk, hi = lo + chunk < npix ? lo + chunk : npix;
            if (lo < hi) th.emplace_back(work, lo, hi);
        }
        for (auto& t : th) t.join();
    }
    return 0;
}
