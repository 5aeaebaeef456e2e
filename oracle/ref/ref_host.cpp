// oracle/ref/ref_host.cpp -- TEST INFRASTRUCTURE (oracle side), never linked into the product.
//
// Thin driver around the REFERENCE's own host code, compiled from the sources where they
// lie under /root/reference (nothing is copied into this repository):
//   include/Scene/scene.h      host_scene::load()          (JSON -> Mesh[] / Material / limits)
//   include/CL/cl_kernel.h     cl_help::kernel::parse()    (#FILE include + #TOKEN# specialiser)
//   include/Camera/camera.h, src/Camera/camera.cpp         (InteractiveCamera -> 80-byte Camera)
// It replays what src/main.cpp:323-427 does before the first enqueue and writes
//   (1) the scene-specialised kernel text to <out.cl>   (a temp file, deleted by the recipe)
//   (2) a section file <out.blob> with the exact host buffers main.cpp would upload.
//
// usage: ref_host <scene.json> <width> <height> <alpha 0|1> <out.cl> <out.blob>
// cwd must be a directory whose "../kernels" resolves to the reference kernels
// (the recipe builds a symlink farm so that "#FILE:bxdf/materials/.." resolves).
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <Camera/camera.h>
#include <Scene/scene.h>
#include <CL/cl_kernel.h>

std::string scene_filepath;      // extern in Scene/scene.h:16
bool ALPHA_TESTING = false;      // extern in CL/cl_kernel.h:6

static void put(FILE* f, const char* name, const void* data, size_t n) {
    char tag[16] = {0};
    strncpy(tag, name, 15);
    unsigned long long nb = n;
    fwrite(tag, 1, 16, f);
    fwrite(&nb, 8, 1, f);
    if (n) fwrite(data, 1, n, f);
}

int main(int argc, char** argv) {
    if (argc < 7) { fprintf(stderr, "usage\n"); return 2; }
    scene_filepath = argv[1];
    int W = atoi(argv[2]), H = atoi(argv[3]);
    ALPHA_TESTING = atoi(argv[4]) != 0;

    host_scene* scene = new host_scene();           // src/main.cpp:375 (value-init => object_count zeroed? no: see below)
    memset(&scene->object_count, 0, sizeof(scene->object_count)); // heap garbage guard; main.cpp relies on fresh zero pages
    scene->load();

    // src/main.cpp:312-319 initCamera + :294 buildRenderCamera
    InteractiveCamera ic;
    ic.setResolution(W, H);
    ic.setFOVX(45.0f);
    Camera cam;
    memset(&cam, 0, sizeof(cam));
    ic.buildRenderCamera(&cam);

    std::string src = cl_help::kernel::parse("../kernels/main.cl", scene);
    FILE* fc = fopen(argv[5], "wb");
    if (!fc) { perror("out.cl"); return 1; }
    fwrite(src.data(), 1, src.size(), fc);
    fclose(fc);

    FILE* fb = fopen(argv[6], "wb");
    if (!fb) { perror("out.blob"); return 1; }
    put(fb, "meshes", scene->cpu_meshes.data(), scene->cpu_meshes.size() * sizeof(Mesh));
    put(fb, "counts", &scene->object_count, sizeof(scene->object_count));
    put(fb, "objmat", scene->obj_mat, sizeof(Material));
    put(fb, "camera", &cam, sizeof(Camera));
    int ints[16] = { scene->MAX_BOUNCES, scene->MAX_DIFF_BOUNCES, scene->MAX_SPEC_BOUNCES,
                     scene->MAX_TRANS_BOUNCES, scene->MAX_SCATTERING_EVENTS, scene->MARCHING_STEPS,
                     scene->SHADOW_MARCHING_STEPS, scene->ACTIVE_MATS, scene->H_SPHERE, scene->H_SDF,
                     scene->H_BOX, scene->H_QUAD, (int)scene->LIGHT_COUNT, (int)scene->HAS_GLOBAL_MEDIUM,
                     (int)scene->BUILD_BVH, (int)ALPHA_TESTING };
    put(fb, "ints", ints, sizeof(ints));
    put(fb, "lights", scene->LIGHT_INDICES.data(), scene->LIGHT_INDICES.size() * sizeof(cl_uint));
    float med[5] = {0, 0, 0, 0, 0};
    if (scene->HAS_GLOBAL_MEDIUM) {
        med[0] = scene->GLOBAL_MEDIUM.density; med[1] = scene->GLOBAL_MEDIUM.sigmaA;
        med[2] = scene->GLOBAL_MEDIUM.sigmaS;  med[3] = scene->GLOBAL_MEDIUM.sigmaT;
        med[4] = scene->GLOBAL_MEDIUM.absorptionOnly ? 1.0f : 0.0f;
    }
    put(fb, "medium", med, sizeof(med));
    put(fb, "objpath", scene->obj_path.data(), scene->obj_path.size());
    int sizes[4] = { (int)sizeof(Mesh), (int)sizeof(Material), (int)sizeof(Camera), 0 };
    put(fb, "sizes", sizes, sizeof(sizes));
    fclose(fb);
    return 0;
}
