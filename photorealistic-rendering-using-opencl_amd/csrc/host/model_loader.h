// model_loader.h -- triangle-soup loader with the reference's Scene{meshes{faces{points{pos,nor}}}}
// view (include/Model/model_loader.h:21-62).  The reference imports through assimp
// (src/Models/model_loader.cpp:38), which is not available; this loader reads Wavefront OBJ
// (v / vn / f, fan-triangulated polygons, negative indices), Stanford PLY (ascii / binary little-endian), STL (ascii / binary) and
// the ".prtmesh" binary soup ("PRTMESH1", uint32 triangle_count, then per triangle 3 x {pos xyz, nor xyz} float32).  An OBJ file with
// several objects / groups / materials gives several meshes, in file order, as assimp's importer does (normals that have to be generated
// are smoothed within a mesh); every consumer walks all meshes (src/Models/model_loader.cpp:58-74, src/main.cpp:93-119), so the soup is
// their concatenation.
#pragma once
#include <array>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace prt {
namespace IO {

struct float3 { float x = 0.f, y = 0.f, z = 0.f; };
struct Vertex { float3 pos, nor; };
struct Face { std::array<Vertex, 3> points; };
struct Mesh { std::vector<Face> faces; };
struct Scene { std::vector<Mesh> meshes; };

class ModelLoader {
public:
    // true on success; on failure last_error() says why (the reference prints and carries on)
    bool ImportFromFile(const std::string& filepath);
    const Scene& getFaces() const { return scene_; }
    size_t triangleCount() const;
    const std::string& last_error() const { return err_; }

    // flattened buffers exactly as src/main.cpp:93-119 uploads them: float4 per corner (w = 0)
    void flatten(std::vector<float>& vertices4, std::vector<float>& normals4) const;
    bool SaveSoup(const std::string& filepath) const;
    // the indexed view of one mesh (aiProcess_JoinIdenticalVertices): unique (position, normal) vertices in order of first use + 3 indices per face
    void weld(size_t mesh_index, std::vector<Vertex>& vertices, std::vector<uint32_t>& indices) const;

private:
    bool load_obj(const std::string& path);
    bool load_ply(const std::string& path);
    bool load_stl(const std::string& path);
    bool load_soup(const std::string& path);
    Scene scene_;
    std::string err_;
};

}  // namespace IO
}  // namespace prt
