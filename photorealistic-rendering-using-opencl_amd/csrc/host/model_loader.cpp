#include "model_loader.h"

#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

namespace prt {
namespace IO {

static bool ends_with(const std::string& s, const char* suf) {
    size_t n = std::strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

bool ModelLoader::ImportFromFile(const std::string& path) {
    scene_.meshes.clear();
    err_.clear();
    if (ends_with(path, ".prtmesh")) return load_soup(path);
    std::ifstream probe(path);
    if (!probe) {
        // "<name>.obj" missing: fall back to the pre-flattened soup next to it
        std::string alt = path.substr(0, path.find_last_of('.')) + ".prtmesh";
        std::ifstream p2(alt, std::ios::binary);
        if (p2) return load_soup(alt);
        err_ = "cannot open model '" + path + "'";
        return false;
    }
    std::string low = path;
    for (char& ch : low) ch = (char)std::tolower((unsigned char)ch);
    if (ends_with(low, ".ply")) return load_ply(path);
    if (ends_with(low, ".stl")) return load_stl(path);
    return load_obj(path);
}

static float3 unit_face_normal(const float3& a, const float3& b, const float3& c) {
    float ux = b.x - a.x, uy = b.y - a.y, uz = b.z - a.z, vx = c.x - a.x, vy = c.y - a.y, vz = c.z - a.z;
    float3 n{uy * vz - uz * vy, uz * vx - ux * vz, ux * vy - uy * vx};
    float l = sqrtf(n.x * n.x + n.y * n.y + n.z * n.z);
    if (l > 0.f) { n.x /= l; n.y /= l; n.z /= l; }
    return n;
}

// Faces that came without normals (their corners hold the unit geometric normal): the reference imports with
// aiProcessPreset_TargetRealtime_Quality (src/Models/model_loader.cpp:38), whose GenSmoothNormals step gives every vertex the
// normalised sum of the unit normals of the faces that meet at its position (smoothing-angle limit 175 degrees).  Same rule here, on
// bit-identical positions (assimp is an absent submodule, so this step has no oracle: SURVEY s8f row N2).  The preset's
// JoinIdenticalVertices has no effect on this path: the renderer de-indexes every mesh into a triangle soup (src/main.cpp:93-119).
static void smooth_missing_normals(Mesh& mesh, const std::vector<size_t>& needs_normals) {
    if (needs_normals.empty()) return;
    struct Key { uint32_t x, y, z; bool operator<(const Key& o) const { return x != o.x ? x < o.x : (y != o.y ? y < o.y : z < o.z); } };
    auto key = [](const float3& p) { Key k; float z0 = p.x + 0.0f, z1 = p.y + 0.0f, z2 = p.z + 0.0f;      // -0 -> +0
                                     std::memcpy(&k.x, &z0, 4); std::memcpy(&k.y, &z1, 4); std::memcpy(&k.z, &z2, 4); return k; };
    std::map<Key, std::vector<float3>> at;               // position -> unit normals of the faces around it
    for (size_t fi : needs_normals)
        for (auto& p : mesh.faces[fi].points) at[key(p.pos)].push_back(p.nor);
    const float cos_limit = -0.99619470f;                // cos(175 deg)
    for (size_t fi : needs_normals)
        for (auto& p : mesh.faces[fi].points) {
            const float3 own = p.nor;
            float sx = 0.f, sy = 0.f, sz = 0.f;
            for (const float3& n : at[key(p.pos)])
                if (n.x * own.x + n.y * own.y + n.z * own.z >= cos_limit) { sx += n.x; sy += n.y; sz += n.z; }
            const float l = sqrtf(sx * sx + sy * sy + sz * sz);
            if (l > 0.f) p.nor = {sx / l, sy / l, sz / l};
        }
}

size_t ModelLoader::triangleCount() const {
    size_t n = 0;
    for (auto& m : scene_.meshes) n += m.faces.size();
    return n;
}

bool ModelLoader::load_obj(const std::string& path) {
    std::ifstream f(path);
    std::vector<float3> pos, nor;
    Mesh mesh;
    std::vector<size_t> needs_normals;          // faces whose corners came without a vn record
    std::string line;
    struct Corner { long v, n; };
    // A file with several objects / groups / materials is several MESHES: assimp's OBJ importer opens a new mesh at every `o`, `g` (it maps
    // groups onto objects) and `usemtl` that is followed by faces, and the reference walks them all in that order
    // (src/Models/model_loader.cpp:58-74, updateSceneData) -- the soup is their concatenation, in file order.  What the split changes is
    // GenSmoothNormals, which works mesh by mesh: corners of two groups that share a position do not share a normal.
    auto flush = [&]() {
        if (mesh.faces.empty()) return;
        smooth_missing_normals(mesh, needs_normals);       // faces without vn records, within this mesh only
        scene_.meshes.push_back(std::move(mesh));
        mesh = Mesh();
        needs_normals.clear();
    };
    while (std::getline(f, line)) {
        const char* s = line.c_str();
        while (*s == ' ' || *s == '\t') ++s;
        if (((s[0] == 'o' || s[0] == 'g') && (s[1] == ' ' || s[1] == '\t' || s[1] == '\r' || s[1] == 0)) || std::strncmp(s, "usemtl", 6) == 0) {
            flush();
        } else if (s[0] == 'v' && (s[1] == ' ' || s[1] == '\t')) {
            float3 p; char* e;
            p.x = strtof(s + 2, &e); p.y = strtof(e, &e); p.z = strtof(e, &e);
            // strtof accepts "nan" / "inf": a non-finite position would make the BVH builder's ordering undefined
            if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) { err_ = "non-finite vertex position in '" + path + "'"; return false; }
            pos.push_back(p);
        } else if (s[0] == 'v' && s[1] == 'n' && (s[2] == ' ' || s[2] == '\t')) {
            float3 p; char* e;
            p.x = strtof(s + 3, &e); p.y = strtof(e, &e); p.z = strtof(e, &e);
            nor.push_back(p);
        } else if (s[0] == 'f' && (s[1] == ' ' || s[1] == '\t')) {
            std::vector<Corner> cs;
            const char* q = s + 2;
            while (*q) {
                while (*q == ' ' || *q == '\t' || *q == '\r') ++q;
                if (!*q) break;
                char* e;
                Corner c{0, 0};
                c.v = strtol(q, &e, 10);
                if (e == q) break;
                q = e;
                if (*q == '/') {
                    ++q;
                    if (*q != '/') { strtol(q, &e, 10); q = e; }       // vt, ignored
                    if (*q == '/') { ++q; c.n = strtol(q, &e, 10); q = e; }
                }
                cs.push_back(c);
            }
            auto fetch = [&](const Corner& c, Vertex& out) -> bool {
                long vi = c.v > 0 ? c.v - 1 : (long)pos.size() + c.v;
                if (vi < 0 || vi >= (long)pos.size()) return false;
                out.pos = pos[(size_t)vi];
                if (c.n != 0) {
                    long ni = c.n > 0 ? c.n - 1 : (long)nor.size() + c.n;
                    if (ni < 0 || ni >= (long)nor.size()) return false;
                    out.nor = nor[(size_t)ni];
                }
                return true;
            };
            for (size_t k = 1; k + 1 < cs.size(); ++k) {              // fan triangulation
                Face face;
                if (!fetch(cs[0], face.points[0]) || !fetch(cs[k], face.points[1]) || !fetch(cs[k + 1], face.points[2])) {
                    err_ = "face index out of range in '" + path + "'";
                    return false;
                }
                if (cs[0].n == 0 || cs[k].n == 0 || cs[k + 1].n == 0) {
                    // no vn: unit geometric normal for now, smoothed over the faces that share a position below
                    const float3 n = unit_face_normal(face.points[0].pos, face.points[1].pos, face.points[2].pos);
                    for (auto& p : face.points) p.nor = n;
                    needs_normals.push_back(mesh.faces.size());
                }
                mesh.faces.push_back(face);
            }
        }
    }
    flush();
    if (scene_.meshes.empty()) { err_ = "no faces in '" + path + "'"; return false; }
    return true;
}

// aiProcess_JoinIdenticalVertices (part of the preset of src/Models/model_loader.cpp:38): vertices of a mesh that agree in every
// attribute become one, the faces index them.  On the render path nothing can see it -- the renderer de-indexes every mesh again
// (src/main.cpp:93-119) -- so the soup above is what ships; this is the indexed view for callers that want assimp's (exporters, a
// builder that shares vertices).  Identity is bit identity of (position, normal) with -0 == +0, first occurrence first: de-indexing
// gives the soup back float for float.
void ModelLoader::weld(size_t mesh_index, std::vector<Vertex>& vertices, std::vector<uint32_t>& indices) const {
    vertices.clear(); indices.clear();
    if (mesh_index >= scene_.meshes.size()) return;
    struct Key { uint32_t w[6]; bool operator<(const Key& o) const { return std::memcmp(w, o.w, sizeof(w)) < 0; } };
    std::map<Key, uint32_t> seen;
    for (const Face& face : scene_.meshes[mesh_index].faces)
        for (const Vertex& v : face.points) {
            const float c[6] = {v.pos.x + 0.0f, v.pos.y + 0.0f, v.pos.z + 0.0f, v.nor.x + 0.0f, v.nor.y + 0.0f, v.nor.z + 0.0f};
            Key k;
            std::memcpy(k.w, c, sizeof(c));
            auto it = seen.find(k);
            if (it == seen.end()) { it = seen.emplace(k, (uint32_t)vertices.size()).first; vertices.push_back(v); }
            indices.push_back(it->second);
        }
}

// ---- Stanford PLY (ascii 1.0 / binary_little_endian 1.0): element vertex {x y z [nx ny nz] ...}, element face {list <count> <index>
// vertex_indices | vertex_index ...}; other elements and properties are skipped; polygons fan-triangulated; without normals: as an
// OBJ without vn.  One of the formats the reference reaches through assimp (src/Models/model_loader.cpp:38).
namespace {
struct PlyProp { std::string name; int type = 0, count_type = -1; };      // type: index into PLY_SIZES; count_type >= 0: a list
const char* const PLY_TYPES[] = {"char", "int8", "uchar", "uint8", "short", "int16", "ushort", "uint16", "int", "int32", "uint", "uint32", "float", "float32", "double", "float64"};
const int PLY_SIZES[] = {1, 1, 1, 1, 2, 2, 2, 2, 4, 4, 4, 4, 4, 4, 8, 8};
int ply_type(const std::string& t) { for (int k = 0; k < 16; ++k) if (t == PLY_TYPES[k]) return k; return -1; }
struct PlyElem { std::string name; size_t count = 0; std::vector<PlyProp> props; };
// one scalar of PLY type `t`, from text or from little-endian bytes; false at the end of the data
bool ply_scalar(std::istream& f, bool ascii, int t, double& out) {
    if (ascii) { return (bool)(f >> out); }
    unsigned char b[8];
    f.read(reinterpret_cast<char*>(b), PLY_SIZES[t]);
    if (!f) return false;
    switch (t / 2) {
        case 0: out = (double)(int8_t)b[0]; break;
        case 1: out = (double)b[0]; break;
        case 2: { int16_t v; std::memcpy(&v, b, 2); out = v; break; }
        case 3: { uint16_t v; std::memcpy(&v, b, 2); out = v; break; }
        case 4: { int32_t v; std::memcpy(&v, b, 4); out = v; break; }
        case 5: { uint32_t v; std::memcpy(&v, b, 4); out = v; break; }
        case 6: { float v; std::memcpy(&v, b, 4); out = v; break; }
        default: { double v; std::memcpy(&v, b, 8); out = v; break; }
    }
    return true;
}
}  // namespace

bool ModelLoader::load_ply(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    std::string line;
    if (!std::getline(f, line) || line.substr(0, 3) != "ply") { err_ = "not a PLY file: '" + path + "'"; return false; }
    bool ascii = false, have_format = false;
    std::vector<PlyElem> elems;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::istringstream ls(line);
        std::string w;
        ls >> w;
        if (w == "format") {
            std::string fmt; ls >> fmt;
            if (fmt == "ascii") ascii = true;
            else if (fmt != "binary_little_endian") { err_ = "unsupported PLY format '" + fmt + "' in '" + path + "'"; return false; }
            have_format = true;
        } else if (w == "element") {
            PlyElem e; ls >> e.name >> e.count;
            elems.push_back(e);
        } else if (w == "property") {
            if (elems.empty()) { err_ = "PLY property before any element in '" + path + "'"; return false; }
            PlyProp p; std::string t; ls >> t;
            if (t == "list") { std::string ct, it; ls >> ct >> it >> p.name; p.count_type = ply_type(ct); p.type = ply_type(it); if (p.count_type < 0) p.type = -1; }
            else { p.type = ply_type(t); ls >> p.name; }
            if (p.type < 0) { err_ = "unknown PLY property type in '" + path + "'"; return false; }
            elems.back().props.push_back(p);
        } else if (w == "end_header") break;
    }
    if (!have_format) { err_ = "PLY header without a format line in '" + path + "'"; return false; }
    std::vector<float3> pos, nor;
    bool have_normals = false;
    Mesh mesh;
    std::vector<size_t> needs_normals;
    for (const PlyElem& e : elems) {
        if (e.count > (size_t)1 << 31) { err_ = "unreasonable element count in '" + path + "'"; return false; }
        int ix = -1, iy = -1, iz = -1, inx = -1, iny = -1, inz = -1;
        if (e.name == "vertex") {
            for (size_t k = 0; k < e.props.size(); ++k) {
                const std::string& n = e.props[k].name;
                if (n == "x") ix = (int)k; else if (n == "y") iy = (int)k; else if (n == "z") iz = (int)k;
                else if (n == "nx") inx = (int)k; else if (n == "ny") iny = (int)k; else if (n == "nz") inz = (int)k;
            }
            if (ix < 0 || iy < 0 || iz < 0) { err_ = "PLY vertex element without x / y / z in '" + path + "'"; return false; }
            have_normals = inx >= 0 && iny >= 0 && inz >= 0;
        }
        for (size_t r = 0; r < e.count; ++r) {
            float3 p, n;
            std::vector<long> idx;
            for (size_t k = 0; k < e.props.size(); ++k) {
                const PlyProp& pr = e.props[k];
                double v = 0.0;
                if (pr.count_type >= 0) {
                    if (!ply_scalar(f, ascii, pr.count_type, v) || v < 0 || v > 1e6) { err_ = "truncated or corrupt PLY data in '" + path + "'"; return false; }
                    const long cnt = (long)v;
                    const bool is_index = e.name == "face" && (pr.name == "vertex_indices" || pr.name == "vertex_index");
                    for (long c = 0; c < cnt; ++c) {
                        if (!ply_scalar(f, ascii, pr.type, v)) { err_ = "truncated PLY data in '" + path + "'"; return false; }
                        if (is_index) idx.push_back((long)v);
                    }
                    continue;
                }
                if (!ply_scalar(f, ascii, pr.type, v)) { err_ = "truncated PLY data in '" + path + "'"; return false; }
                const int kk = (int)k;
                if (kk == ix) p.x = (float)v; else if (kk == iy) p.y = (float)v; else if (kk == iz) p.z = (float)v;
                else if (kk == inx) n.x = (float)v; else if (kk == iny) n.y = (float)v; else if (kk == inz) n.z = (float)v;
            }
            if (e.name == "vertex") {
                if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) { err_ = "non-finite vertex position in '" + path + "'"; return false; }
                pos.push_back(p); nor.push_back(n);
            } else if (e.name == "face") {
                for (size_t k = 1; k + 1 < idx.size(); ++k) {
                    Face face;
                    const long tri[3] = {idx[0], idx[k], idx[k + 1]};
                    for (int c = 0; c < 3; ++c) {
                        if (tri[c] < 0 || tri[c] >= (long)pos.size()) { err_ = "face index out of range in '" + path + "'"; return false; }
                        face.points[c].pos = pos[(size_t)tri[c]];
                        face.points[c].nor = nor[(size_t)tri[c]];
                    }
                    if (!have_normals) {
                        const float3 gn = unit_face_normal(face.points[0].pos, face.points[1].pos, face.points[2].pos);
                        for (auto& q : face.points) q.nor = gn;
                        needs_normals.push_back(mesh.faces.size());
                    }
                    mesh.faces.push_back(face);
                }
            }
        }
    }
    if (mesh.faces.empty()) { err_ = "no faces in '" + path + "'"; return false; }
    smooth_missing_normals(mesh, needs_normals);
    scene_.meshes.push_back(std::move(mesh));
    return true;
}

// ---- STL: binary (80-byte header, uint32 count, 50 bytes per facet) or ASCII ("facet normal .. outer loop vertex ..").  A facet's
// own normal is used for its three corners when it is finite and non-zero (flat shading, what an importer hands on for this format),
// the unit geometric normal otherwise.
bool ModelLoader::load_stl(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    f.seekg(0, std::ios::end);
    const std::streamoff size = f.tellg();
    f.seekg(0);
    Mesh mesh;
    auto add = [&](const float* n, const float* v) -> bool {
        Face face;
        for (int c = 0; c < 3; ++c) {
            if (!std::isfinite(v[3 * c]) || !std::isfinite(v[3 * c + 1]) || !std::isfinite(v[3 * c + 2])) return false;
            face.points[c].pos = {v[3 * c], v[3 * c + 1], v[3 * c + 2]};
        }
        float3 fn{n[0], n[1], n[2]};
        const float l = sqrtf(fn.x * fn.x + fn.y * fn.y + fn.z * fn.z);
        if (!(l > 0.f) || !std::isfinite(l)) fn = unit_face_normal(face.points[0].pos, face.points[1].pos, face.points[2].pos);
        for (auto& p : face.points) p.nor = fn;
        mesh.faces.push_back(face);
        return true;
    };
    uint32_t count = 0;
    if (size >= 84) {
        char header[80];
        f.read(header, 80);
        f.read(reinterpret_cast<char*>(&count), 4);
    }
    if (size >= 84 && (uint64_t)count * 50u + 84u == (uint64_t)size) {           // binary: the size says so (an ASCII file may start with "solid" too)
        std::vector<unsigned char> buf((size_t)count * 50);
        f.read(reinterpret_cast<char*>(buf.data()), (std::streamsize)buf.size());
        if (!f) { err_ = "truncated STL '" + path + "'"; return false; }
        for (uint32_t i = 0; i < count; ++i) {
            float v[12];
            std::memcpy(v, &buf[(size_t)i * 50], 48);
            if (!add(v, v + 3)) { err_ = "non-finite vertex position in '" + path + "'"; return false; }
        }
    } else {
        f.clear();
        f.seekg(0);
        std::string w;
        float n[3] = {0, 0, 0}, v[9];
        int nv = 0;
        while (f >> w) {
            if (w == "facet") { std::string kw; f >> kw >> n[0] >> n[1] >> n[2]; nv = 0; if (!f) break; }
            else if (w == "vertex") {
                if (nv >= 3) { err_ = "STL facet with more than three vertices in '" + path + "'"; return false; }
                f >> v[3 * nv] >> v[3 * nv + 1] >> v[3 * nv + 2];
                if (!f) break;
                ++nv;
            } else if (w == "endfacet") {
                if (nv != 3) { err_ = "STL facet without three vertices in '" + path + "'"; return false; }
                if (!add(n, v)) { err_ = "non-finite vertex position in '" + path + "'"; return false; }
                nv = 0;
            }
        }
    }
    if (mesh.faces.empty()) { err_ = "no facets in '" + path + "'"; return false; }
    scene_.meshes.push_back(std::move(mesh));
    return true;
}

bool ModelLoader::load_soup(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { err_ = "cannot open model '" + path + "'"; return false; }
    char magic[8];
    uint32_t n = 0;
    f.read(magic, 8);
    f.read(reinterpret_cast<char*>(&n), 4);
    if (!f || std::memcmp(magic, "PRTMESH1", 8) != 0) { err_ = "bad soup header in '" + path + "'"; return false; }
    {   // the header's triangle count against the file itself, BEFORE it sizes any allocation (a damaged count asks for ~300 GB)
        const std::streampos here = f.tellg();
        f.seekg(0, std::ios::end);
        const std::streamoff size = f.tellg();
        f.seekg(here);
        if (size < 12 || (uint64_t)n * 72u + 12u > (uint64_t)size) { err_ = "truncated soup '" + path + "' (header announces more triangles than the file holds)"; return false; }
    }
    Mesh mesh;
    mesh.faces.resize(n);
    std::vector<float> buf((size_t)n * 18);
    f.read(reinterpret_cast<char*>(buf.data()), (std::streamsize)(buf.size() * 4));
    if (!f) { err_ = "truncated soup '" + path + "'"; return false; }
    for (uint32_t i = 0; i < n; ++i)
        for (int c = 0; c < 3; ++c) {
            const float* p = &buf[(size_t)i * 18 + c * 6];
            if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) { err_ = "non-finite vertex position in '" + path + "'"; return false; }
            mesh.faces[i].points[c].pos = {p[0], p[1], p[2]};
            mesh.faces[i].points[c].nor = {p[3], p[4], p[5]};
        }
    scene_.meshes.push_back(std::move(mesh));
    return true;
}

bool ModelLoader::SaveSoup(const std::string& path) const {
    std::ofstream f(path, std::ios::binary);
    if (!f) return false;
    uint32_t n = (uint32_t)triangleCount();
    f.write("PRTMESH1", 8);
    f.write(reinterpret_cast<const char*>(&n), 4);
    for (auto& m : scene_.meshes)
        for (auto& face : m.faces)
            for (auto& p : face.points) {
                float v[6] = {p.pos.x, p.pos.y, p.pos.z, p.nor.x, p.nor.y, p.nor.z};
                f.write(reinterpret_cast<const char*>(v), sizeof(v));
            }
    return (bool)f;
}

void ModelLoader::flatten(std::vector<float>& v4, std::vector<float>& n4) const {
    v4.clear(); n4.clear();
    for (auto& m : scene_.meshes)
        for (auto& face : m.faces)
            for (auto& p : face.points) {
                v4.insert(v4.end(), {p.pos.x, p.pos.y, p.pos.z, 0.f});
                n4.insert(n4.end(), {p.nor.x, p.nor.y, p.nor.z, 0.f});
            }
}

}  // namespace IO
}  // namespace prt
