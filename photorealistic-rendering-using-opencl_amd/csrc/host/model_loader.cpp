#include "model_loader.h"

#include <cmath>
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
    return load_obj(path);
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
    while (std::getline(f, line)) {
        const char* s = line.c_str();
        while (*s == ' ' || *s == '\t') ++s;
        if (s[0] == 'v' && (s[1] == ' ' || s[1] == '\t')) {
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
                    const float3 &a = face.points[0].pos, &b = face.points[1].pos, &c = face.points[2].pos;
                    float ux = b.x - a.x, uy = b.y - a.y, uz = b.z - a.z, vx = c.x - a.x, vy = c.y - a.y, vz = c.z - a.z;
                    float3 n{uy * vz - uz * vy, uz * vx - ux * vz, ux * vy - uy * vx};
                    float l = sqrtf(n.x * n.x + n.y * n.y + n.z * n.z);
                    if (l > 0.f) { n.x /= l; n.y /= l; n.z /= l; }
                    for (auto& p : face.points) p.nor = n;
                    needs_normals.push_back(mesh.faces.size());
                }
                mesh.faces.push_back(face);
            }
        }
    }
    if (mesh.faces.empty()) { err_ = "no faces in '" + path + "'"; return false; }
    // Files without vn records: the reference imports with aiProcessPreset_TargetRealtime_Quality
    // (src/Models/model_loader.cpp:38), whose GenSmoothNormals step gives every vertex the normalised sum of the
    // unit normals of the faces that meet at its position (smoothing-angle limit 175 degrees).  Same rule here, on
    // bit-identical positions (assimp is an absent submodule, so this step has no oracle: SURVEY s8f row N2).
    if (!needs_normals.empty()) {
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
