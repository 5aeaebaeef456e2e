#include "bvh.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <numeric>

namespace prt {

namespace {
struct Box {
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    void grow(const float* b6) {   // b6 = minx maxx miny maxy minz maxz
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], b6[2 * a]); hi[a] = std::max(hi[a], b6[2 * a + 1]); }
    }
    float half_area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return (dx + dy) * dz + dx * dy;
    }
    int largest_axis() const {
        float d[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
        return d[0] > d[1] ? (d[0] > d[2] ? 0 : 2) : (d[1] > d[2] ? 1 : 2);
    }
};
}  // namespace

BVH::BVH(const std::shared_ptr<IO::ModelLoader>& ml, unsigned max_leaf_size, float traversal_cost)
    : max_leaf_size_(max_leaf_size), traversal_cost_(traversal_cost) {
    std::vector<float> tb, ce;
    for (auto& mesh : ml->getFaces().meshes)
        for (auto& f : mesh.faces) {
            float b[6] = {FLT_MAX, -FLT_MAX, FLT_MAX, -FLT_MAX, FLT_MAX, -FLT_MAX};
            for (auto& p : f.points) {
                const float c[3] = {p.pos.x, p.pos.y, p.pos.z};
                for (int a = 0; a < 3; ++a) { b[2 * a] = std::min(b[2 * a], c[a]); b[2 * a + 1] = std::max(b[2 * a + 1], c[a]); }
            }
            tb.insert(tb.end(), b, b + 6);
            for (int a = 0; a < 3; ++a) ce.push_back((b[2 * a] + b[2 * a + 1]) * 0.5f);
        }
    if (traversal_cost_ <= 0.0f) traversal_cost_ = (tb.size() / 6 >= 65536) ? 1.5f : 1.0f;
    build(tb, ce);
}

void BVH::build(const std::vector<float>& tb, const std::vector<float>& ce) {
    const size_t n = tb.size() / 6;
    nodes_.clear();
    prim_indices_.clear();
    if (n == 0) {                      // "no OBJ" == an empty leaf root
        cl_BVHnode root{};
        root.is_leaf = 1;
        nodes_.push_back(root);
        return;
    }
    // three orders of the primitive ids, each sorted by the centre on one axis
    std::vector<uint32_t> order[3];
    for (int a = 0; a < 3; ++a) {
        order[a].resize(n);
        std::iota(order[a].begin(), order[a].end(), 0u);
        std::stable_sort(order[a].begin(), order[a].end(), [&](uint32_t x, uint32_t y) { return ce[3 * x + a] < ce[3 * y + a]; });
    }
    std::vector<float> right_cost(n);
    std::vector<uint8_t> goes_left(n);
    std::vector<uint32_t> scratch(n);

    auto set_bounds = [&](cl_BVHnode& nd, const Box& b) {
        for (int a = 0; a < 3; ++a) { nd.bounds[2 * a] = b.lo[a]; nd.bounds[2 * a + 1] = b.hi[a]; }
    };
    struct Task { uint32_t node, begin, end, depth; };
    std::vector<Task> stack;
    nodes_.push_back(cl_BVHnode{});
    {
        Box b;
        for (size_t i = 0; i < n; ++i) b.grow(&tb[6 * i]);
        set_bounds(nodes_[0], b);
    }
    stack.push_back({0, 0, (uint32_t)n, 0});
    while (!stack.empty()) {
        Task t = stack.back();
        stack.pop_back();
        max_depth_ = std::max(max_depth_, t.depth);
        const uint32_t count = t.end - t.begin;
        Box nb;
        for (int a = 0; a < 3; ++a) { nb.lo[a] = nodes_[t.node].bounds[2 * a]; nb.hi[a] = nodes_[t.node].bounds[2 * a + 1]; }

        auto make_leaf = [&]() {
            cl_BVHnode& nd = nodes_[t.node];
            nd.is_leaf = 1;
            nd.first_child_or_primitive = t.begin;
            nd.primitive_count = count;
        };
        if (count <= 1) { make_leaf(); continue; }

        // full sweep on every axis
        float best_cost = FLT_MAX;
        int best_axis = -1;
        uint32_t best_split = 0;
        for (int a = 0; a < 3; ++a) {
            const uint32_t* o = order[a].data();
            Box rb;
            for (uint32_t i = t.end - 1; i > t.begin; --i) {
                rb.grow(&tb[6 * o[i]]);
                right_cost[i] = rb.half_area() * (float)(t.end - i);
            }
            Box lb;
            for (uint32_t i = t.begin; i + 1 < t.end; ++i) {
                lb.grow(&tb[6 * o[i]]);
                float c = lb.half_area() * (float)(i + 1 - t.begin) + right_cost[i + 1];
                if (c < best_cost) { best_cost = c; best_axis = a; best_split = i + 1; }
            }
        }
        const float leaf_limit = nb.half_area() * ((float)count - traversal_cost_);
        if (best_axis < 0 || best_cost >= leaf_limit) {
            if (count <= max_leaf_size_) { make_leaf(); continue; }
            best_axis = nb.largest_axis();              // too big for a leaf: median split
            best_split = t.begin + count / 2;
        }
        // partition the other two orders stably so they stay sorted inside each child
        for (uint32_t i = t.begin; i < t.end; ++i) goes_left[order[best_axis][i]] = (i < best_split);
        for (int a = 0; a < 3; ++a) {
            if (a == best_axis) continue;
            uint32_t* o = order[a].data();
            uint32_t l = t.begin, r = 0;
            for (uint32_t i = t.begin; i < t.end; ++i) {
                if (goes_left[o[i]]) o[l++] = o[i]; else scratch[r++] = o[i];
            }
            std::copy(scratch.begin(), scratch.begin() + r, o + l);
        }
        Box lb, rb;
        for (uint32_t i = t.begin; i < best_split; ++i) lb.grow(&tb[6 * order[best_axis][i]]);
        for (uint32_t i = best_split; i < t.end; ++i) rb.grow(&tb[6 * order[best_axis][i]]);
        const uint32_t first = (uint32_t)nodes_.size();
        nodes_.push_back(cl_BVHnode{});
        nodes_.push_back(cl_BVHnode{});
        set_bounds(nodes_[first], lb);
        set_bounds(nodes_[first + 1], rb);
        cl_BVHnode& nd = nodes_[t.node];
        nd.is_leaf = 0;
        nd.first_child_or_primitive = first;
        nd.primitive_count = 0;
        stack.push_back({first + 1, best_split, t.end, t.depth + 1});
        stack.push_back({first, t.begin, best_split, t.depth + 1});
    }
    prim_indices_ = order[0];       // all three orders agree inside every leaf range as SETS; use axis 0's
    // leaves were cut out of ranges that are identical index ranges in all orders, so order[0]
    // restricted to a leaf range holds exactly that leaf's primitives.
}

std::unique_ptr<std::vector<uint64_t>> BVH::GetPrimitiveIndices() const {
    auto res = std::make_unique<std::vector<uint64_t>>();
    res->reserve(prim_indices_.size());
    for (uint32_t i : prim_indices_) res->push_back(i);
    return res;
}

std::unique_ptr<std::vector<cl_BVHnode>> BVH::PrepareData() const {
    return std::make_unique<std::vector<cl_BVHnode>>(nodes_);
}

}  // namespace prt
