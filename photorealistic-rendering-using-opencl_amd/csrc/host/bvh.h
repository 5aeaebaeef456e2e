// bvh.h -- SAH BVH over a triangle soup with the reference's BVH API
// (include/BVH/bvh.h:38-56: BVH(ml), PrepareData(), GetPrimitiveIndices()) and its flattened
// output layout (36-byte cl_BVHnode, node 0 = root, the two children of an inner node adjacent
// at first_child_or_primitive + {0,1}, leaves index a permutation array of primitive ids).
// The reference delegates the build to madmann91/bvh's SweepSahBuilder (src/BVH/bvh.cpp:58-63),
// an un-vendored dependency; this is an independent full-sweep SAH builder (three presorted
// axis orders, stable partition per split, O(n log n)).  Closest-hit results do not depend on
// the topology, so any valid tree is a drop-in.
#pragma once
#include <cstdint>
#include <memory>
#include <vector>

#include "model_loader.h"
#include "prt_types.h"

namespace prt {

using cl_BVHnode = prt_bvh_node;

class BVH {
public:
    // traversal_cost <= 0 picks the measured optimum, 1: fatter leaves are slower on MI355X at every mesh size since the walk
    // defers its leaves (teapot: monotonically; 871 k-triangle mesh, 4K frame, 512 spp: 4.12 / 4.10 / 4.10 / 3.98 / 3.87 G segments/s
    // at cost 0.5 / 0.7 / 1 / 1.5 / 2 -- the round-2 kernel, which tested leaves inside the box loop, preferred 1.5)
    explicit BVH(const std::shared_ptr<IO::ModelLoader>& ml, unsigned max_leaf_size = 16, float traversal_cost = 0.0f);
    std::unique_ptr<std::vector<uint64_t>> GetPrimitiveIndices() const;
    std::unique_ptr<std::vector<cl_BVHnode>> PrepareData() const;
    size_t node_count() const { return nodes_.size(); }
    unsigned max_depth() const { return max_depth_; }

private:
    void build(const std::vector<float>& tri_bounds /*6 per tri*/, const std::vector<float>& centers /*3 per tri*/);
    std::vector<cl_BVHnode> nodes_;
    std::vector<uint32_t> prim_indices_;
    unsigned max_leaf_size_;
    float traversal_cost_;
    unsigned max_depth_ = 0;
};

}  // namespace prt
