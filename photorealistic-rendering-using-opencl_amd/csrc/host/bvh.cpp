#include "bvh.h"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <mutex>
#include <numeric>
#include <system_error>
#include <thread>

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
    if (traversal_cost_ <= 0.0f) traversal_cost_ = 1.0f;
    build(tb, ce);
}

// Full-sweep SAH, top-down.  The tree is a pure function of the input (split decisions depend only on a range's
// primitives and their three presorted orders, kept sorted by stable partitions), so the ranges are built by a pool of
// threads -- big ranges are handed to the shared queue, small ones finished locally -- into an arena in whatever order
// the threads get there, and a final sequential pass renumbers the nodes into the canonical layout: node 0 = root, the
// two children of an inner node adjacent, a left subtree's nodes before its sibling's (the layout of the sequential
// builder this replaces, node for node).  871 k triangles: 6.6 s on one thread, see profiles/README.md for the pool.
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
    const auto t_start = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (std::getenv("PRT_BVH_TIMING")) std::fprintf(stderr, "BVH %s: %.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count());
    };
    // the machine's threads, shared with the other ranks of a multi-process launch (torchrun exports LOCAL_WORLD_SIZE: eight ranks that
    // each build the same 871 k-triangle tree must not start 8 x 64 threads); PRT_BVH_THREADS overrides
    unsigned n_threads = std::thread::hardware_concurrency();
    if (const char* e = std::getenv("LOCAL_WORLD_SIZE")) { const int k = std::atoi(e); if (k > 1) n_threads = (n_threads + (unsigned)k - 1u) / (unsigned)k; }
    if (const char* e = std::getenv("PRT_BVH_THREADS")) n_threads = (unsigned)std::atoi(e);
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 64) n_threads = 64;
    if (n < 16384) n_threads = 1;
    // three orders of the primitive ids, each sorted by the centre on one axis
    std::vector<uint32_t> order[3];
    {
        auto sort_axis = [&](int a) {
            order[a].resize(n);
            std::iota(order[a].begin(), order[a].end(), 0u);
            std::stable_sort(order[a].begin(), order[a].end(), [&](uint32_t x, uint32_t y) { return ce[3 * x + a] < ce[3 * y + a]; });
        };
        // (a thread that cannot be started -- a process / thread limit on the box -- is work done here instead: the tree does not
        // depend on who builds what)
        auto spawn = [](std::thread& th, auto&& fn, int a) { try { th = std::thread(fn, a); return true; } catch (const std::system_error&) { return false; } };
        if (n_threads > 1) {
            std::thread t1, t2;
            const bool s1 = spawn(t1, sort_axis, 1), s2 = spawn(t2, sort_axis, 2);
            sort_axis(0);
            if (!s1) sort_axis(1);
            if (!s2) sort_axis(2);
            if (s1) t1.join();
            if (s2) t2.join();
        } else {
            for (int a = 0; a < 3; ++a) sort_axis(a);
        }
    }
    lap("sorted");
    // per-position / per-primitive work arrays: a task only touches the positions [begin, end) of its range and the
    // primitives in it, and ranges of concurrent tasks are disjoint
    std::vector<float> right_cost[3] = {std::vector<float>(n), std::vector<float>(n), std::vector<float>(n)};
    std::vector<uint8_t> goes_left(n);
    std::vector<uint32_t> scratch(n);

    auto set_bounds = [&](cl_BVHnode& nd, const Box& b) {
        for (int a = 0; a < 3; ++a) { nd.bounds[2 * a] = b.lo[a]; nd.bounds[2 * a + 1] = b.hi[a]; }
    };
    struct Task { uint32_t node, begin, end, depth; };
    std::vector<cl_BVHnode> arena(2 * n + 1);          // a binary tree over n leaves has at most 2n - 1 nodes
    std::atomic<uint32_t> arena_next{1};
    std::atomic<unsigned> depth_max{0};
    {
        Box b;
        for (size_t i = 0; i < n; ++i) b.grow(&tb[6 * i]);
        set_bounds(arena[0], b);
    }
    std::mutex mu;
    std::condition_variable cv;
    std::vector<Task> shared;
    unsigned busy = 0;                                   // tasks taken from `shared` and not finished yet
    shared.push_back({0, 0, (uint32_t)n, 0});
    const uint32_t share_above = n_threads > 1 ? 4096u : 0xFFFFFFFFu;

    auto run_range = [&](const Task& first_task) {
        std::vector<Task> stack{first_task};
        while (!stack.empty()) {
            const Task t = stack.back();
            stack.pop_back();
            for (unsigned d = depth_max.load(); d < t.depth && !depth_max.compare_exchange_weak(d, t.depth);) {}
            const uint32_t count = t.end - t.begin;
            Box nb;
            for (int a = 0; a < 3; ++a) { nb.lo[a] = arena[t.node].bounds[2 * a]; nb.hi[a] = arena[t.node].bounds[2 * a + 1]; }

            auto make_leaf = [&]() {
                cl_BVHnode& nd = arena[t.node];
                nd.is_leaf = 1;
                nd.first_child_or_primitive = t.begin;
                nd.primitive_count = count;
            };
            if (count <= 1) { make_leaf(); continue; }

            // full sweep on every axis (the three axes of a big range on three threads: the top of the tree is a chain of
            // big ranges that no amount of task sharing shortens)
            float axis_cost[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
            uint32_t axis_split[3] = {0, 0, 0};
            auto sweep = [&](int a) {
                const uint32_t* o = order[a].data();
                float* rc = right_cost[a].data();
                Box rb;
                for (uint32_t i = t.end - 1; i > t.begin; --i) {
                    rb.grow(&tb[6 * o[i]]);
                    rc[i] = rb.half_area() * (float)(t.end - i);
                }
                Box lb;
                for (uint32_t i = t.begin; i + 1 < t.end; ++i) {
                    lb.grow(&tb[6 * o[i]]);
                    float c = lb.half_area() * (float)(i + 1 - t.begin) + rc[i + 1];
                    if (c < axis_cost[a]) { axis_cost[a] = c; axis_split[a] = i + 1; }
                }
            };
            if (n_threads > 1 && count > 65536u) {
                auto spawn = [](std::thread& th, auto&& fn, int a) { try { th = std::thread(fn, a); return true; } catch (const std::system_error&) { return false; } };
                std::thread t1, t2;
                const bool s1 = spawn(t1, sweep, 1), s2 = spawn(t2, sweep, 2);
                sweep(0);
                if (!s1) sweep(1);
                if (!s2) sweep(2);
                if (s1) t1.join();
                if (s2) t2.join();
            } else {
                for (int a = 0; a < 3; ++a) sweep(a);
            }
            float best_cost = FLT_MAX;
            int best_axis = -1;
            uint32_t best_split = 0;
            for (int a = 0; a < 3; ++a)                      // first axis wins ties, as in a sequential sweep over a = 0, 1, 2
                if (axis_cost[a] < best_cost) { best_cost = axis_cost[a]; best_axis = a; best_split = axis_split[a]; }
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
                uint32_t* sc = scratch.data() + t.begin;
                for (uint32_t i = t.begin; i < t.end; ++i) {
                    if (goes_left[o[i]]) o[l++] = o[i]; else sc[r++] = o[i];
                }
                std::copy(sc, sc + r, o + l);
            }
            Box lb, rb;
            for (uint32_t i = t.begin; i < best_split; ++i) lb.grow(&tb[6 * order[best_axis][i]]);
            for (uint32_t i = best_split; i < t.end; ++i) rb.grow(&tb[6 * order[best_axis][i]]);
            const uint32_t first = arena_next.fetch_add(2);
            arena[first] = cl_BVHnode{};
            arena[first + 1] = cl_BVHnode{};
            set_bounds(arena[first], lb);
            set_bounds(arena[first + 1], rb);
            cl_BVHnode& nd = arena[t.node];
            nd.is_leaf = 0;
            nd.first_child_or_primitive = first;
            nd.primitive_count = 0;
            const Task right{first + 1, best_split, t.end, t.depth + 1}, left{first, t.begin, best_split, t.depth + 1};
            if (right.end - right.begin > share_above) {           // hand the bigger pieces to whoever is idle
                { std::lock_guard<std::mutex> lk(mu); shared.push_back(right); }
                cv.notify_one();
            } else {
                stack.push_back(right);
            }
            stack.push_back(left);                                 // (numbering does not depend on who builds what: see the renumbering pass)
        }
    };
    auto worker = [&]() {
        for (;;) {
            Task t;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !shared.empty() || busy == 0; });
                if (shared.empty()) return;                          // nothing queued and nobody who could queue more
                t = shared.back();
                shared.pop_back();
                ++busy;
            }
            run_range(t);
            {
                std::lock_guard<std::mutex> lk(mu);
                --busy;
            }
            cv.notify_all();
        }
    };
    if (n_threads > 1) {
        std::vector<std::thread> pool;
        for (unsigned k = 1; k < n_threads; ++k) {
            try { pool.emplace_back(worker); } catch (const std::system_error&) { break; }      // fewer workers, the same tree
        }
        worker();
        for (auto& th : pool) th.join();
    } else {
        worker();
    }
    max_depth_ = depth_max.load();
    lap("split");
    // canonical numbering: exactly the order in which the one-thread builder allocates its nodes
    nodes_.reserve(arena_next.load());
    nodes_.push_back(arena[0]);
    struct Ren { uint32_t new_index, old_index; };
    std::vector<Ren> st{{0u, 0u}};
    while (!st.empty()) {
        const Ren r = st.back();
        st.pop_back();
        const cl_BVHnode& src = arena[r.old_index];
        if (src.is_leaf) continue;
        const uint32_t first_old = src.first_child_or_primitive, first_new = (uint32_t)nodes_.size();
        nodes_[r.new_index].first_child_or_primitive = first_new;
        nodes_.push_back(arena[first_old]);
        nodes_.push_back(arena[first_old + 1]);
        st.push_back({first_new + 1, first_old + 1});
        st.push_back({first_new, first_old});
    }
    lap("renumbered");
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
