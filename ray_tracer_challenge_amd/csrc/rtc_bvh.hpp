// rtc_bvh.hpp -- host-side AABB BVH builder (binned SAH, median fallback).
//
// Replaces nothing in the reference's results: its Group.local_intersect tests every child
// regardless of bounds (shapes.py:454), so any conservative hierarchy over the leaves returns
// the same hit lists.  What it does replace is the role Group.divide (shapes.py:480-507) was
// written for.  Built once per scene on the host, over the world-space boxes the scene
// compiler supplies; output is the linear two-children-per-node layout rtc_device.hpp walks.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <limits>
#include <thread>
#include <vector>

#ifndef RTC_SAH_BINS
#define RTC_SAH_BINS 16
#endif

namespace rtc {

struct Aabb {
    double lo[3], hi[3];
    void reset() { for (int k = 0; k < 3; k++) { lo[k] = std::numeric_limits<double>::infinity(); hi[k] = -lo[k]; } }
    void grow(const Aabb& o) { for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], o.lo[k]); hi[k] = std::max(hi[k], o.hi[k]); } }
    double area() const {
        double d[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
        if (d[0] < 0 || d[1] < 0 || d[2] < 0) return 0;
        return 2 * (d[0] * d[1] + d[1] * d[2] + d[2] * d[0]);
    }
};

struct Bvh {
    std::vector<double> box;        // [n_nodes][12] child0 lo,hi, child1 lo,hi
    std::vector<int32_t> child;     // [n_nodes][2]
    std::vector<int32_t> order;     // item indices in leaf order
    int depth = 0, max_leaf = 0;
    int n_nodes() const { return (int)(child.size() / 2); }
};

class BvhBuilder {
public:
    static constexpr int32_t EMPTY = INT32_MIN;
    BvhBuilder(const std::vector<Aabb>& boxes, int max_leaf, bool median_only)
        : b_(boxes), max_leaf_(max_leaf), median_(median_only) {}

    Bvh build() {
        Bvh out;
        int n = (int)b_.size();
        idx_.resize(n);
        for (int i = 0; i < n; i++) idx_[i] = i;
        cen_.resize(3 * (size_t)n);
        for (int i = 0; i < n; i++)
            for (int k = 0; k < 3; k++) cen_[3 * (size_t)i + k] = 0.5 * (b_[i].lo[k] + b_[i].hi[k]);
        if (n == 0) return out;
        if (n <= max_leaf_) {   // a single leaf: node 0 with one child
            int node = new_node(out);
            Aabb bb = bounds(0, n);
            set_child(out, node, 0, leaf_code(0, n), bb);
            Aabb e; e.reset();
            set_child(out, node, 1, EMPTY, e);
            out.depth = 1;
            out.max_leaf = n;
        } else {
            // Large inputs: the top of the tree is built here; every range of at most `grain` items is set aside as a task;
            // the tasks are built by a few threads into trees of their own (they partition disjoint ranges of idx_) and
            // spliced in behind the top in the order they were set aside.  The tree is the one a single thread builds --
            // only the numbering of its nodes differs, and that depends on the input alone, not on the number of threads.
            const int threads = std::max(1, std::min<int>(getenv("RTC_BUILD_THREADS") ? atoi(getenv("RTC_BUILD_THREADS")) : 16, (int)std::thread::hardware_concurrency()));
            std::vector<Task> tasks;
            Target top = {&out, n >= 32768 ? &tasks : nullptr, std::max(4096, n / 64)};
            Aabb bb;
            split(0, n, 1, bb, top);
            if (!tasks.empty()) {
                std::vector<Bvh> subs(tasks.size());
                std::atomic<size_t> next(0);
                auto work = [&]() {
                    for (size_t t = next.fetch_add(1); t < tasks.size(); t = next.fetch_add(1)) {
                        Target mine = {&subs[t], nullptr, 0};
                        Aabb unused;
                        split(tasks[t].lo, tasks[t].hi, tasks[t].depth, unused, mine);
                    }
                };
                std::vector<std::thread> pool;
                for (int t = 1; t < threads && t < (int)tasks.size(); t++) pool.emplace_back(work);
                work();
                for (std::thread& th : pool) th.join();
                for (size_t t = 0; t < tasks.size(); t++) {
                    const int32_t offset = out.n_nodes();
                    out.box.insert(out.box.end(), subs[t].box.begin(), subs[t].box.end());
                    for (int32_t c : subs[t].child) out.child.push_back(c >= 0 ? c + offset : c);
                    out.child[2 * (size_t)tasks[t].node + tasks[t].side] = offset;
                    out.depth = std::max(out.depth, subs[t].depth);
                    out.max_leaf = std::max(out.max_leaf, subs[t].max_leaf);
                }
            }
        }
        out.order = idx_;
        return out;
    }

private:
    const std::vector<Aabb>& b_;
    int max_leaf_;
    bool median_;
    std::vector<int32_t> idx_;
    std::vector<double> cen_;
    struct Task { int lo, hi, depth, node, side; };
    struct Target { Bvh* out; std::vector<Task>* tasks; int grain; };      // where split() puts its nodes, and which ranges it sets aside instead

    static int32_t leaf_code(int first, int count) { return ~((first << 4) | (count - 1)); }
    static int new_node(Bvh& out) {
        out.box.resize(out.box.size() + 12);
        out.child.resize(out.child.size() + 2);
        return out.n_nodes() - 1;
    }
    static void set_child(Bvh& out, int node, int k, int32_t ref, const Aabb& bb) {
        double* p = out.box.data() + 12 * (size_t)node + 6 * k;
        for (int a = 0; a < 3; a++) { p[a] = bb.lo[a]; p[3 + a] = bb.hi[a]; }
        out.child[2 * (size_t)node + k] = ref;
    }
    Aabb bounds(int lo, int hi) const {
        Aabb bb; bb.reset();
        for (int i = lo; i < hi; i++) bb.grow(b_[idx_[i]]);
        return bb;
    }

    // builds the inner node for items [lo, hi) (hi - lo > max_leaf) into tg.out and returns its index there
    int split(int lo, int hi, int depth, Aabb& bb_out, Target& tg) {
        Bvh& out = *tg.out;
        int node = new_node(out);
        out.depth = std::max(out.depth, depth);
        int n = hi - lo;
        Aabb cb; cb.reset();
        for (int i = lo; i < hi; i++)
            for (int k = 0; k < 3; k++) {
                double c = cen_[3 * (size_t)idx_[i] + k];
                cb.lo[k] = std::min(cb.lo[k], c); cb.hi[k] = std::max(cb.hi[k], c);
            }
        int axis = 0;
        for (int k = 1; k < 3; k++) if (cb.hi[k] - cb.lo[k] > cb.hi[axis] - cb.lo[axis]) axis = k;
        int mid = -1;
        if (!median_ && cb.hi[axis] > cb.lo[axis]) {
            const int NB = RTC_SAH_BINS;
            int best_axis = -1, best_bin = -1;
            double best_cost = std::numeric_limits<double>::infinity();
            for (int ax = 0; ax < 3; ax++) {
                double ext = cb.hi[ax] - cb.lo[ax];
                if (!(ext > 0)) continue;
                Aabb bins[NB]; int cnt[NB];
                for (int q = 0; q < NB; q++) { bins[q].reset(); cnt[q] = 0; }
                double scale = NB / ext;
                for (int i = lo; i < hi; i++) {
                    int q = (int)((cen_[3 * (size_t)idx_[i] + ax] - cb.lo[ax]) * scale);
                    q = std::min(std::max(q, 0), NB - 1);
                    bins[q].grow(b_[idx_[i]]); cnt[q]++;
                }
                double right_area[NB]; int right_cnt[NB];
                Aabb acc; acc.reset(); int c = 0;
                for (int q = NB - 1; q > 0; q--) { acc.grow(bins[q]); c += cnt[q]; right_area[q] = acc.area(); right_cnt[q] = c; }
                acc.reset(); c = 0;
                for (int q = 0; q < NB - 1; q++) {
                    acc.grow(bins[q]); c += cnt[q];
                    if (c == 0 || right_cnt[q + 1] == 0) continue;
                    double cost = acc.area() * c + right_area[q + 1] * right_cnt[q + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = ax; best_bin = q; }
                }
            }
            if (best_axis >= 0) {
                double ext = cb.hi[best_axis] - cb.lo[best_axis], scale = NB / ext, base = cb.lo[best_axis];
                int ax = best_axis, bb = best_bin;
                auto it = std::partition(idx_.begin() + lo, idx_.begin() + hi, [&](int32_t i) {
                    int q = (int)((cen_[3 * (size_t)i + ax] - base) * scale);
                    q = std::min(std::max(q, 0), NB - 1);
                    return q <= bb;
                });
                mid = (int)(it - idx_.begin());
                if (mid == lo || mid == hi) mid = -1;
            }
        }
        if (mid < 0) {   // median split on the widest centroid axis (also the degenerate-SAH fallback)
            mid = lo + n / 2;
            int ax = axis;
            std::nth_element(idx_.begin() + lo, idx_.begin() + mid, idx_.begin() + hi, [&](int32_t a, int32_t b) {
                double ca = cen_[3 * (size_t)a + ax], cb2 = cen_[3 * (size_t)b + ax];
                return ca < cb2 || (ca == cb2 && a < b);
            });
        }
        int range[2][2] = {{lo, mid}, {mid, hi}};
        bb_out.reset();
        for (int k = 0; k < 2; k++) {
            int a = range[k][0], b = range[k][1];
            Aabb bb;
            int32_t ref;
            if (b - a <= max_leaf_) {
                bb = bounds(a, b);
                ref = leaf_code(a, b - a);
                out.max_leaf = std::max(out.max_leaf, b - a);
                out.depth = std::max(out.depth, depth);
            } else if (tg.tasks && b - a <= tg.grain) {
                bb = bounds(a, b);                     // = what the subtree's own boxes add up to (min / max are exact)
                ref = 0;                               // filled in when the task's tree is spliced in
                tg.tasks->push_back({a, b, depth + 1, node, k});
            } else {
                ref = split(a, b, depth + 1, bb, tg);
            }
            set_child(*tg.out, node, k, ref, bb);
            bb_out.grow(bb);
        }
        return node;
    }
};

inline Bvh build_bvh(const std::vector<Aabb>& boxes, int max_leaf, int max_depth) {
    Bvh b = BvhBuilder(boxes, max_leaf, false).build();
    if (b.depth > max_depth) b = BvhBuilder(boxes, max_leaf, true).build();
    return b;
}

// ---- four children per node ------------------------------------------------------------------------------------
// The binary tree collapsed two levels at a time: a wide node starts as a binary node's two children, and the child box
// with the largest area is replaced by its own two children until there are four (or only leaves are left).  A ray then
// meets about half as many nodes -- half as many DEPENDENT fetches, which is what the walk waits for -- at four box tests
// per node.  The traversal stack holds up to k - 1 pending children per level of a k-wide node, so a wide tree needs
// more stack per level than a binary one: `budget` is the number of entries the walk may have pending at once, and a
// node is only widened as far as the worst-case need of everything below it still fits (a binary node pushes one entry
// per level, so a binary tree of depth <= budget always fits; deep, thin parts of a tree stay binary, the rest goes
// four-wide).  need4(node) = (k - 1) + max over children need4(child) <= budget by construction.
struct Bvh4 {
    std::vector<double> box;        // [n_nodes][24]: child k lo xyz, hi xyz
    std::vector<int32_t> child;     // [n_nodes][4]: >= 0 inner node, < 0 leaf code (as in Bvh), EMPTY = none
    std::vector<int32_t> src;       // [n_nodes][4]: where the child's box lives in the binary tree, (binary node << 1) | side; -1 = none
    int stack_need = 0;             // worst-case pending entries of a walk
    int n_nodes() const { return (int)(child.size() / 4); }
};

class Bvh4Collapser {
public:
    static constexpr int32_t EMPTY = INT32_MIN;
    Bvh4Collapser(const Bvh& b, int budget) : b_(b), budget_(budget) {}
    Bvh4 run() {
        Bvh4 out;
        out_ = &out;
        if (b_.n_nodes() == 0) return out;
        need2_.assign((size_t)b_.n_nodes(), 0);
        height(0);
        emit(0, std::max(budget_, need2_[0]));
        out.stack_need = need4(0);
        return out;
    }

private:
    const Bvh& b_;
    int budget_;
    Bvh4* out_ = nullptr;
    std::vector<int> need2_;       // pending entries a binary walk below (and including) the node can have
    struct Slot { int32_t ref; int32_t src; double box[6]; };

    int height(int node) {
        int h = 0;
        for (int side = 0; side < 2; side++) {
            const int32_t c = b_.child[2 * (size_t)node + side];
            if (c >= 0) h = std::max(h, height(c));
        }
        return need2_[node] = h + 1;
    }
    static double area(const Slot& s) {
        const double d[3] = {s.box[3] - s.box[0], s.box[4] - s.box[1], s.box[5] - s.box[2]};
        if (d[0] < 0 || d[1] < 0 || d[2] < 0) return 0;
        return 2 * (d[0] * d[1] + d[1] * d[2] + d[2] * d[0]);
    }
    Slot slot_of(int node, int side) const {
        Slot s;
        s.ref = b_.child[2 * (size_t)node + side];
        s.src = (node << 1) | side;
        for (int a = 0; a < 6; a++) s.box[a] = b_.box[12 * (size_t)node + 6 * side + a];
        return s;
    }
    int emit(int bnode, int B) {
        const int idx = out_->n_nodes();
        out_->box.resize(out_->box.size() + 24);
        out_->child.resize(out_->child.size() + 4);
        out_->src.resize(out_->src.size() + 4);
        Slot s[4];
        int k = 0;
        for (int side = 0; side < 2; side++)
            if (b_.child[2 * (size_t)bnode + side] != EMPTY) s[k++] = slot_of(bnode, side);
        while (k < 4) {
            int best = -1;
            double best_area = -1;
            const int bound = B - k;            // what every child may still need once there are k + 1 of them
            for (int i = 0; i < k; i++) {
                if (s[i].ref < 0) continue;
                bool ok = true;
                for (int j = 0; j < k && ok; j++)
                    if (j != i && s[j].ref >= 0 && need2_[s[j].ref] > bound) ok = false;
                int grand = 0;
                for (int side = 0; side < 2 && ok; side++) {
                    const int32_t c = b_.child[2 * (size_t)s[i].ref + side];
                    if (c == EMPTY) continue;
                    grand++;
                    if (c >= 0 && need2_[c] > bound) ok = false;
                }
                if (grand != 2) ok = false;
                if (ok && area(s[i]) > best_area) { best_area = area(s[i]); best = i; }
            }
            if (best < 0) break;
            const int node = s[best].ref;
            for (int j = k; j > best + 1; j--) s[j] = s[j - 1];
            s[best] = slot_of(node, 0);
            s[best + 1] = slot_of(node, 1);
            k++;
        }
        const int child_budget = B - (k - 1);
        for (int i = 0; i < 4; i++) {
            int32_t ref = EMPTY, src = -1;
            double box[6] = {std::numeric_limits<double>::infinity(), std::numeric_limits<double>::infinity(), std::numeric_limits<double>::infinity(),
                             -std::numeric_limits<double>::infinity(), -std::numeric_limits<double>::infinity(), -std::numeric_limits<double>::infinity()};
            if (i < k) {
                for (int a = 0; a < 6; a++) box[a] = s[i].box[a];
                src = s[i].src;
                ref = s[i].ref >= 0 ? emit(s[i].ref, child_budget) : s[i].ref;
            }
            out_->child[4 * (size_t)idx + i] = ref;
            out_->src[4 * (size_t)idx + i] = src;
            for (int a = 0; a < 6; a++) out_->box[24 * (size_t)idx + 6 * i + a] = box[a];
        }
        return idx;
    }
    int need4(int node) const {
        int k = 0, deepest = 0;
        for (int i = 0; i < 4; i++) {
            const int32_t c = out_->child[4 * (size_t)node + i];
            if (c == EMPTY) continue;
            k++;
            if (c >= 0) deepest = std::max(deepest, need4(c));
        }
        return std::max(0, k - 1) + deepest;
    }
};

inline Bvh4 collapse_bvh4(const Bvh& b, int budget) { return Bvh4Collapser(b, budget).run(); }

}  // namespace rtc
