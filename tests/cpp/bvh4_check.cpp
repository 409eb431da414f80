// Host check of the four-wide collapse of the triangle BVH (csrc/rtc_bvh.hpp, collapse_bvh4): for random, clustered and
// degenerate box sets and several stack budgets, the wide tree must hold every leaf range of the binary tree exactly once,
// every child box must be the box the binary tree had for that subtree (hence contain everything below it), and the
// worst-case number of pending stack entries of a walk -- (children - 1) per level -- must fit the budget it was built for.
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <vector>
#include "../../ray_tracer_challenge_amd/csrc/rtc_bvh.hpp"

using namespace rtc;

static void binary_leaves(const Bvh& b, int node, std::map<int32_t, int>& out) {
    for (int side = 0; side < 2; side++) {
        int32_t c = b.child[2 * (size_t)node + side];
        if (c == BvhBuilder::EMPTY) continue;
        if (c >= 0) binary_leaves(b, c, out); else out[c]++;
    }
}

// returns the pending-entry need below `node`, checks boxes on the way; box of the items below a child must sit inside the child's box
static int walk4(const Bvh4& t, const std::vector<Aabb>& boxes, const Bvh& b, int node, std::map<int32_t, int>& leaves, int& visited, bool& ok, Aabb& bound_out) {
    visited++;
    int k = 0, deepest = 0;
    bound_out.reset();
    for (int i = 0; i < 4; i++) {
        int32_t c = t.child[4 * (size_t)node + i];
        if (c == Bvh4Collapser::EMPTY) continue;
        k++;
        Aabb below;
        if (c >= 0) deepest = std::max(deepest, walk4(t, boxes, b, c, leaves, visited, ok, below));
        else {
            leaves[c]++;
            int code = ~c, first = code >> 4, count = (code & 15) + 1;
            below.reset();
            for (int q = first; q < first + count; q++) below.grow(boxes[b.order[q]]);
        }
        const double* cb = &t.box[24 * (size_t)node + 6 * i];
        for (int a = 0; a < 3; a++) if (below.lo[a] < cb[a] || below.hi[a] > cb[3 + a]) ok = false;
        Aabb mine; for (int a = 0; a < 3; a++) { mine.lo[a] = cb[a]; mine.hi[a] = cb[3 + a]; }
        bound_out.grow(mine);
    }
    return (k > 0 ? k - 1 : 0) + deepest;
}

int main() {
    std::mt19937_64 rng(2024);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    long long k_hist[5] = {0, 0, 0, 0, 0};
    int worst_need = 0;
    for (int trial = 0; trial < 120; trial++) {
        int n = trial < 6 ? trial + 1 : 2 + (int)(rng() % (trial < 60 ? 300 : 20000));
        std::vector<Aabb> boxes(n);
        for (int i = 0; i < n; i++) {
            double c[3];
            for (int a = 0; a < 3; a++) {
                if (trial % 4 == 0) c[a] = U(rng) * 10;                                  // uniform
                else if (trial % 4 == 1) c[a] = (double)(rng() % 5) * 3 + U(rng) * 0.2;  // clusters
                else if (trial % 4 == 2) c[a] = a == 0 ? U(rng) * 100 : 0.0;             // a line: deep, thin tree
                else c[a] = a == 0 ? (double)i * (1.0 + (double)i * 1e-3) : 0.0;         // spacing that grows: SAH peels one item at a time
            }
            for (int a = 0; a < 3; a++) { double e = U(rng) * 0.05; boxes[i].lo[a] = c[a] - e; boxes[i].hi[a] = c[a] + e; }
        }
        const int leaf = 1 + trial % 3;
        Bvh b = build_bvh(boxes, leaf, 30);
        if (b.depth > 31) { printf("FAIL binary depth %d\n", b.depth); return 1; }
        for (int budget : {31, 35, b.depth}) {
            Bvh4 t = collapse_bvh4(b, budget);
            std::map<int32_t, int> want, got;
            binary_leaves(b, 0, want);
            int visited = 0; bool ok = true; Aabb all;
            int need = walk4(t, boxes, b, 0, got, visited, ok, all);
            if (!ok) { printf("FAIL trial %d: a child box does not contain what is below it\n", trial); return 1; }
            if (got != want) { printf("FAIL trial %d: leaf ranges differ (%zu vs %zu)\n", trial, got.size(), want.size()); return 1; }
            for (auto& kv : got) if (kv.second != 1) { printf("FAIL trial %d: leaf range %d times\n", trial, kv.second); return 1; }
            if (visited != t.n_nodes()) { printf("FAIL trial %d: %d of %d nodes reachable\n", trial, visited, t.n_nodes()); return 1; }
            if (need != t.stack_need) { printf("FAIL trial %d: stack_need %d, recomputed %d\n", trial, t.stack_need, need); return 1; }
            if (need > std::max(budget, b.depth)) { printf("FAIL trial %d: need %d over budget %d (binary depth %d)\n", trial, need, budget, b.depth); return 1; }
            if (t.n_nodes() > b.n_nodes()) { printf("FAIL trial %d: more wide nodes than binary ones\n", trial); return 1; }
            worst_need = std::max(worst_need, need);
            if (budget == 31)
                for (int i = 0; i < t.n_nodes(); i++) { int k = 0; for (int q = 0; q < 4; q++) k += t.child[4 * (size_t)i + q] != Bvh4Collapser::EMPTY; k_hist[k]++; }
        }
    }
    printf("OK worst stack need %d; nodes by children: 1:%lld 2:%lld 3:%lld 4:%lld\n", worst_need, k_hist[1], k_hist[2], k_hist[3], k_hist[4]);
    return 0;
}
