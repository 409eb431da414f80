// Host check of the radix-tree index arithmetic the GPU BVH build uses (csrc/rtc_lbvh.hpp):
// for random and heavily clustered key sets, the inner nodes must form one binary tree over all
// leaves -- every leaf and every inner node but the root has exactly one parent, a node's children
// split its leaf range in two, and the range of a node is exactly the leaves below it.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../../ray_tracer_challenge_amd/csrc/rtc_lbvh.hpp"

static int check(const std::vector<uint64_t>& keys) {
    const int n = (int)keys.size();
    std::vector<int> parent_inner(n - 1, -1), parent_leaf(n, -1), first(n - 1), last(n - 1), split(n - 1);
    for (int i = 0; i < n - 1; i++) {
        rtc::lbvh_inner_node(keys.data(), n, i, first[i], last[i], split[i]);
        if (first[i] < 0 || last[i] >= n || first[i] >= last[i] || split[i] < first[i] || split[i] >= last[i]) return 1;
        if (i != first[i] && i != last[i]) return 2;                    // Karras: node i sits at one end of its range
        int kids[2][2] = {{first[i], split[i]}, {split[i] + 1, last[i]}};
        for (int k = 0; k < 2; k++) {
            if (kids[k][0] == kids[k][1]) { if (parent_leaf[kids[k][0]] != -1) return 3; parent_leaf[kids[k][0]] = i; }
            else { int c = k == 0 ? split[i] : split[i] + 1; if (c < 0 || c >= n - 1 || parent_inner[c] != -1) return 4; parent_inner[c] = i; }
        }
    }
    if (parent_inner[0] != -1) return 5;
    if (first[0] != 0 || last[0] != n - 1) return 6;
    for (int i = 1; i < n - 1; i++) if (parent_inner[i] < 0) return 7;
    for (int j = 0; j < n; j++) if (parent_leaf[j] < 0) return 8;
    int max_depth = 0;
    for (int i = 0; i < n - 1; i++) {                                   // child ranges = the two halves of the parent's
        int k0 = split[i], k1 = split[i] + 1;
        if (first[i] != split[i]) { if (first[k0] != first[i] || last[k0] != split[i]) return 9; }
        if (split[i] + 1 != last[i]) { if (first[k1] != split[i] + 1 || last[k1] != last[i]) return 10; }
    }
    for (int j = 0; j < n; j++) {
        int d = 1, p = parent_leaf[j];
        while (parent_inner[p] >= 0) { p = parent_inner[p]; if (++d > 200) return 11; }
        if (p != 0) return 12;
        max_depth = std::max(max_depth, d);
    }
    return -max_depth;      // <= 0: fine, the depth
}

int main() {
    std::mt19937_64 rng(12345);
    int worst = 0;
    for (int trial = 0; trial < 300; trial++) {
        int n = 2 + (int)(rng() % (trial < 100 ? 40 : 5000));
        std::vector<uint64_t> keys(n);
        for (int i = 0; i < n; i++) {
            uint32_t code;
            if (trial % 3 == 0) code = (uint32_t)(rng() & 0x3FFFFFFF);                                   // uniform
            else if (trial % 3 == 1) code = (uint32_t)((rng() % 7) * 0x01234567u + (rng() & 0xFF)) & 0x3FFFFFFF;   // clusters
            else code = (uint32_t)(rng() % 3);                                                           // almost all duplicates
            keys[i] = trial % 2 ? ((uint64_t)code << 32) | (uint32_t)i : (uint64_t)code;           // unique keys, or ties broken by position
        }
        std::sort(keys.begin(), keys.end());
        int rc = check(keys);
        if (rc > 0) { printf("FAIL trial %d n %d code %d\n", trial, n, rc); return 1; }
        worst = std::max(worst, -rc);
    }
    // Morton code sanity: monotone in each coordinate's leading bits, within 30 bits
    if (rtc::lbvh_morton30(0, 0, 0) != 0 || rtc::lbvh_morton30(1, 1, 1) != 0x3FFFFFFFu) { printf("FAIL morton ends\n"); return 1; }
    if (rtc::lbvh_morton63(0, 0, 0) != 0 || rtc::lbvh_morton63(1, 1, 1) != 0x7FFFFFFFFFFFFFFFull) { printf("FAIL morton63 ends\n"); return 1; }
    if (!(rtc::lbvh_morton63(0.75f, 0, 0) > rtc::lbvh_morton63(0.25f, 1, 1))) { printf("FAIL morton63 order\n"); return 1; }
    for (int a = 0; a < 3; a++)                       // one axis at a time: its 21 bits land on every third bit
        for (int bit = 0; bit < 21; bit++) {
            float v = (float)((1u << bit) + 0.5) / 2097152.0f;
            uint64_t m = rtc::lbvh_morton63(a == 0 ? v : 0, a == 1 ? v : 0, a == 2 ? v : 0);
            if (m != (1ull << (3 * bit + (2 - a)))) { printf("FAIL morton63 bit %d axis %d\n", bit, a); return 1; }
        }
    if (!(rtc::lbvh_morton30(0.75f, 0, 0) > rtc::lbvh_morton30(0.25f, 1, 1))) { printf("FAIL morton order\n"); return 1; }
    printf("OK deepest tree %d\n", worst);
    return 0;
}
