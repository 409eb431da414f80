// rtc_lbvh_device.hpp -- kernels of the GPU triangle-BVH build (see rtc_lbvh.hpp for what and why).
// Output: the same BvhNode records (both child boxes as padded fp32 + two child refs) and the same
// BVH-ordered triangle rows the host builder produces, so the trace kernels do not know the difference.
#pragma once
#include "rtc_device.hpp"
#include "rtc_lbvh.hpp"

#ifndef RTC_GPU_LEAF
// Triangles per leaf of a GPU-built tree = consecutive rows of the Morton order.  ONE: with two, which triangles share a leaf is
// decided by the Morton order alone, and that pairing -- not the tree above it -- was most of what the GPU-built trees lost to
// the host's SAH tree (dragons 1080p fp64, frame on the GPU tree / frame on the host tree: radix tree 1.57 -> 1.18, PLOC 1.35 ->
// 1.12; teapot 1.42 -> 1.06, 1.30 -> 1.03; profiles/r03_experiments/gpu_tree_builders.txt).  Costs twice the nodes.
#define RTC_GPU_LEAF 1
#endif

namespace rtc {

// floats <-> unsigned ints with the same order (for atomicMin / atomicMax)
__device__ __forceinline__ uint32_t lbvh_ordered(float f) { uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float lbvh_unordered(uint32_t o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o); }

// item k = triangle row items[k] & 0x07FFFFFF: its box, padded outward like the host builder's boxes
// (2e-5 relative, >= 2e-5 absolute, rounded away), and the bounds of all box centres
__global__ void lbvh_boxes_kernel(int n, const int32_t* items, const double* tri_geom, float* boxes, uint32_t* bounds /* [6] min xyz, max xyz (ordered) */, double pad_abs) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const double* g = tri_geom + 12 * (size_t)((uint32_t)items[k] & 0x07FFFFFFu);
    float lo[3], hi[3];
    for (int a = 0; a < 3; a++) {
        double p1 = g[a], p2 = g[a] + g[3 + a], p3 = g[a] + g[6 + a];
        double l = fmin(p1, fmin(p2, p3)), h = fmax(p1, fmax(p2, p3));
        l -= fmax(2e-5 * fmax(1.0, fabs(l)), pad_abs); h += fmax(2e-5 * fmax(1.0, fabs(h)), pad_abs);
        lo[a] = __double2float_rd(l); hi[a] = __double2float_ru(h);
        boxes[6 * (size_t)k + a] = lo[a]; boxes[6 * (size_t)k + 3 + a] = hi[a];
        float c = 0.5f * (lo[a] + hi[a]);
        atomicMin(bounds + a, lbvh_ordered(c));
        atomicMax(bounds + 3 + a, lbvh_ordered(c));
    }
}

// key = 63-bit Morton code of the box centre, its lowest `drop_bits` bits cleared (a coarser grid makes a
// shallower tree); value = item index
__global__ void lbvh_keys_kernel(int n, const float* boxes, const uint32_t* bounds, int drop_bits, uint64_t* keys, uint32_t* index) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    float u[3];
    for (int a = 0; a < 3; a++) {
        float lo = lbvh_unordered(bounds[a]), hi = lbvh_unordered(bounds[3 + a]);
        float c = 0.5f * (boxes[6 * (size_t)k + a] + boxes[6 * (size_t)k + 3 + a]);
        float ext = hi - lo;
        u[a] = ext > 0.0f ? (c - lo) / ext : 0.0f;
    }
    keys[k] = (lbvh_morton63(u[0], u[1], u[2]) >> drop_bits) << drop_bits;
    index[k] = (uint32_t)k;
}

// leaf j = sorted items 2j and 2j + 1; its key is the first one's
__global__ void lbvh_leaf_keys_kernel(int n_leaves, const uint64_t* sorted, uint64_t* leaf_keys) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n_leaves) leaf_keys[j] = sorted[RTC_GPU_LEAF * (size_t)j];
}

__device__ __forceinline__ int32_t lbvh_leaf_code(int j, int n_items) {
    int first = RTC_GPU_LEAF * j, count = n_items - first < RTC_GPU_LEAF ? n_items - first : RTC_GPU_LEAF;
    return ~((first << 4) | (count - 1));
}

// inner node i: children and parent links ((parent << 1) | side)
__global__ void lbvh_inner_kernel(int n_leaves, int n_items, const uint64_t* leaf_keys, BvhNode* nodes, int32_t* parent_inner, int32_t* parent_leaf) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_leaves - 1) return;
    int first, last, split;
    lbvh_inner_node(leaf_keys, n_leaves, i, first, last, split);
    if (i == 0) parent_inner[0] = -1;
    int32_t c0, c1;
    if (split == first) { c0 = lbvh_leaf_code(split, n_items); parent_leaf[split] = (i << 1); }
    else { c0 = split; parent_inner[split] = (i << 1); }
    if (split + 1 == last) { c1 = lbvh_leaf_code(split + 1, n_items); parent_leaf[split + 1] = (i << 1) | 1; }
    else { c1 = split + 1; parent_inner[split + 1] = (i << 1) | 1; }
    nodes[i].child[0] = c0; nodes[i].child[1] = c1; nodes[i].pad[0] = 0; nodes[i].pad[1] = 0;
}

// boxes bottom-up: every leaf writes its box into its parent's record and climbs while it is the
// second child to arrive (the first one stops: its sibling will carry the union up)
__global__ void lbvh_refit_kernel(int n_leaves, int n_items, const uint32_t* sorted, const float* boxes, BvhNode* nodes, const int32_t* parent_inner,
                                  const int32_t* parent_leaf, uint32_t* arrived) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_leaves) return;
    float b[6] = {3.0e38f, 3.0e38f, 3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
    for (int q = RTC_GPU_LEAF * j; q < RTC_GPU_LEAF * j + RTC_GPU_LEAF && q < n_items; q++) {
        const float* t = boxes + 6 * (size_t)sorted[q];
        for (int a = 0; a < 3; a++) { b[a] = fminf(b[a], t[a]); b[3 + a] = fmaxf(b[3 + a], t[3 + a]); }
    }
    int32_t link = parent_leaf[j];
    while (link >= 0) {
        const int p = link >> 1, side = link & 1;
        float* slot = nodes[p].box + 6 * side;
        for (int a = 0; a < 6; a++) __hip_atomic_store(slot + a, b[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        if (atomicAdd(arrived + p, 1u) == 0u) return;
        const float* other = nodes[p].box + 6 * (1 - side);           // written by the sibling's carrier before its atomicAdd
        for (int a = 0; a < 3; a++) {
            b[a] = fminf(b[a], __hip_atomic_load(other + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            b[3 + a] = fmaxf(b[3 + a], __hip_atomic_load(other + 3 + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
        link = parent_inner[p];
    }
}

// ---- PLOC: the topology of the binary tree by parallel locally-ordered clustering (Meister & Bittner 2017) ----
// The Karras radix tree above splits wherever the Morton codes' bits say; its boxes overlap where a mesh crosses a cell
// boundary, and rays pay for that (frame 1.65 x the host SAH tree's on the dragons).  PLOC builds bottom-up instead: the
// clusters -- at first the leaves, in Morton order -- each look for the neighbour within `radius` places whose union with
// them has the smallest surface area; two clusters that choose each other merge into a node; the survivors are compacted
// in order; repeat until one cluster is left.  Local in memory like the LBVH, but every merge is chosen by area.
// Cluster ids: < n_leaves = leaf j (sorted items 2j, 2j + 1), else inner node id - n_leaves.  Inner nodes are numbered from
// the back (the k-th merge of the whole build gets node n_leaves - 2 - k), so that the last merge, the root, is node 0 as
// the walks and the kernels below expect.  Boxes of clusters: cbox[cid][6].
struct PlocState { int32_t n; int32_t merges; int32_t iterations; int32_t pad; };      // clusters left, merges done so far

__device__ __forceinline__ float ploc_area(const float* a, const float* b) {
    const float dx = fmaxf(a[3], b[3]) - fminf(a[0], b[0]), dy = fmaxf(a[4], b[4]) - fminf(a[1], b[1]), dz = fmaxf(a[5], b[5]) - fminf(a[2], b[2]);
    return dx * dy + dy * dz + dz * dx;
}
__global__ void ploc_init_kernel(int n_leaves, int n_items, const uint32_t* sorted, const float* boxes, int32_t* cluster, float* cbox, PlocState* st) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0) { st->n = n_leaves; st->merges = 0; st->iterations = 0; st->pad = 0; }
    if (j >= n_leaves) return;
    float b[6] = {3.0e38f, 3.0e38f, 3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
    for (int q = RTC_GPU_LEAF * j; q < RTC_GPU_LEAF * j + RTC_GPU_LEAF && q < n_items; q++) {
        const float* t = boxes + 6 * (size_t)sorted[q];
        for (int a = 0; a < 3; a++) { b[a] = fminf(b[a], t[a]); b[3 + a] = fmaxf(b[3 + a], t[3 + a]); }
    }
    for (int a = 0; a < 6; a++) cbox[6 * (size_t)j + a] = b[a];
    cluster[j] = j;
}
// nearest neighbour of every cluster within `radius` places: smallest area of the union.  Equal areas go to the pair with the
// smaller hash of its two places -- symmetric in i and j, so the pair that is smallest for both of its members chooses itself
// from both sides.  (Ties to the lower place, as this first was, make a chain i -> i-1 -> i-2 ... out of equal boxes -- three
// thousand copies of one triangle merged ONE pair per iteration and ran into the iteration limit, tools/builder_stress.py.)
__device__ __forceinline__ uint32_t ploc_pair_key(int a, int b) {
    uint32_t lo = (uint32_t)(a < b ? a : b), hi = (uint32_t)(a < b ? b : a);
    uint32_t h = lo * 0x9E3779B1u ^ (hi + 0x7F4A7C15u) * 0x85EBCA6Bu;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
    return h;
}
__global__ void ploc_nearest_kernel(const PlocState* st, int radius, const int32_t* cluster, const float* cbox, int32_t* nearest) {
    const int n = st->n;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (n <= 1 || i >= n) return;
    float mine[6];
    for (int a = 0; a < 6; a++) mine[a] = cbox[6 * (size_t)cluster[i] + a];
    float best = 3.0e38f; int at = -1; uint32_t best_key = 0xFFFFFFFFu;
    const int lo = i - radius < 0 ? 0 : i - radius, hi = i + radius >= n ? n - 1 : i + radius;
    for (int j = lo; j <= hi; j++) {
        if (j == i) continue;
        const float area = ploc_area(mine, cbox + 6 * (size_t)cluster[j]);
        if (area < best) { best = area; at = j; best_key = ploc_pair_key(i, j); }
        else if (area == best && at >= 0) {
            const uint32_t key = ploc_pair_key(i, j);
            if (key < best_key) { at = j; best_key = key; }
        }
    }
    nearest[i] = at;
}
// per place: low word 1 if the cluster survives at its place (merged pairs survive at the lower place), high word 1 if a
// node is made there; scanned by the caller (exclusive, 64-bit adds = both counts at once)
__global__ void ploc_flags_kernel(const PlocState* st, int n_max, const int32_t* nearest, unsigned long long* flags) {
    const int n = st->n;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_max) return;
    unsigned long long f = 0ull;
    if (n > 1 && i < n) {
        const int j = nearest[i];
        const bool mutual = j >= 0 && nearest[j] == i;
        if (!mutual) f = 1ull;
        else if (i < j) f = 1ull | (1ull << 32);
    }
    flags[i] = f;
}
__global__ void ploc_merge_kernel(PlocState* st, int n_leaves, int n_items, const int32_t* nearest, const unsigned long long* flags, const unsigned long long* offsets,
                                  const int32_t* cluster, int32_t* cluster_out, float* cbox, BvhNode* nodes, int32_t* parent_inner, int32_t* parent_leaf) {
    const int n = st->n;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (n <= 1 || i >= n) return;
    const unsigned long long f = flags[i], off = offsets[i];
    if (!(f & 1ull)) return;                                             // the upper half of a merged pair: its partner carries it
    const int pos = (int)(uint32_t)off;
    if (!(f >> 32)) { cluster_out[pos] = cluster[i]; return; }
    const int node = n_leaves - 2 - (st->merges + (int)(off >> 32));
    const int32_t c[2] = {cluster[i], cluster[nearest[i]]};
    float b[6];
    for (int s = 0; s < 2; s++) {
        const float* cb = cbox + 6 * (size_t)c[s];
        for (int a = 0; a < 6; a++) nodes[node].box[6 * s + a] = cb[a];
        for (int a = 0; a < 3; a++) { b[a] = s ? fminf(b[a], cb[a]) : cb[a]; b[3 + a] = s ? fmaxf(b[3 + a], cb[3 + a]) : cb[3 + a]; }
        if (c[s] < n_leaves) { nodes[node].child[s] = lbvh_leaf_code(c[s], n_items); parent_leaf[c[s]] = (node << 1) | s; }
        else { nodes[node].child[s] = c[s] - n_leaves; parent_inner[c[s] - n_leaves] = (node << 1) | s; }
    }
    nodes[node].pad[0] = nodes[node].pad[1] = 0;
    for (int a = 0; a < 6; a++) cbox[6 * (size_t)(n_leaves + node) + a] = b[a];
    cluster_out[pos] = n_leaves + node;
    if (node == 0) parent_inner[0] = -1;
}
// after the merges of an iteration: the counts move on (one thread; the scan's last element + its flag = the totals)
__global__ void ploc_advance_kernel(PlocState* st, int n_max, const unsigned long long* flags, const unsigned long long* offsets) {
    if (blockIdx.x != 0 || threadIdx.x != 0 || st->n <= 1) return;
    const unsigned long long total = offsets[n_max - 1] + flags[n_max - 1];
    st->n = (int)(uint32_t)total;
    st->merges += (int)(total >> 32);
    st->iterations += 1;
}

// a single leaf (1 or 2 triangles): node 0 with one child
__global__ void lbvh_single_kernel(int n_items, const uint32_t* sorted, const float* boxes, BvhNode* nodes) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    float b[6] = {3.0e38f, 3.0e38f, 3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
    for (int q = 0; q < n_items && q < RTC_GPU_LEAF; q++) {
        const float* t = boxes + 6 * (size_t)sorted[q];
        for (int a = 0; a < 3; a++) { b[a] = fminf(b[a], t[a]); b[3 + a] = fmaxf(b[3 + a], t[3 + a]); }
    }
    for (int a = 0; a < 6; a++) { nodes[0].box[a] = b[a]; nodes[0].box[6 + a] = a < 3 ? 3.0e38f : -3.0e38f; }
    nodes[0].child[0] = lbvh_leaf_code(0, n_items); nodes[0].child[1] = RTC_EMPTY_CHILD; nodes[0].pad[0] = nodes[0].pad[1] = 0;
}

// triangle rows into BVH (= sorted) order, both precisions, leaf id | does-not-cast-shadows << 31, and
// which row went where (for refits)
__global__ void lbvh_reorder_kernel(int n, const uint32_t* sorted, const int32_t* items, const double* geom64, const float* geom32, const int32_t* tri_leaf,
                                    double* out64, float* out32, int32_t* out_leaf, int32_t* out_row) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 12LL * n) return;
    int pos = (int)(t / 12), c = (int)(t % 12);
    uint32_t code = (uint32_t)items[sorted[pos]];
    uint32_t row = code & 0x07FFFFFFu;
    out64[t] = geom64[12 * (size_t)row + c];
    out32[t] = geom32[12 * (size_t)row + c];
    if (c == 0) { out_leaf[pos] = (int32_t)((uint32_t)tri_leaf[row] | (((code >> 27) & 1u) << 31)); out_row[pos] = (int32_t)row; }
}

// deepest leaf (inner nodes above it), for the traversal-stack check
__global__ void lbvh_depth_kernel(int n_leaves, const int32_t* parent_inner, const int32_t* parent_leaf, int32_t* deepest) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_leaves) return;
    int d = 0;
    for (int32_t link = parent_leaf[j]; link >= 0 && d < 4096; link = parent_inner[link >> 1]) d++;
    atomicMax(deepest, d);
}

// triangle table (both precisions) refreshed from fp64 rows already on the device
__global__ void tri_rows_from_device_kernel(long long n12, const double* geom, const double* norm, double* geom64, double* norm64, float* geom32, float* norm32) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n12) return;
    double g = geom[t], m = norm[t];
    geom64[t] = g; norm64[t] = m; geom32[t] = (float)g; norm32[t] = (float)m;
}

// ---- refit: the tree keeps its shape (whichever builder made it), only the boxes follow the triangles ----

// BVH-ordered rows refreshed from the triangle table: position pos holds row row_of[pos]
__global__ void bvh_gather_rows_kernel(int n, const int32_t* row_of, const double* geom64, const float* geom32, double* out64, float* out32) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 12LL * n) return;
    int pos = (int)(t / 12), c = (int)(t % 12);
    size_t src = 12 * (size_t)row_of[pos] + c;
    out64[t] = geom64[src];
    out32[t] = geom32[src];
}

// One thread per node: it boxes the node's leaf children from their (BVH-ordered) rows, padded like the builders'
// boxes, and counts itself as arrived; a node is complete once its own thread and the carriers of its inner
// children have arrived, and whoever arrives last carries the union of its two boxes up into the parent's record.
// parent[i] = (parent node << 1) | side, -1 for the root; arrived[] zeroed by the caller.
__global__ void bvh_refit_kernel(int n_nodes, BvhNode* nodes, const int32_t* parent, const double* tb_geom, uint32_t* arrived, double pad_abs) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    for (int side = 0; side < 2; side++) {
        const int32_t c = nodes[i].child[side];
        if (c >= 0 || c == RTC_EMPTY_CHILD) continue;
        const int first = (~c) >> 4, count = ((~c) & 15) + 1;
        float* slot = nodes[i].box + 6 * side;
        for (int a = 0; a < 3; a++) {
            double l = 1.0e300, h = -1.0e300;
            for (int q = first; q < first + count; q++) {
                const double* g = tb_geom + 12 * (size_t)q;
                double p1 = g[a], p2 = g[a] + g[3 + a], p3 = g[a] + g[6 + a];
                l = fmin(l, fmin(p1, fmin(p2, p3))); h = fmax(h, fmax(p1, fmax(p2, p3)));
            }
            l -= fmax(2e-5 * fmax(1.0, fabs(l)), pad_abs); h += fmax(2e-5 * fmax(1.0, fabs(h)), pad_abs);
            __hip_atomic_store(slot + a, __double2float_rd(l), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(slot + 3 + a, __double2float_ru(h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    int p = i;
    for (int guard = 0; guard < 4096; guard++) {
        __threadfence();
        const uint32_t needed = 1u + (nodes[p].child[0] >= 0 ? 1u : 0u) + (nodes[p].child[1] >= 0 ? 1u : 0u);
        if (atomicAdd(arrived + p, 1u) + 1u != needed) return;
        const int32_t link = parent[p];
        if (link < 0) return;
        float b[6];
        for (int a = 0; a < 3; a++) {
            b[a] = fminf(__hip_atomic_load(nodes[p].box + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                         __hip_atomic_load(nodes[p].box + 6 + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            b[3 + a] = fmaxf(__hip_atomic_load(nodes[p].box + 3 + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                             __hip_atomic_load(nodes[p].box + 9 + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
        const int q = link >> 1;
        float* slot = nodes[q].box + 6 * (link & 1);
        for (int a = 0; a < 6; a++) __hip_atomic_store(slot + a, b[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        p = q;
    }
}

// ---- the four-wide form rays walk (BvhNode4), kept in step with the binary tree builds and refits work on ----

// after a refit of the binary tree: every wide child box copied from where it lives in the binary tree (BvhNode4::src)
__global__ void bvh4_refresh_kernel(int n4, BvhNode4* nodes4, const BvhNode* nodes2) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    for (int k = 0; k < 4; k++) {
        const int32_t src = nodes4[i].src[k];
        if (src < 0) continue;
        const float* from = nodes2[src >> 1].box + 6 * (src & 1);
        for (int a = 0; a < 6; a++) nodes4[i].box[6 * k + a] = from[a];
    }
}

// height[p] of every inner node of the GPU-built binary tree = the pending stack entries a binary walk can have at and
// below it (1 + the larger of its children's; leaves 0): every leaf climbs, raising the heights on its way, and stops
// where somebody else has already reported at least as much (who then carries at least that much further up)
__global__ void lbvh_height_kernel(int n_leaves, const int32_t* parent_inner, const int32_t* parent_leaf, int32_t* height) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_leaves) return;
    int level = 1;
    for (int32_t link = parent_leaf[j]; link >= 0 && level < 4096; link = parent_inner[link >> 1], level++)
        if (atomicMax(height + (link >> 1), level) >= level) return;
}

// What a wide node at binary node `a` with `budget` pending entries to spare takes in: both children replaced by their own
// children (four slots) if everything below those still fits budget - 3, else the child with the larger box alone (three
// slots, budget - 2) if that fits, else the other, else the binary node as it is (two slots, budget - 1: always fits, since
// height[a] <= budget).  The same invariant as the host's collapse_bvh4 (rtc_bvh.hpp): need4 <= budget by construction.
struct Collapse4 { bool expand[2]; int k; };
__device__ __forceinline__ Collapse4 lbvh_collapse_decision(const BvhNode* nodes2, const int32_t* height, int a, int budget) {
    const int32_t c[2] = {nodes2[a].child[0], nodes2[a].child[1]};
    int need[2], gneed[2];
    float area[2];
    for (int s = 0; s < 2; s++) {
        need[s] = c[s] >= 0 ? height[c[s]] : 0;
        gneed[s] = 0;
        if (c[s] >= 0)
            for (int q = 0; q < 2; q++) { const int32_t g = nodes2[c[s]].child[q]; if (g >= 0) gneed[s] = max(gneed[s], height[g]); }
        const float* b = nodes2[a].box + 6 * s;
        const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
        area[s] = (dx < 0.0f || dy < 0.0f || dz < 0.0f) ? 0.0f : dx * dy + dy * dz + dz * dx;
    }
    Collapse4 d;
    d.expand[0] = d.expand[1] = false; d.k = 2;
    if (c[0] >= 0 && c[1] >= 0 && max(gneed[0], gneed[1]) <= budget - 3) { d.expand[0] = d.expand[1] = true; d.k = 4; return d; }
    const bool ok0 = c[0] >= 0 && gneed[0] <= budget - 2 && need[1] <= budget - 2;
    const bool ok1 = c[1] >= 0 && gneed[1] <= budget - 2 && need[0] <= budget - 2;
    if (ok0 && (!ok1 || area[0] >= area[1])) { d.expand[0] = true; d.k = 3; }
    else if (ok1) { d.expand[1] = true; d.k = 3; }
    return d;
}

// The GPU-built binary tree collapsed into wide nodes, one thread per binary inner node.  Wide node i sits at binary node
// i's index (nodes that were taken into their parent's record leave a hole: an all-empty record nothing refers to).  Whether
// a node keeps a record of its own, and with what budget, follows from its ancestors' decisions: the thread replays them
// from the root down along its own path (<= 31 steps).
__global__ void lbvh_collapse4_kernel(int n_inner, const BvhNode* nodes2, const int32_t* parent_inner, const int32_t* height, int budget, BvhNode4* nodes4) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_inner) return;
    int path[64];
    int len = 0;
    for (int a = i; a >= 0 && len < 64; a = parent_inner[a] >= 0 ? (parent_inner[a] >> 1) : -1) path[len++] = a;      // i ... root
    BvhNode4 out;
    for (int k = 0; k < 4; k++) {
        out.child[k] = RTC_EMPTY_CHILD; out.src[k] = -1;
        for (int a = 0; a < 3; a++) { out.box[6 * k + a] = 3.0e38f; out.box[6 * k + 3 + a] = -3.0e38f; }
    }
    int t = len - 1, B = budget;
    bool kept = len < 64;
    Collapse4 d = {{false, false}, 2};
    while (kept) {
        d = lbvh_collapse_decision(nodes2, height, path[t], B);
        if (t == 0) break;                                               // path[0] = i keeps a record, with decision d
        const int side = nodes2[path[t]].child[0] == path[t - 1] ? 0 : 1;
        B -= d.k - 1;
        if (d.expand[side]) {
            if (t == 1) kept = false;                                    // i itself went into its parent's record
            t -= 2;
        } else t -= 1;
    }
    if (kept) {
        int k = 0;
        for (int s = 0; s < 2; s++) {
            const int32_t c = nodes2[i].child[s];
            if (d.expand[s]) {
                for (int q = 0; q < 2; q++, k++) {
                    out.child[k] = nodes2[c].child[q]; out.src[k] = (c << 1) | q;
                    for (int a = 0; a < 6; a++) out.box[6 * k + a] = nodes2[c].box[6 * q + a];
                }
            } else {
                out.child[k] = c; out.src[k] = c == RTC_EMPTY_CHILD ? -1 : ((i << 1) | s);
                for (int a = 0; a < 6; a++) out.box[6 * k + a] = nodes2[i].box[6 * s + a];
                k++;
            }
        }
    }
    nodes4[i] = out;
}

}  // namespace rtc
