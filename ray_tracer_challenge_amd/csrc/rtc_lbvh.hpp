// rtc_lbvh.hpp -- the triangle BVH built on the GPU: Morton codes of the triangle
// centroids, a radix sort, Karras' binary radix tree (one thread per inner node, no recursion), a
// bottom-up pass for the boxes.  Milliseconds instead of the host builder's ~100 ms, at the price
// of a tree a ray needs more steps through than the binned-SAH one; meant for scenes whose
// triangles move every frame (main.py:240-265's animation loops), not for a still.
//
// Any conservative hierarchy returns the reference's hit lists (its Group.local_intersect never
// culls, shapes.py:454), so the image does not depend on which builder made the tree.
//
// The index arithmetic lives in plain functions that also compile for the host (RTC_HD), where
// tests/test_lbvh_cpu.py runs them against brute force; the kernels only wrap them.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define RTC_HD __host__ __device__
#else
#define RTC_HD
#endif

namespace rtc {

RTC_HD inline int lbvh_clz64(uint64_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return v ? __clzll((long long)v) : 64;
#else
    return v ? __builtin_clzll(v) : 64;
#endif
}

// 10 bits per axis, x in the top bits
RTC_HD inline uint32_t lbvh_expand10(uint32_t v) {
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
RTC_HD inline uint32_t lbvh_morton30(float x, float y, float z) {   // x, y, z in [0, 1]
    float fx = x * 1024.0f, fy = y * 1024.0f, fz = z * 1024.0f;
    uint32_t ix = fx >= 1023.0f ? 1023u : (fx > 0.0f ? (uint32_t)fx : 0u);
    uint32_t iy = fy >= 1023.0f ? 1023u : (fy > 0.0f ? (uint32_t)fy : 0u);
    uint32_t iz = fz >= 1023.0f ? 1023u : (fz > 0.0f ? (uint32_t)fz : 0u);
    return (lbvh_expand10(ix) << 2) | (lbvh_expand10(iy) << 1) | lbvh_expand10(iz);
}

// length of the common prefix of leaf keys i and j (-1 outside the array); keys are unique and sorted
RTC_HD inline int lbvh_delta(const uint64_t* keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    return lbvh_clz64(keys[i] ^ keys[j]);
}

// Karras 2012: inner node i of the radix tree over n sorted leaf keys (n >= 2, 0 <= i < n - 1) covers
// leaves [first, last] and splits after leaf `split`: left child = leaf `split` if split == first else
// inner node `split`; right child = leaf split + 1 if split + 1 == last else inner node split + 1.
RTC_HD inline void lbvh_inner_node(const uint64_t* keys, int n, int i, int& first, int& last, int& split) {
    const int d = lbvh_delta(keys, n, i, i + 1) - lbvh_delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = lbvh_delta(keys, n, i, i - d);
    int lmax = 2;
    while (lbvh_delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (lbvh_delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = lbvh_delta(keys, n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) >> 1;
        if (lbvh_delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    split = i + s * d + (d < 0 ? -1 : 0);
    first = d > 0 ? i : j;
    last = d > 0 ? j : i;
}

}  // namespace rtc
