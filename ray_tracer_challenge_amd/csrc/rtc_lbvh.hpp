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

// 21 bits per axis (63-bit code): ten bits are too coarse once a scene holds several detailed meshes
RTC_HD inline uint64_t lbvh_expand21(uint64_t v) {
    v &= 0x1FFFFFull;
    v = (v | (v << 32)) & 0x001F00000000FFFFull;
    v = (v | (v << 16)) & 0x001F0000FF0000FFull;
    v = (v | (v << 8)) & 0x100F00F00F00F00Full;
    v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}
RTC_HD inline uint64_t lbvh_morton63(float x, float y, float z) {   // x, y, z in [0, 1]
    const double s = 2097152.0;                                        // 2^21
    double fx = (double)x * s, fy = (double)y * s, fz = (double)z * s;
    uint64_t ix = fx >= s - 1.0 ? 2097151ull : (fx > 0.0 ? (uint64_t)fx : 0ull);
    uint64_t iy = fy >= s - 1.0 ? 2097151ull : (fy > 0.0 ? (uint64_t)fy : 0ull);
    uint64_t iz = fz >= s - 1.0 ? 2097151ull : (fz > 0.0 ? (uint64_t)fz : 0ull);
    return (lbvh_expand21(ix) << 2) | (lbvh_expand21(iy) << 1) | lbvh_expand21(iz);
}

// length of the common prefix of leaf keys i and j (-1 outside the array).  Keys are sorted; equal
// keys are told apart by their position (as if the index were appended to the key), so the tree has no ties.
RTC_HD inline int lbvh_delta(const uint64_t* keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const uint64_t x = keys[i] ^ keys[j];
    if (x) return lbvh_clz64(x);
    uint32_t d = (uint32_t)i ^ (uint32_t)j;
#if defined(__HIP_DEVICE_COMPILE__)
    return 64 + (d ? __clz((int)d) : 32);
#else
    return 64 + (d ? __builtin_clz(d) : 32);
#endif
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
