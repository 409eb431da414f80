// rtc_qnode.hpp -- the 32-byte BVH node the trace kernels walk.
//
// The trace loop on a triangle mesh is bound by the CU's vector-memory pipe, not by arithmetic
// (profiles/r02_*: the texture-data unit is busy 67-81 % of a launch, ~30 cycles per wave-level load):
// every lane fetches its own node, and a node of two fp32 child boxes + two child references is 64 bytes =
// four 16-byte loads per lane and visit.  Here both child boxes are quantised to 8 bits per plane on a frame
// the node carries itself, which makes a node 32 bytes = two loads:
//
//   w0, w1   frame origin on the tree's 2^21 grid: gx | gy << 21 | gz << 42  (origin = grid.lo + g * grid.cell)
//   w2       per-axis scale exponents: ex | ey << 8 | ez << 16  (scale = 2^(e - 127), i.e. the byte IS the fp32 exponent)
//   w3..w5   q[12]: child0 lo xyz, child0 hi xyz, child1 lo xyz, child1 hi xyz; plane = origin + q * scale
//   w6, w7   child references, as in BvhNode (>= 0 inner node; < 0: ~c = first << 4 | (count - 1); RTC_EMPTY_CHILD = none)
//
// Quantisation only ever moves a plane outward (lo down, hi up), so a quantised box contains the fp32 box it
// was made from, which in turn contains the exact box padded outward (pad_boxes): the hierarchy stays
// conservative, and the reference never culls at all (shapes.py:454), so hit lists are unchanged.
// An absent child is the inverted box lo = 255, hi = 0 on every axis: no ray meets it.
// The same encoder runs on the host (rtc_scene_create) and on the device (GPU rebuilds and refits).
#pragma once
#include <stdint.h>
#include <math.h>

#ifndef RTC_HD
#ifdef __HIPCC__
#define RTC_HD __host__ __device__
#else
#define RTC_HD
#endif
#endif

namespace rtc {

struct alignas(32) QNode { uint32_t w[8]; };

// per tree: the grid its node origins live on and the (padded) box of the whole tree
struct TreeGrid { float lo[3], cell[3], root[6]; };

#define RTC_QGRID_MAX ((1 << 21) - 1)

// grid for a tree whose boxes all lie inside [lo, hi] (already padded fp32 values)
RTC_HD inline TreeGrid make_tree_grid(const float lo[3], const float hi[3]) {
    TreeGrid G;
    for (int k = 0; k < 3; k++) {
        float l = lo[k], h = hi[k];
        if (!(h >= l)) { l = 0.0f; h = 0.0f; }
        double ext = (double)h - (double)l;
        double margin = 1e-6 * (fabs((double)l) + fabs((double)h) + ext) + 1e-30;
        G.lo[k] = (float)((double)l - margin);
        if ((double)G.lo[k] > (double)l - 0.5 * margin) G.lo[k] = nextafterf(G.lo[k], -INFINITY);
        double cell = ((double)h - (double)G.lo[k]) / (double)(RTC_QGRID_MAX - 2);
        G.cell[k] = (float)cell;
        if ((double)G.cell[k] < cell) G.cell[k] = nextafterf(G.cell[k], INFINITY);
        if (!(G.cell[k] > 0.0f)) G.cell[k] = 1e-30f;
        G.root[k] = l; G.root[3 + k] = h;
    }
    return G;
}

RTC_HD inline float qnode_origin(uint32_t g, float cell, float lo) { return fmaf((float)g, cell, lo); }

// box[12] = child0 lo xyz, hi xyz, child1 lo xyz, hi xyz (fp32, padded); an absent child's box is ignored
RTC_HD inline void qnode_encode(const float box[12], int32_t c0, int32_t c1, int32_t empty_code, const TreeGrid& G, QNode& out) {
    const bool has[2] = {c0 != empty_code, c1 != empty_code};
    uint32_t g[3], ex[3], q[12];
    for (int k = 0; k < 3; k++) {
        float L = INFINITY, H = -INFINITY;
        for (int c = 0; c < 2; c++)
            if (has[c]) { L = fminf(L, box[6 * c + k]); H = fmaxf(H, box[6 * c + 3 + k]); }
        if (!(H >= L)) { L = G.root[k]; H = L; }                        // no child at all (never built)
        double gd = floor(((double)L - (double)G.lo[k]) / (double)G.cell[k]);
        if (gd < 0) gd = 0;
        if (gd > RTC_QGRID_MAX) gd = RTC_QGRID_MAX;
        uint32_t gi = (uint32_t)gd;
        float origin = qnode_origin(gi, G.cell[k], G.lo[k]);
        while (origin > L && gi > 0) { gi--; origin = qnode_origin(gi, G.cell[k], G.lo[k]); }
        g[k] = gi;
        const double o = (double)origin;
        double ext = (double)H - o;
        int e = -126;
        if (ext > 0) { int fe; (void)frexp(ext / 255.0, &fe); e = fe; }   // 2^fe > ext / 255 >= 2^(fe-1)
        if (e < -126) e = -126;
        for (;;) {
            if (e > 127) e = 127;
            const double scale = ldexp(1.0, e);
            bool fits = true;
            for (int c = 0; c < 2; c++) {
                if (!has[c]) { q[6 * c + k] = 255; q[6 * c + 3 + k] = 0; continue; }
                double lo = (double)box[6 * c + k], hi = (double)box[6 * c + 3 + k];
                double ql = floor((lo - o) / scale), qh = ceil((hi - o) / scale);
                if (ql < 0) ql = 0;
                while (ql > 0 && o + ql * scale > lo) ql -= 1;
                while (o + qh * scale < hi) qh += 1;
                if (ql > 255) ql = 255;
                if (qh > 255) { fits = false; break; }
                if (qh < 0) qh = 0;
                q[6 * c + k] = (uint32_t)ql; q[6 * c + 3 + k] = (uint32_t)qh;
            }
            if (fits || e >= 127) break;
            e++;
        }
        ex[k] = (uint32_t)(e + 127);
    }
    const uint64_t w = (uint64_t)g[0] | ((uint64_t)g[1] << 21) | ((uint64_t)g[2] << 42);
    out.w[0] = (uint32_t)w; out.w[1] = (uint32_t)(w >> 32);
    out.w[2] = ex[0] | (ex[1] << 8) | (ex[2] << 16);
    out.w[3] = q[0] | (q[1] << 8) | (q[2] << 16) | (q[3] << 24);
    out.w[4] = q[4] | (q[5] << 8) | (q[6] << 16) | (q[7] << 24);
    out.w[5] = q[8] | (q[9] << 8) | (q[10] << 16) | (q[11] << 24);
    out.w[6] = (uint32_t)c0; out.w[7] = (uint32_t)c1;
}

// the boxes a node decodes to (tests and tools): out[12] in BvhNode order
RTC_HD inline void qnode_decode(const QNode& n, const TreeGrid& G, double out[12]) {
    const uint64_t w = (uint64_t)n.w[0] | ((uint64_t)n.w[1] << 32);
    const uint32_t g[3] = {(uint32_t)(w & RTC_QGRID_MAX), (uint32_t)((w >> 21) & RTC_QGRID_MAX), (uint32_t)((w >> 42) & RTC_QGRID_MAX)};
    const uint32_t qw[3] = {n.w[3], n.w[4], n.w[5]};
    for (int k = 0; k < 3; k++) {
        const float origin = qnode_origin(g[k], G.cell[k], G.lo[k]);
        const double scale = ldexp(1.0, (int)((n.w[2] >> (8 * k)) & 255u) - 127);
        for (int c = 0; c < 2; c++)
            for (int side = 0; side < 2; side++) {
                const int i = 6 * c + 3 * side + k;
                const uint32_t q = (qw[i >> 2] >> (8 * (i & 3))) & 255u;
                out[i] = (double)origin + q * scale;
            }
    }
}

}  // namespace rtc
