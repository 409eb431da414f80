// rtc_device.hpp -- device side of the MI355X trace loop (gfx950, wave64).
//
// Everything the reference does per ray in Python lives here as inlined device functions,
// templated on the arithmetic type R (double = the reference's fp64, float = the fp32 mode: half the registers and bytes, gated at 1/255 against fp64).
// Citations are reference file:line.  Differences from the reference's *structure* (none of
// them change a hit list):
//   * one collapsed world->object 3x4 per leaf instead of an inverse per tree level;
//   * triangles are pre-transformed to world space (their det-epsilon is rescaled to match);
//   * a BVH over world-space AABBs replaces "every leaf for every ray" (shapes.py:454);
//   * nearest-hit / any-hit queries replace full sorted hit lists; the refraction container
//     walk (renderer.py:224-242) is a second, order-free pass over the line's leaves;
//   * recursion (renderer.py:400-406, 336-353) is an explicit per-lane stack.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/rtc_api.h"

#ifndef RTC_BLOCK
#define RTC_BLOCK 256
#endif
// Threads per workgroup of the trace kernel variants without CSG: one wave.  A workgroup is admitted when ONE CU has a free
// wave slot for each of its waves; the waves of a 256-thread workgroup end at different times, and the slots the early ones
// free stay unusable until four are free on the same CU -- a quarter of the slots between 240 and 440 us of a dragons
// launch (wave timelines, profiles/r02_experiments).  One-wave workgroups are admitted slot by slot: dragons 1080p fp64
// 0.634 -> 0.60 ms, teapot 0.355 -> 0.326, earth 0.227 -> 0.189.  (Measured slower in round 1 and again early in round 2 --
// "dispatch-bound" -- because every workgroup then added its ray counts with five atomics on one cache line, four times as
// many of them; the counts now go through per-wave slots, DFrame::work.)  The CSG variants, which still count through
// workgroup atomics, stay at RTC_BLOCK.  RTC_BLOCK counts units in the host code (4 units = 256 pixels of padding granularity).
#ifndef RTC_BLOCK_PLAIN
#define RTC_BLOCK_PLAIN 64
#endif
// __launch_bounds__ minimum waves per SIMD of the single-kernel form (register budget = 512 / waves).
// Measured on MI355X (dragons / teapot / cover at 1080p): fp32 is fastest at 4 (128 VGPRs).  fp64 without CSG, built without
// machine LICM (rtc_trace_f64.hip), needs ~150 registers: at 4 (128 VGPRs, 18 - 40 spilled) it is as fast as at 3 on the
// dragons and faster elsewhere (teapot 0.25 -> 0.22 ms); the variant compiled "for 2" uses 150 registers, spills nothing and
// runs at 3 waves per SIMD in effect, and launch() times the two for every view (Workspace::tune_ev).
#ifndef RTC_WAVES_F64
#define RTC_WAVES_F64 4
#endif
#ifndef RTC_WAVES_F32
#define RTC_WAVES_F32 4
#endif
// The variants for scenes with CSG objects or perturbed patterns ran 256-thread workgroups that counted rays through workgroup
// atomics until round 3 (a first attempt to give them the per-wave slots had faulted in round 2 and was shelved; rebuilt in
// round 3 with a bounds guard on the slot index and a host-side wait after every kernel, the same frames ran clean --
// DESIGN.md).  Development overrides to get the old form back: -DRTC_CSG_SLOTS=0 -DRTC_BLOCK_CSG=256.
#ifndef RTC_CSG_SLOTS
#define RTC_CSG_SLOTS 1
#endif
#ifndef RTC_BLOCK_CSG
#define RTC_BLOCK_CSG RTC_BLOCK_PLAIN
#endif
#ifndef RTC_BVH4_SORT_ANYHIT
#define RTC_BVH4_SORT_ANYHIT 0   // development: 1 = shadow rays and container passes sort the children of a four-wide node like nearest-hit queries do
#endif
#ifndef RTC_NODE_SPLIT
#define RTC_NODE_SPLIT 1      // development: 0 = one copy of a node visit's arithmetic behind both ways a record is fetched
#endif
#ifndef RTC_FOLD_RANGE
#define RTC_FOLD_RANGE 2      // development: 0 = a box is hit if n <= f, f >= tlo, n <= lim (four comparisons); 1 = max(n, tlo) <= min(f, lim); 2 = and the sort key is that max
#endif
#ifndef RTC_PUSH_BRANCHFREE
#define RTC_PUSH_BRANCHFREE 1 // development: 0 = a four-wide visit pushes each child under a branch of its own
#endif
#ifndef RTC_TRI_SPLIT
#define RTC_TRI_SPLIT 0       // development: 1 = the same split for triangle rows (measured: teapot -4 %, dragons 1080p fp64 +12 %: off)
#endif
#ifndef RTC_BVH_STACK
#define RTC_BVH_STACK 32      // per-lane traversal stack entries, in LDS
#endif
#define RTC_MAX_CSG_HITS 32   // hits gathered per CSG root per ray
#define RTC_MAX_BOUNCE_STACK 8
#define RTC_PAT_STACK 6
#define RTC_PERTURB_DEPTH 2
#define RTC_EMPTY_CHILD INT32_MIN

namespace rtc {

template <class R> struct V3 { R x, y, z; };
template <class R> __device__ __forceinline__ V3<R> mk(R x, R y, R z) { V3<R> v = {x, y, z}; return v; }
template <class R> __device__ __forceinline__ V3<R> operator+(V3<R> a, V3<R> b) { return mk<R>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <class R> __device__ __forceinline__ V3<R> operator-(V3<R> a, V3<R> b) { return mk<R>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <class R> __device__ __forceinline__ V3<R> operator*(V3<R> a, R s) { return mk<R>(a.x * s, a.y * s, a.z * s); }
template <class R> __device__ __forceinline__ V3<R> operator*(V3<R> a, V3<R> b) { return mk<R>(a.x * b.x, a.y * b.y, a.z * b.z); }
template <class R> __device__ __forceinline__ V3<R> neg(V3<R> a) { return mk<R>(-a.x, -a.y, -a.z); }
template <class R> __device__ __forceinline__ R dot(V3<R> a, V3<R> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class R> __device__ __forceinline__ V3<R> cross(V3<R> a, V3<R> b) {
    return mk<R>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float rsqrt_(float x) { return 1.0f / sqrtf(x); }
__device__ __forceinline__ double rsqrt_(double x) { return 1.0 / sqrt(x); }
__device__ __forceinline__ float sqrt_(float x) { return sqrtf(x); }
__device__ __forceinline__ double sqrt_(double x) { return sqrt(x); }
__device__ __forceinline__ float abs_(float x) { return fabsf(x); }
__device__ __forceinline__ double abs_(double x) { return fabs(x); }
__device__ __forceinline__ float floor_(float x) { return floorf(x); }
__device__ __forceinline__ double floor_(double x) { return floor(x); }
__device__ __forceinline__ float pow_(float x, float y) { return powf(x, y); }
__device__ __forceinline__ double pow_(double x, double y) { return pow(x, y); }
__device__ __forceinline__ float atan2_(float y, float x) { return atan2f(y, x); }
__device__ __forceinline__ double atan2_(double y, double x) { return atan2(y, x); }
__device__ __forceinline__ float acos_(float x) { return acosf(x); }
__device__ __forceinline__ double acos_(double x) { return acos(x); }
__device__ __forceinline__ float fmod_(float x, float y) { return fmodf(x, y); }
__device__ __forceinline__ double fmod_(double x, double y) { return fmod(x, y); }
__device__ __forceinline__ float rint_(float x) { return rintf(x); }
__device__ __forceinline__ double rint_(double x) { return rint(x); }
__device__ __forceinline__ float min_(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ double min_(double a, double b) { return fmin(a, b); }
__device__ __forceinline__ float max_(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ double max_(double a, double b) { return fmax(a, b); }
template <class R> __device__ __forceinline__ R inf_() { return (R)__builtin_huge_val(); }

// vector.py:170 normalize = v * (1 / |v|)
template <class R> __device__ __forceinline__ V3<R> normalize(V3<R> a) { return a * (R(1) / sqrt_(dot(a, a))); }
// renderer.py:893
template <class R> __device__ __forceinline__ V3<R> reflect(V3<R> in, V3<R> n) { return in - n * (R(2) * dot(in, n)); }
// np.isclose(a, b), rtol 1e-5, atol 1e-8
template <class R> __device__ __forceinline__ bool isclose(R a, R b) { return abs_(a - b) <= R(1e-8) + R(1e-5) * abs_(b); }
// numpy / Python floor-mod
template <class R> __device__ __forceinline__ R pymod(R a, R b) {
    R m = fmod_(a, b);
    if (m != R(0)) { if ((b < R(0)) != (m < R(0))) m += b; } else m = R(0);
    return m;
}
// Python max(a,b,c)/min(a,b,c) comparison chains (NaN of 0*inf survives in first position)
template <class R> __device__ __forceinline__ R pymax3(R a, R b, R c) { R m = a; if (b > m) m = b; if (c > m) m = c; return m; }
template <class R> __device__ __forceinline__ R pymin3(R a, R b, R c) { R m = a; if (b < m) m = b; if (c < m) m = c; return m; }

template <class R> struct Ray { V3<R> o, d; };

template <class R> __device__ __forceinline__ V3<R> xpoint(const R* __restrict__ m, V3<R> p) {
    return mk<R>(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
                 m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
}
template <class R> __device__ __forceinline__ V3<R> xvector(const R* __restrict__ m, V3<R> p) {
    return mk<R>(m[0] * p.x + m[1] * p.y + m[2] * p.z, m[4] * p.x + m[5] * p.y + m[6] * p.z,
                 m[8] * p.x + m[9] * p.y + m[10] * p.z);
}
template <class R> __device__ __forceinline__ V3<R> xtvector(const R* __restrict__ m, V3<R> p) {
    return mk<R>(m[0] * p.x + m[4] * p.y + m[8] * p.z, m[1] * p.x + m[5] * p.y + m[9] * p.z,
                 m[2] * p.x + m[6] * p.y + m[10] * p.z);
}

// ------------------------------------------------------------------------------------------
// device scene (struct of device pointers; one instance per precision)
// ------------------------------------------------------------------------------------------
struct alignas(64) BvhNode { float box[12]; int32_t child[2]; int32_t pad[2]; };
// the triangle BVH's four-wide form (rtc_bvh.hpp, collapse_bvh4): four child boxes (lo xyz, hi xyz each, padded fp32) and four
// child references in one 128-byte record = one L2 line per visit
// src[k]: where child k's box lives in the binary tree the wide one was collapsed from, (binary node << 1) | side, -1 = none
// (refits recompute the binary tree's boxes and copy them over, bvh4_refresh_kernel)
struct alignas(128) BvhNode4 { float box[24]; int32_t child[4]; int32_t src[4]; };

template <class R> struct DScene {
    int32_t n_leaves;
    const int32_t* leaf_kind; const int32_t* leaf_xform; const int32_t* leaf_material;
    const int32_t* leaf_tri;       // triangles: row of tri_geom / tri_norm, or -2 - row of mesh_norm (instanced); cylinders / cones: row of the two tables below
    const int32_t* leaf_closed; const R* leaf_minmax;       // [cylinders + cones], [..][2]
    const R* xform;
    const R* tri_geom; const R* tri_norm; const int32_t* tri_leaf;
    int32_t n_bvh_nodes;           // 0 = no bounded items
    // one 64-byte record per node = one cache line per visit: boxes of both children in fp32 (in
    // BOTH precisions; padded outward, see pad_boxes) and the two child references
    // (>= 0 inner node; < 0: ~c = first<<4 | (count-1); RTC_EMPTY_CHILD = none)
    const BvhNode* nodes;
    const int32_t* bvh_prims;      // prim codes in BVH leaf order (analytic leaves and CSG roots only)
    // triangles have a BVH of their own whose leaves index tb_* directly (no prim-code indirection):
    float obox[6], tbox[6];        // the (padded) box of the whole object BVH and of the whole triangle BVH: where far rays are re-based
    float far_origin;              // 2 x the largest coordinate of any box: a ray that starts beyond it is re-based (traverse_bvh)
    int32_t n_tnodes;
    const BvhNode* tnodes;
    const BvhNode4* tnodes4;       // the same tree four-wide: what rays walk (tnodes is what builds and refits work on)
    const R* tb_geom;              // [n_tris][12] tri_geom rows in BVH leaf order
    const int32_t* tb_leaf;        // [n_tris] leaf id, bit 31 set = does not cast shadows
    // instanced meshes (rtc_scene_desc::inst): one four-wide tree and one set of rows per mesh, in the mesh's object space
    int32_t n_inst;
    const int32_t* inst;           // [n_inst][8] root node (mnodes4), xform row, first leaf, first row of mesh_norm, material, 0, 0, 0
    const float* inst_box;         // [n_inst][6] the (padded) box of the mesh's tree, object space: where far rays are re-based
    const float* inst_far;         // [n_inst] far_origin in the mesh's object space
    const BvhNode4* mnodes4;
    const R* mb_geom;              // [n_mesh_tris][12] rows in the order of their mesh's tree
    const int32_t* mb_leaf;        // [n_mesh_tris] row within the mesh = leaf - first leaf of the instance
    const R* mesh_norm;            // [n_mesh_tris][12] object-space vertex normals, row order
    int32_t n_unbounded; const int32_t* unbounded_prims;
    const int32_t* csg_root; const int32_t* csg_node; const R* csg_bbox;
    const R* mat; const int32_t* mat_shadow; const int32_t* mat_program;
    const int32_t* pat_type; const R* pat_inv; const R* pat_ab; const int32_t* pat_child;
    const R* pat_param; const R* pat_colors; const int32_t* pat_tex;
    const int32_t* prog_off; const int32_t* prog_len; const int32_t* instr;
    int32_t n_flat_prims;                  // > 0: the object BVH is one leaf of this many prims, walked as a plain list
    const int64_t* tex_off; const int32_t* tex_w; const int32_t* tex_h; const R* tex_scale; const uint8_t* tex_data;
    int32_t n_lights;
    int32_t refraction_matters;    // 0 = every material's refractive index is 1: n1 = n2 = 1 at every hit, no container pass needed
    const int32_t* light_type; const R* light_geom; const int32_t* light_steps; const R* light_intensity;
};

template <class R> struct DCamera {
    int32_t hsize, vsize;
    R half_width, half_height, pixel_size;
    R inv[12];
};

struct DFrame {
    int32_t n_tiles;             // tiles to render
    int32_t tile_size, tiles_x;
    const int32_t* tile_ids;     // NULL: tile k = k, output in [x][y][c]; else compact tile layout
    double* rgb; int32_t* hit_id; unsigned long long* rays;
    // `work`: in the PROF variants the eight work counters.  In the other variants without CSG, if not NULL: the slots for ray
    // counts without same-address atomics -- wave w of the grid stores its five counts (primary, shadow, reflection,
    // refraction, shadow queries answered without a ray) as 32-bit numbers at ((uint32_t*)work)[8 w .. 8 w + 4] and
    // sum_rays_kernel adds them to `rays` after the launch.  NULL: every workgroup adds its counts to `rays` itself (40 000
    // atomics on one cache line per 1080p frame, which the memory side serialises: 25 us of a 0.2 ... 0.6 ms frame).
    unsigned long long* work;
    int32_t n_slots;             // how many waves `work` has a slot for, when it holds slots (a wave beyond them adds its counts with atomics)
    int32_t max_depth;
    uint64_t seed;
    // Launch order of the pixel kernels: unit = 64 work items = one 8x8 pixel block.  The k-th wave of the
    // grid renders entry order[k] = unit * 32 + part (workgroups are dispatched in index order, so the first
    // entries start first): part 0 = the whole block, 1..4 = one 4x4 quadrant of it, 5..20 = one 2x2 piece
    // (the dearest blocks are spread over 4 or 16 waves when the GPU has slots to spare), -1 = nothing.
    int32_t n_units, n_entries;
    const int32_t* order;        // [n_entries], or NULL = units 0, 1, 2, ... whole
    uint32_t* cost;              // [n_units] wall-clock ticks spent on the unit (parts add up; zeroed by the host), or NULL
    // A part has 16 or 4 pixels for 64 lanes.  With `replicas` set every pixel of a part is traced by 4 (16) adjacent lanes
    // that do the same thing -- except in the light loop, where they share the shadow rays out among themselves
    // (shade_ray).  Without it the spare lanes of a part are idle.  The host sets it unless the frame counts work or the
    // scene's lights have more than 128 cells together.
    int32_t replicas;
};

// the unit of this wave (-1 if there is none), which of its lanes take part, the pixel of the 8x8 block each lane has
// (0..63, row-major) and which replica of that pixel the lane is (rep of nrep: 0 of 1 unless the part is replicated)
__device__ __forceinline__ long long wave_unit(const DFrame& F, bool& lane_on, int& pixel_lane, int& rep, int& nrep, int waves_per_block = RTC_BLOCK / 64) {
    long long entry = (long long)blockIdx.x * waves_per_block + (threadIdx.x >> 6);
    lane_on = true;
    pixel_lane = threadIdx.x & 63; rep = 0; nrep = 1;
    if (!F.order) return entry < F.n_units ? entry : -1;
    if (entry >= F.n_entries) return -1;
    int e = F.order[entry];
    if (e < 0) return -1;
    int u = e >> 5, part = e & 31;
    if (u >= F.n_units || part > 20) return -1;
    const int lane = threadIdx.x & 63, lx = lane & 7, ly = lane >> 3;
    if (part != 0 && F.replicas) {
        int bx, by;
        if (part >= 5) { const int q = part - 5, p = lane >> 4; rep = lane & 15; nrep = 16; bx = (q & 3) * 2 + (p & 1); by = (q >> 2) * 2 + (p >> 1); }
        else { const int q = part - 1, p = lane >> 2; rep = lane & 3; nrep = 4; bx = (q & 1) * 4 + (p & 3); by = (q >> 1) * 4 + (p >> 2); }
        pixel_lane = by * 8 + bx;
    } else if (part >= 5) lane_on = (lx >> 1) == ((part - 5) & 3) && (ly >> 1) == ((part - 5) >> 2);
    else if (part != 0) lane_on = (((ly >> 2) << 1) | (lx >> 2)) == part - 1;
    return u;
}

// c: rays traced (primary, shadow, reflection, refraction).  w (PROF builds only): BVH node
// visits (2 box tests each), triangle tests, analytic leaf tests, CSG root evaluations, shaded
// hits, light samples shaded, container passes, pattern-node evaluations.
// W_NODE4: how many of the W_NODE visits were to four-wide (triangle BVH) nodes, 4 box tests each (kept outside rtc_stats::work:
// rays_dev[13]).
enum { W_NODE = 0, W_TRI = 1, W_LEAF = 2, W_CSG = 3, W_SHADE = 4, W_LIGHT = 5, W_CONTAINER = 6, W_PATTERN = 7, W_NODE4 = 8, W_COUNT = 9 };
// CSG = false compiles the CSG branches out of every visitor (scenes without CSG objects then
// carry none of its per-lane hit lists) and the Perlin noise of perturbed patterns out of the pattern
// interpreter: the host picks the CSG = true variants for scenes that have either.
// FAR = the camera is farther from the scene than the boxes' padding covers: rays are re-based where they enter a tree
// (traverse_bvh); the host picks it from the camera alone, since every other ray starts on a surface of the scene.
// STREAM = some CSG root of the scene is evaluated list-free (csg_stream): only those scenes' variants carry that code.
// REPL = the launch order has split parts whose pixels are traced by several lanes each (DFrame::replicas): only those
// launches' variants carry the replicated light loop (inlined into every variant it cost the fp64 kernel 18 more spilled
// registers: teapot +9 %, earth +15 % at 1080p, where nothing is split).
// INST = the scene holds instanced meshes (and no CSG objects: the scene compiler flattens them otherwise): only those scenes'
// variants carry the walk through instances (traverse) and what it keeps live.
template <bool PROF, bool CSG = true, bool FAR = false, bool STREAM = false, bool REPL = false, bool INST = false> struct Counters {
    static constexpr bool kCsg = CSG;
    static constexpr bool kInst = INST;
    static constexpr bool kFar = FAR;
    static constexpr bool kStream = CSG && STREAM;
    static constexpr bool kReplicas = REPL;
    static constexpr int kBlock = CSG ? RTC_BLOCK_CSG : RTC_BLOCK_PLAIN;      // threads per workgroup = stride of the per-lane traversal stack in LDS
    int rep, nrep;             // this lane is replica `rep` of `nrep` of its pixel (wave_unit); nrep is the same for the whole wave
    unsigned int c[5];         // primary, shadow, reflection, refraction rays traced; shadow rays not traced (light behind the surface)
    unsigned int w[PROF ? W_COUNT : 1];
#ifdef RTC_LANE_STATS   /* development build (tools/lane_stats.py): every counter counts WAVE-level executions, x 65536 */
    __device__ __forceinline__ void work(int k) { if (PROF) w[k] += 65536u / (unsigned)__popcll(__ballot(1)); }
#else
    __device__ __forceinline__ void work(int k) { if (PROF) w[k]++; }
#endif
};
#define RTC_EPS R(0.000001) /* renderer.py:15, shapes.py:8 */

// ------------------------------------------------------------------------------------------
// local intersections: each returns the number of hits written to t[] (unsorted)
// ------------------------------------------------------------------------------------------
// shapes.py:597-618 / 87-108
template <class R> __device__ __forceinline__ void check_axis(R origin, R direction, R lo, R hi, R& tmin, R& tmax) {
    R n0 = lo - origin, n1 = hi - origin, a, b;
    if (abs_(direction) >= RTC_EPS) { a = n0 / direction; b = n1 / direction; }
    else { a = n0 * inf_<R>(); b = n1 * inf_<R>(); }
    if (a > b) { tmin = b; tmax = a; } else { tmin = a; tmax = b; }
}
template <class R> __device__ __forceinline__ bool box_hit(const Ray<R>& r, const R* lo, const R* hi, R& tmin, R& tmax) {
    R xa, xb, ya, yb, za, zb;
    check_axis(r.o.x, r.d.x, lo[0], hi[0], xa, xb);
    check_axis(r.o.y, r.d.y, lo[1], hi[1], ya, yb);
    check_axis(r.o.z, r.d.z, lo[2], hi[2], za, zb);
    tmin = pymax3(xa, ya, za);
    tmax = pymin3(xb, yb, zb);
    return !(tmin > tmax);
}
// shapes.py:525-535
template <class R> __device__ __forceinline__ int sphere_hits(const Ray<R>& r, R* t) {
    R a = dot(r.d, r.d), b = R(2) * dot(r.d, r.o), c = dot(r.o, r.o) - R(1);
    R disc = b * b - R(4) * a * c;
    if (disc < R(0)) return 0;
    R sq = sqrt_(disc);
    t[0] = (-b - sq) / (R(2) * a);
    t[1] = (-b + sq) / (R(2) * a);
    return 2;
}
// shapes.py:550-555
template <class R> __device__ __forceinline__ int plane_hits(const Ray<R>& r, R* t) {
    if (abs_(r.d.y) < RTC_EPS) return 0;
    t[0] = -r.o.y / r.d.y;
    return 1;
}
// shapes.py:572-582
template <class R> __device__ __forceinline__ int cube_hits(const Ray<R>& r, R* t) {
    const R lo[3] = {R(-1), R(-1), R(-1)}, hi[3] = {R(1), R(1), R(1)};
    R tmin, tmax;
    if (!box_hit(r, lo, hi, tmin, tmax)) return 0;
    t[0] = tmin; t[1] = tmax;
    return 2;
}
// shapes.py:734-792 (cone = false) and 636-705 (cone = true, quirks kept)
template <class R, bool CONE> __device__ __forceinline__ int quadric_hits(const Ray<R>& r, R minimum, R maximum, bool closed, R* t) {
    int n = 0;
    R a = r.d.x * r.d.x + r.d.z * r.d.z;
    R b = R(2) * r.o.x * r.d.x + R(2) * r.o.z * r.d.z;
    R c = r.o.x * r.o.x + r.o.z * r.o.z;
    if (CONE) { a -= r.d.y * r.d.y; b -= R(2) * r.o.y * r.d.y; c -= r.o.y * r.o.y; } else { c -= R(1); }
    bool a0 = abs_(a) <= R(1e-8);   // np.isclose(a, 0)
    if (a0) {
        if (CONE) t[n++] = -c / (R(2) * b);  // even when b ~ 0 (shapes.py:647-650)
    } else {
        R disc = b * b - R(4) * a * c;
        if (disc < R(0)) return 0;
        R sq = sqrt_(disc);
        R t0 = (-b - sq) / (R(2) * a), t1 = (-b + sq) / (R(2) * a);
        if (CONE && sizeof(R) == 4) {
            // fp32 cone: a = dx^2 + dz^2 - dy^2 goes through zero for rays along the cone's side (every view into a cone has
            // them), where -b -+ sqrt(disc) cancels for one root and the division by a small `a` multiplies what is left of
            // it.  That root comes from c / q instead (q = -(b + sign(b) sqrt(disc)) / 2, the sum that does not cancel):
            // same roots, conditioned like a cylinder's.  fp64 stays literal.
            R q = R(-0.5) * (b + (b < R(0) ? -sq : sq));
            if (q != R(0)) { t0 = q / a; t1 = c / q; }
        }
        if (t0 > t1) { R tmp = t0; t0 = t1; t1 = tmp; }
        R y0 = r.o.y + t0 * r.d.y;
        if (minimum < y0 && y0 < maximum) t[n++] = t0;
        R y1 = r.o.y + t1 * r.d.y;
        if (minimum < y1 && y1 < maximum) t[n++] = t1;
    }
    if (closed && !(abs_(r.d.y) <= R(1e-8))) {
        R lim[2] = {minimum, maximum};
#pragma unroll
        for (int k = 0; k < 2; k++) {
            R tc = (lim[k] - r.o.y) / r.d.y;
            R x = r.o.x + tc * r.d.x, z = r.o.z + tc * r.d.z;
            R rad = CONE ? abs_(lim[k]) : R(1);   // |y_cap|, not y_cap^2 (shapes.py:689-693)
            if (!(x * x + z * z > rad)) t[n++] = tc;
        }
    }
    return n;
}
// A pointer every active lane holds the same value of, moved to scalar registers (inside a called function even a
// wave-uniform pointer lives in vector registers): the operand of a scalar load.
template <class T> __device__ __forceinline__ const T* uniform_ptr(const T* p) {
    const unsigned long long a = (unsigned long long)p;
    return reinterpret_cast<const T*>(((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                                      (unsigned int)__builtin_amdgcn_readfirstlane((int)a));
}
// shapes.py:812-830 / 860-878 on world-space data; row = p1, e1, e2, det_eps
template <class R, class ROW> __device__ __forceinline__ bool tri_test(const ROW& g, const Ray<R>& r, R& t, R& u, R& v) {
    V3<R> p1 = mk<R>(g[0], g[1], g[2]), e1 = mk<R>(g[3], g[4], g[5]), e2 = mk<R>(g[6], g[7], g[8]);
    V3<R> dir_cross_e2 = cross(r.d, e2);
    R det = dot(e1, dir_cross_e2);
    if (abs_(det) < g[9]) return false;
    R f = R(1) / det;
    V3<R> p1_to_origin = r.o - p1;
    u = f * dot(p1_to_origin, dir_cross_e2);
    if (u < R(0) || u > R(1)) return false;
    V3<R> origin_cross_e1 = cross(p1_to_origin, e1);
    v = f * dot(r.d, origin_cross_e1);
    if (v < R(0) || (u + v) > R(1)) return false;
    t = f * dot(e2, origin_cross_e1);
    return true;
}
template <class R> struct TriRow { R v[10]; __device__ __forceinline__ R operator[](int k) const { return v[k]; } };
template <class R> __device__ __forceinline__ bool tri_hit(const R* __restrict__ g, const Ray<R>& r, R& t, R& u, R& v, bool uniform = false) {
    // `uniform`: every active lane of the wave tests this same row (traverse_bvh4 found them at the same leaf): the row
    // comes through the scalar unit, like the node records.  (RTC_TRI_SPLIT: a copy of the arithmetic of its own for the
    // scalar row, reading it from scalar registers, as the node visits have -- measured slower on the dragons in fp64, off.)
    TriRow<R> row;
    if (uniform) {
        const R* g0 = uniform_ptr(g);
        if (sizeof(R) == 8) {
            typedef double d8v __attribute__((ext_vector_type(8)));
            typedef double d2v __attribute__((ext_vector_type(2)));
            d8v a; d2v b;
            // early-clobber outputs: both loads read the address pair, which must not be handed out as (part of) a destination
            asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx4 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a), "=&s"(b) : "s"(g0) : "memory");
#pragma unroll
            for (int k = 0; k < 8; k++) row.v[k] = (R)a[k];
            row.v[8] = (R)b[0]; row.v[9] = (R)b[1];
        } else {
            typedef float f8v __attribute__((ext_vector_type(8)));
            typedef float f2v __attribute__((ext_vector_type(2)));
            f8v a; f2v b;
            asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dwordx2 %1, %2, 0x20\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a), "=&s"(b) : "s"(g0) : "memory");
#pragma unroll
            for (int k = 0; k < 8; k++) row.v[k] = (R)a[k];
            row.v[8] = (R)b[0]; row.v[9] = (R)b[1];
        }
        if (RTC_TRI_SPLIT) return tri_test(row, r, t, u, v);
    } else {
#pragma unroll
        for (int k = 0; k < 10; k++) row.v[k] = g[k];
    }
    return tri_test(row, r, t, u, v);
}

// hits of one leaf (world-space ray in, t's out, unsorted); u,v only for triangles
template <class R> __device__ __forceinline__ int leaf_hits(const DScene<R>& S, int leaf, const Ray<R>& ray, R* t, R& u, R& v) {
    int kind = S.leaf_kind[leaf];
    if (kind == RTC_TRIANGLE) {
        return tri_hit(S.tri_geom + 12 * (size_t)S.leaf_tri[leaf], ray, t[0], u, v) ? 1 : 0;
    }
    const R* m = S.xform + 12 * (size_t)S.leaf_xform[leaf];
    if (kind == RTC_PLANE) {
        // the plane test reads only the y row of the object-space ray (shapes.py:546-552): the same two expressions
        // xpoint / xvector evaluate for it, without the other four
        Ray<R> ry = {mk<R>(R(0), m[4] * ray.o.x + m[5] * ray.o.y + m[6] * ray.o.z + m[7], R(0)),
                     mk<R>(R(0), m[4] * ray.d.x + m[5] * ray.d.y + m[6] * ray.d.z, R(0))};
        return plane_hits(ry, t);
    }
    Ray<R> r = {xpoint(m, ray.o), xvector(m, ray.d)};   // shapes.py:271-273, 1447-1448
    if (kind == RTC_CUBE) return cube_hits(r, t);
    // fp32 only: solve the quadratic from the point of the ray nearest the object's origin and add
    // the shift back.  Same roots; it keeps b^2 - 4ac from cancelling when the eye is far from a
    // small object (an fp64 solve does not need it and stays literal).
    R shift = R(0);
    if (sizeof(R) == 4) {
        shift = -dot(r.o, r.d) / dot(r.d, r.d);
        r.o = r.o + r.d * shift;
    }
    int n;
    if (kind == RTC_SPHERE) n = sphere_hits(r, t);
    else {
        const int q = S.leaf_tri[leaf];             // a cylinder's / cone's row of the minimum / maximum / closed tables
        if (kind == RTC_CYLINDER) n = quadric_hits<R, false>(r, S.leaf_minmax[2 * q], S.leaf_minmax[2 * q + 1], S.leaf_closed[q] != 0, t);
        else n = quadric_hits<R, true>(r, S.leaf_minmax[2 * q], S.leaf_minmax[2 * q + 1], S.leaf_closed[q] != 0, t);
    }
    if (sizeof(R) == 4)
        for (int k = 0; k < n; k++) t[k] += shift;
    return n;
}

// ------------------------------------------------------------------------------------------
// CSG (shapes.py:363-422): hits of all leaves of a root, sorted, then filtered node by node
// (children before parents, so a parent only sees what its children let through).
// ------------------------------------------------------------------------------------------
template <class R> struct CsgHits { R t[RTC_MAX_CSG_HITS]; int leaf[RTC_MAX_CSG_HITS]; int n; };

__device__ __forceinline__ bool csg_allowed(int op, bool lhit, bool inl, bool inr) {
    if (op == RTC_CSG_UNION) return (lhit && !inr) || (!lhit && !inl);
    if (op == RTC_CSG_INTERSECT) return (lhit && inr) || (!lhit && inl);
    return (lhit && !inr) || (!lhit && inl);
}

template <class R> __device__ __noinline__ void csg_hits(const DScene<R>& S, int root, const Ray<R>& ray, CsgHits<R>& H) {
    const int32_t* rt = S.csg_root + 4 * root;
    H.n = 0;
    for (int leaf = rt[2]; leaf < rt[3]; leaf++) {
        R t[4], u, v;
        int n = leaf_hits(S, leaf, ray, t, u, v);
        for (int k = 0; k < n && H.n < RTC_MAX_CSG_HITS; k++) {   // stable insertion by t
            int j = H.n++;
            while (j > 0 && H.t[j - 1] > t[k]) { H.t[j] = H.t[j - 1]; H.leaf[j] = H.leaf[j - 1]; j--; }
            H.t[j] = t[k]; H.leaf[j] = leaf;
        }
    }
    for (int nd = rt[0]; nd < rt[0] + rt[1]; nd++) {
        const int32_t* c = S.csg_node + 5 * nd;
        int op = c[0], l0 = c[1], l1 = c[2], l2 = c[3];
        const R* m = S.xform + 12 * (size_t)c[4];
        Ray<R> r = {xpoint(m, ray.o), xvector(m, ray.d)};
        R tmin, tmax;
        bool boxed = box_hit(r, S.csg_bbox + 6 * nd, S.csg_bbox + 6 * nd + 3, tmin, tmax);   // shapes.py:367
        bool inl = false, inr = false;
        int w = 0;
        for (int i = 0; i < H.n; i++) {
            int leaf = H.leaf[i];
            bool keep = true;
            if (leaf >= l0 && leaf < l2) {
                bool lhit = leaf < l1;
                keep = boxed && csg_allowed(op, lhit, inl, inr);
                if (lhit) inl = !inl; else inr = !inr;
            }
            if (keep) { H.t[w] = H.t[i]; H.leaf[w] = leaf; w++; }
        }
        H.n = w;
    }
}

// A root whose leaves could produce more than RTC_MAX_CSG_HITS hits on one ray (two groups of a dozen spheres, a CSG
// over a mesh) has RTC_CSG_STREAMED set in its node count and is evaluated without a list: its intersections are
// enumerated in sorted order by repeated minimum scans over its leaves -- key (t, leaf, position in the leaf's own hit
// list), the order a stable sort of the leaves' lists gives -- and each one is passed up the chain of CSG nodes above
// its leaf, every node keeping its two `in left` / `in right` flags (shapes.py:375-388) in a bit mask.  f(t, leaf) is
// called for every intersection the root lets through, in order, until it returns true.  Costs (hits + 1) x leaves
// leaf tests per ray -- the reference tests every leaf too -- and no memory; at most RTC_MAX_CSG_STREAM_NODES nodes.
#define RTC_CSG_STREAMED 0x40000000
#define RTC_MAX_CSG_STREAM_NODES 64
#define RTC_CSG_OPEN 32       /* leaves of one streamed root a ray can be inside of at once (container walk) */
template <class R> __device__ __forceinline__ bool csg_key_less(R t0, int l0, int k0, R t1, int l1, int k1) {
    return t0 < t1 || (t0 == t1 && (l0 < l1 || (l0 == l1 && k0 < k1)));
}
template <class R, class F> __device__ __noinline__ void csg_stream(const DScene<R>& S, int root, const Ray<R>& ray, F& f) {
    const int32_t* rt = S.csg_root + 4 * root;
    const int n0 = rt[0], nn = rt[1] & ~RTC_CSG_STREAMED;
    unsigned long long boxed = 0, inl = 0, inr = 0;
    for (int j = 0; j < nn; j++) {
        const R* m = S.xform + 12 * (size_t)S.csg_node[5 * (n0 + j) + 4];
        Ray<R> r = {xpoint(m, ray.o), xvector(m, ray.d)};
        R tmin, tmax;
        if (box_hit(r, S.csg_bbox + 6 * (n0 + j), S.csg_bbox + 6 * (n0 + j) + 3, tmin, tmax)) boxed |= 1ULL << j;   // shapes.py:367
    }
    R pt = R(0); int pl = -1, pk = -1;
    bool started = false;
    while (true) {
        R bt = R(0); int bl = -1, bk = -1;
        for (int leaf = rt[2]; leaf < rt[3]; leaf++) {
            R t[4], u, v;
            const int n = leaf_hits(S, leaf, ray, t, u, v);
            for (int k = 0; k < n; k++) {
                if (!(t[k] == t[k])) continue;                                   // NaN (0 * inf, 0 / 0): no place in any order
                if (started && !csg_key_less(pt, pl, pk, t[k], leaf, k)) continue;
                if (bl < 0 || csg_key_less(t[k], leaf, k, bt, bl, bk)) { bt = t[k]; bl = leaf; bk = k; }
            }
        }
        if (bl < 0) return;
        started = true; pt = bt; pl = bl; pk = bk;
        bool keep = true;
        for (int j = 0; j < nn && keep; j++) {                                   // children before parents
            const int32_t* c = S.csg_node + 5 * (n0 + j);
            if (bl < c[1] || bl >= c[3]) continue;
            const bool lhit = bl < c[2];
            const unsigned long long bit = 1ULL << j;
            keep = (boxed & bit) != 0 && csg_allowed(c[0], lhit, (inl & bit) != 0, (inr & bit) != 0);
            if (lhit) inl ^= bit; else inr ^= bit;
        }
        if (keep && f(bt, bl)) return;
    }
}
__device__ __forceinline__ bool csg_streamed(const int32_t* csg_root, int root) { return (csg_root[4 * root + 1] & RTC_CSG_STREAMED) != 0; }

// ------------------------------------------------------------------------------------------
// queries over the whole scene
// ------------------------------------------------------------------------------------------
struct PrimCode {
    int kind, idx; bool noshadow;
    __device__ __forceinline__ explicit PrimCode(int32_t c)
        : kind((int)((uint32_t)c >> 28)), idx((int)((uint32_t)c & 0x07FFFFFFu)), noshadow((((uint32_t)c >> 27) & 1u) != 0) {}
};

// What the box tests of a walk need of the ray: reciprocal direction, -origin / direction, the near bound (fp32, see traverse_bvh)
struct SlabRay { float ix, iy, iz, ox, oy, oz, tlo32; };
// The ray's share of that, made once per ray and used by every walk of it (object tree, triangle tree, the box around a flat
// list): slab distance (b - o) / d as one fused multiply-add b * (1/d) + (-o/d); a direction component of 0 gets a huge finite
// reciprocal instead of inf, so that no 0 * inf appears (the sign of b - o then decides, as it should, and the boxes' outward
// padding covers an origin that sits on the slab plane).  (Each walk used to make its own: three sets of reciprocals a ray.)
struct RayInv { float ix, iy, iz, ox, oy, oz; };
template <class R> __device__ __forceinline__ RayInv ray_inv(const Ray<R>& ray) {
    RayInv q;
    q.ix = 1.0f / (float)ray.d.x; q.iy = 1.0f / (float)ray.d.y; q.iz = 1.0f / (float)ray.d.z;
    if (!(fabsf(q.ix) <= 1e30f)) q.ix = copysignf(1e30f, (float)ray.d.x);
    if (!(fabsf(q.iy) <= 1e30f)) q.iy = copysignf(1e30f, (float)ray.d.y);
    if (!(fabsf(q.iz) <= 1e30f)) q.iz = copysignf(1e30f, (float)ray.d.z);
    q.ox = -(float)ray.o.x * q.ix; q.oy = -(float)ray.o.y * q.iy; q.oz = -(float)ray.o.z * q.iz;
    return q;
}
// entry and exit distance of the ray in one box (lo xyz, hi xyz), with the outward slack of traverse_bvh
__device__ __forceinline__ void slab_test(float lx, float ly, float lz, float hx, float hy, float hz, const SlabRay& s, float& n, float& f) {
    const float x0 = fmaf(lx, s.ix, s.ox), x1 = fmaf(hx, s.ix, s.ox);
    const float y0 = fmaf(ly, s.iy, s.oy), y1 = fmaf(hy, s.iy, s.oy);
    const float z0 = fmaf(lz, s.iz, s.oz), z1 = fmaf(hz, s.iz, s.oz);
    n = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
    f = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
    n -= fabsf(n) * 1e-5f;
    f += fabsf(f) * 1e-5f;
}
// One visit to a binary node: both child boxes tested, the nearer child that is hit returned (RTC_EMPTY_CHILD: none), the
// other pushed.  The record's 12 box values and two child references are taken wherever they arrived: each way of fetching
// a record (vector loads; one scalar load when the wave is at one node) calls its own copy of this, so that the scalar one
// reads its operands from scalar registers -- merged behind the fetch, every scalar record was first moved to vector
// registers (16 v_mov a binary visit, 28 a four-wide one).
struct NodeRec2 { float b[12]; int c0, c1; };
template <int STRIDE> __device__ __forceinline__ int bvh2_visit(const NodeRec2& r, const SlabRay& s, float lim, int* stack, int& sp) {
    float n0, f0, n1, f1;
    slab_test(r.b[0], r.b[1], r.b[2], r.b[3], r.b[4], r.b[5], s, n0, f0);
    slab_test(r.b[6], r.b[7], r.b[8], r.b[9], r.b[10], r.b[11], s, n1, f1);
    // (`&`, not `&&`: the comparisons are data, not control flow -- with `&&` the compiler put every box test under a branch of its own)
    // a box is entered if n <= f, f >= tlo and n <= lim: with tlo <= lim (always: an empty range has nothing to find) that is
    // max(n, tlo) <= min(f, lim) -- one comparison and no mask arithmetic instead of three and two (measured: -4 ... 5 % of a frame)
#if RTC_FOLD_RANGE
    const bool h0 = (r.c0 != RTC_EMPTY_CHILD) & (fmaxf(n0, s.tlo32) <= fminf(f0, lim));
    const bool h1 = (r.c1 != RTC_EMPTY_CHILD) & (fmaxf(n1, s.tlo32) <= fminf(f1, lim));
#else
    const bool h0 = (r.c0 != RTC_EMPTY_CHILD) & (n0 <= f0) & (f0 >= s.tlo32) & (n0 <= lim);
    const bool h1 = (r.c1 != RTC_EMPTY_CHILD) & (n1 <= f1) & (f1 >= s.tlo32) & (n1 <= lim);
#endif
    if (h0 && h1) {
        const bool swap = n1 < n0;
        stack[sp * STRIDE] = swap ? r.c0 : r.c1; sp++;
        return swap ? r.c1 : r.c0;
    }
    return h0 ? r.c0 : h1 ? r.c1 : RTC_EMPTY_CHILD;
}
// One visit to a four-wide node.  UNORDERED (any-hit queries, passes with a fixed far bound): a child that is hit is
// entered, the others pushed.  Otherwise the children that are hit are visited nearest first: their entry distances
// (clamped at 0, the child's slot in the two lowest mantissa bits) and references are sorted together by a five-exchange
// network -- branch-free; picking the references by slot after sorting the keys alone compiled to nested branches -- the
// nearest is returned, the others pushed farthest first.
struct NodeRec4 { float b[24]; int c[4]; };
template <int STRIDE, bool UNORDERED> __device__ __forceinline__ int bvh4_visit(const NodeRec4& r, const SlabRay& s, float lim, int* stack, int& sp) {
    unsigned int key[4];
    bool hit[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float n, f;
        slab_test(r.b[6 * k], r.b[6 * k + 1], r.b[6 * k + 2], r.b[6 * k + 3], r.b[6 * k + 4], r.b[6 * k + 5], s, n, f);
#if RTC_FOLD_RANGE >= 2
        // (the folded test of bvh2_visit.  The key is the clamped entry distance itself: tlo is 0 for the walks that sort; where it
        // is negative -- rays re-based for a far camera -- the order is off for boxes entered behind it, the result is not)
        const float m = fmaxf(n, s.tlo32);
        hit[k] = (r.c[k] != RTC_EMPTY_CHILD) & (m <= fminf(f, lim));
        key[k] = hit[k] ? ((__float_as_uint(m) & ~3u) | (unsigned)k) : 0xFFFFFFFFu;
#elif RTC_FOLD_RANGE
        hit[k] = (r.c[k] != RTC_EMPTY_CHILD) & (fmaxf(n, s.tlo32) <= fminf(f, lim));
        key[k] = hit[k] ? ((__float_as_uint(fmaxf(n, 0.0f)) & ~3u) | (unsigned)k) : 0xFFFFFFFFu;
#else
        hit[k] = (r.c[k] != RTC_EMPTY_CHILD) & (n <= f) & (f >= s.tlo32) & (n <= lim);
        key[k] = hit[k] ? ((__float_as_uint(fmaxf(n, 0.0f)) & ~3u) | (unsigned)k) : 0xFFFFFFFFu;
#endif
    }
    if (UNORDERED) {
        int next = RTC_EMPTY_CHILD;
#if RTC_PUSH_BRANCHFREE
#pragma unroll
        for (int k = 3; k >= 0; k--) {
            stack[sp * STRIDE] = next;
            sp += hit[k] & (next != RTC_EMPTY_CHILD);
            next = hit[k] ? r.c[k] : next;
        }
#else
#pragma unroll
        for (int k = 3; k >= 0; k--)
            if (hit[k]) {
                if (next != RTC_EMPTY_CHILD) { stack[sp * STRIDE] = next; sp++; }
                next = r.c[k];
            }
#endif
        return next;
    }
    int ref[4] = {r.c[0], r.c[1], r.c[2], r.c[3]};
#define RTC_CE(i, j) { const bool sw_ = key[j] < key[i]; const unsigned int ki_ = key[i], kj_ = key[j]; const int ri_ = ref[i], rj_ = ref[j]; \
                       key[i] = sw_ ? kj_ : ki_; key[j] = sw_ ? ki_ : kj_; ref[i] = sw_ ? rj_ : ri_; ref[j] = sw_ ? ri_ : rj_; }
    RTC_CE(0, 1) RTC_CE(2, 3) RTC_CE(0, 2) RTC_CE(1, 3) RTC_CE(1, 2)
#undef RTC_CE
#if RTC_PUSH_BRANCHFREE
    // pushes without branches: every store lands on the first free row (sp <= the tree's stack need < RTC_BVH_STACK rows, checked
    // on the host when the tree is made), the pointer moves on only past a child that was hit, and a store that was not wanted is
    // overwritten by the next push (measured: fp64 -1.5 %, fp32 -6 % on the dragons)
    stack[sp * STRIDE] = ref[3]; sp += key[3] != 0xFFFFFFFFu;
    stack[sp * STRIDE] = ref[2]; sp += key[2] != 0xFFFFFFFFu;
    stack[sp * STRIDE] = ref[1]; sp += key[1] != 0xFFFFFFFFu;
    return key[0] == 0xFFFFFFFFu ? RTC_EMPTY_CHILD : ref[0];
#else
    if (key[0] == 0xFFFFFFFFu) return RTC_EMPTY_CHILD;
    if (key[3] != 0xFFFFFFFFu) { stack[sp * STRIDE] = ref[3]; sp++; }
    if (key[2] != 0xFFFFFFFFu) { stack[sp * STRIDE] = ref[2]; sp++; }
    if (key[1] != 0xFFFFFFFFu) { stack[sp * STRIDE] = ref[1]; sp++; }
    return ref[0];
#endif
}

// Visitor V: bool prim(code) -> true = stop the traversal; R limit() = current far bound.
// Boxes are visited when their slab interval overlaps (tlo, limit()).
// The OBJECT BVH (binary; leaves hold prim codes -> vis.prim); the triangle BVH is walked four-wide, traverse_bvh4.
// Returns true if the visitor stopped it.
// FARW: this walk re-bases rays that start far outside the tree's box (see below)
template <class R, class V, bool FARW>
__device__ __forceinline__ bool traverse_bvh(const DScene<R>& S, const Ray<R>& ray, const RayInv& ri, R tlo, V& vis, int* stack /* LDS, stride = threads per workgroup */) {
    constexpr int kStride = std::remove_reference_t<decltype(vis.cnt)>::kBlock;
    const BvhNode* __restrict__ nodes = S.nodes;
    // Box tests run in fp32 whatever R is: the boxes are padded outward by 2e-5 (relative, >= 2e-5
    // absolute) on the host, which covers rounding the ray to fp32, and the slab distances get a
    // 1e-5 relative slack before every comparison -- the test can only accept MORE than an exact
    // one would, never less, so no leaf the leaf test would accept is culled.
    // (the reciprocal direction and -origin / direction: ray_inv)
    const float ix = ri.ix, iy = ri.iy, iz = ri.iz;
    // The fp32 slab arithmetic rounds relative to the ray's ORIGIN.  Boxes are padded (pad_boxes) by at least 8e-7 of
    // the largest coordinate in the scene, which covers rounding an origin of up to twice that size; a ray that starts
    // farther out (a camera thousands of units from a small mesh) would see the boxes displaced by more than their
    // padding, so it is re-based, in R, at (just before) the point where it enters the tree's own box: box tests then
    // run on origin' = origin + base * direction with distances measured from there.  `base` only has to be near the
    // entry point, not exact, so it is found in fp32.  Only the kernel variants for far cameras (Counters::kFar) carry this:
    // every other ray of a frame starts on a surface of the scene.  base = 0 changes nothing.
    float base32 = 0.0f, lim_pad = 0.0f;
    float ox = ri.ox, oy = ri.oy, oz = ri.oz;
    float tlo32 = (float)tlo;
    if (FARW && fmaxf(fmaxf(fabsf((float)ray.o.x), fabsf((float)ray.o.y)), fabsf((float)ray.o.z)) > S.far_origin) {
        const float* rb = S.obox;
        const float px = (float)ray.o.x, py = (float)ray.o.y, pz = (float)ray.o.z;
        const float x0 = (rb[0] - px) * ix, x1 = (rb[3] - px) * ix, y0 = (rb[1] - py) * iy, y1 = (rb[4] - py) * iy, z0 = (rb[2] - pz) * iz, z1 = (rb[5] - pz) * iz;
        float enter = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
        enter -= 1e-3f * fabsf(enter) + 1e-3f;
        enter = fmaxf(enter, (float)tlo);
        if (fabsf(enter) < 1e30f) {
            const R base = R(enter);
            base32 = enter; lim_pad = 2.4e-7f * fabsf(enter);
            ox = -(float)(ray.o.x + ray.d.x * base) * ix; oy = -(float)(ray.o.y + ray.d.y * base) * iy; oz = -(float)(ray.o.z + ray.d.z * base) * iz;
            tlo32 = (float)(tlo - base);
        }
    }
    int sp = 0;
    int node = 0;
    while (true) {
        if (node >= 0) {
            vis.cnt.work(W_NODE);
#ifdef RTC_LANE_STATS_DETAIL   /* one-off measurement (tools/lane_stats.py): pattern_evals = wave-uniform binary visits */
            if (__builtin_amdgcn_ballot_w64(node != __builtin_amdgcn_readfirstlane(node)) == 0ull) vis.cnt.work(W_PATTERN);
#endif
            // every lane of the wave at the same node (the top of the tree, for the rays of one pixel block): the record
            // comes through the scalar unit (one 64-byte scalar load) and leaves the vector memory pipe alone
            float lim = (float)vis.limit();
            if (FARW) lim -= base32;                    // (float)limit is off by 6e-8 |limit| <= 6e-8 (|base| + |lim|): lim_pad + the relative slack
            lim += fabsf(lim) * 1e-6f;
            if (FARW) lim += lim_pad;
            const SlabRay sr = {ix, iy, iz, ox, oy, oz, tlo32};
            int next;
            NodeRec2 r;
            const int node0 = __builtin_amdgcn_readfirstlane(node);
            if (__builtin_amdgcn_ballot_w64(node != node0) == 0ull) {
                typedef float f16v __attribute__((ext_vector_type(16)));
                f16v rec;
                const BvhNode* p0 = uniform_ptr(nodes + node0);
                asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rec) : "s"(p0) : "memory");
                const NodeRec2 rs = {{rec[0], rec[1], rec[2], rec[3], rec[4], rec[5], rec[6], rec[7], rec[8], rec[9], rec[10], rec[11]},
                                     __float_as_int(rec[12]), __float_as_int(rec[13])};
                if (RTC_NODE_SPLIT) next = bvh2_visit<kStride>(rs, sr, lim, stack, sp); else r = rs;
            } else {
                const float4* q = reinterpret_cast<const float4*>(nodes + node);
                const float4 q0 = q[0], q1 = q[1], q2 = q[2];
                const float2 q3 = *reinterpret_cast<const float2*>(q + 3);
                const NodeRec2 rv = {{q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w}, __float_as_int(q3.x), __float_as_int(q3.y)};
                if (RTC_NODE_SPLIT) next = bvh2_visit<kStride>(rv, sr, lim, stack, sp); else r = rv;
            }
            if (!RTC_NODE_SPLIT) next = bvh2_visit<kStride>(r, sr, lim, stack, sp);
            if (next != RTC_EMPTY_CHILD) { node = next; continue; }
        } else {
            int code = ~node;
            int first = code >> 4, count = (code & 15) + 1;
            for (int i = 0; i < count; i++)
                if (vis.prim(S.bvh_prims[first + i])) return true;
        }
        if (sp == 0) return false;
        sp--;
        node = stack[sp * kStride];
    }
}

// The triangle BVH in its four-wide form.  Same conservative fp32 slab test per child box as traverse_bvh; the children
// that are hit are visited nearest first: their entry distances (clamped at 0, the child's slot in the two lowest mantissa
// bits) are sorted as unsigned integers by a five-exchange network, the nearest is entered, the others are pushed farthest
// first.  The order changes no result (every child that is hit is visited unless the limit has shrunk below its entry
// distance by then -- the visitor's own comparison decides, as in the binary walk).
// INST: the tree of an instanced mesh (`ray` already in its object space; triangles go to vis.tri<true>): nodes, root and the
// box / far_origin for re-basing are the instance's.
template <class R, class V, bool INST = false, bool FARW = V::Cnt::kFar>
__device__ __forceinline__ bool traverse_bvh4(const DScene<R>& S, const Ray<R>& ray, const RayInv& ri, R tlo, V& vis, int* stack /* LDS, stride = threads per workgroup */,
                                              int inst = 0, int root = 0) {
    constexpr int kStride = std::remove_reference_t<decltype(vis.cnt)>::kBlock;
    const BvhNode4* __restrict__ nodes = INST ? S.mnodes4 : S.tnodes4;
    const float ix = ri.ix, iy = ri.iy, iz = ri.iz;
    float base32 = 0.0f, lim_pad = 0.0f;
    float ox = ri.ox, oy = ri.oy, oz = ri.oz;
    float tlo32 = (float)tlo;
    if (FARW && fmaxf(fmaxf(fabsf((float)ray.o.x), fabsf((float)ray.o.y)), fabsf((float)ray.o.z)) > (INST ? S.inst_far[inst] : S.far_origin)) {      // see traverse_bvh
        const float* rb = INST ? S.inst_box + 6 * inst : S.tbox;
        const float px = (float)ray.o.x, py = (float)ray.o.y, pz = (float)ray.o.z;
        const float x0 = (rb[0] - px) * ix, x1 = (rb[3] - px) * ix, y0 = (rb[1] - py) * iy, y1 = (rb[4] - py) * iy, z0 = (rb[2] - pz) * iz, z1 = (rb[5] - pz) * iz;
        float enter = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
        enter -= 1e-3f * fabsf(enter) + 1e-3f;
        enter = fmaxf(enter, (float)tlo);
        if (fabsf(enter) < 1e30f) {
            const R base = R(enter);
            base32 = enter; lim_pad = 2.4e-7f * fabsf(enter);
            ox = -(float)(ray.o.x + ray.d.x * base) * ix; oy = -(float)(ray.o.y + ray.d.y * base) * iy; oz = -(float)(ray.o.z + ray.d.z * base) * iz;
            tlo32 = (float)(tlo - base);
        }
    }
    int sp = 0;
    int node = INST ? root : 0;
    while (true) {
        if (node >= 0) {
            vis.cnt.work(W_NODE);
            vis.cnt.work(W_NODE4);
#ifdef RTC_LANE_STATS_DETAIL   /* one-off measurement: csg_evals = four-wide visits by unordered walks, container_passes = wave-uniform four-wide visits */
            if (V::kUnordered) vis.cnt.work(W_CSG);
            if (__builtin_amdgcn_ballot_w64(node != __builtin_amdgcn_readfirstlane(node)) == 0ull) vis.cnt.work(W_CONTAINER);
#endif
            float lim = (float)vis.limit();
            if (FARW) lim -= base32;
            lim += fabsf(lim) * 1e-6f;
            if (FARW) lim += lim_pad;
            const SlabRay sr = {ix, iy, iz, ox, oy, oz, tlo32};
            constexpr bool kAnyOrder = !RTC_BVH4_SORT_ANYHIT && V::kUnordered;
            int next;
            NodeRec4 r;
            const int node0 = __builtin_amdgcn_readfirstlane(node);
            if (__builtin_amdgcn_ballot_w64(node != node0) == 0ull) {
                // every active lane at the same node: the record comes through the scalar unit (28 of its 32 dwords) and the
                // visit's arithmetic reads it from scalar registers (a copy of bvh4_visit of its own)
                typedef float f16v __attribute__((ext_vector_type(16)));
                typedef float f8v __attribute__((ext_vector_type(8)));
                typedef int i4v __attribute__((ext_vector_type(4)));
                f16v r0; f8v r1; i4v r2;
                const BvhNode4* p0 = uniform_ptr(nodes + node0);
                asm volatile("s_load_dwordx16 %0, %3, 0x0\n\ts_load_dwordx8 %1, %3, 0x40\n\ts_load_dwordx4 %2, %3, 0x60\n\ts_waitcnt lgkmcnt(0)"
                             : "=&s"(r0), "=&s"(r1), "=&s"(r2) : "s"(p0) : "memory");
                const NodeRec4 rs = {{r0[0], r0[1], r0[2], r0[3], r0[4], r0[5], r0[6], r0[7], r0[8], r0[9], r0[10], r0[11], r0[12], r0[13], r0[14], r0[15],
                                      r1[0], r1[1], r1[2], r1[3], r1[4], r1[5], r1[6], r1[7]}, {r2[0], r2[1], r2[2], r2[3]}};
                if (RTC_NODE_SPLIT) next = bvh4_visit<kStride, kAnyOrder>(rs, sr, lim, stack, sp); else r = rs;
            } else {
                const float4* q = reinterpret_cast<const float4*>(nodes + node);
                const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4], q5 = q[5], q6 = q[6];
                const NodeRec4 rv = {{q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w,
                                      q4.x, q4.y, q4.z, q4.w, q5.x, q5.y, q5.z, q5.w},
                                     {__float_as_int(q6.x), __float_as_int(q6.y), __float_as_int(q6.z), __float_as_int(q6.w)}};
                if (RTC_NODE_SPLIT) next = bvh4_visit<kStride, kAnyOrder>(rv, sr, lim, stack, sp); else r = rv;
            }
            if (!RTC_NODE_SPLIT) next = bvh4_visit<kStride, kAnyOrder>(r, sr, lim, stack, sp);
            if (next != RTC_EMPTY_CHILD) { node = next; continue; }
        } else {
            int code = ~node;
            int first = code >> 4, count = (code & 15) + 1;
            const bool uniform = __builtin_amdgcn_ballot_w64(code != __builtin_amdgcn_readfirstlane(code)) == 0ull;
            for (int i = 0; i < count; i++)
                if (vis.template tri<INST>(first + i, uniform)) return true;
        }
        if (sp == 0) return false;
        sp--;
        node = stack[sp * kStride];
    }
}

// the whole scene: unbounded prims, then the object BVH, then the triangle BVH (objects first, so
// that the mesh walk starts with the tightest bound the cheap part can give)
template <class R, class V, bool FARW> __device__ __forceinline__ void traverse_objects(const DScene<R>& S, const Ray<R>& ray, const RayInv& ri, R tlo, V& vis, int* stack) {
    for (int i = 0; i < S.n_unbounded; i++)
        if (vis.prim(S.unbounded_prims[i])) return;
    if (S.n_flat_prims > 0) {
        // two to four objects (one BVH leaf): every lane walks the same short list in the same order, so prim codes,
        // kinds and transforms come through wave-uniform (scalar) loads and no box test or stack is involved.
        // (Measured on shadow_glamour_shot: -11 % fp64, -16 % fp32; a per-object box test in front of each leaf test
        // gives all of it back -- the shadow rays of that scene pass through the boxes anyway.  A single object keeps
        // its BVH: there the root's box test spares the background pixels their fp64 leaf test.)
        if (V::kFlatRootBox) {   // one box around all of them (the tree's own): the same conservative fp32 slab test as traverse_bvh's
            const float ix = ri.ix, iy = ri.iy, iz = ri.iz;
            const float* pb = S.obox;
            const bool far_away = FARW && fmaxf(fmaxf(fabsf((float)ray.o.x), fabsf((float)ray.o.y)), fabsf((float)ray.o.z)) > S.far_origin;
            if (!far_away) {     // (a ray from farther out is not culled here: its leaf tests, in R, decide)
                const float ox = ri.ox, oy = ri.oy, oz = ri.oz;
                float x0 = fmaf(pb[0], ix, ox), x1 = fmaf(pb[3], ix, ox);
                float y0 = fmaf(pb[1], iy, oy), y1 = fmaf(pb[4], iy, oy);
                float z0 = fmaf(pb[2], iz, oz), z1 = fmaf(pb[5], iz, oz);
                float n = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
                float f = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
                n -= fabsf(n) * 1e-5f; f += fabsf(f) * 1e-5f;
                float lim = (float)vis.limit();
                lim += fabsf(lim) * 1e-6f;
                if (!(n <= f && f >= (float)tlo && n <= lim)) return;
            }
        }
        for (int i = 0; i < S.n_flat_prims; i++)
            if (vis.prim(S.bvh_prims[i])) return;
        return;
    }
    if (S.n_bvh_nodes > 0) traverse_bvh<R, V, FARW>(S, ray, ri, tlo, vis, stack);
}
template <class R, class V, bool FARW> __device__ __forceinline__ void traverse_tris(const DScene<R>& S, const Ray<R>& ray, const RayInv& ri, R tlo, V& vis, int* stack) {
    if (S.n_tnodes > 0) traverse_bvh4<R, V, false, FARW>(S, ray, ri, tlo, vis, stack);
}
// Instances of meshes are prims of the object BVH: the visitors do not walk a mesh where they meet its box (the outer walk's
// state would stay live through the inner one) but note the instance -- a byte each in one 64-bit word, in the order met,
// i.e. roughly near to far -- and the meshes are walked here afterwards, the ray taken to each one's object space by the
// instance's collapsed transform (direction not renormalised, so t is the world's t: shapes.py:271-273, 1447-1448).
struct PendingInstances {
    unsigned long long list; int n;
    __device__ __forceinline__ void clear() { list = 0ull; n = 0; }
    __device__ __forceinline__ void add(int idx) { if (n < RTC_MAX_INSTANCES) { list |= (unsigned long long)(idx + 1) << (8 * n); n++; } }
};
// FARW: whether THIS ray can start far outside the trees' boxes.  In a kernel variant for far cameras (Counters::kFar) that is the
// camera ray alone -- every other ray starts on a surface of the scene (or far out on an unbounded plane, which the variants for
// near cameras meet as well: the slab distances' relative slack grows with the origin) -- so only the camera ray's walks carry the
// re-basing.  Instanced meshes are walked in their own object space, where any ray may start far out: their walks follow the
// kernel's flag.
template <class R, class V, bool FARW> __device__ __forceinline__ void traverse(const DScene<R>& S, const Ray<R>& ray, R tlo, V& vis, int* stack) {
    // (a scene whose few objects are a flat list has no box test for this visitor's rays to make: the 56 M shadow rays of a
    // shadow_glamour_shot frame need no reciprocals)
    const bool boxes = S.n_tnodes > 0 || (S.n_flat_prims > 0 ? V::kFlatRootBox : S.n_bvh_nodes > 0);
    RayInv ri = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (boxes) ri = ray_inv(ray);
    traverse_objects<R, V, FARW>(S, ray, ri, tlo, vis, stack);
    if (!vis.done()) traverse_tris<R, V, FARW>(S, ray, ri, tlo, vis, stack);
    if (V::Cnt::kInst) {
        unsigned long long list = vis.pending.list;
        vis.pending.clear();
        while (list != 0ull && !vis.done()) {
            const int inst = (int)(list & 255ull) - 1;
            list >>= 8;
            const int32_t* I = S.inst + 8 * inst;
            const R* m = S.xform + 12 * (size_t)I[1];
            vis.oray.o = xpoint(m, ray.o); vis.oray.d = xvector(m, ray.d);
            vis.inst_leaf0 = I[2];
            traverse_bvh4<R, V, true, V::Cnt::kFar>(S, vis.oray, ray_inv(vis.oray), tlo, vis, stack, inst, I[0]);
        }
    }
}

// hit(intersect_world(ray)) (renderer.py:75-79, shapes.py:1422-1425): smallest t > 0
template <class R, class CNT> struct NearestVisitor {
    typedef CNT Cnt;
    const DScene<R>& S; const Ray<R>& ray; CNT& cnt;
    R best; int leaf; R u, v;
    PendingInstances pending; Ray<R> oray; int inst_leaf0;       // traverse(): the instances met, the ray in the object space of the one being walked
    __device__ __forceinline__ NearestVisitor(const DScene<R>& s, const Ray<R>& r, CNT& c) : S(s), ray(r), cnt(c), best(inf_<R>()), leaf(-1), u(0), v(0) { pending.clear(); }
    static constexpr bool kFlatRootBox = true;
    static constexpr bool kUnordered = false;
    __device__ __forceinline__ R limit() const { return best; }
    __device__ __forceinline__ bool done() const { return false; }
    template <bool INST> __device__ __forceinline__ bool tri(int pos, bool uniform = false) {
        R t, tu, tv;
        cnt.work(W_TRI);
        if (tri_hit((INST ? S.mb_geom : S.tb_geom) + 12 * (size_t)pos, INST ? oray : ray, t, tu, tv, uniform) && t > R(0) && t < best) {
            best = t; leaf = INST ? inst_leaf0 + S.mb_leaf[pos] : (S.tb_leaf[pos] & 0x7FFFFFFF); u = tu; v = tv;
        }
        return false;
    }
    __device__ __forceinline__ bool prim(int32_t code) {
        PrimCode p(code);
        if (CNT::kInst && p.kind == RTC_INSTANCE) { pending.add(p.idx); return false; }
        if (CNT::kStream && p.kind == RTC_CSG_ROOT && csg_streamed(S.csg_root, p.idx)) {
            cnt.work(W_CSG);
            struct First { R t; int leaf; __device__ bool operator()(R tt, int l) { if (tt > R(0)) { t = tt; leaf = l; return true; } return false; } } first = {R(0), -1};
            csg_stream(S, p.idx, ray, first);
            if (first.leaf >= 0 && first.t < best) {
                best = first.t; leaf = first.leaf;
                if (S.leaf_kind[leaf] == RTC_TRIANGLE) { R t; tri_hit(S.tri_geom + 12 * (size_t)S.leaf_tri[leaf], ray, t, u, v); }
            }
        } else if (CNT::kCsg && p.kind == RTC_CSG_ROOT) {
            CsgHits<R> H;
            cnt.work(W_CSG);
            csg_hits(S, p.idx, ray, H);
            for (int i = 0; i < H.n; i++)
                if (H.t[i] > R(0)) {
                    if (H.t[i] < best) {
                        best = H.t[i]; leaf = H.leaf[i];
                        if (S.leaf_kind[leaf] == RTC_TRIANGLE) { R t; tri_hit(S.tri_geom + 12 * (size_t)S.leaf_tri[leaf], ray, t, u, v); }
                    }
                    break;
                }
        } else {
            R t[4], tu, tv;
            cnt.work(W_LEAF);
            int n = leaf_hits(S, p.idx, ray, t, tu, tv);
            for (int k = 0; k < n; k++)
                if (t[k] > R(0) && t[k] < best) { best = t[k]; leaf = p.idx; }
        }
        return false;
    }
};

// is_shadowed (renderer.py:552-560): any hit of a shadow-casting leaf with 0 < t < distance
template <class R, class CNT> struct ShadowVisitor {
    typedef CNT Cnt;
    const DScene<R>& S; const Ray<R>& ray; CNT& cnt; R dist; bool shadowed;
    PendingInstances pending; Ray<R> oray; int inst_leaf0;
    __device__ __forceinline__ ShadowVisitor(const DScene<R>& s, const Ray<R>& r, R d, CNT& c) : S(s), ray(r), cnt(c), dist(d), shadowed(false) { pending.clear(); }
    static constexpr bool kFlatRootBox = false;    // shadow rays start on the objects: most of them are inside the box anyway
    static constexpr bool kUnordered = true;
    __device__ __forceinline__ R limit() const { return dist; }
    __device__ __forceinline__ bool done() const { return shadowed; }
    template <bool INST> __device__ __forceinline__ bool tri(int pos, bool uniform = false) {
        R t, tu, tv;
        cnt.work(W_TRI);
        // (an instance that casts no shadow is never walked by a shadow ray, prim())
        if (tri_hit((INST ? S.mb_geom : S.tb_geom) + 12 * (size_t)pos, INST ? oray : ray, t, tu, tv, uniform) && t > R(0) && t < dist && (INST || S.tb_leaf[pos] >= 0)) { shadowed = true; return true; }
        return false;
    }
    __device__ __forceinline__ bool prim(int32_t code) {
        PrimCode p(code);
        if (CNT::kInst && p.kind == RTC_INSTANCE) { if (!p.noshadow) pending.add(p.idx); return false; }
        if (CNT::kStream && p.kind == RTC_CSG_ROOT && csg_streamed(S.csg_root, p.idx)) {
            cnt.work(W_CSG);
            struct Occluder {
                const DScene<R>& S; R dist; bool found;
                __device__ bool operator()(R t, int l) {
                    if (!(t < dist)) return true;                      // sorted: nothing nearer than the light is left
                    if (t > R(0) && S.mat_shadow[S.leaf_material[l]]) { found = true; return true; }
                    return false;
                }
            } occ = {S, dist, false};
            csg_stream(S, p.idx, ray, occ);
            if (occ.found) { shadowed = true; return true; }
            return false;
        }
        if (CNT::kCsg && p.kind == RTC_CSG_ROOT) {
            CsgHits<R> H;
            cnt.work(W_CSG);
            csg_hits(S, p.idx, ray, H);
            for (int i = 0; i < H.n; i++)
                if (H.t[i] > R(0) && H.t[i] < dist && S.mat_shadow[S.leaf_material[H.leaf[i]]]) { shadowed = true; return true; }
            return false;
        }
        if (p.noshadow) return false;
        R t[4], tu, tv;
        cnt.work(W_LEAF);
        int n = leaf_hits(S, p.idx, ray, t, tu, tv);
        for (int k = 0; k < n; k++)
            if (t[k] > R(0) && t[k] < dist) { shadowed = true; return true; }
        return false;
    }
};

// The container walk of prepare_computations (renderer.py:224-242) without a sorted list.
// An object is in `containers` just before the hit iff an odd number of its intersections
// precede the hit, and its place in the list is that of its LAST such intersection.  So one
// order-free pass over every leaf the LINE meets gives: parity and last-t of the hit leaf X,
// and among the others the odd-parity leaf Y with the largest last-t.
template <class R, class CNT> struct ContainerVisitor {
    typedef CNT Cnt;
    const DScene<R>& S; const Ray<R>& ray; CNT& cnt;
    R t_hit; int x_leaf;
    bool x_odd; R x_last;
    int y_leaf; R y_last;
    PendingInstances pending; Ray<R> oray; int inst_leaf0;
    __device__ __forceinline__ ContainerVisitor(const DScene<R>& s, const Ray<R>& r, R th, int xl, CNT& c)
        : S(s), ray(r), cnt(c), t_hit(th), x_leaf(xl), x_odd(false), x_last(-inf_<R>()), y_leaf(-1), y_last(-inf_<R>()) { pending.clear(); }
    static constexpr bool kFlatRootBox = true;
    static constexpr bool kUnordered = true;          // the far bound is fixed and every leaf in range is visited: no order among children needed
    __device__ __forceinline__ R limit() const { return t_hit; }
    __device__ __forceinline__ bool done() const { return false; }
    template <bool INST> __device__ __forceinline__ bool tri(int pos, bool = false) {
        R t, tu, tv;
        const int leaf = INST ? inst_leaf0 + S.mb_leaf[pos] : (S.tb_leaf[pos] & 0x7FFFFFFF);
        if (leaf == x_leaf) return false;        // the hit triangle's one intersection is the hit itself: nothing precedes it
        cnt.work(W_TRI);
        if (tri_hit((INST ? S.mb_geom : S.tb_geom) + 12 * (size_t)pos, INST ? oray : ray, t, tu, tv) && t < t_hit) account(leaf, 1, t);
        return false;
    }
    // Which intersections of the hit leaf X precede the hit must not hang on this pass recomputing the hit's t to the
    // last bit (one ulp below t_hit would count the hit as its own predecessor and turn an entry into an exit): the hit
    // is X's intersection nearest t_hit, and what precedes it is what is smaller than THAT value.
    __device__ __forceinline__ void account(int leaf, int count, R last) {
        if (!(count & 1)) return;
        if (leaf == x_leaf) { x_odd = true; x_last = last; }
        else if (last > y_last || y_leaf < 0) { y_last = last; y_leaf = leaf; }
    }
    __device__ __forceinline__ bool prim(int32_t code) {
        PrimCode p(code);
        if (CNT::kInst && p.kind == RTC_INSTANCE) { pending.add(p.idx); return false; }
        if (CNT::kStream && p.kind == RTC_CSG_ROOT && csg_streamed(S.csg_root, p.idx)) {
            cnt.work(W_CSG);
            // Where the root's sorted list is cut: at X's entry nearest t_hit if X belongs to this root (first pass: its
            // position among the entries let through), else by value.  Second pass: the reference's `containers` list
            // itself (renderer.py:224-242) over the entries before the cut -- a leaf met again is taken out, a new one
            // is appended -- which needs room only for the leaves the ray is inside of at once.
            int cut = -1;
            if (x_leaf >= S.csg_root[4 * p.idx + 2] && x_leaf < S.csg_root[4 * p.idx + 3]) {
                struct Nearest { int x; R t_hit; R gap; int at, seen; __device__ bool operator()(R t, int l) {
                    if (l == x && abs_(t - t_hit) < gap) { gap = abs_(t - t_hit); at = seen; }
                    seen++; return false; } } nearest = {x_leaf, t_hit, inf_<R>(), -1, 0};
                csg_stream(S, p.idx, ray, nearest);
                cut = nearest.at;
            }
            struct Open {
                R t_hit; int cut, seen, n; int leaf[RTC_CSG_OPEN]; R t[RTC_CSG_OPEN];
                __device__ bool operator()(R tt, int l) {
                    if (cut >= 0 ? seen == cut : !(tt < t_hit)) return true;
                    seen++;
                    for (int i = 0; i < n; i++)
                        if (leaf[i] == l) { for (int j = i + 1; j < n; j++) { leaf[j - 1] = leaf[j]; t[j - 1] = t[j]; } n--; return false; }
                    if (n == RTC_CSG_OPEN) { for (int j = 1; j < n; j++) { leaf[j - 1] = leaf[j]; t[j - 1] = t[j]; } n--; }   // deeper than the list: the oldest goes
                    leaf[n] = l; t[n] = tt; n++;
                    return false;
                }
            } open;
            open.t_hit = t_hit; open.cut = cut; open.seen = 0; open.n = 0;
            csg_stream(S, p.idx, ray, open);
            bool other = false;
            for (int i = open.n - 1; i >= 0; i--) {
                if (open.leaf[i] == x_leaf) account(x_leaf, 1, open.t[i]);
                else if (!other) { account(open.leaf[i], 1, open.t[i]); other = true; }
            }
            return false;
        }
        if (CNT::kCsg && p.kind == RTC_CSG_ROOT) {
            CsgHits<R> H;
            cnt.work(W_CSG);
            csg_hits(S, p.idx, ray, H);
            // the sorted list is cut at the hit: at X's entry nearest t_hit if X belongs to this root, else by value
            int cut = -1;
            R nearest = inf_<R>();
            for (int i = 0; i < H.n; i++)
                if (H.leaf[i] == x_leaf && abs_(H.t[i] - t_hit) < nearest) { nearest = abs_(H.t[i] - t_hit); cut = i; }
            if (cut < 0) { cut = 0; while (cut < H.n && H.t[cut] < t_hit) cut++; }
            for (int i = 0; i < cut; i++) {
                int leaf = H.leaf[i];
                bool seen = false;
                for (int j = 0; j < i; j++) seen = seen || H.leaf[j] == leaf;
                if (seen) continue;
                int cnt = 0; R last = -inf_<R>();
                for (int j = i; j < cut; j++) if (H.leaf[j] == leaf) { cnt++; last = H.t[j]; }
                account(leaf, cnt, last);
            }
            return false;
        }
        R t[4], tu, tv;
        cnt.work(W_LEAF);
        int leaf = p.idx, n = leaf_hits(S, leaf, ray, t, tu, tv);
        R bound = t_hit;
        if (leaf == x_leaf) {
            R nearest = inf_<R>();
            for (int k = 0; k < n; k++) if (abs_(t[k] - t_hit) < nearest) { nearest = abs_(t[k] - t_hit); bound = t[k]; }
        }
        int cnt = 0; R last = -inf_<R>();
        for (int k = 0; k < n; k++) if (t[k] < bound) { cnt++; last = max_(last, t[k]); }
        account(leaf, cnt, last);
        return false;
    }
};

// ------------------------------------------------------------------------------------------
// normals (shapes.py:275-296 collapsed; local normals :537-858)
// ------------------------------------------------------------------------------------------
template <class R> __device__ __forceinline__ V3<R> normal_at(const DScene<R>& S, int leaf, V3<R> wp, R hu, R hv) {
    int kind = S.leaf_kind[leaf];
    if (kind == RTC_TRIANGLE) {
        const int row = S.leaf_tri[leaf];
        if (row <= -2) {
            // a triangle of an instanced mesh: the object-space normal (shapes.py:857-858) taken to world space by the transposed
            // collapsed transform (normal_to_world, shapes.py:288-296) -- what the world-space rows below hold ready-made
            const R* nn = S.mesh_norm + 12 * (size_t)(-2 - row);
            V3<R> n1 = mk<R>(nn[0], nn[1], nn[2]), n2 = mk<R>(nn[3], nn[4], nn[5]), n3 = mk<R>(nn[6], nn[7], nn[8]);
            return normalize(xtvector(S.xform + 12 * (size_t)S.leaf_xform[leaf], n2 * hu + n3 * hv + n1 * (R(1) - hu - hv)));
        }
        const R* nn = S.tri_norm + 12 * (size_t)row;
        V3<R> n1 = mk<R>(nn[0], nn[1], nn[2]), n2 = mk<R>(nn[3], nn[4], nn[5]), n3 = mk<R>(nn[6], nn[7], nn[8]);
        return normalize(n2 * hu + n3 * hv + n1 * (R(1) - hu - hv));
    }
    const R* m = S.xform + 12 * (size_t)S.leaf_xform[leaf];
    V3<R> p = xpoint(m, wp);
    V3<R> n;
    switch (kind) {
    case RTC_SPHERE: n = p; break;
    case RTC_PLANE: n = mk<R>(0, 1, 0); break;
    case RTC_CUBE: {
        R ax = abs_(p.x), ay = abs_(p.y), az = abs_(p.z);
        R maxc = pymax3(ax, ay, az);
        if (isclose(maxc, ax)) n = mk<R>(p.x, 0, 0);
        else if (isclose(maxc, ay)) n = mk<R>(0, p.y, 0);
        else n = mk<R>(0, 0, p.z);
        break;
    }
    default: {   // cylinder / cone: caps by `dist < 1` in both (shapes.py:676-687, 767-774)
        R dist = p.x * p.x + p.z * p.z;
        const int q = S.leaf_tri[leaf];
        R minimum = S.leaf_minmax[2 * q], maximum = S.leaf_minmax[2 * q + 1];
        // fp32 only: a hit point brought back to object space is off by ~1e-6 x its world coordinates, which is the
        // reference's whole EPSILON -- a point on a cap would pass for a wall point and get the wall's normal
        const R cap_eps = sizeof(R) == 4 ? R(1e-4) : RTC_EPS;
        if (dist < R(1) && (maximum - cap_eps) < p.y) n = mk<R>(0, 1, 0);
        else if (dist < R(1) && (minimum + cap_eps) > p.y) n = mk<R>(0, -1, 0);
        else if (kind == RTC_CYLINDER) n = mk<R>(p.x, 0, p.z);
        else { R y = sqrt_(dist); if (p.y > R(0)) y = -y; n = mk<R>(p.x, y, p.z); }
        break;
    }
    }
    return normalize(xtvector(m, n));
}

// ------------------------------------------------------------------------------------------
// patterns (shapes.py:1614-2033)
// ------------------------------------------------------------------------------------------
template <class R> __device__ __forceinline__ V3<R> uv_pattern_at(const DScene<R>& S, int p, R u, R v) {
    int type = S.pat_type[p];
    if (type == RTC_UV_CHECKER) {
        R u2 = floor_(u * S.pat_param[4 * p]), v2 = floor_(v * S.pat_param[4 * p + 1]);
        const R* q = S.pat_ab + 6 * p + (pymod(u2 + v2, R(2)) == R(0) ? 0 : 3);
        return mk<R>(q[0], q[1], q[2]);
    }
    if (type == RTC_UV_ALIGN) {
        int k = 0;
        if (v > R(0.8)) { if (u < R(0.2)) k = 1; else if (u > R(0.8)) k = 2; }
        else if (v < R(0.2)) { if (u < R(0.2)) k = 3; else if (u > R(0.8)) k = 4; }
        const R* q = S.pat_colors + 15 * p + 3 * k;
        return mk<R>(q[0], q[1], q[2]);
    }
    // RTC_UV_IMAGE: shapes.py:1975-1979, canvas.py:49-55 (np.round = half-to-even = rint)
    int tx = S.pat_tex[p];
    int w = S.tex_w[tx], h = S.tex_h[tx];
    R y = u * R(w - 1), x = (R(1) - v) * R(h - 1);
    int row = (int)rint_(x), col = (int)rint_(y);
    if (row < 0) row += h;
    if (col < 0) col += w;
    row = row < 0 ? 0 : (row >= h ? h - 1 : row);   // the reference raises IndexError out of range
    col = col < 0 ? 0 : (col >= w ? w - 1 : col);
    const uint8_t* px = S.tex_data + S.tex_off[tx] + 3 * ((size_t)row * w + col);
    R sc = S.tex_scale[tx];
    return mk<R>(R(px[0]) / sc, R(px[1]) / sc, R(px[2]) / sc);
}

template <class R> __device__ __forceinline__ V3<R> pattern_at(const DScene<R>& S, int p, V3<R> pt, V3<R> a, V3<R> b) {
    const R PI = R(3.141592653589793238462643383279502884);
    const int32_t* ch = S.pat_child + 6 * p;
    switch (S.pat_type[p]) {
    case RTC_PAT_TEST: return pt;
    case RTC_PAT_CHECKERS: {
        R s = floor_(pt.x) + floor_(pt.y) + floor_(pt.z);
        return abs_(pymod(s, R(2))) <= R(1e-8) ? a : b;
    }
    case RTC_PAT_GRADIENT: return a + (b - a) * (pt.x - floor_(pt.x));
    case RTC_PAT_RADIAL: { R mag = sqrt_(pt.x * pt.x + pt.z * pt.z); return a + (b - a) * (mag - floor_(mag)); }
    case RTC_PAT_RING: return pymod(floor_(sqrt_(pt.x * pt.x + pt.z * pt.z)), R(2)) == R(0) ? a : b;
    case RTC_PAT_STRIPE: return pymod(floor_(pt.x), R(2)) == R(0) ? a : b;
    case RTC_PAT_MAP_PLANAR: return uv_pattern_at(S, ch[0], pymod(pt.x, R(1)), pymod(pt.z, R(1)));
    case RTC_PAT_MAP_SPHERICAL: {
        R theta = atan2_(pt.x, pt.z);
        R radius = sqrt_(dot(pt, pt));
        R phi = acos_(pt.y / radius);
        R raw_u = theta / (R(2) * PI);
        return uv_pattern_at(S, ch[0], R(1) - (raw_u + R(0.5)), R(1) - phi / PI);
    }
    case RTC_PAT_MAP_CYLINDRICAL: {
        R radius = sqrt_(pt.x * pt.x + pt.z * pt.z), abs_y = abs_(pt.y);
        R coord = radius; if (abs_y > coord) coord = abs_y;
        int face = (coord == radius) ? 0 : 1;
        R raw_u = atan2_(pt.z, pt.x) / (R(2) * PI);
        return uv_pattern_at(S, ch[face], R(1) - (raw_u + R(0.5)), pymod(pt.y, R(1)));
    }
    default: {   // RTC_PAT_MAP_CUBE
        R coord = pymax3(abs_(pt.x), abs_(pt.y), abs_(pt.z));
        int face; R u, v;
        if (coord == pt.x) face = 0; else if (coord == -pt.x) face = 1;
        else if (coord == pt.y) face = 2; else if (coord == -pt.y) face = 3;
        else if (coord == pt.z) face = 4; else face = 5;
        switch (face) {
        case 0: u = pymod(R(1) - pt.z, R(2)) / R(2); v = pymod(pt.y + R(1), R(2)) / R(2); break;
        case 1: u = pymod(pt.z + R(1), R(2)) / R(2); v = pymod(pt.y + R(1), R(2)) / R(2); break;
        case 2: u = pymod(pt.x + R(1), R(2)) / R(2); v = pymod(R(1) - pt.z, R(2)) / R(2); break;
        case 3: u = pymod(pt.x + R(1), R(2)) / R(2); v = pymod(pt.z + R(1), R(2)) / R(2); break;
        case 4: u = pymod(pt.x + R(1), R(2)) / R(2); v = pymod(pt.y + R(1), R(2)) / R(2); break;
        default: u = pymod(R(1) - pt.x, R(2)) / R(2); v = pymod(pt.y + R(1), R(2)) / R(2); break;
        }
        return uv_pattern_at(S, ch[face], u, v);
    }
    }
}

// ------------------------------------------------------------------------------------------
// PerturbedPattern (shapes.py:1898-1925): noise.pnoise3 of the PyPI `noise` package, which is in neither the
// reference tree nor this image -- Ken Perlin's improved noise as that package publishes it, restated: float
// arithmetic whatever R is (the package narrows its arguments to float), no contraction (an x86 build of it has
// none), reference permutation, 16 gradients, octaves with persistence 0.5 / lacunarity 2 over the sum of
// amplitudes.  PARITY UNPINNED: checked against oracle/'s restatement of the same text only.
// ------------------------------------------------------------------------------------------
static __device__ const unsigned char PN_PERM[256] = {
    151, 160, 137, 91, 90, 15, 131, 13, 201, 95, 96, 53, 194, 233, 7, 225, 140, 36, 103, 30, 69, 142, 8, 99, 37, 240, 21, 10, 23,
    190, 6, 148, 247, 120, 234, 75, 0, 26, 197, 62, 94, 252, 219, 203, 117, 35, 11, 32, 57, 177, 33, 88, 237, 149, 56, 87, 174,
    20, 125, 136, 171, 168, 68, 175, 74, 165, 71, 134, 139, 48, 27, 166, 77, 146, 158, 231, 83, 111, 229, 122, 60, 211, 133,
    230, 220, 105, 92, 41, 55, 46, 245, 40, 244, 102, 143, 54, 65, 25, 63, 161, 1, 216, 80, 73, 209, 76, 132, 187, 208, 89, 18,
    169, 200, 196, 135, 130, 116, 188, 159, 86, 164, 100, 109, 198, 173, 186, 3, 64, 52, 217, 226, 250, 124, 123, 5, 202, 38,
    147, 118, 126, 255, 82, 85, 212, 207, 206, 59, 227, 47, 16, 58, 17, 182, 189, 28, 42, 223, 183, 170, 213, 119, 248, 152, 2,
    44, 154, 163, 70, 221, 153, 101, 155, 167, 43, 172, 9, 129, 22, 39, 253, 19, 98, 108, 110, 79, 113, 224, 232, 178, 185, 112,
    104, 218, 246, 97, 228, 251, 34, 242, 193, 238, 210, 144, 12, 191, 179, 162, 241, 81, 51, 145, 235, 249, 14, 239, 107, 49,
    192, 214, 31, 181, 199, 106, 157, 184, 84, 204, 176, 115, 121, 50, 45, 127, 4, 150, 254, 138, 236, 205, 93, 222, 114, 67, 29,
    24, 72, 243, 141, 128, 195, 78, 66, 215, 61, 156, 180};
// gradient h & 15 as three signed selectors packed two bits each (0 -> 0, 1 -> +1, 2 -> -1): x | y << 2 | z << 4
static __device__ const unsigned char PN_GRAD[16] = {0x05, 0x06, 0x09, 0x0A, 0x11, 0x12, 0x21, 0x22, 0x14, 0x18, 0x24, 0x28, 0x21, 0x22, 0x18, 0x14};

__device__ __forceinline__ int pn_perm(int i) { return PN_PERM[i & 255]; }
__device__ __forceinline__ float pn_grad3(int hash, float x, float y, float z) {
#pragma clang fp contract(off)
    const int g = PN_GRAD[hash & 15];
    const int gx = g & 3, gy = (g >> 2) & 3, gz = (g >> 4) & 3;
    const float cx = gx == 0 ? 0.0f : (gx == 1 ? 1.0f : -1.0f), cy = gy == 0 ? 0.0f : (gy == 1 ? 1.0f : -1.0f), cz = gz == 0 ? 0.0f : (gz == 1 ? 1.0f : -1.0f);
    return x * cx + y * cy + z * cz;
}
__device__ __forceinline__ float pn_lerp(float t, float a, float b) {
#pragma clang fp contract(off)
    return a + t * (b - a);
}
__device__ inline float pn_noise3(float x, float y, float z, int repeat) {
#pragma clang fp contract(off)
    const float rep = (float)repeat;
    int i = (int)floorf(fmodf(x, rep)), j = (int)floorf(fmodf(y, rep)), k = (int)floorf(fmodf(z, rep));
    int ii = (int)fmodf((float)(i + 1), rep), jj = (int)fmodf((float)(j + 1), rep), kk = (int)fmodf((float)(k + 1), rep);
    i &= 255; j &= 255; k &= 255; ii &= 255; jj &= 255; kk &= 255;
    x -= floorf(x); y -= floorf(y); z -= floorf(z);
    const float fx = x * x * x * (x * (x * 6 - 15) + 10), fy = y * y * y * (y * (y * 6 - 15) + 10), fz = z * z * z * (z * (z * 6 - 15) + 10);
    const int A = pn_perm(i), AA = pn_perm(A + j), AB = pn_perm(A + jj), B = pn_perm(ii), BA = pn_perm(B + j), BB = pn_perm(B + jj);
    return pn_lerp(fz,
                   pn_lerp(fy, pn_lerp(fx, pn_grad3(pn_perm(AA + k), x, y, z), pn_grad3(pn_perm(BA + k), x - 1, y, z)),
                           pn_lerp(fx, pn_grad3(pn_perm(AB + k), x, y - 1, z), pn_grad3(pn_perm(BB + k), x - 1, y - 1, z))),
                   pn_lerp(fy, pn_lerp(fx, pn_grad3(pn_perm(AA + kk), x, y, z - 1), pn_grad3(pn_perm(BA + kk), x - 1, y, z - 1)),
                           pn_lerp(fx, pn_grad3(pn_perm(AB + kk), x, y - 1, z - 1), pn_grad3(pn_perm(BB + kk), x - 1, y - 1, z - 1))));
}
__device__ inline float pn_pnoise3(float x, float y, float z, int octaves) {
#pragma clang fp contract(off)
    if (octaves == 1) return pn_noise3(x, y, z, 1024);
    float freq = 1.0f, amp = 1.0f, top = 0.0f, total = 0.0f;
    for (int o = 0; o < octaves; o++) {
        total += pn_noise3(x * freq, y * freq, z * freq, (int)(1024 * freq)) * amp;
        top += amp;
        freq *= 2.0f;
        amp *= 0.5f;
    }
    return total / top;
}
// PerturbedPattern._perturb (shapes.py:1911-1922): the world point moved by three noise samples one apart in z
template <class R> __device__ __noinline__ V3<R> perturb_point(const DScene<R>& S, int node, V3<R> pt) {
    const R freq = S.pat_param[4 * node], scale = S.pat_param[4 * node + 1];
    const int octaves = (int)S.pat_param[4 * node + 2];
    const R x = pt.x / freq, y = pt.y / freq;
    R z = pt.z / freq;
    V3<R> out;
    out.x = pt.x + R(pn_pnoise3((float)x, (float)y, (float)z, octaves)) * scale;
    z += R(1);
    out.y = pt.y + R(pn_pnoise3((float)x, (float)y, (float)z, octaves)) * scale;
    z += R(1);
    out.z = pt.z + R(pn_pnoise3((float)x, (float)y, (float)z, octaves)) * scale;
    return out;
}

// material colour at a world point: plain colour, or the pattern program (postfix over a colour stack)
// EXT = the kernel variant for scenes with CSG objects or perturbed patterns (rtc_api.hip): only it carries the noise code
template <class R, bool EXT> __device__ __forceinline__ V3<R> surface_color(const DScene<R>& S, int material, int leaf, V3<R> wp) {
    int prog = S.mat_program[material];
    const R* mc = S.mat + 10 * (size_t)material;
    if (prog < 0) return mk<R>(mc[0], mc[1], mc[2]);
    V3<R> op = xpoint(S.xform + 12 * (size_t)S.leaf_xform[leaf], wp);   // shape.world_to_object (shapes.py:1664)
    int off = S.prog_off[prog], len = S.prog_len[prog];
    V3<R> stack[RTC_PAT_STACK];
    V3<R> outer0 = wp, outer1 = wp;        // the points PERTURB replaced (RTC_PERTURB_DEPTH = 2)
    int sp = 0, moved = 0;
    for (int i = 0; i < len; i++) {
        int opc = S.instr[2 * (off + i)], node = S.instr[2 * (off + i) + 1];
        if (EXT && opc >= RTC_OP_PERTURB) {
            if (opc == RTC_OP_PERTURB) {
                if (moved++ == 0) outer0 = wp; else outer1 = wp;
                wp = perturb_point(S, node, wp);
            } else {
                wp = --moved == 0 ? outer0 : outer1;
            }
            op = xpoint(S.xform + 12 * (size_t)S.leaf_xform[leaf], wp);
            continue;
        }
        if (opc == RTC_OP_BLEND) {
            V3<R> q = stack[--sp], p = stack[--sp];
            stack[sp++] = (p + q) * R(0.5);
        } else {
            V3<R> a, b;
            if (opc == RTC_OP_EVAL_AB) { b = stack[--sp]; a = stack[--sp]; }
            else { const R* q = S.pat_ab + 6 * node; a = mk<R>(q[0], q[1], q[2]); b = mk<R>(q[3], q[4], q[5]); }
            V3<R> pp = xpoint(S.pat_inv + 12 * (size_t)node, op);   // inverse(pattern.transform) * object_point
            V3<R> c = pattern_at(S, node, pp, a, b);
            if (len == 1) return c;
            stack[sp++] = c;
        }
    }
    return stack[0];
}

// ------------------------------------------------------------------------------------------
// counter-based RNG for jittered area lights (the reference's own jitter is not reproducible)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
template <class R> __device__ __forceinline__ R jitter01(uint64_t seed, uint32_t pixel, uint32_t ray_no, uint32_t light, uint32_t cell, uint32_t stream) {
    uint32_t h = hash32((uint32_t)seed ^ hash32(pixel + 0x9e3779b9U * (ray_no + 1)));
    h = hash32(h ^ hash32(light * 0x85ebca6bU + cell * 4u + stream));
    return R(h >> 8) * R(1.0 / 16777216.0);
}

// ------------------------------------------------------------------------------------------
// the trace loop for one pixel
// ------------------------------------------------------------------------------------------



template <class R, class CNT>
__device__ __forceinline__ bool is_shadowed(const DScene<R>& S, V3<R> light_position, V3<R> point, int* stack, CNT& cnt) {
    V3<R> v = light_position - point;
    R distance = sqrt_(dot(v, v));
    Ray<R> r = {point, v * (R(1) / distance)};
    cnt.c[1]++;
    ShadowVisitor<R, CNT> vis(S, r, distance, cnt);
    traverse<R, ShadowVisitor<R, CNT>, false>(S, r, R(0), vis, stack);
    return vis.shadowed;
}

template <class R>
__device__ __forceinline__ V3<R> light_point(const DScene<R>& S, int l, int u, int v, R ju, R jv) {
    const R* g = S.light_geom + 9 * l;
    return mk<R>(g[0], g[1], g[2]) + mk<R>(g[3], g[4], g[5]) * (R(u) + ju) + mk<R>(g[6], g[7], g[8]) * (R(v) + jv);
}

// prepare_computations (renderer.py:206-222) for the hit (t, leaf, u, v) of `ray`
template <class R> struct HitCtx {
    V3<R> point, normalv, eyev, reflectv, over_point, under_point;
    int leaf, material;
    R t;
};

template <class R>
__device__ __forceinline__ HitCtx<R> make_ctx(const DScene<R>& S, const Ray<R>& ray, R t, int leaf, R hu, R hv, R shade_eps, R shade_eps_rel) {
    HitCtx<R> c;
    c.t = t; c.leaf = leaf;
    c.point = ray.o + ray.d * t;
    c.normalv = normal_at(S, leaf, c.point, hu, hv);
    c.eyev = neg(ray.d);
    c.reflectv = reflect(ray.d, c.normalv);           // before the inside flip, like the reference
    if (dot(c.normalv, c.eyev) < R(0)) c.normalv = neg(c.normalv);
    // fp32: the hit point carries the rounding of o + d*t, i.e. a few ulp of the larger of |o| and |p|
    R scale = max_(max_(max_(abs_(c.point.x), abs_(c.point.y)), abs_(c.point.z)), max_(max_(abs_(ray.o.x), abs_(ray.o.y)), abs_(ray.o.z)));
    R eps = max_(shade_eps, shade_eps_rel * scale);
    c.over_point = c.point + c.normalv * eps;
    c.under_point = c.point - c.normalv * eps;
    c.material = S.leaf_material[leaf];
    return c;
}

struct LightShape { bool area, jitter; int usteps, vsteps, samples; };
template <class R> __device__ __forceinline__ LightShape light_shape(const DScene<R>& S, int l) {
    LightShape s;
    s.area = S.light_type[l] == 1;
    s.usteps = s.area ? S.light_steps[3 * l] : 1;
    s.vsteps = s.area ? S.light_steps[3 * l + 1] : 1;
    s.jitter = s.area && S.light_steps[3 * l + 2] != 0;
    s.samples = s.usteps * s.vsteps;
    return s;
}
template <class R>
__device__ __forceinline__ V3<R> light_sample(const DScene<R>& S, int l, const LightShape& ls, int u, int v, uint64_t seed, uint32_t pixel,
                                              uint32_t path, uint32_t stream) {
    if (!ls.area) return mk<R>(S.light_geom[9 * l], S.light_geom[9 * l + 1], S.light_geom[9 * l + 2]);
    R ju = R(0.5), jv = R(0.5);
    if (ls.jitter) { ju = jitter01<R>(seed, pixel, path, l, v * ls.usteps + u, stream); jv = jitter01<R>(seed, pixel, path, l, v * ls.usteps + u, stream + 1); }
    return light_point(S, l, u, v, ju, jv);
}

// intensity_at * samples (renderer.py:919-922, 964-973): how many of the light's cells see `over_point`
template <class R, class CNT>
__device__ __forceinline__ int count_lit(const DScene<R>& S, V3<R> over_point, int l, int* stack, CNT& cnt, uint64_t seed, uint32_t pixel, uint32_t path) {
    LightShape ls = light_shape(S, l);
    int lit = 0;
    for (int v = 0; v < ls.vsteps; v++)
        for (int u = 0; u < ls.usteps; u++)
            if (!is_shadowed(S, light_sample<R>(S, l, ls, u, v, seed, pixel, path, 0), over_point, stack, cnt)) lit++;
    return lit;
}

// lighting (renderer.py:1066-1090) for light l, in two halves.  light_terms: the ambient term and the mean over the
// light's cells of the diffuse + specular terms (a FRESH set of surface points, renderer.py:1077 -> 957-962); returns
// whether any cell contributes at all.  lighting: ambient + mean x (lit cells / cells), ambient alone when no cell is lit
// (shade_intensity == 0.0).  When no cell contributes -- the light is behind the surface -- the result is the ambient term
// whatever intensity_at would say, so the caller need not trace those shadow rays.
template <class R, class CNT>
__device__ __forceinline__ bool light_terms(const DScene<R>& S, const HitCtx<R>& c, V3<R> pcolor, int l, CNT& cnt, uint64_t seed, uint32_t pixel,
                                            uint32_t path, V3<R>& amb, V3<R>& mean) {
    const R* m = S.mat + 10 * (size_t)c.material;
    R ambient = m[3], diffuse = m[4], specular = m[5], shininess = m[6];
    const R* li = S.light_intensity + 3 * l;
    V3<R> lint = mk<R>(li[0], li[1], li[2]);
    V3<R> effective = pcolor * lint;
    amb = effective * ambient;
    LightShape ls = light_shape(S, l);
    V3<R> acc = mk<R>(0, 0, 0);
    // a coefficient of exactly 0 makes its term +0 whatever the geometry (renderer.py:1083-1090): the specular pow() --
    // the dearest thing in this loop -- is skipped for matte materials, the whole loop for materials that are neither
    const bool has_specular = specular != R(0);
    if (diffuse == R(0) && !has_specular) { mean = acc; return false; }
    for (int v = 0; v < ls.vsteps; v++)
        for (int u = 0; u < ls.usteps; u++) {
            V3<R> pos = light_sample<R>(S, l, ls, u, v, seed, pixel, path, 2);
            cnt.work(W_LIGHT);
            V3<R> lightv = normalize(pos - c.over_point);
            R ldn = dot(lightv, c.normalv);
            if (ldn >= R(0)) {
                acc = acc + effective * diffuse * ldn;
                if (has_specular) {
                    V3<R> reflectv = reflect(neg(lightv), c.normalv);
                    R rde = dot(reflectv, c.eyev);
                    if (rde > R(0)) acc = acc + lint * specular * pow_(rde, shininess);
                }
            }
        }
    mean = mk<R>(acc.x / R(ls.samples), acc.y / R(ls.samples), acc.z / R(ls.samples));
    return acc.x != R(0) || acc.y != R(0) || acc.z != R(0);
}
template <class R>
__device__ __forceinline__ V3<R> lighting(const DScene<R>& S, V3<R> amb, V3<R> mean, int l, int lit) {
    if (lit == 0) return amb;
    LightShape ls = light_shape(S, l);
    return amb + mean * (R(lit) / R(ls.samples));
}

// The light loop of a replicated part (DFrame::replicas): the nrep lanes of a pixel hold the same hit.  The shadow rays of
// all lights -- "items", numbered light by light, cell by cell in count_lit's order -- are dealt out round robin: in round k
// replica r traces item k * nrep + r, a ballot collects the round's answers, and every replica keeps all of them (128 bits).
// Then each replica adds the lights up in the reference's order with the lit counts read from the bits: the same
// arithmetic on the same values as the one-lane loop, so the same colour to the last bit -- in a quarter (a sixteenth)
// of the dependent tree walks.  A light that cannot contribute is not traced here either (its items count as culled).
__device__ __forceinline__ int count_bits(unsigned long long b0, unsigned long long b1, int start, int n) {
    int c = 0;
    if (start < 64) {
        const int e = start + n < 64 ? start + n : 64, w = e - start;
        c += __popcll(b0 & ((w >= 64 ? ~0ull : ((1ull << w) - 1ull)) << start));
    }
    if (start + n > 64) {
        const int s1 = (start > 64 ? start : 64) - 64, w = start + n - 64 - s1;
        c += __popcll(b1 & ((w >= 64 ? ~0ull : ((1ull << w) - 1ull)) << s1));
    }
    return c;
}
template <class R, class CNT>
__device__ __forceinline__ V3<R> light_loop_replicated(const DScene<R>& S, const HitCtx<R>& c, V3<R> pcolor, int* stack, CNT& cnt, uint64_t seed,
                                                       uint32_t pixel, uint32_t path) {
    const int nrep = cnt.nrep, rep = cnt.rep, lane = threadIdx.x & 63;
    int items = 0;
    for (int l = 0; l < S.n_lights; l++) items += light_shape(S, l).samples;
    unsigned long long bits0 = 0, bits1 = 0;
    int l = 0, start_l = 0, samples_l = S.n_lights > 0 ? light_shape(S, 0).samples : 0, known_l = -1;
    bool matters = false;
    for (int k0 = 0; k0 < items; k0 += nrep) {
        const int i = k0 + rep;
        bool lit = false;
        if (i < items) {
            while (i >= start_l + samples_l) { start_l += samples_l; l++; samples_l = light_shape(S, l).samples; }
            if (l != known_l) { V3<R> amb, mean; matters = light_terms(S, c, pcolor, l, cnt, seed, pixel, path, amb, mean); known_l = l; }
            if (matters) {
                const LightShape ls = light_shape(S, l);
                const int cell = i - start_l;
                lit = !is_shadowed(S, light_sample<R>(S, l, ls, cell % ls.usteps, cell / ls.usteps, seed, pixel, path, 0), c.over_point, stack, cnt);
            } else cnt.c[4]++;
        }
        const unsigned long long round = (__builtin_amdgcn_ballot_w64(lit) >> (lane & ~(nrep - 1))) & ((1ull << nrep) - 1ull);
        if (k0 < 64) bits0 |= round << k0; else bits1 |= round << (k0 - 64);        // nrep divides 64: a round never straddles
    }
    V3<R> surface = mk<R>(0, 0, 0);
    int start = 0;
    for (int m = 0; m < S.n_lights; m++) {
        V3<R> amb, mean;
        const bool contributes = light_terms(S, c, pcolor, m, cnt, seed, pixel, path, amb, mean);
        const int samples = light_shape(S, m).samples;
        surface = surface + lighting(S, amb, mean, m, contributes ? count_bits(bits0, bits1, start, samples) : 0);
        start += samples;
    }
    return surface;
}

// n1 / n2 from what the container pass found: X = the hit leaf (in `containers` iff x_odd, entered
// last at x_last), Y = the other odd-parity leaf entered last (at y_last), or none
template <class R>
__device__ __forceinline__ void container_result(const DScene<R>& S, int leaf, bool x_odd, R x_last, int y_leaf, R y_last, R& n1, R& n2) {
    R ri_x = S.mat[10 * (size_t)S.leaf_material[leaf] + 9];
    R ri_y = y_leaf >= 0 ? S.mat[10 * (size_t)S.leaf_material[y_leaf] + 9] : R(1);
    if (x_odd) { n1 = (y_leaf >= 0 && y_last > x_last) ? ri_y : ri_x; n2 = ri_y; }
    else { n1 = ri_y; n2 = ri_x; }
}

// n1 / n2 of prepare_computations (renderer.py:224-242) via the order-free container pass
template <class R, class CNT>
__device__ __forceinline__ void container_n1n2(const DScene<R>& S, const Ray<R>& ray, bool primary, R t, int leaf, int* stack, CNT& cnt, R& n1, R& n2) {
    cnt.work(W_CONTAINER);
    ContainerVisitor<R, CNT> cv(S, ray, t, leaf, cnt);
    if (CNT::kFar && primary) traverse<R, ContainerVisitor<R, CNT>, true>(S, ray, -inf_<R>(), cv, stack);
    else traverse<R, ContainerVisitor<R, CNT>, false>(S, ray, -inf_<R>(), cv, stack);
    container_result(S, leaf, cv.x_odd, cv.x_last, cv.y_leaf, cv.y_last, n1, n2);
}

// What shade_hit adds on top of the surface colour (renderer.py:517-525): which secondary rays
// exist and the factor each one's colour is multiplied by -- reflective [* reflectance] and
// transparency [* (1 - reflectance)] -- with refracted_color's direction (renderer.py:339-350)
// and schlick (renderer.py:272-283).
template <class R> struct Secondary { bool refl, refr; R w_refl, w_refr; V3<R> refr_dir; };

template <class R>
__device__ __forceinline__ Secondary<R> secondary_rays(const DScene<R>& S, const HitCtx<R>& c, int remaining, R n1, R n2) {
    const R* m = S.mat + 10 * (size_t)c.material;
    R reflective = m[7], transparency = m[8];
    Secondary<R> s;
    s.refl = remaining > 0 && reflective != R(0);
    s.refr = false;
    s.w_refl = reflective; s.w_refr = transparency;
    s.refr_dir = mk<R>(0, 0, 0);
    if (remaining > 0 && transparency != R(0)) {
        R n_ratio = n1 / n2;
        R cos_i = dot(c.eyev, c.normalv);
        R sin2_t = n_ratio * n_ratio * (R(1) - cos_i * cos_i);
        if (!(sin2_t > R(1))) {
            R cos_t = sqrt_(R(1) - sin2_t);
            s.refr_dir = c.normalv * (n_ratio * cos_i - cos_t) - c.eyev * n_ratio;
            s.refr = true;
        }
        if (reflective > R(0) && transparency > R(0)) {
            R cosv = cos_i, reflectance;
            bool tir = false;
            if (n1 > n2) {
                R n = n1 / n2;
                R s2 = n * n * (R(1) - cosv * cosv);
                if (s2 > R(1)) tir = true; else cosv = sqrt_(R(1) - s2);
            }
            if (tir) reflectance = R(1);
            else {
                R r0 = (n1 - n2) / (n1 + n2); r0 = r0 * r0;
                R om = R(1) - cosv;
                reflectance = r0 + (R(1) - r0) * (om * om * om * om * om);
            }
            s.w_refl *= reflectance;
            s.w_refr *= (R(1) - reflectance);
        }
    }
    return s;
}
__device__ __forceinline__ bool needs_containers(float transparency, int remaining) { return remaining > 0 && transparency != 0.0f; }
__device__ __forceinline__ bool needs_containers(double transparency, int remaining) { return remaining > 0 && transparency != 0.0; }

// Rays of one pixel are numbered like a binary heap: primary 1, reflection of p = 2p, refraction
// of p = 2p + 1 (only the jitter hash uses the number).

// color_at's body for ONE ray (renderer.py:527-566 without the recursion): nearest hit,
// prepare_computations, the secondary rays shade_hit would recurse into -- handed to `emit`
// BEFORE the light loop, so that what only they need (reflectv, under_point, the ray itself, t)
// is dead while the shadow rays are walked -- and the surface colour from every light.
// emit(has_refl, refl_origin, refl_dir, refl_factor, has_refr, refr_origin, refr_dir, refr_factor)
// is called once for every ray that hits something; if it returns false the ray is given up here
// (its caller will have it traced again some other way) and the light loop is skipped.
template <class R, class CNT, class EMIT>
__device__ __forceinline__ V3<R> shade_ray(const DScene<R>& S, const Ray<R>& ray, bool primary /* the camera ray */, int remaining, uint32_t path, R shade_eps, R shade_eps_rel,
                                           int* stack, CNT& cnt, uint64_t seed, uint32_t pixel, int& hit_leaf, EMIT& emit) {
    NearestVisitor<R, CNT> nv(S, ray, cnt);
    if (CNT::kFar && primary) traverse<R, NearestVisitor<R, CNT>, true>(S, ray, R(0), nv, stack);
    else traverse<R, NearestVisitor<R, CNT>, false>(S, ray, R(0), nv, stack);
    hit_leaf = nv.leaf;
    V3<R> surface = mk<R>(0, 0, 0);
    if (nv.leaf >= 0) {
        cnt.work(W_SHADE);
        HitCtx<R> c = make_ctx(S, ray, nv.best, nv.leaf, nv.u, nv.v, shade_eps, shade_eps_rel);
        if (S.mat_program[c.material] >= 0) cnt.work(W_PATTERN);
        R n1 = R(1), n2 = R(1);
        if (S.refraction_matters && needs_containers(S.mat[10 * (size_t)c.material + 8], remaining)) container_n1n2(S, ray, primary, c.t, c.leaf, stack, cnt, n1, n2);
        Secondary<R> sec = secondary_rays(S, c, remaining, n1, n2);
        if (!emit(sec.refl, c.over_point, c.reflectv, sec.w_refl, sec.refr, c.under_point, sec.refr_dir, sec.w_refr)) return surface;
        V3<R> pcolor = surface_color<R, CNT::kCsg>(S, c.material, c.leaf, c.over_point);
        if (CNT::kReplicas && cnt.nrep > 1) return light_loop_replicated(S, c, pcolor, stack, cnt, seed, pixel, path);
        for (int l = 0; l < S.n_lights; l++) {
            V3<R> amb, mean;
            const bool matters = light_terms(S, c, pcolor, l, cnt, seed, pixel, path, amb, mean);
#ifdef RTC_TRACE_ALL_SHADOW_RAYS
            const int lit = count_lit(S, c.over_point, l, stack, cnt, seed, pixel, path);
#else
            int lit = 0;
            if (matters) lit = count_lit(S, c.over_point, l, stack, cnt, seed, pixel, path);
            else cnt.c[4] += (unsigned int)light_shape(S, l).samples;
#endif
            surface = surface + lighting(S, amb, mean, l, lit);
        }
    }
    return surface;
}

// The whole ray tree under one ray in ONE lane: recursion unrolled into a stack of pending rays;
// weights are multiplied down the tree.  Returns sum of weight * surface over the tree.
template <class R> struct Bounce { V3<R> o, d, w; int remaining; uint32_t path; };

template <class R, class CNT> struct PushPending {
    Bounce<R>* pending; int& np; const V3<R>& weight; const int& remaining; const uint32_t& path; CNT& cnt;
    __device__ __forceinline__ bool operator()(bool refl, V3<R> refl_o, V3<R> refl_d, R w_refl, bool refr, V3<R> refr_o, V3<R> refr_d, R w_refr) {
        if (refr && np < RTC_MAX_BOUNCE_STACK) {
            Bounce<R> b = {refr_o, refr_d, weight * w_refr, remaining - 1, 2 * path + 1};
            pending[np++] = b;
            cnt.c[3]++;
        }
        if (refl && np < RTC_MAX_BOUNCE_STACK) {
            Bounce<R> b = {refl_o, refl_d, weight * w_refl, remaining - 1, 2 * path};
            pending[np++] = b;
            cnt.c[2]++;
        }
        return true;
    }
};

template <class R, class CNT>
__device__ __forceinline__ V3<R> trace_tree(const DScene<R>& S, Ray<R> ray, V3<R> weight, int remaining, uint32_t path, R shade_eps,
                                            R shade_eps_rel, int* stack, CNT& cnt, uint64_t seed, uint32_t pixel, int& first_leaf) {
    Bounce<R> pending[RTC_MAX_BOUNCE_STACK];
    int np = 0;
    V3<R> color = mk<R>(0, 0, 0);
    bool first = true;
    first_leaf = -1;
    while (true) {
        int leaf;
        PushPending<R, CNT> push = {pending, np, weight, remaining, path, cnt};
        V3<R> surface = shade_ray(S, ray, first, remaining, path, shade_eps, shade_eps_rel, stack, cnt, seed, pixel, leaf, push);
        if (first) { first_leaf = leaf; first = false; }
        if (leaf >= 0) color = color + weight * surface;
        if (np == 0) break;
        np--;
        ray.o = pending[np].o; ray.d = pending[np].d; weight = pending[np].w; remaining = pending[np].remaining; path = pending[np].path;
    }
    return color;
}

// Can a camera ray meet anything at all?  The conservative fp32 slab test of traverse_bvh against the boxes of the two trees
// (a scene with unbounded things -- a floor plane -- always can).  A block of sky pixels then skips the trace code altogether:
// its colour is the reference's black for a ray that hits nothing (renderer.py:606-611), and it never executes the register
// spills at the head of that code -- which every wave of a 1080p dragons frame used to write to scratch, sky waves (60 % of
// them) included: two thirds of the frame's HBM write traffic.  Not for the far-camera variants (the fp32 test is relative to
// the ray's origin; their rays are re-based inside the walks).
template <class R> __device__ __forceinline__ bool may_hit_box(const Ray<R>& ray, const float* b) {
    float ix = 1.0f / (float)ray.d.x, iy = 1.0f / (float)ray.d.y, iz = 1.0f / (float)ray.d.z;
    if (!(fabsf(ix) <= 1e30f)) ix = copysignf(1e30f, (float)ray.d.x);
    if (!(fabsf(iy) <= 1e30f)) iy = copysignf(1e30f, (float)ray.d.y);
    if (!(fabsf(iz) <= 1e30f)) iz = copysignf(1e30f, (float)ray.d.z);
    const float ox = -(float)ray.o.x * ix, oy = -(float)ray.o.y * iy, oz = -(float)ray.o.z * iz;
    const float x0 = fmaf(b[0], ix, ox), x1 = fmaf(b[3], ix, ox), y0 = fmaf(b[1], iy, oy), y1 = fmaf(b[4], iy, oy), z0 = fmaf(b[2], iz, oz), z1 = fmaf(b[5], iz, oz);
    float n = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
    float f = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
    n -= fabsf(n) * 1e-5f; f += fabsf(f) * 1e-5f;
    return n <= f && f >= 0.0f;
}
template <class R, class CNT> __device__ __forceinline__ bool may_hit_scene(const DScene<R>& S, const Ray<R>& ray) {
    if (CNT::kFar || S.n_unbounded > 0) return true;
    return (S.n_bvh_nodes > 0 && may_hit_box(ray, S.obox)) || (S.n_tnodes > 0 && may_hit_box(ray, S.tbox));
}

// color_at for one primary ray (the single-kernel form)
template <class R, class CNT>
__device__ __forceinline__ V3<R> trace_pixel(const DScene<R>& S, Ray<R> primary, int max_depth, R shade_eps, R shade_eps_rel, int* stack,
                                             CNT& cnt, uint64_t seed, uint32_t pixel, int& hit_leaf_out) {
    V3<R> c = trace_tree(S, primary, mk<R>(1, 1, 1), max_depth, 1u, shade_eps, shade_eps_rel, stack, cnt, seed, pixel, hit_leaf_out);
    if (CNT::kReplicas && cnt.rep != 0) cnt.c[2] = cnt.c[3] = 0;      // replicas trace the same reflection and refraction rays: counted once
    return c;
}

// renderer.py:742-751
template <class R> __device__ __forceinline__ Ray<R> ray_for_pixel(const DCamera<R>& cam, int px, int py) {
    R xoffset = (R(px) + R(0.5)) * cam.pixel_size, yoffset = (R(py) + R(0.5)) * cam.pixel_size;
    R world_x = cam.half_width - xoffset, world_y = cam.half_height - yoffset;
    V3<R> pixel = xpoint(cam.inv, mk<R>(world_x, world_y, R(-1)));
    V3<R> origin = xpoint(cam.inv, mk<R>(0, 0, 0));
    Ray<R> r = {origin, normalize(pixel - origin)};
    return r;
}

// block-level reduction of the ray / work counters, then a few atomics per block
template <bool PROF, bool CSG, bool FAR, bool STREAM, bool REPL, bool INST>
__device__ __forceinline__ void flush_counters(Counters<PROF, CSG, FAR, STREAM, REPL, INST>& cnt, const DFrame& F, unsigned int* red /* LDS, >= 5 * waves per workgroup */) {
    constexpr int kWaves = Counters<PROF, CSG, FAR, STREAM, REPL, INST>::kBlock / 64;
    if (!PROF && (!CSG || RTC_CSG_SLOTS) && F.rays && F.work) {          // per-wave slots (DFrame::work)
        uint32_t* parts = reinterpret_cast<uint32_t*>(F.work);
        const int ln = threadIdx.x & 63;
        unsigned int mine = 0;
        for (int k = 0; k < 5; k++) {
            unsigned int v = cnt.c[k];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            v = __builtin_amdgcn_readfirstlane(v);
            if (ln == k) mine = v;
        }
        const size_t slot = (size_t)blockIdx.x * kWaves + (threadIdx.x >> 6);
        if (slot < (size_t)F.n_slots) { if (ln < 8) parts[slot * 8 + ln] = mine; }      // one 32-byte store per wave
        else if (ln < 5 && mine) atomicAdd(F.rays + (ln < 4 ? ln : 12), (unsigned long long)mine);      // no slot for this wave (never, by the host's arithmetic)
    } else if (F.rays) {
        __syncthreads();
        for (int k = 0; k < 5; k++) {
            unsigned int v = cnt.c[k];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if ((threadIdx.x & 63) == 0) red[k * kWaves + (threadIdx.x >> 6)] = v;
        }
        __syncthreads();
        if (threadIdx.x < 5) {
            unsigned long long s = 0;
            for (int q = 0; q < kWaves; q++) s += red[threadIdx.x * kWaves + q];
            if (s) atomicAdd(F.rays + (threadIdx.x < 4 ? threadIdx.x : 12), s);      // [4..11] are the work counters
        }
    }
    if (PROF && F.work) {
        for (int k = 0; k < W_COUNT; k++) {
            unsigned int v = cnt.w[k];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if ((threadIdx.x & 63) == 0 && v) atomicAdd(k < 8 ? F.work + k : F.rays + 13, (unsigned long long)v);      // [4..11] = rtc_stats::work, [13] = wide node visits
        }
    }
}
template <bool PROF, bool CSG, bool FAR, bool STREAM, bool REPL, bool INST> __device__ __forceinline__ void zero_counters(Counters<PROF, CSG, FAR, STREAM, REPL, INST>& cnt) {
    cnt.rep = 0; cnt.nrep = 1;
    for (int k = 0; k < 5; k++) cnt.c[k] = 0;
    for (int k = 0; k < (PROF ? W_COUNT : 1); k++) cnt.w[k] = 0;
}

// pixel of work item w of the frame (tile order: one wave = one 8x8 block of a T x T tile)
__device__ __forceinline__ bool work_pixel(const DFrame& F, int hsize, int vsize, long long w, int& x, int& y, size_t& out_index) {
    const int T = F.tile_size, per_tile = T * T, sub = T / 8;
    int slot = (int)(w / per_tile), within = (int)(w % per_tile);
    if (slot >= F.n_tiles) return false;
    int tile = F.tile_ids ? F.tile_ids[slot] : slot;
    int tx0 = (tile % F.tiles_x) * T, ty0 = (tile / F.tiles_x) * T;
    int sb = within >> 6, l = within & 63;
    x = tx0 + (sb % sub) * 8 + (l & 7);
    y = ty0 + (sb / sub) * 8 + (l >> 3);
    if (x >= hsize || y >= vsize) return false;
    out_index = F.tile_ids ? (size_t)slot * per_tile + (size_t)(y - ty0) * T + (x - tx0) : (size_t)x * vsize + y;
    return true;
}

}  // namespace rtc
