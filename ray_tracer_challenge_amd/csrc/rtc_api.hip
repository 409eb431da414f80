// rtc_api.hip -- the C ABI of include/rtc_api.h: scene upload, BVH build, kernel launches.
// Built by hipcc for gfx950 into librtc_hip.so together with rtc_trace_f64.hip (the fp64 trace kernels without CSG).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/rtc_api.h"
#include "rtc_bvh.hpp"
#include "rtc_device.hpp"
#include "rtc_trace_kernel.hpp"
#include "rtc_lbvh_device.hpp"
#include <rocprim/rocprim.hpp>

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return fail(RTC_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
// The per-wave ray counts of a launch (DFrame::work) added to the caller's five counters: a few blocks, five atomics each.
__global__ __launch_bounds__(256) void sum_rays_kernel(const uint4* __restrict__ parts, int n_waves, unsigned long long* rays) {
    __shared__ unsigned long long red[5][4];
    unsigned long long s[5] = {0, 0, 0, 0, 0};
    for (int w = blockIdx.x * 256 + threadIdx.x; w < n_waves; w += gridDim.x * 256) {
        const uint4 a = parts[2 * (size_t)w];
        const unsigned int e = reinterpret_cast<const unsigned int*>(parts + 2 * (size_t)w + 1)[0];
        s[0] += a.x; s[1] += a.y; s[2] += a.z; s[3] += a.w; s[4] += e;
    }
    for (int k = 0; k < 5; k++) {
        for (int off = 32; off > 0; off >>= 1) s[k] += __shfl_down(s[k], off, 64);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = s[k];
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        const unsigned long long t = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
        if (t) atomicAdd(rays + (threadIdx.x < 4 ? threadIdx.x : 12), t);
    }
}

// Next frame's launch order from this frame's unit costs (one block; n is a few 10^4): entries in
// descending cost by buckets of a quarter octave -- the launch cannot end before its longest wave
// does, so the long ones go first -- and, from the second measured frame on, with the dearest units
// split (see order_kernel).  The order inside a bucket is whatever the atomics give; any order and
// any split render the same image.
#define RTC_ORDER_BUCKETS 128
__device__ __forceinline__ int order_bucket(uint32_t c) {
    int lg = 31 - __clz((int)(c | 1u));                              // floor(log2 c)
    int frac = lg >= 2 ? (int)((c >> (lg - 2)) & 3u) : 0;            // next two bits: quarter octaves
    int b = 4 * lg + frac;
    return RTC_ORDER_BUCKETS - 1 - (b > RTC_ORDER_BUCKETS - 1 ? RTC_ORDER_BUCKETS - 1 : b);   // dearest first
}
__device__ __forceinline__ double order_bucket_floor(int bucket) {     // smallest cost that lands in the bucket
    int j = RTC_ORDER_BUCKETS - 1 - bucket, lg = j >> 2, frac = j & 3;
    return lg >= 2 ? (double)(4 + frac) * (double)(1ULL << (lg - 2)) : (double)(1ULL << lg);
}
// Splitting.  A launch cannot end before its longest wave does, and on the mesh scenes the longest waves (blocks of glass and
// mirror pixels with deep ray trees) run for most of the frame: dragons 1080p fp64, 3 072 wave slots, sum of all wave times
// / slots = 455 us but the dearest wave alone 700 us.  A unit can be launched as four 4x4-quadrant waves or sixteen 2x2
// waves instead; a part is shorter than the whole, not proportionally (a wave runs as long as its slowest lane's chain of
// dependent fetches: part4 ~ 0.65 x, part16 ~ 0.5 x the whole), so splitting buys a shorter critical path with more total
// wave time.  The rule: for every target T among the bucket ceilings, the buckets above T are split the least that brings
// their parts under T; the launch is then expected to take max(T, 1.1 x total wave time / slots); the T with the smallest
// expectation wins (T = the dearest unit: nothing is split).  The same rule serves a full GPU (few units split) and a
// rank's share of a frame on many GPUs (few waves, so almost everything dear is split).  Parts take their place in the
// order by their own expected cost.
__global__ __launch_bounds__(1024) void order_kernel(const uint32_t* cost, int32_t* order, int n, int n_entries, int n_slots, float part4,
                                                     float part16, int32_t* used, float force_frac) {
    __shared__ unsigned int hist[RTC_ORDER_BUCKETS], cursor[RTC_ORDER_BUCKETS];
    __shared__ unsigned long long bsum[RTC_ORDER_BUCKETS];
    __shared__ unsigned char ways[RTC_ORDER_BUCKETS];     // 1, 4 or 16 waves for the units of the bucket
    __shared__ unsigned long long total;
    for (int b = threadIdx.x; b < RTC_ORDER_BUCKETS; b += blockDim.x) { hist[b] = 0; ways[b] = 1; bsum[b] = 0; }
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    unsigned long long mine = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        uint32_t c = cost[i]; mine += c;
        int b = order_bucket(c);
        atomicAdd(&hist[b], 1u); atomicAdd(&bsum[b], (unsigned long long)c);
    }
    atomicAdd(&total, mine);
    __syncthreads();
    if (threadIdx.x == 0 && part4 > 0.0f) {
        const double slots = (double)(n_slots > 0 ? n_slots : 1), f4 = part4, f16 = part16 > 0.0f ? part16 : 2.0;
        const long long spare = (long long)n_entries - n;
        int first = 0;
        while (first < RTC_ORDER_BUCKETS && hist[first] == 0) first++;
        double best = 1e300; int best_cut = first;
        // cut: buckets first .. cut-1 are split, T = ceiling of bucket `cut` (= floor of bucket cut-1; the dearest cost for cut = first)
        for (int cut = first; cut < RTC_ORDER_BUCKETS; cut++) {
            const double T = cut == first ? order_bucket_floor(first) * 1.19 : order_bucket_floor(cut - 1);
            if (T < 1000.0) break;                                              // never below 10 us
            double sum = (double)total; long long need = 0; bool ok = true;
            for (int b = first; b < cut && ok; b++) {
                if (!hist[b]) continue;
                const double ceil_b = order_bucket_floor(b) * 1.19;             // a quarter octave
                const int k = ceil_b * f4 <= T ? 4 : 16;
                if (k == 16 && ceil_b * f16 > T) ok = false;
                need += (long long)hist[b] * (k - 1);
                sum += (double)bsum[b] * (k == 4 ? 4.0 * f4 - 1.0 : 16.0 * f16 - 1.0);
            }
            if (!ok || need > spare) break;
            const double est = fmax(T, 1.1 * sum / slots);
            if (force_frac > 0.0f) { if (T >= force_frac * order_bucket_floor(first) * 1.19) best_cut = cut; continue; }   // development: RTC_SPLIT_T
            if (est < best * 0.97) { best = est; best_cut = cut; }             // a split has to promise 3 % to be worth doing
        }
        const double T = best_cut == first ? 1e300 : order_bucket_floor(best_cut - 1);
        for (int b = first; b < best_cut; b++) ways[b] = (unsigned char)(order_bucket_floor(b) * 1.19 * f4 <= T ? 4 : 16);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < RTC_ORDER_BUCKETS; b += blockDim.x) hist[b] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        uint32_t c = cost[i];
        unsigned int k = ways[order_bucket(c)];
        uint32_t pc = k == 1 ? c : (uint32_t)((float)c * (k == 4 ? part4 : part16));
        atomicAdd(&hist[order_bucket(pc)], k);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int run = 0;
        for (int b = 0; b < RTC_ORDER_BUCKETS; b++) { cursor[b] = run; run += hist[b]; }
        total = run;
    }
    __syncthreads();
    // (Tried, not kept: spreading the cheapest workgroups -- sky, a few us each, the last 100 us of a descending order at a third
    // of the wave slots -- among the middle ones: every cheap workgroup dispatched earlier delays a dearer one behind it; and lists
    // of four cheap units behind one entry, rendered by one wave in a loop: the loop around the pixel code cost the fp64 kernel
    // 50 more spilled registers, dragons -2 %, teapot / earth / cover +4 ... +13 %.)
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        uint32_t c = cost[i];
        unsigned int k = ways[order_bucket(c)];
        uint32_t pc = k == 1 ? c : (uint32_t)((float)c * (k == 4 ? part4 : part16));
        unsigned int at = atomicAdd(&cursor[order_bucket(pc)], k);
        if (k == 1) order[at] = i * 32;
        else for (unsigned int q = 0; q < k; q++) order[at + q] = i * 32 + (k == 4 ? 1 : 5) + (int)q;
    }
    for (int i = (int)total + threadIdx.x; i < n_entries; i += blockDim.x) order[i] = -1;
    if (threadIdx.x == 0) *used = (int32_t)total;      // the host sizes later grids by it once it has arrived
}

// canvas.py:133-156
__global__ void quantize_kernel(const double* rgb, int hsize, int vsize, double maxc, uint8_t* out) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long n = (long long)hsize * vsize * 3;
    if (i >= n) return;
    int c = (int)(i % 3);
    long long p = i / 3;
    int x = (int)(p % hsize), y = (int)(p / hsize);
    double v = rgb[((size_t)x * vsize + y) * 3 + c];
    int q;
    if (v >= maxc) q = 255;
    else if (v <= 0) q = 0;
    else q = (int)(v * 255 / maxc + 0.5);
    out[i] = (uint8_t)q;
}

// canvas.py:137-145: np.max over every component.  The result starts as component 0; then one
// block-level max per workgroup and a compare-and-swap loop that only ever raises it.
__global__ void frame_max_init_kernel(const double* rgb, double* out) { *out = rgb[0]; }
__global__ void frame_max_kernel(const double* rgb, long long n, double* out) {
    __shared__ double red[256 / 64];
    double m = -INFINITY;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        double v = rgb[i];
        if (v > m) m = v;
    }
    for (int off = 32; off > 0; off >>= 1) { double o = __shfl_down(m, off, 64); if (o > m) m = o; }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int q = 1; q < 256 / 64; q++) if (red[q] > m) m = red[q];
        // CAS loop: *out = max(*out, m)
        unsigned long long* p = reinterpret_cast<unsigned long long*>(out);
        unsigned long long seen = *p;
        while (m > __longlong_as_double((long long)seen)) {
            unsigned long long prev = atomicCAS(p, seen, (unsigned long long)__double_as_longlong(m));
            if (prev == seen) break;
            seen = prev;
        }
    }
}

// canvas.py:133-156 with max_component on the device
__global__ void ppm_body_kernel(const double* rgb, int hsize, int vsize, const double* max_dev, uint8_t* out) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long n = (long long)hsize * vsize * 3;
    if (i >= n) return;
    const double maxc = max_dev ? *max_dev : 1.0;
    int c = (int)(i % 3);
    long long p = i / 3;
    int x = (int)(p % hsize), y = (int)(p / hsize);
    double v = rgb[((size_t)x * vsize + y) * 3 + c];
    int q;
    if (v >= maxc) q = 255;
    else if (v <= 0) q = 0;
    else q = (int)(v * 255 / maxc + 0.5);
    out[i] = (uint8_t)q;
}

// compact tiles -> canvas (one thread = one pixel of one slot)
__global__ void scatter_tiles_kernel(const double* tiles, const int32_t* tile_ids, int n_slots, int T, int tiles_x, int hsize, int vsize, double* frame) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int per = T * T;
    if (i >= (long long)n_slots * per) return;
    int slot = (int)(i / per), within = (int)(i % per);
    int tile = tile_ids[slot];
    if (tile < 0) return;
    int x = (tile % tiles_x) * T + within % T, y = (tile / tiles_x) * T + within / T;
    if (x >= hsize || y >= vsize) return;
    const double* src = tiles + 3 * (size_t)i;
    double* dst = frame + 3 * ((size_t)x * vsize + y);
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
}

// the same for the hit-id map (one int32 per pixel)
__global__ void scatter_id_tiles_kernel(const int32_t* tiles, const int32_t* tile_ids, int n_slots, int T, int tiles_x, int hsize, int vsize, int32_t* frame) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int per = T * T;
    if (i >= (long long)n_slots * per) return;
    int slot = (int)(i / per), within = (int)(i % per);
    int tile = tile_ids[slot];
    if (tile < 0) return;
    int x = (tile % tiles_x) * T + within % T, y = (tile / tiles_x) * T + within / T;
    if (x >= hsize || y >= vsize) return;
    frame[(size_t)x * vsize + y] = tiles[i];
}

// ---------------------------------------------------------------------------------------------
// device buffers
// ---------------------------------------------------------------------------------------------
struct DeviceArena {
    std::vector<void*> ptrs;
    size_t bytes = 0;
    template <class T> int upload(const T* host, size_t n, const T** out) {
        void* d = nullptr;
        size_t sz = (n ? n : 1) * sizeof(T);
        HIP_TRY(hipMalloc(&d, sz));
        ptrs.push_back(d);
        bytes += sz;
        if (n) HIP_TRY(hipMemcpy(d, host, n * sizeof(T), hipMemcpyHostToDevice));
        *out = static_cast<const T*>(d);
        return 0;
    }
    void release() {
        for (void* p : ptrs) (void)hipFree(p);
        ptrs.clear();
    }
};

template <class R> std::vector<R> cast_vec(const double* p, size_t n) {
    std::vector<R> v(n);
    for (size_t i = 0; i < n; i++) v[i] = (R)p[i];
    return v;
}

// boxes padded outward so that neither rounding in the compiler nor in the slab test can
// reject a leaf the leaf test itself would accept
#define RTC_PAD_OF_EXTENT 8e-7     /* boxes are padded by at least this much of the largest coordinate of the scene (see traverse_bvh) */
template <class R> std::vector<R> pad_boxes(const std::vector<double>& box, double pad_abs) {
    const double rel = sizeof(R) == 4 ? 2e-5 : 1e-9;
    std::vector<R> out(box.size());
    for (size_t i = 0; i < box.size(); i++) {
        bool is_hi = (i % 6) >= 3;
        double v = box[i];
        if (std::isfinite(v)) {
            double pad = std::max(rel * std::max(1.0, std::fabs(v)), pad_abs);
            v = is_hi ? v + pad : v - pad;
            R r = (R)v;
            if (is_hi && (double)r < v) r = std::nextafter(r, std::numeric_limits<R>::infinity());
            if (!is_hi && (double)r > v) r = std::nextafter(r, -std::numeric_limits<R>::infinity());
            out[i] = r;
        } else {
            out[i] = (R)v;
        }
    }
    return out;
}

}  // namespace

// What one render call needs beyond the scene: the launch-order feedback of its view.  A scene
// handle keeps a list of these; a call takes one that is free -- last used on the call's own stream (stream order
// protects it) or finished (its event has completed) -- so calls from several host threads on several streams
// never share one.
struct FbKey { rtc_camera cam; int32_t precision, depth, tile_size, n_tiles; const int32_t* tile_ids; };
struct Workspace {
    bool claimed = false;                 // a host thread is enqueueing with it right now
    hipStream_t stream = nullptr; hipEvent_t done = nullptr; bool used = false;
    // launch-order feedback: per-unit cost of the last frame, two order buffers
    DeviceArena fb_mem;
    uint32_t* fb_cost = nullptr; int32_t* fb_order[2] = {nullptr, nullptr};
    long long fb_units = 0; int fb_cur = 0; bool fb_valid = false;
    int32_t* fb_used_dev = nullptr; int32_t* fb_used_host = nullptr;   // entries the current order really has (host copy: pinned)
    uint32_t* ray_parts = nullptr;        // [n_entries][8]: the ray counts of every wave of the last launch (DFrame::work in the variants that count no work)
    hipEvent_t fb_used_event = nullptr; bool fb_used_known = false;
    // fp64 without CSG exists for 3 and for 2 waves per SIMD (trace_kernel's WAVES): once a view's order has settled, four
    // frames are timed -- 3, 2, 3, 2 waves; events around the launch, read when they have completed -- and the budget with the
    // faster best frame is kept for the view (the first launch of a variant is slower than its later ones)
    hipEvent_t tune_ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int tune_stage = 0;          // 0: order not settled yet, 1..4: the next frame is timed (odd: 3 waves, even: 2), 5: waiting for the events, 6: decided
    bool tune_two_waves = false;
    // what the current order was measured on, and how many frames of it have been sorted: the order is rebuilt
    // after the first two frames of a view (the second one measured under the new order) and then kept
    FbKey fb_key{}; int fb_age = 0;
    void release() {
        fb_mem.release();
        if (fb_used_host) (void)hipHostFree(fb_used_host);
        if (fb_used_event) (void)hipEventDestroy(fb_used_event);
        if (done) (void)hipEventDestroy(done);
        for (hipEvent_t e : tune_ev) if (e) (void)hipEventDestroy(e);
    }
};

struct rtc_scene {
    int device = 0;
    int n_leaves = 0;
    rtc::Bvh bvh_info;   // sizes only (vectors cleared after upload)
    int bvh_nodes = 0, bvh_depth = 0, bvh_max_leaf = 0, obj_depth = 0, obj_max_leaf = 0;
    DeviceArena mem64, mem32, shared;
    rtc::DScene<double> d64;
    rtc::DScene<float> d32;
    bool has_jitter = false, has_csg = false, has_csg_stream = false;
    long long light_cells = 0;      // shadow rays per shaded hit: 1 per point light, usteps x vsteps per area light
    std::mutex ws_mutex;      // guards the workspace list
    std::vector<Workspace*> workspaces;
    int n_cus = 0;
    bool has_secondary = false;   // some material reflects or refracts
    // what a GPU rebuild of the triangle BVH needs: the triangle items (prim codes, item order) and the
    // arena that owns the current triangle BVH (nodes, BVH-ordered rows in both precisions, leaf ids)
    int32_t* d_titems = nullptr; int n_titems = 0; long long n_tris = 0;
    DeviceArena tbvh_mem;
    // for refits (same arena): triangle row at each BVH position, parent link of each node, arrival counters
    int32_t* d_tb_row = nullptr; int32_t* d_tparent = nullptr; uint32_t* d_tarrived = nullptr;
    rtc_launch_info last_launch{};      // what the most recent launch was (rtc_scene_last_launch; reporting only)
    int n_tnodes4 = 0;       // records in d64.tnodes4 (host-built: compact; GPU-built: one per binary node, unused ones empty)
    double max_abs = 0;      // largest coordinate of any box the scene's trees were padded for (RTC_PAD_OF_EXTENT)
    // device-side frame, hit ids, counters and tile list of the host-buffer entry points, kept between calls
    std::mutex out_mutex;
    double* out_rgb = nullptr; int32_t* out_id = nullptr; int64_t* out_rays = nullptr; int32_t* out_tiles = nullptr;
    size_t out_rgb_px = 0, out_id_px = 0, out_tiles_n = 0;
    uint8_t* out_ppm = nullptr; size_t out_ppm_bytes = 0;
    hipEvent_t out_e0 = nullptr, out_e1 = nullptr;
};

namespace {

template <class R>
int upload_real(const rtc_scene_desc& d, const std::vector<double>& quadric_minmax, DeviceArena& A, rtc::DScene<R>& S) {
#define UP_REAL(field, src, count)                                             \
    do {                                                                       \
        std::vector<R> tmp_ = cast_vec<R>((src), (size_t)(count));             \
        if (int rc_ = A.upload(tmp_.data(), tmp_.size(), &S.field)) return rc_; \
    } while (0)
    UP_REAL(leaf_minmax, quadric_minmax.data(), quadric_minmax.size());      // one row per cylinder / cone (create_impl), not per leaf
    UP_REAL(xform, d.xform, 12 * (size_t)d.n_xforms);
    UP_REAL(tri_geom, d.tri_geom, 12 * (size_t)d.n_tris);
    UP_REAL(tri_norm, d.tri_norm, 12 * (size_t)d.n_tris);
    UP_REAL(mesh_norm, d.mesh_norm, 12 * (size_t)d.n_mesh_tris);
    UP_REAL(csg_bbox, d.csg_bbox, 6 * (size_t)d.n_csg_nodes);
    UP_REAL(mat, d.mat, 10 * (size_t)d.n_materials);
    UP_REAL(pat_inv, d.pat_inv, 12 * (size_t)d.n_pat_nodes);
    UP_REAL(pat_ab, d.pat_ab, 6 * (size_t)d.n_pat_nodes);
    UP_REAL(pat_param, d.pat_param, 4 * (size_t)d.n_pat_nodes);
    UP_REAL(pat_colors, d.pat_colors, 15 * (size_t)d.n_pat_nodes);
    UP_REAL(tex_scale, d.tex_scale, (size_t)d.n_textures);
    UP_REAL(light_geom, d.light_geom, 9 * (size_t)d.n_lights);
    UP_REAL(light_intensity, d.light_intensity, 3 * (size_t)d.n_lights);
#undef UP_REAL
    return 0;
}

// the box of a whole tree = the union of its root's two child boxes (all zero for an empty tree)
void root_box(const rtc::BvhNode* root, float out[6]) {
    for (int a = 0; a < 6; a++) out[a] = 0.0f;
    if (!root) return;
    bool any = false;
    for (int side = 0; side < 2; side++) {
        if (root->child[side] == RTC_EMPTY_CHILD) continue;
        for (int a = 0; a < 3; a++) {
            out[a] = any ? std::min(out[a], root->box[6 * side + a]) : root->box[6 * side + a];
            out[3 + a] = any ? std::max(out[3 + a], root->box[6 * side + 3 + a]) : root->box[6 * side + 3 + a];
        }
        any = true;
    }
}
// after a GPU build or refit: the new root boxes come back to the host (64 bytes; the builders have been waited for)
// *grew: the triangles now reach coordinates the boxes' padding was not made for (the caller pads and builds once more)
int refresh_triangle_root_box(rtc_scene* sc, bool* grew) {
    rtc::BvhNode root;
    HIP_TRY(hipMemcpy(&root, sc->d64.tnodes, sizeof(root), hipMemcpyDeviceToHost));
    root_box(&root, sc->d64.tbox);
    memcpy(sc->d32.tbox, sc->d64.tbox, sizeof(sc->d64.tbox));
    double m = 0;
    for (int a = 0; a < 6; a++) m = std::max(m, (double)std::fabs(sc->d64.tbox[a]));
    if (!std::isfinite(m)) return fail(RTC_ERR_INVALID, "triangle rows are not finite");
    // max_abs was taken from the boxes before padding, the root box here is padded (by at most 2e-5 relative, 2e-5 absolute,
    // pad_boxes / lbvh_boxes_kernel): only an extent beyond that margin counts as growth.  far_origin stays where it is: it
    // is tied to the extent the OBJECT boxes were padded for, which a triangle update does not re-pad -- a lower far_origin
    // only re-bases more rays (FAR variants), it never culls.
    *grew = m > sc->max_abs + 4e-5 * std::max(1.0, sc->max_abs);
    if (*grew) sc->max_abs = 1.5 * m;
    return 0;
}

int validate(const rtc_scene_desc& d) {
    if (d.n_leaves < 0 || d.n_xforms < 0 || d.n_tris < 0 || d.n_items < 0 || d.n_unbounded < 0 || d.n_materials < 0 ||
        d.n_lights < 0 || d.n_csg_roots < 0 || d.n_csg_nodes < 0 || d.n_pat_nodes < 0 || d.n_programs < 0 ||
        d.n_instr < 0 || d.n_textures < 0)
        return fail(RTC_ERR_INVALID, "negative count in scene description");
    if (d.n_leaves > 0x07FFFFFF || d.n_tris > 0x07FFFFFF) return fail(RTC_ERR_LIMIT, "more than 2^27-1 leaves");
    if (d.n_items > (1 << 27)) return fail(RTC_ERR_LIMIT, "more than 2^27 BVH items");
    for (int i = 0; i < d.n_leaves; i++) {
        int k = d.leaf_kind[i];
        if (k < RTC_SPHERE || k > RTC_TRIANGLE) return fail(RTC_ERR_INVALID, "leaf %d: unknown kind %d", i, k);
        if (d.leaf_xform[i] < 0 || d.leaf_xform[i] >= d.n_xforms) return fail(RTC_ERR_INVALID, "leaf %d: xform row out of range", i);
        if (d.leaf_material[i] < 0 || d.leaf_material[i] >= d.n_materials) return fail(RTC_ERR_INVALID, "leaf %d: material out of range", i);
        if (k == RTC_TRIANGLE && d.leaf_tri[i] <= -2) {                      // a triangle of an instanced mesh: -2 - (mesh row)
            if (-2 - (long long)d.leaf_tri[i] >= d.n_mesh_tris) return fail(RTC_ERR_INVALID, "leaf %d: mesh row out of range", i);
        } else if (k == RTC_TRIANGLE && (d.leaf_tri[i] < 0 || d.leaf_tri[i] >= d.n_tris)) return fail(RTC_ERR_INVALID, "leaf %d: triangle row out of range", i);
    }
    for (int i = 0; i < d.n_tris; i++)
        if (d.tri_leaf[i] < 0 || d.tri_leaf[i] >= d.n_leaves) return fail(RTC_ERR_INVALID, "triangle %d: leaf out of range", i);
    auto check_prim = [&](int32_t code, const char* what, int i) -> int {
        rtc::PrimCode* dummy = nullptr; (void)dummy;
        int kind = (int)((uint32_t)code >> 28), idx = (int)((uint32_t)code & 0x07FFFFFFu);
        if (kind == RTC_TRIANGLE) { if (idx >= d.n_tris) return fail(RTC_ERR_INVALID, "%s %d: triangle row out of range", what, i); }
        else if (kind == RTC_CSG_ROOT) { if (idx >= d.n_csg_roots) return fail(RTC_ERR_INVALID, "%s %d: CSG root out of range", what, i); }
        else if (kind == RTC_INSTANCE) { if (idx >= d.n_instances) return fail(RTC_ERR_INVALID, "%s %d: instance out of range", what, i); }
        else if (kind <= RTC_CONE) { if (idx >= d.n_leaves || d.leaf_kind[idx] != kind) return fail(RTC_ERR_INVALID, "%s %d: leaf/kind mismatch", what, i); }
        else return fail(RTC_ERR_INVALID, "%s %d: unknown prim kind %d", what, i, kind);
        return 0;
    };
    for (int i = 0; i < d.n_items; i++) {
        if (int rc = check_prim(d.item_prim[i], "item", i)) return rc;
        for (int k = 0; k < 3; k++)
            if (!std::isfinite(d.item_lo[3 * i + k]) || !std::isfinite(d.item_hi[3 * i + k]))
                return fail(RTC_ERR_INVALID, "item %d: non-finite bounds (unbounded things go in unbounded_prim)", i);
    }
    for (int i = 0; i < d.n_unbounded; i++)
        if (int rc = check_prim(d.unbounded_prim[i], "unbounded", i)) return rc;
    if (d.n_mesh_tris < 0 || d.n_meshes < 0 || d.n_instances < 0) return fail(RTC_ERR_INVALID, "negative count in scene description");
    if (d.n_instances > 0 && d.n_csg_roots > 0) return fail(RTC_ERR_UNSUPPORTED, "instanced meshes in a scene with CSG objects (the scene compiler flattens them)");
    for (int i = 0; i < d.n_instr && d.n_instances > 0; i++)
        if (d.instr[2 * (size_t)i] == RTC_OP_PERTURB) return fail(RTC_ERR_UNSUPPORTED, "instanced meshes in a scene with perturbed patterns");
    if (d.n_instances > RTC_MAX_INSTANCES) return fail(RTC_ERR_LIMIT, "%d mesh instances, limit %d (flatten the rest)", d.n_instances, RTC_MAX_INSTANCES);
    for (int i = 0; i < d.n_meshes; i++) {
        const int32_t* r = d.mesh_range + 2 * i;
        if (r[0] < 0 || r[1] < 1 || (long long)r[0] + r[1] > d.n_mesh_tris) return fail(RTC_ERR_INVALID, "mesh %d: rows out of range", i);
    }
    for (int i = 0; i < d.n_instances; i++) {
        const int32_t* r = d.inst + 4 * i;
        if (r[0] < 0 || r[0] >= d.n_meshes || r[1] < 0 || r[1] >= d.n_xforms || r[3] < 0 || r[3] >= d.n_materials ||
            r[2] < 0 || (long long)r[2] + d.mesh_range[2 * r[0] + 1] > d.n_leaves)
            return fail(RTC_ERR_INVALID, "instance %d malformed", i);
    }
    for (int i = 0; i < d.n_unbounded; i++)
        if (((uint32_t)d.unbounded_prim[i] >> 28) == RTC_INSTANCE) return fail(RTC_ERR_INVALID, "an instance cannot be unbounded");
    for (int i = 0; i < d.n_csg_roots; i++) {
        const int32_t* r = d.csg_root + 4 * i;
        if (r[0] < 0 || r[1] < 1 || r[0] + r[1] > d.n_csg_nodes || r[2] < 0 || r[3] > d.n_leaves || r[2] > r[3])
            return fail(RTC_ERR_INVALID, "CSG root %d: ranges out of bounds", i);
        int max_hits = 0;
        for (int l = r[2]; l < r[3]; l++) {
            int k = d.leaf_kind[l];
            max_hits += (k == RTC_PLANE || k == RTC_TRIANGLE) ? 1 : ((k == RTC_SPHERE || k == RTC_CUBE) ? 2 : 4);
        }
        if (r[1] > RTC_MAX_CSG_STREAM_NODES && max_hits > RTC_MAX_CSG_HITS)      // (such roots are evaluated without a hit list, csg_stream)
            return fail(RTC_ERR_LIMIT, "CSG root %d nests %d CSG operations over leaves that can produce %d hits per ray: limit %d operations",
                        i, r[1], max_hits, RTC_MAX_CSG_STREAM_NODES);
    }
    for (int i = 0; i < d.n_csg_nodes; i++) {
        const int32_t* c = d.csg_node + 5 * i;
        if (c[0] < 0 || c[0] > 2 || c[1] > c[2] || c[2] > c[3] || c[1] < 0 || c[3] > d.n_leaves || c[4] < 0 || c[4] >= d.n_xforms)
            return fail(RTC_ERR_INVALID, "CSG node %d malformed", i);
    }
    for (int i = 0; i < d.n_materials; i++)
        if (d.mat_program[i] >= d.n_programs) return fail(RTC_ERR_INVALID, "material %d: program out of range", i);
    for (int i = 0; i < d.n_programs; i++) {
        if (d.prog_off[i] < 0 || d.prog_len[i] < 1 || d.prog_off[i] + d.prog_len[i] > d.n_instr)
            return fail(RTC_ERR_INVALID, "pattern program %d out of range", i);
        int sp = 0, peak = 0, moved = 0;
        for (int k = d.prog_off[i]; k < d.prog_off[i] + d.prog_len[i]; k++) {
            int op = d.instr[2 * k], node = d.instr[2 * k + 1];
            if (op == RTC_OP_PERTURB) {
                if (node < 0 || node >= d.n_pat_nodes || d.pat_type[node] != RTC_PAT_PERTURBED)
                    return fail(RTC_ERR_INVALID, "pattern program %d: PERTURB needs a perturbed-pattern node", i);
                const double* q = d.pat_param + 4 * (size_t)node;
                if (!(q[0] != 0.0) || !(q[2] >= 1.0 && q[2] <= 16.0) || q[2] != std::floor(q[2]))
                    return fail(RTC_ERR_INVALID, "pattern node %d: frequency must not be 0, octaves must be 1..16", node);
                if (++moved > RTC_PERTURB_DEPTH) return fail(RTC_ERR_LIMIT, "pattern program %d nests perturbed patterns deeper than %d", i, RTC_PERTURB_DEPTH);
                continue;
            }
            if (op == RTC_OP_RESTORE) {
                if (--moved < 0) return fail(RTC_ERR_INVALID, "pattern program %d: RESTORE without PERTURB", i);
                continue;
            }
            if (op == RTC_OP_BLEND) { sp -= 1; }
            else if (op == RTC_OP_EVAL || op == RTC_OP_EVAL_AB) {
                if (node < 0 || node >= d.n_pat_nodes) return fail(RTC_ERR_INVALID, "pattern program %d: node out of range", i);
                int t = d.pat_type[node];
                bool known = (t >= RTC_PAT_TEST && t <= RTC_PAT_STRIPE) || (t >= RTC_PAT_MAP_PLANAR && t <= RTC_PAT_MAP_CUBE);
                if (!known) return fail(RTC_ERR_UNSUPPORTED, "pattern program %d: node type %d cannot be evaluated at a point", i, t);
                int nch = t == RTC_PAT_MAP_CUBE ? 6 : (t == RTC_PAT_MAP_CYLINDRICAL ? 2 : (t >= RTC_PAT_MAP_PLANAR ? 1 : 0));
                for (int q = 0; q < nch; q++) {
                    int ch = d.pat_child[6 * node + q];
                    if (ch < 0 || ch >= d.n_pat_nodes) return fail(RTC_ERR_INVALID, "pattern node %d: uv child out of range", node);
                    int ct = d.pat_type[ch];
                    if (ct < RTC_UV_CHECKER || ct > RTC_UV_IMAGE) return fail(RTC_ERR_INVALID, "pattern node %d: child is not a uv pattern", node);
                    if (ct == RTC_UV_IMAGE && (d.pat_tex[ch] < 0 || d.pat_tex[ch] >= d.n_textures))
                        return fail(RTC_ERR_INVALID, "pattern node %d: texture out of range", ch);
                }
                sp += (op == RTC_OP_EVAL) ? 1 : -1;
            } else return fail(RTC_ERR_INVALID, "pattern program %d: unknown opcode %d", i, op);
            if (sp < 1) return fail(RTC_ERR_INVALID, "pattern program %d underflows its stack", i);
            peak = std::max(peak, sp);
        }
        if (sp != 1) return fail(RTC_ERR_INVALID, "pattern program %d leaves %d values", i, sp);
        if (moved != 0) return fail(RTC_ERR_INVALID, "pattern program %d: PERTURB without RESTORE", i);
        if (peak > RTC_PAT_STACK) return fail(RTC_ERR_LIMIT, "pattern program %d needs a stack of %d, limit %d", i, peak, RTC_PAT_STACK);
    }
    for (int i = 0; i < d.n_textures; i++)
        if (d.tex_w[i] < 1 || d.tex_h[i] < 1 || d.tex_off[i] < 0 || d.tex_off[i] + 3LL * d.tex_w[i] * d.tex_h[i] > d.tex_bytes)
            return fail(RTC_ERR_INVALID, "texture %d out of range", i);
    for (int i = 0; i < d.n_lights; i++)
        if (d.light_type[i] == 1 && (d.light_steps[3 * i] < 1 || d.light_steps[3 * i + 1] < 1))
            return fail(RTC_ERR_INVALID, "area light %d: steps must be >= 1", i);
    return 0;
}

// split units dearer than these x (total cost / wave slots) into 4 / 16 waves; 0 = never (development overrides: RTC_SPLIT4, RTC_SPLIT16)
float g_split4 = [] { const char* e = getenv("RTC_SPLIT4"); return e ? (float)atof(e) : 0.5f; }();     // cost of a 4x4 part / cost of its unit; 0 = never split
const bool g_replicas = [] { const char* e = getenv("RTC_REPLICAS"); return !e || atoi(e) != 0; }();     // RTC_REPLICAS=0: parts leave their spare lanes idle
const bool g_ray_parts = [] { const char* e = getenv("RTC_RAY_PARTS"); return !e || atoi(e) != 0; }();   // RTC_RAY_PARTS=0: ray counts through atomics from every workgroup
const bool g_tune_waves = [] { const char* e = getenv("RTC_TUNE_WAVES"); return !e || atoi(e) != 0; }();   // RTC_TUNE_WAVES=0: fp64 always the variant built for RTC_WAVES_F64 waves per SIMD
float g_split_t = [] { const char* e = getenv("RTC_SPLIT_T"); return e ? (float)atof(e) : 0.0f; }();
float g_split16 = [] { const char* e = getenv("RTC_SPLIT16"); return e ? (float)atof(e) : 0.35f; }();     // the same for a 2x2 part

int rebuild_triangle_bvh_device(rtc_scene* sc, bool again = false);

// development (RTC_BVH_INFO=1): where rtc_scene_create's time goes
struct CreateClock {
    const bool on = getenv("RTC_BVH_INFO") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void tick(const char* what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[rtc] create: %-28s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};

int create_impl(const rtc_scene_desc& d, rtc_scene* sc) {
    CreateClock clk;
    {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, sc->device));
        sc->n_cus = prop.multiProcessorCount;
    }
    // ---- two BVHs over the bounded items (host): analytic leaves + CSG roots, and triangles ----
    std::vector<rtc::Aabb> obox, tbox;
    std::vector<int32_t> oprim, tprim;
    for (int i = 0; i < d.n_items; i++) {
        rtc::Aabb b;
        for (int k = 0; k < 3; k++) { b.lo[k] = d.item_lo[3 * i + k]; b.hi[k] = d.item_hi[3 * i + k]; }
        bool is_tri = ((uint32_t)d.item_prim[i] >> 28) == RTC_TRIANGLE;
        (is_tri ? tbox : obox).push_back(b);
        (is_tri ? tprim : oprim).push_back(d.item_prim[i]);
    }
    int tri_leaf = 2;                                   // triangles per BVH leaf (development override: RTC_BVH_LEAF)
    if (const char* e = getenv("RTC_BVH_LEAF")) { int v = atoi(e); if (v >= 1 && v <= 16) tri_leaf = v; }
    // analytic prims per object-BVH leaf: an analytic test (fp64 ray transform + quadric / slab arithmetic) costs several
    // node visits, so one prim per leaf -- unless the handful of objects fits one leaf anyway (measured: cover -14 %
    // with 1, shadow_glamour_shot +7 % when its four objects are spread over a tree).  Development override: RTC_OBJ_LEAF.
    int obj_leaf = obox.size() <= 4 ? 4 : 1;
    if (const char* e = getenv("RTC_OBJ_LEAF")) { int v = atoi(e); if (v >= 1 && v <= 16) obj_leaf = v; }
    rtc::Bvh bvh = rtc::build_bvh(obox, obj_leaf, RTC_BVH_STACK - 2);
    // RTC_BUILD_ON_DEVICE: no triangle tree is built here; the rows go up and rebuild_triangle_bvh_device makes it (end of this function)
    // (only for meshes the host build takes longer over than the GPU build's fixed costs -- some thirty allocations, fifty rounds of
    // small kernels, ~10 ms the first time: teapot, 6 320 triangles, host 3.5 ms)
    const bool device_build = (d.build_flags & RTC_BUILD_ON_DEVICE) != 0 && tprim.size() >= 16384;
    double tri_max_abs = 0;
    if (device_build) {
        for (const rtc::Aabb& b : tbox)
            for (int k = 0; k < 3; k++) tri_max_abs = std::max(tri_max_abs, std::max(std::fabs(b.lo[k]), std::fabs(b.hi[k])));
        tbox.clear();
    }
    std::vector<int32_t> tprim_all;
    if (device_build) tprim_all.swap(tprim);
    rtc::Bvh tbvh = rtc::build_bvh(tbox, tri_leaf, RTC_BVH_STACK - 2);
    if (bvh.depth > RTC_BVH_STACK - 1 || tbvh.depth > RTC_BVH_STACK - 1) return fail(RTC_ERR_LIMIT, "BVH depth exceeds traversal stack");
    sc->bvh_nodes = bvh.n_nodes() + tbvh.n_nodes(); sc->bvh_depth = std::max(bvh.depth, tbvh.depth);
    sc->bvh_max_leaf = std::max(bvh.max_leaf, tbvh.max_leaf);
    sc->obj_depth = bvh.depth; sc->obj_max_leaf = bvh.max_leaf;
    clk.tick("host trees");
    std::vector<int32_t> prims(oprim.size());
    for (size_t i = 0; i < oprim.size(); i++) prims[i] = oprim[bvh.order[i]];
    // triangles in BVH leaf order: geometry rows + (leaf id | noshadow << 31)
    std::vector<double> tb_geom(12 * tprim.size());
    std::vector<int32_t> tb_leaf(tprim.size()), tb_row(tprim.size());
    for (size_t i = 0; i < tprim.size(); i++) {
        uint32_t code = (uint32_t)tprim[tbvh.order[i]];
        uint32_t row = code & 0x07FFFFFFu;
        tb_row[i] = (int32_t)row;
        memcpy(&tb_geom[12 * i], d.tri_geom + 12 * (size_t)row, 12 * sizeof(double));
        tb_leaf[i] = (int32_t)((uint32_t)d.tri_leaf[row] | (((code >> 27) & 1u) << 31));
    }

    // ---- precision-independent arrays (uploaded once, shared by both DScene views) ----
    rtc::DScene<double>& S = sc->d64;
    memset(&sc->d64, 0, sizeof(sc->d64));
    memset(&sc->d32, 0, sizeof(sc->d32));
    DeviceArena& A = sc->shared;
#define UP(field, src, count) do { if (int rc_ = A.upload((src), (size_t)(count), &S.field)) return rc_; } while (0)
    UP(leaf_kind, d.leaf_kind, d.n_leaves);
    UP(leaf_xform, d.leaf_xform, d.n_leaves);
    UP(leaf_material, d.leaf_material, d.n_leaves);
    // minimum / maximum / closed exist for cylinders and cones only: one row each in compact tables, found through the leaf's
    // (otherwise unused) leaf_tri entry -- per-leaf arrays of them were 28 bytes a leaf, 4 MB of the dragons' device memory
    std::vector<int32_t> leaf_tri_dev(d.leaf_tri, d.leaf_tri + d.n_leaves), quadric_closed;
    std::vector<double> quadric_minmax;
    for (int i = 0; i < d.n_leaves; i++)
        if (d.leaf_kind[i] == RTC_CYLINDER || d.leaf_kind[i] == RTC_CONE) {
            leaf_tri_dev[i] = (int32_t)quadric_closed.size();
            quadric_minmax.push_back(d.leaf_minmax[2 * (size_t)i]); quadric_minmax.push_back(d.leaf_minmax[2 * (size_t)i + 1]);
            quadric_closed.push_back(d.leaf_closed[i]);
        }
    UP(leaf_tri, leaf_tri_dev.data(), leaf_tri_dev.size());
    UP(leaf_closed, quadric_closed.data(), quadric_closed.size());
    UP(tri_leaf, d.tri_leaf, d.n_tris);
    UP(bvh_prims, prims.data(), prims.size());
    if (int rc_ = sc->tbvh_mem.upload(tb_leaf.data(), tb_leaf.size(), &S.tb_leaf)) return rc_;
    {
        std::vector<int32_t> tparent((size_t)tbvh.n_nodes(), -1);
        for (int i = 0; i < tbvh.n_nodes(); i++)
            for (int side = 0; side < 2; side++) {
                int32_t c = tbvh.child[2 * (size_t)i + side];
                if (c >= 0) tparent[c] = (i << 1) | side;
            }
        std::vector<uint32_t> zeros((size_t)tbvh.n_nodes(), 0u);
        const int32_t* p_row = nullptr; const int32_t* p_par = nullptr; const uint32_t* p_arr = nullptr;
        if (int rc_ = sc->tbvh_mem.upload(tb_row.data(), tb_row.size(), &p_row)) return rc_;
        if (int rc_ = sc->tbvh_mem.upload(tparent.data(), tparent.size(), &p_par)) return rc_;
        if (int rc_ = sc->tbvh_mem.upload(zeros.data(), zeros.size(), &p_arr)) return rc_;
        sc->d_tb_row = const_cast<int32_t*>(p_row); sc->d_tparent = const_cast<int32_t*>(p_par); sc->d_tarrived = const_cast<uint32_t*>(p_arr);
    }
    {
        const int32_t* items_dev = nullptr;
        const std::vector<int32_t>& titems = device_build ? tprim_all : tprim;
        if (int rc_ = A.upload(titems.data(), titems.size(), &items_dev)) return rc_;
        sc->d_titems = const_cast<int32_t*>(items_dev); sc->n_titems = (int)titems.size(); sc->n_tris = d.n_tris;
    }
    UP(unbounded_prims, d.unbounded_prim, d.n_unbounded);
    {
        // roots whose leaves could produce more hits on one ray than the per-lane list holds are evaluated without one
        std::vector<int32_t> roots(d.csg_root, d.csg_root + 4 * (size_t)d.n_csg_roots);
        for (int i = 0; i < d.n_csg_roots; i++) {
            int max_hits = 0;
            for (int l = roots[4 * i + 2]; l < roots[4 * i + 3]; l++) {
                int k = d.leaf_kind[l];
                max_hits += (k == RTC_PLANE || k == RTC_TRIANGLE) ? 1 : ((k == RTC_SPHERE || k == RTC_CUBE) ? 2 : 4);
            }
            if (max_hits > RTC_MAX_CSG_HITS || getenv("RTC_CSG_STREAM_ALL")) { roots[4 * i + 1] |= RTC_CSG_STREAMED; sc->has_csg_stream = true; }
        }
        UP(csg_root, roots.data(), roots.size());
    }
    UP(csg_node, d.csg_node, 5 * (size_t)d.n_csg_nodes);
    UP(mat_shadow, d.mat_shadow, d.n_materials);
    UP(mat_program, d.mat_program, d.n_materials);
    UP(pat_type, d.pat_type, d.n_pat_nodes);
    UP(pat_child, d.pat_child, 6 * (size_t)d.n_pat_nodes);
    UP(pat_tex, d.pat_tex, d.n_pat_nodes);
    UP(prog_off, d.prog_off, d.n_programs);
    UP(prog_len, d.prog_len, d.n_programs);
    UP(instr, d.instr, 2 * (size_t)d.n_instr);
    UP(tex_off, d.tex_off, d.n_textures);
    UP(tex_w, d.tex_w, d.n_textures);
    UP(tex_h, d.tex_h, d.n_textures);
    UP(tex_data, d.tex_data, (size_t)d.tex_bytes);
    UP(light_type, d.light_type, d.n_lights);
    UP(light_steps, d.light_steps, 3 * (size_t)d.n_lights);
#undef UP
    clk.tick("index arrays up");
    S.n_leaves = d.n_leaves;
    S.n_bvh_nodes = bvh.n_nodes();
    S.n_tnodes = tbvh.n_nodes();
    S.n_unbounded = d.n_unbounded;
    S.n_lights = d.n_lights;
    S.n_flat_prims = (oprim.size() >= 2 && oprim.size() <= 4 && obj_leaf >= 4 && bvh.n_nodes() == 1) ? (int32_t)oprim.size() : 0;
    if (getenv("RTC_NO_FLAT")) S.n_flat_prims = 0;          // development: force the BVH walk
    S.refraction_matters = 0;
    for (int i = 0; i < d.n_materials; i++) if (d.mat[10 * (size_t)i + 9] != 1.0) S.refraction_matters = 1;

    rtc::DScene<float>& F = sc->d32;
#define CP(field) F.field = S.field
    CP(n_leaves); CP(leaf_kind); CP(leaf_xform); CP(leaf_material); CP(leaf_tri); CP(leaf_closed); CP(tri_leaf);
    CP(n_bvh_nodes); CP(bvh_prims); CP(n_flat_prims); CP(n_tnodes); CP(tb_leaf); CP(n_unbounded); CP(unbounded_prims); CP(csg_root); CP(csg_node);
    CP(mat_shadow); CP(mat_program); CP(pat_type); CP(pat_child); CP(pat_tex); CP(prog_off); CP(prog_len); CP(instr);
    CP(tex_off); CP(tex_w); CP(tex_h); CP(tex_data); CP(n_lights); CP(refraction_matters); CP(light_type); CP(light_steps);
#undef CP
    {
        double max_abs = 0;
        for (const rtc::Bvh* b : {&bvh, &tbvh})
            for (double v : b->box) if (std::isfinite(v)) max_abs = std::max(max_abs, std::fabs(v));
        if (std::isfinite(tri_max_abs)) max_abs = std::max(max_abs, tri_max_abs);
        sc->max_abs = max_abs;
        const double pad_abs = RTC_PAD_OF_EXTENT * max_abs;
        auto pack = [pad_abs](const rtc::Bvh& b) {
            std::vector<float> boxes = pad_boxes<float>(b.box, pad_abs);
            std::vector<rtc::BvhNode> out((size_t)b.n_nodes());
            for (int i = 0; i < b.n_nodes(); i++) {
                memcpy(out[i].box, &boxes[12 * (size_t)i], 12 * sizeof(float));
                out[i].child[0] = b.child[2 * (size_t)i]; out[i].child[1] = b.child[2 * (size_t)i + 1];
                out[i].pad[0] = out[i].pad[1] = 0;
            }
            return out;
        };
        std::vector<rtc::BvhNode> on = pack(bvh), tn = pack(tbvh);
        root_box(on.empty() ? nullptr : &on[0], S.obox);
        root_box(tn.empty() ? nullptr : &tn[0], S.tbox);
        memcpy(F.obox, S.obox, sizeof(S.obox)); memcpy(F.tbox, S.tbox, sizeof(S.tbox));
        S.far_origin = F.far_origin = (float)(2.0 * std::max(max_abs, 1e-30));
        if (int rc = A.upload(on.data(), on.size(), &S.nodes)) return rc;
        if (int rc = sc->tbvh_mem.upload(tn.data(), tn.size(), &S.tnodes)) return rc;
        F.nodes = S.nodes; F.tnodes = S.tnodes;
        if (tbvh.n_nodes() > 0) {
            // what rays walk: the same tree four children wide (rtc_bvh.hpp collapse_bvh4); the binary one stays for builds and refits
            rtc::Bvh4 t4 = rtc::collapse_bvh4(tbvh, RTC_BVH_STACK - 1);
            if (t4.stack_need > RTC_BVH_STACK - 1) return fail(RTC_ERR_LIMIT, "four-wide triangle BVH needs %d stack entries", t4.stack_need);
            std::vector<float> b4 = pad_boxes<float>(t4.box, pad_abs);
            std::vector<rtc::BvhNode4> n4((size_t)t4.n_nodes());
            for (int i = 0; i < t4.n_nodes(); i++) {
                memcpy(n4[i].box, &b4[24 * (size_t)i], 24 * sizeof(float));
                for (int k = 0; k < 4; k++) { n4[i].child[k] = t4.child[4 * (size_t)i + k]; n4[i].src[k] = t4.src[4 * (size_t)i + k]; }
            }
            if (getenv("RTC_BVH_INFO")) fprintf(stderr, "[rtc] triangle BVH: %d binary nodes (depth %d) -> %d four-wide nodes, stack need %d\n", tbvh.n_nodes(), tbvh.depth, t4.n_nodes(), t4.stack_need);
            if (int rc = sc->tbvh_mem.upload(n4.data(), n4.size(), &S.tnodes4)) return rc;
            F.tnodes4 = S.tnodes4;
            sc->n_tnodes4 = t4.n_nodes();
        }
        if (int rc = sc->tbvh_mem.upload(tb_geom.data(), tb_geom.size(), &sc->d64.tb_geom)) return rc;
        std::vector<float> g32 = cast_vec<float>(tb_geom.data(), tb_geom.size());
        if (int rc = sc->tbvh_mem.upload(g32.data(), g32.size(), &sc->d32.tb_geom)) return rc;
    }
    clk.tick("nodes and rows up");
    // ---- instanced meshes: one tree and one set of rows per mesh, in its object space ----
    if (d.n_instances > 0) {
        std::vector<rtc::BvhNode4> mnodes;
        std::vector<double> mb_geom(12 * (size_t)d.n_mesh_tris);
        std::vector<int32_t> mb_leaf((size_t)d.n_mesh_tris);
        std::vector<int32_t> mesh_root((size_t)d.n_meshes);
        std::vector<float> mesh_box(6 * (size_t)d.n_meshes), mesh_far((size_t)d.n_meshes);
        const double W = sc->max_abs;                  // rays start inside [-W, W]^3 (farther cameras: the FAR variants re-base)
        for (int m = 0; m < d.n_meshes; m++) {
            const int r0 = d.mesh_range[2 * m], n = d.mesh_range[2 * m + 1];
            std::vector<rtc::Aabb> boxes((size_t)n);
            double max_abs_obj = 0;
            for (int i = 0; i < n; i++) {
                const double* g = d.mesh_geom + 12 * (size_t)(r0 + i);
                for (int k = 0; k < 3; k++) {
                    const double p1 = g[k], p2 = g[k] + g[3 + k], p3 = g[k] + g[6 + k];
                    boxes[i].lo[k] = std::min(p1, std::min(p2, p3)); boxes[i].hi[k] = std::max(p1, std::max(p2, p3));
                    max_abs_obj = std::max(max_abs_obj, std::max(std::fabs(boxes[i].lo[k]), std::fabs(boxes[i].hi[k])));
                }
            }
            // where rays that walk this tree can start, in its object space: the corners of the scene's extent through every instance's transform
            for (int q = 0; q < d.n_instances; q++) {
                if (d.inst[4 * q] != m) continue;
                const double* M = d.xform + 12 * (size_t)d.inst[4 * q + 1];
                for (int c = 0; c < 8; c++) {
                    const double x = (c & 1) ? W : -W, y = (c & 2) ? W : -W, z = (c & 4) ? W : -W;
                    for (int k = 0; k < 3; k++) max_abs_obj = std::max(max_abs_obj, std::fabs(M[4 * k] * x + M[4 * k + 1] * y + M[4 * k + 2] * z + M[4 * k + 3]));
                }
            }
            if (!std::isfinite(max_abs_obj)) return fail(RTC_ERR_INVALID, "mesh %d: coordinates are not finite", m);
            rtc::Bvh mb = rtc::build_bvh(boxes, tri_leaf, RTC_BVH_STACK - 2);
            if (mb.depth > RTC_BVH_STACK - 1) return fail(RTC_ERR_LIMIT, "BVH depth exceeds traversal stack");
            rtc::Bvh4 m4 = rtc::collapse_bvh4(mb, RTC_BVH_STACK - 1);
            if (m4.stack_need > RTC_BVH_STACK - 1) return fail(RTC_ERR_LIMIT, "four-wide mesh BVH needs %d stack entries", m4.stack_need);
            std::vector<float> b4 = pad_boxes<float>(m4.box, RTC_PAD_OF_EXTENT * max_abs_obj);
            const int node_base = (int)mnodes.size();
            mesh_root[m] = node_base;
            float* rb = &mesh_box[6 * (size_t)m];
            for (int a = 0; a < 3; a++) { rb[a] = 3.0e38f; rb[3 + a] = -3.0e38f; }
            for (int i = 0; i < m4.n_nodes(); i++) {
                rtc::BvhNode4 nd;
                memcpy(nd.box, &b4[24 * (size_t)i], 24 * sizeof(float));
                for (int k = 0; k < 4; k++) {
                    int32_t c = m4.child[4 * (size_t)i + k];
                    if (c >= 0) c += node_base;
                    else if (c != RTC_EMPTY_CHILD) { const int code = ~c; c = ~((((code >> 4) + r0) << 4) | (code & 15)); }      // rows of this mesh start at r0
                    nd.child[k] = c; nd.src[k] = -1;
                    if (i == 0 && c != RTC_EMPTY_CHILD)
                        for (int a = 0; a < 3; a++) { rb[a] = std::min(rb[a], nd.box[6 * k + a]); rb[3 + a] = std::max(rb[3 + a], nd.box[6 * k + 3 + a]); }
                }
                mnodes.push_back(nd);
            }
            mesh_far[m] = (float)(2.0 * std::max(max_abs_obj, 1e-30));
            for (int i = 0; i < n; i++) {
                memcpy(&mb_geom[12 * (size_t)(r0 + i)], d.mesh_geom + 12 * (size_t)(r0 + mb.order[i]), 12 * sizeof(double));
                mb_leaf[(size_t)r0 + i] = mb.order[i];
            }
            sc->bvh_nodes += m4.n_nodes(); sc->bvh_depth = std::max(sc->bvh_depth, mb.depth);
        }
        std::vector<int32_t> inst(8 * (size_t)d.n_instances, 0);
        std::vector<float> inst_box(6 * (size_t)d.n_instances), inst_far((size_t)d.n_instances);
        for (int q = 0; q < d.n_instances; q++) {
            const int m = d.inst[4 * q];
            inst[8 * q] = mesh_root[m]; inst[8 * q + 1] = d.inst[4 * q + 1]; inst[8 * q + 2] = d.inst[4 * q + 2];
            inst[8 * q + 3] = d.mesh_range[2 * m]; inst[8 * q + 4] = d.inst[4 * q + 3];
            memcpy(&inst_box[6 * (size_t)q], &mesh_box[6 * (size_t)m], 6 * sizeof(float));
            inst_far[q] = mesh_far[m];
        }
        if (int rc = A.upload(mnodes.data(), mnodes.size(), &S.mnodes4)) return rc;
        if (int rc = A.upload(mb_leaf.data(), mb_leaf.size(), &S.mb_leaf)) return rc;
        if (int rc = A.upload(inst.data(), inst.size(), &S.inst)) return rc;
        if (int rc = A.upload(inst_box.data(), inst_box.size(), &S.inst_box)) return rc;
        if (int rc = A.upload(inst_far.data(), inst_far.size(), &S.inst_far)) return rc;
        if (int rc = sc->mem64.upload(mb_geom.data(), mb_geom.size(), &sc->d64.mb_geom)) return rc;
        std::vector<float> g32 = cast_vec<float>(mb_geom.data(), mb_geom.size());
        if (int rc = sc->mem32.upload(g32.data(), g32.size(), &sc->d32.mb_geom)) return rc;
        S.n_inst = d.n_instances;
        F.n_inst = S.n_inst; F.mnodes4 = S.mnodes4; F.mb_leaf = S.mb_leaf; F.inst = S.inst; F.inst_box = S.inst_box; F.inst_far = S.inst_far;
    }
    clk.tick("instanced meshes");
    if (int rc = upload_real<double>(d, quadric_minmax, sc->mem64, sc->d64)) return rc;
    clk.tick("fp64 arrays up");
    if (int rc = upload_real<float>(d, quadric_minmax, sc->mem32, sc->d32)) return rc;
    clk.tick("fp32 arrays cast and up");
    if ((sc->d64.n_bvh_nodes > 0 && (!sc->d64.nodes || !sc->d64.bvh_prims || !sc->d32.nodes)) ||
        (sc->d64.n_tnodes > 0 && (!sc->d64.tnodes || !sc->d64.tnodes4 || !sc->d32.tnodes4 || !sc->d64.tb_geom || !sc->d64.tb_leaf || !sc->d32.tnodes || !sc->d32.tb_geom)))
        return fail(RTC_ERR_HIP, "internal: BVH arrays missing on the device");
    sc->has_csg = d.n_csg_roots > 0;
    for (int i = 0; i < d.n_instr; i++)
        if (d.instr[2 * (size_t)i] == RTC_OP_PERTURB) sc->has_csg = true;      // the noise code lives in the same kernel variants
    for (int i = 0; i < d.n_materials; i++)
        if (d.mat[10 * (size_t)i + 7] != 0.0 || d.mat[10 * (size_t)i + 8] != 0.0) sc->has_secondary = true;
    for (int i = 0; i < d.n_lights; i++) {
        sc->light_cells += d.light_type[i] == 1 ? (long long)d.light_steps[3 * i] * d.light_steps[3 * i + 1] : 1;
        if (d.light_type[i] == 1 && d.light_steps[3 * i + 2]) sc->has_jitter = true;
    }
    if (device_build) { const int rc = rebuild_triangle_bvh_device(sc, false); clk.tick("tree built on the device"); return rc; }
    return 0;
}

template <class R> rtc::DCamera<R> make_camera(const rtc_camera& c) {
    rtc::DCamera<R> o;
    o.hsize = c.hsize; o.vsize = c.vsize;
    o.half_width = (R)c.half_width; o.half_height = (R)c.half_height; o.pixel_size = (R)c.pixel_size;
    for (int i = 0; i < 12; i++) o.inv[i] = (R)c.inv_transform[i];
    return o;
}

int check_render_args(const rtc_scene* scene, const rtc_camera* cam, const rtc_render_opts* opts, int* tile_size, int* precision,
                      int* depth, double* eps, double* eps_rel) {
    if (!scene || !cam) return fail(RTC_ERR_INVALID, "null scene or camera");
    if (cam->hsize < 1 || cam->vsize < 1 || (long long)cam->hsize * cam->vsize > (1LL << 30))
        return fail(RTC_ERR_INVALID, "bad image size %d x %d", cam->hsize, cam->vsize);
    *precision = opts ? opts->precision : 64;
    if (*precision == 0) *precision = 64;
    if (*precision != 64 && *precision != 32) return fail(RTC_ERR_INVALID, "precision must be 64 or 32, got %d", *precision);
    *depth = opts ? opts->max_depth : 5;
    if (*depth < 0 || *depth > RTC_MAX_BOUNCE_STACK - 2) return fail(RTC_ERR_LIMIT, "max_depth %d outside 0..%d", *depth, RTC_MAX_BOUNCE_STACK - 2);
    *tile_size = (opts && opts->tile_size) ? opts->tile_size : 32;
    if (*tile_size < 8 || *tile_size % 8 || *tile_size > 1024) return fail(RTC_ERR_INVALID, "tile_size must be a multiple of 8 in 8..1024");
    *eps = (opts && opts->shade_eps > 0) ? opts->shade_eps : (*precision == 64 ? 0.000001 : 0.00001);
    *eps_rel = (opts && opts->shade_eps_rel >= 0) ? opts->shade_eps_rel : (*precision == 64 ? 0.0 : 0.000001);
    return 0;
}

// The triangle BVH rebuilt on the GPU from the resident triangle rows (rtc_lbvh.hpp).  Builds into
// fresh buffers and swaps them in only if the tree fits the traversal stack.
int rebuild_triangle_bvh_device(rtc_scene* sc, bool again) {
    const int n = sc->n_titems;
    if (n == 0) return 0;
    const int n_leaves = (n + RTC_GPU_LEAF - 1) / RTC_GPU_LEAF;
    DeviceArena fresh, temp;
    auto alloc = [](DeviceArena& A, size_t bytes, void** out) -> int {
        HIP_TRY(hipMalloc(out, bytes ? bytes : 1));
        A.ptrs.push_back(*out); A.bytes += bytes;
        return 0;
    };
    void *p_nodes4, *p_height, *p_nodes, *p_g64, *p_g32, *p_leaf, *p_row, *p_arr2, *p_boxes, *p_bounds, *p_keys, *p_sorted, *p_idx, *p_sidx, *p_lkeys, *p_pin, *p_pleaf, *p_arr, *p_deep, *p_tmp = nullptr;
    const size_t n_nodes = (size_t)std::max(1, n_leaves - 1);
    int rc = 0;
    auto bail = [&](int code) { fresh.release(); temp.release(); return code; };
    if ((rc = alloc(fresh, n_nodes * sizeof(rtc::BvhNode), &p_nodes)) || (rc = alloc(fresh, 12 * (size_t)n * sizeof(double), &p_g64)) ||
        (rc = alloc(fresh, 12 * (size_t)n * sizeof(float), &p_g32)) || (rc = alloc(fresh, (size_t)n * sizeof(int32_t), &p_leaf)) ||
        (rc = alloc(fresh, (size_t)n * sizeof(int32_t), &p_row)) || (rc = alloc(fresh, n_nodes * sizeof(uint32_t), &p_arr2)) ||
        (rc = alloc(temp, 6 * (size_t)n * sizeof(float), &p_boxes)) || (rc = alloc(temp, 6 * sizeof(uint32_t), &p_bounds)) ||
        (rc = alloc(temp, (size_t)n * sizeof(uint64_t), &p_keys)) || (rc = alloc(temp, (size_t)n * sizeof(uint64_t), &p_sorted)) ||
        (rc = alloc(temp, (size_t)n * sizeof(uint32_t), &p_idx)) || (rc = alloc(temp, (size_t)n * sizeof(uint32_t), &p_sidx)) ||
        (rc = alloc(temp, (size_t)n_leaves * sizeof(uint64_t), &p_lkeys)) || (rc = alloc(fresh, n_nodes * sizeof(int32_t), &p_pin)) ||
        (rc = alloc(temp, (size_t)n_leaves * sizeof(int32_t), &p_pleaf)) || (rc = alloc(temp, n_nodes * sizeof(uint32_t), &p_arr)) ||
        (rc = alloc(temp, sizeof(int32_t), &p_deep)) || (rc = alloc(fresh, n_nodes * sizeof(rtc::BvhNode4), &p_nodes4)) ||
        (rc = alloc(temp, n_nodes * sizeof(int32_t), &p_height)))
        return bail(rc);
#define TRY_B(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return bail(fail(RTC_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_))); } while (0)
    const uint32_t bounds_init[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
    TRY_B(hipMemcpy(p_bounds, bounds_init, sizeof(bounds_init), hipMemcpyHostToDevice));
    const dim3 blk(256);
    auto grid_for = [](long long items) { return dim3((unsigned)((items + 255) / 256)); };
    hipLaunchKernelGGL(rtc::lbvh_boxes_kernel, grid_for(n), blk, 0, nullptr, n, sc->d_titems, sc->d64.tri_geom, (float*)p_boxes, (uint32_t*)p_bounds,
                       RTC_PAD_OF_EXTENT * sc->max_abs);
    int32_t deepest = 0;
    // The topology: PLOC over the Morton order of the finest grid (21 bits per axis) first -- merges chosen by surface area,
    // rtc_lbvh_device.hpp -- then, should that tree be too deep for the traversal stack or not finish, the Karras radix tree on
    // the same grid, then on coarser ones.  (Development: RTC_GPU_BUILDER=lbvh skips PLOC, RTC_PLOC_RADIUS sets its search radius.)
    struct Attempt { bool ploc; int drop_bits; };
    const Attempt attempts[4] = {{true, 0}, {false, 0}, {false, 18}, {false, 33}};
    static const bool use_ploc = [] { const char* e = getenv("RTC_GPU_BUILDER"); return !e || strcmp(e, "lbvh") != 0; }();
    static const int ploc_radius = [] { const char* e = getenv("RTC_PLOC_RADIUS"); int v = e ? atoi(e) : 16; return v < 1 ? 1 : v > 256 ? 256 : v; }();
    void *p_cl0 = nullptr, *p_cl1 = nullptr, *p_cbox = nullptr, *p_near = nullptr, *p_flags = nullptr, *p_offs = nullptr, *p_st = nullptr, *p_scan = nullptr;
    size_t scan_bytes = 0;
    int last_drop = -1;
    for (int attempt = 0; attempt < 4; attempt++) {
        const Attempt at = attempts[attempt];
        if (at.ploc && (!use_ploc || n_leaves < 2)) continue;
        TRY_B(hipMemset(p_pin, 0xFF, n_nodes * sizeof(int32_t)));
        TRY_B(hipMemset(p_pleaf, 0xFF, (size_t)n_leaves * sizeof(int32_t)));
        TRY_B(hipMemset(p_arr, 0, n_nodes * sizeof(uint32_t)));
        TRY_B(hipMemset(p_deep, 0, sizeof(int32_t)));
        if (at.drop_bits != last_drop) {
            hipLaunchKernelGGL(rtc::lbvh_keys_kernel, grid_for(n), blk, 0, nullptr, n, (const float*)p_boxes, (const uint32_t*)p_bounds, at.drop_bits,
                               (uint64_t*)p_keys, (uint32_t*)p_idx);
            size_t tmp_bytes = 0;
            TRY_B(rocprim::radix_sort_pairs(nullptr, tmp_bytes, (uint64_t*)p_keys, (uint64_t*)p_sorted, (uint32_t*)p_idx, (uint32_t*)p_sidx, (size_t)n, 0, 64,
                                            (hipStream_t) nullptr));
            if (!p_tmp && (rc = alloc(temp, tmp_bytes, &p_tmp))) return bail(rc);
            TRY_B(rocprim::radix_sort_pairs(p_tmp, tmp_bytes, (uint64_t*)p_keys, (uint64_t*)p_sorted, (uint32_t*)p_idx, (uint32_t*)p_sidx, (size_t)n, 0, 64,
                                            (hipStream_t) nullptr));
            last_drop = at.drop_bits;
        }
        if (n_leaves == 1) {
            hipLaunchKernelGGL(rtc::lbvh_single_kernel, dim3(1), dim3(64), 0, nullptr, n, (const uint32_t*)p_sidx, (const float*)p_boxes, (rtc::BvhNode*)p_nodes);
        } else if (at.ploc) {
            if (!p_cl0) {
                TRY_B(rocprim::exclusive_scan(nullptr, scan_bytes, (unsigned long long*)nullptr, (unsigned long long*)nullptr, 0ull, (size_t)n_leaves,
                                              rocprim::plus<unsigned long long>(), (hipStream_t) nullptr));
                if ((rc = alloc(temp, (size_t)n_leaves * sizeof(int32_t), &p_cl0)) || (rc = alloc(temp, (size_t)n_leaves * sizeof(int32_t), &p_cl1)) ||
                    (rc = alloc(temp, 6 * (2 * (size_t)n_leaves) * sizeof(float), &p_cbox)) || (rc = alloc(temp, (size_t)n_leaves * sizeof(int32_t), &p_near)) ||
                    (rc = alloc(temp, (size_t)n_leaves * sizeof(unsigned long long), &p_flags)) || (rc = alloc(temp, (size_t)n_leaves * sizeof(unsigned long long), &p_offs)) ||
                    (rc = alloc(temp, sizeof(rtc::PlocState), &p_st)) || (rc = alloc(temp, scan_bytes ? scan_bytes : 1, &p_scan)))
                    return bail(rc);
            }
            hipLaunchKernelGGL(rtc::ploc_init_kernel, grid_for(n_leaves), blk, 0, nullptr, n_leaves, n, (const uint32_t*)p_sidx, (const float*)p_boxes, (int32_t*)p_cl0,
                               (float*)p_cbox, (rtc::PlocState*)p_st);
            rtc::PlocState st = {n_leaves, 0, 0, 0};
            void* cl[2] = {p_cl0, p_cl1};
            int cur = 0;
            // batches of iterations between looks at the count (every look is a wait); a mesh takes ~ 2 log2(leaves) iterations
            for (int batch = 0; batch < 64 && st.n > 1; batch++) {
                for (int it = 0; it < 16; it++) {
                    hipLaunchKernelGGL(rtc::ploc_nearest_kernel, grid_for(n_leaves), blk, 0, nullptr, (const rtc::PlocState*)p_st, ploc_radius, (const int32_t*)cl[cur],
                                       (const float*)p_cbox, (int32_t*)p_near);
                    hipLaunchKernelGGL(rtc::ploc_flags_kernel, grid_for(n_leaves), blk, 0, nullptr, (const rtc::PlocState*)p_st, n_leaves, (const int32_t*)p_near,
                                       (unsigned long long*)p_flags);
                    TRY_B(rocprim::exclusive_scan(p_scan, scan_bytes, (unsigned long long*)p_flags, (unsigned long long*)p_offs, 0ull, (size_t)n_leaves,
                                                  rocprim::plus<unsigned long long>(), (hipStream_t) nullptr));
                    hipLaunchKernelGGL(rtc::ploc_merge_kernel, grid_for(n_leaves), blk, 0, nullptr, (rtc::PlocState*)p_st, n_leaves, n, (const int32_t*)p_near,
                                       (const unsigned long long*)p_flags, (const unsigned long long*)p_offs, (const int32_t*)cl[cur], (int32_t*)cl[1 - cur],
                                       (float*)p_cbox, (rtc::BvhNode*)p_nodes, (int32_t*)p_pin, (int32_t*)p_pleaf);
                    hipLaunchKernelGGL(rtc::ploc_advance_kernel, dim3(1), dim3(1), 0, nullptr, (rtc::PlocState*)p_st, n_leaves, (const unsigned long long*)p_flags,
                                       (const unsigned long long*)p_offs);
                    cur = 1 - cur;
                }
                TRY_B(hipMemcpy(&st, p_st, sizeof(st), hipMemcpyDeviceToHost));
            }
            if (getenv("RTC_BVH_INFO")) fprintf(stderr, "[rtc] PLOC: %d leaves, radius %d, %d iterations, %d merges, %d clusters left\n", n_leaves, ploc_radius, st.iterations, st.merges, st.n);
            if (st.n != 1 || st.merges != n_leaves - 1) continue;            // did not finish: the radix tree instead
            hipLaunchKernelGGL(rtc::lbvh_depth_kernel, grid_for(n_leaves), blk, 0, nullptr, n_leaves, (const int32_t*)p_pin, (const int32_t*)p_pleaf, (int32_t*)p_deep);
        } else {
            hipLaunchKernelGGL(rtc::lbvh_leaf_keys_kernel, grid_for(n_leaves), blk, 0, nullptr, n_leaves, (const uint64_t*)p_sorted, (uint64_t*)p_lkeys);
            hipLaunchKernelGGL(rtc::lbvh_inner_kernel, grid_for(n_leaves - 1), blk, 0, nullptr, n_leaves, n, (const uint64_t*)p_lkeys, (rtc::BvhNode*)p_nodes,
                               (int32_t*)p_pin, (int32_t*)p_pleaf);
            hipLaunchKernelGGL(rtc::lbvh_refit_kernel, grid_for(n_leaves), blk, 0, nullptr, n_leaves, n, (const uint32_t*)p_sidx, (const float*)p_boxes,
                               (rtc::BvhNode*)p_nodes, (const int32_t*)p_pin, (const int32_t*)p_pleaf, (uint32_t*)p_arr);
            hipLaunchKernelGGL(rtc::lbvh_depth_kernel, grid_for(n_leaves), blk, 0, nullptr, n_leaves, (const int32_t*)p_pin, (const int32_t*)p_pleaf, (int32_t*)p_deep);
        }
        TRY_B(hipGetLastError());
        TRY_B(hipMemcpy(&deepest, p_deep, sizeof(deepest), hipMemcpyDeviceToHost));      // also waits for the build
        if (getenv("RTC_BVH_INFO")) fprintf(stderr, "[rtc] GPU build attempt %d (%s, grid - %d bits): depth %d\n", attempt, at.ploc ? "PLOC" : "radix tree", at.drop_bits, deepest);
        if (n_leaves == 1 || deepest <= RTC_BVH_STACK - 1) break;
    }
    // the four-wide form rays walk, collapsed from the binary tree on the device (heights first: what fits the traversal stack)
    TRY_B(hipMemset(p_height, 0, n_nodes * sizeof(int32_t)));
    if (n_leaves > 1)
        hipLaunchKernelGGL(rtc::lbvh_height_kernel, grid_for(n_leaves), blk, 0, nullptr, n_leaves, (const int32_t*)p_pin, (const int32_t*)p_pleaf, (int32_t*)p_height);
    hipLaunchKernelGGL(rtc::lbvh_collapse4_kernel, grid_for((long long)n_nodes), blk, 0, nullptr, (int)n_nodes, (const rtc::BvhNode*)p_nodes, (const int32_t*)p_pin,
                       (const int32_t*)p_height, RTC_BVH_STACK - 1, (rtc::BvhNode4*)p_nodes4);
    hipLaunchKernelGGL(rtc::lbvh_reorder_kernel, grid_for(12LL * n), blk, 0, nullptr, n, (const uint32_t*)p_sidx, sc->d_titems, sc->d64.tri_geom,
                       sc->d32.tri_geom, sc->d64.tri_leaf, (double*)p_g64, (float*)p_g32, (int32_t*)p_leaf, (int32_t*)p_row);
    TRY_B(hipGetLastError());
    TRY_B(hipDeviceSynchronize());
#undef TRY_B
    if (n_leaves == 1) deepest = 1;
    if (deepest > RTC_BVH_STACK - 1) { bail(0); return fail(RTC_ERR_LIMIT, "GPU-built triangle BVH is %d levels deep, the traversal stack holds %d", deepest, RTC_BVH_STACK - 1); }
    temp.release();
    sc->tbvh_mem.release();
    sc->tbvh_mem = fresh;
    fresh.ptrs.clear();
    sc->d64.tnodes = sc->d32.tnodes = (const rtc::BvhNode*)p_nodes;
    sc->d64.n_tnodes = sc->d32.n_tnodes = (int32_t)n_nodes;
    sc->d64.tnodes4 = sc->d32.tnodes4 = (const rtc::BvhNode4*)p_nodes4;
    sc->n_tnodes4 = (int)n_nodes;
    sc->d64.tb_geom = (const double*)p_g64; sc->d32.tb_geom = (const float*)p_g32;
    sc->d64.tb_leaf = sc->d32.tb_leaf = (const int32_t*)p_leaf;
    sc->d_tb_row = (int32_t*)p_row; sc->d_tparent = (int32_t*)p_pin; sc->d_tarrived = (uint32_t*)p_arr2;
    sc->bvh_nodes = sc->d64.n_bvh_nodes + (int)n_nodes;
    sc->bvh_depth = std::max(sc->obj_depth, (int)deepest);
    sc->bvh_max_leaf = std::max(sc->obj_max_leaf, std::min(n, RTC_GPU_LEAF));
    bool grew = false;
    if (int rc2 = refresh_triangle_root_box(sc, &grew)) return rc2;
    if (grew && !again) return rebuild_triangle_bvh_device(sc, true);
    for (Workspace* w : sc->workspaces) w->fb_age = std::min(w->fb_age, 1);   // the launch order stays (any order renders the same image); costs are recorded anew
    return 0;
}

// The triangle BVH refitted on the GPU: same nodes, same leaves, new boxes around the rows now in the triangle table.
int refit_triangle_bvh_device(rtc_scene* sc, bool again = false) {
    const int n = sc->n_titems, n_nodes = sc->d64.n_tnodes;
    if (n == 0 || n_nodes == 0) return 0;
    if (!sc->d_tb_row || !sc->d_tparent || !sc->d_tarrived) return fail(RTC_ERR_HIP, "internal: refit tables missing");
    const dim3 blk(256);
    HIP_TRY(hipMemset(sc->d_tarrived, 0, (size_t)n_nodes * sizeof(uint32_t)));
    hipLaunchKernelGGL(rtc::bvh_gather_rows_kernel, dim3((unsigned)((12LL * n + 255) / 256)), blk, 0, nullptr, n, (const int32_t*)sc->d_tb_row,
                       sc->d64.tri_geom, sc->d32.tri_geom, const_cast<double*>(sc->d64.tb_geom), const_cast<float*>(sc->d32.tb_geom));
    hipLaunchKernelGGL(rtc::bvh_refit_kernel, dim3((unsigned)((n_nodes + 255) / 256)), blk, 0, nullptr, n_nodes, const_cast<rtc::BvhNode*>(sc->d64.tnodes),
                       (const int32_t*)sc->d_tparent, sc->d64.tb_geom, sc->d_tarrived, RTC_PAD_OF_EXTENT * sc->max_abs);
    hipLaunchKernelGGL(rtc::bvh4_refresh_kernel, dim3((unsigned)((sc->n_tnodes4 + 255) / 256)), blk, 0, nullptr, sc->n_tnodes4,
                       const_cast<rtc::BvhNode4*>(sc->d64.tnodes4), sc->d64.tnodes);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    bool grew = false;
    if (int rc2 = refresh_triangle_root_box(sc, &grew)) return rc2;
    if (grew && !again) return refit_triangle_bvh_device(sc, true);
    for (Workspace* w : sc->workspaces) w->fb_age = std::min(w->fb_age, 1);
    return 0;
}

// ---- workspaces ----
Workspace* acquire_workspace(rtc_scene* sc, hipStream_t stream, const FbKey& key) {
    std::lock_guard<std::mutex> lock(sc->ws_mutex);
    Workspace* pick = nullptr;
    for (Workspace* w : sc->workspaces) {
        if (w->claimed) continue;
        bool free_now = !w->used || w->stream == stream;
        if (!free_now) { free_now = hipEventQuery(w->done) == hipSuccess; (void)hipGetLastError(); }
        if (!free_now) continue;
        if (memcmp(&w->fb_key, &key, sizeof(key)) == 0) { pick = w; break; }      // the one that knows this view
        if (!pick) pick = w;
    }
    if (!pick) {
        pick = new (std::nothrow) Workspace();
        if (!pick) return nullptr;
        if (hipEventCreateWithFlags(&pick->done, hipEventDisableTiming) != hipSuccess) { delete pick; return nullptr; }
        sc->workspaces.push_back(pick);
    }
    pick->claimed = true;
    return pick;
}
struct WorkspaceLease {
    rtc_scene* sc; Workspace* ws; hipStream_t stream;
    ~WorkspaceLease() {
        if (!ws) return;
        (void)hipEventRecord(ws->done, stream);
        std::lock_guard<std::mutex> lock(sc->ws_mutex);
        ws->stream = stream; ws->used = true; ws->claimed = false;
    }
};

// development (RTC_DEBUG_SYNC=1): wait for the stream after every kernel of a frame and say which one failed, with the
// launch it was part of -- how round 3 looked for the CSG variants' fault of round 2
const bool g_debug_sync = [] { const char* e = getenv("RTC_DEBUG_SYNC"); return e && atoi(e) != 0; }();
int debug_sync(hipStream_t stream, const char* what, unsigned grid_x, const rtc::DFrame& F) {
    if (!g_debug_sync) return 0;
    hipError_t e = hipStreamSynchronize(stream);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) return 0;
    fprintf(stderr, "[rtc debug] %s failed: %s (grid %u, n_units %d, n_entries %d, n_slots %d, order %p, cost %p, work %p, replicas %d, tiles %d)\n", what,
            hipGetErrorString(e), grid_x, F.n_units, F.n_entries, F.n_slots, (const void*)F.order, (void*)F.cost, (void*)F.work, F.replicas, F.n_tiles);
    return fail(RTC_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

// the trace kernel variant for (precision, counting, CSG / perturbed patterns) on `grid` workgroups
void launch_trace(rtc_scene* scene, const rtc_camera* cam, const rtc::DFrame& F, dim3 grid, int precision, bool prof, double eps, double eps_rel,
                  hipStream_t stream, bool two_waves = false, int feedback_stage = 0) {
    const bool csg = scene->has_csg;
    const int mode = csg ? (scene->has_csg_stream ? 2 : 1) : (scene->d64.n_inst > 0 ? 3 : 0);       // trace_kernel's CSG parameter
    // `grid` counts RTC_BLOCK-thread blocks (4 entries of the launch order each); the variants without CSG run one entry per workgroup
    const dim3 block(csg ? RTC_BLOCK_CSG : RTC_BLOCK_PLAIN);
    grid = dim3(grid.x * (RTC_BLOCK / block.x));
    // a camera farther out than the boxes' padding covers (DScene::far_origin) gets the variants that re-base rays
    const double far_origin = scene->d64.far_origin;
    const bool far_cam = std::fabs(cam->inv_transform[3]) > far_origin || std::fabs(cam->inv_transform[7]) > far_origin || std::fabs(cam->inv_transform[11]) > far_origin;
    const bool repl = F.replicas != 0 && !prof;
    {
        rtc_launch_info& L = scene->last_launch;
        L.precision = precision; L.csg = mode; L.far_camera = far_cam; L.replicas = repl; L.counting_work = prof;
        L.waves_per_simd = csg ? 2 : precision == 64 ? ((two_waves && !prof) ? 2 : RTC_WAVES_F64) : RTC_WAVES_F32;
        L.workgroup_threads = (int32_t)block.x; L.workgroups = (int32_t)grid.x; L.n_units = F.n_units; L.n_entries = F.order ? F.n_entries : F.n_units;
        L.feedback_stage = feedback_stage;
    }
#define RTC_LAUNCH_W(R, PROF, CSG, FAR, REPL, W, SC) \
    hipLaunchKernelGGL((rtc::trace_kernel<R, PROF, CSG, FAR, REPL, W>), grid, block, 0, stream, SC, make_camera<R>(*cam), F, (R)eps, (R)eps_rel)
#define RTC_LAUNCH_K(R, PROF, CSG, FAR, REPL, SC) RTC_LAUNCH_W(R, PROF, CSG, FAR, REPL, RTC_WAVES_OF(R, CSG), SC)
#define RTC_LAUNCH_REPL(R, CSG, FAR, SC) do { if (prof) RTC_LAUNCH_K(R, true, CSG, FAR, false, SC); \
    else if (repl) RTC_LAUNCH_K(R, false, CSG, FAR, true, SC); else RTC_LAUNCH_K(R, false, CSG, FAR, false, SC); } while (0)
#ifdef RTC_DEV_PLAIN_ONLY   /* development builds (tools/build_variant.sh): only the variants the mesh scenes use -- seconds instead of minutes */
#define RTC_LAUNCH_FAR(R, CSG, SC) do { (void)far_cam; RTC_LAUNCH_REPL(R, CSG, false, SC); } while (0)
#define RTC_LAUNCH_CSG12(R, SC) do { } while (0)
#else
#define RTC_LAUNCH_FAR(R, CSG, SC) do { if (far_cam) RTC_LAUNCH_REPL(R, CSG, true, SC); else RTC_LAUNCH_REPL(R, CSG, false, SC); } while (0)
#define RTC_LAUNCH_CSG12(R, SC) do { if (!scene->has_csg_stream) RTC_LAUNCH_FAR(R, 1, SC); else RTC_LAUNCH_FAR(R, 2, SC); } while (0)
#endif
    if (precision == 64) {
        // the variants of rtc_trace_f64.hip (built without machine LICM, which parks hoisted fp64 constants in scratch)
        rtc::launch_trace_f64(mode, prof, far_cam, repl, two_waves, grid, stream, scene->d64, make_camera<double>(*cam), F, eps, eps_rel);
    } else {
        if (mode == 3) RTC_LAUNCH_FAR(float, 3, scene->d32); else if (!csg) RTC_LAUNCH_FAR(float, 0, scene->d32); else RTC_LAUNCH_CSG12(float, scene->d32);
    }
#undef RTC_LAUNCH_CSG12
#undef RTC_LAUNCH_FAR
#undef RTC_LAUNCH_REPL
#undef RTC_LAUNCH_K
#undef RTC_LAUNCH_W
}

int launch(rtc_scene* scene, const rtc_camera* cam, const rtc_render_opts* opts, int32_t n_tiles, const int32_t* tile_ids_dev,
           double* rgb_dev, int32_t* hit_id_dev, int64_t* rays_dev, int64_t* work_dev, hipStream_t stream) {
    int T, precision, depth;
    double eps, eps_rel;
    if (int rc = check_render_args(scene, cam, opts, &T, &precision, &depth, &eps, &eps_rel)) return rc;
    if (!rgb_dev) return fail(RTC_ERR_INVALID, "null output buffer");
    int tiles_x = (cam->hsize + T - 1) / T, tiles_y = (cam->vsize + T - 1) / T;
    if (!tile_ids_dev) n_tiles = tiles_x * tiles_y;
    if (n_tiles < 0) return fail(RTC_ERR_INVALID, "negative tile count");
    if (n_tiles == 0) return 0;
    HIP_TRY(hipSetDevice(scene->device));
    rtc::DFrame F;
    F.n_tiles = n_tiles; F.tile_size = T; F.tiles_x = tiles_x; F.tile_ids = tile_ids_dev;
    F.rgb = rgb_dev; F.hit_id = hit_id_dev; F.rays = reinterpret_cast<unsigned long long*>(rays_dev);
    F.max_depth = depth; F.seed = opts ? opts->seed : 0;
    F.replicas = 0; F.n_slots = 0;
    long long work = (long long)n_tiles * T * T;
    long long blocks = (work + RTC_BLOCK - 1) / RTC_BLOCK;
    if (blocks > 0x7FFFFFFFLL) return fail(RTC_ERR_LIMIT, "too many blocks");
    F.work = reinterpret_cast<unsigned long long*>(work_dev);
    F.order = nullptr; F.cost = nullptr;
    const long long n_units = blocks * (RTC_BLOCK / 64);
    if (n_units > 0x7FFFFFFFLL) return fail(RTC_ERR_LIMIT, "too many blocks");
    F.n_units = (int32_t)n_units;
    const bool csg = scene->has_csg;
    const bool feedback = !(opts && (opts->flags & RTC_FLAG_NO_FEEDBACK));
    const int waves_per_simd = csg ? 2 : (precision == 64 ? RTC_WAVES_F64 : RTC_WAVES_F32);
    dim3 grid((unsigned)blocks);
    // spare launch-order entries for the quadrants of split units (3 more per split unit)
    const long long n_entries = ((n_units + 3 * (n_units / 4 + 64)) + RTC_BLOCK / 64 - 1) / (RTC_BLOCK / 64) * (RTC_BLOCK / 64);
    F.n_entries = (int32_t)n_units;
    bool resort = false;
    Workspace* ws = nullptr;
    WorkspaceLease lease = {scene, nullptr, stream};
    // per-wave slots for the ray counts (DFrame::work) in the variants without CSG, which count no work: a launch without
    // feedback takes a workspace for them too (one-wave workgroups would otherwise mean five atomics per wave)
    const bool want_parts = rays_dev && g_ray_parts && (!csg || RTC_CSG_SLOTS) && !work_dev;
    if (feedback || want_parts) {
        FbKey key;
        memset(&key, 0, sizeof(key));
        key.cam = *cam; key.precision = precision; key.depth = depth; key.tile_size = T; key.n_tiles = n_tiles; key.tile_ids = tile_ids_dev;
        ws = acquire_workspace(scene, stream, key);
        if (!ws) return fail(RTC_ERR_NOMEM, "no workspace for this call");
        lease.ws = ws;
        if (ws->fb_units != n_units) {
            if (ws->used) HIP_TRY(hipStreamSynchronize(stream));        // frames that still read the old buffers
            ws->fb_mem.release(); ws->fb_mem.bytes = 0; ws->fb_units = 0; ws->fb_valid = false;
            // the buffers of the old view are gone: whatever view comes next on this workspace starts over (costs, order, splits, tuning)
            memset(&ws->fb_key, 0, sizeof(ws->fb_key)); ws->fb_age = 0; ws->tune_stage = 0; ws->tune_two_waves = false;
            void* p = nullptr;
            HIP_TRY(hipMalloc(&p, (size_t)n_units * sizeof(uint32_t))); ws->fb_mem.ptrs.push_back(p); ws->fb_cost = (uint32_t*)p;
            for (int k = 0; k < 2; k++) {
                HIP_TRY(hipMalloc(&p, (size_t)n_entries * sizeof(int32_t))); ws->fb_mem.ptrs.push_back(p); ws->fb_order[k] = (int32_t*)p;
            }
            HIP_TRY(hipMalloc(&p, sizeof(int32_t))); ws->fb_mem.ptrs.push_back(p); ws->fb_used_dev = (int32_t*)p;
            HIP_TRY(hipMalloc(&p, (size_t)n_entries * 8 * sizeof(uint32_t))); ws->fb_mem.ptrs.push_back(p); ws->ray_parts = (uint32_t*)p;
            if (!ws->fb_used_host) HIP_TRY(hipHostMalloc((void**)&ws->fb_used_host, sizeof(int32_t), hipHostMallocDefault));
            if (!ws->fb_used_event) HIP_TRY(hipEventCreateWithFlags(&ws->fb_used_event, hipEventDisableTiming));
            ws->fb_used_known = false;
            ws->fb_mem.bytes = (size_t)n_units * 4 + (size_t)n_entries * 40;
            ws->fb_units = n_units;
        }
    }
    if (feedback) {
        FbKey key;
        memset(&key, 0, sizeof(key));
        key.cam = *cam; key.precision = precision; key.depth = depth; key.tile_size = T; key.n_tiles = n_tiles; key.tile_ids = tile_ids_dev;
        if (memcmp(&key, &ws->fb_key, sizeof(key)) != 0) { ws->fb_key = key; ws->fb_age = 0; ws->tune_stage = 0; ws->tune_two_waves = false; }
        resort = ws->fb_age < 2;
        ws->fb_age++;
        if (resort) {
            HIP_TRY(hipMemsetAsync(ws->fb_cost, 0, (size_t)n_units * sizeof(uint32_t), stream));
            F.cost = ws->fb_cost;
        }
        if (ws->fb_valid) {
            F.order = ws->fb_order[ws->fb_cur];
            long long used = n_entries;
            if (!ws->fb_used_known && ws->fb_used_event && hipEventQuery(ws->fb_used_event) == hipSuccess) ws->fb_used_known = true;
            if (ws->fb_used_known && *ws->fb_used_host >= n_units && *ws->fb_used_host <= n_entries) used = *ws->fb_used_host;
            // more entries than units = some units are split: their parts' pixels get several lanes each (the REPL kernel variants)
            if (ws->fb_used_known && used > n_units && !work_dev && scene->light_cells >= 1 && scene->light_cells <= 128 && g_replicas) F.replicas = 1;
            (void)hipGetLastError();       // a not-ready query is not an error
            used = (used + RTC_BLOCK / 64 - 1) / (RTC_BLOCK / 64) * (RTC_BLOCK / 64);
            F.n_entries = (int32_t)used;
            grid = dim3((unsigned)(used / (RTC_BLOCK / 64)));
        }
    }
    // which of the two fp64 register budgets for this view (see Workspace::tune_ev)
    bool two_waves = false;
    int timing = -1;
    if (ws && feedback && g_tune_waves && precision == 64 && !csg && !work_dev) {
        if (ws->tune_stage == 0 && !resort && ws->fb_valid && ws->fb_used_known) ws->tune_stage = 1;
        if (ws->tune_stage >= 1 && ws->tune_stage <= 4) {
            if (!ws->tune_ev[0]) {
                bool made = true;
                for (hipEvent_t& e : ws->tune_ev) made = made && hipEventCreate(&e) == hipSuccess;
                if (!made) {         // all eight or none: this view keeps the default budget
                    for (hipEvent_t& e : ws->tune_ev) { if (e) (void)hipEventDestroy(e); e = nullptr; }
                    (void)hipGetLastError();
                    ws->tune_stage = 6; ws->tune_two_waves = false;
                }
            }
            if (ws->tune_stage <= 4) {
                timing = 2 * (ws->tune_stage - 1);
                two_waves = (ws->tune_stage & 1) == 0;
                ws->tune_stage++;
            }
        } else if (ws->tune_stage == 5 && hipEventQuery(ws->tune_ev[7]) == hipSuccess) {
            float ms[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            bool ok = true;
            for (int k = 0; k < 4; k++) ok = ok && hipEventElapsedTime(&ms[k], ws->tune_ev[2 * k], ws->tune_ev[2 * k + 1]) == hipSuccess;
            ws->tune_two_waves = ok && std::min(ms[1], ms[3]) < 0.95f * std::min(ms[0], ms[2]);       // the wider budget has to win by 5 %
            ws->tune_stage = 6;
        }
        (void)hipGetLastError();       // a not-ready query is not an error
        if (ws->tune_stage == 6) two_waves = ws->tune_two_waves;
    }
    // ray counts through per-wave slots instead of atomics (DFrame::work; every wave of the grid writes its slot, so nothing
    // to clear) -- in the variants without CSG, which count no work
    uint32_t* ray_parts = (ws && want_parts) ? ws->ray_parts : nullptr;
    if (ray_parts) { F.work = reinterpret_cast<unsigned long long*>(ray_parts); F.n_slots = (int32_t)n_entries; }
    if (ray_parts && (long long)grid.x * (RTC_BLOCK / 64) > n_entries)      // every wave of the grid has a slot (the kernel's own guard never fires)
        return fail(RTC_ERR_HIP, "internal: %lld waves for %lld ray count slots", (long long)grid.x * (RTC_BLOCK / 64), n_entries);
    if (timing >= 0) HIP_TRY(hipEventRecord(ws->tune_ev[timing], stream));
    launch_trace(scene, cam, F, grid, precision, work_dev != nullptr, eps, eps_rel, stream, two_waves, (ws && feedback) ? ws->fb_age : 0);
    if (int rc = debug_sync(stream, "trace_kernel", grid.x, F)) return rc;
    if (ray_parts) {
        const int n_waves = (int)std::min<long long>((long long)grid.x * (RTC_BLOCK / 64), n_entries);
        hipLaunchKernelGGL(sum_rays_kernel, dim3((unsigned)std::min(64, (n_waves + 255) / 256)), dim3(256), 0, stream,
                           reinterpret_cast<const uint4*>(ray_parts), n_waves, F.rays);
        if (int rc = debug_sync(stream, "sum_rays_kernel", grid.x, F)) return rc;
    }
    if (timing >= 0) HIP_TRY(hipEventRecord(ws->tune_ev[timing + 1], stream));
    HIP_TRY(hipGetLastError());
    if (feedback && resort) {
        // this frame's costs -> the order the next frame of this view launches in (other buffer: the
        // kernel above may still be reading the current one)
        hipLaunchKernelGGL(order_kernel, dim3(1), dim3(1024), 0, stream, ws->fb_cost, ws->fb_order[1 - ws->fb_cur], (int)n_units,
                           (int)n_entries, scene->n_cus * 4 * waves_per_simd, ws->fb_age >= 2 ? g_split4 : 0.0f, g_split16, ws->fb_used_dev, g_split_t);   // the first measured frame ran in plain order: its costs only sort; the second ran sorted and unsplit: its costs decide the splits
        if (int rc = debug_sync(stream, "order_kernel", grid.x, F)) return rc;
        ws->fb_used_known = false;
        HIP_TRY(hipMemcpyAsync(ws->fb_used_host, ws->fb_used_dev, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipEventRecord(ws->fb_used_event, stream));
        HIP_TRY(hipGetLastError());
        ws->fb_cur = 1 - ws->fb_cur;
        ws->fb_valid = true;
    }
    return 0;
}

// host-buffer path shared by rtc_render and rtc_render_tiles
int render_host(rtc_scene* scene, const rtc_camera* cam, const rtc_render_opts* opts, int32_t n_tiles, const int32_t* tile_ids,
                bool tiled, double* rgb_out, int32_t* hit_id_out, rtc_stats* stats_out) {
    int T, precision, depth;
    double eps, eps_rel;
    if (int rc = check_render_args(scene, cam, opts, &T, &precision, &depth, &eps, &eps_rel)) return rc;
    if (!rgb_out) return fail(RTC_ERR_INVALID, "null rgb_out");
    if (tiled && (n_tiles < 0 || (n_tiles > 0 && !tile_ids))) return fail(RTC_ERR_INVALID, "bad tile list");
    int tiles_x = (cam->hsize + T - 1) / T, tiles_y = (cam->vsize + T - 1) / T;
    if (tiled)
        for (int i = 0; i < n_tiles; i++)
            if (tile_ids[i] < 0 || tile_ids[i] >= tiles_x * tiles_y) return fail(RTC_ERR_INVALID, "tile id %d out of range", tile_ids[i]);
    size_t n_px = tiled ? (size_t)n_tiles * T * T : (size_t)cam->hsize * cam->vsize;
    if (stats_out) memset(stats_out, 0, sizeof(*stats_out));
    if (n_px == 0) return 0;
    HIP_TRY(hipSetDevice(scene->device));
    std::lock_guard<std::mutex> out_lock(scene->out_mutex);
    // buffers grown on demand and kept in the handle: a frame-sized hipMalloc + hipFree per call costs more than the kernel
    auto grow = [&](auto*& ptr, size_t& have, size_t want, size_t elem) -> int {
        if (have >= want) return 0;
        if (ptr) { (void)hipFree(ptr); ptr = nullptr; have = 0; }
        void* p = nullptr;
        HIP_TRY(hipMalloc(&p, want * elem));
        ptr = static_cast<std::remove_reference_t<decltype(ptr)>>(p); have = want;
        return 0;
    };
    if (int rc = grow(scene->out_rgb, scene->out_rgb_px, n_px, 3 * sizeof(double))) return rc;
    if (hit_id_out) if (int rc = grow(scene->out_id, scene->out_id_px, n_px, sizeof(int32_t))) return rc;
    if (tiled) if (int rc = grow(scene->out_tiles, scene->out_tiles_n, (size_t)n_tiles, sizeof(int32_t))) return rc;
    if (!scene->out_rays) HIP_TRY(hipMalloc((void**)&scene->out_rays, 14 * sizeof(int64_t)));
    if (!scene->out_e0) HIP_TRY(hipEventCreate(&scene->out_e0));
    if (!scene->out_e1) HIP_TRY(hipEventCreate(&scene->out_e1));
    double* d_rgb = scene->out_rgb; int32_t* d_id = hit_id_out ? scene->out_id : nullptr; int64_t* d_rays = scene->out_rays;
    int32_t* d_tiles = tiled ? scene->out_tiles : nullptr;
    hipEvent_t e0 = scene->out_e0, e1 = scene->out_e1;
    int rc = 0;
#define TRY_C(expr) HIP_TRY(expr)
    if (tiled) TRY_C(hipMemsetAsync(d_rgb, 0, n_px * 3 * sizeof(double), nullptr));
    if (hit_id_out) TRY_C(hipMemsetAsync(d_id, 0xFF, n_px * sizeof(int32_t), nullptr));
    const bool want_work = stats_out && opts && (opts->flags & RTC_FLAG_COUNT_WORK);
    TRY_C(hipMemsetAsync(d_rays, 0, 14 * sizeof(int64_t), nullptr));
    if (tiled) TRY_C(hipMemcpy(d_tiles, tile_ids, (size_t)n_tiles * sizeof(int32_t), hipMemcpyHostToDevice));
    TRY_C(hipEventRecord(e0, nullptr));
    rc = launch(scene, cam, opts, n_tiles, d_tiles, d_rgb, d_id, d_rays, want_work ? d_rays + 4 : nullptr, nullptr);
    if (rc) return rc;
    TRY_C(hipEventRecord(e1, nullptr));
    TRY_C(hipMemcpy(rgb_out, d_rgb, n_px * 3 * sizeof(double), hipMemcpyDeviceToHost));     // null stream: after the kernel
    if (hit_id_out) TRY_C(hipMemcpy(hit_id_out, d_id, n_px * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (stats_out) {
        int64_t host_rays[14];
        TRY_C(hipMemcpy(host_rays, d_rays, sizeof(host_rays), hipMemcpyDeviceToHost));
        float ms = 0;
        TRY_C(hipEventElapsedTime(&ms, e0, e1));
        memcpy(stats_out->rays, host_rays, 4 * sizeof(int64_t));
        if (want_work) { memcpy(stats_out->work, host_rays + 4, 8 * sizeof(int64_t)); stats_out->wide_node_visits = host_rays[13]; }
        stats_out->shadow_rays_culled = host_rays[12];
        stats_out->kernel_ms = ms;
    }
#undef TRY_C
    return 0;
}

}  // namespace

extern "C" {

int rtc_version(void) { return RTC_VERSION; }

int rtc_host_alloc(size_t bytes, void** out) {
    if (!out) return fail(RTC_ERR_INVALID, "null argument");
    *out = nullptr;
    HIP_TRY(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return 0;
}

void rtc_host_free(void* ptr) {
    if (ptr) (void)hipHostFree(ptr);
}

int rtc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

const char* rtc_last_error(void) { return g_err.c_str(); }

int rtc_scene_create(const rtc_scene_desc* desc, int device, rtc_scene** out) {
    if (!desc || !out) return fail(RTC_ERR_INVALID, "null argument");
    *out = nullptr;
    if (int rc = validate(*desc)) return rc;
    int n = rtc_device_count();
    if (n <= 0) return fail(RTC_ERR_NO_DEVICE, "no GPU visible to HIP");
    if (device < 0 || device >= n) return fail(RTC_ERR_NO_DEVICE, "device %d out of range (%d visible)", device, n);
    HIP_TRY(hipSetDevice(device));
    rtc_scene_desc d = *desc;
    for (int attempt = 0; ; attempt++) {
        rtc_scene* sc = nullptr;
        int rc = 0;
        try {                                           // the BVH build and the staging copies allocate host memory
            sc = new rtc_scene();
            sc->device = device;
            sc->n_leaves = d.n_leaves;
            rc = create_impl(d, sc);
        } catch (const std::bad_alloc&) {
            rc = fail(RTC_ERR_NOMEM, "out of host memory while building the scene");
        } catch (const std::exception& e) {
            rc = fail(RTC_ERR_INVALID, "scene build failed: %s", e.what());
        }
        if (!rc) { *out = sc; return 0; }
        if (sc) rtc_scene_destroy(sc);
        // RTC_BUILD_ON_DEVICE is a request for speed, not a different scene: a mesh whose GPU-built tree comes out deeper than
        // the traversal stack (tens of millions of triangles) gets the host's build, which is made to fit (build_bvh's depth budget)
        if (rc == RTC_ERR_LIMIT && attempt == 0 && (d.build_flags & RTC_BUILD_ON_DEVICE)) {
            if (getenv("RTC_BVH_INFO")) fprintf(stderr, "[rtc] create: %s -- building the tree on the host instead\n", rtc_last_error());
            d.build_flags &= ~RTC_BUILD_ON_DEVICE;
            continue;
        }
        return rc;
    }
}

void rtc_scene_destroy(rtc_scene* scene) {
    if (!scene) return;
    (void)hipSetDevice(scene->device);
    scene->shared.release(); scene->mem64.release(); scene->mem32.release();
    scene->tbvh_mem.release();
    for (Workspace* w : scene->workspaces) { w->release(); delete w; }
    scene->workspaces.clear();
    if (scene->out_rgb) (void)hipFree(scene->out_rgb);
    if (scene->out_id) (void)hipFree(scene->out_id);
    if (scene->out_ppm) (void)hipFree(scene->out_ppm);
    if (scene->out_rays) (void)hipFree(scene->out_rays);
    if (scene->out_tiles) (void)hipFree(scene->out_tiles);
    if (scene->out_e0) (void)hipEventDestroy(scene->out_e0);
    if (scene->out_e1) (void)hipEventDestroy(scene->out_e1);
    delete scene;
}

int rtc_scene_build_bvh_device(rtc_scene* scene) {
    if (!scene) return fail(RTC_ERR_INVALID, "null scene");
    HIP_TRY(hipSetDevice(scene->device));
    std::lock_guard<std::mutex> lock(scene->ws_mutex);
    HIP_TRY(hipDeviceSynchronize());              // no frame of this scene may still be reading the old tree
    return rebuild_triangle_bvh_device(scene);
}

namespace {
int update_triangles_impl(rtc_scene* scene, int64_t n_tris, const double* tri_geom, const double* tri_norm, bool refit, bool on_device);
}

int rtc_scene_update_triangles(rtc_scene* scene, int64_t n_tris, const double* tri_geom, const double* tri_norm) {
    return update_triangles_impl(scene, n_tris, tri_geom, tri_norm, false, false);
}

int rtc_scene_refit_triangles(rtc_scene* scene, int64_t n_tris, const double* tri_geom, const double* tri_norm) {
    return update_triangles_impl(scene, n_tris, tri_geom, tri_norm, true, false);
}

int rtc_scene_update_triangles_device(rtc_scene* scene, int64_t n_tris, const double* tri_geom_dev, const double* tri_norm_dev, int refit) {
    return update_triangles_impl(scene, n_tris, tri_geom_dev, tri_norm_dev, refit != 0, true);
}

}  // extern "C"

namespace {
int update_triangles_impl(rtc_scene* scene, int64_t n_tris, const double* tri_geom, const double* tri_norm, bool refit, bool on_device) {
    if (!scene || !tri_geom || !tri_norm) return fail(RTC_ERR_INVALID, "null argument");
    if (n_tris != scene->n_tris) return fail(RTC_ERR_INVALID, "scene has %lld triangles, got %lld", (long long)scene->n_tris, (long long)n_tris);
    if (n_tris == 0) return 0;
    HIP_TRY(hipSetDevice(scene->device));
    std::lock_guard<std::mutex> lock(scene->ws_mutex);
    HIP_TRY(hipDeviceSynchronize());
    const size_t n = 12 * (size_t)n_tris;
    try {
    if (on_device) {
        hipLaunchKernelGGL(rtc::tri_rows_from_device_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, (long long)n, tri_geom, tri_norm,
                           const_cast<double*>(scene->d64.tri_geom), const_cast<double*>(scene->d64.tri_norm),
                           const_cast<float*>(scene->d32.tri_geom), const_cast<float*>(scene->d32.tri_norm));
        HIP_TRY(hipGetLastError());
        return refit ? refit_triangle_bvh_device(scene) : rebuild_triangle_bvh_device(scene);
    }
    std::vector<float> g32 = cast_vec<float>(tri_geom, n), n32 = cast_vec<float>(tri_norm, n);
    HIP_TRY(hipMemcpy(const_cast<double*>(scene->d64.tri_geom), tri_geom, n * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(const_cast<double*>(scene->d64.tri_norm), tri_norm, n * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(const_cast<float*>(scene->d32.tri_geom), g32.data(), n * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(const_cast<float*>(scene->d32.tri_norm), n32.data(), n * sizeof(float), hipMemcpyHostToDevice));
    return refit ? refit_triangle_bvh_device(scene) : rebuild_triangle_bvh_device(scene);
    } catch (const std::bad_alloc&) {
        return fail(RTC_ERR_NOMEM, "out of host memory while staging the triangle rows");
    }
}
}  // namespace

extern "C" {

int rtc_scene_get_info(const rtc_scene* scene, rtc_scene_info* out) {
    if (!scene || !out) return fail(RTC_ERR_INVALID, "null argument");
    out->device = scene->device;
    out->bvh_nodes = scene->bvh_nodes; out->bvh_depth = scene->bvh_depth; out->bvh_max_leaf = scene->bvh_max_leaf;
    out->device_bytes_fp64 = (int64_t)(scene->shared.bytes + scene->mem64.bytes);
    out->device_bytes_fp32 = (int64_t)(scene->shared.bytes + scene->mem32.bytes);
    return 0;
}

int rtc_scene_last_launch(const rtc_scene* scene, rtc_launch_info* out) {
    if (!scene || !out) return fail(RTC_ERR_INVALID, "null argument");
    *out = scene->last_launch;
    return 0;
}

int rtc_render(rtc_scene* scene, const rtc_camera* cam, const rtc_render_opts* opts, double* rgb_out, int32_t* hit_id_out,
               rtc_stats* stats_out) {
    return render_host(scene, cam, opts, 0, nullptr, false, rgb_out, hit_id_out, stats_out);
}

int rtc_render_tiles(rtc_scene* scene, const rtc_camera* cam, const rtc_render_opts* opts, int32_t n_tiles, const int32_t* tile_ids,
                     double* rgb_out, int32_t* hit_id_out, rtc_stats* stats_out) {
    return render_host(scene, cam, opts, n_tiles, tile_ids, true, rgb_out, hit_id_out, stats_out);
}

int rtc_render_device(rtc_scene* scene, const rtc_camera* cam, const rtc_render_opts* opts, int32_t n_tiles, const int32_t* tile_ids_dev,
                      double* rgb_dev, int32_t* hit_id_dev, int64_t* rays_dev, void* stream) {
    int64_t* work = (rays_dev && opts && (opts->flags & RTC_FLAG_COUNT_WORK)) ? rays_dev + 4 : nullptr;
    return launch(scene, cam, opts, n_tiles, tile_ids_dev, rgb_dev, hit_id_dev, rays_dev, work, static_cast<hipStream_t>(stream));
}

}  // extern "C"

// ---- the entry point that allocates nothing ----
// Its launch-order buffers (per-unit wave costs, two order lists) live in a file-scope device array instead of a
// workspace: RTC_NOALLOC_MAX_UNITS 8x8 pixel blocks at most.  Each GPU holds its own copy of the array at its own
// address, looked up once per device with hipGetSymbolAddress; host code only ever uses that pointer.
#define RTC_NOALLOC_MAX_UNITS 65536LL                 /* 4 194 304 pixels */
#define RTC_NOALLOC_ENTRIES(units) ((((units) + 3 * ((units) / 4 + 64)) + RTC_BLOCK / 64 - 1) / (RTC_BLOCK / 64) * (RTC_BLOCK / 64))
#define RTC_NOALLOC_MAX_DEVICES 64
namespace {
struct NoAllocScratch {
    uint32_t cost[RTC_NOALLOC_MAX_UNITS];
    int32_t order[2][RTC_NOALLOC_ENTRIES(RTC_NOALLOC_MAX_UNITS)];
    int32_t used;
    uint32_t ray_parts[RTC_NOALLOC_ENTRIES(RTC_NOALLOC_MAX_UNITS)][8];      // per-wave ray counts (DFrame::work in the variants without CSG)
};
}  // namespace
static __device__ __align__(256) unsigned char scratch[sizeof(NoAllocScratch)];
namespace {
struct NoAllocState {
    std::mutex m;
    NoAllocScratch* ptr = nullptr;        // this device's copy of `scratch`
    const rtc_scene* scene = nullptr; FbKey key{}; int age = 0, cur = 0; bool valid = false;
} g_noalloc[RTC_NOALLOC_MAX_DEVICES];
}  // namespace

extern "C" {

int rtc_render_device_noalloc(rtc_scene* scene, const rtc_camera* cam, const rtc_render_opts* opts, int32_t n_tiles, const int32_t* tile_ids_dev,
                              double* rgb_dev, int32_t* hit_id_dev, int64_t* rays_dev, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    int T, precision, depth;
    double eps, eps_rel;
    if (int rc = check_render_args(scene, cam, opts, &T, &precision, &depth, &eps, &eps_rel)) return rc;
    if (!rgb_dev) return fail(RTC_ERR_INVALID, "null output buffer");
    if (scene->device < 0 || scene->device >= RTC_NOALLOC_MAX_DEVICES) return fail(RTC_ERR_LIMIT, "device %d beyond the %d this entry point keeps state for", scene->device, RTC_NOALLOC_MAX_DEVICES);
    const int tiles_x = (cam->hsize + T - 1) / T, tiles_y = (cam->vsize + T - 1) / T;
    if (!tile_ids_dev) n_tiles = tiles_x * tiles_y;
    if (n_tiles < 0) return fail(RTC_ERR_INVALID, "negative tile count");
    if (n_tiles == 0) return 0;
    const long long blocks = ((long long)n_tiles * T * T + RTC_BLOCK - 1) / RTC_BLOCK, n_units = blocks * (RTC_BLOCK / 64);
    if (n_units > RTC_NOALLOC_MAX_UNITS)
        return fail(RTC_ERR_LIMIT, "%lld pixels in %d tiles: rtc_render_device_noalloc renders at most %lld pixels per call (use rtc_render_device)",
                    (long long)n_tiles * T * T, n_tiles, RTC_NOALLOC_MAX_UNITS * 64);
    HIP_TRY(hipSetDevice(scene->device));
    NoAllocState& st = g_noalloc[scene->device];
    std::lock_guard<std::mutex> lock(st.m);
    if (!st.ptr) {                                     // first call on this device (it is current): where its copy of the array lives
        void* p = nullptr;
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return fail(RTC_ERR_HIP, "hipGetSymbolAddress(scratch) returned no address");
        st.ptr = static_cast<NoAllocScratch*>(p);
    }
    NoAllocScratch* const sp = st.ptr;
    rtc::DFrame F;
    F.n_tiles = n_tiles; F.tile_size = T; F.tiles_x = tiles_x; F.tile_ids = tile_ids_dev;
    F.rgb = rgb_dev; F.hit_id = hit_id_dev; F.rays = reinterpret_cast<unsigned long long*>(rays_dev);
    const bool prof = rays_dev && opts && (opts->flags & RTC_FLAG_COUNT_WORK);
    F.work = prof ? reinterpret_cast<unsigned long long*>(rays_dev + 4) : nullptr;
    F.max_depth = depth; F.seed = opts ? opts->seed : 0;
    F.n_slots = 0;
    F.replicas = 0;      // this entry point never reads back whether the order holds split units, so their parts leave the spare lanes idle
    F.order = nullptr; F.cost = nullptr;
    F.n_units = (int32_t)n_units; F.n_entries = (int32_t)n_units;
    dim3 grid((unsigned)blocks);
    const long long n_entries = RTC_NOALLOC_ENTRIES(n_units);
    bool feedback = !(opts && (opts->flags & RTC_FLAG_NO_FEEDBACK));
    {
        // Under HIP graph capture nothing executes when it is enqueued, so the host-side record of "the order kernel of call
        // k has run before call k + 1 reads its list" does not hold (a captured call may be replayed later, never, or many
        // times): captured calls render in plain order and leave the launch-order state alone.
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (stream && hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) feedback = false;
        (void)hipGetLastError();
    }
    bool resort = false;
    if (feedback) {
        FbKey key;
        memset(&key, 0, sizeof(key));
        key.cam = *cam; key.precision = precision; key.depth = depth; key.tile_size = T; key.n_tiles = n_tiles; key.tile_ids = tile_ids_dev;
        if (st.scene != scene || memcmp(&key, &st.key, sizeof(key)) != 0) { st.scene = scene; st.key = key; st.age = 0; st.valid = false; }
        resort = st.age < 2;
        st.age++;
        if (resort) {
            // the array keeps whatever the last call left in it: the costs must start at zero
            HIP_TRY(hipMemsetAsync(sp->cost, 0, (size_t)n_units * sizeof(uint32_t), stream));
            F.cost = sp->cost;
        }
        if (st.valid) {       // every entry the order kernel wrote, unused ones included (they hold -1): no read-back of how many are used
            F.order = sp->order[st.cur];
            F.n_entries = (int32_t)n_entries;
            grid = dim3((unsigned)(n_entries / (RTC_BLOCK / 64)));
        }
    }
    uint32_t* ray_parts = (rays_dev && g_ray_parts && (!scene->has_csg || RTC_CSG_SLOTS) && !prof) ? &sp->ray_parts[0][0] : nullptr;      // every wave of the grid writes its slot
    if (ray_parts) { F.work = reinterpret_cast<unsigned long long*>(ray_parts); F.n_slots = (int32_t)RTC_NOALLOC_ENTRIES(RTC_NOALLOC_MAX_UNITS); }
    launch_trace(scene, cam, F, grid, precision, prof, eps, eps_rel, stream);
    if (ray_parts) {
        const int n_waves = (int)grid.x * (RTC_BLOCK / 64);
        hipLaunchKernelGGL(sum_rays_kernel, dim3((unsigned)std::min(64, (n_waves + 255) / 256)), dim3(256), 0, stream,
                           reinterpret_cast<const uint4*>(ray_parts), n_waves, F.rays);
    }
    HIP_TRY(hipGetLastError());
    if (feedback && resort) {
        const int waves_per_simd = scene->has_csg ? 2 : (precision == 64 ? RTC_WAVES_F64 : RTC_WAVES_F32);
        hipLaunchKernelGGL(order_kernel, dim3(1), dim3(1024), 0, stream, sp->cost, sp->order[1 - st.cur], (int)n_units, (int)n_entries,
                           scene->n_cus * 4 * waves_per_simd, st.age >= 2 ? g_split4 : 0.0f, g_split16, &sp->used, g_split_t);
        HIP_TRY(hipGetLastError());
        st.cur = 1 - st.cur;
        st.valid = true;
    }
    return 0;
}

int rtc_quantize_device(const double* rgb_dev, int32_t hsize, int32_t vsize, double max_component, uint8_t* out_dev, void* stream) {
    if (!rgb_dev || !out_dev || hsize < 1 || vsize < 1) return fail(RTC_ERR_INVALID, "bad argument");
    long long n = (long long)hsize * vsize * 3;
    hipLaunchKernelGGL(quantize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), rgb_dev, hsize,
                       vsize, max_component, out_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

int rtc_frame_max_device(const double* rgb_dev, int64_t n_values, double* max_out_dev, void* stream) {
    if (!rgb_dev || !max_out_dev || n_values < 1) return fail(RTC_ERR_INVALID, "bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(frame_max_init_kernel, dim3(1), dim3(1), 0, st, rgb_dev, max_out_dev);
    unsigned blocks = (unsigned)std::min<long long>((n_values + 255) / 256, 2048);
    hipLaunchKernelGGL(frame_max_kernel, dim3(blocks), dim3(256), 0, st, rgb_dev, (long long)n_values, max_out_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

int rtc_ppm_body_device(const double* rgb_dev, int32_t hsize, int32_t vsize, const double* max_component_dev, uint8_t* out_dev, void* stream) {
    if (!rgb_dev || !out_dev || hsize < 1 || vsize < 1) return fail(RTC_ERR_INVALID, "bad argument");
    long long n = (long long)hsize * vsize * 3;
    hipLaunchKernelGGL(ppm_body_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), rgb_dev, hsize,
                       vsize, max_component_dev, out_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

int64_t rtc_ppm_size(int32_t hsize, int32_t vsize) {
    if (hsize < 1 || vsize < 1) return 0;
    char header[64];
    int h = snprintf(header, sizeof(header), "P6\n%d %d\n255\n", hsize, vsize);
    return (int64_t)h + 3LL * hsize * vsize + 1;
}

int rtc_render_ppm(rtc_scene* scene, const rtc_camera* cam, const rtc_render_opts* opts, uint8_t* clamped_out, uint8_t* scaled_out,
                   int64_t capacity, int64_t* length_out, rtc_stats* stats_out) {
    int T, precision, depth;
    double eps, eps_rel;
    if (int rc = check_render_args(scene, cam, opts, &T, &precision, &depth, &eps, &eps_rel)) return rc;
    char header[64];
    const int hlen = snprintf(header, sizeof(header), "P6\n%d %d\n255\n", cam->hsize, cam->vsize);
    const size_t n_px = (size_t)cam->hsize * cam->vsize, body = 3 * n_px, total = (size_t)hlen + body + 1;
    if (length_out) *length_out = (int64_t)total;
    if ((clamped_out || scaled_out) && capacity < (int64_t)total) return fail(RTC_ERR_INVALID, "ppm buffer too small: %lld < %zu", (long long)capacity, total);
    if (stats_out) memset(stats_out, 0, sizeof(*stats_out));
    HIP_TRY(hipSetDevice(scene->device));
    std::lock_guard<std::mutex> out_lock(scene->out_mutex);
    // frame, counters, the maximum and the encoded file stay in the handle between calls (see render_host)
    if (scene->out_rgb_px < n_px) {
        if (scene->out_rgb) { (void)hipFree(scene->out_rgb); scene->out_rgb = nullptr; scene->out_rgb_px = 0; }
        HIP_TRY(hipMalloc((void**)&scene->out_rgb, body * sizeof(double)));
        scene->out_rgb_px = n_px;
    }
    if (scene->out_ppm_bytes < total + sizeof(double)) {
        if (scene->out_ppm) { (void)hipFree(scene->out_ppm); scene->out_ppm = nullptr; scene->out_ppm_bytes = 0; }
        HIP_TRY(hipMalloc((void**)&scene->out_ppm, total + 2 * sizeof(double)));
        scene->out_ppm_bytes = total + sizeof(double);
    }
    if (!scene->out_rays) HIP_TRY(hipMalloc((void**)&scene->out_rays, 14 * sizeof(int64_t)));
    double* d_rgb = scene->out_rgb; int64_t* d_rays = scene->out_rays;
    double* d_max = reinterpret_cast<double*>(scene->out_ppm);             // 8 bytes in front of the file
    uint8_t* d_out = scene->out_ppm + sizeof(double);
    int rc = 0;
#define TRY_C(expr) HIP_TRY(expr)
    TRY_C(hipMemsetAsync(d_rays, 0, 14 * sizeof(int64_t), nullptr));
    const bool want_work = stats_out && opts && (opts->flags & RTC_FLAG_COUNT_WORK);
    rc = launch(scene, cam, opts, 0, nullptr, d_rgb, nullptr, d_rays, want_work ? d_rays + 4 : nullptr, nullptr);
    if (rc) return rc;
    TRY_C(hipMemcpy(d_out, header, (size_t)hlen, hipMemcpyHostToDevice));
    const uint8_t newline = 10;
    TRY_C(hipMemcpy(d_out + hlen + body, &newline, 1, hipMemcpyHostToDevice));
    for (int variant = 0; variant < 2; variant++) {
        uint8_t* dst = variant == 0 ? clamped_out : scaled_out;
        if (!dst) continue;
        if (variant == 1) { rc = rtc_frame_max_device(d_rgb, (int64_t)body, d_max, nullptr); if (rc) return rc; }
        rc = rtc_ppm_body_device(d_rgb, cam->hsize, cam->vsize, variant == 1 ? d_max : nullptr, d_out + hlen, nullptr);
        if (rc) return rc;
        TRY_C(hipMemcpy(dst, d_out, total, hipMemcpyDeviceToHost));
    }
    if (stats_out) {
        TRY_C(hipMemcpy(stats_out->rays, d_rays, 4 * sizeof(int64_t), hipMemcpyDeviceToHost));
        if (want_work) TRY_C(hipMemcpy(stats_out->work, d_rays + 4, 8 * sizeof(int64_t), hipMemcpyDeviceToHost));
        if (want_work) TRY_C(hipMemcpy(&stats_out->wide_node_visits, d_rays + 13, sizeof(int64_t), hipMemcpyDeviceToHost));
        TRY_C(hipMemcpy(&stats_out->shadow_rays_culled, d_rays + 12, sizeof(int64_t), hipMemcpyDeviceToHost));
    }
#undef TRY_C
    return 0;
}

int rtc_scatter_tiles_device(const double* tiles_dev, const int32_t* tile_ids_dev, int32_t n_slots, int32_t tile_size, int32_t hsize,
                             int32_t vsize, double* frame_dev, void* stream) {
    if (!tiles_dev || !tile_ids_dev || !frame_dev || n_slots < 0 || hsize < 1 || vsize < 1 || tile_size < 1 || tile_size > 1024)
        return fail(RTC_ERR_INVALID, "bad argument");
    if (n_slots == 0) return 0;
    long long n = (long long)n_slots * tile_size * tile_size;
    if ((n + 255) / 256 > 0x7FFFFFFFLL) return fail(RTC_ERR_LIMIT, "too many tiles");
    int tiles_x = (hsize + tile_size - 1) / tile_size;
    hipLaunchKernelGGL(scatter_tiles_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), tiles_dev,
                       tile_ids_dev, (int)n_slots, (int)tile_size, tiles_x, (int)hsize, (int)vsize, frame_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

int rtc_scatter_id_tiles_device(const int32_t* tiles_dev, const int32_t* tile_ids_dev, int32_t n_slots, int32_t tile_size, int32_t hsize,
                                int32_t vsize, int32_t* frame_dev, void* stream) {
    if (!tiles_dev || !tile_ids_dev || !frame_dev || n_slots < 0 || hsize < 1 || vsize < 1 || tile_size < 1 || tile_size > 1024)
        return fail(RTC_ERR_INVALID, "bad argument");
    if (n_slots == 0) return 0;
    long long n = (long long)n_slots * tile_size * tile_size;
    if ((n + 255) / 256 > 0x7FFFFFFFLL) return fail(RTC_ERR_LIMIT, "too many tiles");
    int tiles_x = (hsize + tile_size - 1) / tile_size;
    hipLaunchKernelGGL(scatter_id_tiles_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), tiles_dev,
                       tile_ids_dev, (int)n_slots, (int)tile_size, tiles_x, (int)hsize, (int)vsize, frame_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"
