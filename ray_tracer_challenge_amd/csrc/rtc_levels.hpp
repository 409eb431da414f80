// rtc_levels.hpp -- the trace loop one bounce level at a time.
//
// In the single kernel a pixel's whole ray tree (up to 2^(depth+1) - 1 rays) runs in one lane, one
// ray after the other, so a frame cannot end before its dearest pixel has walked ~60 rays in series,
// and every other lane of that wave idles meanwhile.  Here each ray is its own work item:
//
//   lv_primary   one lane per pixel: camera ray, nearest hit, shadow rays, surface colour -> canvas;
//                the reflection / refraction rays it would recurse into are appended to the ray pool
//                (level 1), ballot-compacted, one atomic per wave
//   lv_level L   one lane per ray of level L (dense: every lane has a ray): the same, own colour
//                (weight x surface) -> col[], children -> level L + 1
//   lv_gather    one lane per pixel that has children: canvas += the subtree sums, depth-first in the
//                fixed order own + reflection subtree + refraction subtree, whatever order the
//                atomics appended the rays in: frames are bit-reproducible
//
// Same device functions as the single kernel (shade_ray), so the same arithmetic per ray; only the
// order in which a pixel's contributions are added differs (a tree instead of a running sum).
// A full pool is not an error: a ray that finds no room walks its subtree in-lane (trace_tree).
#pragma once
#include "rtc_device.hpp"

#define RTC_LV_MAX_LEVELS 8

namespace rtc {

template <class R> struct LvRay { R o[3], d[3], w[3]; int32_t pixel; int32_t meta; };   // meta = remaining | path << 3
struct LvCtrl {
    int32_t count[RTC_LV_MAX_LEVELS];   // rays appended per level (may run past the pool: readers clamp)
    int32_t overflow;                   // rays that found the pool full and were traced in-lane
    int32_t pad[7];
};
template <class R> struct LvSpace {
    LvRay<R>* ray;        // [cap]   the pool: level 1 from 0, level L + 1 right after level L
    double* col;          // [cap][3] own colour of the ray (weight x surface)
    int32_t* kids;        // [cap]   -1 = no children, else first_child * 4 | has_reflection | has_refraction << 1
    int32_t* kids0;       // [n_work] the same for the primary rays (work item order)
    LvCtrl* ctrl;
    int32_t cap;
};

// first pool index and (clamped) ray count of a level >= 1
__device__ __forceinline__ void lv_range(const LvCtrl* c, int cap, int level, int& start, int& count) {
    start = 0;
    for (int l = 1; l < level; l++) { int n = c->count[l]; start += n; if (start > cap) start = cap; }
    count = c->count[level];
    if (count > cap - start) count = cap - start;
}

// Appends this lane's children to level `level + 1` (callable from divergent code: the ballots see
// the active lanes).  If they do not fit, the ray is given up (kids = RTC_LV_REDO, returns false) and
// lv_redo walks its whole subtree in one lane afterwards.
#define RTC_LV_REDO (-2)
template <class R, class CNT> struct EmitToPool {
    const LvSpace<R>& Q; int next_start; int level;
    V3<R> weight; int remaining; uint32_t path; uint32_t pixel; CNT& cnt;
    int32_t kids;          // out
    __device__ __forceinline__ void store(int idx, V3<R> o, V3<R> d, V3<R> w, int rem, uint32_t p) const {
        LvRay<R> r;
        r.o[0] = o.x; r.o[1] = o.y; r.o[2] = o.z; r.d[0] = d.x; r.d[1] = d.y; r.d[2] = d.z; r.w[0] = w.x; r.w[1] = w.y; r.w[2] = w.z;
        r.pixel = (int32_t)pixel; r.meta = rem | (int32_t)(p << 3);
        Q.ray[idx] = r;
    }
    __device__ __forceinline__ bool operator()(bool refl, V3<R> refl_o, V3<R> refl_d, R w_refl, bool refr, V3<R> refr_o, V3<R> refr_d, R w_refr) {
        const int n = (refl ? 1 : 0) + (refr ? 1 : 0);
        // wave-aggregated reservation of n consecutive slots per lane
        const unsigned long long active = __ballot(1);
        const unsigned long long one = __ballot(n >= 1), two = __ballot(n == 2);
        const int lane = threadIdx.x & 63;
        const unsigned long long below = (1ULL << lane) - 1ULL;
        const int total = __popcll(one) + __popcll(two);
        const int leader = __ffsll((long long)active) - 1;
        int base = 0;
        if (total > 0) {
            if (lane == leader) base = atomicAdd(&Q.ctrl->count[level + 1], total);
            base = __shfl(base, leader, 64);
        }
        if (n == 0) return true;
        const long long at = (long long)next_start + base + __popcll(one & below) + __popcll(two & below);
        if (base < 0 || at + n > Q.cap) {          // pool full (or the counter wrapped): lv_redo takes this ray
            atomicAdd(&Q.ctrl->overflow, 1);
            kids = RTC_LV_REDO;
            return false;
        }
        int k = (int)at;
        if (refl) { store(k++, refl_o, refl_d, weight * w_refl, remaining - 1, 2 * path); cnt.c[2]++; }
        if (refr) { store(k, refr_o, refr_d, weight * w_refr, remaining - 1, 2 * path + 1); cnt.c[3]++; }
        kids = (int)at * 4 + (refl ? 1 : 0) + (refr ? 2 : 0);
        return true;
    }
};

template <class R, bool PROF, bool CSG>
__global__ __launch_bounds__(RTC_BLOCK, (CSG ? 2 : sizeof(R) == 8 ? RTC_WAVES_F64 : RTC_WAVES_F32)) void lv_primary(
    DScene<R> S, DCamera<R> cam, DFrame F, R shade_eps, R shade_eps_rel, LvSpace<R> Q) {
    __shared__ int s_stack[RTC_BVH_STACK * RTC_BLOCK];
    int* stack = s_stack + threadIdx.x;
    const int lane = threadIdx.x & 63;
    typedef Counters<PROF, CSG> CNT;
    CNT cnt;
    zero_counters(cnt);
    const long long t_start = F.cost ? (long long)wall_clock64() : 0;
    bool lane_on;
    const long long unit = wave_unit(F, lane_on);
    if (unit >= 0 && lane_on) {
        const long long w = unit * 64 + lane;
        int x, y;
        size_t o;
        int32_t kids = -1;
        if (work_pixel(F, cam.hsize, cam.vsize, w, x, y, o)) {
            Ray<R> ray = ray_for_pixel(cam, x, y);
            cnt.c[0] = 1;
            const uint32_t pixel = (uint32_t)(y * cam.hsize + x);
            EmitToPool<R, CNT> emit = {Q, 0, 0, mk<R>(1, 1, 1), F.max_depth, 1u, pixel, cnt, -1};
            int leaf;
            V3<R> c = shade_ray(S, ray, F.max_depth, 1u, shade_eps, shade_eps_rel, stack, cnt, F.seed, pixel, leaf, emit);
            kids = emit.kids;
            F.rgb[3 * o] = (double)c.x; F.rgb[3 * o + 1] = (double)c.y; F.rgb[3 * o + 2] = (double)c.z;
            if (F.hit_id) F.hit_id[o] = leaf;
        }
        Q.kids0[w] = kids;
    }
    if (F.cost && unit >= 0 && lane == 0) {
        long long dt = (long long)wall_clock64() - t_start;
        atomicAdd(F.cost + unit, (uint32_t)(dt < 0 ? 0 : dt > 0x0FFFFFFFLL ? 0x0FFFFFFFLL : dt));
    }
    flush_counters(cnt, F, reinterpret_cast<unsigned int*>(s_stack));
}

template <class R, bool PROF, bool CSG>
__global__ __launch_bounds__(RTC_BLOCK, (CSG ? 2 : sizeof(R) == 8 ? RTC_WAVES_F64 : RTC_WAVES_F32)) void lv_level(
    DScene<R> S, DFrame F, R shade_eps, R shade_eps_rel, LvSpace<R> Q, int level) {
    __shared__ int s_stack[RTC_BVH_STACK * RTC_BLOCK];
    int* stack = s_stack + threadIdx.x;
    typedef Counters<PROF, CSG> CNT;
    CNT cnt;
    zero_counters(cnt);
    int start, count;
    lv_range(Q.ctrl, Q.cap, level, start, count);
    const int next_start = start + count;
    // whole waves walk the level together (the append ballots want every lane of a wave in the loop)
    for (long long base = (long long)blockIdx.x * RTC_BLOCK; base < count; base += (long long)gridDim.x * RTC_BLOCK) {
        const long long i = base + threadIdx.x;
        if (i < count) {
            const int idx = start + (int)i;
            const LvRay<R> r = Q.ray[idx];
            Ray<R> ray = {mk<R>(r.o[0], r.o[1], r.o[2]), mk<R>(r.d[0], r.d[1], r.d[2])};
            const V3<R> weight = mk<R>(r.w[0], r.w[1], r.w[2]);
            const int remaining = r.meta & 7;
            const uint32_t path = (uint32_t)r.meta >> 3;
            EmitToPool<R, CNT> emit = {Q, next_start, level, weight, remaining, path, (uint32_t)r.pixel, cnt, -1};
            int leaf;
            V3<R> c = shade_ray(S, ray, remaining, path, shade_eps, shade_eps_rel, stack, cnt, F.seed, (uint32_t)r.pixel, leaf, emit);
            c = weight * c;
            Q.col[3 * (size_t)idx] = (double)c.x; Q.col[3 * (size_t)idx + 1] = (double)c.y; Q.col[3 * (size_t)idx + 2] = (double)c.z;
            Q.kids[idx] = emit.kids;
        }
    }
    flush_counters(cnt, F, reinterpret_cast<unsigned int*>(s_stack));
}

// Rays whose children found the pool full: their whole subtree in one lane, like the single kernel.
// (Their nearest-hit traversal is walked a second time, so work counters -- not ray counters --
// count it twice.)  Exits at once when nothing overflowed.
template <class R, bool PROF, bool CSG>
__global__ __launch_bounds__(RTC_BLOCK, (CSG ? 2 : sizeof(R) == 8 ? RTC_WAVES_F64 : RTC_WAVES_F32)) void lv_redo(
    DScene<R> S, DCamera<R> cam, DFrame F, R shade_eps, R shade_eps_rel, LvSpace<R> Q, long long n_work, int max_level) {
    __shared__ int s_stack[RTC_BVH_STACK * RTC_BLOCK];
    if (Q.ctrl->overflow == 0) return;
    int* stack = s_stack + threadIdx.x;
    typedef Counters<PROF, CSG> CNT;
    CNT cnt;
    zero_counters(cnt);
    int start, count;
    lv_range(Q.ctrl, Q.cap, max_level, start, count);
    const long long used = (long long)start + count;
    for (long long i = (long long)blockIdx.x * RTC_BLOCK + threadIdx.x; i < n_work + used; i += (long long)gridDim.x * RTC_BLOCK) {
        int leaf;
        if (i < n_work) {
            if (Q.kids0[i] != RTC_LV_REDO) continue;
            int x, y; size_t o;
            if (!work_pixel(F, cam.hsize, cam.vsize, i, x, y, o)) continue;
            V3<R> c = trace_tree(S, ray_for_pixel(cam, x, y), mk<R>(1, 1, 1), F.max_depth, 1u, shade_eps, shade_eps_rel, stack, cnt, F.seed,
                                 (uint32_t)(y * cam.hsize + x), leaf);
            F.rgb[3 * o] = (double)c.x; F.rgb[3 * o + 1] = (double)c.y; F.rgb[3 * o + 2] = (double)c.z;
            Q.kids0[i] = -1;
        } else {
            const int idx = (int)(i - n_work);
            if (Q.kids[idx] != RTC_LV_REDO) continue;
            const LvRay<R> r = Q.ray[idx];
            Ray<R> ray = {mk<R>(r.o[0], r.o[1], r.o[2]), mk<R>(r.d[0], r.d[1], r.d[2])};
            V3<R> c = trace_tree(S, ray, mk<R>(r.w[0], r.w[1], r.w[2]), r.meta & 7, (uint32_t)r.meta >> 3, shade_eps, shade_eps_rel, stack, cnt,
                                 F.seed, (uint32_t)r.pixel, leaf);
            Q.col[3 * (size_t)idx] = (double)c.x; Q.col[3 * (size_t)idx + 1] = (double)c.y; Q.col[3 * (size_t)idx + 2] = (double)c.z;
            Q.kids[idx] = -1;
        }
    }
    flush_counters(cnt, F, reinterpret_cast<unsigned int*>(s_stack));
}

// canvas[pixel] += subtree sums, depth-first, own + reflection subtree + refraction subtree
template <class R>
__global__ __launch_bounds__(RTC_BLOCK) void lv_gather(DFrame F, int hsize, int vsize, LvSpace<R> Q, long long n_work) {
    const long long w = (long long)blockIdx.x * RTC_BLOCK + threadIdx.x;
    if (w >= n_work) return;
    const int32_t k0 = Q.kids0[w];
    if (k0 < 0) return;
    int x, y;
    size_t o;
    if (!work_pixel(F, hsize, vsize, w, x, y, o)) return;
    double acc[RTC_LV_MAX_LEVELS][3];
    int32_t kid[RTC_LV_MAX_LEVELS], next[RTC_LV_MAX_LEVELS], stage[RTC_LV_MAX_LEVELS];
    int d = 0;
    acc[0][0] = F.rgb[3 * o]; acc[0][1] = F.rgb[3 * o + 1]; acc[0][2] = F.rgb[3 * o + 2];
    kid[0] = k0; next[0] = k0 >> 2; stage[0] = 0;
    while (true) {
        int go = -1;
        if (stage[d] == 0) { stage[d] = 1; if (kid[d] >= 0 && (kid[d] & 1)) go = next[d]++; }
        if (go < 0 && stage[d] == 1) { stage[d] = 2; if (kid[d] >= 0 && (kid[d] & 2)) go = next[d]++; }
        if (go >= 0 && go < Q.cap && d + 1 < RTC_LV_MAX_LEVELS) {
            d++;
            acc[d][0] = Q.col[3 * (size_t)go]; acc[d][1] = Q.col[3 * (size_t)go + 1]; acc[d][2] = Q.col[3 * (size_t)go + 2];
            kid[d] = Q.kids[go]; next[d] = kid[d] >> 2; stage[d] = 0;
            continue;
        }
        if (go >= 0) continue;          // malformed link: skip it
        if (d == 0) break;
        acc[d - 1][0] += acc[d][0]; acc[d - 1][1] += acc[d][1]; acc[d - 1][2] += acc[d][2];
        d--;
    }
    F.rgb[3 * o] = acc[0][0]; F.rgb[3 * o + 1] = acc[0][1]; F.rgb[3 * o + 2] = acc[0][2];
}

}  // namespace rtc
