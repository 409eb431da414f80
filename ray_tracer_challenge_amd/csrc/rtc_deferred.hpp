// rtc_deferred.hpp -- the trace loop as three kernels around a pool of shaded hits in HBM.
//
// The reference shades a hit by tracing a shadow ray per light sample and adding the light's terms
// right away (shade_hit, renderer.py:506-526 -> intensity_at / is_shadowed :919-922, 964-973, 552-560).
// Shadow rays are most of a frame's rays (dragons: 2.06 M of 4.62 M) and all of them are independent of
// one another, yet inside a pixel's lane they sit between the dependent steps of the pixel's ray tree and
// their traversal state shares one register allocation with the shading state.  Here they are deferred:
//
//   trace_kernel    one lane per pixel: the pixel's whole ray tree of NEAREST-hit queries (explicit bounce
//                   stack, renderer.py:400-406 / 336-353 unrolled).  Every shaded hit becomes a record in the
//                   hit pool -- its weight in the pixel, surface colour, over_point and, per light, the
//                   mean diffuse + specular term of `lighting` (renderer.py:1066-1090) -- with a PENDING
//                   mark for every light whose shadow rays matter.  No shadow ray is traced here.
//   shadow_kernel   is_shadowed for every pending (record, light, group of light cells): waves read rows of
//                   64 candidates, ballot the pending ones, compact them into a wave-local queue in LDS and
//                   trace full wavefronts of any-hit rays; the lit-cell counts are added into the records.
//   resolve_kernel  one lane per pixel: walks the pixel's records in the order the tree produced them and
//                   adds weight x sum over lights (ambient + mean x lit / cells) -- the same running sum, in
//                   the same order, as tracing the shadow rays in place: frames are bit-reproducible.
//
// Records: the first hit of pixel slot s is record s (no allocation); later hits of a tree take records
// from the secondary pool, one wave-aggregated atomic per wave and bounce (RTC_POOL_SHARDS counters, by
// workgroup, so that the atomics do not meet on one address).  A pixel that finds the pool full gives its
// tree up (state 2): redo_kernel traces such pixels in one lane, shadow rays in place, afterwards.
#pragma once
#include "rtc_device.hpp"

#define RTC_POOL_SHARDS 8
#define RTC_LIT_PENDING (-(1 << 30))   /* lit[] of a (record, light) whose shadow rays are still to be traced; counts are added to it */
#ifndef RTC_SHADOW_ROWS
#define RTC_SHADOW_ROWS 4              /* consecutive rows of 64 candidates one shadow wave compacts and traces */
#endif

namespace rtc {

struct PoolCtrl {
    uint32_t count[RTC_POOL_SHARDS];   // records taken from each shard of the secondary pool (may run past sub_cap: readers clamp)
    uint32_t failed;                   // pixels given up for redo_kernel
    uint32_t pad[7];
};

// SoA over record index r in [0, cap): r < n_slots = first hit of pixel slot r, the rest = secondary pool
template <class R> struct HitPool {
    R* rec;            // [10][cap] weight xyz, surface colour xyz, material ambient, over_point xyz
    R* mean;           // [n_lights][3][cap] mean over the light's cells of diffuse + specular (written where lit is pending)
    int32_t* lit;      // [n_lights][cap] 0 = ambient only (no cell contributes), else RTC_LIT_PENDING + lit cells
    int32_t* next;     // [cap] next record of the same pixel, -1 = none
    uint32_t* tag;     // [2][cap] pixel, ray number (the jitter hash's inputs)
    int32_t* state;    // [n_slots] 0 = no hit, 1 = records from slot on, 2 = given up (redo)
    PoolCtrl* ctrl;    // this frame's counters
    PoolCtrl* ctrl_next;   // the next frame's, zeroed by this frame's trace kernel
    int32_t cap, n_slots, sub_cap, n_lights;
    int32_t cells_per_group;   // area lights: light cells one shadow work item covers
};

enum { REC_WEIGHT = 0, REC_COLOR = 3, REC_AMBIENT = 6, REC_OVER = 7, REC_FIELDS = 10 };

// A record of the secondary pool for every active lane with `need`, -1 where the shard is full.
// Callable from divergent code: the ballot sees the active lanes.
template <class R> __device__ __forceinline__ int pool_alloc(const HitPool<R>& P, int shard, bool need) {
    const unsigned long long mask = __ballot(need);
    int r = -1;
    if (need) {
        const int lane = threadIdx.x & 63;
        const int n = __popcll(mask), rank = __popcll(mask & ((1ULL << lane) - 1ULL));
        uint32_t base = 0;
        if (rank == 0) base = atomicAdd(&P.ctrl->count[shard], (uint32_t)n);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);      // the first active lane here is the one with rank 0
        const uint32_t idx = base + (uint32_t)rank;
        if (idx < (uint32_t)P.sub_cap) r = P.n_slots + shard * P.sub_cap + (int)idx;
    }
    return r;
}

// The pixel's ray tree, nearest hits only; every shaded hit -> one record.  Returns the slot's state.
template <class R, class CNT>
__device__ __forceinline__ int trace_tree_deferred(const DScene<R>& S, Ray<R> ray, int remaining, R shade_eps, R shade_eps_rel, int* stack, CNT& cnt,
                                                   uint64_t seed, uint32_t pixel, int slot, const HitPool<R>& P, int& first_leaf) {
    Bounce<R> pending[RTC_MAX_BOUNCE_STACK];
    int np = 0;
    V3<R> weight = mk<R>(1, 1, 1);
    uint32_t path = 1u;
    bool first = true;
    int prev = -1, state = 0;
    first_leaf = -1;
    const int shard = blockIdx.x & (RTC_POOL_SHARDS - 1);
    const size_t cap = (size_t)P.cap;
    while (true) {
        NearestVisitor<R, CNT> nv(S, ray, cnt);
        traverse(S, ray, R(0), nv, stack);
        const bool need = nv.leaf >= 0;
        int r = -1;
        if (first) { first_leaf = nv.leaf; if (need) r = slot; }
        else r = pool_alloc(P, shard, need);
        first = false;
        if (need) {
            if (r < 0) { state = 2; break; }
            state = 1;
            cnt.work(W_SHADE);
            HitCtx<R> c = make_ctx(S, ray, nv.best, nv.leaf, nv.u, nv.v, shade_eps, shade_eps_rel);
            if (S.mat_program[c.material] >= 0) cnt.work(W_PATTERN);
            R n1 = R(1), n2 = R(1);
            if (S.refraction_matters && needs_containers(S.mat[10 * (size_t)c.material + 8], remaining)) container_n1n2(S, ray, c.t, c.leaf, stack, cnt, n1, n2);
            Secondary<R> sec = secondary_rays(S, c, remaining, n1, n2);
            PushPending<R, CNT> push = {pending, np, weight, remaining, path, cnt};
            push(sec.refl, c.over_point, c.reflectv, sec.w_refl, sec.refr, c.under_point, sec.refr_dir, sec.w_refr);
            V3<R> pcolor = surface_color<R, CNT::kCsg>(S, c.material, c.leaf, c.over_point);
            R* rec = P.rec + r;
            rec[(REC_WEIGHT + 0) * cap] = weight.x; rec[(REC_WEIGHT + 1) * cap] = weight.y; rec[(REC_WEIGHT + 2) * cap] = weight.z;
            rec[(REC_COLOR + 0) * cap] = pcolor.x; rec[(REC_COLOR + 1) * cap] = pcolor.y; rec[(REC_COLOR + 2) * cap] = pcolor.z;
            rec[REC_AMBIENT * cap] = S.mat[10 * (size_t)c.material + 3];
            rec[(REC_OVER + 0) * cap] = c.over_point.x; rec[(REC_OVER + 1) * cap] = c.over_point.y; rec[(REC_OVER + 2) * cap] = c.over_point.z;
            P.tag[r] = pixel; P.tag[cap + r] = path;
            for (int l = 0; l < S.n_lights; l++) {
                V3<R> amb, mean;
                if (light_terms(S, c, pcolor, l, cnt, seed, pixel, path, amb, mean)) {
                    R* m = P.mean + (size_t)(3 * l) * cap + r;
                    m[0] = mean.x; m[cap] = mean.y; m[2 * cap] = mean.z;
                    P.lit[(size_t)l * cap + r] = RTC_LIT_PENDING;
                } else {
                    // no cell of the light contributes (it is behind the surface): `lighting` gives the ambient term whatever
                    // is_shadowed would say, so no shadow ray is needed
                    P.lit[(size_t)l * cap + r] = 0;
                    cnt.c[4] += (unsigned int)light_shape(S, l).samples;
                }
            }
            P.next[r] = -1;
            if (prev >= 0) P.next[prev] = r;
            prev = r;
        }
        if (np == 0) break;
        np--;
        ray.o = pending[np].o; ray.d = pending[np].d; weight = pending[np].w; remaining = pending[np].remaining; path = pending[np].path;
    }
    if (state == 2) {
        // given up: nothing of this pixel is left for the other kernels (its records' shadow work is withdrawn, its
        // counters start over in redo_kernel)
        if (prev >= 0)
            for (int r = slot; r >= 0; r = P.next[r])
                for (int l = 0; l < S.n_lights; l++) P.lit[(size_t)l * cap + r] = 0;
        atomicAdd(&P.ctrl->failed, 1u);
        zero_counters(cnt);
        cnt.c[0] = 1;
    }
    return state;
}

// lighting's result for light l of record r (renderer.py:1088-1090): ambient alone when no cell is lit
template <class R>
__device__ __forceinline__ V3<R> record_light(const DScene<R>& S, const HitPool<R>& P, int r, int l, V3<R> pcolor, R ambient) {
    const size_t cap = (size_t)P.cap;
    const R* li = S.light_intensity + 3 * l;
    V3<R> effective = pcolor * mk<R>(li[0], li[1], li[2]);
    V3<R> amb = effective * ambient;
    int lit = P.lit[(size_t)l * cap + r];
    if (lit < 0) lit -= RTC_LIT_PENDING;
    if (lit == 0) return amb;
    const R* m = P.mean + (size_t)(3 * l) * cap + r;
    V3<R> mean = mk<R>(m[0], m[cap], m[2 * cap]);
    return amb + mean * (R(lit) / R(light_shape(S, l).samples));
}

// the colour of pixel slot `slot` from its records (state 1)
template <class R> __device__ __forceinline__ V3<R> resolve_pixel(const DScene<R>& S, const HitPool<R>& P, int slot) {
    const size_t cap = (size_t)P.cap;
    V3<R> color = mk<R>(0, 0, 0);
    for (int r = slot; r >= 0; r = P.next[r]) {
        const R* rec = P.rec + r;
        V3<R> weight = mk<R>(rec[(REC_WEIGHT + 0) * cap], rec[(REC_WEIGHT + 1) * cap], rec[(REC_WEIGHT + 2) * cap]);
        V3<R> pcolor = mk<R>(rec[(REC_COLOR + 0) * cap], rec[(REC_COLOR + 1) * cap], rec[(REC_COLOR + 2) * cap]);
        R ambient = rec[REC_AMBIENT * cap];
        V3<R> surface = mk<R>(0, 0, 0);
        for (int l = 0; l < S.n_lights; l++) surface = surface + record_light(S, P, r, l, pcolor, ambient);
        color = color + weight * surface;
    }
    return color;
}

// shadow work items are (light, group of light cells, record): an area light's cells are dealt out in groups of
// cells_per_group so that one lane never walks a whole 10 x 10 light alone
__device__ __forceinline__ int light_groups(int samples, int cells_per_group) { return (samples + cells_per_group - 1) / cells_per_group; }

}  // namespace rtc
