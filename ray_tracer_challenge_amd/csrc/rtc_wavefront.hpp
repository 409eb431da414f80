// rtc_wavefront.hpp -- the trace loop as a wavefront pipeline: rays live in HBM queues, one
// queue segment per bounce level, and small kernels sweep a level at a time:
//
//   level L:  k_trace   nearest hit of every ray of the level            (BVH traversal only)
//             k_shadow  one work item per (ray, light): how many of the light's cells see the
//                       hit; plus one item per ray for the refraction container pass (n1, n2)
//             k_shade   patterns + lighting from those counts; appends the reflection and
//                       refraction rays to level L+1 (ballot-compacted, one atomic per wave)
//   then      k_reduce  L = deepest-1 .. 0: colour += child colour * factor, in the reference's
//                       order surface + reflected*reflectance + refracted*(1-reflectance)
//             k_output  level-0 colours and hit ids -> framebuffer
//
// Same device functions as the megakernel (rtc_device.hpp), so the same arithmetic; what changes
// is that each kernel keeps only its own state in registers (occupancy) and that every level is
// dense (no lane idles while its neighbours chase a secondary ray).  The ray tree is kept
// (parent -> child links), so colours are summed in a fixed order whatever order the atomics
// appended the children in: frames are bit-reproducible.
#pragma once
#include "rtc_device.hpp"

#define RTC_WF_MAX_LEVELS 8

namespace rtc {

template <class R> struct WfRay { R o[3], d[3]; };
template <class R> struct WfHit { R t, u, v; int32_t leaf; int32_t meta; };     // meta = remaining | path << 4; leaf -2 = lane outside the image
template <class R> struct WfLink { int32_t refl, refr; R w_refl, w_refr; };
template <class R> struct WfCol { R c[3]; };
template <class R> struct WfCst { R x_last, y_last; int32_t x_odd, y_leaf; };   // ContainerVisitor state between the two passes
template <class R> struct WfPt { R p[3]; };

struct WfCtrl {
    int32_t count[RTC_WF_MAX_LEVELS];   // rays per level (count[0] set by the host)
    int32_t overflow;                   // children dropped because the pool was full
    int32_t pad[7];
};

template <class R> struct WfSpace {
    WfRay<R>* ray; WfHit<R>* hit; int32_t* pix; WfLink<R>* link; WfCol<R>* col; WfCst<R>* cst; WfPt<R>* opt;
    uint32_t* mask;                     // [cap][n_lights][mask_words] bit set = that light cell is shadowed
    int32_t mask_words;
    WfCtrl* ctrl;
    int32_t cap;
};

template <class R> struct WfParams {
    WfSpace<R> W;
    DFrame F;
    R shade_eps, shade_eps_rel;
    int32_t level;
};

__device__ __forceinline__ int wf_start(const WfCtrl* c, int level) {
    int s = 0;
    for (int l = 0; l < level; l++) s += c->count[l];
    return s;
}

// block-level reduction of the ray / work counters, then a few atomics per block
template <bool PROF, bool CSG>
__device__ __forceinline__ void flush_counters(Counters<PROF, CSG>& cnt, const DFrame& F, unsigned int* red /* LDS, >= 4 * RTC_BLOCK/64 */) {
    if (F.rays) {
        __syncthreads();
        for (int k = 0; k < 4; k++) {
            unsigned int v = cnt.c[k];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if ((threadIdx.x & 63) == 0) red[k * (RTC_BLOCK / 64) + (threadIdx.x >> 6)] = v;
        }
        __syncthreads();
        if (threadIdx.x < 4) {
            unsigned long long s = 0;
            for (int q = 0; q < RTC_BLOCK / 64; q++) s += red[threadIdx.x * (RTC_BLOCK / 64) + q];
            if (s) atomicAdd(F.rays + threadIdx.x, s);
        }
    }
    if (PROF && F.work) {
        for (int k = 0; k < W_COUNT; k++) {
            unsigned int v = cnt.w[k];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if ((threadIdx.x & 63) == 0 && v) atomicAdd(F.work + k, (unsigned long long)v);
        }
    }
}
template <bool PROF, bool CSG> __device__ __forceinline__ void zero_counters(Counters<PROF, CSG>& cnt) {
    for (int k = 0; k < 4; k++) cnt.c[k] = 0;
    for (int k = 0; k < (PROF ? W_COUNT : 1); k++) cnt.w[k] = 0;
}

// pixel of work item w of the frame (tile order: one wave = one 8x8 block of a T x T tile)
__device__ __forceinline__ bool wf_pixel(const DFrame& F, int hsize, int vsize, long long w, int& x, int& y, size_t& out_index) {
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

template <class R> __device__ __forceinline__ Ray<R> load_ray(const WfRay<R>& r) {
    Ray<R> ray = {mk<R>(r.o[0], r.o[1], r.o[2]), mk<R>(r.d[0], r.d[1], r.d[2])};
    return ray;
}
template <class R> __device__ __forceinline__ void store_ray(WfRay<R>& r, V3<R> o, V3<R> d) {
    r.o[0] = o.x; r.o[1] = o.y; r.o[2] = o.z; r.d[0] = d.x; r.d[1] = d.y; r.d[2] = d.z;
}

// ---------------------------------------------------------------- k_trace
template <class R, bool PROF>
__global__ __launch_bounds__(RTC_BLOCK) void wf_trace(DScene<R> S, DCamera<R> cam, WfParams<R> P) {
    __shared__ int s_stack[RTC_BVH_STACK * RTC_BLOCK];
    int* stack = s_stack + threadIdx.x;
    Counters<PROF> cnt; zero_counters(cnt);
    const WfSpace<R>& W = P.W;
    const int count = W.ctrl->count[P.level], start = wf_start(W.ctrl, P.level);
    for (long long i = (long long)blockIdx.x * RTC_BLOCK + threadIdx.x; i < count; i += (long long)gridDim.x * RTC_BLOCK) {
        int idx = start + (int)i;
        Ray<R> ray;
        if (P.level == 0) {
            int x, y; size_t o;
            if (!wf_pixel(P.F, cam.hsize, cam.vsize, i, x, y, o)) { W.hit[idx].leaf = -2; continue; }
            ray = ray_for_pixel(cam, x, y);
            store_ray(W.ray[idx], ray.o, ray.d);
            W.hit[idx].meta = P.F.max_depth | (1 << 4);
            W.pix[idx] = y * cam.hsize + x;
            cnt.c[0]++;
        } else {
            ray = load_ray(W.ray[idx]);
        }
        NearestVisitor<R, Counters<PROF>> nv(S, ray, cnt);
        traverse_objects(S, ray, R(0), nv, stack);
        WfHit<R>& h = W.hit[idx];
        h.t = nv.best; h.u = nv.u; h.v = nv.v; h.leaf = nv.leaf;
    }
    flush_counters(cnt, P.F, reinterpret_cast<unsigned int*>(s_stack));
}

// ---------------------------------------------------------------- lean triangle-BVH kernels
// Same visitors, but only their tri() side is instantiated here: no analytic leaves, no CSG, no
// normals -- so these kernels keep few registers and run at several waves per SIMD.
template <class R, bool PROF>
__global__ __launch_bounds__(RTC_BLOCK) void wf_trace_tri(DScene<R> S, WfParams<R> P) {
    __shared__ int s_stack[RTC_BVH_STACK * RTC_BLOCK];
    int* stack = s_stack + threadIdx.x;
    Counters<PROF> cnt; zero_counters(cnt);
    const WfSpace<R>& W = P.W;
    const int count = W.ctrl->count[P.level], start = wf_start(W.ctrl, P.level);
    for (long long i = (long long)blockIdx.x * RTC_BLOCK + threadIdx.x; i < count; i += (long long)gridDim.x * RTC_BLOCK) {
        int idx = start + (int)i;
        WfHit<R>& h = W.hit[idx];
        if (h.leaf == -2) continue;
        Ray<R> ray = load_ray(W.ray[idx]);
        NearestVisitor<R, Counters<PROF>> nv(S, ray, cnt);
        nv.best = h.t; nv.leaf = h.leaf;
        traverse_tris(S, ray, R(0), nv, stack);
        if (nv.leaf != h.leaf || nv.best != h.t) { h.t = nv.best; h.u = nv.u; h.v = nv.v; h.leaf = nv.leaf; }
    }
    flush_counters(cnt, P.F, reinterpret_cast<unsigned int*>(s_stack));
}

template <class R, bool PROF>
__global__ __launch_bounds__(RTC_BLOCK) void wf_shadow_tri(DScene<R> S, WfParams<R> P) {
    __shared__ int s_stack[RTC_BVH_STACK * RTC_BLOCK];
    int* stack = s_stack + threadIdx.x;
    Counters<PROF> cnt; zero_counters(cnt);
    const WfSpace<R>& W = P.W;
    const int count = W.ctrl->count[P.level], start = wf_start(W.ctrl, P.level);
    const long long items = (long long)count * (S.n_lights + 1);
    for (long long it = (long long)blockIdx.x * RTC_BLOCK + threadIdx.x; it < items; it += (long long)gridDim.x * RTC_BLOCK) {
        int l = (int)(it / count), idx = start + (int)(it % count);
        const int leaf = W.hit[idx].leaf;
        if (leaf < 0) continue;
        if (l < S.n_lights) {
            V3<R> over_point = mk<R>(W.opt[idx].p[0], W.opt[idx].p[1], W.opt[idx].p[2]);
            LightShape ls = light_shape(S, l);
            uint32_t* m = W.mask + ((size_t)idx * S.n_lights + l) * W.mask_words;
            const int meta = W.hit[idx].meta;
            for (int w = 0; w * 32 < ls.samples; w++) {
                uint32_t bits = m[w], add = 0;
                for (int b = 0; b < 32 && w * 32 + b < ls.samples; b++) {
                    if ((bits >> b) & 1u) continue;
                    int cell = w * 32 + b;
                    V3<R> pos = light_sample<R>(S, l, ls, cell % ls.usteps, cell / ls.usteps, P.F.seed, (uint32_t)W.pix[idx], (uint32_t)(meta >> 4), 0);
                    V3<R> v = pos - over_point;
                    R distance = sqrt_(dot(v, v));
                    Ray<R> r = {over_point, v * (R(1) / distance)};
                    ShadowVisitor<R, Counters<PROF>> vis(S, r, distance, cnt);
                    traverse_tris(S, r, R(0), vis, stack);
                    if (vis.shadowed) add |= 1u << b;
                }
                if (add) m[w] = bits | add;
            }
        } else {
            WfCst<R>& st = W.cst[idx];
            if (st.x_odd < 0) continue;                       // no container pass for this hit
            Ray<R> ray = load_ray(W.ray[idx]);
            ContainerVisitor<R, Counters<PROF>> cv(S, ray, W.hit[idx].t, leaf, cnt);
            cv.x_odd = st.x_odd != 0; cv.x_last = st.x_last; cv.y_leaf = st.y_leaf; cv.y_last = st.y_last;
            traverse_tris(S, ray, -inf_<R>(), cv, stack);
            st.x_odd = cv.x_odd ? 1 : 0; st.x_last = cv.x_last; st.y_leaf = cv.y_leaf; st.y_last = cv.y_last;
        }
    }
    flush_counters(cnt, P.F, reinterpret_cast<unsigned int*>(s_stack));
}

// ---------------------------------------------------------------- k_shadow over the objects (+ container pass)
template <class R, bool PROF>
__global__ __launch_bounds__(RTC_BLOCK) void wf_shadow(DScene<R> S, WfParams<R> P) {
    __shared__ int s_stack[RTC_BVH_STACK * RTC_BLOCK];
    int* stack = s_stack + threadIdx.x;
    Counters<PROF> cnt; zero_counters(cnt);
    const WfSpace<R>& W = P.W;
    const int count = W.ctrl->count[P.level], start = wf_start(W.ctrl, P.level);
    const long long items = (long long)count * (S.n_lights + 1);
    for (long long it = (long long)blockIdx.x * RTC_BLOCK + threadIdx.x; it < items; it += (long long)gridDim.x * RTC_BLOCK) {
        int l = (int)(it / count), idx = start + (int)(it % count);   // neighbouring lanes: neighbouring rays, same light
        const WfHit<R> h = W.hit[idx];
        if (h.leaf < 0) continue;
        Ray<R> ray = load_ray(W.ray[idx]);
        if (l < S.n_lights) {
            HitCtx<R> c = make_ctx(S, ray, h.t, h.leaf, h.u, h.v, P.shade_eps, P.shade_eps_rel);
            if (l == 0) { W.opt[idx].p[0] = c.over_point.x; W.opt[idx].p[1] = c.over_point.y; W.opt[idx].p[2] = c.over_point.z; }
            LightShape ls = light_shape(S, l);
            uint32_t* m = W.mask + ((size_t)idx * S.n_lights + l) * W.mask_words;
            for (int w = 0; w * 32 < ls.samples; w++) {
                uint32_t bits = 0;
                for (int b = 0; b < 32 && w * 32 + b < ls.samples; b++) {
                    int cell = w * 32 + b;
                    V3<R> pos = light_sample<R>(S, l, ls, cell % ls.usteps, cell / ls.usteps, P.F.seed, (uint32_t)W.pix[idx], (uint32_t)(h.meta >> 4), 0);
                    V3<R> v = pos - c.over_point;
                    R distance = sqrt_(dot(v, v));
                    Ray<R> r = {c.over_point, v * (R(1) / distance)};
                    cnt.c[1]++;
                    ShadowVisitor<R, Counters<PROF>> vis(S, r, distance, cnt);
                    traverse_objects(S, r, R(0), vis, stack);
                    if (vis.shadowed) bits |= 1u << b;
                }
                m[w] = bits;
            }
        } else {
            int material = S.leaf_material[h.leaf];
            WfCst<R> st;
            st.x_odd = -1; st.y_leaf = -1; st.x_last = st.y_last = -inf_<R>();
            if (needs_containers(S.mat[10 * (size_t)material + 8], h.meta & 15)) {
                cnt.work(W_CONTAINER);
                ContainerVisitor<R, Counters<PROF>> cv(S, ray, h.t, h.leaf, cnt);
                traverse_objects(S, ray, -inf_<R>(), cv, stack);
                st.x_odd = cv.x_odd ? 1 : 0; st.x_last = cv.x_last; st.y_leaf = cv.y_leaf; st.y_last = cv.y_last;
            }
            W.cst[idx] = st;
        }
    }
    flush_counters(cnt, P.F, reinterpret_cast<unsigned int*>(s_stack));
}

// ---------------------------------------------------------------- k_shade
// appends `want` lanes' children to the next level: one atomic per wave
__device__ __forceinline__ int wf_append(int32_t* counter, bool want) {
    unsigned long long mask = __ballot(want);
    if (mask == 0) return -1;
    int lane = threadIdx.x & 63;
    int leader = __ffsll((long long)mask) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, __popcll(mask));
    base = __shfl(base, leader, 64);
    return want ? base + __popcll(mask & ((1ull << lane) - 1ull)) : -1;
}

template <class R, bool PROF>
__global__ __launch_bounds__(RTC_BLOCK) void wf_shade(DScene<R> S, WfParams<R> P) {
    __shared__ unsigned int s_red[4 * (RTC_BLOCK / 64)];
    Counters<PROF> cnt; zero_counters(cnt);
    const WfSpace<R>& W = P.W;
    const int count = W.ctrl->count[P.level], start = wf_start(W.ctrl, P.level);
    const int next_start = start + count;
    const long long padded = ((long long)count + 63) & ~63LL;     // whole waves stay together for the ballots
    for (long long i = (long long)blockIdx.x * RTC_BLOCK + threadIdx.x; i < padded; i += (long long)gridDim.x * RTC_BLOCK) {
        int idx = start + (int)i;
        bool live = i < count;
        WfHit<R> h;
        h.leaf = -1;
        if (live) h = W.hit[idx];
        V3<R> surface = mk<R>(0, 0, 0);
        Secondary<R> sec;
        sec.refl = sec.refr = false;
        HitCtx<R> c;
        int remaining = 0; uint32_t path = 0;
        if (live && h.leaf >= 0) {
            remaining = h.meta & 15; path = (uint32_t)(h.meta >> 4);
            Ray<R> ray = load_ray(W.ray[idx]);
            cnt.work(W_SHADE);
            c = make_ctx(S, ray, h.t, h.leaf, h.u, h.v, P.shade_eps, P.shade_eps_rel);
            if (S.mat_program[c.material] >= 0) cnt.work(W_PATTERN);
            V3<R> pcolor = surface_color(S, c.material, c.leaf, c.over_point);
            for (int l = 0; l < S.n_lights; l++) {
                LightShape ls = light_shape(S, l);
                const uint32_t* m = W.mask + ((size_t)idx * S.n_lights + l) * W.mask_words;
                int shadowed = 0;
                for (int w = 0; w * 32 < ls.samples; w++) shadowed += __popc(m[w]);
                surface = surface + lighting(S, c, pcolor, l, ls.samples - shadowed, cnt, P.F.seed, (uint32_t)W.pix[idx], path);
            }
            R n1 = R(1), n2 = R(1);
            if (needs_containers(S.mat[10 * (size_t)c.material + 8], remaining)) {
                const WfCst<R> st = W.cst[idx];
                container_result(S, h.leaf, st.x_odd != 0, st.x_last, st.y_leaf, st.y_last, n1, n2);
            }
            sec = secondary_rays(S, c, remaining, n1, n2);
        }
        int k_refl = wf_append(&W.ctrl->count[P.level + 1], sec.refl);
        int k_refr = wf_append(&W.ctrl->count[P.level + 1], sec.refr);
        if (!live) continue;
        WfLink<R> link;
        link.refl = link.refr = -1;
        link.w_refl = sec.w_refl; link.w_refr = sec.w_refr;
        if (sec.refl) {
            int child = next_start + k_refl;
            if (child < W.cap) {
                store_ray(W.ray[child], c.over_point, c.reflectv);
                W.hit[child].meta = (remaining - 1) | (int)((2 * path) << 4);
                W.pix[child] = W.pix[idx];
                link.refl = child;
                cnt.c[2]++;
            } else atomicAdd(&W.ctrl->overflow, 1);
        }
        if (sec.refr) {
            int child = next_start + k_refr;
            if (child < W.cap) {
                store_ray(W.ray[child], c.under_point, sec.refr_dir);
                W.hit[child].meta = (remaining - 1) | (int)((2 * path + 1) << 4);
                W.pix[child] = W.pix[idx];
                link.refr = child;
                cnt.c[3]++;
            } else atomicAdd(&W.ctrl->overflow, 1);
        }
        W.link[idx] = link;
        W.col[idx].c[0] = surface.x; W.col[idx].c[1] = surface.y; W.col[idx].c[2] = surface.z;
    }
    flush_counters(cnt, P.F, s_red);
}

// a level's count may exceed what the pool holds: clamp it before the level is swept
__global__ void wf_clamp(WfCtrl* ctrl, int level, int cap) {
    int s = wf_start(ctrl, level);
    if (s + ctrl->count[level] > cap) ctrl->count[level] = cap > s ? cap - s : 0;
}

// ---------------------------------------------------------------- k_reduce / k_output
template <class R>
__global__ __launch_bounds__(RTC_BLOCK) void wf_reduce(WfParams<R> P) {
    const WfSpace<R>& W = P.W;
    const int count = W.ctrl->count[P.level], start = wf_start(W.ctrl, P.level);
    for (long long i = (long long)blockIdx.x * RTC_BLOCK + threadIdx.x; i < count; i += (long long)gridDim.x * RTC_BLOCK) {
        int idx = start + (int)i;
        if (W.hit[idx].leaf < 0) continue;
        WfLink<R> k = W.link[idx];
        if (k.refl < 0 && k.refr < 0) continue;
        // surface += reflected * reflectance + refracted * (1 - reflectance)   (renderer.py:523, 525)
        V3<R> add = mk<R>(0, 0, 0);
        if (k.refl >= 0) add = add + mk<R>(W.col[k.refl].c[0], W.col[k.refl].c[1], W.col[k.refl].c[2]) * k.w_refl;
        if (k.refr >= 0) add = add + mk<R>(W.col[k.refr].c[0], W.col[k.refr].c[1], W.col[k.refr].c[2]) * k.w_refr;
        W.col[idx].c[0] += add.x; W.col[idx].c[1] += add.y; W.col[idx].c[2] += add.z;
    }
}

template <class R>
__global__ __launch_bounds__(RTC_BLOCK) void wf_output(WfParams<R> P, int hsize, int vsize) {
    const WfSpace<R>& W = P.W;
    const int count = W.ctrl->count[0];
    for (long long i = (long long)blockIdx.x * RTC_BLOCK + threadIdx.x; i < count; i += (long long)gridDim.x * RTC_BLOCK) {
        int x, y; size_t o;
        if (!wf_pixel(P.F, hsize, vsize, i, x, y, o)) continue;
        int leaf = W.hit[i].leaf;
        double r = 0, g = 0, b = 0;
        if (leaf >= 0) { r = (double)W.col[i].c[0]; g = (double)W.col[i].c[1]; b = (double)W.col[i].c[2]; }
        P.F.rgb[3 * o] = r; P.F.rgb[3 * o + 1] = g; P.F.rgb[3 * o + 2] = b;
        if (P.F.hit_id) P.F.hit_id[o] = leaf;
    }
}

}  // namespace rtc
