// rtc_trace_kernel.hpp -- the trace kernel template, shared by the two translation units of librtc_hip.so:
// rtc_api.hip instantiates the fp32 variants, rtc_trace_f64.hip the fp64 variants (compiled with other optimiser settings:
// see the Makefile).
#pragma once
#include "rtc_device.hpp"

namespace rtc {

// CSG: 0 = the scene has no CSG object and no perturbed pattern, 1 = it has, 2 = and some CSG root is evaluated list-free,
// 3 = like 0, with instanced meshes (rtc_scene_desc::inst)
// REPL: the pixels of split parts are traced by 4 or 16 lanes each (never with PROF)
// WAVES: resident waves per SIMD the register budget is cut for (512 / WAVES registers per lane).  fp64 without CSG is built
// for 4 (128 registers, 18 - 40 spilled) and for 2 (its ~150 registers, nothing spilled, three waves resident in effect): which
// is faster depends on the scene -- tree walks want the waves, fp64 pattern arithmetic wants the registers -- and launch()
// times both for every view.
#define RTC_WAVES_OF(R, CSG) (((CSG) == 1 || (CSG) == 2) ? 2 : sizeof(R) == 8 ? RTC_WAVES_F64 : RTC_WAVES_F32)
template <class R, bool PROF, int CSG, bool FAR, bool REPL, int WAVES>
__global__ __launch_bounds__((CSG == 1 || CSG == 2) ? RTC_BLOCK_CSG : RTC_BLOCK_PLAIN, WAVES) void trace_kernel(rtc::DScene<R> S, rtc::DCamera<R> cam, rtc::DFrame F, R shade_eps, R shade_eps_rel) {
    using CNT = rtc::Counters<PROF, (CSG == 1 || CSG == 2), FAR, (CSG == 2), REPL, (CSG == 3)>;
    __shared__ int s_stack[RTC_BVH_STACK * CNT::kBlock];
    int* stack = s_stack + threadIdx.x;
    const int lane = threadIdx.x & 63;
    CNT cnt;
    rtc::zero_counters(cnt);
    const long long t_start = F.cost ? (long long)wall_clock64() : 0;
    bool lane_on;
    int pixel_lane;
    const long long unit = rtc::wave_unit(F, lane_on, pixel_lane, cnt.rep, cnt.nrep, CNT::kBlock / 64);
    int x, y;
    size_t o;
    if (unit >= 0 && lane_on && rtc::work_pixel(F, cam.hsize, cam.vsize, unit * 64 + pixel_lane, x, y, o)) {
        rtc::Ray<R> ray = rtc::ray_for_pixel(cam, x, y);
        int hit_leaf;
#ifdef RTC_TIMELINE
        long long tl_start = wall_clock64();
#endif
        rtc::V3<R> c = rtc::mk<R>(0, 0, 0);
        hit_leaf = -1;
        if (rtc::may_hit_scene<R, CNT>(S, ray))          // (a ray that meets neither tree's box is black: the sky waves end here)
            c = rtc::trace_pixel(S, ray, F.max_depth, shade_eps, shade_eps_rel, stack, cnt, F.seed, (uint32_t)(y * cam.hsize + x), hit_leaf);
#ifdef RTC_TIMELINE   /* development build only (tools/timeline.py): start/end ticks (100 MHz) instead of the colour */
        c.x = (R)(double)(tl_start & 0xFFFFFFFFFFLL); c.y = (R)(double)(wall_clock64() & 0xFFFFFFFFFFLL); c.z = R(0);
#endif
        if (cnt.rep == 0) {
            cnt.c[0] = 1;
            F.rgb[3 * o] = (double)c.x; F.rgb[3 * o + 1] = (double)c.y; F.rgb[3 * o + 2] = (double)c.z;
            if (F.hit_id) F.hit_id[o] = hit_leaf;
        }                                             // (replicas: reflection / refraction rays are counted once, trace_pixel sees to it)
    }
    if (F.cost && unit >= 0 && lane == 0) {
        long long dt = (long long)wall_clock64() - t_start;
        atomicAdd(F.cost + unit, (uint32_t)(dt < 0 ? 0 : dt > 0x0FFFFFFFLL ? 0x0FFFFFFFLL : dt));
    }
    rtc::flush_counters(cnt, F, reinterpret_cast<unsigned int*>(s_stack));
}

// the fp64 variants, launched from the translation unit that holds them (rtc_trace_f64.hip); csg = 0, 1, 2 as the template's
// CSG; `grid` counts workgroups of RTC_BLOCK_PLAIN (csg = 0) / RTC_BLOCK_CSG threads
void launch_trace_f64(int csg, bool prof, bool far_cam, bool repl, bool two_waves, dim3 grid, hipStream_t stream, const DScene<double>& S,
                      const DCamera<double>& cam, const DFrame& F, double eps, double eps_rel);

}  // namespace rtc
