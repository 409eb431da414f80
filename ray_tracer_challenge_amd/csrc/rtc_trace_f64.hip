// rtc_trace_f64.hip -- the fp64 trace kernel variants, in a translation unit of their own because they are
// compiled with `-mllvm -disable-machine-licm` (Makefile).  With machine LICM on, the fp64 constants of the inlined math
// (pow, acos, atan2 polynomials: two 32-bit moves each) are hoisted out of the ray loop into registers the loop has no room
// for, spilled in the kernel's first instructions by every wave -- sky waves included -- and re-read from scratch inside
// the loop: ~100 spilled VGPRs, ~70 scratch stores per wave, 1 GB of scratch written back to HBM per dragons frame.
// Without it the same kernels spill nothing (scratch = the 640-byte bounce stack): dragons 1080p 0.60 -> 0.53 ms, teapot
// 0.33 -> 0.25, earth 0.19 -> 0.14; frames bit-identical.  The fp32 variants are faster WITH it on the dragons (0.43 vs 0.47 ms,
// they have the registers) and stay in rtc_api.hip.  The CSG variants (CSG = 1, 2) joined this unit in round 3.
#include <hip/hip_runtime.h>

#include "../../include/rtc_api.h"
#include "rtc_trace_kernel.hpp"

namespace rtc {

void launch_trace_f64(int csg, bool prof, bool far_cam, bool repl, bool two_waves, dim3 grid, hipStream_t stream, const DScene<double>& S,
                      const DCamera<double>& cam, const DFrame& F, double eps, double eps_rel) {
    const dim3 block(csg ? RTC_BLOCK_CSG : RTC_BLOCK_PLAIN);
#define RTC_F64(PROF, CSG, FAR, REPL, W) hipLaunchKernelGGL((trace_kernel<double, PROF, CSG, FAR, REPL, W>), grid, block, 0, stream, S, cam, F, eps, eps_rel)
    // without CSG: two register budgets (launch() times them per view); with: one
#define RTC_F64_W(FAR, REPL) do { if (csg == 3) { if (two_waves) RTC_F64(false, 3, FAR, REPL, 2); else RTC_F64(false, 3, FAR, REPL, RTC_WAVES_F64); } \
    else if (two_waves) RTC_F64(false, 0, FAR, REPL, 2); else RTC_F64(false, 0, FAR, REPL, RTC_WAVES_F64); } while (0)
#define RTC_F64_PROF(FAR) do { if (csg == 3) RTC_F64(true, 3, FAR, false, RTC_WAVES_F64); else RTC_F64(true, 0, FAR, false, RTC_WAVES_F64); } while (0)
#define RTC_F64_CSG(CSG, FAR) do { if (prof) RTC_F64(true, CSG, FAR, false, RTC_WAVES_OF(double, CSG)); else if (repl) RTC_F64(false, CSG, FAR, true, RTC_WAVES_OF(double, CSG)); \
    else RTC_F64(false, CSG, FAR, false, RTC_WAVES_OF(double, CSG)); } while (0)
#ifdef RTC_DEV_PLAIN_ONLY   /* development builds: no far-camera variants, no CSG */
    (void)far_cam;
    if (prof) RTC_F64_PROF(false);
    else if (repl) RTC_F64_W(false, true);
    else RTC_F64_W(false, false);
#else
    if (csg == 1) { if (far_cam) RTC_F64_CSG(1, true); else RTC_F64_CSG(1, false); }
    else if (csg == 2) { if (far_cam) RTC_F64_CSG(2, true); else RTC_F64_CSG(2, false); }
    else if (prof) { if (far_cam) RTC_F64_PROF(true); else RTC_F64_PROF(false); }
    else if (far_cam) { if (repl) RTC_F64_W(true, true); else RTC_F64_W(true, false); }
    else { if (repl) RTC_F64_W(false, true); else RTC_F64_W(false, false); }
#endif
#undef RTC_F64_PROF
#undef RTC_F64_CSG
#undef RTC_F64_W
#undef RTC_F64
}

}  // namespace rtc
