// pt_kernels_small.hip — the small-list kernels (PT_GEOM_SMALL: lists of at most 16 spheres, the reference's own
// scene size: `uniform Sphere[15] u_sphere_list`, static/shader.frag:103), a translation unit and gfx950 code object
// of their own: a context whose scene has more than 16 spheres never loads it, one whose scene is the reference's
// never loads the walk kernels' share of a cold start twice over.
//
// No LDS in the scan, no candidate queue: the list reaches the VALU group by group from SGPRs (pt_list.hpp
// small_scan).  FOUR builds, one per list length modulo four: the last group of a list of 4 q + r spheres tests
// exactly r of them, so the padding entries are never looked at and nothing is masked — the reference's nine
// spheres (State::default; BASELINE config 4 has nine too) pay one test in their last group instead of four:
// config 4 -6.2 %, State::default 16 x 25 spp -5.6 %, its 1-spp frame -2.7 % against the build that tests the
// padding (A/B on one device, profiles/r04_ab_runs.txt).
#include "pt_trace_body.hpp"
#include "pt_extra.h"

extern "C" __global__ __launch_bounds__(1024) PT_BUILT_FOR(PT_WAVES_SMALL) void pt_trace_kernel_small_t0(const PtKernelArgs A) {
  pt_trace_body<false, true, 7, false, false, 0>(A);
}
extern "C" __global__ __launch_bounds__(1024) PT_BUILT_FOR(PT_WAVES_SMALL) void pt_trace_kernel_small_t1(const PtKernelArgs A) {
  pt_trace_body<false, true, 7, false, false, 1>(A);
}
extern "C" __global__ __launch_bounds__(1024) PT_BUILT_FOR(PT_WAVES_SMALL) void pt_trace_kernel_small_t2(const PtKernelArgs A) {
  pt_trace_body<false, true, 7, false, false, 2>(A);
}
extern "C" __global__ __launch_bounds__(1024) PT_BUILT_FOR(PT_WAVES_SMALL) void pt_trace_kernel_small_t3(const PtKernelArgs A) {
  pt_trace_body<false, true, 7, false, false, 3>(A);
}

extern "C" const void* pt_small_kernel(unsigned n_spheres) {
  switch (n_spheres & 3u) {
    case 0: return reinterpret_cast<const void*>(pt_trace_kernel_small_t0);
    case 1: return reinterpret_cast<const void*>(pt_trace_kernel_small_t1);
    case 2: return reinterpret_cast<const void*>(pt_trace_kernel_small_t2);
    default: return reinterpret_cast<const void*>(pt_trace_kernel_small_t3);
  }
}
