// pt_kernels_extra.hip — the OPT-IN builds of the trace kernel, a translation unit (and therefore a gfx950
// code object) of their own: the Russian-roulette kernels (PT_OPT_RUSSIAN_ROULETTE, pt_shade.hpp) and the
// measuring twins (PT_OPT_COUNT_WORK: the same bodies with the executed-work tallies and the phase clock
// live).  The HIP runtime loads a code object when one of its kernels is first asked for, so a context that
// never turns these options on never pays for the sixteen kernels in here (round 3: one 557 KB code object
// with 22 instantiations of the body, loaded by every context's first launch).  pt_api.hip reaches them
// through pt_extra_kernel() only.
#include "pt_trace_body.hpp"
#include "pt_extra.h"

// measuring twins (PT_OPT_COUNT_WORK): the same walks with the executed-work tallies
extern "C" __global__ __launch_bounds__(1024) PT_BUILT_FOR(PT_WAVES_WALK) void pt_trace_kernel_bvh_count(const PtKernelArgs A) {
  pt_trace_body<false, false, 1, true>(A);
}
extern "C" __global__ __launch_bounds__(1024) PT_BUILT_FOR(PT_WAVES_WALK) void pt_trace_kernel_grid_count(const PtKernelArgs A) {
  pt_trace_body<false, false, 4, true>(A);
}
extern "C" __global__ __launch_bounds__(1024) PT_BUILT_FOR(PT_WAVES_TWIN_CELLS) void pt_trace_kernel_grid_cells_count(const PtKernelArgs A) {
  pt_trace_body<false, false, 5, true>(A);
}
// (the small-list kernel's twin: the phase clock of config 4 and State::default)
extern "C" __global__ __launch_bounds__(1024) PT_BUILT_FOR(PT_WAVES_SMALL) void pt_trace_kernel_small_count(const PtKernelArgs A) {
  pt_trace_body<false, true, 7, true>(A);
}

// Russian-roulette builds (PT_OPT_RUSSIAN_ROULETTE, opt-in; same launch shapes as their namesakes)
// (the small-list kernel: one build per list length modulo four, like pt_kernels_small.hip)
extern "C" __global__ __launch_bounds__(1024) PT_BUILT_FOR(PT_WAVES_SMALL) void pt_trace_kernel_small_t0_rr(const PtKernelArgs A) {
  pt_trace_body<false, true, 7, false, true, 0>(A);
}
extern "C" __global__ __launch_bounds__(1024) PT_BUILT_FOR(PT_WAVES_SMALL) void pt_trace_kernel_small_t1_rr(const PtKernelArgs A) {
  pt_trace_body<false, true, 7, false, true, 1>(A);
}
extern "C" __global__ __launch_bounds__(1024) PT_BUILT_FOR(PT_WAVES_SMALL) void pt_trace_kernel_small_t2_rr(const PtKernelArgs A) {
  pt_trace_body<false, true, 7, false, true, 2>(A);
}
extern "C" __global__ __launch_bounds__(1024) PT_BUILT_FOR(PT_WAVES_SMALL) void pt_trace_kernel_small_t3_rr(const PtKernelArgs A) {
  pt_trace_body<false, true, 7, false, true, 3>(A);
}
extern "C" __global__ __launch_bounds__(1024) PT_BUILT_FOR(PT_WAVES_LIST) void pt_trace_kernel_scalar_rr(const PtKernelArgs A) {
  pt_trace_body<false, true, 0, false, true>(A);
}
extern "C" __global__ __launch_bounds__(1024) PT_BUILT_FOR(PT_WAVES_LIST) void pt_trace_kernel_scalar_nolds_rr(const PtKernelArgs A) {
  pt_trace_body<false, false, 0, false, true>(A);
}
extern "C" __global__ __launch_bounds__(1024) PT_BUILT_FOR(PT_WAVES_WALK) void pt_trace_kernel_bvh_rr(const PtKernelArgs A) {
  pt_trace_body<false, false, 1, false, true>(A);
}
extern "C" __global__ __launch_bounds__(1024) PT_BUILT_FOR(PT_WAVES_WALK) void pt_trace_kernel_bvh_nodes_rr(const PtKernelArgs A) {
  pt_trace_body<false, false, 2, false, true>(A);
}
extern "C" __global__ __launch_bounds__(1024) PT_BUILT_FOR(PT_WAVES_WALK) void pt_trace_kernel_bvh_gmem_rr(const PtKernelArgs A) {
  pt_trace_body<false, false, 3, false, true>(A);
}
extern "C" __global__ __launch_bounds__(1024) PT_BUILT_FOR(PT_WAVES_WALK) void pt_trace_kernel_grid_rr(const PtKernelArgs A) {
  pt_trace_body<false, false, 4, false, true>(A);
}
extern "C" __global__ __launch_bounds__(1024) PT_BUILT_FOR(PT_WAVES_WALK) void pt_trace_kernel_grid_cells_rr(const PtKernelArgs A) {
  pt_trace_body<false, false, 5, false, true>(A);
}
extern "C" __global__ __launch_bounds__(1024) PT_BUILT_FOR(PT_WAVES_WALK) void pt_trace_kernel_grid_gmem_rr(const PtKernelArgs A) {
  pt_trace_body<false, false, 6, false, true>(A);
}


extern "C" const void* pt_extra_kernel(int id) {
  switch (id) {
    case PT_X_BVH_COUNT: return reinterpret_cast<const void*>(pt_trace_kernel_bvh_count);
    case PT_X_GRID_COUNT: return reinterpret_cast<const void*>(pt_trace_kernel_grid_count);
    case PT_X_GRID_CELLS_COUNT: return reinterpret_cast<const void*>(pt_trace_kernel_grid_cells_count);
    case PT_X_SMALL_COUNT: return reinterpret_cast<const void*>(pt_trace_kernel_small_count);
    case PT_X_SMALL_RR + 0: return reinterpret_cast<const void*>(pt_trace_kernel_small_t0_rr);
    case PT_X_SMALL_RR + 1: return reinterpret_cast<const void*>(pt_trace_kernel_small_t1_rr);
    case PT_X_SMALL_RR + 2: return reinterpret_cast<const void*>(pt_trace_kernel_small_t2_rr);
    case PT_X_SMALL_RR + 3: return reinterpret_cast<const void*>(pt_trace_kernel_small_t3_rr);
    case PT_X_SCALAR_RR: return reinterpret_cast<const void*>(pt_trace_kernel_scalar_rr);
    case PT_X_SCALAR_NOLDS_RR: return reinterpret_cast<const void*>(pt_trace_kernel_scalar_nolds_rr);
    case PT_X_BVH_RR: return reinterpret_cast<const void*>(pt_trace_kernel_bvh_rr);
    case PT_X_BVH_NODES_RR: return reinterpret_cast<const void*>(pt_trace_kernel_bvh_nodes_rr);
    case PT_X_BVH_GMEM_RR: return reinterpret_cast<const void*>(pt_trace_kernel_bvh_gmem_rr);
    case PT_X_GRID_RR: return reinterpret_cast<const void*>(pt_trace_kernel_grid_rr);
    case PT_X_GRID_CELLS_RR: return reinterpret_cast<const void*>(pt_trace_kernel_grid_cells_rr);
    case PT_X_GRID_GMEM_RR: return reinterpret_cast<const void*>(pt_trace_kernel_grid_gmem_rr);
    default: return nullptr;
  }
}
