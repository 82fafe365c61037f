// pt_extra.h — the kernels of pt_kernels_extra.hip and pt_kernels_small.hip (code objects of their own): how pt_api.hip reaches them.
#pragma once
enum {
  PT_X_BVH_COUNT = 0, PT_X_GRID_COUNT, PT_X_GRID_CELLS_COUNT, PT_X_SMALL_COUNT,
  PT_X_SMALL_RR /* + list length % 4: four builds */, PT_X_SCALAR_RR = PT_X_SMALL_RR + 4, PT_X_SCALAR_NOLDS_RR, PT_X_BVH_RR, PT_X_BVH_NODES_RR, PT_X_BVH_GMEM_RR,
  PT_X_GRID_RR, PT_X_GRID_CELLS_RR, PT_X_GRID_GMEM_RR, PT_X_COUNT
};
// the kernel's host-side handle (what hipLaunchKernel takes); asking for it loads nothing yet
extern "C" const void* pt_extra_kernel(int id);
// pt_kernels_small.hip (a code object of its own): the small-list kernel built for this list length's remainder modulo four
extern "C" const void* pt_small_kernel(unsigned n_spheres);
