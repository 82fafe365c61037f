/*
 * ptrace_dev.h — developer diagnostics of libptrace.so.  NOT part of the versioned C ABI of
 * ptrace.h (PT_ABI_VERSION does not cover it, bindings do not list it): these entry points read the
 * raw tallies of the measuring twins (PT_OPT_COUNT_WORK) for tools/wave_log.py and may change or
 * disappear with the kernels.  A product integration needs none of them.
 */
#ifndef PTRACE_DEV_H
#define PTRACE_DEV_H

#include "ptrace.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Raw device counters (pt_kernel_args.h PT_CTR_*: queue head, segments, executed-work tallies, phase
 * clocks, segments per time bin).  Synchronises the stream.  Returns the number of values written
 * (at most cap), < 0 on error. */
long pt_debug_counters(pt_ctx* ctx, unsigned long long* out, size_t cap);
/* Per wave FOUR u64 of the last counted launch: {start, queue dry (0 = never saw it dry), end} in 100 MHz
 * ticks, then where the wave ran: HW_REG_HW_ID (wave slot [3:0], SIMD [5:4], CU [11:8], SH [12], SE [15:13])
 * | HW_REG_XCC_ID << 32.  `out` holds 4 * cap_waves values.  Returns the number of waves written (at most
 * cap_waves), < 0 when there is no log. */
long pt_debug_wave_log(pt_ctx* ctx, unsigned long long* out, size_t cap_waves);
/* The grid twins' gather statistics of the last counted launch (every 8th wave sampled): out[0 .. n_entries) = leaf-round
 * lanes per entry-run start (index into the grid's entry array), then 66 values: bins 0..64 = leaf rounds with that many
 * DISTINCT runs among their lanes, [65] = the sampled rounds' lanes.  Returns the number of values written (n_entries + 66
 * when cap allows), < 0 when there is none.  The histogram costs the twin several times its run time (atomics, a loop over the
 * distinct runs of every sampled round), so it is filled only at pt_set_option(ctx, PT_OPT_COUNT_WORK, 2); 1 is the plain twin. */
long pt_debug_cell_hist(pt_ctx* ctx, uint32_t* out, size_t cap);
/* Where the set-up calls of this context spent their time, host clock, milliseconds (bench.py's `first_frame`):
 * pt_create: the process's first HIP call (runtime start; ~0 in a process that has used HIP before), device selection
 * + properties, stream + counter allocations, the kernel attributes (the first of them loads the library's main code
 * object onto the device), accumulation / slab / texture buffers, total; the last pt_set_spheres: splitting the records,
 * hierarchy build, grid build, allocations + uploads, total; the last pt_reserve_passes.  Returns the number written. */
enum { PT_SETUP_CREATE_RUNTIME = 0, PT_SETUP_CREATE_DEVICE, PT_SETUP_CREATE_STREAM_ALLOCS, PT_SETUP_CREATE_CODE_OBJECT,
       PT_SETUP_CREATE_BUFFERS, PT_SETUP_CREATE_TOTAL, PT_SETUP_SPHERES_SPLIT, PT_SETUP_SPHERES_BVH_BUILD,
       PT_SETUP_SPHERES_GRID_BUILD, PT_SETUP_SPHERES_UPLOAD, PT_SETUP_SPHERES_TOTAL, PT_SETUP_RESERVE_TOTAL,
       /* inside PT_SETUP_CREATE_STREAM_ALLOCS: */ PT_SETUP_CREATE_STREAM, PT_SETUP_CREATE_FIRST_MALLOC, PT_SETUP_CREATE_FIRST_MEMSET, PT_SETUP_COUNT };
long pt_debug_setup_times(pt_ctx* ctx, double* out_ms, size_t cap);
/* Watchdog for the A/B tools (tools/ab_kernels.py, tools/sweep_knobs.py, ...): waits for everything
 * enqueued on the context's stream WITHOUT blocking in the driver — an event is recorded and polled
 * (hipEventQuery, 1 ms apart) until it has fired or timeout_ms have passed.  Returns 0 when the stream
 * is idle, 1 on timeout (the caller reports and exits non-zero: an experimental kernel that never
 * finishes must not hold a GPU box until gpurun's own limit, round 3), < 0 on a HIP error. */
int pt_debug_wait(pt_ctx* ctx, unsigned timeout_ms);
/* pt_build_grid (ptrace.h) in the layout the grid kernels use when a scene's entries do not fit the LDS and
 * are gathered from global memory: the cells' runs of entries in Morton order of their cells
 * (csrc/pt_grid.hpp morton_runs).  Same arguments, counts and return codes.  For tests/test_grid.py. */
int pt_build_grid_runs(const PtSphere* spheres, uint32_t n, uint32_t* counts8, float* geom12, float* margin4,
                       float* delta_g, uint32_t* cells, size_t n_cells, float* entries, size_t entry_floats,
                       uint32_t* entry_index, size_t n_index);

#ifdef __cplusplus
}
#endif
#endif /* PTRACE_DEV_H */
