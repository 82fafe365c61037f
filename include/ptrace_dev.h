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
/* Per wave {start, queue dry (0 = never saw it dry), end} of the last counted launch, in 100 MHz
 * ticks.  Returns the number of waves written (at most cap_waves), < 0 when there is no log. */
long pt_debug_wave_log(pt_ctx* ctx, unsigned long long* out, size_t cap_waves);

#ifdef __cplusplus
}
#endif
#endif /* PTRACE_DEV_H */
