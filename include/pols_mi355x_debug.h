/* pols_mi355x_debug.h -- measurement aids of libpols_mi355x.so: NOT part of the reference interface (include/pols_mi355x.h is).
 * bench.py's roofline leg, scripts/ and the tests use them; a plugin built on the C-ABI never needs this header. */
#ifndef POLS_MI355X_DEBUG_H
#define POLS_MI355X_DEBUG_H

#include "pols_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Kernel timing with HIP events on the context's stream (used by bench.py's roofline leg).
 * While enabled every compute entry brackets its dominant kernel with an event pair; enable = n > 1 times every n-th
 * call only (an event pair costs ~5 us on the stream's timeline, which matters next to a 75 us kernel). */
int pols_timing_enable(pols_ctx *ctx, int enable);
/* Synchronises, copies up to `max` per-launch durations (ms) recorded since the last call, returns the count. */
int pols_timing_collect(pols_ctx *ctx, float *ms_out, int max);
/* Name of the kernel variant the last compute entry launched (for profiles / DESIGN.md). */
const char *pols_last_kernel_name(pols_ctx *ctx);

/* Measurement aid, NOT part of the reference interface (bench.py's `roofline.stream_ceiling`): one pass over the batch's columns with
 * the arithmetic removed -- every feature column, the target and the weights read with 16-byte streaming loads down the row axis,
 * their sum written over `pred_out` (n_rows values, batch dtype) with streaming stores: the rate HBM admits for this traffic mix on
 * this device, beside which a static kernel's achieved rate on the same buffers is read.  DEVICE batches, up to POLS_MAX_FEATURES
 * columns; timed like every launch (pols_timing_enable / pols_timing_collect). */
int pols_stream_probe(pols_ctx *ctx, const pols_batch *b, void *pred_out);
/* The same with a choice of launch shape.  mode 0: one workgroup per 256-lane piece, one piece per lane, exits (the launch shape of the
 * resident static kernels: "the kernel without its arithmetic").  mode 1: a PERSISTENT grid (4 workgroups per CU) striding over the
 * pieces with the next piece's loads issued before the current one is stored -- no dispatch ramp, no tail, 18 x 16-byte loads in
 * flight per lane: what the memory system admits for this traffic mix when nothing else is in the way.  (Falls back to mode 0 when
 * n_rows is not a whole number of pieces.) */
int pols_stream_probe_ex(pols_ctx *ctx, const pols_batch *b, void *pred_out, int mode);

#ifdef __cplusplus
}
#endif
#endif /* POLS_MI355X_DEBUG_H */
