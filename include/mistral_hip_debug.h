/*
 * mistral_hip_debug.h -- diagnostics of libmistral_hip.so's persistent decode engine.
 *
 * NOT part of the drop-in boundary (include/mistral_hip.h): nothing a caller of the hot path needs, no reference
 * counterpart.  Used by scripts/engine_trace.py (phase timeline), the A/B scripts (loader knobs) and
 * tests/test_gpu_greedy.py (residency-gate sabotage).  Results of mi_forward never depend on any of these.
 */
#ifndef MISTRAL_HIP_DEBUG_H
#define MISTRAL_HIP_DEBUG_H

#include "mistral_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Debug timeline of the engine (scripts/engine_trace.py): while a zero-filled device buffer of
 * mi_debug_engine_trace_bytes() bytes is registered, consumer wave 0 and the loader wave of every workgroup stamp a
 * 100 MHz clock at each phase boundary of each layer: trace[cu][layer (32)][event (26)] uint64.  NULL unregisters. */
size_t mi_debug_engine_trace_bytes(void);
int mi_debug_set_engine_trace(void* dev_buffer);
/* Tuning knobs of the engine's loader wave (results never depend on them): while the workgroup's consumers
 * sweep hand-off granules the loader wave stops (thin = 2, shipped), keeps one 16 KiB fill outstanding (1) or streams on
 * (0); depth = fills in flight otherwise (2 or 3).
 * Initial values: MI_ENGINE_THIN / MI_ENGINE_DEPTH, else the shipped defaults.  MI_ENGINE_HOLDERS=0 (environment, read
 * once) runs the engine without its holder waves. */
int mi_debug_set_engine_knobs(int thin, int depth);
/* holder waves on (1) / off (0) / environment default (-1); results never depend on it (bit-identical either way) */
int mi_debug_set_engine_holders(int on);
/* The engine source is compiled four times: the default build (frozen since round 3: dense models), the `next` build (round 5:
 * the dense GQA-4 shapes whose rows are multiples of 4 pieces, i.e. the headline model), a "wide" build for the shapes those
 * decline (GQA ratio 6 with a 32 KiB hid vector; rows of an even number of pieces that is not a multiple of 4) and a MoE build
 * (Mixtral-8x7B shapes).  0 (default; environment MI_ENGINE_VARIANT): dense models the `next` build where it applies, else the
 * default build; MoE models the MoE build, else the wide build; 1: the wide build wherever it applies (tests compare its code
 * paths with the launch path at small sizes; also admits shapes that measured slower than the launch path); 2: the default
 * (frozen) build first for every model - the A/B partner of `next`, and the build that carries the timeline stamp sites.
 * Returns the previous setting. */
int mi_debug_set_engine_variant(int variant);
/* Test hook: the next `launches` engine launches on this workspace (hipGraph replays included: the count lives in the
 * workspace) wait for one workgroup more than exist, i.e. fail their residency gate after its ~50 ms bound exactly as a
 * launch with a missing workgroup would (status 0x700, nothing written).  Synchronises the stream. */
int mi_debug_engine_sabotage(void* workspace, int launches, mi_stream_t stream);
/* Same-process A/B of the prefill kernels' alternative forms (negative = keep the current setting): attn_waves 0 =
 * automatic, 4 | 8 = pin the attention block shape (MI_ATTN_PREFILL_WAVES); gemm_tail 0 = no tail split, 1 = the last
 * partly filled round of a 256x256-tile GEMM on the 128x128 kernel, 2 = as 128x256 tiles of the 256 kernel (default,
 * bit-identical to 0; MI_GEMM_TAIL). */
int mi_debug_set_prefill_kernels(int attn_waves, int gemm_tail);

#ifdef __cplusplus
}
#endif
#endif /* MISTRAL_HIP_DEBUG_H */
