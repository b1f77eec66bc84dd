/* Unit-test and profiling hooks of the MI355X NVSM / LSE engine — libcunvsm_amd_testhooks.so.
 *
 * NOT part of the drop-in surface (include/cunvsm_amd.h) and not exported by libcunvsm_amd.so: tests/, bench.py's --gate-us
 * profiling aid and tools/exp/ load this library behind the product library (cunvsm_amd/_lib.py does it on first use of a hook);
 * it calls the product's own kernel launchers and carries no kernels of its own. */
#ifndef CUNVSM_AMD_TEST_HOOKS_H
#define CUNVSM_AMD_TEST_HOOKS_H

#include "cunvsm_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Debug / unit-test hooks for individual kernels (tests only; not part of the drop-in surface). */
int nvsm_debug_gemm(int variant, int M, int N, int K, const float* hostA, const float* hostB, float* hostC);
/* Queues a kernel on the handle's stream that spins for `microseconds` of GPU wall clock: profiling runs put it in
 * front of a step so that the host has queued the whole step before the GPU starts it (tools/rocprof_summary.py timeline). */
int nvsm_debug_delay(nvsm_model* m, int microseconds);
/* table passes of the update in one launch (1, the default) or as the three launches chunk / level-2 / rows (0): the two
 * forms are bit-identical (tests/test_gpu_parity.py); process-wide */
int nvsm_debug_set_table_pass_form(int one_launch);
/* the stable (row, entry) radix sort alone: keys of `bits` significant bits in, sorted keys + their original positions out */
/* average ms per launch of a batch-sized projection product on device operands (extras: 1 = column statistics, 2 = row sums of squares) */
int nvsm_debug_gemm_time(int b_layout, int M, int N, int K, int extras, int repeats, float* avg_ms);
/* the projection-gradient product alone (which 0 = the split-bf16 split-K kernel, 1 = its wave-sized form gemm_dtw.hip, 2 = tiled exact-fp32 kernel): average ms of the product and of its slab reduce */
int nvsm_debug_dt_time(int M, int N, int rows, int slabs, int repeats, int which, float* kernel_ms, float* reduce_ms);
int nvsm_debug_sort(int64_t n, int bits, const int32_t* keys, int32_t* keys_out, int32_t* vals_out, int repeats, float* avg_ms);
int nvsm_debug_gather_mean(int64_t num_rows, int dim, const float* table, const int64_t* idx, const float* wts,
                           int window, int64_t num_out, float* out);

#ifdef __cplusplus
}
#endif
#endif /* CUNVSM_AMD_TEST_HOOKS_H */
