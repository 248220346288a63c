/*
 * sfm_gpu_debug.h — test hooks of libsfm_gpu.  None of these has a counterpart
 * in the reference; they exist so that tests can pin hardware assumptions and
 * exercise every kernel shape.
 */
#ifndef SFM_GPU_DEBUG_H
#define SFM_GPU_DEBUG_H

#include "sfm_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Forces the shape of the distance kernel (-1: planner decides): 0..3 = 32x32x32
 * MFMAs with {4x4, 2x4, 2x8, 4x8} {query blocks per wave x waves}, 4 = 16x16x64
 * MFMAs.  Every shape gives the same (exact) results.  The timing build of the
 * library (libsfm_gpu_timing.so, never the one a host links) also takes
 * shape + 16*a, which switches parts of the kernel off for timing and gives WRONG
 * results: a = bit 0 no digest of the accumulators | bit 1 no DMA of further
 * stages | bit 2 no stage barrier; the default build answers
 * SFM_GPU_ERR_UNSUPPORTED. */
int sfm_gpu_debug_set_knn_variant(sfm_gpu_ctx* ctx, int variant);

/* Enqueues a kernel that overwrites the LDS of every CU with a pattern.  A kernel
 * that misses a DMA, or reads a stage before it has landed, otherwise reads what
 * the previous kernel left in LDS, which in a test is often the right data. */
int sfm_gpu_debug_poison_lds(sfm_gpu_ctx* ctx, uint32_t pattern);

/* The first `bytes` of the sweep records the last k-NN call on this context left
 * ([train part][padded query row][4] x {best group, second group, third group,
 * first unit of the part}; a group is (x << 9 | unit within the part)). */
int sfm_gpu_debug_copy_records(sfm_gpu_ctx* ctx, void* out, size_t bytes);

/* Timing build only (SFM_GPU_ERR_UNSUPPORTED otherwise): per workgroup of the last
 * 16x16 sweep launch, 8 words: wall clock at entry (100 MHz ticks), shader cycle
 * counter at entry, after the prologue, after the last stage, at exit, wall clock
 * at exit, XCC id, unused.  In the default build the kernel carries no stamp. */
int sfm_gpu_debug_sweep_stamps(sfm_gpu_ctx* ctx, uint64_t* out, int n_wg);

/* Number of queries that the last k-NN call on this context (its first job)
 * sent through the exhaustive float-ranked pass. */
int sfm_gpu_debug_redo_count(sfm_gpu_ctx* ctx, int32_t* out);

/* Timing build only (SFM_GPU_ERR_UNSUPPORTED otherwise): per-wavefront stamps of
 * the scoring launches since the last call, four 64-bit words each -- realtime
 * counter (100 MHz) at entry and at exit, hardware id | XCC id << 32, the
 * launch's stage tag -- at most n_max of them; *count = how many were copied. */
int sfm_gpu_debug_score_stamps(sfm_gpu_ctx* ctx, uint64_t* out, int n_max, int32_t* count);

/* c[32][32] (int32) = a[32][32] * b^T with b given as [32 columns][32 k], all
 * host pointers, through one v_mfma_i32_32x32x32_i8 loaded the way the
 * distance kernel loads its operands. */
int sfm_gpu_debug_mfma_probe(sfm_gpu_ctx* ctx, const int8_t* a, const int8_t* b, int32_t* c);

/* c[16][16] (int32) = a[16][64] * b^T with b given as [16 columns][64 k], through
 * one v_mfma_i32_16x16x64_i8 loaded the way the 16x16 form of the distance kernel
 * loads its operands. */
int sfm_gpu_debug_mfma16_probe(sfm_gpu_ctx* ctx, const int8_t* a, const int8_t* b, int32_t* c);

/* Per-hypothesis data of the last sfm_gpu_ransac call on this context: the
 * sample table [n_hyp][m], the number of candidate models per hypothesis, the
 * inlier counts [n_hyp][10] (-1 = no such model) and the models
 * [n_hyp][10][9].  Host pointers; any may be NULL. */
int sfm_gpu_debug_ransac_dump(sfm_gpu_ctx* ctx, int n_hyp, int32_t* table, int32_t* nmodels,
                              int32_t* counts, double* models);

/* Counts the integers n < 2^24 for which the device's plain sqrtf((float)n)
 * differs from the correctly rounded value the match records carry. */
int sfm_gpu_debug_sqrt_probe(sfm_gpu_ctx* ctx, int32_t* mismatches);

/* Per context.  0: the RANSAC scoring kernel evaluates every (model, point) pair
 * in double precision only; 1 (default): a float32 filter with a rigorous error
 * bound settles the clear cases first.  Both give identical counts; the switch
 * lets tests prove it.  (The timing build also takes 2 and 3, experiments with
 * WRONG counts: 2 drops the tests the filter cannot settle, 3 also the point
 * loads inside the loop; SFM_GPU_ERR_UNSUPPORTED in the default build.) */
int sfm_gpu_debug_set_score_filter(sfm_gpu_ctx* ctx, int on);

/* Per context.  Fixed-count RANSAC scores its candidate models in stages over
 * the points and, between stages, drops the models that provably cannot reach
 * the best count (their entries in the dumped counts stay partial, never above
 * the true count).  0 switches the pruning off so that every dumped count is
 * complete; 1 (default) on: for a batch of eight problems or more the list is
 * ordered by the counts of an opening stretch and stages follow over which whole
 * wavefronts of models that cannot reach the bound any more stop at once (4: the
 * same for any batch); smaller batches, and 2 and 3 always, take two stages
 * sized from that stretch with the list compacted between them (two cuts, 3: one).
 * The selected model, its inlier count and the mask are the same
 * in every mode. */
int sfm_gpu_debug_set_score_prune(sfm_gpu_ctx* ctx, int on);

/* Per context.  The seven-point elimination exists with one lane per hypothesis
 * and with two lanes sharing a hypothesis (half the LDS per wavefront; same
 * models bit for bit): -1 (default) picks by batch size, 0 / 1 force one / two. */
int sfm_gpu_debug_set_solver_split(sfm_gpu_ctx* ctx, int mode);

/* Environment, read once per process: SFM_GPU_ROCTX=1 brackets every stage of a
 * call (the names sfm_gpu_profile_query knows: "knn2_mfma", "knn_finalize",
 * "ratio_compact", "ransac_solve", "ransac_score", ...) in a rocTX range
 * (roctxRangePush / Pop from librocprofiler-sdk-roctx.so or libroctx64.so, loaded
 * on demand), so that `rocprofv3 --marker-trace --kernel-trace` shows which launches
 * belong to which stage.  Without the variable, or without the library, nothing is
 * loaded and nothing is called. */

/* Vote counts of the four pose candidates in the last sfm_gpu_recover_pose call
 * on this context, in the order (R1,t) (R2,t) (R1,-t) (R2,-t). */
int sfm_gpu_debug_pose_votes(sfm_gpu_ctx* ctx, int32_t* votes4);

#ifdef __cplusplus
}
#endif
#endif
