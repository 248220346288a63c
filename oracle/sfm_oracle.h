/*
 * CPU ORACLE — TEST INFRASTRUCTURE ONLY.
 *
 * A dependency-free CPU restatement of the semantics behind the two library
 * calls on the reference's matching path:
 *     matcher->knnMatch(descL, descR, matches, 2)      /root/reference/main.cpp:184
 *     ratio test in double                              /root/reference/main.cpp:189-197
 *     findEssentialMat(..., FM_RANSAC, 0.99, 1.0, mask) /root/reference/main.cpp:224
 *     recoverPose(E, ptsL, ptsR, K, R, t)               /root/reference/main.cpp:420
 *
 * The arithmetic of those calls lives in OpenCV 4.5.5 (pinned only by
 * /root/reference/README.md:5), which is NOT vendored in the reference and is
 * not installed here.  The reference holds no tests, fixtures or golden
 * vectors for this path, so:  PARITY UNPINNED.  This file restates the
 * published algorithms (brute-force L2 k-NN with OpenCV's insertion order,
 * cv::RNG, the RANSAC point-set registrator loop, Nister's five-point
 * solver, Sampson error) and is pinned only by known-answer tests that can be
 * derived without OpenCV (tests/test_oracle_kat.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The shipped GPU path never links or calls it.
 */
#ifndef SFM_ORACLE_H
#define SFM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Layout-identical to cv::DMatch (queryIdx, trainIdx, imgIdx, distance). */
typedef struct {
    int32_t queryIdx;
    int32_t trainIdx;
    int32_t imgIdx;
    float distance;
} oracle_dmatch;

/* Brute-force k-NN, NORM_L2, float32 rows with byte strides.  Writes
 * nq*k records (unused slots: trainIdx=-1) and nq neighbour counts
 * (=min(k,nt)).  Returns 0, or -1 on bad arguments. */
int oracle_knn_match(const float* query, int nq, size_t qstep_bytes,
                     const float* train, int nt, size_t tstep_bytes,
                     int dim, int k, oracle_dmatch* out, int32_t* counts,
                     int nthreads);

/* The same result, byte for byte, by integer arithmetic when every input value
 * is an integer in 0..255 (cv::SIFT descriptors): the variant bench.py times as
 * the CPU baseline.  Other inputs fall through to oracle_knn_match. */
int oracle_knn_match_u8path(const float* query, int nq, size_t qstep_bytes,
                            const float* train, int nt, size_t tstep_bytes,
                            int dim, int k, oracle_dmatch* out, int32_t* counts,
                            int nthreads);

/* Ratio test of main.cpp:189-197: keep row r iff
 * (double)m[r][0].distance < ratio * (double)m[r][1].distance.  Rows with
 * fewer than 2 neighbours are skipped (the reference reads m[r][1]
 * unconditionally: undefined behaviour there).  Returns survivor count and
 * writes survivor row numbers ascending. */
int oracle_ratio_test(const oracle_dmatch* matches, const int32_t* counts,
                      int nq, int k, double ratio, int32_t* keep_rows);

/* cv::RNG stream from a given 64-bit seed. */
void oracle_rng_next(uint64_t seed, int n, uint32_t* out);
void oracle_rng_uniform(uint64_t seed, int a, int b, int n, int32_t* out);

/* Sample table of the RANSAC loop: n_iters subsets of m distinct indices in
 * [0,count) drawn exactly as RANSACPointSetRegistrator::getSubset does with
 * RNG(-1) (no subset check: the essential-matrix callback accepts every
 * subset).  Returns 0, -1 if count < m. */
int oracle_sample_table(int count, int m, int n_iters, int32_t* out);

/* Same with the fundamental-matrix subset check (collinearity in either
 * image, float coordinates); pts are n x 2 float32 pixel coordinates.
 * Returns number of subsets produced (may stop early when 10000 attempts
 * fail). */
int oracle_sample_table_f(int count, const float* pts1, const float* pts2,
                          int n_iters, int32_t* out);

/* Sample table of the fixed-count mode: n_rows independent rows from the
 * counter-based generator (no OpenCV counterpart).  pts1/pts2 (float32 pixels)
 * enable the fundamental-matrix collinearity redraw; NULL for the essential
 * model.  A row that cannot be drawn has out[row][0] = -1. */
int oracle_sample_table_fixed(int count, int m, int n_rows, const float* pts1, const float* pts2,
                              int32_t* out);

int oracle_ransac_update_num_iters(double p, double ep, int model_points,
                                   int max_iters);

/* All real roots (ascending) of coef[0] + coef[1] z + ... + coef[degree] z^degree,
 * degree 1..10: the root finder used by the two minimal solvers.  Returns the
 * number of roots, -1 on a bad degree. */
int oracle_real_roots(const double* coef, int degree, double* roots);

/* Five-point solver on 5 normalised correspondences (x1: 5x2, x2: 5x2).
 * Writes up to 10 row-major 3x3 models with unit Frobenius norm, in
 * ascending order of the real root z.  Returns the model count. */
int oracle_five_point(const double* x1, const double* x2, double* models);

/* Seven-point solver on 7 float32 pixel correspondences.  Up to 3 models,
 * F33 = 1 when possible.  Returns the model count. */
int oracle_seven_point(const float* x1, const float* x2, double* models);

/* Sampson error of model E at n normalised correspondences, double
 * arithmetic cast to float (no fused multiply-add). */
void oracle_sampson_errors(const double* E, const double* x1, const double* x2,
                           int n, float* err);

/* Fundamental-matrix error: max of the two squared point-to-epipolar-line
 * distances, float32 pixel points, double arithmetic cast to float. */
void oracle_symepi_errors(const double* F, const float* x1, const float* x2,
                          int n, float* err);

/* Normalisation done by findEssentialMat before RANSAC:
 * x=(u-cx)/fx, y=(v-cy)/fy ; returns threshold/((fx+fy)/2). */
double oracle_normalize_points(const double* pts_px, int n, const double* K,
                               double threshold_px, double* out);

/*
 * RANSAC essential matrix (model_kind=0, m=5, Sampson) or fundamental matrix
 * (model_kind=1, m=7, symmetric epipolar).
 *   pts1/pts2     n x 2 double pixel coordinates
 *   K             3x3 row-major (ignored for model_kind=1)
 *   fixed_count   0: OpenCV-adaptive loop (max_iters cap, niters shrinks);
 *                 >0: evaluate exactly that many hypotheses, no early stop
 *   sample_table  optional fixed_count*m (or max_iters*m) indices; NULL =>
 *                 adaptive: generated from RNG(-1) like OpenCV; fixed count:
 *                 the counter-based table of oracle_sample_table_fixed
 *   mask          n bytes; untouched when the call fails (n < m)
 *   hyp_counts    optional [n_hyp][10] inlier counts (-1 = no such model),
 *                 filled for the hypotheses actually evaluated
 * Returns 1 on success, 0 on failure (OpenCV's `false`), -1 bad arguments.
 */
int oracle_ransac(int model_kind, const double* pts1, const double* pts2, int n,
                  const double* K, double prob, double threshold_px,
                  int max_iters, int fixed_count, const int32_t* sample_table,
                  uint8_t* mask, double* model_out, int32_t* n_inliers,
                  int32_t* iters_used, int32_t* hyp_counts);

/* oracle_ransac in fixed-count mode with the hypotheses spread over `nthreads`
 * threads; same winner, mask, model and count as the sequential loop (own
 * counter-based sample table only).  Used as the timed CPU baseline. */
int oracle_ransac_fixed_mt(int model_kind, const double* pts1, const double* pts2, int n,
                           const double* K, double threshold_px, int fixed_count, int nthreads,
                           uint8_t* mask, double* model_out, int32_t* n_inliers);

/* With n == m points cv::findEssentialMat / findFundamentalMat return every
 * candidate of the one direct solve stacked (3k x 3); oracle_ransac reports the
 * first in model_out, this returns all k of the calling thread's last such call
 * (capacity in models; -1 when it is too small). */
int oracle_ransac_direct_models(double* models_out, int capacity);

/* ---- resection: cv::solvePnPRansac(points_3d, pts, K, zeros, r, t, false,
 * SOLVEPNP_EPNP), /root/reference/bundle_adjustment.h:139-155 (pnp_oracle.cpp) ---- */

/* EPnP pose of n >= 4 correspondences (object points n x 3, pixel coordinates
 * n x 2, all double): x_cam = R x_world + t.  1 ok, 0 degenerate, -1 bad n. */
int oracle_epnp(const double* object_pts, const double* image_pts, int n, const double* K,
                double* R, double* t);

/* Rotation vector (axis * angle) of a rotation matrix. */
int oracle_rotation_vector(const double* R, double* rvec);

/* The whole call.  iterations_count 100, reprojection_error 8.0, confidence 0.99
 * are OpenCV's defaults, which the reference takes.  mask: n bytes (0/1), left
 * untouched on failure.  Returns 1 ok, 0 no pose, -1 bad arguments (n < 5: four
 * points go to P3P in OpenCV, which is not restated). */
int oracle_pnp_ransac(const double* object_pts, const double* image_pts, int n, const double* K,
                      int iterations_count, double reprojection_error, double confidence,
                      double* rvec, double* tvec, double* R_out, uint8_t* mask,
                      int32_t* n_inliers, int32_t* iters_used);

/* 3x3 singular value decomposition A = U diag(S) V^T by one-sided Jacobi
 * rotations (row-major, |S| descending, U[:,2] = U[:,0] x U[:,1] with S[2]
 * carrying the sign).  Returns 1, or 0 when A has rank < 2. */
int oracle_svd3(const double* A, double* U, double* S, double* V);

/* cv::decomposeEssentialMat: the two rotations and the unit translation. */
int oracle_decompose_essential(const double* E, double* R1, double* R2, double* t);

/*
 * cv::recoverPose(E, pts1, pts2, K, R, t, distanceThresh, mask)
 * (/root/reference/main.cpp:420, /root/reference/bundle_adjustment.h:80 with
 * the default distanceThresh = 50 and no mask).  pts are n x 2 double pixel
 * coordinates.  mask (optional, n bytes) is in/out: rows with mask 0 do not
 * vote; on return it marks the points in front of both cameras of the chosen
 * pose.  votes (optional) receives the four candidates' counts in OpenCV's
 * order (R1,t) (R2,t) (R1,-t) (R2,-t).  Returns the winning count, -1 on bad
 * arguments or a rank-deficient E.
 */
int oracle_recover_pose(const double* E, const double* pts1, const double* pts2, int n,
                        const double* K, double distance_thresh, double* R_out, double* t_out,
                        uint8_t* mask, int32_t* votes);

/*
 * The three-view join of /root/reference/main.cpp:442-456: for the inlier
 * matches ab of image pair (i, i+1) and bc of pair (i+1, j), every (m, n) with
 * ab[m].trainIdx == bc[n].queryIdx, in the order of the reference's nested loop
 * (m outer, n inner).  out receives (ab[m].queryIdx, ab[m].trainIdx,
 * bc[n].trainIdx) triples: the keypoint of one track in images i, i+1 and j.
 * Returns the number of triples (which may exceed capacity; only `capacity`
 * are written), -1 on bad arguments.  O(n_ab * n_bc) like the reference.
 */
int oracle_track_join(const oracle_dmatch* ab, int n_ab, const oracle_dmatch* bc, int n_bc, int32_t* out,
                      int capacity);

#ifdef __cplusplus
}
#endif
#endif
