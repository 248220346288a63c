/*
 * sfm_gpu.h — C ABI of the MI355X (gfx950) matching path.
 *
 * The reference (Suxilan/SFM) has no plugin or FFI layer; its matching hot
 * path is two C++ library calls inside match() (/root/reference/main.cpp:159).
 * Each entry point below names the reference line(s) whose work it replaces.
 * All pointers are plain host or device addresses, all sizes are plain
 * integers, nothing throws: every call returns SFM_GPU_OK (0) or a negative
 * SFM_GPU_ERR_* code, and sfm_gpu_last_error_string() explains the last
 * failure on that context.  The caller owns every buffer it passes in; the
 * library keeps no reference to caller memory after a call returns.
 *
 * Threading: a context is bound to one GPU and one HIP stream and must be used
 * by one host thread at a time.  Calls taking host pointers are synchronous
 * (they return when the outputs are written), like the calls they replace.
 */
#ifndef SFM_GPU_H
#define SFM_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SFM_GPU_OK 0
#define SFM_GPU_ERR_ARG (-1)          /* null pointer, negative size, dim != 128, ...   */
#define SFM_GPU_ERR_HIP (-2)          /* a HIP runtime call failed                       */
#define SFM_GPU_ERR_NOT_U8 (-3)       /* a descriptor value is not an integer in 0..255  */
#define SFM_GPU_ERR_TOO_MANY (-4)     /* n_train >= 2^18 (cv::BFMatcher asserts the same)*/
#define SFM_GPU_ERR_UNSUPPORTED (-5)  /* k > 2, unknown dtype / model / mode             */
#define SFM_GPU_ERR_NOMEM (-6)
#define SFM_GPU_ERR_CAPACITY (-7)     /* a size exceeds a documented fixed maximum       */

/* descriptor element types */
#define SFM_GPU_DESC_F32 0 /* float32 holding integers 0..255 (cv::SIFT output, CV_32F) */
#define SFM_GPU_DESC_U8 1  /* uint8 (CV_8U)                                             */

/* RANSAC models */
#define SFM_GPU_MODEL_ESSENTIAL 0   /* 5-point, Sampson error (cv::findEssentialMat)          */
#define SFM_GPU_MODEL_FUNDAMENTAL 1 /* 7-point, symmetric epipolar (cv::findFundamentalMat)   */

/* RANSAC modes */
#define SFM_GPU_RANSAC_ADAPTIVE 0 /* OpenCV loop: max_iters cap, iteration count shrinks   */
#define SFM_GPU_RANSAC_FIXED 1    /* evaluate exactly n_hyp hypotheses, no early stop      */

/* Largest n_hyp (fixed mode) / max_iters (adaptive mode) a call accepts;
 * SFM_GPU_ERR_CAPACITY beyond. */
#define SFM_GPU_MAX_HYPOTHESES (1 << 20)

/* Layout-identical to cv::DMatch: the output rows of knnMatch can be
 * memcpy'd into std::vector<cv::DMatch>. */
typedef struct sfm_gpu_dmatch {
    int32_t queryIdx;
    int32_t trainIdx;
    int32_t imgIdx; /* always 0: one train image per call, as at main.cpp:184 */
    float distance;
} sfm_gpu_dmatch;

typedef struct sfm_gpu_ctx sfm_gpu_ctx;
typedef struct sfm_gpu_image sfm_gpu_image;

/* ---- context --------------------------------------------------------- */

/* Creates a context on GPU `device_id`.  `hip_stream` is a hipStream_t to
 * launch on (NULL: the library creates its own stream).  Replaces nothing in
 * the reference (which has no device); it stands where main.cpp:181 builds a
 * matcher object per pair. */
int sfm_gpu_create(int device_id, void* hip_stream, sfm_gpu_ctx** out);
void sfm_gpu_destroy(sfm_gpu_ctx* ctx);
const char* sfm_gpu_last_error_string(const sfm_gpu_ctx* ctx);
/* Waits for everything queued on the context's stream. */
int sfm_gpu_synchronize(sfm_gpu_ctx* ctx);
/* Library build identification, e.g. "sfm_gpu 0.1 gfx950". */
const char* sfm_gpu_version(void);

/* ---- k-NN matching: replaces main.cpp:181-184 ------------------------- */

/*
 * matcher->knnMatch(descriptorsL, descriptorsR, matches, 2)   main.cpp:184
 * with cv::BFMatcher(NORM_L2) semantics: for each query row the k nearest
 * train rows by L2 distance, ascending by (distance as float32, trainIdx).
 *
 *   query/train   host pointers to row-major descriptors, `*_step` bytes
 *                 between rows (cv::Mat::step), dim must be 128
 *   dtype         SFM_GPU_DESC_F32 or SFM_GPU_DESC_U8
 *   k             1 or 2
 *   matches       out, n_query*k records; slots beyond counts[i] have
 *                 trainIdx = -1
 *   counts        out, n_query entries = min(k, n_train)  (the reference reads
 *                 matches[i][1] unconditionally at main.cpp:192; callers should
 *                 check counts)
 * n_query == 0 or n_train == 0 is not an error: counts are all 0.
 */
int sfm_gpu_knn_match(sfm_gpu_ctx* ctx, const void* query, int n_query, size_t query_step,
                      const void* train, int n_train, size_t train_step, int dim, int dtype,
                      int k, sfm_gpu_dmatch* matches, int32_t* counts);

/*
 * Ratio test of main.cpp:189-197 on the output of sfm_gpu_knn_match:
 * keeps row i iff counts[i] >= 2 and
 *   (double)matches[i*k].distance < ratio * (double)matches[i*k+1].distance.
 * Writes the surviving row numbers ascending into keep_rows (capacity
 * n_query) and their number into *n_keep.  Host-side, O(n_query).
 */
int sfm_gpu_ratio_test(const sfm_gpu_dmatch* matches, const int32_t* counts, int n_query, int k,
                       double ratio, int32_t* keep_rows, int32_t* n_keep);

/* ---- RANSAC: replaces main.cpp:222-224 -------------------------------- */

/*
 * findEssentialMat(ptsL, ptsR, K, FM_RANSAC, prob, threshold, mask)   main.cpp:224
 * (and main.cpp:418, bundle_adjustment.h:79 without a mask), or the
 * findFundamentalMat(FM_RANSAC) counterpart when model = FUNDAMENTAL.
 *
 *   pts1, pts2    host, n x 2 doubles, pixel coordinates (Point2d)
 *   K             host, 3x3 row-major camera matrix (ESSENTIAL only; may be
 *                 NULL for FUNDAMENTAL)
 *   prob          RANSAC confidence in (0,1)        (0.99 at main.cpp:224)
 *   threshold_px  inlier threshold in pixels, > 0   (1.0 at main.cpp:224)
 *   max_iters     iteration cap of the adaptive loop (OpenCV default 1000)
 *   mode          SFM_GPU_RANSAC_ADAPTIVE reproduces OpenCV's sequential,
 *                 self-shortening loop; SFM_GPU_RANSAC_FIXED solves exactly
 *                 n_hyp hypotheses and returns what scoring every one of their
 *                 models on every point returns (models that provably cannot
 *                 reach the best count are not scored to the end)
 *   sample_table  host, optional n_hyp x m int32 point indices (m = 5 or 7),
 *                 FIXED mode only; NULL: the library's own seeded table, whose
 *                 rows are independent draws of a counter-based generator
 *                 (ADAPTIVE always uses the stream OpenCV's RNG(-1) draws)
 *   n_hyp         hypotheses in FIXED mode (ignored in ADAPTIVE)
 *   mask          out, n bytes 0/1; left untouched when *ok == 0
 *   model         out, 9 doubles row-major (E on normalised coordinates with
 *                 unit Frobenius norm, or F on pixels with F33 = 1)
 *   n_inliers, iters_used   out, optional
 *   ok            out: 1 = a model was found, 0 = none (n < m, or no
 *                 hypothesis reached m inliers) — OpenCV returns an empty Mat
 * n == m (5 or 7): no RANSAC, one direct solve, mask all ones, `model` = the
 * first candidate (all of them: sfm_gpu_ransac_candidates).  FUNDAMENTAL with
 * 8 <= n <= 14 in ADAPTIVE mode returns SFM_GPU_ERR_UNSUPPORTED:
 * cv::findFundamentalMat switches to LMedS there, which is not on this path.
 */
int sfm_gpu_ransac(sfm_gpu_ctx* ctx, const double* pts1, const double* pts2, int n,
                   const double* K, int model, double prob, double threshold_px, int max_iters,
                   int mode, const int32_t* sample_table, int n_hyp, uint8_t* mask,
                   double* model_out, int32_t* n_inliers, int32_t* iters_used, int32_t* ok);

/*
 * Every candidate model of the minimal sample that won the last sfm_gpu_ransac
 * call on this context, in solver order (at most 10 for the five-point solver,
 * 3 for the seven-point solver).  With n == 5 (or 7) correspondences
 * cv::findEssentialMat / cv::findFundamentalMat run no RANSAC and return all
 * candidates stacked in one 3k x 3 matrix (main.cpp:224 would receive that
 * matrix); sfm_gpu_ransac then puts the first one into `model` and this call
 * delivers the whole stack.
 *   models_out   host, capacity x 9 doubles
 *   n_models     out; SFM_GPU_ERR_CAPACITY (and *n_models set) when capacity is
 *                too small
 */
int sfm_gpu_ransac_candidates(sfm_gpu_ctx* ctx, double* models_out, int capacity, int32_t* n_models);

/* ---- pose recovery: replaces main.cpp:420 ----------------------------- */

/*
 * recoverPose(EssentialMat, pointsL, pointsR, K, R, T)            main.cpp:420
 * (and bundle_adjustment.h:80): decomposes E into its four (R, t) candidates
 * and keeps the one that puts the most correspondences in front of both
 * cameras at a depth below distance_thresh (in units of the baseline).
 *
 *   E             host, 9 doubles row-major, on normalised coordinates (the
 *                 output of sfm_gpu_ransac with model = ESSENTIAL)
 *   pts1, pts2    host, n x 2 doubles, pixel coordinates
 *   K             host, 3x3 row-major camera matrix
 *   distance_thresh   OpenCV's default is 50.0 (the reference passes none)
 *   R_out, t_out  out, 9 and 3 doubles: x2 ~ R x1 + t, |t| = 1
 *   mask          optional, n bytes, in/out as in OpenCV: rows holding 0 do
 *                 not vote; on return 1 marks the correspondences that pass
 *                 the test under the chosen pose.  NULL: every row votes.
 *   n_good        out, optional: recoverPose's return value (the vote count)
 * Ties between candidates resolve in OpenCV's order (R1,t) (R2,t) (R1,-t)
 * (R2,-t).  SFM_GPU_ERR_ARG when E has fewer than two non-zero singular values.
 */
int sfm_gpu_recover_pose(sfm_gpu_ctx* ctx, const double* E, const double* pts1,
                         const double* pts2, int n, const double* K, double distance_thresh,
                         double* R_out, double* t_out, uint8_t* mask, int32_t* n_good);

/* ---- resection: replaces bundle_adjustment.h:139-155 -------------------- */

/*
 * solvePnPRansac(points_3d, pts, cameraMatrix, distortion_coeffs, r, t, false,
 *                SOLVEPNP_EPNP)                         bundle_adjustment.h:154
 * (Resection(), called per new image at main.cpp:520; distortion_coeffs is all
 * zeros there, and this entry point takes none): RANSAC over five-point EPnP
 * hypotheses drawn like OpenCV draws them (cv::RNG(-1), getSubset), inliers by
 * squared reprojection distance <= reprojection_error^2 evaluated in float32,
 * the self-shortening iteration count of RANSACUpdateNumIters, then one EPnP
 * over all inliers.
 *
 *   object_pts    host, n x 3 doubles (vector<Point3d>)
 *   image_pts     host, n x 2 doubles, pixels (vector<Point2d>)
 *   K             host, 3x3 row-major camera matrix
 *   iterations_count, reprojection_error, confidence
 *                 OpenCV's defaults, which the reference takes: 100, 8.0, 0.99
 *   rvec_out      out, 3 doubles: rotation vector (axis * angle), the `r` the
 *                 reference feeds to Rodrigues() at main.cpp:522
 *   tvec_out      out, 3 doubles
 *   R_out         out, optional, 9 doubles row-major: the rotation matrix itself
 *   inlier_mask   out, optional, n bytes 0/1 (OpenCV returns the inlier indices)
 *   n_inliers, iters_used   out, optional
 *   ok            out: 1 = a pose was found (solvePnPRansac's return value)
 * n < 4: SFM_GPU_ERR_ARG (OpenCV asserts); n == 4: SFM_GPU_ERR_UNSUPPORTED
 * (OpenCV switches to P3P); n == 5: one direct solve, every point an inlier.
 */
int sfm_gpu_pnp_ransac(sfm_gpu_ctx* ctx, const double* object_pts, const double* image_pts, int n,
                       const double* K, int iterations_count, double reprojection_error,
                       double confidence, double* rvec_out, double* tvec_out, double* R_out,
                       uint8_t* inlier_mask, int32_t* n_inliers, int32_t* iters_used, int32_t* ok);

/* ---- three-view track join: replaces main.cpp:442-456 ----------------- */

/* Keypoint indices of one track in images i, i+1 and j. */
typedef struct sfm_gpu_track3 {
    int32_t a, b, c;
} sfm_gpu_track3;

/*
 * The nested loop of main.cpp:442-456 over matches[i] (pair (i, i+1)) and
 * matches[i+1] (pair (i+1, j)): every (m, n) with
 * matches_ab[m].trainIdx == matches_bc[n].queryIdx gives the triple
 * (matches_ab[m].queryIdx, matches_ab[m].trainIdx, matches_bc[n].trainIdx), in
 * the order of that loop (m outer, n inner) -- the indices of kp1, kp2, kp3.
 *
 *   matches_ab, matches_bc   host, any order for ab; bc in ascending queryIdx
 *                            (equal values allowed), which is how match()
 *                            (main.cpp:189-231) and sfm_gpu_match_pairs emit it;
 *                            SFM_GPU_ERR_UNSUPPORTED otherwise
 *   tracks, capacity         out, room for `capacity` triples
 *   n_tracks                 out, number of triples found; when it exceeds
 *                            capacity the call returns SFM_GPU_ERR_CAPACITY and
 *                            writes none
 * The reference spends O(n_ab * n_bc) here; this is a bisection join.
 */
int sfm_gpu_track_join(sfm_gpu_ctx* ctx, const sfm_gpu_dmatch* matches_ab, int n_ab,
                       const sfm_gpu_dmatch* matches_bc, int n_bc, sfm_gpu_track3* tracks,
                       int capacity, int32_t* n_tracks);

/*
 * The loop of main.cpp:432-456 around that join, for a whole sequence at once:
 * list l (the selectMatches of pair l, as sfm_gpu_match_pairs returns them:
 * n_matches[l] records at matches + l * match_capacity) is joined with list
 * l + 1 for l = 0 .. n_lists - 2, in one upload, three launches and one
 * download instead of a call per image.
 *
 *   tracks, track_capacity   out, room for `track_capacity` triples in all
 *   track_offsets            out, n_lists int32: join l's triples are
 *                            tracks[track_offsets[l] .. track_offsets[l+1]), in the
 *                            order sfm_gpu_track_join gives; the last entry is
 *                            their total.  When that exceeds track_capacity the
 *                            call returns SFM_GPU_ERR_CAPACITY with the offsets
 *                            set (the total tells how much room a second call
 *                            needs) and writes no triples.
 * Every list but the first must be in ascending queryIdx order
 * (SFM_GPU_ERR_UNSUPPORTED otherwise).
 */
int sfm_gpu_track_join_sequence(sfm_gpu_ctx* ctx, const sfm_gpu_dmatch* matches, const int32_t* n_matches,
                                int match_capacity, int n_lists, sfm_gpu_track3* tracks,
                                int track_capacity, int32_t* track_offsets);

/* ---- device-resident images: the all-pairs driver --------------------- */

/*
 * Uploads and prepares one image's descriptors (and optionally its keypoint
 * coordinates, n x 2 float32 like cv::KeyPoint::pt) once, so that exhaustive
 * pair matching does not re-upload per pair.  Stands for the per-image
 * cv::Mat descriptors[i] / keypoints[i] of main.cpp:175-178.
 * `desc_on_device` != 0 means `desc` and `keypoints_xy` are device pointers.
 */
int sfm_gpu_image_create(sfm_gpu_ctx* ctx, const void* desc, int n, size_t step, int dim,
                         int dtype, const float* keypoints_xy, int desc_on_device,
                         sfm_gpu_image** out);
void sfm_gpu_image_destroy(sfm_gpu_ctx* ctx, sfm_gpu_image* img);
int sfm_gpu_image_rows(const sfm_gpu_image* img);

/*
 * The same preparation into caller-owned device memory, for callers that keep
 * many images in one allocation and exchange them between GPUs (the
 * all-gather of the exhaustive matcher moves these blocks as they are).
 *   sfm_gpu_image_packed_bytes(n)   size of the block for n rows
 *   sfm_gpu_image_prepare_into      enqueues conversion of descriptors[i]
 *                                   (main.cpp:175) into `packed_dev`; returns
 *                                   without waiting; value validation is
 *                                   reported by sfm_gpu_check_inputs
 *   sfm_gpu_image_wrap              non-owning image handle over such a block
 *   sfm_gpu_check_inputs            waits for the stream; SFM_GPU_ERR_NOT_U8 if
 *                                   any block prepared since the last check
 *                                   held a value outside the integers 0..255
 * The block layout is position independent (offsets only), so a block may be
 * copied to another GPU and wrapped there.
 */
size_t sfm_gpu_image_packed_bytes(int n);
int sfm_gpu_image_prepare_into(sfm_gpu_ctx* ctx, const void* desc, int n, size_t step, int dim,
                               int dtype, const float* keypoints_xy, int desc_on_device,
                               void* packed_dev);
/* sfm_gpu_image_prepare_into for `count` images already in device memory, in
 * one launch (all arrays are host arrays of device pointers / sizes). */
int sfm_gpu_image_prepare_many(sfm_gpu_ctx* ctx, const void* const* desc_dev, const int32_t* n,
                               size_t step, int dim, int dtype, const float* const* keypoints_dev,
                               void* const* packed_dev, int count);
int sfm_gpu_image_wrap(sfm_gpu_ctx* ctx, const void* packed_dev, int n, int has_keypoints,
                       sfm_gpu_image** out);
int sfm_gpu_check_inputs(sfm_gpu_ctx* ctx);

/*
 * knnMatch (main.cpp:184) between two prepared images.  matches/counts are
 * host pointers, or device pointers when out_on_device != 0 (then the call
 * only enqueues work on the context's stream).
 */
int sfm_gpu_knn_match_images(sfm_gpu_ctx* ctx, const sfm_gpu_image* query,
                             const sfm_gpu_image* train, int k, sfm_gpu_dmatch* matches,
                             int32_t* counts, int out_on_device);

/*
 * sfm_gpu_knn_match_images with device outputs, for callers that cannot
 * allocate device memory: code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc.  No counterpart in the reference.  The call
 * allocates nothing and copies nothing: the sweep records, the list of queries
 * for the float-ranked pass and the job descriptor live in a static device
 * array that is part of the library's code object (16.2 MiB per GPU, held until
 * the process ends), and the call only enqueues kernels on the context's
 * stream.
 *
 *   matches_dev, counts_dev   device pointers, n_query*k records / n_query
 *                             counts, as sfm_gpu_knn_match_images writes them
 * Limits: at most 32768 query rows, and train parts x query rows (padded to
 * 1024) at most 262144 -- 32768 x 32768 fits (8 parts), as does any smaller
 * pair; SFM_GPU_ERR_CAPACITY beyond.  All calls on one GPU share that GPU's array:
 * two such calls that overlap in time on one GPU (different streams or
 * contexts) race; calls on one stream are ordered and safe.
 */
int sfm_gpu_knn_match_images_noalloc(sfm_gpu_ctx* ctx, const sfm_gpu_image* query,
                                     const sfm_gpu_image* train, int k,
                                     sfm_gpu_dmatch* matches_dev, int32_t* counts_dev);

/* Parameters of one match() iteration (main.cpp:181-231). */
typedef struct sfm_gpu_match_params {
    double ratio;        /* 0.7 at main.cpp:189                                */
    int32_t model;       /* SFM_GPU_MODEL_*                                    */
    int32_t mode;        /* SFM_GPU_RANSAC_*                                   */
    double K[9];         /* camera matrix of the query image (main.cpp:179)    */
    double prob;         /* 0.99                                               */
    double threshold_px; /* 1.0                                                */
    int32_t max_iters;   /* 1000                                               */
    int32_t n_hyp;       /* FIXED mode hypothesis count                        */
} sfm_gpu_match_params;

/*
 * The body of the reference's pair loop (main.cpp:175-231, 267) for a list of
 * image pairs: kNN(k=2) -> ratio test -> RANSAC mask -> inlier matches.
 *
 *   images        array of n_images prepared images
 *   pairs         host, n_pairs x 2 int32 (query image, train image), each in [0, n_images)
 *   n_good        out, per pair: ratio-test survivors  ("goodMatches.size()")
 *   n_inliers     out, per pair: RANSAC inliers        ("selectMatches.size()")
 *   inlier_matches  out, optional: n_pairs * match_capacity records; pair p's
 *                 selectMatches occupy [p*match_capacity, +n_inliers[p])
 *   match_capacity  records reserved per pair (>= the largest query image when
 *                 inlier_matches != NULL)
 * Every image named by a pair must carry keypoints (SFM_GPU_ERR_ARG otherwise):
 * they are the points RANSAC runs on.
 */
int sfm_gpu_match_pairs(sfm_gpu_ctx* ctx, sfm_gpu_image* const* images, int n_images, const int32_t* pairs,
                        int n_pairs, const sfm_gpu_match_params* params, int32_t* n_good,
                        int32_t* n_inliers, sfm_gpu_dmatch* inlier_matches,
                        int match_capacity);

/*
 * sfm_gpu_match_pairs followed, for every pair, by what main.cpp:400-420 does
 * with the first pair: the points of the pair's selectMatches are gathered
 * (main.cpp:400-416), findEssentialMat(pointsL, pointsR, K, RANSAC) runs on them
 * with OpenCV's defaults -- prob 0.999, threshold 1.0, at most 1000 iterations,
 * the adaptive loop (main.cpp:418) -- and recoverPose(E, pointsL, pointsR, K, R, T)
 * takes that matrix and lets every selected match vote (main.cpp:420).  The
 * points never leave the GPU.  params->model must be SFM_GPU_MODEL_ESSENTIAL.
 *
 *   distance_thresh   as in sfm_gpu_recover_pose (OpenCV's default: 50.0)
 *   models_out    out, optional: n_pairs x 9 doubles, the essential matrix of each
 *                 pair's SECOND RANSAC (the one the pose is decomposed from)
 *   R_out, t_out  out: n_pairs x 9 and n_pairs x 3 doubles
 *   n_pose_good   out, per pair: recoverPose's return value; -1 (and R, t, model
 *                 all zero) for a pair whose first or second RANSAC found no model
 */
int sfm_gpu_match_pairs_pose(sfm_gpu_ctx* ctx, sfm_gpu_image* const* images, int n_images,
                             const int32_t* pairs, int n_pairs, const sfm_gpu_match_params* params,
                             int32_t* n_good, int32_t* n_inliers, sfm_gpu_dmatch* inlier_matches,
                             int match_capacity, double distance_thresh, double* models_out,
                             double* R_out, double* t_out, int32_t* n_pose_good);

/*
 * The same work split into "enqueue" and "collect", so that a host can put the
 * next batch on the stream while the previous one is still running and the GPU
 * never waits for the host between batches.  No counterpart in the reference,
 * whose loop finishes a pair before it starts the next (main.cpp:175-231).
 *
 *   sfm_gpu_match_pairs_begin  enqueues one batch (every pair must fit one
 *                     launch group: SFM_GPU_ERR_CAPACITY otherwise, as when two
 *                     batches are already in flight).  It returns without
 *                     waiting for the device.  (With mode ADAPTIVE the loop's
 *                     first 32 sample rows are enqueued, which is all it takes
 *                     whenever the inlier share is above about 67 %, or below 5
 *                     inliers; a batch in which some pair wants more is run
 *                     again, waiting between the rounds, inside _end.)
 *                     The images must stay alive and unchanged until the
 *                     matching _end; other calls on the context may be made in
 *                     between (they queue behind the batch).
 *   sfm_gpu_match_pairs_end    waits for the OLDEST batch in flight and returns
 *                     its counts (capacity = room in both arrays, >= that
 *                     batch's n_pairs).
 * sfm_gpu_match_pairs_begin leaves counts only (what the match-count matrix of
 * exhaustive matching needs); sfm_gpu_match_pairs_begin_records also leaves what
 * match() itself returns per pair, the selectMatches of main.cpp:225-231, 267:
 *
 *   inlier_matches_dev   caller-owned DEVICE memory, n_pairs * match_capacity
 *                     records; pair p's selectMatches occupy
 *                     [p*match_capacity, +n_inliers[p]), complete once the matching
 *                     _end has returned (or for any work queued on the context's
 *                     stream behind the _begin).  Must stay untouched until then.
 *   match_capacity    records reserved per pair, >= the largest query image
 *   inlier_matches_host  optional PINNED host memory of the same shape
 *                     (hipHostMalloc / hipHostRegister; SFM_GPU_ERR_ARG otherwise):
 *                     behind the batch, on a stream of its own so that the next
 *                     batch is not held up, the records travel into it; valid once
 *                     the matching _end has returned.  Entries at and beyond a pair's
 *                     n_inliers are unspecified (left alone, or equal to the device
 *                     buffer's: when another batch has been begun meanwhile the whole
 *                     record area crosses as one DMA copy beside that batch's sweep,
 *                     otherwise a kernel writes just the records that exist).
 */
int sfm_gpu_match_pairs_begin(sfm_gpu_ctx* ctx, sfm_gpu_image* const* images, int n_images,
                              const int32_t* pairs, int n_pairs, const sfm_gpu_match_params* params);
int sfm_gpu_match_pairs_begin_records(sfm_gpu_ctx* ctx, sfm_gpu_image* const* images, int n_images,
                                      const int32_t* pairs, int n_pairs, const sfm_gpu_match_params* params,
                                      sfm_gpu_dmatch* inlier_matches_dev, int match_capacity,
                                      sfm_gpu_dmatch* inlier_matches_host);
int sfm_gpu_match_pairs_end(sfm_gpu_ctx* ctx, int32_t* n_good, int32_t* n_inliers, int capacity);

/*
 * Ordering between two contexts (two streams) of one device: work enqueued on
 * `waiter` after this call starts only when the batch most recently enqueued on
 * `source` (sfm_gpu_match_pairs / _begin) has finished its distance sweep and
 * its neighbour extraction; a no-op until `source` has enqueued a batch after
 * the first such call.  For hosts that convert the next batch's descriptors on
 * a second stream: the conversion is bound by HBM and shares the chip well with
 * what follows in the batch (ratio test, hypothesis generation and solving:
 * latency-bound; scoring: bound by vector issue), while beside the sweep or the
 * neighbour extraction it would slow those down by as much as it saves.  No
 * counterpart in the reference.
 */
int sfm_gpu_stream_wait_front(sfm_gpu_ctx* waiter, sfm_gpu_ctx* source);

/* ---- measurement support --------------------------------------------- */

/*
 * Per-kernel device time.  With profiling enabled the library brackets each
 * launch of its named kernels with HIP events on the context's stream; the
 * query returns accumulated milliseconds and launch count since the last
 * reset.  Kernel names: "knn2_mfma", "prep_rows", "knn_finalize",
 * "ratio_compact", "ransac_table", "ransac_solve", "ransac_score",
 * "ransac_select", "pose_vote", "track_join", "pnp_prep", "pnp_solve",
 * "pnp_score".  Has no counterpart in the reference.
 */
int sfm_gpu_profile_enable(sfm_gpu_ctx* ctx, int on);
/* Restricts the timing to one named kernel (NULL: all of them), so that a
 * throughput measurement can carry the timer of its dominant kernel without
 * paying for events around every launch. */
int sfm_gpu_profile_select(sfm_gpu_ctx* ctx, const char* kernel);
int sfm_gpu_profile_reset(sfm_gpu_ctx* ctx);
int sfm_gpu_profile_query(sfm_gpu_ctx* ctx, const char* kernel, double* total_ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif
