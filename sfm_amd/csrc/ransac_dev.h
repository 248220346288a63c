// Device-side description of one RANSAC problem and the launchers of the
// kernels in ransac_kernels.hip / pair_kernels.hip.
#pragma once

#include "sfm_internal.h"

struct sfm_gpu_ctx;

namespace sfm {

// slots of RansacProb::result (int32)
enum {
    R_OK = 0,      // 1 when a model was selected
    R_HYP = 1,     // winning hypothesis (sample-table row)
    R_Q = 2,       // winning candidate model of that hypothesis
    R_NINL = 3,    // its inlier count
    R_ITERS = 4,   // iterations the loop consumed
    R_VALID = 5,   // sample-table rows generated so far
    R_DIRECT = 6,  // n == m: single direct solve, mask all ones
    R_RNG_HI = 7,  // generator state parked between table batches
    R_RNG_LO = 8,
    R_N = 9,       // copy of the point count
    R_NLIST = 10,  // candidate models listed for the batch being scored
    R_MAXABS = 11, // largest |coordinate| of the problem's points, float32 bits
    R_BOUND = 12,  // largest complete inlier count known so far in the batch being scored
    R_S1 = 13,     // staged scoring: end of the stage that follows the opening stretch (a multiple of 128, or n)
    R_MID = 14,    // place of a further cut between R_S1 and n (a multiple of 128, or n)
    R_S0 = 15,     // the opening stretch [0, R_S0): every listed model is counted there before anything is planned
    R_S3 = 16,     // a third cut between R_MID and n
    R_LIVE = 17,   // R_LIVE + k, k = 0..4: leading groups of 64 of the ordered list in which some model could
                   // still reach the bound when stage k of the scoring began (an upper bound; stage k writes k + 1)
    R_TOP = 22,    // models at the head of the ordered list (list2) that were completed up front: the stages start behind them
    R_COUNT = 24,
    // adaptive mode (the staged scoring of the fixed-count mode does not run there):
    R_NITERS = R_S1,  // the loop bound as it stands, shrinking with every better model
    R_STATE = R_MID   // R_STATE_* bits
};
enum { R_STATE_DONE = 1, R_STATE_FAILED = 2 };

constexpr int MAX_MODELS = 10;  // candidate models per hypothesis (5-point: <= 10)

struct RansacProb {
    const double* px1;     // [n][2] pixel coordinates, image 1
    const double* px2;     // [n][2] pixel coordinates, image 2
    const int32_t* n_ptr;  // device: number of correspondences
    double* x1;            // [cap][2] normalised (essential model) / float32 pixels widened (fundamental)
    double* x2;
    float* f1;             // [cap][2] float32 pixels (fundamental model)
    float* f2;
    float* soa;            // [ceil(cap/4)][X1 Y1 X2 Y2][4] float32 copies of x1/x2, four points per 64-byte block
    int32_t* table;        // [hyp_cap][m]
    double* models;        // [hyp_cap][MAX_MODELS][9]
    int32_t* nmodels;      // [hyp_cap]
    int32_t* counts;       // [hyp_cap][MAX_MODELS], -1 = no such model
    int32_t* list;         // [hyp_cap * MAX_MODELS] dense list of hyp*MAX_MODELS + q of one batch
    int32_t* list2;        // [hyp_cap * MAX_MODELS] where the list is put in order (ransac_sortplan_kernel)
    int32_t* result;       // [R_COUNT]
    uint8_t* mask;         // [cap]
    double* model_out;     // [9]
    double fx, fy, cx, cy;
    double tlo, thi;       // (double)tf * (1 -/+ 2^-20)
    float tf;              // (float)(threshold^2), threshold already scaled
    int32_t model;         // SFM_GPU_MODEL_*
    int32_t m;             // minimal sample size: 5 or 7
};

// What RANSAC keeps of correspondence i, given its pixel coordinates in both
// images: normalised coordinates for the essential model (x = (u-cx)/fx,
// y = (v-cy)/fy), float32-narrowed pixels for the fundamental model, as the two
// OpenCV front ends prepare them, plus the float32 planes the scoring filter
// reads (blocks of four points, [block][X1 Y1 X2 Y2][4]: one 64-byte scalar load
// brings four points with pairs of equal coordinates adjacent).
__device__ __forceinline__ void ransac_prep_point(const RansacProb& pb, int i, double u1, double v1, double u2,
                                                  double v2)
{
    double a1, b1, a2, b2;
    if (pb.model == SFM_GPU_MODEL_ESSENTIAL) {
        a1 = (u1 - pb.cx) / pb.fx;
        b1 = (v1 - pb.cy) / pb.fy;
        a2 = (u2 - pb.cx) / pb.fx;
        b2 = (v2 - pb.cy) / pb.fy;
    } else {
        const float f1x = (float)u1, f1y = (float)v1, f2x = (float)u2, f2y = (float)v2;
        pb.f1[2 * i] = f1x;
        pb.f1[2 * i + 1] = f1y;
        pb.f2[2 * i] = f2x;
        pb.f2[2 * i + 1] = f2y;
        // the same values widened again: what the error function computes on
        a1 = (double)f1x;
        b1 = (double)f1y;
        a2 = (double)f2x;
        b2 = (double)f2y;
    }
    pb.x1[2 * i] = a1;
    pb.x1[2 * i + 1] = b1;
    pb.x2[2 * i] = a2;
    pb.x2[2 * i + 1] = b2;
    float* blk = pb.soa + (size_t)(i >> 2) * 16 + (i & 3);
    blk[0] = (float)a1;
    blk[4] = (float)b1;
    blk[8] = (float)a2;
    blk[12] = (float)b2;
}

// Clears a problem's result block (every slot but the point count R_N).
__device__ __forceinline__ void ransac_reset_result(const RansacProb& pb)
{
    pb.result[R_OK] = 0;
    pb.result[R_HYP] = -1;
    pb.result[R_Q] = -1;
    pb.result[R_NINL] = 0;
    pb.result[R_ITERS] = 0;
    pb.result[R_VALID] = 0;
    pb.result[R_DIRECT] = 0;
    pb.result[R_NLIST] = 0;
    pb.result[R_MAXABS] = 0;
}

hipError_t launch_ransac_prep(hipStream_t s, const RansacProb* probs, int n_prob, int cap);
hipError_t launch_ransac_table(hipStream_t s, const RansacProb* probs, int n_prob, int row0, int nrows);
hipError_t launch_ransac_table_fixed(hipStream_t s, const RansacProb* probs, int n_prob, int nrows);
// gen_rows > 0 (fundamental model only, hyp0 = 0): the solver draws the fixed-count
// sample rows itself instead of reading a table made by launch_ransac_table_fixed
// split (seven-point only): -1 = one or two lanes per hypothesis by batch size, 0 / 1 = one / two
hipError_t launch_ransac_solve(hipStream_t s, const RansacProb* probs, int n_prob, int model, int hyp0, int n_hyp,
                               int gen_rows = 0, int split = -1);
// staged (fixed-count mode): every model is scored on an opening stretch of the
// points; the list is then ordered by those counts and its best few are completed
// (the largest of their counts bounds the winner's from below); four stages
// follow, their ends chosen per problem on the device, in which a wavefront of
// models stops at once when none of them can reach the bound any more (their
// counts stay partial).  The selected model, its count and the mask are those of
// staged = false.
// filter_mode: 1 = float32 filter with a rigorous bound in front of the double
// precision test, 0 = double precision only (same counts).  prune_mode: 0 = never
// drop a model, 1 = as described above for batches of eight problems or more (4: for any batch),
// 2 / 3 (and 1 for smaller batches) = two stages sized from the opening stretch's best count, the
// list compacted between them, two cuts / one cut.
hipError_t launch_ransac_score(hipStream_t s, const RansacProb* probs, int n_prob, int cap, int model, int hyp0,
                               int n_hyp, bool staged, int filter_mode, int prune_mode);
hipError_t launch_ransac_mask(hipStream_t s, const RansacProb* probs, int n_prob, int cap);
// timing build: per-wavefront stamps of the scoring launches since the last call (hipErrorNotSupported otherwise)
hipError_t read_score_stamps(unsigned long long* out, int n_max, int* count);
// Adaptive mode: replays RANSACPointSetRegistrator::run's loop over the counts of
// sample rows [row0, row0 + nrows) on the device (one wavefront per problem); a
// problem's progress lives in its result block (R_ITERS, R_NINL, R_HYP, R_Q,
// R_NITERS, R_STATE), and R_OK is set once it is done.
hipError_t launch_ransac_replay(hipStream_t s, const RansacProb* probs, int n_prob, int row0, int nrows, int max_iters,
                                double confidence);

// One image pair between the k-NN output and RANSAC (pair_kernels.hip).
struct PairDev {
    const sfm_gpu_dmatch* knn;  // [n_query][2]
    const int32_t* counts;      // [n_query]
    const float2* kp_q;         // query image keypoints (may be null)
    const float2* kp_t;         // train image keypoints (may be null)
    sfm_gpu_dmatch* good;       // [n_query] ratio-test survivors, ascending query row
    double* px1;                // [n_query][2]
    double* px2;
    int32_t* n_good;            // [1]
    const uint8_t* mask;        // RANSAC mask over the survivors
    const int32_t* result;      // RANSAC result block
    sfm_gpu_dmatch* inliers;    // [capacity] or null
    int32_t* n_inliers;         // [1]
    int32_t* host_counts;       // [2] pinned host memory: (n_good, n_inliers) for the caller, or null
    const RansacProb* prob;     // the pair's RANSAC problem: the ratio kernel prepares its points itself; or null
    int32_t n_query;
    int32_t capacity;
};

hipError_t launch_ratio_compact(hipStream_t s, const PairDev* pairs, int n_pairs, int n_query_max, double ratio);
// dst and src_pinned are 16-byte aligned and padded to a multiple of 16 bytes
hipError_t launch_upload_block(hipStream_t s, void* dst, const void* src_pinned, size_t bytes, int32_t* redo,
                               size_t redo_stride, int n_jobs);
// check_state: pairs whose adaptive loop is not finished report -1 inliers instead
hipError_t launch_inlier_compact(hipStream_t s, const PairDev* pairs, int n_pairs, bool check_state = false);
// pair p's first counts[2p+1] records from src + p*stride to dst_pinned + p*stride (device-visible host memory)
hipError_t launch_records_to_host(hipStream_t s, const sfm_gpu_dmatch* src, sfm_gpu_dmatch* dst_pinned, size_t stride,
                                  const int32_t* counts, int n_pairs);
// pairs != null: the kernel also gathers each pair's inlier matches and reports
// its two counts (what launch_inlier_compact does as a launch of its own)
hipError_t launch_ransac_select(hipStream_t s, const RansacProb* probs, int n_prob, int n_hyp,
                                const PairDev* pairs);


// recoverPose for the g problems at dprobs (essential model, stream drained,
// masks final): pose.hip.  Pairs with n_inliers[i] <= 0 get n_pose_good = -1.
// every_point: all of a problem's correspondences vote, not only its inliers.
int pose_for_pairs(sfm_gpu_ctx* ctx, const RansacProb* dprobs, int g, const int32_t* n_inliers, const double* K,
                   double distance_thresh, double* models_out, double* R_out, double* t_out, int32_t* n_pose_good,
                   int nq_max, bool every_point);
// px1/px2 of every pair = the keypoint coordinates of its inlier matches (Point2f
// widened to double, main.cpp:400-416): the input of a RANSAC over them
hipError_t launch_gather_inlier_points(hipStream_t s, const PairDev* pairs, int n_pairs, int n_query_max);
}  // namespace sfm
