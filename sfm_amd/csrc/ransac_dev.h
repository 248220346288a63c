// Device-side description of one RANSAC problem and the launchers of the
// kernels in ransac_kernels.hip / pair_kernels.hip.
#pragma once

#include "sfm_internal.h"

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
    R_COUNT = 16
};

constexpr int MAX_MODELS = 10;  // candidate models per hypothesis (5-point: <= 10)

struct RansacProb {
    const double* px1;     // [n][2] pixel coordinates, image 1
    const double* px2;     // [n][2] pixel coordinates, image 2
    const int32_t* n_ptr;  // device: number of correspondences
    double* x1;            // [cap][2] normalised (essential model) / float32 pixels widened (fundamental)
    double* x2;
    float* f1;             // [cap][2] float32 pixels (fundamental model)
    float* f2;
    float* soa;            // [4][cap_pad] float32 planes X1, Y1, X2, Y2 of x1/x2
    int32_t* table;        // [hyp_cap][m]
    double* models;        // [hyp_cap][MAX_MODELS][9]
    int32_t* nmodels;      // [hyp_cap]
    int32_t* counts;       // [hyp_cap][MAX_MODELS], -1 = no such model
    int32_t* list;         // [hyp_cap * MAX_MODELS] dense list of hyp*MAX_MODELS + q of one batch
    int32_t* result;       // [R_COUNT]
    uint8_t* mask;         // [cap]
    double* model_out;     // [9]
    double fx, fy, cx, cy;
    double tlo, thi;       // (double)tf * (1 -/+ 2^-20)
    float tf;              // (float)(threshold^2), threshold already scaled
    int32_t model;         // SFM_GPU_MODEL_*
    int32_t m;             // minimal sample size: 5 or 7
    int32_t cap_pad;       // plane stride of soa (even, >= cap)
};

hipError_t launch_ransac_prep(hipStream_t s, const RansacProb* probs, int n_prob, int cap);
hipError_t launch_ransac_table(hipStream_t s, const RansacProb* probs, int n_prob, int row0, int nrows);
hipError_t launch_ransac_table_fixed(hipStream_t s, const RansacProb* probs, int n_prob, int nrows);
hipError_t launch_ransac_solve(hipStream_t s, const RansacProb* probs, int n_prob, int model, int hyp0, int n_hyp);
// prune > 0: every model is first scored on the first `prune` points, then the
// models that provably cannot reach the best count are dropped (their counts
// stay partial) and the rest go on; the selected model, its count and the mask
// are the same as with prune = 0.
hipError_t launch_ransac_score(hipStream_t s, const RansacProb* probs, int n_prob, int cap, int model, int hyp0,
                               int n_hyp, int prune, int mid);
void set_ransac_score_filter(int on);
void set_ransac_score_prune(int on);
hipError_t launch_ransac_select(hipStream_t s, const RansacProb* probs, int n_prob, int n_hyp);
hipError_t launch_ransac_mask(hipStream_t s, const RansacProb* probs, int n_prob, int cap);

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
    int32_t n_query;
    int32_t capacity;
};

hipError_t launch_ratio_compact(hipStream_t s, const PairDev* pairs, int n_pairs, int n_query_max, double ratio);
// dst and src_pinned are 16-byte aligned and padded to a multiple of 16 bytes
hipError_t launch_upload_block(hipStream_t s, void* dst, const void* src_pinned, size_t bytes, int32_t* redo,
                               size_t redo_stride, int n_jobs);
hipError_t launch_inlier_compact(hipStream_t s, const PairDev* pairs, int n_pairs);

}  // namespace sfm
