// Internal declarations shared by the HIP translation units of libsfm_gpu.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "sfm_gpu.h"

namespace sfm {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int DIM = 128;
// Granule in which the train rows are divided among workgroups.
constexpr int CHUNK = 256;
// One workgroup sweeps at most this many chunks (512 tiles of 32 rows: the
// tile number is packed into 9 bits under the 23-bit x).
constexpr int MAX_SWEEP_CHUNKS = 64;
// Descriptor stores are allocated in multiples of this many rows so that every
// workgroup shape may read whole query blocks and whole train chunks.
constexpr int ROW_ALIGN = 1024;
// x = dot + cin is what the MFMA chain leaves in the accumulator.
// cin = floor(Tn/2) + KEY_OFFSET keeps x in [64, 5226560] for every pair of
// 0..255-valued rows: 23 bits, so (x << 9) | tile fits 32 bits.
constexpr int KEY_OFFSET = 2080896;
constexpr int PAD_CIN = 0x7FFFFF;  // padding rows rank behind every real row
// Sweep records per (train part, query row): one per lane that holds the query in
// the sweep (4 with 16x16 MFMAs, 2 of the 4 slots used with 32x32).
constexpr int REC_SUBS = 4;

// One image's descriptors as the distance kernel wants them.
//   desc[r][d]  = value - 128 as int8 (byte = value ^ 0x80)
//   qn[r]       = sum_d (value-128)^2                (query-side norm)
//   cin[r]      = floor(Tn/2) + KEY_OFFSET, Tn = sum_d (value-127)^2 - 128
// so that for query q and train t:
//   d2(q,t) = qn[q] + 2*(dot(~q, t) + cin[t]) + (Tn & 1) - 2*KEY_OFFSET
struct ImageDev {
    int n = 0;
    int rows_alloc = 0;
    int8_t* desc = nullptr;
    int32_t* cin = nullptr;
    int32_t* qn = nullptr;
    float2* kp = nullptr;
};

// One image to convert: source rows (float32 or uint8, device memory),
// optional keypoints to copy beside them, destination layout.
struct PrepJob {
    const unsigned char* src;
    size_t step;
    const float2* kp_src;
    ImageDev img;
};

struct KnnJob {
    const int8_t* q_desc;
    const int8_t* t_desc;
    const int32_t* t_cin;
    const int32_t* q_qn;
    const int32_t* t_qn;
    uint4* partial;       // [n_splits][q_rows_alloc][REC_SUBS] {best group, 2nd group, 3rd minimum, first unit}
    sfm_gpu_dmatch* out;  // [n_query][k]
    int32_t* counts;      // [n_query]
    int32_t* redo;        // [0] = count, [1..] = query rows needing the float-ranked pass
    int n_query;
    int n_train;
    int q_rows_alloc;
    int t_chunks;
};

struct KernelTimer {
    std::string name;
    double total_ms = 0.0;
    int64_t launches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

}  // namespace sfm

struct sfm_gpu_image {
    sfm::ImageDev d;
};

// launchers implemented in knn_kernels.hip
namespace sfm {

hipError_t launch_prep_rows(hipStream_t s, const PrepJob* jobs_dev, int n_jobs, int rows_alloc_max,
                            int dtype, int32_t* err_flag);
// Picks the workgroup shape and split count for a batch of equally sized jobs.
struct KnnPlan {
    int n_splits;
    int chunks_per_split;
    int q_per_block;
    int variant;
    // timing experiments only, honoured by the timing build alone (wrong results): bit 0 = no
    // digest of the accumulators, bit 1 = no DMA of further stages, bit 2 = no stage barrier
    int ablate;
    int group_mode;  // geometry of the row groups in the sweep's records: 0 = 32x32 tiles, 1 = 16x16 tiles
};
KnnPlan plan_knn(int n_query_max, int n_train_max, int n_jobs, int force_variant);
// ev_start / ev_stop (may be null): events attached to the launch itself -- they take the dispatch's own
// start and end times, and unlike events recorded around it they put no barrier into the queue
hipError_t launch_knn2(hipStream_t s, const KnnJob* jobs_dev, int n_jobs, int q_rows_alloc_max,
                       const KnnPlan& plan, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
// skip_ratio > 0 (k = 2 only): queries that fail the ratio test
// d1 < skip_ratio * d2 end with counts = 0 (those that certainly fail it get no
// exact neighbours at all); counts = 2 then means "passes".  launch_knn_redo
// must be given the same value.
hipError_t launch_knn_finalize(hipStream_t s, const KnnJob* jobs_dev, int n_jobs, int n_query_max,
                               int n_splits, int k, int group_mode, double skip_ratio);
hipError_t launch_knn_redo(hipStream_t s, const KnnJob* jobs_dev, int n_jobs, int n_splits, int k,
                           double skip_ratio);
// timing build only: the in-kernel stamps of the last 16x16 sweep, 8 words per workgroup
hipError_t read_sweep_stamps(unsigned long long* out, int n_wg);
// overwrites the LDS of every CU with a pattern (test hook)
hipError_t launch_lds_poison(hipStream_t s, uint32_t pattern);
hipError_t launch_mfma_layout_probe(hipStream_t s, const int8_t* a, const int8_t* b, int32_t* c);
hipError_t launch_sqrt_probe(hipStream_t s, int32_t* mismatches);
hipError_t launch_mfma16_layout_probe(hipStream_t s, const int8_t* a, const int8_t* b, int32_t* c);

}  // namespace sfm
