// Glue between the k-NN output and RANSAC for one image pair, on the device:
// the ratio test with point gathering (/root/reference/main.cpp:187-198) and
// the final inlier selection (/root/reference/main.cpp:225-231).
#include <algorithm>

#include "ransac_dev.h"

namespace sfm {

namespace {

// Order-preserving compaction of the rows with flag set, 1024 rows per step.
// Returns this thread's output slot (or -1) and advances *base.
__device__ __forceinline__ int compact_slot(bool flag, int* wsum, int* base)
{
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(flag);
    const int lane_prefix = __popcll(mask & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(mask);
    __syncthreads();
    int wave_prefix = 0, total = 0;
    for (int w = 0; w < 16; ++w) {
        const int c = wsum[w];
        if (w < wave) wave_prefix += c;
        total += c;
    }
    const int b = *base;
    __syncthreads();
    if (tid == 0) *base = b + total;
    __syncthreads();
    return flag ? b + wave_prefix + lane_prefix : -1;
}

// Ratio test verdict of query row q.  With ratio > 0 the k-NN kernels have
// already folded it into the count (2 = passes, 0 = fails).
__device__ __forceinline__ bool ratio_keep(const PairDev& pd, int q, double ratio)
{
    if (pd.counts[q] < 2) return false;
    if (ratio > 0.0) return true;
    return (double)pd.knn[2 * q].distance < ratio * (double)pd.knn[2 * q + 1].distance;
}

// Ratio test and gather; grid = (tiles of 1024 query rows, pairs).  A workgroup
// compacts its own tile; the slot of its first survivor is the number of
// survivors in the tiles before it, which it simply counts again itself (reads
// that hit L2, no communication between workgroups, deterministic order).
__global__ void __launch_bounds__(1024) ratio_compact_kernel(const PairDev* __restrict__ pairs, double ratio)
{
    __shared__ int wsum[16];
    __shared__ int base;
    const PairDev pd = pairs[blockIdx.y];
    const int tile = blockIdx.x;
    const int n_tiles = (pd.n_query + 1023) / 1024;
    if (tile >= n_tiles && !(tile == 0 && n_tiles == 0)) return;
    const int tid = threadIdx.x;
    int before = 0;
    // (eight tiles' loads at a time: a loop of load -> test pays a round trip per tile)
    for (int j0 = 0; j0 < tile; j0 += 8) {
        bool keep[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) keep[k] = j0 + k < tile && ratio_keep(pd, (j0 + k) * 1024 + tid, ratio);
#pragma unroll
        for (int k = 0; k < 8; ++k) before += keep[k] ? 1 : 0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off);
    if ((tid & 63) == 0) wsum[tid >> 6] = before;
    __syncthreads();
    if (tid == 0) {
        int sum = 0;
        for (int w = 0; w < 16; ++w) sum += wsum[w];
        base = sum;
    }
    __syncthreads();
    const int q = tile * 1024 + tid;
    const bool flag = q < pd.n_query && ratio_keep(pd, q, ratio);
    const int slot = compact_slot(flag, wsum, &base);
    // with a RANSAC problem attached, its points are prepared here (Point2f
    // widened to double as at main.cpp:194-195, then what ransac_prep_kernel
    // would do) instead of travelling through the pixel arrays
    const bool fused = pd.prob != nullptr;
    RansacProb pb;
    if (fused) pb = *pd.prob;
    if (slot >= 0) {
        const sfm_gpu_dmatch m0 = pd.knn[2 * q];
        pd.good[slot] = m0;
        if (pd.kp_q && pd.kp_t) {
            const float2 a = pd.kp_q[m0.queryIdx];
            const float2 b = pd.kp_t[m0.trainIdx];
            if (fused) {
                ransac_prep_point(pb, slot, (double)a.x, (double)a.y, (double)b.x, (double)b.y);
            } else {
                pd.px1[2 * slot] = (double)a.x;
                pd.px1[2 * slot + 1] = (double)a.y;
                pd.px2[2 * slot] = (double)b.x;
                pd.px2[2 * slot + 1] = (double)b.y;
            }
        }
    }
    if (fused && tid == 0 && tile == 0) ransac_reset_result(pb);
    if (tid == 0 && (tile == n_tiles - 1 || n_tiles == 0)) {
        *pd.n_good = base;
        if (fused) pb.result[R_N] = base;
    }
}

// check_state (adaptive RANSAC enqueued without waiting for its outcome): a pair
// whose loop has not finished reports -1 inliers, and the caller runs it again.
__global__ void __launch_bounds__(1024) inlier_compact_kernel(const PairDev* __restrict__ pairs, int check_state)
{
    __shared__ int wsum[16];
    __shared__ int base;
    const PairDev pd = pairs[blockIdx.x];
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    const int n = *pd.n_good;
    if (check_state && !(pd.result[R_STATE] & R_STATE_DONE)) {
        if (threadIdx.x == 0) {
            *pd.n_inliers = 0;
            if (pd.host_counts) {
                pd.host_counts[0] = n;
                pd.host_counts[1] = -1;
            }
        }
        return;
    }
    const bool ok = pd.result[R_OK] != 0;
    if (ok) {
        for (int start = 0; start < n; start += 1024) {
            const int i = start + threadIdx.x;
            const bool flag = i < n && pd.mask[i] != 0;
            const int slot = compact_slot(flag, wsum, &base);
            if (slot >= 0 && pd.inliers && slot < pd.capacity) pd.inliers[slot] = pd.good[i];
        }
    }
    if (threadIdx.x == 0) {
        *pd.n_inliers = base;
        // the caller's copy goes straight to its pinned buffer: visible to the
        // host once it has waited for this kernel, and one copy less in the stream
        if (pd.host_counts) {
            pd.host_counts[0] = n;
            pd.host_counts[1] = base;
        }
    }
}

// Brings one batch's descriptor block from pinned host memory (which the device
// reads directly) into device memory and clears the jobs' redo counters: one
// short launch in the stream instead of three copies and a fill, each of which
// costs more in queue hand-over than in transfer.
__global__ void __launch_bounds__(256) upload_block_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src,
                                                           int n16, int32_t* redo, size_t redo_stride, int n_jobs)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n16) dst[i] = src[i];
    if (i < n_jobs) *(int32_t*)((char*)redo + (size_t)i * redo_stride) = 0;
}

// A batch's inlier records on their way to the host: pair p's first
// counts[2p+1] records go from src + p*stride to dst + p*stride, where dst is
// pinned host memory the device writes directly.  Only the records that exist
// travel; the host never has to know the counts before the copy is enqueued.
// grid = (slices, pairs).
__global__ void __launch_bounds__(256) records_to_host_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst,
                                                              size_t stride, const int32_t* __restrict__ counts)
{
    const int n = counts[2 * blockIdx.y + 1];
    const uint4* s = src + (size_t)blockIdx.y * stride;
    uint4* d = dst + (size_t)blockIdx.y * stride;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) d[i] = s[i];
}

// The points of a pair's selectMatches, as main.cpp:400-416 gathers them for the
// pose of the first pair: keypoints[.][m.queryIdx].pt and keypoints[.][m.trainIdx].pt
// widened to double.  grid = (tiles of 256 matches, pairs).
__global__ void __launch_bounds__(256) gather_inlier_points_kernel(const PairDev* __restrict__ pairs)
{
    const PairDev pd = pairs[blockIdx.y];
    const int n = *pd.n_inliers;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n || !pd.inliers) return;
    const sfm_gpu_dmatch m = pd.inliers[j];
    const float2 a = pd.kp_q[m.queryIdx];
    const float2 b = pd.kp_t[m.trainIdx];
    pd.px1[2 * j] = (double)a.x;
    pd.px1[2 * j + 1] = (double)a.y;
    pd.px2[2 * j] = (double)b.x;
    pd.px2[2 * j + 1] = (double)b.y;
}

}  // namespace

hipError_t launch_gather_inlier_points(hipStream_t s, const PairDev* pairs, int n_pairs, int n_query_max)
{
    const int tiles = std::max(1, (n_query_max + 255) / 256);
    hipLaunchKernelGGL(gather_inlier_points_kernel, dim3(tiles, n_pairs), dim3(256), 0, s, pairs);
    return hipGetLastError();
}

hipError_t launch_records_to_host(hipStream_t s, const sfm_gpu_dmatch* src, sfm_gpu_dmatch* dst_pinned, size_t stride,
                                  const int32_t* counts, int n_pairs)
{
    static_assert(sizeof(sfm_gpu_dmatch) == sizeof(uint4), "one record per 16-byte access");
    hipLaunchKernelGGL(records_to_host_kernel, dim3(4, n_pairs), dim3(256), 0, s, (const uint4*)src,
                       (uint4*)dst_pinned, stride, counts);
    return hipGetLastError();
}

hipError_t launch_upload_block(hipStream_t s, void* dst, const void* src_pinned, size_t bytes, int32_t* redo,
                               size_t redo_stride, int n_jobs)
{
    const int n16 = (int)((bytes + 15) / 16);
    const int n = std::max(n16, n_jobs);
    hipLaunchKernelGGL(upload_block_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (uint4*)dst,
                       (const uint4*)src_pinned, n16, redo, redo_stride, n_jobs);
    return hipGetLastError();
}

hipError_t launch_ratio_compact(hipStream_t s, const PairDev* pairs, int n_pairs, int n_query_max, double ratio)
{
    const int tiles = std::max(1, (n_query_max + 1023) / 1024);
    hipLaunchKernelGGL(ratio_compact_kernel, dim3(tiles, n_pairs), dim3(1024), 0, s, pairs, ratio);
    return hipGetLastError();
}

hipError_t launch_inlier_compact(hipStream_t s, const PairDev* pairs, int n_pairs, bool check_state)
{
    hipLaunchKernelGGL(inlier_compact_kernel, dim3(n_pairs), dim3(1024), 0, s, pairs, check_state ? 1 : 0);
    return hipGetLastError();
}

}  // namespace sfm
