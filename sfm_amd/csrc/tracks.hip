// Three-view track join: the nested loop of /root/reference/main.cpp:442-456
// (every inlier match m of image pair (i, i+1) against every inlier match n of
// pair (i+1, j), kept when matches[i][m].trainIdx == matches[i+1][n].queryIdx)
// as a sorted join.  match() emits the matches of a pair in ascending queryIdx
// (main.cpp:189-197 walks the k-NN rows in order and the inlier filter keeps
// the order), so the second list is searched by bisection instead of scanned:
// O((m + n) log n) instead of O(m n), same triples in the same order.
#include <cstring>

#include "sfm_ctx.h"

namespace sfm {

namespace {

// first index in [0, n) with q[idx].queryIdx >= key (n when there is none)
__device__ __forceinline__ int lower_bound_q(const sfm_gpu_dmatch* q, int n, int key)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (q[mid].queryIdx < key) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// One workgroup.  result[0] = number of triples, result[1] = 1 when bc is not
// in ascending queryIdx order (then nothing else is meaningful).
__global__ void __launch_bounds__(1024) track_join_kernel(const sfm_gpu_dmatch* __restrict__ ab, int n_ab,
                                                          const sfm_gpu_dmatch* __restrict__ bc, int n_bc,
                                                          int32_t* __restrict__ out, int capacity,
                                                          int32_t* __restrict__ result)
{
    __shared__ int wsum[16];
    __shared__ int s_base, s_bad;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) { s_base = 0; s_bad = 0; }
    __syncthreads();
    for (int n = tid + 1; n < n_bc; n += 1024)
        if (bc[n - 1].queryIdx > bc[n].queryIdx) s_bad = 1;
    __syncthreads();
    if (s_bad) {
        if (tid == 0) { result[0] = 0; result[1] = 1; }
        return;
    }
    for (int m0 = 0; m0 < n_ab; m0 += 1024) {
        const int m = m0 + tid;
        int lo = 0, cnt = 0, key = 0, qa = 0;
        if (m < n_ab) {
            key = ab[m].trainIdx;
            qa = ab[m].queryIdx;
            lo = lower_bound_q(bc, n_bc, key);
            int hi = lo;
            while (hi < n_bc && bc[hi].queryIdx == key) ++hi;  // runs of equal keys are short (normally 1)
            cnt = hi - lo;
        }
        // exclusive scan of cnt over the tile, in match order
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off);
            if (lane >= off) incl += o;
        }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        int pos = s_base + incl - cnt, total = 0;
        for (int w = 0; w < 16; ++w) {
            if (w < wv) pos += wsum[w];
            total += wsum[w];
        }
        for (int t = 0; t < cnt; ++t)
            if (pos + t < capacity) {
                out[3 * (pos + t) + 0] = qa;
                out[3 * (pos + t) + 1] = key;
                out[3 * (pos + t) + 2] = bc[lo + t].trainIdx;
            }
        __syncthreads();
        if (tid == 0) s_base += total;
        __syncthreads();
    }
    if (tid == 0) { result[0] = s_base; result[1] = 0; }
}

}  // namespace

}  // namespace sfm

using namespace sfm;

extern "C" int sfm_gpu_track_join(sfm_gpu_ctx* ctx, const sfm_gpu_dmatch* matches_ab, int n_ab,
                                  const sfm_gpu_dmatch* matches_bc, int n_bc, sfm_gpu_track3* tracks,
                                  int capacity, int32_t* n_tracks)
{
    if (!ctx) return SFM_GPU_ERR_ARG;
    if (n_ab < 0 || n_bc < 0 || capacity < 0 || !n_tracks || (n_ab > 0 && !matches_ab) ||
        (n_bc > 0 && !matches_bc) || (capacity > 0 && !tracks))
        return ctx->fail(SFM_GPU_ERR_ARG, "sfm_gpu_track_join: null pointer or negative size");
    *n_tracks = 0;
    if (n_ab == 0 || n_bc == 0) return SFM_GPU_OK;
    SFM_HIP(ctx, hipSetDevice(ctx->device));
    const size_t a_bytes = (size_t)n_ab * sizeof(sfm_gpu_dmatch), b_bytes = (size_t)n_bc * sizeof(sfm_gpu_dmatch);
    const size_t o_bytes = 16 + (size_t)capacity * sizeof(sfm_gpu_track3);
    SFM_HIP(ctx, ctx->pin_a.reserve(a_bytes + b_bytes));
    SFM_HIP(ctx, ctx->pin_b.reserve(o_bytes));
    SFM_HIP(ctx, ctx->pose_in.reserve(a_bytes + b_bytes));
    SFM_HIP(ctx, ctx->pose_out.reserve(o_bytes));
    char* hin = ctx->pin_a.as<char>();
    std::memcpy(hin, matches_ab, a_bytes);
    std::memcpy(hin + a_bytes, matches_bc, b_bytes);
    char* din = ctx->pose_in.as<char>();
    char* dout = ctx->pose_out.as<char>();
    SFM_HIP(ctx, hipMemcpyAsync(din, hin, a_bytes + b_bytes, hipMemcpyHostToDevice, ctx->stream));
    {
        ProfScope ps(ctx, "track_join");
        track_join_kernel<<<1, 1024, 0, ctx->stream>>>((const sfm_gpu_dmatch*)din, n_ab,
                                                       (const sfm_gpu_dmatch*)(din + a_bytes), n_bc,
                                                       (int32_t*)(dout + 16), capacity, (int32_t*)dout);
        SFM_HIP(ctx, hipGetLastError());
    }
    // the count first, then only as many triples as there are
    SFM_HIP(ctx, hipMemcpyAsync(ctx->pin_b.p, dout, 16, hipMemcpyDeviceToHost, ctx->stream));
    SFM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int32_t* res = ctx->pin_b.as<int32_t>();
    if (res[1])
        return ctx->fail(SFM_GPU_ERR_UNSUPPORTED,
                         "sfm_gpu_track_join: matches_bc is not in ascending queryIdx order (match() emits it so)");
    const int found = res[0];
    *n_tracks = found;
    if (found > capacity)
        return ctx->fail(SFM_GPU_ERR_CAPACITY, "sfm_gpu_track_join: %d tracks, capacity %d", found, capacity);
    if (found > 0) {
        SFM_HIP(ctx, hipMemcpyAsync(ctx->pin_b.as<char>() + 16, dout + 16, (size_t)found * sizeof(sfm_gpu_track3),
                                    hipMemcpyDeviceToHost, ctx->stream));
        SFM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        std::memcpy(tracks, ctx->pin_b.as<char>() + 16, (size_t)found * sizeof(sfm_gpu_track3));
    }
    return SFM_GPU_OK;
}
