// Three-view track join: the nested loop of /root/reference/main.cpp:442-456
// (every inlier match m of image pair (i, i+1) against every inlier match n of
// pair (i+1, j), kept when matches[i][m].trainIdx == matches[i+1][n].queryIdx)
// as a sorted join.  match() emits the matches of a pair in ascending queryIdx
// (main.cpp:189-197 walks the k-NN rows in order and the inlier filter keeps
// the order), so the second list is searched by bisection instead of scanned:
// O((m + n) log n) instead of O(m n), same triples in the same order.
#include <cstring>
#include <vector>

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

// One workgroup per join j of list j (any order) with list j+1 (ascending
// queryIdx).  Two passes over the same search: COUNT stores the number of
// triples of the join in counts[j] (and sets *bad when list j+1 is out of
// order), WRITE stores the triples from offsets[j] on, which a scan of the
// counts has set in between, as far as `capacity` triples in all.
// lists: the match lists back to back; starts[l], sizes[l]: list l.
template <bool WRITE>
__global__ void __launch_bounds__(1024) track_join_kernel(const sfm_gpu_dmatch* __restrict__ lists,
                                                          const int32_t* __restrict__ starts,
                                                          const int32_t* __restrict__ sizes,
                                                          int32_t* __restrict__ counts, int32_t* __restrict__ bad,
                                                          const int32_t* __restrict__ offsets,
                                                          int32_t* __restrict__ out, int capacity)
{
    __shared__ int wsum[16];
    __shared__ int s_base, s_bad;
    const int j = blockIdx.x;
    const sfm_gpu_dmatch* ab = lists + starts[j];
    const sfm_gpu_dmatch* bc = lists + starts[j + 1];
    const int n_ab = sizes[j], n_bc = sizes[j + 1];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) { s_base = WRITE ? offsets[j] : 0; s_bad = 0; }
    __syncthreads();
    if (!WRITE) {
        for (int n = tid + 1; n < n_bc; n += 1024)
            if (bc[n - 1].queryIdx > bc[n].queryIdx) s_bad = 1;
        __syncthreads();
        if (s_bad) {
            if (tid == 0) { counts[j] = 0; *bad = 1; }
            return;
        }
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
        if (WRITE)
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
    if (!WRITE && tid == 0) counts[j] = s_base;
}

// offsets[0..n] = exclusive prefix sums of counts[0..n), one workgroup
__global__ void __launch_bounds__(1024) track_offsets_kernel(const int32_t* __restrict__ counts, int n,
                                                             int32_t* __restrict__ offsets)
{
    __shared__ int wsum[16];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int j0 = 0; j0 < n; j0 += 1024) {
        const int j = j0 + tid;
        const int c = j < n ? counts[j] : 0;
        int incl = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off);
            if (lane >= off) incl += o;
        }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        int pos = s_base + incl - c, total = 0;
        for (int w = 0; w < 16; ++w) {
            if (w < wv) pos += wsum[w];
            total += wsum[w];
        }
        if (j < n) offsets[j] = pos;
        __syncthreads();
        if (tid == 0) s_base += total;
        __syncthreads();
    }
    if (tid == 0) offsets[n] = s_base;
}

// Joins list j with list j+1 for j = 0..n_lists-2.  list_ptr[l] / sizes[l]: the
// host lists.  offsets_out[n_lists]: join j's triples are tracks[offsets[j] ..
// offsets[j+1]) (offsets_out[n_lists-1] = their total).
int join_lists(sfm_gpu_ctx* ctx, const sfm_gpu_dmatch* const* list_ptr, const int32_t* sizes, int n_lists,
               sfm_gpu_track3* tracks, int capacity, int32_t* offsets_out, const char* who)
{
    const int n_joins = n_lists - 1;
    size_t total = 0;
    for (int l = 0; l < n_lists; ++l) total += (size_t)sizes[l];
    if (total > 0x7FFFFFFFu) return ctx->fail(SFM_GPU_ERR_CAPACITY, "%s: %zu matches in all", who, total);
    SFM_HIP(ctx, hipSetDevice(ctx->device));
    // up: [starts | sizes] (2*n_lists ints, padded to 16 bytes), then the records
    const size_t head = ((size_t)2 * n_lists * 4 + 15) & ~(size_t)15;
    const size_t in_bytes = head + total * sizeof(sfm_gpu_dmatch);
    // device out: [bad | pad][counts n_joins][offsets n_lists] then the triples
    const size_t ohead = ((size_t)(4 + n_joins + n_lists) * 4 + 15) & ~(size_t)15;
    const size_t o_bytes = ohead + (size_t)capacity * sizeof(sfm_gpu_track3);
    SFM_HIP(ctx, ctx->pin_a.reserve(in_bytes));
    SFM_HIP(ctx, ctx->pin_b.reserve(o_bytes));
    SFM_HIP(ctx, ctx->pose_in.reserve(in_bytes));
    SFM_HIP(ctx, ctx->pose_out.reserve(o_bytes));
    char* hin = ctx->pin_a.as<char>();
    int32_t* hstarts = (int32_t*)hin;
    int32_t* hsizes = hstarts + n_lists;
    size_t at = 0;
    for (int l = 0; l < n_lists; ++l) {
        hstarts[l] = (int32_t)at;
        hsizes[l] = sizes[l];
        if (sizes[l] > 0)
            std::memcpy(hin + head + at * sizeof(sfm_gpu_dmatch), list_ptr[l], (size_t)sizes[l] * sizeof(sfm_gpu_dmatch));
        at += (size_t)sizes[l];
    }
    char* din = ctx->pose_in.as<char>();
    char* dout = ctx->pose_out.as<char>();
    int32_t* d_bad = (int32_t*)dout;
    int32_t* d_counts = d_bad + 4;
    int32_t* d_offsets = d_counts + n_joins;
    int32_t* d_tracks = (int32_t*)(dout + ohead);
    const sfm_gpu_dmatch* d_lists = (const sfm_gpu_dmatch*)(din + head);
    const int32_t* d_starts = (const int32_t*)din;
    const int32_t* d_sizes = d_starts + n_lists;
    SFM_HIP(ctx, hipMemcpyAsync(din, hin, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    SFM_HIP(ctx, hipMemsetAsync(d_bad, 0, 16, ctx->stream));
    {
        ProfScope ps(ctx, "track_join");
        track_join_kernel<false><<<n_joins, 1024, 0, ctx->stream>>>(d_lists, d_starts, d_sizes, d_counts, d_bad,
                                                                    nullptr, nullptr, 0);
        track_offsets_kernel<<<1, 1024, 0, ctx->stream>>>(d_counts, n_joins, d_offsets);
        track_join_kernel<true><<<n_joins, 1024, 0, ctx->stream>>>(d_lists, d_starts, d_sizes, d_counts, d_bad,
                                                                   d_offsets, d_tracks, capacity);
        SFM_HIP(ctx, hipGetLastError());
    }
    // flags and offsets first, then only as many triples as there are
    SFM_HIP(ctx, hipMemcpyAsync(ctx->pin_b.p, dout, ohead, hipMemcpyDeviceToHost, ctx->stream));
    SFM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int32_t* res = ctx->pin_b.as<int32_t>();
    if (res[0])
        return ctx->fail(SFM_GPU_ERR_UNSUPPORTED,
                         "%s: a second list is not in ascending queryIdx order (match() emits it so)", who);
    const int32_t* off = res + 4 + n_joins;
    std::memcpy(offsets_out, off, (size_t)n_lists * 4);
    const int found = off[n_joins];
    if (found > capacity) return ctx->fail(SFM_GPU_ERR_CAPACITY, "%s: %d tracks, capacity %d", who, found, capacity);
    if (found > 0) {
        SFM_HIP(ctx, hipMemcpyAsync(ctx->pin_b.as<char>() + ohead, d_tracks, (size_t)found * sizeof(sfm_gpu_track3),
                                    hipMemcpyDeviceToHost, ctx->stream));
        SFM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        std::memcpy(tracks, ctx->pin_b.as<char>() + ohead, (size_t)found * sizeof(sfm_gpu_track3));
    }
    return SFM_GPU_OK;
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
    const sfm_gpu_dmatch* lists[2] = {matches_ab, matches_bc};
    const int32_t sizes[2] = {n_ab, n_bc};
    int32_t offsets[2] = {0, 0};
    const int rc = join_lists(ctx, lists, sizes, 2, tracks, capacity, offsets, "sfm_gpu_track_join");
    *n_tracks = offsets[1];
    return rc;
}

extern "C" int sfm_gpu_track_join_sequence(sfm_gpu_ctx* ctx, const sfm_gpu_dmatch* matches,
                                           const int32_t* n_matches, int match_capacity, int n_lists,
                                           sfm_gpu_track3* tracks, int track_capacity, int32_t* track_offsets)
{
    if (!ctx) return SFM_GPU_ERR_ARG;
    if (n_lists < 0 || match_capacity < 0 || track_capacity < 0 || !track_offsets ||
        (n_lists > 0 && (!n_matches || (match_capacity > 0 && !matches))) || (track_capacity > 0 && !tracks))
        return ctx->fail(SFM_GPU_ERR_ARG, "sfm_gpu_track_join_sequence: null pointer or negative size");
    for (int l = 0; l < n_lists; ++l) {
        track_offsets[l] = 0;
        if (n_matches[l] < 0 || n_matches[l] > match_capacity)
            return ctx->fail(SFM_GPU_ERR_ARG, "sfm_gpu_track_join_sequence: list %d has %d matches, capacity %d", l,
                             n_matches[l], match_capacity);
    }
    if (n_lists < 2) return SFM_GPU_OK;
    if (n_lists > (1 << 20)) return ctx->fail(SFM_GPU_ERR_CAPACITY, "sfm_gpu_track_join_sequence: %d lists", n_lists);
    std::vector<const sfm_gpu_dmatch*> lists((size_t)n_lists);
    for (int l = 0; l < n_lists; ++l) lists[l] = matches + (size_t)l * match_capacity;
    return join_lists(ctx, lists.data(), n_matches, n_lists, tracks, track_capacity, track_offsets,
                      "sfm_gpu_track_join_sequence");
}
