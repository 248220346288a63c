// Pose recovery from an essential matrix: the work of
//   recoverPose(EssentialMat, pointsL, pointsR, K, R, T)
// at /root/reference/main.cpp:420 and /root/reference/bundle_adjustment.h:80
// (cv::decomposeEssentialMat + a cheirality vote over the correspondences).
//
// The 3x3 decomposition runs once per call on the host; the vote, whose cost
// grows with the number of correspondences, runs on the GPU with one thread per
// correspondence, each evaluating the four (R, t) candidates.
#include <cmath>
#include <cstring>

#include <vector>

#include "ransac_dev.h"
#include "sfm_ctx.h"
#include "sfm_gpu_debug.h"

namespace sfm {

struct PoseVoteArgs {
    double R[2][9];
    double t[3];
    double fx, fy, cx, cy;
    double dist;
};

// Votes of one correspondence: a = R (x1, y1, 1) and b = (x2, y2, 1) are the
// two viewing rays in the second camera's frame; |z1 a + t - z2 b|^2 is
// minimal at z1 = n1 / det, z2 = n2 / det (det >= 0 by Cauchy-Schwarz), the
// depths along the two rays.  The candidate gets the vote when both depths lie
// in (0, dist).  Written without the divisions; unfused double arithmetic.
__device__ __forceinline__ int cheirality_vote(const double* R, double t0, double t1, double t2,
                                               double x1, double y1, double x2, double y2,
                                               double bb, double dist)
{
    const double a0 = (R[0] * x1 + R[1] * y1) + R[2];
    const double a1 = (R[3] * x1 + R[4] * y1) + R[5];
    const double a2 = (R[6] * x1 + R[7] * y1) + R[8];
    const double aa = (a0 * a0 + a1 * a1) + a2 * a2;
    const double ab = (a0 * x2 + a1 * y2) + a2;
    const double at = (a0 * t0 + a1 * t1) + a2 * t2;
    const double bt = (x2 * t0 + y2 * t1) + t2;
    const double det = aa * bb - ab * ab;
    const double n1 = ab * bt - at * bb;
    const double n2 = aa * bt - ab * at;
    const double lim = dist * det;
    return det > 0 && n1 > 0 && n1 < lim && n2 > 0 && n2 < lim;
}

// flags[i] bit k = correspondence i votes for candidate k, in OpenCV's order
// (R1,t) (R2,t) (R1,-t) (R2,-t); votes[k] += number of set bits (zeroed by the
// caller).  Rows with mask_in[i] == 0 do not vote.
__global__ __launch_bounds__(256) void pose_vote_kernel(const double2* __restrict__ pts1,
                                                        const double2* __restrict__ pts2,
                                                        const uint8_t* __restrict__ mask_in, int n,
                                                        PoseVoteArgs pa, uint8_t* __restrict__ flags,
                                                        int32_t* __restrict__ votes)
{
    __shared__ int32_t s_votes[4];
    if (threadIdx.x < 4) s_votes[threadIdx.x] = 0;
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    unsigned f = 0;
    if (i < n && (!mask_in || mask_in[i])) {
        const double2 p = pts1[i], q = pts2[i];
        const double x1 = (p.x - pa.cx) / pa.fx, y1 = (p.y - pa.cy) / pa.fy;
        const double x2 = (q.x - pa.cx) / pa.fx, y2 = (q.y - pa.cy) / pa.fy;
        const double bb = (x2 * x2 + y2 * y2) + 1.0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const double sg = k < 2 ? 1.0 : -1.0;  // exact negation of t
            f |= (unsigned)cheirality_vote(pa.R[k & 1], sg * pa.t[0], sg * pa.t[1], sg * pa.t[2], x1,
                                           y1, x2, y2, bb, pa.dist)
                 << k;
        }
    }
    if (i < n) flags[i] = (uint8_t)f;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const unsigned long long b = __ballot((f >> k) & 1);
        if ((threadIdx.x & 63) == 0 && b) atomicAdd(&s_votes[k], __popcll(b));
    }
    __syncthreads();
    if (threadIdx.x < 4 && s_votes[threadIdx.x]) atomicAdd(&votes[threadIdx.x], s_votes[threadIdx.x]);
}

// The same vote for every pair of a sfm_gpu_match_pairs group at once: pair
// blockIdx.y votes with the correspondences its RANSAC kept (mask), read in the
// normalised form the essential-matrix problem already holds ((u-cx)/fx, the
// very values pose_vote_kernel computes from the pixels).  A pair whose
// args[].dist is negative has no model and is skipped.  votes: [pairs][4], zeroed
// by the caller.
// every_point != 0: every correspondence of the problem votes, whatever its mask
// says (recoverPose called without a mask, as at main.cpp:420).
__global__ __launch_bounds__(256) void pose_vote_pairs_kernel(const RansacProb* __restrict__ probs,
                                                              const PoseVoteArgs* __restrict__ args,
                                                              int32_t* __restrict__ votes, int every_point)
{
    __shared__ int32_t s_votes[4];
    const RansacProb pb = probs[blockIdx.y];
    const int n = *pb.n_ptr;
    if (blockIdx.x * 256 >= n) return;
    const PoseVoteArgs pa = args[blockIdx.y];
    if (pa.dist < 0) return;
    if (threadIdx.x < 4) s_votes[threadIdx.x] = 0;
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    unsigned f = 0;
    if (i < n && (every_point || pb.mask[i])) {
        const double x1 = pb.x1[2 * i], y1 = pb.x1[2 * i + 1], x2 = pb.x2[2 * i], y2 = pb.x2[2 * i + 1];
        const double bb = (x2 * x2 + y2 * y2) + 1.0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const double sg = k < 2 ? 1.0 : -1.0;
            f |= (unsigned)cheirality_vote(pa.R[k & 1], sg * pa.t[0], sg * pa.t[1], sg * pa.t[2], x1, y1, x2, y2,
                                           bb, pa.dist)
                 << k;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const unsigned long long b = __ballot((f >> k) & 1);
        if ((threadIdx.x & 63) == 0 && b) atomicAdd(&s_votes[k], __popcll(b));
    }
    __syncthreads();
    if (threadIdx.x < 4 && s_votes[threadIdx.x]) atomicAdd(&votes[4 * blockIdx.y + threadIdx.x], s_votes[threadIdx.x]);
}

// models[p][9] = the model the selection kernel left for problem p
__global__ __launch_bounds__(64) void gather_models_kernel(const RansacProb* __restrict__ probs, double* __restrict__ models)
{
    if (threadIdx.x < 9) models[9 * blockIdx.x + threadIdx.x] = probs[blockIdx.x].model_out[threadIdx.x];
}

// A = U diag(S) V^T of a row-major 3x3 matrix by one-sided Jacobi rotations of
// the columns (rotations accumulated in V), columns ordered by descending
// length, third left vector = u1 x u2 so that det U = +1 and a rank-2 matrix
// (an essential matrix) needs no division by its zero singular value.
// Returns false when the second singular value is below 2^-40 of the first.
static bool svd3_rank2(const double* A, double* U, double* V)
{
    double G[9], W[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    std::memcpy(G, A, sizeof(G));
    for (int sweep = 0; sweep < 30; sweep++) {
        bool rotated = false;
        for (int pq = 0; pq < 3; pq++) {
            const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
            double alpha = 0, beta = 0, gamma = 0;
            alpha = (G[p] * G[p] + G[3 + p] * G[3 + p]) + G[6 + p] * G[6 + p];
            beta = (G[q] * G[q] + G[3 + q] * G[3 + q]) + G[6 + q] * G[6 + q];
            gamma = (G[p] * G[q] + G[3 + p] * G[3 + q]) + G[6 + p] * G[6 + q];
            if (!(std::fabs(gamma) > 0x1p-52 * std::sqrt(alpha * beta))) continue;
            rotated = true;
            const double zeta = (beta - alpha) / (2.0 * gamma);
            const double tt = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
            const double c = 1.0 / std::sqrt(1.0 + tt * tt);
            const double s = c * tt;
            for (int r = 0; r < 3; r++) {
                double* g = G + 3 * r;
                double* w = W + 3 * r;
                const double gp = g[p], gq = g[q], wp = w[p], wq = w[q];
                g[p] = c * gp - s * gq;
                g[q] = s * gp + c * gq;
                w[p] = c * wp - s * wq;
                w[q] = s * wp + c * wq;
            }
        }
        if (!rotated) break;
    }
    double len[3];
    for (int j = 0; j < 3; j++) len[j] = std::sqrt((G[j] * G[j] + G[3 + j] * G[3 + j]) + G[6 + j] * G[6 + j]);
    int ord[3] = {0, 1, 2};
    auto sift = [&](int a, int b) {
        if (len[ord[b]] > len[ord[a]]) { const int tmp = ord[a]; ord[a] = ord[b]; ord[b] = tmp; }
    };
    sift(0, 1);
    sift(1, 2);
    sift(0, 1);
    if (!(len[ord[1]] > 0x1p-40 * len[ord[0]])) return false;  // numerically rank < 2
    for (int j = 0; j < 3; j++)
        for (int r = 0; r < 3; r++) V[3 * r + j] = W[3 * r + ord[j]];
    for (int j = 0; j < 2; j++)
        for (int r = 0; r < 3; r++) U[3 * r + j] = G[3 * r + ord[j]] / len[ord[j]];
    U[2] = U[3] * U[7] - U[6] * U[4];
    U[5] = U[6] * U[1] - U[0] * U[7];
    U[8] = U[0] * U[4] - U[3] * U[1];
    return true;
}

// cv::decomposeEssentialMat: R1 = U W V^T, R2 = U W^T V^T, t = third column of
// U, W = [0 1 0; -1 0 0; 0 0 1], both orthogonal factors made proper first.
static bool decompose_essential(const double* E, double* R1, double* R2, double* t)
{
    double U[9], V[9];
    if (!svd3_rank2(E, U, V)) return false;
    const double detV = (V[0] * (V[4] * V[8] - V[5] * V[7]) - V[1] * (V[3] * V[8] - V[5] * V[6])) +
                        V[2] * (V[3] * V[7] - V[4] * V[6]);
    if (detV < 0)
        for (int i = 0; i < 9; i++) V[i] = -V[i];
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) {
            const double a = U[3 * i + 0] * V[3 * j + 1] - U[3 * i + 1] * V[3 * j + 0];
            const double b = U[3 * i + 2] * V[3 * j + 2];
            R1[3 * i + j] = a + b;
            R2[3 * i + j] = b - a;
        }
        t[i] = U[3 * i + 2];
    }
    return true;
}

// OpenCV's order of preference: the first candidate not beaten by another.
static int best_candidate(const int32_t* votes)
{
    for (int k = 0; k < 4; k++) {
        bool top = true;
        for (int j = 0; j < 4; j++) top = top && votes[k] >= votes[j];
        if (top) return k;
    }
    return 3;
}

// recoverPose for the g pairs of the group sfm_gpu_match_pairs has just finished
// (stream drained): each pair's RANSAC model against that pair's inliers.
int pose_for_pairs(sfm_gpu_ctx* ctx, const RansacProb* dprobs, int g, const int32_t* n_inliers, const double* K,
                   double distance_thresh, double* models_out, double* R_out, double* t_out, int32_t* n_pose_good,
                   int nq_max, bool every_point)
{
    hipStream_t s = ctx->stream;
    const size_t args_b = sizeof(PoseVoteArgs) * (size_t)g;
    const size_t dev_b = args_b + (size_t)g * 72 + (size_t)g * 16;
    SFM_HIP(ctx, ctx->pose_in.reserve(dev_b));
    SFM_HIP(ctx, ctx->pin_a.reserve(dev_b));
    char* dbase = ctx->pose_in.as<char>();
    PoseVoteArgs* dargs = (PoseVoteArgs*)dbase;
    double* dmodels = (double*)(dbase + args_b);
    int32_t* dvotes = (int32_t*)(dbase + args_b + (size_t)g * 72);
    char* hbase = ctx->pin_a.as<char>();
    PoseVoteArgs* hargs = (PoseVoteArgs*)hbase;
    double* hmodels = (double*)(hbase + args_b);
    int32_t* hvotes = (int32_t*)(hbase + args_b + (size_t)g * 72);
    gather_models_kernel<<<g, 64, 0, s>>>(dprobs, dmodels);
    SFM_HIP(ctx, hipGetLastError());
    SFM_HIP(ctx, hipMemcpyAsync(hmodels, dmodels, (size_t)g * 72, hipMemcpyDeviceToHost, s));
    SFM_HIP(ctx, hipStreamSynchronize(s));
    for (int i = 0; i < g; ++i) {
        PoseVoteArgs& pa = hargs[i];
        std::memset(&pa, 0, sizeof(pa));
        pa.dist = -1.0;
        if (n_inliers[i] <= 0) {
            std::memset(hmodels + 9 * i, 0, 72);
            continue;
        }
        if (!decompose_essential(hmodels + 9 * i, pa.R[0], pa.R[1], pa.t)) continue;
        pa.fx = K[0];
        pa.fy = K[4];
        pa.cx = K[2];
        pa.cy = K[5];
        pa.dist = distance_thresh;
    }
    SFM_HIP(ctx, hipMemcpyAsync(dargs, hargs, args_b, hipMemcpyHostToDevice, s));
    SFM_HIP(ctx, hipMemsetAsync(dvotes, 0, (size_t)g * 16, s));
    {
        ProfScope ps(ctx, "pose_vote");
        pose_vote_pairs_kernel<<<dim3((nq_max + 255) / 256 > 0 ? (nq_max + 255) / 256 : 1, g), 256, 0, s>>>(dprobs, dargs, dvotes, every_point ? 1 : 0);
        SFM_HIP(ctx, hipGetLastError());
    }
    SFM_HIP(ctx, hipMemcpyAsync(hvotes, dvotes, (size_t)g * 16, hipMemcpyDeviceToHost, s));
    SFM_HIP(ctx, hipStreamSynchronize(s));
    for (int i = 0; i < g; ++i) {
        const PoseVoteArgs& pa = hargs[i];
        if (models_out) std::memcpy(models_out + 9 * i, hmodels + 9 * i, 72);
        if (pa.dist < 0) {
            std::memset(R_out + 9 * i, 0, 72);
            std::memset(t_out + 3 * i, 0, 24);
            n_pose_good[i] = -1;
            continue;
        }
        const int best = best_candidate(hvotes + 4 * i);
        std::memcpy(R_out + 9 * i, pa.R[best & 1], 72);
        for (int k = 0; k < 3; k++) t_out[3 * i + k] = best < 2 ? pa.t[k] : -pa.t[k];
        n_pose_good[i] = hvotes[4 * i + best];
    }
    return SFM_GPU_OK;
}

}  // namespace sfm

using namespace sfm;

extern "C" int sfm_gpu_recover_pose(sfm_gpu_ctx* ctx, const double* E, const double* pts1,
                                    const double* pts2, int n, const double* K,
                                    double distance_thresh, double* R_out, double* t_out,
                                    uint8_t* mask, int32_t* n_good)
{
    if (!ctx) return SFM_GPU_ERR_ARG;
    if (!E || !K || !R_out || !t_out || n < 0 || (n > 0 && (!pts1 || !pts2)))
        return ctx->fail(SFM_GPU_ERR_ARG, "sfm_gpu_recover_pose: null pointer or negative size");
    if (!(K[0] != 0) || !(K[4] != 0))
        return ctx->fail(SFM_GPU_ERR_ARG, "sfm_gpu_recover_pose: zero focal length");
    PoseVoteArgs pa;
    if (!decompose_essential(E, pa.R[0], pa.R[1], pa.t))
        return ctx->fail(SFM_GPU_ERR_ARG, "sfm_gpu_recover_pose: E has rank < 2");
    pa.fx = K[0];
    pa.fy = K[4];
    pa.cx = K[2];
    pa.cy = K[5];
    pa.dist = distance_thresh;

    int32_t votes[4] = {0, 0, 0, 0};
    const uint8_t* flags_host = nullptr;
    if (n > 0) {
        SFM_HIP(ctx, hipSetDevice(ctx->device));
        // staging: [pts1 | pts2 | mask] up, [votes (16 B) | flags] down
        const size_t pbytes = (size_t)n * 16;
        const size_t in_bytes = 2 * pbytes + (mask ? (size_t)n : 0);
        const size_t out_bytes = 16 + (size_t)n;
        SFM_HIP(ctx, ctx->pin_a.reserve(in_bytes));
        SFM_HIP(ctx, ctx->pin_b.reserve(out_bytes));
        SFM_HIP(ctx, ctx->pose_in.reserve(in_bytes));
        SFM_HIP(ctx, ctx->pose_out.reserve(out_bytes));
        char* hin = ctx->pin_a.as<char>();
        std::memcpy(hin, pts1, pbytes);
        std::memcpy(hin + pbytes, pts2, pbytes);
        if (mask) std::memcpy(hin + 2 * pbytes, mask, (size_t)n);
        char* din = ctx->pose_in.as<char>();
        char* dout = ctx->pose_out.as<char>();
        SFM_HIP(ctx, hipMemcpyAsync(din, hin, in_bytes, hipMemcpyHostToDevice, ctx->stream));
        SFM_HIP(ctx, hipMemsetAsync(dout, 0, 16, ctx->stream));
        {
            ProfScope ps(ctx, "pose_vote");
            pose_vote_kernel<<<(n + 255) / 256, 256, 0, ctx->stream>>>(
                (const double2*)din, (const double2*)(din + pbytes),
                mask ? (const uint8_t*)(din + 2 * pbytes) : nullptr, n, pa, (uint8_t*)(dout + 16),
                (int32_t*)dout);
            SFM_HIP(ctx, hipGetLastError());
        }
        SFM_HIP(ctx, hipMemcpyAsync(ctx->pin_b.p, dout, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
        SFM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        std::memcpy(votes, ctx->pin_b.p, sizeof(votes));
        flags_host = ctx->pin_b.as<uint8_t>() + 16;
    }
    const int best = best_candidate(votes);
    std::memcpy(R_out, pa.R[best & 1], 9 * sizeof(double));
    for (int i = 0; i < 3; i++) t_out[i] = best < 2 ? pa.t[i] : -pa.t[i];
    if (mask)
        for (int i = 0; i < n; i++) mask[i] = (uint8_t)((flags_host[i] >> best) & 1);
    if (n_good) *n_good = votes[best];
    std::memcpy(ctx->dbg_pose_votes, votes, sizeof(votes));
    return SFM_GPU_OK;
}

extern "C" int sfm_gpu_debug_pose_votes(sfm_gpu_ctx* ctx, int32_t* votes4)
{
    if (!ctx || !votes4) return SFM_GPU_ERR_ARG;
    std::memcpy(votes4, ctx->dbg_pose_votes, sizeof(ctx->dbg_pose_votes));
    return SFM_GPU_OK;
}
