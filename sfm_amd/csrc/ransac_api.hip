// Host side of the RANSAC path and of the pair-list driver.
//
// sfm_gpu_ransac     <- findEssentialMat(...)            /root/reference/main.cpp:224
// sfm_gpu_match_pairs<- the loop body of match()          /root/reference/main.cpp:175-231, 267
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

#include "ransac_dev.h"
#include "sfm_gpu_debug.h"
#include "sfm_ctx.h"

using namespace sfm;

namespace {

int cv_round(double v) { return (int)std::nearbyint(v); }

// cv::RANSACUpdateNumIters (OpenCV 4.5.5 calib3d ptsetreg.cpp).
int update_num_iters(double p, double ep, int model_points, int max_iters)
{
    p = std::max(p, 0.);
    p = std::min(p, 1.);
    ep = std::max(ep, 0.);
    ep = std::min(ep, 1.);
    double num = std::max(1. - p, DBL_MIN);
    double denom = 1. - std::pow(1. - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = std::log(num);
    denom = std::log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : cv_round(num / denom);
}

struct RansacParams {
    int model;
    int mode;
    double K[9];
    double prob;
    double threshold_px;
    int max_iters;
    int n_hyp;
};

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Bytes of per-problem RANSAC workspace for `cap` points and `hyp_cap` hypotheses.
size_t ransac_ws_bytes(int cap, int hyp_cap, int m)
{
    size_t b = 0;
    b += align_up((size_t)cap * 16, 256) * 2;            // x1, x2
    b += align_up((size_t)cap * 8, 256) * 2;             // f1, f2
    b += align_up(((size_t)cap + 7) / 4 * 64, 256);      // soa: blocks of four points
    b += align_up((size_t)hyp_cap * m * 4, 256);         // table
    b += align_up((size_t)hyp_cap * 90 * 8, 256);        // models
    b += align_up((size_t)hyp_cap * 4, 256);             // nmodels
    b += align_up((size_t)hyp_cap * 10 * 4, 256);        // counts
    b += align_up((size_t)hyp_cap * 10 * 4, 256) * 2;    // list, list2
    b += align_up((size_t)cap, 256);                     // mask
    b += 256;                                            // model_out
    return b;
}

void carve_ransac(RansacProb& pb, char* base, int cap, int hyp_cap, int m)
{
    char* p = base;
    auto take = [&](size_t bytes) { char* r = p; p += align_up(bytes, 256); return r; };
    pb.x1 = (double*)take((size_t)cap * 16);
    pb.x2 = (double*)take((size_t)cap * 16);
    pb.f1 = (float*)take((size_t)cap * 8);
    pb.f2 = (float*)take((size_t)cap * 8);
    pb.soa = (float*)take(((size_t)cap + 7) / 4 * 64);
    pb.table = (int32_t*)take((size_t)hyp_cap * m * 4);
    pb.models = (double*)take((size_t)hyp_cap * 90 * 8);
    pb.nmodels = (int32_t*)take((size_t)hyp_cap * 4);
    pb.counts = (int32_t*)take((size_t)hyp_cap * 10 * 4);
    pb.list = (int32_t*)take((size_t)hyp_cap * 10 * 4);
    pb.list2 = (int32_t*)take((size_t)hyp_cap * 10 * 4);
    pb.mask = (uint8_t*)take((size_t)cap);
    pb.model_out = (double*)take(256);
}

int fill_params(sfm_gpu_ctx* ctx, RansacProb& pb, const RansacParams& rp)
{
    if (rp.model != SFM_GPU_MODEL_ESSENTIAL && rp.model != SFM_GPU_MODEL_FUNDAMENTAL)
        return ctx->fail(SFM_GPU_ERR_UNSUPPORTED, "unknown RANSAC model %d", rp.model);
    if (rp.mode != SFM_GPU_RANSAC_ADAPTIVE && rp.mode != SFM_GPU_RANSAC_FIXED)
        return ctx->fail(SFM_GPU_ERR_UNSUPPORTED, "unknown RANSAC mode %d", rp.mode);
    if (!(rp.prob > 0 && rp.prob < 1)) return ctx->fail(SFM_GPU_ERR_ARG, "prob must be in (0,1)");
    if (rp.mode == SFM_GPU_RANSAC_FIXED && rp.n_hyp < 1)
        return ctx->fail(SFM_GPU_ERR_ARG, "n_hyp must be >= 1 in fixed mode");
    if (rp.mode == SFM_GPU_RANSAC_FIXED && rp.n_hyp > SFM_GPU_MAX_HYPOTHESES)
        return ctx->fail(SFM_GPU_ERR_CAPACITY, "n_hyp %d exceeds SFM_GPU_MAX_HYPOTHESES (%d)", rp.n_hyp,
                         SFM_GPU_MAX_HYPOTHESES);
    if (rp.mode == SFM_GPU_RANSAC_ADAPTIVE && rp.max_iters > SFM_GPU_MAX_HYPOTHESES)
        return ctx->fail(SFM_GPU_ERR_CAPACITY, "max_iters %d exceeds SFM_GPU_MAX_HYPOTHESES (%d)", rp.max_iters,
                         SFM_GPU_MAX_HYPOTHESES);
    if (!(rp.threshold_px > 0) || !std::isfinite(rp.threshold_px))
        return ctx->fail(SFM_GPU_ERR_ARG, "threshold_px must be positive and finite");
    if (rp.model == SFM_GPU_MODEL_ESSENTIAL) {
        const double f = (rp.K[0] + rp.K[4]) / 2;
        if (!std::isfinite(f) || f == 0.0 || rp.K[0] == 0.0 || rp.K[4] == 0.0 || !std::isfinite(rp.K[2]) ||
            !std::isfinite(rp.K[5]))
            return ctx->fail(SFM_GPU_ERR_ARG, "camera matrix needs finite, non-zero focal lengths");
    }
    pb.model = rp.model;
    pb.m = rp.model == SFM_GPU_MODEL_ESSENTIAL ? 5 : 7;
    double thr = rp.threshold_px;
    pb.fx = pb.fy = 1.0;
    pb.cx = pb.cy = 0.0;
    if (rp.model == SFM_GPU_MODEL_ESSENTIAL) {
        pb.fx = rp.K[0];
        pb.fy = rp.K[4];
        pb.cx = rp.K[2];
        pb.cy = rp.K[5];
        thr /= (pb.fx + pb.fy) / 2;  // findEssentialMat: threshold /= (fx+fy)/2
    }
    pb.tf = (float)(thr * thr);
    pb.tlo = (double)pb.tf * (1.0 - 1.0 / 1048576.0);
    pb.thi = (double)pb.tf * (1.0 + 1.0 / 1048576.0);
    return SFM_GPU_OK;
}

// Runs RANSAC for n_prob problems whose descriptors (with every device pointer
// set) are in `probs`.  `results` receives n_prob * R_COUNT ints.  On return
// the masks and models are on the device; the stream has been synchronised.
// optimistic (adaptive mode): only the first batch of sample rows is enqueued,
// nothing is waited for, and `results` stays empty: a problem whose loop wants
// more rows than that is left with R_STATE_DONE clear and R_OK zero, for the
// caller to find and to run again without this flag.
int ransac_batch(sfm_gpu_ctx* ctx, std::vector<RansacProb>& probs, int cap, int hyp_cap,
                 const RansacParams& rp, const int32_t* custom_table_host, std::vector<int32_t>& results,
                 bool need_results, RansacProb* dprobs_ready = nullptr, bool points_ready = false,
                 const PairDev* dpairs = nullptr, bool* pairs_done = nullptr, bool optimistic = false)
{
    const int n_prob = (int)probs.size();
    const int m = probs[0].m;
    hipStream_t s = ctx->stream;
    RansacProb* dprobs = dprobs_ready;  // the caller may have uploaded `probs` with its own descriptors
    if (!dprobs) {
        SFM_HIP(ctx, ctx->rs_misc.reserve(sizeof(RansacProb) * (size_t)n_prob));
        dprobs = ctx->rs_misc.as<RansacProb>();
        SFM_HIP(ctx, ctx->upload(dprobs, probs.data(), sizeof(RansacProb) * n_prob));
    }
    if (!points_ready) {  // (the pair pipeline's ratio kernel prepares the points itself)
        ProfScope ps(ctx, "ransac_prep");
        SFM_HIP(ctx, launch_ransac_prep(s, dprobs, n_prob, cap));
    }
    results.assign((size_t)n_prob * R_COUNT, 0);
    // the result blocks of a batch are contiguous (probs[p].result = base + p * R_COUNT)
    auto fetch_results = [&]() -> int {
        const size_t bytes = (size_t)n_prob * R_COUNT * 4;
        SFM_HIP(ctx, ctx->pin_a.reserve(bytes));
        SFM_HIP(ctx, hipMemcpyAsync(ctx->pin_a.p, probs[0].result, bytes, hipMemcpyDeviceToHost, s));
        SFM_HIP(ctx, hipStreamSynchronize(s));
        std::memcpy(results.data(), ctx->pin_a.p, bytes);
        return SFM_GPU_OK;
    };

    if (rp.mode == SFM_GPU_RANSAC_FIXED) {
        const int nh = rp.n_hyp;
        if (custom_table_host) {
            // caller-supplied table (single problem, n > m checked by the caller)
            SFM_HIP(ctx, hipMemcpyAsync(probs[0].table, custom_table_host, (size_t)nh * m * 4,
                                        hipMemcpyHostToDevice, s));
            const int32_t valid = nh;
            SFM_HIP(ctx, hipMemcpyAsync(probs[0].result + R_VALID, &valid, 4, hipMemcpyHostToDevice, s));
        } else if (rp.model != SFM_GPU_MODEL_FUNDAMENTAL) {
            ProfScope ps(ctx, "ransac_table");
            SFM_HIP(ctx, launch_ransac_table_fixed(s, dprobs, n_prob, nh));
        }
        {
            // the seven-point solver draws its own rows unless the caller brought a table
            const int gen = (!custom_table_host && rp.model == SFM_GPU_MODEL_FUNDAMENTAL) ? nh : 0;
            ProfScope ps(ctx, "ransac_solve");
            SFM_HIP(ctx, launch_ransac_solve(s, dprobs, n_prob, rp.model, 0, nh, gen, ctx->solver_split));
        }
        {
            // staged scoring with an exact bound; the stages are sized per problem
            // on the device (ransac_plan_kernel), nothing here depends on earlier calls
            ProfScope ps(ctx, "ransac_score");
            SFM_HIP(ctx, launch_ransac_score(s, dprobs, n_prob, cap, rp.model, 0, nh, true, ctx->score_filter,
                                             ctx->score_prune));
        }
        {
            // winner, then its mask and model copy, by the same workgroup per problem
            ProfScope ps(ctx, "ransac_select");
            SFM_HIP(ctx, launch_ransac_select(s, dprobs, n_prob, nh, dpairs));
            if (pairs_done) *pairs_done = dpairs != nullptr;
        }
        // the whole fixed-count path runs without the host; callers that keep
        // working on the device need not wait here
        return need_results ? fetch_results() : SFM_GPU_OK;
    }

    // Adaptive mode: hypotheses are scored in batches on the GPU and OpenCV's
    // sequential loop (RANSACPointSetRegistrator::run) is replayed over the
    // counts by ransac_replay_kernel, which reproduces its order-dependent early
    // termination exactly.  Between batches the host only looks at the problems'
    // result blocks: who is done, and how many more rows the others still want.
    const int max_it = std::max(rp.max_iters, 1);
    int done_rows = 0;
    int want = max_it;
    while (done_rows < max_it && want > 0) {
        int b = done_rows == 0 ? 32 : 128;
        b = std::min(b, std::min(want, max_it - done_rows));
        {
            ProfScope ps(ctx, "ransac_table");
            SFM_HIP(ctx, launch_ransac_table(s, dprobs, n_prob, done_rows, b));
        }
        {
            ProfScope ps(ctx, "ransac_solve");
            SFM_HIP(ctx, launch_ransac_solve(s, dprobs, n_prob, rp.model, done_rows, b, 0, ctx->solver_split));
        }
        {
            ProfScope ps(ctx, "ransac_score");
            SFM_HIP(ctx, launch_ransac_score(s, dprobs, n_prob, cap, rp.model, done_rows, b, false, ctx->score_filter,
                                             ctx->score_prune));
        }
        SFM_HIP(ctx, launch_ransac_replay(s, dprobs, n_prob, done_rows, b, max_it, rp.prob));
        done_rows += b;
        if (optimistic) break;  // the caller looks at the states later
        const int rc = fetch_results();
        if (rc) return rc;
        want = 0;
        for (int p = 0; p < n_prob; ++p) {
            const int32_t* r = &results[(size_t)p * R_COUNT];
            if (!(r[R_STATE] & R_STATE_DONE)) want = std::max(want, r[R_NITERS] - done_rows);
        }
    }
    {
        ProfScope ps(ctx, "ransac_mask");
        SFM_HIP(ctx, launch_ransac_mask(s, dprobs, n_prob, cap));
    }
    if (optimistic) return SFM_GPU_OK;
    SFM_HIP(ctx, hipStreamSynchronize(s));
    return SFM_GPU_OK;
}

}  // namespace

extern "C" {

int sfm_gpu_debug_set_score_filter(sfm_gpu_ctx* ctx, int on)
{
    if (!ctx) return SFM_GPU_ERR_ARG;
#ifdef SFM_GPU_TIMING_BUILD
    const int max_mode = 3;  // 2, 3: timing experiments with wrong counts
#else
    const int max_mode = 1;
#endif
    if (on < 0 || on > max_mode) return ctx->fail(SFM_GPU_ERR_UNSUPPORTED, "score filter mode %d", on);
    ctx->score_filter = on;
    return SFM_GPU_OK;
}

int sfm_gpu_debug_set_solver_split(sfm_gpu_ctx* ctx, int mode)
{
    if (!ctx) return SFM_GPU_ERR_ARG;
    if (mode < -1 || mode > 1) return ctx->fail(SFM_GPU_ERR_UNSUPPORTED, "solver split mode %d", mode);
    ctx->solver_split = mode;
    return SFM_GPU_OK;
}

int sfm_gpu_debug_set_score_prune(sfm_gpu_ctx* ctx, int on)
{
    if (!ctx) return SFM_GPU_ERR_ARG;
    if (on < 0 || on > 4) return ctx->fail(SFM_GPU_ERR_UNSUPPORTED, "score prune mode %d", on);
    ctx->score_prune = on;
    return SFM_GPU_OK;
}

int sfm_gpu_ransac(sfm_gpu_ctx* ctx, const double* pts1, const double* pts2, int n, const double* K,
                   int model, double prob, double threshold_px, int max_iters, int mode,
                   const int32_t* sample_table, int n_hyp, uint8_t* mask, double* model_out,
                   int32_t* n_inliers, int32_t* iters_used, int32_t* ok)
{
    if (!ctx) return SFM_GPU_ERR_ARG;
    if (n < 0 || !ok || (n > 0 && (!pts1 || !pts2 || !mask)) || !model_out)
        return ctx->fail(SFM_GPU_ERR_ARG, "null pointer or negative count");
    if (model == SFM_GPU_MODEL_ESSENTIAL && !K) return ctx->fail(SFM_GPU_ERR_ARG, "K is required");
    if (sample_table && mode != SFM_GPU_RANSAC_FIXED)
        return ctx->fail(SFM_GPU_ERR_ARG, "a sample table needs SFM_GPU_RANSAC_FIXED");
    *ok = 0;
    if (n_inliers) *n_inliers = 0;
    if (iters_used) *iters_used = 0;
    RansacParams rp;
    rp.model = model;
    rp.mode = mode;
    if (K) memcpy(rp.K, K, sizeof(rp.K));
    else memset(rp.K, 0, sizeof(rp.K));
    rp.prob = prob;
    rp.threshold_px = threshold_px;
    rp.max_iters = max_iters;
    rp.n_hyp = n_hyp;
    std::vector<RansacProb> probs(1);
    int rc = fill_params(ctx, probs[0], rp);
    if (rc) return rc;
    const int m = probs[0].m;
    ctx->last_hyp = -1;
    if (n < m) return SFM_GPU_OK;  // OpenCV: returns false, mask untouched
    // cv::findFundamentalMat runs RANSAC only from 15 points on: 8..14 points go to
    // LMedS, which is not on this path.  (7 points: one direct solve, as here.  The
    // fixed-count mode is a benchmark mode of this library and takes any n >= 7.)
    if (model == SFM_GPU_MODEL_FUNDAMENTAL && mode == SFM_GPU_RANSAC_ADAPTIVE && n > m && n < 15)
        return ctx->fail(SFM_GPU_ERR_UNSUPPORTED,
                         "fundamental matrix from %d points: OpenCV uses LMedS for 8..14 points, not RANSAC", n);
    if (sample_table && n > m)
        for (long long i = 0; i < (long long)n_hyp * m; ++i)
            if (sample_table[i] < 0 || sample_table[i] >= n)
                return ctx->fail(SFM_GPU_ERR_ARG, "sample table entry %lld out of range", i);
    SFM_HIP(ctx, hipSetDevice(ctx->device));
    const int hyp_cap = mode == SFM_GPU_RANSAC_FIXED ? n_hyp : std::max(max_iters, 1);
    const size_t px_bytes = align_up((size_t)n * 16, 256);
    SFM_HIP(ctx, ctx->rs_pts.reserve(px_bytes * 2 + 256));
    SFM_HIP(ctx, ctx->rs_models.reserve(ransac_ws_bytes(n, hyp_cap, m)));
    char* pbase = ctx->rs_pts.as<char>();
    double* dpx1 = (double*)pbase;
    double* dpx2 = (double*)(pbase + px_bytes);
    int32_t* dn = (int32_t*)(pbase + 2 * px_bytes);
    SFM_HIP(ctx, hipMemcpyAsync(dpx1, pts1, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
    SFM_HIP(ctx, hipMemcpyAsync(dpx2, pts2, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
    SFM_HIP(ctx, hipMemcpyAsync(dn, &n, 4, hipMemcpyHostToDevice, ctx->stream));
    probs[0].px1 = dpx1;
    probs[0].px2 = dpx2;
    probs[0].n_ptr = dn;
    carve_ransac(probs[0], ctx->rs_models.as<char>(), n, hyp_cap, m);
    SFM_HIP(ctx, ctx->rs_counts.reserve(R_COUNT * 4));
    probs[0].result = ctx->rs_counts.as<int32_t>();
    ctx->dbg_table = probs[0].table;
    ctx->dbg_counts = probs[0].counts;
    ctx->dbg_nmodels = probs[0].nmodels;
    ctx->dbg_models = probs[0].models;
    ctx->dbg_hyp_cap = hyp_cap;
    ctx->dbg_m = m;
    std::vector<int32_t> results;
    rc = ransac_batch(ctx, probs, n, hyp_cap, rp, (sample_table && n > m) ? sample_table : nullptr, results, true);
    if (rc) return rc;
    *ok = results[R_OK];
    ctx->last_hyp = results[R_OK] ? results[R_HYP] : -1;
    if (iters_used) *iters_used = results[R_ITERS];
    if (results[R_OK]) {
        if (n_inliers) *n_inliers = results[R_NINL];
        SFM_HIP(ctx, hipMemcpyAsync(mask, probs[0].mask, n, hipMemcpyDeviceToHost, ctx->stream));
        SFM_HIP(ctx, hipMemcpyAsync(model_out, probs[0].model_out, 72, hipMemcpyDeviceToHost, ctx->stream));
        SFM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return SFM_GPU_OK;
}

int sfm_gpu_ransac_candidates(sfm_gpu_ctx* ctx, double* models_out, int capacity, int32_t* n_models)
{
    if (!ctx || !n_models || capacity < 0 || (capacity > 0 && !models_out)) return SFM_GPU_ERR_ARG;
    *n_models = 0;
    if (ctx->last_hyp < 0 || !ctx->dbg_models || ctx->last_hyp >= ctx->dbg_hyp_cap)
        return ctx->fail(SFM_GPU_ERR_ARG, "the last sfm_gpu_ransac call on this context found no model");
    SFM_HIP(ctx, hipSetDevice(ctx->device));
    int32_t nm = 0;
    SFM_HIP(ctx, hipMemcpyAsync(&nm, ctx->dbg_nmodels + ctx->last_hyp, 4, hipMemcpyDeviceToHost, ctx->stream));
    SFM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (nm < 0 || nm > MAX_MODELS) return ctx->fail(SFM_GPU_ERR_HIP, "corrupt candidate count %d", nm);
    *n_models = nm;
    if (nm > capacity)
        return ctx->fail(SFM_GPU_ERR_CAPACITY, "%d candidate models, room for %d", nm, capacity);
    if (nm > 0) {
        SFM_HIP(ctx, hipMemcpyAsync(models_out, ctx->dbg_models + (size_t)ctx->last_hyp * MAX_MODELS * 9,
                                    (size_t)nm * 72, hipMemcpyDeviceToHost, ctx->stream));
        SFM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return SFM_GPU_OK;
}

// Test hook: per-hypothesis data of the last sfm_gpu_ransac call on this
// context.  Any output pointer may be null.  n_hyp rows are copied.
int sfm_gpu_debug_ransac_dump(sfm_gpu_ctx* ctx, int n_hyp, int32_t* table, int32_t* nmodels,
                              int32_t* counts, double* models)
{
    if (!ctx || n_hyp < 0 || n_hyp > ctx->dbg_hyp_cap || !ctx->dbg_table)
        return ctx ? ctx->fail(SFM_GPU_ERR_ARG, "no RANSAC run to dump, or n_hyp too large") : SFM_GPU_ERR_ARG;
    hipStream_t s = ctx->stream;
    if (table) SFM_HIP(ctx, hipMemcpyAsync(table, ctx->dbg_table, (size_t)n_hyp * ctx->dbg_m * 4, hipMemcpyDeviceToHost, s));
    if (nmodels) SFM_HIP(ctx, hipMemcpyAsync(nmodels, ctx->dbg_nmodels, (size_t)n_hyp * 4, hipMemcpyDeviceToHost, s));
    if (counts) SFM_HIP(ctx, hipMemcpyAsync(counts, ctx->dbg_counts, (size_t)n_hyp * 40, hipMemcpyDeviceToHost, s));
    if (models) SFM_HIP(ctx, hipMemcpyAsync(models, ctx->dbg_models, (size_t)n_hyp * 720, hipMemcpyDeviceToHost, s));
    SFM_HIP(ctx, hipStreamSynchronize(s));
    return SFM_GPU_OK;
}

// Everything sfm_gpu_match_pairs needs to know before it touches the stream.
struct MpPlan {
    RansacParams rp;
    RansacProb proto;
    int m = 0, hyp_cap = 0, nq_max = 1, nt_max = 1, qalloc_max = ROW_ALIGN, group = 1;
    size_t per_pair_ransac = 0, out_b = 0, cnt_b = 0, redo_b = 0, good_b = 0, px_b = 0, inl_b = 0;
};

// want_inliers: the pairs' inlier records are wanted; own_records: in a buffer of
// the context (false: the caller brings device memory for them).
static int mp_plan(sfm_gpu_ctx* ctx, MpPlan& pl, sfm_gpu_image* const* images, int n_images, const int32_t* pairs,
                   int n_pairs, const sfm_gpu_match_params* params, bool want_inliers, int match_capacity,
                   bool own_records = true)
{
    for (int p = 0; p < 2 * n_pairs; ++p)
        if (pairs[p] < 0 || pairs[p] >= n_images)
            return ctx->fail(SFM_GPU_ERR_ARG, "pair %d names image %d of %d", p / 2, pairs[p], n_images);
    if (!std::isfinite(params->ratio))
        return ctx->fail(SFM_GPU_ERR_ARG, "ratio must be finite");
    RansacParams& rp = pl.rp;
    rp.model = params->model;
    rp.mode = params->mode;
    memcpy(rp.K, params->K, sizeof(rp.K));
    rp.prob = params->prob;
    rp.threshold_px = params->threshold_px;
    rp.max_iters = params->max_iters;
    rp.n_hyp = params->n_hyp;
    int rc = fill_params(ctx, pl.proto, rp);
    if (rc) return rc;
    pl.m = pl.proto.m;
    pl.hyp_cap = rp.mode == SFM_GPU_RANSAC_FIXED ? rp.n_hyp : std::max(rp.max_iters, 1);
    for (int p = 0; p < n_pairs; ++p) {
        const sfm_gpu_image* iq = images[pairs[2 * p]];
        const sfm_gpu_image* it = images[pairs[2 * p + 1]];
        if (!iq || !it) return ctx->fail(SFM_GPU_ERR_ARG, "null image in pair %d", p);
        if (!iq->d.kp || !it->d.kp)
            return ctx->fail(SFM_GPU_ERR_ARG, "pair %d: images need keypoints for RANSAC", p);
        pl.nq_max = std::max(pl.nq_max, iq->d.n);
        pl.nt_max = std::max(pl.nt_max, it->d.n);
        pl.qalloc_max = std::max(pl.qalloc_max, iq->d.rows_alloc);
    }
    if (want_inliers && match_capacity < pl.nq_max)
        return ctx->fail(SFM_GPU_ERR_CAPACITY, "match_capacity %d < largest query image %d",
                         match_capacity, pl.nq_max);
    // pairs are processed in groups that share one launch per stage
    pl.per_pair_ransac = ransac_ws_bytes(pl.nq_max, pl.hyp_cap, pl.m);
    int group = (int)std::min<size_t>((size_t)n_pairs,
                                      std::max<size_t>(1, ((size_t)3 << 30) / (pl.per_pair_ransac + 1)));
    pl.group = std::min(group, 256);
    pl.out_b = align_up((size_t)pl.nq_max * 2 * sizeof(sfm_gpu_dmatch), 256);
    pl.cnt_b = align_up((size_t)pl.nq_max * 4, 256);
    pl.redo_b = align_up(((size_t)pl.nq_max + 1) * 4, 256);
    pl.good_b = align_up((size_t)pl.nq_max * sizeof(sfm_gpu_dmatch), 256);
    pl.px_b = align_up((size_t)pl.nq_max * 16, 256);
    pl.inl_b = want_inliers && own_records ? align_up((size_t)match_capacity * sizeof(sfm_gpu_dmatch), 256) : 0;
    return SFM_GPU_OK;
}

// Bytes of one group's descriptor block: the three arrays back to back, each
// starting on a 64-byte boundary (device block and pinned staging alike).
static size_t mp_stage_bytes(int g)
{
    return align_up(sizeof(KnnJob) * (size_t)g, 64) + align_up(sizeof(PairDev) * (size_t)g, 64) +
           align_up(sizeof(RansacProb) * (size_t)g, 64) + 64;
}

static int mp_reserve(sfm_gpu_ctx* ctx, const MpPlan& pl)
{
    const size_t group = (size_t)pl.group;
    SFM_HIP(ctx, hipSetDevice(ctx->device));
    SFM_HIP(ctx, ctx->out.reserve(pl.out_b * group));
    SFM_HIP(ctx, ctx->counts.reserve(pl.cnt_b * group));
    SFM_HIP(ctx, ctx->redo.reserve(pl.redo_b * group));
    SFM_HIP(ctx, ctx->mp_good.reserve((pl.good_b + pl.inl_b) * group));
    SFM_HIP(ctx, ctx->mp_pts.reserve(pl.px_b * 2 * group));
    SFM_HIP(ctx, ctx->mp_ngood.reserve(8 * group));
    SFM_HIP(ctx, ctx->rs_counts.reserve((size_t)R_COUNT * 4 * group));
    SFM_HIP(ctx, ctx->rs_models.reserve(pl.per_pair_ransac * group));
    // the RANSAC workspace is about to be carved anew: what the last sfm_gpu_ransac
    // call left there (sfm_gpu_ransac_candidates, the test hooks) is gone
    ctx->last_hyp = -1;
    ctx->dbg_table = ctx->dbg_counts = ctx->dbg_nmodels = nullptr;
    ctx->dbg_models = nullptr;
    ctx->dbg_hyp_cap = 0;
    SFM_HIP(ctx, ctx->jobs.reserve(mp_stage_bytes((int)group)));
    return SFM_GPU_OK;
}

// Enqueues one group of pairs (g0 .. g0+g) on the context's stream; the last
// kernel leaves the (n_good, n_inliers) words in `res_pinned` (pinned host memory).  In
// fixed-count mode nothing here waits for the device.
// records_dev: caller-owned device memory for the inlier records (pair p's at
// records_dev + p * match_capacity), or null: a buffer of the context when
// want_inliers, none otherwise.
// optimistic (adaptive RANSAC only): the first batch of sample rows is enqueued and
// nothing is waited for; a pair whose loop is not over by then reports -1 inliers
// in res_pinned and has to be run again without this flag.
static int mp_enqueue(sfm_gpu_ctx* ctx, const MpPlan& pl, sfm_gpu_image* const* images, const int32_t* pairs,
                      int g0, int g, const sfm_gpu_match_params* params, bool want_inliers, int match_capacity,
                      std::vector<PairDev>& pds, void* res_pinned, sfm_gpu_dmatch* records_dev = nullptr,
                      bool optimistic = false, int records_g0 = -1)
{
    if (records_g0 < 0) records_g0 = g0;  // where the group's first pair sits in records_dev
    hipStream_t s = ctx->stream;
    std::vector<KnnJob> jobs(g);
    std::vector<RansacProb> probs(g, pl.proto);
    std::vector<int32_t> results;
    pds.assign(g, PairDev());
    // device layout of the descriptor block: jobs | pairs | problems
    const size_t off_pd = align_up(sizeof(KnnJob) * (size_t)g, 64);
    const size_t off_pb = off_pd + align_up(sizeof(PairDev) * (size_t)g, 64);
    const KnnPlan plan = plan_knn(pl.nq_max, pl.nt_max, g, ctx->knn_variant);
    // a smaller (last) group may be planned with more train parts per job: size
    // the record buffer from this group's own plan
    const size_t partial_b = align_up((size_t)plan.n_splits * pl.qalloc_max * REC_SUBS * sizeof(uint4), 256);
    SFM_HIP(ctx, ctx->partial.reserve(partial_b * g));
    for (int i = 0; i < g; ++i) {
        const ImageDev& q = images[pairs[2 * (g0 + i)]]->d;
        const ImageDev& t = images[pairs[2 * (g0 + i) + 1]]->d;
        KnnJob& j = jobs[i];
        j.q_desc = q.desc;
        j.t_desc = t.desc;
        j.t_cin = t.cin;
        j.q_qn = q.qn;
        j.t_qn = t.qn;
        j.partial = (uint4*)(ctx->partial.as<char>() + partial_b * i);
        j.out = (sfm_gpu_dmatch*)(ctx->out.as<char>() + pl.out_b * i);
        j.counts = (int32_t*)(ctx->counts.as<char>() + pl.cnt_b * i);
        j.redo = (int32_t*)(ctx->redo.as<char>() + pl.redo_b * i);
        j.n_query = q.n;
        j.n_train = t.n;
        j.q_rows_alloc = q.rows_alloc;
        j.t_chunks = t.n > 0 ? (t.n + CHUNK - 1) / CHUNK : 0;
        PairDev& pd = pds[i];
        pd.knn = j.out;
        pd.counts = j.counts;
        pd.kp_q = q.kp;
        pd.kp_t = t.kp;
        pd.good = (sfm_gpu_dmatch*)(ctx->mp_good.as<char>() + (pl.good_b + pl.inl_b) * i);
        pd.inliers = !want_inliers ? nullptr
                     : records_dev ? records_dev + (size_t)(records_g0 + i) * match_capacity
                                   : (sfm_gpu_dmatch*)((char*)pd.good + pl.good_b);
        pd.px1 = (double*)(ctx->mp_pts.as<char>() + pl.px_b * 2 * i);
        pd.px2 = (double*)(ctx->mp_pts.as<char>() + pl.px_b * 2 * i + pl.px_b);
        pd.n_good = ctx->mp_ngood.as<int32_t>() + 2 * i;
        pd.n_inliers = pd.n_good + 1;
        pd.host_counts = (int32_t*)res_pinned + 2 * i;
        pd.prob = (const RansacProb*)(ctx->jobs.as<char>() + off_pb) + i;
        pd.n_query = q.n;
        pd.capacity = match_capacity;
        RansacProb& pb = probs[i];
        pb.px1 = pd.px1;
        pb.px2 = pd.px2;
        pb.n_ptr = pd.n_good;
        carve_ransac(pb, ctx->rs_models.as<char>() + pl.per_pair_ransac * i, pl.nq_max, pl.hyp_cap, pl.m);
        pb.result = ctx->rs_counts.as<int32_t>() + (size_t)R_COUNT * i;
        pd.mask = pb.mask;
        pd.result = pb.result;
    }
    // the three descriptor arrays travel as one block: one copy on the stream
    const size_t blob_b = off_pb + sizeof(RansacProb) * (size_t)g;
    std::vector<char> blob(blob_b);
    memcpy(blob.data(), jobs.data(), sizeof(KnnJob) * (size_t)g);
    memcpy(blob.data() + off_pd, pds.data(), sizeof(PairDev) * (size_t)g);
    memcpy(blob.data() + off_pb, probs.data(), sizeof(RansacProb) * (size_t)g);
    KnnJob* djobs = ctx->jobs.as<KnnJob>();
    PairDev* dpds = (PairDev*)(ctx->jobs.as<char>() + off_pd);
    RansacProb* dprobs = (RansacProb*)(ctx->jobs.as<char>() + off_pb);
    const size_t blob_pad = align_up(blob_b, 64);
    if (ctx->up_stage && ctx->up_used + blob_pad <= ctx->up_stage->cap) {
        // staged in pinned memory: one short kernel copies the block and clears
        // the redo counters that head each job's list
        char* at = ctx->up_stage->as<char>() + ctx->up_used;
        memcpy(at, blob.data(), blob_b);
        ctx->up_used += blob_pad;
        SFM_HIP(ctx, launch_upload_block(s, djobs, at, blob_b, ctx->redo.as<int32_t>(), pl.redo_b, g));
        if (ctx->want_up_ev) {
            SFM_HIP(ctx, hipEventRecord(ctx->up_ev, s));
            ctx->want_up_ev = false;
        }
    } else {
        SFM_HIP(ctx, hipMemsetAsync(ctx->redo.p, 0, pl.redo_b * g, s));
        SFM_HIP(ctx, hipMemcpyAsync(djobs, blob.data(), blob_b, hipMemcpyHostToDevice, s));
    }
    {
        ProfScope ps(ctx, "knn2_mfma", true);
        SFM_HIP(ctx, launch_knn2(s, djobs, g, pl.qalloc_max, plan, ps.a, ps.b));
    }
    {
        ProfScope ps(ctx, "knn_finalize");
        // the pair loop keeps only the ratio-test survivors: queries that
        // certainly fail the test are not given exact neighbours
        SFM_HIP(ctx, launch_knn_finalize(s, djobs, g, pl.nq_max, plan.n_splits, 2, plan.group_mode, params->ratio));
    }
    if (ctx->front_ev) {
        // the sweep and the neighbour extraction are behind (see sfm_gpu_stream_wait_front)
        SFM_HIP(ctx, hipEventRecord(ctx->front_ev, s));
        ctx->front_recorded = true;
    }
    {
        ProfScope ps(ctx, "knn_redo");
        SFM_HIP(ctx, launch_knn_redo(s, djobs, g, plan.n_splits, 2, params->ratio));
    }
    {
        ProfScope ps(ctx, "ratio_compact");
        SFM_HIP(ctx, launch_ratio_compact(s, dpds, g, pl.nq_max, params->ratio));
    }
    bool pairs_done = false;  // fixed-count mode: the selection kernel has gathered the inliers too
    const bool opt = optimistic && pl.rp.mode == SFM_GPU_RANSAC_ADAPTIVE;
    int rc = ransac_batch(ctx, probs, pl.nq_max, pl.hyp_cap, pl.rp, nullptr, results, false, dprobs, true, dpds,
                          &pairs_done, opt);
    if (rc) return rc;
    if (!pairs_done) {
        ProfScope ps(ctx, "inlier_compact");
        SFM_HIP(ctx, launch_inlier_compact(s, dpds, g, opt));
    }
    return SFM_GPU_OK;  // inlier_compact has written the counts into res_pinned
}

// Copies one group's counts out of `res`.
static void mp_collect(sfm_gpu_ctx*, const int32_t* res, int g, int32_t* n_good, int32_t* n_inliers)
{
    for (int i = 0; i < g; ++i) {
        n_good[i] = res[2 * i];
        n_inliers[i] = res[2 * i + 1];
    }
}

// What sfm_gpu_match_pairs_pose adds to sfm_gpu_match_pairs.
struct PoseOut {
    double distance_thresh;
    double *models, *R, *t;
    int32_t* n_pose_good;
};

static int match_pairs_impl(sfm_gpu_ctx* ctx, sfm_gpu_image* const* images, int n_images, const int32_t* pairs,
                            int n_pairs, const sfm_gpu_match_params* params, int32_t* n_good, int32_t* n_inliers,
                            sfm_gpu_dmatch* inlier_matches, int match_capacity, const PoseOut* pose)
{
    if (n_pairs < 0 || n_images < 0 || !params ||
        (n_pairs > 0 && (!images || !pairs || !n_good || !n_inliers)))
        return ctx->fail(SFM_GPU_ERR_ARG, "null pointer or negative count");
    if (n_pairs == 0) return SFM_GPU_OK;
    MpPlan pl;
    // the pose needs every pair's inlier records on the device, wanted by the caller or not
    const bool want_records = inlier_matches != nullptr || pose != nullptr;
    if (!inlier_matches) {
        match_capacity = 1;
        for (int p = 0; p < n_pairs; ++p) {
            const int q = pairs[2 * p];
            if (q >= 0 && q < n_images && images[q]) match_capacity = std::max(match_capacity, images[q]->d.n);
        }
    }
    int rc = mp_plan(ctx, pl, images, n_images, pairs, n_pairs, params, want_records, match_capacity);
    if (rc) return rc;
    rc = mp_reserve(ctx, pl);
    if (rc) return rc;
    hipStream_t s = ctx->stream;
    // second RANSAC of the pose (main.cpp:418): findEssentialMat(pointsL, pointsR, K, RANSAC) with
    // OpenCV's defaults prob 0.999, threshold 1.0, at most 1000 iterations
    RansacParams rp2 = pl.rp;
    RansacProb proto2;
    const int hyp_cap2 = 1000;
    size_t per_pair2 = 0;
    if (pose) {
        rp2.model = SFM_GPU_MODEL_ESSENTIAL;
        rp2.mode = SFM_GPU_RANSAC_ADAPTIVE;
        rp2.prob = 0.999;
        rp2.threshold_px = 1.0;
        rp2.max_iters = hyp_cap2;
        rc = fill_params(ctx, proto2, rp2);
        if (rc) return rc;
        per_pair2 = ransac_ws_bytes(pl.nq_max, hyp_cap2, proto2.m);
        SFM_HIP(ctx, ctx->rs_models2.reserve(per_pair2 * (size_t)pl.group + (size_t)R_COUNT * 4 * pl.group));
    }
    for (auto& slot : ctx->mp_sync) {
        SFM_HIP(ctx, slot.stage.reserve(mp_stage_bytes(pl.group)));
        SFM_HIP(ctx, slot.res.reserve(8 * (size_t)pl.group));
        if (!slot.ev) SFM_HIP(ctx, hipEventCreateWithFlags(&slot.ev, hipEventDisableTiming));
        // the group's inlier records, when they are to be copied out behind a later group
        if (inlier_matches && !pose)
            SFM_HIP(ctx, slot.rec.reserve((size_t)pl.group * match_capacity * sizeof(sfm_gpu_dmatch)));
    }
    // Enqueues group k (pairs g0 ..) in slot k & 1.
    auto enqueue = [&](int k, bool optimistic) -> int {
        sfm_gpu_ctx::MpSlot& slot = ctx->mp_sync[k & 1];
        const int g0 = k * pl.group, g = std::min(pl.group, n_pairs - g0);
        ctx->up_stage = &slot.stage;  // this slot's previous group has been waited for
        ctx->up_used = 0;
        sfm_gpu_dmatch* rec = inlier_matches && !pose ? slot.rec.as<sfm_gpu_dmatch>() : nullptr;
        const int rc2 = mp_enqueue(ctx, pl, images, pairs, g0, g, params, want_records, match_capacity, slot.pds,
                                   slot.res.p, rec, optimistic, 0);
        ctx->up_stage = nullptr;
        if (rc2) return rc2;
        SFM_HIP(ctx, hipEventRecord(slot.ev, s));
        slot.n_pairs = g;
        return SFM_GPU_OK;
    };
    // Waits for group k and hands its results to the caller.  *again = true (and
    // nothing handed over) when a pair's adaptive loop wanted more sample rows than
    // the optimistic enqueue gave it.
    auto collect = [&](int k, bool* again) -> int {
        sfm_gpu_ctx::MpSlot& slot = ctx->mp_sync[k & 1];
        const int g0 = k * pl.group, g = slot.n_pairs;
        SFM_HIP(ctx, hipEventSynchronize(slot.ev));
        const int32_t* res = slot.res.as<int32_t>();
        if (again) {
            *again = false;
            for (int i = 0; i < g; ++i)
                if (res[2 * i + 1] < 0) *again = true;
            if (*again) return SFM_GPU_OK;
        }
        mp_collect(ctx, res, g, n_good + g0, n_inliers + g0);
        if (inlier_matches) {
            for (int i = 0; i < g; ++i)
                if (n_inliers[g0 + i] > 0)
                    SFM_HIP(ctx, hipMemcpyAsync(inlier_matches + (size_t)(g0 + i) * match_capacity,
                                                slot.pds[i].inliers, (size_t)n_inliers[g0 + i] * sizeof(sfm_gpu_dmatch),
                                                hipMemcpyDeviceToHost, s));
            if (pose) SFM_HIP(ctx, hipStreamSynchronize(s));
        }
        return SFM_GPU_OK;
    };
    const int n_groups = (n_pairs + pl.group - 1) / pl.group;
    if (!pose) {
        // Two groups on the stream: group k+1 is enqueued before the host waits for
        // group k, so the device does not idle through the host's work between
        // groups.  The adaptive loop is enqueued optimistically (its first batch of
        // sample rows, which is all it needs on image pairs that overlap or plainly
        // do not); a group in which some pair wants more is run again, in the blocking
        // way, once the stream has drained -- by then the group behind it is complete
        // and everything it produced sits in its own slot.
        for (int k = 0; k <= n_groups; ++k) {
            if (k < n_groups) {
                rc = enqueue(k, true);
                if (rc) return rc;
            }
            if (k == 0) continue;
            bool again = false;
            rc = collect(k - 1, &again);
            if (rc) return rc;
            if (again) {
                SFM_HIP(ctx, hipStreamSynchronize(s));
                rc = enqueue(k - 1, false);
                if (rc) return rc;
                rc = collect(k - 1, nullptr);
                if (rc) return rc;
                // (the shared workspaces now hold group k-1 again; group k, which ran in
                // between, is complete, and pairs of it that want more rows are found when
                // it is collected, as usual)
            }
        }
        if (inlier_matches) SFM_HIP(ctx, hipStreamSynchronize(s));  // the last records are on the host
        return SFM_GPU_OK;
    }
    for (int k = 0; k < n_groups; ++k) {
        const int g0 = k * pl.group, g = std::min(pl.group, n_pairs - g0);
        rc = enqueue(k, false);
        if (rc) return rc;
        rc = collect(k, nullptr);
        if (rc) return rc;
        std::vector<PairDev>& pds = ctx->mp_sync[k & 1].pds;
        {
            // main.cpp:400-420 for every pair: a second essential-matrix RANSAC over the
            // pair's selectMatches, then recoverPose with that matrix on all of them.
            // The group's pair descriptors are still on the device (pds[0].prob points
            // into the same block, behind them).
            const PairDev* dpds = (const PairDev*)(ctx->jobs.as<char>() + align_up(sizeof(KnnJob) * (size_t)g, 64));
            SFM_HIP(ctx, launch_gather_inlier_points(s, dpds, g, pl.nq_max));
            std::vector<RansacProb> probs2(g, proto2);
            int32_t* res2 = (int32_t*)(ctx->rs_models2.as<char>() + per_pair2 * (size_t)pl.group);
            for (int i = 0; i < g; ++i) {
                RansacProb& pb = probs2[i];
                pb.px1 = pds[i].px1;
                pb.px2 = pds[i].px2;
                pb.n_ptr = pds[i].n_inliers;
                carve_ransac(pb, ctx->rs_models2.as<char>() + per_pair2 * i, pl.nq_max, hyp_cap2, proto2.m);
                pb.result = res2 + (size_t)R_COUNT * i;
            }
            std::vector<int32_t> results2;
            rc = ransac_batch(ctx, probs2, pl.nq_max, hyp_cap2, rp2, nullptr, results2, true);
            if (rc) return rc;
            std::vector<int32_t> ok2(g);
            for (int i = 0; i < g; ++i) ok2[i] = results2[(size_t)i * R_COUNT + R_OK] ? results2[(size_t)i * R_COUNT + R_NINL] : 0;
            rc = pose_for_pairs(ctx, ctx->rs_misc.as<RansacProb>(), g, ok2.data(), params->K, pose->distance_thresh,
                                pose->models ? pose->models + 9 * (size_t)g0 : nullptr, pose->R + 9 * (size_t)g0,
                                pose->t + 3 * (size_t)g0, pose->n_pose_good + g0, pl.nq_max, true);
            if (rc) return rc;
        }
    }
    return SFM_GPU_OK;
}

int sfm_gpu_match_pairs(sfm_gpu_ctx* ctx, sfm_gpu_image* const* images, int n_images, const int32_t* pairs,
                        int n_pairs, const sfm_gpu_match_params* params, int32_t* n_good,
                        int32_t* n_inliers, sfm_gpu_dmatch* inlier_matches, int match_capacity)
{
    if (!ctx) return SFM_GPU_ERR_ARG;
    return match_pairs_impl(ctx, images, n_images, pairs, n_pairs, params, n_good, n_inliers, inlier_matches,
                            match_capacity, nullptr);
}

int sfm_gpu_match_pairs_pose(sfm_gpu_ctx* ctx, sfm_gpu_image* const* images, int n_images, const int32_t* pairs,
                             int n_pairs, const sfm_gpu_match_params* params, int32_t* n_good,
                             int32_t* n_inliers, sfm_gpu_dmatch* inlier_matches, int match_capacity,
                             double distance_thresh, double* models_out, double* R_out, double* t_out,
                             int32_t* n_pose_good)
{
    if (!ctx) return SFM_GPU_ERR_ARG;
    if (n_pairs > 0 && (!R_out || !t_out || !n_pose_good))
        return ctx->fail(SFM_GPU_ERR_ARG, "sfm_gpu_match_pairs_pose: null pose output");
    if (params && params->model != SFM_GPU_MODEL_ESSENTIAL)
        return ctx->fail(SFM_GPU_ERR_ARG, "sfm_gpu_match_pairs_pose: recoverPose needs the essential model");
    if (!(distance_thresh >= 0))
        return ctx->fail(SFM_GPU_ERR_ARG, "sfm_gpu_match_pairs_pose: distance_thresh must be >= 0");
    const PoseOut po = {distance_thresh, models_out, R_out, t_out, n_pose_good};
    return match_pairs_impl(ctx, images, n_images, pairs, n_pairs, params, n_good, n_inliers, inlier_matches,
                            match_capacity, &po);
}

// Puts the copy of a batch's records to the host on the side stream, behind the batch (and, with
// after_upload, behind the descriptor upload of the batch that follows it on the context's stream).
static int mp_enqueue_record_copy(sfm_gpu_ctx* ctx, sfm_gpu_ctx::MpSlot& slot, bool after_upload)
{
    SFM_HIP(ctx, hipStreamWaitEvent(ctx->rec_stream, slot.ev, 0));
    if (after_upload) {
        // Beside the next batch's sweep: the whole record area by a plain copy (the DMA engines; the
        // records that exist are a third of it on the bench's data), because a kernel that writes just
        // those would take wavefront slots from the sweep for as long as the bytes take to cross the bus
        // (measured: the sweep 35-40 us longer per batch).
        SFM_HIP(ctx, hipStreamWaitEvent(ctx->rec_stream, ctx->up_ev, 0));
        SFM_HIP(ctx, hipMemcpyAsync(slot.records_host, slot.records_dev,
                                    sizeof(sfm_gpu_dmatch) * (size_t)slot.match_capacity * (size_t)slot.n_pairs,
                                    hipMemcpyDeviceToHost, ctx->rec_stream));
    } else {
        SFM_HIP(ctx, launch_records_to_host(ctx->rec_stream, slot.records_dev, slot.records_host_dev,
                                            (size_t)slot.match_capacity, slot.res.as<int32_t>(), slot.n_pairs));
    }
    SFM_HIP(ctx, hipEventRecord(slot.rec_ev, ctx->rec_stream));
    return SFM_GPU_OK;
}

static int match_pairs_begin_impl(sfm_gpu_ctx* ctx, sfm_gpu_image* const* images, int n_images, const int32_t* pairs,
                                  int n_pairs, const sfm_gpu_match_params* params, sfm_gpu_dmatch* records_dev,
                                  int match_capacity, sfm_gpu_dmatch* records_host)
{
    if (n_pairs < 1 || n_images < 0 || !params || !images || !pairs)
        return ctx->fail(SFM_GPU_ERR_ARG, "null pointer or no pairs");
    if (records_host && !records_dev)
        return ctx->fail(SFM_GPU_ERR_ARG, "records on the host need the device buffer they are copied from");
    if (ctx->mp_outstanding >= 2)
        return ctx->fail(SFM_GPU_ERR_CAPACITY, "two batches are already in flight: call sfm_gpu_match_pairs_end first");
    MpPlan pl;
    int rc = mp_plan(ctx, pl, images, n_images, pairs, n_pairs, params, records_dev != nullptr, match_capacity, false);
    if (rc) return rc;
    if (pl.group < n_pairs)
        return ctx->fail(SFM_GPU_ERR_CAPACITY, "%d pairs of this size do not fit one batch (%d at most)", n_pairs,
                         pl.group);
    SFM_HIP(ctx, hipSetDevice(ctx->device));
    sfm_gpu_dmatch* host_dev = nullptr;  // the device's address of the caller's pinned buffer
    if (records_host) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, records_host) != hipSuccess || at.type != hipMemoryTypeHost ||
            !at.devicePointer) {
            (void)hipGetLastError();
            return ctx->fail(SFM_GPU_ERR_ARG, "inlier_matches_host must be pinned host memory (hipHostMalloc / "
                                              "hipHostRegister): the device writes it directly");
        }
        host_dev = (sfm_gpu_dmatch*)at.devicePointer;
        if (!ctx->rec_stream) SFM_HIP(ctx, hipStreamCreateWithFlags(&ctx->rec_stream, hipStreamNonBlocking));
    }
    rc = mp_reserve(ctx, pl);
    if (rc) return rc;
    sfm_gpu_ctx::MpSlot& slot = ctx->mp_slot[ctx->mp_next];
    if (!slot.ev) SFM_HIP(ctx, hipEventCreateWithFlags(&slot.ev, hipEventDisableTiming));
    if (records_host && !slot.rec_ev) SFM_HIP(ctx, hipEventCreateWithFlags(&slot.rec_ev, hipEventDisableTiming));
    SFM_HIP(ctx, slot.stage.reserve(mp_stage_bytes(n_pairs)));
    SFM_HIP(ctx, slot.res.reserve(8 * (size_t)n_pairs));
    std::vector<PairDev> pds;
    ctx->up_stage = &slot.stage;  // this slot's previous batch has been ended
    ctx->up_used = 0;
    if (ctx->rec_deferred) {
        if (!ctx->up_ev)
            SFM_HIP(ctx, hipEventCreateWithFlags(&ctx->up_ev, hipEventDisableTiming | hipEventDisableSystemFence));
        ctx->want_up_ev = true;
    }
    rc = mp_enqueue(ctx, pl, images, pairs, 0, n_pairs, params, records_dev != nullptr, match_capacity, pds,
                    slot.res.p, records_dev, true);
    ctx->up_stage = nullptr;
    const bool up_recorded = ctx->rec_deferred && !ctx->want_up_ev;
    ctx->want_up_ev = false;
    if (ctx->rec_deferred) {
        // the batch before this one owes its records to the host: they travel while this batch's sweep runs
        sfm_gpu_ctx::MpSlot& prev = *ctx->rec_deferred;
        ctx->rec_deferred = nullptr;
        const int rc2 = mp_enqueue_record_copy(ctx, prev, up_recorded);
        if (rc2) return rc2;
    }
    if (rc) return rc;
    SFM_HIP(ctx, hipEventRecord(slot.ev, ctx->stream));
    slot.rec_pending = false;
    // kept for _end: an adaptive batch in which some pair wants more sample rows than
    // the first batch of them is run again there
    slot.images.assign(images, images + n_images);
    slot.pairs.assign(pairs, pairs + 2 * (size_t)n_pairs);
    slot.params = *params;
    slot.records_dev = records_dev;
    slot.records_host_dev = host_dev;
    slot.records_host = records_host;
    slot.match_capacity = match_capacity;
    slot.n_pairs = n_pairs;
    if (records_host) {
        // Behind the batch, on a stream of its own: the next batch runs on the context's stream
        // while these records travel.  The copy is enqueued by the next _begin, or by _end if
        // that comes first (see rec_deferred).  The counts are read from this slot's pinned
        // block, which nothing overwrites before _end.
        ctx->rec_deferred = &slot;
        slot.rec_pending = true;
    }
    ctx->mp_next ^= 1;
    ++ctx->mp_outstanding;
    return SFM_GPU_OK;
}

int sfm_gpu_match_pairs_begin(sfm_gpu_ctx* ctx, sfm_gpu_image* const* images, int n_images,
                              const int32_t* pairs, int n_pairs, const sfm_gpu_match_params* params)
{
    if (!ctx) return SFM_GPU_ERR_ARG;
    return match_pairs_begin_impl(ctx, images, n_images, pairs, n_pairs, params, nullptr, 0, nullptr);
}

int sfm_gpu_match_pairs_begin_records(sfm_gpu_ctx* ctx, sfm_gpu_image* const* images, int n_images,
                                      const int32_t* pairs, int n_pairs, const sfm_gpu_match_params* params,
                                      sfm_gpu_dmatch* inlier_matches_dev, int match_capacity,
                                      sfm_gpu_dmatch* inlier_matches_host)
{
    if (!ctx) return SFM_GPU_ERR_ARG;
    if (!inlier_matches_dev || match_capacity < 1)
        return ctx->fail(SFM_GPU_ERR_ARG, "sfm_gpu_match_pairs_begin_records: no device buffer for the records");
    return match_pairs_begin_impl(ctx, images, n_images, pairs, n_pairs, params, inlier_matches_dev, match_capacity,
                                  inlier_matches_host);
}

int sfm_gpu_stream_wait_front(sfm_gpu_ctx* waiter, sfm_gpu_ctx* source)
{
    if (!waiter || !source) return SFM_GPU_ERR_ARG;
    if (waiter->device != source->device)
        return waiter->fail(SFM_GPU_ERR_ARG, "the two contexts are on different devices");
    SFM_HIP(waiter, hipSetDevice(source->device));
    // (orders work of one device only: no system-scope release, whose write-back and invalidation
    // would be paid by the kernels behind the record)
    if (!source->front_ev)
        SFM_HIP(waiter, hipEventCreateWithFlags(&source->front_ev, hipEventDisableTiming | hipEventDisableSystemFence));
    if (source->front_recorded) SFM_HIP(waiter, hipStreamWaitEvent(waiter->stream, source->front_ev, 0));
    return SFM_GPU_OK;
}

int sfm_gpu_match_pairs_end(sfm_gpu_ctx* ctx, int32_t* n_good, int32_t* n_inliers, int capacity)
{
    if (!ctx) return SFM_GPU_ERR_ARG;
    if (!n_good || !n_inliers) return ctx->fail(SFM_GPU_ERR_ARG, "null pointer");
    if (ctx->mp_outstanding < 1) return ctx->fail(SFM_GPU_ERR_ARG, "no batch in flight");
    sfm_gpu_ctx::MpSlot& slot = ctx->mp_slot[ctx->mp_outstanding == 2 ? ctx->mp_next : ctx->mp_next ^ 1];
    if (capacity < slot.n_pairs)
        return ctx->fail(SFM_GPU_ERR_CAPACITY, "the oldest batch has %d pairs, room for %d", slot.n_pairs, capacity);
    if (ctx->rec_deferred == &slot) {
        ctx->rec_deferred = nullptr;
        const int rc0 = mp_enqueue_record_copy(ctx, slot, false);
        if (rc0) return rc0;
    }
    SFM_HIP(ctx, hipEventSynchronize(slot.ev));
    if (slot.rec_pending) SFM_HIP(ctx, hipEventSynchronize(slot.rec_ev));
    bool again = false;
    for (int i = 0; i < slot.n_pairs; ++i)
        if (slot.res.as<int32_t>()[2 * i + 1] < 0) again = true;
    if (again) {
        // Some pair's adaptive loop was not over after the rows enqueued by _begin:
        // the batch runs again, this time waiting between the batches of rows.  A
        // younger batch in flight finishes first (the workspaces are shared); what
        // it produced sits in its own slot and in the caller's memory.
        SFM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        MpPlan pl;
        const bool rec = slot.records_dev != nullptr;
        int rc = mp_plan(ctx, pl, slot.images.data(), (int)slot.images.size(), slot.pairs.data(), slot.n_pairs,
                         &slot.params, rec, slot.match_capacity, false);
        if (rc) return rc;
        rc = mp_reserve(ctx, pl);
        if (rc) return rc;
        std::vector<PairDev> pds;
        ctx->up_stage = &slot.stage;
        ctx->up_used = 0;
        rc = mp_enqueue(ctx, pl, slot.images.data(), slot.pairs.data(), 0, slot.n_pairs, &slot.params, rec,
                        slot.match_capacity, pds, slot.res.p, slot.records_dev, false);
        ctx->up_stage = nullptr;
        if (rc) return rc;
        if (slot.records_host_dev)
            SFM_HIP(ctx, launch_records_to_host(ctx->stream, slot.records_dev, slot.records_host_dev,
                                                (size_t)slot.match_capacity, slot.res.as<int32_t>(), slot.n_pairs));
        SFM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    slot.rec_pending = false;
    --ctx->mp_outstanding;
    mp_collect(ctx, slot.res.as<int32_t>(), slot.n_pairs, n_good, n_inliers);
    return SFM_GPU_OK;
}

}  // extern "C"
