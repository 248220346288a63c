// C ABI of libsfm_gpu: context, image preparation, k-NN matching entry points.
#include <algorithm>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <vector>

#include "ransac_dev.h"
#include "sfm_ctx.h"

using namespace sfm;

namespace {

// sfm_gpu_image handed to callers is the first member of its holder.
struct ImageHolder {
    sfm_gpu_image img;
    ImageStore st;
};

int check_knn_args(sfm_gpu_ctx* ctx, int n_query, int n_train, int dim, int dtype, int k)
{
    if (n_query < 0 || n_train < 0) return ctx->fail(SFM_GPU_ERR_ARG, "negative row count");
    if (dim != DIM) return ctx->fail(SFM_GPU_ERR_ARG, "dim must be 128, got %d", dim);
    if (dtype != SFM_GPU_DESC_F32 && dtype != SFM_GPU_DESC_U8)
        return ctx->fail(SFM_GPU_ERR_UNSUPPORTED, "unknown descriptor dtype %d", dtype);
    if (k < 1 || k > 2) return ctx->fail(SFM_GPU_ERR_UNSUPPORTED, "k must be 1 or 2, got %d", k);
    if (n_train >= (1 << 18))
        return ctx->fail(SFM_GPU_ERR_TOO_MANY, "n_train %d >= 2^18 (cv::BFMatcher limit)", n_train);
    return SFM_GPU_OK;
}

// Enqueues the preparation kernel for a batch of images.
int run_prep_jobs(sfm_gpu_ctx* ctx, const std::vector<PrepJob>& jobs, int dtype, int flag_slot)
{
    if (jobs.empty()) return SFM_GPU_OK;
    int rows_max = 0;
    for (const PrepJob& j : jobs) rows_max = std::max(rows_max, j.img.rows_alloc);
    const size_t bytes = sizeof(PrepJob) * jobs.size();
    SFM_HIP(ctx, ctx->prepjobs.reserve(bytes + 16));
    {
        // job descriptors: pinned staging + a one-workgroup copy kernel, so that
        // the call never waits for work already on the stream
        sfm_gpu_ctx::UpSlot& u = ctx->up_ring[ctx->up_ring_next];
        ctx->up_ring_next = (ctx->up_ring_next + 1) % 4;
        if (u.armed) SFM_HIP(ctx, hipEventSynchronize(u.ev));
        u.armed = false;
        SFM_HIP(ctx, u.buf.reserve(bytes + 64));
        if (!u.ev) SFM_HIP(ctx, hipEventCreateWithFlags(&u.ev, hipEventDisableTiming));
        std::memcpy(u.buf.p, jobs.data(), bytes);
        SFM_HIP(ctx, launch_upload_block(ctx->stream, ctx->prepjobs.p, u.buf.p, bytes, nullptr, 0, 0));
        SFM_HIP(ctx, hipEventRecord(u.ev, ctx->stream));
        u.armed = true;
    }
    ProfScope ps(ctx, "prep_rows");
    SFM_HIP(ctx, launch_prep_rows(ctx->stream, ctx->prepjobs.as<PrepJob>(), (int)jobs.size(), rows_max,
                                  dtype, ctx->flags.as<int32_t>() + flag_slot));
    return SFM_GPU_OK;
}

// Host rows -> contiguous device rows in `raw`.  A pitched copy from pageable
// memory goes row by row and is far slower than one flat copy, so strided input
// (cv::Mat with step > row bytes) is first gathered into pinned memory.
int upload_rows(sfm_gpu_ctx* ctx, DevBuf& raw, const void* desc, int n, size_t step, size_t rowbytes)
{
    SFM_HIP(ctx, raw.reserve((size_t)n * rowbytes));
    if (step == rowbytes) {
        SFM_HIP(ctx, hipMemcpyAsync(raw.p, desc, (size_t)n * rowbytes, hipMemcpyHostToDevice, ctx->stream));
        return SFM_GPU_OK;
    }
    SFM_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the staging buffer may still be in flight
    SFM_HIP(ctx, ctx->pin_b.reserve((size_t)n * rowbytes));
    for (int r = 0; r < n; ++r)
        std::memcpy(ctx->pin_b.as<char>() + (size_t)r * rowbytes, (const char*)desc + (size_t)r * step, rowbytes);
    SFM_HIP(ctx, hipMemcpyAsync(raw.p, ctx->pin_b.p, (size_t)n * rowbytes, hipMemcpyHostToDevice, ctx->stream));
    return SFM_GPU_OK;
}

// Uploads (or reads in place) n rows and runs the preparation kernel into st.
int prepare_image(sfm_gpu_ctx* ctx, ImageStore& st, DevBuf& raw, const void* desc, int n,
                  size_t step, int dtype, bool on_device, bool with_kp)
{
    SFM_HIP(ctx, st.shape(n, with_kp));
    const size_t rowbytes = dtype == SFM_GPU_DESC_F32 ? DIM * 4 : DIM;
    if (n > 0 && step < rowbytes) return ctx->fail(SFM_GPU_ERR_ARG, "row step %zu < row bytes", step);
    const void* src = desc;
    size_t src_step = step;
    if (!on_device && n > 0) {
        const int urc = upload_rows(ctx, raw, desc, n, step, rowbytes);
        if (urc) return urc;
        src = raw.p;
        src_step = rowbytes;
    }
    std::vector<PrepJob> jobs(1);
    jobs[0].src = (const unsigned char*)src;
    jobs[0].step = src_step;
    jobs[0].kp_src = nullptr;
    jobs[0].img = st.d;
    // the job array buffer is reused by the next call: order it behind this one
    int rc = run_prep_jobs(ctx, jobs, dtype, 0);
    if (rc) return rc;
    SFM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SFM_GPU_OK;
}

// Stores one k-NN job descriptor in device memory and clears the counter of the
// float-ranked pass, so that a call needs neither a copy nor a fill.
__global__ void store_knn_job_kernel(KnnJob job, KnnJob* dst, int32_t* redo_count)
{
    if (threadIdx.x == 0) {
        *dst = job;
        *redo_count = 0;
    }
}

// Device memory a k-NN call works in besides its inputs and outputs.
struct KnnWorkspace {
    uint4* partial;   // [n_splits][q.rows_alloc][REC_SUBS] sweep records
    int32_t* redo;    // [1 + n_query]
    KnnJob* job;
};

KnnJob make_knn_job(const ImageDev& q, const ImageDev& t, const KnnWorkspace& ws, sfm_gpu_dmatch* out_dev,
                    int32_t* cnt_dev)
{
    KnnJob job;
    job.q_desc = q.desc;
    job.t_desc = t.desc;
    job.t_cin = t.cin;
    job.q_qn = q.qn;
    job.t_qn = t.qn;
    job.partial = ws.partial;
    job.out = out_dev;
    job.counts = cnt_dev;
    job.redo = ws.redo;
    job.n_query = q.n;
    job.n_train = t.n;
    job.q_rows_alloc = q.rows_alloc;
    job.t_chunks = t.n > 0 ? (t.n + CHUNK - 1) / CHUNK : 0;
    return job;
}

// sweep -> neighbour extraction -> float-ranked pass for one job already stored at ws.job
int enqueue_knn(sfm_gpu_ctx* ctx, const ImageDev& q, const KnnWorkspace& ws, const KnnPlan& plan, int k,
                bool timers)
{
    {
        ProfScope ps(timers ? ctx : nullptr, "knn2_mfma");
        SFM_HIP(ctx, launch_knn2(ctx->stream, ws.job, 1, q.rows_alloc, plan));
    }
    {
        ProfScope ps(timers ? ctx : nullptr, "knn_finalize");
        SFM_HIP(ctx, launch_knn_finalize(ctx->stream, ws.job, 1, q.n, plan.n_splits, k, plan.group_mode, 0.0));
    }
    {
        ProfScope ps(timers ? ctx : nullptr, "knn_redo");
        SFM_HIP(ctx, launch_knn_redo(ctx->stream, ws.job, 1, plan.n_splits, k, 0.0));
    }
    return SFM_GPU_OK;
}

int knn_on_images(sfm_gpu_ctx* ctx, const ImageDev& q, const ImageDev& t, int k,
                  sfm_gpu_dmatch* matches, int32_t* counts, bool out_on_device)
{
    const int nq = q.n, nt = t.n;
    if (nq == 0) return SFM_GPU_OK;
    const KnnPlan plan = plan_knn(nq, nt, 1, ctx->knn_variant);
    SFM_HIP(ctx, ctx->partial.reserve((size_t)plan.n_splits * q.rows_alloc * REC_SUBS * sizeof(uint4)));
    SFM_HIP(ctx, ctx->redo.reserve(((size_t)nq + 1) * 4));
    SFM_HIP(ctx, ctx->jobs.reserve(sizeof(KnnJob)));
    sfm_gpu_dmatch* out_dev = matches;
    int32_t* cnt_dev = counts;
    if (!out_on_device) {
        SFM_HIP(ctx, ctx->out.reserve((size_t)nq * k * sizeof(sfm_gpu_dmatch)));
        SFM_HIP(ctx, ctx->counts.reserve((size_t)nq * 4));
        out_dev = ctx->out.as<sfm_gpu_dmatch>();
        cnt_dev = ctx->counts.as<int32_t>();
    }
    const KnnWorkspace ws = {ctx->partial.as<uint4>(), ctx->redo.as<int32_t>(), ctx->jobs.as<KnnJob>()};
    const KnnJob job = make_knn_job(q, t, ws, out_dev, cnt_dev);
    SFM_HIP(ctx, hipMemcpyAsync(ws.job, &job, sizeof(job), hipMemcpyHostToDevice, ctx->stream));
    SFM_HIP(ctx, hipMemsetAsync(ws.redo, 0, 4, ctx->stream));
    const int rc = enqueue_knn(ctx, q, ws, plan, k, true);
    if (rc) return rc;
    if (!out_on_device) {
        SFM_HIP(ctx, hipMemcpyAsync(matches, out_dev, (size_t)nq * k * sizeof(sfm_gpu_dmatch),
                                    hipMemcpyDeviceToHost, ctx->stream));
        SFM_HIP(ctx, hipMemcpyAsync(counts, cnt_dev, (size_t)nq * 4, hipMemcpyDeviceToHost, ctx->stream));
        SFM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return SFM_GPU_OK;
}

// ---- working memory of sfm_gpu_knn_match_images_noalloc --------------------
// One array per GPU the library runs on, part of the code object, so the entry
// point never allocates.  Layout: sweep records | redo list | job descriptor.
constexpr size_t NOALLOC_MAX_RECORDS = 1048576;  // uint4 records: n_splits * padded query rows * REC_SUBS
constexpr int NOALLOC_MAX_QUERY = 32768;
constexpr size_t NOALLOC_REDO_OFF = NOALLOC_MAX_RECORDS * sizeof(uint4);
constexpr size_t NOALLOC_JOB_OFF = NOALLOC_REDO_OFF + ((size_t)NOALLOC_MAX_QUERY + 64) * 4;
constexpr size_t NOALLOC_BYTES = NOALLOC_JOB_OFF + 256;
static_assert(sizeof(KnnJob) <= 256, "job descriptor must fit its slot");
static __device__ __align__(256) unsigned char scratch[NOALLOC_BYTES];

// Address of this device's copy of `scratch` (the array's name is only a handle
// on the host side; every GPU has its own copy at its own address).  The
// caller has made `device` current.
constexpr int MAX_DEVICES = 64;
void* g_scratch_ptr[MAX_DEVICES] = {};
std::mutex g_scratch_mutex;  // contexts may be created from several host threads
hipError_t resolve_scratch(int device, void** out)
{
    if (device < 0 || device >= MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        g_scratch_ptr[device] = p;
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}

}  // namespace

namespace sfm {
const RoctxApi& roctx_api()
{
    static const RoctxApi api = [] {
        RoctxApi a;
        const char* on = getenv("SFM_GPU_ROCTX");
        if (!on || on[0] != '1') return a;
        for (const char* name : {"librocprofiler-sdk-roctx.so", "libroctx64.so"}) {
            if (void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
                a.push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
                a.pop = (int (*)())dlsym(h, "roctxRangePop");
                if (a.push && a.pop) return a;
                a.push = nullptr;
                a.pop = nullptr;
            }
        }
        return a;
    }();
    return api;
}
}  // namespace sfm

extern "C" {

const char* sfm_gpu_version(void)
{
#ifdef SFM_GPU_TIMING_BUILD
    return "sfm_gpu 0.2 gfx950 timing-build";
#else
    return "sfm_gpu 0.2 gfx950";
#endif
}

int sfm_gpu_create(int device_id, void* hip_stream, sfm_gpu_ctx** out)
{
    if (!out) return SFM_GPU_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev)
        return SFM_GPU_ERR_HIP;
    int prev_dev = -1;
    if (hipGetDevice(&prev_dev) != hipSuccess) prev_dev = -1;
    if (hipSetDevice(device_id) != hipSuccess) return SFM_GPU_ERR_HIP;
    // the caller's current device is left as it was
    struct Restore {
        int d;
        ~Restore() { if (d >= 0) (void)hipSetDevice(d); }
    } restore{prev_dev};
    sfm_gpu_ctx* c = new (std::nothrow) sfm_gpu_ctx();
    if (!c) return SFM_GPU_ERR_NOMEM;
    c->device = device_id;
    if (hip_stream) {
        c->stream = (hipStream_t)hip_stream;
    } else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
            delete c;
            return SFM_GPU_ERR_HIP;
        }
        c->own_stream = true;
    }
    if (c->flags.reserve(64) != hipSuccess || hipMemsetAsync(c->flags.p, 0, 64, c->stream) != hipSuccess ||
        resolve_scratch(device_id, &c->static_scratch) != hipSuccess) {
        c->flags.release();
        if (c->own_stream) (void)hipStreamDestroy(c->stream);
        delete c;
        return SFM_GPU_ERR_HIP;
    }
    *out = c;
    return SFM_GPU_OK;
}

void sfm_gpu_destroy(sfm_gpu_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto& kv : c->timers)
        for (auto& pr : kv.second.pending) {
            c->event_pool.push_back(pr.first);
            c->event_pool.push_back(pr.second);
        }
    for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
    DevBuf* bufs[] = {&c->raw_q, &c->raw_t, &c->partial, &c->out, &c->counts, &c->redo, &c->jobs,
                      &c->flags, &c->prepjobs, &c->rs_pts, &c->rs_table, &c->rs_models, &c->rs_counts, &c->rs_mask,
                      &c->rs_misc, &c->rs_models2, &c->mp_good, &c->mp_pts, &c->mp_ngood, &c->pose_in, &c->pose_out, &c->pnp_ws};
    for (DevBuf* b : bufs) b->release();
    c->tmp_q.release();
    c->tmp_t.release();
    c->pin_a.release();
    c->pin_b.release();
    for (auto& slot : c->mp_sync) {
        if (slot.ev) (void)hipEventDestroy(slot.ev);
        slot.stage.release();
        slot.res.release();
        slot.rec.release();
    }
    if (c->front_ev) (void)hipEventDestroy(c->front_ev);
    if (c->up_ev) (void)hipEventDestroy(c->up_ev);
    for (auto& u : c->up_ring) {
        if (u.ev) (void)hipEventDestroy(u.ev);
        u.buf.release();
    }
    for (auto& slot : c->mp_slot) {
        if (slot.ev) (void)hipEventDestroy(slot.ev);
        if (slot.rec_ev) (void)hipEventDestroy(slot.rec_ev);
        slot.stage.release();
        slot.res.release();
    }
    if (c->rec_stream) {
        (void)hipStreamSynchronize(c->rec_stream);
        (void)hipStreamDestroy(c->rec_stream);
    }
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* sfm_gpu_last_error_string(const sfm_gpu_ctx* c) { return c ? c->err.c_str() : "null context"; }

int sfm_gpu_synchronize(sfm_gpu_ctx* c)
{
    if (!c) return SFM_GPU_ERR_ARG;
    SFM_HIP(c, hipStreamSynchronize(c->stream));
    return SFM_GPU_OK;
}

int sfm_gpu_knn_match(sfm_gpu_ctx* ctx, const void* query, int n_query, size_t query_step,
                      const void* train, int n_train, size_t train_step, int dim, int dtype, int k,
                      sfm_gpu_dmatch* matches, int32_t* counts)
{
    if (!ctx) return SFM_GPU_ERR_ARG;
    int rc = check_knn_args(ctx, n_query, n_train, dim, dtype, k);
    if (rc) return rc;
    if (n_query == 0) return SFM_GPU_OK;
    if (!query || !matches || !counts || (n_train > 0 && !train))
        return ctx->fail(SFM_GPU_ERR_ARG, "null pointer argument");
    SFM_HIP(ctx, hipSetDevice(ctx->device));
    SFM_HIP(ctx, hipMemsetAsync(ctx->flags.p, 0, 4, ctx->stream));
    rc = prepare_image(ctx, ctx->tmp_q, ctx->raw_q, query, n_query, query_step, dtype, false, false);
    if (rc) return rc;
    rc = prepare_image(ctx, ctx->tmp_t, ctx->raw_t, train, n_train, train_step, dtype, false, false);
    if (rc) return rc;
    // both conversions have been waited for (prepare_image synchronises): read the
    // validation flag now, so that no copy into this frame is in flight on any path out
    int32_t flag = 0;
    SFM_HIP(ctx, hipMemcpyAsync(&flag, ctx->flags.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    SFM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (flag)
        return ctx->fail(SFM_GPU_ERR_NOT_U8, "descriptor values must be integers in 0..255");
    return knn_on_images(ctx, ctx->tmp_q.d, ctx->tmp_t.d, k, matches, counts, false);
}

int sfm_gpu_ratio_test(const sfm_gpu_dmatch* matches, const int32_t* counts, int n_query, int k,
                       double ratio, int32_t* keep_rows, int32_t* n_keep)
{
    if (!matches || !counts || !keep_rows || !n_keep || n_query < 0 || k < 2) return SFM_GPU_ERR_ARG;
    int n = 0;
    for (int r = 0; r < n_query; r++) {
        if (counts[r] < 2) continue;
        const sfm_gpu_dmatch* m = matches + (size_t)r * k;
        if ((double)m[0].distance < ratio * (double)m[1].distance) keep_rows[n++] = r;
    }
    *n_keep = n;
    return SFM_GPU_OK;
}

int sfm_gpu_image_create(sfm_gpu_ctx* ctx, const void* desc, int n, size_t step, int dim, int dtype,
                         const float* keypoints_xy, int desc_on_device, sfm_gpu_image** out)
{
    if (!ctx || !out) return SFM_GPU_ERR_ARG;
    *out = nullptr;
    int rc = check_knn_args(ctx, n, 0, dim, dtype, 2);
    if (rc) return rc;
    if (n >= (1 << 18)) return ctx->fail(SFM_GPU_ERR_TOO_MANY, "image rows %d >= 2^18", n);
    if (n > 0 && !desc) return ctx->fail(SFM_GPU_ERR_ARG, "null descriptor pointer");
    SFM_HIP(ctx, hipSetDevice(ctx->device));
    SFM_HIP(ctx, hipMemsetAsync(ctx->flags.p, 0, 4, ctx->stream));
    ImageHolder* h = new (std::nothrow) ImageHolder();
    if (!h) return ctx->fail(SFM_GPU_ERR_NOMEM, "host allocation failed");
    rc = prepare_image(ctx, h->st, ctx->raw_q, desc, n, step, dtype, desc_on_device != 0,
                       keypoints_xy != nullptr);
    if (rc == SFM_GPU_OK && keypoints_xy && n > 0) {
        hipError_t e = hipMemcpyAsync(h->st.d.kp, keypoints_xy, (size_t)n * sizeof(float2),
                                      desc_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                                      ctx->stream);
        if (e != hipSuccess) rc = ctx->fail(SFM_GPU_ERR_HIP, "keypoint copy failed: %s", hipGetErrorString(e));
    }
    int32_t flag = 0;
    if (rc == SFM_GPU_OK) {
        hipError_t e = hipMemcpyAsync(&flag, ctx->flags.p, 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = ctx->fail(SFM_GPU_ERR_HIP, "sync failed: %s", hipGetErrorString(e));
        else if (flag) rc = ctx->fail(SFM_GPU_ERR_NOT_U8, "descriptor values must be integers in 0..255");
    }
    if (rc != SFM_GPU_OK) {
        (void)hipStreamSynchronize(ctx->stream);
        h->st.release();
        delete h;
        return rc;
    }
    h->img.d = h->st.d;
    *out = &h->img;
    return SFM_GPU_OK;
}

void sfm_gpu_image_destroy(sfm_gpu_ctx* ctx, sfm_gpu_image* img)
{
    if (!img) return;
    ImageHolder* h = (ImageHolder*)img;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    h->st.release();
    delete h;
}

int sfm_gpu_image_rows(const sfm_gpu_image* img) { return img ? img->d.n : -1; }

int sfm_gpu_knn_match_images(sfm_gpu_ctx* ctx, const sfm_gpu_image* query, const sfm_gpu_image* train,
                             int k, sfm_gpu_dmatch* matches, int32_t* counts, int out_on_device)
{
    if (!ctx || !query || !train) return SFM_GPU_ERR_ARG;
    int rc = check_knn_args(ctx, query->d.n, train->d.n, DIM, SFM_GPU_DESC_U8, k);
    if (rc) return rc;
    if (query->d.n > 0 && (!matches || !counts)) return ctx->fail(SFM_GPU_ERR_ARG, "null output pointer");
    SFM_HIP(ctx, hipSetDevice(ctx->device));
    return knn_on_images(ctx, query->d, train->d, k, matches, counts, out_on_device != 0);
}

int sfm_gpu_knn_match_images_noalloc(sfm_gpu_ctx* ctx, const sfm_gpu_image* query, const sfm_gpu_image* train,
                                     int k, sfm_gpu_dmatch* matches_dev, int32_t* counts_dev)
{
    if (!ctx || !query || !train) return SFM_GPU_ERR_ARG;
    int rc = check_knn_args(ctx, query->d.n, train->d.n, DIM, SFM_GPU_DESC_U8, k);
    if (rc) return rc;
    const ImageDev& q = query->d;
    const ImageDev& t = train->d;
    if (q.n == 0) return SFM_GPU_OK;
    if (!matches_dev || !counts_dev) return ctx->fail(SFM_GPU_ERR_ARG, "null output pointer");
    if (!ctx->static_scratch) return ctx->fail(SFM_GPU_ERR_HIP, "static working memory was not resolved");
    const KnnPlan plan = plan_knn(q.n, t.n, 1, ctx->knn_variant);
    const size_t records = (size_t)plan.n_splits * q.rows_alloc * REC_SUBS;
    if (q.n > NOALLOC_MAX_QUERY || records > NOALLOC_MAX_RECORDS)
        return ctx->fail(SFM_GPU_ERR_CAPACITY,
                         "sfm_gpu_knn_match_images_noalloc: %d query rows x %d train parts = %zu records exceed "
                         "the static working memory (at most %d query rows and %zu records of 16 bytes)",
                         q.n, plan.n_splits, records, NOALLOC_MAX_QUERY, NOALLOC_MAX_RECORDS);
    // launches go to the context's device; the caller's current device is put back
    int prev_dev = -1;
    if (hipGetDevice(&prev_dev) != hipSuccess) prev_dev = -1;
    SFM_HIP(ctx, hipSetDevice(ctx->device));
    struct Restore {
        int d;
        ~Restore() { if (d >= 0) (void)hipSetDevice(d); }
    } restore{prev_dev == ctx->device ? -1 : prev_dev};
    unsigned char* base = (unsigned char*)ctx->static_scratch;
    const KnnWorkspace ws = {(uint4*)base, (int32_t*)(base + NOALLOC_REDO_OFF), (KnnJob*)(base + NOALLOC_JOB_OFF)};
    const KnnJob job = make_knn_job(q, t, ws, matches_dev, counts_dev);
    // the job descriptor and the cleared redo counter reach the device as kernel
    // arguments: no copy, no fill, nothing staged
    hipLaunchKernelGGL(store_knn_job_kernel, dim3(1), dim3(64), 0, ctx->stream, job, ws.job, ws.redo);
    SFM_HIP(ctx, hipGetLastError());
    return enqueue_knn(ctx, q, ws, plan, k, false);
}

size_t sfm_gpu_image_packed_bytes(int n) { return n < 0 ? 0 : image_block_bytes(n); }

int sfm_gpu_image_prepare_into(sfm_gpu_ctx* ctx, const void* desc, int n, size_t step, int dim,
                               int dtype, const float* keypoints_xy, int desc_on_device,
                               void* packed_dev)
{
    if (!ctx || !packed_dev) return SFM_GPU_ERR_ARG;
    int rc = check_knn_args(ctx, n, 0, dim, dtype, 2);
    if (rc) return rc;
    if (n >= (1 << 18)) return ctx->fail(SFM_GPU_ERR_TOO_MANY, "image rows %d >= 2^18", n);
    if (n > 0 && !desc) return ctx->fail(SFM_GPU_ERR_ARG, "null descriptor pointer");
    SFM_HIP(ctx, hipSetDevice(ctx->device));
    const ImageDev d = image_layout(packed_dev, n);
    const size_t rowbytes = dtype == SFM_GPU_DESC_F32 ? DIM * 4 : DIM;
    if (n > 0 && step < rowbytes) return ctx->fail(SFM_GPU_ERR_ARG, "row step %zu < row bytes", step);
    const void* src = desc;
    size_t src_step = step;
    if (!desc_on_device && n > 0) {
        // staging buffer is reused: wait for earlier users of it
        SFM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        rc = upload_rows(ctx, ctx->raw_q, desc, n, step, rowbytes);
        if (rc) return rc;
        src = ctx->raw_q.p;
        src_step = rowbytes;
    }
    std::vector<PrepJob> jobs(1);
    jobs[0].src = (const unsigned char*)src;
    jobs[0].step = src_step;
    jobs[0].kp_src = (desc_on_device && keypoints_xy) ? (const float2*)keypoints_xy : nullptr;
    jobs[0].img = d;
    rc = run_prep_jobs(ctx, jobs, dtype, 1);
    if (rc) return rc;
    if (keypoints_xy && n > 0 && !desc_on_device)
        SFM_HIP(ctx, hipMemcpyAsync(d.kp, keypoints_xy, (size_t)n * sizeof(float2), hipMemcpyHostToDevice,
                                    ctx->stream));
    if (!desc_on_device) SFM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SFM_GPU_OK;
}

int sfm_gpu_image_prepare_many(sfm_gpu_ctx* ctx, const void* const* desc_dev, const int32_t* n,
                               size_t step, int dim, int dtype, const float* const* keypoints_dev,
                               void* const* packed_dev, int count)
{
    if (!ctx || count < 0 || (count > 0 && (!desc_dev || !n || !packed_dev)))
        return ctx ? ctx->fail(SFM_GPU_ERR_ARG, "null pointer or negative count") : SFM_GPU_ERR_ARG;
    if (count == 0) return SFM_GPU_OK;
    SFM_HIP(ctx, hipSetDevice(ctx->device));
    const size_t rowbytes = dtype == SFM_GPU_DESC_F32 ? DIM * 4 : DIM;
    if (step < rowbytes) return ctx->fail(SFM_GPU_ERR_ARG, "row step %zu < row bytes", step);
    std::vector<PrepJob> jobs(count);
    for (int i = 0; i < count; ++i) {
        int rc = check_knn_args(ctx, n[i], 0, dim, dtype, 2);
        if (rc) return rc;
        if (n[i] >= (1 << 18)) return ctx->fail(SFM_GPU_ERR_TOO_MANY, "image rows %d >= 2^18", n[i]);
        if ((n[i] > 0 && !desc_dev[i]) || !packed_dev[i])
            return ctx->fail(SFM_GPU_ERR_ARG, "null pointer for image %d", i);
        jobs[i].src = (const unsigned char*)desc_dev[i];
        jobs[i].step = step;
        jobs[i].kp_src = keypoints_dev ? (const float2*)keypoints_dev[i] : nullptr;
        jobs[i].img = image_layout(packed_dev[i], n[i]);
    }
    return run_prep_jobs(ctx, jobs, dtype, 1);
}

int sfm_gpu_image_wrap(sfm_gpu_ctx* ctx, const void* packed_dev, int n, int has_keypoints,
                       sfm_gpu_image** out)
{
    if (!ctx || !out || !packed_dev || n < 0) return SFM_GPU_ERR_ARG;
    *out = nullptr;
    if (n >= (1 << 18)) return ctx->fail(SFM_GPU_ERR_TOO_MANY, "image rows %d >= 2^18", n);
    ImageHolder* h = new (std::nothrow) ImageHolder();
    if (!h) return ctx->fail(SFM_GPU_ERR_NOMEM, "host allocation failed");
    h->img.d = image_layout((void*)packed_dev, n);
    if (!has_keypoints) h->img.d.kp = nullptr;
    *out = &h->img;  // h->st owns nothing: destroy only frees the handle
    return SFM_GPU_OK;
}

int sfm_gpu_check_inputs(sfm_gpu_ctx* ctx)
{
    if (!ctx) return SFM_GPU_ERR_ARG;
    int32_t flag = 0;
    SFM_HIP(ctx, hipMemcpyAsync(&flag, ctx->flags.as<int32_t>() + 1, 4, hipMemcpyDeviceToHost, ctx->stream));
    SFM_HIP(ctx, hipMemsetAsync(ctx->flags.as<int32_t>() + 1, 0, 4, ctx->stream));
    SFM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (flag) return ctx->fail(SFM_GPU_ERR_NOT_U8, "descriptor values must be integers in 0..255");
    return SFM_GPU_OK;
}

// ---- measurement ------------------------------------------------------

int sfm_gpu_profile_enable(sfm_gpu_ctx* c, int on)
{
    if (!c) return SFM_GPU_ERR_ARG;
    c->profiling = on != 0;
    return SFM_GPU_OK;
}

int sfm_gpu_profile_select(sfm_gpu_ctx* c, const char* kernel)
{
    if (!c) return SFM_GPU_ERR_ARG;
    c->profile_only = kernel ? kernel : "";
    return SFM_GPU_OK;
}

static int drain_timers(sfm_gpu_ctx* c)
{
    SFM_HIP(c, hipStreamSynchronize(c->stream));
    for (auto& kv : c->timers) {
        for (auto& pr : kv.second.pending) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
                kv.second.total_ms += ms;
                kv.second.launches++;
            }
            c->event_pool.push_back(pr.first);
            c->event_pool.push_back(pr.second);
        }
        kv.second.pending.clear();
    }
    return SFM_GPU_OK;
}

int sfm_gpu_profile_reset(sfm_gpu_ctx* c)
{
    if (!c) return SFM_GPU_ERR_ARG;
    int rc = drain_timers(c);
    if (rc) return rc;
    c->timers.clear();
    return SFM_GPU_OK;
}

int sfm_gpu_profile_query(sfm_gpu_ctx* c, const char* kernel, double* total_ms, int64_t* launches)
{
    if (!c || !kernel) return SFM_GPU_ERR_ARG;
    int rc = drain_timers(c);
    if (rc) return rc;
    auto it = c->timers.find(kernel);
    if (total_ms) *total_ms = it == c->timers.end() ? 0.0 : it->second.total_ms;
    if (launches) *launches = it == c->timers.end() ? 0 : it->second.launches;
    return SFM_GPU_OK;
}

// ---- test hooks (declared in sfm_gpu_debug.h) ---------------------------

int sfm_gpu_debug_set_knn_variant(sfm_gpu_ctx* c, int variant)
{
    if (!c) return SFM_GPU_ERR_ARG;
#ifndef SFM_GPU_TIMING_BUILD
    if (variant >= 16) return c->fail(SFM_GPU_ERR_UNSUPPORTED, "timing ablations exist in the timing build only");
#endif
    if (variant >= 0 && variant % 16 > 5) return c->fail(SFM_GPU_ERR_UNSUPPORTED, "unknown sweep shape %d", variant % 16);
    c->knn_variant = variant < 0 ? -1 : variant;
    return SFM_GPU_OK;
}

// Test hook: overwrites the LDS of every CU with a pattern, in stream order.
int sfm_gpu_debug_poison_lds(sfm_gpu_ctx* ctx, uint32_t pattern)
{
    if (!ctx) return SFM_GPU_ERR_ARG;
    SFM_HIP(ctx, hipSetDevice(ctx->device));
    SFM_HIP(ctx, launch_lds_poison(ctx->stream, pattern));
    return SFM_GPU_OK;
}

// Test hook: the first `bytes` of the context's sweep-record buffer.
int sfm_gpu_debug_copy_records(sfm_gpu_ctx* ctx, void* out, size_t bytes)
{
    if (!ctx || !out || bytes > ctx->partial.cap) return SFM_GPU_ERR_ARG;
    SFM_HIP(ctx, hipSetDevice(ctx->device));
    SFM_HIP(ctx, hipMemcpyAsync(out, ctx->partial.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    SFM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SFM_GPU_OK;
}

// Timing build only: in-kernel stamps of the last 16x16 sweep (8 words per workgroup).
int sfm_gpu_debug_sweep_stamps(sfm_gpu_ctx* ctx, uint64_t* out, int n_wg)
{
    if (!ctx || !out || n_wg < 0) return SFM_GPU_ERR_ARG;
    SFM_HIP(ctx, hipSetDevice(ctx->device));
    SFM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const hipError_t e = read_sweep_stamps((unsigned long long*)out, n_wg);
    if (e == hipErrorNotSupported) return ctx->fail(SFM_GPU_ERR_UNSUPPORTED, "stamps exist in the timing build only");
    SFM_HIP(ctx, e);
    return SFM_GPU_OK;
}

// Timing build only: stamps of the scoring wavefronts since the last call (4 words each).
int sfm_gpu_debug_score_stamps(sfm_gpu_ctx* ctx, uint64_t* out, int n_max, int32_t* count)
{
    if (!ctx || !out || !count || n_max < 0) return SFM_GPU_ERR_ARG;
    SFM_HIP(ctx, hipSetDevice(ctx->device));
    SFM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int c = 0;
    const hipError_t e = read_score_stamps((unsigned long long*)out, n_max, &c);
    if (e == hipErrorNotSupported) return ctx->fail(SFM_GPU_ERR_UNSUPPORTED, "stamps exist in the timing build only");
    SFM_HIP(ctx, e);
    *count = c;
    return SFM_GPU_OK;
}

// Number of queries the last k-NN call on this context sent through the
// exhaustive float-ranked pass (first job of the call).
int sfm_gpu_debug_redo_count(sfm_gpu_ctx* ctx, int32_t* out)
{
    if (!ctx || !out || !ctx->redo.p) return SFM_GPU_ERR_ARG;
    SFM_HIP(ctx, hipMemcpyAsync(out, ctx->redo.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    SFM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SFM_GPU_OK;
}

// c[32*32] = a[32][32] * b^T where b is given as [32 columns][32 k]; host pointers.
int sfm_gpu_debug_mfma_probe(sfm_gpu_ctx* ctx, const int8_t* a, const int8_t* b, int32_t* c)
{
    if (!ctx || !a || !b || !c) return SFM_GPU_ERR_ARG;
    SFM_HIP(ctx, hipSetDevice(ctx->device));
    SFM_HIP(ctx, ctx->rs_misc.reserve(1024 + 1024 + 4096));
    int8_t* da = ctx->rs_misc.as<int8_t>();
    int8_t* db = da + 1024;
    int32_t* dc = (int32_t*)(da + 2048);
    SFM_HIP(ctx, hipMemcpyAsync(da, a, 1024, hipMemcpyHostToDevice, ctx->stream));
    SFM_HIP(ctx, hipMemcpyAsync(db, b, 1024, hipMemcpyHostToDevice, ctx->stream));
    SFM_HIP(ctx, launch_mfma_layout_probe(ctx->stream, da, db, dc));
    SFM_HIP(ctx, hipMemcpyAsync(c, dc, 4096, hipMemcpyDeviceToHost, ctx->stream));
    SFM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SFM_GPU_OK;
}

// Number of integers n < 2^24 whose plain device sqrtf differs from the
// correctly rounded float square root.
int sfm_gpu_debug_sqrt_probe(sfm_gpu_ctx* ctx, int32_t* mismatches)
{
    if (!ctx || !mismatches) return SFM_GPU_ERR_ARG;
    SFM_HIP(ctx, hipSetDevice(ctx->device));
    SFM_HIP(ctx, ctx->rs_misc.reserve(256));
    SFM_HIP(ctx, hipMemsetAsync(ctx->rs_misc.p, 0, 4, ctx->stream));
    SFM_HIP(ctx, launch_sqrt_probe(ctx->stream, ctx->rs_misc.as<int32_t>()));
    SFM_HIP(ctx, hipMemcpyAsync(mismatches, ctx->rs_misc.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    SFM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SFM_GPU_OK;
}

// c[16][16] = a[16][64] * b^T, b given as [16 columns][64 k]; host pointers.
int sfm_gpu_debug_mfma16_probe(sfm_gpu_ctx* ctx, const int8_t* a, const int8_t* b, int32_t* c)
{
    if (!ctx || !a || !b || !c) return SFM_GPU_ERR_ARG;
    SFM_HIP(ctx, hipSetDevice(ctx->device));
    SFM_HIP(ctx, ctx->rs_misc.reserve(1024 + 1024 + 1024));
    int8_t* da = ctx->rs_misc.as<int8_t>();
    int8_t* db = da + 1024;
    int32_t* dc = (int32_t*)(da + 2048);
    SFM_HIP(ctx, hipMemcpyAsync(da, a, 1024, hipMemcpyHostToDevice, ctx->stream));
    SFM_HIP(ctx, hipMemcpyAsync(db, b, 1024, hipMemcpyHostToDevice, ctx->stream));
    SFM_HIP(ctx, launch_mfma16_layout_probe(ctx->stream, da, db, dc));
    SFM_HIP(ctx, hipMemcpyAsync(c, dc, 1024, hipMemcpyDeviceToHost, ctx->stream));
    SFM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SFM_GPU_OK;
}

}  // extern "C"
