// Context object behind the C ABI: stream, grow-only device workspaces,
// per-kernel event timers.
#pragma once

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>

#include <vector>

#include "sfm_gpu.h"
#include "ransac_dev.h"
#include "sfm_internal.h"

namespace sfm {

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    // Grow-only.  Contents are not preserved across a growth.
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) {
            hipError_t e = hipFree(p);
            p = nullptr;
            cap = 0;
            if (e != hipSuccess) return e;
        }
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return e; }
        cap = want;
        return hipSuccess;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T> T* as() const { return (T*)p; }
};

// Pinned host staging memory (grow-only): small device-to-host copies into
// pageable memory are staged by the runtime and cost far more than the copy.
struct PinnedBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e != hipSuccess) { p = nullptr; return e; }
        cap = want;
        return hipSuccess;
    }
    void release()
    {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T> T* as() const { return (T*)p; }
};

// Bytes of the device block that holds an image of n rows:
// [desc rows*128 | cin rows*4 | qn rows*4 | keypoints rows*8].
inline int image_rows_alloc(int n) { return ((n > 0 ? n : 1) + ROW_ALIGN - 1) / ROW_ALIGN * ROW_ALIGN; }
inline size_t image_block_bytes(int n) { return (size_t)image_rows_alloc(n) * (DIM + 8 + 8); }
inline ImageDev image_layout(void* base, int n)
{
    ImageDev d;
    const int rows = image_rows_alloc(n);
    d.n = n;
    d.rows_alloc = rows;
    d.desc = (int8_t*)base;
    d.cin = (int32_t*)(d.desc + (size_t)rows * DIM);
    d.qn = d.cin + rows;
    d.kp = (float2*)(d.qn + rows);
    return d;
}

// Device image whose storage is one DevBuf (used for the temporaries of the
// host-pointer entry points and for sfm_gpu_image).
struct ImageStore {
    DevBuf buf;
    DevBuf kpbuf;
    ImageDev d;
    hipError_t shape(int n, bool with_kp)
    {
        int rows = ((n > 0 ? n : 1) + ROW_ALIGN - 1) / ROW_ALIGN * ROW_ALIGN;
        size_t bytes = (size_t)rows * (DIM + 8);
        hipError_t e = buf.reserve(bytes);
        if (e != hipSuccess) return e;
        d.n = n;
        d.rows_alloc = rows;
        d.desc = buf.as<int8_t>();
        d.cin = (int32_t*)(d.desc + (size_t)rows * DIM);
        d.qn = d.cin + rows;
        d.kp = nullptr;
        if (with_kp) {
            e = kpbuf.reserve((size_t)(n > 0 ? n : 1) * sizeof(float2));
            if (e != hipSuccess) return e;
            d.kp = kpbuf.as<float2>();
        }
        return hipSuccess;
    }
    void release()
    {
        buf.release();
        kpbuf.release();
    }
};

}  // namespace sfm

struct sfm_gpu_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    int knn_variant = -1;  // -1: planner decides
    int score_filter = 1;  // test hooks of the RANSAC scoring pass (sfm_gpu_debug.h)
    int score_prune = 1;
    int solver_split = -1;  // seven-point elimination: -1 by batch size, 0 one lane per hypothesis, 1 two
    // this device's copy of the library's static working memory (the calls that
    // must not allocate work there); resolved once per device at creation
    void* static_scratch = nullptr;

    sfm::DevBuf raw_q, raw_t;
    sfm::ImageStore tmp_q, tmp_t;
    sfm::DevBuf partial, out, counts, redo, jobs, flags, prepjobs;
    // ransac workspaces
    sfm::DevBuf rs_pts, rs_table, rs_models, rs_counts, rs_mask, rs_misc;
    sfm::DevBuf rs_models2;  // workspace of the pose's second RANSAC (sfm_gpu_match_pairs_pose)
    sfm::PinnedBuf pin_a, pin_b;  // host staging for results / counts / descriptors of jobs
    // match_pairs workspaces
    sfm::DevBuf mp_good, mp_pts, mp_ngood;
    // recover_pose workspaces
    sfm::DevBuf pose_in, pose_out;
    // pnp_ransac workspace
    sfm::DevBuf pnp_ws;
    int32_t dbg_pose_votes[4] = {0, 0, 0, 0};
    // Small host-to-device uploads (job descriptors) go through pinned staging
    // when a stage is set: a copy from pageable memory makes the host wait for
    // the stream, which would defeat queueing one batch behind another.
    sfm::PinnedBuf* up_stage = nullptr;
    size_t up_used = 0;
    hipError_t upload(void* dst, const void* src, size_t bytes)
    {
        const size_t need = (bytes + 63) & ~(size_t)63;
        if (up_stage && up_used + need <= up_stage->cap) {
            char* at = up_stage->as<char>() + up_used;
            memcpy(at, src, bytes);
            up_used += need;
            return hipMemcpyAsync(dst, at, bytes, hipMemcpyHostToDevice, stream);
        }
        return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream);
    }
    // Ring of pinned staging blocks for calls that only enqueue (image
    // conversion): a block is reused once the event recorded behind its upload
    // has passed.
    struct UpSlot {
        sfm::PinnedBuf buf;
        hipEvent_t ev = nullptr;
        bool armed = false;
    };
    UpSlot up_ring[4];
    int up_ring_next = 0;
    // sfm_gpu_match_pairs_begin / _end: up to two batches in flight
    struct MpSlot {
        hipEvent_t ev = nullptr;
        hipEvent_t rec_ev = nullptr;  // behind the batch's copy of its records to the host, when it has one
        bool rec_pending = false;
        sfm::PinnedBuf stage, res;
        int n_pairs = 0;
        // what it takes to run the batch again (adaptive RANSAC enqueued without
        // waiting: a pair whose loop wants more sample rows than the first batch)
        std::vector<sfm_gpu_image*> images;
        std::vector<int32_t> pairs;
        sfm_gpu_match_params params;
        sfm_gpu_dmatch* records_dev = nullptr;
        sfm_gpu_dmatch* records_host_dev = nullptr;
        sfm_gpu_dmatch* records_host = nullptr;  // the same buffer by the caller's address
        int match_capacity = 0;
        // sfm_gpu_match_pairs' own groups: device addresses of the group's pairs, and
        // the group's inlier records when the caller wants them on the host
        std::vector<sfm::PairDev> pds;
        sfm::DevBuf rec;
    };
    MpSlot mp_slot[2];
    int mp_next = 0;         // slot the next begin takes
    int mp_outstanding = 0;  // begun and not yet ended (the oldest is mp_next - mp_outstanding)
    MpSlot mp_sync[2];       // sfm_gpu_match_pairs keeps two of its launch groups on the stream
    // side stream on which a batch's inlier records travel to the host while the
    // next batch already runs on `stream` (created on first use)
    hipStream_t rec_stream = nullptr;
    // The copy of a batch's records is put on rec_stream when the NEXT batch begins, gated on
    // up_ev (recorded behind that batch's descriptor upload), or by _end if no batch followed:
    // the upload reads pinned host memory, and a read request issued while the records are
    // being written to the host queues behind all of those writes.
    MpSlot* rec_deferred = nullptr;
    hipEvent_t up_ev = nullptr;
    bool want_up_ev = false;
    // recorded behind the sweep and the neighbour extraction of every pair batch
    // once another context has asked to wait for it (sfm_gpu_stream_wait_front)
    hipEvent_t front_ev = nullptr;
    bool front_recorded = false;

    // device pointers of the last sfm_gpu_ransac call, for the test hooks
    const int32_t* dbg_table = nullptr;
    const int32_t* dbg_counts = nullptr;
    const int32_t* dbg_nmodels = nullptr;
    const double* dbg_models = nullptr;
    int dbg_hyp_cap = 0, dbg_m = 0;
    int last_hyp = -1;  // winning sample row of the last sfm_gpu_ransac call, -1: none

    bool profiling = false;
    std::string profile_only;  // empty: every named kernel is timed
    std::map<std::string, sfm::KernelTimer> timers;
    std::vector<hipEvent_t> event_pool;  // events are reused, not created per launch
    bool take_event(hipEvent_t* e)
    {
        if (!event_pool.empty()) {
            *e = event_pool.back();
            event_pool.pop_back();
            return true;
        }
        return hipEventCreate(e) == hipSuccess;
    }

    int fail(int code, const char* fmt, ...)
    {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        err = buf;
        return code;
    }
};

namespace sfm {

// rocTX ranges around the stages of a call, for `rocprofv3 --marker-trace`: on when
// the environment has SFM_GPU_ROCTX=1 and a rocTX library can be loaded
// (librocprofiler-sdk-roctx.so, else libroctx64.so); otherwise two null pointers.
struct RoctxApi {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
};
const RoctxApi& roctx_api();  // sfm_gpu_api.hip

// RAII bracket around one kernel launch for the per-kernel timers.
struct ProfScope {
    sfm_gpu_ctx* c;
    KernelTimer* t = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    bool marked = false;
    // attach: the scope holds one launch that takes the two events itself
    // (hipExtLaunchKernelGGL): nothing is recorded around it, the queue gets no barrier
    bool attach = false;
    ProfScope(sfm_gpu_ctx* ctx, const char* name, bool attach_ = false) : c(ctx), attach(attach_)
    {
        if (roctx_api().push) {
            roctx_api().push(name);
            marked = true;
        }
        if (!c || !c->profiling) return;
        if (!c->profile_only.empty() && c->profile_only != name) return;
        t = &c->timers[name];
        t->name = name;
        if (!c->take_event(&a) || !c->take_event(&b)) { t = nullptr; a = b = nullptr; return; }
        if (!attach) (void)hipEventRecord(a, c->stream);
    }
    ~ProfScope()
    {
        if (marked) roctx_api().pop();
        if (!t) return;
        if (!attach) (void)hipEventRecord(b, c->stream);
        t->pending.emplace_back(a, b);
    }
};

#define SFM_HIP(ctx, call)                                                                 \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess)                                                              \
            return (ctx)->fail(e_ == hipErrorOutOfMemory ? SFM_GPU_ERR_NOMEM : SFM_GPU_ERR_HIP, \
                               "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),     \
                               __FILE__, __LINE__);                                        \
    } while (0)

}  // namespace sfm
