// Descriptor preparation, int8-MFMA distance + fused top-2, and match
// finalisation for gfx950.
//
// Replaces the arithmetic behind matcher->knnMatch(descL, descR, matches, 2)
// (/root/reference/main.cpp:184) with cv::BFMatcher(NORM_L2) semantics.
//
// Arithmetic.  Descriptor values v are integers 0..255.  Every row is stored
// once as b = v - 128 (int8).  A train row enters the MFMA as b, a query row as
// ~b = 127 - v, so the int32 accumulator holds dot = sum (127-q)(t-128) and
//     d2(q,t) = sum (q-t)^2 = Qn + Tn + 2*dot,
//     Qn = sum (q-128)^2,   Tn = sum (t-127)^2 - 128.
// For one query the ranking key is Tn + 2*dot.  The accumulator is preloaded
// with cin[t] = floor(Tn/2) + KEY_OFFSET, so the MFMA chain leaves
// x = dot + cin in registers and key = 2x + (Tn & 1) (+ a constant).  Rows with
// x above the lane's current second-best x cannot enter the top-2, which a
// min3 tree over the 16 accumulators and one wave-wide branch decide; only
// tiles that pass take the update path, where
//     p = (x << 9) + low[t],  low[t] = (Tn & 1) << 8 | (t & 255)
// packs key and row-in-chunk into one u32 and (min, med3) keep the two
// smallest.  Ties on the key resolve to the lowest train row, as OpenCV's
// ascending insertion scan does.
#include <type_traits>

#include "sfm_internal.h"

namespace sfm {

namespace {

__device__ __forceinline__ uint32_t umin32(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t umax32(uint32_t a, uint32_t b) { return a > b ? a : b; }
// middle of (a, b, p); lowers to v_med3_u32
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t p)
{
    return umax32(umin32(a, b), umin32(umax32(a, b), p));
}

// ------------------------------------------------------------ prep_rows ----
// One wave per row: converts 128 values to int8 (v - 128), validates them and
// writes the per-row constants.  Rows n..rows_alloc-1 are written as padding.
template <int DTYPE>
__global__ void __launch_bounds__(256) prep_rows_kernel(const PrepJob* __restrict__ jobs, int32_t* err_flag)
{
    const PrepJob job = jobs[blockIdx.y];
    const ImageDev img = job.img;
    const unsigned char* src = job.src;
    const size_t step = job.step;
    const int n = img.n;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= img.rows_alloc) return;
    int v0 = 128, v1 = 128;
    bool bad = false;
    if (row < n) {
        if (DTYPE == SFM_GPU_DESC_F32) {
            const float2 f = *(const float2*)(src + (size_t)row * step + lane * 8);
            v0 = (int)f.x;
            v1 = (int)f.y;
            bad = !(f.x >= 0.f && f.x <= 255.f && f.y >= 0.f && f.y <= 255.f) ||
                  (float)v0 != f.x || (float)v1 != f.y;
        } else {
            const unsigned short u = *(const unsigned short*)(src + (size_t)row * step + lane * 2);
            v0 = u & 255;
            v1 = u >> 8;
        }
    }
    if (bad) {
        v0 = v1 = 128;
        atomicOr(err_flag, 1);
    }
    const int b0 = v0 - 128, b1 = v1 - 128;
    const unsigned short packed = (unsigned short)((b0 & 255) | ((b1 & 255) << 8));
    *(unsigned short*)(img.desc + (size_t)row * DIM + lane * 2) = packed;
    int n2 = b0 * b0 + b1 * b1;
    int s = b0 + b1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        n2 += __shfl_xor(n2, off);
        s += __shfl_xor(s, off);
    }
    if (lane == 0) {
        if (row < n) {
            const int tn = n2 + 2 * s;  // = sum (v-127)^2 - 128, in [-128, 2097024]
            img.qn[row] = n2;
            img.cin[row] = (tn >> 1) + KEY_OFFSET;  // arithmetic shift = floor
            img.low[row] = ((tn & 1) << 8) | (row & 255);
            if (job.kp_src && img.kp) img.kp[row] = job.kp_src[row];
        } else {
            img.qn[row] = 0;
            img.cin[row] = PAD_CIN;
            img.low[row] = row & 255;
        }
    }
}

// ------------------------------------------------------------- knn2_mfma ----
constexpr int ST = 128;                      // train rows staged in LDS per step
constexpr int LDS_DESC = ST * DIM;           // descriptors of one stage
constexpr int LDS_STAGE = LDS_DESC + ST * 8; // + cin[ST] + low[ST]

// Compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>).
// Keeps per-query-block state arrays indexed by constants only, so they stay
// in registers.
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// 16 bytes per lane from global memory straight into LDS at
// lds_wave_base + 16 * lane (the LDS side of the instruction is not a scatter).
__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int QB, int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
knn2_mfma_kernel(const KnnJob* __restrict__ jobs, int n_splits, int chunks_per_split, int ablate)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * LDS_STAGE];

    const KnnJob job = jobs[blockIdx.z];
    constexpr int QPB = WAVES * QB * 32;
    if ((int)blockIdx.x * QPB >= job.n_query) return;
    const int c0 = blockIdx.y * chunks_per_split;
    const int c1 = min(c0 + chunks_per_split, job.t_chunks);

    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int col = lane & 31;  // query within a block of 32 (MFMA column)
    const int h = lane >> 5;    // which half of the K bytes / which row quartets
    const int q_base = (blockIdx.x * WAVES + wave) * (QB * 32);

    // Query fragments stay in registers for the whole sweep: lane (col, h)
    // holds bytes [64h, 64h+64) of its query row, complemented.
    v4i bq[QB][4];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const v4i* qp = (const v4i*)(job.q_desc + (size_t)(q_base + qb * 32 + col) * DIM + 64 * h);
#pragma unroll
        for (int s = 0; s < 4; ++s) bq[qb][s] = ~qp[s];
    }
    // per-query-block state as register vectors (constant indices only)
    typedef uint32_t uQ __attribute__((ext_vector_type(QB)));
    typedef int32_t iQ __attribute__((ext_vector_type(QB)));
    uQ m1, m2, thr;
    iQ g1, g2;
    static_for<QB>([&](auto qbc) {
        constexpr int qb = decltype(qbc)::value;
        m1[qb] = m2[qb] = 0xFFFFFFFFu;
        thr[qb] = 0x7FFFFFu;
        g1[qb] = g2[qb] = -1;
    });

    // Stage image in LDS: row-major 128-B rows of eight 16-B slots, slot cs of
    // row r holding source chunk cs ^ ((r >> 1) & 7), which makes the
    // ds_read_b128 fragment reads conflict-free; then cin[ST], low[ST].  The
    // DMA writes LDS linearly, so the permutation is applied to the source
    // address.  One wave-instruction moves 8 rows.
    const int dma_r = lane >> 3;                 // row within the 8-row piece
    const int dma_cs = lane & 7;
    auto issue_stage = [&](int buf, int row0) {
        unsigned char* dst = lds + buf * LDS_STAGE;
#pragma unroll
        for (int k = wave; k < ST / 8; k += WAVES) {
            const int r = 8 * k + dma_r;
            const int cc = dma_cs ^ ((r >> 1) & 7);
            glds16(job.t_desc + (size_t)(row0 + r) * DIM + cc * 16, dst + k * 1024);
        }
        if (wave == WAVES - 1) {
            const int32_t* src = lane < 32 ? job.t_cin + row0 + lane * 4 : job.t_low + row0 + (lane - 32) * 4;
            glds16(src, dst + LDS_DESC);
        }
    };

    // When the train sweep is split over several workgroups they tighten each
    // other's thresholds through job.thr_glob[q], an upper bound of query q's
    // final second-best x: a row above it cannot be in the top-2 whichever
    // workgroup saw the rows that set it.  Values only ever decrease and a
    // stale read is merely a looser bound, so no ordering is needed.
    const bool coop = n_splits > 1;
    uQ pend, pub;
    static_for<QB>([&](auto qbc) {
        constexpr int qb = decltype(qbc)::value;
        pend[qb] = pub[qb] = 0x7FFFFFu;
    });

    unsigned dbg_checks = 0, dbg_trig = 0, dbg_lanes = 0;  // ablate >= 4: update-path statistics
    struct Frag {
        v4i a[4];
        v4i c[4];
    };
    auto load_frag = [&](Frag& f, const unsigned char* buf, int tile) {
        const int r = tile * 32 + col;
        const int swz = (r >> 1) & 7;
#pragma unroll
        for (int k = 0; k < 4; ++k) f.a[k] = *(const v4i*)(buf + (r * 8 + ((4 * h + k) ^ swz)) * 16);
        // accumulator register 4g+i of this lane is train row 8g + 4h + i
#pragma unroll
        for (int g = 0; g < 4; ++g)
            f.c[g] = *(const v4i*)(buf + LDS_DESC + (tile * 32 + 8 * g + 4 * h) * 4);
    };
    auto compute_tile = [&](const Frag& f, const unsigned char* low_tile, int32_t chunk_base) {
        v16i cinv;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            cinv[4 * g + 0] = f.c[g][0];
            cinv[4 * g + 1] = f.c[g][1];
            cinv[4 * g + 2] = f.c[g][2];
            cinv[4 * g + 3] = f.c[g][3];
        }
        static_for<QB>([&](auto qbc) {
            constexpr int qb = decltype(qbc)::value;
            v16i acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.a[0], bq[qb][0], cinv, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.a[1], bq[qb][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.a[2], bq[qb][2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.a[3], bq[qb][3], acc, 0, 0, 0);
            if (ablate == 2) {  // timing experiment: MFMA only
                asm volatile("" ::"v"(acc));
                return;
            }
            uint32_t tg[4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
                tg[g] = umin32(umin32(umin32((uint32_t)acc[4 * g], (uint32_t)acc[4 * g + 1]),
                                      (uint32_t)acc[4 * g + 2]), (uint32_t)acc[4 * g + 3]);
            const uint32_t t = umin32(umin32(umin32(tg[0], tg[1]), tg[2]), tg[3]);
            if (ablate == 1) {  // timing experiment: never update
                asm volatile("" ::"v"(t));
                return;
            }
            if (ablate >= 4) dbg_checks++;
            if (ablate == 3 || __builtin_amdgcn_ballot_w64(t <= thr[qb]) != 0) {
                if (ablate >= 4) {
                    dbg_trig++;
                    dbg_lanes += __popcll(__builtin_amdgcn_ballot_w64(t <= thr[qb]));
                }
                // Update path, one quartet of rows at a time, skipping quartets
                // that hold no candidate for any lane.
                const uint32_t om1 = m1[qb], om2 = m2[qb];
                const int32_t og1 = g1[qb], og2 = g2[qb];
                uint32_t ma = om1, mb = om2;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (__builtin_amdgcn_ballot_w64(tg[g] <= thr[qb]) != 0) {
                        const v4i lw = *(const v4i*)(low_tile + (8 * g + 4 * h) * 4);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const uint32_t p = ((uint32_t)acc[4 * g + i] << 9) + (uint32_t)lw[i];
                            const uint32_t nb = umed3(ma, mb, p);
                            ma = umin32(ma, p);
                            mb = nb;
                        }
                    }
                }
                // Recover which rows the survivors came from.  Entries carried over
                // from earlier chunks have their index field cleared, so an equal key
                // from the current chunk never displaces them (lowest train row wins
                // ties).
                const int32_t ia = chunk_base + (int32_t)(ma & 255u);
                const int32_t ib = chunk_base + (int32_t)(mb & 255u);
                const bool new1 = ma != om1;
                const int32_t ng2_if_new1 = (mb == om1) ? og1 : ib;
                const int32_t ng2_if_old1 = (mb != om2) ? ib : og2;
                g1[qb] = new1 ? ia : og1;
                g2[qb] = new1 ? ng2_if_new1 : ng2_if_old1;
                m1[qb] = ma;
                m2[qb] = mb;
                // Lanes l and l+32 serve the same query: a row that cannot beat the
                // better of their two second-best keys cannot be in the query's top-2.
                const uint32_t mo = (uint32_t)__shfl_xor((int)mb, 32);
                thr[qb] = umin32(thr[qb], umin32(mb, mo) >> 9);
            }
        });
    };

    const int n_stage = (c1 - c0) * (CHUNK / ST);
    if (n_stage > 0) issue_stage(0, c0 * CHUNK);
    for (int s = 0; s < n_stage; ++s) {
        // Hand-placed wait + raw barrier (the compiler does not order LDS-DMA by
        // itself): this wave's DMA of stage s and its threshold loads have
        // landed (vmcnt), then everyone's have (barrier), and every wave is done
        // reading the buffer that stage s+1 is about to overwrite.  The
        // threshold merge sits between the two so that no wait for it can end
        // up behind the next DMA issue, where it would wait for that DMA too.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        static_for<QB>([&](auto qbc) {
            constexpr int qb = decltype(qbc)::value;
            uint32_t v = umin32(thr[qb], pend[qb]);
            asm volatile("" : "+v"(v));  // pins the merge (and its operand wait) here
            thr[qb] = v;
        });
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < n_stage) issue_stage((s + 1) & 1, c0 * CHUNK + (s + 1) * ST);
        if (coop) {
            static_for<QB>([&](auto qbc) {
                constexpr int qb = decltype(qbc)::value;
                typedef __attribute__((address_space(1))) uint32_t gu32;
                gu32* gp = (gu32*)(job.thr_glob + q_base + qb * 32 + col);
                if (thr[qb] < pub[qb]) {
                    __hip_atomic_fetch_min(gp, thr[qb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    pub[qb] = thr[qb];
                }
                pend[qb] = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            });
        }
        const unsigned char* buf = lds + (s & 1) * LDS_STAGE;
        const int32_t chunk_base = (c0 + s / (CHUNK / ST)) * CHUNK;
        const unsigned char* buf_low = buf + LDS_DESC + ST * 4;
        Frag f0, f1;
        load_frag(f0, buf, 0);
        load_frag(f1, buf, 1);
        compute_tile(f0, buf_low, chunk_base);
        load_frag(f0, buf, 2);
        compute_tile(f1, buf_low + 32 * 4, chunk_base);
        load_frag(f1, buf, 3);
        compute_tile(f0, buf_low + 64 * 4, chunk_base);
        compute_tile(f1, buf_low + 96 * 4, chunk_base);
        if ((s + 1) % (CHUNK / ST) == 0) {
            // entries now belong to an earlier chunk: clear their row-in-chunk
            static_for<QB>([&](auto qbc) {
                constexpr int qb = decltype(qbc)::value;
                m1[qb] &= ~0xFFu;
                m2[qb] &= ~0xFFu;
            });
        }
    }

    if (ablate >= 4 && lane == 0) {
        atomicAdd(job.thr_glob + job.q_rows_alloc + 0, dbg_checks);
        atomicAdd(job.thr_glob + job.q_rows_alloc + 1, dbg_trig);
        atomicAdd(job.thr_glob + job.q_rows_alloc + 2, dbg_lanes);
    }
    // Lanes l and l+32 hold the same query over disjoint rows: merge, then
    // lane l < 32 writes the split's partial result.
    static_for<QB>([&](auto qbc) {
        constexpr int qb = decltype(qbc)::value;
        const unsigned long long a1 = ((unsigned long long)(m1[qb] >> 8) << 32) | (uint32_t)g1[qb];
        const unsigned long long a2 = ((unsigned long long)(m2[qb] >> 8) << 32) | (uint32_t)g2[qb];
        const unsigned long long b1 = __shfl_xor(a1, 32);
        const unsigned long long b2 = __shfl_xor(a2, 32);
        const bool afirst = a1 < b1;
        const unsigned long long f = afirst ? a1 : b1;
        const unsigned long long s2 = afirst ? (b1 < a2 ? b1 : a2) : (a1 < b2 ? a1 : b2);
        if (h == 0) {
            const int q = q_base + qb * 32 + col;
            uint4 o;
            o.x = (uint32_t)(f >> 32);
            o.y = (uint32_t)f;
            o.z = (uint32_t)(s2 >> 32);
            o.w = (uint32_t)s2;
            job.partial[(size_t)blockIdx.y * job.q_rows_alloc + q] = o;
        }
    });
}

// ---------------------------------------------------------- finalisation ----

// sqrtf of an integer < 2^24, correctly rounded, independent of the accuracy of
// the device's double sqrt: the candidate is checked against the two
// neighbouring rounding boundaries in exact double arithmetic.
__device__ __forceinline__ float sqrt_exact_u32(uint32_t n)
{
    const double dn = (double)n;
    float f = (float)sqrt(dn);
    if (n == 0) return 0.f;
    const float up = __uint_as_float(__float_as_uint(f) + 1u);
    const float dnf = __uint_as_float(__float_as_uint(f) - 1u);
    const double hi = 0.5 * ((double)f + (double)up);
    const double lo = 0.5 * ((double)f + (double)dnf);
    if (hi * hi < dn) f = up;
    else if (lo * lo > dn) f = dnf;
    return f;
}

__global__ void __launch_bounds__(256)
knn_finalize_kernel(const KnnJob* __restrict__ jobs, int n_splits, int k)
{
    const KnnJob job = jobs[blockIdx.y];
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= job.n_query) return;
    unsigned long long f = ~0ull, s = ~0ull;
    for (int sp = 0; sp < n_splits; ++sp) {
        const uint4 p = job.partial[(size_t)sp * job.q_rows_alloc + q];
        const unsigned long long b1 = ((unsigned long long)p.x << 32) | p.y;
        const unsigned long long b2 = ((unsigned long long)p.z << 32) | p.w;
        const bool afirst = f < b1;
        const unsigned long long nf = afirst ? f : b1;
        const unsigned long long ns = afirst ? (b1 < s ? b1 : s) : (f < b2 ? f : b2);
        f = nf;
        s = ns;
    }
    const int qn = job.q_qn[q];
    int cnt = 0;
    bool redo = false;
    const unsigned long long e[2] = {f, s};
    for (int j = 0; j < k; ++j) {
        sfm_gpu_dmatch m;
        m.queryIdx = q;
        m.imgIdx = 0;
        const uint32_t key = (uint32_t)(e[j] >> 32);
        const int32_t idx = (int32_t)(uint32_t)e[j];
        if (key != KEY_NONE && idx >= 0 && idx < job.n_train) {
            const int32_t d2 = qn + (int32_t)key - 2 * KEY_OFFSET;
            m.trainIdx = idx;
            m.distance = sqrt_exact_u32((uint32_t)d2);
            // From 2^22 upwards different integers can share one float sqrt,
            // and the ranking is defined on the float: re-rank those queries.
            if (j == 1 && d2 >= (1 << 22)) redo = true;
            cnt++;
        } else {
            m.trainIdx = -1;
            m.distance = __uint_as_float(0x7f7fffffu);  // FLT_MAX
        }
        job.out[(size_t)q * k + j] = m;
    }
    // with k == 1 the second neighbour still decides whether rank 1 is safe
    if (k == 1) {
        const uint32_t key2 = (uint32_t)(s >> 32);
        const int32_t idx2 = (int32_t)(uint32_t)s;
        if (key2 != KEY_NONE && idx2 >= 0 && idx2 < job.n_train &&
            qn + (int32_t)key2 - 2 * KEY_OFFSET >= (1 << 22))
            redo = true;
    }
    job.counts[q] = cnt;
    if (redo) {
        const int slot = atomicAdd(job.redo, 1);
        job.redo[1 + slot] = q;
    }
}

// Float-ranked exhaustive pass for the queries flagged above: ranks train rows
// by (sqrtf(d2) as float, row), exactly the reference ordering, at any d2.
__global__ void __launch_bounds__(256) knn_redo_kernel(const KnnJob* __restrict__ jobs, int k)
{
    __shared__ unsigned long long sh[2 * 256];
    const KnnJob job = jobs[blockIdx.y];
    const int n_redo = job.redo[0];
    for (int item = blockIdx.x; item < n_redo; item += gridDim.x) {
        const int q = job.redo[1 + item];
        const int8_t* qrow = job.q_desc + (size_t)q * DIM;
        unsigned long long f = ~0ull, s = ~0ull;
        for (int t = threadIdx.x; t < job.n_train; t += 256) {
            const int8_t* trow = job.t_desc + (size_t)t * DIM;
            int d2 = 0;
            for (int d = 0; d < DIM; d += 4) {
                const int qa = *(const int*)(qrow + d);
                const int ta = *(const int*)(trow + d);
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int df = (int)(int8_t)(qa >> (8 * b)) - (int)(int8_t)(ta >> (8 * b));
                    d2 += df * df;
                }
            }
            const unsigned long long key =
                ((unsigned long long)__float_as_uint(sqrt_exact_u32((uint32_t)d2)) << 32) | (uint32_t)t;
            if (key < f) { s = f; f = key; }
            else if (key < s) { s = key; }
        }
        sh[2 * threadIdx.x] = f;
        sh[2 * threadIdx.x + 1] = s;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if ((int)threadIdx.x < off) {
                const unsigned long long a1 = sh[2 * threadIdx.x], a2 = sh[2 * threadIdx.x + 1];
                const unsigned long long b1 = sh[2 * (threadIdx.x + off)], b2 = sh[2 * (threadIdx.x + off) + 1];
                const bool afirst = a1 < b1;
                sh[2 * threadIdx.x] = afirst ? a1 : b1;
                sh[2 * threadIdx.x + 1] = afirst ? (b1 < a2 ? b1 : a2) : (a1 < b2 ? a1 : b2);
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            for (int j = 0; j < k; ++j) {
                const unsigned long long e = sh[j];
                if (e != ~0ull) {
                    sfm_gpu_dmatch m;
                    m.queryIdx = q;
                    m.imgIdx = 0;
                    m.trainIdx = (int32_t)(uint32_t)e;
                    m.distance = __uint_as_float((uint32_t)(e >> 32));
                    job.out[(size_t)q * k + j] = m;
                }
            }
        }
        __syncthreads();
    }
}

// C = A(32x32 i8) * B(32x32 i8) through one v_mfma_i32_32x32x32_i8, operands
// loaded exactly like the distance kernel loads them; used by a test that
// checks the lane/register mapping assumed above against a plain product.
__global__ void __launch_bounds__(64) mfma_layout_probe_kernel(const int8_t* A, const int8_t* B, int32_t* C)
{
    const int lane = threadIdx.x, col = lane & 31, h = lane >> 5;
    // A: row-major [32 rows][32 k]; B given as [32 cols][32 k] (each "column" contiguous)
    const v4i a = *(const v4i*)(A + col * 32 + 16 * h);
    const v4i b = *(const v4i*)(B + col * 32 + 16 * h);
    v16i c;
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = 0;
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        C[row * 32 + col] = c[i];
    }
}

}  // namespace

// ---------------------------------------------------------------- hosts ----

hipError_t launch_prep_rows(hipStream_t s, const PrepJob* jobs_dev, int n_jobs, int rows_alloc_max,
                            int dtype, int32_t* err_flag)
{
    const dim3 grid((rows_alloc_max + 3) / 4, n_jobs);
    if (dtype == SFM_GPU_DESC_F32)
        hipLaunchKernelGGL(prep_rows_kernel<SFM_GPU_DESC_F32>, grid, dim3(256), 0, s, jobs_dev, err_flag);
    else
        hipLaunchKernelGGL(prep_rows_kernel<SFM_GPU_DESC_U8>, grid, dim3(256), 0, s, jobs_dev, err_flag);
    return hipGetLastError();
}

// Workgroup shapes of the distance kernel: {query blocks per wave, waves}.
//   0: 4 x 4 (512 queries / workgroup)   1: 2 x 4 (256)
//   2: 2 x 8 (512)                       3: 4 x 8 (1024)
static const int kVariantQPB[4] = {512, 256, 512, 1024};

KnnPlan plan_knn(int n_query_max, int n_train_max, int n_jobs, int force_variant)
{
    KnnPlan p;
    p.ablate = force_variant >= 0 ? force_variant / 16 : 0;
    if (force_variant >= 0) force_variant %= 16;
    p.variant = (force_variant >= 0 && force_variant < 4) ? force_variant : 0;
    p.q_per_block = kVariantQPB[p.variant];
    const int t_chunks = (n_train_max + CHUNK - 1) / CHUNK;
    const long long qblocks = (long long)((n_query_max + p.q_per_block - 1) / p.q_per_block) * n_jobs;
    // The train sweep is split only as far as needed to give every CU work:
    // short sweeps start with cold top-2 thresholds and update far more often.
    int splits = 1;
    while (qblocks * splits < 256 && splits * 2 <= t_chunks && t_chunks / (splits * 2) >= 4)
        splits *= 2;
    p.n_splits = splits;
    p.chunks_per_split = (t_chunks + splits - 1) / splits;
    if (p.chunks_per_split < 1) p.chunks_per_split = 1;
    p.n_splits = t_chunks > 0 ? (t_chunks + p.chunks_per_split - 1) / p.chunks_per_split : 1;
    return p;
}

hipError_t launch_knn2(hipStream_t s, const KnnJob* jobs_dev, int n_jobs, int q_rows_alloc_max,
                       const KnnPlan& plan)
{
    const int qblocks = (q_rows_alloc_max + plan.q_per_block - 1) / plan.q_per_block;
    const dim3 grid(qblocks, plan.n_splits, n_jobs);
    switch (plan.variant) {
        case 1:
            hipLaunchKernelGGL((knn2_mfma_kernel<2, 4>), grid, dim3(256), 0, s, jobs_dev, plan.n_splits,
                               plan.chunks_per_split, plan.ablate);
            break;
        case 2:
            hipLaunchKernelGGL((knn2_mfma_kernel<2, 8>), grid, dim3(512), 0, s, jobs_dev, plan.n_splits,
                               plan.chunks_per_split, plan.ablate);
            break;
        case 3:
            hipLaunchKernelGGL((knn2_mfma_kernel<4, 8>), grid, dim3(512), 0, s, jobs_dev, plan.n_splits,
                               plan.chunks_per_split, plan.ablate);
            break;
        default:
            hipLaunchKernelGGL((knn2_mfma_kernel<4, 4>), grid, dim3(256), 0, s, jobs_dev, plan.n_splits,
                               plan.chunks_per_split, plan.ablate);
            break;
    }
    return hipGetLastError();
}

hipError_t launch_knn_finalize(hipStream_t s, const KnnJob* jobs_dev, int n_jobs, int n_query_max,
                               int n_splits, int k)
{
    const dim3 grid((n_query_max + 255) / 256, n_jobs);
    hipLaunchKernelGGL(knn_finalize_kernel, grid, dim3(256), 0, s, jobs_dev, n_splits, k);
    return hipGetLastError();
}

hipError_t launch_knn_redo(hipStream_t s, const KnnJob* jobs_dev, int n_jobs, int k)
{
    hipLaunchKernelGGL(knn_redo_kernel, dim3(32, n_jobs), dim3(256), 0, s, jobs_dev, k);
    return hipGetLastError();
}

hipError_t launch_mfma_layout_probe(hipStream_t s, const int8_t* a, const int8_t* b, int32_t* c)
{
    hipLaunchKernelGGL(mfma_layout_probe_kernel, dim3(1), dim3(64), 0, s, a, b, c);
    return hipGetLastError();
}

}  // namespace sfm
