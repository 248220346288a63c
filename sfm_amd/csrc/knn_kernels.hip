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
// x = dot + cin = floor(key/2) + const in registers, already comparable across
// train rows.  The sweep keeps, per query, the groups of 16 rows with the
// smallest minimum x; a second kernel computes exact integer distances for the
// rows of the two best groups and ranks them by (d2, row), which is OpenCV's
// (float distance, ascending train index) order whenever d2 < 2^22.  The rare
// queries where that shortcut is not provably exact go through an exhaustive
// float-ranked pass.
#include <algorithm>
#include <cstdlib>
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
// 8 lanes per row, 32 rows per workgroup: converts 128 values to int8
// (v - 128), validates them and writes the per-row constants.  Rows
// n..rows_alloc-1 are written as padding.  Every load instruction reads whole
// 128-B pieces of 8 rows and every store writes whole 128-B rows.
// Four float32 values that should be integers 0..255 -> their bytes, minus 128.
// v + 2^23 has the bit pattern 0x4B0000vv exactly when v is such an integer (or
// was rounded to one): the low byte is the value, the upper 24 bits are checked
// through the running OR / AND of all patterns, and (v + 2^23) - 2^23 - v, which
// is +0 only when nothing was rounded away, through the running OR of its bits
// (a NaN or an infinity fails the first check, -0 passes as 0).  Two packed
// additions per pair of values instead of a conversion each way and three
// compares per value: the conversion kernel runs beside other work and its
// vector instructions are what it takes from it.
typedef float pf2 __attribute__((ext_vector_type(2)));
struct ValueCheck {
    uint32_t any = 0u, all = 0xFFFFFFFFu, resid = 0u;
    __device__ __forceinline__ bool bad() const
    {
        return (any & 0xFFFFFF00u) != 0x4B000000u || (all & 0xFFFFFF00u) != 0x4B000000u || resid != 0u;
    }
};
__device__ __forceinline__ uint32_t pack4_checked(float4 f, ValueCheck& chk)
{
    const pf2 magic = {8388608.f, 8388608.f};
    const pf2 a = {f.x, f.y}, b = {f.z, f.w};
    const pf2 ta = a + magic, tb = b + magic;
    const pf2 da = (ta - magic) - a, db = (tb - magic) - b;
    const uint32_t t0 = __float_as_uint(ta.x), t1 = __float_as_uint(ta.y), t2 = __float_as_uint(tb.x),
                   t3 = __float_as_uint(tb.y);
    chk.any |= (t0 | t1) | (t2 | t3);
    chk.all &= (t0 & t1) & (t2 & t3);
    chk.resid |= (__float_as_uint(da.x) | __float_as_uint(da.y)) | (__float_as_uint(db.x) | __float_as_uint(db.y));
    const uint32_t lo = __builtin_amdgcn_perm(t1, t0, 0x0c0c0400u);  // (t0.b0, t1.b0, 0, 0)
    const uint32_t hi = __builtin_amdgcn_perm(t3, t2, 0x04000c0cu);  // (0, 0, t2.b0, t3.b0)
    return (lo | hi) ^ 0x80808080u;
}

template <int DTYPE, int RPL>
__global__ void __launch_bounds__(256) prep_rows_kernel(const PrepJob* __restrict__ jobs, int32_t* err_flag)
{
    // RPL rows per lane (rows 32 apart), all their loads issued before anything else.  Measured
    // with 1 / 2 / 4: 107 / 114.5 / 122.5 us alone, 0.805 / 0.842 / 0.830 ms per bench step: one row
    // per lane it is, with the loads ahead of the row's bounds branch (112 us and 0.796 ms before).
    constexpr int TS = 36;  // dwords per transposition row (16-B aligned, spreads banks)
    __shared__ __attribute__((aligned(16))) uint32_t tr[32 * TS];
    const PrepJob job = jobs[blockIdx.y];
    const ImageDev img = job.img;
    if ((int)blockIdx.x * 32 * RPL >= img.rows_alloc) return;
    const unsigned char* src = job.src;
    const size_t step = job.step;
    const int n = img.n;
    const int rl = threadIdx.x >> 3;  // row within the workgroup's block of 32
    const int seg = threadIdx.x & 7;
    const bool wide = (((uintptr_t)src | step) & 15) == 0;  // 16-B loads allowed
    typedef float f4v __attribute__((ext_vector_type(4)));
    f4v early[RPL][4] = {};
    if (DTYPE == SFM_GPU_DESC_F32 && wide && n > 0) {  // (rows past the end read row 0 and are ignored)
        // read once, never again: streaming loads leave the caches to the kernels this one runs
        // beside (123 -> 114.5 us alone, 0.8175 -> 0.8135 ms per bench step)
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
            const int row = blockIdx.x * 32 * RPL + 32 * r + rl;
            const float* fp0 = (const float*)(src + (size_t)(row < n ? row : 0) * step) + seg * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) early[r][j] = __builtin_nontemporal_load((const f4v*)(fp0 + j * 32));
        }
    }
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
        const int row = blockIdx.x * 32 * RPL + 32 * r + rl;
        uint32_t P[4] = {0u, 0u, 0u, 0u};  // value 128 -> byte 0
        bool bad = false;
        ValueCheck chk;
        if (row < n) {
            const unsigned char* rp = src + (size_t)row * step;
            if (DTYPE == SFM_GPU_DESC_F32) {
                // piece j of the lane: values 32j + 4seg .. +3
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float* fp = (const float*)rp + j * 32 + seg * 4;
                    float4 f;
                    if (wide) f = make_float4(early[r][j].x, early[r][j].y, early[r][j].z, early[r][j].w);
                    else f = make_float4(fp[0], fp[1], fp[2], fp[3]);
                    P[j] = pack4_checked(f, chk);
                }
                bad = chk.bad();
            } else {
                // the lane's 16 consecutive bytes
                const unsigned char* bp = rp + seg * 16;
                if (wide) {
                    const uint4 u = *(const uint4*)bp;
                    P[0] = u.x; P[1] = u.y; P[2] = u.z; P[3] = u.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        P[j] = (uint32_t)bp[4 * j] | ((uint32_t)bp[4 * j + 1] << 8) | ((uint32_t)bp[4 * j + 2] << 16) |
                               ((uint32_t)bp[4 * j + 3] << 24);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) P[j] ^= 0x80808080u;
            }
        }
        if (bad) {
            P[0] = P[1] = P[2] = P[3] = 0u;
            atomicOr(err_flag, 1);
        }
        int n2 = 0, sm = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            n2 = __builtin_amdgcn_sdot4((int)P[j], (int)P[j], n2, false);
            sm = __builtin_amdgcn_sdot4((int)P[j], 0x01010101, sm, false);
        }
#pragma unroll
        for (int off = 4; off > 0; off >>= 1) {
            n2 += __shfl_xor(n2, off);
            sm += __shfl_xor(sm, off);
        }
        if (row >= img.rows_alloc) continue;
        uint4 o;
        if (DTYPE == SFM_GPU_DESC_F32) {
            // the 8 lanes of a row belong to one wave: exchange through LDS needs
            // no workgroup barrier
            if (r > 0) __builtin_amdgcn_wave_barrier();  // (the row before has been read out)
#pragma unroll
            for (int j = 0; j < 4; ++j) tr[rl * TS + j * 8 + seg] = P[j];
            __builtin_amdgcn_wave_barrier();
            o = *(const uint4*)&tr[rl * TS + seg * 4];
        } else {
            o = make_uint4(P[0], P[1], P[2], P[3]);
        }
        *(uint4*)(img.desc + (size_t)row * DIM + seg * 16) = o;
        if (seg == 0) {
            if (row < n) {
                const int tn = n2 + 2 * sm;  // = sum (v-127)^2 - 128, in [-128, 2097024]
                img.qn[row] = n2;
                img.cin[row] = (tn >> 1) + KEY_OFFSET;  // arithmetic shift = floor
                if (job.kp_src && img.kp) img.kp[row] = job.kp_src[row];
            } else {
                img.qn[row] = 0;
                img.cin[row] = PAD_CIN;
            }
        }
    }
}

// ------------------------------------------------------------- knn2_mfma ----
constexpr int ST = 128;                      // train rows staged in LDS per step
constexpr int LDS_DESC = ST * DIM;           // descriptors of one stage
constexpr int LDS_STAGE = LDS_DESC + ST * 4; // + cin[ST]
constexpr int TILE_BITS = 9;                 // a sweep holds at most 512 tiles of 32 rows

// 16 bytes per lane from global memory straight into LDS at
// lds_wave_base + 16 * lane (the LDS side of the instruction is not a scatter).
__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Workgroup id -> (query block, train part, job).  Consecutive workgroup ids
// are dealt to the 8 XCDs round-robin and each XCD has its own L2, so the
// workgroups that stream the same train rows are given ids that are equal
// modulo 8 whenever the launch allows it:
//   mode 1 (jobs % 8 == 0)    all workgroups of a job run on one XCD
//   mode 2 (parts % 8 == 0)   all workgroups of a train part run on one XCD
//   mode 0                    plain order
struct WgCoord {
    int qblock, split, job;
};
__device__ __forceinline__ WgCoord decode_wg(uint32_t id, uint32_t qblocks, uint32_t splits, int mode)
{
    WgCoord c;
    if (mode == 1) {
        const uint32_t per_job = qblocks * splits;
        const uint32_t j = id >> 3;
        const uint32_t w = j % per_job;
        c.job = (int)((j / per_job) * 8 + (id & 7));
        c.qblock = (int)(w % qblocks);
        c.split = (int)(w / qblocks);
    } else if (mode == 2) {
        const uint32_t r = id / splits;
        c.split = (int)(id % splits);
        c.qblock = (int)(r % qblocks);
        c.job = (int)(r / qblocks);
    } else {
        const uint32_t r = id / qblocks;
        c.qblock = (int)(id % qblocks);
        c.split = (int)(r % splits);
        c.job = (int)(r / splits);
    }
    return c;
}

#define SFM_SB() __builtin_amdgcn_sched_barrier(0)

// In-kernel stamps of the 16x16 sweep, timing build only (in the product build
// no stamp executes): per workgroup {realtime at entry, cycle counter at entry,
// after the prologue, after the last stage, at exit, realtime at exit, XCC id}.
#ifdef SFM_GPU_TIMING_BUILD
constexpr int SWEEP_STAMP_WGS = 8192;
__device__ unsigned long long g_sweep_stamps[SWEEP_STAMP_WGS * 8];
#define SFM_STAMP(slot)                                                                         \
    do {                                                                                        \
        if (threadIdx.x == 0 && blockIdx.x < SWEEP_STAMP_WGS)                                   \
            g_sweep_stamps[blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memtime();             \
    } while (0)
#define SFM_STAMP_REAL(slot)                                                                    \
    do {                                                                                        \
        if (threadIdx.x == 0 && blockIdx.x < SWEEP_STAMP_WGS)                                   \
            g_sweep_stamps[blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memrealtime();         \
    } while (0)
#else
#define SFM_STAMP(slot) do { } while (0)
#define SFM_STAMP_REAL(slot) do { } while (0)
#endif

// Distance sweep.  Workgroup = WAVES waves, each owning QB blocks of 32
// queries whose fragments stay in registers; the workgroup streams one part of
// the train rows (at most 512 tiles of 32 rows) through LDS in stages of ST
// rows, three buffers deep.
//
// Per 32x32 tile the accumulators of one lane are x for ONE query against 16
// train rows (a "group": rows 8g+4h+i of the tile, g,i = 0..3).  The lane keeps
// the three smallest group minima seen so far, the best two tagged with their
// tile: 8 min/min3 + 1 pack + 4 min/max/med3 per tile and query block, no
// branch, no cross-lane traffic.  The exact neighbours are then picked by
// knn_rescan_kernel among the rows of the two best groups of each query.
//
// Schedule.  A "unit" is one train tile against one query block: 4 chained
// MFMAs.  The instruction stream is pinned (sched_barrier after every group):
// each MFMA of unit u is followed by a share of the 13 VALU operations that
// digest the accumulators of unit u-1 and by the LDS reads of the next tile's
// fragments, so the matrix pipe always has the next MFMA queued behind the one
// executing.  The stage hand-over needs one barrier per stage, placed inside
// the stream (after the first unit) where no wave has to wait for data: at that
// point every wave has finished reading stage s-1, stage s+1 landed long ago,
// and the DMA of stage s+2 may overwrite the buffer of stage s-1.
template <int QB, int WAVES, int ABLATE>
__global__ void __launch_bounds__(WAVES * 64, 8 / WAVES)  // two waves per SIMD: accumulators stay in VGPRs
knn2_mfma_kernel(const KnnJob* __restrict__ jobs, int chunks_per_split, int qblocks, int splits,
                 int map_mode)
{
    static_assert(QB == 1 || QB == 2 || QB == 4 || QB == 8, "QB must divide 8");
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * LDS_STAGE];

    const WgCoord wg = decode_wg(blockIdx.x, (uint32_t)qblocks, (uint32_t)splits, map_mode);
    const KnnJob job = jobs[wg.job];
    constexpr int QPB = WAVES * QB * 32;
    if (wg.qblock * QPB >= job.n_query) return;
    const int c0 = wg.split * chunks_per_split;
    const int c1 = min(c0 + chunks_per_split, job.t_chunks);

    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int col = lane & 31;  // query within a block of 32 (MFMA column)
    const int h = lane >> 5;    // which half of the K bytes / which row quartets
    const int q_base = (wg.qblock * WAVES + wave) * (QB * 32);

    // Query fragments: lane (col, h) holds bytes [64h, 64h+64) of its query
    // row, complemented.
    v4i bq[QB][4];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const v4i* qp = (const v4i*)(job.q_desc + (size_t)(q_base + qb * 32 + col) * DIM + 64 * h);
#pragma unroll
        for (int k = 0; k < 4; ++k) bq[qb][k] = ~qp[k];
    }
    uint32_t m1[QB], m2[QB], m3[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) m1[qb] = m2[qb] = m3[qb] = 0xFFFFFFFFu;

    // Stage image in LDS: row-major 128-B rows of eight 16-B slots, slot cs of
    // row r holding source chunk cs ^ ((r >> 1) & 7), which makes the
    // ds_read_b128 fragment reads conflict-free; then cin[ST].  The DMA writes
    // LDS linearly, so the permutation is applied to the source address.  One
    // wave-instruction moves 8 rows.
    const int dma_r = lane >> 3;
    const int dma_cs = lane & 7;
    constexpr int PIECES = ST / 8 / WAVES;  // 1-KiB pieces of a stage moved by each wave
    static_assert(WAVES % 2 == 0, "the swizzle term of a lane must not depend on the piece");
    // piece k = wave + j*WAVES covers rows 8k..8k+7; for r = 8k + dma_r the term
    // (r >> 1) & 7 is 4*(k & 1) | (dma_r >> 1), the same for every j: one lane
    // offset serves all pieces, the rest of the address is uniform
    const uint32_t dma_off =
        (uint32_t)((8 * wave + dma_r) * DIM + ((dma_cs ^ (4 * (wave & 1) | (dma_r >> 1))) << 4));
    const uint32_t cin_off = (uint32_t)(lane & (ST / 4 - 1)) * 16u;
    auto issue_piece = [&](int bufi, int row0, int j) {
        unsigned char* dst = lds + bufi * LDS_STAGE;
        const int8_t* base = job.t_desc + (size_t)row0 * DIM + (size_t)j * (WAVES * 1024);
        glds16(base + dma_off, dst + (wave + j * WAVES) * 1024);
    };
    auto issue_cin = [&](int bufi, int row0) {
        unsigned char* dst = lds + bufi * LDS_STAGE;
        const unsigned char* base = (const unsigned char*)(job.t_cin + row0);
        if (wave == WAVES - 1 && lane < ST / 4) glds16(base + cin_off, dst + LDS_DESC);
    };
    auto issue_stage = [&](int bufi, int row0) {
#pragma unroll
        for (int j = 0; j < PIECES; ++j) issue_piece(bufi, row0, j);
        issue_cin(bufi, row0);
    };

    // Fragment addresses inside a stage buffer.  The swizzle term of row
    // 32*tile + col does not depend on the tile.
    const int swz = (col >> 1) & 7;
    int a_off[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) a_off[k] = col * DIM + (((4 * h + k) ^ swz) << 4);
    // accumulator register 4g+i of this lane is train row 8g + 4h + i
    const int c_off = LDS_DESC + 16 * h;

    v4i fa[2][4];  // train fragments of the current / next tile
    v16i fc[2];    // their cin values in accumulator layout
    v16i acc[2];   // accumulators of the current / previous unit
    constexpr int U = (ST / 32) * QB;  // units per stage
    constexpr int PARTS = 8 / QB;      // LDS reads issued per unit
    constexpr int DMA_STEP = (U - 2) / (ST / 8 / WAVES + 1) > 0 ? (U - 2) / (ST / 8 / WAVES + 1) : 1;
    static_assert(2 + (ST / 8 / WAVES) * DMA_STEP < U, "not enough units to spread the DMA over");

    const int n_stage = (c1 - c0) * (CHUNK / ST);
    const int row_base = c0 * CHUNK;

    // digest of a finished unit: three smallest group minima per query block
    auto minima_lo = [&](const v16i& a) -> uint32_t {
        uint32_t t = umin32((uint32_t)a[0], (uint32_t)a[1]);
        t = umin32(umin32(t, (uint32_t)a[2]), (uint32_t)a[3]);
        t = umin32(umin32(t, (uint32_t)a[4]), (uint32_t)a[5]);
        t = umin32(umin32(t, (uint32_t)a[6]), (uint32_t)a[7]);
        t = umin32(umin32(t, (uint32_t)a[8]), (uint32_t)a[9]);
        return t;
    };
    auto minima_hi = [&](uint32_t t, const v16i& a) -> uint32_t {
        t = umin32(umin32(t, (uint32_t)a[10]), (uint32_t)a[11]);
        t = umin32(umin32(t, (uint32_t)a[12]), (uint32_t)a[13]);
        t = umin32(umin32(t, (uint32_t)a[14]), (uint32_t)a[15]);
        return t;
    };

    if (n_stage > 0) {
        issue_stage(0, row_base);
        if (n_stage > 1) issue_stage(1, row_base + ST);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        SFM_SB();
#pragma unroll
        for (int k = 0; k < 4; ++k) fa[0][k] = *(const v4i*)(lds + a_off[k]);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const v4i c = *(const v4i*)(lds + c_off + g * 32);
            fc[0][4 * g + 0] = c[0];
            fc[0][4 * g + 1] = c[1];
            fc[0][4 * g + 2] = c[2];
            fc[0][4 * g + 3] = c[3];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[1][i] = 0;

        int cur = 0;
        uint32_t first = 0xFFFFFFFFu;  // masks the digest of the unit before the first one
        for (int s = 0; s < n_stage; ++s) {
            const int nxt = cur == 2 ? 0 : cur + 1;
            const int nx2 = nxt == 2 ? 0 : nxt + 1;
            const unsigned char* buf = lds + cur * LDS_STAGE;
            const unsigned char* nbuf = lds + nxt * LDS_STAGE;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = u / QB, qb = u % QB, set = t & 1, nset = set ^ 1;
                const int pqb = (u + U - 1) % U % QB;
                // tile of the previous unit within the sweep
                const uint32_t ptile = (uint32_t)(s * (ST / 32) + (u == 0 ? -1 : (u - 1) / QB));
                const v16i& pa = acc[(u + 1) & 1];
                // next tile's fragments: next tile of this stage, or tile 0 of the
                // next stage (a stale buffer after the last stage: never used)
                const unsigned char* nd = t < ST / 32 - 1 ? buf + (t + 1) * 32 * DIM : nbuf;
                const unsigned char* nc = t < ST / 32 - 1 ? buf + (t + 1) * 32 * 4 : nbuf;
                uint32_t tmin = 0, p = 0;

                acc[u & 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[set][0], bq[qb][0], fc[set], 0, 0, 0);
                SFM_SB();
#pragma unroll
                for (int part = qb * PARTS; part < (qb + 1) * PARTS; ++part) {
                    if (part < 4) {
                        fa[nset][part] = *(const v4i*)(nd + a_off[part]);
                    } else {
                        const v4i c = *(const v4i*)(nc + c_off + (part - 4) * 32);
                        fc[nset][4 * (part - 4) + 0] = c[0];
                        fc[nset][4 * (part - 4) + 1] = c[1];
                        fc[nset][4 * (part - 4) + 2] = c[2];
                        fc[nset][4 * (part - 4) + 3] = c[3];
                    }
                }
                if (u == 1 && !(ABLATE & 4)) {
                    // stage s+1 has landed (this wave's share, then everyone's), and
                    // every wave is past its last read of stage s-1
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    SFM_SB();
                }
                // the DMA of stage s+2 is spread over the units after the barrier, one
                // piece at a time, instead of stalling this wave's MFMA stream in a burst
                // (past the end the last stage is fetched again into the free buffer
                // rather than branching around the issue)
                if (u == 1 && !(ABLATE & 2)) issue_stage(nx2, row_base + min(s + 2, n_stage - 1) * ST);
                SFM_SB();
                acc[u & 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[set][1], bq[qb][1], acc[u & 1], 0, 0, 0);
                SFM_SB();
                if (!(ABLATE & 1)) tmin = minima_lo(pa);
                SFM_SB();
                acc[u & 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[set][2], bq[qb][2], acc[u & 1], 0, 0, 0);
                SFM_SB();
                if (!(ABLATE & 1)) {
                    tmin = minima_hi(tmin, pa);
                    p = (tmin << TILE_BITS) | ptile;
                    if (u == 0) p |= first;
                }
                SFM_SB();
                acc[u & 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[set][3], bq[qb][3], acc[u & 1], 0, 0, 0);
                SFM_SB();
                if (!(ABLATE & 1)) {
                    const uint32_t n3 = umed3(m2[pqb], m3[pqb], p);
                    const uint32_t n2 = umed3(m1[pqb], m2[pqb], p);
                    m1[pqb] = umin32(m1[pqb], p);
                    m2[pqb] = n2;
                    m3[pqb] = n3;
                } else {
                    asm volatile("" ::"v"(pa));
                }
                SFM_SB();
            }
            first = 0;
            cur = nxt;
        }
        // nothing of this workgroup may still be in flight towards LDS when it ends
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // digest of the very last unit
        if (!(ABLATE & 1)) {
            const v16i& pa = acc[(U - 1) & 1];
            const uint32_t tmin = minima_hi(minima_lo(pa), pa);
            const uint32_t p = (tmin << TILE_BITS) | (uint32_t)(n_stage * (ST / 32) - 1);
            const uint32_t n3 = umed3(m2[QB - 1], m3[QB - 1], p);
            const uint32_t n2 = umed3(m1[QB - 1], m2[QB - 1], p);
            m1[QB - 1] = umin32(m1[QB - 1], p);
            m2[QB - 1] = n2;
            m3[QB - 1] = n3;
        } else {
            asm volatile("" ::"v"(acc[(U - 1) & 1]));
        }
    }

    // Lanes l and l+32 hold the same query over disjoint groups; each writes its
    // own record (sub = half h), the neighbour extraction merges them:
    //   x = best group (x << 9 | tile), y = second group, z = third minimum,
    //   w = first tile of the sweep
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int q = q_base + qb * 32 + col;
        const uint4 o = make_uint4(m1[qb], m2[qb], m3[qb], (uint32_t)(c0 * (CHUNK / 32)));
        job.partial[((size_t)wg.split * job.q_rows_alloc + q) * REC_SUBS + h] = o;
    }
}

// ------------------------------------------------------- knn2_mfma (16x16) ----
// The same sweep on v_mfma_i32_16x16x64_i8.  That shape sustains a higher clock
// under load than 32x32x32 (measured 80 % against 67 % of the nominal peak for
// bare MFMA loops), at the price of less idle issue time beside it.
//
// A wave owns 8 blocks of 16 queries (lane = query l&15, k-quarter l>>4; two
// fragments per block for K = 128).  Train rows are taken in blocks of 64 = four
// tiles of 16; a unit is one block against one query block: 8 MFMAs leaving 16
// accumulators per lane (tile j, register i = row 16j + 4*(l>>4) + i of the
// block), i.e. again 16 rows of ONE query per lane, digested exactly like a
// 32x32 unit.  Records carry the block number and the lane quarter.
typedef int v4acc __attribute__((ext_vector_type(4)));

template <int WAVES, int ABLATE>
__global__ void __launch_bounds__(WAVES * 64, 2)  // two waves per SIMD, as one or two workgroups per CU
knn2_mfma16_kernel(const KnnJob* __restrict__ jobs, int chunks_per_split, int qblocks, int splits, int map_mode)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * LDS_STAGE];
    constexpr int QB = 8;            // query blocks of 16 per wave
    constexpr int BLK = 64;          // train rows per block
    constexpr int NBLK = ST / BLK;   // blocks per stage
    constexpr int U = NBLK * QB;     // units per stage
    constexpr int QPB = WAVES * QB * 16;

    const WgCoord wg = decode_wg(blockIdx.x, (uint32_t)qblocks, (uint32_t)splits, map_mode);
    SFM_STAMP_REAL(0);
    SFM_STAMP(1);
    const KnnJob job = jobs[wg.job];
    if (wg.qblock * QPB >= job.n_query) return;
    const int c0 = wg.split * chunks_per_split;
    const int c1 = min(c0 + chunks_per_split, job.t_chunks);

    // the wave number as a scalar: everything derived from it (LDS bases, DMA
    // offsets, the cin branch) then stays in scalar registers
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int c16 = lane & 15;  // query within a block of 16 / train row within a tile
    const int kq = lane >> 4;   // which 16 bytes of each 64-byte half row / which row quartet
    const int q_base = (wg.qblock * WAVES + wave) * (QB * 16);

    v4i bq[QB][2];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int8_t* qp = job.q_desc + (size_t)(q_base + qb * 16 + c16) * DIM + 16 * kq;
        bq[qb][0] = ~*(const v4i*)(qp);
        bq[qb][1] = ~*(const v4i*)(qp + 64);
    }
    uint32_t m1[QB], m2[QB], m3[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) m1[qb] = m2[qb] = m3[qb] = 0xFFFFFFFFu;

    // stage image in LDS and its DMA: as in the 32x32 kernel
    const int dma_r = lane >> 3;
    const int dma_cs = lane & 7;
    constexpr int PIECES = ST / 8 / WAVES;
    static_assert(WAVES % 2 == 0, "the swizzle term of a lane must not depend on the piece");
    const uint32_t dma_off =
        (uint32_t)((8 * wave + dma_r) * DIM + ((dma_cs ^ (4 * (wave & 1) | (dma_r >> 1))) << 4));
    const uint32_t cin_off = (uint32_t)(lane & (ST / 4 - 1)) * 16u;
    // LDS-DMA through buffer descriptors: the row base travels in a scalar
    // offset, the lane part in one constant VGPR, so an issue costs no vector
    // arithmetic (buffer_load_dwordx4 ... offen lds)
    const __amdgpu_buffer_rsrc_t rs_desc =
        __builtin_amdgcn_make_buffer_rsrc((void*)job.t_desc, 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_cin =
        __builtin_amdgcn_make_buffer_rsrc((void*)job.t_cin, 0, 0x7FFFFFFF, 0x00020000);
    const unsigned long long cin_exec = wave == WAVES - 1 ? (1ull << (ST / 4)) - 1ull : 0ull;
    auto issue_piece = [&](int bufi, int row0, int j) {
        unsigned char* dst = lds + bufi * LDS_STAGE;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            rs_desc, (__attribute__((address_space(3))) void*)(dst + (wave + j * WAVES) * 1024), 16,
            (int)dma_off, row0 * DIM + j * (WAVES * 1024), 0, 0);
    };
    auto issue_cin = [&](int bufi, int row0) {
        unsigned char* dst = lds + bufi * LDS_STAGE;
        // cin[ST]: 32 lanes of the last wave.  Predicated through EXEC without a
        // branch (a branch here would split the pinned instruction stream into
        // basic blocks and let the compiler move the digest across the MFMAs);
        // every lane is active at this point of the kernel.
        const unsigned cin_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)(dst + LDS_DESC);
        unsigned saved_m0;  // M0 is restored: the compiler keeps track of what it put there for its own DMAs
        asm volatile(
            "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_mov_b64 exec, %2\n\ts_nop 0\n\t"
            "buffer_load_dwordx4 %3, %4, %5 offen lds\n\ts_mov_b64 exec, -1\n\ts_mov_b32 m0, %0"
            : "=&s"(saved_m0)
            : "s"(cin_lds), "s"(cin_exec), "v"(cin_off), "s"(rs_cin), "s"(row0 * 4)
            : "memory");
    };
    auto issue_stage = [&](int bufi, int row0) {
#pragma unroll
        for (int j = 0; j < PIECES; ++j) issue_piece(bufi, row0, j);
        issue_cin(bufi, row0);
    };

    // fragment addresses: row 16*tile + c16 has swizzle term (c16 >> 1) & 7 for
    // every tile; slot 4*half + kq
    const int swz = (c16 >> 1) & 7;
    int a_off[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) a_off[hh] = c16 * DIM + (((4 * hh + kq) ^ swz) << 4);
    const int c_off = LDS_DESC + 16 * kq;  // accumulator register i of tile j is row 16j + 4kq + i

    v4i fa[2][4][2];   // [set][tile][half] train fragments of the current / next block
    v4acc fc[2][4];    // [set][tile] their cin values in accumulator layout
    v4acc acc[2][4];   // [set][tile] accumulators of the current / previous unit
    const int n_stage = (c1 - c0) * (CHUNK / ST);
    const int row_base = c0 * CHUNK;

    auto digest_a = [&](const v4acc* a) -> uint32_t {  // first 8 of the 16 values
        uint32_t t = umin32((uint32_t)a[0][0], (uint32_t)a[0][1]);
        t = umin32(umin32(t, (uint32_t)a[0][2]), (uint32_t)a[0][3]);
        t = umin32(umin32(t, (uint32_t)a[1][0]), (uint32_t)a[1][1]);
        t = umin32(umin32(t, (uint32_t)a[1][2]), (uint32_t)a[1][3]);
        return t;
    };
    auto digest_b = [&](uint32_t t, const v4acc* a) -> uint32_t {
        t = umin32(umin32(t, (uint32_t)a[2][0]), (uint32_t)a[2][1]);
        t = umin32(umin32(t, (uint32_t)a[2][2]), (uint32_t)a[2][3]);
        t = umin32(umin32(t, (uint32_t)a[3][0]), (uint32_t)a[3][1]);
        t = umin32(umin32(t, (uint32_t)a[3][2]), (uint32_t)a[3][3]);
        return t;
    };
    // reads of block `blk` of the stage at `buf` into fragment set `set`: part 0..11
    auto read_part = [&](int set, const unsigned char* buf, int blk, int part) {
        if (part < 8) {
            const int j = part >> 1, hh = part & 1;
            fa[set][j][hh] = *(const v4i*)(buf + (blk * BLK + j * 16) * DIM + a_off[hh]);
        } else {
            const int j = part - 8;
            fc[set][j] = *(const v4acc*)(buf + c_off + (blk * BLK + j * 16) * 4);
        }
    };

    if (n_stage > 0) {
        issue_stage(0, row_base);
        if (n_stage > 1) issue_stage(1, row_base + ST);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        SFM_SB();
#pragma unroll
        for (int part = 0; part < 12; ++part) read_part(0, lds, 0, part);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[1][j][i] = 0;
        SFM_STAMP(2);

        int cur = 0;
        uint32_t first = 0xFFFFFFFFu;
        for (int s = 0; s < n_stage; ++s) {
            const int nxt = cur == 2 ? 0 : cur + 1;
            const int nx2 = nxt == 2 ? 0 : nxt + 1;
            const unsigned char* buf = lds + cur * LDS_STAGE;
            const unsigned char* nbuf = lds + nxt * LDS_STAGE;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int b = u / QB, qb = u % QB, set = b & 1, nset = set ^ 1;
                const int pqb = (u + U - 1) % U % QB;
                const uint32_t ptile = (uint32_t)(s * NBLK + (u == 0 ? -1 : (u - 1) / QB));
                const v4acc* pa = acc[(u + 1) & 1];
                v4acc* ca = acc[u & 1];
                // next block: the next one of this stage, or block 0 of the next stage
                const unsigned char* nb = b < NBLK - 1 ? buf : nbuf;
                const int nblk = b < NBLK - 1 ? b + 1 : 0;
                uint32_t tmin = 0, p = 0;
                // first halves of the four tiles, then second halves: dependent MFMAs
                // are four issues apart
                ca[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[set][0][0], bq[qb][0], fc[set][0], 0, 0, 0);
                SFM_SB();
                // 12 reads per block over 8 units: 2,2,2,2,1,1,1,1
                if (qb < 4) {
                    read_part(nset, nb, nblk, 2 * qb);
                    read_part(nset, nb, nblk, 2 * qb + 1);
                } else {
                    read_part(nset, nb, nblk, 4 + qb);
                }
                if (u == 1 && !(ABLATE & 4)) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    SFM_SB();
                }
                // the DMA of stage s+2, one piece per unit behind the barrier (a burst
                // of them stalls the wave's MFMA stream for hundreds of cycles)
                if (!(ABLATE & 2) && u >= 1 && u <= PIECES + 1) {
                    const int row2 = row_base + min(s + 2, n_stage - 1) * ST;
                    if (u <= PIECES) issue_piece(nx2, row2, u - 1);
                    else issue_cin(nx2, row2);
                }
                SFM_SB();
                ca[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[set][1][0], bq[qb][0], fc[set][1], 0, 0, 0);
                SFM_SB();
                // the digest of the previous unit, two instructions per MFMA gap
                if (!(ABLATE & 1)) {
                    tmin = umin32((uint32_t)pa[0][0], (uint32_t)pa[0][1]);
                    tmin = umin32(umin32(tmin, (uint32_t)pa[0][2]), (uint32_t)pa[0][3]);
                }
                SFM_SB();
                ca[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[set][2][0], bq[qb][0], fc[set][2], 0, 0, 0);
                SFM_SB();
                if (!(ABLATE & 1)) {
                    tmin = umin32(umin32(tmin, (uint32_t)pa[1][0]), (uint32_t)pa[1][1]);
                    tmin = umin32(umin32(tmin, (uint32_t)pa[1][2]), (uint32_t)pa[1][3]);
                }
                SFM_SB();
                ca[3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[set][3][0], bq[qb][0], fc[set][3], 0, 0, 0);
                SFM_SB();
                if (!(ABLATE & 1)) {
                    tmin = umin32(umin32(tmin, (uint32_t)pa[2][0]), (uint32_t)pa[2][1]);
                    tmin = umin32(umin32(tmin, (uint32_t)pa[2][2]), (uint32_t)pa[2][3]);
                }
                SFM_SB();
                ca[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[set][0][1], bq[qb][1], ca[0], 0, 0, 0);
                SFM_SB();
                if (!(ABLATE & 1)) {
                    tmin = umin32(umin32(tmin, (uint32_t)pa[3][0]), (uint32_t)pa[3][1]);
                    tmin = umin32(umin32(tmin, (uint32_t)pa[3][2]), (uint32_t)pa[3][3]);
                }
                SFM_SB();
                ca[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[set][1][1], bq[qb][1], ca[1], 0, 0, 0);
                SFM_SB();
                if (!(ABLATE & 1)) {
                    p = (tmin << TILE_BITS) | ptile;
                    if (u == 0) p |= first;
                }
                SFM_SB();
                ca[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[set][2][1], bq[qb][1], ca[2], 0, 0, 0);
                SFM_SB();
                // m1 <= m2 <= m3 are the three smallest so far: the new third is the
                // middle of (m2, m3, p), the new second the middle of (m1, m2, p)
                uint32_t n2 = 0;
                if (!(ABLATE & 1)) {
                    m3[pqb] = umed3(m2[pqb], m3[pqb], p);
                    n2 = umed3(m1[pqb], m2[pqb], p);
                }
                SFM_SB();
                ca[3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[set][3][1], bq[qb][1], ca[3], 0, 0, 0);
                SFM_SB();
                if (!(ABLATE & 1)) {
                    m1[pqb] = umin32(m1[pqb], p);
                    m2[pqb] = n2;
                } else {
                    asm volatile("" ::"v"(pa[0]), "v"(pa[1]), "v"(pa[2]), "v"(pa[3]));
                }
                SFM_SB();
            }
            first = 0;
            cur = nxt;
        }
        SFM_STAMP(3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(ABLATE & 1)) {
            const v4acc* pa = acc[(U - 1) & 1];
            const uint32_t tmin = digest_b(digest_a(pa), pa);
            const uint32_t p = (tmin << TILE_BITS) | (uint32_t)(n_stage * NBLK - 1);
            const uint32_t n3 = umed3(m2[QB - 1], m3[QB - 1], p);
            const uint32_t n2 = umed3(m1[QB - 1], m2[QB - 1], p);
            m1[QB - 1] = umin32(m1[QB - 1], p);
            m2[QB - 1] = n2;
            m3[QB - 1] = n3;
        } else {
            const v4acc* pa = acc[(U - 1) & 1];
            asm volatile("" ::"v"(pa[0]), "v"(pa[1]), "v"(pa[2]), "v"(pa[3]));
        }
    }

    // The four lanes l&15 + 16*quarter hold the same query over disjoint groups;
    // each writes its own record (sub = quarter), 1 KiB contiguous per wave and
    // query block, and the neighbour extraction merges them:
    //   x = best group (x << 9 | block), y = second group, z = third minimum,
    //   w = first block of the sweep
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int q = q_base + qb * 16 + c16;
        const uint4 o = make_uint4(m1[qb], m2[qb], m3[qb], (uint32_t)(c0 * (CHUNK / BLK)));
        job.partial[((size_t)wg.split * job.q_rows_alloc + q) * REC_SUBS + kq] = o;
    }
    SFM_STAMP(4);
    SFM_STAMP_REAL(5);
#ifdef SFM_GPU_TIMING_BUILD
    if (threadIdx.x == 0 && blockIdx.x < SWEEP_STAMP_WGS) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_sweep_stamps[blockIdx.x * 8 + 6] = xcc;
    }
#endif
}

// ---------------------------------------------------------------- rescan ----

// sqrtf of an integer < 2^24, correctly rounded, independent of the accuracy of
// the device's double sqrt: the candidate is checked against the two
// neighbouring rounding boundaries in exact double arithmetic.
__device__ __forceinline__ float sqrt_exact_u32(uint32_t n)
{
    const double dn = (double)n;
    float f = sqrtf((float)n);  // n < 2^24 is exact in float
    if (n == 0) return 0.f;
    const float up = __uint_as_float(__float_as_uint(f) + 1u);
    const float dnf = __uint_as_float(__float_as_uint(f) - 1u);
    const double hi = 0.5 * ((double)f + (double)up);
    const double lo = 0.5 * ((double)f + (double)dnf);
    if (hi * hi < dn) f = up;
    else if (lo * lo > dn) f = dnf;
    return f;
}

// Float distance of a squared distance n < 2^24 as the match records carry it:
// the correctly rounded square root.  The device's sqrtf is correctly rounded
// on every such n (a test counts the differences from sqrt_exact_u32 over all
// 2^24 of them and requires zero), and costs a quarter of the checked form.
__device__ __forceinline__ float sqrt_d2(uint32_t n) { return sqrtf((float)n); }

typedef unsigned long long u64;

// keeps the three smallest of (a1<=a2<=a3) and v
__device__ __forceinline__ void insert3(u64& a1, u64& a2, u64& a3, u64 v)
{
    if (v < a3) {
        a3 = v;
        if (a3 < a2) { const u64 t = a2; a2 = a3; a3 = t; }
        if (a2 < a1) { const u64 t = a1; a1 = a2; a2 = t; }
    }
}

// The three smallest of two ascending triples (a1<=a2<=a3, b1<=b2<=b3): the k-th
// smallest of the union is the minimum over i+j=k of max(a_i, b_j).
__device__ __forceinline__ void merge3(u64& a1, u64& a2, u64& a3, u64 b1, u64 b2, u64 b3)
{
    auto mn = [](u64 x, u64 y) { return x < y ? x : y; };
    auto mx = [](u64 x, u64 y) { return x < y ? y : x; };
    const u64 r1 = mn(a1, b1);
    const u64 r2 = mn(mn(a2, b2), mx(a1, b1));
    const u64 r3 = mn(mn(a3, b3), mn(mx(a2, b1), mx(a1, b2)));
    a1 = r1;
    a2 = r2;
    a3 = r3;
}

// Exact neighbours from the sweep records.  32 lanes per query:
//  1. the query's records (one per part and sweep lane) are reduced to the two best groups
//     by (minimum x, tile, half) and to the smallest x of all the other groups;
//  2. the 32 rows of the two groups are laid over the 32 lanes in ascending row
//     order and every lane computes the exact squared distance of its row;
//  3. the two smallest (d2, lane) are the neighbours.  Every row outside the two
//     groups has x >= the third minimum, hence d2-key >= 2*third; if that does
//     not exceed the key of the second neighbour found (a tie between group
//     minima), or if d2 >= 2^22 where different d2 share one float sqrt, the
//     query is queued for the exhaustive float-ranked pass instead.
// GM selects the group geometry of the sweep that wrote the records:
//   GM = 0  32x32 tiles: unit = tile of 32 rows, sub = half h (1 bit),
//           group rows = 32*tile + 8g + 4h + i            (g, i = 0..3)
//   GM = 1  16x16 tiles: unit = block of 64 rows, sub = quarter (2 bits),
//           group rows = 64*block + 16j + 4*quarter + i   (j, i = 0..3)
constexpr int RESCAN_QPW = 32;  // queries per workgroup
template <int GM>
__global__ void __launch_bounds__(256)
knn_rescan_kernel(const KnnJob* __restrict__ jobs, int n_splits, int k, int qblocks, int map_mode, double ratio)
{
    const double skip_r2 = ratio * ratio;  // 0 when the caller wants every neighbour
    constexpr int SB = GM ? 2 : 1;           // bits of the sub-unit index
    constexpr int SUBS = 1 << SB;            // records per (part, query): one per lane holding the query
    constexpr int UROWS = GM ? 64 : 32;      // rows per unit
    constexpr int JSTRIDE = GM ? 16 : 8;     // row stride between the four runs of a group
    // same workgroup -> XCD affinity as the sweep: the gathers of one job's
    // train rows then hit a single L2
    const WgCoord wg = decode_wg(blockIdx.x, (uint32_t)qblocks, 1u, map_mode);
    const KnnJob job = jobs[wg.job];
    const u64 NONE = ~0ull;
    const uint32_t tmask = (1u << TILE_BITS) - 1u;
    // Step 1, two or eight threads per query of the workgroup's 32: merge the
    // query's records (one per train part and sweep lane) into the two best groups
    // and the third minimum, decide whether the query needs neighbours at all, and
    // list those that do.  The merge is all 64-bit compares: with few records per
    // query two threads each (one wavefront for the 32 queries) issue a quarter of
    // the instructions eight would; with many records eight shorten the chain.
    __shared__ u64 s_c[RESCAN_QPW][3];
    __shared__ int s_list[RESCAN_QPW];
    __shared__ int s_wcount[4];
    {
        const int n_rec = n_splits * SUBS;
        const int tpq_log = n_rec >= 16 ? 3 : 1, tpq = 1 << tpq_log;
        const bool worker = (int)threadIdx.x < tpq * RESCAN_QPW;  // whole wavefronts
        const int qi = (threadIdx.x >> tpq_log) & (RESCAN_QPW - 1), part = threadIdx.x & (tpq - 1);
        const int q = wg.qblock * RESCAN_QPW + qi;
        bool need = false;
        u64 c1 = NONE, c2 = NONE, c3 = NONE;
        if (worker && q < job.n_query) {
            // key = x << 32 | (unit << SB | sub): ascending (x, unit, sub)
            auto key_of = [&](uint32_t p, uint32_t t0, uint32_t sub) -> u64 {
                return ((u64)(p >> TILE_BITS) << 32) | (((t0 + (p & tmask)) << SB) | sub);
            };
            // One record per (part, lane of the query in the sweep), its three groups
            // already in ascending order.  A lane's third group may be the second
            // best of all, so all three keep their identity.
            for (int j = part; j < n_rec; j += tpq) {
                const int sp = j / SUBS;
                const uint32_t sub = (uint32_t)(j - sp * SUBS);
                const uint4 r = job.partial[((size_t)sp * job.q_rows_alloc + q) * REC_SUBS + sub];
                const uint32_t t0 = r.w;
                const u64 b1 = r.x != 0xFFFFFFFFu ? key_of(r.x, t0, sub) : NONE;
                const u64 b2 = r.y != 0xFFFFFFFFu ? key_of(r.y, t0, sub) : NONE;
                const u64 b3 = r.z != 0xFFFFFFFFu ? key_of(r.z, t0, sub) : NONE;
                merge3(c1, c2, c3, b1, b2, b3);
            }
        }
        if (worker) {  // every lane of these wavefronts: a partner lane may hold a query past the end
            for (int off = 1; off < tpq; off <<= 1) {
                const u64 b1 = __shfl_xor(c1, off), b2 = __shfl_xor(c2, off), b3 = __shfl_xor(c3, off);
                merge3(c1, c2, c3, b1, b2, b3);
            }
        }
        if (worker && part == 0 && q < job.n_query) {
            need = true;
            // Pipelines that only keep the queries passing the ratio test
            // d1 < r * d2 (skip_r2 = r^2 > 0) need no neighbours for a query that
            // certainly fails it.  The best group's minimum bounds the nearest
            // distance from below, D1 >= qn + 2 x1 - 2 KEY, and the second group's
            // minimum belongs to another row, so the second nearest distance is at
            // most U = qn + 2 x2 - 2 KEY + 1: when D1 >= r^2 U, with a margin far
            // wider than the float rounding of the two square roots, the test fails
            // whatever the exact neighbours are.  Such a query reports none.
            if (skip_r2 > 0.0 && c2 != NONE) {
                const double qn0 = (double)job.q_qn[q] - 2.0 * (double)KEY_OFFSET;
                const double d1_lo = qn0 + 2.0 * (double)(uint32_t)(c1 >> 32);
                const double d2_hi = qn0 + 2.0 * (double)(uint32_t)(c2 >> 32) + 1.0;
                if (d1_lo >= skip_r2 * d2_hi * (1.0 + 1e-6) + 2.0) {
                    need = false;
                    job.counts[q] = 0;
                }
            }
            s_c[qi][0] = c1;
            s_c[qi][1] = c2;
            s_c[qi][2] = c3;
        }
        // list in ascending query order: per wavefront (8 queries), then across the 4
        const unsigned long long m = __builtin_amdgcn_ballot_w64(need);
        const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
        if (lane == 0) s_wcount[wv] = __popcll(m);
        __syncthreads();
        int base = 0;
        for (int w = 0; w < wv; ++w) base += s_wcount[w];
        if (need) s_list[base + __popcll(m & ((1ull << lane) - 1ull))] = qi;
    }
    __shared__ int s_list2[RESCAN_QPW];
    __shared__ int s_n2;
    if (threadIdx.x == 0) s_n2 = 0;
    __syncthreads();
    const int s_count = s_wcount[0] + s_wcount[1] + s_wcount[2] + s_wcount[3];
    int n_need = s_count;
    const int* list = s_list;
    // Step 2a, pair pipeline only (what is kept of a query is its best match and
    // the ratio verdict): sixteen lanes per query compute the exact distances of
    // the BEST group's 16 rows only.  With dA1 <= dA2 the two smallest of those and
    // L = qn + 2 x2 - 2 KEY (x2 = minimum of the second group): every row outside
    // the best group has d2 >= L, and the second group's minimum row has d2 = L or
    // L + 1.  So when dA1 < L the best match is the best of the group, and the
    // second distance is min(dA2, L) or min(dA2, L + 1): the verdict of
    // main.cpp:192 is evaluated on both, exactly as the reference does (float
    // square roots, compare in double), and when they agree the query is done
    // with half the rows gathered.  Anything else (no second group, a tie between
    // the groups, distances at 2^22 where float square roots collide, a verdict
    // that hangs on that one unit) goes through step 2b below.
    if (skip_r2 > 0.0) {
        const int l16 = threadIdx.x & 15;
        const int sg = l16 & 7, rq = l16 >> 3;
        for (int it = threadIdx.x >> 4; it < n_need; it += 16) {
            const int qi = s_list[it];
            const int q = wg.qblock * RESCAN_QPW + qi;
            const u64 c1 = s_c[qi][0], c2 = s_c[qi][1];
            // (a query without any record -- no train rows, or a sweep run with its digest
            // switched off for timing -- has no group to gather: row 0 is read and ignored)
            const bool have_a = c1 != NONE, have_b = c2 != NONE;
            const uint32_t ida = (uint32_t)c1;
            const int base_a = have_a ? (int)(ida >> SB) * UROWS + 4 * (int)(ida & ((1u << SB) - 1u)) : 0;
            const v4i qv = *(const v4i*)(job.q_desc + (size_t)q * DIM + sg * 16);
            // candidate c = 2 * it2 + rq of the group is row base_a + JSTRIDE * (c >> 2) + (c & 3)
            v4i tv[8];
#pragma unroll
            for (int it2 = 0; it2 < 8; ++it2) {
                const int r = base_a + JSTRIDE * (it2 >> 1) + 2 * (it2 & 1) + rq;
                tv[it2] = *(const v4i*)(job.t_desc + (size_t)r * DIM + sg * 16);
            }
            int part[8];
#pragma unroll
            for (int it2 = 0; it2 < 8; ++it2) {
                int acc = __builtin_amdgcn_sdot4(qv[0], tv[it2][0], 0, false);
                acc = __builtin_amdgcn_sdot4(qv[1], tv[it2][1], acc, false);
                acc = __builtin_amdgcn_sdot4(qv[2], tv[it2][2], acc, false);
                acc = __builtin_amdgcn_sdot4(qv[3], tv[it2][3], acc, false);
                part[it2] = acc;
            }
            // reduce over the 8 segments while scattering the 8 candidates of this half:
            // afterwards the lane holds the whole dot product of candidate 2 * sg + rq
            int r4[4], r2[2];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool up = (sg & 4) != 0;
                const int mine = up ? part[j + 4] : part[j];
                const int send = up ? part[j] : part[j + 4];
                r4[j] = mine + __shfl_xor(send, 4);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bool up = (sg & 2) != 0;
                const int mine = up ? r4[j + 2] : r4[j];
                const int send = up ? r4[j] : r4[j + 2];
                r2[j] = mine + __shfl_xor(send, 2);
            }
            const int qt = ((sg & 1) ? r2[1] : r2[0]) + __shfl_xor((sg & 1) ? r2[0] : r2[1], 1);
            const int cand = 2 * sg + rq;
            const int row = base_a + JSTRIDE * (sg >> 1) + 2 * (sg & 1) + rq;
            const bool row_ok = row < job.n_train;
            const int qn = job.q_qn[q];
            const int tn = job.t_qn[row_ok ? row : 0];
            uint32_t e = 0xFFFFFFFFu;  // (d2 << 5) | candidate: ascending (d2, row)
            if (row_ok) e = ((uint32_t)(qn + tn - 2 * qt) << 5) | (uint32_t)cand;
            uint32_t e1 = e, e2 = 0xFFFFFFFFu;
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                const uint32_t b1 = (uint32_t)__shfl_xor((int)e1, off), b2 = (uint32_t)__shfl_xor((int)e2, off);
                const uint32_t lo = umin32(e1, b1);
                e2 = umin32(umax32(e1, b1), umin32(e2, b2));
                e1 = lo;
            }
            const int p1 = (int)(e1 & 31u);
            const int row1 = __shfl(row, ((p1 & 1) << 3) | (p1 >> 1), 16);
            if (l16 == 0) {
                bool done = false;
                if (have_a && have_b && e1 != 0xFFFFFFFFu) {
                    const int32_t dA1 = (int32_t)(e1 >> 5);
                    const int32_t dA2 = e2 != 0xFFFFFFFFu ? (int32_t)(e2 >> 5) : 0x7FFFFFFF;
                    const int32_t L = qn - 2 * KEY_OFFSET + 2 * (int32_t)(uint32_t)(c2 >> 32);
                    if (dA1 < L && L + 1 < (1 << 22)) {
                        const float f1 = sqrt_d2((uint32_t)dA1);
                        const float f_lo = sqrt_d2((uint32_t)min(dA2, L)), f_hi = sqrt_d2((uint32_t)min(dA2, L + 1));
                        const bool v_lo = (double)f1 < ratio * (double)f_lo, v_hi = (double)f1 < ratio * (double)f_hi;
                        if (v_lo == v_hi) {
                            done = true;
                            sfm_gpu_dmatch m;
                            m.queryIdx = q;
                            m.trainIdx = row1;
                            m.imgIdx = 0;
                            m.distance = f1;
                            job.out[(size_t)q * k] = m;
                            job.counts[q] = v_lo ? 2 : 0;
                        }
                    }
                }
                if (!done) s_list2[atomicAdd(&s_n2, 1)] = qi;
            }
        }
        __syncthreads();
        n_need = s_n2;
        list = s_list2;
    }
    const int l = threadIdx.x & 31;
    // Step 2b: the eight 32-lane groups take the listed queries in turn.
    for (int it = threadIdx.x >> 5; it < n_need; it += 8) {
    const int qi = list[it];
    const int q = wg.qblock * RESCAN_QPW + qi;
    const u64 c1 = s_c[qi][0], c2 = s_c[qi][1], c3 = s_c[qi][2];
    const uint32_t third_x = (uint32_t)(c3 >> 32);  // 0xFFFFFFFF when there is no third group
    // c1 < c2 compares (x, unit, sub).  For the lane order to be the row order
    // the group with the smaller (unit, sub) must come first.
    const uint32_t ida = (uint32_t)c1, idb = (uint32_t)c2;
    const bool have_a = c1 != NONE, have_b = c2 != NONE;
    const uint32_t id_lo = (!have_b || ida < idb) ? ida : idb;
    const uint32_t id_hi = (!have_b || ida < idb) ? idb : ida;
    const bool same_unit = have_b && (id_lo >> SB) == (id_hi >> SB);
    const int unit_lo = (int)(id_lo >> SB) * UROWS, unit_hi = (int)(id_hi >> SB) * UROWS;
    const int sub_lo = 4 * (int)(id_lo & ((1u << SB) - 1u)), sub_hi = 4 * (int)(id_hi & ((1u << SB) - 1u));
    // candidate position 0..31 -> train row (ascending), -1 when there is none;
    // written without branches (selects only).  Two groups of one unit interleave:
    // position = 8*run + 4*which + i.
    auto row_of = [&](int pos) -> int {
        const int i = pos & 15;
        const bool hi = pos >= 16;
        const int split_row = (hi ? unit_hi + sub_hi : unit_lo + sub_lo) + JSTRIDE * (i >> 2) + (i & 3);
        const int same_row = unit_lo + JSTRIDE * (pos >> 3) + ((pos >> 2) & 1 ? sub_hi : sub_lo) + (pos & 3);
        const int r = same_unit ? same_row : split_row;
        const bool none = !have_a || (hi && !have_b);
        return none ? -1 : r;
    };
    // Dot products, 8 lanes per row so that each load instruction fetches whole
    // 128-B rows: lane l works on 16-byte segment l&7 of candidates 4*it + (l>>3).
    // All loads are issued up front on clamped rows (padding rows are readable)
    // so that none of them waits behind another's latency; rows that do not
    // exist are masked out at the end.
    const int sg = l & 7, rq = l >> 3;
    const int pos = 4 * sg + rq;
    const int row = row_of(pos);
    const bool row_ok = row >= 0 && row < job.n_train;
    const int qn = job.q_qn[q];
    const int tn = job.t_qn[row_ok ? row : 0];
    const v4i qv = *(const v4i*)(job.q_desc + (size_t)q * DIM + sg * 16);
    // rows of candidates 4*it + rq (i = rq): in two different units the position
    // is in the low group for it < 4 (run = it) and in the high one after; in
    // one unit run = it >> 1 and the group alternates with it
    const int base_lo = have_a ? unit_lo + sub_lo + rq : 0;
    const int base_hi = have_b ? unit_hi + sub_hi + rq : 0;
    v4i tv[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int r = same_unit ? ((it & 1) ? base_hi : base_lo) + JSTRIDE * (it >> 1)
                                : (it < 4 ? base_lo : base_hi) + JSTRIDE * (it & 3);
        tv[it] = *(const v4i*)(job.t_desc + (size_t)r * DIM + sg * 16);
    }
    int part[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        int acc = __builtin_amdgcn_sdot4(qv[0], tv[it][0], 0, false);
        acc = __builtin_amdgcn_sdot4(qv[1], tv[it][1], acc, false);
        acc = __builtin_amdgcn_sdot4(qv[2], tv[it][2], acc, false);
        acc = __builtin_amdgcn_sdot4(qv[3], tv[it][3], acc, false);
        part[it] = acc;
    }
    // reduce over the 8 segments while scattering the 8 candidates: afterwards
    // lane l holds the whole dot product of candidate 4*sg + rq
    int r4[4], r2[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool up = (sg & 4) != 0;
        const int mine = up ? part[j + 4] : part[j];
        const int send = up ? part[j] : part[j + 4];
        r4[j] = mine + __shfl_xor(send, 4);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const bool up = (sg & 2) != 0;
        const int mine = up ? r4[j + 2] : r4[j];
        const int send = up ? r4[j] : r4[j + 2];
        r2[j] = mine + __shfl_xor(send, 2);
    }
    const int qt = ((sg & 1) ? r2[1] : r2[0]) + __shfl_xor((sg & 1) ? r2[0] : r2[1], 1);
    uint32_t e = 0xFFFFFFFFu;  // (d2 << 5) | position; d2 <= 8323200 < 2^23
    if (row_ok) {
        const int d2 = qn + tn - 2 * qt;
        e = ((uint32_t)d2 << 5) | (uint32_t)pos;
    }
    // two smallest over the 32 lanes
    uint32_t e1 = e, e2 = 0xFFFFFFFFu;
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) {
        const uint32_t b1 = (uint32_t)__shfl_xor((int)e1, off), b2 = (uint32_t)__shfl_xor((int)e2, off);
        const uint32_t lo = umin32(e1, b1);
        e2 = umin32(umax32(e1, b1), umin32(e2, b2));
        e1 = lo;
    }
    // the rows of the two winners, from the lanes that hold them (lane of
    // position p: segment p >> 2 is the lane's low three bits, p & 3 its high two)
    const int p1 = (int)(e1 & 31u), p2 = (int)(e2 & 31u);
    const int row1 = __shfl(row, ((p1 & 3) << 3) | (p1 >> 2), 32);
    const int row2 = __shfl(row, ((p2 & 3) << 3) | (p2 >> 2), 32);
    // lane j < k writes neighbour j
    const uint32_t ej = l == 0 ? e1 : e2;
    const float dist_j = sqrt_d2(ej >> 5);
    const float dist_second = __shfl(dist_j, 1, 32);  // for lane 0's ratio verdict below
    if (l < k) {
        sfm_gpu_dmatch m;
        m.queryIdx = q;
        m.imgIdx = 0;
        if (ej != 0xFFFFFFFFu) {
            m.trainIdx = l == 0 ? row1 : row2;
            m.distance = dist_j;
        } else {
            m.trainIdx = -1;
            m.distance = __uint_as_float(0x7f7fffffu);  // FLT_MAX
        }
        job.out[(size_t)q * k + l] = m;
    }
    if (l == 0) {
        int cnt = (e1 != 0xFFFFFFFFu ? 1 : 0) + ((k > 1 && e2 != 0xFFFFFFFFu) ? 1 : 0);
        if (e2 != 0xFFFFFFFFu) {
            const int32_t d2 = (int32_t)(e2 >> 5);
            const uint32_t x2 = (uint32_t)(d2 - qn + 2 * KEY_OFFSET) >> 1;
            bool redo = d2 >= (1 << 22) || third_x <= x2;
            bool settled = false;  // verdict already known from the bounds
            if (redo && skip_r2 > 0.0 && d2 < (1 << 22)) {
                // A row outside the two groups may be the true second neighbour:
                // its squared distance is at least L3 = qn + 2 third - 2 KEY and
                // the true second distance lies in [L3, d2].  The pair pipeline
                // only keeps (best match, ratio verdict): when the best match is
                // strictly nearer than L3 and the verdict is the same at both ends
                // of that range (margins as in the screening above), the exhaustive
                // pass has nothing to add.
                const double d1 = (double)(e1 >> 5);
                const double l3 = (double)qn - 2.0 * (double)KEY_OFFSET + 2.0 * (double)third_x;
                if (d1 < l3 && (d1 + 2.0) * (1.0 + 1e-6) < skip_r2 * l3) {
                    redo = false;  // passes whatever the second neighbour is; the record of the best is exact
                    settled = true;
                } else if (d1 >= skip_r2 * (double)d2 * (1.0 + 1e-6) + 2.0) {
                    redo = false;  // fails even against the farthest candidate
                    settled = true;
                    cnt = 0;
                }
            }
            if (redo) {
                const int slot = atomicAdd(job.redo, 1);
                job.redo[1 + slot] = q;
            } else if (skip_r2 > 0.0 && !settled) {
                // both records are final: the ratio verdict itself (main.cpp:192, in
                // double on the float distances) goes into the count, 0 = fails, so
                // that the compaction after this kernel reads one word per query
                if (!((double)dist_j < ratio * (double)dist_second)) cnt = 0;
            }
        }
        job.counts[q] = cnt;
    }
    }  // listed queries
}

// Float-ranked exhaustive pass for the queries flagged above: ranks train rows
// by (sqrtf(d2) as float, row), exactly the reference ordering, at any d2.
// Each flagged query is cut into segments of train rows handled by different
// workgroups (knn_redo_scan_kernel, partial results parked in the sweep's
// record buffer, which is free by now) and merged by knn_redo_merge_kernel.
// 8 lanes share a row so that each load instruction fetches whole 128-B rows.
constexpr int REDO_MAX_SEG = 64;

__device__ __forceinline__ int redo_segments(const KnnJob& job, int n_redo, int n_splits, int max_seg)
{
    const long long cap = (long long)job.q_rows_alloc * n_splits;  // uint4 slots in job.partial
    long long nseg = n_redo > 0 ? cap / n_redo : 1;
    if (nseg > max_seg) nseg = max_seg;
    return nseg < 1 ? 1 : (int)nseg;
}

__device__ __forceinline__ void merge2(u64& a1, u64& a2, u64 b1, u64 b2)
{
    const bool afirst = a1 < b1;
    const u64 nf = afirst ? a1 : b1;
    const u64 ns = afirst ? (b1 < a2 ? b1 : a2) : (a1 < b2 ? a1 : b2);
    a1 = nf;
    a2 = ns;
}

__global__ void __launch_bounds__(256)
knn_redo_scan_kernel(const KnnJob* __restrict__ jobs, int n_splits, int max_seg)
{
    __shared__ u64 sh[2 * 4];
    const KnnJob job = jobs[blockIdx.y];
    const int n_redo = job.redo[0];
    if (n_redo == 0) return;
    const int nseg = redo_segments(job, n_redo, n_splits, max_seg);
    const int seg_rows = ((job.n_train + nseg - 1) / nseg + 31) & ~31;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int sg = lane & 7, rsub = lane >> 3;
    const u64 NONE = ~0ull;
    for (int u = blockIdx.x; u < n_redo * nseg; u += gridDim.x) {
        const int item = u / nseg, seg = u - item * nseg;
        const int q = job.redo[1 + item];
        const int r0 = seg * seg_rows;
        const int r1 = min(r0 + seg_rows, job.n_train);
        const v4i qv = *(const v4i*)(job.q_desc + (size_t)q * DIM + sg * 16);
        const int qn = job.q_qn[q];
        u64 f = NONE, s = NONE;
        int d2f = 0x7FFFFFFF, lim = 0x7FFFFFFF;  // d2 of this lane's best / second best so far
        // four row groups per step, their loads issued together
        for (int base = r0 + wave * 8; base < r1; base += 4 * 4 * 8) {
            v4i tv[4];
            int tn[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int row = base + g * 32 + rsub;
                const int rr = row < r1 ? row : 0;
                tv[g] = *(const v4i*)(job.t_desc + (size_t)rr * DIM + sg * 16);
                tn[g] = job.t_qn[rr];
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int row = base + g * 32 + rsub;
                const bool ok = row < r1;
                int acc = __builtin_amdgcn_sdot4(qv[0], tv[g][0], 0, false);
                acc = __builtin_amdgcn_sdot4(qv[1], tv[g][1], acc, false);
                acc = __builtin_amdgcn_sdot4(qv[2], tv[g][2], acc, false);
                acc = __builtin_amdgcn_sdot4(qv[3], tv[g][3], acc, false);
                acc += __shfl_xor(acc, 1);
                acc += __shfl_xor(acc, 2);
                acc += __shfl_xor(acc, 4);
                const int d2 = qn + tn[g] - 2 * acc;
                // rows come in ascending order, so a later row needs a float distance
                // strictly below the second best's, hence a smaller d2
                if (ok && sg == 0 && d2 < lim) {
                    const u64 key = ((u64)__float_as_uint(sqrt_d2((uint32_t)d2)) << 32) | (uint32_t)row;
                    if (key < f) { s = f; f = key; lim = d2f; d2f = d2; }
                    else if (key < s) { s = key; lim = d2; }
                }
            }
        }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) merge2(f, s, __shfl_xor(f, off), __shfl_xor(s, off));
        if (lane == 0) {
            sh[2 * wave] = f;
            sh[2 * wave + 1] = s;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            u64 a1 = NONE, a2 = NONE;
            for (int w = 0; w < 4; ++w) merge2(a1, a2, sh[2 * w], sh[2 * w + 1]);
            uint4 o;
            o.x = (uint32_t)a1;
            o.y = (uint32_t)(a1 >> 32);
            o.z = (uint32_t)a2;
            o.w = (uint32_t)(a2 >> 32);
            job.partial[u] = o;
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256)
knn_redo_merge_kernel(const KnnJob* __restrict__ jobs, int n_splits, int k, int max_seg, double ratio)
{
    const KnnJob job = jobs[blockIdx.y];
    const int n_redo = job.redo[0];
    const int nseg = redo_segments(job, n_redo, n_splits, max_seg);
    const u64 NONE = ~0ull;
    for (int item = blockIdx.x * 256 + threadIdx.x; item < n_redo; item += gridDim.x * 256) {
        const int q = job.redo[1 + item];
        u64 a1 = NONE, a2 = NONE;
        for (int seg = 0; seg < nseg; ++seg) {
            const uint4 o = job.partial[(size_t)item * nseg + seg];
            merge2(a1, a2, ((u64)o.y << 32) | o.x, ((u64)o.w << 32) | o.z);
        }
        const u64 e[2] = {a1, a2};
        for (int j = 0; j < k; ++j) {
            if (e[j] != NONE) {
                sfm_gpu_dmatch m;
                m.queryIdx = q;
                m.imgIdx = 0;
                m.trainIdx = (int32_t)(uint32_t)e[j];
                m.distance = __uint_as_float((uint32_t)(e[j] >> 32));
                job.out[(size_t)q * k + j] = m;
            }
        }
        // ratio verdict in the count, as the rescan kernel leaves it
        if (ratio > 0.0 && k == 2 && a2 != NONE &&
            !((double)__uint_as_float((uint32_t)(a1 >> 32)) < ratio * (double)__uint_as_float((uint32_t)(a2 >> 32))))
            job.counts[q] = 0;
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

// Counts the integers n in [0, 2^24) for which plain sqrtf((float)n) differs
// from the verified correctly rounded value (test hook).
__global__ void __launch_bounds__(256) sqrt_probe_kernel(int32_t* mismatches)
{
    const uint32_t n = blockIdx.x * 256u + threadIdx.x;
    if (n >= (1u << 24)) return;
    const float a = sqrtf((float)n);
    const float b = sqrt_exact_u32(n);
    if (__float_as_uint(a) != __float_as_uint(b)) atomicAdd(mismatches, 1);
}

// Fills the whole LDS of the CU the workgroup lands on with a byte pattern.  A
// kernel that skips a DMA or reads a stage before it landed would otherwise see
// whatever the previous kernel left there -- often the right data of the same test.
__global__ void __launch_bounds__(256) lds_poison_kernel(uint32_t pattern, uint32_t* sink)
{
    __shared__ uint32_t buf[163840 / 4];
    for (int i = threadIdx.x; i < 163840 / 4; i += 256) buf[i] = pattern + (uint32_t)i * 2654435761u;
    __syncthreads();
    if (buf[(threadIdx.x * 97u + blockIdx.x) % (163840u / 4u)] == 0x12345678u && sink) sink[0] = 1;  // keeps the stores
}

// C(16x16) = A(16 rows x 64 k) * B^T (B given as [16 cols][64 k]) through one
// v_mfma_i32_16x16x64_i8 with the operand and accumulator mapping the 16x16
// sweep assumes.
__global__ void __launch_bounds__(64) mfma16_layout_probe_kernel(const int8_t* A, const int8_t* B, int32_t* C)
{
    const int lane = threadIdx.x, c16 = lane & 15, kq = lane >> 4;
    const v4i a = *(const v4i*)(A + c16 * 64 + 16 * kq);
    const v4i b = *(const v4i*)(B + c16 * 64 + 16 * kq);
    v4acc c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) C[(4 * kq + i) * 16 + c16] = c[i];
}

}  // namespace

// ---------------------------------------------------------------- hosts ----

hipError_t launch_prep_rows(hipStream_t s, const PrepJob* jobs_dev, int n_jobs, int rows_alloc_max,
                            int dtype, int32_t* err_flag)
{
#ifndef SFM_PREP_RPL
#define SFM_PREP_RPL 1
#endif
    if (dtype == SFM_GPU_DESC_F32)
        hipLaunchKernelGGL((prep_rows_kernel<SFM_GPU_DESC_F32, SFM_PREP_RPL>),
                           dim3((rows_alloc_max + 32 * SFM_PREP_RPL - 1) / (32 * SFM_PREP_RPL), n_jobs), dim3(256), 0, s,
                           jobs_dev, err_flag);
    else
        hipLaunchKernelGGL((prep_rows_kernel<SFM_GPU_DESC_U8, 1>), dim3((rows_alloc_max + 31) / 32, n_jobs), dim3(256),
                           0, s, jobs_dev, err_flag);
    return hipGetLastError();
}

// Workgroup shapes of the distance kernel: {query blocks per wave, waves}.
//   0: 4 x 4 (512 queries / workgroup)   1: 2 x 4 (256)
//   2: 2 x 8 (512)                       3: 4 x 8 (1024)
//   4: 16x16x64 MFMAs, 8 x 4 waves (512 queries / workgroup), groups over 64-row blocks
//   5: the same with 8 waves (1024 queries / workgroup, one workgroup per CU): half the
//      LDS-DMA instructions per MFMA
static const int kVariantQPB[6] = {512, 256, 512, 1024, 512, 1024};

KnnPlan plan_knn(int n_query_max, int n_train_max, int n_jobs, int force_variant)
{
    KnnPlan p;
    p.ablate = force_variant >= 0 ? force_variant / 16 : 0;
    if (force_variant >= 0) force_variant %= 16;
    p.variant = (force_variant >= 0 && force_variant < 6) ? force_variant : 4;  // 16x16x64 unless forced
    p.group_mode = p.variant >= 4 ? 1 : 0;
    p.q_per_block = kVariantQPB[p.variant];
    const int t_chunks = (n_train_max + CHUNK - 1) / CHUNK;
    const long long qblocks = (long long)((n_query_max + p.q_per_block - 1) / p.q_per_block) * n_jobs;
    // The sweep carries no data-dependent state between tiles, so the train rows
    // can be split freely into parts (never more than MAX_SWEEP_CHUNKS chunks
    // each).  The number of parts is chosen to minimise
    //     rounds of resident workgroups x (chunks per part + 1 for set-up),
    // with `slots` workgroups resident at a time (two per CU for the 4-wave
    // shapes, one for the 8-wave ones); ties go to fewer parts (fewer records).
    const long long slots = (p.variant == 0 || p.variant == 1 || p.variant == 4) ? 512 : 256;
    const int s_min = std::max(1, (t_chunks + MAX_SWEEP_CHUNKS - 1) / MAX_SWEEP_CHUNKS);
    const int s_max = std::max(s_min, std::min(t_chunks, 64));
    int splits = s_min;
    long long best_cost = -1;
    for (int sp = s_min; sp <= s_max && t_chunks > 0; ++sp) {
        const long long cps = (t_chunks + sp - 1) / sp;
        const long long parts = (t_chunks + cps - 1) / cps;  // what the launch will really use
        const long long rounds = (qblocks * parts + slots - 1) / slots;
        const long long cost = rounds * (cps + 1);
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            splits = sp;
        }
    }
#ifdef SFM_GPU_TIMING_BUILD
    if (const char* e = getenv("SFM_KNN_SPLITS")) splits = std::max(s_min, std::min(atoi(e), std::max(t_chunks, 1)));  // experiments
#endif
    p.n_splits = splits;
    p.chunks_per_split = (t_chunks + splits - 1) / splits;
    if (p.chunks_per_split < 1) p.chunks_per_split = 1;
    p.n_splits = t_chunks > 0 ? (t_chunks + p.chunks_per_split - 1) / p.chunks_per_split : 1;
    return p;
}

hipError_t launch_knn2(hipStream_t s, const KnnJob* jobs_dev, int n_jobs, int q_rows_alloc_max,
                       const KnnPlan& plan, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    const int qblocks = (q_rows_alloc_max + plan.q_per_block - 1) / plan.q_per_block;
    const int splits = plan.n_splits;
    const long long total = (long long)qblocks * splits * n_jobs;
    if (total <= 0) return hipSuccess;
    if (total > 0x7FFFFFFFll) return hipErrorInvalidValue;
    // a lone job runs fastest in plain order (its workgroups load their query blocks
    // in one burst at the start: with the query block varying fastest each XCD
    // fetches an eighth of them instead of all)
    int mode = (n_jobs % 8 == 0) ? 1 : (n_jobs > 1 && splits % 8 == 0) ? 2 : 0;
#ifdef SFM_GPU_TIMING_BUILD
    if (const char* e = getenv("SFM_KNN_MAP_MODE")) mode = atoi(e);  // placement experiments
#endif
    const dim3 grid((unsigned)total);
#define SFM_KNN2(QB, WAVES, ABL)                                                                  \
    hipExtLaunchKernelGGL((knn2_mfma_kernel<QB, WAVES, ABL>), grid, dim3(WAVES * 64), 0, s, ev_start, ev_stop, 0, \
                          jobs_dev, plan.chunks_per_split, qblocks, splits, mode)
#define SFM_KNN16W(W, ABL)                                                                       \
    hipExtLaunchKernelGGL((knn2_mfma16_kernel<W, ABL>), grid, dim3(W * 64), 0, s, ev_start, ev_stop, 0, jobs_dev, \
                          plan.chunks_per_split, qblocks, splits, mode)
#define SFM_KNN16(ABL)              \
    do {                            \
        if (plan.variant == 5)      \
            SFM_KNN16W(8, ABL);     \
        else                        \
            SFM_KNN16W(4, ABL);     \
    } while (0)
    if (plan.ablate != 0) {
#ifdef SFM_GPU_TIMING_BUILD
        // timing experiments (wrong results): parts of the kernel switched off
        if (plan.variant == 4 || plan.variant == 5) {
            switch (plan.ablate) {
                case 1: SFM_KNN16(1); break;
                case 2: SFM_KNN16(2); break;
                case 3: SFM_KNN16(3); break;
                case 4: SFM_KNN16(4); break;
                case 6: SFM_KNN16(6); break;
                case 7: SFM_KNN16(7); break;
                default: return hipErrorInvalidValue;
            }
        } else if (plan.variant == 0) {
            switch (plan.ablate) {
                case 1: SFM_KNN2(4, 4, 1); break;
                case 3: SFM_KNN2(4, 4, 3); break;
                case 7: SFM_KNN2(4, 4, 7); break;
                default: return hipErrorInvalidValue;
            }
        } else {
            return hipErrorInvalidValue;
        }
        return hipGetLastError();
#else
        return hipErrorInvalidValue;
#endif
    }
    switch (plan.variant) {
        case 1: SFM_KNN2(2, 4, 0); break;
        case 2: SFM_KNN2(2, 8, 0); break;
        case 3: SFM_KNN2(4, 8, 0); break;
        case 4:
        case 5: SFM_KNN16(0); break;
        default: SFM_KNN2(4, 4, 0); break;
    }
#undef SFM_KNN16
#undef SFM_KNN16W
#undef SFM_KNN2
    return hipGetLastError();
}

hipError_t launch_knn_finalize(hipStream_t s, const KnnJob* jobs_dev, int n_jobs, int n_query_max,
                               int n_splits, int k, int group_mode, double skip_ratio)
{
    const double ratio = skip_ratio > 0.0 && k == 2 ? skip_ratio : 0.0;
    const int qblocks = (n_query_max + RESCAN_QPW - 1) / RESCAN_QPW;
    if (qblocks <= 0 || n_jobs <= 0) return hipSuccess;
    const long long total = (long long)qblocks * n_jobs;
    if (total > 0x7FFFFFFFll) return hipErrorInvalidValue;
    const int mode = n_jobs % 8 == 0 ? 1 : 0;
    if (group_mode)
        hipLaunchKernelGGL(knn_rescan_kernel<1>, dim3((unsigned)total), dim3(256), 0, s, jobs_dev, n_splits, k,
                           qblocks, mode, ratio);
    else
        hipLaunchKernelGGL(knn_rescan_kernel<0>, dim3((unsigned)total), dim3(256), 0, s, jobs_dev, n_splits, k,
                           qblocks, mode, ratio);
    return hipGetLastError();
}

hipError_t launch_knn_redo(hipStream_t s, const KnnJob* jobs_dev, int n_jobs, int n_splits, int k, double skip_ratio)
{
    // flagged queries are rare: about 2048 workgroups in all are plenty, and a
    // larger grid only costs launch time
    const int gx = std::min(512, std::max(8, 2048 / (n_jobs > 0 ? n_jobs : 1)));
    // with many jobs in flight a flagged query is cut into fewer, longer segments
    const int max_seg = std::min(REDO_MAX_SEG, std::max(1, 512 / (n_jobs > 0 ? n_jobs : 1)));
    hipLaunchKernelGGL(knn_redo_scan_kernel, dim3(gx, n_jobs), dim3(256), 0, s, jobs_dev, n_splits, max_seg);
    hipLaunchKernelGGL(knn_redo_merge_kernel, dim3(4, n_jobs), dim3(256), 0, s, jobs_dev, n_splits, k, max_seg,
                       skip_ratio > 0.0 && k == 2 ? skip_ratio : 0.0);
    return hipGetLastError();
}

hipError_t read_sweep_stamps(unsigned long long* out, int n_wg)
{
#ifdef SFM_GPU_TIMING_BUILD
    if (n_wg > SWEEP_STAMP_WGS) n_wg = SWEEP_STAMP_WGS;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sweep_stamps), (size_t)n_wg * 64);
#else
    (void)out;
    (void)n_wg;
    return hipErrorNotSupported;
#endif
}

hipError_t launch_mfma_layout_probe(hipStream_t s, const int8_t* a, const int8_t* b, int32_t* c)
{
    hipLaunchKernelGGL(mfma_layout_probe_kernel, dim3(1), dim3(64), 0, s, a, b, c);
    return hipGetLastError();
}
hipError_t launch_sqrt_probe(hipStream_t s, int32_t* mismatches)
{
    hipLaunchKernelGGL(sqrt_probe_kernel, dim3((1u << 24) / 256), dim3(256), 0, s, mismatches);
    return hipGetLastError();
}
hipError_t launch_lds_poison(hipStream_t s, uint32_t pattern)
{
    hipLaunchKernelGGL(lds_poison_kernel, dim3(2048), dim3(256), 0, s, pattern, (uint32_t*)nullptr);
    return hipGetLastError();
}
hipError_t launch_mfma16_layout_probe(hipStream_t s, const int8_t* a, const int8_t* b, int32_t* c)
{
    hipLaunchKernelGGL(mfma16_layout_probe_kernel, dim3(1), dim3(64), 0, s, a, b, c);
    return hipGetLastError();
}

}  // namespace sfm
