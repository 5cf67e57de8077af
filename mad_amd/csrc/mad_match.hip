// mad_match.hip -- descriptor correlation (a11), pose scoring (a12) and top-k for gfx950.
// Reference: MaD._match_dsc (mad/MaD.py:414-453) and the stable sort of
// MaD._filter_dsc_pairs (mad/MaD.py:480).
//
//  correlate : descriptor counts are <= 64, so rows are packed to int8 and the
//              N_hi x N_lo x 1024 contraction runs on v_mfma_i32_16x16x64_i8 with exact
//              int32 accumulation; score = dot / (|h| |l|) in float64.
//  pairs     : per hi row, ordered compaction of the columns whose score exceeds cc
//              (row-major order of np.where, MaD.py:423).
//  pose      : one wavefront per pair; the used hi anchors, the used lo anchors (binned into
//              cells of edge > 2 dist) and the cell offsets all sit in LDS.
//  top-k     : histogram of the integer match counts -> threshold count -> ordered
//              pick of the ties -> one-workgroup bitonic sort of the k survivors.
//
// The whole match is enqueued without a host round trip: row counts, pair counts and cloud
// sizes stay on the device (kernels read them through pointers and are persistent / grid-
// stride), buffers are sized from capacity hints, and the device raises a flag when a hint
// was too small -- the host then grows the buffer and re-runs (only ever on a first call).
#include <vector>

#include "mad_common.h"

// status words of one match, on the device (int32)
// [ST_NHI .. +3] and [ST_NLO .. +3] mirror the two sets' dev_n words {rows, range flag, border rejects, describe overflow}
// ST_NSEL: pairs whose exact count is computed when the pose search prunes by bounds (k_prune_select)
// ST_FLAG_SEL: more pairs were selected than the one-workgroup top-k over the selection holds (the match is repeated with the general one)
enum { ST_NPAIRS = 0, ST_LHI, ST_LLO, ST_NKEYS, ST_FLAG_C, ST_FLAG_PAIRS, ST_BAD, ST_NSEL, ST_NHI = 8, ST_NLO = 12, ST_FLAG_SEL = 16, ST_COUNT = 20 };

// ---------------------------------------------------------------------------
// per-row auxiliaries: int8 rows + norms, inverse rotations, result-row meta
// ---------------------------------------------------------------------------

// one wave per row: int8 copy, sqrt of the exact integer sum of squares, range check;
// rows in [n, roundup(n, 128)) are zero-filled so that the GEMM may read whole tiles
__global__ __launch_bounds__(256) void k_pack_rows(const int16_t *__restrict__ src, const int32_t *__restrict__ n_ptr, int D,
                                                   int8_t *__restrict__ dst, double *__restrict__ norm,
                                                   int32_t *__restrict__ bad) {
    const int64_t n = *n_ptr;
    const int64_t n_pad = (n + 127) / 128 * 128;
    const int lane = lane_id();
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
    for (int64_t row = wave; row < n_pad; row += nw) {
        if (row >= n) {
            for (int k = lane; k < D; k += MAD_WAVE) dst[row * D + k] = 0;
            if (lane == 0) norm[row] = 0.0;
            continue;
        }
        long long ss = 0;
        int oob = 0;
        for (int k = lane; k < D; k += MAD_WAVE) {
            const int v = src[row * D + k];
            if (v > 127 || v < -128) oob = 1;
            dst[row * D + k] = (int8_t)v;
            ss += (long long)v * v;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, MAD_WAVE);
        if (bad && __any(oob) && lane == 0) atomicExch(bad, 1);
        if (lane == 0) norm[row] = sqrt((double)ss);
    }
}

__device__ __forceinline__ void mat3_inv(const double *m, double *o) { mad_mat3_inv(m, o); }

__device__ __forceinline__ void mat3_mul(const double *a, const double *b, double *o) {
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) o[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}

// inverse of every row's Rfinal; meta = {anchor index, octave, main bin} (MaD.py:451) when the set supplies them
__global__ void k_row_aux(const double *__restrict__ row_R, const int32_t *__restrict__ n_ptr, double *__restrict__ row_Rinv,
                          const int32_t *__restrict__ row_anchor, const int32_t *__restrict__ row_main,
                          const int32_t *__restrict__ anc_index, const int32_t *__restrict__ anc_octave,
                          int32_t *__restrict__ meta) {
    const int64_t n = *n_ptr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        mat3_inv(row_R + 9 * i, row_Rinv + 9 * i);
        if (meta) {
            const int a = row_anchor[i];
            meta[3 * i] = anc_index[a]; meta[3 * i + 1] = anc_octave[a]; meta[3 * i + 2] = row_main[i];
        }
    }
}

// ---------------------------------------------------------------------------
// int8 MFMA correlation: C[hi][lo] = sum_k A[hi][k] * B[lo][k]
// ---------------------------------------------------------------------------

#define GEMM_BM 128
#define GEMM_BN 128
#define GEMM_BK 128                 // bytes of K per LDS stage
#define GEMM_LDA (GEMM_BK + 16)     // padded row: 144 B -> conflict-free ds_read_b128
#define GEMM_THREADS 256

typedef int v4i __attribute__((ext_vector_type(4)));

// the low (ll) / high (hh) 16 bits of two scalar words side by side: one scalar instruction each
__device__ __forceinline__ unsigned s_pack_ll(unsigned a, unsigned b) { unsigned r; asm("s_pack_ll_b32_b16 %0, %1, %2" : "=s"(r) : "s"(a), "s"(b)); return r; }
__device__ __forceinline__ unsigned s_pack_hh(unsigned a, unsigned b) { unsigned r; asm("s_pack_hh_b32_b16 %0, %1, %2" : "=s"(r) : "s"(a), "s"(b)); return r; }

// ---------------------------------------------------------------------------
// The contraction of a11 (MaD.py:420) on the matrix cores: 256 x 128 tiles (a wave owns 128 x 64: 12 fragment reads per 32 MFMAs instead of 8 per
// 16), operands staged global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass), three
// stages of 64 bytes of K in a ring with ONE barrier per stage and the loads of two stages in flight across it, and the tiles of
// SEVERAL matches (jobs) in one persistent grid: the matches of a step share the lo (map) rows, and the ~570 tiles of one
// match do not divide over 512 resident workgroups, the ~2 300 of four do.
//
// LDS image of a stage: 24 blocks of 16 rows x 64 B (16 of A, 8 of B), each written by one wave-instruction (lane l: row
// l / 4, 16-byte position l % 4) and read back as one MFMA fragment (lane L: row L % 16, K chunk L / 16).  Rows are 64 B
// apart, so rows r and r + 4 share banks: the chunk is stored at position chunk ^ (row / 4 % 4) -- applied to the SOURCE
// address, because the LDS side of an LDS-DMA is lane-linear -- which makes every quarter-wave of a read hit all 64 banks once.
// ---------------------------------------------------------------------------

#define G2_BM 256
#define G2_BN 128
#define G2_BK 64
#define G2_NSTAGE 3
#define G2_STAGE ((G2_BM + G2_BN) * G2_BK)
#define G2_LDS (G2_NSTAGE * G2_STAGE + 2 * (G2_BM + G2_BN) * 4)      // three stages + two norm buffers (the tile in its K loop, the next one)
static_assert(G2_BM == 256 && G2_BN == 128, "the tile arithmetic of k_corr_gemm2 shifts by these");

// row pitch (bytes) of the candidate flags: one byte per mask word (32 columns) of the padded matrix, rows aligned for 4-byte reads
__host__ __device__ __forceinline__ int64_t mad_cflag_pitch(int64_t ldc) { return (ldc / 32 + 3) & ~(int64_t)3; }

struct GemmJob {
    const int8_t *A, *B;           // hi rows, lo rows (int8, K bytes each, zero-padded to multiples of 128 rows)
    int32_t *C;
    const int32_t *n_hi, *n_lo;    // device: row counts
    int64_t cap_c;
    int32_t *status;
    const double *hn, *ln;
    uint32_t *mask;
    uint8_t *cflag;                // zeroed by the caller: one byte per mask word, set where the word has a bit (the pair kernels look nowhere else)
};
struct GemmBatch {
    int n_jobs;
    int K;
    int split_tail;                // the tiles of a last, short round as 128 x 128 halves (MAD_GEMM_NO_SPLIT: whole tiles, as in round 3)
    double cc;
    GemmJob job[MAD_BATCH_MAX];
};

__device__ __forceinline__ void glds16(const int8_t *g, int8_t *l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}

#ifdef MAD_PROBE_STAMPS      // diagnostic build: s_memtime at the phases of each workgroup's first tile (tools/probe_gemm.py)
__device__ long long g2_stamps[1024 * 8];
#define G2_STAMP(k) do { if (tid == 0 && first_tile) g2_stamps[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int mad_debug_g2_stamps(long long *out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g2_stamps), (size_t)n * 8) == hipSuccess ? 0 : -1;
}
#else
#define G2_STAMP(k) do { } while (0)
#endif

__global__ __launch_bounds__(GEMM_THREADS, 2) void k_corr_gemm2(GemmBatch G) {
    extern __shared__ __align__(16) int8_t g2_smem[];      // the stages and, behind them, the norms of the tile: ONE object
    float *sT = (float *)(g2_smem + G2_NSTAGE * G2_STAGE);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w >> 1, wn = w & 1;
    const int K = G.K, n_k = K / G2_BK;
    const int xcd = blockIdx.x & 7;
    const int64_t nslot = gridDim.x >> 3;
    int64_t t = blockIdx.x >> 3, base = 0;
#ifdef MAD_PROBE_STAMPS
    bool first_tile = true;
#endif
    G2_STAMP(0);
    const int ld_row = lane >> 2, ld_chunk = ((lane & 3) ^ ((lane >> 4) & 3)) * 16;
    const int rd_off = (lane & 15) * 64 + ((((lane >> 4) ^ ((lane & 15) >> 2)) & 3) * 16);
    const bool g2_split = G.split_tail != 0;
    int nbuf = 0;      // which of the two norm buffers the tile in its K loop uses
    for (int j = 0; j < G.n_jobs; j++) {
        const GemmJob &J = G.job[j];
        const int64_t hp = (((int64_t)*J.n_hi + 127) >> 7) << 7, lp = (((int64_t)*J.n_lo + 127) >> 7) << 7;
        if (hp * lp > J.cap_c) {
            if (blockIdx.x == 0 && tid == 0) J.status[ST_FLAG_C] = 1;
            continue;
        }
        // XCD x takes a contiguous run of the job's tiles in column-major order: ~1/8 of the lo rows against all hi rows, for
        // every job of the batch in turn -- the lo slice stays in its L2 from one match to the next
        const int64_t tiles_m = (hp + G2_BM - 1) >> 8, tiles = tiles_m * (lp >> 7), per_xcd = (tiles + 7) >> 3;      // (G2_BM 256, G2_BN 128)
        const int64_t begin = per_xcd * xcd, end = begin + per_xcd < tiles ? begin + per_xcd : tiles;
        const int64_t cnt = end > begin ? end - begin : 0;
        const int64_t ldm = lp / 32, ldf = mad_cflag_pitch(lp);
        // One tile of MT x 16 rows per wave-row (MT = 8: the 256 x 128 tile; MT = 4: its upper or lower half, 128 x 128, a wave owning
        // 64 x 64) -- the same stages, fragment layout and epilogue.  A tile goes through three phases:
        //   start(tile)    its norms into one of two norm buffers, this lane's source rows, the LDS-DMA pieces of its first two K stages
        //   the K loop     (its first barrier is where the norms and stage 0 become visible to everybody)
        //   the epilogue   registers and the norm buffer only -- no stage of LDS
        // and the workgroup's NEXT tile is started between the K loop and the epilogue of the current one, so that its first stages
        // (3 700 clocks of L2 / HBM latency, in-kernel stamps) fly under the epilogue (6 500 clocks) instead of in front of its K loop.
        const int8_t *srcA[4], *srcB[2];      // this lane's source rows of the tile in its K loop (rewritten by start(next) once that loop is over)
        auto start = [&](const int na, const int64_t row0, const int64_t col0, const int buf, auto norms_last) {      // na = MT / 2: 16-row blocks of A a wave stages per K stage
            float *const sN = sT + buf * (G2_BM + G2_BN);
            // |h| per row, cc |l| per column (zero rows count as norm 1, MaD.py:416).  Plain loads: the compiler drains the
            // vector-memory counter completely at their first use, so they come BEFORE the LDS-DMA pieces when an epilogue follows (it
            // must not wait for the pieces), and AFTER them for a workgroup's first tile of a job (nothing to do but wait: one round trip
            // instead of two).
            auto norms = [&]() {
                const int64_t r = row0 + tid < hp ? row0 + tid : hp - 1;
                const double v = J.hn[r];
                sN[tid] = (float)(v > 0 ? v : 1.0);
                if (tid < G2_BN) {
                    const double u = J.ln[col0 + tid];
                    sN[G2_BM + tid] = (float)(G.cc * (u > 0 ? u : 1.0));
                }
            };
            if (!decltype(norms_last)::value) norms();
            // blocks w, w + 4, ... of A (rows past the last 128-row block of a set with an odd number of them are read from its
            // last row and never stored), blocks w, w + 4 of B
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int64_t r = row0 + (w + 4 * i) * 16 + ld_row;
                srcA[i] = J.A + (r < hp ? r : hp - 1) * K + ld_chunk;
            }
#pragma unroll
            for (int i = 0; i < 2; i++) srcB[i] = J.B + (col0 + (w + 4 * i) * 16 + ld_row) * K + ld_chunk;
#pragma unroll
            for (int s = 0; s < 2; s++) {
                if (s >= n_k) break;
                int8_t *slot = g2_smem + s * G2_STAGE;
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if (i < na) glds16(srcA[i] + s * G2_BK, slot + (w + 4 * i) * 1024);
#pragma unroll
                for (int i = 0; i < 2; i++) glds16(srcB[i] + s * G2_BK, slot + (16 + w + 4 * i) * 1024);
            }
            if (decltype(norms_last)::value) norms();
        };
        // K loop + start of the next tile (next_na == 0: none) + epilogue
        auto run_tile = [&](auto mt_tag, const int64_t row0, const int64_t col0, const int buf, const int next_na, const int64_t next_row0, const int64_t next_col0) {
            constexpr int MT = decltype(mt_tag)::value, NA = MT / 2;
            const float *const sN = sT + buf * (G2_BM + G2_BN);
            G2_STAMP(1);
            auto issue = [&](int s) {
                int8_t *slot = g2_smem + (s % G2_NSTAGE) * G2_STAGE;
                const int k0 = s * G2_BK;
#pragma unroll
                for (int i = 0; i < NA; i++) glds16(srcA[i] + k0, slot + (w + 4 * i) * 1024);
#pragma unroll
                for (int i = 0; i < 2; i++) glds16(srcB[i] + k0, slot + (16 + w + 4 * i) * 1024);
            };
            v4i acc[MT][4];
#pragma unroll
            for (int m = 0; m < MT; m++)
#pragma unroll
                for (int n = 0; n < 4; n++) acc[m][n] = (v4i){0, 0, 0, 0};
            G2_STAMP(2);
            // One stage: wait for its bytes, barrier, the fragment reads up front, then the MFMAs with the LDS-DMA pieces of stage
            // s + 2 spread between them.  The order is pinned with sched_group_barrier: left alone the
            // compiler keeps two A fragments live, waits for lgkmcnt(0) five times a stage, and issues the pieces in a burst
            // right behind the barrier, next to the reads, where a piece costs most to issue (1 640 clocks per stage for the two
            // workgroups of a CU against 1 024 of MFMA issue, in-kernel stamps).
            // (the counted waits: the pieces of the stage after this one may still be in flight -- NA + 2 per wave; stores of the
            // previous tile's epilogue that are still on their way only make a wait longer, the loads return in order)
            auto stage = [&](int s, auto more) {
                if (decltype(more)::value || s + 1 < n_k) {
                    if (MT == 8) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
                } else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();      // stage s has landed for everybody; everybody has read stage s - 1, whose slot is filled next
                const int8_t *slot = g2_smem + (s % G2_NSTAGE) * G2_STAGE;
                v4i fa[MT], fb[4];
#pragma unroll
                for (int n = 0; n < 4; n++) fb[n] = *(const v4i *)(slot + (16 + wn * 4 + n) * 1024 + rd_off);
#pragma unroll
                for (int m = 0; m < MT; m++) fa[m] = *(const v4i *)(slot + (wm * MT + m) * 1024 + rd_off);
                if (decltype(more)::value) issue(s + 2);
#pragma unroll
                for (int m = 0; m < MT; m++)
#pragma unroll
                    for (int n = 0; n < 4; n++)
                        acc[m][n] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[m], fb[n], acc[m][n], 0, 0, 0);
                // the B fragments and the first two of A, then per A fragment: its four MFMAs, the read of the fragment after next, a piece
                __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
                for (int m = 0; m < MT; m++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                   // MFMA
                    if (m + 2 < MT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                   // DS read
                    if (decltype(more)::value && m < NA + 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // one LDS-DMA piece (VMEM read)
                }
            };
            for (int s = 0; s + 2 < n_k; s++) {
                if (s == 1) G2_STAMP(3);
                stage(s, std::true_type());
            }
            stage(n_k - 2, std::false_type());      // (K is a multiple of 128: at least two stages)
            stage(n_k - 1, std::false_type());
            G2_STAMP(4);
            __syncthreads();      // the last stages have been read by everybody: the next tile's first stages may land on them
            if (next_na) start(next_na, next_row0, next_col0, buf ^ 1, std::false_type());
            G2_STAMP(5);
            // Epilogue: besides the int32 dot products of the candidates, every tile leaves one bit per correlation in `mask`
            // ([row][ldc / 32] words): set when the score MAY exceed cc.  The test is a float32 product with a relative margin of 4e-6
            // (dot > cc |h| |l| (1 - margin)), i.e. a superset of the reference's float64 `dot / (|h| |l|) > cc` (MaD.py:423); the pair
            // kernels apply the exact expression to the flagged entries only (~0.2 % of the matrix) instead of dividing and comparing
            // N_hi x N_lo times, and never read the rest of C.
            // Round 4: a mask word is WRITTEN only where it has a bit, next to its candidate flag (`cflag`, zeroed by the caller; the pair
            // kernels read a mask word only behind a set flag, so the rest of `mask` is never looked at and need not be initialised).  Per
            // group of 4 rows x 64 columns the wave spends 12 vector instructions (convert, multiply, compare per entry; the compares land
            // in scalar registers), three scalar ORs and ONE branch; two groups in three have no candidate and end there.  Before, every
            // group paid four branches, eight scalar packs, eight v_writelane and a store: the epilogue was 36 % of a tile.
            if (row0 + wm * (MT * 16) < hp) {
                const unsigned lp32 = (unsigned)lp, ldm32 = (unsigned)ldm, ldf32 = (unsigned)ldf;
                // the tile's bases, formed once and held in scalar registers (left to itself the compiler re-reads the job from the kernel
                // arguments and repeats the 64-bit products in every rare branch)
                unsigned long long ctb = (unsigned long long)(J.C + row0 * lp + col0), mtb = (unsigned long long)(J.mask + row0 * ldm + col0 / 32),
                                   ftb = (unsigned long long)(J.cflag + row0 * ldf + col0 / 32);
                asm volatile("" : "+s"(ctb), "+s"(mtb), "+s"(ftb));
                float tl[4];
#pragma unroll
                for (int n = 0; n < 4; n++) {      // cc |l| moved towards "candidate" by the margin (|h| > 0, so the product moves with it)
                    const float v = sN[G2_BM + wn * 64 + n * 16 + (lane & 15)];
                    tl[n] = v - fabsf(v) * 4e-6f;
                }
#pragma unroll
                for (int m = 0; m < MT; m++) {
                    const float4 th4 = *(const float4 *)&sN[wm * (MT * 16) + m * 16 + (lane >> 4) * 4];
                    const float thv[4] = {th4.x, th4.y, th4.z, th4.w};
#pragma unroll
                    for (int jj = 0; jj < 4; jj++) {
                        unsigned long long bal[4];
                        bool cand[4];
#pragma unroll
                        for (int n = 0; n < 4; n++) {
                            cand[n] = (float)acc[m][n][jj] > thv[jj] * tl[n];
                            bal[n] = __ballot(cand[n]);
                        }
                        if ((bal[0] | bal[1] | bal[2] | bal[3]) == 0ull) continue;      // (uniform)
                        // the rest is the rare part: every address is formed here, from the tile's (uniform) bases and 32-bit offsets -- a tile
                        // spans 256 rows of at most 2^20 columns
                        unsigned rowt = (unsigned)(m * 16 + jj);      // (+ 4 x row group: the row within the wave's part of the tile)
                        asm volatile("" : "+s"(rowt));      // formed here, on the scalar unit, and not hoisted: 32 x 3 precomputed offsets spill
                        rowt += (unsigned)(wm * (MT * 16));
                        typedef __attribute__((address_space(1))) char gchar;      // (global, not flat, stores)
                        gchar *const crow = (gchar *)ctb + ((rowt + (unsigned)(lane >> 4) * 4u) * lp32 + (unsigned)(wn * 64 + (lane & 15))) * 4u;
#pragma unroll
                        for (int n = 0; n < 4; n++)
                            if (cand[n]) *(__attribute__((address_space(1))) int32_t *)(crow + n * 64) = acc[m][n][jj];
                        // the eight mask words of the group (4 row groups x 2 words of 32 columns) are cut from the four ballots on the
                        // scalar unit and handed to lanes 0..7
                        int wd = 0;
#define MAD_MASK_WORD(Gg, Hh) ((Gg) & 1 ? s_pack_hh((unsigned)(bal[2 * (Hh)] >> (32 * ((Gg) >> 1))), (unsigned)(bal[2 * (Hh) + 1] >> (32 * ((Gg) >> 1)))) \
                                        : s_pack_ll((unsigned)(bal[2 * (Hh)] >> (32 * ((Gg) >> 1))), (unsigned)(bal[2 * (Hh) + 1] >> (32 * ((Gg) >> 1)))))
#define MAD_WRITELANE(Gg, Hh, LANE) asm volatile("v_writelane_b32 %0, %1, " #LANE : "+v"(wd) : "s"(MAD_MASK_WORD(Gg, Hh)))
                        MAD_WRITELANE(0, 0, 0); MAD_WRITELANE(1, 0, 1); MAD_WRITELANE(2, 0, 2); MAD_WRITELANE(3, 0, 3);
                        MAD_WRITELANE(0, 1, 4); MAD_WRITELANE(1, 1, 5); MAD_WRITELANE(2, 1, 6); MAD_WRITELANE(3, 1, 7);
#undef MAD_WRITELANE
#undef MAD_MASK_WORD
                        if (lane < 8 && wd != 0) {      // lane: row group lane & 3, word lane >> 2
                            const unsigned r = rowt + (unsigned)(lane & 3) * 4u, wcol = (unsigned)(wn * 2 + (lane >> 2));
                            *(__attribute__((address_space(1))) uint32_t *)((gchar *)mtb + (r * ldm32 + wcol) * 4u) = (unsigned)wd;
                            ((__attribute__((address_space(1))) uint8_t *)ftb)[r * ldf32 + wcol] = 1;
                        }
                    }
                }
            }
            G2_STAMP(6);
#ifdef MAD_PROBE_STAMPS
            first_tile = false;
#endif
        };
        // Whole rounds of the XCD's workgroups take whole tiles.  What is left over -- 7 tiles for 64 workgroups on a C3 match, a
        // second round that costs a third of the launch (tools/probe_gemm_sizes.py: 512 tiles 26.5 us, 568 tiles 39.3 us) -- is dealt
        // in halves of a tile (128 x 128) when that spreads it over more of the XCD's workgroups: twice as many, each on a CU of its
        // own for half as long.  int32 sums are exact under any split.
        // (tile numbers fit 32 bits -- the matrix has fewer than 2^31 entries -- and a 32-bit division is a tenth of a 64-bit one)
        const int64_t full = g2_split ? (int64_t)((unsigned)cnt / (unsigned)nslot * (unsigned)nslot) : cnt, rem = cnt - full;
        // (quarters -- a third instantiation of the tile -- were built and measured twice: 37.0 us per C3 launch against 36.1 with halves
        // only in the first form of this kernel, 32.7 against 32.8 in this one: the third instantiation brings 8 spilled registers back,
        // which cost what the finer deal gains)
        const int parts = rem > 0 && 2 * rem <= nslot ? 2 : 1;
        const int64_t units = full + rem * parts;
        // this workgroup's tiles of the job, one after the other: unit t -> (rows, columns, MT / 2), na == 0 when the job has no more
        auto tile_of = [&](int64_t &tt, int64_t &row0, int64_t &col0) -> int {
            for (; tt < base + units; tt += nslot) {
                const int64_t u = tt - base;
                if (parts == 1 || u < full) {
                    const unsigned tile = (unsigned)(begin + u), tcol = tile / (unsigned)tiles_m;
                    row0 = (int64_t)(tile - tcol * (unsigned)tiles_m) * G2_BM; col0 = (int64_t)tcol * G2_BN;
                    return row0 + G2_BM / 2 >= hp ? 2 : 4;      // (a set's last, odd block of 128 rows: the upper half of a tile is all there is)
                }
                const unsigned h = (unsigned)(u - full), tile = (unsigned)(begin + full) + h / 2u, tcol = tile / (unsigned)tiles_m;      // (parts == 2)
                row0 = (int64_t)(tile - tcol * (unsigned)tiles_m) * G2_BM + (h & 1u) * (G2_BM / 2); col0 = (int64_t)tcol * G2_BN;
                if (row0 < hp) return 2;      // (else: below a set's last, odd block of 128 rows -- nothing)
            }
            return 0;
        };
        int64_t row0 = 0, col0 = 0, nrow0 = 0, ncol0 = 0;
        int na = tile_of(t, row0, col0);
        if (na) start(na, row0, col0, nbuf, std::false_type());      // (std::true_type() measured: 33.4 us per C3 launch against 32.8 -- two registers spill)
        while (na) {
            int64_t tn = t + nslot;
            const int nna = tile_of(tn, nrow0, ncol0);
            if (na == 4) run_tile(std::integral_constant<int, 8>(), row0, col0, nbuf, nna, nrow0, ncol0);
            else run_tile(std::integral_constant<int, 4>(), row0, col0, nbuf, nna, nrow0, ncol0);
            t = tn; na = nna; row0 = nrow0; col0 = ncol0; nbuf ^= 1;
        }
        base += units;
    }
}

// ---------------------------------------------------------------------------
// threshold + ordered compaction (np.where(preds > cc), MaD.py:423)
// ---------------------------------------------------------------------------

__device__ __forceinline__ double corr_score(int dot, double nh, double nl) {
    // zero rows stay un-normalised in the reference (MaD.py:416) -> divide by 1
    return (double)dot / ((nh > 0 ? nh : 1.0) * (nl > 0 ? nl : 1.0));
}

// The flagged entries of row i behind the candidate flags f .. f + 3 of this lane (a byte per mask word, set by the GEMM where the
// word has a bit: 0.2 % of the entries are candidates), gathered IN COLUMN ORDER into the wave's list in LDS: the four mask words of a lane
// are requested together (a word behind a clear flag is 0 without a load), a wave scan of the bit counts places every lane's
// columns.  Returns how many; beyond PAIR_LIST entries per pass the caller walks the words itself.  -- Rounds 1-3 read every word of
// the mask in both pair kernels, a workgroup and several barriers per row, and each thread chased its own word's bits one after the
// other; a list lets the dependent loads behind every candidate (dot product, norm) go out side by side, one candidate per lane.
#define PAIR_LIST 1024
#define PAIR_WORDS 4      // mask words (= candidate flags) of a lane per pass: a wave covers 64 x 4 x 32 = 8 192 columns
__device__ __forceinline__ int pair_list_row(const uint32_t *__restrict__ mask_row, const uint8_t *__restrict__ flag_row, int64_t f0, int64_t ldf,
                                             unsigned *__restrict__ list, unsigned mw[PAIR_WORDS]) {
    const int lane = (int)lane_id();
    const int64_t f = f0 + PAIR_WORDS * lane;
    const unsigned fl = f < ldf ? *(const unsigned *)(flag_row + f) : 0u;
    int pc = 0;
#pragma unroll
    for (int q = 0; q < PAIR_WORDS; q++) {
        mw[q] = ((fl >> (8 * q)) & 0xffu) ? mask_row[f + q] : 0u;
        pc += __popc(mw[q]);
    }
    const int inc = wave_incl_scan_i32(pc);
    const int total = __builtin_amdgcn_readlane(inc, MAD_WAVE - 1);
    if (total <= PAIR_LIST) {
        int o = inc - pc;
#pragma unroll
        for (int q = 0; q < PAIR_WORDS; q++) {
            unsigned m = mw[q];
            while (m) {
                const int b = __ffs(m) - 1;
                m &= m - 1;
                list[o++] = (unsigned)((f + q) * 32 + b);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // the lanes' writes above before the wave's reads of the list (LDS operations of a wave execute in order)
    }
    return total;
}

// Exact test of the entries the GEMM flagged (its float32 test with a margin is a superset of MaD.py:423's float64 one): a bit that
// fails is cleared in the mask, the row's count goes to row_cnt.  One WAVE per hi row, four rows per workgroup, persistent.
__global__ __launch_bounds__(256) void k_pair_count(const int32_t *__restrict__ C, uint32_t *__restrict__ mask, const uint8_t *__restrict__ cflag,
                                                    const int32_t *__restrict__ n_hi_ptr, const int32_t *__restrict__ n_lo_ptr,
                                                    const double *__restrict__ hn, const double *__restrict__ ln, double cc,
                                                    int32_t *__restrict__ row_cnt, const int32_t *__restrict__ status) {
    __shared__ unsigned s_list[4][PAIR_LIST];
    if (status[ST_FLAG_C]) return;
    const int64_t n_hi = *n_hi_ptr, n_lo = *n_lo_ptr;
    const int64_t ldc = (n_lo + GEMM_BN - 1) / GEMM_BN * GEMM_BN, ldm = ldc / 32, ldf = mad_cflag_pitch(ldc);
    const int lane = (int)lane_id();
    unsigned *const list = s_list[threadIdx.x >> 6];
    for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < n_hi; i += (int64_t)gridDim.x * 4) {
        const double nh = hn[i];
        uint32_t *const mrow = mask + i * ldm;
        int c = 0;
        for (int64_t f0 = 0; f0 < ldf; f0 += PAIR_WORDS * MAD_WAVE) {
            unsigned mw[PAIR_WORDS];
            const int total = pair_list_row(mrow, cflag + i * ldf, f0, ldf, list, mw);
            if (total > PAIR_LIST) {      // a pass with more candidates than the list holds: every lane walks its own words
#pragma unroll
                for (int q = 0; q < PAIR_WORDS; q++) {
                    unsigned m = mw[q], exact = 0;
                    if (!m) continue;
                    const int64_t w = f0 + PAIR_WORDS * lane + q;
                    while (m) {
                        const int b = __ffs(m) - 1;
                        m &= m - 1;
                        const int64_t j = w * 32 + b;
                        if (j < n_lo && corr_score(C[i * ldc + j], nh, ln[j]) > cc) exact |= 1u << b;
                    }
                    mrow[w] = exact;
                    c += __popc(exact);
                }
                continue;
            }
            for (int k = lane; k < total; k += MAD_WAVE) {      // (the wave's own LDS writes above are visible to it: LDS operations of a wave execute in order)
                const int64_t j = list[k];
                const bool ok = j < n_lo && corr_score(C[i * ldc + j], nh, ln[j]) > cc;
                if (!ok) atomicAnd(&mrow[j >> 5], ~(1u << (j & 31)));
                c += ok ? 1 : 0;
            }
        }
        c = wave_sum_i32(c);
        if (lane == 0) row_cnt[i] = c;
    }
}

// Ordered compaction of the flagged entries into the pair list (np.where's row-major order, MaD.py:423), the prefix sum of the rows'
// counts inside: a workgroup takes the rows 4 b .. 4 b + 3, 4 (b + G) .., one wave each; the number of pairs before its first row is
// a block reduction over the counts of the rows in between (a handful of cached loads per thread instead of a one-workgroup scan
// launch between count and emit: rounds 1-2), a wave's own offset adds the counts of the workgroup's earlier rows.  Inside a row the
// candidates stand in the wave's list in column order (pair_list_row): the k-th goes to position base + k, one candidate per lane.
// Workgroup 0 also forms the total (status[ST_NPAIRS], the overflow flag).  Marks the anchors that take part in a pair (the clouds
// of MaD.py:427-428).
// (Also tried, round 3: count + scan + emit as ONE launch, rows ticketed in order and the offsets by a decoupled look-back over 8-byte row
// descriptors -- correct, and 75-85 us per match against 28 for the three launches: 2 200 workgroups that reach the look-back together
// find no inclusive prefix nearby and poll each other's descriptors through the fabric.)
__global__ __launch_bounds__(256) void k_pair_emit2(const int32_t *__restrict__ C, const uint32_t *__restrict__ mask, const uint8_t *__restrict__ cflag,
                                                    const int32_t *__restrict__ n_hi_ptr, const int32_t *__restrict__ n_lo_ptr,
                                                    const double *__restrict__ hn, const double *__restrict__ ln,
                                                    const int32_t *__restrict__ row_cnt, int64_t cap_pairs,
                                                    int32_t *__restrict__ pair_hi, int32_t *__restrict__ pair_lo,
                                                    double *__restrict__ pair_score, const int32_t *__restrict__ hi_row_anchor,
                                                    const int32_t *__restrict__ lo_row_anchor, const int32_t *__restrict__ hi_canon,
                                                    const int32_t *__restrict__ lo_canon, uint8_t *__restrict__ used_hi,
                                                    uint8_t *__restrict__ used_lo, int32_t *__restrict__ status) {
    __shared__ unsigned s_list[4][PAIR_LIST];
    __shared__ long long s_sum[4];
    __shared__ int s_cnt[4];
    if (status[ST_FLAG_C]) return;
    const int64_t n_hi = *n_hi_ptr, n_lo = *n_lo_ptr;
    const int64_t ldc = (n_lo + GEMM_BN - 1) / GEMM_BN * GEMM_BN, ldm = ldc / 32, ldf = mad_cflag_pitch(ldc);
    const int lane = (int)lane_id(), wv = threadIdx.x >> 6;
    unsigned *const list = s_list[wv];
    // sum of row_cnt[a .. b) over the workgroup
    auto block_sum = [&](int64_t a, int64_t b) -> long long {
        long long v = 0;
        for (int64_t q = a + threadIdx.x; q < b; q += 256) v += row_cnt[q];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, MAD_WAVE);
        __syncthreads();
        if (lane == 0) s_sum[wv] = v;
        __syncthreads();
        return s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
    };
    if (blockIdx.x == 0) {
        const long long total = block_sum(0, n_hi);
        if (threadIdx.x == 0) {
            status[ST_NPAIRS] = (int32_t)min(total, (long long)INT32_MAX);
            if (total > cap_pairs) status[ST_FLAG_PAIRS] = 1;
        }
    }
    int64_t done = 0;      // rows whose counts are in `before`
    long long before = 0;
    for (int64_t i0 = (int64_t)blockIdx.x * 4; i0 < n_hi; i0 += (int64_t)gridDim.x * 4) {      // (uniform over the workgroup)
        before += block_sum(done, i0);
        const int64_t i = i0 + wv;
        const int count = i < n_hi ? row_cnt[i] : 0;
        if (lane == 0) s_cnt[wv] = count;
        __syncthreads();
        long long mine = before;
        for (int q = 0; q < wv; q++) mine += s_cnt[q];
        before += s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        done = min(i0 + 4, n_hi);
        if (count == 0 || mine + count > cap_pairs) continue;      // (a list that overflows is never read: the match is repeated)
        const double nh = hn[i];
        int64_t base = mine;
        auto put = [&](int64_t o, int64_t j) {
            pair_hi[o] = (int32_t)i;
            pair_lo[o] = (int32_t)j;
            pair_score[o] = corr_score(C[i * ldc + j], nh, ln[j]);
            if (used_lo) { const int a = lo_row_anchor ? lo_row_anchor[j] : (int)j; used_lo[lo_canon ? lo_canon[a] : a] = 1; }
        };
        for (int64_t f0 = 0; f0 < ldf; f0 += PAIR_WORDS * MAD_WAVE) {
            unsigned mw[PAIR_WORDS];
            const int total = pair_list_row(mask + i * ldm, cflag + i * ldf, f0, ldf, list, mw);
            if (total > PAIR_LIST) {      // more candidates in this pass than the list holds: every lane places its own, in order
                int pc = 0;
#pragma unroll
                for (int q = 0; q < PAIR_WORDS; q++) pc += __popc(mw[q]);
                int64_t o = base + wave_incl_scan_i32(pc) - pc;
#pragma unroll
                for (int q = 0; q < PAIR_WORDS; q++) {
                    unsigned m = mw[q];
                    while (m) {      // ascending columns: the row-major order of np.where (MaD.py:423)
                        const int b = __ffs(m) - 1;
                        m &= m - 1;
                        put(o++, (f0 + PAIR_WORDS * lane + q) * 32 + b);
                    }
                }
            } else {
                for (int k = lane; k < total; k += MAD_WAVE) put(base + k, (int64_t)list[k]);
            }
            base += total;
        }
        if (lane == 0 && used_hi) { const int a = hi_row_anchor ? hi_row_anchor[i] : (int)i; used_hi[hi_canon ? hi_canon[a] : a] = 1; }
    }
}

// ---------------------------------------------------------------------------
// pose scoring
// ---------------------------------------------------------------------------

// compact the used anchors' coordinates (order immaterial for the count); one workgroup of 1024 threads
struct CloudJob {
    const double *subv;          // n x 3 sub-voxel coordinates of the hi anchors; nullptr = nothing to do
    const uint8_t *used;
    int n;
    double *cloud;               // out: the used ones
    int32_t *count;              // out: how many (status[ST_LHI])
    const int32_t *hi_words, *lo_words;      // the two sets' dev_n words, mirrored into status for the host's read-back
    int32_t *status;
};

__device__ __forceinline__ void compact_cloud_block(const CloudJob &J, int *wt /* 17 */, int *s_base) {
    if (threadIdx.x == 0) *s_base = 0;
    if (J.hi_words && threadIdx.x < 4) J.status[ST_NHI + threadIdx.x] = J.hi_words[threadIdx.x];      // for the host's read-back
    if (J.lo_words && threadIdx.x >= 4 && threadIdx.x < 8) J.status[ST_NLO + threadIdx.x - 4] = J.lo_words[threadIdx.x - 4];
    __syncthreads();
    for (int b = 0; b < J.n; b += 1024) {
        const int i = b + threadIdx.x;
        const bool p = i < J.n && (!J.used || J.used[i]);
        int tot;
        const int pos = block_excl_scan(p ? 1 : 0, wt, &tot);
        if (p) {
            const int o = *s_base + pos;
            J.cloud[3 * o] = J.subv[3 * i]; J.cloud[3 * o + 1] = J.subv[3 * i + 1]; J.cloud[3 * o + 2] = J.subv[3 * i + 2];
        }
        __syncthreads();
        if (threadIdx.x == 0) *s_base += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *J.count = *s_base;
}

__global__ __launch_bounds__(1024) void k_compact_cloud(CloudJob J) {
    __shared__ int wt[17];
    __shared__ int s_base;
    compact_cloud_block(J, wt, &s_base);
}

__global__ void k_count_flags(const uint8_t *__restrict__ used, int n, int32_t *__restrict__ count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int v = (i < n && used[i]) ? 1 : 0;
    const int s = wave_sum_i32(v);
    if (lane_id() == 0 && s) atomicAdd(count, s);
}

// The lo cloud of one match (the used map anchors, a few thousand points at most) is binned into
// cells of edge > 2 * reach, so the ball around a query point meets at most 2 cells per axis
// (<= 4 z-runs).  Cloud, cell offsets (uint16) and the hi cloud all sit in LDS.
struct PoseGrid {
    double mn[3];
    double inv_cell[3];
    float inv_cell_f[3];
    int dim[3];
    int ncell;
};

__device__ __forceinline__ int pg_cell(double v, double mn, double inv, int dim) {
    const int c = (int)floor((v - mn) * inv);
    return min(max(c, 0), dim - 1);
}

// one workgroup: counting sort of the used points into cells; sorted points and offsets to global
// (1024 threads; cnt = G.ncell ints of LDS, wt = 17 ints, carry = 1 int)
__device__ __forceinline__ void pose_grid_block(const double *__restrict__ pts, const uint8_t *__restrict__ used, int n, const PoseGrid &G,
                                                int32_t *__restrict__ cell_start, unsigned short *__restrict__ cell_start16,
                                                double *__restrict__ sorted, float4 *__restrict__ sorted_f, int32_t *__restrict__ n_used,
                                                int *cnt, int *wt, int *carry_p) {
    int &carry = *carry_p;
    for (int c = threadIdx.x; c < G.ncell; c += 1024) cnt[c] = 0;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 1024)
        if (!used || used[i]) {
            const int c = (pg_cell(pts[3 * i], G.mn[0], G.inv_cell[0], G.dim[0]) * G.dim[1] +
                           pg_cell(pts[3 * i + 1], G.mn[1], G.inv_cell[1], G.dim[1])) * G.dim[2] +
                          pg_cell(pts[3 * i + 2], G.mn[2], G.inv_cell[2], G.dim[2]);
            atomicAdd(&cnt[c], 1);
        }
    __syncthreads();
    {   // exclusive scan of the cell counts: a thread sums a run of consecutive cells, ONE block scan over the runs (round 3; a block
        // scan per 1 024 cells before: sixteen of them, ~8 us of barriers, for the 25^3 cells of a 256^3 map)
        const int per = (G.ncell + 1023) / 1024;
        const int c0 = min((int)threadIdx.x * per, G.ncell), c1 = min(c0 + per, G.ncell);
        int sum = 0;
        for (int c = c0; c < c1; c++) sum += cnt[c];
        int tot;
        int run = block_excl_scan(sum, wt, &tot);
        for (int c = c0; c < c1; c++) {
            const int v = cnt[c];
            cnt[c] = run; cell_start[c] = run; cell_start16[c] = (unsigned short)run;
            run += v;
        }
        if (threadIdx.x == 0) carry = tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        cell_start[G.ncell] = carry; *n_used = carry;
        cell_start16[G.ncell] = (unsigned short)carry; cell_start16[G.ncell + 1] = 0;      // the search kernels copy whole dwords
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 1024)
        if (!used || used[i]) {
            const int c = (pg_cell(pts[3 * i], G.mn[0], G.inv_cell[0], G.dim[0]) * G.dim[1] +
                           pg_cell(pts[3 * i + 1], G.mn[1], G.inv_cell[1], G.dim[1])) * G.dim[2] +
                          pg_cell(pts[3 * i + 2], G.mn[2], G.inv_cell[2], G.dim[2]);
            const int o = atomicAdd(&cnt[c], 1);
            sorted[3 * o] = pts[3 * i]; sorted[3 * o + 1] = pts[3 * i + 1]; sorted[3 * o + 2] = pts[3 * i + 2];
            // float32 copy relative to the grid origin: what the candidate filter of k_pose_lds32 reads
            if (sorted_f)
                sorted_f[o] = make_float4((float)(pts[3 * i] - G.mn[0]), (float)(pts[3 * i + 1] - G.mn[1]), (float)(pts[3 * i + 2] - G.mn[2]), 0.f);
        }
}

__global__ __launch_bounds__(1024) void k_pose_grid_build(const double *__restrict__ pts, const uint8_t *__restrict__ used, int n,
                                                          PoseGrid G, int32_t *__restrict__ cell_start,
                                                          unsigned short *__restrict__ cell_start16, double *__restrict__ sorted,
                                                          float4 *__restrict__ sorted_f, int32_t *__restrict__ n_used, CloudJob J) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ int wt[17];
    __shared__ int carry;
    if (J.subv) {      // the hi cloud of the same match, compacted by this workgroup too: one launch fewer on the match's critical path
        compact_cloud_block(J, wt, &carry);
        __syncthreads();
    }
    pose_grid_block(pts, used, n, G, cell_start, cell_start16, sorted, sorted_f, n_used, (int *)smem, wt, &carry);
}

__host__ __device__ __forceinline__ size_t pad16(size_t b) { return (b + 15) & ~(size_t)15; }

// Occupancy bitmaps of the lo cloud over voxels of edge h (origin mn), two planes interleaved word by word:
//   outer: bit set iff some lo point lies within dist + h sqrt(3)/2 + slack of the voxel centre.  A transformed hi point that
//          falls into a clear voxel -- or outside the bitmap -- has no lo point within dist;
//   inner: bit set iff some lo point lies within dist - h sqrt(3)/2 - slack of the voxel centre.  A point in such a voxel has
//          a lo point within dist for certain and is counted without a search.
// Both hold whatever the float32 rounding of the point's voxel coordinates does (slack = 0.02 A against < 1e-3 A of error),
// so only the points in the shell between the two (about one in eleven on the 256^3 workload) go through the exact float64
// search, and the counts are those of the unfiltered search.  z-rows are padded to whole 32-bit words.
struct PoseBits {
    double mn[3];
    double h;
    int dim[3];
    int wz;      // words per z-row
};

// one launch marks up to four bitmaps (blockIdx.y): the two planes of the fine bitmap and of the coarse one (k_pose_bounds)
struct PoseBitsJobs {
    PoseBits B[4];
    double rad[4];
    int plane[4];
    int planes[4];      // planes interleaved in the job's bitmap: 2 (fine: outer, inner) or 1 (coarse: outer only)
    unsigned *bits[4];
};

// The marking of ONE point p into one bitmap job, by a group of `nthr` threads (thread t of the group): shared by k_pose_bits
// (one 256-thread workgroup per point) and k_pose_setup (four 256-thread groups per workgroup).
__device__ __forceinline__ void pose_bits_point(const PoseBits &B, double rad, int plane, int planes, unsigned *__restrict__ bits, double qx,
                                                double qy, double qz, int t0, int nthr) {
    const double inv_h = 1.0 / B.h, rad2 = rad * rad;
    const double px = qx - B.mn[0], py = qy - B.mn[1], pz = qz - B.mn[2];
    // voxel k has its centre at (k + 0.5) h; the index ranges below are supersets, the row test is exact
    const int x0 = max((int)floor((px - rad) * inv_h - 0.5), 0), x1 = min((int)ceil((px + rad) * inv_h - 0.5), B.dim[0] - 1);
    const int y0 = max((int)floor((py - rad) * inv_h - 0.5), 0), y1 = min((int)ceil((py + rad) * inv_h - 0.5), B.dim[1] - 1);
    const int ny = y1 - y0 + 1, nrow = (x1 - x0 + 1) * ny;
    for (int t = t0; t < nrow; t += nthr) {
        const int kx = x0 + t / ny, ky = y0 + t % ny;
        const double dx = (kx + 0.5) * B.h - px, dy = (ky + 0.5) * B.h - py;
        const double rem = rad2 - dx * dx - dy * dy;
        if (rem < 0.0) continue;
        const double sq = sqrt(rem);
        // outer plane: a superset of the voxels within rad is harmless; inner plane: it must be a subset
        const int z0 = max(plane == 0 ? (int)floor((pz - sq) * inv_h - 0.5) : (int)ceil((pz - sq) * inv_h - 0.5 + 1e-9), 0);
        const int z1 = min(plane == 0 ? (int)ceil((pz + sq) * inv_h - 0.5) : (int)floor((pz + sq) * inv_h - 0.5 - 1e-9), B.dim[2] - 1);
        if (z1 < z0) continue;
        unsigned *row = bits + planes * ((size_t)kx * B.dim[1] + ky) * B.wz + plane;      // planes interleaved: [outer][inner] per word
        for (int w = z0 >> 5; w <= (z1 >> 5); w++) {
            const int lo = max(z0 - 32 * w, 0), hi = min(z1 - 32 * w, 31);
            const unsigned m = (hi == 31 ? 0xffffffffu : ((1u << (hi + 1)) - 1u)) & ~((1u << lo) - 1u);
            atomicOr(&row[planes * w], m);
        }
    }
}

__global__ __launch_bounds__(256) void k_pose_bits(const double *__restrict__ sorted, const int32_t *__restrict__ cell_start, int ncell,
                                                   PoseBitsJobs J) {
    const PoseBits B = J.B[blockIdx.y];
    const int l_lo = cell_start[ncell];
    for (int p = blockIdx.x; p < l_lo; p += gridDim.x)
        pose_bits_point(B, J.rad[blockIdx.y], J.plane[blockIdx.y], J.planes[blockIdx.y], J.bits[blockIdx.y], sorted[3 * p], sorted[3 * p + 1],
                        sorted[3 * p + 2], (int)threadIdx.x, 256);
}

// float32 map from a hi-cloud point to bitmap voxel coordinates for one pair: v = M c + t
struct PoseVox {
    float m[9], t[3];
};

__device__ __forceinline__ void pose_vox_setup(const double *R, double ph0, double ph1, double ph2, double pl0, double pl1, double pl2,
                                               const PoseBits &B, PoseVox *V) {
    const double ih = 1.0 / B.h;
    for (int i = 0; i < 9; i++) V->m[i] = (float)(R[i] * ih);
    // x = R (c - ph) + pl = R c + (pl - R ph)
    V->t[0] = (float)(((pl0 - (ph0 * R[0] + ph1 * R[1] + ph2 * R[2])) - B.mn[0]) * ih);
    V->t[1] = (float)(((pl1 - (ph0 * R[3] + ph1 * R[4] + ph2 * R[5])) - B.mn[1]) * ih);
    V->t[2] = (float)(((pl2 - (ph0 * R[6] + ph1 * R[7] + ph2 * R[8])) - B.mn[2]) * ih);
}

// the two bitmap words (outer, inner) of a hi-cloud point and the bit to test in them; zero for points outside the bitmap
__device__ __forceinline__ void pose_vox_fetch(const PoseVox &V, float cx, float cy, float cz, const PoseBits &B,
                                               const unsigned *bits, uint2 *word, int *bit) {
    const float vx = fmaf(cz, V.m[2], fmaf(cy, V.m[1], fmaf(cx, V.m[0], V.t[0])));
    const float vy = fmaf(cz, V.m[5], fmaf(cy, V.m[4], fmaf(cx, V.m[3], V.t[1])));
    const float vz = fmaf(cz, V.m[8], fmaf(cy, V.m[7], fmaf(cx, V.m[6], V.t[2])));
    const int ix = cvt_floor(vx), iy = cvt_floor(vy), iz = cvt_floor(vz);
    *bit = iz & 31;
    *word = make_uint2(0u, 0u);
    // one test, 24-bit multiplies (full rate; a 32-bit integer multiply costs four issue slots): the bitmap holds at most 2^21 word pairs
    const bool in = ((unsigned)ix < (unsigned)B.dim[0]) & ((unsigned)iy < (unsigned)B.dim[1]) & ((unsigned)iz < (unsigned)B.dim[2]);
    if (in) *word = ((const uint2 *)bits)[mad_u24(mad_u24((unsigned)ix, (unsigned)B.dim[1], (unsigned)iy), (unsigned)B.wz, (unsigned)(iz >> 5))];
}

// Two-phase count for one pair (one wave): the bitmap test for every hi point, survivors collected in the wave's LDS
// stack, and the exact search `exact(a)` run on them 64 at a time, so that its lanes stay full.  The bitmap words of
// POSE_BATCH x 64 points are requested before the first one is looked at: the loads are scattered L2 hits, and a wave
// that waited for each in turn would spend most of its time doing so.
#define POSE_STACK 128
#define POSE_BATCH 4
template <class Cloud32, class Exact>
__device__ __forceinline__ int pose_count_filtered(int l_hi, const PoseVox &V, const PoseBits &B, const unsigned *bits,
                                                   unsigned short *stack, Cloud32 cloud32, Exact exact) {
    const int lane = lane_id();
    int nq = 0, cnt = 0;
    for (int a0 = 0; a0 < l_hi; a0 += MAD_WAVE * POSE_BATCH) {
        uint2 word[POSE_BATCH];
        int bit[POSE_BATCH];
#pragma unroll
        for (int u = 0; u < POSE_BATCH; u++) {
            const int a = a0 + u * MAD_WAVE + lane;
            word[u] = make_uint2(0u, 0u); bit[u] = 0;
            if (a < l_hi) {
                float cx, cy, cz;
                cloud32(a, cx, cy, cz);
                pose_vox_fetch(V, cx, cy, cz, B, bits, &word[u], &bit[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < POSE_BATCH; u++) {
            if (a0 + u * MAD_WAVE >= l_hi) break;      // wave-uniform
            const bool certain = (word[u].y >> bit[u]) & 1u;      // inner plane: counted without a search
            cnt += certain ? 1 : 0;
            const bool sv = ((word[u].x >> bit[u]) & 1u) && !certain;
            const unsigned long long bal = __ballot(sv);
            if (sv) stack[nq + __popcll(bal & lanemask_lt())] = (unsigned short)(a0 + u * MAD_WAVE + lane);
            nq += __popcll(bal);
            if (nq >= MAD_WAVE) {      // wave-uniform
                __builtin_amdgcn_wave_barrier();
                const int a2 = stack[nq - MAD_WAVE + lane];
                __builtin_amdgcn_wave_barrier();
                cnt += exact(a2) ? 1 : 0;
                nq -= MAD_WAVE;
            }
        }
    }
    if (nq > 0) {
        __builtin_amdgcn_wave_barrier();
        const int a2 = lane < nq ? stack[lane] : 0;
        __builtin_amdgcn_wave_barrier();
        if (lane < nq) cnt += exact(a2) ? 1 : 0;
    }
    return cnt;
}

// Per-pair rigid transform, written once by k_pose_prep (a thread per pair: the three-level chain pair -> row -> anchor
// of dependent scattered loads is hidden by the width of that launch) and read back as one contiguous 120-byte record
// by the search kernels, which fetch the record of their next pair while they work on the current one.
struct PosePair {
    double R[9];      // inv(lo.Rfinal) @ hi.Rfinal (MaD.py:438)
    double ph[3], pl[3];
    PoseVox vf, vc;   // hi-cloud point -> voxel coordinates of the fine / of the coarse bitmap (float32; pose_vox_setup)
};

__device__ __forceinline__ void pose_prep_range(const int32_t *__restrict__ pair_hi, const int32_t *__restrict__ pair_lo, int64_t n_pairs,
                                                int64_t first, int64_t stride, const double *__restrict__ hi_p, const double *__restrict__ hi_R,
                                                const double *__restrict__ lo_p, const double *__restrict__ lo_Rinv,
                                                const int32_t *__restrict__ hi_row_anchor, const int32_t *__restrict__ lo_row_anchor,
                                                const PoseBits &Bf, const PoseBits &Bc, PosePair *__restrict__ rec) {
    for (int64_t p = first; p < n_pairs; p += stride) {
        const int ih = pair_hi[p], il = pair_lo[p];
        PosePair P;
        mat3_mul(lo_Rinv + 9 * il, hi_R + 9 * ih, P.R);
        const int ah = hi_row_anchor ? hi_row_anchor[ih] : ih, al = lo_row_anchor ? lo_row_anchor[il] : il;
        for (int d = 0; d < 3; d++) { P.ph[d] = hi_p[3 * ah + d]; P.pl[d] = lo_p[3 * al + d]; }
        // the float64 part of the bitmap maps here, one thread per pair, not once per wave in the search kernels
        pose_vox_setup(P.R, P.ph[0], P.ph[1], P.ph[2], P.pl[0], P.pl[1], P.pl[2], Bf, &P.vf);
        pose_vox_setup(P.R, P.ph[0], P.ph[1], P.ph[2], P.pl[0], P.pl[1], P.pl[2], Bc, &P.vc);
        rec[p] = P;
    }
}

__global__ __launch_bounds__(256) void k_pose_prep(const int32_t *__restrict__ pair_hi, const int32_t *__restrict__ pair_lo,
                                                   const int32_t *__restrict__ status, int64_t cap_pairs,
                                                   const double *__restrict__ hi_p, const double *__restrict__ hi_R,
                                                   const double *__restrict__ lo_p, const double *__restrict__ lo_Rinv,
                                                   const int32_t *__restrict__ hi_row_anchor, const int32_t *__restrict__ lo_row_anchor,
                                                   PoseBits Bf, PoseBits Bc, PosePair *__restrict__ rec) {
    if (status[ST_FLAG_C] || status[ST_FLAG_PAIRS]) return;
    const int64_t n_pairs = min((int64_t)status[ST_NPAIRS], cap_pairs);
    pose_prep_range(pair_hi, pair_lo, n_pairs, (int64_t)blockIdx.x * 256 + threadIdx.x, (int64_t)gridDim.x * 256, hi_p, hi_R, lo_p, lo_Rinv,
                    hi_row_anchor, lo_row_anchor, Bf, Bc, rec);
}

// What a match needs between its pair list and its pose search, in ONE launch of 1024-thread workgroups with different roles
// (they depend on the pair list and on the "takes part in a pair" flags only, not on each other):
//   workgroup 0                : the used lo anchors binned into the search grid (k_pose_grid_build)
//   workgroup 1                : the used hi anchors compacted into the hi cloud; the sets' row counts mirrored into the status words
//   workgroups 2 .. 2 + n_bits : the occupancy bitmaps of the lo cloud (k_pose_bits), a 256-thread group per point, straight from the
//                                anchor list and its flags (the marks do not depend on the order of the points)
//   the rest                   : the per-pair records (k_pose_prep)
// -- four launches before (grid, zero fill of the bitmaps, bits, prep; the zero fill now travels with the match's status words).
struct PoseSetup {
    const double *pts; const uint8_t *used; int n;      // the lo anchors and their flags (nullptr: all)
    PoseGrid G;
    int32_t *cell_start; unsigned short *cell_start16; double *sorted; float4 *sorted_f; int32_t *n_used;
    CloudJob J;
    PoseBitsJobs bits; int n_bit_jobs; int n_bits_wgs;
    const int32_t *pair_hi, *pair_lo; const int32_t *status; int64_t cap_pairs;
    const double *hi_p, *hi_R, *lo_p, *lo_Rinv; const int32_t *hi_row_anchor, *lo_row_anchor;
    PoseBits Bf, Bc; PosePair *rec;
    int roles;      // bit mask, 15 = all (diagnostic: MAD_PROBE_SETUP repeats the launch role by role for a kernel trace)
};

__global__ __launch_bounds__(1024, 8) void k_pose_setup(PoseSetup S) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ int wt[17];
    __shared__ int carry;
    const int b = (int)blockIdx.x;
    if (b == 0) {
        if (S.roles & 1) pose_grid_block(S.pts, S.used, S.n, S.G, S.cell_start, S.cell_start16, S.sorted, S.sorted_f, S.n_used, (int *)smem, wt, &carry);
        return;
    }
    if (b == 1) {
        if (S.J.subv && (S.roles & 2)) compact_cloud_block(S.J, wt, &carry);
        return;
    }
    if (b < 2 + S.n_bits_wgs) {
        if (!(S.roles & 4)) return;
        const int groups = S.n_bits_wgs * 4 / S.n_bit_jobs;              // 256-thread groups per bitmap job
        const int g = (b - 2) * 4 + ((int)threadIdx.x >> 8);
        const int job = g / groups, first = g % groups;
        if (job >= S.n_bit_jobs) return;
        const PoseBits B = S.bits.B[job];
        for (int p = first; p < S.n; p += groups) {
            if (S.used && !S.used[p]) continue;
            pose_bits_point(B, S.bits.rad[job], S.bits.plane[job], S.bits.planes[job], S.bits.bits[job], S.pts[3 * p], S.pts[3 * p + 1],
                            S.pts[3 * p + 2], (int)threadIdx.x & 255, 256);
        }
        return;
    }
    if (S.status[ST_FLAG_C] || S.status[ST_FLAG_PAIRS] || !(S.roles & 8)) return;
    const int64_t n_pairs = min((int64_t)S.status[ST_NPAIRS], S.cap_pairs);
    const int64_t wg = b - 2 - S.n_bits_wgs, n_wg = (int64_t)gridDim.x - 2 - S.n_bits_wgs;
    pose_prep_range(S.pair_hi, S.pair_lo, n_pairs, wg * 1024 + threadIdx.x, n_wg * 1024, S.hi_p, S.hi_R, S.lo_p, S.lo_Rinv, S.hi_row_anchor,
                    S.lo_row_anchor, S.Bf, S.Bc, S.rec);
}

#define POSE_LDS_THREADS 1024
__device__ __forceinline__ int topk_threshold(const int32_t *__restrict__ hist, int nbins, int64_t k, int *wt, int *sh, int *need);
#define POSE_PARTS 4                        // waves that share one listed pair of a pruned search
#define POSE_RING 256                       // survivor queue of a wave (entries), a power of two >= 2 x 64
#define POSE_WAVE_LDS (POSE_RING * 2 + 256) // bytes per wave: the queue + two pair records of 128 bytes
#define POSE_OWN_CAP 2048                   // pairs a workgroup of k_pose_lds may list for itself (PoseOwnSel)

// k_pose_lds selecting its own pairs (see there); upper == nullptr: the pairs are all, or listed in `sel`
struct PoseOwnSel {
    const unsigned short *upper;      // per pair: upper bound of its count (k_pose_bounds)
    const int32_t *hist;              // histogram of the lower bounds, nbins bins
    int nbins;
    int64_t k;
    int32_t *sel_out;                 // the listed pairs of all workgroups, in no particular order; status_w[ST_NSEL] counts them
    int64_t sel_cap;
    int32_t *status_w;
};

// MaD.py:433-448.  One wave per pair, lanes over the hi cloud.  `dd_lim` is the smallest double whose square root is >=
// dist, so dd < dd_lim is exactly the reference's sqrt(dd) < dist without the root.
// Two phases per pair: every hi point is tested against the occupancy bitmap of the lo cloud (float32, ~40 instructions
// per 64 points), the survivors (about one in eight) are queued, and the exact float64 search runs on 64 queued points
// at a time.  A pair leaves fewer than 64 behind; they stay queued and share a round with the first survivors of the
// wave's NEXT pair (each entry carries a slot bit, the two pairs' transforms sit side by side in LDS), so that the rounds
// run with full lanes: ~0.9 rounds per pair instead of ~1.3.
__global__ __launch_bounds__(POSE_LDS_THREADS) void k_pose_lds(const int32_t *__restrict__ status, int64_t cap_pairs,
                                                               const PosePair *__restrict__ rec,
                                                               const double *__restrict__ hi_cloud, const double *__restrict__ lo_sorted,
                                                               const int32_t *__restrict__ cell_start,
                                                               const unsigned short *__restrict__ cell_start16, PoseGrid G, int l_hi_cap,
                                                               int l_lo_cap, float reach, double dd_lim, PoseBits B,
                                                               const unsigned *__restrict__ bits, int32_t *__restrict__ counts,
                                                               const int32_t *__restrict__ sel, PoseOwnSel own) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ int t_wt[POSE_LDS_THREADS / MAD_WAVE + 1];
    __shared__ int t_sh[3];
    __shared__ int s_nloc, s_gbase;
    if (status[ST_FLAG_C] || status[ST_FLAG_PAIRS]) return;
    // LDS regions, each a multiple of 16 bytes (pose_device on the host mirrors this)
    double *cl = (double *)smem;                                                     // hi cloud
    double *lp = (double *)(smem + pad16((size_t)l_hi_cap * 24));                    // sorted lo cloud
    float4 *clf = (float4 *)((unsigned char *)lp + pad16((size_t)l_lo_cap * 24));    // hi cloud in float32: the bitmap test reads it
    unsigned short *cs = (unsigned short *)(clf + l_hi_cap);                         // cell offsets
    unsigned char *wave_lds = (unsigned char *)cs + pad16((size_t)(G.ncell + 1) * 2) + (threadIdx.x >> 6) * POSE_WAVE_LDS;
    unsigned short *ring = (unsigned short *)wave_lds;                               // this wave's survivor queue
    double *recs = (double *)(wave_lds + POSE_RING * 2);                             // this wave's two pair records, 16 doubles each
    // sel != nullptr: only the pairs listed there (status[ST_NSEL] of them; the others keep the lower bound k_pose_bounds left in
    // counts).  The listed pairs are few (hundreds) and heavy (good poses: most hi points reach the exact search), so each is cut
    // into POSE_PARTS runs of hi points, one wave each, whose counts add up atomically in counts[pair] (zeroed by k_prune_select).
    // own.upper != nullptr: the selection itself happens here as well (k_prune_select's work, one launch fewer).  Every workgroup
    // derives T, the k-th largest lower bound, from the histogram of k_pose_bounds, takes every gridDim.x-th pair, lists
    // those whose upper bound reaches T in LDS (and, for the top-k kernel behind it, in own.sel_out), and searches ITS list with its
    // 16 waves.  A workgroup whose list outgrows
    // POSE_OWN_CAP raises ST_FLAG_SEL -- the host repeats the match with the separate selection kernel.
    const bool own_sel = own.upper != nullptr;
    const bool listed = own_sel || sel != nullptr;
    const int parts = listed ? POSE_PARTS : 1;
    const int l_hi = status[ST_LHI];
    const int l_lo = cell_start[G.ncell];
    stage_lds(lp, lo_sorted, (size_t)l_lo * 24);
    stage_lds(cs, cell_start16, (size_t)((G.ncell + 2) & ~1) * 2);
    stage_lds(cl, hi_cloud, (size_t)l_hi * 24);
    const int lane = lane_id();
    int *own_list = (int *)((unsigned char *)cs + pad16((size_t)(G.ncell + 1) * 2) + (POSE_LDS_THREADS / MAD_WAVE) * POSE_WAVE_LDS);
    if (own_sel) {
        const int64_t n_all = min((int64_t)status[ST_NPAIRS], cap_pairs);
        int need;
        const int T = max(topk_threshold(own.hist, own.nbins, min(own.k, n_all), t_wt, t_sh, &need), 0);
        if (threadIdx.x == 0) s_nloc = 0;
        __syncthreads();
        // pair i belongs to workgroup i % gridDim.x: good pairs cluster by hi row (runs of consecutive pairs), and a run that
        // stayed in one workgroup made it the launch's tail (runs of 64: 28 us against 11 for the same pairs dealt evenly)
        const int64_t per_wg = (n_all + gridDim.x - 1) / gridDim.x;
        for (int64_t j0 = 0; j0 < per_wg; j0 += POSE_LDS_THREADS) {      // (workgroup-uniform trip count: the ballots need whole waves)
            const int64_t i = (int64_t)blockIdx.x + (int64_t)gridDim.x * (j0 + threadIdx.x);
            const bool take = i < n_all && (int)own.upper[i] >= T;
            const unsigned long long bal = __ballot(take);
            if (bal == 0ull) continue;
            int base = 0;
            if (lane == 0) base = atomicAdd(&s_nloc, __popcll(bal));
            base = __shfl(base, 0, MAD_WAVE);
            const int o = base + __popcll(bal & lanemask_lt());
            if (take) {
                if (o < POSE_OWN_CAP) own_list[o] = (int)i;
                counts[i] = 0;      // the parts of a listed pair add up atomically
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const int n_loc = s_nloc;
            if (n_loc > POSE_OWN_CAP) { own.status_w[ST_FLAG_SEL] = 1; s_gbase = -1; }
            else s_gbase = n_loc ? atomicAdd(own.status_w + ST_NSEL, n_loc) : 0;
        }
    }
    __syncthreads();
    if (own_sel) {
        if (s_gbase < 0) return;      // (workgroup-uniform)
        for (int i = threadIdx.x; i < s_nloc; i += POSE_LDS_THREADS)
            if (s_gbase + i < own.sel_cap) own.sel_out[s_gbase + i] = own_list[i];
            else own.status_w[ST_FLAG_SEL] = 1;
    }
    const int *lst = own_sel ? own_list : sel;      // (LDS or global: generic loads)
    const int64_t n_pairs = own_sel ? (int64_t)s_nloc * parts
                                    : (sel ? (int64_t)status[ST_NSEL] * parts : min((int64_t)status[ST_NPAIRS], cap_pairs));      // work items
    for (int i = threadIdx.x; i < l_hi; i += POSE_LDS_THREADS)
        clf[i] = make_float4((float)cl[3 * i], (float)cl[3 * i + 1], (float)cl[3 * i + 2], 0.f);
    __syncthreads();
    const int64_t wave = own_sel ? (int64_t)(threadIdx.x >> 6)
                                 : (int64_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (POSE_LDS_THREADS / MAD_WAVE) + (threadIdx.x >> 6)));
    const int64_t nwaves = own_sel ? (int64_t)(POSE_LDS_THREADS / MAD_WAVE) : (int64_t)gridDim.x * (POSE_LDS_THREADS / MAD_WAVE);
    PosePair cur;
    if (wave < n_pairs) cur = rec[listed ? lst[wave / parts] : wave];
    const float mnx = (float)G.mn[0], mny = (float)G.mn[1], mnz = (float)G.mn[2];
    const int part_len = (l_hi + parts - 1) / parts;
    auto emit = [&](int64_t p, int c) {      // a pair's (partial) count
        if (listed) atomicAdd(&counts[p], c);
        else counts[p] = c;
    };

    // the exact search for queue entry e = slot << 15 | hi point
    auto exact = [&](int e) -> bool {
        const double *P = recs + 16 * (e >> 15);      // R[9], ph[3], pl[3] of the entry's pair
        const int a = e & 0x7fff;
        const double d0 = cl[3 * a] - P[9], d1 = cl[3 * a + 1] - P[10], d2 = cl[3 * a + 2] - P[11];
        const double x = (d0 * P[0] + d1 * P[1] + d2 * P[2]) + P[12];      // MaD.py:440-444
        const double y = (d0 * P[3] + d1 * P[4] + d2 * P[5]) + P[13];
        const double z = (d0 * P[6] + d1 * P[7] + d2 * P[8]) + P[14];
        // cells met by a slightly larger ball (reach = dist + 0.01), in float32: a superset is harmless
        const float xf = (float)x - mnx, yf = (float)y - mny, zf = (float)z - mnz;
        const int x0 = cvt_floor((xf - reach) * G.inv_cell_f[0]), x1 = cvt_floor((xf + reach) * G.inv_cell_f[0]);
        const int y0 = cvt_floor((yf - reach) * G.inv_cell_f[1]), y1 = cvt_floor((yf + reach) * G.inv_cell_f[1]);
        const int z0 = cvt_floor((zf - reach) * G.inv_cell_f[2]), z1 = cvt_floor((zf + reach) * G.inv_cell_f[2]);
        bool hit = false;
        if (x1 >= 0 && x0 < G.dim[0] && y1 >= 0 && y0 < G.dim[1] && z1 >= 0 && z0 < G.dim[2]) {
            const int zz0 = max(z0, 0), zz1 = min(z1, G.dim[2] - 1) + 1;
            const int xa = max(x0, 0), xb = min(x1, G.dim[0] - 1), ya = max(y0, 0), yb = min(y1, G.dim[1] - 1);
            // the ball meets at most 2 x 2 columns: fetch all four z-runs before walking any of them
            const int c00 = __mul24(__mul24(xa, G.dim[1]) + ya, G.dim[2]), c01 = __mul24(__mul24(xa, G.dim[1]) + yb, G.dim[2]);      // <= 30 000 cells
            const int c10 = __mul24(__mul24(xb, G.dim[1]) + ya, G.dim[2]), c11 = __mul24(__mul24(xb, G.dim[1]) + yb, G.dim[2]);
            int s[4], e4[4];
            s[0] = cs[c00 + zz0]; e4[0] = cs[c00 + zz1];
            s[1] = cs[c01 + zz0]; e4[1] = (yb != ya) ? cs[c01 + zz1] : s[1];
            s[2] = cs[c10 + zz0]; e4[2] = (xb != xa) ? cs[c10 + zz1] : s[2];
            s[3] = cs[c11 + zz0]; e4[3] = (xb != xa && yb != ya) ? cs[c11 + zz1] : s[3];
            // one loop over the concatenation of the four runs: the wave then iterates max-over-lanes of the
            // TOTAL candidate count instead of the sum over runs of the per-run maxima
            const int n0 = e4[0] - s[0], n1 = n0 + (e4[1] - s[1]), n2 = n1 + (e4[2] - s[2]), n3 = n2 + (e4[3] - s[3]);
            const int b1 = s[1] - n0, b2 = s[2] - n1, b3 = s[3] - n2;
            for (int t = 0; t < n3 && !hit; t++) {
                const int q = t + (t < n0 ? s[0] : (t < n1 ? b1 : (t < n2 ? b2 : b3)));
                const double e0 = lp[3 * q] - x, e1 = lp[3 * q + 1] - y, e2 = lp[3 * q + 2] - z;
                hit = (e0 * e0 + e1 * e1 + e2 * e2) < dd_lim;      // MaD.py:447-448
            }
        }
        return hit;
    };

    // queue state, all wave-uniform
    int head = 0, count = 0;          // ring[(head + i) & (POSE_RING - 1)], i < count, oldest first
    int slot = 0;                     // slot bit of the pair being filtered
    int cnt_cur = 0, cnt_old = 0;     // hits so far of that pair / of the previous one
    int old_left = 0;                 // queued entries that still belong to the previous pair (they are at the head)
    int64_t p_old = -1;
    auto round = [&](int n_take) {    // the exact search for the n_take <= 64 oldest entries
        __builtin_amdgcn_wave_barrier();
        const int e = lane < n_take ? ring[(head + lane) & (POSE_RING - 1)] : 0;
        __builtin_amdgcn_wave_barrier();
        bool hit = false;
        if (lane < n_take) hit = exact(e);
        const unsigned long long bal = __ballot(hit), mine = __ballot(lane < n_take && (e >> 15) == slot);
        cnt_cur += __popcll(bal & mine);
        cnt_old += __popcll(bal & ~mine);
        head = (head + n_take) & (POSE_RING - 1);
        count -= n_take;
        if (old_left > 0) {
            old_left = max(old_left - n_take, 0);
            if (old_left == 0 && lane == 0) emit(p_old, cnt_old);      // the previous pair is complete
        }
    };
    for (int64_t it = wave; it < n_pairs; it += nwaves) {
        const int64_t p = listed ? lst[it / parts] : it;
        const int a_begin = listed ? (int)(it % parts) * part_len : 0, a_end = listed ? min(a_begin + part_len, l_hi) : l_hi;      // this item's hi points
        PosePair nxt;      // requested now, needed one iteration later
        if (it + nwaves < n_pairs) nxt = rec[listed ? lst[(it + nwaves) / parts] : it + nwaves];
        const PoseVox V = cur.vf;
        if (lane == 0) {      // this pair's transform for the exact search (the slot's previous user is complete by now)
            double *P = recs + 16 * slot;
            for (int i = 0; i < 9; i++) P[i] = cur.R[i];
            for (int i = 0; i < 3; i++) { P[9 + i] = cur.ph[i]; P[12 + i] = cur.pl[i]; }
        }
        for (int a0 = a_begin; a0 < a_end; a0 += MAD_WAVE * POSE_BATCH) {
            uint2 word[POSE_BATCH];
            int bit[POSE_BATCH];
            if (a0 + MAD_WAVE * POSE_BATCH <= a_end) {      // wave-uniform: a full batch, in straight-line code -- the four LDS reads and transforms overlap
                unsigned idx[POSE_BATCH];
                bool in[POSE_BATCH];
#pragma unroll
                for (int u = 0; u < POSE_BATCH; u++) {
                    const float4 c = clf[a0 + u * MAD_WAVE + lane];
                    const float vx = fmaf(c.z, V.m[2], fmaf(c.y, V.m[1], fmaf(c.x, V.m[0], V.t[0])));
                    const float vy = fmaf(c.z, V.m[5], fmaf(c.y, V.m[4], fmaf(c.x, V.m[3], V.t[1])));
                    const float vz = fmaf(c.z, V.m[8], fmaf(c.y, V.m[7], fmaf(c.x, V.m[6], V.t[2])));
                    const int ix = cvt_floor(vx), iy = cvt_floor(vy), iz = cvt_floor(vz);
                    bit[u] = iz & 31;
                    in[u] = ((unsigned)ix < (unsigned)B.dim[0]) & ((unsigned)iy < (unsigned)B.dim[1]) & ((unsigned)iz < (unsigned)B.dim[2]);
                    idx[u] = mad_u24(mad_u24((unsigned)ix, (unsigned)B.dim[1], (unsigned)iy), (unsigned)B.wz, (unsigned)(iz >> 5));
                }
#pragma unroll
                for (int u = 0; u < POSE_BATCH; u++) {
                    word[u] = make_uint2(0u, 0u);
                    if (in[u]) word[u] = ((const uint2 *)bits)[idx[u]];      // a point outside the bitmap box pays nothing
                }
            } else {
#pragma unroll
                for (int u = 0; u < POSE_BATCH; u++) {      // the bitmap words of 4 x 64 points are requested before any is looked at
                    const int a = a0 + u * MAD_WAVE + lane;
                    word[u] = make_uint2(0u, 0u); bit[u] = 0;
                    if (a < a_end) {
                        const float4 c = clf[a];
                        pose_vox_fetch(V, c.x, c.y, c.z, B, bits, &word[u], &bit[u]);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < POSE_BATCH; u++) {
                if (a0 + u * MAD_WAVE >= a_end) break;      // wave-uniform
                const bool certain = (word[u].y >> bit[u]) & 1u;      // inner plane: a lo point within dist for certain
                cnt_cur += __popcll(__ballot(certain));
                const bool sv = ((word[u].x >> bit[u]) & 1u) && !certain;
                const unsigned long long bal = __ballot(sv);
                if (sv)
                    ring[(head + count + __popcll(bal & lanemask_lt())) & (POSE_RING - 1)] =
                        (unsigned short)((slot << 15) | (a0 + u * MAD_WAVE + lane));
                count += __popcll(bal);
                if (count >= MAD_WAVE) round(MAD_WAVE);      // wave-uniform
            }
        }
        if (old_left > 0) round(count);      // rare: too few survivors to reach the previous pair's leftovers; take everything
        // this pair becomes the previous one; what it left in the queue waits for company
        p_old = p; cnt_old = cnt_cur; cnt_cur = 0; old_left = count;
        if (old_left == 0 && lane == 0) emit(p, cnt_old);
        slot ^= 1;
        cur = nxt;
    }
    if (count > 0) round(count);      // completes the last pair
}


// The same search for lo clouds too large for float64 points in LDS (maps of ~512^3: 5 000+ anchors).  The lo points
// sit in LDS as float32 offsets from the grid origin (16 B each) and the hi cloud is read from global memory (a
// coalesced, cache-resident stream).  Candidates are tested in two tiers, two per iteration in float32:
// d2 < lim_in decides "within dist", d2 > lim_out decides "outside"; lim_in / lim_out bracket
// dist^2 by a margin far above the float32 error of the offsets (host: pose_device).  Only a candidate whose float32
// distance falls inside that band (about one sample in 10^4) is re-evaluated with the reference's float64 expression
// on the float64 point in global memory, so the count is the float64 count.  On clouds that fit both kernels this one
// is ~20 % slower than k_pose_lds, and ~10 x faster than the global cell list it replaces for the big ones.
template <bool HI_LDS>
__global__ __launch_bounds__(POSE_LDS_THREADS) void k_pose_lds32(const int32_t *__restrict__ status, int64_t cap_pairs,
                                                                 const PosePair *__restrict__ rec,
                                                                 const double *__restrict__ hi_cloud, const double *__restrict__ lo_sorted,
                                                                 const float4 *__restrict__ lo_sorted_f,
                                                                 const int32_t *__restrict__ cell_start,
                                                                 const unsigned short *__restrict__ cell_start16, PoseGrid G, int l_lo_cap, int l_hi_cap,
                                                                 float reach, double dd_lim, float lim_in, float lim_out, PoseBits B,
                                                                 const unsigned *__restrict__ bits, int32_t *__restrict__ counts,
                                                                 const int32_t *__restrict__ sel) {
    extern __shared__ __align__(16) unsigned char smem[];
    if (status[ST_FLAG_C] || status[ST_FLAG_PAIRS]) return;
    float4 *lpf = (float4 *)smem;                                  // sorted lo cloud, float32 offsets from G.mn
    unsigned short *cs = (unsigned short *)(lpf + l_lo_cap + 1);   // cell offsets
    unsigned short *stack = (unsigned short *)((unsigned char *)cs + pad16((size_t)(G.ncell + 1) * 2)) + (threadIdx.x >> 6) * POSE_STACK;
    const int64_t n_pairs = sel ? (int64_t)status[ST_NSEL] : min((int64_t)status[ST_NPAIRS], cap_pairs);      // as in k_pose_lds
    const int l_hi = status[ST_LHI];
    const int l_lo = cell_start[G.ncell];
    // HI_LDS: the hi cloud too, as float64 for the exact search and as float32 for the bitmap test (when it fits beside the lo cloud)
    double *cl = (double *)((unsigned char *)cs + pad16((size_t)(G.ncell + 1) * 2) + (POSE_LDS_THREADS / MAD_WAVE) * POSE_STACK * 2);
    float4 *clf = (float4 *)((unsigned char *)cl + pad16((size_t)l_hi_cap * 24));
    stage_lds(lpf, lo_sorted_f, (size_t)l_lo * 16);
    stage_lds(cs, cell_start16, (size_t)((G.ncell + 2) & ~1) * 2);
    if (HI_LDS) stage_lds(cl, hi_cloud, (size_t)l_hi * 24);
    if (threadIdx.x == 0) lpf[l_lo] = make_float4(1e30f, 1e30f, 1e30f, 0.f);      // pad: the odd partner of a run's last point
    __syncthreads();
    if (HI_LDS) {
        for (int i = threadIdx.x; i < l_hi; i += POSE_LDS_THREADS)
            clf[i] = make_float4((float)cl[3 * i], (float)cl[3 * i + 1], (float)cl[3 * i + 2], 0.f);
        __syncthreads();
    }
    const double *hc = HI_LDS ? (const double *)cl : hi_cloud;      // where the exact search reads the hi cloud
    const int lane = lane_id();
    const int64_t wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (POSE_LDS_THREADS / MAD_WAVE) + (threadIdx.x >> 6)));
    const int64_t nwaves = (int64_t)gridDim.x * (POSE_LDS_THREADS / MAD_WAVE);
    PosePair cur;
    if (wave < n_pairs) cur = rec[sel ? sel[wave] : wave];
    for (int64_t it = wave; it < n_pairs; it += nwaves) {
        const int64_t p = sel ? sel[it] : it;
        PosePair nxt;      // requested now, needed one iteration later
        if (it + nwaves < n_pairs) nxt = rec[sel ? sel[it + nwaves] : it + nwaves];
        const double *R = cur.R;
        const double ph0 = cur.ph[0], ph1 = cur.ph[1], ph2 = cur.ph[2], pl0 = cur.pl[0], pl1 = cur.pl[1], pl2 = cur.pl[2];
        const PoseVox V = cur.vf;
        auto cloud32 = [&](int a, float &cx, float &cy, float &cz) {
            if (HI_LDS) { const float4 c = clf[a]; cx = c.x; cy = c.y; cz = c.z; }
            else { cx = (float)hi_cloud[3 * a]; cy = (float)hi_cloud[3 * a + 1]; cz = (float)hi_cloud[3 * a + 2]; }
        };

        auto exact = [&](int a) -> bool {
            const double d0 = hc[3 * a] - ph0, d1 = hc[3 * a + 1] - ph1, d2 = hc[3 * a + 2] - ph2;
            const double x = (d0 * R[0] + d1 * R[1] + d2 * R[2]) + pl0;      // MaD.py:440-444
            const double y = (d0 * R[3] + d1 * R[4] + d2 * R[5]) + pl1;
            const double z = (d0 * R[6] + d1 * R[7] + d2 * R[8]) + pl2;
            // offsets from the grid origin, rounded once to float32
            const float xf = (float)(x - G.mn[0]), yf = (float)(y - G.mn[1]), zf = (float)(z - G.mn[2]);
            // cells met by a slightly larger ball (reach = dist + 0.01), in float32: a superset is harmless
            const int x0 = cvt_floor((xf - reach) * G.inv_cell_f[0]), x1 = cvt_floor((xf + reach) * G.inv_cell_f[0]);
            const int y0 = cvt_floor((yf - reach) * G.inv_cell_f[1]), y1 = cvt_floor((yf + reach) * G.inv_cell_f[1]);
            const int z0 = cvt_floor((zf - reach) * G.inv_cell_f[2]), z1 = cvt_floor((zf + reach) * G.inv_cell_f[2]);
            bool hit = false;
            if (x1 >= 0 && x0 < G.dim[0] && y1 >= 0 && y0 < G.dim[1] && z1 >= 0 && z0 < G.dim[2]) {
                const int zz0 = max(z0, 0), zz1 = min(z1, G.dim[2] - 1) + 1;
                const int xa = max(x0, 0), xb = min(x1, G.dim[0] - 1), ya = max(y0, 0), yb = min(y1, G.dim[1] - 1);
                const int c00 = __mul24(__mul24(xa, G.dim[1]) + ya, G.dim[2]), c01 = __mul24(__mul24(xa, G.dim[1]) + yb, G.dim[2]);      // <= 30 000 cells
                const int c10 = __mul24(__mul24(xb, G.dim[1]) + ya, G.dim[2]), c11 = __mul24(__mul24(xb, G.dim[1]) + yb, G.dim[2]);
                int s[4], e[4];
                s[0] = cs[c00 + zz0]; e[0] = cs[c00 + zz1];
                s[1] = cs[c01 + zz0]; e[1] = (yb != ya) ? cs[c01 + zz1] : s[1];
                s[2] = cs[c10 + zz0]; e[2] = (xb != xa) ? cs[c10 + zz1] : s[2];
                s[3] = cs[c11 + zz0]; e[3] = (xb != xa && yb != ya) ? cs[c11 + zz1] : s[3];
                // one loop over the concatenation of the four runs, two candidates (a "slot") per iteration
                const int m0 = (e[0] - s[0] + 1) >> 1, m1 = m0 + ((e[1] - s[1] + 1) >> 1), m2 = m1 + ((e[2] - s[2] + 1) >> 1),
                          m3 = m2 + ((e[3] - s[3] + 1) >> 1);
                for (int t = 0; t < m3 && !hit; t++) {
                    const int c = t < m0 ? 0 : (t < m1 ? 1 : (t < m2 ? 2 : 3));
                    const int q = (c == 0 ? s[0] + 2 * t : (c == 1 ? s[1] + 2 * (t - m0) : (c == 2 ? s[2] + 2 * (t - m1) : s[3] + 2 * (t - m2))));
                    const int qe = c == 0 ? e[0] : (c == 1 ? e[1] : (c == 2 ? e[2] : e[3]));
                    const float4 pa = lpf[q], pb = lpf[q + 1];      // q + 1 may belong to the next cell (or be the pad): masked below
                    // plain float32 on purpose: packed float32 VALU instructions are banned in this library (Makefile)
                    const float ax = pa.x - xf, ay = pa.y - yf, az = pa.z - zf, bx = pb.x - xf, by = pb.y - yf, bz = pb.z - zf;
                    const float dda = fmaf(az, az, fmaf(ay, ay, ax * ax)), ddb = fmaf(bz, bz, fmaf(by, by, bx * bx));
                    const bool vb = q + 1 < qe;
                    hit = dda < lim_in || (vb && ddb < lim_in);
                    if (!hit && ((dda <= lim_out) || (vb && ddb <= lim_out))) {
                        // inside the band: the reference's float64 expression (MaD.py:447-448)
                        if (dda <= lim_out) {
                            const double e0 = lo_sorted[3 * q] - x, e1 = lo_sorted[3 * q + 1] - y, e2 = lo_sorted[3 * q + 2] - z;
                            hit = (e0 * e0 + e1 * e1 + e2 * e2) < dd_lim;
                        }
                        if (!hit && vb && ddb <= lim_out) {
                            const double e0 = lo_sorted[3 * q + 3] - x, e1 = lo_sorted[3 * q + 4] - y, e2 = lo_sorted[3 * q + 5] - z;
                            hit = (e0 * e0 + e1 * e1 + e2 * e2) < dd_lim;
                        }
                    }
                }
            }
            return hit;
        };
        int cnt = pose_count_filtered(l_hi, V, B, bits, stack, cloud32, exact);
        cnt = wave_sum_i32(cnt);
        if (lane == 0) counts[p] = cnt;
        cur = nxt;
    }
}

// ---- pruning by bounds: the bitmap phase alone brackets a pair's count ------------------------------------------
//
// For one pair, the number of hi points whose voxel has the INNER bit set is a lower bound L of its match count, and L plus
// the number of points in the shell (outer bit set, inner clear) an upper bound U.  A match only reports the k best pairs
// (count descending, pair order ascending, MaD.py:480), so with T = the k-th largest L over all pairs, a pair with U < T
// cannot be among them: at least k pairs have a count >= T.  k_pose_bounds brackets every pair (the bitmap phase of
// k_pose_lds: float32, ~40 % of its instructions, and it needs neither the lo cloud nor float64 points in LDS);
// k_prune_select lists the pairs with U >= T; the exact search then runs on those alone and overwrites their entry of
// `counts`.  The others keep L <= count < T there, which the top-k selection ranks behind every listed pair exactly as
// it would rank their true counts: the k rows and their order are those of the unpruned search.
__device__ __forceinline__ int topk_threshold(const int32_t *__restrict__ hist, int nbins, int64_t k, int *wt, int *sh, int *need);
__global__ __launch_bounds__(256) void k_count_hist(const int32_t *__restrict__ counts, const int32_t *__restrict__ status, int64_t cap_pairs,
                                                    int32_t *__restrict__ hist, int nbins);

#define PB_THREADS 1024
#define PB_MAX_SETS 16      // hi clouds of up to 16 x 64 points are bracketed (the per-set lookup state lives in registers: ~7 per set)
#define PB_SAMPLE 2048          // scores sampled for the threshold between the two phases of the bounds pass
#define PB_SCORE_BINS 1024      // ... and the bins of [0, 1] they are counted in
struct PoseCoarse {
    PoseBits B;
    int n_words;
};

// NB x 64 >= l_hi: the point sets of one pair.  Per pair and wave:
//   coarse phase : every hi point is mapped into the COARSE outer-plane bitmap, which sits in LDS (one LDS lookup per point);
//                  a clear voxel proves that no lo point lies within dist -- 3 points in 4 end here.  The others are queued
//                  (ballot compaction, per-wave queue in LDS);
//   fine phase   : the queued points, now in full lanes, are mapped into the FINE two-plane bitmap in global memory (one
//                  scattered L2 request per lane: what bounded the search when every point made one) and tallied:
//                  inner bit = certainly within dist (lower bound), outer bit = possibly (upper bound).
// median of (x, 0, hi) = x clamped into [0, hi], hi >= 0 and wave-uniform: one instruction
__device__ __forceinline__ int clamp0(int x, int hi) {
    int r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "s"(hi));
    return r;
}
// a * b + c in 24-bit arithmetic with a wave-uniform b straight from its scalar register
__device__ __forceinline__ unsigned mad_u24s(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}

// Software-pipelined over pairs: the fine lookups of pair i are in flight while the coarse phase of pair i + 1 runs.
template <int NB, bool SPLIT>
__global__ __launch_bounds__(PB_THREADS) void k_pose_bounds(const int32_t *__restrict__ status, int64_t cap_pairs,
                                                            const PosePair *__restrict__ rec, const double *__restrict__ hi_cloud,
                                                            PoseBits B, const unsigned *__restrict__ bits, PoseCoarse C,
                                                            const unsigned *__restrict__ bits_c, int32_t *__restrict__ lower,
                                                            unsigned short *__restrict__ upper, int32_t *__restrict__ hist, int nbins,
                                                            const double *__restrict__ score, int phase, int64_t target_a, int64_t k_stop) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ int t_wt[PB_THREADS / MAD_WAVE + 1];
    __shared__ int t_sh[3];
    __shared__ int s_hist[PB_SCORE_BINS];
    if (status[ST_FLAG_C] || status[ST_FLAG_PAIRS]) return;
    unsigned *lb = (unsigned *)smem;                                           // coarse bitmap
    float4 *clf = (float4 *)(smem + pad16((size_t)C.n_words * 4));             // hi cloud, float32
    const int64_t n_pairs = min((int64_t)status[ST_NPAIRS], cap_pairs);
    const int l_hi = status[ST_LHI];
    // Two launches per match, best-scoring pairs first.  A pair's correlation score predicts its match count well (on the C3
    // workload the 60 largest counts all lie within the best-scoring 15 % of the pairs, all but a few within the best 1 %), so
    // phase 1 brackets the pairs whose score lies in the top ~target_a (a threshold taken from PB_SAMPLE scores at a fixed
    // stride, which every workgroup derives for itself) and phase 2 the others, with k_stop > 0: a pair is ABANDONED as soon
    // as (coarse hits so far) + (points not yet looked at) -- an upper bound of its upper bound -- falls below T_stop = the
    // k_stop-th largest lower bound already in the histogram (phase 1's): at least k_stop pairs have a count >= T_stop, so such a
    // pair cannot be among the k_stop best, and the final threshold of k_prune_select (over all pairs) can only be higher.  It
    // gets lower bound 0 and that upper bound.  Nothing depends on the prediction being right: a poor one only prunes less.
    // phase 0: all pairs in one launch, nothing abandoned.
    auto bin_of = [](double v) { return min(max((int)(v * PB_SCORE_BINS), 0), PB_SCORE_BINS - 1); };
    int b_star = 0;      // phase 1 takes the pairs whose score bin is >= b_star, phase 2 the others
    int t_stop = 0;
    if (SPLIT && phase != 0) {
        if (target_a < n_pairs) {
            for (int i = threadIdx.x; i < PB_SCORE_BINS; i += PB_THREADS) s_hist[i] = 0;
            __syncthreads();
            const int64_t n_s = min((int64_t)PB_SAMPLE, n_pairs), stride = n_pairs / n_s;
            for (int64_t i = threadIdx.x; i < n_s; i += PB_THREADS) atomicAdd(&s_hist[bin_of(score[i * stride])], 1);
            __syncthreads();
            int need;
            b_star = max(topk_threshold(s_hist, PB_SCORE_BINS, max((target_a * n_s + n_pairs - 1) / n_pairs, (int64_t)1), t_wt, t_sh, &need), 0);
            __syncthreads();
        }
        if (phase == 2 && b_star == 0) return;      // everything went through in phase 1
        // T_stop: what phase 1 left in hist[nbins + 1] (its last workgroup, below) -- NOT derived here from the histogram, into which
        // the workgroups of this very launch flush their own lower bounds as they finish: a workgroup that starts late (the lanes
        // overlap: a CU may be busy with another match) would see a higher threshold and abandon other pairs than in the run
        // before.  The top-k does not depend on it, the selected count and the launch sizes derived from it did.
        if (phase == 2 && k_stop > 0) t_stop = hist[nbins + 1];
    }
    // histogram of the lower bounds (k_prune_select takes its threshold from it): per workgroup in LDS, flushed once at the end
    int *lh = (int *)(smem + pad16((size_t)C.n_words * 4) + pad16((size_t)(l_hi + 4) * 16) + (size_t)(PB_THREADS / MAD_WAVE) * ((NB + 1) * MAD_WAVE) * 2);
    for (int i = threadIdx.x; i < nbins; i += PB_THREADS) lh[i] = 0;
    // one queue of NB x 64 point ids (+ 64 dump slots) per wave (LDS operations of a wave execute in order: the lookups of a pair have read it
    // before the next pair's filter writes it)
    unsigned short *queue = (unsigned short *)(smem + pad16((size_t)C.n_words * 4) + pad16((size_t)(l_hi + 4) * 16)) + (threadIdx.x >> 6) * ((NB + 1) * MAD_WAVE);
    stage_lds(lb, bits_c, (size_t)C.n_words * 4);
    for (int i = threadIdx.x; i < l_hi; i += PB_THREADS)
        clf[i] = make_float4((float)hi_cloud[3 * i], (float)hi_cloud[3 * i + 1], (float)hi_cloud[3 * i + 2], 0.f);
    __syncthreads();
    const int lane = lane_id();
    const int64_t wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (PB_THREADS / MAD_WAVE) + (threadIdx.x >> 6)));
    const int64_t nwaves = (int64_t)gridDim.x * (PB_THREADS / MAD_WAVE);

    // coarse phase of one pair -> number of queued points (wave-uniform).  Straight-line per set of 64 points, ~26 vector
    // instructions and no branch: the lane's points stay in registers from pair to pair (up to 8 sets; a lane beyond the cloud
    // holds NaNs, which convert to voxel 0), a coordinate outside the bitmap is clamped onto its outermost layer -- never
    // marked, the box is a whole voxel wider than any marked one on every side (pose_device) -- instead of being tested, the
    // bit comes out with one v_bfe, and a lane that does not pass writes its id to a dump slot instead of being masked off.
    constexpr bool CACHE = NB <= 8;
    float pcx[CACHE ? NB : 1], pcy[CACHE ? NB : 1], pcz[CACHE ? NB : 1];
    if (CACHE) {
#pragma unroll NB
        for (int u = 0; u < NB; u++) {
            const int a = u * MAD_WAVE + lane;
            const float4 c = clf[min(a, max(l_hi - 1, 0))];
            const float bad = __int_as_float(0x7fc00000);
            pcx[u] = a < l_hi ? c.x : bad; pcy[u] = a < l_hi ? c.y : bad; pcz[u] = a < l_hi ? c.z : bad;
        }
    }
    unsigned short *dump = queue + NB * MAD_WAVE;      // 64 entries behind the wave's queue, never read
    const int dx1 = C.B.dim[0] - 1, dy1 = C.B.dim[1] - 1, dz1 = C.B.dim[2] - 1;
    auto filter = [&](const PoseVox &V, unsigned short *q) -> int {
        int nq = 0;
        // the translation in vector registers, once per pair (a v_fma takes one scalar operand: as an addend beside a scalar
        // matrix entry it would be copied for every set)
        float t0, t1, t2;
        asm volatile("v_mov_b32 %0, %3\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %5" : "=&v"(t0), "=&v"(t1), "=&v"(t2) : "s"(V.t[0]), "s"(V.t[1]), "s"(V.t[2]));
#pragma unroll NB
        for (int u = 0; u < NB; u++) {
            if (u * MAD_WAVE >= l_hi) break;      // wave-uniform
            const int a = u * MAD_WAVE + lane;
            float cx, cy, cz;
            if (CACHE) { cx = pcx[u]; cy = pcy[u]; cz = pcz[u]; }
            else {
                const float4 c = clf[min(a, l_hi - 1)];
                const float bad = __int_as_float(0x7fc00000);
                cx = a < l_hi ? c.x : bad; cy = a < l_hi ? c.y : bad; cz = a < l_hi ? c.z : bad;
            }
            const float vx = fmaf(cz, V.m[2], fmaf(cy, V.m[1], fmaf(cx, V.m[0], t0)));
            const float vy = fmaf(cz, V.m[5], fmaf(cy, V.m[4], fmaf(cx, V.m[3], t1)));
            const float vz = fmaf(cz, V.m[8], fmaf(cy, V.m[7], fmaf(cx, V.m[6], t2)));
            const int jx = clamp0(cvt_floor(vx), dx1), jy = clamp0(cvt_floor(vy), dy1), jz = clamp0(cvt_floor(vz), dz1);
            const unsigned cw = lb[mad_u24s(mad_u24s((unsigned)jx, (unsigned)C.B.dim[1], (unsigned)jy), (unsigned)C.B.wz, (unsigned)(jz >> 5))];
            const bool pass = __builtin_amdgcn_ubfe(cw, (unsigned)jz, 1u) != 0u;      // v_bfe_u32 takes the offset from the low five bits
            const unsigned long long bal = __ballot(pass);
            // set bits below this lane: two instructions (v_mbcnt_lo / _hi) where a lane mask costs a 64-bit shift and two population counts
            const int below = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
            unsigned short *dst = pass ? q + nq + below : dump + lane;
            *dst = (unsigned short)a;
            nq += __popcll(bal);
            if (SPLIT) {
                const int bound = nq + max(l_hi - (u + 1) * MAD_WAVE, 0);      // wave-uniform
                if (bound < t_stop) return -bound - 1;
            }
        }
        return nq;
    };
    // fine phase, first half: the lookups of the queued points go out.  Coordinates are clamped onto the fine bitmap's outermost
    // layer (never marked either: same construction), so the only predicate left is "this lane holds a queued point".
    const int fx1 = B.dim[0] - 1, fy1 = B.dim[1] - 1, fz1 = B.dim[2] - 1;
    auto lookup = [&](const PoseVox &V, const unsigned short *q, int nq, uint2 (&w)[NB], int (&bit)[NB]) {
        __builtin_amdgcn_wave_barrier();      // the queue was written by this wave's own lanes
        float t0, t1, t2;
        asm volatile("v_mov_b32 %0, %3\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %5" : "=&v"(t0), "=&v"(t1), "=&v"(t2) : "s"(V.t[0]), "s"(V.t[1]), "s"(V.t[2]));
        const float m0 = V.m[0], m1 = V.m[1], m2 = V.m[2], m3 = V.m[3], m4 = V.m[4], m5 = V.m[5], m6 = V.m[6], m7 = V.m[7], m8 = V.m[8];
#pragma unroll NB
        for (int u = 0; u < NB; u++) {
            w[u] = make_uint2(0u, 0u); bit[u] = 0;
            if (u * MAD_WAVE >= nq) continue;      // wave-uniform
            const int e = u * MAD_WAVE + lane;
            const float4 c = clf[q[min(e, nq - 1)]];
            const float vx = fmaf(c.z, m2, fmaf(c.y, m1, fmaf(c.x, m0, t0)));
            const float vy = fmaf(c.z, m5, fmaf(c.y, m4, fmaf(c.x, m3, t1)));
            const float vz = fmaf(c.z, m8, fmaf(c.y, m7, fmaf(c.x, m6, t2)));
            const int ix = clamp0(cvt_floor(vx), fx1), iy = clamp0(cvt_floor(vy), fy1), iz = clamp0(cvt_floor(vz), fz1);
            bit[u] = iz;      // the bit tests use its low five bits
            const unsigned fi = mad_u24s(mad_u24s((unsigned)ix, (unsigned)B.dim[1], (unsigned)iy), (unsigned)B.wz, (unsigned)(iz >> 5));
            if (e < nq) w[u] = ((const uint2 *)bits)[fi];
        }
    };
    // second half: the inner ball lies inside the outer one, so L = the inner count (low half), U = the outer count (high half)
    auto tally = [&](const uint2 (&w)[NB], const int (&bit)[NB], int nq) -> int {
        int acc = 0;
#pragma unroll NB
        for (int u = 0; u < NB; u++) {
            if (u * MAD_WAVE >= nq) break;      // wave-uniform
            const unsigned in = __builtin_amdgcn_ubfe(w[u].y, (unsigned)bit[u], 1u), out = __builtin_amdgcn_ubfe(w[u].x | w[u].y, (unsigned)bit[u], 1u);
            acc += (int)(in + (out << 16));
        }
        return wave_sum_i32(acc);
    };

    uint2 wA[NB], wB[NB];
    int bitA[NB], bitB[NB];
    int nqA = 0, nqB = 0;      // < 0: abandoned, -(upper bound) - 1
    auto put = [&](int64_t p, int acc, int nq) {
        if (nq < 0) acc = (-nq - 1) << 16;      // lower bound 0
        if (lane == 0) { lower[p] = acc & 0xffff; upper[p] = (unsigned short)(acc >> 16); atomicAdd(&lh[min(acc & 0xffff, nbins - 1)], 1); }
    };
    // The wave's pairs are wave, wave + nwaves, ...; which of them belong to this launch's phase is found 64 at a time (one
    // strided load of their scores, one ballot), then the set bits are walked.
    int64_t chunk = 0;                 // the next 64 candidates start at index wave + chunk * nwaves
    unsigned long long todo = 0ull;    // members of the current chunk not yet handed out
    int64_t todo_base = 0;
    int64_t plain = wave;      // SPLIT = false: the pairs wave, wave + nwaves, ... in turn
    auto next_pair = [&]() -> int64_t {
        if (!SPLIT) {
            const int64_t r = plain < n_pairs ? plain : -1;
            plain += nwaves;
            return r;
        }
        while (todo == 0ull) {
            const int64_t first = wave + chunk * nwaves;
            if (first >= n_pairs) return -1;
            const int64_t mine = first + (int64_t)lane * nwaves;
            bool take = mine < n_pairs;
            if (take && phase != 0) take = (bin_of(score[mine]) >= b_star) == (phase == 1);
            todo = __ballot(take);
            todo_base = first;
            chunk += MAD_WAVE;
        }
        const int j = __builtin_ctzll(todo);
        todo &= todo - 1ull;
        return todo_base + (int64_t)j * nwaves;
    };
    int64_t p = next_pair(), pb = -1, pa = -1;
    if (p >= 0) {
        nqA = filter(rec[p].vc, queue);
        lookup(rec[p].vf, queue, max(nqA, 0), wA, bitA);
    }
    while (p >= 0) {      // two pairs per trip (A, then B): the buffers swap roles without register copies
        pb = next_pair();
        if (pb >= 0) nqB = filter(rec[pb].vc, queue);      // while the lookups of pair p are in flight
        put(p, tally(wA, bitA, max(nqA, 0)), nqA);
        if (pb < 0) break;
        lookup(rec[pb].vf, queue, max(nqB, 0), wB, bitB);
        pa = next_pair();
        if (pa >= 0) nqA = filter(rec[pa].vc, queue);
        put(pb, tally(wB, bitB, max(nqB, 0)), nqB);
        if (pa < 0) break;
        lookup(rec[pa].vf, queue, max(nqA, 0), wA, bitA);
        p = pa;
    }
    __syncthreads();
    const bool leave_t = SPLIT && phase == 1 && k_stop > 0;
    int came_back = 0;
    for (int i = threadIdx.x; i < nbins; i += PB_THREADS)
        if (lh[i]) {
            if (leave_t) came_back += atomicAdd(&hist[i], lh[i]);      // returning: the add has been performed when the value is here
            else atomicAdd(&hist[i], lh[i]);
        }
    if (leave_t) {
        // The workgroup that finishes last takes T_stop = the k_stop-th largest lower bound of phase 1 for phase 2.
        // What this hand-off relies on, and what it does not:
        //  * ordering: every thread's adds above are RETURNING atomics (the value is back when the add has been performed at
        //    the coherence point, the L2 of this agent), the barrier collects the workgroup's, and thread 0 then takes the ticket with
        //    an acq_rel read-modify-write at agent scope -- a release for this workgroup's adds, an acquire for the last workgroup --
        //    so the last one's loads of the bins (agent scope, below) see every workgroup's adds.  One ordered RMW per workgroup, no
        //    __threadfence(): an agent-scope fence in every thread writes the L2 back in every workgroup, +60 us per match.
        //  * and if a bin WERE read stale (a count too low), the threshold derived from it could only come out LOWER: T is the
        //    k_stop-th largest lower bound, monotonic in every bin.  A lower T_stop abandons fewer pairs in phase 2 -- less pruning,
        //    the same selection {U >= final T} and the same top-k ("score first, then abandon").  So the
        //    result never depended on this ordering; the number of pairs selected, and with it the sizes of later launches, did.
        //  * hist[nbins + 1] (T_stop) and hist[nbins + 2] (the ticket) lie inside the region the match's zero fill clears before
        //    every attempt (zr_hist2: n_hi_anchors + 17 words; nbins = l_hi + 1 <= n_hi_anchors + 1), asserted on the host.
        asm volatile("" :: "v"(came_back));
        __syncthreads();
        if (threadIdx.x == 0) t_sh[0] = __hip_atomic_fetch_add(&hist[nbins + 2], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
        __syncthreads();
        if (t_sh[0]) {      // (uniform)
            __syncthreads();
            for (int i = threadIdx.x; i < nbins; i += PB_THREADS) lh[i] = __hip_atomic_load(&hist[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            int need;
            const int T = max(topk_threshold(lh, nbins, k_stop, t_wt, t_sh, &need), 0);
            if (threadIdx.x == 0) hist[nbins + 1] = T;
        }
    }
}

// ---- the same bracket with the coarse map on the matrix cores ---------------------------------------------------------------
//
// The coarse phase of k_pose_bounds spends 9 of its ~26 vector instructions per 64 points on v = M c + t.  That map is a small
// matrix product, and v_mfma_f32_16x16x4_f32 does 16 x 16 of them per instruction if the operands are laid out for it:
//   A (16 x 4), row 4 q + i = row i of [M | t] of pair q of FOUR pairs (i = 3: zeros);  B (4 x 16), column n = (x, y, z, 1) of point n
//   D (16 x 16): lane l holds D[4 (l / 16) + i][l % 16], i = 0..3 = the voxel coordinates of point l % 16 under pair l / 16
// so a wave works on four pairs at once, one pair per group of 16 lanes, 16 points per instruction, and every lane ends up with one
// whole mapped point -- the rest of the phase (voxel, LDS bitmap, ballot compaction) runs on full lanes as before, minus the nine
// fused multiply-adds.  float32 with another summation order than the fmaf chain: the bitmaps' slack (0.02 A against < 1e-3 A,
// PoseBits) covers any of them, and the bracket L <= count <= U is all the selection needs (a voxel of difference moves a bound by
// one, never the top-k).  The wave's queue holds (point | pair << 10) of all four pairs; the fine phase reads each entry's map from
// the four records in LDS.  A group of four pairs is worked in two halves of the point blocks, so that the queue (one half, all
// passing: 2 NB sets) stays small and the lookups of one half fly under the coarse phase of the next.  SPLIT = false only (nothing
// is abandoned); NB <= 8.
typedef float v4f __attribute__((ext_vector_type(4)));

template <int NB>
__global__ __launch_bounds__(PB_THREADS) void k_pose_bounds_mx(const int32_t *__restrict__ status, int64_t cap_pairs, const PosePair *__restrict__ rec,
                                                               const double *__restrict__ hi_cloud, PoseBits B, const unsigned *__restrict__ bits,
                                                               PoseCoarse C, const unsigned *__restrict__ bits_c, int32_t *__restrict__ lower,
                                                               unsigned short *__restrict__ upper, int32_t *__restrict__ hist, int nbins) {
    extern __shared__ __align__(16) unsigned char smem[];
    if (status[ST_FLAG_C] || status[ST_FLAG_PAIRS]) return;
    constexpr int NBLK = NB * 4, HALF = NBLK / 2;      // blocks of 16 points; a half of them per coarse / fine turn
    constexpr int QSETS = 2 * NB;                      // the queue: one half with every point passing
    constexpr int NF = NB / 2;                         // sets of 64 queued entries whose lookups are in flight (more: looked up on the spot)
    constexpr int NWV = PB_THREADS / MAD_WAVE;
    unsigned *lb = (unsigned *)smem;                                           // coarse bitmap
    float4 *clf = (float4 *)(smem + pad16((size_t)C.n_words * 4));             // hi cloud, float32
    const int64_t n_pairs = min((int64_t)status[ST_NPAIRS], cap_pairs);
    const int l_hi = status[ST_LHI];
    unsigned char *after = smem + pad16((size_t)C.n_words * 4) + pad16((size_t)(l_hi + 4) * 16);
    const int wv = threadIdx.x >> 6;
    unsigned short *queue = (unsigned short *)after + wv * ((QSETS + 1) * MAD_WAVE);
    unsigned short *dump = queue + QSETS * MAD_WAVE;      // 64 entries behind the wave's queue, never read
    float *wrec = (float *)(after + (size_t)NWV * (QSETS + 1) * MAD_WAVE * 2) + wv * 48;      // the fine maps of the wave's four pairs
    int *lh = (int *)(after + (size_t)NWV * (QSETS + 1) * MAD_WAVE * 2 + (size_t)NWV * 192);
    for (int i = threadIdx.x; i < nbins; i += PB_THREADS) lh[i] = 0;
    stage_lds(lb, bits_c, (size_t)C.n_words * 4);
    for (int i = threadIdx.x; i < l_hi; i += PB_THREADS)
        clf[i] = make_float4((float)hi_cloud[3 * i], (float)hi_cloud[3 * i + 1], (float)hi_cloud[3 * i + 2], 0.f);
    __syncthreads();
    const int lane = lane_id(), g = lane >> 4, c16 = lane & 15;
    const int64_t wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * NWV + wv));
    const int64_t nwaves = (int64_t)gridDim.x * NWV;

    // B operands: component g of point 16 j + c16 (g = 3: the 1 that takes the translation); a lane beyond the cloud holds a NaN,
    // which maps to NaN and converts to voxel 0 -- on the bitmap's outermost layer, never marked
    float bp[NBLK];
#pragma unroll
    for (int j = 0; j < NBLK; j++) {
        const int a = 16 * j + c16;
        const float4 c = clf[min(a, max(l_hi - 1, 0))];
        const float v = g == 0 ? c.x : (g == 1 ? c.y : (g == 2 ? c.z : 1.0f));
        bp[j] = a < l_hi ? v : __int_as_float(0x7fc00000);
    }
    // A operand of this lane: entry (i, k = g) of [M | t] of pair q_a, i = c16 % 4, q_a = c16 / 4 -- one float of PoseVox per lane
    const int q_a = c16 >> 2, i_a = c16 & 3;
    const int a_off = g < 3 ? i_a * 3 + g : 9 + i_a;      // PoseVox: m[9] (row-major), t[3]
    auto pair_of = [&](int64_t grp, int q) -> int64_t { return wave + (4 * grp + q) * nwaves; };
    // (both loads are unconditional, at a clamped pair: a load inside a branch makes the compiler lose count of what is in flight and
    // wait for everything -- the lookups of the previous half included -- at the next use of any loaded value)
    auto load_a = [&](int64_t grp) -> float {
        const int64_t p = min(pair_of(grp, q_a), n_pairs - 1);
        return ((const float *)&rec[p].vc)[i_a < 3 ? a_off : 0];
    };
    auto valid_a = [&](int64_t grp) -> bool { return pair_of(grp, q_a) < n_pairs && i_a < 3; };      // else 0: a pair past the end maps every point to voxel 0
    auto load_f = [&](int64_t grp) -> float {      // lanes 0..47: float lane % 12 of the fine map of pair lane / 12
        const int64_t p = min(pair_of(grp, min(lane / 12, 3)), n_pairs - 1);
        return ((const float *)&rec[p].vf)[lane % 12];
    };
    const int dx1 = C.B.dim[0] - 1, dy1 = C.B.dim[1] - 1, dz1 = C.B.dim[2] - 1;
    const int fx1 = B.dim[0] - 1, fy1 = B.dim[1] - 1, fz1 = B.dim[2] - 1;
    const unsigned ent0 = (unsigned)c16 | ((unsigned)g << 10);
    const v4f zero4 = {0.f, 0.f, 0.f, 0.f};

    // coarse phase of one half of a group -> number of queued entries (wave-uniform)
    auto coarse = [&](const float a_op, auto half_tag) -> int {
        constexpr int half = decltype(half_tag)::value;      // (a constant: bp[] must be indexed with constants to stay in registers)
        int nq = 0;
        // four blocks (64 points x 4 pairs) at a time in straight-line code: the four matrix instructions first, then the four chains
        // voxel -> LDS word -> bit side by side, then the queue writes
#pragma unroll
        for (int jb = 0; jb < HALF / 4; jb++) {
            const int j0 = half * HALF + 4 * jb;
            if (16 * j0 >= l_hi) break;           // wave-uniform; a block past the cloud inside a batch holds NaNs (voxel 0, never marked)
            v4f d[4];
#pragma unroll
            for (int t = 0; t < 4; t++) d[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_op, bp[half * HALF + 4 * jb + t], zero4, 0, 0, 0);
            asm volatile("" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]));      // four results in four register quads: issued back to back (left alone the compiler reuses one quad and waits four times)
            unsigned cw[4];
            int jz[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int jx = clamp0(cvt_floor(d[t][0]), dx1), jy = clamp0(cvt_floor(d[t][1]), dy1);
                jz[t] = clamp0(cvt_floor(d[t][2]), dz1);
                cw[t] = lb[mad_u24s(mad_u24s((unsigned)jx, (unsigned)C.B.dim[1], (unsigned)jy), (unsigned)C.B.wz, (unsigned)(jz[t] >> 5))];
            }
            asm volatile("" : "+v"(cw[0]), "+v"(cw[1]), "+v"(cw[2]), "+v"(cw[3]));      // (likewise: the four LDS reads go out together)
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const bool pass = __builtin_amdgcn_ubfe(cw[t], (unsigned)jz[t], 1u) != 0u;
                const unsigned long long bal = __ballot(pass);
                const int below = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
                const int idx = pass ? nq + below : QSETS * MAD_WAVE + lane;      // a lane that does not pass writes to its dump slot
                queue[idx] = (unsigned short)(ent0 + 16u * (unsigned)(j0 + t));
                nq += __popcll(bal);
            }
        }
        return nq;
    };
    // fine phase of the queue entries [64 s0, min(64 (s0 + NF), n)): the lookups go out ...
    auto lookup = [&](const int s0, const int n, uint2 (&w)[NF], int (&bit)[NF]) {
        __builtin_amdgcn_wave_barrier();      // the queue and the records were written by this wave's own lanes
        // straight-line over all NF sets, no branch: a lane without an entry reads the queue's entry 0 (whatever it holds: a point id
        // below 1 024, a pair below 4 -- LDS reads past the cloud return what lies there), looks up word 0 of the bitmap and tests its
        // bit 0 -- voxel (0, 0, 0), on the outermost layer, never marked -- so it counts nothing.  (With a branch per set the compiler
        // waits for ALL lookups in flight at the top of every set.)
#pragma unroll
        for (int u = 0; u < NF; u++) {
            const int e = (s0 + u) * MAD_WAVE + lane;
            const bool have = e < n;
            const unsigned ent = queue[have ? e : 0];
            const unsigned q = (ent >> 10) & 3u;
            const float4 c = clf[ent & 1023u];
            const float4 *mv = (const float4 *)(wrec + q * 12);
            const float4 r0 = mv[0], r1 = mv[1], r2 = mv[2];      // m0..m3 | m4..m7 | m8 t0 t1 t2
            const float vx = fmaf(c.z, r0.z, fmaf(c.y, r0.y, fmaf(c.x, r0.x, r2.y)));
            const float vy = fmaf(c.z, r1.y, fmaf(c.y, r1.x, fmaf(c.x, r0.w, r2.z)));
            const float vz = fmaf(c.z, r2.x, fmaf(c.y, r1.w, fmaf(c.x, r1.z, r2.w)));
            const int ix = clamp0(cvt_floor(vx), fx1), iy = clamp0(cvt_floor(vy), fy1), iz = clamp0(cvt_floor(vz), fz1);
            const unsigned fi = mad_u24s(mad_u24s((unsigned)ix, (unsigned)B.dim[1], (unsigned)iy), (unsigned)B.wz, (unsigned)(iz >> 5));
            bit[u] = have ? (iz & 31) | (int)(q << 5) : 0;
            w[u] = ((const uint2 *)bits)[have ? fi : 0u];
        }
    };
    // ... and are counted: per lane, the inner / outer hits of pair q in bits 8 q .. 8 q + 7 of two words (a lane sees one entry per
    // set, at most 2 x 2 NB sets per group: the fields cannot overflow)
    unsigned acc_in = 0u, acc_out = 0u;
    auto tally = [&](const uint2 (&w)[NF], const int (&bit)[NF]) {
#pragma unroll
        for (int u = 0; u < NF; u++) {
            const unsigned in = __builtin_amdgcn_ubfe(w[u].y, (unsigned)bit[u], 1u), out = __builtin_amdgcn_ubfe(w[u].x | w[u].y, (unsigned)bit[u], 1u);
            const unsigned sh = ((unsigned)bit[u] >> 2) & 0x18u;
            acc_in += in << sh;
            acc_out += out << sh;
        }
    };
    auto put = [&](const int64_t grp) {      // the group's four brackets: lane q writes pair q
        // 8-bit fields widened to 16 before the sum over the lanes: pairs 0 and 2 in one word, 1 and 3 in the other
        const unsigned i_02 = (unsigned)wave_sum_i32((int)(acc_in & 0x00ff00ffu)), i_13 = (unsigned)wave_sum_i32((int)((acc_in >> 8) & 0x00ff00ffu));
        const unsigned o_02 = (unsigned)wave_sum_i32((int)(acc_out & 0x00ff00ffu)), o_13 = (unsigned)wave_sum_i32((int)((acc_out >> 8) & 0x00ff00ffu));
        acc_in = 0u; acc_out = 0u;
        if (lane < 4) {
            const int64_t p = pair_of(grp, lane);
            const unsigned iw = (lane & 1) ? i_13 : i_02, ow = (lane & 1) ? o_13 : o_02;
            const int L = (int)((iw >> (16 * (lane >> 1))) & 0xffffu), U = (int)((ow >> (16 * (lane >> 1))) & 0xffffu);
            if (p < n_pairs) { lower[p] = L; upper[p] = (unsigned short)U; atomicAdd(&lh[min(L, nbins - 1)], 1); }
        }
    };

    uint2 wA[NF], wB[NF];
    int bitA[NF], bitB[NF];
    // entries beyond the NF sets in flight (a half where more than ~3 points in 8 pass): looked up and counted on the spot, in the
    // buffer that has just been counted
    auto overflow = [&](const int n, uint2 (&w)[NF], int (&bit)[NF]) {
        for (int s0 = NF; s0 * MAD_WAVE < n; s0 += NF) {
            lookup(s0, n, w, bit);
            tally(w, bit);
        }
    };
    // Turns: (group, half) = (0, 0), (0, 1), (1, 0), ...  Per turn: the coarse phase of this half, the count of the previous turn's
    // lookups, this half's lookups into the other buffer -- two turns per trip so that the buffers swap without copies.
    if (wave < n_pairs) {
#pragma unroll
        for (int u = 0; u < NF; u++) { wB[u] = make_uint2(0u, 0u); bitB[u] = 0; }      // "the lookups of the group before the first": nothing
        float a_op = valid_a(0) ? load_a(0) : 0.f, f_op = load_f(0);
        for (int64_t grp = 0;; grp++) {
            float a_next = load_a(grp + 1), f_next = load_f(grp + 1);      // (clamped: harmless past the end)
            // half 0 of the group -> buffer A (the previous turn's lookups, half 1 of the group before, sit in buffer B)
            const int n0 = coarse(a_op, std::integral_constant<int, 0>());
            tally(wB, bitB);
            if (grp > 0) put(grp - 1);
            else { acc_in = 0u; acc_out = 0u; }
            // the next group's operands have arrived by now (a coarse phase ago) and nothing else is in flight: taken here, before
            // the overflow loop, whose loads the compiler cannot count
            asm volatile("" : "+v"(a_next), "+v"(f_next));
            if (lane < 48) wrec[lane] = f_op;      // this group's fine maps (the lookups of the group before are out)
            overflow(n0, wB, bitB);
            lookup(0, n0, wA, bitA);
            // half 1 -> buffer B
            const int n1 = coarse(a_op, std::integral_constant<int, 1>());
            tally(wA, bitA);
            overflow(n1, wA, bitA);
            lookup(0, n1, wB, bitB);
            if (pair_of(grp + 1, 0) >= n_pairs) {      // wave-uniform
                tally(wB, bitB);
                put(grp);
                break;
            }
            a_op = valid_a(grp + 1) ? a_next : 0.f;
            f_op = f_next;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nbins; i += PB_THREADS)
        if (lh[i]) atomicAdd(&hist[i], lh[i]);
}

// T = the k-th largest lower bound (from its histogram; 0 when there are fewer than k pairs), then the pairs whose upper bound
// reaches it, in no particular order.  Every workgroup derives T for itself, as k_tie_chunks does.
__global__ __launch_bounds__(256) void k_prune_select(const int32_t *__restrict__ status_in, int64_t cap_pairs, const int32_t *__restrict__ hist,
                                                      int nbins, int64_t k, const unsigned short *__restrict__ upper, int32_t *__restrict__ sel,
                                                      int32_t *__restrict__ n_sel, int32_t *__restrict__ thr_out, int32_t *__restrict__ zero_counts) {
    __shared__ int wt[5];
    __shared__ int sh[3];
    __shared__ int s_base;
    const int64_t n = (status_in[ST_FLAG_C] || status_in[ST_FLAG_PAIRS]) ? 0 : min((int64_t)status_in[ST_NPAIRS], cap_pairs);
    if (k > n) k = n;
    int need;
    const int cstar = topk_threshold(hist, nbins, k, wt, sh, &need);
    const int T = max(cstar, 0);
    if (blockIdx.x == 0 && threadIdx.x == 0) *thr_out = T;
    for (int64_t i0 = (int64_t)blockIdx.x * 256; i0 < n; i0 += (int64_t)gridDim.x * 256) {
        const int64_t i = i0 + threadIdx.x;
        const bool take = i < n && (int)upper[i] >= T;
        int tot;
        const int pos = block_excl_scan(take ? 1 : 0, wt, &tot);      // one returning atomic per workgroup: 2 000 of them on one word cost 20 us
        if (threadIdx.x == 0) s_base = tot ? atomicAdd(n_sel, tot) : 0;
        __syncthreads();
        if (take) { sel[s_base + pos] = (int32_t)i; if (zero_counts) zero_counts[i] = 0; }
        __syncthreads();
    }
}

// ---- fallback for clouds that do not fit LDS: uniform cell list (cell = dist) in global memory -------------

struct CellGrid {
    const int32_t *start;      // ncell + 1 offsets into pts
    const double *pts;         // sorted points, xyz
    const int32_t *ids;        // sorted point -> index in the unsorted array
    const uint8_t *used;       // per point: takes part in the lo cloud (or nullptr = all)
    double mn[3];
    double cell;
    int dim[3];
};

#define POSE_THREADS 256

__global__ __launch_bounds__(POSE_THREADS) void k_pose(const int32_t *__restrict__ pair_hi, const int32_t *__restrict__ pair_lo,
                                                       const int32_t *__restrict__ status, int64_t cap_pairs,
                                                       const double *__restrict__ hi_p, const double *__restrict__ hi_R,
                                                       const double *__restrict__ lo_p, const double *__restrict__ lo_Rinv,
                                                       const int32_t *__restrict__ hi_row_anchor,
                                                       const int32_t *__restrict__ lo_row_anchor,
                                                       const double *__restrict__ hi_cloud, CellGrid G, double dd_lim,
                                                       int32_t *__restrict__ counts) {
    extern __shared__ __align__(16) unsigned char smem[];
    if (status[ST_FLAG_C] || status[ST_FLAG_PAIRS]) return;
    double *cl = (double *)smem;
    const int64_t n_pairs = min((int64_t)status[ST_NPAIRS], cap_pairs);
    const int l_hi = status[ST_LHI];
    for (int i = threadIdx.x; i < 3 * l_hi; i += POSE_THREADS) cl[i] = hi_cloud[i];
    __syncthreads();
    const int lane = lane_id();
    const int64_t wave = (int64_t)blockIdx.x * (POSE_THREADS / MAD_WAVE) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (POSE_THREADS / MAD_WAVE);
    const double inv_cell = 1.0 / G.cell;
    for (int64_t p = wave; p < n_pairs; p += nwaves) {
        const int ih = pair_hi[p], il = pair_lo[p];
        double R[9];
        mat3_mul(lo_Rinv + 9 * il, hi_R + 9 * ih, R);
        const int ah = hi_row_anchor ? hi_row_anchor[ih] : ih, al = lo_row_anchor ? lo_row_anchor[il] : il;
        const double ph0 = hi_p[3 * ah], ph1 = hi_p[3 * ah + 1], ph2 = hi_p[3 * ah + 2];
        const double pl0 = lo_p[3 * al], pl1 = lo_p[3 * al + 1], pl2 = lo_p[3 * al + 2];
        int cnt = 0;
        for (int a = lane; a < l_hi; a += MAD_WAVE) {
            const double d0 = cl[3 * a] - ph0, d1 = cl[3 * a + 1] - ph1, d2 = cl[3 * a + 2] - ph2;
            const double x = (d0 * R[0] + d1 * R[1] + d2 * R[2]) + pl0;
            const double y = (d0 * R[3] + d1 * R[4] + d2 * R[5]) + pl1;
            const double z = (d0 * R[6] + d1 * R[7] + d2 * R[8]) + pl2;
            const int cx = (int)floor((x - G.mn[0]) * inv_cell), cy = (int)floor((y - G.mn[1]) * inv_cell),
                      cz = (int)floor((z - G.mn[2]) * inv_cell);
            bool hit = false;
            if (cx >= -1 && cx <= G.dim[0] && cy >= -1 && cy <= G.dim[1] && cz >= -1 && cz <= G.dim[2]) {
                const int z0 = max(cz - 1, 0), z1 = min(cz + 1, G.dim[2] - 1);
                if (z0 <= z1) {
                    for (int ex = max(cx - 1, 0); ex <= min(cx + 1, G.dim[0] - 1) && !hit; ex++)
                        for (int ey = max(cy - 1, 0); ey <= min(cy + 1, G.dim[1] - 1) && !hit; ey++) {
                            const size_t col = ((size_t)ex * G.dim[1] + ey) * G.dim[2];
                            const int s0 = G.start[col + z0], s1 = G.start[col + z1 + 1];
                            for (int q = s0; q < s1; q++) {
                                if (G.used && !G.used[G.ids[q]]) continue;
                                const double e0 = G.pts[3 * q] - x, e1 = G.pts[3 * q + 1] - y, e2 = G.pts[3 * q + 2] - z;
                                if ((e0 * e0 + e1 * e1 + e2 * e2) < dd_lim) { hit = true; break; }
                            }
                        }
                }
            }
            cnt += hit ? 1 : 0;
        }
        cnt = wave_sum_i32(cnt);
        if (lane == 0) counts[p] = cnt;
    }
}

// What a result row (MaD.py:451) is made from; out = [n_cap x 23 rows][n_cap int64 pair ranks][ST_COUNT int32 status] when a kernel
// writes the tail as well (out may be pinned HOST memory: the k rows of a match go straight to where the host reads them).
struct ResultArgs {
    const int32_t *pair_hi, *pair_lo;
    const double *pair_score;
    const int32_t *counts;
    const double *hi_p, *hi_R;
    const int32_t *hi_meta;
    const double *lo_p, *lo_Rinv;
    const int32_t *lo_meta, *hi_row_anchor, *lo_row_anchor;
    double *out;
    int64_t n_cap;
};
__device__ __forceinline__ void result_row(const ResultArgs &R, int64_t p, int l_hi, double *__restrict__ o) {
    const int ih = R.pair_hi[p], il = R.pair_lo[p];
    const int ah = R.hi_row_anchor ? R.hi_row_anchor[ih] : ih, al = R.lo_row_anchor ? R.lo_row_anchor[il] : il;
    o[0] = R.pair_score[p];
    o[1] = 100.0 * (double)R.counts[p] / (double)l_hi;      // MaD.py:448
    o[2] = R.lo_meta[3 * il]; o[3] = R.lo_meta[3 * il + 1]; o[4] = R.lo_meta[3 * il + 2];
    o[5] = R.hi_meta[3 * ih]; o[6] = R.hi_meta[3 * ih + 1]; o[7] = R.hi_meta[3 * ih + 2];
    o[8] = R.hi_p[3 * ah]; o[9] = R.hi_p[3 * ah + 1]; o[10] = R.hi_p[3 * ah + 2];
    o[11] = R.lo_p[3 * al]; o[12] = R.lo_p[3 * al + 1]; o[13] = R.lo_p[3 * al + 2];
    mat3_mul(R.lo_Rinv + 9 * il, R.hi_R + 9 * ih, o + 14);
}

// rows of MaD.py:451 for the pairs listed in sel (or all pairs when sel == nullptr)
__global__ void k_results(const int64_t *__restrict__ sel, const int32_t *__restrict__ n_sel_ptr, int64_t n_sel_cap,
                          const int32_t *__restrict__ pair_hi, const int32_t *__restrict__ pair_lo,
                          const double *__restrict__ pair_score, const int32_t *__restrict__ counts,
                          const int32_t *__restrict__ status, const double *__restrict__ hi_p, const double *__restrict__ hi_R,
                          const int32_t *__restrict__ hi_meta, const double *__restrict__ lo_p, const double *__restrict__ lo_Rinv,
                          const int32_t *__restrict__ lo_meta, const int32_t *__restrict__ hi_row_anchor,
                          const int32_t *__restrict__ lo_row_anchor, double *__restrict__ out, int tail) {
    const bool failed = status[ST_FLAG_C] || status[ST_FLAG_PAIRS];
    const int64_t n_sel = failed ? 0 : min((int64_t)*n_sel_ptr, n_sel_cap);
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_sel; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = sel ? sel[t] : t;
        const int ih = pair_hi[p], il = pair_lo[p];
        double *o = out + MAD_RESULT_COLS * t;
        const int ah = hi_row_anchor ? hi_row_anchor[ih] : ih, al = lo_row_anchor ? lo_row_anchor[il] : il;
        o[0] = pair_score[p];
        o[1] = 100.0 * (double)counts[p] / (double)status[ST_LHI];      // MaD.py:448
        o[2] = lo_meta[3 * il]; o[3] = lo_meta[3 * il + 1]; o[4] = lo_meta[3 * il + 2];
        o[5] = hi_meta[3 * ih]; o[6] = hi_meta[3 * ih + 1]; o[7] = hi_meta[3 * ih + 2];
        o[8] = hi_p[3 * ah]; o[9] = hi_p[3 * ah + 1]; o[10] = hi_p[3 * ah + 2];
        o[11] = lo_p[3 * al]; o[12] = lo_p[3 * al + 1]; o[13] = lo_p[3 * al + 2];
        mat3_mul(lo_Rinv + 9 * il, hi_R + 9 * ih, o + 14);
    }
    if (tail) {      // [n_sel_cap x 23 rows][n_sel_cap int64 pair ranks][ST_COUNT int32 status]
        int64_t *ti = (int64_t *)(out + MAD_RESULT_COLS * n_sel_cap);
        for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_sel; t += (int64_t)gridDim.x * blockDim.x) ti[t] = sel[t];
        if (blockIdx.x == 0 && threadIdx.x < ST_COUNT) ((int32_t *)(ti + n_sel_cap))[threadIdx.x] = status[threadIdx.x];
    }
}

// ---------------------------------------------------------------------------
// top-k by (count desc, pair index asc), lengths on the device
// ---------------------------------------------------------------------------

#define TK_CHUNK 1024      // pairs per workgroup in the tie-ranking passes

// per-workgroup LDS histogram first: the counts crowd into a few bins, global atomics on them serialise
__global__ __launch_bounds__(256) void k_count_hist(const int32_t *__restrict__ counts, const int32_t *__restrict__ status,
                                                    int64_t cap_pairs, int32_t *__restrict__ hist, int nbins) {
    extern __shared__ __align__(16) unsigned char smem[];
    if (status[ST_FLAG_C] || status[ST_FLAG_PAIRS]) return;
    int *h = (int *)smem;
    const int64_t n = min((int64_t)status[ST_NPAIRS], cap_pairs);
    for (int b = threadIdx.x; b < nbins; b += 256) h[b] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        atomicAdd(&h[min(max(counts[i], 0), nbins - 1)], 1);
    __syncthreads();
    for (int b = threadIdx.x; b < nbins; b += 256)
        if (h[b]) atomicAdd(&hist[b], h[b]);
}

// Threshold count c* from the histogram: the bin, walking down from the top, where the running total reaches k.
// Every workgroup computes it for itself (a few hundred bins) -- cheaper than a launch of its own.
// -> c* (nbins when k <= 0, -1 when fewer than k pairs exist), *need = ties to take at c*
__device__ __forceinline__ int topk_threshold(const int32_t *__restrict__ hist, int nbins, int64_t k, int *wt, int *sh, int *need) {
    // sh: 3 ints of shared memory {c*, need, carry}
    if (threadIdx.x == 0) { sh[0] = k <= 0 ? nbins : -1; sh[1] = 0; sh[2] = 0; }
    __syncthreads();
    if (k > 0) {
        for (int base = nbins - 1; base >= 0; base -= (int)blockDim.x) {
            const int b = base - (int)threadIdx.x;
            const int v = b >= 0 ? hist[b] : 0;
            int tot;
            const int ex = sh[2] + block_excl_scan(v, wt, &tot);
            if (b >= 0 && ex < k && (int64_t)ex + v >= k) { sh[0] = b; sh[1] = (int)(k - ex); }
            __syncthreads();
            if (threadIdx.x == 0) sh[2] += tot;
            __syncthreads();
            if (sh[0] >= 0) break;
        }
    }
    *need = sh[1];
    return sh[0];
}

// ties per chunk of TK_CHUNK consecutive pairs; info[0] = c*, info[2] = ties to take at c*, info[3] = min(k, n)
__global__ __launch_bounds__(256) void k_tie_chunks(const int32_t *__restrict__ counts, const int32_t *__restrict__ status,
                                                    int64_t cap_pairs, const int32_t *__restrict__ hist, int nbins, int64_t k,
                                                    int32_t *__restrict__ info, int32_t *__restrict__ chunk_cnt,
                                                    int32_t *__restrict__ n_chunks) {
    __shared__ int wt[5];
    __shared__ int sh[3];
    const int64_t n = (status[ST_FLAG_C] || status[ST_FLAG_PAIRS]) ? 0 : min((int64_t)status[ST_NPAIRS], cap_pairs);
    const int64_t nch = (n + TK_CHUNK - 1) / TK_CHUNK;
    if (k > n) k = n;
    int need;
    const int cstar = topk_threshold(hist, nbins, k, wt, sh, &need);
    if (blockIdx.x == 0 && threadIdx.x == 0) { *n_chunks = (int32_t)nch; info[0] = cstar; info[2] = need; info[3] = (int32_t)k; }
    for (int64_t ch = blockIdx.x; ch < nch; ch += gridDim.x) {
        int c = 0;
        for (int t = threadIdx.x; t < TK_CHUNK; t += 256) {
            const int64_t i = ch * TK_CHUNK + t;
            c += (i < n && counts[i] == cstar) ? 1 : 0;
        }
        c = wave_sum_i32(c);
        __syncthreads();
        if (lane_id() == 0) wt[threadIdx.x >> 6] = c;
        __syncthreads();
        if (threadIdx.x == 0) chunk_cnt[ch] = wt[0] + wt[1] + wt[2] + wt[3];
    }
}

// keys = ((maxc - count) << 40) | pair index; survivors appended in any order.  A tie is taken when its rank among
// the ties (in pair order) is below info[2]; a chunk's first rank is the sum of the tie counts of the chunks before it.
__global__ __launch_bounds__(TK_CHUNK) void k_topk_select(const int32_t *__restrict__ counts, const int32_t *__restrict__ status,
                                                          int64_t cap_pairs, const int32_t *__restrict__ info,
                                                          const int32_t *__restrict__ chunk_cnt, int maxc,
                                                          unsigned long long *__restrict__ keys, int32_t *__restrict__ n_keys) {
    __shared__ int wt[TK_CHUNK / MAD_WAVE + 1];
    const int64_t n = (status[ST_FLAG_C] || status[ST_FLAG_PAIRS]) ? 0 : min((int64_t)status[ST_NPAIRS], cap_pairs);
    const int64_t nch = (n + TK_CHUNK - 1) / TK_CHUNK;
    const int cstar = info[0], need = info[2];
    for (int64_t ch = blockIdx.x; ch < nch; ch += gridDim.x) {
        int before = 0;
        for (int64_t q = threadIdx.x; q < ch; q += TK_CHUNK) before += chunk_cnt[q];
        int off;
        (void)block_excl_scan(before, wt, &off);
        __syncthreads();
        const int64_t i = ch * TK_CHUNK + threadIdx.x;
        const int c = i < n ? counts[i] : -1;
        const bool tie = i < n && c == cstar;
        int tot;
        const int rank = off + block_excl_scan(tie ? 1 : 0, wt, &tot);
        const bool take = i < n && (c > cstar || (tie && rank < need));
        if (take) {
            const int o = atomicAdd(n_keys, 1);
            keys[o] = ((unsigned long long)(maxc - c) << 40) | (unsigned long long)i;
        }
        __syncthreads();
    }
}

// one workgroup: bitonic sort of up to `cap` (power of two) 64-bit keys in LDS, emit pair indices.  (Sorting from
// the last workgroup of k_topk_select to finish was tried: the agent-scope fences it needs write back the whole
// L2 of every XCD and cost 60 us.)
__global__ __launch_bounds__(1024) void k_topk_sort(const unsigned long long *__restrict__ keys,
                                                    const int32_t *__restrict__ n_keys, int cap, int64_t *__restrict__ order) {
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned long long *s = (unsigned long long *)smem;
    const int n = min(*n_keys, cap);
    for (int i = threadIdx.x; i < cap; i += 1024) s[i] = i < n ? keys[i] : ~0ull;
    __syncthreads();
    for (int k = 2; k <= cap; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < cap; i += 1024) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = s[i], b = s[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { s[i] = b; s[ixj] = a; }
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < n; i += 1024) order[i] = (int64_t)(s[i] & ((1ull << 40) - 1));
}

// The k best of the pairs a pruned pose search has LISTED (sel[0 .. status[ST_NSEL])): every pair that is not listed has a count
// strictly below the k-th best (k_prune_select), so the first k of the stable order (count descending, pair index ascending,
// MaD.py:480) over the list are those over all pairs.  One workgroup: keys (maxc - count) << 40 | pair, bitonic sort in LDS over
// the next power of two >= the list length, the first k out.  A list longer than TKS_CAP raises ST_FLAG_SEL instead.
#define TKS_CAP 8192
// R.out != nullptr: the k result rows, their pair ranks and the status words are written here as well (k_results' work for a
// match, one launch fewer).
__global__ __launch_bounds__(1024) void k_topk_selected(const int32_t *__restrict__ counts, const int32_t *__restrict__ sel,
                                                        int32_t *__restrict__ status, int64_t k, int maxc, int64_t *__restrict__ order,
                                                        ResultArgs R) {
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned long long *s = (unsigned long long *)smem;
    const bool failed = status[ST_FLAG_C] || status[ST_FLAG_PAIRS];
    const int n = failed ? 0 : status[ST_NSEL];
    const int l_hi = status[ST_LHI];
    int64_t *ti = R.out ? (int64_t *)(R.out + MAD_RESULT_COLS * R.n_cap) : nullptr;
    // the status words as the host will read them: the two this kernel sets itself are substituted, not re-read
    auto tail_status = [&](int n_keys, int flag_sel) {
        if (R.out && threadIdx.x < ST_COUNT) {
            int v = status[threadIdx.x];
            if (threadIdx.x == ST_NKEYS) v = n_keys;
            if (threadIdx.x == ST_FLAG_SEL) v = v | flag_sel;
            ((int32_t *)(ti + R.n_cap))[threadIdx.x] = v;
        }
    };
    if (n > TKS_CAP) {
        if (threadIdx.x == 0) { status[ST_FLAG_SEL] = 1; status[ST_NKEYS] = 0; }
        tail_status(0, 1);
        return;
    }
    int cap = 64;
    while (cap < n) cap <<= 1;
    for (int i = threadIdx.x; i < cap; i += 1024) {
        unsigned long long key = ~0ull;
        if (i < n) { const int p = sel[i]; key = ((unsigned long long)(maxc - counts[p]) << 40) | (unsigned long long)p; }
        s[i] = key;
    }
    __syncthreads();
    if (n <= 1024) {
        // a short list (the usual case: a few hundred pairs): every key is distinct, so a key's place in the order is the number
        // of smaller keys -- n comparisons per thread against ~50 barrier-separated passes of the sorting network
        const int n_out = (int)min((int64_t)n, k);
        if ((int)threadIdx.x < n) {
            const unsigned long long mine = s[threadIdx.x];
            int rank = 0;
            for (int j = 0; j < n; j++) rank += s[j] < mine ? 1 : 0;
            if (rank < n_out) {
                const int64_t p = (int64_t)(mine & ((1ull << 40) - 1));
                order[rank] = p;
                if (R.out) { result_row(R, p, l_hi, R.out + MAD_RESULT_COLS * rank); ti[rank] = p; }
            }
        }
        if (threadIdx.x == 0) status[ST_NKEYS] = n_out;
        tail_status(n_out, 0);
        return;
    }
    for (int kk = 2; kk <= cap; kk <<= 1)
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < cap; i += 1024) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = s[i], b = s[ixj];
                    const bool up = (i & kk) == 0;
                    if ((a > b) == up) { s[i] = b; s[ixj] = a; }
                }
            }
            __syncthreads();
        }
    const int n_out = (int)min((int64_t)n, k);
    for (int i = threadIdx.x; i < n_out; i += 1024) {
        const int64_t p = (int64_t)(s[i] & ((1ull << 40) - 1));
        order[i] = p;
        if (R.out) { result_row(R, p, l_hi, R.out + MAD_RESULT_COLS * i); ti[i] = p; }
    }
    if (threadIdx.x == 0) status[ST_NKEYS] = n_out;
    tail_status(n_out, 0);
}

// Selects the first k pairs of the (count desc, index asc) order into d_order (sorted); their number goes to
// status[ST_NKEYS].  Everything is enqueued; nothing is read back.
static int topk_device(mad_ctx *ctx, const int32_t *d_counts, int32_t *d_status, int64_t cap_pairs, int64_t k, int maxc,
                       int64_t *d_order, int32_t *hist_zeroed = nullptr) {
    if (k > 8192) return mad_fail(ctx, MAD_EINVAL, "top-k: k = %lld exceeds 8192", (long long)k);
    if (cap_pairs >= ((int64_t)1 << 31)) return mad_fail(ctx, MAD_EINVAL, "top-k: %lld pairs", (long long)cap_pairs);
    const int nbins = maxc + 1;
    if (nbins > 16384) return mad_fail(ctx, MAD_EINVAL, "top-k: %d count bins", nbins);
    const int64_t max_chunks = mad_ceil_div(cap_pairs, TK_CHUNK) + 1;
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_HIST), (size_t)(nbins + 16) * 4));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_TIE_FLAG), (size_t)(max_chunks + 2) * 4));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_SEL), (size_t)(k + 8) * 8));
    int32_t *hist = hist_zeroed ? hist_zeroed : scratch<int32_t>(ctx, S_HIST);      // nbins + 16 zeroed ints
    int32_t *info = hist + nbins;            // 4 ints, then the chunk count
    int32_t *n_chunks = info + 4;
    if (!hist_zeroed) MAD_HIP(hipMemsetAsync(hist, 0, (size_t)(nbins + 16) * 4, ctx->stream));
    int cap = 1;
    while (cap < k) cap <<= 1;
    mad_timer_begin(ctx, MAD_T_TOPK);
    const int gs = ctx->n_cu * 2;
    hipLaunchKernelGGL(k_count_hist, dim3(gs), dim3(256), (size_t)nbins * 4, ctx->stream, d_counts, d_status, cap_pairs, hist, nbins);
    hipLaunchKernelGGL(k_tie_chunks, dim3(gs), dim3(256), 0, ctx->stream, d_counts, d_status, cap_pairs, hist, nbins, k, info,
                       scratch<int32_t>(ctx, S_TIE_FLAG), n_chunks);
    hipLaunchKernelGGL(k_topk_select, dim3(ctx->n_cu), dim3(TK_CHUNK), 0, ctx->stream, d_counts, d_status, cap_pairs, info,
                       scratch<int32_t>(ctx, S_TIE_FLAG), maxc, scratch<unsigned long long>(ctx, S_SEL), d_status + ST_NKEYS);
    hipLaunchKernelGGL(k_topk_sort, dim3(1), dim3(1024), (size_t)cap * 8, ctx->stream, scratch<unsigned long long>(ctx, S_SEL),
                       d_status + ST_NKEYS, cap, d_order);
    mad_timer_end(ctx, MAD_T_TOPK);
    MAD_HIP(hipGetLastError());
    return MAD_OK;
}

// ---------------------------------------------------------------------------
// global cell list (fallback path only)
// ---------------------------------------------------------------------------

__device__ __forceinline__ int cell_of(double v, double mn, double inv_cell, int dim) {
    int c = (int)floor((v - mn) * inv_cell);
    return min(max(c, 0), dim - 1);
}

__global__ void k_cell_count(const double *__restrict__ pts, int n, double m0, double m1, double m2, double inv_cell, int d0,
                             int d1, int d2, int32_t *__restrict__ cell_cnt, int32_t *__restrict__ pt_cell) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = (cell_of(pts[3 * i], m0, inv_cell, d0) * d1 + cell_of(pts[3 * i + 1], m1, inv_cell, d1)) * d2 +
                  cell_of(pts[3 * i + 2], m2, inv_cell, d2);
    pt_cell[i] = c;
    atomicAdd(&cell_cnt[c], 1);
}

__global__ void k_cell_fill(const double *__restrict__ pts, int n, const int32_t *__restrict__ pt_cell,
                            const int32_t *__restrict__ cell_start, int32_t *__restrict__ cursor,
                            double *__restrict__ sorted, int32_t *__restrict__ ids) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = pt_cell[i];
    const int o = cell_start[c] + atomicAdd(&cursor[c], 1);
    sorted[3 * o] = pts[3 * i]; sorted[3 * o + 1] = pts[3 * i + 1]; sorted[3 * o + 2] = pts[3 * i + 2];
    ids[o] = i;
}

static int build_cells(mad_ctx *ctx, const double bb_min[3], const double bb_max[3], const double *d_pts, int n, double cell,
                       DevBuf &b_start, DevBuf &b_pts, DevBuf &b_ids, double mn_out[3], int dim_out[3]) {
    size_t ncell = 1;
    for (int d = 0; d < 3; d++) {
        dim_out[d] = (int)floor((bb_max[d] - bb_min[d]) / cell) + 1;
        if (dim_out[d] < 1) dim_out[d] = 1;
        mn_out[d] = bb_min[d];
        ncell *= (size_t)dim_out[d];
    }
    if (ncell > ((size_t)1 << 28)) return mad_fail(ctx, MAD_EINVAL, "cell list of %zu cells is too large", ncell);
    MAD_TRY(mad_reserve(ctx, b_start, (ncell + 1) * 4));
    MAD_TRY(mad_reserve(ctx, b_pts, (size_t)(n > 0 ? n : 1) * 24));
    MAD_TRY(mad_reserve(ctx, b_ids, (size_t)(n > 0 ? n : 1) * 4));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_TMP_C), (ncell + 1) * 4));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_TMP_D), (size_t)(n > 0 ? n : 1) * 4));
    int32_t *cnt = scratch<int32_t>(ctx, S_TMP_C);
    int32_t *pt_cell = scratch<int32_t>(ctx, S_TMP_D);
    MAD_HIP(hipMemsetAsync(cnt, 0, (ncell + 1) * 4, ctx->stream));
    if (n > 0) {
        const unsigned nb = (unsigned)mad_ceil_div(n, 256);
        hipLaunchKernelGGL(k_cell_count, dim3(nb), dim3(256), 0, ctx->stream, d_pts, n, bb_min[0], bb_min[1], bb_min[2], 1.0 / cell,
                           dim_out[0], dim_out[1], dim_out[2], cnt, pt_cell);
        MAD_TRY(mad_scan_i32(ctx, cnt, (int32_t *)b_start.p, (int64_t)ncell));
        MAD_HIP(hipMemsetAsync(cnt, 0, (ncell + 1) * 4, ctx->stream));
        hipLaunchKernelGGL(k_cell_fill, dim3(nb), dim3(256), 0, ctx->stream, d_pts, n, pt_cell, (const int32_t *)b_start.p,
                           cnt, (double *)b_pts.p, (int32_t *)b_ids.p);
    } else {
        MAD_HIP(hipMemsetAsync(b_start.p, 0, (ncell + 1) * 4, ctx->stream));
    }
    MAD_HIP(hipGetLastError());
    return MAD_OK;
}

int mad_build_cells(mad_ctx *ctx, mad_set *set, double cell) {
    MAD_TRY(build_cells(ctx, set->bb_min, set->bb_max, (const double *)set->anc_subv.p, set->n_anchors, cell, set->cell_start,
                        set->cell_pts, set->cell_ids, set->cell_min, set->cell_dim));
    set->cell_size = cell;
    set->cells_ready = true;
    return MAD_OK;
}

// smallest double whose (correctly rounded) square root is >= dist: dd < limit  <=>  sqrt(dd) < dist
static double sqrt_limit(double dist) {
    double t = dist * dist;
    while (sqrt(nextafter(t, 0.0)) >= dist) t = nextafter(t, 0.0);
    while (sqrt(t) < dist) t = nextafter(t, INFINITY);
    return t;
}

// ---------------------------------------------------------------------------
// the match pipeline (shared by the stage API and the set API)
// ---------------------------------------------------------------------------

struct Side {      // one side of a match, all device pointers
    const int8_t *dsc8;
    const double *norm;
    const double *R;            // per row
    const double *Rinv;         // per row
    const int32_t *meta;        // per row x 3
    const int32_t *row_anchor;  // row -> entry of `p` (nullptr: identity)
    const int32_t *anc_canon;   // anchor -> first anchor with the same coordinates (nullptr: identity)
    const double *p;            // sub-voxel coordinates (per anchor, or per row when row_anchor == nullptr)
    const int32_t *n_rows;      // device
    int64_t cap_rows;           // upper bound of *n_rows
};

// correlate + compact: fills S_PAIR_*; status[ST_NPAIRS]; used flags (nullable)
static int gemm2_launch(mad_ctx *ctx, const GemmBatch &G) {
    static bool attr = false;
    if (!attr) {
        MAD_HIP(hipFuncSetAttribute((const void *)k_corr_gemm2, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS));
        attr = true;
    }
    static const int per_cu = getenv("MAD_GEMM_WG_PER_CU") ? std::max(1, std::min(2, atoi(getenv("MAD_GEMM_WG_PER_CU")))) : 2;      // probe: 1 leaves half of every CU to the kernels of other lanes
    hipLaunchKernelGGL(k_corr_gemm2, dim3(ctx->n_cu * per_cu), dim3(GEMM_THREADS), G2_LDS, ctx->stream, G);      // persistent: two per CU
    return MAD_OK;
}

// bytes of the candidate flags of a score matrix of cap_c entries (rows x pitch <= cap_c / 32 + 3 rows' worth: twice the entries / 32 covers it)
static size_t cflag_bytes(int64_t cap_c) { return (size_t)cap_c / 16 + 64; }

// The correlation stage in three pieces, so that the GEMMs of several matches can go out as ONE launch (mad_match_topk_many):
// the scratch of this lane sized and the GEMM's arguments filled in; the GEMM; threshold + ordered compaction of the pairs.
static int correlate_reserve(mad_ctx *ctx, const Side &hi, const Side &lo, int D, double cc, int32_t *d_status, int64_t cap_c,
                             int64_t cap_pairs, GemmJob *job) {
    if (D % GEMM_BK) return mad_fail(ctx, MAD_EINVAL, "correlate: D = %d is not a multiple of %d", D, GEMM_BK);
    if (hi.cap_rows > 65000) return mad_fail(ctx, MAD_EINVAL, "correlate: %lld hi rows exceed the single-launch scan", (long long)hi.cap_rows);
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_CMAT), (size_t)cap_c * 4));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_ROWCNT), (size_t)(hi.cap_rows + 2) * 4));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_ROWOFF), (size_t)(hi.cap_rows + 2) * 4));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_PAIR_HI), (size_t)cap_pairs * 4));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_PAIR_LO), (size_t)cap_pairs * 4));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_PAIR_SCORE), (size_t)cap_pairs * 8));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_CMASK), (size_t)cap_c / 8 + 64));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_CFLAG), cflag_bytes(cap_c)));
    *job = GemmJob{hi.dsc8, lo.dsc8, scratch<int32_t>(ctx, S_CMAT), hi.n_rows, lo.n_rows, cap_c, d_status, hi.norm, lo.norm,
                   scratch<uint32_t>(ctx, S_CMASK), scratch<uint8_t>(ctx, S_CFLAG)};
    return MAD_OK;
}

static int correlate_gemm(mad_ctx *ctx, int n_jobs, const GemmJob *jobs, int D, double cc) {
    mad_timer_begin(ctx, MAD_T_CORRELATE);
    for (int j0 = 0; j0 < n_jobs; j0 += MAD_BATCH_MAX) {
        GemmBatch G;
        G.n_jobs = std::min(n_jobs - j0, MAD_BATCH_MAX); G.K = D; G.cc = cc;
        static const bool no_split = getenv("MAD_GEMM_NO_SPLIT") != nullptr;      // diagnostic switch: whole tiles only, as in round 3
        G.split_tail = no_split ? 0 : 1;
        for (int j = 0; j < G.n_jobs; j++) G.job[j] = jobs[j0 + j];
        MAD_TRY(gemm2_launch(ctx, G));
    }
    mad_timer_end(ctx, MAD_T_CORRELATE);
    MAD_HIP(hipGetLastError());
    return MAD_OK;
}

static int correlate_pairs(mad_ctx *ctx, const Side &hi, const Side &lo, double cc, int32_t *d_status, int64_t cap_pairs, uint8_t *d_used_hi,
                           uint8_t *d_used_lo) {
    int32_t *C = scratch<int32_t>(ctx, S_CMAT);
    uint32_t *mask = scratch<uint32_t>(ctx, S_CMASK);
    mad_timer_begin(ctx, MAD_T_PAIRS);
    const uint8_t *cflag = scratch<uint8_t>(ctx, S_CFLAG);
    hipLaunchKernelGGL(k_pair_count, dim3(ctx->n_cu * 4), dim3(256), 0, ctx->stream, C, mask, cflag, hi.n_rows, lo.n_rows, hi.norm, lo.norm, cc,
                       scratch<int32_t>(ctx, S_ROWCNT), d_status);
    hipLaunchKernelGGL(k_pair_emit2, dim3(ctx->n_cu * 4), dim3(256), 0, ctx->stream, C, mask, cflag, hi.n_rows, lo.n_rows, hi.norm, lo.norm,
                       scratch<int32_t>(ctx, S_ROWCNT), cap_pairs, scratch<int32_t>(ctx, S_PAIR_HI), scratch<int32_t>(ctx, S_PAIR_LO),
                       scratch<double>(ctx, S_PAIR_SCORE), hi.row_anchor, lo.row_anchor, hi.anc_canon, lo.anc_canon, d_used_hi, d_used_lo,
                       d_status);
    mad_timer_end(ctx, MAD_T_PAIRS);
    MAD_HIP(hipGetLastError());
    return MAD_OK;
}

static int correlate_device(mad_ctx *ctx, const Side &hi, const Side &lo, int D, double cc, int32_t *d_status, int64_t cap_c,
                            int64_t cap_pairs, uint8_t *d_used_hi, uint8_t *d_used_lo) {
    GemmJob job;
    MAD_TRY(correlate_reserve(ctx, hi, lo, D, cc, d_status, cap_c, cap_pairs, &job));
    mad_zero_words(ctx, job.cflag, cflag_bytes(cap_c));
    MAD_TRY(correlate_gemm(ctx, 1, &job, D, cc));
    return correlate_pairs(ctx, hi, lo, cc, d_status, cap_pairs, d_used_hi, d_used_lo);
}

// Workgroups of the persistent pose search per CU.  Two fill a CU completely (16 waves x 64 registers per SIMD, 130 KB of
// LDS) and run the kernel 4 % faster on its own; one leaves half of the registers and 95 KB of LDS to the kernels of the
// other lanes (the orient / describe launches of the next batch are latency-bound and fill the issue slots the pose search
// leaves), which is worth 3-4 % of the whole step.  MAD_POSE_WGS=2 restores the former.
static int pose_wgs_per_cu() {
    static const int v = getenv("MAD_POSE_WGS") ? atoi(getenv("MAD_POSE_WGS")) : 1;
    return v == 2 ? 2 : 1;
}

// Everything about the pose stage of one match that follows from the host's knowledge alone (cloud capacities, the lo cloud's
// bounding box, dist): the search grid, the bitmaps, the LDS budgets, which kernels run.  Computed before the match's first launch,
// so that the zero fill of the bitmaps can travel with the zero fill of the match's status words.
struct PosePlan {
    PoseGrid G;
    PoseBits B;                      // fine bitmap, two planes
    PoseCoarse PC;                   // coarse outer plane of the pruning pass
    double bits_rad, bits_rad_in, rad_c_out, dd_lim, reach;
    size_t n_words, fine_bytes, coarse_bytes, lds, lds32, lds32_hi;
    float lim_in, lim_out;
    bool fits64, fits32, hi_in_lds, lds_path, prune;
    int l_hi_max, n_cloud;
};

static void pose_plan(const mad_ctx *ctx, int l_hi_max, int n_cloud, const double bb_min[3], const double bb_max[3], bool have_fallback,
                      double dist, int64_t prune_k, bool have_hist2, PosePlan *Q) {
    PosePlan &P = *Q;
    P.l_hi_max = l_hi_max; P.n_cloud = n_cloud;
    P.dd_lim = sqrt_limit(dist);
    P.reach = dist + 0.01;
    const double reach = P.reach;
    PoseGrid &G = P.G;
    G.ncell = 1;
    for (int d = 0; d < 3; d++) {
        const double ext = bb_max[d] - bb_min[d];
        double cell = 2.0 * reach + 0.05;      // > 2 * reach: the enlarged ball meets at most 2 cells per axis
        if (ext / cell > 24.0) cell = ext / 24.0;
        G.mn[d] = bb_min[d];
        G.inv_cell[d] = 1.0 / cell;
        G.inv_cell_f[d] = (float)(1.0 / cell);
        G.dim[d] = (int)floor(ext / cell) + 1;
        if (G.dim[d] < 1) G.dim[d] = 1;
        G.ncell *= G.dim[d];
    }
    const size_t stacks = (size_t)(POSE_LDS_THREADS / MAD_WAVE) * POSE_WAVE_LDS;      // k_pose_lds: per-wave queue + two pair records
    const size_t stacks32 = (size_t)(POSE_LDS_THREADS / MAD_WAVE) * POSE_STACK * 2;   // k_pose_lds32: per-wave survivor stack only
    P.lds32 = (size_t)(n_cloud + 1) * 16 + pad16((size_t)(G.ncell + 1) * 2) + stacks32 + 16;
    P.lds32_hi = P.lds32 + pad16((size_t)l_hi_max * 24) + (size_t)l_hi_max * 16;      // with the hi cloud in LDS as well
    P.hi_in_lds = P.lds32_hi <= 150 * 1024;
    // occupancy bitmap of the lo cloud (k_pose_bits), in global memory: voxel edge MAD_POSE_VOXEL (default 0.8 A: 0.6-0.8 measure the same, 1.0 is 3 % slower) unless that
    // needs more than 16 MB.  It was tried in LDS too: there it has to be coarser (1.7 A beside the C3 clouds), lets 74
    // instead of 57 points per pair through, and the second round of the exact search that this costs outweighs the
    // cheaper lookup (0.77 against 0.66 ms per C3 step).
    const size_t lds_base = pad16((size_t)l_hi_max * 24) + pad16((size_t)n_cloud * 24) + (size_t)l_hi_max * 16 + pad16((size_t)(G.ncell + 1) * 2) +
                            stacks + (size_t)POSE_OWN_CAP * 4 + 16;      // the regions of k_pose_lds (the last: its own selection list)
    const bool base64 = lds_base <= 150 * 1024;
    const size_t bits_budget = (size_t)16 << 20;
    static const double h0 = getenv("MAD_POSE_VOXEL") ? atof(getenv("MAD_POSE_VOXEL")) : 0.8;
    PoseBits &B = P.B;
    const double slack = 0.02;
    for (B.h = (h0 >= 0.25 && h0 <= 8.0) ? h0 : 0.8;; B.h *= 1.08) {
        const double guard = dist + B.h * 0.8660254037844387 + slack + B.h;
        for (int d = 0; d < 3; d++) {
            B.mn[d] = bb_min[d] - guard;
            B.dim[d] = (int)ceil((bb_max[d] - bb_min[d] + 2.0 * guard) / B.h) + 1;
        }
        B.wz = (B.dim[2] + 31) / 32;
        P.n_words = 2 * (size_t)B.dim[0] * B.dim[1] * B.wz;      // two planes
        if (P.n_words * 4 <= bits_budget) break;
    }
    P.lds = lds_base;
    P.bits_rad = dist + B.h * 0.8660254037844387 + slack;
    P.bits_rad_in = dist - B.h * 0.8660254037844387 - slack;
    // The coarse outer-plane bitmap of the pruning pass (k_pose_bounds keeps it in LDS beside the float32 hi cloud): the finest voxel
    // with which it fits.  A coarser voxel lets more points through to the global lookup, it never changes a count.
    {
        PoseBits &Bc = P.PC.B;
        // beside the bitmap: the float32 hi cloud and one queue of 2-byte point ids per wave
        const int nb_sets = (l_hi_max + MAD_WAVE - 1) / MAD_WAVE;
        // (clouds of up to 8 sets go through k_pose_bounds_mx: a queue of 2 nbv sets + the dump slots and four fine maps per wave)
        const size_t q_sets = nb_sets <= 8 ? (size_t)2 * ((nb_sets + 1) & ~1) + 1 : (size_t)((nb_sets + 1) & ~1) + 5;      // else: up to 4 sets of rounding (nbv) + the dump slots
        const size_t budget = (size_t)150 * 1024 - pad16((size_t)(l_hi_max + 4) * 16) - (size_t)(PB_THREADS / MAD_WAVE) * (q_sets * MAD_WAVE * 2 + 192) - pad16((size_t)(l_hi_max + 1) * 4) - 64;
        size_t n_words_c = 0;
        for (Bc.h = std::max(1.2, B.h);; Bc.h *= 1.05) {
            const double guard = dist + Bc.h * 0.8660254037844387 + slack + Bc.h;
            for (int d = 0; d < 3; d++) {
                Bc.mn[d] = bb_min[d] - guard;
                Bc.dim[d] = (int)ceil((bb_max[d] - bb_min[d] + 2.0 * guard) / Bc.h) + 1;
            }
            Bc.wz = (Bc.dim[2] + 31) / 32;
            n_words_c = (size_t)Bc.dim[0] * Bc.dim[1] * Bc.wz;
            if (n_words_c * 4 <= budget || Bc.h > 16.0) break;
        }
        P.PC.n_words = (int)n_words_c;
        P.rad_c_out = dist + Bc.h * 0.8660254037844387 + slack;      // float32 voxel coordinates, as for the fine bitmap: same slack
    }
    // float32 tier of k_pose_lds32: offsets from the grid origin are below M = extent + reach, each rounded once (error
    // <= ulp(M) / 2); a squared distance near dist^2 is then off by < 2 sqrt(3) (dist + 1) ulp(M) plus ~1e-5 of float32
    // arithmetic.  The band is 4 x that bound.
    double M = reach;
    for (int d = 0; d < 3; d++) M = std::max(M, bb_max[d] - bb_min[d] + 2.0 * reach);
    const double ulpM = ldexp(1.0, (int)ceil(log2(M)) - 23);
    const double band = 4.0 * (2.0 * sqrt(3.0) * (dist + 1.0) * ulpM + 1e-5 * dist * dist);
    P.lim_in = nextafterf((float)(dist * dist - band), 0.f);
    P.lim_out = nextafterf((float)(dist * dist + band), 1e30f);
    // queue entries are 15 (k_pose_lds) / 16 (k_pose_lds32) bits of hi-cloud index
    P.fits64 = base64 && l_hi_max < 32768;
    P.fits32 = P.lds32 <= 150 * 1024 && P.lim_in > 0.f && l_hi_max < 65536;
    P.lds_path = !have_fallback && (P.fits64 || P.fits32) && n_cloud < 65535 && G.ncell <= 30000;
    static const bool no_prune = getenv("MAD_NO_PRUNE") != nullptr;      // diagnostic switch
    P.prune = P.lds_path && prune_k > 0 && have_hist2 && P.bits_rad_in > 0.5 && P.PC.B.h <= 16.0 && l_hi_max <= PB_MAX_SETS * MAD_WAVE && !no_prune;
    P.fine_bytes = P.lds_path ? pad16(P.n_words * 4) : 0;
    P.coarse_bytes = P.prune ? pad16((size_t)P.PC.n_words * 4) : 0;
    (void)ctx;
}

// Scores the pairs in S_PAIR_* into S_COUNTS.  The lo cloud is the set of points `d_cloud[0..n_cloud)` whose flag in
// `d_cloud_used` is set (all when nullptr); `fallback` (cell = dist, built over the same points) is used when the clouds
// do not fit LDS.
// prune_k > 0 (with hist2 = l_hi_max + 1 zeroed ints, and status[ST_NSEL] zero): only the k best pairs will be asked for, so the
// exact search runs on the pairs the bounds cannot exclude (k_pose_bounds, k_prune_select) and S_COUNTS holds lower bounds for
// the rest.  *pruned (nullable) tells whether that happened (it needs the LDS path and both bitmap planes).
// pre (nullable): the plan of this very call, made by the caller, who has also reserved S_PG_BITS and enqueued its zero fill
// (match_enqueue_head: one fill for status words and bitmaps).  fused: the fewest launches -- k_pose_setup, and the exact search
// selecting its own pairs when a previous match of this lane has told how many to expect (*own_sel reports that).
static int pose_device(mad_ctx *ctx, const Side &hi, const Side &lo, int32_t *d_status, int64_t cap_pairs,
                       const double *d_hi_cloud, int l_hi_max, const double *d_cloud, int n_cloud, const uint8_t *d_cloud_used,
                       const double bb_min[3], const double bb_max[3], const CellGrid *fallback, double dist, int64_t prune_k = 0,
                       int32_t *hist2 = nullptr, bool *pruned = nullptr, const CloudJob *job = nullptr, const PosePlan *pre = nullptr,
                       bool fused = false, bool *own_sel = nullptr) {
    if (pruned) *pruned = false;
    if (own_sel) *own_sel = false;
    const CloudJob no_job = {nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr};
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_COUNTS), (size_t)cap_pairs * 4));
    PosePlan local;
    if (!pre) pose_plan(ctx, l_hi_max, n_cloud, bb_min, bb_max, fallback != nullptr, dist, prune_k, hist2 != nullptr, &local);
    const PosePlan &P = pre ? *pre : local;
    const PoseGrid &G = P.G;
    const PoseBits &B = P.B;
    const PoseCoarse &PC = P.PC;
    const double dd_lim = P.dd_lim, reach = P.reach;
    const bool fits64 = P.fits64;
    if (P.lds_path) {
        MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_PG_START), (size_t)(G.ncell + 2) * 4 + pad16((size_t)(G.ncell + 2) * 2) + 16));
        unsigned short *d_start16 = (unsigned short *)(scratch<char>(ctx, S_PG_START) + pad16((size_t)(G.ncell + 2) * 4));
        MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_PG_PTS), (size_t)(n_cloud + 2) * 24));
        if (!fits64) MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_PG_PTSF), (size_t)(n_cloud + 2) * 16));
        static bool attr_set = false;
        if (!attr_set) {
            MAD_HIP(hipFuncSetAttribute((const void *)k_pose_lds32<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            MAD_HIP(hipFuncSetAttribute((const void *)k_pose_lds32<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            MAD_HIP(hipFuncSetAttribute((const void *)k_pose_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));      // (+ its static LDS)
            MAD_HIP(hipFuncSetAttribute((const void *)k_pose_grid_build, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            MAD_HIP(hipFuncSetAttribute((const void *)k_pose_setup, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            attr_set = true;
        }
        const bool prune = P.prune;
        static const bool dbg_prune = getenv("MAD_DEBUG_PRUNE") != nullptr;
        if (dbg_prune)
            fprintf(stderr, "pose_device: prune=%d prune_k=%lld hist2=%p rad_in=%g coarse h=%g l_hi_max=%d fits64=%d fits32=%d fine h=%g n_cloud=%d\n", (int)prune,
                    (long long)prune_k, (void *)hist2, P.bits_rad_in, PC.B.h, l_hi_max, (int)fits64, (int)P.fits32, B.h, n_cloud);
        // fine bitmap, then (when pruning) the coarse one, in one buffer: one zero fill, one launch marks all planes
        const size_t fine_bytes = P.fine_bytes, coarse_bytes = P.coarse_bytes;
        if (!pre) {
            MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_PG_BITS), fine_bytes + coarse_bytes + 16));
            mad_zero_words(ctx, scratch<unsigned>(ctx, S_PG_BITS), fine_bytes + coarse_bytes);
        }
        unsigned *d_bits = scratch<unsigned>(ctx, S_PG_BITS);
        unsigned *d_bits_c = (unsigned *)(scratch<char>(ctx, S_PG_BITS) + fine_bytes);
        PoseBitsJobs J;
        int n_jobs = 0;
        auto bjob = [&](const PoseBits &b, double rad, int plane, int planes, unsigned *bits) {
            J.B[n_jobs] = b; J.rad[n_jobs] = rad; J.plane[n_jobs] = plane; J.planes[n_jobs] = planes; J.bits[n_jobs] = bits; n_jobs++;
        };
        bjob(B, P.bits_rad, 0, 2, d_bits);
        if (P.bits_rad_in > 0.5) bjob(B, P.bits_rad_in, 1, 2, d_bits);
        if (prune) bjob(PC.B, P.rad_c_out, 0, 1, d_bits_c);
        for (int j = n_jobs; j < 4; j++) { J.B[j] = B; J.rad[j] = 0; J.plane[j] = 0; J.planes[j] = 2; J.bits[j] = nullptr; }
        MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_PG_PAIRS), (size_t)cap_pairs * sizeof(PosePair)));
        PosePair *d_rec = scratch<PosePair>(ctx, S_PG_PAIRS);
        if (fused) {
            PoseSetup S;
            S.pts = d_cloud; S.used = d_cloud_used; S.n = n_cloud; S.G = G;
            S.cell_start = scratch<int32_t>(ctx, S_PG_START); S.cell_start16 = d_start16; S.sorted = scratch<double>(ctx, S_PG_PTS);
            S.sorted_f = fits64 ? (float4 *)nullptr : scratch<float4>(ctx, S_PG_PTSF); S.n_used = d_status + ST_LLO;
            S.J = job ? *job : no_job;
            S.bits = J; S.n_bit_jobs = n_jobs; S.n_bits_wgs = n_jobs * 64;
            S.pair_hi = scratch<int32_t>(ctx, S_PAIR_HI); S.pair_lo = scratch<int32_t>(ctx, S_PAIR_LO); S.status = d_status; S.cap_pairs = cap_pairs;
            S.hi_p = hi.p; S.hi_R = hi.R; S.lo_p = lo.p; S.lo_Rinv = lo.Rinv; S.hi_row_anchor = hi.row_anchor; S.lo_row_anchor = lo.row_anchor;
            S.Bf = B; S.Bc = PC.B; S.rec = d_rec;
            const int prep_wgs = std::max(ctx->n_cu * 2 - 2 - S.n_bits_wgs, ctx->n_cu / 2);
            S.roles = 15;
            hipLaunchKernelGGL(k_pose_setup, dim3((unsigned)(2 + S.n_bits_wgs + prep_wgs)), dim3(1024), (size_t)G.ncell * 4, ctx->stream, S);
            static const bool probe_setup = getenv("MAD_PROBE_SETUP") != nullptr;      // every role once more on its own (idempotent)
            if (probe_setup)
                for (int r = 1; r < 16; r <<= 1) {
                    S.roles = r;
                    hipLaunchKernelGGL(k_pose_setup, dim3((unsigned)(2 + S.n_bits_wgs + prep_wgs)), dim3(1024), (size_t)G.ncell * 4, ctx->stream, S);
                }
        } else {
            hipLaunchKernelGGL(k_pose_grid_build, dim3(1), dim3(1024), (size_t)G.ncell * 4, ctx->stream, d_cloud, d_cloud_used, n_cloud, G,
                               scratch<int32_t>(ctx, S_PG_START), d_start16, scratch<double>(ctx, S_PG_PTS),
                               fits64 ? (float4 *)nullptr : scratch<float4>(ctx, S_PG_PTSF), d_status + ST_LLO, job ? *job : no_job);
            hipLaunchKernelGGL(k_pose_bits, dim3((unsigned)std::min(std::max(n_cloud, 1), ctx->n_cu * 4), n_jobs), dim3(256), 0, ctx->stream,
                               (const double *)scratch<double>(ctx, S_PG_PTS), (const int32_t *)scratch<int32_t>(ctx, S_PG_START), G.ncell, J);
            hipLaunchKernelGGL(k_pose_prep, dim3(ctx->n_cu * 8), dim3(256), 0, ctx->stream, scratch<int32_t>(ctx, S_PAIR_HI),
                               scratch<int32_t>(ctx, S_PAIR_LO), d_status, cap_pairs, hi.p, hi.R, lo.p, lo.Rinv, hi.row_anchor, lo.row_anchor, B, PC.B, d_rec);
        }
        ctx->last_pose_kernel = fits64 ? 0 : 1;
        mad_timer_begin(ctx, MAD_T_POSE);      // the pose stage: bounds + selection (when pruning) + the exact search
        const int32_t *d_sel = nullptr;
        PoseOwnSel own = {nullptr, nullptr, 0, 0, nullptr, 0, nullptr};
        if (prune) {
            MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_TMP_C), (size_t)cap_pairs * 2 + 64));
            MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_TMP_D), (size_t)cap_pairs * 4 + 64));
            const int nbins = l_hi_max + 1;
            static bool attr_b = false;
            if (!attr_b) {      // (155 KB: the kernel has 4.2 KB of static LDS beside the dynamic part)
                MAD_HIP(hipFuncSetAttribute((const void *)k_pose_bounds<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
                MAD_HIP(hipFuncSetAttribute((const void *)k_pose_bounds<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
                MAD_HIP(hipFuncSetAttribute((const void *)k_pose_bounds<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
                MAD_HIP(hipFuncSetAttribute((const void *)k_pose_bounds<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
                MAD_HIP(hipFuncSetAttribute((const void *)k_pose_bounds<6, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
                MAD_HIP(hipFuncSetAttribute((const void *)k_pose_bounds<6, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
                MAD_HIP(hipFuncSetAttribute((const void *)k_pose_bounds<8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
                MAD_HIP(hipFuncSetAttribute((const void *)k_pose_bounds<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
                MAD_HIP(hipFuncSetAttribute((const void *)k_pose_bounds<10, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
                MAD_HIP(hipFuncSetAttribute((const void *)k_pose_bounds<10, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
                MAD_HIP(hipFuncSetAttribute((const void *)k_pose_bounds<12, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
                MAD_HIP(hipFuncSetAttribute((const void *)k_pose_bounds<12, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
                MAD_HIP(hipFuncSetAttribute((const void *)k_pose_bounds<16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
                MAD_HIP(hipFuncSetAttribute((const void *)k_pose_bounds<16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
                MAD_HIP(hipFuncSetAttribute((const void *)k_pose_bounds_mx<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
                MAD_HIP(hipFuncSetAttribute((const void *)k_pose_bounds_mx<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
                MAD_HIP(hipFuncSetAttribute((const void *)k_pose_bounds_mx<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
                MAD_HIP(hipFuncSetAttribute((const void *)k_pose_bounds_mx<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
                attr_b = true;
            }
            // two launches of the bounds kernel, best-scoring pairs first: the second abandons pairs early (k_pose_bounds)
            // ... when a pair has more than eight sets of 64 points to lose: with fewer (C3, C4: ~430 points) the second launch's
            // fixed cost (staging the bitmap again, the score sample) eats what abandoning saves -- measured: C5 (700 points) 8.1 ->
            // 6.8 ms per step, C4 2.28 -> 2.20, C3 1.05 -> 1.05 with its pose stage 0.307 -> 0.342 ms when serialised.
            // mad_set_option "pose_split" (MAD_POSE_SPLIT in the environment) = 1 / 0 forces it on / off.
            const bool split = ctx->pose_split >= 0 ? ctx->pose_split != 0 : l_hi_max > 8 * MAD_WAVE;
            // phase 1: ~5 % of the pairs at the default, at least 64 k of them (fewer pairs than that: everything in one phase)
            static const int split_factor = getenv("MAD_POSE_SPLIT_FACTOR") ? atoi(getenv("MAD_POSE_SPLIT_FACTOR")) : 64;      // diagnostic switch
            const int64_t target_a = std::max<int64_t>(ctx->pose_split_min, (int64_t)split_factor * prune_k);
            const int nb = (l_hi_max + MAD_WAVE - 1) / MAD_WAVE;      // point sets of 64 an entire hi cloud needs
            const int nbv = nb <= 2 ? 2 : (nb <= 4 ? 4 : (nb <= 6 ? 6 : (nb <= 8 ? 8 : (nb <= 10 ? 10 : (nb <= 12 ? 12 : 16)))));
            const size_t lds_b = pad16((size_t)PC.n_words * 4) + pad16((size_t)(l_hi_max + 4) * 16) + (size_t)(PB_THREADS / MAD_WAVE) * (nbv + 1) * MAD_WAVE * 2 +
                                 pad16((size_t)nbins * 4) + 16;
#define MAD_PB_LAUNCH(NBV, SPL)                                                                                                              \
    hipLaunchKernelGGL((k_pose_bounds<NBV, SPL>), dim3(ctx->n_cu), dim3(PB_THREADS), lds_b, ctx->stream, d_status, cap_pairs, d_rec, d_hi_cloud, B, \
                       d_bits, PC, d_bits_c, scratch<int32_t>(ctx, S_COUNTS), scratch<unsigned short>(ctx, S_TMP_C), hist2, nbins,              \
                       (const double *)scratch<double>(ctx, S_PAIR_SCORE), pb_phase, target_a, pb_stop)
            // (T_stop and the ticket of phase 1, hist2[nbins + 1] and [nbins + 2], must lie inside the zero-filled region of the match)
            static_assert(3 <= 17, "hist2 has l_hi_max + 17 words (zr_hist2): bins 0 .. l_hi_max, then T_stop and the ticket");
            if (split && nbins + 3 > l_hi_max + 17) return mad_fail(ctx, MAD_EINVAL, "pose bounds: %d bins for %d hi anchors", nbins, l_hi_max);
            // one phase, at most 8 sets: the coarse map on the matrix cores (k_pose_bounds_mx; MAD_POSE_MX=0: the vector form)
            // Built for the round-3 review and measured slower on C3 (32.4 against 29.2 us per launch for clouds of 7-8 sets, 25.5 against 26.2
            // for 6; DESIGN.md section 6d: the coarse phase loses 7 of its 26 vector instructions per 64 points, the fine phase -- whose map
            // now is a per-lane operand read from LDS -- gains as many per 64 points again): OFF unless mad_set_option "pose_mx" (MAD_POSE_MX=1).
            const bool mx = ctx->pose_mx != 0 && !split && nbv <= 8;
            if (mx) {
                const size_t lds_mx = pad16((size_t)PC.n_words * 4) + pad16((size_t)(l_hi_max + 4) * 16) +
                                      (size_t)(PB_THREADS / MAD_WAVE) * ((size_t)(2 * nbv + 1) * MAD_WAVE * 2 + 192) + pad16((size_t)nbins * 4) + 16;
                if (lds_mx > (size_t)155 * 1024) return mad_fail(ctx, MAD_EINVAL, "pose bounds: %zu bytes of LDS", lds_mx);
#define MAD_PBX_LAUNCH(NBV)                                                                                                                        \
    hipLaunchKernelGGL((k_pose_bounds_mx<NBV>), dim3(ctx->n_cu), dim3(PB_THREADS), lds_mx, ctx->stream, d_status, cap_pairs, d_rec, d_hi_cloud, B, \
                       d_bits, PC, d_bits_c, scratch<int32_t>(ctx, S_COUNTS), scratch<unsigned short>(ctx, S_TMP_C), hist2, nbins)
                if (nbv == 2) MAD_PBX_LAUNCH(2);
                else if (nbv == 4) MAD_PBX_LAUNCH(4);
                else if (nbv == 6) MAD_PBX_LAUNCH(6);
                else MAD_PBX_LAUNCH(8);
#undef MAD_PBX_LAUNCH
            }
            for (int pass = 0; pass < (mx ? 0 : (split ? 2 : 1)); pass++) {
                const int pb_phase = split ? pass + 1 : 0;
                const int64_t pb_stop = split ? prune_k : 0;      // phase 1 leaves T_stop for phase 2 (hist2[nbins + 1]; [nbins + 2]: its workgroups' tickets)
                if (split) {
                    if (nbv == 2) MAD_PB_LAUNCH(2, true);
                    else if (nbv == 4) MAD_PB_LAUNCH(4, true);
                    else if (nbv == 6) MAD_PB_LAUNCH(6, true);
                    else if (nbv == 8) MAD_PB_LAUNCH(8, true);
                    else if (nbv == 10) MAD_PB_LAUNCH(10, true);
                    else if (nbv == 12) MAD_PB_LAUNCH(12, true);
                    else MAD_PB_LAUNCH(16, true);
                } else {
                    if (nbv == 2) MAD_PB_LAUNCH(2, false);
                    else if (nbv == 4) MAD_PB_LAUNCH(4, false);
                    else if (nbv == 6) MAD_PB_LAUNCH(6, false);
                    else if (nbv == 8) MAD_PB_LAUNCH(8, false);
                    else if (nbv == 10) MAD_PB_LAUNCH(10, false);
                    else if (nbv == 12) MAD_PB_LAUNCH(12, false);
                    else MAD_PB_LAUNCH(16, false);
                }
            }
#undef MAD_PB_LAUNCH
            // the selection: inside the exact search itself (k_pose_lds, own) when this lane's previous match has told how many pairs
            // to expect, else a launch of its own
            if (fused && fits64 && ctx->lane_sel_hint[ctx->lane] > 0) {
                own.upper = scratch<unsigned short>(ctx, S_TMP_C); own.hist = hist2; own.nbins = nbins; own.k = prune_k;
                own.sel_out = scratch<int32_t>(ctx, S_TMP_D); own.sel_cap = cap_pairs; own.status_w = d_status;
                if (own_sel) *own_sel = true;
            } else {
                hipLaunchKernelGGL(k_prune_select, dim3(ctx->n_cu / 2), dim3(256), 0, ctx->stream, d_status, cap_pairs, hist2, nbins, prune_k,
                                   scratch<unsigned short>(ctx, S_TMP_C), scratch<int32_t>(ctx, S_TMP_D), d_status + ST_NSEL, hist2 + nbins,
                                   fits64 ? scratch<int32_t>(ctx, S_COUNTS) : (int32_t *)nullptr);      // k_pose_lds adds up partial counts
                d_sel = scratch<int32_t>(ctx, S_TMP_D);
            }
            if (pruned) *pruned = true;
        }
        // the exact search of a pruned match sees a few hundred pairs: a grid sized from what the previous match in this lane
        // selected (one pair per wave, 25 % spare) instead of one workgroup per CU staging both clouds for nothing.  A larger
        // selection than expected is still searched completely, the kernels are persistent.
        unsigned wgs_sel = (unsigned)ctx->n_cu;
        if ((d_sel || own.upper) && ctx->lane_sel_hint[ctx->lane] > 0)
            wgs_sel = (unsigned)std::min<int64_t>(ctx->n_cu, std::max<int64_t>(16, (ctx->lane_sel_hint[ctx->lane] * (fits64 ? POSE_PARTS : 1) * 5 / 4) / (POSE_LDS_THREADS / MAD_WAVE) + 4));
        const bool listed = d_sel || own.upper;
        if (fits64)
            hipLaunchKernelGGL(k_pose_lds, dim3(listed ? wgs_sel : ctx->n_cu * (P.lds > 80 * 1024 ? 1 : pose_wgs_per_cu())), dim3(POSE_LDS_THREADS), P.lds, ctx->stream, d_status, cap_pairs, d_rec,
                               d_hi_cloud, scratch<double>(ctx, S_PG_PTS), scratch<int32_t>(ctx, S_PG_START), d_start16, G,
                               l_hi_max, n_cloud, (float)reach, dd_lim, B, d_bits, scratch<int32_t>(ctx, S_COUNTS), d_sel, own);
        else if (P.hi_in_lds)
            hipLaunchKernelGGL(k_pose_lds32<true>, dim3(d_sel ? wgs_sel : ctx->n_cu), dim3(POSE_LDS_THREADS), P.lds32_hi, ctx->stream, d_status, cap_pairs, d_rec,
                               d_hi_cloud, scratch<double>(ctx, S_PG_PTS), scratch<float4>(ctx, S_PG_PTSF),
                               scratch<int32_t>(ctx, S_PG_START), d_start16, G, n_cloud, l_hi_max, (float)reach, dd_lim, P.lim_in, P.lim_out, B, d_bits,
                               scratch<int32_t>(ctx, S_COUNTS), d_sel);
        else
            hipLaunchKernelGGL(k_pose_lds32<false>, dim3(d_sel ? wgs_sel : ctx->n_cu), dim3(POSE_LDS_THREADS), P.lds32, ctx->stream, d_status, cap_pairs, d_rec,
                               d_hi_cloud, scratch<double>(ctx, S_PG_PTS), scratch<float4>(ctx, S_PG_PTSF),
                               scratch<int32_t>(ctx, S_PG_START), d_start16, G, n_cloud, l_hi_max, (float)reach, dd_lim, P.lim_in, P.lim_out, B, d_bits,
                               scratch<int32_t>(ctx, S_COUNTS), d_sel);
        mad_timer_end(ctx, MAD_T_POSE);
        MAD_HIP(hipGetLastError());
        return MAD_OK;
    }
    if (!fallback) return mad_fail(ctx, MAD_EINVAL, "pose: clouds of %d + %d points need the global cell list", l_hi_max, n_cloud);
    if (job) hipLaunchKernelGGL(k_compact_cloud, dim3(1), dim3(1024), 0, ctx->stream, *job);
    const size_t lds2 = (size_t)l_hi_max * 24;
    if (lds2 > 150 * 1024) return mad_fail(ctx, MAD_EINVAL, "pose: hi cloud of %d anchors does not fit LDS", l_hi_max);
    MAD_HIP(hipFuncSetAttribute((const void *)k_pose, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MAD_HIP(hipMemsetAsync(d_status + ST_LLO, 0, 4, ctx->stream));
    if (d_cloud_used)
        hipLaunchKernelGGL(k_count_flags, dim3((unsigned)mad_ceil_div(n_cloud, 256)), dim3(256), 0, ctx->stream, d_cloud_used, n_cloud,
                           d_status + ST_LLO);
    else
        MAD_HIP(hipMemcpyAsync(d_status + ST_LLO, &n_cloud, 4, hipMemcpyHostToDevice, ctx->stream));
    ctx->last_pose_kernel = 2;
    mad_timer_begin(ctx, MAD_T_POSE);
    hipLaunchKernelGGL(k_pose, dim3(ctx->n_cu * 8), dim3(POSE_THREADS), lds2, ctx->stream, scratch<int32_t>(ctx, S_PAIR_HI),
                       scratch<int32_t>(ctx, S_PAIR_LO), d_status, cap_pairs, hi.p, hi.R, lo.p, lo.Rinv, hi.row_anchor, lo.row_anchor,
                       d_hi_cloud, *fallback, dd_lim, scratch<int32_t>(ctx, S_COUNTS));
    mad_timer_end(ctx, MAD_T_POSE);
    MAD_HIP(hipGetLastError());
    return MAD_OK;
}

static bool clouds_fit_lds(int64_t l_hi, int64_t l_lo) {      // with the largest cell table pose_device makes (25^3 cells)
    // as pose_device sizes them (+ its 16-byte paddings): the float64 kernel carries a queue and two pair records per wave, the
    // float32 one a survivor stack
    const size_t cells = 15632 * 2 + 16 + 64;      // pose_device caps the grid at 25 cells per axis
    const size_t fixed = cells + (POSE_LDS_THREADS / MAD_WAVE) * POSE_WAVE_LDS + (size_t)POSE_OWN_CAP * 4, fixed32 = cells + (POSE_LDS_THREADS / MAD_WAVE) * POSE_STACK * 2;
    return ((size_t)(l_hi + l_lo) * 24 + (size_t)l_hi * 16 + fixed <= 150 * 1024 || (size_t)(l_lo + 1) * 16 + fixed32 <= 150 * 1024) && l_lo < 65535 &&
           l_hi < 65536;      // (the first alternative implies l_hi < 32768)
}

static int32_t *status_words(mad_ctx *ctx) {      // inside S_MISC
    return scratch<int32_t>(ctx, S_MISC) + 64;
}

// ---------------------------------------------------------------------------
// stage API
// ---------------------------------------------------------------------------

// int16 rows from the host -> padded int8 + norms on the device; n to a device word
static int stage_rows(mad_ctx *ctx, const int16_t *h_rows, int64_t n, int D, int slot16, int slot8, int slotn, int32_t *d_n,
                      int32_t *d_bad) {
    const int64_t n_pad = mad_ceil_div(n > 0 ? n : 1, GEMM_BM) * GEMM_BM;
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, slot16), (size_t)n * D * 2));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, slot8), (size_t)n_pad * D));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, slotn), (size_t)n_pad * 8));
    const int32_t n32 = (int32_t)n;
    MAD_HIP(hipMemcpyAsync(d_n, &n32, 4, hipMemcpyHostToDevice, ctx->stream));
    MAD_HIP(hipMemcpyAsync(mad_sb(ctx, slot16).p, h_rows, (size_t)n * D * 2, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_pack_rows, dim3((unsigned)std::min<int64_t>(mad_ceil_div(n_pad, 4), 4096)), dim3(256), 0, ctx->stream,
                       scratch<int16_t>(ctx, slot16), d_n, D, scratch<int8_t>(ctx, slot8), scratch<double>(ctx, slotn), d_bad);
    MAD_HIP(hipGetLastError());
    return MAD_OK;
}

extern "C" int mad_correlate(mad_ctx *ctx, const int16_t *hi, int64_t n_hi, const int16_t *lo, int64_t n_lo, int D,
                             double cc, int32_t *pair_hi, int32_t *pair_lo, double *pair_score, int64_t *n_pairs,
                             int64_t cap) {
    if (ctx) mad_use_lane(ctx, 0);
    if (!ctx || !n_pairs) return MAD_EINVAL;
    *n_pairs = 0;
    if (n_hi <= 0 || n_lo <= 0) return MAD_OK;
    if (!hi || !lo) return mad_fail(ctx, MAD_EINVAL, "mad_correlate: NULL descriptors");
    if (n_hi * n_lo >= ((int64_t)1 << 31)) return mad_fail(ctx, MAD_EINVAL, "mad_correlate: %lld x %lld too large", (long long)n_hi, (long long)n_lo);
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_MISC), 4096));
    int32_t *st = status_words(ctx);
    MAD_HIP(hipMemsetAsync(st, 0, ST_COUNT * 4, ctx->stream));
    MAD_TRY(stage_rows(ctx, hi, n_hi, D, S_HI16, S_HI8, S_HNORM, st + ST_NHI, st + ST_BAD));
    MAD_TRY(stage_rows(ctx, lo, n_lo, D, S_LO16, S_LO8, S_LNORM, st + ST_NLO, st + ST_BAD));
    const int64_t hp = mad_ceil_div(n_hi, GEMM_BM) * GEMM_BM, lp = mad_ceil_div(n_lo, GEMM_BN) * GEMM_BN;
    const Side H = {scratch<int8_t>(ctx, S_HI8), scratch<double>(ctx, S_HNORM), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, st + ST_NHI, n_hi};
    const Side L = {scratch<int8_t>(ctx, S_LO8), scratch<double>(ctx, S_LNORM), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, st + ST_NLO, n_lo};
    const int32_t *hs = (const int32_t *)&ctx->pinned[0];
    int64_t cap_pairs = std::max<int64_t>(cap, 1);
    for (int attempt = 0; attempt < 2; attempt++) {
        MAD_TRY(correlate_device(ctx, H, L, D, cc, st, hp * lp, cap_pairs, nullptr, nullptr));
        MAD_HIP(hipMemcpyAsync(&ctx->pinned[0], st, ST_COUNT * 4, hipMemcpyDeviceToHost, ctx->stream));
        MAD_HIP(hipStreamSynchronize(ctx->stream));
        if (hs[ST_BAD]) return mad_fail(ctx, MAD_EDOM, "descriptor count outside the int8 range");
        *n_pairs = hs[ST_NPAIRS];
        if (!hs[ST_FLAG_PAIRS]) break;
        if (*n_pairs > cap) return mad_fail(ctx, MAD_ENOSPC, "mad_correlate: %lld pairs, capacity %lld", (long long)*n_pairs, (long long)cap);
        cap_pairs = *n_pairs;
        MAD_HIP(hipMemsetAsync(st + ST_FLAG_PAIRS, 0, 4, ctx->stream));
    }
    const int64_t np = *n_pairs;
    if (np > cap) return mad_fail(ctx, MAD_ENOSPC, "mad_correlate: %lld pairs, capacity %lld", (long long)np, (long long)cap);
    if (np > 0) {
        if (pair_hi) MAD_HIP(hipMemcpyAsync(pair_hi, mad_sb(ctx, S_PAIR_HI).p, np * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (pair_lo) MAD_HIP(hipMemcpyAsync(pair_lo, mad_sb(ctx, S_PAIR_LO).p, np * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (pair_score) MAD_HIP(hipMemcpyAsync(pair_score, mad_sb(ctx, S_PAIR_SCORE).p, np * 8, hipMemcpyDeviceToHost, ctx->stream));
        MAD_HIP(hipStreamSynchronize(ctx->stream));
    }
    return MAD_OK;
}

extern "C" int mad_pose_score(mad_ctx *ctx, const int32_t *pair_hi, const int32_t *pair_lo, const double *pair_score,
                              int64_t n_pairs, const double *hi_p, const double *hi_R, const int32_t *hi_meta, int64_t n_hi,
                              const double *lo_p, const double *lo_R, const int32_t *lo_meta, int64_t n_lo,
                              const double *hi_cloud, int64_t l_hi, const double *lo_cloud, int64_t l_lo, double dist,
                              double *results, int32_t *counts) {
    if (ctx) mad_use_lane(ctx, 0);
    if (!ctx) return MAD_EINVAL;
    if (n_pairs <= 0) return MAD_OK;
    if (!pair_hi || !pair_lo || !pair_score || !hi_p || !hi_R || !hi_meta || !lo_p || !lo_R || !lo_meta || !hi_cloud || !lo_cloud)
        return mad_fail(ctx, MAD_EINVAL, "mad_pose_score: NULL argument");
    if (l_hi <= 0 || l_lo <= 0 || !(dist > 0)) return mad_fail(ctx, MAD_EINVAL, "mad_pose_score: empty cloud or dist <= 0");
    struct Up { int slot; const void *src; size_t bytes; };
    const Up ups[] = {
        {S_PAIR_HI, pair_hi, (size_t)n_pairs * 4}, {S_PAIR_LO, pair_lo, (size_t)n_pairs * 4},
        {S_PAIR_SCORE, pair_score, (size_t)n_pairs * 8},
        {S_TMP_E, hi_p, (size_t)n_hi * 24}, {S_TMP_F, hi_R, (size_t)n_hi * 72}, {S_TMP_G, hi_meta, (size_t)n_hi * 12},
        {S_TMP_H, lo_p, (size_t)n_lo * 24}, {S_TMP_I, lo_R, (size_t)n_lo * 72}, {S_TMP_J, lo_meta, (size_t)n_lo * 12},
        {S_HI_CLOUD, hi_cloud, (size_t)l_hi * 24}, {S_USED_LO, lo_cloud, (size_t)l_lo * 24},
    };
    for (const Up &u : ups) {
        MAD_TRY(mad_reserve(ctx, mad_sb(ctx, u.slot), u.bytes));
        MAD_HIP(hipMemcpyAsync(mad_sb(ctx, u.slot).p, u.src, u.bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_TMP_A), (size_t)n_lo * 72));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_MISC), 4096));
    int32_t *st = status_words(ctx);
    int32_t hs[ST_COUNT] = {0};
    hs[ST_NPAIRS] = (int32_t)n_pairs; hs[ST_LHI] = (int32_t)l_hi; hs[ST_NHI] = (int32_t)n_hi; hs[ST_NLO] = (int32_t)n_lo;
    MAD_HIP(hipMemcpyAsync(st, hs, sizeof(hs), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_row_aux, dim3(64), dim3(256), 0, ctx->stream, scratch<double>(ctx, S_TMP_I), st + ST_NLO,
                       scratch<double>(ctx, S_TMP_A), (const int32_t *)nullptr, (const int32_t *)nullptr, (const int32_t *)nullptr,
                       (const int32_t *)nullptr, (int32_t *)nullptr);
    double bmn[3] = {0, 0, 0}, bmx[3] = {0, 0, 0};      // bounding box of the lo cloud
    for (int64_t i = 0; i < l_lo; i++)
        for (int d = 0; d < 3; d++) {
            const double v = lo_cloud[3 * i + d];
            if (i == 0 || v < bmn[d]) bmn[d] = v;
            if (i == 0 || v > bmx[d]) bmx[d] = v;
        }
    const Side H = {nullptr, nullptr, scratch<double>(ctx, S_TMP_F), nullptr, scratch<int32_t>(ctx, S_TMP_G), nullptr,
                    nullptr, scratch<double>(ctx, S_TMP_E), st + ST_NHI, n_hi};
    const Side L = {nullptr, nullptr, scratch<double>(ctx, S_TMP_I), scratch<double>(ctx, S_TMP_A), scratch<int32_t>(ctx, S_TMP_J),
                    nullptr, nullptr, scratch<double>(ctx, S_TMP_H), st + ST_NLO, n_lo};
    CellGrid G;
    const bool fits = clouds_fit_lds(l_hi, l_lo);
    if (!fits) {      // global cell list (cell = dist) over the lo cloud
        double mn[3];
        int dim[3];
        MAD_TRY(build_cells(ctx, bmn, bmx, scratch<double>(ctx, S_USED_LO), (int)l_lo, dist, mad_sb(ctx, S_CELL_START),
                            mad_sb(ctx, S_CELL_PTS), mad_sb(ctx, S_CELL_IDS), mn, dim));
        G.start = scratch<int32_t>(ctx, S_CELL_START); G.pts = scratch<double>(ctx, S_CELL_PTS); G.ids = scratch<int32_t>(ctx, S_CELL_IDS);
        G.used = nullptr;
        for (int d = 0; d < 3; d++) { G.mn[d] = mn[d]; G.dim[d] = dim[d]; }
        G.cell = dist;
    }
    MAD_TRY(pose_device(ctx, H, L, st, n_pairs, scratch<double>(ctx, S_HI_CLOUD), (int)l_hi, scratch<double>(ctx, S_USED_LO), (int)l_lo,
                        nullptr, bmn, bmx, fits ? nullptr : &G, dist));
    if (results) {
        MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_RESULTS), (size_t)n_pairs * MAD_RESULT_COLS * 8));
        hipLaunchKernelGGL(k_results, dim3((unsigned)std::min<int64_t>(mad_ceil_div(n_pairs, 256), 4096)), dim3(256), 0, ctx->stream,
                           (const int64_t *)nullptr, st + ST_NPAIRS, n_pairs, scratch<int32_t>(ctx, S_PAIR_HI),
                           scratch<int32_t>(ctx, S_PAIR_LO), scratch<double>(ctx, S_PAIR_SCORE), scratch<int32_t>(ctx, S_COUNTS), st,
                           H.p, H.R, H.meta, L.p, L.Rinv, L.meta, (const int32_t *)nullptr, (const int32_t *)nullptr,
                           scratch<double>(ctx, S_RESULTS), 0);
        MAD_HIP(hipGetLastError());
        MAD_HIP(hipMemcpyAsync(results, mad_sb(ctx, S_RESULTS).p, (size_t)n_pairs * MAD_RESULT_COLS * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (counts) MAD_HIP(hipMemcpyAsync(counts, mad_sb(ctx, S_COUNTS).p, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, ctx->stream));
    MAD_HIP(hipStreamSynchronize(ctx->stream));
    return MAD_OK;
}

extern "C" int mad_topk(mad_ctx *ctx, const int32_t *counts, int64_t n, int64_t k, int64_t *order) {
    if (ctx) mad_use_lane(ctx, 0);
    if (!ctx) return MAD_EINVAL;
    if (n <= 0 || k <= 0) return MAD_OK;
    if (!counts || !order) return mad_fail(ctx, MAD_EINVAL, "mad_topk: NULL argument");
    int maxc = 0;
    for (int64_t i = 0; i < n; i++) {
        if (counts[i] < 0) return mad_fail(ctx, MAD_EINVAL, "mad_topk: negative count");
        if (counts[i] > maxc) maxc = counts[i];
    }
    if (k > n) k = n;
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_COUNTS), (size_t)n * 4));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_SEL_OUT), (size_t)(k + 8) * 8));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_MISC), 4096));
    int32_t *st = status_words(ctx);
    int32_t hs[ST_COUNT] = {0};
    hs[ST_NPAIRS] = (int32_t)n;
    MAD_HIP(hipMemcpyAsync(st, hs, sizeof(hs), hipMemcpyHostToDevice, ctx->stream));
    MAD_HIP(hipMemcpyAsync(mad_sb(ctx, S_COUNTS).p, counts, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    MAD_TRY(topk_device(ctx, scratch<int32_t>(ctx, S_COUNTS), st, n, k, maxc, scratch<int64_t>(ctx, S_SEL_OUT)));
    MAD_HIP(hipMemcpyAsync(order, mad_sb(ctx, S_SEL_OUT).p, (size_t)k * 8, hipMemcpyDeviceToHost, ctx->stream));
    MAD_HIP(hipStreamSynchronize(ctx->stream));
    return MAD_OK;
}

// ---------------------------------------------------------------------------
// device-resident sets
// ---------------------------------------------------------------------------

extern "C" int mad_set_create(mad_ctx *ctx, mad_set **out) {
    if (!ctx || !out) return MAD_EINVAL;
    mad_set *s = new mad_set();
    if (hipEventCreateWithFlags(&s->ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s->uploaded, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s->built, hipEventDisableTiming) != hipSuccess) {
        if (s->ready) (void)hipEventDestroy(s->ready);
        if (s->uploaded) (void)hipEventDestroy(s->uploaded);
    if (s->built) (void)hipEventDestroy(s->built);
        delete s;
        return mad_fail(ctx, MAD_EHIP, "mad_set_create: event creation failed");
    }
    s->lane = ctx->next_set_lane;      // sets take turns on the lanes, so that their builds overlap
    ctx->next_set_lane = (ctx->next_set_lane + 1) % MAD_LANES;
    s->pinned_slot = ctx->next_pinned;
    ctx->next_pinned = 64 + (ctx->next_pinned - 64 + 2) % 900;      // two 8-byte slots per set
    *out = s;
    return MAD_OK;
}

extern "C" void mad_set_destroy(mad_ctx *ctx, mad_set *s) {
    if (!s) return;
    if (ctx) (void)mad_synchronize(ctx);
    if (ctx) {      // results that still refer to this set (mad_match_fetch(counts), mad_match_results) are refused from now on
        if (ctx->match.hi == s) ctx->match.hi = nullptr;
        if (ctx->match.lo == s) ctx->match.lo = nullptr;
        if (ctx->match.shard_hi == s) ctx->match.shard_hi = nullptr;
        if (ctx->match.shard_lo == s) ctx->match.shard_lo = nullptr;
    }
    DevBuf *bufs[] = {&s->anc_blob, &s->row_anchor, &s->row_main, &s->row_sec, &s->row_R, &s->row_Rinv, &s->row_meta, &s->dsc,
                      &s->dsc8, &s->norm, &s->row_perm, &s->row_rec, &s->anc_rows, &s->cell_start, &s->cell_pts, &s->cell_ids};      // anc_* and dev_n are views
    for (DevBuf *b : bufs) mad_release(*b);
    if (s->host_stage) (void)hipHostFree(s->host_stage);
    if (s->ready) (void)hipEventDestroy(s->ready);
    if (s->uploaded) (void)hipEventDestroy(s->uploaded);
    if (s->built) (void)hipEventDestroy(s->built);
    delete s;
}

// the row count of a set on the host: read back on demand (the pipeline itself never needs it); repeats the
// describe stage if its launch had been sized too small
static int set_rows(mad_ctx *ctx, const mad_set *cs, int64_t *n_rows) {
    mad_set *s = const_cast<mad_set *>(cs);
    if (s->n_rows_host < 0) {
        const int lane_before = ctx->lane;
        mad_use_lane(ctx, s->lane);
        struct Back { mad_ctx *c; int l; ~Back() { mad_use_lane(c, l); } } back{ctx, lane_before};
        MAD_HIP(hipStreamWaitEvent(ctx->stream, s->built, 0));      // a batched build runs on the lane of the batch's first set
        const int32_t *h = (const int32_t *)&ctx->pinned[s->pinned_slot];
        MAD_HIP(hipMemcpyAsync(&ctx->pinned[s->pinned_slot], s->dev_n.p, 16, hipMemcpyDeviceToHost, ctx->stream));
        MAD_HIP(hipStreamSynchronize(ctx->stream));
        if (h[3] && s->last_r > 0) {
            MAD_HIP(hipMemsetAsync((int32_t *)s->dev_n.p + 3, 0, 4, ctx->stream));
            DescribeJob J;
            J.f[0] = s->last_f[0]; J.f[1] = s->last_f[1];
            J.d_anc_coords = (const int32_t *)s->anc_coords.p; J.d_anc_octave = (const int32_t *)s->anc_octave.p; J.uniform_octave = 0;
            J.d_row_anchor = (const int32_t *)s->row_anchor.p; J.d_row_R = (const double *)s->row_R.p; J.d_row_Rinv = (const double *)s->row_Rinv.p;
            J.d_row_perm = s->last_perm ? (const int32_t *)s->row_perm.p : nullptr;
            J.d_row_rec = s->last_rec ? (const DscRowRec *)s->row_rec.p : nullptr;
            if (s->last_perm && s->last_rec && s->anc_rows.p && s->ball_dims[0] > 0 && s->ball_dims[0] == s->last_f[1].nx &&
                s->ball_dims[1] == s->last_f[1].ny && s->ball_dims[2] == s->last_f[1].nz) {
                J.d_anc_rows = (const int32_t *)s->anc_rows.p; J.n_anchors = s->n_anchors; J.n_rowwise = s->n_rowwise; J.fan = s->last_fan;
            }
            J.d_n_rows = (const int32_t *)s->dev_n.p; J.grid_rows = s->cap_rows; J.d_overflow = (int32_t *)s->dev_n.p + 3;
            J.d_dsc = (int16_t *)s->dsc.p; J.d_dsc8 = (int8_t *)s->dsc8.p; J.d_norm = (double *)s->norm.p;
            MAD_TRY(mad_describe_device_many(ctx, 1, &J, s->last_r));
            MAD_HIP(hipEventRecord(s->built, ctx->stream));
            MAD_HIP(hipMemcpyAsync(&ctx->pinned[s->pinned_slot], s->dev_n.p, 16, hipMemcpyDeviceToHost, ctx->stream));
            MAD_HIP(hipStreamSynchronize(ctx->stream));
            if (h[3]) return mad_fail(ctx, MAD_EINVAL, "mad_set_build: the describe stage could not be completed (flag %d)", h[3]);
        }
        if (h[3] && s->last_r == 0) {      // an imported set (mad_set_import) that could not be completed
            if (h[3] < 0) return mad_fail(ctx, MAD_EINVAL, "mad_set_import: malformed wire image (size, capacity or anchor ids do not match)");
            return mad_fail(ctx, MAD_ENOSPC, "mad_set_import: a share has %d rows, more than the wire images hold; export again with a larger cap_rows", h[3]);
        }
        s->n_rows_host = h[0];
        s->rows_hint = h[0];
        s->range_bad = h[1] != 0;
    }
    *n_rows = s->n_rows_host;
    return MAD_OK;
}

// ball_dims (nullable): the base-octave grid; base-octave anchors whose sample ball lies inside it (mad_ball_interior) are sorted
// behind all others in working order -- k_describe_ball takes them, s->n_rowwise anchors stay with k_describe
static int set_upload_anchors(mad_ctx *ctx, mad_set *s, const int32_t *anc_coords, const int32_t *anc_octave,
                              const double *anc_subv, const int32_t *anc_index, int n, int32_t rows0 = 0, const int *ball_dims = nullptr) {
    s->n_anchors = n;
    const int bd[3] = {ball_dims && anc_coords ? ball_dims[0] : 0, ball_dims && anc_coords ? ball_dims[1] : 0, ball_dims && anc_coords ? ball_dims[2] : 0};
    s->gen++;
    const size_t m = (size_t)(n > 0 ? n : 1);
    const size_t o_subv = 64, o_coords = o_subv + m * 24, o_oct = o_coords + m * 12, o_idx = o_oct + m * 4, o_canon = o_idx + m * 4,
                 o_order = o_canon + m * 4, total = (o_order + m * 4 + 15) / 16 * 16;
    if (s->host_stage_cap < total) {
        if (s->host_stage) {
            MAD_HIP(hipEventSynchronize(s->uploaded));
            (void)hipHostFree(s->host_stage);
            s->host_stage = nullptr;
            s->host_stage_cap = 0;
        }
        const size_t want = total + total / 2;
        if (hipHostMalloc(&s->host_stage, want) != hipSuccess) return mad_fail(ctx, MAD_ENOMEM, "pinned anchor staging of %zu bytes", want);
        s->host_stage_cap = want;
        s->staged_n = -1;      // a fresh buffer holds nothing
    } else {
        MAD_HIP(hipEventSynchronize(s->uploaded));      // the previous copy out of the staging buffer has finished
    }
    const void *blob_before = s->anc_blob.p;
    MAD_TRY(mad_reserve(ctx, s->anc_blob, total));
    char *h = (char *)s->host_stage, *d = (char *)s->anc_blob.p;
    s->dev_n.p = d;
    s->anc_subv.p = d + o_subv; s->anc_coords.p = d + o_coords; s->anc_octave.p = d + o_oct; s->anc_index.p = d + o_idx;
    s->anc_canon.p = d + o_canon; s->anc_order.p = d + o_order;
    // The same anchors as last time (a set rebuilt in place, step after step): the staging buffer and the device copy already
    // hold them -- only the counters are reset.  Decided by comparing the bytes, not by trusting the caller.
    const bool same = n > 0 && s->staged_n == n && s->staged_coords == (anc_coords != nullptr) && blob_before == s->anc_blob.p &&
                      bd[0] == s->ball_dims[0] && bd[1] == s->ball_dims[1] && bd[2] == s->ball_dims[2] &&
                      memcmp(h + o_subv, anc_subv, (size_t)n * 24) == 0 && (!anc_coords || memcmp(h + o_coords, anc_coords, (size_t)n * 12) == 0) &&
                      memcmp(h + o_oct, anc_octave, (size_t)n * 4) == 0 && memcmp(h + o_idx, anc_index, (size_t)n * 4) == 0;
    memset(h, 0, 64);      // the device counters start from zero
    ((int32_t *)h)[0] = rows0;
    if (same) {
        mad_copy_words(ctx, d, h, 64);
        MAD_HIP(hipGetLastError());
        MAD_HIP(hipEventRecord(s->uploaded, ctx->stream));
        s->n_rows_host = -1;
        return MAD_OK;
    }
    s->staged_n = n; s->staged_coords = anc_coords != nullptr;
    s->ball_dims[0] = bd[0]; s->ball_dims[1] = bd[1]; s->ball_dims[2] = bd[2];
    s->n_rowwise = n;
    if (n > 0) {
        memcpy(h + o_subv, anc_subv, (size_t)n * 24);
        if (anc_coords) memcpy(h + o_coords, anc_coords, (size_t)n * 12);
        else memset(h + o_coords, 0, (size_t)n * 12);
        memcpy(h + o_oct, anc_octave, (size_t)n * 4);
        memcpy(h + o_idx, anc_index, (size_t)n * 4);
        // Anchors with identical coordinates are ONE point of a cloud: the reference builds its clouds with
        // np.unique(subv_map_coords, axis=0) (MaD.py:427-428), and two detector peaks do now and then converge on the same
        // sub-voxel position.  canon[i] = the first anchor with the coordinates of anchor i; the "takes part in a pair" flags
        // are raised on the canonical anchor only, so the cloud and its size l count such a position once.
        int32_t *canon = (int32_t *)(h + o_canon);
        // (the bytes of an anchor's three coordinates hashed to 40 bits and sorted with the anchor's number in the low 24: one sort of
        // plain 64-bit words -- the lists arrive fresh every step, this runs on the host in front of every build -- and the bytes
        // themselves compared only inside a run of equal hashes)
        static thread_local std::vector<uint64_t> keys;
        keys.resize((size_t)n);
        const bool packable = n < (1 << 24);
        if (packable) {
            for (int i = 0; i < n; i++) {
                uint64_t w[3];
                memcpy(w, anc_subv + 3 * i, 24);
                uint64_t x = w[0] * 0x9E3779B97F4A7C15ull;
                x = (x ^ (x >> 29) ^ w[1]) * 0xBF58476D1CE4E5B9ull;
                x = (x ^ (x >> 32) ^ w[2]) * 0x94D049BB133111EBull;
                x ^= x >> 31;
                keys[i] = (x & ~(uint64_t)0xffffff) | (uint64_t)i;
            }
            std::sort(keys.begin(), keys.end());
            for (int i = 0; i < n;) {
                int e = i + 1;
                while (e < n && (keys[e] >> 24) == (keys[i] >> 24)) e++;
                // anchors i .. e - 1 (ascending numbers) share a hash: each takes the first of them with its very bytes (nearly always: itself)
                for (int a = i; a < e; a++) {
                    const int ia = (int)(keys[a] & 0xffffff);
                    int first = ia;
                    for (int b = i; b < a; b++) {
                        const int ib = (int)(keys[b] & 0xffffff);
                        if (memcmp(anc_subv + 3 * ia, anc_subv + 3 * ib, 24) == 0) { first = canon[ib]; break; }
                    }
                    canon[ia] = first;
                }
                i = e;
            }
        } else {
            std::vector<int32_t> order(n);
            for (int i = 0; i < n; i++) order[i] = i;
            std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
                const int c = memcmp(anc_subv + 3 * a, anc_subv + 3 * b, 24);
                return c != 0 ? c < 0 : a < b;
            });
            for (int i = 0; i < n; i++)
                canon[order[i]] = (i > 0 && memcmp(anc_subv + 3 * order[i], anc_subv + 3 * order[i - 1], 24) == 0) ? canon[order[i - 1]] : order[i];
        }
        // The order the build kernels work in: by octave (one texture each), then along a Morton curve through the voxel
        // coordinates.  The reference lists anchors by detector response; workgroups that run side by side then sample balls
        // all over a 1-2 GB texture.  In Morton order they sample overlapping balls and meet in the XCD's L2.
        int32_t *work = (int32_t *)(h + o_order);
        for (int i = 0; i < n; i++) work[i] = i;
        if (anc_coords) {
            auto spread = [](uint64_t v) {      // 21 bits -> every third bit
                v &= 0x1fffff;
                v = (v | v << 32) & 0x1f00000000ffffull; v = (v | v << 16) & 0x1f0000ff0000ffull; v = (v | v << 8) & 0x100f00f00f00f00full;
                v = (v | v << 4) & 0x10c30c30c30c30c3ull; v = (v | v << 2) & 0x1249249249249249ull;
                return v;
            };
            bool small = packable;      // coordinates below 2^12 (after the octave's shift): class (2 bits) | Morton (36 bits) | number (24 bits) in one word
            for (int i = 0; i < n; i++) {
                const int sh = anc_octave[i] == 0 ? 1 : 0;      // the same physical cell size in both octaves
                const uint64_t x = (uint64_t)std::max(anc_coords[3 * i], 0) >> sh, y = (uint64_t)std::max(anc_coords[3 * i + 1], 0) >> sh,
                               z = (uint64_t)std::max(anc_coords[3 * i + 2], 0) >> sh;
                // (bit 63: a base-octave anchor whose ball of samples lies inside the grid -- k_describe_ball's, behind all others)
                const bool ball = bd[0] > 0 && anc_octave[i] == 1 &&
                                  mad_ball_interior(anc_coords[3 * i], anc_coords[3 * i + 1], anc_coords[3 * i + 2], bd[0], bd[1], bd[2]);
                if (ball) s->n_rowwise--;
                small = small && (x | y | z) < 4096;
                keys[i] = ((uint64_t)ball << 63) | ((uint64_t)(anc_octave[i] != 0) << 62) | ((spread(x) << 2 | spread(y) << 1 | spread(z)) & ~(3ull << 62));
            }
            if (small) {      // equal keys (anchors in one cell) keep their list order, as below: the number is the low end of the word
                for (int i = 0; i < n; i++) keys[i] = (keys[i] & (3ull << 62)) | ((keys[i] & ((1ull << 36) - 1)) << 24) | (uint64_t)i;
                std::sort(keys.begin(), keys.end());
                for (int i = 0; i < n; i++) work[i] = (int32_t)(keys[i] & 0xffffff);
            } else {
                std::sort(work, work + n, [&](int32_t a, int32_t b) { return keys[a] != keys[b] ? keys[a] < keys[b] : a < b; });
            }
        }
    }
    // a kernel reads the pinned buffer directly: in stream order, without the copy engine's start-up latency
    mad_copy_words(ctx, d, h, n > 0 ? total : 64);
    MAD_HIP(hipGetLastError());
    MAD_HIP(hipEventRecord(s->uploaded, ctx->stream));
    for (int i = 0; i < n; i++)
        for (int d3 = 0; d3 < 3; d3++) {
            const double v = anc_subv[3 * i + d3];
            if (i == 0 || v < s->bb_min[d3]) s->bb_min[d3] = v;
            if (i == 0 || v > s->bb_max[d3]) s->bb_max[d3] = v;
        }
    s->cells_ready = false;
    s->n_rows_host = -1;
    return MAD_OK;
}

static int set_reserve_rows(mad_ctx *ctx, mad_set *s, int64_t cap) {
    const int64_t cap_pad = mad_ceil_div(cap > 0 ? cap : 1, GEMM_BM) * GEMM_BM;
    s->cap_rows = cap;
    MAD_TRY(mad_reserve(ctx, s->row_anchor, (size_t)cap_pad * 4));
    MAD_TRY(mad_reserve(ctx, s->row_main, (size_t)cap_pad * 4));
    MAD_TRY(mad_reserve(ctx, s->row_sec, (size_t)cap_pad * 4));
    MAD_TRY(mad_reserve(ctx, s->row_R, (size_t)cap_pad * 72));
    MAD_TRY(mad_reserve(ctx, s->row_Rinv, (size_t)cap_pad * 72));
    MAD_TRY(mad_reserve(ctx, s->row_meta, (size_t)cap_pad * 12));
    MAD_TRY(mad_reserve(ctx, s->dsc, (size_t)cap_pad * s->D * 2));
    MAD_TRY(mad_reserve(ctx, s->dsc8, (size_t)cap_pad * s->D));
    MAD_TRY(mad_reserve(ctx, s->norm, (size_t)cap_pad * 8));
    MAD_TRY(mad_reserve(ctx, s->row_perm, (size_t)cap_pad * 4));
    MAD_TRY(mad_reserve(ctx, s->row_rec, (size_t)cap_pad * sizeof(DscRowRec)));
    return MAD_OK;
}

// for rows that came from the host (mad_set_load): int8 rows + norms with the range check, inverse rotations,
// result meta.  Rows built on the device get all of that from k_orient_rows / k_describe directly.
static int set_finish_rows(mad_ctx *ctx, mad_set *s) {
    int32_t *d_n = (int32_t *)s->dev_n.p;
    int32_t *bad = d_n + 1;      // zeroed with the other counters by the anchor upload
    const int64_t cap_pad = mad_ceil_div(s->cap_rows > 0 ? s->cap_rows : 1, GEMM_BM) * GEMM_BM;
    hipLaunchKernelGGL(k_pack_rows, dim3((unsigned)std::min<int64_t>(mad_ceil_div(cap_pad, 4), (int64_t)ctx->n_cu * 8)), dim3(256), 0,
                       ctx->stream, (const int16_t *)s->dsc.p, d_n, s->D, (int8_t *)s->dsc8.p, (double *)s->norm.p, bad);
    hipLaunchKernelGGL(k_row_aux, dim3((unsigned)std::min<int64_t>(mad_ceil_div(cap_pad, 256), 1024)), dim3(256), 0, ctx->stream,
                       (const double *)s->row_R.p, d_n, (double *)s->row_Rinv.p, (const int32_t *)s->row_anchor.p,
                       (const int32_t *)s->row_main.p, (const int32_t *)s->anc_index.p, (const int32_t *)s->anc_octave.p,
                       (int32_t *)s->row_meta.p);
    MAD_HIP(hipGetLastError());
    s->n_rows_host = -1;
    return MAD_OK;
}

// Orientation + description of the anchors of n_sets structures, each into its own set, with ONE launch per stage for all of
// them (k_orient, scan, row expansion, k_describe): the structures of a step fill the chip together.  Everything is enqueued
// on the lane of the first set; every set's `built` event is recorded behind the last launch.
extern "C" int mad_set_build_many(mad_ctx *ctx, int n_sets, mad_set *const *sets, const int *slot_of_octave,
                                  const int32_t *const *anc_coords, const int32_t *const *anc_octave, const double *const *anc_subv,
                                  const int32_t *const *anc_index, const int *n_anchors, int r, int lim_main, int lim_sec) {
    if (!ctx || n_sets < 0 || (n_sets > 0 && (!sets || !slot_of_octave || !anc_coords || !anc_octave || !anc_subv || !anc_index || !n_anchors)))
        return MAD_EINVAL;
    if (n_sets == 0) return MAD_OK;
    for (int i = 0; i < n_sets; i++) {
        if (!sets[i]) return mad_fail(ctx, MAD_EINVAL, "mad_set_build: set %d is NULL", i);
        for (int k = 0; k < i; k++)
            if (sets[k] == sets[i]) return mad_fail(ctx, MAD_EINVAL, "mad_set_build: set %d is listed twice", i);
    }
    mad_use_lane(ctx, sets[0]->lane);
    if (lim_main < 1 || lim_sec < 1 || lim_main * lim_sec > 64) return mad_fail(ctx, MAD_EINVAL, "mad_set_build: lim_main=%d lim_sec=%d", lim_main, lim_sec);
    for (int i = 1; i < n_sets; i++)      // an export still reading a set on its own lane
        if (sets[i]->lane != sets[0]->lane) MAD_HIP(hipStreamWaitEvent(ctx->stream, sets[i]->ready, 0));
    std::vector<OrientJob> oj(n_sets);
    std::vector<DescribeJob> dj(n_sets);
    for (int i = 0; i < n_sets; i++) {
        const int n = n_anchors[i];
        if (n > 0 && (!anc_coords[i] || !anc_octave[i] || !anc_subv[i] || !anc_index[i])) return mad_fail(ctx, MAD_EINVAL, "mad_set_build: NULL anchors");
        FieldDev *f = oj[i].f;
        f[0] = f[1] = FieldDev{nullptr, 0, 0, 0};
        for (int o = 0; o < 2; o++) {
            const int sl = slot_of_octave[2 * i + o];
            if (sl >= 0) {
                if (sl >= MAD_MAX_FIELDS || !ctx->fields[sl].tex) return mad_fail(ctx, MAD_EINVAL, "mad_set_build: field slot %d is empty", sl);
                f[o] = ctx->fields[sl];
            }
        }
        for (int a = 0; a < n; a++) {
            const int o = anc_octave[i][a];
            if ((o != 0 && o != 1) || !f[o].tex) return mad_fail(ctx, MAD_EINVAL, "mad_set_build: anchor %d has octave %d without a field", a, o);
        }
    }
    for (int i = 0; i < n_sets; i++) {
        mad_set *s = sets[i];
        const int n = n_anchors[i];
        // (r = 8, the default patch: the only size k_describe_ball is built for)
        const int bdims[3] = {oj[i].f[1].nx, oj[i].f[1].ny, oj[i].f[1].nz};
        const bool sort_ball = ctx->dsc_ball && r == 8 && oj[i].f[1].tex4 != nullptr && ctx->spatial_order;
        MAD_TRY(set_upload_anchors(ctx, s, anc_coords[i], anc_octave[i], anc_subv[i], anc_index[i], n, 0, sort_ball ? bdims : nullptr));
        s->D = 64 * ctx->eq_host[1].Z;
        MAD_TRY(set_reserve_rows(ctx, s, (int64_t)n * lim_main * lim_sec));
        MAD_TRY(mad_reserve(ctx, s->anc_rows, (size_t)(n > 0 ? n : 1) * MAD_ANCROW_WORDS * 4));
        s->last_fan = lim_main * lim_sec;
        OrientJob &J = oj[i];
        J.d_coords = (const int32_t *)s->anc_coords.p; J.d_octave = (const int32_t *)s->anc_octave.p; J.uniform_octave = 0; J.n = n;
        OrientOut &out = J.out;
        out.row_anchor = (int32_t *)s->row_anchor.p; out.row_main = (int32_t *)s->row_main.p; out.row_sec = (int32_t *)s->row_sec.p;
        out.row_R = (double *)s->row_R.p; out.row_count = nullptr;
        out.d_n_rows = (int32_t *)s->dev_n.p; out.d_n_reject = (int32_t *)s->dev_n.p + 2;
        out.row_Rinv = (double *)s->row_Rinv.p; out.row_meta = (int32_t *)s->row_meta.p;
        out.anc_index = (const int32_t *)s->anc_index.p; out.anc_octave = (const int32_t *)s->anc_octave.p;
        out.anc_order = ctx->spatial_order ? (const int32_t *)s->anc_order.p : nullptr;
        out.row_perm = ctx->spatial_order ? (int32_t *)s->row_perm.p : nullptr;
        out.row_rec = (DscRowRec *)s->row_rec.p;
        out.anc_rows = out.anc_order && sort_ball ? (int32_t *)s->anc_rows.p : nullptr;
        out.counters_zeroed = true;
        DescribeJob &Q = dj[i];
        Q.f[0] = J.f[0]; Q.f[1] = J.f[1];
        Q.d_anc_coords = J.d_coords; Q.d_anc_octave = J.d_octave; Q.uniform_octave = 0;
        Q.d_row_anchor = out.row_anchor; Q.d_row_R = out.row_R; Q.d_row_Rinv = out.row_Rinv; Q.d_row_perm = out.row_perm; Q.d_n_rows = out.d_n_rows;
        Q.d_row_rec = out.row_rec;
        // the anchors sorted behind n_rowwise go through k_describe_ball (when the sort was made for this base-octave grid)
        if (out.anc_rows && s->ball_dims[0] == bdims[0] && s->ball_dims[1] == bdims[1] && s->ball_dims[2] == bdims[2] && s->ball_dims[0] > 0) {
            Q.d_anc_rows = out.anc_rows; Q.n_anchors = n; Q.n_rowwise = s->n_rowwise; Q.fan = lim_main * lim_sec;
        }
        s->last_perm = out.row_perm != nullptr;
        s->last_rec = out.row_rec != nullptr;
        // the describe launch is sized from the row count of this set's previous build when there is one
        Q.grid_rows = n <= 0 ? 0 : (s->rows_hint > 0 ? std::min<int64_t>(s->cap_rows, s->rows_hint + s->rows_hint / 8 + 64) : s->cap_rows);
        Q.d_overflow = (int32_t *)s->dev_n.p + 3;
        Q.d_dsc = (int16_t *)s->dsc.p; Q.d_dsc8 = (int8_t *)s->dsc8.p; Q.d_norm = (double *)s->norm.p;      // int8 copy + norms included: counts are <= 64 by construction
        s->last_f[0] = J.f[0]; s->last_f[1] = J.f[1]; s->last_r = r;
        s->n_rows_host = -1;
    }
    MAD_TRY(mad_orient_device_many(ctx, n_sets, oj.data(), r, lim_main, lim_sec));
    MAD_TRY(mad_describe_device_many(ctx, n_sets, dj.data(), r));
    for (int i = 0; i < n_sets; i++) {
        MAD_HIP(hipEventRecord(sets[i]->built, ctx->stream));
    }
    return MAD_OK;
}

extern "C" int mad_set_build(mad_ctx *ctx, mad_set *s, const int *slot_of_octave, const int32_t *anc_coords,
                             const int32_t *anc_octave, const double *anc_subv, const int32_t *anc_index, int n, int r,
                             int lim_main, int lim_sec) {
    if (!ctx || !s || !slot_of_octave) return MAD_EINVAL;
    return mad_set_build_many(ctx, 1, &s, slot_of_octave, &anc_coords, &anc_octave, &anc_subv, &anc_index, &n, r, lim_main, lim_sec);
}

extern "C" int mad_set_load(mad_ctx *ctx, mad_set *s, int64_t n_rows, const int32_t *row_anchor, const int32_t *row_main,
                            const double *row_R, const int16_t *dsc, int D, const double *anc_subv, const int32_t *anc_index,
                            const int32_t *anc_octave, int n_anchors) {
    if (!ctx || !s) return MAD_EINVAL;
    mad_use_lane(ctx, s->lane);
    if (n_rows > 0 && (!row_anchor || !row_main || !row_R || !dsc)) return mad_fail(ctx, MAD_EINVAL, "mad_set_load: NULL rows");
    if (n_anchors > 0 && (!anc_subv || !anc_index || !anc_octave)) return mad_fail(ctx, MAD_EINVAL, "mad_set_load: NULL anchors");
    for (int64_t i = 0; i < n_rows; i++)
        if (row_anchor[i] < 0 || row_anchor[i] >= n_anchors) return mad_fail(ctx, MAD_EINVAL, "mad_set_load: row %lld -> anchor %d", (long long)i, row_anchor[i]);
    MAD_TRY(set_upload_anchors(ctx, s, nullptr, anc_octave, anc_subv, anc_index, n_anchors, (int32_t)n_rows));
    s->D = D;
    MAD_TRY(set_reserve_rows(ctx, s, n_rows));
    if (n_rows > 0) {
        MAD_HIP(hipMemcpyAsync(s->row_anchor.p, row_anchor, n_rows * 4, hipMemcpyHostToDevice, ctx->stream));
        MAD_HIP(hipMemcpyAsync(s->row_main.p, row_main, n_rows * 4, hipMemcpyHostToDevice, ctx->stream));
        MAD_HIP(hipMemsetAsync(s->row_sec.p, 0, n_rows * 4, ctx->stream));
        MAD_HIP(hipMemcpyAsync(s->row_R.p, row_R, n_rows * 72, hipMemcpyHostToDevice, ctx->stream));
        MAD_HIP(hipMemcpyAsync(s->dsc.p, dsc, (size_t)n_rows * D * 2, hipMemcpyHostToDevice, ctx->stream));
    }
    MAD_TRY(set_finish_rows(ctx, s));
    MAD_HIP(hipEventRecord(s->built, ctx->stream));
    int64_t n_dev = 0;
    MAD_TRY(set_rows(ctx, s, &n_dev));      // synchronises (the host arrays may go away) and fetches the range check
    if (s->range_bad) return mad_fail(ctx, MAD_EDOM, "descriptor count outside the int8 range");
    return MAD_OK;
}

extern "C" int mad_set_size(mad_ctx *ctx, const mad_set *s, int64_t *n_rows, int32_t *n_anchors) {
    if (!ctx || !s) return MAD_EINVAL;
    int64_t n = 0;
    MAD_TRY(set_rows(ctx, s, &n));
    if (n_rows) *n_rows = n;
    if (n_anchors) *n_anchors = s->n_anchors;
    return MAD_OK;
}

extern "C" int mad_set_download(mad_ctx *ctx, const mad_set *s, int32_t *row_anchor, int32_t *row_main, int32_t *row_sec,
                                double *row_R, int16_t *dsc) {
    if (!ctx || !s) return MAD_EINVAL;
    mad_use_lane(ctx, s->lane);
    int64_t n = 0;
    MAD_TRY(set_rows(ctx, s, &n));      // waits for the build, on whichever lane it ran
    if (n <= 0) return MAD_OK;
    if (row_anchor) MAD_HIP(hipMemcpyAsync(row_anchor, s->row_anchor.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (row_main) MAD_HIP(hipMemcpyAsync(row_main, s->row_main.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (row_sec) MAD_HIP(hipMemcpyAsync(row_sec, s->row_sec.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (row_R) MAD_HIP(hipMemcpyAsync(row_R, s->row_R.p, n * 72, hipMemcpyDeviceToHost, ctx->stream));
    if (dsc) MAD_HIP(hipMemcpyAsync(dsc, s->dsc.p, (size_t)n * s->D * 2, hipMemcpyDeviceToHost, ctx->stream));
    MAD_HIP(hipStreamSynchronize(ctx->stream));
    return MAD_OK;
}

static Side side_of(const mad_set *s) {
    Side x;
    x.dsc8 = (const int8_t *)s->dsc8.p; x.norm = (const double *)s->norm.p; x.R = (const double *)s->row_R.p;
    x.Rinv = (const double *)s->row_Rinv.p; x.meta = (const int32_t *)s->row_meta.p; x.row_anchor = (const int32_t *)s->row_anchor.p;
    x.p = (const double *)s->anc_subv.p; x.n_rows = (const int32_t *)s->dev_n.p; x.cap_rows = s->cap_rows;
    x.anc_canon = (const int32_t *)s->anc_canon.p;
    return x;
}

// capacity state of one match attempt
struct MatchPlan {
    int64_t cap_c, cap_pairs, k;
    bool fits;
    CellGrid G;
    double dist;
    bool no_small;      // the one-workgroup top-k over the pruned selection overflowed: use the general selection
    bool pruned;        // this attempt's pose search was pruned by bounds (set when it is enqueued; per bracket and lane, not per lane)
    PosePlan pose;      // made by match_enqueue_head (which also zeroes the bitmaps), used by match_enqueue_tail
};

// enqueue a11 + a12 + top-k (+ the result rows) of one (hi, lo) pair in the CURRENT lane; no host round trip.
// layout of the zero region of a match: [status ST_COUNT int32][hist (n_hi_anchors + 17) int32: match counts, top-k selection]
// [hist2 (n_hi_anchors + 17) int32: lower bounds, pose pruning][used flags of the hi anchors, padded to 32][of the lo anchors]
static int32_t *zero_status(mad_ctx *ctx) { return scratch<int32_t>(ctx, S_ZERO); }
static inline int32_t *zr_hist(int32_t *st) { return st + ST_COUNT; }
static inline int32_t *zr_hist2(int32_t *st, int n_hi) { return st + ST_COUNT + n_hi + 17; }
static inline uint8_t *zr_used_hi(int32_t *st, int n_hi) { return (uint8_t *)(st + ST_COUNT + 2 * (n_hi + 17)); }
static inline uint8_t *zr_used_lo(int32_t *st, int n_hi) { return zr_used_hi(st, n_hi) + ((n_hi + 31) & ~31); }
static size_t zero_bytes(const mad_set *hi, const mad_set *lo) {
    return ((size_t)(ST_COUNT + 2 * (hi->n_anchors + 17)) * 4 + (size_t)hi->n_anchors + lo->n_anchors + 64 + 15) / 16 * 16;
}
static size_t tail_bytes(int64_t k) { return (size_t)k * (MAD_RESULT_COLS * 8 + 8) + ST_COUNT * 4; }

// A match is enqueued in two halves: what precedes the GEMM (waits, zeroed status, the GEMM's arguments) and what follows it.
// Between them the caller launches the GEMM -- of this match alone, or of all matches of a bracket in one grid.
static int match_enqueue_head(mad_ctx *ctx, const mad_set *hi, const mad_set *lo, double cc, MatchPlan &P, GemmJob *job) {
    int32_t *st = zero_status(ctx);
    const Side H = side_of(hi), L = side_of(lo);
    // the sets may have been built on other lanes
    MAD_HIP(hipStreamWaitEvent(ctx->stream, hi->built, 0));
    MAD_HIP(hipStreamWaitEvent(ctx->stream, lo->built, 0));
    // status, histograms and flags, and the occupancy bitmaps of the pose search, zeroed by ONE launch
    pose_plan(ctx, hi->n_anchors, lo->n_anchors, lo->bb_min, lo->bb_max, !P.fits, P.dist, P.k, true, &P.pose);
    const size_t bits_bytes = P.pose.fine_bytes + P.pose.coarse_bytes;
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_PG_BITS), bits_bytes + 16));
    MAD_TRY(correlate_reserve(ctx, H, L, hi->D, cc, st, P.cap_c, P.cap_pairs, job));
    mad_zero_words3(ctx, st, zero_bytes(hi, lo), mad_sb(ctx, S_PG_BITS).p, bits_bytes, job->cflag, cflag_bytes(P.cap_c));
    return MAD_OK;
}

static int match_enqueue_tail(mad_ctx *ctx, const mad_set *hi, const mad_set *lo, double cc, double dist, MatchPlan &P) {
    int32_t *st = zero_status(ctx);
    int32_t *hist = zr_hist(st);
    uint8_t *used_hi = zr_used_hi(st, hi->n_anchors), *used_lo = zr_used_lo(st, hi->n_anchors);
    const Side H = side_of(hi), L = side_of(lo);
    MAD_TRY(correlate_pairs(ctx, H, L, cc, st, P.cap_pairs, used_hi, used_lo));
    // clouds: anchors that take part in at least one pair (MaD.py:427-428)
    const CloudJob job = {(const double *)hi->anc_subv.p, used_hi, hi->n_anchors, scratch<double>(ctx, S_HI_CLOUD), st + ST_LHI,
                          (const int32_t *)hi->dev_n.p, (const int32_t *)lo->dev_n.p, st};      // compacted by the first pose kernel
    CellGrid G = P.G;
    G.used = used_lo;
    bool pruned = false;      // the caller gets k rows: pairs that cannot be among them need no exact count
    MAD_TRY(pose_device(ctx, H, L, st, P.cap_pairs, scratch<double>(ctx, S_HI_CLOUD), hi->n_anchors, (const double *)lo->anc_subv.p,
                        lo->n_anchors, used_lo, lo->bb_min, lo->bb_max, P.fits ? nullptr : &G, dist, P.k, zr_hist2(st, hi->n_anchors), &pruned, &job,
                        &P.pose, true));
    P.pruned = pruned;
    // A pruned search has listed every pair that can be among the k best (all others lie strictly below the k-th count): when the
    // previous match of this lane listed few enough, the k best are taken from that list by ONE workgroup instead of four
    // launches over all pairs.  Should the list outgrow the kernel (ST_FLAG_SEL), the match is repeated with the general selection.
    const int64_t hint = ctx->lane_sel_hint[ctx->lane];
    const ResultArgs RA = {scratch<int32_t>(ctx, S_PAIR_HI), scratch<int32_t>(ctx, S_PAIR_LO), scratch<double>(ctx, S_PAIR_SCORE),
                           scratch<int32_t>(ctx, S_COUNTS), H.p, H.R, H.meta, L.p, L.Rinv, L.meta, H.row_anchor, L.row_anchor,
                           (double *)ctx->host_res[ctx->res_slot][ctx->res_idx], P.k};
    if (pruned && !P.no_small && hint > 0 && hint <= TKS_CAP / 2 && P.k <= TKS_CAP) {
        mad_timer_begin(ctx, MAD_T_TOPK);
        static bool attr_t = false;
        if (!attr_t) {
            MAD_HIP(hipFuncSetAttribute((const void *)k_topk_selected, hipFuncAttributeMaxDynamicSharedMemorySize, TKS_CAP * 8));
            attr_t = true;
        }
        hipLaunchKernelGGL(k_topk_selected, dim3(1), dim3(1024), (size_t)TKS_CAP * 8, ctx->stream, scratch<int32_t>(ctx, S_COUNTS),
                           scratch<int32_t>(ctx, S_TMP_D), st, P.k, hi->n_anchors, scratch<int64_t>(ctx, S_SEL_OUT), RA);
        mad_timer_end(ctx, MAD_T_TOPK);
    } else {
        MAD_TRY(topk_device(ctx, scratch<int32_t>(ctx, S_COUNTS), st, P.cap_pairs, P.k, hi->n_anchors, scratch<int64_t>(ctx, S_SEL_OUT), hist));
        hipLaunchKernelGGL(k_results, dim3((unsigned)mad_ceil_div(P.k, 256)), dim3(256), 0, ctx->stream,
                           scratch<int64_t>(ctx, S_SEL_OUT), st + ST_NKEYS, P.k, scratch<int32_t>(ctx, S_PAIR_HI),
                           scratch<int32_t>(ctx, S_PAIR_LO), scratch<double>(ctx, S_PAIR_SCORE), scratch<int32_t>(ctx, S_COUNTS),
                           st, H.p, H.R, H.meta, L.p, L.Rinv, L.meta, H.row_anchor, L.row_anchor, RA.out, 1);
    }
    MAD_HIP(hipGetLastError());
    // rows, pair ranks and status words were written by the kernel straight into the pinned staging of this (bracket, lane): no copy
    // engine, no blit kernel; the host reads them once the event has passed
    MAD_HIP(hipEventRecord(ctx->lane_done[ctx->res_slot][ctx->res_idx], ctx->stream));
    return MAD_OK;
}

static int match_enqueue(mad_ctx *ctx, const mad_set *hi, const mad_set *lo, double cc, double dist, MatchPlan &P) {
    GemmJob job;
    MAD_TRY(match_enqueue_head(ctx, hi, lo, cc, P, &job));
    MAD_TRY(correlate_gemm(ctx, 1, &job, hi->D, cc));
    return match_enqueue_tail(ctx, hi, lo, cc, dist, P);
}

static int match_prepare(mad_ctx *ctx, const mad_set *hi, const mad_set *lo, double dist, int64_t k, MatchPlan *P) {
    if (hi->D != lo->D) return mad_fail(ctx, MAD_EINVAL, "mad_match_topk: descriptor lengths %d vs %d", hi->D, lo->D);
    if (!(dist > 0)) return mad_fail(ctx, MAD_EINVAL, "mad_match_topk: dist must be positive");
    P->k = k < 1 ? 1 : k;
    P->dist = dist;
    P->no_small = false;
    P->pruned = false;
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_ZERO), zero_bytes(hi, lo)));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_HI_CLOUD), (size_t)hi->n_anchors * 24 + 24));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_SEL_OUT), (size_t)(P->k + 8) * 8));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_RESULTS), tail_bytes(P->k) + 64));
    if (ctx->host_res_cap[ctx->res_slot][ctx->res_idx] < tail_bytes(P->k)) {
        if (ctx->host_res[ctx->res_slot][ctx->res_idx]) {
            MAD_HIP(hipStreamSynchronize(ctx->stream));
            (void)hipHostFree(ctx->host_res[ctx->res_slot][ctx->res_idx]);
            ctx->host_res[ctx->res_slot][ctx->res_idx] = nullptr;
        }
        const size_t want = tail_bytes(P->k) * 2;
        if (hipHostMalloc(&ctx->host_res[ctx->res_slot][ctx->res_idx], want) != hipSuccess) return mad_fail(ctx, MAD_ENOMEM, "pinned result staging of %zu bytes", want);
        ctx->host_res_cap[ctx->res_slot][ctx->res_idx] = want;
    }
    // capacity hints: the score matrix for ~8 rows per anchor, pairs for 2 % of the matrix; both grow on demand
    const int64_t full_c = (mad_ceil_div(hi->cap_rows, 128) * 128) * (mad_ceil_div(lo->cap_rows, 128) * 128);
    P->cap_c = std::max<int64_t>(ctx->match.cap_c, std::min<int64_t>((int64_t)(hi->n_anchors * 8 + 128) * (lo->n_anchors * 8 + 128), full_c));
    P->cap_pairs = std::max<int64_t>(ctx->match.cap_pairs, std::max<int64_t>(P->cap_c / 50, 1 << 16));
    // the global cell list is only needed when the clouds cannot live in LDS
    P->fits = clouds_fit_lds(hi->n_anchors, lo->n_anchors);
    if (!P->fits) {
        if (!lo->cells_ready || lo->cell_size != dist) {      // shared by every lane afterwards: finish it here
            MAD_HIP(hipStreamWaitEvent(ctx->stream, lo->built, 0));
            MAD_TRY(mad_build_cells(ctx, const_cast<mad_set *>(lo), dist));
            MAD_HIP(hipStreamSynchronize(ctx->stream));
        }
        P->G.start = (const int32_t *)lo->cell_start.p; P->G.pts = (const double *)lo->cell_pts.p; P->G.ids = (const int32_t *)lo->cell_ids.p;
        P->G.used = nullptr;
        for (int d = 0; d < 3; d++) { P->G.mn[d] = lo->cell_min[d]; P->G.dim[d] = lo->cell_dim[d]; }
        P->G.cell = lo->cell_size;
    }
    return MAD_OK;
}

// read the status of the match that ran in `lane` (already complete).  Returns 1 when it has to be repeated with
// larger capacities (updated in P), 0 when it is final (outputs filled), negative on error.
static int match_finish(mad_ctx *ctx, int ri, const mad_set *hi, const mad_set *lo, MatchPlan *P, double *results,
                        int64_t *pair_index, int64_t *n_out, int64_t *stats) {
    const int lane = ri % MAD_LANES;      // ri: the match's result index in its bracket (its lane, or MAD_LANES + lane for the lane's second match)
    const char *base = (const char *)ctx->host_res[ctx->res_slot][ri];
    const int64_t *h_idx = (const int64_t *)(base + (size_t)P->k * MAD_RESULT_COLS * 8);
    const int32_t *hs = (const int32_t *)(h_idx + P->k);
    if (hs[ST_NHI + 3] || hs[ST_NLO + 3]) {
        // a set's describe launch had been sized from a stale hint: repair the set(s), then match again
        int64_t dummy;
        const_cast<mad_set *>(hi)->n_rows_host = -1;
        const_cast<mad_set *>(lo)->n_rows_host = -1;
        MAD_TRY(set_rows(ctx, hi, &dummy));
        MAD_TRY(set_rows(ctx, lo, &dummy));
        return 1;
    }
    if (hs[ST_FLAG_C]) {
        const int64_t hp = mad_ceil_div((int64_t)hs[ST_NHI], 128) * 128, lp = mad_ceil_div((int64_t)hs[ST_NLO], 128) * 128;
        P->cap_c = hp * lp;
        if (P->cap_c >= ((int64_t)1 << 31)) return mad_fail(ctx, MAD_EINVAL, "mad_match_topk: score matrix of %lld entries", (long long)P->cap_c);
        return 1;
    }
    if (hs[ST_FLAG_PAIRS]) {
        P->cap_pairs = (int64_t)hs[ST_NPAIRS] + hs[ST_NPAIRS] / 8 + 1024;
        return 1;
    }
    if (hs[ST_FLAG_SEL]) {
        P->no_small = true;
        ctx->lane_sel_hint[lane] = 0;
        return 1;
    }
    ctx->match.cap_c = std::max(ctx->match.cap_c, P->cap_c);
    ctx->match.cap_pairs = std::max(ctx->match.cap_pairs, P->cap_pairs);
    ctx->match.n_pairs = hs[ST_NPAIRS];
    ctx->match.l_hi = hs[ST_LHI];
    ctx->match.l_lo = hs[ST_LLO];
    ctx->match.lane = lane;
    ctx->match.n_hi_anchors = hi->n_anchors;
    ctx->match.n_lo_anchors = lo->n_anchors;
    ctx->match.pruned = P->pruned;
    ctx->match.n_sel = ctx->match.pruned ? hs[ST_NSEL] : hs[ST_NPAIRS];
    ctx->lane_sel_hint[lane] = ctx->match.pruned ? std::max<int64_t>(hs[ST_NSEL], 1) : 0;
    ctx->match.hi = hi; ctx->match.lo = lo; ctx->match.hi_gen = hi->gen; ctx->match.lo_gen = lo->gen; ctx->match.cap_pairs_used = P->cap_pairs; ctx->match.fits = P->fits; ctx->match.dist = P->dist;
    const_cast<mad_set *>(hi)->n_rows_host = hs[ST_NHI];
    const_cast<mad_set *>(lo)->n_rows_host = hs[ST_NLO];
    const_cast<mad_set *>(hi)->rows_hint = hs[ST_NHI];
    const_cast<mad_set *>(lo)->rows_hint = hs[ST_NLO];
    if (stats) { stats[0] = hs[ST_NPAIRS]; stats[1] = hs[ST_LHI]; stats[2] = hs[ST_LLO]; stats[3] = (int64_t)hs[ST_NHI] * hs[ST_NLO]; }
    const int64_t got = hs[ST_NPAIRS] > 0 ? hs[ST_NKEYS] : 0;
    if (results && got > 0) memcpy(results, base, (size_t)got * MAD_RESULT_COLS * 8);
    if (pair_index && got > 0) memcpy(pair_index, h_idx, (size_t)got * 8);
    *n_out = got;
    return 0;
}

static bool match_trivial(const mad_set *hi, const mad_set *lo) {
    return hi->cap_rows <= 0 || lo->cap_rows <= 0 || hi->n_anchors <= 0 || lo->n_anchors <= 0;
}

extern "C" int mad_match_topk(mad_ctx *ctx, const mad_set *hi, const mad_set *lo, double cc, double dist, int64_t k,
                              double *results, int64_t *pair_index, int64_t *n_out, int64_t *stats) {
    if (!ctx || !hi || !lo || !n_out) return MAD_EINVAL;
    *n_out = 0;
    if (stats) { stats[0] = 0; stats[1] = 0; stats[2] = 0; stats[3] = 0; }
    ctx->match.n_pairs = 0;
    if (match_trivial(hi, lo)) return MAD_OK;
    mad_use_lane(ctx, 0);
    ctx->res_idx = 0;
    MatchPlan P;
    MAD_TRY(match_prepare(ctx, hi, lo, dist, k, &P));
    for (int attempt = 0; attempt < 5; attempt++) {
        MAD_TRY(match_enqueue(ctx, hi, lo, cc, dist, P));
        MAD_HIP(hipStreamSynchronize(ctx->stream));      // the one host round trip of a match
        const int rc = match_finish(ctx, 0, hi, lo, &P, results, pair_index, n_out, stats);
        if (rc <= 0) return rc;
    }
    return mad_fail(ctx, MAD_EHIP, "mad_match_topk: capacity negotiation did not converge");
}

// Several subunits against one map.  Match i runs in the lane of its hi set, so up to MAD_LANES matches are in
// flight before the host waits for the oldest: the GPU never idles on a result read-back.
// results: n x k x 23, pair_index (nullable): n x k, n_out: n, stats (nullable): n x 4.
// State of one mad_match_topk_many_begin .. _finish bracket (ctx->many).
struct ManyState {
    int n = 0;
    std::vector<const mad_set *> hi;
    const mad_set *lo = nullptr;
    double cc = 0, dist = 0;
    int64_t k = 1;
    double *results = nullptr;
    int64_t *pair_index = nullptr, *n_out = nullptr, *stats = nullptr;
    MatchPlan plans[MAD_RES];      // by result index: ri = the match's lane, or MAD_LANES + lane for the second match of a lane
    int pending[MAD_RES];
    int slot = 0;      // result slot (pinned staging + completion events) of this bracket
};

static int many_retire(mad_ctx *ctx, ManyState &M, int ri) {
    const int i = M.pending[ri];
    if (i < 0) return MAD_OK;
    const int lane = ri % MAD_LANES;
    M.pending[ri] = -1;
    ctx->res_slot = M.slot;
    ctx->res_idx = ri;
    MAD_HIP(hipEventSynchronize(ctx->lane_done[M.slot][ri]));
    double *res_i = M.results ? M.results + (size_t)i * M.k * MAD_RESULT_COLS : nullptr;
    int64_t *idx_i = M.pair_index ? M.pair_index + (size_t)i * M.k : nullptr;
    int64_t *st_i = M.stats ? M.stats + 4 * i : nullptr;
    int rc = match_finish(ctx, ri, M.hi[i], M.lo, &M.plans[ri], res_i, idx_i, &M.n_out[i], st_i);
    for (int attempt = 0; rc == 1 && attempt < 5; attempt++) {      // rare: repeat this one synchronously
        mad_use_lane(ctx, lane);
        MAD_TRY(match_enqueue(ctx, M.hi[i], M.lo, M.cc, M.dist, M.plans[ri]));
        MAD_HIP(hipStreamSynchronize(ctx->stream));
        rc = match_finish(ctx, ri, M.hi[i], M.lo, &M.plans[ri], res_i, idx_i, &M.n_out[i], st_i);
    }
    if (rc == 1) return mad_fail(ctx, MAD_EHIP, "mad_match_topk_many: capacity negotiation did not converge");
    return rc;
}

extern "C" int mad_match_topk_many_begin(mad_ctx *ctx, int n, const mad_set *const *hi, const mad_set *lo, double cc, double dist, int64_t k,
                                         double *results, int64_t *pair_index, int64_t *n_out, int64_t *stats) {
    if (!ctx || !hi || !lo || !n_out || n < 0) return MAD_EINVAL;
    if (ctx->many_open >= MAD_BRACKETS) return mad_fail(ctx, MAD_EINVAL, "mad_match_topk_many_begin: %d brackets are open already", MAD_BRACKETS);
    if (k < 1) k = 1;
    ManyState *Mp = new ManyState();
    ManyState &M = *Mp;
    M.slot = (ctx->many_oldest + ctx->many_open) % MAD_BRACKETS;      // the slots form a ring: brackets finish in the order they began
    ctx->many_open++;
    ctx->res_slot = M.slot;
    M.n = n; M.hi.assign(hi, hi + n); M.lo = lo; M.cc = cc; M.dist = dist; M.k = k;
    M.results = results; M.pair_index = pair_index; M.n_out = n_out; M.stats = stats;
    for (int l = 0; l < MAD_RES; l++) M.pending[l] = -1;
    ctx->many[M.slot] = Mp;
    int rc_all = MAD_OK;
    // The first match of every lane goes out in a batch: each lane enqueues what precedes its GEMM, ONE grid then computes the
    // score tiles of all of them (they share the lo rows, and together their tiles fill the chip's workgroup slots evenly), and
    // each lane continues behind it with its own pairs, poses and top-k.  Matches that find their lane taken wait their turn below.
    // Off unless asked for (mad_set_batching): with the lanes overlapped, four separate GEMM launches fill each other's tails and
    // leave every match free to start as soon as its own sets are ready -- measured on C3: one batched launch 0.106 ms of device
    // time per step against 0.151 for four, but 1.21 ms per overlapped step against 1.07.
    const bool no_batch = !ctx->batch_gemm;
    std::vector<int> batch;
    std::vector<GemmJob> jobs;
    bool taken[MAD_LANES] = {};
    for (int i = 0; i < n && rc_all == MAD_OK; i++) {
        n_out[i] = 0;
        if (stats) { stats[4 * i] = stats[4 * i + 1] = stats[4 * i + 2] = stats[4 * i + 3] = 0; }
        if (!hi[i]) { rc_all = mad_fail(ctx, MAD_EINVAL, "mad_match_topk_many: set %d is NULL", i); break; }
        if (match_trivial(hi[i], lo) || no_batch) continue;
        const int lane = hi[i]->lane;      // where its hi set was built: the matches spread over the lanes like the sets
        if (taken[lane] || hi[i]->D != hi[0]->D) continue;
        taken[lane] = true;
        mad_use_lane(ctx, lane);
        ctx->res_idx = lane;
        rc_all = match_prepare(ctx, hi[i], lo, dist, k, &M.plans[lane]);
        if (rc_all != MAD_OK) break;
        GemmJob job;
        rc_all = match_enqueue_head(ctx, hi[i], lo, cc, M.plans[lane], &job);
        if (rc_all != MAD_OK) break;
        MAD_HIP(hipEventRecord(ctx->lane_pre[lane], ctx->stream));
        batch.push_back(i);
        jobs.push_back(job);
    }
    if (rc_all == MAD_OK && !batch.empty()) {
        const int g = hi[batch[0]]->lane;
        mad_use_lane(ctx, g);
        for (int i : batch)
            if (hi[i]->lane != g) MAD_HIP(hipStreamWaitEvent(ctx->stream, ctx->lane_pre[hi[i]->lane], 0));
        rc_all = correlate_gemm(ctx, (int)jobs.size(), jobs.data(), hi[batch[0]]->D, cc);
        if (rc_all == MAD_OK) MAD_HIP(hipEventRecord(ctx->gemm_done[M.slot], ctx->stream));
        for (size_t b = 0; b < batch.size() && rc_all == MAD_OK; b++) {
            const int i = batch[b], lane = hi[i]->lane;
            mad_use_lane(ctx, lane);
            ctx->res_idx = lane;
            if (lane != g) MAD_HIP(hipStreamWaitEvent(ctx->stream, ctx->gemm_done[M.slot], 0));
            rc_all = match_enqueue_tail(ctx, hi[i], lo, cc, dist, M.plans[lane]);
            M.pending[lane] = i;
        }
    }
    for (int i = 0; i < n && rc_all == MAD_OK; i++) {
        if (!hi[i] || match_trivial(hi[i], lo)) continue;
        if (std::find(batch.begin(), batch.end(), i) != batch.end()) continue;
        const int lane = hi[i]->lane;
        // A lane that already carries a match of this bracket takes a second one BEHIND it on its stream, with result staging of
        // its own (more subunits than lanes -- C5: 12 -- used to make the host wait here for the first match's results); only a
        // third match of one lane in one bracket waits for the first to be collected.
        int ri = lane;
        if (M.pending[ri] >= 0) ri = MAD_LANES + lane;
        if (M.pending[ri] >= 0) {
            ri = lane;
            rc_all = many_retire(ctx, M, ri);
            if (rc_all != MAD_OK) break;
            if (M.pending[MAD_LANES + lane] >= 0) {      // (keep the lane's matches in order: the waiting one becomes its first)
                rc_all = many_retire(ctx, M, MAD_LANES + lane);
                if (rc_all != MAD_OK) break;
            }
        }
        mad_use_lane(ctx, lane);
        ctx->res_slot = M.slot;
        ctx->res_idx = ri;
        rc_all = match_prepare(ctx, hi[i], lo, dist, k, &M.plans[ri]);
        if (rc_all != MAD_OK) break;
        rc_all = match_enqueue(ctx, hi[i], lo, cc, dist, M.plans[ri]);
        M.pending[ri] = i;
    }
    mad_use_lane(ctx, ctx->match.lane);
    if (rc_all != MAD_OK) {      // leave nothing in flight behind a failed call
        for (int l = 0; l < MAD_RES; l++)
            if (M.pending[l] >= 0) { (void)hipEventSynchronize(ctx->lane_done[M.slot][l]); M.pending[l] = -1; }
        const int slot = M.slot;
        delete Mp;
        ctx->many[slot] = nullptr;
        ctx->many_open--;      // the newest bracket: the ring just shrinks again
    }
    ctx->res_slot = 0;
    ctx->res_idx = 0;
    return rc_all;
}

extern "C" int mad_set_option(mad_ctx *ctx, const char *name, double value) {
    if (!ctx || !name) return MAD_EINVAL;
    if (!strcmp(name, "pose_split_min")) {      // (MAD_POSE_SPLIT_MIN in the environment sets the initial value)
        if (!(value >= 0) || !(value < 1e12)) return mad_fail(ctx, MAD_EINVAL, "mad_set_option: pose_split_min = %g", value);
        ctx->pose_split_min = (int64_t)value;
        return MAD_OK;
    }
    if (!strcmp(name, "pose_mx")) {      // 1: the bounds pass of clouds of up to 512 points with its coarse map on the matrix cores (k_pose_bounds_mx)
        ctx->pose_mx = value != 0 ? 1 : 0;
        return MAD_OK;
    }
    if (!strcmp(name, "pose_split")) {      // -1: on for hi clouds of more than 512 points (default), 0: off, 1: on
        ctx->pose_split = value < 0 ? -1 : (value != 0 ? 1 : 0);
        return MAD_OK;
    }
    if (!strcmp(name, "ori_queue") || !strcmp(name, "dsc_queue")) {      // test hooks: a small cap drives the kernels' full-queue paths
        if (!(value >= 0) || !(value < 1e9)) return mad_fail(ctx, MAD_EINVAL, "mad_set_option: %s = %g", name, value);
        (name[0] == 'o' ? ctx->ori_queue_cap : ctx->dsc_queue_cap) = (int)value;
        return MAD_OK;
    }
    if (!strcmp(name, "dsc_ball")) {      // 1: the base-octave anchors through k_describe_ball; 0 (default): every row through k_describe
        ctx->dsc_ball = value != 0;
        return MAD_OK;
    }
    return mad_fail(ctx, MAD_EINVAL, "mad_set_option: unknown option '%s'", name);
}

extern "C" int mad_set_batching(mad_ctx *ctx, int on) {
    if (!ctx) return MAD_EINVAL;
    ctx->batch_gemm = on != 0;
    return MAD_OK;
}

extern "C" int mad_last_pose_kernel(mad_ctx *ctx) { return ctx ? ctx->last_pose_kernel : -1; }

extern "C" int64_t mad_last_pose_selected(mad_ctx *ctx) { return ctx ? ctx->match.n_sel : -1; }

extern "C" int mad_match_topk_many_finish(mad_ctx *ctx) {
    if (!ctx) return MAD_EINVAL;
    if (ctx->many_open <= 0) return mad_fail(ctx, MAD_EINVAL, "mad_match_topk_many_finish: nothing was begun");
    const int slot = ctx->many_oldest;      // the oldest of the open brackets
    ManyState *Mp = (ManyState *)ctx->many[slot];
    int rc_all = MAD_OK;
    for (int l = 0; l < MAD_RES; l++) {
        const int rc = many_retire(ctx, *Mp, l);
        if (rc_all == MAD_OK) rc_all = rc;
    }
    delete Mp;
    ctx->many[slot] = nullptr;
    ctx->many_oldest = (slot + 1) % MAD_BRACKETS;
    ctx->many_open--;
    ctx->res_slot = 0;
    ctx->res_idx = 0;
    mad_use_lane(ctx, ctx->match.lane);
    return rc_all;
}

void mad_many_abandon(mad_ctx *ctx) {      // mad_destroy: a bracket left open
    for (int r = 0; ctx && r < MAD_BRACKETS; r++)
        if (ctx->many[r]) { delete (ManyState *)ctx->many[r]; ctx->many[r] = nullptr; }
    if (ctx) ctx->many_open = 0;
}

extern "C" int mad_match_topk_many(mad_ctx *ctx, int n, const mad_set *const *hi, const mad_set *lo, double cc, double dist, int64_t k,
                                   double *results, int64_t *pair_index, int64_t *n_out, int64_t *stats) {
    MAD_TRY(mad_match_topk_many_begin(ctx, n, hi, lo, cc, dist, k, results, pair_index, n_out, stats));
    return mad_match_topk_many_finish(ctx);
}

// ---------------------------------------------------------------------------
// one subunit's pair grid sharded over ranks by blocks of lo rows (stages B and C)
// ---------------------------------------------------------------------------

static Side side_block(const mad_set *s, int64_t begin, const int32_t *d_n_rows, int64_t n) {
    Side x = side_of(s);
    x.dsc8 += begin * s->D; x.norm += begin; x.R += 9 * begin; x.Rinv += 9 * begin; x.meta += 3 * begin; x.row_anchor += begin;
    x.n_rows = d_n_rows; x.cap_rows = n;
    return x;
}

// Stage B: correlate hi against the lo rows [lo_begin, lo_end) and list the pairs above cc (kept on the device for
// mad_match_shard_topk).  used_hi / used_lo (one byte per anchor of hi / lo) receive this shard's "anchor takes part
// in a pair" flags: the caller ORs them over all shards (Exchange 1) -- the clouds and the repeatability denominator
// of MaD.py:427-428,448 are global.
extern "C" int mad_match_shard_pairs(mad_ctx *ctx, const mad_set *hi, const mad_set *lo, int64_t lo_begin, int64_t lo_end, double cc,
                                     uint8_t *used_hi, uint8_t *used_lo, int64_t *n_pairs) {
    if (!ctx || !hi || !lo || !used_hi || !used_lo || !n_pairs) return MAD_EINVAL;
    mad_use_lane(ctx, 0);
    *n_pairs = 0;
    ctx->match.shard_hi = nullptr;
    int64_t n_hi = 0, n_lo = 0;
    MAD_TRY(set_rows(ctx, hi, &n_hi));
    MAD_TRY(set_rows(ctx, lo, &n_lo));
    if (hi->D != lo->D) return mad_fail(ctx, MAD_EINVAL, "mad_match_shard_pairs: descriptor lengths %d vs %d", hi->D, lo->D);
    if (lo_begin < 0 || lo_end < lo_begin || lo_end > n_lo) return mad_fail(ctx, MAD_EINVAL, "mad_match_shard_pairs: lo rows [%lld, %lld) of %lld", (long long)lo_begin, (long long)lo_end, (long long)n_lo);
    memset(used_hi, 0, (size_t)hi->n_anchors);
    memset(used_lo, 0, (size_t)lo->n_anchors);
    const int64_t nb = lo_end - lo_begin;
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_ZERO), zero_bytes(hi, lo)));
    int32_t *st = zero_status(ctx);
    uint8_t *d_used_hi = zr_used_hi(st, hi->n_anchors), *d_used_lo = zr_used_lo(st, hi->n_anchors);
    MAD_HIP(hipStreamWaitEvent(ctx->stream, hi->built, 0));
    MAD_HIP(hipStreamWaitEvent(ctx->stream, lo->built, 0));
    int64_t cap_pairs = std::max<int64_t>(ctx->match.cap_pairs, 1 << 16);
    const int64_t cap_c = (mad_ceil_div(std::max<int64_t>(n_hi, 1), 128) * 128) * (mad_ceil_div(std::max<int64_t>(nb, 1), 128) * 128);
    if (cap_c >= ((int64_t)1 << 31)) return mad_fail(ctx, MAD_EINVAL, "mad_match_shard_pairs: score matrix of %lld entries", (long long)cap_c);
    const int32_t *hs = (const int32_t *)&ctx->pinned[0];
    for (int attempt = 0; attempt < 4 && n_hi > 0 && nb > 0; attempt++) {
        mad_zero_words(ctx, st, zero_bytes(hi, lo));
        const int32_t nb32 = (int32_t)nb;
        MAD_HIP(hipMemcpyAsync(st + ST_NLO, &nb32, 4, hipMemcpyHostToDevice, ctx->stream));
        const Side H = side_of(hi), L = side_block(lo, lo_begin, st + ST_NLO, nb);
        MAD_TRY(correlate_device(ctx, H, L, hi->D, cc, st, cap_c, cap_pairs, d_used_hi, d_used_lo));
        MAD_HIP(hipMemcpyAsync(&ctx->pinned[0], st, ST_COUNT * 4, hipMemcpyDeviceToHost, ctx->stream));
        MAD_HIP(hipStreamSynchronize(ctx->stream));
        if (hs[ST_FLAG_C]) return mad_fail(ctx, MAD_EHIP, "mad_match_shard_pairs: score matrix capacity");
        if (!hs[ST_FLAG_PAIRS]) break;
        cap_pairs = (int64_t)hs[ST_NPAIRS] + 1024;
        if (attempt == 3) return mad_fail(ctx, MAD_EHIP, "mad_match_shard_pairs: pair capacity did not converge");
    }
    const int64_t np = (n_hi > 0 && nb > 0) ? hs[ST_NPAIRS] : 0;
    if (np > 0) {
        MAD_HIP(hipMemcpyAsync(used_hi, d_used_hi, (size_t)hi->n_anchors, hipMemcpyDeviceToHost, ctx->stream));
        MAD_HIP(hipMemcpyAsync(used_lo, d_used_lo, (size_t)lo->n_anchors, hipMemcpyDeviceToHost, ctx->stream));
        MAD_HIP(hipStreamSynchronize(ctx->stream));
    }
    ctx->match.cap_pairs = std::max(ctx->match.cap_pairs, cap_pairs);
    ctx->match.shard_hi = hi; ctx->match.shard_lo = lo;
    ctx->match.shard_begin = lo_begin; ctx->match.shard_end = lo_end; ctx->match.shard_pairs = np; ctx->match.shard_cap_pairs = cap_pairs;
    *n_pairs = np;
    return MAD_OK;
}

// Stage C: score this shard's pairs against the GLOBAL clouds (used_*_all = OR over the shards) and return its k best:
// result rows (MaD.py:451), match counts and the global row-major pair rank hi_row * N_lo + lo_row -- the key that lets
// mad_amd.dist.merge_topk reproduce python's stable sort over the unsharded pair list (Exchange 2).
extern "C" int mad_match_shard_topk(mad_ctx *ctx, const mad_set *hi, const mad_set *lo, const uint8_t *used_hi_all,
                                    const uint8_t *used_lo_all, double dist, int64_t k, double *results, int64_t *pair_rank,
                                    int32_t *counts, int64_t *n_out, int64_t *l_hi) {
    if (!ctx || !hi || !lo || !used_hi_all || !used_lo_all || !results || !pair_rank || !counts || !n_out) return MAD_EINVAL;
    mad_use_lane(ctx, 0);
    *n_out = 0;
    if (l_hi) *l_hi = 0;
    if (ctx->match.shard_hi != hi || ctx->match.shard_lo != lo) return mad_fail(ctx, MAD_EINVAL, "mad_match_shard_topk: call mad_match_shard_pairs for these sets first");
    if (!(dist > 0) || k < 1) return mad_fail(ctx, MAD_EINVAL, "mad_match_shard_topk: dist %g k %lld", dist, (long long)k);
    int64_t n_lo = 0;
    MAD_TRY(set_rows(ctx, lo, &n_lo));
    const int64_t np = ctx->match.shard_pairs, begin = ctx->match.shard_begin, nb = ctx->match.shard_end - begin;
    int32_t *st = zero_status(ctx);
    int32_t *hist = zr_hist(st);
    uint8_t *d_used_hi = zr_used_hi(st, hi->n_anchors), *d_used_lo = zr_used_lo(st, hi->n_anchors);
    // the global flags replace the shard's own; the histogram of the top-k selection starts from zero
    MAD_HIP(hipMemcpyAsync(d_used_hi, used_hi_all, (size_t)hi->n_anchors, hipMemcpyHostToDevice, ctx->stream));
    MAD_HIP(hipMemcpyAsync(d_used_lo, used_lo_all, (size_t)lo->n_anchors, hipMemcpyHostToDevice, ctx->stream));
    MAD_HIP(hipMemsetAsync(hist, 0, (size_t)(hi->n_anchors + 17) * 8, ctx->stream));      // hist and hist2
    MAD_HIP(hipMemsetAsync(st + ST_NKEYS, 0, 4, ctx->stream));
    MAD_HIP(hipMemsetAsync(st + ST_NSEL, 0, 4, ctx->stream));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_HI_CLOUD), (size_t)hi->n_anchors * 24 + 24));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_SEL_OUT), (size_t)(k + 8) * 8));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_RESULTS), (size_t)(k + 1) * MAD_RESULT_COLS * 8));
    hipLaunchKernelGGL(k_compact_cloud, dim3(1), dim3(1024), 0, ctx->stream,
                       CloudJob{(const double *)hi->anc_subv.p, d_used_hi, hi->n_anchors, scratch<double>(ctx, S_HI_CLOUD), st + ST_LHI, nullptr, nullptr, st});
    MAD_HIP(hipGetLastError());
    if (np > 0) {
        const Side H = side_of(hi), L = side_block(lo, begin, st + ST_NLO, nb);
        const int64_t cap_pairs = ctx->match.shard_cap_pairs;
        const bool fits = clouds_fit_lds(hi->n_anchors, lo->n_anchors);
        CellGrid G;
        if (!fits) {
            if (!lo->cells_ready || lo->cell_size != dist) {
                MAD_TRY(mad_build_cells(ctx, const_cast<mad_set *>(lo), dist));
                MAD_HIP(hipStreamSynchronize(ctx->stream));
            }
            G.start = (const int32_t *)lo->cell_start.p; G.pts = (const double *)lo->cell_pts.p; G.ids = (const int32_t *)lo->cell_ids.p;
            for (int d = 0; d < 3; d++) { G.mn[d] = lo->cell_min[d]; G.dim[d] = lo->cell_dim[d]; }
            G.cell = lo->cell_size;
            G.used = d_used_lo;
        }
        MAD_TRY(pose_device(ctx, H, L, st, cap_pairs, scratch<double>(ctx, S_HI_CLOUD), hi->n_anchors, (const double *)lo->anc_subv.p,
                            lo->n_anchors, d_used_lo, lo->bb_min, lo->bb_max, fits ? nullptr : &G, dist, k, zr_hist2(st, hi->n_anchors)));
        MAD_TRY(topk_device(ctx, scratch<int32_t>(ctx, S_COUNTS), st, cap_pairs, k, hi->n_anchors, scratch<int64_t>(ctx, S_SEL_OUT), hist));
        hipLaunchKernelGGL(k_results, dim3((unsigned)mad_ceil_div(k, 256)), dim3(256), 0, ctx->stream, scratch<int64_t>(ctx, S_SEL_OUT),
                           st + ST_NKEYS, k, scratch<int32_t>(ctx, S_PAIR_HI), scratch<int32_t>(ctx, S_PAIR_LO),
                           scratch<double>(ctx, S_PAIR_SCORE), scratch<int32_t>(ctx, S_COUNTS), st, H.p, H.R, H.meta, L.p, L.Rinv, L.meta,
                           H.row_anchor, L.row_anchor, scratch<double>(ctx, S_RESULTS), 0);
        MAD_HIP(hipGetLastError());
    }
    MAD_HIP(hipMemcpyAsync(&ctx->pinned[0], st, ST_COUNT * 4, hipMemcpyDeviceToHost, ctx->stream));
    MAD_HIP(hipStreamSynchronize(ctx->stream));
    const int32_t *hs = (const int32_t *)&ctx->pinned[0];
    if (l_hi) *l_hi = hs[ST_LHI];
    const int64_t got = np > 0 ? std::min<int64_t>(hs[ST_NKEYS], k) : 0;
    if (got > 0) {
        std::vector<int64_t> sel((size_t)got);
        std::vector<int32_t> ph((size_t)np), pl((size_t)np), cn((size_t)np);
        MAD_HIP(hipMemcpyAsync(sel.data(), mad_sb(ctx, S_SEL_OUT).p, (size_t)got * 8, hipMemcpyDeviceToHost, ctx->stream));
        MAD_HIP(hipMemcpyAsync(ph.data(), mad_sb(ctx, S_PAIR_HI).p, (size_t)np * 4, hipMemcpyDeviceToHost, ctx->stream));
        MAD_HIP(hipMemcpyAsync(pl.data(), mad_sb(ctx, S_PAIR_LO).p, (size_t)np * 4, hipMemcpyDeviceToHost, ctx->stream));
        MAD_HIP(hipMemcpyAsync(cn.data(), mad_sb(ctx, S_COUNTS).p, (size_t)np * 4, hipMemcpyDeviceToHost, ctx->stream));
        MAD_HIP(hipMemcpyAsync(results, mad_sb(ctx, S_RESULTS).p, (size_t)got * MAD_RESULT_COLS * 8, hipMemcpyDeviceToHost, ctx->stream));
        MAD_HIP(hipStreamSynchronize(ctx->stream));
        for (int64_t t = 0; t < got; t++) {
            const int64_t p = sel[(size_t)t];
            counts[t] = cn[(size_t)p];
            pair_rank[t] = (int64_t)ph[(size_t)p] * n_lo + (begin + pl[(size_t)p]);
        }
    }
    *n_out = got;
    return MAD_OK;
}

// ---------------------------------------------------------------------------
// The same two stages without a host round trip (round 4): everything is enqueued on the lane of the hi set, the flags and the
// shard's list stay in DEVICE memory of the caller (torch tensors: mad_amd/dist.py orders its two collectives -- an OR all-reduce,
// an all-gather -- on that lane's stream between and behind the two calls, the ExternalStream pattern of ShardedSetBuild), and the
// host reads one small record per shard when the step is collected.  What a host decision settled in the synchronous form is a
// flag in that record instead: a pair list or score matrix beyond its capacity hint, a lo set whose row count is not the one the
// block bounds were computed from.  A flagged shard is repeated by the caller through the synchronous calls above.
// ---------------------------------------------------------------------------

#define ST_SHARD_NLO 17      // status word: the lo set has another number of rows than the caller assumed when it cut the blocks

__global__ void k_shard_head(int32_t *st, int32_t nb, const int32_t *lo_n_rows, int32_t n_lo_assumed) {
    if (threadIdx.x == 0) {
        st[ST_NLO] = nb;
        if (*lo_n_rows != n_lo_assumed) st[ST_SHARD_NLO] = 1;
    }
}

// out: [0] rows m, [1] flags (1 score matrix capacity, 2 pair capacity, 4 lo row count, 8 selection list), [2] size of the global hi
// cloud, [3] pairs of the shard, then k x 23 result rows, k match counts, k global pair ranks (hi_row * N_lo + lo_row), all float64
__global__ void k_shard_pack(const int64_t *__restrict__ sel, const int32_t *__restrict__ st, int64_t k, const int32_t *__restrict__ ph,
                             const int32_t *__restrict__ pl, const int32_t *__restrict__ cnt, const double *__restrict__ rows, int64_t n_lo,
                             int64_t begin, double *__restrict__ out) {
    const int flags = (st[ST_FLAG_C] ? 1 : 0) | (st[ST_FLAG_PAIRS] ? 2 : 0) | (st[ST_SHARD_NLO] ? 4 : 0) | (st[ST_FLAG_SEL] ? 8 : 0);
    const int64_t m = (st[ST_NPAIRS] > 0 && !flags) ? min((int64_t)st[ST_NKEYS], k) : 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = (double)m; out[1] = (double)flags; out[2] = (double)st[ST_LHI]; out[3] = (double)st[ST_NPAIRS]; }
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < k; t += (int64_t)gridDim.x * blockDim.x) {
        const bool have = t < m;
        const int64_t p = have ? sel[t] : 0;
        for (int c = 0; c < MAD_RESULT_COLS; c++) out[4 + t * MAD_RESULT_COLS + c] = have ? rows[t * MAD_RESULT_COLS + c] : 0.0;
        out[4 + k * MAD_RESULT_COLS + t] = have ? (double)cnt[p] : 0.0;
        out[4 + k * (MAD_RESULT_COLS + 1) + t] = have ? (double)((int64_t)ph[p] * n_lo + begin + pl[p]) : 0.0;
    }
}

extern "C" int64_t mad_match_shard_record_doubles(int64_t k) { return 4 + (k < 1 ? 1 : k) * (MAD_RESULT_COLS + 2); }

// Stage B, asynchronous: hi against the lo rows [lo_begin, lo_end) of a lo set assumed to have n_lo rows.  d_flags (device,
// hi->n_anchors + lo->n_anchors bytes) receives this shard's "anchor takes part in a pair" flags, hi's first.
extern "C" int mad_match_shard_begin(mad_ctx *ctx, const mad_set *hi, const mad_set *lo, int64_t lo_begin, int64_t lo_end, int64_t n_lo,
                                     double cc, uint8_t *d_flags) {
    if (!ctx || !hi || !lo || !d_flags) return MAD_EINVAL;
    if (hi->D != lo->D) return mad_fail(ctx, MAD_EINVAL, "mad_match_shard_begin: descriptor lengths %d vs %d", hi->D, lo->D);
    if (lo_begin < 0 || lo_end < lo_begin || lo_end > n_lo || n_lo > lo->cap_rows)
        return mad_fail(ctx, MAD_EINVAL, "mad_match_shard_begin: lo rows [%lld, %lld) of %lld (capacity %lld)", (long long)lo_begin, (long long)lo_end, (long long)n_lo, (long long)lo->cap_rows);
    const int lane = hi->lane;
    mad_use_lane(ctx, lane);
    ShardAsync &S = ctx->shard_async[lane];
    S = ShardAsync();
    const int64_t nb = lo_end - lo_begin;
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_ZERO), zero_bytes(hi, lo)));
    int32_t *st = zero_status(ctx);
    uint8_t *d_used_hi = zr_used_hi(st, hi->n_anchors), *d_used_lo = zr_used_lo(st, hi->n_anchors);
    MAD_HIP(hipStreamWaitEvent(ctx->stream, hi->built, 0));
    MAD_HIP(hipStreamWaitEvent(ctx->stream, lo->built, 0));
    const int64_t cap_pairs = std::max<int64_t>(ctx->match.cap_pairs, 1 << 16);
    // the score matrix for the rows hi had at its previous build (a hint: the device flags a matrix that does not fit)
    const int64_t hi_rows = std::min<int64_t>(hi->cap_rows, hi->rows_hint > 0 ? hi->rows_hint + hi->rows_hint / 8 + 64 : (int64_t)hi->n_anchors * 8 + 128);
    const int64_t cap_c = (mad_ceil_div(std::max<int64_t>(hi_rows, 1), 128) * 128) * (mad_ceil_div(std::max<int64_t>(nb, 1), 128) * 128);
    if (cap_c >= ((int64_t)1 << 31)) return mad_fail(ctx, MAD_EINVAL, "mad_match_shard_begin: score matrix of %lld entries", (long long)cap_c);
    mad_zero_words(ctx, st, zero_bytes(hi, lo));
    hipLaunchKernelGGL(k_shard_head, dim3(1), dim3(64), 0, ctx->stream, st, (int32_t)nb, (const int32_t *)lo->dev_n.p, (int32_t)n_lo);
    const Side H = side_of(hi), L = side_block(lo, lo_begin, st + ST_NLO, nb);
    MAD_TRY(correlate_device(ctx, H, L, hi->D, cc, st, cap_c, cap_pairs, d_used_hi, d_used_lo));
    if (hi->n_anchors > 0) MAD_HIP(hipMemcpyAsync(d_flags, d_used_hi, (size_t)hi->n_anchors, hipMemcpyDeviceToDevice, ctx->stream));
    if (lo->n_anchors > 0) MAD_HIP(hipMemcpyAsync(d_flags + hi->n_anchors, d_used_lo, (size_t)lo->n_anchors, hipMemcpyDeviceToDevice, ctx->stream));
    S.hi = hi; S.lo = lo; S.hi_gen = hi->gen; S.lo_gen = lo->gen; S.begin = lo_begin; S.nb = nb; S.n_lo = n_lo; S.cap_pairs = cap_pairs;
    return MAD_OK;
}

// Stage C, asynchronous: the shard's pairs scored against the GLOBAL clouds -- d_flags_all = the OR of all shards' flags, same
// layout, device memory -- and its k best packed into d_out (mad_match_shard_record_doubles(k) float64, device memory).
extern "C" int mad_match_shard_score(mad_ctx *ctx, const mad_set *hi, const mad_set *lo, const uint8_t *d_flags_all, double dist, int64_t k,
                                     double *d_out) {
    if (!ctx || !hi || !lo || !d_flags_all || !d_out) return MAD_EINVAL;
    if (!(dist > 0) || k < 1) return mad_fail(ctx, MAD_EINVAL, "mad_match_shard_score: dist %g k %lld", dist, (long long)k);
    const int lane = hi->lane;
    mad_use_lane(ctx, lane);
    const ShardAsync S = ctx->shard_async[lane];
    if (S.hi != hi || S.lo != lo || S.hi_gen != hi->gen || S.lo_gen != lo->gen)
        return mad_fail(ctx, MAD_EINVAL, "mad_match_shard_score: call mad_match_shard_begin for these sets (as they are now) first");
    ctx->shard_async[lane] = ShardAsync();
    int32_t *st = zero_status(ctx);
    int32_t *hist = zr_hist(st);
    uint8_t *d_used_hi = zr_used_hi(st, hi->n_anchors), *d_used_lo = zr_used_lo(st, hi->n_anchors);
    // the global flags replace the shard's own; the histograms of the selection start from zero
    if (hi->n_anchors > 0) MAD_HIP(hipMemcpyAsync(d_used_hi, d_flags_all, (size_t)hi->n_anchors, hipMemcpyDeviceToDevice, ctx->stream));
    if (lo->n_anchors > 0) MAD_HIP(hipMemcpyAsync(d_used_lo, d_flags_all + hi->n_anchors, (size_t)lo->n_anchors, hipMemcpyDeviceToDevice, ctx->stream));
    MAD_HIP(hipMemsetAsync(hist, 0, (size_t)(hi->n_anchors + 17) * 8, ctx->stream));      // hist and hist2
    MAD_HIP(hipMemsetAsync(st + ST_NKEYS, 0, 4, ctx->stream));
    MAD_HIP(hipMemsetAsync(st + ST_NSEL, 0, 4, ctx->stream));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_HI_CLOUD), (size_t)hi->n_anchors * 24 + 24));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_SEL_OUT), (size_t)(k + 8) * 8));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_RESULTS), (size_t)(k + 1) * MAD_RESULT_COLS * 8));
    hipLaunchKernelGGL(k_compact_cloud, dim3(1), dim3(1024), 0, ctx->stream,
                       CloudJob{(const double *)hi->anc_subv.p, d_used_hi, hi->n_anchors, scratch<double>(ctx, S_HI_CLOUD), st + ST_LHI, nullptr, nullptr, st});
    MAD_HIP(hipGetLastError());
    const Side H = side_of(hi), L = side_block(lo, S.begin, st + ST_NLO, S.nb);
    const bool fits = clouds_fit_lds(hi->n_anchors, lo->n_anchors);
    CellGrid G;
    if (!fits) {
        if (!lo->cells_ready || lo->cell_size != dist) {      // (once per lo set and distance: shared by every later match)
            MAD_TRY(mad_build_cells(ctx, const_cast<mad_set *>(lo), dist));
            MAD_HIP(hipStreamSynchronize(ctx->stream));
        }
        G.start = (const int32_t *)lo->cell_start.p; G.pts = (const double *)lo->cell_pts.p; G.ids = (const int32_t *)lo->cell_ids.p;
        for (int d = 0; d < 3; d++) { G.mn[d] = lo->cell_min[d]; G.dim[d] = lo->cell_dim[d]; }
        G.cell = lo->cell_size;
        G.used = d_used_lo;
    }
    MAD_TRY(pose_device(ctx, H, L, st, S.cap_pairs, scratch<double>(ctx, S_HI_CLOUD), hi->n_anchors, (const double *)lo->anc_subv.p,
                        lo->n_anchors, d_used_lo, lo->bb_min, lo->bb_max, fits ? nullptr : &G, dist, k, zr_hist2(st, hi->n_anchors)));
    MAD_TRY(topk_device(ctx, scratch<int32_t>(ctx, S_COUNTS), st, S.cap_pairs, k, hi->n_anchors, scratch<int64_t>(ctx, S_SEL_OUT), hist));
    hipLaunchKernelGGL(k_results, dim3((unsigned)mad_ceil_div(k, 256)), dim3(256), 0, ctx->stream, scratch<int64_t>(ctx, S_SEL_OUT),
                       st + ST_NKEYS, k, scratch<int32_t>(ctx, S_PAIR_HI), scratch<int32_t>(ctx, S_PAIR_LO),
                       scratch<double>(ctx, S_PAIR_SCORE), scratch<int32_t>(ctx, S_COUNTS), st, H.p, H.R, H.meta, L.p, L.Rinv, L.meta,
                       H.row_anchor, L.row_anchor, scratch<double>(ctx, S_RESULTS), 0);
    hipLaunchKernelGGL(k_shard_pack, dim3((unsigned)mad_ceil_div(k, 256)), dim3(256), 0, ctx->stream, scratch<int64_t>(ctx, S_SEL_OUT), st, k,
                       scratch<int32_t>(ctx, S_PAIR_HI), scratch<int32_t>(ctx, S_PAIR_LO), scratch<int32_t>(ctx, S_COUNTS),
                       scratch<double>(ctx, S_RESULTS), S.n_lo, S.begin, d_out);
    MAD_HIP(hipGetLastError());
    return MAD_OK;
}

// The records of a group's shards (d_all: n float64 in device memory, e.g. what the all-gather behind mad_match_shard_score left) on
// their way to the host: a copy into pinned memory of the library on the lane of `hi`, an event behind it.  *ticket names the
// copy for mad_match_shard_wait, which blocks until it has arrived and hands it out.  The staging and the events are the library's
// own: the caller's framework never records an event on, or ties memory to, a stream the library destroys.
extern "C" int mad_match_shard_collect(mad_ctx *ctx, const mad_set *hi, const double *d_all, int64_t n, int *ticket) {
    if (!ctx || !hi || !d_all || !ticket || n < 1) return MAD_EINVAL;
    const int lane = hi->lane;
    mad_use_lane(ctx, lane);
    int slot = -1;
    for (int t = 0; t < MAD_SHARD_RING && slot < 0; t++) {
        const int c = (ctx->shard_next[lane] + t) % MAD_SHARD_RING;
        if (!ctx->shard_busy[lane][c]) slot = c;
    }
    if (slot < 0) return mad_fail(ctx, MAD_EINVAL, "mad_match_shard_collect: %d records of lane %d have not been waited for", MAD_SHARD_RING, lane);
    ctx->shard_next[lane] = (slot + 1) % MAD_SHARD_RING;
    const size_t bytes = (size_t)n * 8;
    if (ctx->shard_host_cap[lane][slot] < bytes) {
        if (ctx->shard_host[lane][slot]) (void)hipHostFree(ctx->shard_host[lane][slot]);
        ctx->shard_host[lane][slot] = nullptr;
        ctx->shard_host_cap[lane][slot] = 0;
        if (hipHostMalloc(&ctx->shard_host[lane][slot], bytes * 2) != hipSuccess) return mad_fail(ctx, MAD_ENOMEM, "pinned shard records of %zu bytes", bytes * 2);
        ctx->shard_host_cap[lane][slot] = bytes * 2;
    }
    if (!ctx->shard_ev[lane][slot]) MAD_HIP(hipEventCreateWithFlags(&ctx->shard_ev[lane][slot], hipEventDisableTiming));
    MAD_HIP(hipMemcpyAsync(ctx->shard_host[lane][slot], d_all, bytes, hipMemcpyDeviceToHost, ctx->stream));
    MAD_HIP(hipEventRecord(ctx->shard_ev[lane][slot], ctx->stream));
    ctx->shard_busy[lane][slot] = true;
    *ticket = lane * MAD_SHARD_RING + slot;
    return MAD_OK;
}

extern "C" int mad_match_shard_wait(mad_ctx *ctx, int ticket, double *out, int64_t n) {
    if (!ctx || !out || n < 1 || ticket < 0 || ticket >= MAD_LANES * MAD_SHARD_RING) return MAD_EINVAL;
    const int lane = ticket / MAD_SHARD_RING, slot = ticket % MAD_SHARD_RING;
    if (!ctx->shard_busy[lane][slot] || ctx->shard_host_cap[lane][slot] < (size_t)n * 8) return mad_fail(ctx, MAD_EINVAL, "mad_match_shard_wait: ticket %d is not pending", ticket);
    MAD_HIP(hipEventSynchronize(ctx->shard_ev[lane][slot]));
    memcpy(out, ctx->shard_host[lane][slot], (size_t)n * 8);
    ctx->shard_busy[lane][slot] = false;
    return MAD_OK;
}

// The match counts of ALL pairs of the last match: when its pose search was pruned by bounds (only the k best were asked for),
// run the exact search over every pair now.  The pair list, the clouds' flags and the status words of that match are still in its
// lane's scratch (the condition mad_match_fetch has always had); the two sets must still exist.
static int complete_counts(mad_ctx *ctx) {
    if (!ctx->match.pruned || ctx->match.n_pairs <= 0) return MAD_OK;
    const mad_set *hi = (const mad_set *)ctx->match.hi, *lo = (const mad_set *)ctx->match.lo;
    if (!hi || !lo) return mad_fail(ctx, MAD_EINVAL, "match counts: a set of the last match has been destroyed; its pruned pose search cannot be completed");
    if (hi->gen != ctx->match.hi_gen || lo->gen != ctx->match.lo_gen)
        return mad_fail(ctx, MAD_EINVAL, "match counts: a set of the last match has been rebuilt since; its pruned pose search cannot be completed");
    mad_use_lane(ctx, ctx->match.lane);
    int32_t *st = zero_status(ctx);
    const Side H = side_of(hi), L = side_of(lo);
    MAD_TRY(pose_device(ctx, H, L, st, ctx->match.cap_pairs_used, scratch<double>(ctx, S_HI_CLOUD), hi->n_anchors, (const double *)lo->anc_subv.p,
                        lo->n_anchors, zr_used_lo(st, hi->n_anchors), lo->bb_min, lo->bb_max, nullptr, ctx->match.dist));
    MAD_HIP(hipStreamSynchronize(ctx->stream));
    ctx->match.pruned = false;
    return MAD_OK;
}

extern "C" int mad_match_fetch(mad_ctx *ctx, int32_t *pair_hi, int32_t *pair_lo, double *pair_score, int32_t *counts,
                               int64_t cap) {
    if (!ctx) return MAD_EINVAL;
    if (counts) MAD_TRY(complete_counts(ctx));
    mad_use_lane(ctx, ctx->match.lane);
    const int64_t np = ctx->match.n_pairs;
    if (np > cap) return mad_fail(ctx, MAD_ENOSPC, "mad_match_fetch: %lld pairs, capacity %lld", (long long)np, (long long)cap);
    if (np <= 0) return MAD_OK;
    if (pair_hi) MAD_HIP(hipMemcpyAsync(pair_hi, mad_sb(ctx, S_PAIR_HI).p, np * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (pair_lo) MAD_HIP(hipMemcpyAsync(pair_lo, mad_sb(ctx, S_PAIR_LO).p, np * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (pair_score) MAD_HIP(hipMemcpyAsync(pair_score, mad_sb(ctx, S_PAIR_SCORE).p, np * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (counts) MAD_HIP(hipMemcpyAsync(counts, mad_sb(ctx, S_COUNTS).p, np * 4, hipMemcpyDeviceToHost, ctx->stream));
    MAD_HIP(hipStreamSynchronize(ctx->stream));
    return MAD_OK;
}

extern "C" int mad_match_results(mad_ctx *ctx, const mad_set *hi, const mad_set *lo, double *results, int64_t cap) {
    if (!ctx || !hi || !lo || !results) return MAD_EINVAL;
    MAD_TRY(complete_counts(ctx));
    mad_use_lane(ctx, ctx->match.lane);
    const int64_t np = ctx->match.n_pairs;
    if (np > cap) return mad_fail(ctx, MAD_ENOSPC, "mad_match_results: %lld pairs, capacity %lld", (long long)np, (long long)cap);
    if (np <= 0) return MAD_OK;
    if (hi->n_anchors != ctx->match.n_hi_anchors || lo->n_anchors != ctx->match.n_lo_anchors)
        return mad_fail(ctx, MAD_EINVAL, "mad_match_results: sets differ from the last mad_match_topk call");
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_RESULTS), (size_t)np * MAD_RESULT_COLS * 8));
    int32_t *st = zero_status(ctx);
    const Side H = side_of(hi), L = side_of(lo);
    hipLaunchKernelGGL(k_results, dim3((unsigned)std::min<int64_t>(mad_ceil_div(np, 256), 4096)), dim3(256), 0, ctx->stream,
                       (const int64_t *)nullptr, st + ST_NPAIRS, np, scratch<int32_t>(ctx, S_PAIR_HI), scratch<int32_t>(ctx, S_PAIR_LO),
                       scratch<double>(ctx, S_PAIR_SCORE), scratch<int32_t>(ctx, S_COUNTS), st, H.p, H.R, H.meta, L.p, L.Rinv, L.meta,
                       H.row_anchor, L.row_anchor, scratch<double>(ctx, S_RESULTS), 0);
    MAD_HIP(hipGetLastError());
    MAD_HIP(hipMemcpyAsync(results, mad_sb(ctx, S_RESULTS).p, (size_t)np * MAD_RESULT_COLS * 8, hipMemcpyDeviceToHost, ctx->stream));
    MAD_HIP(hipStreamSynchronize(ctx->stream));
    return MAD_OK;
}

extern "C" int mad_match_used(mad_ctx *ctx, uint8_t *hi_used, int32_t n_hi_anchors, uint8_t *lo_used, int32_t n_lo_anchors) {
    if (!ctx) return MAD_EINVAL;
    if (n_hi_anchors != ctx->match.n_hi_anchors || n_lo_anchors != ctx->match.n_lo_anchors)
        return mad_fail(ctx, MAD_EINVAL, "mad_match_used: anchor counts do not match the last mad_match_topk call");
    mad_use_lane(ctx, ctx->match.lane);
    const uint8_t *d_hi = zr_used_hi(zero_status(ctx), n_hi_anchors), *d_lo = zr_used_lo(zero_status(ctx), n_hi_anchors);
    if (hi_used && n_hi_anchors > 0) MAD_HIP(hipMemcpyAsync(hi_used, d_hi, n_hi_anchors, hipMemcpyDeviceToHost, ctx->stream));
    if (lo_used && n_lo_anchors > 0) MAD_HIP(hipMemcpyAsync(lo_used, d_lo, n_lo_anchors, hipMemcpyDeviceToHost, ctx->stream));
    MAD_HIP(hipStreamSynchronize(ctx->stream));
    return MAD_OK;
}

// ---------------------------------------------------------------------------
// a structure's rows built in shares on several GPUs (stage A)
// ---------------------------------------------------------------------------
//
// Orientation and description are independent per anchor (Orientator.py:80-108, Descriptor.py:106-116), so the anchors
// of one structure are dealt round-robin to the ranks (anchor a -> share a % n_shares, local position a / n_shares), each
// rank runs mad_set_build on its share, and the shares travel as "wire images" (what one all-gather moves):
//   [header 64 B][main int32 C][sec int32 C][local anchor int32 C][norm double C][dsc8 int8 C x D],   C = cap_rows
// every section padded to 16 bytes.  Rfinal, its inverse, the int16 descriptors and the result metadata are NOT sent:
// the importer re-derives them with the expressions the builder used (mad_rfinal, mad_mat3_inv, a widening copy).
// mad_set_import scatters the rows of all shares into the reference's order: anchor order x main x sec
// (Orientator.py:90-106) -- a set bit-identical to the one mad_set_build makes from the whole anchor list.

struct WireHeader {
    int32_t n_rows, n_anchors, overflow, range_bad, rejects, D, cap_rows, magic;
    int32_t pad[8];
};
static_assert(sizeof(WireHeader) == 64, "wire header is 64 bytes");
#define MAD_WIRE_MAGIC 0x4d614431

struct WireLayout {
    size_t o_main, o_sec, o_anchor, o_norm, o_dsc8, total;
};

__host__ __device__ static inline WireLayout wire_layout(int D, int64_t C) {
    WireLayout L;
    L.o_main = 64;
    L.o_sec = L.o_main + pad16((size_t)C * 4);
    L.o_anchor = L.o_sec + pad16((size_t)C * 4);
    L.o_norm = L.o_anchor + pad16((size_t)C * 4);
    L.o_dsc8 = L.o_norm + pad16((size_t)C * 8);
    L.total = L.o_dsc8 + pad16((size_t)C * D);
    return L;
}

extern "C" int64_t mad_set_wire_bytes(int D, int64_t cap_rows) {
    if (D <= 0 || cap_rows < 0) return -1;
    return (int64_t)wire_layout(D, cap_rows > 0 ? cap_rows : 1).total;
}

extern "C" int mad_set_lane(mad_ctx *ctx, const mad_set *s) { return (ctx && s) ? s->lane : -1; }

extern "C" void *mad_set_stream(mad_ctx *ctx, const mad_set *s) {
    return (ctx && s) ? (void *)ctx->lane_stream[ctx->overlap ? s->lane : 0] : nullptr;
}

extern "C" int mad_set_bind_lane(mad_ctx *ctx, mad_set *s, int lane) {
    if (!ctx || !s || lane < 0 || lane >= MAD_LANES) return MAD_EINVAL;
    MAD_TRY(mad_synchronize(ctx));      // nothing of the set may be in flight on its old lane
    s->lane = lane;
    return MAD_OK;
}

__global__ __launch_bounds__(256) void k_set_export(const int32_t *__restrict__ dev_n, const int32_t *__restrict__ row_main,
                                                    const int32_t *__restrict__ row_sec, const int32_t *__restrict__ row_anchor,
                                                    const double *__restrict__ norm, const int8_t *__restrict__ dsc8, int D, int64_t C,
                                                    int n_anchors, unsigned char *__restrict__ wire) {
    const WireLayout L = wire_layout(D, C);
    const int64_t n_all = dev_n[0];
    const bool over = n_all > C || dev_n[3] != 0;      // more rows than the wire holds, or the share's own describe launch fell short
    const int64_t n = over ? 0 : n_all;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        WireHeader h;
        h.n_rows = (int32_t)n; h.n_anchors = n_anchors; h.overflow = over ? (int32_t)max(n_all, (int64_t)1) : 0; h.range_bad = dev_n[1];
        h.rejects = dev_n[2]; h.D = D; h.cap_rows = (int32_t)C; h.magic = MAD_WIRE_MAGIC;
        for (int i = 0; i < 8; i++) h.pad[i] = 0;
        *(WireHeader *)wire = h;
    }
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = tid; i < n; i += nt) {
        ((int32_t *)(wire + L.o_main))[i] = row_main[i];
        ((int32_t *)(wire + L.o_sec))[i] = row_sec[i];
        ((int32_t *)(wire + L.o_anchor))[i] = row_anchor[i];
        ((double *)(wire + L.o_norm))[i] = norm[i];
    }
    const int64_t n16 = n * D / 16;      // D is a multiple of 16
    const uint4 *src = (const uint4 *)dsc8;
    uint4 *dst = (uint4 *)(wire + L.o_dsc8);
    for (int64_t i = tid; i < n16; i += nt) dst[i] = src[i];
}

// Packs the rows of `share` (built with mad_set_build) into a wire image of capacity cap_rows.  wire_on_device != 0:
// `wire` is device memory and the call is asynchronous on the set's lane (mad_set_stream); otherwise host memory,
// synchronous.  A share with more than cap_rows rows travels as an empty image whose header says how many it had.
extern "C" int mad_set_export(mad_ctx *ctx, const mad_set *s, void *wire, int wire_on_device, int64_t cap_rows) {
    if (!ctx || !s || !wire || cap_rows < 1) return MAD_EINVAL;
    if (!s->dev_n.p || s->D <= 0 || (s->D % 16)) return mad_fail(ctx, MAD_EINVAL, "mad_set_export: the set has not been built");
    mad_use_lane(ctx, s->lane);
    MAD_HIP(hipStreamWaitEvent(ctx->stream, s->built, 0));      // the build may have run on another lane (mad_set_build_many)
    const WireLayout L = wire_layout(s->D, cap_rows);
    unsigned char *d_wire = (unsigned char *)wire;
    if (!wire_on_device) {
        MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_TMP_B), L.total));
        d_wire = scratch<unsigned char>(ctx, S_TMP_B);
    }
    const int64_t work = std::max<int64_t>(std::min<int64_t>(s->cap_rows, cap_rows) * s->D / 16, 256);
    hipLaunchKernelGGL(k_set_export, dim3((unsigned)std::min<int64_t>(mad_ceil_div(work, 256), (int64_t)ctx->n_cu * 8)), dim3(256), 0,
                       ctx->stream, (const int32_t *)s->dev_n.p, (const int32_t *)s->row_main.p, (const int32_t *)s->row_sec.p,
                       (const int32_t *)s->row_anchor.p, (const double *)s->norm.p, (const int8_t *)s->dsc8.p, s->D, cap_rows, s->n_anchors,
                       d_wire);
    MAD_HIP(hipGetLastError());
    MAD_HIP(hipEventRecord(s->ready, ctx->stream));      // a later build of this set on another lane waits for this read
    if (!wire_on_device) {
        MAD_HIP(hipMemcpyAsync(wire, d_wire, L.total, hipMemcpyDeviceToHost, ctx->stream));
        MAD_HIP(hipStreamSynchronize(ctx->stream));
    }
    return MAD_OK;
}

// rows per anchor of the whole structure (anchor = local anchor * n_shares + share), counted from the shares' row lists
__global__ __launch_bounds__(256) void k_import_count(const unsigned char *__restrict__ wires, size_t wire_bytes, int n_shares, int D, int64_t C,
                                                      int n_anchors, int32_t *__restrict__ cnt, int32_t *__restrict__ flags) {
    const WireLayout L = wire_layout(D, C);
    for (int r = blockIdx.y; r < n_shares; r += gridDim.y) {
        const unsigned char *w = wires + (size_t)r * wire_bytes;
        const WireHeader *h = (const WireHeader *)w;
        const bool bad = h->magic != MAD_WIRE_MAGIC || h->D != D || h->cap_rows != (int32_t)C || h->n_rows < 0 || h->n_rows > C;
        if (bad || h->overflow) {
            if (blockIdx.x == 0 && threadIdx.x == 0) atomicMax(&flags[bad ? 1 : 0], bad ? 1 : h->overflow);
            continue;
        }
        const int32_t *la = (const int32_t *)(w + L.o_anchor);
        for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < h->n_rows; j += (int64_t)gridDim.x * blockDim.x) {
            const int64_t a = (int64_t)la[j] * n_shares + r;
            if (la[j] < 0 || a >= n_anchors) atomicMax(&flags[1], 1);
            else atomicAdd(&cnt[a], 1);
        }
    }
}

// one wave per source row: its place in the reference's order, the copied and the re-derived fields
__global__ __launch_bounds__(256) void k_import_rows(const unsigned char *__restrict__ wires, size_t wire_bytes, int n_shares, int D, int64_t C,
                                                     const int32_t *__restrict__ row_off, int n_anchors, const int32_t *__restrict__ flags,
                                                     const EqspDev *__restrict__ eq, const int32_t *__restrict__ anc_index,
                                                     const int32_t *__restrict__ anc_octave, int32_t *__restrict__ dev_n,
                                                     int32_t *__restrict__ row_anchor, int32_t *__restrict__ row_main, int32_t *__restrict__ row_sec,
                                                     double *__restrict__ row_R, double *__restrict__ row_Rinv, int32_t *__restrict__ row_meta,
                                                     int16_t *__restrict__ dsc, int8_t *__restrict__ dsc8, double *__restrict__ norm) {
    const WireLayout L = wire_layout(D, C);
    const bool failed = flags[0] != 0 || flags[1] != 0;
    const int64_t total = failed ? 0 : row_off[n_anchors];
    const int lane = lane_id();
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
    if (wave == 0 && lane == 0) {
        int32_t rej = 0, bad = 0;
        for (int r = 0; r < n_shares; r++) {
            const WireHeader *h = (const WireHeader *)(wires + (size_t)r * wire_bytes);
            rej += h->rejects; bad |= h->range_bad;
        }
        // {rows, rows out of the int8 range, border rejects, incomplete}: an incomplete import reads as an empty set whose
        // fourth word tells mad_set_size / the match calls why
        dev_n[0] = (int32_t)total; dev_n[1] = bad; dev_n[2] = rej; dev_n[3] = failed ? (flags[1] ? -1 : flags[0]) : 0;
    }
    // rows up to the next multiple of 128 are zero: the GEMM reads whole tiles
    const int64_t n_pad = (total + 127) / 128 * 128;
    for (int64_t r = total + wave; r < n_pad; r += nw) {
        for (int i = lane; i < D / 16; i += MAD_WAVE) ((uint4 *)(dsc8 + r * D))[i] = make_uint4(0, 0, 0, 0);
        if (lane == 0) norm[r] = 0.0;
    }
    if (failed) return;
    const int64_t slots = (int64_t)n_shares * C;
    for (int64_t s = wave; s < slots; s += nw) {
        const int r = (int)(s / C);
        const int64_t j = s % C;
        const unsigned char *w = wires + (size_t)r * wire_bytes;
        if (j >= ((const WireHeader *)w)->n_rows) continue;
        const int32_t *la = (const int32_t *)(w + L.o_anchor);
        const int me = la[j];
        int64_t first = j;      // rows of one anchor are consecutive in a share (at most lim_main x lim_sec of them)
        while (first > 0 && la[first - 1] == me) first--;
        const int a = me * n_shares + r;
        const int64_t dst = (int64_t)row_off[a] + (j - first);
        const uint4 *src8 = (const uint4 *)(w + L.o_dsc8 + (size_t)j * D);
        for (int i = lane; i < D / 16; i += MAD_WAVE) {
            const uint4 v = src8[i];
            ((uint4 *)(dsc8 + dst * D))[i] = v;
            // the int16 counts the builder also keeps (mad_set_download): a widening copy
            const int8_t *b = (const int8_t *)&v;
            short4 lo4[2], hi4[2];
            lo4[0] = make_short4(b[0], b[1], b[2], b[3]); lo4[1] = make_short4(b[4], b[5], b[6], b[7]);
            hi4[0] = make_short4(b[8], b[9], b[10], b[11]); hi4[1] = make_short4(b[12], b[13], b[14], b[15]);
            short4 *o = (short4 *)(dsc + dst * D + 16 * i);
            o[0] = lo4[0]; o[1] = lo4[1]; o[2] = hi4[0]; o[3] = hi4[1];
        }
        if (lane == 0) {
            const int mb = ((const int32_t *)(w + L.o_main))[j], sb = ((const int32_t *)(w + L.o_sec))[j];
            norm[dst] = ((const double *)(w + L.o_norm))[j];
            row_anchor[dst] = a; row_main[dst] = mb; row_sec[dst] = sb;
            double R[9];
            mad_rfinal(eq, mb, sb, R);
            for (int i = 0; i < 9; i++) row_R[9 * dst + i] = R[i];
            mad_mat3_inv(R, row_Rinv + 9 * dst);
            row_meta[3 * dst] = anc_index[a]; row_meta[3 * dst + 1] = anc_octave[a]; row_meta[3 * dst + 2] = mb;
        }
    }
}

// Assembles `s` from the wire images of n_shares shares (contiguous, share r at wires + r * mad_set_wire_bytes(D, cap_rows)).
// The anchors are those of the WHOLE structure, in the reference's order (anc_coords may be NULL: an imported set is not
// described again).  wires_on_device != 0: device memory, asynchronous on the set's lane; otherwise host memory.
extern "C" int mad_set_import(mad_ctx *ctx, mad_set *s, const void *wires, int wires_on_device, int n_shares, int64_t cap_rows,
                              const int32_t *anc_coords, const int32_t *anc_octave, const double *anc_subv, const int32_t *anc_index,
                              int n_anchors) {
    if (!ctx || !s || !wires || n_shares < 1 || cap_rows < 1) return MAD_EINVAL;
    if (n_anchors > 0 && (!anc_octave || !anc_subv || !anc_index)) return mad_fail(ctx, MAD_EINVAL, "mad_set_import: NULL anchors");
    if (n_anchors > 65536) return mad_fail(ctx, MAD_EINVAL, "mad_set_import: %d anchors exceed the single-launch scan", n_anchors);
    if (!ctx->eq_set[0] || !ctx->eq_set[1]) return mad_fail(ctx, MAD_EINVAL, "mad_set_import: EQSP tables not set");
    mad_use_lane(ctx, s->lane);
    const int D = 64 * ctx->eq_host[1].Z;
    const WireLayout L = wire_layout(D, cap_rows);
    const unsigned char *d_wires = (const unsigned char *)wires;
    if (!wires_on_device) {
        MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_TMP_B), L.total * n_shares));
        MAD_HIP(hipMemcpyAsync(mad_sb(ctx, S_TMP_B).p, wires, L.total * n_shares, hipMemcpyHostToDevice, ctx->stream));
        d_wires = scratch<unsigned char>(ctx, S_TMP_B);
    }
    MAD_TRY(set_upload_anchors(ctx, s, anc_coords, anc_octave, anc_subv, anc_index, n_anchors));
    s->D = D;
    const int64_t cap_all = std::min<int64_t>((int64_t)n_shares * cap_rows, std::max<int64_t>((int64_t)n_anchors * 64, 1));
    MAD_TRY(set_reserve_rows(ctx, s, cap_all));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_SLOT_CNT), (size_t)(n_anchors + 8) * 4));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_ROW_OFF), (size_t)(n_anchors + 8) * 4));
    int32_t *cnt = scratch<int32_t>(ctx, S_SLOT_CNT), *off = scratch<int32_t>(ctx, S_ROW_OFF);
    int32_t *flags = cnt + n_anchors;      // {overflow: rows the largest share had, malformed image}
    mad_zero_words(ctx, cnt, (size_t)(n_anchors + 8) * 4);
    const unsigned gx = (unsigned)std::min<int64_t>(mad_ceil_div(cap_rows, 256), 64);
    hipLaunchKernelGGL(k_import_count, dim3(gx, (unsigned)std::min(n_shares, 64)), dim3(256), 0, ctx->stream, d_wires, L.total, n_shares, D,
                       cap_rows, n_anchors, cnt, flags);
    mad_scan_small(ctx, cnt, off, nullptr, nullptr, n_anchors);
    hipLaunchKernelGGL(k_import_rows, dim3((unsigned)std::min<int64_t>(mad_ceil_div((int64_t)n_shares * cap_rows, 4) + 1, (int64_t)ctx->n_cu * 16)),
                       dim3(256), 0, ctx->stream, d_wires, L.total, n_shares, D, cap_rows, off, n_anchors, flags, ctx->eq[0],
                       (const int32_t *)s->anc_index.p, (const int32_t *)s->anc_octave.p, (int32_t *)s->dev_n.p, (int32_t *)s->row_anchor.p,
                       (int32_t *)s->row_main.p, (int32_t *)s->row_sec.p, (double *)s->row_R.p, (double *)s->row_Rinv.p,
                       (int32_t *)s->row_meta.p, (int16_t *)s->dsc.p, (int8_t *)s->dsc8.p, (double *)s->norm.p);
    MAD_HIP(hipGetLastError());
    s->last_r = 0;      // nothing to repeat locally: an incomplete import is the caller's to redo with larger images
    s->n_rows_host = -1;
    MAD_HIP(hipEventRecord(s->built, ctx->stream));
    if (!wires_on_device) MAD_HIP(hipStreamSynchronize(ctx->stream));      // the host images may go away
    return MAD_OK;
}
