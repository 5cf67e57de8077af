// mad_match.hip -- descriptor correlation (a11), pose scoring (a12) and top-k for gfx950.
// Reference: MaD._match_dsc (mad/MaD.py:414-453) and the stable sort of
// MaD._filter_dsc_pairs (mad/MaD.py:480).
//
//  correlate : descriptor counts are <= 64, so rows are packed to int8 and the
//              N_hi x N_lo x 1024 contraction runs on v_mfma_i32_16x16x64_i8 with exact
//              int32 accumulation; score = dot / (|h| |l|) in float64.
//  pairs     : per hi row, ordered compaction of the columns whose score exceeds cc
//              (row-major order of np.where, MaD.py:423).
//  pose      : one wavefront per pair; the used hi anchors sit in LDS, the lo anchors
//              in a uniform cell list (cell = dist) so that "nearest lo anchor closer
//              than dist" needs 9 contiguous cell runs instead of a k-d tree.
//  top-k     : histogram of the integer match counts -> threshold count -> ordered
//              pick of the ties -> one-workgroup bitonic sort of the k survivors.
#include "mad_common.h"

// ---------------------------------------------------------------------------
// int16 -> int8 packing + row norms
// ---------------------------------------------------------------------------

// one wave per row: int8 copy, sqrt of the exact integer sum of squares, range check
__global__ __launch_bounds__(256) void k_pack_rows(const int16_t *__restrict__ src, int64_t n_rows, int D,
                                                   int8_t *__restrict__ dst, double *__restrict__ norm,
                                                   int32_t *__restrict__ bad) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int lane = lane_id();
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
    if (__any(oob) && lane == 0) atomicExch(bad, 1);
    if (lane == 0) norm[row] = sqrt((double)ss);
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

__global__ __launch_bounds__(GEMM_THREADS) void k_corr_gemm(const int8_t *__restrict__ A, const int8_t *__restrict__ B,
                                                            int K, int32_t *__restrict__ C, int64_t ldc) {
    __shared__ __align__(16) int8_t sA[GEMM_BM * GEMM_LDA];
    __shared__ __align__(16) int8_t sB[GEMM_BN * GEMM_LDA];
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int64_t row0 = (int64_t)blockIdx.y * GEMM_BM, col0 = (int64_t)blockIdx.x * GEMM_BN;
    v4i acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < 4; n++) acc[m][n] = (v4i){0, 0, 0, 0};

    for (int k0 = 0; k0 < K; k0 += GEMM_BK) {
        // stage 128 rows x 128 B of each operand: 1024 16-byte chunks per operand, 4 per thread
        v4i ra[4], rb[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int c = tid + GEMM_THREADS * i;
            const int r = c >> 3, q = c & 7;
            ra[i] = *(const v4i *)(A + (row0 + r) * K + k0 + q * 16);
            rb[i] = *(const v4i *)(B + (col0 + r) * K + k0 + q * 16);
        }
        __syncthreads();      // previous stage fully consumed
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int c = tid + GEMM_THREADS * i;
            const int r = c >> 3, q = c & 7;
            *(v4i *)(sA + r * GEMM_LDA + q * 16) = ra[i];
            *(v4i *)(sB + r * GEMM_LDA + q * 16) = rb[i];
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < GEMM_BK / 64; kk++) {
            v4i fa[4], fb[4];
            const int koff = kk * 64 + (lane >> 4) * 16;
#pragma unroll
            for (int m = 0; m < 4; m++) fa[m] = *(const v4i *)(sA + (wm * 64 + m * 16 + (lane & 15)) * GEMM_LDA + koff);
#pragma unroll
            for (int n = 0; n < 4; n++) fb[n] = *(const v4i *)(sB + (wn * 64 + n * 16 + (lane & 15)) * GEMM_LDA + koff);
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int n = 0; n < 4; n++)
                    acc[m][n] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[m], fb[n], acc[m][n], 0, 0, 0);
        }
    }
    // C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < 4; n++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int64_t r = row0 + wm * 64 + m * 16 + (lane >> 4) * 4 + j;
                const int64_t c = col0 + wn * 64 + n * 16 + (lane & 15);
                C[r * ldc + c] = acc[m][n][j];
            }
}

// ---------------------------------------------------------------------------
// threshold + ordered compaction (np.where(preds > cc), MaD.py:423)
// ---------------------------------------------------------------------------

__device__ __forceinline__ double corr_score(int dot, double nh, double nl) {
    // zero rows stay un-normalised in the reference (MaD.py:416) -> divide by 1
    return (double)dot / ((nh > 0 ? nh : 1.0) * (nl > 0 ? nl : 1.0));
}

__global__ __launch_bounds__(256) void k_pair_count(const int32_t *__restrict__ C, int64_t ldc, int64_t n_hi, int64_t n_lo,
                                                    const double *__restrict__ hn, const double *__restrict__ ln, double cc,
                                                    int32_t *__restrict__ row_cnt) {
    __shared__ int wt[4];
    const int64_t i = blockIdx.x;
    const double nh = hn[i];
    int c = 0;
    for (int64_t j = threadIdx.x; j < n_lo; j += 256) c += corr_score(C[i * ldc + j], nh, ln[j]) > cc ? 1 : 0;
    c = wave_sum_i32(c);
    if (lane_id() == 0) wt[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) row_cnt[i] = wt[0] + wt[1] + wt[2] + wt[3];
}

__global__ __launch_bounds__(256) void k_pair_emit(const int32_t *__restrict__ C, int64_t ldc, int64_t n_hi, int64_t n_lo,
                                                   const double *__restrict__ hn, const double *__restrict__ ln, double cc,
                                                   const int32_t *__restrict__ row_off, int32_t *__restrict__ pair_hi,
                                                   int32_t *__restrict__ pair_lo, double *__restrict__ pair_score,
                                                   const int32_t *__restrict__ hi_row_anchor,
                                                   const int32_t *__restrict__ lo_row_anchor,
                                                   uint8_t *__restrict__ used_hi, uint8_t *__restrict__ used_lo) {
    __shared__ int wt[5];
    const int64_t i = blockIdx.x;
    const double nh = hn[i];
    int64_t base = row_off[i];
    bool any = false;
    for (int64_t j0 = 0; j0 < n_lo; j0 += 256) {
        const int64_t j = j0 + threadIdx.x;
        double s = 0;
        bool p = false;
        if (j < n_lo) {
            s = corr_score(C[i * ldc + j], nh, ln[j]);
            p = s > cc;
        }
        int tot;
        const int pos = block_excl_scan(p ? 1 : 0, wt, &tot);
        if (p) {
            const int64_t o = base + pos;
            pair_hi[o] = (int32_t)i;
            pair_lo[o] = (int32_t)j;
            pair_score[o] = s;
            if (used_lo) used_lo[lo_row_anchor ? lo_row_anchor[j] : j] = 1;
            any = true;
        }
        base += tot;
    }
    if (any && used_hi) used_hi[hi_row_anchor ? hi_row_anchor[i] : i] = 1;
}

// ---------------------------------------------------------------------------
// pose scoring
// ---------------------------------------------------------------------------

struct CellGrid {
    const int32_t *start;      // ncell + 1 offsets into pts
    const double *pts;         // sorted points, xyz
    const int32_t *ids;        // sorted point -> anchor id
    const uint8_t *used;       // per anchor: takes part in the lo cloud (or nullptr = all)
    double mn[3];
    double cell;
    int dim[3];
};

// compact the used anchors' coordinates (order immaterial for the count)
__global__ __launch_bounds__(1024) void k_compact_cloud(const double *__restrict__ subv, const uint8_t *__restrict__ used,
                                                        int n, double *__restrict__ cloud, int32_t *__restrict__ count) {
    __shared__ int wt[17];
    __shared__ int s_base;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    for (int b = 0; b < n; b += 1024) {
        const int i = b + threadIdx.x;
        const bool p = i < n && (!used || used[i]);
        int tot;
        const int pos = block_excl_scan(p ? 1 : 0, wt, &tot);
        if (p) {
            const int o = s_base + pos;
            cloud[3 * o] = subv[3 * i]; cloud[3 * o + 1] = subv[3 * i + 1]; cloud[3 * o + 2] = subv[3 * i + 2];
        }
        __syncthreads();
        if (threadIdx.x == 0) s_base += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *count = s_base;
}

__global__ void k_count_flags(const uint8_t *__restrict__ used, int n, int32_t *__restrict__ count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int v = (i < n && used[i]) ? 1 : 0;
    const int s = wave_sum_i32(v);
    if (lane_id() == 0 && s) atomicAdd(count, s);
}

#define POSE_THREADS 256

// MaD.py:433-448.  One wave per pair, lanes over the hi cloud (held in LDS).
__global__ __launch_bounds__(POSE_THREADS) void k_pose(const int32_t *__restrict__ pair_hi, const int32_t *__restrict__ pair_lo,
                                                       int64_t n_pairs, const double *__restrict__ hi_p,
                                                       const double *__restrict__ hi_R, const double *__restrict__ lo_p,
                                                       const double *__restrict__ lo_R, const int32_t *__restrict__ hi_row_anchor,
                                                       const int32_t *__restrict__ lo_row_anchor,
                                                       const double *__restrict__ hi_cloud, const int32_t *__restrict__ l_hi_ptr,
                                                       CellGrid G, double dist, int32_t *__restrict__ counts) {
    extern __shared__ __align__(16) unsigned char smem[];
    double *cl = (double *)smem;
    const int l_hi = *l_hi_ptr;
    for (int i = threadIdx.x; i < 3 * l_hi; i += POSE_THREADS) cl[i] = hi_cloud[i];
    __syncthreads();
    const int lane = lane_id();
    const int64_t wave = (int64_t)blockIdx.x * (POSE_THREADS / MAD_WAVE) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (POSE_THREADS / MAD_WAVE);
    const double inv_cell = 1.0 / G.cell;
    for (int64_t p = wave; p < n_pairs; p += nwaves) {
        const int ih = pair_hi[p], il = pair_lo[p];
        // R = inv(lo.Rfinal) @ hi.Rfinal (MaD.py:438); every lane computes it (uniform values)
        const double *m = lo_R + 9 * il;
        const double *h = hi_R + 9 * ih;
        const double c00 = m[4] * m[8] - m[5] * m[7];
        const double c01 = m[5] * m[6] - m[3] * m[8];
        const double c02 = m[3] * m[7] - m[4] * m[6];
        const double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
        const double id = 1.0 / det;
        double iv[9];
        iv[0] = c00 * id; iv[1] = (m[2] * m[7] - m[1] * m[8]) * id; iv[2] = (m[1] * m[5] - m[2] * m[4]) * id;
        iv[3] = c01 * id; iv[4] = (m[0] * m[8] - m[2] * m[6]) * id; iv[5] = (m[2] * m[3] - m[0] * m[5]) * id;
        iv[6] = c02 * id; iv[7] = (m[1] * m[6] - m[0] * m[7]) * id; iv[8] = (m[0] * m[4] - m[1] * m[3]) * id;
        double R[9];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) R[3 * i + j] = iv[3 * i] * h[j] + iv[3 * i + 1] * h[3 + j] + iv[3 * i + 2] * h[6 + j];
        const int ah = hi_row_anchor ? hi_row_anchor[ih] : ih, al = lo_row_anchor ? lo_row_anchor[il] : il;
        const double ph0 = hi_p[3 * ah], ph1 = hi_p[3 * ah + 1], ph2 = hi_p[3 * ah + 2];
        const double pl0 = lo_p[3 * al], pl1 = lo_p[3 * al + 1], pl2 = lo_p[3 * al + 2];
        int cnt = 0;
        for (int a = lane; a < l_hi; a += MAD_WAVE) {
            const double d0 = cl[3 * a] - ph0, d1 = cl[3 * a + 1] - ph1, d2 = cl[3 * a + 2] - ph2;
            const double x = (d0 * R[0] + d1 * R[1] + d2 * R[2]) + pl0;      // MaD.py:440-444
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
                                const double dd = e0 * e0 + e1 * e1 + e2 * e2;
                                if (sqrt(dd) < dist) { hit = true; break; }      // MaD.py:447-448
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


// ---- LDS-resident variant -----------------------------------------------------------------
// The lo cloud of one match (the used map anchors, a few thousand points at most) is binned
// into cells of edge >= 2 * dist, so the ball of radius dist around a query point meets at
// most 2 cells per axis (<= 4 z-runs).  Cloud, cell offsets (uint16) and the hi cloud all sit
// in LDS: a query costs LDS reads only.  Same predicate as k_pose, hence the same counts.
#define POSE_LDS_THREADS 1024

struct PoseGrid {
    double mn[3];
    double inv_cell[3];
    int dim[3];
    int ncell;
};

__device__ __forceinline__ int pg_cell(double v, double mn, double inv, int dim) {
    const int c = (int)floor((v - mn) * inv);
    return min(max(c, 0), dim - 1);
}

// one workgroup: counting sort of the used points into cells; sorted points and offsets to global
__global__ __launch_bounds__(1024) void k_pose_grid_build(const double *__restrict__ pts, const uint8_t *__restrict__ used, int n,
                                                          PoseGrid G, int32_t *__restrict__ cell_start, double *__restrict__ sorted,
                                                          int32_t *__restrict__ n_used) {
    extern __shared__ __align__(16) unsigned char smem[];
    int *cnt = (int *)smem;
    __shared__ int wt[17];
    __shared__ int carry;
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
    for (int base = 0; base < G.ncell; base += 1024) {
        const int c = base + threadIdx.x;
        const int v = c < G.ncell ? cnt[c] : 0;
        int tot;
        const int ex = block_excl_scan(v, wt, &tot);
        if (c < G.ncell) { cnt[c] = carry + ex; cell_start[c] = carry + ex; }
        __syncthreads();
        if (threadIdx.x == 0) carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) { cell_start[G.ncell] = carry; *n_used = carry; }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 1024)
        if (!used || used[i]) {
            const int c = (pg_cell(pts[3 * i], G.mn[0], G.inv_cell[0], G.dim[0]) * G.dim[1] +
                           pg_cell(pts[3 * i + 1], G.mn[1], G.inv_cell[1], G.dim[1])) * G.dim[2] +
                          pg_cell(pts[3 * i + 2], G.mn[2], G.inv_cell[2], G.dim[2]);
            const int o = atomicAdd(&cnt[c], 1);
            sorted[3 * o] = pts[3 * i]; sorted[3 * o + 1] = pts[3 * i + 1]; sorted[3 * o + 2] = pts[3 * i + 2];
        }
}

__global__ __launch_bounds__(POSE_LDS_THREADS) void k_pose_lds(const int32_t *__restrict__ pair_hi, const int32_t *__restrict__ pair_lo,
                                                           int64_t n_pairs, const double *__restrict__ hi_p,
                                                           const double *__restrict__ hi_R, const double *__restrict__ lo_p,
                                                           const double *__restrict__ lo_R, const int32_t *__restrict__ hi_row_anchor,
                                                           const int32_t *__restrict__ lo_row_anchor,
                                                           const double *__restrict__ hi_cloud, const int32_t *__restrict__ l_hi_ptr,
                                                           const double *__restrict__ lo_sorted, const int32_t *__restrict__ cell_start,
                                                           PoseGrid G, int l_hi_cap, int l_lo_cap, double dist,
                                                           int32_t *__restrict__ counts) {
    extern __shared__ __align__(16) unsigned char smem[];
    double *cl = (double *)smem;                                   // hi cloud
    double *lp = cl + 3 * (size_t)l_hi_cap;                        // sorted lo cloud
    unsigned short *cs = (unsigned short *)(lp + 3 * (size_t)l_lo_cap);      // cell offsets
    const int l_hi = *l_hi_ptr;
    const int l_lo = cell_start[G.ncell];
    for (int i = threadIdx.x; i < 3 * l_hi; i += POSE_LDS_THREADS) cl[i] = hi_cloud[i];
    for (int i = threadIdx.x; i < 3 * l_lo; i += POSE_LDS_THREADS) lp[i] = lo_sorted[i];
    for (int i = threadIdx.x; i <= G.ncell; i += POSE_LDS_THREADS) cs[i] = (unsigned short)cell_start[i];
    __syncthreads();
    const int lane = lane_id();
    const int64_t wave = (int64_t)blockIdx.x * (POSE_LDS_THREADS / MAD_WAVE) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (POSE_LDS_THREADS / MAD_WAVE);
    for (int64_t p = wave; p < n_pairs; p += nwaves) {
        const int ih = pair_hi[p], il = pair_lo[p];
        const double *m = lo_R + 9 * il;
        const double *h = hi_R + 9 * ih;
        const double c00 = m[4] * m[8] - m[5] * m[7];
        const double c01 = m[5] * m[6] - m[3] * m[8];
        const double c02 = m[3] * m[7] - m[4] * m[6];
        const double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
        const double id = 1.0 / det;
        double iv[9];
        iv[0] = c00 * id; iv[1] = (m[2] * m[7] - m[1] * m[8]) * id; iv[2] = (m[1] * m[5] - m[2] * m[4]) * id;
        iv[3] = c01 * id; iv[4] = (m[0] * m[8] - m[2] * m[6]) * id; iv[5] = (m[2] * m[3] - m[0] * m[5]) * id;
        iv[6] = c02 * id; iv[7] = (m[1] * m[6] - m[0] * m[7]) * id; iv[8] = (m[0] * m[4] - m[1] * m[3]) * id;
        double R[9];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) R[3 * i + j] = iv[3 * i] * h[j] + iv[3 * i + 1] * h[3 + j] + iv[3 * i + 2] * h[6 + j];
        const int ah = hi_row_anchor ? hi_row_anchor[ih] : ih, al = lo_row_anchor ? lo_row_anchor[il] : il;
        const double ph0 = hi_p[3 * ah], ph1 = hi_p[3 * ah + 1], ph2 = hi_p[3 * ah + 2];
        const double pl0 = lo_p[3 * al], pl1 = lo_p[3 * al + 1], pl2 = lo_p[3 * al + 2];
        int cnt = 0;
        for (int a = lane; a < l_hi; a += MAD_WAVE) {
            const double d0 = cl[3 * a] - ph0, d1 = cl[3 * a + 1] - ph1, d2 = cl[3 * a + 2] - ph2;
            const double x = (d0 * R[0] + d1 * R[1] + d2 * R[2]) + pl0;
            const double y = (d0 * R[3] + d1 * R[4] + d2 * R[5]) + pl1;
            const double z = (d0 * R[6] + d1 * R[7] + d2 * R[8]) + pl2;
            // cells met by the ball of radius dist (monotone in the coordinate, so no neighbour is missed)
            const int x0 = (int)floor(((x - dist) - G.mn[0]) * G.inv_cell[0]), x1 = (int)floor(((x + dist) - G.mn[0]) * G.inv_cell[0]);
            const int y0 = (int)floor(((y - dist) - G.mn[1]) * G.inv_cell[1]), y1 = (int)floor(((y + dist) - G.mn[1]) * G.inv_cell[1]);
            const int z0 = (int)floor(((z - dist) - G.mn[2]) * G.inv_cell[2]), z1 = (int)floor(((z + dist) - G.mn[2]) * G.inv_cell[2]);
            bool hit = false;
            if (x1 >= 0 && x0 < G.dim[0] && y1 >= 0 && y0 < G.dim[1] && z1 >= 0 && z0 < G.dim[2]) {
                const int zz0 = max(z0, 0), zz1 = min(z1, G.dim[2] - 1) + 1;
                const int xa = max(x0, 0), xb = min(x1, G.dim[0] - 1), ya = max(y0, 0), yb = min(y1, G.dim[1] - 1);
                // the ball meets at most 2 x 2 columns: fetch all four z-runs before walking any of them
                const int c00 = (xa * G.dim[1] + ya) * G.dim[2], c01 = (xa * G.dim[1] + yb) * G.dim[2];
                const int c10 = (xb * G.dim[1] + ya) * G.dim[2], c11 = (xb * G.dim[1] + yb) * G.dim[2];
                int s[4], e[4];
                s[0] = cs[c00 + zz0]; e[0] = cs[c00 + zz1];
                s[1] = cs[c01 + zz0]; e[1] = (yb != ya) ? cs[c01 + zz1] : s[1];
                s[2] = cs[c10 + zz0]; e[2] = (xb != xa) ? cs[c10 + zz1] : s[2];
                s[3] = cs[c11 + zz0]; e[3] = (xb != xa && yb != ya) ? cs[c11 + zz1] : s[3];
#pragma unroll
                for (int c = 0; c < 4; c++)
                    for (int q = s[c]; q < e[c] && !hit; q++) {
                        const double e0 = lp[3 * q] - x, e1 = lp[3 * q + 1] - y, e2 = lp[3 * q + 2] - z;
                        const double dd = e0 * e0 + e1 * e1 + e2 * e2;
                        hit = sqrt(dd) < dist;
                    }
            }
            cnt += hit ? 1 : 0;
        }
        cnt = wave_sum_i32(cnt);
        if (lane == 0) counts[p] = cnt;
    }
}

// rows of MaD.py:451 for the pairs listed in sel (or all pairs when sel == nullptr)
__global__ void k_results(const int64_t *__restrict__ sel, int64_t n_sel, const int32_t *__restrict__ pair_hi,
                          const int32_t *__restrict__ pair_lo, const double *__restrict__ pair_score,
                          const int32_t *__restrict__ counts, const int32_t *__restrict__ l_hi_ptr,
                          const double *__restrict__ hi_p, const double *__restrict__ hi_R, const int32_t *__restrict__ hi_meta,
                          const double *__restrict__ lo_p, const double *__restrict__ lo_R, const int32_t *__restrict__ lo_meta,
                          const int32_t *__restrict__ hi_row_anchor, const int32_t *__restrict__ lo_row_anchor,
                          double *__restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_sel) return;
    const int64_t p = sel ? sel[t] : t;
    const int ih = pair_hi[p], il = pair_lo[p];
    const double *m = lo_R + 9 * il, *h = hi_R + 9 * ih;
    const double c00 = m[4] * m[8] - m[5] * m[7];
    const double c01 = m[5] * m[6] - m[3] * m[8];
    const double c02 = m[3] * m[7] - m[4] * m[6];
    const double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    const double id = 1.0 / det;
    double iv[9];
    iv[0] = c00 * id; iv[1] = (m[2] * m[7] - m[1] * m[8]) * id; iv[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    iv[3] = c01 * id; iv[4] = (m[0] * m[8] - m[2] * m[6]) * id; iv[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    iv[6] = c02 * id; iv[7] = (m[1] * m[6] - m[0] * m[7]) * id; iv[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    double *o = out + MAD_RESULT_COLS * t;
    const int ah = hi_row_anchor ? hi_row_anchor[ih] : ih, al = lo_row_anchor ? lo_row_anchor[il] : il;
    o[0] = pair_score[p];
    o[1] = 100.0 * (double)counts[p] / (double)(*l_hi_ptr);
    // meta is {index, oct_scale, main_bin}: index and octave per anchor, main bin per row
    o[2] = lo_meta[3 * il]; o[3] = lo_meta[3 * il + 1]; o[4] = lo_meta[3 * il + 2];
    o[5] = hi_meta[3 * ih]; o[6] = hi_meta[3 * ih + 1]; o[7] = hi_meta[3 * ih + 2];
    o[8] = hi_p[3 * ah]; o[9] = hi_p[3 * ah + 1]; o[10] = hi_p[3 * ah + 2];
    o[11] = lo_p[3 * al]; o[12] = lo_p[3 * al + 1]; o[13] = lo_p[3 * al + 2];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) o[14 + 3 * i + j] = iv[3 * i] * h[j] + iv[3 * i + 1] * h[3 + j] + iv[3 * i + 2] * h[6 + j];
}

// ---------------------------------------------------------------------------
// top-k by (count desc, pair index asc)
// ---------------------------------------------------------------------------

// per-workgroup LDS histogram first: the counts crowd into a few bins, global atomics on them serialise
__global__ __launch_bounds__(256) void k_count_hist(const int32_t *__restrict__ counts, int64_t n, int32_t *__restrict__ hist, int nbins) {
    extern __shared__ __align__(16) unsigned char smem[];
    int *h = (int *)smem;
    for (int b = threadIdx.x; b < nbins; b += 256) h[b] = 0;
    __syncthreads();
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += step) atomicAdd(&h[min(max(counts[i], 0), nbins - 1)], 1);
    __syncthreads();
    for (int b = threadIdx.x; b < nbins; b += 256)
        if (h[b]) atomicAdd(&hist[b], h[b]);
}

// info[0] = threshold count c*, info[1] = number of pairs with count > c*, info[2] = ties to take at c*
__global__ void k_topk_threshold(const int32_t *__restrict__ hist, int nbins, int64_t k, int32_t *__restrict__ info) {
    if (threadIdx.x || blockIdx.x) return;
    int64_t above = 0;
    int c = nbins - 1;
    for (; c >= 0; c--) {
        if (above + hist[c] >= k) break;
        above += hist[c];
    }
    if (c < 0) { info[0] = -1; info[1] = (int32_t)above; info[2] = 0; return; }      // fewer than k pairs: take all
    info[0] = c; info[1] = (int32_t)above; info[2] = (int32_t)(k - above);
}

__global__ void k_tie_flags(const int32_t *__restrict__ counts, int64_t n, const int32_t *__restrict__ info,
                            int32_t *__restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = (counts[i] == info[0]) ? 1 : 0;
}

// keys = ((maxc - count) << 40) | pair index; survivors appended in any order, sorted afterwards
__global__ void k_topk_select(const int32_t *__restrict__ counts, int64_t n, const int32_t *__restrict__ info,
                              const int32_t *__restrict__ tie_rank, int maxc, unsigned long long *__restrict__ keys,
                              int32_t *__restrict__ n_keys) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = counts[i];
    const bool take = (c > info[0]) || (c == info[0] && tie_rank[i] < info[2]);
    if (take) {
        const int o = atomicAdd(n_keys, 1);
        keys[o] = ((unsigned long long)(maxc - c) << 40) | (unsigned long long)i;
    }
}

// one workgroup: bitonic sort of up to `cap` (power of two) 64-bit keys in LDS, emit pair indices
__global__ __launch_bounds__(1024) void k_topk_sort(const unsigned long long *__restrict__ keys,
                                                    const int32_t *__restrict__ n_keys, int cap, int64_t *__restrict__ order) {
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned long long *s = (unsigned long long *)smem;
    const int n = *n_keys;
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

// Selects the first k pairs of the (count desc, index asc) order; order_dev gets them sorted.
static int topk_device(mad_ctx *ctx, const int32_t *d_counts, int64_t n, int64_t k, int maxc, int64_t *d_order,
                       int64_t *n_out) {
    *n_out = 0;
    if (n <= 0 || k <= 0) return MAD_OK;
    if (k > n) k = n;
    if (k > 8192) return mad_fail(ctx, MAD_EINVAL, "top-k: k = %lld exceeds 8192", (long long)k);
    if (n >= ((int64_t)1 << 31)) return mad_fail(ctx, MAD_EINVAL, "top-k: %lld pairs", (long long)n);
    const int nbins = maxc + 1;
    MAD_TRY(mad_reserve(ctx, ctx->scratch[S_HIST], (size_t)(nbins + 8) * 4));
    MAD_TRY(mad_reserve(ctx, ctx->scratch[S_TIE_FLAG], (size_t)n * 4));
    MAD_TRY(mad_reserve(ctx, ctx->scratch[S_TIE_OFF], (size_t)(n + 1) * 4));
    MAD_TRY(mad_reserve(ctx, ctx->scratch[S_SEL], (size_t)(k + 8) * 8));
    int32_t *hist = scratch<int32_t>(ctx, S_HIST);
    int32_t *info = hist + nbins;            // 3 ints + n_keys
    int32_t *n_keys = info + 3;
    MAD_HIP(hipMemsetAsync(hist, 0, (size_t)(nbins + 8) * 4, ctx->stream));
    mad_timer_begin(ctx, MAD_T_TOPK);
    if (nbins > 16384) return mad_fail(ctx, MAD_EINVAL, "top-k: %d count bins", nbins);
    const int blocks = (int)std::min<int64_t>(mad_ceil_div(n, 1024), (int64_t)ctx->n_cu * 2);
    hipLaunchKernelGGL(k_count_hist, dim3(blocks), dim3(256), (size_t)nbins * 4, ctx->stream, d_counts, n, hist, nbins);
    hipLaunchKernelGGL(k_topk_threshold, dim3(1), dim3(64), 0, ctx->stream, hist, nbins, k, info);
    const unsigned nb = (unsigned)mad_ceil_div(n, 256);
    hipLaunchKernelGGL(k_tie_flags, dim3(nb), dim3(256), 0, ctx->stream, d_counts, n, info, scratch<int32_t>(ctx, S_TIE_FLAG));
    MAD_TRY(mad_scan_i32(ctx, scratch<int32_t>(ctx, S_TIE_FLAG), scratch<int32_t>(ctx, S_TIE_OFF), n));
    hipLaunchKernelGGL(k_topk_select, dim3(nb), dim3(256), 0, ctx->stream, d_counts, n, info,
                       scratch<int32_t>(ctx, S_TIE_OFF), maxc, scratch<unsigned long long>(ctx, S_SEL), n_keys);
    int cap = 1;
    while (cap < k) cap <<= 1;
    hipLaunchKernelGGL(k_topk_sort, dim3(1), dim3(1024), (size_t)cap * 8, ctx->stream,
                       scratch<unsigned long long>(ctx, S_SEL), n_keys, cap, d_order);
    mad_timer_end(ctx, MAD_T_TOPK);
    MAD_HIP(hipGetLastError());
    *n_out = k;
    return MAD_OK;
}

// ---------------------------------------------------------------------------
// cell list over a set of points (built on the device, dimensions from the host)
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

// h_pts: n x 3 host copy (for the bounding box); d_pts: the same on the device
static int build_cells(mad_ctx *ctx, const double *h_pts, const double *d_pts, int n, double cell, DevBuf &b_start,
                       DevBuf &b_pts, DevBuf &b_ids, double mn_out[3], int dim_out[3]) {
    double mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
    for (int i = 0; i < n; i++)
        for (int d = 0; d < 3; d++) {
            const double v = h_pts[3 * i + d];
            if (i == 0 || v < mn[d]) mn[d] = v;
            if (i == 0 || v > mx[d]) mx[d] = v;
        }
    size_t ncell = 1;
    for (int d = 0; d < 3; d++) {
        dim_out[d] = (int)floor((mx[d] - mn[d]) / cell) + 1;
        if (dim_out[d] < 1) dim_out[d] = 1;
        mn_out[d] = mn[d];
        ncell *= (size_t)dim_out[d];
    }
    if (ncell > ((size_t)1 << 28)) return mad_fail(ctx, MAD_EINVAL, "cell list of %zu cells is too large", ncell);
    MAD_TRY(mad_reserve(ctx, b_start, (ncell + 1) * 4));
    MAD_TRY(mad_reserve(ctx, b_pts, (size_t)(n > 0 ? n : 1) * 24));
    MAD_TRY(mad_reserve(ctx, b_ids, (size_t)(n > 0 ? n : 1) * 4));
    MAD_TRY(mad_reserve(ctx, ctx->scratch[S_TMP_C], (ncell + 1) * 4));
    MAD_TRY(mad_reserve(ctx, ctx->scratch[S_TMP_D], (size_t)(n > 0 ? n : 1) * 4));
    int32_t *cnt = scratch<int32_t>(ctx, S_TMP_C);
    int32_t *pt_cell = scratch<int32_t>(ctx, S_TMP_D);
    MAD_HIP(hipMemsetAsync(cnt, 0, (ncell + 1) * 4, ctx->stream));
    if (n > 0) {
        const unsigned nb = (unsigned)mad_ceil_div(n, 256);
        hipLaunchKernelGGL(k_cell_count, dim3(nb), dim3(256), 0, ctx->stream, d_pts, n, mn[0], mn[1], mn[2], 1.0 / cell,
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

int mad_build_cells(mad_ctx *ctx, mad_set *set, const double *h_subv, double cell) {
    for (int i = 0; i < set->n_anchors; i++)
        for (int d = 0; d < 3; d++) {
            const double v = h_subv[3 * i + d];
            if (i == 0 || v < set->bb_min[d]) set->bb_min[d] = v;
            if (i == 0 || v > set->bb_max[d]) set->bb_max[d] = v;
        }
    MAD_TRY(build_cells(ctx, h_subv, (const double *)set->anc_subv.p, set->n_anchors, cell, set->cell_start, set->cell_pts,
                        set->cell_ids, set->cell_min, set->cell_dim));
    set->cell_size = cell;
    set->cells_ready = true;
    return MAD_OK;
}

// ---------------------------------------------------------------------------
// correlation driver shared by the stage API and the set API
// ---------------------------------------------------------------------------

struct PairBufs {
    int64_t n_pairs;
};

// d_hi8/d_lo8 padded to the GEMM tile (zero rows), norms per row.  Fills S_PAIR_* and S_ROWOFF.
static int correlate_device(mad_ctx *ctx, const int8_t *d_hi8, int64_t n_hi, const int8_t *d_lo8, int64_t n_lo, int D,
                            const double *d_hn, const double *d_ln, double cc, const int32_t *d_hi_row_anchor,
                            const int32_t *d_lo_row_anchor, uint8_t *d_used_hi, uint8_t *d_used_lo, int64_t *n_pairs_out) {
    const int64_t hp = mad_ceil_div(n_hi, GEMM_BM) * GEMM_BM, lp = mad_ceil_div(n_lo, GEMM_BN) * GEMM_BN;
    if (D % GEMM_BK) return mad_fail(ctx, MAD_EINVAL, "correlate: D = %d is not a multiple of %d", D, GEMM_BK);
    MAD_TRY(mad_reserve(ctx, ctx->scratch[S_CMAT], (size_t)hp * lp * 4));
    MAD_TRY(mad_reserve(ctx, ctx->scratch[S_ROWCNT], (size_t)(n_hi + 1) * 4));
    MAD_TRY(mad_reserve(ctx, ctx->scratch[S_ROWOFF], (size_t)(n_hi + 2) * 4));
    int32_t *C = scratch<int32_t>(ctx, S_CMAT);
    mad_timer_begin(ctx, MAD_T_CORRELATE);
    hipLaunchKernelGGL(k_corr_gemm, dim3((unsigned)(lp / GEMM_BN), (unsigned)(hp / GEMM_BM)), dim3(GEMM_THREADS), 0,
                       ctx->stream, d_hi8, d_lo8, D, C, lp);
    mad_timer_end(ctx, MAD_T_CORRELATE);
    mad_timer_begin(ctx, MAD_T_PAIRS);
    hipLaunchKernelGGL(k_pair_count, dim3((unsigned)n_hi), dim3(256), 0, ctx->stream, C, lp, n_hi, n_lo, d_hn, d_ln, cc,
                       scratch<int32_t>(ctx, S_ROWCNT));
    MAD_TRY(mad_scan_i32(ctx, scratch<int32_t>(ctx, S_ROWCNT), scratch<int32_t>(ctx, S_ROWOFF), n_hi));
    MAD_HIP(hipMemcpyAsync(&ctx->pinned[0], scratch<int32_t>(ctx, S_ROWOFF) + n_hi, 4, hipMemcpyDeviceToHost, ctx->stream));
    MAD_HIP(hipStreamSynchronize(ctx->stream));
    const int64_t np = *(int32_t *)&ctx->pinned[0];
    *n_pairs_out = np;
    if (np > 0) {
        MAD_TRY(mad_reserve(ctx, ctx->scratch[S_PAIR_HI], (size_t)np * 4));
        MAD_TRY(mad_reserve(ctx, ctx->scratch[S_PAIR_LO], (size_t)np * 4));
        MAD_TRY(mad_reserve(ctx, ctx->scratch[S_PAIR_SCORE], (size_t)np * 8));
        hipLaunchKernelGGL(k_pair_emit, dim3((unsigned)n_hi), dim3(256), 0, ctx->stream, C, lp, n_hi, n_lo, d_hn, d_ln, cc,
                           scratch<int32_t>(ctx, S_ROWOFF), scratch<int32_t>(ctx, S_PAIR_HI),
                           scratch<int32_t>(ctx, S_PAIR_LO), scratch<double>(ctx, S_PAIR_SCORE), d_hi_row_anchor,
                           d_lo_row_anchor, d_used_hi, d_used_lo);
    }
    mad_timer_end(ctx, MAD_T_PAIRS);
    MAD_HIP(hipGetLastError());
    return MAD_OK;
}

// int16 rows on the device -> padded int8 + norms
static int pack_rows(mad_ctx *ctx, const int16_t *d_src, int64_t n, int D, DevBuf &b8, DevBuf &bn, int64_t tile) {
    const int64_t np = mad_ceil_div(n > 0 ? n : 1, tile) * tile;
    MAD_TRY(mad_reserve(ctx, b8, (size_t)np * D));
    MAD_TRY(mad_reserve(ctx, bn, (size_t)np * 8));
    MAD_TRY(mad_reserve(ctx, ctx->scratch[S_MISC], 256));
    int32_t *bad = scratch<int32_t>(ctx, S_MISC);
    MAD_HIP(hipMemsetAsync(bad, 0, 4, ctx->stream));
    MAD_HIP(hipMemsetAsync((int8_t *)b8.p + (size_t)n * D, 0, (size_t)(np - n) * D, ctx->stream));
    if (n > 0)
        hipLaunchKernelGGL(k_pack_rows, dim3((unsigned)mad_ceil_div(n, 4)), dim3(256), 0, ctx->stream, d_src, n, D,
                           (int8_t *)b8.p, (double *)bn.p, bad);
    MAD_HIP(hipMemcpyAsync(&ctx->pinned[2], bad, 4, hipMemcpyDeviceToHost, ctx->stream));
    MAD_HIP(hipStreamSynchronize(ctx->stream));
    if (*(int32_t *)&ctx->pinned[2]) return mad_fail(ctx, MAD_EDOM, "descriptor count outside the int8 range");
    return MAD_OK;
}

extern "C" int mad_correlate(mad_ctx *ctx, const int16_t *hi, int64_t n_hi, const int16_t *lo, int64_t n_lo, int D,
                             double cc, int32_t *pair_hi, int32_t *pair_lo, double *pair_score, int64_t *n_pairs,
                             int64_t cap) {
    if (!ctx || !n_pairs) return MAD_EINVAL;
    *n_pairs = 0;
    if (n_hi <= 0 || n_lo <= 0) return MAD_OK;
    if (!hi || !lo) return mad_fail(ctx, MAD_EINVAL, "mad_correlate: NULL descriptors");
    if (n_hi * n_lo >= ((int64_t)1 << 31)) return mad_fail(ctx, MAD_EINVAL, "mad_correlate: %lld x %lld too large", (long long)n_hi, (long long)n_lo);
    MAD_TRY(mad_reserve(ctx, ctx->scratch[S_HI16], (size_t)n_hi * D * 2));
    MAD_TRY(mad_reserve(ctx, ctx->scratch[S_LO16], (size_t)n_lo * D * 2));
    MAD_HIP(hipMemcpyAsync(ctx->scratch[S_HI16].p, hi, (size_t)n_hi * D * 2, hipMemcpyHostToDevice, ctx->stream));
    MAD_HIP(hipMemcpyAsync(ctx->scratch[S_LO16].p, lo, (size_t)n_lo * D * 2, hipMemcpyHostToDevice, ctx->stream));
    MAD_TRY(pack_rows(ctx, scratch<int16_t>(ctx, S_HI16), n_hi, D, ctx->scratch[S_HI8], ctx->scratch[S_HNORM], GEMM_BM));
    MAD_TRY(pack_rows(ctx, scratch<int16_t>(ctx, S_LO16), n_lo, D, ctx->scratch[S_LO8], ctx->scratch[S_LNORM], GEMM_BN));
    int64_t np = 0;
    MAD_TRY(correlate_device(ctx, scratch<int8_t>(ctx, S_HI8), n_hi, scratch<int8_t>(ctx, S_LO8), n_lo, D,
                             scratch<double>(ctx, S_HNORM), scratch<double>(ctx, S_LNORM), cc, nullptr, nullptr, nullptr,
                             nullptr, &np));
    *n_pairs = np;
    if (np > cap) return mad_fail(ctx, MAD_ENOSPC, "mad_correlate: %lld pairs, capacity %lld", (long long)np, (long long)cap);
    if (np > 0) {
        if (pair_hi) MAD_HIP(hipMemcpyAsync(pair_hi, ctx->scratch[S_PAIR_HI].p, np * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (pair_lo) MAD_HIP(hipMemcpyAsync(pair_lo, ctx->scratch[S_PAIR_LO].p, np * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (pair_score) MAD_HIP(hipMemcpyAsync(pair_score, ctx->scratch[S_PAIR_SCORE].p, np * 8, hipMemcpyDeviceToHost, ctx->stream));
        MAD_HIP(hipStreamSynchronize(ctx->stream));
    }
    return MAD_OK;
}

// ---------------------------------------------------------------------------
// pose scoring drivers
// ---------------------------------------------------------------------------

// lo cloud = the points of d_lo_pts whose flag is set (all if d_lo_used == nullptr); bounding box from the host.
static int pose_device(mad_ctx *ctx, const int32_t *d_pair_hi, const int32_t *d_pair_lo, int64_t n_pairs,
                       const double *d_hi_p, const double *d_hi_R, const double *d_lo_p, const double *d_lo_R,
                       const int32_t *d_hi_row_anchor, const int32_t *d_lo_row_anchor, const double *d_hi_cloud,
                       const int32_t *d_l_hi, int l_hi_max, const double *d_lo_pts, const uint8_t *d_lo_used, int n_lo_pts,
                       const double bb_min[3], const double bb_max[3], const CellGrid *fallback, double dist,
                       int32_t *d_counts, int32_t *d_l_lo) {
    if (n_pairs <= 0) return MAD_OK;
    // cells of edge >= 2 dist, at most 24 per axis
    PoseGrid G;
    G.ncell = 1;
    for (int d = 0; d < 3; d++) {
        const double ext = bb_max[d] - bb_min[d];
        double cell = 2.0 * dist;
        if (ext / cell > 24.0) cell = ext / 24.0;
        G.mn[d] = bb_min[d];
        G.inv_cell[d] = 1.0 / cell;
        G.dim[d] = (int)floor(ext / cell) + 1;
        if (G.dim[d] < 1) G.dim[d] = 1;
        G.ncell *= G.dim[d];
    }
    const size_t lds = (size_t)(l_hi_max + n_lo_pts) * 24 + (size_t)(G.ncell + 1) * 2 + 16;
    int blocks = (int)std::min<int64_t>(mad_ceil_div(n_pairs, POSE_THREADS / MAD_WAVE), (int64_t)ctx->n_cu * 8);
    if (blocks < 1) blocks = 1;
    if (lds <= 150 * 1024 && n_lo_pts < 65535 && G.ncell <= 30000) {
        MAD_TRY(mad_reserve(ctx, ctx->scratch[S_PG_START], (size_t)(G.ncell + 2) * 4));
        MAD_TRY(mad_reserve(ctx, ctx->scratch[S_PG_PTS], (size_t)(n_lo_pts + 1) * 24));
        mad_timer_begin(ctx, MAD_T_POSE);
        hipLaunchKernelGGL(k_pose_grid_build, dim3(1), dim3(1024), (size_t)G.ncell * 4, ctx->stream, d_lo_pts, d_lo_used, n_lo_pts, G,
                           scratch<int32_t>(ctx, S_PG_START), scratch<double>(ctx, S_PG_PTS), d_l_lo);
        static bool attr_set = false;
        if (!attr_set) {
            MAD_HIP(hipFuncSetAttribute((const void *)k_pose_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            MAD_HIP(hipFuncSetAttribute((const void *)k_pose_grid_build, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            attr_set = true;
        }
        const int lblocks = (int)std::max<int64_t>(1, std::min<int64_t>(mad_ceil_div(n_pairs, POSE_LDS_THREADS / MAD_WAVE), (int64_t)ctx->n_cu * 2));
        hipLaunchKernelGGL(k_pose_lds, dim3(lblocks), dim3(POSE_LDS_THREADS), lds, ctx->stream, d_pair_hi, d_pair_lo, n_pairs, d_hi_p,
                           d_hi_R, d_lo_p, d_lo_R, d_hi_row_anchor, d_lo_row_anchor, d_hi_cloud, d_l_hi,
                           scratch<double>(ctx, S_PG_PTS), scratch<int32_t>(ctx, S_PG_START), G, l_hi_max, n_lo_pts, dist,
                           d_counts);
        mad_timer_end(ctx, MAD_T_POSE);
        MAD_HIP(hipGetLastError());
        return MAD_OK;
    }
    // clouds too large for LDS: global cell list (cell = dist) with the used flags
    if (!fallback) return mad_fail(ctx, MAD_EINVAL, "pose: clouds of %d + %d points need the global cell list", l_hi_max, n_lo_pts);
    const size_t lds2 = (size_t)l_hi_max * 24;
    if (lds2 > 150 * 1024) return mad_fail(ctx, MAD_EINVAL, "pose: hi cloud of %d anchors does not fit LDS", l_hi_max);
    if (lds2 > 64 * 1024)
        MAD_HIP(hipFuncSetAttribute((const void *)k_pose, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    MAD_HIP(hipMemsetAsync(d_l_lo, 0, 4, ctx->stream));
    if (d_lo_used)
        hipLaunchKernelGGL(k_count_flags, dim3((unsigned)mad_ceil_div(n_lo_pts, 256)), dim3(256), 0, ctx->stream, d_lo_used, n_lo_pts, d_l_lo);
    else
        MAD_HIP(hipMemcpyAsync(d_l_lo, &n_lo_pts, 4, hipMemcpyHostToDevice, ctx->stream));
    mad_timer_begin(ctx, MAD_T_POSE);
    hipLaunchKernelGGL(k_pose, dim3(blocks), dim3(POSE_THREADS), lds2, ctx->stream, d_pair_hi, d_pair_lo, n_pairs, d_hi_p,
                       d_hi_R, d_lo_p, d_lo_R, d_hi_row_anchor, d_lo_row_anchor, d_hi_cloud, d_l_hi, *fallback, dist, d_counts);
    mad_timer_end(ctx, MAD_T_POSE);
    MAD_HIP(hipGetLastError());
    return MAD_OK;
}

extern "C" int mad_pose_score(mad_ctx *ctx, const int32_t *pair_hi, const int32_t *pair_lo, const double *pair_score,
                              int64_t n_pairs, const double *hi_p, const double *hi_R, const int32_t *hi_meta, int64_t n_hi,
                              const double *lo_p, const double *lo_R, const int32_t *lo_meta, int64_t n_lo,
                              const double *hi_cloud, int64_t l_hi, const double *lo_cloud, int64_t l_lo, double dist,
                              double *results, int32_t *counts) {
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
        MAD_TRY(mad_reserve(ctx, ctx->scratch[u.slot], u.bytes));
        MAD_HIP(hipMemcpyAsync(ctx->scratch[u.slot].p, u.src, u.bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    MAD_TRY(mad_reserve(ctx, ctx->scratch[S_COUNTS], (size_t)n_pairs * 4));
    MAD_TRY(mad_reserve(ctx, ctx->scratch[S_MISC], 256));
    int32_t *d_lhi = scratch<int32_t>(ctx, S_MISC) + 8;
    const int32_t lh = (int32_t)l_hi;
    MAD_HIP(hipMemcpyAsync(d_lhi, &lh, 4, hipMemcpyHostToDevice, ctx->stream));
    // bounding box of the lo cloud; the global cell list (cell = dist) is only built when the clouds do not fit LDS
    double bmn[3] = {0, 0, 0}, bmx[3] = {0, 0, 0};
    for (int64_t i = 0; i < l_lo; i++)
        for (int d = 0; d < 3; d++) {
            const double v = lo_cloud[3 * i + d];
            if (i == 0 || v < bmn[d]) bmn[d] = v;
            if (i == 0 || v > bmx[d]) bmx[d] = v;
        }
    DevBuf &b_start = ctx->scratch[S_CELL_START], &b_pts = ctx->scratch[S_CELL_PTS], &b_ids = ctx->scratch[S_CELL_IDS];
    double mn[3];
    int dim[3];
    MAD_TRY(build_cells(ctx, lo_cloud, scratch<double>(ctx, S_USED_LO), (int)l_lo, dist, b_start, b_pts, b_ids, mn, dim));
    CellGrid G;
    G.start = (const int32_t *)b_start.p; G.pts = (const double *)b_pts.p; G.ids = (const int32_t *)b_ids.p; G.used = nullptr;
    for (int d = 0; d < 3; d++) { G.mn[d] = mn[d]; G.dim[d] = dim[d]; }
    G.cell = dist;
    MAD_TRY(pose_device(ctx, scratch<int32_t>(ctx, S_PAIR_HI), scratch<int32_t>(ctx, S_PAIR_LO), n_pairs,
                        scratch<double>(ctx, S_TMP_E), scratch<double>(ctx, S_TMP_F), scratch<double>(ctx, S_TMP_H),
                        scratch<double>(ctx, S_TMP_I), nullptr, nullptr, scratch<double>(ctx, S_HI_CLOUD), d_lhi, (int)l_hi,
                        scratch<double>(ctx, S_USED_LO), nullptr, (int)l_lo, bmn, bmx, &G, dist, scratch<int32_t>(ctx, S_COUNTS),
                        d_lhi + 1));
    if (results) {
        MAD_TRY(mad_reserve(ctx, ctx->scratch[S_RESULTS], (size_t)n_pairs * MAD_RESULT_COLS * 8));
        hipLaunchKernelGGL(k_results, dim3((unsigned)mad_ceil_div(n_pairs, 256)), dim3(256), 0, ctx->stream, nullptr, n_pairs,
                           scratch<int32_t>(ctx, S_PAIR_HI), scratch<int32_t>(ctx, S_PAIR_LO),
                           scratch<double>(ctx, S_PAIR_SCORE), scratch<int32_t>(ctx, S_COUNTS), d_lhi,
                           scratch<double>(ctx, S_TMP_E), scratch<double>(ctx, S_TMP_F), scratch<int32_t>(ctx, S_TMP_G),
                           scratch<double>(ctx, S_TMP_H), scratch<double>(ctx, S_TMP_I), scratch<int32_t>(ctx, S_TMP_J),
                           nullptr, nullptr, scratch<double>(ctx, S_RESULTS));
        MAD_HIP(hipGetLastError());
        MAD_HIP(hipMemcpyAsync(results, ctx->scratch[S_RESULTS].p, (size_t)n_pairs * MAD_RESULT_COLS * 8,
                               hipMemcpyDeviceToHost, ctx->stream));
    }
    if (counts) MAD_HIP(hipMemcpyAsync(counts, ctx->scratch[S_COUNTS].p, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, ctx->stream));
    MAD_HIP(hipStreamSynchronize(ctx->stream));
    return MAD_OK;
}

extern "C" int mad_topk(mad_ctx *ctx, const int32_t *counts, int64_t n, int64_t k, int64_t *order) {
    if (!ctx) return MAD_EINVAL;
    if (n <= 0 || k <= 0) return MAD_OK;
    if (!counts || !order) return mad_fail(ctx, MAD_EINVAL, "mad_topk: NULL argument");
    int maxc = 0;
    for (int64_t i = 0; i < n; i++) {
        if (counts[i] < 0) return mad_fail(ctx, MAD_EINVAL, "mad_topk: negative count");
        if (counts[i] > maxc) maxc = counts[i];
    }
    MAD_TRY(mad_reserve(ctx, ctx->scratch[S_COUNTS], (size_t)n * 4));
    MAD_TRY(mad_reserve(ctx, ctx->scratch[S_SEL_OUT], (size_t)(k + 8) * 8));
    MAD_HIP(hipMemcpyAsync(ctx->scratch[S_COUNTS].p, counts, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    int64_t got = 0;
    MAD_TRY(topk_device(ctx, scratch<int32_t>(ctx, S_COUNTS), n, k, maxc, scratch<int64_t>(ctx, S_SEL_OUT), &got));
    MAD_HIP(hipMemcpyAsync(order, ctx->scratch[S_SEL_OUT].p, (size_t)got * 8, hipMemcpyDeviceToHost, ctx->stream));
    MAD_HIP(hipStreamSynchronize(ctx->stream));
    return MAD_OK;
}

// ---------------------------------------------------------------------------
// device-resident sets
// ---------------------------------------------------------------------------

extern "C" int mad_set_create(mad_ctx *ctx, mad_set **out) {
    if (!ctx || !out) return MAD_EINVAL;
    *out = new mad_set();
    return MAD_OK;
}

extern "C" void mad_set_destroy(mad_ctx *ctx, mad_set *s) {
    if (!s) return;
    if (ctx) (void)hipStreamSynchronize(ctx->stream);
    DevBuf *bufs[] = {&s->anc_coords, &s->anc_octave, &s->anc_subv, &s->anc_index, &s->row_anchor, &s->row_main, &s->row_sec,
                      &s->row_R, &s->dsc, &s->dsc8, &s->norm, &s->cell_start, &s->cell_pts, &s->cell_ids};
    for (DevBuf *b : bufs) mad_release(*b);
    delete s;
}

static int set_upload_anchors(mad_ctx *ctx, mad_set *s, const int32_t *anc_coords, const int32_t *anc_octave,
                              const double *anc_subv, const int32_t *anc_index, int n) {
    s->n_anchors = n;
    const size_t m = (size_t)(n > 0 ? n : 1);
    MAD_TRY(mad_reserve(ctx, s->anc_coords, m * 12));
    MAD_TRY(mad_reserve(ctx, s->anc_octave, m * 4));
    MAD_TRY(mad_reserve(ctx, s->anc_subv, m * 24));
    MAD_TRY(mad_reserve(ctx, s->anc_index, m * 4));
    if (n > 0) {
        if (anc_coords) MAD_HIP(hipMemcpyAsync(s->anc_coords.p, anc_coords, (size_t)n * 12, hipMemcpyHostToDevice, ctx->stream));
        MAD_HIP(hipMemcpyAsync(s->anc_octave.p, anc_octave, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
        MAD_HIP(hipMemcpyAsync(s->anc_subv.p, anc_subv, (size_t)n * 24, hipMemcpyHostToDevice, ctx->stream));
        MAD_HIP(hipMemcpyAsync(s->anc_index.p, anc_index, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    s->cells_ready = false;
    return MAD_OK;
}

// per-row meta {anchor index, octave, main bin} gathered on the fly by k_results through
// row_anchor; stored packed here so that one kernel serves both APIs
__global__ void k_row_meta(const int32_t *row_anchor, const int32_t *row_main, const int32_t *anc_index,
                           const int32_t *anc_octave, int64_t n, int32_t *meta) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int a = row_anchor[i];
    meta[3 * i] = anc_index[a]; meta[3 * i + 1] = anc_octave[a]; meta[3 * i + 2] = row_main[i];
}

static int set_finish_rows(mad_ctx *ctx, mad_set *s) {
    s->n_rows_pad = mad_ceil_div(s->n_rows > 0 ? s->n_rows : 1, GEMM_BM) * GEMM_BM;
    MAD_TRY(pack_rows(ctx, (const int16_t *)s->dsc.p, s->n_rows, s->D, s->dsc8, s->norm, GEMM_BM));
    return MAD_OK;
}

extern "C" int mad_set_build(mad_ctx *ctx, mad_set *s, const int *slot_of_octave, const int32_t *anc_coords,
                             const int32_t *anc_octave, const double *anc_subv, const int32_t *anc_index, int n, int r,
                             int lim_main, int lim_sec) {
    if (!ctx || !s || !slot_of_octave) return MAD_EINVAL;
    if (n > 0 && (!anc_coords || !anc_octave || !anc_subv || !anc_index)) return mad_fail(ctx, MAD_EINVAL, "mad_set_build: NULL anchors");
    FieldDev f[2] = {FieldDev{nullptr, 0, 0, 0}, FieldDev{nullptr, 0, 0, 0}};
    for (int o = 0; o < 2; o++) {
        const int sl = slot_of_octave[o];
        if (sl >= 0) {
            if (sl >= MAD_MAX_FIELDS || !ctx->fields[sl].tex) return mad_fail(ctx, MAD_EINVAL, "mad_set_build: field slot %d is empty", sl);
            f[o] = ctx->fields[sl];
        }
    }
    for (int i = 0; i < n; i++) {
        const int o = anc_octave[i];
        if ((o != 0 && o != 1) || !f[o].tex) return mad_fail(ctx, MAD_EINVAL, "mad_set_build: anchor %d has octave %d without a field", i, o);
    }
    MAD_TRY(set_upload_anchors(ctx, s, anc_coords, anc_octave, anc_subv, anc_index, n));
    s->D = 64 * ctx->eq_host[1].Z;
    s->n_rows = 0;
    int64_t rows = 0;
    MAD_TRY(mad_orient_device(ctx, f[0], f[1], (const int32_t *)s->anc_coords.p, (const int32_t *)s->anc_octave.p, 0, n, r,
                              lim_main, lim_sec, false, &rows, nullptr));
    s->n_rows = rows;
    const size_t m = (size_t)(rows > 0 ? rows : 1);
    MAD_TRY(mad_reserve(ctx, s->row_anchor, m * 4));
    MAD_TRY(mad_reserve(ctx, s->row_main, m * 4));
    MAD_TRY(mad_reserve(ctx, s->row_sec, m * 4));
    MAD_TRY(mad_reserve(ctx, s->row_R, m * 72));
    MAD_TRY(mad_reserve(ctx, s->dsc, m * s->D * 2));
    if (rows > 0) {
        MAD_HIP(hipMemcpyAsync(s->row_anchor.p, ctx->scratch[S_ROW_ANCHOR].p, rows * 4, hipMemcpyDeviceToDevice, ctx->stream));
        MAD_HIP(hipMemcpyAsync(s->row_main.p, ctx->scratch[S_ROW_MAIN].p, rows * 4, hipMemcpyDeviceToDevice, ctx->stream));
        MAD_HIP(hipMemcpyAsync(s->row_sec.p, ctx->scratch[S_ROW_SEC].p, rows * 4, hipMemcpyDeviceToDevice, ctx->stream));
        MAD_HIP(hipMemcpyAsync(s->row_R.p, ctx->scratch[S_ROW_R].p, rows * 72, hipMemcpyDeviceToDevice, ctx->stream));
        MAD_TRY(mad_describe_device(ctx, f[0], f[1], (const int32_t *)s->anc_coords.p, (const int32_t *)s->anc_octave.p, 0,
                                    (const int32_t *)s->row_anchor.p, (const double *)s->row_R.p, rows, r, (int16_t *)s->dsc.p));
    }
    MAD_TRY(set_finish_rows(ctx, s));
    MAD_TRY(mad_build_cells(ctx, s, anc_subv, 4.0));
    return MAD_OK;
}

extern "C" int mad_set_load(mad_ctx *ctx, mad_set *s, int64_t n_rows, const int32_t *row_anchor, const int32_t *row_main,
                            const double *row_R, const int16_t *dsc, int D, const double *anc_subv, const int32_t *anc_index,
                            const int32_t *anc_octave, int n_anchors) {
    if (!ctx || !s) return MAD_EINVAL;
    if (n_rows > 0 && (!row_anchor || !row_main || !row_R || !dsc)) return mad_fail(ctx, MAD_EINVAL, "mad_set_load: NULL rows");
    if (n_anchors > 0 && (!anc_subv || !anc_index || !anc_octave)) return mad_fail(ctx, MAD_EINVAL, "mad_set_load: NULL anchors");
    for (int64_t i = 0; i < n_rows; i++)
        if (row_anchor[i] < 0 || row_anchor[i] >= n_anchors) return mad_fail(ctx, MAD_EINVAL, "mad_set_load: row %lld -> anchor %d", (long long)i, row_anchor[i]);
    MAD_TRY(set_upload_anchors(ctx, s, nullptr, anc_octave, anc_subv, anc_index, n_anchors));
    s->D = D;
    s->n_rows = n_rows;
    const size_t m = (size_t)(n_rows > 0 ? n_rows : 1);
    MAD_TRY(mad_reserve(ctx, s->row_anchor, m * 4));
    MAD_TRY(mad_reserve(ctx, s->row_main, m * 4));
    MAD_TRY(mad_reserve(ctx, s->row_sec, m * 4));
    MAD_TRY(mad_reserve(ctx, s->row_R, m * 72));
    MAD_TRY(mad_reserve(ctx, s->dsc, m * D * 2));
    if (n_rows > 0) {
        MAD_HIP(hipMemcpyAsync(s->row_anchor.p, row_anchor, n_rows * 4, hipMemcpyHostToDevice, ctx->stream));
        MAD_HIP(hipMemcpyAsync(s->row_main.p, row_main, n_rows * 4, hipMemcpyHostToDevice, ctx->stream));
        MAD_HIP(hipMemsetAsync(s->row_sec.p, 0, n_rows * 4, ctx->stream));
        MAD_HIP(hipMemcpyAsync(s->row_R.p, row_R, n_rows * 72, hipMemcpyHostToDevice, ctx->stream));
        MAD_HIP(hipMemcpyAsync(s->dsc.p, dsc, (size_t)n_rows * D * 2, hipMemcpyHostToDevice, ctx->stream));
    }
    MAD_TRY(set_finish_rows(ctx, s));
    MAD_TRY(mad_build_cells(ctx, s, anc_subv, 4.0));
    return MAD_OK;
}

extern "C" int mad_set_size(mad_ctx *ctx, const mad_set *s, int64_t *n_rows, int32_t *n_anchors) {
    if (!ctx || !s) return MAD_EINVAL;
    if (n_rows) *n_rows = s->n_rows;
    if (n_anchors) *n_anchors = s->n_anchors;
    return MAD_OK;
}

extern "C" int mad_set_download(mad_ctx *ctx, const mad_set *s, int32_t *row_anchor, int32_t *row_main, int32_t *row_sec,
                                double *row_R, int16_t *dsc) {
    if (!ctx || !s) return MAD_EINVAL;
    const int64_t n = s->n_rows;
    if (n <= 0) return MAD_OK;
    if (row_anchor) MAD_HIP(hipMemcpyAsync(row_anchor, s->row_anchor.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (row_main) MAD_HIP(hipMemcpyAsync(row_main, s->row_main.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (row_sec) MAD_HIP(hipMemcpyAsync(row_sec, s->row_sec.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (row_R) MAD_HIP(hipMemcpyAsync(row_R, s->row_R.p, n * 72, hipMemcpyDeviceToHost, ctx->stream));
    if (dsc) MAD_HIP(hipMemcpyAsync(dsc, s->dsc.p, (size_t)n * s->D * 2, hipMemcpyDeviceToHost, ctx->stream));
    MAD_HIP(hipStreamSynchronize(ctx->stream));
    return MAD_OK;
}

extern "C" int mad_match_topk(mad_ctx *ctx, const mad_set *hi, const mad_set *lo, double cc, double dist, int64_t k,
                              double *results, int64_t *pair_index, int64_t *n_out, int64_t *stats) {
    if (!ctx || !hi || !lo || !n_out) return MAD_EINVAL;
    *n_out = 0;
    ctx->match = MatchState();
    ctx->match.n_hi_anchors = hi->n_anchors;
    ctx->match.n_lo_anchors = lo->n_anchors;
    if (stats) { stats[0] = 0; stats[1] = 0; stats[2] = 0; stats[3] = hi->n_rows * lo->n_rows; }
    if (hi->n_rows <= 0 || lo->n_rows <= 0) return MAD_OK;
    if (hi->D != lo->D) return mad_fail(ctx, MAD_EINVAL, "mad_match_topk: descriptor lengths %d vs %d", hi->D, lo->D);
    if (!(dist > 0)) return mad_fail(ctx, MAD_EINVAL, "mad_match_topk: dist must be positive");
    if (hi->n_rows * lo->n_rows >= ((int64_t)1 << 31)) return mad_fail(ctx, MAD_EINVAL, "mad_match_topk: %lld x %lld rows too large", (long long)hi->n_rows, (long long)lo->n_rows);
    if (!lo->cells_ready || lo->cell_size != dist) {
        // the cell list was built for another radius: rebuild it from a host copy of the anchors
        mad_set *l = const_cast<mad_set *>(lo);
        double *h = (double *)malloc((size_t)lo->n_anchors * 24 + 24);
        if (!h) return mad_fail(ctx, MAD_ENOMEM, "host allocation failed");
        hipError_t e = hipMemcpy(h, lo->anc_subv.p, (size_t)lo->n_anchors * 24, hipMemcpyDeviceToHost);
        int rc = e == hipSuccess ? mad_build_cells(ctx, l, h, dist) : mad_fail(ctx, MAD_EHIP, "anchor read-back failed");
        free(h);
        MAD_TRY(rc);
    }
    MAD_TRY(mad_reserve(ctx, ctx->scratch[S_USED_HI], (size_t)hi->n_anchors + 16));
    MAD_TRY(mad_reserve(ctx, ctx->scratch[S_USED_LO], (size_t)lo->n_anchors + 16));
    uint8_t *used_hi = scratch<uint8_t>(ctx, S_USED_HI), *used_lo = scratch<uint8_t>(ctx, S_USED_LO);
    MAD_HIP(hipMemsetAsync(used_hi, 0, (size_t)hi->n_anchors + 16, ctx->stream));
    MAD_HIP(hipMemsetAsync(used_lo, 0, (size_t)lo->n_anchors + 16, ctx->stream));
    int64_t np = 0;
    MAD_TRY(correlate_device(ctx, (const int8_t *)hi->dsc8.p, hi->n_rows, (const int8_t *)lo->dsc8.p, lo->n_rows, hi->D,
                             (const double *)hi->norm.p, (const double *)lo->norm.p, cc, (const int32_t *)hi->row_anchor.p,
                             (const int32_t *)lo->row_anchor.p, used_hi, used_lo, &np));
    ctx->match.n_pairs = np;
    if (stats) stats[0] = np;
    if (np == 0) return MAD_OK;
    // clouds: anchors that take part in at least one pair (MaD.py:427-428)
    MAD_TRY(mad_reserve(ctx, ctx->scratch[S_HI_CLOUD], (size_t)hi->n_anchors * 24 + 24));
    MAD_TRY(mad_reserve(ctx, ctx->scratch[S_MISC], 256));
    int32_t *d_lhi = scratch<int32_t>(ctx, S_MISC) + 8, *d_llo = scratch<int32_t>(ctx, S_MISC) + 9;
    MAD_HIP(hipMemsetAsync(d_llo, 0, 4, ctx->stream));
    hipLaunchKernelGGL(k_compact_cloud, dim3(1), dim3(1024), 0, ctx->stream, (const double *)hi->anc_subv.p, used_hi,
                       hi->n_anchors, scratch<double>(ctx, S_HI_CLOUD), d_lhi);
    MAD_TRY(mad_reserve(ctx, ctx->scratch[S_COUNTS], (size_t)np * 4));
    CellGrid G;
    G.start = (const int32_t *)lo->cell_start.p; G.pts = (const double *)lo->cell_pts.p; G.ids = (const int32_t *)lo->cell_ids.p;
    G.used = used_lo;
    for (int d = 0; d < 3; d++) { G.mn[d] = lo->cell_min[d]; G.dim[d] = lo->cell_dim[d]; }
    G.cell = lo->cell_size;
    MAD_TRY(pose_device(ctx, scratch<int32_t>(ctx, S_PAIR_HI), scratch<int32_t>(ctx, S_PAIR_LO), np,
                        (const double *)hi->anc_subv.p, (const double *)hi->row_R.p, (const double *)lo->anc_subv.p,
                        (const double *)lo->row_R.p, (const int32_t *)hi->row_anchor.p, (const int32_t *)lo->row_anchor.p,
                        scratch<double>(ctx, S_HI_CLOUD), d_lhi, hi->n_anchors, (const double *)lo->anc_subv.p, used_lo,
                        lo->n_anchors, lo->bb_min, lo->bb_max, &G, dist, scratch<int32_t>(ctx, S_COUNTS), d_llo));
    MAD_HIP(hipMemcpyAsync(&ctx->pinned[4], d_lhi, 8, hipMemcpyDeviceToHost, ctx->stream));      // l_hi and l_lo
    // top-k
    MAD_TRY(mad_reserve(ctx, ctx->scratch[S_SEL_OUT], (size_t)(k + 8) * 8));
    int64_t got = 0;
    MAD_TRY(topk_device(ctx, scratch<int32_t>(ctx, S_COUNTS), np, k, hi->n_anchors, scratch<int64_t>(ctx, S_SEL_OUT), &got));
    if (got > 0 && results) {
        // per-row meta for both sides
        MAD_TRY(mad_reserve(ctx, ctx->scratch[S_TMP_G], (size_t)hi->n_rows * 12));
        MAD_TRY(mad_reserve(ctx, ctx->scratch[S_TMP_J], (size_t)lo->n_rows * 12));
        hipLaunchKernelGGL(k_row_meta, dim3((unsigned)mad_ceil_div(hi->n_rows, 256)), dim3(256), 0, ctx->stream,
                           (const int32_t *)hi->row_anchor.p, (const int32_t *)hi->row_main.p, (const int32_t *)hi->anc_index.p,
                           (const int32_t *)hi->anc_octave.p, hi->n_rows, scratch<int32_t>(ctx, S_TMP_G));
        hipLaunchKernelGGL(k_row_meta, dim3((unsigned)mad_ceil_div(lo->n_rows, 256)), dim3(256), 0, ctx->stream,
                           (const int32_t *)lo->row_anchor.p, (const int32_t *)lo->row_main.p, (const int32_t *)lo->anc_index.p,
                           (const int32_t *)lo->anc_octave.p, lo->n_rows, scratch<int32_t>(ctx, S_TMP_J));
        MAD_TRY(mad_reserve(ctx, ctx->scratch[S_RESULTS], (size_t)got * MAD_RESULT_COLS * 8));
        hipLaunchKernelGGL(k_results, dim3((unsigned)mad_ceil_div(got, 256)), dim3(256), 0, ctx->stream,
                           scratch<int64_t>(ctx, S_SEL_OUT), got, scratch<int32_t>(ctx, S_PAIR_HI),
                           scratch<int32_t>(ctx, S_PAIR_LO), scratch<double>(ctx, S_PAIR_SCORE),
                           scratch<int32_t>(ctx, S_COUNTS), d_lhi, (const double *)hi->anc_subv.p, (const double *)hi->row_R.p,
                           scratch<int32_t>(ctx, S_TMP_G), (const double *)lo->anc_subv.p, (const double *)lo->row_R.p,
                           scratch<int32_t>(ctx, S_TMP_J), (const int32_t *)hi->row_anchor.p,
                           (const int32_t *)lo->row_anchor.p, scratch<double>(ctx, S_RESULTS));
        MAD_HIP(hipGetLastError());
        MAD_HIP(hipMemcpyAsync(results, ctx->scratch[S_RESULTS].p, (size_t)got * MAD_RESULT_COLS * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (got > 0 && pair_index) MAD_HIP(hipMemcpyAsync(pair_index, ctx->scratch[S_SEL_OUT].p, (size_t)got * 8, hipMemcpyDeviceToHost, ctx->stream));
    MAD_HIP(hipStreamSynchronize(ctx->stream));
    const int32_t *ll = (const int32_t *)&ctx->pinned[4];
    ctx->match.l_hi = ll[0];
    ctx->match.l_lo = ll[1];
    if (stats) { stats[1] = ll[0]; stats[2] = ll[1]; }
    *n_out = got;
    return MAD_OK;
}

extern "C" int mad_match_fetch(mad_ctx *ctx, int32_t *pair_hi, int32_t *pair_lo, double *pair_score, int32_t *counts,
                               int64_t cap) {
    if (!ctx) return MAD_EINVAL;
    const int64_t np = ctx->match.n_pairs;
    if (np > cap) return mad_fail(ctx, MAD_ENOSPC, "mad_match_fetch: %lld pairs, capacity %lld", (long long)np, (long long)cap);
    if (np <= 0) return MAD_OK;
    if (pair_hi) MAD_HIP(hipMemcpyAsync(pair_hi, ctx->scratch[S_PAIR_HI].p, np * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (pair_lo) MAD_HIP(hipMemcpyAsync(pair_lo, ctx->scratch[S_PAIR_LO].p, np * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (pair_score) MAD_HIP(hipMemcpyAsync(pair_score, ctx->scratch[S_PAIR_SCORE].p, np * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (counts) MAD_HIP(hipMemcpyAsync(counts, ctx->scratch[S_COUNTS].p, np * 4, hipMemcpyDeviceToHost, ctx->stream));
    MAD_HIP(hipStreamSynchronize(ctx->stream));
    return MAD_OK;
}

extern "C" int mad_match_results(mad_ctx *ctx, const mad_set *hi, const mad_set *lo, double *results, int64_t cap) {
    if (!ctx || !hi || !lo || !results) return MAD_EINVAL;
    const int64_t np = ctx->match.n_pairs;
    if (np > cap) return mad_fail(ctx, MAD_ENOSPC, "mad_match_results: %lld pairs, capacity %lld", (long long)np, (long long)cap);
    if (np <= 0) return MAD_OK;
    if (hi->n_anchors != ctx->match.n_hi_anchors || lo->n_anchors != ctx->match.n_lo_anchors)
        return mad_fail(ctx, MAD_EINVAL, "mad_match_results: sets differ from the last mad_match_topk call");
    MAD_TRY(mad_reserve(ctx, ctx->scratch[S_TMP_G], (size_t)hi->n_rows * 12));
    MAD_TRY(mad_reserve(ctx, ctx->scratch[S_TMP_J], (size_t)lo->n_rows * 12));
    MAD_TRY(mad_reserve(ctx, ctx->scratch[S_RESULTS], (size_t)np * MAD_RESULT_COLS * 8));
    int32_t *d_lhi = scratch<int32_t>(ctx, S_MISC) + 8;
    hipLaunchKernelGGL(k_row_meta, dim3((unsigned)mad_ceil_div(hi->n_rows, 256)), dim3(256), 0, ctx->stream,
                       (const int32_t *)hi->row_anchor.p, (const int32_t *)hi->row_main.p, (const int32_t *)hi->anc_index.p,
                       (const int32_t *)hi->anc_octave.p, hi->n_rows, scratch<int32_t>(ctx, S_TMP_G));
    hipLaunchKernelGGL(k_row_meta, dim3((unsigned)mad_ceil_div(lo->n_rows, 256)), dim3(256), 0, ctx->stream,
                       (const int32_t *)lo->row_anchor.p, (const int32_t *)lo->row_main.p, (const int32_t *)lo->anc_index.p,
                       (const int32_t *)lo->anc_octave.p, lo->n_rows, scratch<int32_t>(ctx, S_TMP_J));
    hipLaunchKernelGGL(k_results, dim3((unsigned)mad_ceil_div(np, 256)), dim3(256), 0, ctx->stream, (const int64_t *)nullptr, np,
                       scratch<int32_t>(ctx, S_PAIR_HI), scratch<int32_t>(ctx, S_PAIR_LO), scratch<double>(ctx, S_PAIR_SCORE),
                       scratch<int32_t>(ctx, S_COUNTS), d_lhi, (const double *)hi->anc_subv.p, (const double *)hi->row_R.p,
                       scratch<int32_t>(ctx, S_TMP_G), (const double *)lo->anc_subv.p, (const double *)lo->row_R.p,
                       scratch<int32_t>(ctx, S_TMP_J), (const int32_t *)hi->row_anchor.p, (const int32_t *)lo->row_anchor.p,
                       scratch<double>(ctx, S_RESULTS));
    MAD_HIP(hipGetLastError());
    MAD_HIP(hipMemcpyAsync(results, ctx->scratch[S_RESULTS].p, (size_t)np * MAD_RESULT_COLS * 8, hipMemcpyDeviceToHost, ctx->stream));
    MAD_HIP(hipStreamSynchronize(ctx->stream));
    return MAD_OK;
}

extern "C" int mad_match_used(mad_ctx *ctx, uint8_t *hi_used, int32_t n_hi_anchors, uint8_t *lo_used, int32_t n_lo_anchors) {
    if (!ctx) return MAD_EINVAL;
    if (n_hi_anchors != ctx->match.n_hi_anchors || n_lo_anchors != ctx->match.n_lo_anchors)
        return mad_fail(ctx, MAD_EINVAL, "mad_match_used: anchor counts do not match the last mad_match_topk call");
    if (hi_used && n_hi_anchors > 0) MAD_HIP(hipMemcpyAsync(hi_used, ctx->scratch[S_USED_HI].p, n_hi_anchors, hipMemcpyDeviceToHost, ctx->stream));
    if (lo_used && n_lo_anchors > 0) MAD_HIP(hipMemcpyAsync(lo_used, ctx->scratch[S_USED_LO].p, n_lo_anchors, hipMemcpyDeviceToHost, ctx->stream));
    MAD_HIP(hipStreamSynchronize(ctx->stream));
    return MAD_OK;
}
