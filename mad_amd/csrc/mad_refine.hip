// mad_refine.hip -- rigid-body refinement (a13), density simulation (a14-a15) and
// cross-correlation (a16) for gfx950.  Reference: mad/structure_utils.py:58-161,
// mad/PDB.py:131-292, mad/Dmap.py:153-258.
//
//  refine  : the <= 500 steps are sequentially dependent, so each candidate runs in ONE
//            persistent 1024-thread workgroup that keeps the rigid transform in LDS and
//            walks the steps without returning to the host; parallelism is over atoms
//            (trilinear gather of the np.gradient texels) and over candidates (one
//            workgroup each).  Reductions use a fixed tree, so results are reproducible.
//  density : float64 atomic splat, three separable float64 blur passes (the reference's
//            Gaussian is a product of 1-D Gaussians), float32 normalise + threshold.
//  ccc     : three float64 dot products over the overlap box.
#include <vector>

#include "mad_common.h"

#define RF_THREADS 1024
#define RF_WAVES (RF_THREADS / MAD_WAVE)
#define RF_THREADS_REG 512      // workgroups of the register-resident form: 8 waves, up to 256 registers each
#define RF_MAXA 8               // ... and the atoms a thread then holds

// ---------------------------------------------------------------------------
// np.gradient texels of the density map (structure_utils.py:80)
// ---------------------------------------------------------------------------

__device__ __forceinline__ float grad1(const float *g, int n, int i, size_t c, size_t st) {
    if (i == 0) return __fdiv_rn(__fsub_rn(g[c + st], g[c]), 1.0f);
    if (i == n - 1) return __fdiv_rn(__fsub_rn(g[c], g[c - st]), 1.0f);
    return __fdiv_rn(__fsub_rn(g[c + st], g[c - st]), 2.0f);
}

__global__ __launch_bounds__(256) void k_density_grad(const float *__restrict__ g, int nx, int ny, int nz,
                                                      float4 *__restrict__ out) {
    const size_t n = (size_t)nx * ny * nz;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += step) {
        const int z = (int)(i % nz), y = (int)((i / nz) % ny), x = (int)(i / ((size_t)nz * ny));
        out[i] = make_float4(grad1(g, nx, x, i, (size_t)ny * nz), grad1(g, ny, y, i, (size_t)nz), grad1(g, nz, z, i, 1), 0.f);
    }
}

extern "C" int mad_upload_density(mad_ctx *ctx, const float *grid, int nx, int ny, int nz, double ox, double oy, double oz,
                                  double voxsp) {
    if (!ctx || !grid) return MAD_EINVAL;
    if (nx < 2 || ny < 2 || nz < 2 || !(voxsp > 0)) return mad_fail(ctx, MAD_EINVAL, "mad_upload_density: dims %dx%dx%d vs %g", nx, ny, nz, voxsp);
    DensityDev &d = ctx->dens;
    const size_t n = (size_t)nx * ny * nz;
    MAD_HIP(hipStreamSynchronize(ctx->stream));
    if (d.grid) (void)hipFree(d.grid);
    if (d.grad) (void)hipFree(d.grad);
    d.grid = nullptr; d.grad = nullptr;
    if (hipMalloc((void **)&d.grid, n * sizeof(float)) != hipSuccess || hipMalloc((void **)&d.grad, n * sizeof(float4)) != hipSuccess)
        return mad_fail(ctx, MAD_ENOMEM, "mad_upload_density: %zu voxels", n);
    d.nx = nx; d.ny = ny; d.nz = nz; d.o[0] = ox; d.o[1] = oy; d.o[2] = oz; d.vs = voxsp;
    MAD_HIP(hipMemcpyAsync(d.grid, grid, n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    const int blocks = (int)std::min<size_t>(mad_ceil_div((int64_t)n, 256), (size_t)ctx->n_cu * 16);
    hipLaunchKernelGGL(k_density_grad, dim3(blocks), dim3(256), 0, ctx->stream, d.grid, nx, ny, nz, d.grad);
    MAD_HIP(hipGetLastError());
    MAD_HIP(hipStreamSynchronize(ctx->stream));
    return MAD_OK;
}

// ---------------------------------------------------------------------------
// refinement
// ---------------------------------------------------------------------------

struct RefineArgs {
    const float4 *grad;
    int nx, ny, nz;
    double o[3], del[3], vs;
    double *coords;      // n_cand x n_atoms x 3, in/out (current coordinates)
    double *init;        // n_cand x n_atoms x 3 scratch: start coordinates
    double *prev;        // n_cand x n_atoms x 3 scratch: coordinates at the last batch boundary
    int64_t n_atoms;
    const int64_t *cand_n, *cand_off;      // nullable: candidate c has cand_n[c] atoms starting at atom cand_off[c] of `coords` (else n_atoms each, back to back)
    int n_steps;
    double max_step, min_step;
    int32_t *converged, *last_step;
    // a candidate's atoms split over G workgroups (G > 1): per-candidate arrival counter and two slots of G x 8 partial sums
    int G;
    unsigned long long *arrive;      // n_cand, zeroed before the launch
    double *partial;                 // n_cand x 2 x G x 8
};

// fixed-order block reduction of NV doubles per thread (sum or max); result valid in all threads
template <int NV, bool IS_MAX>
__device__ __forceinline__ void block_reduce(double *v, double *lds /* RF_WAVES * NV + NV */) {
    const int lane = lane_id(), w = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; i++) v[i] = IS_MAX ? wave_max_f64(v[i]) : wave_sum_f64(v[i]);
    __syncthreads();
    if (lane == 0)
#pragma unroll
        for (int i = 0; i < NV; i++) lds[w * NV + i] = v[i];
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 0; i < NV; i++) {
            double a = lds[i];
            for (int k = 1; k < (int)(blockDim.x >> 6); k++) a = IS_MAX ? fmax(a, lds[k * NV + i]) : a + lds[k * NV + i];      // (workgroups of 16 or 8 waves)
            lds[RF_WAVES * NV + i] = a;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; i++) v[i] = lds[RF_WAVES * NV + i];
}

// The same reduction over the G workgroups that share one candidate (G = 1: the workgroup's own).  Every workgroup publishes
// its partial result, arrives on the candidate's counter and waits for the others; each then combines the G partials in the same
// fixed order, so all of them continue with bit-identical values and take the same branches.  Hand-off by 8-byte agent-scope
// atomics on both sides (MI355X_MICROARCH.md, inter-workgroup visibility); two slots, used alternately, because a workgroup may
// publish epoch e + 1 while a slower one still reads epoch e -- never further ahead, it cannot leave e + 1 before that one arrives.
// All G x n_cand workgroups are resident at once (mad_refine sizes G for that), so the wait cannot starve.
template <int NV, bool IS_MAX>
__device__ __forceinline__ void group_reduce(double *v, double *lds, const RefineArgs &A, int cand, int g, unsigned &epoch) {
    block_reduce<NV, IS_MAX>(v, lds);
    if (A.G == 1) return;
    __syncthreads();      // everybody has read the block result out of lds before thread 0 reuses it
    if (threadIdx.x == 0) {
        double *slot = A.partial + ((size_t)(cand * 2 + (epoch & 1)) * A.G) * 8;
        for (int i = 0; i < NV; i++) __hip_atomic_store(slot + g * 8 + i, v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(A.arrive + cand, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long want = (unsigned long long)A.G * (epoch + 1);
        while (__hip_atomic_load(A.arrive + cand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(2);
        for (int i = 0; i < NV; i++) {
            double a = __hip_atomic_load(slot + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int k = 1; k < A.G; k++) {
                const double b = __hip_atomic_load(slot + k * 8 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                a = IS_MAX ? fmax(a, b) : a + b;
            }
            lds[RF_WAVES * NV + i] = a;
        }
    }
    epoch++;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; i++) v[i] = lds[RF_WAVES * NV + i];
}

// math_utils.py:15-27
__device__ __forceinline__ void rod_mat(const double ax[3], double angle, double *m) {
    const double a = cos(angle / 2.0), s = sin(angle / 2.0);
    const double b = -ax[0] * s, c = -ax[1] * s, d = -ax[2] * s;
    const double aa = a * a, bb = b * b, cc = c * c, dd = d * d;
    const double bc = b * c, ad = a * d, ac = a * c, ab = a * b, bd = b * d, cd = c * d;
    m[0] = aa + bb - cc - dd; m[1] = 2 * (bc + ad); m[2] = 2 * (bd - ac);
    m[3] = 2 * (bc - ad); m[4] = aa + cc - bb - dd; m[5] = 2 * (cd + ab);
    m[6] = 2 * (bd + ac); m[7] = 2 * (cd - ab); m[8] = aa + dd - bb - cc;
}

// math_utils.py:5-13: a vector that cannot be normalised is returned as is
__device__ __forceinline__ void unit_vec(const double v[3], double o[3]) {
    const double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    if (n == 0.0 || n != n) { o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; return; }
    o[0] = v[0] / n; o[1] = v[1] / n; o[2] = v[2] / n;
}

// MAXA > 0: a thread keeps its (at most MAXA) atoms -- start, current and batch-boundary coordinates -- in registers for the whole run
// and writes the coordinates once, at the end (round 3; before, every step rewrote all coordinates twice: 2 GB of stores for 32
// candidates of 26 000 atoms).  MAXA = 0: the same loops over global memory, for batches whose atoms per thread exceed the registers.
template <int MAXA, class F>
__device__ __forceinline__ void rf_atoms(int64_t first, int64_t stride, int64_t n, F &&f) {
    if (MAXA > 0) {
#pragma unroll
        for (int s = 0; s < (MAXA > 0 ? MAXA : 1); s++) {
            const int64_t i = first + s * stride;
            if (i < n) f(s, i);
            __builtin_amdgcn_sched_barrier(0);      // one atom at a time: interleaved, the eight gathers of eight atoms spill
        }
    } else {
        for (int64_t i = first; i < n; i += stride) f(0, i);
    }
}

template <int MAXA>
__global__ __launch_bounds__(MAXA > 0 ? RF_THREADS_REG : RF_THREADS) void k_refine(RefineArgs A) {
    constexpr bool REG = MAXA > 0;
    double r_ini[REG ? MAXA : 1][3], r_cur[REG ? MAXA : 1][3], r_prv[REG ? MAXA : 1][3];
    __shared__ double red[RF_WAVES * 7 + 7];
    __shared__ double s_rot[9], s_trans[3], s_upd[12];      // s_upd: step translation (3) or step rotation (9) + centre
    __shared__ double s_step;
    const int cand = blockIdx.x / A.G, grp = blockIdx.x % A.G;
    const int tid = threadIdx.x;
    const int64_t first = (int64_t)grp * blockDim.x + tid, stride = (int64_t)A.G * blockDim.x;      // this thread's atoms: first, first + stride, ...
    unsigned epoch = 0;
    const int64_t n = A.cand_n ? A.cand_n[cand] : A.n_atoms;
    const int64_t off = A.cand_off ? A.cand_off[cand] : (int64_t)cand * n;
    double *cur = A.coords + (size_t)off * 3;
    double *ini = A.init + (size_t)off * 3;
    double *prv = A.prev + (size_t)off * 3;

    // structure_utils.py:65-67: start copy, centroid, farthest atom
    double v[7];
    v[0] = v[1] = v[2] = 0;
    rf_atoms<MAXA>(first, stride, n, [&](int s, int64_t i) {
        const double x = cur[3 * i], y = cur[3 * i + 1], z = cur[3 * i + 2];
        if (REG) {
            r_ini[s][0] = x; r_ini[s][1] = y; r_ini[s][2] = z; r_prv[s][0] = x; r_prv[s][1] = y; r_prv[s][2] = z;
            r_cur[s][0] = x; r_cur[s][1] = y; r_cur[s][2] = z;
        } else {
            ini[3 * i] = x; ini[3 * i + 1] = y; ini[3 * i + 2] = z;
            prv[3 * i] = x; prv[3 * i + 1] = y; prv[3 * i + 2] = z;
        }
        v[0] += x; v[1] += y; v[2] += z;
    });
    auto write_back = [&]() {      // REG: the coordinates leave the registers once
        if (REG) rf_atoms<MAXA>(first, stride, n, [&](int s, int64_t i) { cur[3 * i] = r_cur[s][0]; cur[3 * i + 1] = r_cur[s][1]; cur[3 * i + 2] = r_cur[s][2]; });
    };
    group_reduce<3, false>(v, red, A, cand, grp, epoch);
    const double cen0 = v[0] / (double)n, cen1 = v[1] / (double)n, cen2 = v[2] / (double)n;
    v[0] = 0;
    rf_atoms<MAXA>(first, stride, n, [&](int s, int64_t i) {
        const double a = (REG ? r_ini[s][0] : ini[3 * i]) - cen0, b = (REG ? r_ini[s][1] : ini[3 * i + 1]) - cen1, c = (REG ? r_ini[s][2] : ini[3 * i + 2]) - cen2;
        v[0] = fmax(v[0], sqrt(a * a + b * b + c * c));
    });
    group_reduce<1, true>(v, red, A, cand, grp, epoch);
    const double maxd = v[0];
    if (tid == 0) {
        for (int i = 0; i < 9; i++) s_rot[i] = (i % 4 == 0) ? 1.0 : 0.0;
        s_trans[0] = s_trans[1] = s_trans[2] = 0;
        s_step = A.max_step;
    }
    __syncthreads();

    int batch = 0, step = 0, conv = 0;
    for (step = 0; step < A.n_steps; step++) {
        const double r0 = s_rot[0], r1 = s_rot[1], r2 = s_rot[2], r3 = s_rot[3], r4 = s_rot[4], r5 = s_rot[5], r6 = s_rot[6],
                     r7 = s_rot[7], r8 = s_rot[8];
        const double ct0 = cen0 + s_trans[0], ct1 = cen1 + s_trans[1], ct2 = cen2 + s_trans[2];
        const double step_size = s_step;
        for (int i = 0; i < 7; i++) v[i] = 0;
        rf_atoms<MAXA>(first, stride, n, [&](int s, int64_t i) {
            // :91-96 re-apply the accumulated transform to the start coordinates
            const double a = (REG ? r_ini[s][0] : ini[3 * i]) - cen0, b = (REG ? r_ini[s][1] : ini[3 * i + 1]) - cen1, c = (REG ? r_ini[s][2] : ini[3 * i + 2]) - cen2;
            const double p0 = (a * r0 + b * r3 + c * r6) + ct0;
            const double p1 = (a * r1 + b * r4 + c * r7) + ct1;
            const double p2 = (a * r2 + b * r5 + c * r8) + ct2;
            if (REG) { r_cur[s][0] = p0; r_cur[s][1] = p1; r_cur[s][2] = p2; }
            else { cur[3 * i] = p0; cur[3 * i + 1] = p1; cur[3 * i + 2] = p2; }
            if (p0 != p0 || p1 != p1 || p2 != p2) v[6] = 1.0;
            // :101-103 atoms strictly inside the map
            const bool inside = (p0 > A.o[0]) && (p0 < A.o[0] + A.nx * A.vs - A.vs) && (p1 > A.o[1]) &&
                                (p1 < A.o[1] + A.ny * A.vs - A.vs) && (p2 > A.o[2]) && (p2 < A.o[2] + A.nz * A.vs - A.vs);
            if (!inside) return;
            // :106 trilinear interpolation of the gradient (scipy RegularGridInterpolator, linear)
            const double p[3] = {p0, p1, p2};
            const int dims[3] = {A.nx, A.ny, A.nz};
            int i0[3];
            double y[3];
#pragma unroll
            for (int d = 0; d < 3; d++) {
                int ii = (int)floor((p[d] - A.o[d]) / A.vs);
                ii = max(0, min(ii, dims[d] - 2));
                while (ii > 0 && p[d] < A.o[d] + ii * A.del[d]) ii--;
                while (ii < dims[d] - 2 && p[d] >= A.o[d] + (ii + 1) * A.del[d]) ii++;
                const double g0 = A.o[d] + ii * A.del[d], g1 = A.o[d] + (ii + 1) * A.del[d];
                i0[d] = ii;
                y[d] = (p[d] - g0) / (g1 - g0);
            }
            double g[3] = {0, 0, 0};
#pragma unroll
            for (int cx = 0; cx < 2; cx++)
#pragma unroll
                for (int cy = 0; cy < 2; cy++)
#pragma unroll
                    for (int cz = 0; cz < 2; cz++) {
                        double wgt = 1.0;
                        wgt = wgt * (cx ? y[0] : 1 - y[0]);
                        wgt = wgt * (cy ? y[1] : 1 - y[1]);
                        wgt = wgt * (cz ? y[2] : 1 - y[2]);
                        const float4 t = A.grad[((size_t)(i0[0] + cx) * A.ny + (i0[1] + cy)) * A.nz + (i0[2] + cz)];
                        g[0] = g[0] + (double)t.x * wgt;
                        g[1] = g[1] + (double)t.y * wgt;
                        g[2] = g[2] + (double)t.z * wgt;
                    }
            v[0] += g[0]; v[1] += g[1]; v[2] += g[2];
            // :121-122 torque about the START centroid
            const double c0 = p0 - cen0, c1 = p1 - cen1, c2 = p2 - cen2;
            v[3] += g[1] * c2 - g[2] * c1;
            v[4] += g[2] * c0 - g[0] * c2;
            v[5] += g[0] * c1 - g[1] * c0;
        });
        group_reduce<7, false>(v, red, A, cand, grp, epoch);
        if (v[6] != 0.0) {      // :97-98
            if (tid == 0 && grp == 0) { A.converged[cand] = 0; A.last_step[cand] = step; }
            write_back();
            return;
        }
        const bool is_trans = (step % 2) == 0;
        if (tid == 0) {
            if (is_trans) {      // :111-116
                double u[3];
                unit_vec(v, u);
                for (int d = 0; d < 3; d++) { u[d] *= step_size; s_upd[d] = u[d]; s_trans[d] += u[d]; }
            } else {             // :123-138
                double ax[3], sm[9], nr[9];
                unit_vec(v + 3, ax);
                rod_mat(ax, step_size / maxd, sm);
                for (int i = 0; i < 9; i++) s_upd[i] = sm[i];
                for (int i = 0; i < 3; i++)
                    for (int j = 0; j < 3; j++) nr[3 * i + j] = s_rot[3 * i] * sm[j] + s_rot[3 * i + 1] * sm[3 + j] + s_rot[3 * i + 2] * sm[6 + j];
                for (int i = 0; i < 9; i++) s_rot[i] = nr[i];
            }
        }
        __syncthreads();
        batch++;
        const bool batch_end = batch == 4;
        double mn = 0;
        if (is_trans) {
            const double u0 = s_upd[0], u1 = s_upd[1], u2 = s_upd[2];
            rf_atoms<MAXA>(first, stride, n, [&](int s, int64_t i) {
                const double x = (REG ? r_cur[s][0] : cur[3 * i]) + u0, yv = (REG ? r_cur[s][1] : cur[3 * i + 1]) + u1, z = (REG ? r_cur[s][2] : cur[3 * i + 2]) + u2;
                if (REG) { r_cur[s][0] = x; r_cur[s][1] = yv; r_cur[s][2] = z; }
                else { cur[3 * i] = x; cur[3 * i + 1] = yv; cur[3 * i + 2] = z; }
                if (batch_end) {
                    const double a = (REG ? r_prv[s][0] : prv[3 * i]) - x, b = (REG ? r_prv[s][1] : prv[3 * i + 1]) - yv, c = (REG ? r_prv[s][2] : prv[3 * i + 2]) - z;
                    mn = fmax(mn, sqrt(a * a + b * b + c * c));
                    if (REG) { r_prv[s][0] = x; r_prv[s][1] = yv; r_prv[s][2] = z; }
                    else { prv[3 * i] = x; prv[3 * i + 1] = yv; prv[3 * i + 2] = z; }
                }
            });
        } else {
            const double m0 = s_upd[0], m1 = s_upd[1], m2 = s_upd[2], m3 = s_upd[3], m4 = s_upd[4], m5 = s_upd[5],
                         m6 = s_upd[6], m7 = s_upd[7], m8 = s_upd[8];
            const double n0 = -1 * cen0 - s_trans[0], n1 = -1 * cen1 - s_trans[1], n2 = -1 * cen2 - s_trans[2];
            rf_atoms<MAXA>(first, stride, n, [&](int s, int64_t i) {
                const double a = (REG ? r_cur[s][0] : cur[3 * i]) + n0, b = (REG ? r_cur[s][1] : cur[3 * i + 1]) + n1, c = (REG ? r_cur[s][2] : cur[3 * i + 2]) + n2;
                const double x = (a * m0 + b * m3 + c * m6) + ct0;
                const double yv = (a * m1 + b * m4 + c * m7) + ct1;
                const double z = (a * m2 + b * m5 + c * m8) + ct2;
                if (REG) { r_cur[s][0] = x; r_cur[s][1] = yv; r_cur[s][2] = z; }
                else { cur[3 * i] = x; cur[3 * i + 1] = yv; cur[3 * i + 2] = z; }
                if (batch_end) {
                    const double pa = (REG ? r_prv[s][0] : prv[3 * i]) - x, pb = (REG ? r_prv[s][1] : prv[3 * i + 1]) - yv, pc = (REG ? r_prv[s][2] : prv[3 * i + 2]) - z;
                    mn = fmax(mn, sqrt(pa * pa + pb * pb + pc * pc));
                    if (REG) { r_prv[s][0] = x; r_prv[s][1] = yv; r_prv[s][2] = z; }
                    else { prv[3 * i] = x; prv[3 * i + 1] = yv; prv[3 * i + 2] = z; }
                }
            });
        }
        if (batch_end) {      // :141-147
            double mv[1] = {mn};
            group_reduce<1, true>(mv, red, A, cand, grp, epoch);
            if (tid == 0 && mv[0] < s_step) s_step *= 0.5;
            batch = 0;
        }
        __syncthreads();
        if (s_step < A.min_step) { conv = 1; break; }      // :150-152
    }
    if (step == A.n_steps) step = A.n_steps - 1;
    if (tid == 0 && grp == 0) { A.converged[cand] = conv; A.last_step[cand] = step; }
    write_back();
}

// a13 on coordinates that are already in S_TMP_E (n_cand x n_atoms x 3, refined in place); converged / last_step stay on the
// device (S_MISC + 256 bytes).  Asynchronous.
// cand_n / cand_off (device, nullable): candidates of different sizes (n_atoms is then the largest, total_atoms their sum).
static int refine_device(mad_ctx *ctx, int n_cand, int64_t n_atoms, int n_steps, double max_step, double min_step, int32_t **d_conv,
                         int32_t **d_last, const int64_t *cand_n = nullptr, const int64_t *cand_off = nullptr, int64_t total_atoms = 0) {
    const size_t bytes = (size_t)(cand_n ? total_atoms : (int64_t)n_cand * n_atoms) * 24;
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_MISC), 256 + (size_t)n_cand * 8));
    const DensityDev &d = ctx->dens;
    RefineArgs A;
    A.grad = d.grad; A.nx = d.nx; A.ny = d.ny; A.nz = d.nz; A.vs = d.vs;
    for (int i = 0; i < 3; i++) {
        A.o[i] = d.o[i];
        volatile double t = d.o[i] + d.vs;      // np.arange's element step: (start + step) - start
        A.del[i] = t - d.o[i];
    }
    A.coords = scratch<double>(ctx, S_TMP_E);
    A.n_atoms = n_atoms; A.n_steps = n_steps; A.max_step = max_step; A.min_step = min_step;
    A.cand_n = cand_n; A.cand_off = cand_off;
    A.converged = scratch<int32_t>(ctx, S_MISC) + 64;
    A.last_step = A.converged + n_cand;
    // A candidate is a chain of <= 500 dependent steps, and a batch of them is tens of workgroups on 256 CUs: split each
    // candidate's atoms over G workgroups that meet once per reduction (group_reduce).  G is such that ALL workgroups are
    // resident together even at one per CU -- the wait inside group_reduce relies on it -- and a workgroup keeps >= 2 atoms
    // per thread.  MAD_REFINE_SPLIT=1 forces the one-workgroup form (the summation order, hence the last bits, depend on G).
    static const int split_cap = getenv("MAD_REFINE_SPLIT") ? atoi(getenv("MAD_REFINE_SPLIT")) : 8;
    int G = std::max(1, std::min(std::min(split_cap, 8), ctx->n_cu / n_cand));
    while (G > 1 && n_atoms < (int64_t)G * RF_THREADS * 2) G--;
    A.G = G;
    // at most RF_MAXA atoms per thread of a 512-thread workgroup: they live in registers (k_refine<RF_MAXA>); more: the loops over
    // global memory, with their two scratch copies
    static const bool no_reg = getenv("MAD_REFINE_NO_REG") != nullptr;      // diagnostic switch
    const bool reg = !no_reg && mad_ceil_div(n_atoms, (int64_t)G * RF_THREADS_REG) <= RF_MAXA;
    if (!reg) {
        MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_TMP_F), bytes));
        MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_TMP_H), bytes));
    }
    A.init = reg ? nullptr : scratch<double>(ctx, S_TMP_F); A.prev = reg ? nullptr : scratch<double>(ctx, S_TMP_H);
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_TMP_I), (size_t)n_cand * 8 + (size_t)n_cand * 2 * G * 64 + 64));
    A.arrive = scratch<unsigned long long>(ctx, S_TMP_I);
    A.partial = (double *)(A.arrive + ((n_cand + 1) & ~1));
    MAD_HIP(hipMemsetAsync(A.arrive, 0, (size_t)n_cand * 8 + 16, ctx->stream));
    mad_timer_begin(ctx, MAD_T_REFINE);
    if (reg) hipLaunchKernelGGL(k_refine<RF_MAXA>, dim3(n_cand * G), dim3(RF_THREADS_REG), 0, ctx->stream, A);
    else hipLaunchKernelGGL(k_refine<0>, dim3(n_cand * G), dim3(RF_THREADS), 0, ctx->stream, A);
    mad_timer_end(ctx, MAD_T_REFINE);
    MAD_HIP(hipGetLastError());
    *d_conv = A.converged; *d_last = A.last_step;
    return MAD_OK;
}

extern "C" int mad_refine(mad_ctx *ctx, double *coords, int n_cand, int64_t n_atoms, int n_steps, double max_step,
                          double min_step, int32_t *converged, int32_t *last_step) {
    if (ctx) mad_use_lane(ctx, 0);
    if (!ctx) return MAD_EINVAL;
    if (!ctx->dens.grad) return mad_fail(ctx, MAD_EINVAL, "mad_refine: call mad_upload_density first");
    if (n_cand <= 0) return MAD_OK;
    if (!coords || !converged || !last_step || n_atoms <= 0 || n_steps < 0)
        return mad_fail(ctx, MAD_EINVAL, "mad_refine: bad argument");
    const size_t bytes = (size_t)n_cand * n_atoms * 24;
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_TMP_E), bytes));
    MAD_HIP(hipMemcpyAsync(mad_sb(ctx, S_TMP_E).p, coords, bytes, hipMemcpyHostToDevice, ctx->stream));
    int32_t *d_conv = nullptr, *d_last = nullptr;
    MAD_TRY(refine_device(ctx, n_cand, n_atoms, n_steps, max_step, min_step, &d_conv, &d_last));
    MAD_HIP(hipMemcpyAsync(coords, mad_sb(ctx, S_TMP_E).p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    MAD_HIP(hipMemcpyAsync(converged, d_conv, (size_t)n_cand * 4, hipMemcpyDeviceToHost, ctx->stream));
    MAD_HIP(hipMemcpyAsync(last_step, d_last, (size_t)n_cand * 4, hipMemcpyDeviceToHost, ctx->stream));
    MAD_HIP(hipStreamSynchronize(ctx->stream));
    return MAD_OK;
}

static long py_round(double v) { return (long)nearbyint(v); }      // python round(): half to even

// Dmap.py:163-230: overlap box of two grids in voxel units.  Returns false when the boxes do not overlap
// (Dmap.py:232-234); e[] may still hold a zero extent (0/0 -> NaN, as the reference).
static bool ccc_overlap(const int32_t d1[3], const double o1[3], const int32_t d2[3], const double o2[3], double voxsp, long mn1[3],
                        long mn2[3], long e[3]) {
    long mx1[3], mx2[3];
    bool empty = false;
    for (int d = 0; d < 3; d++) {
        const double a = o1[d] / voxsp, b = o2[d] / voxsp;
        if (a > b) { mn1[d] = 0; mn2[d] = py_round(a - b); }
        else if (a < b) { mn1[d] = py_round(b - a); mn2[d] = 0; }
        else { mn1[d] = 0; mn2[d] = 0; }
        if (a + d1[d] > b + d2[d]) { mx1[d] = py_round(b + d2[d] - a); mx2[d] = d2[d]; }
        else if (a + d1[d] < b + d2[d]) { mx1[d] = d1[d]; mx2[d] = py_round(a + d1[d] - b); }
        else { mx1[d] = d1[d]; mx2[d] = d2[d]; }
        if (mx1[d] - mn1[d] < 0) empty = true;      // Dmap.py:232-234
    }
    e[0] = e[1] = e[2] = 0;
    if (empty) return false;
    for (int d = 0; d < 3; d++) {      // python slice semantics
        const long a0 = mn1[d] < 0 ? 0 : mn1[d], a1 = mx1[d] > d1[d] ? d1[d] : mx1[d];
        const long b0 = mn2[d] < 0 ? 0 : mn2[d], b1 = mx2[d] > d2[d] ? d2[d] : mx2[d];
        const long ea = a1 - a0 > 0 ? a1 - a0 : 0, eb = b1 - b0 > 0 ? b1 - b0 : 0;
        e[d] = ea < eb ? ea : eb;
        mn1[d] = a0; mn2[d] = b0;
    }
    return true;
}

// ---------------------------------------------------------------------------
// density simulation (a14-a15) and cross-correlation (a16), batched: every kernel takes a table of jobs (blockIdx.y = job), so
// that the 32 candidates of a refinement batch -- 0.5 M voxels each, far too few to fill the chip one at a time -- go through
// seven launches together instead of 7 x 32, and nothing but poses, three sums per candidate and (when asked for) the final
// coordinates crosses the PCIe bus.  Round 3; before: one candidate at a time (2.3 + 1.1 ms of kernels for 32 candidates,
// and 40 ms of host work around them).
//
// Determinism: the splat accumulates in 2^-36 fixed point (integer atomics commute: the same grid whatever the order the atoms
// arrive in; |error| < 2^-37 per contribution against masses of 1-32 and a 2e-7 tolerance), the maxima are integer / bit-pattern
// atomics, and the three sums of a CCC come back as per-workgroup partials that the host adds in workgroup order.  Round 2 used
// float64 atomics for both and differed from run to run in the last bits.
// ---------------------------------------------------------------------------

#define SPLAT_FIX_BITS 36
#define CCC_WGS 64      // workgroups (partial sums) per CCC job

struct DJob {
    unsigned long long atom0, mass0, n_atoms;      // ranges in the atom / mass arrays of the batch
    unsigned long long off;                        // element offset of the job's volume in the float64 buffers
    unsigned long long off32;                      // ... and of its float32 output in the pool
    double mn[3];
    int p[3], dims[3];
    // overlap box with the uploaded map for the CCC (Dmap.py:163-230): starts in the map (s1) and in this volume (s2), extent
    int s1[3], s2[3], e[3];
    int ccc;      // 1: compute the three sums
};

// PDB.py:263-288: trilinear splat of the atomic masses; grid [px][py][pz], z fastest, 2^-36 fixed point
__global__ __launch_bounds__(256) void k_splat_b(const DJob *__restrict__ jobs, const double *__restrict__ atoms, const double *__restrict__ mass,
                                                 double vs, int margin, double *__restrict__ vol) {
    const DJob &J = jobs[blockIdx.y];
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= J.n_atoms) return;
    const double *a3 = atoms + 3 * (J.atom0 + i);
    long long *grid = (long long *)vol + J.off;
    const int px = J.p[0], py = J.p[1], pz = J.p[2];
    const double gx = margin + (a3[0] - J.mn[0]) / vs;
    const double gy = margin + (a3[1] - J.mn[1]) / vs;
    const double gz = margin + (a3[2] - J.mn[2]) / vs;
    const int x0 = (int)floor(gx), y0 = (int)floor(gy), z0 = (int)floor(gz);
    const double a = (x0 + 1) - gx, b = (y0 + 1) - gy, c = (z0 + 1) - gz, m = mass[J.mass0 + i];
    const double wx[2] = {a, 1 - a}, wy[2] = {b, 1 - b}, wz[2] = {c, 1 - c};
#pragma unroll
    for (int dx = 0; dx < 2; dx++)
#pragma unroll
        for (int dy = 0; dy < 2; dy++)
#pragma unroll
            for (int dz = 0; dz < 2; dz++) {
                const int x = x0 + dx, y = y0 + dy, z = z0 + dz;
                if (x < 0 || y < 0 || z < 0 || x >= px || y >= py || z >= pz) continue;
                const long long q = __double2ll_rn(ldexp(m * wx[dx] * wy[dy] * wz[dz], SPLAT_FIX_BITS));
                atomicAdd((unsigned long long *)&grid[((size_t)x * py + y) * pz + z], (unsigned long long)q);
            }
}

// maximum of a job's splat grid (fixed point, non-negative) -> mx[job]
__global__ __launch_bounds__(256) void k_max_b(const DJob *__restrict__ jobs, const double *__restrict__ vol, long long *__restrict__ mx) {
    __shared__ long long wt[4];
    const DJob &J = jobs[blockIdx.y];
    const long long *v = (const long long *)vol + J.off;
    const size_t n = (size_t)J.p[0] * J.p[1] * J.p[2];
    long long m = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m = max(m, v[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, MAD_WAVE));
    if (lane_id() == 0) wt[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(&mx[blockIdx.y], max(max(wt[0], wt[1]), max(wt[2], wt[3])));
}

// one separable full-mode pass along `axis` for every job; the job's input dims are p[] grown by 2r along the axes already done.
// FIRST: the input is the fixed-point splat, divided by its maximum first (PDB.py:290 grid / max)
template <bool FIRST>
__global__ __launch_bounds__(256) void k_blur_b(const DJob *__restrict__ jobs, const double *__restrict__ in_all, int axis, int r,
                                                const double *__restrict__ taps, const long long *__restrict__ mx, double *__restrict__ out_all) {
    const DJob &J = jobs[blockIdx.y];
    const int d0 = J.p[0] + (axis > 0 ? 2 * r : 0), d1 = J.p[1] + (axis > 1 ? 2 * r : 0), d2 = J.p[2];
    const int o0 = d0 + (axis == 0 ? 2 * r : 0), o1 = d1 + (axis == 1 ? 2 * r : 0), o2 = d2 + (axis == 2 ? 2 * r : 0);
    const size_t n = (size_t)o0 * o1 * o2;
    const double *in = in_all + J.off;
    double *out = out_all + J.off;
    const double inv = FIRST ? ldexp((double)mx[blockIdx.y], -SPLAT_FIX_BITS) : 1.0;
    const int dn = axis == 0 ? d0 : (axis == 1 ? d1 : d2);
    const size_t st = axis == 0 ? (size_t)d1 * d2 : (axis == 1 ? (size_t)d2 : 1);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int z = (int)(i % o2), y = (int)((i / o2) % o1), x = (int)(i / ((size_t)o2 * o1));
        const int pos = axis == 0 ? x : (axis == 1 ? y : z);
        // out[pos] = sum_t in[pos - t] * g[t], t = 0..2r  (full convolution)
        const int xi = axis == 0 ? 0 : x, yi = axis == 1 ? 0 : y, zi = axis == 2 ? 0 : z;
        const size_t base = ((size_t)xi * d1 + yi) * d2 + zi;
        double acc = 0;
        for (int t = 0; t <= 2 * r; t++) {
            const int s = pos - t;
            if (s < 0 || s >= dn) continue;
            double val;
            if (FIRST) val = ldexp((double)((const long long *)in)[base + (size_t)s * st], -SPLAT_FIX_BITS) / inv;
            else val = in[base + (size_t)s * st];
            acc += val * taps[t];
        }
        out[i] = acc;
    }
}

// float32 copy of the blurred volume into the pool + its maximum (bit pattern: valid for non-negative floats)
__global__ __launch_bounds__(256) void k_to_f32_b(const DJob *__restrict__ jobs, const double *__restrict__ in_all, float *__restrict__ pool,
                                                  unsigned *__restrict__ maxbits) {
    __shared__ float wm[4];
    const DJob &J = jobs[blockIdx.y];
    const double *in = in_all + J.off;
    float *out = pool + J.off32;
    const size_t n = (size_t)J.dims[0] * J.dims[1] * J.dims[2];
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float f = (float)in[i];
        out[i] = f;
        m = fmaxf(m, f);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, MAD_WAVE));
    if (lane_id() == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(&maxbits[blockIdx.y], __float_as_uint(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))));
}

// PDB.py:162-163: / max (float32), then zero below the isovalue; iso2 > -inf: the clamp of Dmap.get_CCC_with_grid on top (Dmap.py:160-161)
__global__ __launch_bounds__(256) void k_norm_b(const DJob *__restrict__ jobs, float *__restrict__ pool, const unsigned *__restrict__ maxbits, float iso,
                                                float iso2) {
    const DJob &J = jobs[blockIdx.y];
    float *g = pool + J.off32;
    const size_t n = (size_t)J.dims[0] * J.dims[1] * J.dims[2];
    const float mx = __uint_as_float(maxbits[blockIdx.y]);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float v = __fdiv_rn(g[i], mx);
        if (v < iso) v = 0.f;
        if (v < iso2) v = 0.f;
        g[i] = v;
    }
}

__global__ void k_clamp_f32(float *__restrict__ g, size_t n, float iso) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (g[i] < iso) g[i] = 0.f;
}

// Dmap.py:249-254 for every job with J.ccc: <a, b>, <a, a>, <b, b> over the overlap box of grid 1 (the map, [.][a1][a2], clamped at
// iso1 on the fly) and the job's volume.  part[job][workgroup][3]: plain stores, added up by the host in workgroup order.
__global__ __launch_bounds__(256) void k_ccc_b(const DJob *__restrict__ jobs, const float *__restrict__ g1, int a1, int a2,
                                               const float *__restrict__ pool, float iso1, double *__restrict__ part) {
    __shared__ double wt[4][3];
    const DJob &J = jobs[blockIdx.y];
    double o = 0, na = 0, nb = 0;
    if (J.ccc) {
        const float *g2 = pool + J.off32;
        const int b1 = J.dims[1], b2 = J.dims[2], e1 = J.e[1], e2 = J.e[2];
        const size_t n = (size_t)J.e[0] * e1 * e2;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
            const int z = (int)(i % e2), y = (int)((i / e2) % e1), x = (int)(i / ((size_t)e2 * e1));
            const float a_raw = g1[((size_t)(J.s1[0] + x) * a1 + (J.s1[1] + y)) * a2 + (J.s1[2] + z)];
            const double a = a_raw < iso1 ? 0.f : a_raw;      // grid 1 clamped on the fly (-inf: already clamped)
            const double b = g2[((size_t)(J.s2[0] + x) * b1 + (J.s2[1] + y)) * b2 + (J.s2[2] + z)];
            o += a * b; na += a * a; nb += b * b;
        }
    }
    o = wave_sum_f64(o); na = wave_sum_f64(na); nb = wave_sum_f64(nb);
    if (lane_id() == 0) { wt[threadIdx.x >> 6][0] = o; wt[threadIdx.x >> 6][1] = na; wt[threadIdx.x >> 6][2] = nb; }
    __syncthreads();
    if (threadIdx.x < 3)
        part[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 3 + threadIdx.x] = ((wt[0][threadIdx.x] + wt[1][threadIdx.x]) + wt[2][threadIdx.x]) + wt[3][threadIdx.x];
}

// geometry of the simulated density of one structure (PDB.py:144-145, 157-159, 237-257)
struct DensityPlan {
    double mn[3];        // lattice-aligned minimum of the atoms
    int p[3];            // splat grid
    int dims[3];         // blurred grid
    double origin[3];
    int r, margin;
    double sig;
};

static int density_plan_box(mad_ctx *ctx, const double mn_in[3], const double mx_in[3], double resolution, double voxsp, int pad, DensityPlan *P) {
    double mx[3];
    P->margin = 2 + pad;
    for (int d = 0; d < 3; d++) {
        if (!(mn_in[d] == mn_in[d]) || !(mx_in[d] == mx_in[d]) || !(fabs(mn_in[d]) < 1e12) || !(fabs(mx_in[d]) < 1e12))
            return mad_fail(ctx, MAD_EDOM, "mad_structure_to_density: NaN coordinate");
        P->mn[d] = voxsp * floor(mn_in[d] / voxsp);
        mx[d] = voxsp * ceil(mx_in[d] / voxsp);
        P->p[d] = (int)ceil((mx[d] - P->mn[d]) / voxsp) + 2 * P->margin + 1;
    }
    P->sig = resolution / (M_PI * sqrt(2.0)) / voxsp;
    P->r = (int)ceil(3.0 * P->sig);
    for (int d = 0; d < 3; d++) {
        P->dims[d] = P->p[d] + 2 * P->r;
        P->origin[d] = P->mn[d] - (P->r + P->margin) * voxsp;
    }
    if (P->r > 64) return mad_fail(ctx, MAD_EINVAL, "mad_structure_to_density: kernel radius %d", P->r);
    return MAD_OK;
}

static int density_plan(mad_ctx *ctx, const double *atoms, int64_t n, double resolution, double voxsp, int pad, DensityPlan *P) {
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = 0; i < n; i++)
        for (int d = 0; d < 3; d++) {
            const double v = atoms[3 * i + d];
            if (!(v == v)) return mad_fail(ctx, MAD_EDOM, "mad_structure_to_density: NaN coordinate");
            mn[d] = v < mn[d] ? v : mn[d];
            mx[d] = v > mx[d] ? v : mx[d];
        }
    return density_plan_box(ctx, mn, mx, resolution, voxsp, pad, P);
}

// 1-D taps; the 3-D kernel of PDB.py:148-150 is their outer product over the cube of their sum
static void density_taps(const DensityPlan &P, double *taps) {
    double ts = 0;
    for (int t = -P.r; t <= P.r; t++) { taps[t + P.r] = exp(-(double)(t * t) / (2.0 * P.sig * P.sig)); ts += taps[t + P.r]; }
    for (int t = 0; t <= 2 * P.r; t++) taps[t] /= ts;
}

// A batch of density simulations (+ CCC against the uploaded map) on atoms that are already on the device.
//   plans[j], atom0[j], mass0[j], n_atoms[j]: job j's geometry and its atoms / masses inside d_atoms / d_mass
//   pool, off32[j]: where job j's float32 volume goes (dims[j] voxels)
//   ccc_iso: clamp for the CCC (only when sums != nullptr: then sums[3 j ..] = <a,b>, <a,a>, <b,b> against ctx->dens, or NaN marks
//            for jobs whose box misses the map: empty[j])
// Scratch: S_TMP_H / S_TMP_I (float64 volumes), S_TMP_G (job table, maxima, taps, partial sums).  One synchronisation at the end
// when sums are asked for, none otherwise.
static int density_batch(mad_ctx *ctx, const std::vector<DensityPlan> &plans, const double *d_atoms, const double *d_mass,
                         const std::vector<unsigned long long> &atom0, const std::vector<unsigned long long> &mass0,
                         const std::vector<unsigned long long> &n_atoms, double voxsp, double isovalue, float *pool,
                         const std::vector<unsigned long long> &off32, double ccc_iso, double *sums, std::vector<char> *empty) {
    const int n_jobs = (int)plans.size();
    if (n_jobs == 0) return MAD_OK;
    const bool want_ccc = sums != nullptr;
    const DensityDev &M = ctx->dens;
    double taps[129];
    density_taps(plans[0], taps);      // sigma and radius depend on resolution and voxel spacing only
    const int r = plans[0].r, margin = plans[0].margin;
    // jobs go through in chunks whose float64 volumes fit 2 x 4 GB
    const size_t chunk_cap = (size_t)512 << 20;
    if (empty) empty->assign(n_jobs, 0);
    std::vector<DJob> jobs(n_jobs);
    for (int j0 = 0; j0 < n_jobs;) {
        size_t tot = 0, n_max = 0, np_max = 0, no_max = 0;
        int j1 = j0;
        while (j1 < n_jobs) {
            const DensityPlan &P = plans[j1];
            const size_t no = (size_t)P.dims[0] * P.dims[1] * P.dims[2];
            if (j1 > j0 && (tot + no > chunk_cap || j1 - j0 >= 32768)) break;
            DJob &J = jobs[j1];
            J.atom0 = atom0[j1]; J.mass0 = mass0[j1]; J.n_atoms = n_atoms[j1]; J.off = tot; J.off32 = off32[j1];
            for (int d = 0; d < 3; d++) { J.mn[d] = P.mn[d]; J.p[d] = P.p[d]; J.dims[d] = P.dims[d]; J.s1[d] = J.s2[d] = J.e[d] = 0; }
            J.ccc = 0;
            if (want_ccc) {
                const int32_t d1[3] = {M.nx, M.ny, M.nz}, d2[3] = {P.dims[0], P.dims[1], P.dims[2]};
                long mn1[3], mn2[3], e[3];
                const bool none = !ccc_overlap(d1, M.o, d2, P.origin, voxsp, mn1, mn2, e);
                if (empty) (*empty)[j1] = none ? 1 : 0;
                if (!none && e[0] > 0 && e[1] > 0 && e[2] > 0) {
                    J.ccc = 1;
                    for (int d = 0; d < 3; d++) { J.s1[d] = (int)mn1[d]; J.s2[d] = (int)mn2[d]; J.e[d] = (int)e[d]; }
                }
            }
            tot += no;
            n_max = std::max(n_max, (size_t)n_atoms[j1]);
            np_max = std::max(np_max, (size_t)P.p[0] * P.p[1] * P.p[2]);
            no_max = std::max(no_max, no);
            j1++;
        }
        const int nj = j1 - j0;
        MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_TMP_H), tot * 8));
        MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_TMP_I), tot * 8));
        const size_t b_jobs = (size_t)nj * sizeof(DJob), b_max = ((size_t)nj * 12 + 15) & ~(size_t)15, b_taps = 129 * 8,
                     b_part = want_ccc ? (size_t)nj * CCC_WGS * 24 : 0;
        MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_TMP_G), b_jobs + b_max + b_taps + b_part + 64));
        char *blk = scratch<char>(ctx, S_TMP_G);
        DJob *d_jobs = (DJob *)blk;
        long long *d_mx = (long long *)(blk + b_jobs);
        unsigned *d_mxf = (unsigned *)(blk + b_jobs + (size_t)nj * 8);
        double *d_taps = (double *)(blk + b_jobs + b_max);
        double *d_part = (double *)(blk + b_jobs + b_max + b_taps);
        double *bufA = scratch<double>(ctx, S_TMP_H), *bufB = scratch<double>(ctx, S_TMP_I);
        MAD_HIP(hipMemcpyAsync(d_jobs, jobs.data() + j0, b_jobs, hipMemcpyHostToDevice, ctx->stream));
        MAD_HIP(hipMemcpyAsync(d_taps, taps, sizeof(double) * (2 * r + 1), hipMemcpyHostToDevice, ctx->stream));
        mad_zero_words3(ctx, bufA, tot * 8, d_mx, b_max, nullptr, 0);
        mad_timer_begin(ctx, MAD_T_DENSITY);
        const unsigned gy = (unsigned)nj;
        const unsigned gb = (unsigned)std::max<size_t>(1, std::min<size_t>(mad_ceil_div((int64_t)no_max, 1024), (size_t)ctx->n_cu * 8 / std::min<size_t>(nj, 64) + 1));
        hipLaunchKernelGGL(k_splat_b, dim3((unsigned)mad_ceil_div((int64_t)n_max, 256), gy), dim3(256), 0, ctx->stream, d_jobs, d_atoms, d_mass, voxsp,
                           margin, bufA);
        hipLaunchKernelGGL(k_max_b, dim3(std::min<unsigned>(gb, (unsigned)mad_ceil_div((int64_t)np_max, 256)), gy), dim3(256), 0, ctx->stream, d_jobs,
                           bufA, d_mx);
        hipLaunchKernelGGL(k_blur_b<true>, dim3(gb, gy), dim3(256), 0, ctx->stream, d_jobs, bufA, 0, r, d_taps, d_mx, bufB);
        hipLaunchKernelGGL(k_blur_b<false>, dim3(gb, gy), dim3(256), 0, ctx->stream, d_jobs, bufB, 1, r, d_taps, d_mx, bufA);
        hipLaunchKernelGGL(k_blur_b<false>, dim3(gb, gy), dim3(256), 0, ctx->stream, d_jobs, bufA, 2, r, d_taps, d_mx, bufB);
        hipLaunchKernelGGL(k_to_f32_b, dim3(gb, gy), dim3(256), 0, ctx->stream, d_jobs, bufB, pool, d_mxf);
        hipLaunchKernelGGL(k_norm_b, dim3(gb, gy), dim3(256), 0, ctx->stream, d_jobs, pool, d_mxf, (float)isovalue,
                           want_ccc ? (float)ccc_iso : -INFINITY);
        mad_timer_end(ctx, MAD_T_DENSITY);
        if (want_ccc) {
            mad_timer_begin(ctx, MAD_T_CCC);
            hipLaunchKernelGGL(k_ccc_b, dim3(CCC_WGS, gy), dim3(256), 0, ctx->stream, d_jobs, (const float *)M.grid, M.ny, M.nz, pool, (float)ccc_iso,
                               d_part);
            mad_timer_end(ctx, MAD_T_CCC);
            std::vector<double> h((size_t)nj * CCC_WGS * 3);
            MAD_HIP(hipGetLastError());
            MAD_HIP(hipMemcpyAsync(h.data(), d_part, b_part, hipMemcpyDeviceToHost, ctx->stream));
            MAD_HIP(hipStreamSynchronize(ctx->stream));
            for (int j = 0; j < nj; j++)
                for (int q = 0; q < 3; q++) {
                    double s = 0;
                    for (int w = 0; w < CCC_WGS; w++) s += h[((size_t)j * CCC_WGS + w) * 3 + q];      // workgroup order: the same sum every run
                    sums[3 * (size_t)(j0 + j) + q] = s;
                }
        }
        MAD_HIP(hipGetLastError());
        j0 = j1;
    }
    return MAD_OK;
}

extern "C" int mad_structure_to_density(mad_ctx *ctx, const double *atoms, const double *mass, int64_t n, double resolution,
                                        double voxsp, double isovalue, int pad, int32_t dims[3], double origin[3], float *grid) {
    if (ctx) mad_use_lane(ctx, 0);
    if (!ctx || !atoms || !mass || !dims || !origin || n <= 0) return ctx ? mad_fail(ctx, MAD_EINVAL, "mad_structure_to_density: bad argument") : MAD_EINVAL;
    if (!(voxsp > 0) || !(resolution > 0) || pad < 0) return mad_fail(ctx, MAD_EINVAL, "mad_structure_to_density: resolution %g voxsp %g pad %d", resolution, voxsp, pad);
    std::vector<DensityPlan> plans(1);
    MAD_TRY(density_plan(ctx, atoms, n, resolution, voxsp, pad, &plans[0]));
    const DensityPlan &P = plans[0];
    for (int d = 0; d < 3; d++) { dims[d] = P.dims[d]; origin[d] = P.origin[d]; }
    if (!grid) return MAD_OK;
    const size_t no = (size_t)dims[0] * dims[1] * dims[2];
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_TMP_E), (size_t)n * 24));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_TMP_F), (size_t)n * 8));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_TMP_J), no * 4));
    double *d_atoms = scratch<double>(ctx, S_TMP_E), *d_mass = scratch<double>(ctx, S_TMP_F);
    float *d_out = scratch<float>(ctx, S_TMP_J);
    MAD_HIP(hipMemcpyAsync(d_atoms, atoms, (size_t)n * 24, hipMemcpyHostToDevice, ctx->stream));
    MAD_HIP(hipMemcpyAsync(d_mass, mass, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    MAD_TRY(density_batch(ctx, plans, d_atoms, d_mass, {0}, {0}, {(unsigned long long)n}, voxsp, isovalue, d_out, {0}, 0.0, nullptr, nullptr));
    MAD_HIP(hipMemcpyAsync(grid, d_out, no * 4, hipMemcpyDeviceToHost, ctx->stream));
    MAD_HIP(hipStreamSynchronize(ctx->stream));
    return MAD_OK;
}


extern "C" int mad_ccc(mad_ctx *ctx, float *grid1, const int32_t d1[3], const double o1[3], float *grid2,
                       const int32_t d2[3], const double o2[3], double voxsp, double isovalue, double *ccc) {
    if (ctx) mad_use_lane(ctx, 0);
    if (!ctx || !grid1 || !grid2 || !d1 || !d2 || !o1 || !o2 || !ccc) return ctx ? mad_fail(ctx, MAD_EINVAL, "mad_ccc: NULL argument") : MAD_EINVAL;
    *ccc = 0.0;
    const size_t n1 = (size_t)d1[0] * d1[1] * d1[2], n2 = (size_t)d2[0] * d2[1] * d2[2];
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_TMP_H), n1 * 4));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_TMP_I), n2 * 4));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_TMP_G), sizeof(DJob) + CCC_WGS * 24 + 64));
    float *g1 = scratch<float>(ctx, S_TMP_H), *g2 = scratch<float>(ctx, S_TMP_I);
    DJob *d_job = scratch<DJob>(ctx, S_TMP_G);
    double *d_part = (double *)(scratch<char>(ctx, S_TMP_G) + ((sizeof(DJob) + 15) & ~(size_t)15));
    MAD_HIP(hipMemcpyAsync(g1, grid1, n1 * 4, hipMemcpyHostToDevice, ctx->stream));
    MAD_HIP(hipMemcpyAsync(g2, grid2, n2 * 4, hipMemcpyHostToDevice, ctx->stream));
    const int gb = ctx->n_cu * 8;
    mad_timer_begin(ctx, MAD_T_CCC);
    // Dmap.py:160-161: both grids are clamped in place
    hipLaunchKernelGGL(k_clamp_f32, dim3(gb), dim3(256), 0, ctx->stream, g1, n1, (float)isovalue);
    hipLaunchKernelGGL(k_clamp_f32, dim3(gb), dim3(256), 0, ctx->stream, g2, n2, (float)isovalue);
    long mn1[3], mn2[3], e[3];
    const bool empty = !ccc_overlap(d1, o1, d2, o2, voxsp, mn1, mn2, e);
    DJob J;
    memset(&J, 0, sizeof(J));
    for (int d = 0; d < 3; d++) { J.dims[d] = d2[d]; J.s1[d] = (int)mn1[d]; J.s2[d] = (int)mn2[d]; J.e[d] = (int)e[d]; }
    J.ccc = (!empty && e[0] > 0 && e[1] > 0 && e[2] > 0) ? 1 : 0;
    MAD_HIP(hipMemcpyAsync(d_job, &J, sizeof(J), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_ccc_b, dim3(CCC_WGS, 1), dim3(256), 0, ctx->stream, d_job, (const float *)g1, d1[1], d1[2], (const float *)g2, -INFINITY, d_part);
    mad_timer_end(ctx, MAD_T_CCC);
    MAD_HIP(hipGetLastError());
    double hp[CCC_WGS * 3];
    MAD_HIP(hipMemcpyAsync(hp, d_part, sizeof(hp), hipMemcpyDeviceToHost, ctx->stream));
    MAD_HIP(hipMemcpyAsync(grid1, g1, n1 * 4, hipMemcpyDeviceToHost, ctx->stream));
    MAD_HIP(hipMemcpyAsync(grid2, g2, n2 * 4, hipMemcpyDeviceToHost, ctx->stream));
    MAD_HIP(hipStreamSynchronize(ctx->stream));
    if (empty) { *ccc = 0.0; return MAD_OK; }
    double h[3] = {0, 0, 0};
    for (int w = 0; w < CCC_WGS; w++)
        for (int q = 0; q < 3; q++) h[q] += hp[3 * w + q];
    *ccc = h[0] / sqrt(h[1] * h[2]);      // 0/0 -> NaN for an empty but non-inverted box, as the reference
    return MAD_OK;
}

// a14-a16 for a batch of placed copies of one structure: each candidate's atoms are turned into a simulated density
// (PDB.structure_to_density, PDB.py:131-208) and scored against the map uploaded with mad_upload_density
// (Dmap.get_CCC_with_grid, Dmap.py:153-258: both grids clamped at ccc_isovalue -- the map on the fly, so the uploaded copy stays
// as it was).  All candidates go through each kernel together; one read-back of the partial sums at the end.
static int density_ccc_device(mad_ctx *ctx, const double *d_atoms, const double *d_mass, int n_cand, int64_t n, const double *bbox /* host, n_cand x 6 */,
                              double resolution, double density_isovalue, double ccc_isovalue, double *ccc) {
    const double voxsp = ctx->dens.vs;
    std::vector<DensityPlan> plans(n_cand);
    std::vector<unsigned long long> atom0(n_cand), mass0(n_cand, 0), na(n_cand, (unsigned long long)n), off32(n_cand);
    size_t tot = 0;
    for (int c = 0; c < n_cand; c++) {
        MAD_TRY(density_plan_box(ctx, bbox + 6 * (size_t)c, bbox + 6 * (size_t)c + 3, resolution, voxsp, 0, &plans[c]));
        atom0[c] = (unsigned long long)c * n;
        off32[c] = tot;
        tot += (size_t)plans[c].dims[0] * plans[c].dims[1] * plans[c].dims[2];
    }
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_TMP_J), tot * 4));
    std::vector<double> sums((size_t)n_cand * 3);
    std::vector<char> empty;
    MAD_TRY(density_batch(ctx, plans, d_atoms, d_mass, atom0, mass0, na, voxsp, density_isovalue, scratch<float>(ctx, S_TMP_J), off32, ccc_isovalue,
                          sums.data(), &empty));
    for (int c = 0; c < n_cand; c++) ccc[c] = empty[c] ? 0.0 : sums[3 * (size_t)c] / sqrt(sums[3 * (size_t)c + 1] * sums[3 * (size_t)c + 2]);
    return MAD_OK;
}

extern "C" int mad_density_ccc(mad_ctx *ctx, const double *atoms, const double *mass, int n_cand, int64_t n, double resolution,
                               double density_isovalue, double ccc_isovalue, double *ccc) {
    if (ctx) mad_use_lane(ctx, 0);
    if (!ctx || !atoms || !mass || !ccc || n_cand < 0 || n <= 0) return ctx ? mad_fail(ctx, MAD_EINVAL, "mad_density_ccc: bad argument") : MAD_EINVAL;
    if (!ctx->dens.grid) return mad_fail(ctx, MAD_EINVAL, "mad_density_ccc: call mad_upload_density first");
    if (!(resolution > 0)) return mad_fail(ctx, MAD_EINVAL, "mad_density_ccc: resolution %g", resolution);
    if (n_cand == 0) return MAD_OK;
    if (n_cand > 4096) return mad_fail(ctx, MAD_EINVAL, "mad_density_ccc: %d candidates in one call", n_cand);
    std::vector<double> bbox((size_t)n_cand * 6);
    for (int c = 0; c < n_cand; c++) {
        double *b = &bbox[6 * (size_t)c];
        b[0] = b[1] = b[2] = INFINITY; b[3] = b[4] = b[5] = -INFINITY;
        const double *a = atoms + (size_t)c * n * 3;
        for (int64_t i = 0; i < n; i++)
            for (int d = 0; d < 3; d++) {
                const double v = a[3 * i + d];
                if (!(v == v)) return mad_fail(ctx, MAD_EDOM, "mad_structure_to_density: NaN coordinate");
                b[d] = v < b[d] ? v : b[d];
                b[3 + d] = v > b[3 + d] ? v : b[3 + d];
            }
    }
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_TMP_E), (size_t)n_cand * n * 24));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_TMP_F), (size_t)n * 8));
    MAD_HIP(hipMemcpyAsync(mad_sb(ctx, S_TMP_E).p, atoms, (size_t)n_cand * n * 24, hipMemcpyHostToDevice, ctx->stream));
    MAD_HIP(hipMemcpyAsync(mad_sb(ctx, S_TMP_F).p, mass, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    return density_ccc_device(ctx, scratch<double>(ctx, S_TMP_E), scratch<double>(ctx, S_TMP_F), n_cand, n, bbox.data(), resolution, density_isovalue,
                              ccc_isovalue, ccc);
}

// ---------------------------------------------------------------------------
// MaD._refine_filtered_solutions (MaD.py:556-629) for a batch of candidate poses of one structure, on the device from the poses to
// the scores: place (MaD.py:566-569: translate by -hi, rotate, translate by lo) -> refine_pdb (a13) -> structure_to_density (a14-a15)
// -> get_CCC_with_grid (a16).  What travels: n_atoms x 3 base coordinates + masses once, 15 doubles per candidate in, and per
// candidate three partial-sum rows, two flags and -- only if `coords` is given -- the refined coordinates out.
// ---------------------------------------------------------------------------

// x' = (x - hi) @ M + lo, the three steps in the reference's order and rounding (PDB.py:98-113: subtract, coords @ R, add)
// candidate c: cand[3 c ..] = {first atom of its structure in `base`, its atom count, first atom of its copy in `out`}
__global__ __launch_bounds__(256) void k_place(const double *__restrict__ base, const int64_t *__restrict__ cand,
                                               const double *__restrict__ pose /* n_cand x 15: hi, lo, M */, double *__restrict__ out) {
    const int c = blockIdx.y;
    const double *P = pose + 15 * (size_t)c;
    const int64_t b0 = cand[3 * c], n = cand[3 * c + 1], o0 = cand[3 * c + 2];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double a = base[3 * (b0 + i)] - P[0], b = base[3 * (b0 + i) + 1] - P[1], cc = base[3 * (b0 + i) + 2] - P[2];
    const double *M = P + 6;
    double *o = out + (size_t)(o0 + i) * 3;
    o[0] = ((a * M[0] + b * M[3]) + cc * M[6]) + P[3];
    o[1] = ((a * M[1] + b * M[4]) + cc * M[7]) + P[4];
    o[2] = ((a * M[2] + b * M[5]) + cc * M[8]) + P[5];
}

// bounding box of every candidate's atoms (NaN-propagating: a NaN coordinate makes the box NaN) -> box[c][6]
__global__ __launch_bounds__(1024) void k_bbox(const double *__restrict__ coords, const int64_t *__restrict__ cand, double *__restrict__ box) {
    __shared__ double wt[16][6];
    const int c = blockIdx.x;
    const int64_t n = cand[3 * c + 1];
    const double *a = coords + (size_t)cand[3 * c + 2] * 3;
    double v[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
    bool bad = false;
    for (int64_t i = threadIdx.x; i < n; i += 1024)
        for (int d = 0; d < 3; d++) {
            const double x = a[3 * i + d];
            bad |= !(x == x);
            v[d] = fmin(v[d], x); v[3 + d] = fmax(v[3 + d], x);
        }
    for (int d = 0; d < 3; d++) { v[d] = -wave_max_f64(-v[d]); v[3 + d] = wave_max_f64(v[3 + d]); }
    const bool any_bad = __any(bad);
    if (lane_id() == 0)
        for (int d = 0; d < 6; d++) wt[threadIdx.x >> 6][d] = any_bad ? NAN : v[d];
    __syncthreads();
    if (threadIdx.x < 6) {
        double r = wt[0][threadIdx.x];
        for (int w = 1; w < 16; w++) {
            const double x = wt[w][threadIdx.x];
            r = (r != r || x != x) ? NAN : (threadIdx.x < 3 ? fmin(r, x) : fmax(r, x));
        }
        box[6 * (size_t)c + threadIdx.x] = r;
    }
}

extern "C" int mad_dock_refine_score(mad_ctx *ctx, int n_struct, const double *base_atoms, const double *mass, const int64_t *first_atom,
                                     int n_cand, const int32_t *cand_struct, const double *hi_p, const double *lo_p, const double *rot,
                                     int n_steps, double max_step, double min_step, double resolution, double density_isovalue,
                                     double ccc_isovalue, double *coords, int32_t *converged, int32_t *last_step, double *ccc) {
    if (ctx) mad_use_lane(ctx, 0);
    if (!ctx) return MAD_EINVAL;
    if (!ctx->dens.grad) return mad_fail(ctx, MAD_EINVAL, "mad_dock_refine_score: call mad_upload_density first");
    if (n_cand <= 0) return MAD_OK;
    if (n_struct <= 0 || !base_atoms || !mass || !first_atom || !cand_struct || !hi_p || !lo_p || !rot || !converged || !last_step || !ccc ||
        n_steps < 0 || !(resolution > 0))
        return mad_fail(ctx, MAD_EINVAL, "mad_dock_refine_score: bad argument");
    if (n_cand > 4096) return mad_fail(ctx, MAD_EINVAL, "mad_dock_refine_score: %d candidates in one call", n_cand);
    const int64_t n_base = first_atom[n_struct];
    std::vector<int64_t> tab((size_t)n_cand * 3), cn(n_cand), co(n_cand);      // {base0, n, out0} per candidate; sizes; offsets
    int64_t total = 0, n_max = 0;
    for (int c = 0; c < n_cand; c++) {
        const int st = cand_struct[c];
        if (st < 0 || st >= n_struct) return mad_fail(ctx, MAD_EINVAL, "mad_dock_refine_score: candidate %d refers to structure %d of %d", c, st, n_struct);
        const int64_t n = first_atom[st + 1] - first_atom[st];
        if (n <= 0) return mad_fail(ctx, MAD_EINVAL, "mad_dock_refine_score: structure %d has no atoms", st);
        tab[3 * (size_t)c] = first_atom[st]; tab[3 * (size_t)c + 1] = n; tab[3 * (size_t)c + 2] = total;
        cn[c] = n; co[c] = total;
        total += n;
        n_max = std::max(n_max, n);
    }
    const size_t bytes = (size_t)total * 24;
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_TMP_E), bytes));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_TMP_A), (size_t)n_base * 24));       // base coordinates of all structures
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_TMP_B), (size_t)n_base * 8));        // their masses
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_TMP_C), (size_t)n_cand * (15 + 6 + 5) * 8 + 64));      // poses, boxes, the three tables
    std::vector<double> pose((size_t)n_cand * 15);
    for (int c = 0; c < n_cand; c++) {
        for (int d = 0; d < 3; d++) { pose[15 * (size_t)c + d] = hi_p[3 * c + d]; pose[15 * (size_t)c + 3 + d] = lo_p[3 * c + d]; }
        for (int d = 0; d < 9; d++) pose[15 * (size_t)c + 6 + d] = rot[9 * c + d];
    }
    double *d_base = scratch<double>(ctx, S_TMP_A), *d_mass = scratch<double>(ctx, S_TMP_B), *d_pose = scratch<double>(ctx, S_TMP_C);
    double *d_box = d_pose + (size_t)n_cand * 15;
    int64_t *d_tab = (int64_t *)(d_box + (size_t)n_cand * 6), *d_cn = d_tab + (size_t)n_cand * 3, *d_co = d_cn + n_cand;
    double *d_coords = scratch<double>(ctx, S_TMP_E);
    MAD_HIP(hipMemcpyAsync(d_base, base_atoms, (size_t)n_base * 24, hipMemcpyHostToDevice, ctx->stream));
    MAD_HIP(hipMemcpyAsync(d_mass, mass, (size_t)n_base * 8, hipMemcpyHostToDevice, ctx->stream));
    MAD_HIP(hipMemcpyAsync(d_pose, pose.data(), pose.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    MAD_HIP(hipMemcpyAsync(d_tab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    MAD_HIP(hipMemcpyAsync(d_cn, cn.data(), cn.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    MAD_HIP(hipMemcpyAsync(d_co, co.data(), co.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_place, dim3((unsigned)mad_ceil_div(n_max, 256), n_cand), dim3(256), 0, ctx->stream, d_base, d_tab, d_pose, d_coords);
    int32_t *d_conv = nullptr, *d_last = nullptr;
    MAD_TRY(refine_device(ctx, n_cand, n_max, n_steps, max_step, min_step, &d_conv, &d_last, d_cn, d_co, total));
    hipLaunchKernelGGL(k_bbox, dim3(n_cand), dim3(1024), 0, ctx->stream, d_coords, d_tab, d_box);
    MAD_HIP(hipGetLastError());
    std::vector<double> box((size_t)n_cand * 6);
    MAD_HIP(hipMemcpyAsync(box.data(), d_box, box.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    MAD_HIP(hipMemcpyAsync(converged, d_conv, (size_t)n_cand * 4, hipMemcpyDeviceToHost, ctx->stream));
    MAD_HIP(hipMemcpyAsync(last_step, d_last, (size_t)n_cand * 4, hipMemcpyDeviceToHost, ctx->stream));
    MAD_HIP(hipStreamSynchronize(ctx->stream));      // the one read-back before the density stage: 48 bytes per candidate
    // a candidate whose refinement ended in NaN coordinates (structure_utils.py:97-98) has no density: scored NaN, the others go on
    std::vector<int> ok;
    for (int c = 0; c < n_cand; c++) {
        bool fin = true;
        for (int d = 0; d < 6; d++) fin = fin && (box[6 * (size_t)c + d] == box[6 * (size_t)c + d]) && fabs(box[6 * (size_t)c + d]) < 1e12;
        if (fin) ok.push_back(c);
        else ccc[c] = NAN;
    }
    if (!ok.empty()) {
        const double voxsp = ctx->dens.vs;
        const int m = (int)ok.size();
        std::vector<DensityPlan> plans(m);
        std::vector<unsigned long long> atom0(m), mass0(m), na(m), off32(m);
        size_t tot = 0;
        for (int j = 0; j < m; j++) {
            const int c = ok[j];
            MAD_TRY(density_plan_box(ctx, &box[6 * (size_t)c], &box[6 * (size_t)c + 3], resolution, voxsp, 0, &plans[j]));
            atom0[j] = (unsigned long long)co[c];
            mass0[j] = (unsigned long long)first_atom[cand_struct[c]];
            na[j] = (unsigned long long)cn[c];
            off32[j] = tot;
            tot += (size_t)plans[j].dims[0] * plans[j].dims[1] * plans[j].dims[2];
        }
        MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_TMP_J), tot * 4));
        std::vector<double> sums((size_t)m * 3);
        std::vector<char> empty;
        MAD_TRY(density_batch(ctx, plans, d_coords, d_mass, atom0, mass0, na, voxsp, density_isovalue, scratch<float>(ctx, S_TMP_J), off32,
                              ccc_isovalue, sums.data(), &empty));
        for (int j = 0; j < m; j++) ccc[ok[j]] = empty[j] ? 0.0 : sums[3 * (size_t)j] / sqrt(sums[3 * (size_t)j + 1] * sums[3 * (size_t)j + 2]);
    }
    if (coords) {
        MAD_HIP(hipMemcpyAsync(coords, d_coords, bytes, hipMemcpyDeviceToHost, ctx->stream));
        MAD_HIP(hipStreamSynchronize(ctx->stream));
    }
    return MAD_OK;
}

// ---- next to the path, downstream: pairwise occupancy overlap of placed structures ---------------------------------
// structure_utils.get_overlap (structure_utils.py:163-259) as MaD._build_from_single / _build_models use it
// (MaD.py:673-686, 775-783): overlap(i, j) = #{voxels of the common box where both grids are > 0} / #{grid_i > 0}.

struct OverlapPair {      // one (i, j) of the table: where the two grids start in the float pool, their strides, the common box
    unsigned long long off1, off2;
    int a1, a2, b1, b2;      // y/z extents of grid i and grid j
    int s1[3], s2[3], e[3];
};

// positives of each grid after the clamp at `iso` (structure_utils.py:171-172, 255); grid g = blockIdx.y
__global__ __launch_bounds__(256) void k_clamp_count(float *__restrict__ pool, const unsigned long long *__restrict__ first, float iso,
                                                     unsigned long long *__restrict__ npos) {
    const size_t b = first[blockIdx.y], n = first[blockIdx.y + 1] - b;
    float *g = pool + b;
    int c = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float v = g[i];
        if (v < iso) { v = 0.f; g[i] = v; }
        c += v > 0.f ? 1 : 0;
    }
    c = wave_sum_i32(c);
    if (lane_id() == 0 && c) atomicAdd(&npos[blockIdx.y], (unsigned long long)c);
}

// co-occupied voxels of the common box (structure_utils.py:246-254); pair p = blockIdx.y
__global__ __launch_bounds__(256) void k_overlap_pairs(const float *__restrict__ pool, const OverlapPair *__restrict__ pairs,
                                                       unsigned long long *__restrict__ common) {
    const OverlapPair P = pairs[blockIdx.y];
    const float *g1 = pool + P.off1, *g2 = pool + P.off2;
    const size_t n = (size_t)P.e[0] * P.e[1] * P.e[2];
    int c = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int z = (int)(i % P.e[2]), y = (int)((i / P.e[2]) % P.e[1]), x = (int)(i / ((size_t)P.e[2] * P.e[1]));
        const float a = g1[((size_t)(P.s1[0] + x) * P.a1 + (P.s1[1] + y)) * P.a2 + (P.s1[2] + z)];
        const float b = g2[((size_t)(P.s2[0] + x) * P.b1 + (P.s2[1] + y)) * P.b2 + (P.s2[2] + z)];
        c += (a > 0.f && b > 0.f) ? 1 : 0;
    }
    c = wave_sum_i32(c);
    if (lane_id() == 0 && c) atomicAdd(&common[blockIdx.y], (unsigned long long)c);
}

// shared tail of the two entry points: grids already in `pool` (float32, unclamped), geometry on the host
static int overlap_run(mad_ctx *ctx, float *pool, const std::vector<unsigned long long> &first, const std::vector<int32_t> &dims,
                       const std::vector<double> &origin, double voxsp, double isovalue, const std::vector<int> &pi,
                       const std::vector<int> &pj, std::vector<unsigned long long> &npos, std::vector<unsigned long long> &common) {
    const int n_grid = (int)first.size() - 1;
    const size_t n_pair = pi.size();
    std::vector<OverlapPair> tab;
    std::vector<long> where(n_pair, -1);
    for (size_t p = 0; p < n_pair; p++) {
        const int i = pi[p], j = pj[p];
        long mn1[3], mn2[3], e[3];
        if (!ccc_overlap(&dims[3 * i], &origin[3 * i], &dims[3 * j], &origin[3 * j], voxsp, mn1, mn2, e)) continue;      // :241-243
        if (e[0] <= 0 || e[1] <= 0 || e[2] <= 0) continue;
        OverlapPair P;
        P.off1 = first[i]; P.off2 = first[j];
        P.a1 = dims[3 * i + 1]; P.a2 = dims[3 * i + 2]; P.b1 = dims[3 * j + 1]; P.b2 = dims[3 * j + 2];
        for (int d = 0; d < 3; d++) { P.s1[d] = (int)mn1[d]; P.s2[d] = (int)mn2[d]; P.e[d] = (int)e[d]; }
        where[p] = (long)tab.size();
        tab.push_back(P);
    }
    const size_t nt = tab.size();
    const size_t bytes_first = (size_t)(n_grid + 1) * 8, bytes_cnt = (size_t)(n_grid + nt) * 8, bytes_tab = nt * sizeof(OverlapPair);
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_TMP_G), bytes_first + bytes_cnt + bytes_tab + 64));
    char *blk = scratch<char>(ctx, S_TMP_G);
    unsigned long long *d_first = (unsigned long long *)blk, *d_cnt = (unsigned long long *)(blk + bytes_first);
    OverlapPair *d_tab = (OverlapPair *)(blk + bytes_first + bytes_cnt);
    MAD_HIP(hipMemcpyAsync(d_first, first.data(), bytes_first, hipMemcpyHostToDevice, ctx->stream));
    if (nt) MAD_HIP(hipMemcpyAsync(d_tab, tab.data(), bytes_tab, hipMemcpyHostToDevice, ctx->stream));
    MAD_HIP(hipMemsetAsync(d_cnt, 0, bytes_cnt, ctx->stream));
    hipLaunchKernelGGL(k_clamp_count, dim3(16, n_grid), dim3(256), 0, ctx->stream, pool, d_first, (float)isovalue, d_cnt);
    for (size_t p0 = 0; p0 < nt; p0 += 32768) {      // gridDim.y limit
        const unsigned np = (unsigned)std::min<size_t>(nt - p0, 32768);
        hipLaunchKernelGGL(k_overlap_pairs, dim3(8, np), dim3(256), 0, ctx->stream, pool, d_tab + p0, d_cnt + n_grid + p0);
    }
    MAD_HIP(hipGetLastError());
    std::vector<unsigned long long> h(n_grid + nt);
    MAD_HIP(hipMemcpyAsync(h.data(), d_cnt, bytes_cnt, hipMemcpyDeviceToHost, ctx->stream));
    MAD_HIP(hipStreamSynchronize(ctx->stream));
    npos.assign(h.begin(), h.begin() + n_grid);
    common.assign(n_pair, 0);
    for (size_t p = 0; p < n_pair; p++)
        if (where[p] >= 0) common[p] = h[n_grid + where[p]];
    return MAD_OK;
}

extern "C" int mad_grid_overlap(mad_ctx *ctx, float *grid1, const int32_t d1[3], const double o1[3], float *grid2, const int32_t d2[3],
                                const double o2[3], double voxsp, double isovalue, int64_t *common, int64_t *n_pos1) {
    if (ctx) mad_use_lane(ctx, 0);
    if (!ctx || !grid1 || !grid2 || !d1 || !d2 || !o1 || !o2 || !common || !n_pos1)
        return ctx ? mad_fail(ctx, MAD_EINVAL, "mad_grid_overlap: NULL argument") : MAD_EINVAL;
    if (!(voxsp > 0)) return mad_fail(ctx, MAD_EINVAL, "mad_grid_overlap: voxsp %g", voxsp);
    for (int d = 0; d < 3; d++)
        if (d1[d] <= 0 || d2[d] <= 0) return mad_fail(ctx, MAD_EINVAL, "mad_grid_overlap: empty grid");
    const size_t n1 = (size_t)d1[0] * d1[1] * d1[2], n2 = (size_t)d2[0] * d2[1] * d2[2];
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_TMP_H), (n1 + n2) * 4));
    float *pool = scratch<float>(ctx, S_TMP_H);
    MAD_HIP(hipMemcpyAsync(pool, grid1, n1 * 4, hipMemcpyHostToDevice, ctx->stream));
    MAD_HIP(hipMemcpyAsync(pool + n1, grid2, n2 * 4, hipMemcpyHostToDevice, ctx->stream));
    std::vector<unsigned long long> first = {0, n1, n1 + n2}, npos, cm;
    std::vector<int32_t> dims = {d1[0], d1[1], d1[2], d2[0], d2[1], d2[2]};
    std::vector<double> org = {o1[0], o1[1], o1[2], o2[0], o2[1], o2[2]};
    MAD_TRY(overlap_run(ctx, pool, first, dims, org, voxsp, isovalue, {0}, {1}, npos, cm));
    // both grids come back clamped, as the reference leaves them (structure_utils.py:171-172)
    MAD_HIP(hipMemcpyAsync(grid1, pool, n1 * 4, hipMemcpyDeviceToHost, ctx->stream));
    MAD_HIP(hipMemcpyAsync(grid2, pool + n1, n2 * 4, hipMemcpyDeviceToHost, ctx->stream));
    MAD_HIP(hipStreamSynchronize(ctx->stream));
    *common = (int64_t)cm[0];
    *n_pos1 = (int64_t)npos[0];
    return MAD_OK;
}

extern "C" int mad_overlap_matrix(mad_ctx *ctx, const double *atoms, const double *mass, const int64_t *first_atom, int n_struct,
                                  double resolution, double voxsp, double density_isovalue, double overlap_isovalue, double *overlap) {
    if (ctx) mad_use_lane(ctx, 0);
    if (!ctx || !atoms || !mass || !first_atom || !overlap || n_struct < 0)
        return ctx ? mad_fail(ctx, MAD_EINVAL, "mad_overlap_matrix: bad argument") : MAD_EINVAL;
    if (!(voxsp > 0) || !(resolution > 0)) return mad_fail(ctx, MAD_EINVAL, "mad_overlap_matrix: resolution %g voxsp %g", resolution, voxsp);
    if (n_struct > 4096) return mad_fail(ctx, MAD_EINVAL, "mad_overlap_matrix: %d structures in one call", n_struct);
    for (size_t i = 0; i < (size_t)n_struct * n_struct; i++) overlap[i] = 0.0;
    if (n_struct < 2) return MAD_OK;
    std::vector<DensityPlan> plans(n_struct);
    std::vector<unsigned long long> first(n_struct + 1, 0);
    std::vector<int32_t> dims(3 * (size_t)n_struct);
    std::vector<double> org(3 * (size_t)n_struct);
    size_t no_max = 0, n_max = 0;
    for (int s = 0; s < n_struct; s++) {
        const int64_t n = first_atom[s + 1] - first_atom[s];
        if (n <= 0) return mad_fail(ctx, MAD_EINVAL, "mad_overlap_matrix: structure %d has no atoms", s);
        MAD_TRY(density_plan(ctx, atoms + 3 * first_atom[s], n, resolution, voxsp, 0, &plans[s]));
        const size_t no = (size_t)plans[s].dims[0] * plans[s].dims[1] * plans[s].dims[2];
        first[s + 1] = first[s] + no;
        no_max = std::max(no_max, no);
        n_max = std::max(n_max, (size_t)n);
        for (int d = 0; d < 3; d++) { dims[3 * s + d] = plans[s].dims[d]; org[3 * s + d] = plans[s].origin[d]; }
    }
    const int64_t n_all = first_atom[n_struct];
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_TMP_E), (size_t)n_all * 24));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_TMP_F), (size_t)n_all * 8));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_TMP_J), (size_t)first[n_struct] * 4));
    double *d_atoms = scratch<double>(ctx, S_TMP_E), *d_mass = scratch<double>(ctx, S_TMP_F);
    float *pool = scratch<float>(ctx, S_TMP_J);
    MAD_HIP(hipMemcpyAsync(d_atoms, atoms, (size_t)n_all * 24, hipMemcpyHostToDevice, ctx->stream));
    MAD_HIP(hipMemcpyAsync(d_mass, mass, (size_t)n_all * 8, hipMemcpyHostToDevice, ctx->stream));
    {   // all structures through each density kernel together (density_batch)
        std::vector<unsigned long long> a0(n_struct), na(n_struct);
        for (int s_ = 0; s_ < n_struct; s_++) { a0[s_] = (unsigned long long)first_atom[s_]; na[s_] = (unsigned long long)(first_atom[s_ + 1] - first_atom[s_]); }
        std::vector<unsigned long long> off32(first.begin(), first.begin() + n_struct);
        MAD_TRY(density_batch(ctx, plans, d_atoms, d_mass, a0, a0, na, voxsp, density_isovalue, pool, off32, 0.0, nullptr, nullptr));
    }
    (void)no_max; (void)n_max;
    std::vector<int> pi, pj;
    for (int i = 0; i < n_struct; i++)
        for (int j = i + 1; j < n_struct; j++) { pi.push_back(i); pj.push_back(j); }
    std::vector<unsigned long long> npos, cm;
    MAD_TRY(overlap_run(ctx, pool, first, dims, org, voxsp, overlap_isovalue, pi, pj, npos, cm));
    for (size_t p = 0; p < pi.size(); p++)
        overlap[(size_t)pi[p] * n_struct + pj[p]] = npos[pi[p]] ? (double)cm[p] / (double)npos[pi[p]] : 0.0;      // :256-258
    return MAD_OK;
}
