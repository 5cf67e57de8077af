// mad_space.hip -- scale-space preparation and peak search on the device.
//
// Replaces the numerical body of MapSpace.build_space (mad/MapSpace.py:116-189): zero padding, the 2x
// upsampling by successive 1-D not-a-knot cubic splines (scipy interp1d(kind="cubic"), :206-214), the light
// pre-smoothing (:144), the scale-normalised LoG volumes (:169-173), the Gaussian-smoothed volumes and
// their gradients (:178-189) -- and the dense part of Detector.find_anchors (mad/Detector.py:28-29): the
// 3x3x3 local-maximum mask.  The gradients are written straight into a field slot as the 16-byte texels the
// orientation / descriptor kernels sample, so a structure never leaves the device between its density grid
// and its descriptors.
//
// Arithmetic contract (what makes the volumes reproduce scipy's):
//  * every 1-D filter pass converts its input line to double, accumulates in double in the order of
//    scipy.ndimage's correlate1d for symmetric kernels -- centre tap first, then (left + right) * w from the
//    outermost tap inwards -- and rounds ONCE to the storage type of the volume (float32 or float64), pass
//    by pass in axis order 0, 1, 2, exactly like gaussian_filter does with its intermediate arrays;
//  * the boundary is scipy's "reflect" (d c b a | a b c d | d c b a);
//  * the kernel weights are inputs (computed by numpy on the host exactly as scipy computes them);
//  * the Laplacian is the storage-type sum of the three second-derivative filters in axis order
//    (generic_laplace), then * -1, * sigma^2, negatives clamped to 0;
//  * np.gradient: central differences / 2 inside, one-sided at the faces, in the storage type;
//  * the spline pass solves the banded collocation system per line (LU factors from the host, no pivoting)
//    and evaluates 4 basis weights per output sample, in double.  This is NOT bit-identical to LAPACK's
//    pivoted banded solve + de Boor, it agrees to ~4e-16 relative; after the float32 rounding of the
//    upsampled volume the difference survives in roughly one voxel per 1e7 as a single float32 ulp.
#include <algorithm>
#include <cstring>
#include <vector>

#include "mad_common.h"

namespace {

struct Dims {
    int n[3];
    __host__ __device__ size_t count() const { return (size_t)n[0] * n[1] * n[2]; }
};

// scipy.ndimage "reflect" (half-sample symmetric) extension of index i into [0, n)
__device__ __forceinline__ int reflect_idx(int i, int n) {
    if (n == 1) return 0;
    const int p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - 1 - i;
}

template <typename T>
__global__ __launch_bounds__(256) void k_pad(const T *__restrict__ src, Dims s, int pad, T *__restrict__ dst, Dims d) {
    const size_t total = d.count();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int z = (int)(i % d.n[2]), y = (int)((i / d.n[2]) % d.n[1]), x = (int)(i / ((size_t)d.n[2] * d.n[1]));
        const int sx = x - pad, sy = y - pad, sz = z - pad;
        T v = (T)0;
        if (sx >= 0 && sx < s.n[0] && sy >= 0 && sy < s.n[1] && sz >= 0 && sz < s.n[2]) v = src[((size_t)sx * s.n[1] + sy) * s.n[2] + sz];
        dst[i] = v;
    }
}

// start offset and element stride of line L along `axis` of a volume with dims d (z fastest)
__device__ __forceinline__ void line_of(size_t L, Dims d, int axis, size_t &off, size_t &stride) {
    if (axis == 0) { off = L; stride = (size_t)d.n[1] * d.n[2]; }
    else if (axis == 1) { off = (L / d.n[2]) * (size_t)d.n[1] * d.n[2] + (L % d.n[2]); stride = (size_t)d.n[2]; }
    else { off = L * (size_t)d.n[2]; stride = 1; }
}

// B-spline coefficients of every line along `axis`: forward / backward substitution with the banded LU
// factors lu = [l2 | l1 | d | u1 | u2] (5 x n) of the not-a-knot collocation matrix.  One thread per line.
template <typename T>
__global__ __launch_bounds__(256) void k_spline_solve(const T *__restrict__ in, Dims d, int axis, const double *__restrict__ lu,
                                                      double *__restrict__ c) {
    const int n = d.n[axis];
    const size_t n_lines = d.count() / n;
    const size_t L = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (L >= n_lines) return;
    size_t off, st;
    line_of(L, d, axis, off, st);
    const double *l2 = lu, *l1 = lu + n, *dg = lu + 2 * (size_t)n, *u1 = lu + 3 * (size_t)n, *u2 = lu + 4 * (size_t)n;
    double z1 = 0.0, z2 = 0.0;      // z[i-1], z[i-2]
    for (int i = 0; i < n; i++) {
        const double z = ((double)in[off + i * st] - l1[i] * z1) - l2[i] * z2;
        c[off + i * st] = z;
        z2 = z1; z1 = z;
    }
    double c1 = 0.0, c2 = 0.0;      // c[i+1], c[i+2]
    for (int i = n - 1; i >= 0; i--) {
        const double v = ((c[off + i * st] - u1[i] * c1) - u2[i] * c2) / dg[i];
        c[off + i * st] = v;
        c2 = c1; c1 = v;
    }
}

// value of the spline at the 2n-1 half-integer sites of every line: 4 basis weights per site
__global__ __launch_bounds__(256) void k_spline_eval(const double *__restrict__ c, Dims d, int axis, const double *__restrict__ ev_w,
                                                     const int32_t *__restrict__ ev_i, double *__restrict__ out, Dims o) {
    const size_t total = o.count();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int q[3];
        q[2] = (int)(i % o.n[2]); q[1] = (int)((i / o.n[2]) % o.n[1]); q[0] = (int)(i / ((size_t)o.n[2] * o.n[1]));
        const int m = q[axis];
        const int i0 = ev_i[m];
        q[axis] = i0;
        const size_t base = ((size_t)q[0] * d.n[1] + q[1]) * d.n[2] + q[2];
        const size_t st = axis == 0 ? (size_t)d.n[1] * d.n[2] : (axis == 1 ? (size_t)d.n[2] : 1);
        const double *w = ev_w + 4 * (size_t)m;
        out[i] = ((w[0] * c[base] + w[1] * c[base + st]) + w[2] * c[base + 2 * st]) + w[3] * c[base + 3 * st];
    }
}

// One pass of scipy.ndimage.correlate1d with a symmetric kernel of radius R along `axis`, "reflect" boundary;
// up to two kernels (w0, w1) applied to the same input.  One thread per voxel.
template <typename Tin, typename Tout>
__global__ __launch_bounds__(256) void k_filter_axis(const Tin *__restrict__ in, Dims d, int axis, int R, const double *__restrict__ w0,
                                                     const double *__restrict__ w1, Tout *__restrict__ out0, Tout *__restrict__ out1) {
    const size_t total = d.count();
    const size_t st = axis == 0 ? (size_t)d.n[1] * d.n[2] : (axis == 1 ? (size_t)d.n[2] : 1);
    const int n = d.n[axis];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int pos = (int)((i / st) % n);
        const size_t base = i - (size_t)pos * st;
        const double centre = (double)in[i];
        double a0 = centre * w0[R], a1 = w1 ? centre * w1[R] : 0.0;
        if (pos >= R && pos + R < n) {
            for (int jj = -R; jj < 0; jj++) {
                const double pr = (double)in[i + (ptrdiff_t)jj * (ptrdiff_t)st] + (double)in[i - (ptrdiff_t)jj * (ptrdiff_t)st];
                a0 += pr * w0[R + jj];
                if (w1) a1 += pr * w1[R + jj];
            }
        } else {
            for (int jj = -R; jj < 0; jj++) {
                const double pr = (double)in[base + (size_t)reflect_idx(pos + jj, n) * st] + (double)in[base + (size_t)reflect_idx(pos - jj, n) * st];
                a0 += pr * w0[R + jj];
                if (w1) a1 += pr * w1[R + jj];
            }
        }
        out0[i] = (Tout)a0;
        if (w1) out1[i] = (Tout)a1;
    }
}

// Last (axis 2) pass of the three second-derivative filters and of the Gaussian, fused with the Laplacian sum:
//   term0 = G0_z(s01), term1 = G0_z(u01), term2 = G2_z(t01), gauss = G0_z(t01)      (each rounded to T)
//   log   = clamp0( (-1 * ((term0 + term1) + term2)) * sig2 )                          (T arithmetic, MapSpace.py:171-172)
template <typename T>
__global__ __launch_bounds__(256) void k_log_gauss(const T *__restrict__ s01, const T *__restrict__ u01, const T *__restrict__ t01, Dims d,
                                                   int R, const double *__restrict__ g0, const double *__restrict__ g2, double sig2,
                                                   T *__restrict__ gauss, T *__restrict__ logv) {
    const size_t total = d.count();
    const int n = d.n[2];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int pos = (int)(i % n);
        const size_t base = i - pos;
        const double cs = (double)s01[i], cu = (double)u01[i], ct = (double)t01[i];
        double a0 = cs * g0[R], a1 = cu * g0[R], a2 = ct * g2[R], ag = ct * g0[R];
        for (int jj = -R; jj < 0; jj++) {
            const size_t il = base + reflect_idx(pos + jj, n), ir = base + reflect_idx(pos - jj, n);
            const double ps = (double)s01[il] + (double)s01[ir], pu = (double)u01[il] + (double)u01[ir], pt = (double)t01[il] + (double)t01[ir];
            a0 += ps * g0[R + jj];
            a1 += pu * g0[R + jj];
            a2 += pt * g2[R + jj];
            ag += pt * g0[R + jj];
        }
        gauss[i] = (T)ag;
        T l = (T)a0;
        l = l + (T)a1;
        l = l + (T)a2;
        l = (-l) * (T)sig2;
        logv[i] = (l < (T)0) ? (T)0 : l;
    }
}

// np.gradient of the smoothed volume (unit spacing, edge_order 1) -> texel {gx, gy, gz, |g|} in float32
template <typename T>
__global__ __launch_bounds__(256) void k_grad_tex(const T *__restrict__ g, Dims d, float4 *__restrict__ tex, unsigned *__restrict__ tex4) {
    const size_t total = d.count();
    const size_t sx = (size_t)d.n[1] * d.n[2], sy = (size_t)d.n[2];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int z = (int)(i % d.n[2]), y = (int)((i / d.n[2]) % d.n[1]), x = (int)(i / sx);
        T gx, gy, gz;
        if (x == 0) gx = g[i + sx] - g[i]; else if (x == d.n[0] - 1) gx = g[i] - g[i - sx]; else gx = (g[i + sx] - g[i - sx]) / (T)2;
        if (y == 0) gy = g[i + sy] - g[i]; else if (y == d.n[1] - 1) gy = g[i] - g[i - sy]; else gy = (g[i + sy] - g[i - sy]) / (T)2;
        if (z == 0) gz = g[i + 1] - g[i]; else if (z == d.n[2] - 1) gz = g[i] - g[i - 1]; else gz = (g[i + 1] - g[i - 1]) / (T)2;
        const float fx = (float)gx, fy = (float)gy, fz = (float)gz;
        const float s = __fadd_rn(__fadd_rn(__fmul_rn(fx, fx), __fmul_rn(fy, fy)), __fmul_rn(fz, fz));
        const float w = sqrtf(s);      // correctly rounded (see k_pack_field)
        tex[i] = make_float4(fx, fy, fz, w);
        tex4[i] = mad_tex4_encode(fx, fy, fz, w);
    }
}

template <typename Tin, typename Tout>
__global__ __launch_bounds__(256) void k_convert(const Tin *__restrict__ in, Tout *__restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = (Tout)in[i];
}

// 3x3x3 local maxima (value == maximum of the zero-extended neighbourhood), strictly above the threshold,
// at least `border` voxels from every face (skimage.feature.peak_local_max, min_distance 1)
template <typename T>
__global__ __launch_bounds__(256) void k_peaks(const T *__restrict__ v, Dims d, double thr, int border, int64_t *__restrict__ out_idx,
                                               double *__restrict__ out_val, int32_t *__restrict__ count, int cap) {
    const size_t total = d.count();
    const size_t sx = (size_t)d.n[1] * d.n[2], sy = (size_t)d.n[2];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int z = (int)(i % d.n[2]), y = (int)((i / d.n[2]) % d.n[1]), x = (int)(i / sx);
        if (x < border || y < border || z < border || x >= d.n[0] - border || y >= d.n[1] - border || z >= d.n[2] - border) continue;
        const T c = v[i];
        if (!((double)c > thr)) continue;
        bool is_max = !(c < (T)0);      // the zero extension takes part in the maximum
        for (int dx = -1; dx <= 1 && is_max; dx++) {
            for (int dy = -1; dy <= 1 && is_max; dy++) {
                for (int dz = -1; dz <= 1; dz++) {
                    const int xx = x + dx, yy = y + dy, zz = z + dz;
                    T o = (T)0;
                    if (xx >= 0 && xx < d.n[0] && yy >= 0 && yy < d.n[1] && zz >= 0 && zz < d.n[2]) o = v[(size_t)xx * sx + (size_t)yy * sy + zz];
                    if (o > c) { is_max = false; break; }
                }
            }
        }
        if (!is_max) continue;
        const int slot = atomicAdd(count, 1);
        if (slot < cap) { out_idx[slot] = (int64_t)i; out_val[slot] = (double)c; }
    }
}

// (2r+1)^3 neighbourhoods of n voxels (zero outside the volume), for the host's sub-voxel fit
template <typename T>
__global__ __launch_bounds__(256) void k_patches(const T *__restrict__ v, Dims d, const int32_t *__restrict__ coords, int n, int r,
                                                 T *__restrict__ out) {
    const int side = 2 * r + 1;
    const size_t per = (size_t)side * side * side, total = per * n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int p = (int)(i / per);
        const int q = (int)(i % per);
        const int x = coords[3 * p] + q / (side * side) - r, y = coords[3 * p + 1] + (q / side) % side - r, z = coords[3 * p + 2] + q % side - r;
        T o = (T)0;
        if (x >= 0 && x < d.n[0] && y >= 0 && y < d.n[1] && z >= 0 && z < d.n[2]) o = v[((size_t)x * d.n[1] + y) * d.n[2] + z];
        out[i] = o;
    }
}

struct Octave {
    Dims d;
    bool f64 = false;
    int kind = 1;          // 0 = upsampled, 1 = base (DensityFeature.oct_scale)
    void *grid = nullptr, *gauss = nullptr, *logv = nullptr;
};

// temporaries of one build, freed together
struct Arena {
    std::vector<void *> ptrs;
    void *get(size_t bytes) {
        void *p = nullptr;
        if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) return nullptr;
        ptrs.push_back(p);
        return p;
    }
    ~Arena() { for (void *p : ptrs) (void)hipFree(p); }
};

unsigned blocks_for(const mad_ctx *ctx, size_t n) { return (unsigned)std::min<size_t>((n + 255) / 256, (size_t)ctx->n_cu * 32); }

}  // namespace

struct mad_space {
    int n_oct = 0;
    Octave oct[2];
};

static void space_release(mad_space *s) {
    for (int o = 0; o < 2; o++) {
        for (void **p : {&s->oct[o].grid, &s->oct[o].gauss, &s->oct[o].logv}) {
            if (*p) (void)hipFree(*p);
            *p = nullptr;
        }
    }
    s->n_oct = 0;
}

extern "C" int mad_space_create(mad_ctx *ctx, mad_space **out) {
    if (!ctx || !out) return MAD_EINVAL;
    *out = new mad_space();
    return MAD_OK;
}

extern "C" void mad_space_destroy(mad_ctx *ctx, mad_space *s) {
    if (!s) return;
    if (ctx) (void)hipStreamSynchronize(ctx->stream);
    space_release(s);
    delete s;
}

template <typename T>
static int build_octave(mad_ctx *ctx, Octave &O, int R, const double *d_g0, const double *d_g2, double sig2, int slot) {
    Arena A;
    const size_t n = O.d.count(), bytes = n * sizeof(T);
    const unsigned nb = blocks_for(ctx, n);
    T *grid = (T *)O.grid;
    T *t0 = (T *)A.get(bytes), *s0 = (T *)A.get(bytes), *t01 = (T *)A.get(bytes), *s01 = (T *)A.get(bytes), *u01 = (T *)A.get(bytes);
    if (!t0 || !s0 || !t01 || !s01 || !u01) return mad_fail(ctx, MAD_ENOMEM, "mad_space_build: %zu bytes of filter temporaries", 5 * bytes);
    if (hipMalloc(&O.gauss, bytes) != hipSuccess || hipMalloc(&O.logv, bytes) != hipSuccess)
        return mad_fail(ctx, MAD_ENOMEM, "mad_space_build: %zu bytes of volumes", 2 * bytes);
    // axis 0: order 0 and order 2 of the same input
    hipLaunchKernelGGL((k_filter_axis<T, T>), dim3(nb), dim3(256), 0, ctx->stream, grid, O.d, 0, R, d_g0, d_g2, t0, s0);
    // axis 1: G0(t0) -> t01 and G2(t0) -> u01 in one pass, G0(s0) -> s01
    hipLaunchKernelGGL((k_filter_axis<T, T>), dim3(nb), dim3(256), 0, ctx->stream, t0, O.d, 1, R, d_g0, d_g2, t01, u01);
    hipLaunchKernelGGL((k_filter_axis<T, T>), dim3(nb), dim3(256), 0, ctx->stream, s0, O.d, 1, R, d_g0, (const double *)nullptr, s01, (T *)nullptr);
    // axis 2 + Laplacian
    hipLaunchKernelGGL((k_log_gauss<T>), dim3(nb), dim3(256), 0, ctx->stream, s01, u01, t01, O.d, R, d_g0, d_g2, sig2, (T *)O.gauss, (T *)O.logv);
    MAD_HIP(hipGetLastError());
    if (slot >= 0) {
        size_t cnt = 0;
        MAD_TRY(mad_field_alloc(ctx, slot, O.d.n[0], O.d.n[1], O.d.n[2], &cnt));
        hipLaunchKernelGGL((k_grad_tex<T>), dim3(nb), dim3(256), 0, ctx->stream, (const T *)O.gauss, O.d, (float4 *)ctx->field_mem[slot],
                           (unsigned *)((float4 *)ctx->field_mem[slot] + cnt));
        MAD_HIP(hipGetLastError());
    }
    MAD_HIP(hipStreamSynchronize(ctx->stream));      // the arena goes away
    return MAD_OK;
}

template <typename T>
static int build_all(mad_ctx *ctx, mad_space *s, const T *h_grid, int nx, int ny, int nz, int pad, int oct_mode, const double *g0,
                     const double *g2, int R, double sig2, const double *pre, int pre_R, const double *const *lu,
                     const double *const *ev_w, const int32_t *const *ev_i, int slot_up, int slot_base) {
    Arena A;
    const Dims src{{nx, ny, nz}};
    const Dims b{{nx + 2 * pad, ny + 2 * pad, nz + 2 * pad}};
    // weights and tables
    double *d_g0 = (double *)A.get((2 * R + 1) * 8), *d_g2 = (double *)A.get((2 * R + 1) * 8), *d_pre = (double *)A.get((2 * pre_R + 1) * 8);
    T *d_src = (T *)A.get(src.count() * sizeof(T));
    if (!d_g0 || !d_g2 || !d_pre || !d_src) return mad_fail(ctx, MAD_ENOMEM, "mad_space_build: staging");
    MAD_HIP(hipMemcpyAsync(d_g0, g0, (2 * R + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    MAD_HIP(hipMemcpyAsync(d_g2, g2, (2 * R + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    if (pre_R > 0) MAD_HIP(hipMemcpyAsync(d_pre, pre, (2 * pre_R + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    MAD_HIP(hipMemcpyAsync(d_src, h_grid, src.count() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    T *base = nullptr;
    if (hipMalloc((void **)&base, b.count() * sizeof(T)) != hipSuccess) return mad_fail(ctx, MAD_ENOMEM, "mad_space_build: base volume");
    hipLaunchKernelGGL((k_pad<T>), dim3(blocks_for(ctx, b.count())), dim3(256), 0, ctx->stream, (const T *)d_src, src, pad, base, b);
    MAD_HIP(hipGetLastError());

    int n_oct = 0;
    if (oct_mode & 2) {
        // successive 1-D splines along axes 0, 1, 2 (MapSpace.interpn_so, :191-214), all in double
        Dims cur = b;
        const void *in = base;
        bool in_is_T = true;
        double *vol = nullptr;
        for (int a = 0; a < 3; a++) {
            const int n = cur.n[a];
            Dims o = cur;
            o.n[a] = 2 * n - 1;
            double *d_lu = (double *)A.get((size_t)5 * n * 8), *d_w = (double *)A.get((size_t)(2 * n - 1) * 32);
            int32_t *d_i = (int32_t *)A.get((size_t)(2 * n - 1) * 4);
            double *c = (double *)A.get(cur.count() * 8);
            double *outv = (double *)A.get(o.count() * 8);
            if (!d_lu || !d_w || !d_i || !c || !outv) return mad_fail(ctx, MAD_ENOMEM, "mad_space_build: spline volumes of %zu voxels", o.count());
            MAD_HIP(hipMemcpyAsync(d_lu, lu[a], (size_t)5 * n * 8, hipMemcpyHostToDevice, ctx->stream));
            MAD_HIP(hipMemcpyAsync(d_w, ev_w[a], (size_t)(2 * n - 1) * 32, hipMemcpyHostToDevice, ctx->stream));
            MAD_HIP(hipMemcpyAsync(d_i, ev_i[a], (size_t)(2 * n - 1) * 4, hipMemcpyHostToDevice, ctx->stream));
            const size_t n_lines = cur.count() / n;
            const unsigned lb = (unsigned)((n_lines + 255) / 256);
            if (in_is_T) hipLaunchKernelGGL((k_spline_solve<T>), dim3(lb), dim3(256), 0, ctx->stream, (const T *)in, cur, a, d_lu, c);
            else hipLaunchKernelGGL((k_spline_solve<double>), dim3(lb), dim3(256), 0, ctx->stream, (const double *)in, cur, a, d_lu, c);
            hipLaunchKernelGGL(k_spline_eval, dim3(blocks_for(ctx, o.count())), dim3(256), 0, ctx->stream, (const double *)c, cur, a, d_w, d_i, outv, o);
            MAD_HIP(hipGetLastError());
            in = outv; in_is_T = false; cur = o; vol = outv;
        }
        Octave &U = s->oct[n_oct];
        U.d = cur; U.f64 = false; U.kind = 0;
        if (hipMalloc(&U.grid, cur.count() * 4) != hipSuccess) return mad_fail(ctx, MAD_ENOMEM, "mad_space_build: upsampled volume");
        const unsigned nb = blocks_for(ctx, cur.count());
        if (pre_R > 0) {      // gaussian_filter(up, sigma=presmooth) in double, then .astype(float32) (MapSpace.py:144)
            double *p1 = (double *)A.get(cur.count() * 8);
            if (!p1) return mad_fail(ctx, MAD_ENOMEM, "mad_space_build: pre-smoothing volume");
            hipLaunchKernelGGL((k_filter_axis<double, double>), dim3(nb), dim3(256), 0, ctx->stream, (const double *)vol, cur, 0, pre_R, d_pre, (const double *)nullptr, p1, (double *)nullptr);
            hipLaunchKernelGGL((k_filter_axis<double, double>), dim3(nb), dim3(256), 0, ctx->stream, (const double *)p1, cur, 1, pre_R, d_pre, (const double *)nullptr, vol, (double *)nullptr);
            hipLaunchKernelGGL((k_filter_axis<double, float>), dim3(nb), dim3(256), 0, ctx->stream, (const double *)vol, cur, 2, pre_R, d_pre, (const double *)nullptr, (float *)U.grid, (float *)nullptr);
        } else {
            hipLaunchKernelGGL((k_convert<double, float>), dim3(nb), dim3(256), 0, ctx->stream, (const double *)vol, (float *)U.grid, cur.count());
        }
        MAD_HIP(hipGetLastError());
        n_oct++;
    }
    if (oct_mode & 1) {
        Octave &B = s->oct[n_oct];
        B.d = b; B.f64 = sizeof(T) == 8; B.kind = 1; B.grid = base;
        n_oct++;
    }
    MAD_HIP(hipStreamSynchronize(ctx->stream));
    if (!(oct_mode & 1)) (void)hipFree(base);
    s->n_oct = n_oct;
    for (int o = 0; o < n_oct; o++) {
        Octave &O = s->oct[o];
        const int slot = O.kind == 0 ? slot_up : slot_base;
        if (O.f64) MAD_TRY(build_octave<double>(ctx, O, R, d_g0, d_g2, sig2, slot));
        else MAD_TRY(build_octave<float>(ctx, O, R, d_g0, d_g2, sig2, slot));
    }
    return MAD_OK;
}

extern "C" int mad_space_build(mad_ctx *ctx, mad_space *s, const void *grid, int is_f64, int nx, int ny, int nz, int pad,
                               int oct_mode, const double *g0, const double *g2, int radius, double sig2,
                               const double *pre, int pre_radius, const double *const *lu, const double *const *ev_w,
                               const int32_t *const *ev_i, int slot_up, int slot_base) {
    if (!ctx || !s || !grid || !g0 || !g2) return MAD_EINVAL;
    if (nx < 2 || ny < 2 || nz < 2 || pad < 0 || radius < 1 || radius > 64 || pre_radius < 0 || pre_radius > 64)
        return mad_fail(ctx, MAD_EINVAL, "mad_space_build: dims %dx%dx%d pad %d radius %d/%d", nx, ny, nz, pad, radius, pre_radius);
    if (oct_mode < 1 || oct_mode > 3) return mad_fail(ctx, MAD_EINVAL, "mad_space_build: oct_mode %d", oct_mode);
    if ((oct_mode & 2) && (!lu || !ev_w || !ev_i || (pre_radius > 0 && !pre))) return mad_fail(ctx, MAD_EINVAL, "mad_space_build: spline tables missing");
    if ((oct_mode & 2) && std::min(nx, std::min(ny, nz)) + 2 * pad < 4) return mad_fail(ctx, MAD_EINVAL, "mad_space_build: a cubic spline needs 4 samples per axis");
    const size_t up = (size_t)(2 * (nx + 2 * pad) - 1) * (2 * (ny + 2 * pad) - 1) * (2 * (nz + 2 * pad) - 1);
    if ((oct_mode & 2) && up >= ((size_t)1 << 32)) return mad_fail(ctx, MAD_EINVAL, "mad_space_build: upsampled volume of %zu voxels exceeds 2^32", up);
    MAD_HIP(hipStreamSynchronize(ctx->stream));
    space_release(s);
    int rc;
    if (is_f64) rc = build_all<double>(ctx, s, (const double *)grid, nx, ny, nz, pad, oct_mode, g0, g2, radius, sig2, pre, pre_radius, lu, ev_w, ev_i, slot_up, slot_base);
    else rc = build_all<float>(ctx, s, (const float *)grid, nx, ny, nz, pad, oct_mode, g0, g2, radius, sig2, pre, pre_radius, lu, ev_w, ev_i, slot_up, slot_base);
    (void)hipStreamSynchronize(ctx->stream);
    if (rc != MAD_OK) space_release(s);
    return rc;
}

extern "C" int mad_space_info(mad_ctx *ctx, const mad_space *s, int *n_octaves, int32_t *dims6, int32_t *kind2, int32_t *is_f64_2) {
    if (!ctx || !s) return MAD_EINVAL;
    if (n_octaves) *n_octaves = s->n_oct;
    for (int o = 0; o < s->n_oct; o++) {
        if (dims6) { dims6[3 * o] = s->oct[o].d.n[0]; dims6[3 * o + 1] = s->oct[o].d.n[1]; dims6[3 * o + 2] = s->oct[o].d.n[2]; }
        if (kind2) kind2[o] = s->oct[o].kind;
        if (is_f64_2) is_f64_2[o] = s->oct[o].f64 ? 1 : 0;
    }
    return MAD_OK;
}

extern "C" int mad_space_download(mad_ctx *ctx, const mad_space *s, int entry, int what, void *out) {
    if (!ctx || !s || !out) return MAD_EINVAL;
    if (entry < 0 || entry >= s->n_oct) return mad_fail(ctx, MAD_EINVAL, "mad_space_download: entry %d of %d", entry, s->n_oct);
    const Octave &O = s->oct[entry];
    const void *src = what == 0 ? O.grid : (what == 1 ? O.logv : (what == 2 ? O.gauss : nullptr));
    if (!src) return mad_fail(ctx, MAD_EINVAL, "mad_space_download: volume %d", what);
    MAD_HIP(hipMemcpyAsync(out, src, O.d.count() * (O.f64 ? 8 : 4), hipMemcpyDeviceToHost, ctx->stream));
    MAD_HIP(hipStreamSynchronize(ctx->stream));
    return MAD_OK;
}

extern "C" int mad_space_peaks(mad_ctx *ctx, const mad_space *s, int entry, double threshold, int border, int64_t *lin_index,
                               double *value, int64_t cap, int64_t *n_out) {
    if (!ctx || !s || !n_out) return MAD_EINVAL;
    mad_use_lane(ctx, 0);
    if (entry < 0 || entry >= s->n_oct) return mad_fail(ctx, MAD_EINVAL, "mad_space_peaks: entry %d of %d", entry, s->n_oct);
    if (border < 0 || cap < 0 || cap >= ((int64_t)1 << 31)) return mad_fail(ctx, MAD_EINVAL, "mad_space_peaks: border %d cap %lld", border, (long long)cap);
    const Octave &O = s->oct[entry];
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_TMP_A), (size_t)(cap + 1) * 8));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_TMP_B), (size_t)(cap + 1) * 8));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_MISC), 256));
    int32_t *d_cnt = scratch<int32_t>(ctx, S_MISC);
    MAD_HIP(hipMemsetAsync(d_cnt, 0, 4, ctx->stream));
    const unsigned nb = blocks_for(ctx, O.d.count());
    if (O.f64) hipLaunchKernelGGL((k_peaks<double>), dim3(nb), dim3(256), 0, ctx->stream, (const double *)O.logv, O.d, threshold, border, scratch<int64_t>(ctx, S_TMP_A), scratch<double>(ctx, S_TMP_B), d_cnt, (int)cap);
    else hipLaunchKernelGGL((k_peaks<float>), dim3(nb), dim3(256), 0, ctx->stream, (const float *)O.logv, O.d, threshold, border, scratch<int64_t>(ctx, S_TMP_A), scratch<double>(ctx, S_TMP_B), d_cnt, (int)cap);
    MAD_HIP(hipGetLastError());
    MAD_HIP(hipMemcpyAsync(&ctx->pinned[0], d_cnt, 4, hipMemcpyDeviceToHost, ctx->stream));
    MAD_HIP(hipStreamSynchronize(ctx->stream));
    const int64_t n = *(const int32_t *)&ctx->pinned[0];
    *n_out = n;
    if (n > cap) return mad_fail(ctx, MAD_ENOSPC, "mad_space_peaks: %lld peaks, capacity %lld", (long long)n, (long long)cap);
    if (n > 0 && lin_index) MAD_HIP(hipMemcpyAsync(lin_index, mad_sb(ctx, S_TMP_A).p, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (n > 0 && value) MAD_HIP(hipMemcpyAsync(value, mad_sb(ctx, S_TMP_B).p, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    MAD_HIP(hipStreamSynchronize(ctx->stream));
    return MAD_OK;
}

extern "C" int mad_space_patches(mad_ctx *ctx, const mad_space *s, int entry, const int32_t *coords, int n, int r, void *out) {
    if (!ctx || !s || (n > 0 && (!coords || !out))) return MAD_EINVAL;
    mad_use_lane(ctx, 0);
    if (entry < 0 || entry >= s->n_oct) return mad_fail(ctx, MAD_EINVAL, "mad_space_patches: entry %d of %d", entry, s->n_oct);
    if (n <= 0) return MAD_OK;
    if (r < 1 || r > 16) return mad_fail(ctx, MAD_EINVAL, "mad_space_patches: radius %d", r);
    const Octave &O = s->oct[entry];
    const int side = 2 * r + 1;
    const size_t total = (size_t)side * side * side * n, esz = O.f64 ? 8 : 4;
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_TMP_A), (size_t)n * 12));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_TMP_B), total * esz));
    MAD_HIP(hipMemcpyAsync(mad_sb(ctx, S_TMP_A).p, coords, (size_t)n * 12, hipMemcpyHostToDevice, ctx->stream));
    const unsigned nb = blocks_for(ctx, total);
    if (O.f64) hipLaunchKernelGGL((k_patches<double>), dim3(nb), dim3(256), 0, ctx->stream, (const double *)O.logv, O.d, scratch<int32_t>(ctx, S_TMP_A), n, r, scratch<double>(ctx, S_TMP_B));
    else hipLaunchKernelGGL((k_patches<float>), dim3(nb), dim3(256), 0, ctx->stream, (const float *)O.logv, O.d, scratch<int32_t>(ctx, S_TMP_A), n, r, scratch<float>(ctx, S_TMP_B));
    MAD_HIP(hipGetLastError());
    MAD_HIP(hipMemcpyAsync(out, mad_sb(ctx, S_TMP_B).p, total * esz, hipMemcpyDeviceToHost, ctx->stream));
    MAD_HIP(hipStreamSynchronize(ctx->stream));
    return MAD_OK;
}
