// mad_ctx.hip -- context, error reporting, device memory, EQSP tables, gradient-field
// upload, timing and the prefix-sum utility of libmad_amd.so (gfx950).
#include "mad_common.h"

static char g_init_err[512] = "";
#define MAD_TWO_PI_H 6.283185307179586476925286766559

int mad_fail(mad_ctx *ctx, int code, const char *fmt, ...) {
    char *dst = ctx ? ctx->err : g_init_err;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 512, fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char *mad_last_error(const mad_ctx *ctx) { return ctx ? ctx->err : g_init_err; }

int mad_reserve(mad_ctx *ctx, DevBuf &b, size_t bytes) {
    if (bytes == 0) bytes = 16;
    if (b.cap >= bytes) return MAD_OK;
    if (b.p) {
        MAD_HIP(hipStreamSynchronize(ctx->stream));
        (void)hipFree(b.p);
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = bytes + bytes / 4 + 256;    // 25 % slack: the buffers only grow
    want = (want + 255) & ~size_t(255);
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        b.p = nullptr;
        return mad_fail(ctx, MAD_ENOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    }
    b.cap = want;
    ctx->n_device_allocs++;
    return MAD_OK;
}

// how many device buffers the context has (re)allocated so far: every one of them is a hipMalloc, and a hipFree that waits for all
// streams when it replaces a smaller buffer -- a pipeline in its steady state must show none (bench.py checks its timed region)
extern "C" int64_t mad_device_allocations(mad_ctx *ctx) { return ctx ? ctx->n_device_allocs : -1; }

void mad_release(DevBuf &b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

extern "C" int mad_init(int device, mad_ctx **out) {
    if (!out) return mad_fail(nullptr, MAD_EINVAL, "mad_init: out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return mad_fail(nullptr, MAD_ENODEV, "mad_init: no HIP device (%s)", hipGetErrorString(e));
    if (device < 0 || device >= ndev) return mad_fail(nullptr, MAD_EINVAL, "mad_init: device %d of %d", device, ndev);
    e = hipSetDevice(device);
    if (e != hipSuccess) return mad_fail(nullptr, MAD_ENODEV, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return mad_fail(nullptr, MAD_ENODEV, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return mad_fail(nullptr, MAD_ENODEV, "mad_init: device %d is %s; this library is built for gfx950 only", device,
                        prop.gcnArchName);
    mad_ctx *ctx = new mad_ctx();
    ctx->spatial_order = getenv("MAD_NO_SPATIAL_ORDER") == nullptr;      // diagnostic switch: build in list order
    ctx->batch_gemm = getenv("MAD_GEMM_BATCH") != nullptr;
    if (getenv("MAD_POSE_SPLIT_MIN")) ctx->pose_split_min = atoll(getenv("MAD_POSE_SPLIT_MIN"));
    if (getenv("MAD_POSE_SPLIT")) ctx->pose_split = atoi(getenv("MAD_POSE_SPLIT"));
    if (getenv("MAD_POSE_MX")) ctx->pose_mx = atoi(getenv("MAD_POSE_MX")) != 0;
    if (getenv("MAD_BALL")) ctx->dsc_ball = atoi(getenv("MAD_BALL")) != 0;      // k_describe_ball for the base-octave anchors (off by default: DESIGN.md section 6d)
    ctx->device = device;
    ctx->n_cu = prop.multiProcessorCount;
    for (int i = 0; i < MAD_MAX_FIELDS; i++) {
        ctx->fields[i] = FieldDev{nullptr, 0, 0, 0};
        ctx->field_mem[i] = nullptr;
    }
    bool streams_ok = true;
    for (int l = 0; l < MAD_LANES; l++) streams_ok &= hipStreamCreateWithFlags(&ctx->lane_stream[l], hipStreamNonBlocking) == hipSuccess;
    ctx->stream = ctx->lane_stream[0];
    if (!streams_ok ||
        hipMalloc((void **)&ctx->eq[0], sizeof(EqspDev)) != hipSuccess ||
        hipMalloc((void **)&ctx->eq[1], sizeof(EqspDev)) != hipSuccess ||
        hipHostMalloc((void **)&ctx->pinned, 8192) != hipSuccess) {
        mad_fail(nullptr, MAD_EHIP, "mad_init: stream / table allocation failed");
        delete ctx;
        return MAD_EHIP;
    }
    for (int g = 0; g < MAD_T_COUNT; g++)
        for (int i = 0; i < MAD_T_RING; i++) {
            (void)hipEventCreate(&ctx->timers[g].start[i]);
            (void)hipEventCreate(&ctx->timers[g].stop[i]);
        }
    for (int r = 0; r < MAD_BRACKETS; r++)
        for (int l = 0; l < MAD_RES; l++) (void)hipEventCreateWithFlags(&ctx->lane_done[r][l], hipEventDisableTiming);
    for (int l = 0; l < MAD_LANES; l++) (void)hipEventCreateWithFlags(&ctx->lane_pre[l], hipEventDisableTiming);
    for (int r = 0; r < MAD_BRACKETS; r++) (void)hipEventCreateWithFlags(&ctx->gemm_done[r], hipEventDisableTiming);
    *out = ctx;
    return MAD_OK;
}

extern "C" void mad_destroy(mad_ctx *ctx) {
    if (ctx) { mad_synchronize(ctx); mad_many_abandon(ctx); }
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    for (int l = 0; l < MAD_LANES; l++) (void)hipStreamSynchronize(ctx->lane_stream[l]);
    for (int i = 0; i < MAD_MAX_FIELDS; i++)
        if (ctx->field_mem[i]) (void)hipFree(ctx->field_mem[i]);
    for (auto &b : ctx->scratch) mad_release(b);
    if (ctx->eq[0]) (void)hipFree(ctx->eq[0]);
    if (ctx->eq[1]) (void)hipFree(ctx->eq[1]);
    if (ctx->mask_off) (void)hipFree(ctx->mask_off);
    if (ctx->gw_tab) (void)hipFree(ctx->gw_tab);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->dens.grid) (void)hipFree(ctx->dens.grid);
    if (ctx->dens.grad) (void)hipFree(ctx->dens.grad);
    for (int g = 0; g < MAD_T_COUNT; g++)
        for (int i = 0; i < MAD_T_RING; i++) {
            (void)hipEventDestroy(ctx->timers[g].start[i]);
            (void)hipEventDestroy(ctx->timers[g].stop[i]);
        }
    for (int r = 0; r < MAD_BRACKETS; r++) (void)hipEventDestroy(ctx->gemm_done[r]);
    for (int l = 0; l < MAD_LANES; l++) (void)hipEventDestroy(ctx->lane_pre[l]);
    for (int l = 0; l < MAD_LANES; l++)
        for (int r = 0; r < MAD_SHARD_RING; r++) {
            if (ctx->shard_ev[l][r]) (void)hipEventDestroy(ctx->shard_ev[l][r]);
            if (ctx->shard_host[l][r]) (void)hipHostFree(ctx->shard_host[l][r]);
        }
    for (int l = 0; l < MAD_RES; l++)
        for (int r = 0; r < MAD_BRACKETS; r++) {
            (void)hipEventDestroy(ctx->lane_done[r][l]);
            if (ctx->host_res[r][l]) (void)hipHostFree(ctx->host_res[r][l]);
        }
    for (int l = 0; l < MAD_LANES; l++)
        if (ctx->lane_stream[l]) (void)hipStreamDestroy(ctx->lane_stream[l]);
    delete ctx;
}

extern "C" int mad_synchronize(mad_ctx *ctx) {
    if (!ctx) return MAD_EINVAL;
    for (int l = 0; l < MAD_LANES; l++) MAD_HIP(hipStreamSynchronize(ctx->lane_stream[l]));
    return MAD_OK;
}

extern "C" int mad_set_overlap(mad_ctx *ctx, int on) {
    if (!ctx) return MAD_EINVAL;
    MAD_TRY(mad_synchronize(ctx));
    ctx->overlap = on != 0;
    mad_use_lane(ctx, 0);
    return MAD_OK;
}

extern "C" void *mad_stream(mad_ctx *ctx) { return ctx ? (void *)ctx->lane_stream[0] : nullptr; }

// ---------------------------------------------------------------------------
// timing: HIP events on the ctx stream around each kernel group
// ---------------------------------------------------------------------------

static void timer_drain(mad_ctx *ctx, TimerGroup &t) {
    for (int i = 0; i < t.pending; i++) {
        float ms = 0;
        if (hipEventSynchronize(t.stop[i]) == hipSuccess && hipEventElapsedTime(&ms, t.start[i], t.stop[i]) == hipSuccess)
            t.total_ms += ms;
        t.launches++;
    }
    t.pending = 0;
}

void mad_timer_begin(mad_ctx *ctx, int group) {
    if (!ctx->timing) return;
    TimerGroup &t = ctx->timers[group];
    if (t.pending == MAD_T_RING) timer_drain(ctx, t);
    (void)hipEventRecord(t.start[t.pending], ctx->stream);
}

void mad_timer_end(mad_ctx *ctx, int group) {
    if (!ctx->timing) return;
    TimerGroup &t = ctx->timers[group];
    (void)hipEventRecord(t.stop[t.pending], ctx->stream);
    t.pending++;
}

static const char *k_timer_names[MAD_T_COUNT] = {"orient", "describe", "correlate", "pairs", "pose",
                                                 "topk",   "refine",   "density",   "ccc"};

extern "C" int mad_timing_enable(mad_ctx *ctx, int on) {
    if (!ctx) return MAD_EINVAL;
    ctx->timing = on != 0;
    return MAD_OK;
}

extern "C" int mad_timing_reset(mad_ctx *ctx) {
    if (!ctx) return MAD_EINVAL;
    for (int g = 0; g < MAD_T_COUNT; g++) {
        timer_drain(ctx, ctx->timers[g]);
        ctx->timers[g].total_ms = 0;
        ctx->timers[g].launches = 0;
    }
    return MAD_OK;
}

extern "C" int mad_timing_get(mad_ctx *ctx, const char *what, double *total_ms, int64_t *launches) {
    if (!ctx || !what) return MAD_EINVAL;
    for (int g = 0; g < MAD_T_COUNT; g++)
        if (strcmp(what, k_timer_names[g]) == 0) {
            timer_drain(ctx, ctx->timers[g]);
            if (total_ms) *total_ms = ctx->timers[g].total_ms;
            if (launches) *launches = ctx->timers[g].launches;
            return MAD_OK;
        }
    return mad_fail(ctx, MAD_EINVAL, "mad_timing_get: unknown group '%s'", what);
}

extern "C" double mad_last_ms(mad_ctx *ctx, const char *what) {
    double t = -1;
    int64_t n = 0;
    if (mad_timing_get(ctx, what, &t, &n) != MAD_OK || n == 0) return -1.0;
    return t / (double)n;
}

// ---------------------------------------------------------------------------
// EQSP tables
// ---------------------------------------------------------------------------

extern "C" int mad_set_eqsp(mad_ctx *ctx, int which, int Z, const double *bounds, const double *to_dom,
                            const double *adj_sec) {
    if (!ctx) return MAD_EINVAL;
    if (which < 0 || which > 1 || Z < 2 || Z > MAD_MAX_Z || !bounds)
        return mad_fail(ctx, MAD_EINVAL, "mad_set_eqsp: which=%d Z=%d", which, Z);
    if (which == 0 && (!to_dom || !adj_sec)) return mad_fail(ctx, MAD_EINVAL, "mad_set_eqsp: orientation table needs matrices");
    EqspDev &h = ctx->eq_host[which];
    memset(&h, 0, sizeof(h));
    h.Z = Z;
    double prev = -1.0;
    int nb = -1;
    for (int a = 0; a < Z; a++) {
        h.th_lo[a] = bounds[4 * a];
        h.th_hi[a] = bounds[4 * a + 2];
        if (nb < 0 || bounds[4 * a + 1] != prev) {      // a new belt starts when phi_min changes (eqsp.py:40)
            nb++;
            h.ph_lo[nb] = bounds[4 * a + 1];
            h.ph_hi[nb] = bounds[4 * a + 3];
            h.belt_first[nb] = a;
            h.belt_count[nb] = 0;
            prev = bounds[4 * a + 1];
        } else if (bounds[4 * a + 3] != h.ph_hi[nb]) {
            return mad_fail(ctx, MAD_EINVAL, "mad_set_eqsp: zone %d does not share its belt's phi_max", a);
        }
        h.belt_count[nb]++;
    }
    h.nbelt = nb + 1;
    if (h.nbelt > MAD_MAX_BELT) return mad_fail(ctx, MAD_EINVAL, "mad_set_eqsp: %d belts", h.nbelt);
    for (int b = 0; b < h.nbelt; b++) {
        // inner thresholds, rounded towards the inside of the belt and kept off the poles
        float lo = nextafterf((float)cos(h.ph_lo[b] + MAD_EQSP_GUARD), -2.0f);
        float hi = nextafterf((float)cos(h.ph_hi[b] - MAD_EQSP_GUARD), 2.0f);
        if (lo > 1.0f - 1e-6f) lo = 1.0f - 1e-6f;
        if (hi < -1.0f + 1e-6f) hi = -1.0f + 1e-6f;
        h.z_in_lo[b] = lo;
        h.z_in_hi[b] = hi;
        h.belt_lo0[b] = (float)h.th_lo[h.belt_first[b]];
        h.belt_inv_w[b] = (float)(h.belt_count[b] / MAD_TWO_PI_H);
        h.belt_first32[b] = h.belt_first[b];
        h.belt_count32[b] = h.belt_count[b];
    }
    for (int i = 0; i < MAD_ZLUT; i++) {
        const double zc = -1.0 + (i + 0.5) * (2.0 / MAD_ZLUT);
        const double ph = acos(zc);
        int best = 0;
        for (int b = 0; b < h.nbelt; b++)
            if (ph >= h.ph_lo[b] && ph < h.ph_hi[b]) best = b;
        h.zlut[i] = (unsigned char)best;
    }
    for (int a = 0; a < Z; a++) {
        h.g32[a][0] = (float)cos(h.th_lo[a] + MAD_EQSP_GUARD); h.g32[a][1] = (float)sin(h.th_lo[a] + MAD_EQSP_GUARD);
        h.g32[a][2] = (float)cos(h.th_hi[a] - MAD_EQSP_GUARD); h.g32[a][3] = (float)sin(h.th_hi[a] - MAD_EQSP_GUARD);
    }
    {      // the image the kernels copy into LDS
        EqspFastLds &im = h.image;
        memset(&im, 0, sizeof(im));
        memcpy(im.g32, h.g32, sizeof(im.g32));
        for (int b = 0; b < MAD_MAX_BELT; b++) {
            im.z_in_lo[b] = h.z_in_lo[b]; im.z_in_hi[b] = h.z_in_hi[b]; im.belt_lo0[b] = h.belt_lo0[b]; im.belt_inv_w[b] = h.belt_inv_w[b];
            im.belt_first[b] = h.belt_first32[b]; im.belt_count[b] = h.belt_count32[b];
            im.belt_f[b] = make_float4(h.z_in_lo[b], h.z_in_hi[b], h.belt_lo0[b], h.belt_inv_w[b]);
            im.belt_i[b] = make_int2(h.belt_first32[b], h.belt_count32[b]);
            im.ph_lo[b] = h.ph_lo[b]; im.ph_hi[b] = h.ph_hi[b];
        }
        memcpy(im.zlut, h.zlut, sizeof(im.zlut));
        for (int a = 0; a < MAD_MAX_Z; a++) { im.th_lo[a] = h.th_lo[a]; im.th_hi[a] = h.th_hi[a]; }
        im.nbelt = h.nbelt;
        im.tier2_ok = getenv("MAD_NO_TIER2") ? 0 : 1;      // diagnostic switch
        for (int a = 0; a < Z; a++) {
            im.dir[a][0] = cos(h.th_lo[a]); im.dir[a][1] = sin(h.th_lo[a]);
            im.dir[a][2] = cos(h.th_hi[a]); im.dir[a][3] = sin(h.th_hi[a]);
        }
        for (int b = 0; b < h.nbelt; b++) {
            const double m[2] = {MAD_T2_MARGIN64, MAD_T2_MARGIN32};
            for (int k = 0; k < 2; k++) {
                // phi > ph_lo  <=>  z < cos(ph_lo); a bound at or beyond a pole is no constraint inside (-1, 1)
                im.zthr[b][2 * k] = h.ph_lo[b] <= 0.0 ? 2.0 : cos(h.ph_lo[b] + m[k]);
                im.zthr[b][2 * k + 1] = h.ph_hi[b] - m[k] >= 3.14159265358979323846 ? -2.0 : cos(h.ph_hi[b] - m[k]);
            }
            if (h.belt_count[b] > 1)
                for (int a = h.belt_first[b]; a < h.belt_first[b] + h.belt_count[b]; a++)
                    if (!(h.th_hi[a] - h.th_lo[a] < 3.0) || !(h.th_hi[a] > h.th_lo[a])) im.tier2_ok = 0;
            if (b + 1 < h.nbelt && h.ph_hi[b] != h.ph_lo[b + 1]) im.tier2_ok = 0;      // belts must share their bounds
        }
    }
    {   // table classifier of the 4-byte texels (EqspTabLds): conservative by construction, see the struct
        EqspTabLds &T = h.tab;
        memset(&T, 255, sizeof(T));
        static const bool no_tab = getenv("MAD_NO_TAB") != nullptr;      // diagnostic switch
        h.tab_ok = (h.nbelt <= MAD_TAB_BELTS && Z <= 127 && !no_tab) ? 1 : 0;
        const double g = MAD_TAB_GUARD, pi = 3.14159265358979323846;
        for (int k = 0; h.tab_ok && k < MAD_TAB_ZBINS; k++) {
            const double zlo = -1.0 + (k - 1) * (2.0 / MAD_TAB_ZBINS), zhi = -1.0 + (k + 2) * (2.0 / MAD_TAB_ZBINS);      // the bin and its neighbours
            if (zlo <= -1.0 || zhi >= 1.0) continue;
            const double ph_min = acos(zhi) - g, ph_max = acos(zlo) + g;
            for (int b = 0; b < h.nbelt; b++)
                if (ph_min > h.ph_lo[b] && ph_max < h.ph_hi[b]) T.zbelt[k] = (unsigned char)b;
        }
        auto theta_of = [&](double p) {      // inverse of eqsp_tab32's pseudo-angle
            if (p <= 2.0) { const double xr = 1.0 - p; return atan2(1.0 - fabs(xr), xr); }
            const double xr = p - 3.0;
            return atan2(-(1.0 - fabs(xr)), xr) + 2.0 * pi;
        };
        for (int b = 0; h.tab_ok && b < h.nbelt; b++) {
            const int a0 = h.belt_first[b], cnt = h.belt_count[b];
            unsigned char *row = T.ptab[b];
            if (cnt == 1) { memset(row, a0, MAD_TAB_PBINS); continue; }      // a polar cap spans every azimuth
            const double s_min = std::min(sin(h.ph_lo[b]), sin(h.ph_hi[b]));        // sin(phi) is concave on [0, pi]
            if (!(s_min > 0.05)) continue;                                          // an angular error is an azimuth error / sin(phi)
            const double gt = g / s_min;
            for (int k = 2; k < MAD_TAB_PBINS - 2; k++) {                           // (the bins at the 0 / 2 pi seam stay undecided)
                const double t0 = theta_of((k - 1) * (4.0 / MAD_TAB_PBINS)) - gt, t1 = theta_of((k + 2) * (4.0 / MAD_TAB_PBINS)) + gt;
                for (int a = a0; a < a0 + cnt; a++) {
                    const bool in = (t0 > h.th_lo[a] && t1 < h.th_hi[a]) || (t0 + 2.0 * pi > h.th_lo[a] && t1 + 2.0 * pi < h.th_hi[a]);
                    if (in) row[k] = (unsigned char)a;
                }
            }
        }
    }
    if (to_dom) memcpy(h.to_dom, to_dom, sizeof(double) * 9 * Z);
    if (adj_sec) memcpy(h.adj_sec, adj_sec, sizeof(double) * 9 * Z);
    MAD_HIP(hipMemcpyAsync(ctx->eq[which], &h, sizeof(EqspDev), hipMemcpyHostToDevice, ctx->stream));
    MAD_HIP(hipStreamSynchronize(ctx->stream));
    ctx->eq_set[which] = true;
    return MAD_OK;
}

// ---------------------------------------------------------------------------
// gradient fields
// ---------------------------------------------------------------------------

// planar {gx,gy,gz} -> texel {gx,gy,gz,|g|}; |g| with one float32 rounding per operation
__global__ __launch_bounds__(256) void k_pack_field(const float *__restrict__ gx, const float *__restrict__ gy,
                                                    const float *__restrict__ gz, float4 *__restrict__ tex, unsigned *__restrict__ tex4, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += step) {
        const float x = gx[i], y = gy[i], z = gz[i];
        const float s = __fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z));
        const float w = sqrtf(s);      // sqrtf is correctly rounded here; the "_rn" sqrt intrinsic is NOT (1 ulp off for 15 % of inputs, measured)
        tex[i] = make_float4(x, y, z, w);
        tex4[i] = mad_tex4_encode(x, y, z, w);
    }
}

int mad_field_alloc(mad_ctx *ctx, int slot, int nx, int ny, int nz, size_t *n_out) {
    if (slot < 0 || slot >= MAD_MAX_FIELDS) return mad_fail(ctx, MAD_EINVAL, "field slot %d out of range", slot);
    if (nx < 2 || ny < 2 || nz < 2) return mad_fail(ctx, MAD_EINVAL, "field dims %dx%dx%d", nx, ny, nz);
    const size_t n = (size_t)nx * ny * nz;
    // the kernels index texels with 24-bit multiplies: (x ny + y) and nz below 2^24, the whole field below 2^32 texels
    if (n >= ((size_t)1 << 32) || (size_t)nx * ny >= ((size_t)1 << 24) || nz >= (1 << 24))
        return mad_fail(ctx, MAD_EINVAL, "field of %dx%dx%d texels: more than 2^32 texels or 2^24 per x-y plane", nx, ny, nz);
    if (ctx->field_mem[slot]) {
        MAD_HIP(hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->field_mem[slot]);
        ctx->field_mem[slot] = nullptr;
        ctx->fields[slot] = FieldDev{nullptr, 0, 0, 0};
    }
    hipError_t e = hipMalloc(&ctx->field_mem[slot], n * (sizeof(float4) + sizeof(unsigned)));      // the 16-byte texels, then the 4-byte ones
    if (e != hipSuccess) {
        ctx->field_mem[slot] = nullptr;
        return mad_fail(ctx, MAD_ENOMEM, "field of %zu texels: %s", n, hipGetErrorString(e));
    }
    ctx->fields[slot] = FieldDev{(const float4 *)ctx->field_mem[slot], nx, ny, nz, (const unsigned *)((const float4 *)ctx->field_mem[slot] + n)};
    *n_out = n;
    return MAD_OK;
}

extern "C" int mad_upload_field_device(mad_ctx *ctx, int slot, const float *g3, int nx, int ny, int nz) {
    if (!ctx || !g3) return MAD_EINVAL;
    size_t n = 0;
    MAD_TRY(mad_field_alloc(ctx, slot, nx, ny, nz, &n));
    const int blocks = (int)std::min<size_t>(mad_ceil_div((int64_t)n, 256), (size_t)ctx->n_cu * 16);
    hipLaunchKernelGGL(k_pack_field, dim3(blocks), dim3(256), 0, ctx->stream, g3, g3 + n, g3 + 2 * n,
                       (float4 *)ctx->field_mem[slot], (unsigned *)((float4 *)ctx->field_mem[slot] + n), n);
    MAD_HIP(hipGetLastError());
    MAD_HIP(hipStreamSynchronize(ctx->stream));
    return MAD_OK;
}

extern "C" int mad_upload_field(mad_ctx *ctx, int slot, const float *gx, const float *gy, const float *gz, int nx,
                                int ny, int nz) {
    if (!ctx || !gx || !gy || !gz) return MAD_EINVAL;
    const size_t n = (size_t)nx * ny * nz;
    float *stage = nullptr;
    hipError_t e = hipMalloc((void **)&stage, 3 * n * sizeof(float));
    if (e != hipSuccess) return mad_fail(ctx, MAD_ENOMEM, "field staging: %s", hipGetErrorString(e));
    int rc = MAD_OK;
    if (hipMemcpyAsync(stage, gx, n * sizeof(float), hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(stage + n, gy, n * sizeof(float), hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(stage + 2 * n, gz, n * sizeof(float), hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        rc = mad_fail(ctx, MAD_EHIP, "field upload failed");
    if (rc == MAD_OK) rc = mad_upload_field_device(ctx, slot, stage, nx, ny, nz);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(stage);
    return rc;
}

extern "C" int mad_free_field(mad_ctx *ctx, int slot) {
    if (!ctx || slot < 0 || slot >= MAD_MAX_FIELDS) return MAD_EINVAL;
    if (ctx->field_mem[slot]) {
        MAD_HIP(hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->field_mem[slot]);
    }
    ctx->field_mem[slot] = nullptr;
    ctx->fields[slot] = FieldDev{nullptr, 0, 0, 0};
    return MAD_OK;
}

// ---------------------------------------------------------------------------
// small fills and copies as ordinary kernels: one launch each, in stream order, no copy-engine hand-over
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_zero_words(uint4 *__restrict__ p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(0, 0, 0, 0);
}

__global__ __launch_bounds__(256) void k_copy_words(uint4 *__restrict__ dst, const uint4 *__restrict__ src, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

void mad_zero_words(mad_ctx *ctx, void *p, size_t bytes) {
    const size_t n16 = (bytes + 15) / 16;
    hipLaunchKernelGGL(k_zero_words, dim3((unsigned)std::min<size_t>((n16 + 255) / 256, 1024)), dim3(256), 0, ctx->stream, (uint4 *)p, n16);
}

// up to three regions in one launch (a match's status words, its bitmaps and the row descriptors of its pair compaction)
__global__ __launch_bounds__(256) void k_zero_words3(uint4 *__restrict__ p, size_t n16, uint4 *__restrict__ q, size_t m16, uint4 *__restrict__ r,
                                                     size_t l16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16 + m16 + l16; i += (size_t)gridDim.x * blockDim.x) {
        if (i < n16) p[i] = make_uint4(0, 0, 0, 0);
        else if (i < n16 + m16) q[i - n16] = make_uint4(0, 0, 0, 0);
        else r[i - n16 - m16] = make_uint4(0, 0, 0, 0);
    }
}

void mad_zero_words3(mad_ctx *ctx, void *p, size_t bytes_p, void *q, size_t bytes_q, void *r, size_t bytes_r) {
    const size_t n16 = (bytes_p + 15) / 16, m16 = (bytes_q + 15) / 16, l16 = (bytes_r + 15) / 16;
    hipLaunchKernelGGL(k_zero_words3, dim3((unsigned)std::min<size_t>((n16 + m16 + l16 + 255) / 256, 1024)), dim3(256), 0, ctx->stream, (uint4 *)p,
                       n16, (uint4 *)q, m16, (uint4 *)r, l16);
}

void mad_copy_words(mad_ctx *ctx, void *dst, const void *src, size_t bytes) {
    const size_t n16 = (bytes + 15) / 16;
    hipLaunchKernelGGL(k_copy_words, dim3((unsigned)std::min<size_t>((n16 + 255) / 256, 1024)), dim3(256), 0, ctx->stream, (uint4 *)dst,
                       (const uint4 *)src, n16);
}

// ---------------------------------------------------------------------------
// device peaks measured on the spot (bench.py reports roofline fractions against these as well as against the spec figures)
// ---------------------------------------------------------------------------

// streaming copy, 16 bytes per lane, grid-stride: the HBM rate a kernel of this library can hope for (read + write counted)
__global__ __launch_bounds__(256) void k_probe_copy(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16) {
    const size_t T = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * T < n16; i += 4 * T) {      // four requests in flight per lane
        const uint4 a = src[i], b = src[i + T], c = src[i + 2 * T], d = src[i + 3 * T];
        dst[i] = a; dst[i + T] = b; dst[i + 2 * T] = c; dst[i + 3 * T] = d;
    }
    for (; i < n16; i += T) dst[i] = src[i];
}
// the same copy with one 16-byte element per thread (the form MI355X_MICROARCH.md quotes 6.29 TB/s for); the probe reports the better
__global__ __launch_bounds__(256) void k_probe_copy1(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}

typedef int probe_v4i __attribute__((ext_vector_type(4)));
// dense int8 MFMA issue rate: every wave keeps eight independent 16x16 accumulators busy with v_mfma_i32_16x16x64_i8 (the
// instruction of k_corr_gemm2), operands in registers, nothing else in the loop
__global__ __launch_bounds__(256) void k_probe_mfma_i8(int iters, int *__restrict__ sink) {
    probe_v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (int)blockIdx.x};
    probe_v4i c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    // (inline assembly: left to itself the compiler rotates the accumulators through v_accvgpr moves, 40 of them per trip)
#define MAD_PROBE_MFMA(c) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
    for (int it = 0; it < iters; it++) {
        MAD_PROBE_MFMA(c0); MAD_PROBE_MFMA(c1); MAD_PROBE_MFMA(c2); MAD_PROBE_MFMA(c3);
        MAD_PROBE_MFMA(c4); MAD_PROBE_MFMA(c5); MAD_PROBE_MFMA(c6); MAD_PROBE_MFMA(c7);
    }
#undef MAD_PROBE_MFMA
    const probe_v4i t = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    const int v = t[0] + t[1] + t[2] + t[3];
    if (v == 0x7fffffff) sink[0] = v;      // keeps the loop alive
}

extern "C" int mad_probe_peaks(mad_ctx *ctx, double *copy_gbs, double *i8_tops) {
    if (!ctx) return MAD_EINVAL;
    mad_use_lane(ctx, 0);
    MAD_TRY(mad_synchronize(ctx));
    hipEvent_t e0, e1;
    MAD_HIP(hipEventCreate(&e0));
    MAD_HIP(hipEventCreate(&e1));
    float ms = 0.f;
    if (copy_gbs) {
        const size_t bytes = (size_t)1 << 30;      // 1 GiB in, 1 GiB out: far beyond the 256 MB of Infinity Cache
        void *src = nullptr, *dst = nullptr;
        if (hipMalloc(&src, bytes) != hipSuccess || hipMalloc(&dst, bytes) != hipSuccess) {
            if (src) (void)hipFree(src);
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
            return mad_fail(ctx, MAD_ENOMEM, "mad_probe_peaks: 2 GiB of probe buffers");
        }
        MAD_HIP(hipMemsetAsync(src, 1, bytes, ctx->stream));
        double best = 0.0;
        for (int rep = 0; rep < 7; rep++) {      // the first pass warms the page tables; then three of each form
            MAD_HIP(hipEventRecord(e0, ctx->stream));
            if (rep & 1) hipLaunchKernelGGL(k_probe_copy, dim3(ctx->n_cu * 16), dim3(256), 0, ctx->stream, (const uint4 *)src, (uint4 *)dst, bytes / 16);
            else hipLaunchKernelGGL(k_probe_copy1, dim3((unsigned)(bytes / 16 / 256)), dim3(256), 0, ctx->stream, (const uint4 *)src, (uint4 *)dst, bytes / 16);
            MAD_HIP(hipEventRecord(e1, ctx->stream));
            MAD_HIP(hipEventSynchronize(e1));
            MAD_HIP(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0) best = std::max(best, 2.0 * (double)bytes / (ms * 1e-3) / 1e9);
        }
        *copy_gbs = best;
        (void)hipFree(src);
        (void)hipFree(dst);
    }
    if (i8_tops) {
        int *sink = nullptr;
        MAD_HIP(hipMalloc((void **)&sink, 64));
        const int iters = 20000, wgs = ctx->n_cu * 4;      // 4 workgroups x 4 waves per CU: four waves per SIMD
        double best = 0.0;
        for (int rep = 0; rep < 3; rep++) {
            MAD_HIP(hipEventRecord(e0, ctx->stream));
            hipLaunchKernelGGL(k_probe_mfma_i8, dim3(wgs), dim3(256), 0, ctx->stream, iters, sink);
            MAD_HIP(hipEventRecord(e1, ctx->stream));
            MAD_HIP(hipEventSynchronize(e1));
            MAD_HIP(hipEventElapsedTime(&ms, e0, e1));
            const double ops = (double)wgs * 4 * (double)iters * 8 * (2.0 * 16 * 16 * 64);
            if (rep > 0) best = std::max(best, ops / (ms * 1e-3) / 1e12);
        }
        *i8_tops = best;
        (void)hipFree(sink);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return MAD_OK;
}

// ---------------------------------------------------------------------------
// exclusive prefix sum (three launches: chunk sums, scan of sums, apply)
// ---------------------------------------------------------------------------

#define SCAN_THREADS 256
#define SCAN_ITEMS 8
#define SCAN_CHUNK (SCAN_THREADS * SCAN_ITEMS)

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_sums(const int32_t *__restrict__ in, int64_t n,
                                                            int32_t *__restrict__ sums) {
    __shared__ int wt[SCAN_THREADS / MAD_WAVE];
    const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK;
    int s = 0;
    for (int i = 0; i < SCAN_ITEMS; i++) {
        const int64_t idx = base + (int64_t)i * SCAN_THREADS + threadIdx.x;
        if (idx < n) s += in[idx];
    }
    s = wave_sum_i32(s);
    if (lane_id() == 0) wt[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int i = 0; i < SCAN_THREADS / MAD_WAVE; i++) t += wt[i];
        sums[blockIdx.x] = t;
    }
}

// one block: exclusive scan of nb chunk sums in place; sums[nb] = total
__global__ __launch_bounds__(1024) void k_scan_top(int32_t *__restrict__ sums, int64_t nb) {
    __shared__ int wt[1024 / MAD_WAVE + 1];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < nb; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const int v = (i < nb) ? sums[i] : 0;
        int tot;
        const int ex = block_excl_scan(v, wt, &tot);
        if (i < nb) sums[i] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[nb] = carry;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(const int32_t *__restrict__ in, int32_t *__restrict__ out,
                                                             int64_t n, const int32_t *__restrict__ sums, int64_t nb) {
    __shared__ int wt[SCAN_THREADS / MAD_WAVE + 1];
    const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS];
    int s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        v[i] = (base + i < n) ? in[base + i] : 0;
        s += v[i];
    }
    int tot;
    int run = block_excl_scan(s, wt, &tot) + sums[blockIdx.x];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        if (base + i < n) out[base + i] = run;
        run += v[i];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = sums[nb];
}

int mad_scan_i32(mad_ctx *ctx, const int32_t *in, int32_t *out, int64_t n) {
    if (n <= 0) {
        MAD_HIP(hipMemsetAsync(out, 0, sizeof(int32_t), ctx->stream));
        return MAD_OK;
    }
    const int64_t nb = mad_ceil_div(n, SCAN_CHUNK);
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_SCAN_TMP), (size_t)(nb + 1) * sizeof(int32_t)));
    int32_t *sums = scratch<int32_t>(ctx, S_SCAN_TMP);
    hipLaunchKernelGGL(k_scan_sums, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, ctx->stream, in, n, sums);
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, ctx->stream, sums, nb);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, ctx->stream, in, out, n, sums, nb);
    MAD_HIP(hipGetLastError());
    return MAD_OK;
}

// one workgroup, length on the device: the launch-cheap form for the short scans of the pipeline
__global__ __launch_bounds__(1024) void k_scan_small(const int32_t *__restrict__ in, int32_t *__restrict__ out,
                                                     const int32_t *__restrict__ d_n, int32_t *__restrict__ total_out,
                                                     int n_host) {
    __shared__ int wt[1024 / MAD_WAVE + 1];
    __shared__ int carry;
    const int n = d_n ? *d_n : n_host;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = (i < n) ? in[i] : 0;
        int tot;
        const int ex = block_excl_scan(v, wt, &tot);
        if (i < n) out[i] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[n] = carry;
        if (total_out) *total_out = carry;
    }
}

void mad_scan_small(mad_ctx *ctx, const int32_t *in, int32_t *out, const int32_t *d_n, int32_t *total_out, int n_host) {
    hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(1024), 0, ctx->stream, in, out, d_n, total_out, n_host);
}
