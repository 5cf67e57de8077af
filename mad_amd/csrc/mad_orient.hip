// mad_orient.hip -- orientation assignment (a1-a8) and descriptor generation (a9-a10)
// for gfx950.  Reference: mad/Orientator.py:68-343, mad/Descriptor.py:106-202.
//
// Orientation: one 256-thread workgroup per anchor.  Only the voxels of the 17^3 box
// that carry weight (inside the 1.05 r sphere, |g| >= 1e-5: 2517 at most for r = 8)
// are fetched; their unit gradients are kept in LDS as float32 and every re-binning
// pass (one in float32 semantics, one per main-bin candidate in float64 after the
// rotation) runs out of LDS.
//
// Description: one 256-thread workgroup per oriented row; each thread owns a (j, k)
// column of the 16^3 sample lattice and gathers its 16 nearest-neighbour texels with
// 16 independent 16-byte loads; the 64 x 16 histogram lives in LDS.
#include <type_traits>

#include "mad_common.h"

#ifndef ORI_THREADS
#define ORI_THREADS 512
#endif
#define ORI_MAX_FAN 64          // lim_main * lim_sec must not exceed this
#define ORI_QUEUE 512           // directions the float32 classifier hands to the exact one, per pass (more: classified in place)
#define ORI_MAX_MAIN 8

// ---------------------------------------------------------------------------
// sphere mask (Orientator.py:38-47)
// ---------------------------------------------------------------------------

static int ensure_mask(mad_ctx *ctx, int r) {
    if (ctx->mask_r == r) return MAD_OK;
    const int B = 2 * r + 1;
    int8_t *h = (int8_t *)malloc((size_t)B * B * B * 4);
    int n = 0;
    for (int i = -r; i <= r; i++)
        for (int j = -r; j <= r; j++)
            for (int k = -r; k <= r; k++)
                if (sqrt((double)(i * i + j * j + k * k)) <= r * 1.05) {
                    h[4 * n] = (int8_t)i; h[4 * n + 1] = (int8_t)j; h[4 * n + 2] = (int8_t)k; h[4 * n + 3] = 0;
                    n++;
                }
    if (ctx->mask_off) {
        (void)hipDeviceSynchronize();
        (void)hipFree(ctx->mask_off);
        ctx->mask_off = nullptr;
    }
    hipError_t e = hipMalloc((void **)&ctx->mask_off, (size_t)n * 4);
    if (e == hipSuccess) e = hipMemcpy(ctx->mask_off, h, (size_t)n * 4, hipMemcpyHostToDevice);
    free(h);
    if (e != hipSuccess) return mad_fail(ctx, MAD_EHIP, "sphere mask upload: %s", hipGetErrorString(e));
    ctx->mask_r = r;
    ctx->mask_n = n;
    return MAD_OK;
}

// ---------------------------------------------------------------------------
// orientation kernel
// ---------------------------------------------------------------------------

struct OrientArgs {
    FieldDev f[2];                 // octave 0 (upsampled, stride 2) and 1 (base, stride 1)
    const int32_t *coords;         // n x 3
    const int32_t *order;          // nullable: workgroup k takes anchor order[k] (spatial order: neighbours share texels in L2)
    const int32_t *octave;         // n, or nullptr -> uniform_octave
    int uniform_octave;
    int n;
    int r;
    int nmask;
    const int8_t *mask_off;        // nmask x {dx,dy,dz,0}
    const EqspDev *eq;
    int lim_main, lim_sec;
    int fan;                       // lim_main * lim_sec
    int32_t *slot_cnt;             // n: rows produced (0 for anchors rejected at the border)
    int32_t *n_reject;             // device counter of border rejects (nullable)
    int32_t *slot_main;            // n x fan
    int32_t *slot_sec;             // n x fan
    int32_t *slot_hist;            // n x lim_main x Z quantised counts per accepted main bin (or nullptr)
    int32_t *slot_hidx;            // n x fan: which of the anchor's hist rows a slot uses
    // Orientator(gw_sig != 0): weight of a voxel at squared offset d2 from the anchor, exp(-d2 / (2 sigma^2)) in 2^-50 units
    // (Orientator.py:49-54); nullptr = every voxel of the sphere counts 1
    const unsigned long long *wfix;
    int queue_cap;                 // <= ORI_QUEUE: entries of the undecided-direction queue in use (mad_set_option "ori_queue": tests of the full-queue path)
};
#define ORI_WFIX_BITS 50


// Quantise `hist` (Z counts in LDS) to 0..50 of its max (Orientator.py:336-340) into q.
// Executed by wave 0 only; returns the max (0 = nothing counted, q left = hist).
// A weighted histogram holds fixed-point sums; the reference stores a zone's float64 sum into an int32 array
// (DensityFeature.py:50, Orientator.py:334), i.e. truncates it, before anything else looks at it.
__device__ __forceinline__ int hist_count(int v) { return v; }
__device__ __forceinline__ int hist_count(unsigned long long v) { return (int)(v >> ORI_WFIX_BITS); }

template <class H, class Q>
__device__ __forceinline__ int quantise_wave0(const H *hist, Q *q, int Z) {
    const int lane = lane_id();
    const int c0 = lane < Z ? hist_count(hist[lane]) : 0;
    const int c1 = lane + 64 < Z ? hist_count(hist[lane + 64]) : 0;
    const int mx = wave_max_i32(max(c0, c1));
    if (lane < Z) q[lane] = (Q)(mx ? (int)((double)c0 / (double)mx * 50.0) : c0);
    if (lane + 64 < Z) q[lane + 64] = (Q)(mx ? (int)((double)c1 / (double)mx * 50.0) : c1);
    return mx;
}

// The atan2 / acos test against the table's bounds -- the reference's own arithmetic, float64 (Descriptor.py:176-187) or with its
// float32 roundings (sem32: Orientator.py:317-331) -- OUT OF LINE: one direction in 1e8 gets this far, and inlined (three times
// in k_orient, twice in k_describe) the polynomial constants of the two functions were materialised, and at k_orient's 80
// registers spilled, on the common path.  Returns the matching zones, ascending, as bytes (zone + 1; at most three can match).
__device__ __noinline__ unsigned eqsp_trig_zones(const EqspFastLds *trig, double x, double y, double z, int sem32) {
    unsigned out = 0;
    int n = 0;
    auto take = [&](int zn) { if (n < 4) out |= (unsigned)(zn + 1) << (8 * n); n++; };
    if (sem32) {
        const float two_pi_f = (float)MAD_TWO_PI;
        float th = (float)atan2(y, x);
        if (th < 0.0f) th = __fadd_rn(th, two_pi_f);
        const float sth = __fadd_rn(th, two_pi_f);
        z = z > 1.0 ? 1.0 : (z < -1.0 ? -1.0 : z);
        const float ph = (float)acos(z);
        eqsp_classify_lds(trig, (double)th, (double)sth, (double)ph, take);
    } else {
        double th = atan2(y, x);
        if (th < 0) th += MAD_TWO_PI;
        const double sth = th + MAD_TWO_PI;
        z = z > 1.0 ? 1.0 : (z < -1.0 ? -1.0 : z);
        const double ph = acos(z);
        eqsp_classify_lds(trig, th, sth, ph, take);
    }
    return out;
}

// exact zone test of one rotated direction in float64 (the reference's arithmetic); f(zone) per match
template <class F>
__device__ __forceinline__ void classify_exact64(const EqspFastLds *eq, const EqspFastLds *trig, double rx, double ry, double rz, F &&f) {
    // eq: the image up to th_lo at least (LDS); trig: a whole image (the bounds the angle test walks: global memory will do)
    if (eqsp_tier2(eq, rx, ry, rz, false, f)) return;      // decided without atan2 / acos (all but ~1e-8 of the calls)
    for (unsigned zs = eqsp_trig_zones(trig, rx, ry, rz, 0); zs; zs >>= 8) f((int)(zs & 255u) - 1);
}

#ifdef MAD_PROBE_STAMPS      // diagnostic build: s_memtime at the phases of every anchor's workgroup (tools/probe_orient.py)
__device__ long long ori_stamps[4096 * 12];
#define ORI_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 4096) ori_stamps[blockIdx.x * 12 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int mad_debug_ori_stamps(long long *out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ori_stamps), (size_t)n * 8) == hipSuccess ? 0 : -1;
}
#else
#define ORI_STAMP(k) do { } while (0)
#endif

template <bool GW>
#ifndef ORI_WPE
#define ORI_WPE 6
#endif
// (registers for three workgroups per CU -- 6 waves per SIMD, 80 registers -- where the LDS allows three: the unweighted form)
__global__ __launch_bounds__(ORI_THREADS, GW ? 4 : ORI_WPE) void k_orient(Batch<OrientArgs> B) {
    ORI_STAMP(0);
    const int job = batch_job(B, (int)blockIdx.x);
    const OrientArgs &A = B.job[job];
    extern __shared__ __align__(16) unsigned char smem[];
    float *vx = (float *)smem;
    float *vy = vx + A.nmask;
    float *vz = vy + A.nmask;
    int *queue = (int *)(vz + A.nmask);                 // (voxel | candidate << 16) the fast classifier could not decide: ORI_QUEUE entries
    // GW (Gaussian window): the kept voxels' squared offsets and the weight table, behind the queue
    unsigned short *vw = (unsigned short *)(queue + ORI_QUEUE);
    unsigned long long *wl = (unsigned long long *)(smem + (((size_t)A.nmask * 14 + ORI_QUEUE * 4 + 15) & ~(size_t)15));
    using H = typename std::conditional<GW, unsigned long long, int>::type;
    __shared__ H hist[ORI_MAX_MAIN + 1][MAD_MAX_Z];     // [0]: first pass; [1 + c]: main-bin candidate c
    __shared__ short qz[ORI_MAX_MAIN + 1][MAD_MAX_Z];   // quantised counts (<= 50), same indexing
    __shared__ int main_list[MAD_MAX_Z];
    __shared__ int sec_list[ORI_MAX_MAIN][ORI_MAX_FAN];
    __shared__ int sec_cnt[ORI_MAX_MAIN];
    __shared__ int s_nvox, s_nmain, s_mx, s_nq;
    __shared__ double s_dom[ORI_MAX_MAIN][9];
    __shared__ float s_domf[ORI_MAX_MAIN][9];
    // The classifier's tables without their last part, the float64 bounds of the atan2 / acos fallback (one direction in 1e8 gets
    // that far: it reads them from the image in global memory).  With the queue cut to ORI_QUEUE entries and the quantised counts
    // as int16 an anchor's workgroup needs 52 KB of LDS instead of 65: three per CU instead of two.
    __shared__ uint4 fast_s[(offsetof(EqspFastLds, th_lo) + 15) / 16];
    const EqspFastLds &fast = *(const EqspFastLds *)fast_s;
    const EqspFastLds *const trig = &A.eq->image;

    // anchor -> octave, coordinates -> texel addresses: every link a scalar load, the field selected from both descriptors
    // (as in k_describe: the conditional-pointer forms compile into flat vector loads and a dependent fetch)
    int a = (int)blockIdx.x - B.first[job];
    if (A.order) a = __builtin_amdgcn_readfirstlane(A.order[a]);
    const int tid = threadIdx.x;
    int oct = A.uniform_octave;
    if (A.octave) oct = __builtin_amdgcn_readfirstlane(A.octave[a]);
    const FieldDev F0 = A.f[0], F1 = A.f[1];
    FieldDev F;
    F.tex = oct == 1 ? F1.tex : F0.tex; F.nx = oct == 1 ? F1.nx : F0.nx; F.ny = oct == 1 ? F1.ny : F0.ny; F.nz = oct == 1 ? F1.nz : F0.nz;
    const int stride = (oct == 1) ? 1 : 2;
    const int r = A.r;
    const int Z = A.eq->Z;
    const int x = __builtin_amdgcn_readfirstlane(A.coords[3 * a]), y = __builtin_amdgcn_readfirstlane(A.coords[3 * a + 1]),
              z = __builtin_amdgcn_readfirstlane(A.coords[3 * a + 2]);

    // step01 border test (Orientator.py:128-135 / 149-155)
    {
        const int xm = x - r * stride, ym = y - r * stride, zm = z - r * stride;
        const int xp = x + r * stride + 1, yp = y + r * stride + 1, zp = z + r * stride + 1;
        if (xm < 0 || ym < 0 || zm < 0 || xp > F.nx - 1 || yp > F.ny - 1 || zp > F.nz - 1) {
            if (tid == 0) {
                A.slot_cnt[a] = 0;
                if (A.n_reject) atomicAdd(A.n_reject, 1);
            }
            return;
        }
    }
    // step01: fetch, normalise (float32, Orientator.py:139-147), keep weighted voxels only.
    // All texels of a thread are requested before the first one is looked at (the box has ceil(nmask / 512) <= ORI_TRIPS of
    // them per thread): a trip that waits for its mask offset, then for its texel, then compacts, costs two memory round trips,
    // and five such trips in a row were a quarter of the anchor's time.  The first batch goes out BEFORE the tables are staged
    // and the histograms zeroed, so that those ~1.5 us run under the texels' flight.
    const float cutoff = 1e-5f;
    constexpr int ORI_TRIPS = 2560 / ORI_THREADS;      // r = 8: 2 517 voxels in the sphere; larger boxes take the loop below more than once
    int packed[ORI_TRIPS];
    float4 tx[ORI_TRIPS];
    auto request = [&](int m00) {
#pragma unroll
        for (int k = 0; k < ORI_TRIPS; k++) {
            const int m = m00 + k * ORI_THREADS + tid;
            packed[k] = ((const int *)A.mask_off)[min(m, A.nmask - 1)];      // {dx, dy, dz, 0} as one load
        }
#pragma unroll
        for (int k = 0; k < ORI_TRIPS; k++) {
            const int dx = (int)(int8_t)(packed[k] & 0xff), dy = (int)(int8_t)((packed[k] >> 8) & 0xff), dz = (int)(int8_t)((packed[k] >> 16) & 0xff);
            // 24-bit multiplies (full rate): nx ny < 2^24 is checked on the host, the sum stays below 2^32 texels
            const unsigned src = __umul24(__umul24((unsigned)(x + __mul24(dx, stride)), (unsigned)F.ny) + (unsigned)(y + __mul24(dy, stride)), (unsigned)F.nz) +
                                 (unsigned)(z + __mul24(dz, stride));
            tx[k] = F.tex[src];
        }
    };
    request(0);
    if (tid == 0) { s_nvox = 0; s_nq = 0; }
    for (int i = tid; i < (ORI_MAX_MAIN + 1) * MAD_MAX_Z; i += ORI_THREADS) (&hist[0][0])[i] = 0;
    if (GW)
        for (int i = tid; i <= 3 * r * r; i += ORI_THREADS) wl[i] = A.wfix[i];
    stage_lds(fast_s, &A.eq->image, sizeof(fast_s));
    // one count (or one weight) of voxel v for zone zn of histogram h
    auto tally = [&](int h, int zn, int v) {
        if (GW) atomicAdd(&hist[h][zn], (H)wl[vw[v]]);
        else atomicAdd(&hist[h][zn], (H)1);
    };
    __syncthreads();
    ORI_STAMP(1);
    for (int m00 = 0; m00 < A.nmask; m00 += ORI_TRIPS * ORI_THREADS) {
        if (m00 > 0) request(m00);
#pragma unroll
        for (int k = 0; k < ORI_TRIPS; k++) {
            if (m00 + k * ORI_THREADS >= A.nmask) break;      // workgroup-uniform: the ballot below needs whole waves
            const int m = m00 + k * ORI_THREADS + tid;
            const float4 t = tx[k];
            const bool keep = m < A.nmask && !(t.w < cutoff);
            float gx = t.x, gy = t.y, gz = t.z;
            if (t.w > cutoff) { gx = __fdiv_rn(gx, t.w); gy = __fdiv_rn(gy, t.w); gz = __fdiv_rn(gz, t.w); }
            // one LDS atomic per wave instead of one per voxel (they all hit the same word)
            const unsigned long long bal = __ballot(keep);
            int base = 0;
            if (lane_id() == 0 && bal) base = atomicAdd(&s_nvox, __popcll(bal));
            base = __shfl(base, 0, MAD_WAVE);
            if (keep) {
                const int slot = base + __popcll(bal & lanemask_lt());
                vx[slot] = gx; vy[slot] = gy; vz[slot] = gz;
                if (GW) {
                    const int dx = (int)(int8_t)(packed[k] & 0xff), dy = (int)(int8_t)((packed[k] >> 8) & 0xff), dz = (int)(int8_t)((packed[k] >> 16) & 0xff);
                    vw[slot] = (unsigned short)(dx * dx + dy * dy + dz * dz);
                }
            }
        }
    }
    __syncthreads();
    ORI_STAMP(2);
    const int nvox = s_nvox;

    // step02: first binning on the float32 box (Orientator.py:307-334).  Directions well inside a zone
    // take the guard-banded float32 path; the few near a bound are queued for the exact test.
    // the reference's float32 arithmetic for one direction of the first pass (Orientator.py:317-331)
    auto exact_first = [&](int v) {
        if (eqsp_tier2(&fast, (double)vx[v], (double)vy[v], (double)vz[v], true, [&](int zn) { tally(0, zn, v); })) return;
        for (unsigned zs = eqsp_trig_zones(trig, (double)vx[v], (double)vy[v], (double)vz[v], 1); zs; zs >>= 8) tally(0, (int)(zs & 255u) - 1, v);
    };
    // (the voxels of a thread classified side by side, in straight-line code: each classification is a chain of three dependent LDS
    // reads, and five of them one after the other were 4 500 of an anchor's 45 600 cycles; the tallies follow)
    for (int v0 = tid; v0 < nvox; v0 += ORI_TRIPS * ORI_THREADS) {
        int zn[ORI_TRIPS];
#pragma unroll
        for (int k = 0; k < ORI_TRIPS; k++) {
            const int v = min(v0 + k * ORI_THREADS, nvox - 1);
            zn[k] = eqsp_fast32<true>(&fast, vx[v], vy[v], vz[v]);
        }
#pragma unroll
        for (int k = 0; k < ORI_TRIPS; k++) {
            const int v = v0 + k * ORI_THREADS;
            if (v >= nvox) continue;
            if (zn[k] >= 0) { tally(0, zn[k], v); continue; }
            const int slot = atomicAdd(&s_nq, 1);
            if (slot < A.queue_cap) queue[slot] = v;      // (a full queue: the whole pass again below)
        }
    }
    __syncthreads();
    if (s_nq > A.queue_cap) {
        // more undecided directions than the queue holds (only if nearly every direction sat on a bound; mad_set_option "ori_queue"
        // drives it): the whole first pass again with the reference's arithmetic (uniform over the workgroup)
        for (int i = tid; i < MAD_MAX_Z; i += ORI_THREADS) hist[0][i] = 0;
        __syncthreads();
        for (int v = tid; v < nvox; v += ORI_THREADS) exact_first(v);
    }
    __syncthreads();
    ORI_STAMP(3);
    // One wave closes the first pass while the others wait at ONE barrier: the few directions the fast classifier declined
    // (the reference's float32 arithmetic), the quantisation, the main bins and their rotations -- four barrier-separated
    // phases before, a third of the anchor's time for a handful of lanes' work.
    if (tid < MAD_WAVE) {
        const int nq = s_nq > A.queue_cap ? 0 : s_nq;
        const int lane = lane_id();
        for (int qi = lane; qi < nq; qi += MAD_WAVE) exact_first(queue[qi]);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // the wave's own LDS atomics above before its reads below
        const int mx = quantise_wave0(hist[0], qz[0], Z);
        // main bins: quantised count > 0.8 * max (Orientator.py:181)
        const bool p0 = lane < Z && (double)qz[0][lane] > 50 * 0.8;
        const bool p1 = lane + 64 < Z && (double)qz[0][lane + 64] > 50 * 0.8;
        const unsigned long long m0 = __ballot(p0), m1 = __ballot(p1);
        const int n0 = __popcll(m0), nm = n0 + __popcll(m1);
        if (p0) main_list[__popcll(m0 & lanemask_lt())] = lane;
        if (p1) main_list[n0 + __popcll(m1 & lanemask_lt())] = lane + 64;
        if (lane == 0) { s_mx = mx; s_nmain = nm; s_nq = 0; }
        // step03's rotations (Orientator.py:204-206) for the accepted candidates
        if (mx != 0 && nm > 0 && nm <= A.lim_main) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            for (int i = lane; i < nm * 9; i += MAD_WAVE) {
                const double d = A.eq->to_dom[main_list[i / 9]][i % 9];
                s_dom[i / 9][i % 9] = d;
                s_domf[i / 9][i % 9] = (float)d;
            }
        }
    }
    __syncthreads();
    ORI_STAMP(4);
    ORI_STAMP(5);
    const int nmain = s_nmain;
    if (s_mx == 0 || nmain == 0 || nmain > A.lim_main) {      // Orientator.py:182-184
        if (tid == 0) A.slot_cnt[a] = 0;
        return;
    }

    // step03 for every main-bin candidate at once: rotate by to_dom (Orientator.py:303) and re-bin
    ORI_STAMP(6);
    // (a voxel's candidates classified side by side, three at a time in straight-line code, so that their chains of dependent LDS
    // reads overlap instead of following each other: this pass was 12 100 of an anchor's 45 600 cycles, two thirds of its
    // instructions -- then the tallies.  A group's last candidates may repeat the anchor's last one: classified, not tallied)
    constexpr int ORI_SIDE = 3;
    for (int v = tid; v < nvox; v += ORI_THREADS) {
        const float g0 = vx[v], g1 = vy[v], g2 = vz[v];
        for (int c0 = 0; c0 < nmain; c0 += ORI_SIDE) {
            int zn[ORI_SIDE];
#pragma unroll
            for (int u = 0; u < ORI_SIDE; u++) {
                const float *d = s_domf[min(c0 + u, nmain - 1)];
                const float rx = g0 * d[0] + g1 * d[1] + g2 * d[2];
                const float ry = g0 * d[3] + g1 * d[4] + g2 * d[5];
                const float rz = g0 * d[6] + g1 * d[7] + g2 * d[8];
                zn[u] = eqsp_fast32<true>(&fast, rx, ry, rz);
            }
#pragma unroll
            for (int u = 0; u < ORI_SIDE; u++) {
                const int c = c0 + u;
                if (c >= nmain || main_list[c] == 0) continue;      // Orientator.py:211: the pole keeps the first binning
                if (zn[u] >= 0) { tally(1 + c, zn[u], v); continue; }
                const int slot = atomicAdd(&s_nq, 1);
                if (slot < A.queue_cap) queue[slot] = v | (c << 16);      // (a full queue: the whole pass again below)
            }
        }
    }
    __syncthreads();
    ORI_STAMP(7);
    if (s_nq > A.queue_cap) {
        // the queue overflowed (see the first pass): every candidate's histogram again, every direction with the exact arithmetic
        for (int i = tid; i < nmain * MAD_MAX_Z; i += ORI_THREADS) (&hist[1][0])[i] = 0;
        __syncthreads();
        for (int e = tid; e < nvox * nmain; e += ORI_THREADS) {
            const int v = e / nmain, c = e - v * nmain;
            if (main_list[c] == 0) continue;
            const double *d = s_dom[c];
            const double g0 = vx[v], g1 = vy[v], g2 = vz[v];
            classify_exact64(&fast, trig, g0 * d[0] + g1 * d[1] + g2 * d[2], g0 * d[3] + g1 * d[4] + g2 * d[5], g0 * d[6] + g1 * d[7] + g2 * d[8],
                             [&](int zn) { tally(1 + c, zn, v); });
        }
    } else {
        const int nq = s_nq;      // ~0.1 % of nvox * nmain in practice
        for (int qi = tid; qi < nq; qi += ORI_THREADS) {
            const int v = queue[qi] & 0xffff, c = queue[qi] >> 16;
            const double *d = s_dom[c];
            const double g0 = vx[v], g1 = vy[v], g2 = vz[v];
            const double rx = g0 * d[0] + g1 * d[1] + g2 * d[2];
            const double ry = g0 * d[3] + g1 * d[4] + g2 * d[5];
            const double rz = g0 * d[6] + g1 * d[7] + g2 * d[8];
            classify_exact64(&fast, trig, rx, ry, rz, [&](int zn) { tally(1 + c, zn, v); });
        }
    }
    __syncthreads();
    ORI_STAMP(8);
    // quantise + step04 (Orientator.py:228-239) per candidate, one wave each
    for (int c = tid >> 6; c < nmain; c += ORI_THREADS / MAD_WAVE) {
        const int lane = lane_id();
        short *q1 = qz[1 + c];
        if (main_list[c] != 0) quantise_wave0(hist[1 + c], q1, Z);
        else { if (lane < Z) q1[lane] = qz[0][lane]; if (lane + 64 < Z) q1[lane + 64] = qz[0][lane + 64]; }
        const int i0 = lane, i1 = lane + 64;
        const bool in0 = i0 >= 1 && i0 < Z - 1, in1 = i1 >= 1 && i1 < Z - 1;
        const int c0 = in0 ? q1[i0] : 0, c1 = in1 ? q1[i1] : 0;
        const int mx = wave_max_i32(max(c0, c1));
        bool p0 = false, p1 = false;
        if (mx > 0) {
            p0 = in0 && (double)((int)((double)c0 / (double)mx * 50.0)) > 50 * 0.8;
            p1 = in1 && (double)((int)((double)c1 / (double)mx * 50.0)) > 50 * 0.8;
        }
        const unsigned long long m0 = __ballot(p0), m1 = __ballot(p1);
        const int n0 = __popcll(m0), ns = n0 + __popcll(m1);
        const bool ok = mx > 0 && ns <= A.lim_sec;
        if (ok) {
            if (p0) sec_list[c][__popcll(m0 & lanemask_lt())] = i0;
            if (p1) sec_list[c][n0 + __popcll(m1 & lanemask_lt())] = i1;
        }
        if (lane == 0) sec_cnt[c] = ok ? ns : -1;
    }
    __syncthreads();
    ORI_STAMP(9);
    // emit rows: candidates in ascending main bin, secondary bins ascending (Orientator.py:90-106)
    int produced = 0, hist_rows = 0;
    for (int c = 0; c < nmain; c++) {
        const int ns = sec_cnt[c];
        if (ns < 0) continue;
        if (tid < ns) {
            const size_t o = (size_t)a * A.fan + produced + tid;
            A.slot_main[o] = main_list[c];
            A.slot_sec[o] = sec_list[c][tid];
            A.slot_hidx[o] = hist_rows;
        }
        if (A.slot_hist)
            for (int i = tid; i < Z; i += ORI_THREADS) A.slot_hist[((size_t)a * A.lim_main + hist_rows) * Z + i] = qz[1 + c][i];
        produced += ns;
        hist_rows++;
    }
    if (tid == 0) A.slot_cnt[a] = produced;
    ORI_STAMP(10);
}

// per job: where the per-anchor slots of k_orient lie and where the job's rows go
struct RowsArgs {
    const int32_t *slot_cnt, *slot_main, *slot_sec, *slot_hist, *slot_hidx;
    int32_t *row_off;              // n + 1
    int n;
    int32_t *row_anchor, *row_main, *row_sec;
    double *row_R;
    int32_t *row_count;
    double *row_Rinv;
    int32_t *row_meta;
    const int32_t *anc_index, *anc_octave;
    int32_t *n_rows;               // device: the job's row count
    const int32_t *order;          // nullable: anchors in working order
    int32_t *perm_off;             // with order: n + 1 row offsets in working order
    int32_t *row_perm;             // with order: the rows in working order
    DscRowRec *row_rec;            // nullable: what k_describe starts a row from, in working order
    int32_t *anc_rows;             // nullable: per anchor in working order MAD_ANCROW_WORDS ints {position of its first row in working order, rows, voxel coordinates}
    const int32_t *coords;         // with row_rec: the anchors' voxel coordinates
    int uniform_octave;            // ... and their octave when anc_octave is null
};

// the record of one row (k_describe reads it with one 128-byte load); inv = inv(Rfinal), still in the thread's registers
__device__ __forceinline__ void put_row_rec(const RowsArgs &A, int64_t pos, int row, int a, const double *inv, const double *R) {
    DscRowRec &q = A.row_rec[pos];
    q.row = row;
    q.c[0] = A.coords[3 * a]; q.c[1] = A.coords[3 * a + 1]; q.c[2] = A.coords[3 * a + 2];
    q.octave = A.anc_octave ? A.anc_octave[a] : A.uniform_octave;
    for (int i = 0; i < 9; i++) q.inv[i] = inv[i];
    if (!A.anc_rows) return;      // (only k_describe_ball reads the rest)
    // the float32 values k_describe's threads form for themselves, once per row
    for (int i = 0; i < 9; i++) q.hf[i] = (float)inv[i];
    for (int i = 0; i < 9; i++) q.rf[i] = i < 6 ? (float)R[i] : (float)R[i] * (1.0f / 511.0f);
    for (int i = 0; i < 3; i++) q.ru[i] = (float)R[6 + i];
}

// row offsets of every job: exclusive scan of its anchors' row counts, one workgroup per job
__global__ __launch_bounds__(1024) void k_orient_scan(Batch<RowsArgs> B) {
    __shared__ int wt[1024 / MAD_WAVE + 1];
    __shared__ int carry;
    const RowsArgs &A = B.job[blockIdx.x];
    const int n = A.n;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = (i < n) ? A.slot_cnt[i] : 0;
        int tot;
        const int ex = block_excl_scan(v, wt, &tot);
        if (i < n) A.row_off[i] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        A.row_off[n] = carry;
        *A.n_rows = carry;
    }
    if (!A.order) return;
    __syncthreads();
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {      // the same in working order
        const int i = base + threadIdx.x;
        const int v = (i < n) ? A.slot_cnt[A.order[i]] : 0;
        int tot;
        const int ex = block_excl_scan(v, wt, &tot);
        if (i < n) A.perm_off[i] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry += tot;
        __syncthreads();
    }
}


// expand the per-anchor slots into the compact row list; Rfinal = adj_sec @ to_dom (Orientator.py:105)
__global__ __launch_bounds__(256) void k_orient_rows(Batch<RowsArgs> B, int fan, int lim_main, const EqspDev *eq) {
    const int job = batch_job(B, (int)blockIdx.x);
    const RowsArgs &A = B.job[job];
    const int64_t gid = (int64_t)((int)blockIdx.x - B.first[job]) * 256 + threadIdx.x;
    const int p = (int)(gid / fan), s = (int)(gid % fan);
    if (p >= A.n) return;
    const int a = A.order ? A.order[p] : p;
    const int c = A.slot_cnt[a];
    if (s == 0 && A.anc_rows) {
        int32_t *q = A.anc_rows + MAD_ANCROW_WORDS * (int64_t)p;
        q[0] = A.order ? A.perm_off[p] : A.row_off[a]; q[1] = c; q[2] = A.coords[3 * a]; q[3] = A.coords[3 * a + 1]; q[4] = A.coords[3 * a + 2];
    }
    if (s >= c) return;
    const int64_t row = (int64_t)A.row_off[a] + s;
    if (A.order) A.row_perm[A.perm_off[p] + s] = (int32_t)row;
    const int mb = A.slot_main[(size_t)a * fan + s], sb = A.slot_sec[(size_t)a * fan + s];
    A.row_anchor[row] = a;
    A.row_main[row] = mb;
    A.row_sec[row] = sb;
    double R9[9], inv9[9];      // in registers: the inverse is formed from what was computed, not from what was just stored
    mad_rfinal(eq, mb, sb, R9);
    for (int i = 0; i < 9; i++) A.row_R[9 * row + i] = R9[i];
    if (A.row_count) {
        const int Z = eq->Z;
        const int32_t *h = A.slot_hist + ((size_t)a * lim_main + A.slot_hidx[(size_t)a * fan + s]) * Z;
        for (int i = 0; i < Z; i++) A.row_count[row * Z + i] = h[i];
    }
    if (A.row_Rinv || A.row_rec) mad_mat3_inv(R9, inv9);      // inv(lo.Rfinal) of MaD.py:438, once per row
    if (A.row_Rinv)
        for (int i = 0; i < 9; i++) A.row_Rinv[9 * row + i] = inv9[i];
    if (A.row_meta) { A.row_meta[3 * row] = A.anc_index[a]; A.row_meta[3 * row + 1] = A.anc_octave[a]; A.row_meta[3 * row + 2] = mb; }
    if (A.row_rec) put_row_rec(A, A.order ? A.perm_off[p] + s : row, (int)row, a, inv9, R9);
}

// k_orient_scan + k_orient_rows in ONE launch (round 3): every workgroup forms the exclusive scan of its job's
// per-anchor row counts for itself, in LDS (n ints: a thread sums a run of consecutive anchors, one block scan over the runs), the
// working-order offset of its first anchor by a block reduction, and then expands its ORS_THREADS / fan anchors.  A few microseconds of
// redundant work per workgroup, all of them side by side, instead of a one-workgroup launch on the build's critical path.
// Dynamic LDS: (max n + 1) ints.  The rows come out exactly as from the two-launch form.
#define ORI_ROWS_MAX_N 30000      // anchors per job the one-launch form takes (LDS); beyond: k_orient_scan + k_orient_rows
// (256 threads since round 4: a 1 024-thread workgroup needs half a CU's wave slots at once, and beside the k_describe workgroups of
// other lanes -- 24 of a CU's 32 waves -- it waited for them: 42 us per launch in the overlapped run against 10 alone)
#define ORS_THREADS 256
__global__ __launch_bounds__(ORS_THREADS) void k_orient_rows_scan(Batch<RowsArgs> B, int fan, int lim_main, const EqspDev *eq) {
    extern __shared__ __align__(16) int s_off[];
    __shared__ int wt[ORS_THREADS / MAD_WAVE + 1];
    __shared__ int s_perm[ORS_THREADS];
    const int job = batch_job(B, (int)blockIdx.x);
    const RowsArgs &A = B.job[job];
    const int n = A.n, tid = (int)threadIdx.x;
    const int per = (n + ORS_THREADS - 1) / ORS_THREADS;
    const int i0 = min(tid * per, n), i1 = min(i0 + per, n);
    int sum = 0;
    for (int i = i0; i < i1; i++) sum += A.slot_cnt[i];
    int total;
    int run = block_excl_scan(sum, wt, &total);
    for (int i = i0; i < i1; i++) { s_off[i] = run; run += A.slot_cnt[i]; }
    const int ppb = ORS_THREADS / fan;                            // anchors per workgroup
    const int p0 = ((int)blockIdx.x - B.first[job]) * ppb;        // its first position in working order
    if (p0 == 0 && tid == 0) *A.n_rows = total;
    // rows of the anchors that precede position p0 in working order, then of this workgroup's own anchors
    int before = 0;
    for (int q = tid; q < p0; q += ORS_THREADS) before += A.slot_cnt[A.order ? A.order[q] : q];
    int base;
    (void)block_excl_scan(before, wt, &base);
    const int mine = (tid < ppb && p0 + tid < n) ? A.slot_cnt[A.order ? A.order[p0 + tid] : p0 + tid] : 0;
    int tot2;
    const int ex = block_excl_scan(mine, wt, &tot2);
    s_perm[tid] = base + ex;
    __syncthreads();
    const int pl = tid / fan, sl = tid % fan, pp = p0 + pl;
    if (pl >= ppb || pp >= n) return;
    const int a = A.order ? A.order[pp] : pp;
    const int c = A.slot_cnt[a];
    if (sl == 0 && A.anc_rows) {
        int32_t *q = A.anc_rows + MAD_ANCROW_WORDS * (int64_t)pp;
        q[0] = s_perm[pl]; q[1] = c; q[2] = A.coords[3 * a]; q[3] = A.coords[3 * a + 1]; q[4] = A.coords[3 * a + 2];
    }
    if (sl >= c) return;
    const int64_t row = (int64_t)s_off[a] + sl;
    if (A.order) A.row_perm[s_perm[pl] + sl] = (int32_t)row;
    const int mb = A.slot_main[(size_t)a * fan + sl], sb = A.slot_sec[(size_t)a * fan + sl];
    A.row_anchor[row] = a;
    A.row_main[row] = mb;
    A.row_sec[row] = sb;
    double R9[9], inv9[9];      // in registers: the inverse is formed from what was computed, not from what was just stored
    mad_rfinal(eq, mb, sb, R9);
    for (int i = 0; i < 9; i++) A.row_R[9 * row + i] = R9[i];
    if (A.row_count) {
        const int Z = eq->Z;
        const int32_t *h = A.slot_hist + ((size_t)a * lim_main + A.slot_hidx[(size_t)a * fan + sl]) * Z;
        for (int i = 0; i < Z; i++) A.row_count[row * Z + i] = h[i];
    }
    if (A.row_Rinv || A.row_rec) mad_mat3_inv(R9, inv9);      // inv(lo.Rfinal) of MaD.py:438, once per row
    if (A.row_Rinv)
        for (int i = 0; i < 9; i++) A.row_Rinv[9 * row + i] = inv9[i];
    if (A.row_meta) { A.row_meta[3 * row] = A.anc_index[a]; A.row_meta[3 * row + 1] = A.anc_octave[a]; A.row_meta[3 * row + 2] = mb; }
    if (A.row_rec) put_row_rec(A, A.order ? s_perm[pl] + sl : row, (int)row, a, inv9, R9);
}

// Runs a1-a8 for the anchor lists of n_jobs structures (coordinates and octaves already on the device) in one k_orient
// grid, one scan and one row-expansion launch, and writes each job's rows to its `out` (capacity n * lim_main * lim_sec rows).
// Asynchronous: the row counts stay on the device.
static int orient_batch(mad_ctx *ctx, int n_jobs, const OrientJob *jobs, int r, int lim_main, int lim_sec) {
    const int Z = ctx->eq_host[0].Z;
    const int fan = lim_main * lim_sec;
    int64_t total = 0;
    bool want_hist = false;
    for (int j = 0; j < n_jobs; j++) { total += jobs[j].n; want_hist |= jobs[j].out.row_count != nullptr; }
    if (total > (int64_t)INT32_MAX / (fan * 4)) return mad_fail(ctx, MAD_EINVAL, "mad_orient: %lld anchors in one batch", (long long)total);
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_SLOT_CNT), (size_t)(total + 4) * 4));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_SLOT_MAIN), (size_t)total * fan * 4));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_SLOT_SEC), (size_t)total * fan * 4));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_TMP_A), (size_t)total * fan * 4));
    if (want_hist) MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_SLOT_HIST), (size_t)total * lim_main * Z * 4));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_ROW_OFF), (size_t)(total + n_jobs + 2) * 4));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_PERM_OFF), (size_t)(total + n_jobs + 2) * 4));

    Batch<OrientArgs> B;
    Batch<RowsArgs> R;
    B.n_jobs = R.n_jobs = n_jobs;
    int64_t a0 = 0, blk = 0;
    int max_n = 0;
    for (int j = 0; j < n_jobs; j++) max_n = std::max(max_n, jobs[j].n);
    const bool one_launch = max_n <= ORI_ROWS_MAX_N && fan <= ORS_THREADS;      // (beyond: a scan of its own and the row expansion, two launches)
    for (int j = 0; j < n_jobs; j++) {
        const OrientJob &J = jobs[j];
        OrientArgs &A = B.job[j];
        A.f[0] = J.f[0]; A.f[1] = J.f[1];
        A.coords = J.d_coords; A.octave = J.d_octave; A.uniform_octave = J.uniform_octave;
        A.order = J.out.row_perm ? J.out.anc_order : nullptr;
        A.n = J.n; A.r = r; A.nmask = ctx->mask_n; A.mask_off = ctx->mask_off; A.eq = ctx->eq[0];
        A.lim_main = lim_main; A.lim_sec = lim_sec; A.fan = fan;
        A.slot_cnt = scratch<int32_t>(ctx, S_SLOT_CNT) + a0;
        A.n_reject = J.out.d_n_reject;
        A.slot_main = scratch<int32_t>(ctx, S_SLOT_MAIN) + a0 * fan;
        A.slot_sec = scratch<int32_t>(ctx, S_SLOT_SEC) + a0 * fan;
        A.slot_hist = J.out.row_count ? scratch<int32_t>(ctx, S_SLOT_HIST) + a0 * lim_main * Z : nullptr;
        A.slot_hidx = scratch<int32_t>(ctx, S_TMP_A) + a0 * fan;
        A.wfix = ctx->gw_sig != 0.0 ? ctx->gw_tab : nullptr;
        A.queue_cap = std::min(std::max(ctx->ori_queue_cap, 0), ORI_QUEUE);
        B.first[j] = (int)a0;
        RowsArgs &Q = R.job[j];
        Q.slot_cnt = A.slot_cnt; Q.slot_main = A.slot_main; Q.slot_sec = A.slot_sec; Q.slot_hist = A.slot_hist; Q.slot_hidx = A.slot_hidx;
        Q.row_off = scratch<int32_t>(ctx, S_ROW_OFF) + a0 + j;
        Q.n = J.n;
        Q.row_anchor = J.out.row_anchor; Q.row_main = J.out.row_main; Q.row_sec = J.out.row_sec; Q.row_R = J.out.row_R;
        Q.row_count = J.out.row_count; Q.row_Rinv = J.out.row_Rinv; Q.row_meta = J.out.row_meta;
        Q.anc_index = J.out.anc_index; Q.anc_octave = J.out.anc_octave;
        Q.n_rows = J.out.d_n_rows;
        Q.order = A.order; Q.perm_off = scratch<int32_t>(ctx, S_PERM_OFF) + a0 + j; Q.row_perm = J.out.row_perm;
        Q.row_rec = J.out.row_rec; Q.coords = J.d_coords; Q.uniform_octave = J.uniform_octave;
        Q.anc_rows = J.out.anc_rows;
        if (!Q.anc_octave) Q.anc_octave = J.d_octave;
        R.first[j] = (int)blk;
        a0 += J.n;
        blk += one_launch ? mad_ceil_div((int64_t)J.n, ORS_THREADS / fan) : mad_ceil_div((int64_t)J.n * fan, 256);
        if (J.out.d_n_reject && !J.out.counters_zeroed) MAD_HIP(hipMemsetAsync(J.out.d_n_reject, 0, 4, ctx->stream));
    }
    B.first[n_jobs] = (int)a0;
    R.first[n_jobs] = (int)blk;
    mad_timer_begin(ctx, MAD_T_ORIENT);
    // unit gradients (SoA) + the undecided-voxel queue; with a window also the voxels' squared offsets and the weight table
    const bool gw = ctx->gw_sig != 0.0;
    const size_t lds = gw ? ((((size_t)ctx->mask_n * 14 + ORI_QUEUE * 4 + 15) & ~(size_t)15) + (size_t)(3 * r * r + 1) * 8)
                          : (size_t)ctx->mask_n * 3 * sizeof(float) + ORI_QUEUE * 4;
    if (lds > 48 * 1024) {      // boxes beyond r = 8 (Orientator(ori_radius > 16)): more dynamic LDS than a kernel gets by default
        static bool attr = false;
        if (!attr) {
            MAD_HIP(hipFuncSetAttribute((const void *)k_orient<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
            MAD_HIP(hipFuncSetAttribute((const void *)k_orient<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
            attr = true;
        }
        if (lds > 128 * 1024) return mad_fail(ctx, MAD_EINVAL, "mad_orient: a box of side %d needs %zu bytes of LDS", r, lds);
    }
    if (gw) hipLaunchKernelGGL(k_orient<true>, dim3((unsigned)a0), dim3(ORI_THREADS), lds, ctx->stream, B);
    else hipLaunchKernelGGL(k_orient<false>, dim3((unsigned)a0), dim3(ORI_THREADS), lds, ctx->stream, B);
    mad_timer_end(ctx, MAD_T_ORIENT);
    if (one_launch) {
        const size_t lds_r = (size_t)(max_n + 4) * 4;
        if (lds_r > 48 * 1024) {
            static bool attr_r = false;
            if (!attr_r) { MAD_HIP(hipFuncSetAttribute((const void *)k_orient_rows_scan, hipFuncAttributeMaxDynamicSharedMemorySize, (ORI_ROWS_MAX_N + 4) * 4)); attr_r = true; }
        }
        hipLaunchKernelGGL(k_orient_rows_scan, dim3((unsigned)blk), dim3(ORS_THREADS), lds_r, ctx->stream, R, fan, lim_main, ctx->eq[0]);
    } else {
        hipLaunchKernelGGL(k_orient_scan, dim3(n_jobs), dim3(1024), 0, ctx->stream, R);
        hipLaunchKernelGGL(k_orient_rows, dim3((unsigned)blk), dim3(256), 0, ctx->stream, R, fan, lim_main, ctx->eq[0]);
    }
    MAD_HIP(hipGetLastError());
    return MAD_OK;
}

int mad_orient_device_many(mad_ctx *ctx, int n_jobs, const OrientJob *jobs, int r, int lim_main, int lim_sec) {
    if (!ctx->eq_set[0]) return mad_fail(ctx, MAD_EINVAL, "mad_orient: orientation EQSP table not set");
    if (r < 1 || r > 10) return mad_fail(ctx, MAD_EINVAL, "mad_orient: box_side %d outside 1..10", r);
    if (lim_main < 1 || lim_main > ORI_MAX_MAIN || lim_sec < 1 || lim_main * lim_sec > ORI_MAX_FAN)
        return mad_fail(ctx, MAD_EINVAL, "mad_orient: lim_main=%d lim_sec=%d unsupported", lim_main, lim_sec);
    MAD_TRY(ensure_mask(ctx, r));
    if (ctx->gw_sig != 0.0 && (ctx->gw_r != r || ctx->gw_built != ctx->gw_sig)) {
        // Orientator(gw_sig): the window's weights for this box size, as 2^-50 fixed point
        unsigned long long tab[3 * 10 * 10 + 1];
        for (int d2 = 0; d2 <= 3 * r * r; d2++)
            tab[d2] = (unsigned long long)llround(ldexp(exp(-1.0 * ((double)d2 / (2.0 * (ctx->gw_sig * ctx->gw_sig)))), ORI_WFIX_BITS));
        if (!ctx->gw_tab && hipMalloc((void **)&ctx->gw_tab, sizeof(tab)) != hipSuccess) return mad_fail(ctx, MAD_ENOMEM, "orientation window table");
        MAD_HIP(hipMemcpy(ctx->gw_tab, tab, sizeof(unsigned long long) * (3 * r * r + 1), hipMemcpyHostToDevice));
        ctx->gw_r = r; ctx->gw_built = ctx->gw_sig;
    }
    // jobs without anchors only have their counters reset; the others go out in batches of MAD_BATCH_MAX
    OrientJob live[MAD_BATCH_MAX];
    int n_live = 0;
    for (int j = 0; j < n_jobs; j++) {
        if (jobs[j].n <= 0) {
            if (!jobs[j].out.counters_zeroed) {
                MAD_HIP(hipMemsetAsync(jobs[j].out.d_n_rows, 0, 4, ctx->stream));
                if (jobs[j].out.d_n_reject) MAD_HIP(hipMemsetAsync(jobs[j].out.d_n_reject, 0, 4, ctx->stream));
            }
            continue;
        }
        live[n_live++] = jobs[j];
        if (n_live == MAD_BATCH_MAX) { MAD_TRY(orient_batch(ctx, n_live, live, r, lim_main, lim_sec)); n_live = 0; }
    }
    if (n_live) MAD_TRY(orient_batch(ctx, n_live, live, r, lim_main, lim_sec));
    return MAD_OK;
}

int mad_orient_device(mad_ctx *ctx, FieldDev f0, FieldDev f1, const int32_t *d_coords, const int32_t *d_octave,
                      int uniform_octave, int n, int r, int lim_main, int lim_sec, OrientOut out) {
    OrientJob J;
    J.f[0] = f0; J.f[1] = f1; J.d_coords = d_coords; J.d_octave = d_octave; J.uniform_octave = uniform_octave; J.n = n; J.out = out;
    return mad_orient_device_many(ctx, 1, &J, r, lim_main, lim_sec);
}

extern "C" int mad_set_orient_window(mad_ctx *ctx, double gw_sig) {
    if (!ctx) return MAD_EINVAL;
    if (!(gw_sig >= 0.0) || !(gw_sig < 1e6)) return mad_fail(ctx, MAD_EINVAL, "mad_set_orient_window: gw_sig = %g", gw_sig);
    MAD_TRY(mad_synchronize(ctx));      // the table may be in use
    ctx->gw_sig = gw_sig;
    return MAD_OK;
}

extern "C" int mad_orient(mad_ctx *ctx, int slot, int octave, const int32_t *coords, int n, int r, int lim_main,
                          int lim_sec, int32_t *row_anchor, int32_t *row_main, int32_t *row_sec, double *row_R,
                          int32_t *row_count, int64_t *n_rows, int64_t cap, int32_t *n_reject) {
    if (ctx) mad_use_lane(ctx, 0);
    if (!ctx || !n_rows) return MAD_EINVAL;
    if (slot < 0 || slot >= MAD_MAX_FIELDS || !ctx->fields[slot].tex)
        return mad_fail(ctx, MAD_EINVAL, "mad_orient: field slot %d is empty", slot);
    if (octave != 0 && octave != 1) return mad_fail(ctx, MAD_EINVAL, "mad_orient: octave %d", octave);
    if (n > 0 && !coords) return mad_fail(ctx, MAD_EINVAL, "mad_orient: coords is NULL");
    *n_rows = 0;
    if (n <= 0) { if (n_reject) *n_reject = 0; return MAD_OK; }
    if (lim_main < 1 || lim_sec < 1 || lim_main * lim_sec > ORI_MAX_FAN) return mad_fail(ctx, MAD_EINVAL, "mad_orient: lim_main=%d lim_sec=%d", lim_main, lim_sec);
    const int Z = ctx->eq_host[0].Z;
    const int64_t rows_cap = (int64_t)n * lim_main * lim_sec;
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_COORDS), (size_t)n * 12));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_ROW_ANCHOR), (size_t)rows_cap * 4));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_ROW_MAIN), (size_t)rows_cap * 4));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_ROW_SEC), (size_t)rows_cap * 4));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_ROW_R), (size_t)rows_cap * 72));
    if (row_count) MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_ROW_COUNT), (size_t)rows_cap * Z * 4));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_MISC), 256));
    MAD_HIP(hipMemcpyAsync(mad_sb(ctx, S_COORDS).p, coords, (size_t)n * 12, hipMemcpyHostToDevice, ctx->stream));
    OrientOut out;
    out.row_anchor = scratch<int32_t>(ctx, S_ROW_ANCHOR); out.row_main = scratch<int32_t>(ctx, S_ROW_MAIN);
    out.row_sec = scratch<int32_t>(ctx, S_ROW_SEC); out.row_R = scratch<double>(ctx, S_ROW_R);
    out.row_count = row_count ? scratch<int32_t>(ctx, S_ROW_COUNT) : nullptr;
    out.d_n_rows = scratch<int32_t>(ctx, S_MISC) + 16; out.d_n_reject = scratch<int32_t>(ctx, S_MISC) + 17;
    FieldDev f = ctx->fields[slot];
    MAD_TRY(mad_orient_device(ctx, f, f, scratch<int32_t>(ctx, S_COORDS), nullptr, octave, n, r, lim_main, lim_sec, out));
    MAD_HIP(hipMemcpyAsync(&ctx->pinned[0], out.d_n_rows, 8, hipMemcpyDeviceToHost, ctx->stream));
    MAD_HIP(hipStreamSynchronize(ctx->stream));
    const int32_t *hv = (const int32_t *)&ctx->pinned[0];
    const int64_t rows = hv[0];
    if (n_reject) *n_reject = hv[1];
    *n_rows = rows;
    if (rows > cap) return mad_fail(ctx, MAD_ENOSPC, "mad_orient: %lld rows, capacity %lld", (long long)rows, (long long)cap);
    if (rows == 0) return MAD_OK;
    if (row_anchor) MAD_HIP(hipMemcpyAsync(row_anchor, out.row_anchor, rows * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (row_main) MAD_HIP(hipMemcpyAsync(row_main, out.row_main, rows * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (row_sec) MAD_HIP(hipMemcpyAsync(row_sec, out.row_sec, rows * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (row_R) MAD_HIP(hipMemcpyAsync(row_R, out.row_R, rows * 72, hipMemcpyDeviceToHost, ctx->stream));
    if (row_count) MAD_HIP(hipMemcpyAsync(row_count, out.row_count, rows * Z * 4, hipMemcpyDeviceToHost, ctx->stream));
    MAD_HIP(hipStreamSynchronize(ctx->stream));
    return MAD_OK;
}

// ---------------------------------------------------------------------------
// descriptor kernel
// ---------------------------------------------------------------------------

#define DSC_THREADS 256
#define DSC_QUEUE 256

struct DescribeArgs {
    FieldDev f[2];
    const int32_t *anc_coords;     // per anchor (or per row when row_anchor == nullptr)
    const int32_t *anc_octave;     // per anchor, or nullptr -> uniform_octave
    int uniform_octave;
    const int32_t *row_anchor;     // row -> anchor, or nullptr (identity)
    const double *row_R;           // n_rows x 9
    const double *row_Rinv;        // n_rows x 9: inv(Rfinal) by cofactors (mad_mat3_inv), or nullptr -> formed here
    const int32_t *row_perm;       // nullable: the k-th workgroup takes row row_perm[k] (rows of neighbouring anchors side by side)
    const DscRowRec *row_rec;      // nullable: the k-th record = everything the k-th row in working order starts from
    int queue_cap;                 // entries of the undecided-sample queue in use (<= its size)
    const int32_t *n_rows;         // device: number of rows
    const int32_t *row_limit;      // device, nullable: this launch takes the first *row_limit rows in working order only (the rest: k_describe_ball)
    int32_t *overflow;             // device: set when the launch was sized for fewer rows than *n_rows
    int r;
    const EqspDev *eq;
    int16_t *dsc;                  // n_rows x 64*Z
    int8_t *dsc8;                  // nullable: the same rows as int8, zero-padded to a multiple of 128 rows (GEMM operand)
    double *norm;                  // with dsc8: |row|_2 (MaD.py:416)
};

// The reference's arithmetic for one sample (Descriptor.py:153-187): float32 normalisation, float64
// rotation by Rfinal, atan2 / arccos against the table; default zone 0, the last matching zone wins.
__device__ __forceinline__ int describe_exact(const EqspFastLds *eq, float4 t, const double *R) {
    float gx = t.x, gy = t.y, gz = t.z;
    if (t.w > 1e-12f) { gx = __fdiv_rn(gx, t.w); gy = __fdiv_rn(gy, t.w); gz = __fdiv_rn(gz, t.w); }
    const double g0 = gx, g1 = gy, g2 = gz;
    const double rx = g0 * R[0] + g1 * R[1] + g2 * R[2];
    const double ry = g0 * R[3] + g1 * R[4] + g2 * R[5];
    const double rz = g0 * R[6] + g1 * R[7] + g2 * R[8];
    int zone = 0;
    classify_exact64(eq, eq, rx, ry, rz, [&](int zn) { zone = zn; });
    return zone;
}

// Nearest voxel of lattice point l (in the anchor's frame) with the reference's float64 expression; false when the
// point leaves the grid (scipy's RegularGridInterpolator, bounds_error=True, MapSpace.py:189).
__device__ __forceinline__ bool lattice_voxel_exact(double l0, double l1, double l2, const double *inv, double c0, double c1, double c2,
                                                    const FieldDev &F, int *v0, int *v1, int *v2) {
    const double p0 = (l0 * inv[0] + l1 * inv[1] + l2 * inv[2]) + c0;      // Descriptor.py:132-133
    const double p1 = (l0 * inv[3] + l1 * inv[4] + l2 * inv[5]) + c1;
    const double p2 = (l0 * inv[6] + l1 * inv[7] + l2 * inv[8]) + c2;
    if (!(p0 >= 0.0) || !(p0 <= (double)(F.nx - 1)) || !(p1 >= 0.0) || !(p1 <= (double)(F.ny - 1)) || !(p2 >= 0.0) ||
        !(p2 <= (double)(F.nz - 1)))
        return false;
    int a0i = min((int)floor(p0), F.nx - 2), a1i = min((int)floor(p1), F.ny - 2), a2i = min((int)floor(p2), F.nz - 2);
    *v0 = (p0 - (double)a0i <= 0.5) ? a0i : a0i + 1;
    *v1 = (p1 - (double)a1i <= 0.5) ? a1i : a1i + 1;
    *v2 = (p2 - (double)a2i <= 0.5) ? a2i : a2i + 1;
    return true;
}
// ... as a texel index; *oob is set when the point leaves the grid
__device__ __forceinline__ unsigned lattice_index_exact(double l0, double l1, double l2, const double *inv, double c0, double c1, double c2,
                                                        const FieldDev &F, bool *oob) {
    int a0i, a1i, a2i;
    if (!lattice_voxel_exact(l0, l1, l2, inv, c0, c1, c2, F, &a0i, &a1i, &a2i)) {
        *oob = true;
        return 0;
    }
    return (unsigned)(((size_t)a0i * F.ny + a1i) * F.nz + a2i);
}

// Descriptor.py:44-93: the sub-region of lattice point (i, j, k) among NSUB, split into the part a thread's (j, k) column fixes
// and the part its sample index i adds.  Order of the reference's sub_slices lists: 64 and 27 -- third axis fastest, then the
// first, then the second; 8 -- its own order ((s1,s1,s2) first); 1 -- the whole cube.
template <int S, int NSUB> __device__ __forceinline__ int sub_blk(int x) {
    if (NSUB == 64) return x / (S / 4);
    if (NSUB == 27) return x < S / 3 ? 0 : (x < 2 * S / 3 ? 1 : 2);
    if (NSUB == 8) return x < S / 2 ? 0 : 1;
    return 0;
}
template <int S, int NSUB> __device__ __forceinline__ int sub_of_jk(int j, int k) {
    if (NSUB == 64) return sub_blk<S, NSUB>(j) * 16 + sub_blk<S, NSUB>(k);
    if (NSUB == 27) return sub_blk<S, NSUB>(j) * 9 + sub_blk<S, NSUB>(k);
    if (NSUB == 8) return sub_blk<S, NSUB>(j) * 2 + (1 - sub_blk<S, NSUB>(k));
    return 0;
}
template <int S, int NSUB> __device__ __forceinline__ int sub_of_i(int i) {
    if (NSUB == 64) return sub_blk<S, NSUB>(i) * 4;
    if (NSUB == 27) return sub_blk<S, NSUB>(i) * 3;
    if (NSUB == 8) return sub_blk<S, NSUB>(i) * 4;
    return 0;
}

#define DSC_CHUNK 8
#ifndef DSC_PASSES
#define DSC_PASSES 1      // texel passes per thread at S = 16 (probe builds: -DDSC_PASSES=2 -DDSC_OCC=5)
#endif
#ifndef DSC_OCC
#define DSC_OCC 4         // workgroups per CU the register budget is set for
#endif
// ZMAX: zones of the descriptor sphere the histogram has room for -- 16 (Descriptor(subeqsp_size=16), what MaD.run uses) or 128
// (subeqsp_size=112, the reference's other table, eqsp.py:16: rows of 64 x 112 = 7 168 counts)
// TAB (round 3): the samples are gathered from the field's 4-byte texels (a unit direction in 3 x 10 bits, FieldDev::tex4) and
// classified through the conservative tables of EqspTabLds -- a quarter of the bytes per gather, ~25 vector instructions per
// sample instead of ~65 for normalisation + rotation + eqsp_fast32, a quarter of the registers for the texels in flight.  What the
// table cannot decide (the bins a zone edge crosses +- MAD_TAB_GUARD, 2-4 % of the samples) is queued BY TEXEL INDEX and goes
// through the former tiers -- float32 with its 1e-4 guard, then float64 -- on the full 16-byte texel, in full lanes.  The result
// is the same descriptor, bit for bit: every tier only ever answers when the exact arithmetic is certain to agree.
#define DSC_QUEUE_TAB 768
#ifdef MAD_PROBE_STAMPS      // diagnostic build: s_memtime at the phases of every row's workgroup (tools/probe_describe.py)
__device__ long long dsc_stamps[16384 * 8];
#define DSC_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0 && blockIdx.x < 16384) dsc_stamps[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
extern "C" int mad_debug_dsc_stamps(long long *out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(dsc_stamps), (size_t)n * 8) == hipSuccess ? 0 : -1;
}
#else
#define DSC_STAMP(k) do { } while (0)
#endif
#ifndef DSC_OCC_TAB
#define DSC_OCC_TAB 4     // ... of the TAB form (5: 96 registers, 32 of them spilled, 88 -> 102 us per launch)
#endif
template <int S, int NSUB = 64, int ZMAX = 16, bool TAB = false>
__global__ __launch_bounds__(DSC_THREADS, TAB ? DSC_OCC_TAB : DSC_OCC) void k_describe(Batch<DescribeArgs> B) {
    const int job = batch_job(B, (int)blockIdx.x);
    const DescribeArgs &A = B.job[job];
    const int bid = (int)blockIdx.x - B.first[job], gdim = B.first[job + 1] - B.first[job];      // this job's part of the grid (multiples of 8)
    __shared__ int hist[NSUB * ZMAX];
    __shared__ int s_oob, s_nq;
    __shared__ int s_part[DSC_THREADS / MAD_WAVE];
    __shared__ __align__(16) int s_rec[32];      // the row's DscRowRec
    // TAB: the float32 / float64 tiers see 3-4 % of the samples, at the end, in full lanes: they read their tables from global
    // memory (cache-resident, shared by every workgroup) and the 12 KB image is not staged per row
    // (TAB: only what eqsp_fast32 reads -- the head of the image: g32, zlut, belt_f, belt_i -- is staged; the float64 tier, a few
    // samples per row, reads its tables from global memory)
    __shared__ typename std::conditional<TAB, uint4[(offsetof(EqspFastLds, dir) + 15) / 16], EqspFastLds>::type fast_s;
    const EqspFastLds *const fastp = (const EqspFastLds *)&fast_s;                             // float32 tier
    const EqspFastLds *const exactp = TAB ? &A.eq->image : (const EqspFastLds *)&fast_s;       // float64 tier
    __shared__ float4 qv[TAB ? 1 : DSC_QUEUE];         // texels the fast classifier could not decide
    __shared__ int qsub[TAB ? 1 : DSC_QUEUE];
    __shared__ typename std::conditional<TAB, EqspTabLds, int>::type tab;
    __shared__ unsigned qidx[TAB ? DSC_QUEUE_TAB : 1];             // TAB: texel indices of the samples the table could not decide
    __shared__ unsigned short qsub16[TAB ? DSC_QUEUE_TAB : 1];
    const int QCAP = min(TAB ? DSC_QUEUE_TAB : DSC_QUEUE, A.queue_cap);      // (mad_set_option "dsc_queue": tests of the full-queue path)
    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs (b and b + 8 share one), so give
    // each XCD a contiguous run of rows.  Consecutive rows belong to the same anchor (fan-out ~5) or to
    // neighbours in the anchor list and sample the same neighbourhood: running side by side on ONE XCD
    // they meet in its L2.  The row count lives on the device; the grid is an upper bound of it.
    const int64_t n_rows = *A.n_rows;
    const int64_t n_work = A.row_limit ? (int64_t)*A.row_limit : n_rows;
    const int64_t chunk = (n_work + 7) / 8;
    const int tid = threadIdx.x;
    if (8 * chunk > (int64_t)gdim) {      // the launch was sized from a stale hint: tell the host
        if (bid == 0 && tid == 0) *A.overflow = 1;
        return;
    }
    if (A.dsc8) {      // zero rows up to the next multiple of 128: the GEMM reads whole tiles
        const int64_t n_pad = (n_rows + 127) / 128 * 128;
        const int Dp = NSUB * A.eq->Z;
        for (int64_t r = n_rows + bid; r < n_pad; r += gdim) {
            for (int i = tid; i < Dp / 4; i += DSC_THREADS) ((int32_t *)(A.dsc8 + r * Dp))[i] = 0;
            if (tid == 0) A.norm[r] = 0.0;
        }
    }
    const int64_t work = (int64_t)(bid & 7) * chunk + (bid >> 3);
    if ((int64_t)(bid >> 3) >= chunk || work >= n_work) return;
    DSC_STAMP(0);
    if (TAB) { stage_lds(&tab, &A.eq->tab, sizeof(EqspTabLds)); stage_lds(&fast_s, &A.eq->image, sizeof(fast_s)); }
    else eqsp_fast_stage(A.eq, (EqspFastLds *)&fast_s);
    const FieldDev F0 = A.f[0], F1 = A.f[1];
    const int Z = A.eq->Z;
    const int D = NSUB * Z;            // S = 2 r samples per axis (16), NSUB sub-regions of Z zones each
    for (int i = tid; i < D; i += DSC_THREADS) hist[i] = 0;
    // What a row starts from -- its index, its anchor's coordinates and octave, inv(Rfinal) -- is one 128-byte record (DscRowRec,
    // written with the rows by k_orient_rows* in WORKING order) that 32 lanes fetch with one vector load and park in LDS, next
    // to the staging of the tables.  Before (round 2) it was a chain of dependent scalar loads, row_perm -> row_anchor -> octave,
    // coordinates, inv(R): with the staging 7 700 of a row's 35 600 cycles (MAD_PROBE_STAMPS).  Without records (mad_describe
    // on rows of the caller) thread 0 walks that chain and forms the record itself.
    // (Round 3 also built this kernel as a loop, twice -- workgroups that fill the chip walk the rows, statically dealt or by
    // tickets from a counter per XCD, tables staged once, the next row's record fetched a row ahead.  Fewer instructions (-13 %)
    // and MORE time: with the registers of this form (80, six workgroups per CU) the same binary took 0.46 ms per C3 step run one
    // row per workgroup and 0.58 looping; its waves spend 60 % of their cycles parked at waits and barriers against 43 %
    // (SQ_WAIT_ANY), L2 misses +26 %.  Neither staggered starts, nor an LDS-only barrier at the top of a row, nor the ticket behind
    // that barrier changed it.  Not understood; dropped.)
    const int32_t *const p_perm = A.row_perm, *const p_anchor = A.row_anchor, *const p_coords = A.anc_coords, *const p_octave = A.anc_octave;
    const double *const p_R = A.row_R, *const p_Rinv = A.row_Rinv;
    const int uni_octave = A.uniform_octave;
    auto chain_record = [=](int64_t w, int *rec) {      // thread 0 only
        const int r = p_perm ? p_perm[w] : (int)w;
        const int an = p_anchor ? p_anchor[r] : r;
        rec[0] = r;
        rec[1] = p_coords[3 * an]; rec[2] = p_coords[3 * an + 1]; rec[3] = p_coords[3 * an + 2];
        rec[4] = p_octave ? p_octave[an] : uni_octave;
        double *inv = (double *)(rec + 8);
        // Rfinal's inverse (np.linalg.inv, Descriptor.py:132; cofactors)
        if (p_Rinv) {
            for (int i = 0; i < 9; i++) inv[i] = p_Rinv[9 * (int64_t)r + i];
        } else {
            double m[9];
            mad_mat3_inv(p_R + 9 * (int64_t)r, m);
            for (int i = 0; i < 9; i++) inv[i] = m[i];
        }
    };
    const int *const recs = (const int *)A.row_rec;
    if (recs) {
        if (tid < 32) s_rec[tid] = recs[work * MAD_ROWREC_WORDS + tid];
    } else if (tid == 0) chain_record(work, s_rec);
    if (tid == 0) { s_oob = 0; s_nq = 0; }
    __syncthreads();      // the histogram is zero, the flags are reset, the record is there
    const int64_t row = __builtin_amdgcn_readfirstlane(s_rec[0]);
    const int oct = __builtin_amdgcn_readfirstlane(s_rec[4]);
    const double *const sInv = (const double *)&s_rec[8];
    FieldDev F;
    F.tex = oct == 1 ? F1.tex : F0.tex; F.nx = oct == 1 ? F1.nx : F0.nx; F.ny = oct == 1 ? F1.ny : F0.ny; F.nz = oct == 1 ? F1.nz : F0.nz;
    F.tex4 = oct == 1 ? F1.tex4 : F0.tex4;
#ifdef DSC_TEX_AUX      // probe builds (tools/build_variant.sh -DDSC_TEX_AUX=n): the texel gathers as buffer loads with cache-policy bits n (1 sc0, 2 nt, 16 sc1)
    const __amdgpu_buffer_rsrc_t tex_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)F.tex4, 0, F.nx * F.ny * F.nz * 4, 0x00020000);
#define DSC_TEX4(at) ((unsigned)__builtin_amdgcn_raw_buffer_load_b32(tex_rsrc, (int)((at) * 4u), 0, DSC_TEX_AUX))
#else
#define DSC_TEX4(at) (F.tex4[at])
#endif
    const double *Rrow = A.row_R + 9 * row;
    DSC_STAMP(1);

    const int ic0 = __builtin_amdgcn_readfirstlane(s_rec[1]), ic1 = __builtin_amdgcn_readfirstlane(s_rec[2]),
              ic2 = __builtin_amdgcn_readfirstlane(s_rec[3]);
    const float h0 = (float)sInv[0], h1 = (float)sInv[1], h2 = (float)sInv[2], h3 = (float)sInv[3], h4 = (float)sInv[4], h5 = (float)sInv[5],
                h6 = (float)sInv[6], h7 = (float)sInv[7], h8 = (float)sInv[8];
    // this thread's (j, k) column of the S^3 lattice; threads beyond S*S idle (S <= 16)
    const int j = tid / S, k = tid % S;
    const bool active = tid < S * S;
    // lattice coordinate along an axis = lbase + lstep * index (Descriptor.py:34-35)
    // (float32 here: half-integers below 32, exact.  The float64 forms the rare exact paths need are made inside those paths, from
    // values the optimiser cannot see through, so that they do not occupy registers outside them)
    const float lbf = oct == 0 ? (float)(-2 * A.r + 1) : (float)-A.r + 0.5f, lsf = oct == 0 ? 2.0f : 1.0f;
    auto exact_index = [&](int i, bool *left) -> unsigned {
        float lb = lbf, ls = lsf;
        asm volatile("" : "+v"(lb), "+v"(ls));
        const double lbase = (double)lb, lstep = (double)ls;
        return lattice_index_exact(lbase + lstep * i, lbase + lstep * j, lbase + lstep * k, sInv, (double)ic0, (double)ic1, (double)ic2, F, left);
    };
    // The rotated lattice stays within `reach` voxels of the anchor.  When that ball lies inside the grid with a voxel to
    // spare (every row of an anchor the Orientator accepted, unless it hugs the border) no sample can leave the grid.
    const int reach = (int)(1.7320508 * fabs((double)lbf)) + 2;
    const bool interior = ic0 - reach >= 1 && ic0 + reach <= F.nx - 2 && ic1 - reach >= 1 && ic1 + reach <= F.ny - 2 &&
                          ic2 - reach >= 1 && ic2 + reach <= F.nz - 2;      // uniform over the workgroup
    // The S samples of a thread go through in DSC_PASSES passes: the texel requests of a pass all go out before its first texel is
    // looked at, then the pass is classified.  DSC_PASSES = 1 keeps all S texels (64 registers at S = 16) in flight at once.
    constexpr int NP = (S % (DSC_CHUNK * DSC_PASSES) == 0) ? DSC_PASSES : 1;
    constexpr int PS = S / NP;
    bool oob = false;
    if (active) {
        // float32 guess of the offset from the anchor voxel, |error| < 1e-5 voxel: the nearest voxel is known unless the
        // fraction is within 2e-4 of the 0.5 tie (or, in a border row, the point is within 1e-3 of the grid edge).
        // Straight-line code: the (j, k) part of inv(R) l is formed once, every sample adds one fused multiply-add per
        // axis, and the texel requests go out as the indices appear (clamped into the grid in a border row: a sample that
        // needed the clamp is marked unsure and fetched again below).
        const float m1 = lbf + lsf * (float)j, m2 = lbf + lsf * (float)k;      // exact
        const float b0 = fmaf(m1, h1, m2 * h2), b1 = fmaf(m1, h4, m2 * h5), b2 = fmaf(m1, h7, m2 * h8);
        const int sub_jk = sub_of_jk<S, NSUB>(j, k);      // Descriptor.py:44-93
#pragma unroll
        for (int pass = 0; pass < NP; pass++) {
            float4 t[TAB ? 1 : PS];
            unsigned q4[TAB ? PS : 1], qi[TAB ? PS : 1];      // TAB: the 4-byte texels and where they came from
            unsigned unsure = 0;      // bit i: the float32 guess of sample i of this pass is too close to a tie to be trusted
            auto guess = [&](auto border) {
                const float fc0 = (float)ic0, fc1 = (float)ic1, fc2 = (float)ic2;
                const float lim0 = (float)(F.nx - 1) - 1e-3f, lim1 = (float)(F.ny - 1) - 1e-3f, lim2 = (float)(F.nz - 1) - 1e-3f;
#pragma unroll
                for (int i = 0; i < PS; i++) {
                    const float m0 = lbf + lsf * (float)(pass * PS + i);      // exact: half-integers below 32
                    const float a0 = fmaf(m0, h0, b0), a1 = fmaf(m0, h3, b1), a2 = fmaf(m0, h6, b2);
                    // nearest voxel = floor(a + 0.5) unless the fraction is within 2e-4 of the tie (then the float64 expression decides)
                    const float fr0 = __builtin_amdgcn_fractf(a0), fr1 = __builtin_amdgcn_fractf(a1), fr2 = __builtin_amdgcn_fractf(a2);
                    bool safe = fminf(fminf(fabsf(fr0 - 0.5f), fabsf(fr1 - 0.5f)), fabsf(fr2 - 0.5f)) > 2e-4f;      // (three subtractions, one v_min3 with |.| modifiers, one compare)
                    int n0 = ic0 + cvt_round(a0), n1 = ic1 + cvt_round(a1), n2 = ic2 + cvt_round(a2);
                    if (decltype(border)::value) {
                        const float q0 = a0 + fc0, q1 = a1 + fc1, q2 = a2 + fc2;
                        safe &= (q0 > 1e-3f) & (q0 < lim0) & (q1 > 1e-3f) & (q1 < lim1) & (q2 > 1e-3f) & (q2 < lim2);
                        n0 = min(max(n0, 0), F.nx - 1); n1 = min(max(n1, 0), F.ny - 1); n2 = min(max(n2, 0), F.nz - 1);
                    }
                    const unsigned at = mad_u24(mad_u24((unsigned)n0, (unsigned)F.ny, (unsigned)n1), (unsigned)F.nz, (unsigned)n2);      // nx ny < 2^24 (checked at allocation)
                    if (TAB) { qi[i] = at; q4[i] = DSC_TEX4(at); }
                    else t[i] = F.tex[at];
                    unsure |= safe ? 0u : (1u << i);
                }
            };
            if (interior) guess(std::false_type()); else guess(std::true_type());
            DSC_STAMP(2);
            if (unsure) {      // rare: the reference's float64 expression for those samples, and their texels again
#pragma unroll
                for (int i = 0; i < PS; i++)
                    if (unsure & (1u << i)) {
                        const unsigned at = exact_index(pass * PS + i, &oob);
                        if (TAB) { qi[i] = at; q4[i] = DSC_TEX4(at); }
                        else t[i] = F.tex[at];
                    }
            }
            // DSC_CHUNK points at a time: first their zones, in straight-line code (approximate unit direction, rotated in
            // float32: a guess, verified with guard bands inside eqsp_fast32), so that the table reads of different points
            // overlap; then the histogram updates.  The few points the fast classifier declines (~2 per row) are collected in
            // a bit mask and handed to the exact path afterwards.  (A sample that left the grid zeroes the whole row below:
            // what its pass counted in the meantime is never written.)
            unsigned undecided = 0;
            // (the rotation is converted here, behind the requests: nine registers fewer while the texels are in flight)
            // (TAB: the third row carries the 1 / 511 of the texel's components: z on the unit scale, x and y on any common one)
            const float zs = TAB ? (1.0f / 511.0f) : 1.0f;
            const float f0 = (float)Rrow[0], f1 = (float)Rrow[1], f2 = (float)Rrow[2], f3 = (float)Rrow[3], f4 = (float)Rrow[4], f5 = (float)Rrow[5],
                        f6 = (float)Rrow[6] * zs, f7 = (float)Rrow[7] * zs, f8 = (float)Rrow[8] * zs;
#pragma unroll
            for (int i0 = 0; i0 < PS; i0 += DSC_CHUNK) {
                int zone[DSC_CHUNK];
                unsigned any_flag = 0;      // TAB: the texels of this chunk or-ed together -- bit 31 says "some texel carries a flag"
#pragma unroll
                for (int u = 0; u < DSC_CHUNK; u++) {
                    if (i0 + u >= PS) { zone[u] = -2; continue; }      // S = 4, 12: the last chunk is short
                    if (TAB) {
                        const unsigned q = q4[i0 + u];
                        // (the builtin's return type is unsigned: without the cast to int a negative component converts as 4e9)
                        const float gx = (float)(int)__builtin_amdgcn_sbfe(q, 0, 10), gy = (float)(int)__builtin_amdgcn_sbfe(q, 10, 10),
                                    gz = (float)(int)__builtin_amdgcn_sbfe(q, 20, 10);
                        const float rx = fmaf(gz, f2, fmaf(gy, f1, gx * f0));
                        const float ry = fmaf(gz, f5, fmaf(gy, f4, gx * f3));
                        const float rz = fmaf(gz, f8, fmaf(gy, f7, gx * f6));
                        const int zn = eqsp_tab32((const EqspTabLds *)&tab, rx, ry, rz);
                        zone[u] = zn;      // (flags: below, once per chunk and only where there are any)
                        any_flag |= q;
                        continue;
                    }
                    const float4 tx = t[i0 + u];
                    const float inv = __builtin_amdgcn_rcpf(fmaxf(tx.w, 1e-30f));
                    const float gx = tx.x * inv, gy = tx.y * inv, gz = tx.z * inv;
                    const float rx = fmaf(gz, f2, fmaf(gy, f1, gx * f0));
                    const float ry = fmaf(gz, f5, fmaf(gy, f4, gx * f3));
                    const float rz = fmaf(gz, f8, fmaf(gy, f7, gx * f6));
                    const int zn = eqsp_fast32(fastp, rx, ry, rz);
                    zone[u] = tx.w < 1e-5f ? -2 : zn;                         // -2: Descriptor.py:190 (zone -1, not counted)
                }
                if (TAB && (int)any_flag < 0) {      // rare: 2 = not finite -> the exact tiers, 3 = below the magnitude cut-off, not counted (Descriptor.py:190)
#pragma unroll
                    for (int u = 0; u < DSC_CHUNK; u++) {
                        if (i0 + u >= PS) continue;
                        const unsigned fl = q4[TAB ? i0 + u : 0] >> 30;
                        zone[u] = fl == 3u ? -2 : (fl == 0u ? zone[u] : -1);
                    }
                }
#pragma unroll
                for (int u = 0; u < DSC_CHUNK; u++) {
                    if (i0 + u >= PS) continue;
                    if (zone[u] >= 0) atomicAdd(&hist[(sub_jk + sub_of_i<S, NSUB>(pass * PS + i0 + u)) * Z + zone[u]], 1);
                    undecided |= zone[u] == -1 ? (1u << (i0 + u)) : 0u;
                }
            }
            DSC_STAMP(3);
            if (TAB) {
                // The table leaves 3-6 % of the samples open: ~200 per row.  ONE queue reservation per wave (a scan of the lanes' counts;
                // every thread of an S = 16 row is active) -- a returning LDS atomic per sample on one address serialises the CU.
                const int cnt = __popc(undecided);
                const int inc = wave_incl_scan_i32(cnt);      // (DPP: a shuffle is an LDS instruction and waits behind the histogram's atomics)
                const int total = __builtin_amdgcn_readlane(inc, MAD_WAVE - 1);
                if (total) {      // (wave-uniform)
                    int base = 0;
                    if (lane_id() == 0) base = atomicAdd(&s_nq, total);
                    int off = __builtin_amdgcn_readfirstlane(base) + inc - cnt;
#pragma unroll
                    for (int i = 0; i < PS; i++)
                        if (undecided & (1u << i)) {
                            if (off < QCAP) { qidx[off] = qi[i]; qsub16[off] = (unsigned short)(sub_jk + sub_of_i<S, NSUB>(pass * PS + i)); }
                            off++;
                        }
                }
            } else if (undecided) {      // decide later with the exact arithmetic, with full lanes
#pragma unroll
                for (int i = 0; i < PS; i++)
                    if (undecided & (1u << i)) {
                        const int slot = atomicAdd(&s_nq, 1);
                        if (slot < QCAP) { qv[slot] = t[i]; qsub[slot] = sub_jk + sub_of_i<S, NSUB>(pass * PS + i); }
                    }
            }
        }
    }
    DSC_STAMP(4);
    if (oob) s_oob = 1;
    __syncthreads();
    DSC_STAMP(5);
    const bool dead = s_oob != 0;      // Descriptor.py:142-149: a sample left the grid -> the whole descriptor is zero
    if (dead) {
        for (int i = tid; i < D; i += DSC_THREADS) A.dsc[row * D + i] = 0;
        if (A.dsc8) {
            for (int i = tid; i < D; i += DSC_THREADS) A.dsc8[row * D + i] = 0;
            if (tid == 0) A.norm[row] = 0.0;
        }
    } else if (s_nq > QCAP) {
        // more undecided points than the queue holds (not seen in practice): redo the whole row with the exact arithmetic
        __syncthreads();
        for (int i = tid; i < D; i += DSC_THREADS) hist[i] = 0;
        __syncthreads();
        if (active)
            for (int i = 0; i < S; i++) {      // not unrolled: indices again from the float64 expression
                bool none = false;
                const float4 tx = F.tex[exact_index(i, &none)];
                if (tx.w < 1e-5f) continue;
                atomicAdd(&hist[(sub_of_jk<S, NSUB>(j, k) + sub_of_i<S, NSUB>(i)) * Z + describe_exact(exactp, tx, Rrow)], 1);
            }
    } else if (TAB) {
        // the samples the table left open: their 16-byte texels, the float32 tier with its 1e-4 guard, the float64 tier behind it
        const int nq = s_nq;
        const float f0 = (float)Rrow[0], f1 = (float)Rrow[1], f2 = (float)Rrow[2], f3 = (float)Rrow[3], f4 = (float)Rrow[4], f5 = (float)Rrow[5],
                    f6 = (float)Rrow[6], f7 = (float)Rrow[7], f8 = (float)Rrow[8];
        for (int e = tid; e < nq; e += DSC_THREADS) {
            const float4 tx = F.tex[qidx[e]];
            if (tx.w < 1e-5f) continue;      // (cannot happen for a queued sample: such texels carry flag 3; kept for symmetry with the slow path)
            const float inv = __builtin_amdgcn_rcpf(fmaxf(tx.w, 1e-30f));
            const float gx = tx.x * inv, gy = tx.y * inv, gz = tx.z * inv;
            int zn = eqsp_fast32(fastp, fmaf(gz, f2, fmaf(gy, f1, gx * f0)), fmaf(gz, f5, fmaf(gy, f4, gx * f3)), fmaf(gz, f8, fmaf(gy, f7, gx * f6)));
            if (zn < 0) zn = describe_exact(exactp, tx, Rrow);
            atomicAdd(&hist[(int)qsub16[e] * Z + zn], 1);
        }
    } else {
        const int nq = s_nq;
        for (int qi = tid; qi < nq; qi += DSC_THREADS) atomicAdd(&hist[qsub[qi] * Z + describe_exact(exactp, qv[qi], Rrow)], 1);
    }
    __syncthreads();
    DSC_STAMP(6);
    int ss = 0;      // counts <= 64, 1024 of them: the sum of squares is exact in int32
    for (int i = tid; i < D; i += DSC_THREADS) {
        const int v = hist[i];
        if (dead) continue;
        A.dsc[row * D + i] = (int16_t)v;
        if (A.dsc8) A.dsc8[row * D + i] = (int8_t)v;
        ss += v * v;
    }
    if (A.dsc8 && !dead) {
        ss = wave_sum_i32(ss);
        if (lane_id() == 0) s_part[tid >> 6] = ss;
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int w = 0; w < DSC_THREADS / MAD_WAVE; w++) tot += s_part[w];
            A.norm[row] = sqrt((double)tot);
        }
    }
    DSC_STAMP(7);
}

// ---------------------------------------------------------------------------
// descriptor kernel of the base octave: the anchor's ball of 4-byte texels in LDS (round 4)
// ---------------------------------------------------------------------------
// k_describe runs at the rate its gathers are accepted: ~3.3 clocks per scattered lane-load per CU, whatever the lanes' addresses have
// in common, 4 096 of them per row.  In the base octave (Descriptor.py:35: lattice -7.5 .. 7.5 voxels) every
// sample of every row of an anchor lies within 7.5 sqrt(3) = 12.99 voxels of it, so its nearest voxel d satisfies
// sum_i max(|d_i| - 0.5, 0)^2 <= 168.75: 11 027 texels = 43 KB as 4-byte texels.  One 1 024-thread workgroup per (anchor, run of
// DSCB_RPB rows) fetches that ball ONCE with coalesced loads (z-runs of up to 27 consecutive texels) and samples the rows from LDS:
// a thread owns four samples of a row -- one (j, k) column, a quarter of the i axis = one sub-region -- and the rows go through
// one after the other, one barrier each: histogram of row r (packed, two 16-bit counters per word: a count is <= 64) | barrier |
// the samples the table left open (their 16-byte texels from global memory, the float32 / float64 tiers, as in k_describe) and,
// by one wave, the write-out of row r - 1.  Same arithmetic per sample as k_describe<16, 64, 16, true>: same descriptors, bit for bit.
// (The upsampled octave's ball is 53^3 texels: it does not fit, and stays with k_describe.)
#define DSCB_THREADS 1024
#define DSCB_M 13
#define DSCB_SIDE (2 * DSCB_M + 1)
#define DSCB_COLS (DSCB_SIDE * DSCB_SIDE)
#define DSCB_E2MAX 676                  // voxel d is in the ball iff sum_i max(2 |d_i| - 1, 0)^2 <= 4 (7.5 sqrt(3) + 0.01)^2 = 676.04
#define DSCB_NBALL 11027
#define DSCB_RPB 4                      // rows of an anchor one workgroup takes, four waves each (an anchor of n rows: ceil(n / 4) workgroups, each with its own ball)
#define DSCB_FAST_BYTES ((offsetof(EqspFastLds, dir) + 15) / 16 * 16)
#define DSCB_OFF_TAB ((DSCB_NBALL + 3) / 4 * 16)
#define DSCB_OFF_FAST (DSCB_OFF_TAB + sizeof(EqspTabLds))
#define DSCB_OFF_COL (DSCB_OFF_FAST + DSCB_FAST_BYTES)
#define DSCB_COL_BYTES ((DSCB_COLS * 2 + 15) / 16 * 16)
#define DSCB_OFF_HIST (DSCB_OFF_COL + DSCB_COL_BYTES)
#define DSCB_OFF_Q (DSCB_OFF_HIST + DSCB_RPB * 512 * 4)
#define DSCB_OFF_FLAGS (DSCB_OFF_Q + DSCB_RPB * 256 * 4)
#define DSCB_ROWC 24                    // floats of a row's constants in LDS (21 used)
#define DSCB_OFF_ROWC (DSCB_OFF_FLAGS + 64)      // (wtot: one int per wave)
#define DSCB_OFF_ROWD (DSCB_OFF_ROWC + DSCB_RPB * DSCB_ROWC * 4 + 16)      // behind the constants: the rows' indices (4 ints)
#define DSCB_LDS_BYTES (DSCB_OFF_ROWD + DSCB_RPB * 9 * 8)
static_assert(DSCB_LDS_BYTES <= 80 * 1024, "two workgroups of k_describe_ball per CU");

// half-length of the z-run of column (cx, cy) of the ball (its voxels: z = DSCB_M - h .. DSCB_M + h), or -1
__host__ __device__ __forceinline__ int dscb_col_h(int cx, int cy) {
    const int ax = cx > DSCB_M ? cx - DSCB_M : DSCB_M - cx, ay = cy > DSCB_M ? cy - DSCB_M : DSCB_M - cy;
    const int ex = ax > 0 ? 2 * ax - 1 : 0, ey = ay > 0 ? 2 * ay - 1 : 0;
    const int rem = DSCB_E2MAX - ex * ex - ey * ey;
    if (rem < 0) return -1;
    const int sq = (int)__builtin_sqrtf((float)rem + 0.5f);      // floor(sqrt(rem)): rem <= 676, the square root is correctly rounded
    return (sq + 1) >> 1;                                         // max(2 |dz| - 1, 0) <= sq
}


// The float64 tier of k_describe_ball, out of line: inlined, its float64 temporaries (and the loop invariants the optimiser
// hoists for them) would set the register count of a kernel that has 64 registers per thread.
__device__ __noinline__ int dscb_describe_exact(const EqspFastLds *eq, float tx, float ty, float tz, float tw, const double *R) {
    return describe_exact(eq, make_float4(tx, ty, tz, tw), R);
}
typedef const __attribute__((address_space(4))) DscRowRec *dscb_rec_t;      // records and rotations were written by earlier launches:
typedef const __attribute__((address_space(4))) double *dscb_f64_t;         // constant for this kernel -> scalar loads

struct DescribeBallArgs {
    FieldDev f;                    // the base octave's field
    const DscRowRec *row_rec;      // the rows' records in working order
    const int32_t *anc_rows;       // per anchor in working order: {position of its first row, rows}
    const double *row_R;           // n_rows x 9
    int n_base, n_rowwise;         // anchors of this kernel, and the anchors before them in working order (k_describe's)
    const EqspDev *eq;
    int32_t *overflow;             // device: -2 when an anchor that is not interior arrives here
    const unsigned *colinfo;       // mad_ctx::ball_colinfo
    int16_t *dsc;                  // n_rows x 1024
    int8_t *dsc8;                  // nullable
    double *norm;
};

#ifdef MAD_PROBE_STAMPS      // diagnostic build: s_memtime at the phases of every workgroup (tools/probe_ball.py)
__device__ long long dscb_stamps[16384 * 8];
__device__ int dscb_stamp_rows[16384];
__device__ int dscb_stamp_n;
__device__ int dscb_dbg[16384 * 4];      // per workgroup: longest dscb_exact_voxel / dscb_describe_exact call of the drain (ticks), calls of each
extern "C" int mad_debug_dscb_dbg(int *out, int n) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(dscb_dbg), (size_t)n * 16) == hipSuccess ? 0 : -1; }
__device__ long long dscb_real[16384 * 2];      // s_memrealtime (100 MHz, one clock for the whole device) at a workgroup's start and end
extern "C" int mad_debug_dscb_real(long long *out, int n) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(dscb_real), (size_t)n * 16) == hipSuccess ? 0 : -1; }
#define DSCB_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0 && st_slot < 16384) dscb_stamps[st_slot * 8 + (k)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
extern "C" int mad_debug_dscb_stamps(long long *out, int *rows, int n) {
    int used = 0;
    if (hipMemcpyFromSymbol(&used, HIP_SYMBOL(dscb_stamp_n), 4) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(dscb_stamps), (size_t)n * 64) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(rows, HIP_SYMBOL(dscb_stamp_rows), (size_t)n * 4) != hipSuccess) return -1;
    const int zero = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(dscb_stamp_n), &zero, 4) != hipSuccess) return -1;
    return used;
}
#else
#define DSCB_STAMP(k) do { } while (0)
#endif
__global__ __launch_bounds__(DSCB_THREADS, 8) void k_describe_ball(Batch<DescribeBallArgs> B, int chunks) {
    extern __shared__ __align__(16) unsigned char dscb_lds[];
    const int job = batch_job(B, (int)blockIdx.x);
    const DescribeBallArgs &A = B.job[job];
    const int bid = (int)blockIdx.x - B.first[job];
    // workgroup -> (run of rows, anchor): the FIRST runs of all anchors come first in the grid (most anchors have no second one), and
    // within a run the anchors are dealt so that each XCD (workgroups b, b + 8, ...) gets a contiguous stretch of the Morton order
    const int per = (A.n_base + 7) / 8, nbp = per * 8;
    const int ch = bid / nbp, bi = bid - ch * nbp;
    const int slot = (bi & 7) * per + (bi >> 3);
    if (ch >= chunks || slot >= A.n_base) return;
    // one scalar load: where the anchor's rows lie and where the anchor is (written by k_orient_rows* of this build: constant here)
    typedef const __attribute__((address_space(4))) int32_t *ci32_t;
    const ci32_t an = (ci32_t)(A.anc_rows + MAD_ANCROW_WORDS * (int64_t)(A.n_rowwise + slot));
    const int pos0 = an[0], cnt = an[1], ic0 = an[2], ic1 = an[3], ic2 = an[4];
    const int r_begin = ch * DSCB_RPB;
    if (r_begin >= cnt) return;
    const int n_here = min(cnt - r_begin, DSCB_RPB);      // rows of this workgroup: r_begin .. r_begin + n_here - 1 of the anchor
#ifdef MAD_PROBE_STAMPS
    __shared__ int st_slot_s;
    if (threadIdx.x == 0) { st_slot_s = atomicAdd(&dscb_stamp_n, 1); if (st_slot_s < 16384) dscb_stamp_rows[st_slot_s] = n_here; }
    __syncthreads();
    const int st_slot = st_slot_s;
    __shared__ int dbg_s[4];
    if (threadIdx.x < 4) dbg_s[threadIdx.x] = 0;
    if (threadIdx.x == 0 && st_slot < 16384) dscb_real[2 * st_slot] = __builtin_amdgcn_s_memrealtime();
#endif
    DSCB_STAMP(0);

    unsigned *const ball = (unsigned *)dscb_lds;
    const EqspTabLds *const tab = (const EqspTabLds *)(dscb_lds + DSCB_OFF_TAB);
    const EqspFastLds *const fastp = (const EqspFastLds *)(dscb_lds + DSCB_OFF_FAST);      // float32 tier (the head of the image)
    const EqspFastLds *const exactp = &A.eq->image;                                          // float64 tier: global memory, a few samples per row
    short *const colb = (short *)(dscb_lds + DSCB_OFF_COL);
    const short *const colc = colb + DSCB_M * (DSCB_SIDE + 1);      // indexed by r0 * 27 + r1 with the offsets r = -13 .. 13 from the anchor voxel
    unsigned *const hist = (unsigned *)(dscb_lds + DSCB_OFF_HIST);      // per row of the run: 512 words of two 16-bit counters
    // per row and column: which of its 16 samples are open (the table could not decide them, or their voxel is next to a tie), and
    // the running count of open samples up to and including the column
    unsigned short *const omask = (unsigned short *)(dscb_lds + DSCB_OFF_Q), *const ocum = omask + DSCB_RPB * 256;
    int *const wtot = (int *)(dscb_lds + DSCB_OFF_FLAGS);                // per wave: its open samples
    float *const rowc = (float *)(dscb_lds + DSCB_OFF_ROWC);            // per row: hf[9], rf[9], ru[3] of its record
    int *const rowi = (int *)(rowc + DSCB_RPB * DSCB_ROWC);             // ... and its row index
    double *const rowd = (double *)(dscb_lds + DSCB_OFF_ROWD);          // ... and inv(Rfinal) in float64 (the voxel of a sample next to a tie)
    const int tid = (int)threadIdx.x;
    const FieldDev F = A.f;
    const DscRowRec *const grec = A.row_rec + pos0 + r_begin;           // the records of this run
    // Only anchors whose ball lies inside the grid with a voxel to spare come here (no sample of theirs can leave the grid, as in
    // k_describe's `interior`); the host sorts the others in front of `n_rowwise` with the same expression (mad_ball_interior).
    if (!mad_ball_interior(ic0, ic1, ic2, F.nx, F.ny, F.nz)) {
        if (tid == 0) *A.overflow = -2;      // refused, loudly: the set's describe stage reports the flag
        return;
    }

    // ---- tables (one 16-byte piece per thread) and the ball's column list first; then half a wave per x-plane of the ball: the 27
    // ---- columns' texels of a thread all requested before the first is stored (z-runs of up to 27 consecutive texels)
    unsigned *const cinfo = (unsigned *)omask;      // (the open-sample lists are not in use yet) per column: LDS offset of its first texel | half-length << 16 (31: none)
    {
        constexpr int N_TAB = (int)sizeof(EqspTabLds) / 16, N_FAST = (int)DSCB_FAST_BYTES / 16;
        static_assert(N_TAB + N_FAST <= DSCB_THREADS && DSCB_COLS <= DSCB_THREADS && DSCB_RPB * 256 * 4 >= DSCB_COLS * 4, "one table piece per thread");
        const uint4 *src = (const uint4 *)&A.eq->tab;      // (a valid address for every thread; stored only where dst4 is set)
        uint4 *dst4 = nullptr;
        if (tid < N_TAB) { src = (const uint4 *)&A.eq->tab + tid; dst4 = (uint4 *)tab + tid; }
        else if (tid < N_TAB + N_FAST) { src = (const uint4 *)&A.eq->image + (tid - N_TAB); dst4 = (uint4 *)fastp + (tid - N_TAB); }
        const uint4 tv = *src;
        const unsigned ci = A.colinfo[min(tid, DSCB_COLS - 1)];
        const int rq = tid / DSCB_ROWC, ri = tid - rq * DSCB_ROWC;
        const bool has_rc = rq < n_here && ri < 21;
        const float rcv = ((const float *)grec[has_rc ? rq : 0].hf)[has_rc ? ri : 0];      // hf, rf, ru are contiguous
        const int riv = grec[tid < n_here ? tid : 0].row;
        const bool has_rd = tid < 9 * n_here;
        const double rdv = grec[has_rd ? tid / 9 : 0].inv[has_rd ? tid % 9 : 0];
        if (dst4) *dst4 = tv;
        if (has_rd) rowd[tid] = rdv;
        if (tid < DSCB_COLS) {
            cinfo[tid] = ci;
            colb[tid] = (short)((ci & 0xffffu) + (ci >> 16));      // the column's middle texel: what a sample's dz is added to
        }
        if (has_rc) rowc[tid] = rcv;
        if (tid < n_here) rowi[tid] = riv;
        for (int i = tid; i < DSCB_RPB * 512; i += DSCB_THREADS) hist[i] = 0;
        __syncthreads();
        const int hw = tid >> 5, zl = tid & 31;      // half-wave hw: the plane x = ic0 - 13 + hw; lane zl: the zl-th texel of a column's run
        if (hw < DSCB_SIDE) {
            unsigned v[DSCB_SIDE];
            const unsigned plane = mad_u24((unsigned)(ic0 - DSCB_M + hw), (unsigned)F.ny, (unsigned)(ic1 - DSCB_M));
#pragma unroll
            for (int u = 0; u < DSCB_SIDE; u++) {
                const int h = (int)(cinfo[hw * DSCB_SIDE + u] >> 16);
                // (the ball lies inside the grid; the clamp keeps the lanes beyond a column's run, or of no column, inside the texture)
                const int gz = min(max(ic2 - h + zl, 0), F.nz - 1);
                v[u] = F.tex4[mad_u24(plane + (unsigned)u, (unsigned)F.nz, (unsigned)gz)];
            }
#pragma unroll
            for (int u = 0; u < DSCB_SIDE; u++) {
                const unsigned c2 = cinfo[hw * DSCB_SIDE + u];
                const int h = (int)(c2 >> 16);
                if (h != 31 && zl <= 2 * h) ball[(c2 & 0xffffu) + zl] = v[u];
            }
        }
    }
    __syncthreads();
    DSCB_STAMP(1);

    // Four waves per row, the rows of the run side by side: wave w belongs to row g = w / 4; its thread owns column (j, k) of that
    // row's lattice and walks i = 0 .. 15 four samples at a time -- a sub-region (Descriptor.py:44-64) per trip.
    const int g = tid >> 8, col = tid & 255, j = col >> 4, k = col & 15;
    const bool live = g < n_here;      // (uniform per wave)
    const int sub_jk = (j >> 2) * 16 + (k >> 2);
    const float lbf = -7.5f;      // Descriptor.py:35, dsc_radius 16: the lattice -7.5 .. 7.5
    const float m1 = lbf + (float)j, m2 = lbf + (float)k;
    const float *const rcp = rowc + (live ? g : 0) * DSCB_ROWC;
    auto rc = [&](int i) {      // constant i of this wave's row: the same value in every lane -> a scalar register
        return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, rcp[i])));
    };
    unsigned *const H = hist + g * 512;

    // ---- table tier.  The float32 guess of a sample's offset from the anchor voxel is formed exactly as in k_describe: the nearest
    // ---- voxel is cvt_round(a) unless a fraction is within 2e-4 of the 0.5 tie -- such a sample is left open with the ones the
    // ---- table cannot decide, and whoever takes it from the row's list lets the reference's float64 expression choose the voxel.
    unsigned openmask = 0;      // bit i: sample i of this thread is open
    int open_inc = 0;
    if (live) {
        const float h0 = rc(0), h1 = rc(1), h2 = rc(2), h3 = rc(3), h4 = rc(4), h5 = rc(5), h6 = rc(6), h7 = rc(7), h8 = rc(8);      // (float)inv(Rfinal)
        // (float)Rfinal, the third row with the 1 / 511 of the texel's components: z on the unit scale, x and y on any common one
        const float f0 = rc(9), f1 = rc(10), f2 = rc(11), f3 = rc(12), f4 = rc(13), f5 = rc(14), f6 = rc(15), f7 = rc(16), f8 = rc(17);
        const float b0 = fmaf(m1, h1, m2 * h2), b1 = fmaf(m1, h4, m2 * h5), b2 = fmaf(m1, h7, m2 * h8);
        // A row whose rotation keeps a grid axis (nearly) fixed has EVERY sample next to a tie along that axis: the lattice sits on
        // half-integers (Descriptor.py:35), and the reference's float64 expression has to place each sample (scipy: a fraction <= 0.5
        // takes the lower voxel).  These are the rows of main bin 0 / 111 -- the poles: Rfinal is a turn about z -- about one row in
        // a hundred.  Such a row takes that expression for all its samples here, in line, instead of leaving 4 096 samples open.
        // (Which rows take this path changes their cost, never their result: both paths place a sample with the float64 expression
        // whenever float32 cannot.)
        auto tiny = [](float x) { return fabsf(x) < 2e-5f ? 1 : 0; };
        const bool tie_row = tiny(h0) + tiny(h1) + tiny(h2) >= 2 || tiny(h3) + tiny(h4) + tiny(h5) >= 2 || tiny(h6) + tiny(h7) + tiny(h8) >= 2;
        const double *const dinv = rowd + 9 * g;
#pragma unroll 1
        for (int i0 = 0; i0 < 16; i0 += 4) {
            unsigned q[4], open = 0;
            if (tie_row) {      // (uniform per wave)
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const double l0 = -7.5 + (double)(i0 + u), l1 = -7.5 + (double)j, l2 = -7.5 + (double)k;
                    int r[3];
#pragma unroll
                    for (int ax = 0; ax < 3; ax++) {
                        const int ic = ax == 0 ? ic0 : (ax == 1 ? ic1 : ic2), nmax = ax == 0 ? F.nx : (ax == 1 ? F.ny : F.nz);
                        const double pp = (l0 * dinv[3 * ax] + l1 * dinv[3 * ax + 1] + l2 * dinv[3 * ax + 2]) + (double)ic;      // Descriptor.py:132-133
                        const int lo = min((int)floor(pp), nmax - 2);
                        r[ax] = ((pp - (double)lo <= 0.5) ? lo : lo + 1) - ic;
                    }
                    const int li = (int)colc[mad_i24(r[0], DSCB_SIDE, r[1])] + r[2];
                    q[u] = ball[min(max(li, 0), DSCB_NBALL - 1)];
                }
            } else {
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const float m0 = lbf + (float)(i0 + u);
                    const float a0 = fmaf(m0, h0, b0), a1 = fmaf(m0, h3, b1), a2 = fmaf(m0, h6, b2);
                    const float fr0 = __builtin_amdgcn_fractf(a0), fr1 = __builtin_amdgcn_fractf(a1), fr2 = __builtin_amdgcn_fractf(a2);
                    const bool safe = fminf(fminf(fabsf(fr0 - 0.5f), fabsf(fr1 - 0.5f)), fabsf(fr2 - 0.5f)) > 2e-4f;
                    // |a| <= 12.991, so the voxel lies in the ball; the clamp of the LDS index keeps a corrupt record from reading outside it
                    const int cidx = mad_i24(cvt_round(a0), DSCB_SIDE, cvt_round(a1));
                    const int li = (int)colc[cidx] + cvt_round(a2);
                    q[u] = ball[min(max(li, 0), DSCB_NBALL - 1)];
                    open |= safe ? 0u : (1u << u);
                }
            }
            // zones of the four samples in straight-line code (4-byte texel -> float32 rotation -> table: k_describe's TAB tier)
            int zone[4];
            unsigned any_flag = 0;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const unsigned t = q[u];
                const float gx = (float)(int)__builtin_amdgcn_sbfe(t, 0, 10), gy = (float)(int)__builtin_amdgcn_sbfe(t, 10, 10),
                            gz = (float)(int)__builtin_amdgcn_sbfe(t, 20, 10);
                const float rx = fmaf(gz, f2, fmaf(gy, f1, gx * f0));
                const float ry = fmaf(gz, f5, fmaf(gy, f4, gx * f3));
                const float rz = fmaf(gz, f8, fmaf(gy, f7, gx * f6));
                zone[u] = eqsp_tab32(tab, rx, ry, rz);
                any_flag |= t;
            }
            if ((int)any_flag < 0) {      // rare: 2 = not finite -> the exact tiers, 3 = below the magnitude cut-off, not counted (Descriptor.py:190)
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const unsigned fl = q[u] >> 30;
                    zone[u] = fl == 3u ? -2 : (fl == 0u ? zone[u] : -1);
                }
            }
            const int sub8 = (sub_jk + i0) * 8;      // sub-region of these four samples: i0 / 4 along i = + 4 (i0 / 4), x 8 words
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const bool op = (open >> u) & 1u;      // voxel not certain: whatever the guessed texel said does not count
                if (zone[u] >= 0 && !op) atomicAdd(&H[sub8 + (zone[u] >> 1)], 1u << ((zone[u] & 1) << 4));
                open |= zone[u] == -1 ? (1u << u) : 0u;
            }
            openmask |= open << i0;
        }
        // where this thread's open samples stand in the row's list: a scan over the wave now, the waves before it after the barrier
        open_inc = wave_incl_scan_i32(__popc(openmask));
        if (lane_id() == MAD_WAVE - 1) wtot[tid >> 6] = open_inc;
    }
    DSCB_STAMP(2);
    __syncthreads();
    DSCB_STAMP(3);

    // ---- the open samples of each row (~4 % of a row; up to most of it where the directions hug a zone's edge), dealt evenly to
    // ---- the row's 256 threads whatever their number: the columns' masks and running counts in LDS, the e-th open sample found by
    // ---- a binary search.  Per sample: its voxel again (float64 next to a tie: Descriptor.py:132-133, scipy's rule), its 16-byte
    // ---- texel, the float32 tier with its 1e-4 guard, the float64 tier behind it -- as k_describe's queue phase.
    if (live) {
        int before = 0;
        for (int w = g * 4; w < (tid >> 6); w++) before += wtot[w];
        ocum[g * 256 + col] = (unsigned short)(open_inc + before);
        omask[g * 256 + col] = (unsigned short)openmask;
    }
    __syncthreads();
    auto settle = [&](int ei, int ej, int ek) {
        const float e1 = lbf + (float)ej, e2 = lbf + (float)ek, e0 = lbf + (float)ei;
        const float a0 = fmaf(e0, rcp[0], fmaf(e1, rcp[1], e2 * rcp[2])), a1 = fmaf(e0, rcp[3], fmaf(e1, rcp[4], e2 * rcp[5])),
                    a2 = fmaf(e0, rcp[6], fmaf(e1, rcp[7], e2 * rcp[8]));
        const bool safe = fminf(fminf(fabsf(__builtin_amdgcn_fractf(a0) - 0.5f), fabsf(__builtin_amdgcn_fractf(a1) - 0.5f)), fabsf(__builtin_amdgcn_fractf(a2) - 0.5f)) > 2e-4f;
        int r0 = cvt_round(a0), r1 = cvt_round(a1), r2 = cvt_round(a2);      // the voxel's offset from the anchor's
        const int sb8 = (((ej >> 2) * 16 + (ek >> 2)) + (ei & ~3)) * 8;
        if (!safe) {
            // Next to a tie the reference's own float64 expression decides (Descriptor.py:132-133; scipy's nearest rule: a fraction
            // <= 0.5 takes the lower voxel).  Rows turned by a multiple of 90 degrees have EVERY sample there (the lattice sits on
            // half-integers), so this is straight-line code, an axis at a time, not a call.  The ball lies inside the grid.
            const double *const dinv = rowd + 9 * g;
            const double l0 = -7.5 + (double)ei, l1 = -7.5 + (double)ej, l2 = -7.5 + (double)ek;
            auto nearest = [&](int ax, int ic, int nmax) {
                const double pp = (l0 * dinv[3 * ax] + l1 * dinv[3 * ax + 1] + l2 * dinv[3 * ax + 2]) + (double)ic;
                const int lo = min((int)floor(pp), nmax - 2);
                const int v = (pp - (double)lo <= 0.5) ? lo : lo + 1;
                return min(max(v - ic, -DSCB_M), DSCB_M);      // (the clamp never acts: |offset| <= 12.991)
            };
            r0 = nearest(0, ic0, F.nx); r1 = nearest(1, ic1, F.ny); r2 = nearest(2, ic2, F.nz);
            // the voxel is settled, its 4-byte texel is in the ball: the table tier, as for every other sample
            const unsigned t = ball[min(max((int)colc[mad_i24(r0, DSCB_SIDE, r1)] + r2, 0), DSCB_NBALL - 1)];
            const unsigned fl = t >> 30;
            if (fl == 3u) return;      // below the magnitude cut-off: not counted (Descriptor.py:190)
            if (fl == 0u) {
                const float gx = (float)(int)__builtin_amdgcn_sbfe(t, 0, 10), gy = (float)(int)__builtin_amdgcn_sbfe(t, 10, 10),
                            gz = (float)(int)__builtin_amdgcn_sbfe(t, 20, 10);
                const int zn = eqsp_tab32(tab, fmaf(gz, rcp[11], fmaf(gy, rcp[10], gx * rcp[9])), fmaf(gz, rcp[14], fmaf(gy, rcp[13], gx * rcp[12])),
                                          fmaf(gz, rcp[17], fmaf(gy, rcp[16], gx * rcp[15])));
                if (zn >= 0) {
                    atomicAdd(&H[sb8 + (zn >> 1)], 1u << ((zn & 1) << 4));
                    return;
                }
            }
        }
        // the 16-byte texel: the float32 tier with its 1e-4 guard, the float64 tier behind it
        const float4 tx = F.tex[mad_u24(mad_u24((unsigned)(ic0 + r0), (unsigned)F.ny, (unsigned)(ic1 + r1)), (unsigned)F.nz, (unsigned)(ic2 + r2))];
        if (tx.w < 1e-5f) return;      // (not counted, Descriptor.py:190 -- such texels carry flag 3 and never get here)
        const float inv = __builtin_amdgcn_rcpf(fmaxf(tx.w, 1e-30f));
        const float gx = tx.x * inv, gy = tx.y * inv, gz = tx.z * inv;
        int zn = eqsp_fast32(fastp, fmaf(gz, rcp[11], fmaf(gy, rcp[10], gx * rcp[9])), fmaf(gz, rcp[14], fmaf(gy, rcp[13], gx * rcp[12])),
                             fmaf(gz, rcp[20], fmaf(gy, rcp[19], gx * rcp[18])));
#ifdef MAD_PROBE_STAMPS
        if (zn < 0) atomicAdd(&dbg_s[3], 1);
#endif
        if (zn < 0) zn = dscb_describe_exact(exactp, tx.x, tx.y, tx.z, tx.w, A.row_R + 9 * (int64_t)rowi[g]);
        atomicAdd(&H[sb8 + (zn >> 1)], 1u << ((zn & 1) << 4));
    };
    if (live) {
        const unsigned short *const cum = ocum + g * 256, *const msk = omask + g * 256;
        const int n_open = cum[255];
#ifdef MAD_PROBE_STAMPS
        if (col == 0) { atomicMax(&dbg_s[0], n_open); atomicAdd(&dbg_s[1], n_open); }
#endif
        for (int e = col; e < n_open; e += 256) {
            int c = 0;      // the first column whose running count exceeds e
#pragma unroll
            for (int st = 128; st; st >>= 1)
                if ((int)cum[c + st - 1] <= e) c += st;
            unsigned m = msk[c];
            for (int skip = e - ((int)cum[c] - __popc(m)); skip > 0; skip--) m &= m - 1;
            settle(__builtin_ctz(m), c >> 4, c & 15);
        }
    }
    DSCB_STAMP(4);
    __syncthreads();
    DSCB_STAMP(5);
    DSCB_STAMP(6);

    // ---- the first wave of a row: its 1 024 counts as int16 (the packed words ARE the row) and int8, its norm
    if (live && (tid & 255) < MAD_WAVE) {
        const int l = (int)lane_id();
        const uint4 *const H4 = (const uint4 *)H;
        const uint4 w0 = H4[2 * l], w1 = H4[2 * l + 1];
        const int64_t row = rowi[g];
        uint4 *const o16 = (uint4 *)(A.dsc + row * 1024);
        o16[2 * l] = w0;
        o16[2 * l + 1] = w1;
        if (A.dsc8) {
            auto b2 = [](unsigned x) { return (x & 0xffu) | ((x >> 8) & 0xff00u); };      // two counts -> two bytes
            auto sq = [](unsigned x) { const int a = (int)(x & 0xffffu), b = (int)(x >> 16); return a * a + b * b; };
            ((uint4 *)(A.dsc8 + row * 1024))[l] = make_uint4(b2(w0.x) | b2(w0.y) << 16, b2(w0.z) | b2(w0.w) << 16, b2(w1.x) | b2(w1.y) << 16, b2(w1.z) | b2(w1.w) << 16);
            int ss = sq(w0.x) + sq(w0.y) + sq(w0.z) + sq(w0.w) + sq(w1.x) + sq(w1.y) + sq(w1.z) + sq(w1.w);      // counts <= 64: exact in int32
            ss = wave_sum_i32(ss);
            if (l == 0) A.norm[row] = sqrt((double)ss);
        }
    }
    DSCB_STAMP(7);
#ifdef MAD_PROBE_STAMPS
    if (threadIdx.x == 0 && st_slot < 16384) dscb_real[2 * st_slot + 1] = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x < 4 && st_slot < 16384) dscb_dbg[4 * st_slot + threadIdx.x] = dbg_s[threadIdx.x];
#endif
}

static int ensure_ball(mad_ctx *ctx) {
    if (ctx->ball_colinfo) return MAD_OK;
    unsigned h[DSCB_COLS];
    int base = 0;
    for (int c = 0; c < DSCB_COLS; c++) {
        const int hh = dscb_col_h(c / DSCB_SIDE, c % DSCB_SIDE);
        h[c] = hh < 0 ? (31u << 16) : ((unsigned)base | (unsigned)hh << 16);      // LDS index of the column's voxel dz = -hh: base; it holds 2 hh + 1 texels
        if (hh >= 0) base += 2 * hh + 1;
    }
    if (base != DSCB_NBALL) return mad_fail(ctx, MAD_EINVAL, "k_describe_ball: the ball has %d texels, built for %d", base, DSCB_NBALL);
    MAD_HIP(hipMalloc((void **)&ctx->ball_colinfo, sizeof(h)));
    MAD_HIP(hipMemcpy(ctx->ball_colinfo, h, sizeof(h), hipMemcpyHostToDevice));
    MAD_HIP(hipFuncSetAttribute((const void *)k_describe_ball, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DSCB_LDS_BYTES));
    return MAD_OK;
}

int mad_describe_device_many(mad_ctx *ctx, int n_jobs, const DescribeJob *jobs, int r, int dsc_size) {
    if (dsc_size != 64 && (2 * r != 16 || (dsc_size != 27 && dsc_size != 8 && dsc_size != 1)))
        return mad_fail(ctx, MAD_EINVAL, "mad_describe: dsc_size %d (27, 8 and 1 are built for the default dsc_radius 16 only; 64 for 4 ... 16)", dsc_size);
    if (!ctx->eq_set[1]) return mad_fail(ctx, MAD_EINVAL, "mad_describe: descriptor EQSP table not set");
    const int Zd = ctx->eq_host[1].Z;
    if (Zd != 16 && !(Zd <= 128 && dsc_size == 64 && 2 * r == 16))
        return mad_fail(ctx, MAD_EINVAL, "mad_describe: %d descriptor zones: 16 for every layout, up to 128 (the 112-zone table) for the default 64 regions and dsc_radius 16", Zd);
    if (r < 2 || r > 8 || (r % 2)) return mad_fail(ctx, MAD_EINVAL, "mad_describe: dsc radius %d must be 2, 4, 6 or 8", r);
    const bool tab = Zd == 16 && ctx->eq_host[1].tab_ok && dsc_size == 64 && 2 * r == 16;
    const bool ball_ok = tab && ctx->dsc_ball;      // (mad_set_option "dsc_ball", MAD_BALL=1: off by default)
    if (ball_ok) MAD_TRY(ensure_ball(ctx));
    for (int j0 = 0; j0 < n_jobs; j0 += MAD_BATCH_MAX) {
        Batch<DescribeArgs> B;
        Batch<DescribeBallArgs> BB;      // the base-octave anchors of the same jobs (k_describe_ball)
        B.n_jobs = 0;
        BB.n_jobs = 0;
        int64_t blk = 0, bblk = 0;
        int max_fan = 1;
        for (int j = j0; j < n_jobs && j < j0 + MAD_BATCH_MAX; j++) {
            const DescribeJob &J = jobs[j];
            if (J.grid_rows <= 0) continue;
            for (int o = 0; o < 2; o++) {
                const FieldDev &f = J.f[o];
                if (f.tex && ((size_t)f.nx * f.ny * f.nz >= (size_t)1 << 32 || (size_t)f.nx * f.ny >= (size_t)1 << 24 || f.nz >= 1 << 24))
                    return mad_fail(ctx, MAD_EINVAL, "mad_describe: field of %dx%dx%d texels exceeds 2^32 (or 2^24 per x-y plane)", f.nx, f.ny, f.nz);
            }
            // the anchors of the base octave (behind those of octave 0 in working order) go through the ball kernel when the set
            // pipeline has told where each anchor's rows lie
            const int n_base = J.n_anchors - J.n_rowwise;
            const bool ball = ball_ok && J.d_anc_rows && J.d_row_rec && J.d_row_perm && n_base > 0 && J.n_rowwise >= 0 && J.fan > 0 && J.f[1].tex4;
            DescribeArgs &A = B.job[B.n_jobs];
            A.f[0] = J.f[0]; A.f[1] = J.f[1];
            A.anc_coords = J.d_anc_coords; A.anc_octave = J.d_anc_octave; A.uniform_octave = J.uniform_octave;
            A.row_anchor = J.d_row_anchor; A.row_R = J.d_row_R; A.row_Rinv = J.d_row_Rinv; A.row_perm = J.d_row_perm; A.n_rows = J.d_n_rows; A.overflow = J.d_overflow;
            A.r = r; A.eq = ctx->eq[1]; A.dsc = J.d_dsc; A.dsc8 = J.d_dsc8; A.norm = J.d_norm; A.row_rec = J.d_row_rec;
            A.queue_cap = std::max(ctx->dsc_queue_cap, 0);
            A.row_limit = ball ? J.d_anc_rows + MAD_ANCROW_WORDS * (int64_t)J.n_rowwise : nullptr;      // first row position of the first base-octave anchor
            B.first[B.n_jobs++] = (int)blk;
            // one workgroup per possible row, a multiple of 8 per job (one share per XCD).  With the ball kernel this grid only has
            // the octave-0 rows to cover: their share of the hint by anchors, with room (a launch that falls short raises `overflow`)
            int64_t rows_here = J.grid_rows;
            if (ball) rows_here = std::min<int64_t>(J.grid_rows, (int64_t)((double)J.grid_rows * J.n_rowwise / J.n_anchors * 1.3) + 64);
            blk += ((rows_here + 7) / 8) * 8 + 8;
            if (blk > INT32_MAX) return mad_fail(ctx, MAD_EINVAL, "mad_describe: %lld rows in one batch", (long long)blk);
            if (ball) {
                DescribeBallArgs &Q = BB.job[BB.n_jobs];
                Q.f = J.f[1]; Q.row_rec = J.d_row_rec; Q.anc_rows = J.d_anc_rows; Q.row_R = J.d_row_R;
                Q.n_base = n_base; Q.n_rowwise = J.n_rowwise;
                Q.eq = ctx->eq[1]; Q.overflow = J.d_overflow; Q.colinfo = ctx->ball_colinfo; Q.dsc = J.d_dsc; Q.dsc8 = J.d_dsc8; Q.norm = J.d_norm;
                BB.first[BB.n_jobs++] = (int)bblk;
                max_fan = std::max(max_fan, J.fan);
            }
        }
        if (B.n_jobs == 0) continue;
        B.first[B.n_jobs] = (int)blk;
        const unsigned nblk = (unsigned)blk;
        static const size_t dsc_pad = getenv("MAD_DSC_PAD_KB") ? (size_t)atoi(getenv("MAD_DSC_PAD_KB")) * 1024 : 0;      // probe: LDS a row's workgroup holds without using it (fewer rows per CU, room for other kernels' workgroups)
        mad_timer_begin(ctx, MAD_T_DESCRIBE);
        // the default layout takes the 4-byte texels and the table classifier when the descriptor table has them (MAD_NO_TAB: never)
        switch (dsc_size == 64 ? 2 * r : -dsc_size) {
            case 4: hipLaunchKernelGGL(k_describe<4>, dim3(nblk), dim3(DSC_THREADS), 0, ctx->stream, B); break;
            case 8: hipLaunchKernelGGL(k_describe<8>, dim3(nblk), dim3(DSC_THREADS), 0, ctx->stream, B); break;
            case 12: hipLaunchKernelGGL(k_describe<12>, dim3(nblk), dim3(DSC_THREADS), 0, ctx->stream, B); break;
            case -27: hipLaunchKernelGGL((k_describe<16, 27>), dim3(nblk), dim3(DSC_THREADS), 0, ctx->stream, B); break;
            case -8: hipLaunchKernelGGL((k_describe<16, 8>), dim3(nblk), dim3(DSC_THREADS), 0, ctx->stream, B); break;
            case -1: hipLaunchKernelGGL((k_describe<16, 1>), dim3(nblk), dim3(DSC_THREADS), 0, ctx->stream, B); break;
            default:
                if (Zd > 16) hipLaunchKernelGGL((k_describe<16, 64, 128>), dim3(nblk), dim3(DSC_THREADS), 0, ctx->stream, B);
                else if (tab) hipLaunchKernelGGL((k_describe<16, 64, 16, true>), dim3(nblk), dim3(DSC_THREADS), dsc_pad, ctx->stream, B);
                else hipLaunchKernelGGL(k_describe<16>, dim3(nblk), dim3(DSC_THREADS), 0, ctx->stream, B);
                break;
        }
        if (BB.n_jobs > 0) {
            // every job's grid: `chunks` runs of DSCB_RPB rows x its base-octave anchors rounded up to 8 (an anchor has at most fan rows;
            // the workgroups of runs an anchor does not have return at once)
            static const int chunks_probe = getenv("MAD_BALL_CHUNKS") ? atoi(getenv("MAD_BALL_CHUNKS")) : 0;      // timing probe only: anchors with more rows lose them
            const int chunks = chunks_probe > 0 ? chunks_probe : (max_fan + DSCB_RPB - 1) / DSCB_RPB;
            int64_t at = 0;
            for (int q = 0; q < BB.n_jobs; q++) {
                BB.first[q] = (int)at;
                at += (int64_t)chunks * ((BB.job[q].n_base + 7) / 8 * 8);
                if (at > INT32_MAX) return mad_fail(ctx, MAD_EINVAL, "mad_describe: %lld ball workgroups in one batch", (long long)at);
            }
            BB.first[BB.n_jobs] = (int)at;
            hipLaunchKernelGGL(k_describe_ball, dim3((unsigned)at), dim3(DSCB_THREADS), DSCB_LDS_BYTES, ctx->stream, BB, chunks);
        }
        mad_timer_end(ctx, MAD_T_DESCRIBE);
        MAD_HIP(hipGetLastError());
    }
    return MAD_OK;
}

int mad_describe_device(mad_ctx *ctx, FieldDev f0, FieldDev f1, const int32_t *d_anc_coords, const int32_t *d_anc_octave,
                        int uniform_octave, const int32_t *d_row_anchor, const double *d_row_R, const double *d_row_Rinv,
                        const int32_t *d_n_rows, int64_t grid_rows, int32_t *d_overflow, int r, int16_t *d_dsc, int8_t *d_dsc8, double *d_norm,
                        int dsc_size) {
    DescribeJob J;
    J.f[0] = f0; J.f[1] = f1; J.d_anc_coords = d_anc_coords; J.d_anc_octave = d_anc_octave; J.uniform_octave = uniform_octave;
    J.d_row_anchor = d_row_anchor; J.d_row_R = d_row_R; J.d_row_Rinv = d_row_Rinv; J.d_n_rows = d_n_rows; J.grid_rows = grid_rows;
    J.d_overflow = d_overflow; J.d_dsc = d_dsc; J.d_dsc8 = d_dsc8; J.d_norm = d_norm;
    return mad_describe_device_many(ctx, 1, &J, r, dsc_size);
}

extern "C" int mad_describe(mad_ctx *ctx, int slot, int octave, const int32_t *coords, const double *R, int64_t n_rows,
                            int r, int16_t *dsc) {
    return mad_describe_sized(ctx, slot, octave, coords, R, n_rows, r, 64, dsc);
}

extern "C" int mad_describe_sized(mad_ctx *ctx, int slot, int octave, const int32_t *coords, const double *R, int64_t n_rows,
                                  int r, int dsc_size, int16_t *dsc) {
    if (ctx) mad_use_lane(ctx, 0);
    if (!ctx) return MAD_EINVAL;
    if (slot < 0 || slot >= MAD_MAX_FIELDS || !ctx->fields[slot].tex)
        return mad_fail(ctx, MAD_EINVAL, "mad_describe: field slot %d is empty", slot);
    if (octave != 0 && octave != 1) return mad_fail(ctx, MAD_EINVAL, "mad_describe: octave %d", octave);
    if (n_rows <= 0) return MAD_OK;
    if (!coords || !R || !dsc) return mad_fail(ctx, MAD_EINVAL, "mad_describe: NULL argument");
    if (dsc_size != 64 && dsc_size != 27 && dsc_size != 8 && dsc_size != 1) return mad_fail(ctx, MAD_EINVAL, "mad_describe: invalid dsc size %d", dsc_size);
    const int D = dsc_size * ctx->eq_host[1].Z;
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_ROW_COORDS), (size_t)n_rows * 12));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_ROW_R), (size_t)n_rows * 72));
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_DSC), (size_t)n_rows * D * 2));
    MAD_HIP(hipMemcpyAsync(mad_sb(ctx, S_ROW_COORDS).p, coords, (size_t)n_rows * 12, hipMemcpyHostToDevice, ctx->stream));
    MAD_HIP(hipMemcpyAsync(mad_sb(ctx, S_ROW_R).p, R, (size_t)n_rows * 72, hipMemcpyHostToDevice, ctx->stream));
    FieldDev f = ctx->fields[slot];
    MAD_TRY(mad_reserve(ctx, mad_sb(ctx, S_MISC), 256));
    int32_t *d_n = scratch<int32_t>(ctx, S_MISC) + 18;
    const int32_t n32 = (int32_t)n_rows;
    MAD_HIP(hipMemcpyAsync(d_n, &n32, 4, hipMemcpyHostToDevice, ctx->stream));
    MAD_TRY(mad_describe_device(ctx, f, f, scratch<int32_t>(ctx, S_ROW_COORDS), nullptr, octave, nullptr,
                                scratch<double>(ctx, S_ROW_R), nullptr, d_n, n_rows, d_n + 1, r, scratch<int16_t>(ctx, S_DSC), nullptr, nullptr, dsc_size));
    MAD_HIP(hipMemcpyAsync(dsc, mad_sb(ctx, S_DSC).p, (size_t)n_rows * D * 2, hipMemcpyDeviceToHost, ctx->stream));
    MAD_HIP(hipStreamSynchronize(ctx->stream));
    return MAD_OK;
}
