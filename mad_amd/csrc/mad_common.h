// mad_common.h -- internal declarations shared by the translation units of libmad_amd.so.
// gfx950 only: 64-lane wavefronts are assumed throughout.
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/mad_amd.h"

#define MAD_WAVE 64
#define MAD_ZLUT 1024
#define MAD_MAX_BELT 32

// ---------------------------------------------------------------------------
// device-side tables
// ---------------------------------------------------------------------------

// One EQSP partition as the kernels use it.  Zones of one belt share their phi
// bounds (mad/eqsp/eqsp.py:37-46), so a direction is classified by finding its
// belt first and then testing only that belt's azimuth sectors -- the same
// strict comparisons against the same table values as the reference's scan of
// all zones (Orientator.py:324-334).
// Tables the fast classifier reads per lane, staged in LDS by the kernels.
struct __attribute__((aligned(16))) EqspFastLds {
    float g32[MAD_MAX_Z][4];
    float z_in_lo[MAD_MAX_BELT], z_in_hi[MAD_MAX_BELT], belt_lo0[MAD_MAX_BELT], belt_inv_w[MAD_MAX_BELT];
    int belt_first[MAD_MAX_BELT], belt_count[MAD_MAX_BELT];
    unsigned char zlut[MAD_ZLUT];
    float4 belt_f[MAD_MAX_BELT];      // (z_in_lo, z_in_hi, belt_lo0, belt_inv_w): one 16-byte read per lookup
    int2 belt_i[MAD_MAX_BELT];        // (first zone, zone count)
    // second tier (eqsp_tier2): float64 edge directions of every zone and z thresholds of every belt, with which a
    // direction that is not within MAD_T2_* radians of a bound is classified by four cross products instead of atan2 / acos
    double dir[MAD_MAX_Z][4];         // cos, sin of theta_min; cos, sin of theta_max
    double zthr[MAD_MAX_BELT][4];     // inside the belt needs z < [0] and z > [1] (float64 semantics), z < [2] and z > [3] (float32 semantics)
    int nbelt;
    int tier2_ok;                     // every belt with more than one zone has zones narrower than 3 rad (else tier 2 is off)
    // LAST: the exact float64 bounds the atan2 / acos fallback walks (eqsp_classify_lds; ~1e-8 of the samples get that far).
    // Kernels that stage a prefix of this image -- k_describe's table form up to `dir`, k_orient up to here -- hand the fallback the
    // image in global memory instead.
    double th_lo[MAD_MAX_Z], th_hi[MAD_MAX_Z], ph_lo[MAD_MAX_BELT], ph_hi[MAD_MAX_BELT];
};

// Table classifier of the 4-byte texels (k_describe, descriptor sphere only): belt from z through `zbelt`, zone inside the belt from
// the pseudo-angle p = 1 - x / (|x| + |y|) (y >= 0) or 3 + x / (|x| + |y|) (y < 0), monotonic in theta, through `ptab`.  An entry is
// a zone only where EVERY direction that can land in the bin -- the bin, its two neighbours and MAD_TAB_GUARD radians around them
// (more than twice the 1.7e-3 rad three 10-bit components can be off together) -- lies strictly inside that zone; everything else is 255 =
// "undecided", and the sample goes through the float32 / float64 tiers on its full texel.
#define MAD_TAB_BELTS 4
#define MAD_TAB_PBINS 2048
#define MAD_TAB_ZBINS 2048
#define MAD_TAB_GUARD 4e-3
struct __attribute__((aligned(16))) EqspTabLds {
    unsigned char zbelt[MAD_TAB_ZBINS];
    unsigned char ptab[MAD_TAB_BELTS][MAD_TAB_PBINS];
};

#define MAD_T2_MARGIN64 1e-9      // >> the ~1e-15 rad of a float64 atan2 / acos
#define MAD_T2_MARGIN32 2e-6      // >> the float32 rounding of theta, of theta + float32(2 pi) and of phi (<= 9e-7 rad together)

struct EqspDev {
    int Z;
    int nbelt;
    double th_lo[MAD_MAX_Z];
    double th_hi[MAD_MAX_Z];
    double ph_lo[MAD_MAX_Z];     // per belt
    double ph_hi[MAD_MAX_Z];     // per belt
    int belt_first[MAD_MAX_Z];   // per belt: first zone
    int belt_count[MAD_MAX_Z];   // per belt: number of zones
    double to_dom[MAD_MAX_Z][9];
    double adj_sec[MAD_MAX_Z][9];
    // Guard-banded float32 tables of the fast classifier (eqsp_fast32): a direction that lies at
    // least MAD_EQSP_GUARD radians inside a zone's open (theta, phi) rectangle is classified with a
    // table look-up, a polynomial angle guess and two cross products, all in float32; everything
    // closer to a bound (or to a pole) goes through the exact float64 path (eqsp_classify).
    unsigned char zlut[MAD_ZLUT];   // z bin -> belt holding the bin centre
    float z_in_lo[MAD_MAX_BELT];    // per belt: inside needs z <  z_in_lo  (cos(ph_lo + guard), rounded inwards)
    float z_in_hi[MAD_MAX_BELT];    // per belt: inside needs z >  z_in_hi  (cos(ph_hi - guard), rounded inwards)
    float belt_lo0[MAD_MAX_BELT];   // per belt: theta_min of its first zone
    float belt_inv_w[MAD_MAX_BELT]; // per belt: zones / 2pi
    int belt_first32[MAD_MAX_BELT];
    int belt_count32[MAD_MAX_BELT];
    float g32[MAD_MAX_Z][4];        // per zone: cos, sin of (theta_min + guard), cos, sin of (theta_max - guard)
    EqspFastLds image;              // the LDS copy of the above, byte for byte (mad_set_eqsp builds it, kernels copy it)
    EqspTabLds tab;                 // table classifier of the 4-byte texels (valid when tab_ok)
    int tab_ok;
};

// How far inside a zone a direction must lie for the float32 classifier to decide it (radians of theta and of phi).  What the guard has
// to cover: (1) the slivers of the 4-decimal tables -- a belt's last zone ends at its tabulated theta_max while the first one, reached
// through theta + 2 pi, begins 1.47e-5 rad below it (6.2832 against 2 pi; the same in every belt of both tables, and the only
// overlap there is: tests/test_eqsp.py::test_zone_bounds_overlap_only_at_the_seam pins it) -- inside the guard a direction must match this zone and no other;
// (2) the float32 noise of the inputs, <= 1e-6 rad of direction = up to 5e-6 rad of theta in the collar next to a polar cap (sin phi
// 0.19); (3) the float32 evaluation of the cross products and of z against the inward-rounded bounds, ~3e-7.  Together 2.0e-5; the
// guard is twice that.  (1e-4 until round 4: with 4e-5 a third fewer samples fall through to the float64 queue that ends every row of
// k_describe -- 79.5 -> 77.2 us per launch on C3, k_orient unchanged.)
#ifndef MAD_EQSP_GUARD
#define MAD_EQSP_GUARD 4e-5
#endif

// One octave's gradient field: a texel is {gx, gy, gz, |g|} (|g| in float32 exactly
// as numpy forms it: sqrt((gx*gx + gy*gy) + gz*gz), Orientator.py:139).
struct FieldDev {
    const float4 *tex;
    int nx, ny, nz;
    // The same field as 4-byte texels (round 3): the unit direction in three signed 10-bit components + two flag bits (mad_tex4_encode).
    // k_describe gathers THESE (a quarter of the bytes per sample) and classifies them through a table with a guard band wider than
    // the quantisation; only the samples the table cannot decide (2-4 %) fetch the 16-byte texel.  Behind `tex` in the same allocation.
    const unsigned *tex4;
};

// k_describe_ball takes a base-octave anchor when its ball of texels (radius 13 voxels) lies inside the grid with a voxel to spare:
// the condition under which k_describe knows that no sample of the anchor's rows can leave the grid.  The host sorts anchors with
// this expression (set_upload_anchors), the kernel checks it.
#ifdef __HIPCC__
__host__ __device__
#endif
static inline bool mad_ball_interior(int c0, int c1, int c2, int nx, int ny, int nz) {
    const int reach = 14;
    return c0 - reach >= 1 && c0 + reach <= nx - 2 && c1 - reach >= 1 && c1 + reach <= ny - 2 && c2 - reach >= 1 && c2 + reach <= nz - 2;
}

// ---------------------------------------------------------------------------
// host-side context
// ---------------------------------------------------------------------------

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

enum { MAD_T_ORIENT = 0, MAD_T_DESCRIBE, MAD_T_CORRELATE, MAD_T_PAIRS, MAD_T_POSE, MAD_T_TOPK, MAD_T_REFINE,
       MAD_T_DENSITY, MAD_T_CCC, MAD_T_COUNT };

#define MAD_T_RING 32
#define MAD_LANES 8
#define MAD_BRACKETS 3      // mad_match_topk_many_begin brackets that may be open at once (steps in flight - 1)
#define MAD_SHARD_RING 8     // shard records of one lane that may be on their way to the host at once
#define MAD_RES (2 * MAD_LANES)      // matches of one bracket that may be in flight: two per lane, the second queued behind the first on the lane's stream

struct TimerGroup {
    hipEvent_t start[MAD_T_RING];
    hipEvent_t stop[MAD_T_RING];
    int pending = 0;
    double total_ms = 0.0;
    int64_t launches = 0;
};

struct DensityDev {
    float *grid = nullptr;       // [nx][ny][nz]
    float4 *grad = nullptr;      // np.gradient texels {d/dx, d/dy, d/dz, 0}
    int nx = 0, ny = 0, nz = 0;
    double o[3] = {0, 0, 0};
    double vs = 0;
};

struct MatchState {            // the most recent mad_match_topk call
    int64_t n_pairs = 0;
    int32_t l_hi = 0, l_lo = 0;
    int32_t n_hi_anchors = 0, n_lo_anchors = 0;
    int64_t cap_c = 0;         // capacity hints carried from call to call (elements of the int32 score matrix,
    int64_t cap_pairs = 0;     // number of pairs); the device raises a flag when one is too small
    int lane = 0;              // scratch copy that holds this match's pairs and counts
    // the pose search of that match was pruned by bounds: S_COUNTS holds exact counts only for the pairs that could reach the
    // top k.  mad_match_fetch / mad_match_results complete it on demand (they need the two sets alive and the distance).
    bool pruned = false, fits = true;
    const void *hi = nullptr, *lo = nullptr;      // cleared by mad_set_destroy; hi_gen / lo_gen: the builds those counts belong to
    uint64_t hi_gen = 0, lo_gen = 0;
    double dist = 0;
    int64_t cap_pairs_used = 0;
    int64_t n_sel = 0;         // pairs that went through the exact search (= n_pairs when nothing was pruned)
    // the shard left in lane 0 by mad_match_shard_pairs, consumed by mad_match_shard_topk
    const void *shard_hi = nullptr, *shard_lo = nullptr;
    int64_t shard_begin = 0, shard_end = 0, shard_pairs = 0, shard_cap_pairs = 0;
};

struct ShardAsync {            // a mad_match_shard_begin waiting for its mad_match_shard_score, per lane
    const void *hi = nullptr, *lo = nullptr;
    uint64_t hi_gen = 0, lo_gen = 0;
    int64_t begin = 0, nb = 0, n_lo = 0, cap_pairs = 0;
};

struct mad_ctx {
    int device = -1;
    hipStream_t stream = nullptr;            // the stream of the current lane (lane_stream[lane])
    hipStream_t lane_stream[MAD_LANES] = {};      // [0] is the stream mad_stream() reports
    int next_set_lane = 0;
    bool overlap = true;                      // false: every lane enqueues on lane_stream[0] (kernels run one at a time)
    int ori_queue_cap = 1 << 20, dsc_queue_cap = 1 << 20;      // test hooks (mad_set_option "ori_queue" / "dsc_queue"): caps of the kernels' undecided queues
    int pose_mx = 0;                          // 1: k_pose_bounds_mx for hi clouds of up to 512 points (mad_set_option "pose_mx"; slower on C3, section 6d)
    int pose_split = -1;                      // pose search in two launches, best-scoring pairs first: -1 = for hi clouds of more than 512 points
    int64_t n_device_allocs = 0;              // device buffers (re)allocated so far (mad_reserve; mad_device_allocations): a steady state has none
    int64_t pose_split_min = 4096;            // pose search: pairs in the list bracketed first (at least; mad_set_option "pose_split_min")
    bool batch_gemm = false;                  // mad_match_topk_many: the GEMMs of a bracket's matches in one launch (mad_set_batching)
    bool spatial_order = true;                // the build kernels take anchors / rows in Morton order (MAD_NO_SPATIAL_ORDER: list order)
    char err[512] = {0};
    FieldDev fields[MAD_MAX_FIELDS];
    void *field_mem[MAD_MAX_FIELDS];
    EqspDev *eq[2] = {nullptr, nullptr};     // device copies
    EqspDev eq_host[2];
    bool eq_set[2] = {false, false};
    double gw_sig = 0.0, gw_built = 0.0;     // Orientator(gw_sig): Gaussian window on the orientation histogram (0 = none); the table's sigma
    unsigned long long *gw_tab = nullptr;    // device: its weights by squared offset, 2^-50 fixed point, for box size gw_r
    int gw_r = -1;
    int8_t *mask_off = nullptr;              // sphere-mask offsets for the current r
    int mask_r = -1;
    int mask_n = 0;
    unsigned *ball_colinfo = nullptr;        // k_describe_ball: LDS offset and half-length of every (x, y) column of the base-octave sample ball (device)
    bool dsc_ball = false;                   // mad_set_option "dsc_ball" / MAD_BALL=1: base-octave rows described anchor by anchor from a ball of texels in LDS
                                             // (k_describe_ball; measured slower than k_describe on C3 / C4 / C5: DESIGN.md section 6d -- off by default)
    // named grow-only scratch buffers
    DevBuf scratch[64 * MAD_LANES];   // MAD_LANES independent copies: matches in flight do not share scratch
    int lane = 0;                    // the copy the current call works in
    // two result slots per lane: one per open mad_match_topk_many bracket, so that a second batch of matches can be enqueued
    // (and deliver into its own pinned staging) before the first one has been collected
    hipEvent_t lane_done[MAD_BRACKETS][MAD_RES];   // recorded behind the last operation of a bracket's match (result index: lane, or MAD_LANES + lane for a lane's second match)
    hipEvent_t lane_pre[MAD_LANES];     // mad_match_topk_many: a lane is ready for the bracket's common GEMM
    hipEvent_t gemm_done[MAD_BRACKETS];            // ... and that GEMM has been enqueued (per open bracket)
    void *host_res[MAD_BRACKETS][MAD_RES] = {};     // pinned staging of a match's results / indices / status
    size_t host_res_cap[MAD_BRACKETS][MAD_RES] = {};
    int res_slot = 0, res_idx = 0;       // the (bracket, result index) the match calls below read and write
    // host pinned staging for small read-backs
    int64_t *pinned = nullptr;     // 1024 slots: [16 * lane ..] read-backs of the lane, [64..] two per mad_set
    int next_pinned = 64;
    DensityDev dens;
    MatchState match;
    ShardAsync shard_async[MAD_LANES];
    // mad_match_shard_collect / _wait: a ring of pinned staging buffers and completion events per lane (steps in flight)
    void *shard_host[MAD_LANES][MAD_SHARD_RING] = {};
    size_t shard_host_cap[MAD_LANES][MAD_SHARD_RING] = {};
    hipEvent_t shard_ev[MAD_LANES][MAD_SHARD_RING] = {};
    bool shard_busy[MAD_LANES][MAD_SHARD_RING] = {};
    int shard_next[MAD_LANES] = {};
    int last_pose_kernel = -1;               // 0 k_pose_lds, 1 k_pose_lds32, 2 k_pose (mad_last_pose_kernel)
    int64_t lane_sel_hint[MAD_LANES] = {};   // pairs the last pruned match of a lane sent to the exact search (sizes the next launch)
    void *many[MAD_BRACKETS] = {};           // open mad_match_topk_many_begin brackets (ManyState, mad_match.hip), by result slot: a ring
    int many_oldest = 0, many_open = 0;      // the slot _finish collects next, and how many are open
    bool timing = false;
    TimerGroup timers[MAD_T_COUNT];
    int n_cu = 256;
};

struct mad_set {
    int32_t n_anchors = 0;
    uint64_t gen = 0;            // counts the builds / loads / imports of this set: results that refer to an earlier one are stale
    int64_t cap_rows = 0;        // capacity of the row buffers (rows are produced on the device; see dev_n)
    int D = 0;
    // per anchor: views into anc_blob = [dev_n 64 B][subv n x 3 f64][coords n x 3 i32][octave n i32][index n i32], which one
    // copy from the pinned staging buffer fills (a copy from pageable memory would block the host on the stream)
    DevBuf anc_blob;
    DevBuf anc_coords, anc_octave, anc_subv, anc_index, anc_canon;      // views into anc_blob
    DevBuf anc_order;            // view: the anchors in spatial (octave, Morton) order -- the order the build kernels WORK in, so that
                                 // workgroups running side by side sample neighbouring texels (the rows keep the reference's order)
    void *host_stage = nullptr;
    size_t host_stage_cap = 0;
    int staged_n = -1;                  // what the staging buffer (and the device blob) hold: anchor count, with / without voxel coordinates
    bool staged_coords = false;
    hipEvent_t uploaded = nullptr;      // recorded behind the staging copy: the buffer may be rewritten after it
    int lane = 0;                       // the lane (stream + scratch) this set is built on
    hipEvent_t built = nullptr;         // recorded behind the last kernel of a build / load: consumers on other lanes wait for it
    // per row
    DevBuf row_anchor, row_main, row_sec, row_R, row_Rinv, row_meta, dsc, dsc8, norm;
    DevBuf row_perm;             // k-th row in working order (rows of spatially neighbouring anchors next to each other)
    DevBuf row_rec;              // DscRowRec of the k-th row in working order
    DevBuf anc_rows;             // per anchor IN WORKING ORDER: MAD_ANCROW_WORDS ints {position of its first row in working order, rows, voxel coordinates} (k_orient_rows* write it)
    int n_rowwise = 0;           // anchors k_describe takes row by row (octave 0, and base-octave ones near the border): they come first in
                                 // working order; the others go through k_describe_ball
    int ball_dims[3] = {0, 0, 0};      // the base-octave grid the anchors were sorted for (0: no anchor was set apart)
    int last_fan = 0;            // lim_main x lim_sec of the last build
    DevBuf dev_n;                // view: int32[4] on the device = {rows, rows out of int8 range, rejects, describe overflow}
    int64_t n_rows_host = -1;    // host copy of dev_n[0]; -1 until the asynchronous read-back has been waited for
    bool range_bad = false;      // a loaded row held a count outside the int8 range
    int64_t rows_hint = 0;       // row count of the previous build of this set (sizes the describe launch)
    // what mad_set_build needs to repeat the describe stage when the hint was too small
    FieldDev last_f[2];
    int last_r = 0;
    bool last_perm = false, last_rec = false;
    int pinned_slot = 0;
    hipEvent_t ready = nullptr;  // recorded behind mad_set_export: a rebuild on another lane (mad_set_build_many) waits for that read
    // cell list over ALL anchors (cell = dist), only built when a cloud does not fit LDS
    DevBuf cell_start, cell_pts, cell_ids;
    double cell_min[3] = {0, 0, 0};
    double bb_min[3] = {0, 0, 0}, bb_max[3] = {0, 0, 0};      // bounding box of all anchors
    double cell_size = 0;
    int cell_dim[3] = {0, 0, 0};
    bool cells_ready = false;
};

// Several jobs (the anchor lists of several structures) in ONE grid: job j owns the workgroups first[j] .. first[j + 1] - 1.
// The tables travel in the kernel arguments, so a workgroup finds its job with a few scalar compares and reads that job's
// arguments with scalar loads.
#define MAD_BATCH_MAX 16
template <class Args> struct Batch {
    int n_jobs;
    int first[MAD_BATCH_MAX + 1];
    Args job[MAD_BATCH_MAX];
};
template <class Args> __device__ __forceinline__ int batch_job(const Batch<Args> &B, int block) {
    int j = 0;
    while (j + 1 < B.n_jobs && block >= B.first[j + 1]) j++;
    return j;
}

// scratch slots
enum {
    S_COORDS = 0, S_OCT, S_SLOT_CNT, S_SLOT_MAIN, S_SLOT_SEC, S_SLOT_HIST, S_ROW_OFF, S_SCAN_TMP,
    S_ROW_ANCHOR, S_ROW_MAIN, S_ROW_SEC, S_ROW_R, S_ROW_COUNT, S_DSC, S_ROW_COORDS,
    S_HI16, S_LO16, S_HI8, S_LO8, S_HNORM, S_LNORM, S_CMAT, S_ROWCNT, S_ROWOFF,
    S_PAIR_HI, S_PAIR_LO, S_PAIR_SCORE, S_COUNTS, S_USED_HI, S_USED_LO, S_HI_CLOUD, S_MISC,
    S_HIST, S_SEL, S_RESULTS, S_TMP_A, S_TMP_B, S_TMP_C, S_TMP_D, S_TMP_E, S_TMP_F, S_TMP_G,
    S_TIE_FLAG, S_TIE_OFF, S_SEL_OUT, S_TMP_H, S_TMP_I, S_TMP_J,
    S_CELL_START, S_CELL_PTS, S_CELL_IDS, S_PG_START, S_PG_PTS, S_PG_PTSF, S_ZERO, S_CMASK, S_PG_BITS, S_PG_PAIRS, S_PERM_OFF, S_CFLAG, S_N_SLOTS
};
static_assert(S_N_SLOTS <= 64, "grow mad_ctx::scratch");

int mad_fail(mad_ctx *ctx, int code, const char *fmt, ...);
int mad_reserve(mad_ctx *ctx, DevBuf &b, size_t bytes);            // grow-only device buffer
void mad_release(DevBuf &b);
int mad_field_alloc(mad_ctx *ctx, int slot, int nx, int ny, int nz, size_t *n_texels);      // (re)allocates the texels of a field slot
static inline DevBuf &mad_sb(mad_ctx *ctx, int slot) { return ctx->scratch[ctx->lane * 64 + slot]; }
// A lane = one HIP stream + one set of scratch buffers.  Independent pieces of work (the builds of different
// sets, the matches of one mad_match_topk_many call) go to different lanes and overlap on the device.
static inline void mad_use_lane(mad_ctx *ctx, int lane) { ctx->lane = lane; ctx->stream = ctx->lane_stream[ctx->overlap ? lane : 0]; }
template <class T> static inline T *scratch(mad_ctx *ctx, int slot) { return (T *)mad_sb(ctx, slot).p; }

void mad_timer_begin(mad_ctx *ctx, int group);
void mad_timer_end(mad_ctx *ctx, int group);

// exclusive prefix sum of n int32 on the ctx stream; out[n] receives the total (out has n+1 entries)
int mad_scan_i32(mad_ctx *ctx, const int32_t *in, int32_t *out, int64_t n);

// single-launch exclusive scan for n <= 65536 with the length read on the device; out[*n] = total, also to *total_out
void mad_scan_small(mad_ctx *ctx, const int32_t *in, int32_t *out, const int32_t *d_n, int32_t *total_out, int n_host = 0);

// What k_describe starts a row from, as one record in WORKING order (the k-th record belongs to the k-th row the kernel takes):
// written with the rows by k_orient_rows*.  k_describe fetches its first 128 bytes with one vector load; k_describe_ball reads the
// float32 copies behind them with scalar loads (they are what every thread of k_describe forms from the float64 values).
struct alignas(16) DscRowRec {
    int32_t row;           // the row (where its descriptor goes)
    int32_t c[3];          // voxel coordinates of its anchor
    int32_t octave;
    int32_t pad0[3];
    double inv[9];         // inv(Rfinal) (np.linalg.inv of Descriptor.py:132, by cofactors)
    int32_t pad1[6];
    float hf[9];           // (float)inv[i]
    float rf[9];           // (float)Rfinal[i], the third row times 1 / 511 (the scale of a 4-byte texel's components)
    float ru[3];           // (float)Rfinal[6 .. 8] as they are (the tiers on the 16-byte texel)
    int32_t pad2[3];
};
#define MAD_ROWREC_WORDS ((int)(sizeof(DscRowRec) / 4))
#define MAD_ANCROW_WORDS 8      // mad_set::anc_rows, per anchor in working order: {position of its first row, rows, voxel coordinates x 3, -, -, -}
static_assert(sizeof(DscRowRec) == 224 && offsetof(DscRowRec, hf) == 128, "k_describe reads the first 32 dwords of a DscRowRec");

// implemented in mad_orient.hip; used by the set API in mad_match.hip.  Both are fully asynchronous.
struct OrientOut {
    int32_t *row_anchor, *row_main, *row_sec;
    double *row_R;
    int32_t *row_count;      // nullable: Z quantised counts per row
    // nullable extras of the set pipeline, written with the rows: inv(Rfinal) and {anchor index, octave, main bin}
    double *row_Rinv = nullptr;
    int32_t *row_meta = nullptr;
    const int32_t *anc_index = nullptr, *anc_octave = nullptr;
    // nullable: the anchors in working order (a permutation of 0 .. n-1) and, written with the rows, the rows in that order
    const int32_t *anc_order = nullptr;
    int32_t *row_perm = nullptr;
    DscRowRec *row_rec = nullptr;      // nullable: the rows' records for k_describe, in working order
    int32_t *anc_rows = nullptr;       // nullable (with anc_order): per anchor in working order {first row position, rows}
    bool counters_zeroed = false;      // the caller has already enqueued the zeroing of d_n_rows / d_n_reject
    int32_t *d_n_rows;       // device: number of rows produced
    int32_t *d_n_reject;     // device, nullable: anchors refused at the border
};
int mad_orient_device(mad_ctx *ctx, FieldDev f0, FieldDev f1, const int32_t *d_coords, const int32_t *d_octave,
                      int uniform_octave, int n, int r, int lim_main, int lim_sec, OrientOut out);
// The same for the anchor lists of several structures in one k_orient grid (+ one scan, one row expansion), and their
// descriptors in one k_describe grid: a step's structures fill the chip together instead of one small launch each.
struct OrientJob {
    FieldDev f[2];
    const int32_t *d_coords, *d_octave;
    int uniform_octave;
    int n;
    OrientOut out;
};
struct DescribeJob {
    FieldDev f[2];
    const int32_t *d_anc_coords, *d_anc_octave;
    int uniform_octave;
    const int32_t *d_row_anchor;
    const double *d_row_R, *d_row_Rinv;
    const int32_t *d_row_perm = nullptr;      // nullable: workgroup k describes row d_row_perm[k]
    const DscRowRec *d_row_rec = nullptr;     // nullable: the same rows' records (k_orient_rows* wrote them), in that order
    // nullable: {first row position, rows} per anchor in working order; with it, n_anchors, n_rowwise (the anchors k_describe takes: the
    // first ones in working order) and fan (rows per anchor at most), the other anchors -- base octave, interior -- go through k_describe_ball
    const int32_t *d_anc_rows = nullptr;
    int n_anchors = 0, n_rowwise = 0, fan = 0;
    const int32_t *d_n_rows;
    int64_t grid_rows;
    int32_t *d_overflow;
    int16_t *d_dsc;
    int8_t *d_dsc8;
    double *d_norm;
};
int mad_orient_device_many(mad_ctx *ctx, int n_jobs, const OrientJob *jobs, int r, int lim_main, int lim_sec);
int mad_describe_device_many(mad_ctx *ctx, int n_jobs, const DescribeJob *jobs, int r, int dsc_size = 64);
int mad_describe_device(mad_ctx *ctx, FieldDev f0, FieldDev f1, const int32_t *d_anc_coords, const int32_t *d_anc_octave,
                        int uniform_octave, const int32_t *d_row_anchor, const double *d_row_R, const double *d_row_Rinv,
                        const int32_t *d_n_rows, int64_t grid_rows, int32_t *d_overflow, int r, int16_t *d_dsc, int8_t *d_dsc8 = nullptr,
                        double *d_norm = nullptr, int dsc_size = 64);
void mad_many_abandon(mad_ctx *ctx);
void mad_zero_words(mad_ctx *ctx, void *p, size_t bytes);              // one-launch zero fill (bytes rounded up to 16)
void mad_zero_words3(mad_ctx *ctx, void *p, size_t bytes_p, void *q, size_t bytes_q, void *r, size_t bytes_r);      // the same for up to three regions, still one launch
void mad_copy_words(mad_ctx *ctx, void *dst, const void *src, size_t bytes);      // kernel copy, e.g. out of pinned host memory
int mad_build_cells(mad_ctx *ctx, mad_set *set, double cell);

#define MAD_HIP(call)                                                                            \
    do {                                                                                         \
        hipError_t e__ = (call);                                                                 \
        if (e__ != hipSuccess)                                                                   \
            return mad_fail(ctx, MAD_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)

#define MAD_TRY(call)            \
    do {                         \
        int rc__ = (call);       \
        if (rc__ != MAD_OK) return rc__; \
    } while (0)

static inline int64_t mad_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------
#ifdef __HIPCC__

// Wave-wide integer collectives on the DPP data path (row shifts, row broadcasts: vector-ALU instructions) rather than
// ds_bpermute: a shuffle is an LDS instruction, and six dependent ones queue behind whatever the workgroup has in flight there
// (3 500 cycles for the scan of k_describe's queue reservation, measured with MAD_PROBE_STAMPS).  Integer sums and maxima are
// order-independent, so the results are those of the shuffle forms.
template <int CTRL, int ROW_MASK, bool BOUND> __device__ __forceinline__ int dpp_i32(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, BOUND);      // lanes without a source (or masked out) read 0
}
// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ int wave_incl_scan_i32(int v) {
    v += dpp_i32<0x111, 0xf, true>(v);      // row_shr:1
    v += dpp_i32<0x112, 0xf, true>(v);      // row_shr:2
    v += dpp_i32<0x114, 0xf, true>(v);      // row_shr:4
    v += dpp_i32<0x118, 0xf, true>(v);      // row_shr:8   -- inclusive within each row of 16
    v += dpp_i32<0x142, 0xa, false>(v);     // row_bcast:15 into rows 1 and 3
    v += dpp_i32<0x143, 0xc, false>(v);     // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ int wave_sum_i32(int v) {
    return __builtin_amdgcn_readlane(wave_incl_scan_i32(v), MAD_WAVE - 1);
}
__device__ __forceinline__ int wave_max_i32(int v) {      // (values >= INT_MIN: lanes without a source contribute the lane's own value)
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false));
    return __builtin_amdgcn_readlane(v, MAD_WAVE - 1);
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, MAD_WAVE);
    return v;
}
__device__ __forceinline__ double wave_max_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, MAD_WAVE));
    return v;
}
__device__ __forceinline__ unsigned lane_id() { return threadIdx.x & (MAD_WAVE - 1); }
__device__ __forceinline__ unsigned long long lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// Block-wide exclusive scan of one int per thread (blockDim.x multiple of 64, <= 1024).
// `warp_tot` must hold blockDim.x/64 + 1 ints of LDS.  Returns the exclusive prefix;
// *total gets the block sum.
__device__ __forceinline__ int block_excl_scan(int v, int *warp_tot, int *total) {
    const int lane = lane_id(), w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int inc = wave_incl_scan_i32(v);
    if (lane == MAD_WAVE - 1) warp_tot[w] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < nw; i++) { int t = warp_tot[i]; warp_tot[i] = run; run += t; }
        warp_tot[nw] = run;
    }
    __syncthreads();
    int res = inc - v + warp_tot[w];
    *total = warp_tot[nw];
    __syncthreads();
    return res;
}

// Classify a direction given as (theta, theta + 2pi, phi) and call f(zone) for every
// zone whose strict bounds contain it (Orientator.py:328-331).  Zones can overlap by
// a sliver next to a sector that wraps past 2pi (the tables are rounded to 4
// decimals), so more than one call is possible; callers that need "last match wins"
// (Descriptor.py:187) keep the last.
template <class F>
__device__ __forceinline__ void eqsp_classify(const EqspDev *t, double th, double sth, double ph, F &&f) {
    const int nb = t->nbelt;
    for (int b = 0; b < nb; b++) {
        if (ph < t->ph_hi[b] && ph > t->ph_lo[b]) {
            const int a0 = t->belt_first[b], a1 = a0 + t->belt_count[b];
            for (int a = a0; a < a1; a++) {
                const double lo = t->th_lo[a], hi = t->th_hi[a];
                if ((th < hi && th > lo) || (sth < hi && sth > lo)) f(a);
            }
            break;
        }
    }
}

#define MAD_TWO_PI 6.283185307179586476925286766559

__device__ __forceinline__ void mad_mat3_inv(const double *m, double *o) {      // cofactors (np.linalg.inv, MaD.py:438)
    const double c00 = m[4] * m[8] - m[5] * m[7];
    const double c01 = m[5] * m[6] - m[3] * m[8];
    const double c02 = m[3] * m[7] - m[4] * m[6];
    const double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    const double id = 1.0 / det;
    o[0] = c00 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c01 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c02 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

// Rfinal = adj_sec_mat @ to_dom_mat (Orientator.py:105) from the table matrices of (main bin, secondary bin).  One
// expression, used by k_orient_rows and by the import of rows built on another GPU (mad_set_import), so that both
// produce the same bits.
__device__ __forceinline__ void mad_rfinal(const EqspDev *eq, int mb, int sb, double *o) {
    const double *A = eq->adj_sec[sb], *B = eq->to_dom[mb];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) o[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

// Copies `bytes` (a multiple of 4) from 16-byte-aligned global memory to 16-byte-aligned LDS with the whole workgroup:
// 16 bytes per lane and four requests in flight per lane, because a workgroup that waits for one 4-byte load per lane
// per trip spends tens of microseconds filling its tables.
__device__ __forceinline__ void stage_lds(void *lds_dst, const void *src, size_t bytes) {
    const uint4 *s4 = (const uint4 *)src;
    uint4 *d4 = (uint4 *)lds_dst;
    const int n4 = (int)(bytes >> 4), T = (int)blockDim.x, last = n4 - 1;
    // loads at clamped (always valid) addresses, stores predicated: four independent requests per trip, all in registers
    for (int i = threadIdx.x; i < n4; i += 4 * T) {
        const uint4 v0 = s4[i], v1 = s4[min(i + T, last)], v2 = s4[min(i + 2 * T, last)], v3 = s4[min(i + 3 * T, last)];
        d4[i] = v0;
        if (i + T < n4) d4[i + T] = v1;
        if (i + 2 * T < n4) d4[i + 2 * T] = v2;
        if (i + 3 * T < n4) d4[i + 3 * T] = v3;
    }
    const int tail = (int)((bytes & 15) >> 2);
    if ((int)threadIdx.x < tail) ((unsigned *)lds_dst)[4 * n4 + threadIdx.x] = ((const unsigned *)src)[4 * n4 + threadIdx.x];
}

// floor / round-half-up of a float32 straight to int32: one instruction each on gfx950 where floorf + a cast take two or three.
// cvt_round is floor(x + 0.5) with the sum rounded to float32: exact except within an ulp of a tie, which every caller treats
// as undecided anyway.  Both saturate (NaN -> 0).
__device__ __forceinline__ int cvt_floor(float x) {
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
// a * b + c with a, b < 2^24: one full-rate instruction (the compiler, unsure of the ranges, otherwise picks a 64-bit multiply-add)
__device__ __forceinline__ unsigned mad_u24(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ int mad_i24(int a, int b, int c) {      // the same for signed values within 24 bits
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ int cvt_round(float x) {
    int r;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

__device__ __forceinline__ void eqsp_fast_stage(const EqspDev *t, EqspFastLds *l) { stage_lds(l, &t->image, sizeof(EqspFastLds)); }

// 4-byte texel of a gradient {x, y, z} with magnitude w (float32, as in the 16-byte texel): bits 0-9 / 10-19 / 20-29 = the unit
// direction's components x 511, rounded, as signed 10-bit integers; bits 30-31: 0 = a direction, 3 = |g| < 1e-5 (the sample is not
// counted, Descriptor.py:190), 2 = not finite (always decided by the exact tiers).
__device__ __forceinline__ unsigned mad_tex4_encode(float x, float y, float z, float w) {
    if (!(fabsf(x) <= 3.0e38f) || !(fabsf(y) <= 3.0e38f) || !(fabsf(z) <= 3.0e38f) || !(w <= 3.0e38f)) return 0x80000000u;
    if (w < 1e-5f) return 0xC0000000u;
    const float inv = 511.0f / w;
    const int ix = min(max((int)rintf(x * inv), -511), 511), iy = min(max((int)rintf(y * inv), -511), 511), iz = min(max((int)rintf(z * inv), -511), 511);
    return ((unsigned)ix & 0x3ffu) | (((unsigned)iy & 0x3ffu) << 10) | (((unsigned)iz & 0x3ffu) << 20);
}

// zone of the direction (x, y, z) -- z on the unit scale, x and y on any common scale -- or -1 (undecided); see EqspTabLds
__device__ __forceinline__ int eqsp_tab32(const EqspTabLds *t, float x, float y, float z) {
    const int bz = min(max(cvt_floor(fmaf(z, 0.5f * MAD_TAB_ZBINS, 0.5f * MAD_TAB_ZBINS)), 0), MAD_TAB_ZBINS - 1);
    const int b = t->zbelt[bz];
    const float xr = x * __builtin_amdgcn_rcpf(fmaxf(fabsf(x) + fabsf(y), 1e-30f));
    // p = 1 - xr (y >= 0) or 3 + xr (y < 0) = 2 - copysign(1 + xr, y); scaled to bins: 2 B/4 - copysign((1 + xr) B/4, y)
    const float u = copysignf(fmaf(xr, 0.25f * MAD_TAB_PBINS, 0.25f * MAD_TAB_PBINS), y);
    const int bp = min(max(cvt_floor(0.5f * MAD_TAB_PBINS - u), 0), MAD_TAB_PBINS - 1);
    const int zn = t->ptab[b & (MAD_TAB_BELTS - 1)][bp];
    return (b | zn) >= 255 ? -1 : zn;      // (both are bytes, zones stay below 128: 255 in either means undecided)
}

// eqsp_classify on the LDS copy of the table
template <class F>
__device__ __forceinline__ void eqsp_classify_lds(const EqspFastLds *t, double th, double sth, double ph, F &&f) {
    const int nb = t->nbelt;
    for (int b = 0; b < nb; b++) {
        if (ph < t->ph_hi[b] && ph > t->ph_lo[b]) {
            const int a0 = t->belt_first[b], a1 = a0 + t->belt_count[b];
            for (int a = a0; a < a1; a++) {
                const double lo = t->th_lo[a], hi = t->th_hi[a];
                if ((th < hi && th > lo) || (sth < hi && sth > lo)) f(a);
            }
            break;
        }
    }
}

// atan2 in [0, 2pi) to ~2e-6 rad (odd minimax polynomial on [0, 1] + octant unfolding): a GUESS only.
// v_rcp_f32 (1 ulp) and fused multiply-adds on purpose: nothing here decides a zone, it only proposes one.
__device__ __forceinline__ float approx_angle(float x, float y) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float q = mn * __builtin_amdgcn_rcpf(fmaxf(mx, 1e-30f));
    const float q2 = q * q;
    float a = fmaf(q2, -0.011769974f, 0.052822195f);
    a = fmaf(q2, a, -0.116649978f);
    a = fmaf(q2, a, 0.193669885f);
    a = fmaf(q2, a, -0.332655430f);
    a = fmaf(q2, a, 0.999979854f);
    a *= q;
    a = ay > ax ? 1.57079637f - a : a;
    a = x < 0.f ? 3.14159274f - a : a;
    a = y < 0.f ? 6.28318548f - a : a;
    return a;
}

// Fast float32 classification of the (approximately unit) direction (x, y, z), z = cos(phi).
// Returns the one zone that contains it with a margin of MAD_EQSP_GUARD on every side, or -1 when
// it is closer than that to a bound, near a pole, or not finite; the caller then runs the exact
// test.  Inputs may carry float32 rounding noise (<= ~1e-6 rad): inside the margin the exact
// float64 test on the un-noised direction gives this same zone and no other, because noise and
// the overlap of neighbouring zones (1.47e-5 rad at a belt's seam) together are half the guard (see MAD_EQSP_GUARD).
// Branch-free on purpose (three dependent LDS reads, everything else selects): callers classify several
// points in a row, and straight-line code lets the scheduler overlap their table reads.
// FLAT: the final test written without short-circuit operators, so that no control flow appears and the table reads of
// several points overlap (k_orient: -3 %); in k_describe, which sits at its register limit, the overlap spills (2x slower),
// so it keeps the short-circuit form.
template <bool FLAT = false>
__device__ __forceinline__ int eqsp_fast32(const EqspFastLds *l, float x, float y, float z) {
    const int bin = min(max((int)((z + 1.0f) * (0.5f * MAD_ZLUT)), 0), MAD_ZLUT - 1);
    const int b = l->zlut[bin];
    const float4 bf = l->belt_f[b];
    const int2 bi = l->belt_i[b];
    const bool in_belt = z < bf.x && z > bf.y;
    float u = approx_angle(x, y) - bf.z;
    u = u < 0.f ? u + 6.28318548f : u;
    const int a = bi.x + min(max((int)(u * bf.w), 0), bi.y - 1);
    const float4 g = *(const float4 *)l->g32[a];
    const float c1 = fmaf(g.x, y, -(g.y * x));      // > 0: counter-clockwise of theta_min + guard
    const float c2 = fmaf(x, g.w, -(y * g.z));      // > 0: clockwise of theta_max - guard
    const bool ok = FLAT ? (in_belt & ((bi.y == 1) | ((c1 > 0.f) & (c2 > 0.f))))
                         : (in_belt && (bi.y == 1 || (c1 > 0.f && c2 > 0.f)));      // a polar cap spans every azimuth
    return ok ? a : -1;
}

// Second tier of the exact classification.  The reference decides zone membership by comparing atan2 / arccos values with
// the table's bounds; wherever the direction (x, y, z) is at least `margin` radians away from every bound that could
// matter, the same decision follows from the signs of cross products with the zone's edge directions and from z against
// cos(bound) -- no transcendental, ~60 instructions instead of ~400, which matters because the few directions that reach
// the exact path are processed by a handful of lanes while the rest of the workgroup waits at a barrier.
//   sem32 = false: the comparison the reference makes in float64 (after a rotation);  margin MAD_T2_MARGIN64
//   sem32 = true : angles rounded to float32 first (the Orientator's first pass);      margin MAD_T2_MARGIN32
// Returns true when decided: f(zone) has then been called for every matching zone, in ascending order (a direction inside
// the sliver where two zones' rounded bounds overlap matches both, as in the reference).  Returns false -- and has
// called nothing -- when some relevant bound is within the margin, |z| >= 1, or the input is not finite: the caller
// then runs the transcendental path, which is the definition.
template <class F>
__device__ __forceinline__ bool eqsp_tier2(const EqspFastLds *t, double x, double y, double z, bool sem32, F &&f) {
    if (!t->tier2_ok || !(z < 1.0) || !(z > -1.0) || !(fabs(x) <= 2.0) || !(fabs(y) <= 2.0)) return false;      // also catches NaN
    const double margin = sem32 ? MAD_T2_MARGIN32 : MAD_T2_MARGIN64;
    const int o = sem32 ? 2 : 0;
    // belt: the one the z look-up proposes or a neighbour (belts share their bounds: inside one by the margin means outside all others)
    const float zf = (float)z;
    const int bin = min(max((int)((zf + 1.0f) * (0.5f * MAD_ZLUT)), 0), MAD_ZLUT - 1);
    const int bg = t->zlut[bin];
    int b = -1;
    for (int c = max(bg - 1, 0); c <= min(bg + 1, t->nbelt - 1); c++)
        if (z < t->zthr[c][o] && z > t->zthr[c][o + 1]) b = c;
    if (b < 0) return false;
    const int first = t->belt_first[b], cnt = t->belt_count[b];
    if (cnt == 1) { f(first); return true; }      // a polar cap spans every azimuth (theta' = theta + 2 pi covers theta = 0)
    // zone: the one the angle guess proposes and its two neighbours in the belt (cyclic)
    float u = approx_angle((float)x, (float)y) - t->belt_lo0[b];
    u = u < 0.f ? u + 6.28318548f : u;
    const int k0 = min(max((int)(u * t->belt_inv_w[b]), 0), cnt - 1);
    int za[3];      // matching zone of candidate d - 1, or "none" (constant indices only: stays in registers)
    const int none = 1 << 20;
#pragma unroll
    for (int d = -1; d <= 1; d++) {
        za[d + 1] = none;
        if (cnt == 2 && d == 1) continue;      // two zones: k0 - 1 and k0 + 1 are the same neighbour
        int k = k0 + d;
        k = k < 0 ? k + cnt : (k >= cnt ? k - cnt : k);
        const int a = first + k;
        const double c1 = t->dir[a][0] * y - t->dir[a][1] * x;      // r sin(theta - theta_min)
        const double c2 = x * t->dir[a][3] - y * t->dir[a][2];      // r sin(theta_max - theta)
        if (c1 > margin && c2 > margin) za[d + 1] = a;              // inside by the margin
        else if (!(c1 < -margin || c2 < -margin)) return false;     // not outside by the margin either: undecided
    }
    // zones further away cannot match: the guess is good to 2e-6 rad and a zone is at least two orders wider than the overlap.
    // ascending zone order (three-element sorting network)
    int s0 = min(za[0], za[1]), s1 = max(za[0], za[1]), s2 = za[2];
    const int u1 = min(s1, s2); s2 = max(s1, s2); s1 = max(s0, u1); s0 = min(s0, u1);
    if (s0 != none) f(s0);
    if (s1 != none) f(s1);
    if (s2 != none) f(s2);
    return true;
}

#endif  // __HIPCC__
