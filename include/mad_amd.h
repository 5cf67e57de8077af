/*
 * mad_amd.h -- C-ABI of libmad_amd.so, the MI355X (gfx950) implementation of the
 * MaD anchor-matching hot path.
 *
 * The reference (LBM-EPFL/MaD) has no FFI: its boundary is a set of Python
 * methods.  Each entry point below replaces the *body* of one of them; the
 * Python mirror in mad_amd/ keeps the reference's names and signatures and calls
 * these through ctypes.  See INTEGRATION.md for the binding a reference
 * maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success or a negative MAD_E* code; nothing
 *     throws across the ABI; mad_last_error(ctx) gives the message;
 *   - unless a parameter says "device", buffers are caller-owned HOST memory
 *     (numpy arrays) and the call is synchronous on return;
 *   - the library owns only what lives inside a mad_ctx / mad_set;
 *   - one ctx per GPU; a ctx is not thread-safe, distinct ctxs may be driven
 *     from distinct threads;
 *   - 3x3 matrices are row-major double[9]; volumes are [nx][ny][nz], z fastest;
 *   - variable-length outputs take a capacity; on overflow the call returns
 *     MAD_ENOSPC and stores the needed size in the count argument.
 *
 * File:line references are into the reference checkout (mad/...).
 */
#ifndef MAD_AMD_H
#define MAD_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MAD_OK        0
#define MAD_EINVAL   -22   /* bad argument */
#define MAD_ENOMEM   -12   /* device allocation failed */
#define MAD_ENOSPC   -28   /* output capacity too small */
#define MAD_ENODEV   -19   /* no usable gfx950 device */
#define MAD_EDOM     -33   /* input outside the supported range */
#define MAD_EHIP     -5    /* HIP runtime error */

#define MAD_MAX_Z      128 /* zones per EQSP table */
#define MAD_MAX_FIELDS 64  /* gradient-field slots per ctx */
#define MAD_RESULT_COLS 23 /* row width of MaD._match_dsc results, MaD.py:451 */

typedef struct mad_ctx mad_ctx;
typedef struct mad_set mad_set;   /* device-resident oriented-anchor rows of one structure */

/* ---- context ----------------------------------------------------------- */

int mad_init(int device, mad_ctx **out);
void mad_destroy(mad_ctx *ctx);
const char *mad_last_error(const mad_ctx *ctx);   /* ctx may be NULL: last init error */
int mad_synchronize(mad_ctx *ctx);
/*
 * A ctx owns 8 "lanes" (a HIP stream + its scratch buffers each).  Independent work -- the builds of different
 * mad_sets, the matches of one mad_match_topk_many call -- is enqueued on different lanes and overlaps on the
 * device; every entry point still returns only results that are complete.  mad_stream returns the stream of
 * lane 0 (the one the stage API uses), for callers that time with events.
 * mad_set_overlap(ctx, 0) makes every lane enqueue on that one stream, so kernels run one at a time: the mode in
 * which the duration of a launch is a property of the kernel alone (profiling, per-kernel rooflines).
 */
void *mad_stream(mad_ctx *ctx);
int mad_set_overlap(mad_ctx *ctx, int on);
/*
 * Per-kernel-group timing with HIP events recorded on the ctx stream around each launch.
 * Groups: "orient", "describe", "correlate", "pairs", "pose", "topk", "refine", "density", "ccc".
 * Off by default; mad_timing_get drains pending events (a sync) and returns the accumulated
 * device time and launch count since the last reset; mad_last_ms = total / launches (<0 if none).
 */
int mad_timing_enable(mad_ctx *ctx, int on);
int mad_timing_reset(mad_ctx *ctx);
int mad_timing_get(mad_ctx *ctx, const char *what, double *total_ms, int64_t *launches);
double mad_last_ms(mad_ctx *ctx, const char *what);

/*
 * Measurement aid (bench.py, SURVEY.md 8(d) "confirm with a device copy microbenchmark in the same run"): the rate of a
 * streaming device copy of 1 GiB (read + write bytes per second, GB/s) and the issue rate of v_mfma_i32_16x16x64_i8 with
 * register operands (the instruction of the correlation kernel, int8 TOP/s), measured on this ctx's device now.  Either
 * pointer may be NULL.  No reference counterpart.
 */
int mad_probe_peaks(mad_ctx *ctx, double *copy_gbs, double *i8_tops);

/*
 * EQSP zone tables.  which = 0: orientation sphere (Orientator.py:16, 112 zones);
 * which = 1: descriptor sphere (Descriptor.py:17, 16 zones).  bounds = Z x
 * [theta_min, phi_min, theta_max, phi_max] (eqsp.py:16-20).  For which = 0 also
 * pass the 3x3 matrices the reference derives from the zone centres:
 * to_dom[a] = rotation bringing the centre of zone a onto +z (Orientator.py:198-205,
 * identity for a = 0) and adj_sec[a] = z-rotation of Orientator.py:259-263.
 */
int mad_set_eqsp(mad_ctx *ctx, int which, int Z, const double *bounds,
                 const double *to_dom, const double *adj_sec);

/* ---- gradient fields (input of the path; produced by MapSpace.py:178-189) ---- */

/*
 * Upload one octave's gradient field into slot (0..MAD_MAX_FIELDS-1): three planar
 * float32 volumes, the layout of the reference's grad_list view (MapSpace.py:187).
 * On the device the field is repacked to one 16-byte texel per voxel
 * {gx, gy, gz, |g|}.
 */
int mad_upload_field(mad_ctx *ctx, int slot, const float *gx, const float *gy, const float *gz,
                     int nx, int ny, int nz);
/* Same, from a DEVICE buffer holding the three planes back to back ([3][nx][ny][nz]). */
int mad_upload_field_device(mad_ctx *ctx, int slot, const float *g3_device, int nx, int ny, int nz);
int mad_free_field(mad_ctx *ctx, int slot);

/* ---- a1-a8: Orientator.assign_orientations (Orientator.py:68-110) ------------ */

/*
 * coords: n x 3 integer voxel positions in the octave of `slot` (DensityFeature.coords).
 * octave: 1 = base grid (unit stride), 0 = upsampled grid (stride 2), Orientator.py:128-158.
 * r = box_side (8 for the default patch of 16).  Rows are emitted in anchor order
 * x main bin ascending x secondary bin ascending (Orientator.py:90-106).
 * row_count (nullable): Z quantised zone counts per row (DensityFeature.ar_count).
 * n_reject (nullable): anchors refused by the border test (Orientator.py:131-135).
 */
/*
 * Orientator(gw_sig = sigma) (Orientator.py:49-54): a Gaussian window exp(-d^2 / (2 sigma^2)) on the orientation histogram,
 * d = a voxel's offset from the anchor in voxels of the box.  0 (the default, and what MaD.run uses) = no window.  Applies
 * to mad_orient and mad_set_build from the next call on.  The reference truncates each zone's float64 weight sum to
 * int32 before it quantises; the kernel sums the weights in 2^-50 fixed point (deterministic, |error| < 1.2e-12 per
 * sum), so a zone count can differ from the reference's only where its weight sum lies that close to an integer.
 */
int mad_set_orient_window(mad_ctx *ctx, double gw_sig);
int mad_orient(mad_ctx *ctx, int slot, int octave, const int32_t *coords, int n, int r,
               int lim_main, int lim_sec,
               int32_t *row_anchor, int32_t *row_main, int32_t *row_sec, double *row_R,
               int32_t *row_count, int64_t *n_rows, int64_t cap, int32_t *n_reject);

/* ---- a9-a10: Descriptor.generate_descriptors (Descriptor.py:106-202) --------- */

/*
 * coords: n_rows x 3 anchor voxel positions, R: n_rows x 9 Rfinal.  dsc: n_rows x
 * (64 * Zd) int16, sub-cube major (id j*16+i*4+k, Descriptor.py:44-64), zone minor.
 * A row whose sample cube leaves the grid is all zero (Descriptor.py:140-149).
 */
int mad_describe(mad_ctx *ctx, int slot, int octave, const int32_t *coords, const double *R,
                 int64_t n_rows, int r, int16_t *dsc);
/*
 * The same with another partition of the sample cube, Descriptor(dsc_size = 64 | 27 | 8 | 1) (Descriptor.py:44-93):
 * dsc is n_rows x (dsc_size * Zd), sub-regions in the order of the reference's sub_slices lists.  MaD.run never selects
 * them (it constructs Descriptor(dsc_radius=patch_size) only, MaD.py:362); 27, 8 and 1 exist for r = 8.
 */
int mad_describe_sized(mad_ctx *ctx, int slot, int octave, const int32_t *coords, const double *R,
                       int64_t n_rows, int r, int dsc_size, int16_t *dsc);

/* ---- a11: MaD._match_dsc part 1 (MaD.py:416-424) ----------------------------- */

/*
 * hi: n_hi x D, lo: n_lo x D int16 descriptor counts (D = 1024; every |count| <= 127,
 * else MAD_EDOM).  Emits every (hi, lo) with normalised correlation > cc, row-major,
 * with its score.
 */
int mad_correlate(mad_ctx *ctx, const int16_t *hi, int64_t n_hi, const int16_t *lo, int64_t n_lo,
                  int D, double cc, int32_t *pair_hi, int32_t *pair_lo, double *pair_score,
                  int64_t *n_pairs, int64_t cap);

/* ---- a12: MaD._match_dsc part 2 (MaD.py:426-451) ----------------------------- */

/*
 * Per-row inputs: p = subv_map_coords (n x 3), R = Rfinal (n x 9), meta = {index,
 * oct_scale, main_bin} (n x 3 int32).  hi_cloud / lo_cloud: the unique sub-voxel
 * coordinates of rows present in at least one pair (MaD.py:427-428).
 * results (nullable): n_pairs x 23 as MaD.py:451.  counts (nullable): number of
 * hi cloud points brought within `dist` of a lo cloud point (repeatability =
 * 100 * count / l_hi).
 */
int mad_pose_score(mad_ctx *ctx, const int32_t *pair_hi, const int32_t *pair_lo, const double *pair_score,
                   int64_t n_pairs,
                   const double *hi_p, const double *hi_R, const int32_t *hi_meta, int64_t n_hi,
                   const double *lo_p, const double *lo_R, const int32_t *lo_meta, int64_t n_lo,
                   const double *hi_cloud, int64_t l_hi, const double *lo_cloud, int64_t l_lo,
                   double dist, double *results, int32_t *counts);

/*
 * Indices of the first k pairs in the order of MaD._filter_dsc_pairs' stable sort
 * (MaD.py:480): count descending, ties in input order.
 */
int mad_topk(mad_ctx *ctx, const int32_t *counts, int64_t n, int64_t k, int64_t *order);

/* ---- device-resident pipeline (same kernels, no host round trips) -------------- */

/*
 * A mad_set holds the oriented-anchor rows of one structure (map or subunit) on
 * the device: descriptors, Rfinal, anchor ids and the anchors' sub-voxel
 * coordinates.  mad_set_build runs a1-a10 for the anchors of both octaves:
 *   slot_of_octave[o] = field slot of octave o (or -1),
 *   anc_coords n x 3 int32, anc_octave n int32, anc_subv n x 3 double (Angstrom),
 *   anc_index n int32 (DensityFeature.index).
 * Row order = anchor order (as given) x main x sec, exactly the reference's list.
 */
int mad_set_create(mad_ctx *ctx, mad_set **out);
void mad_set_destroy(mad_ctx *ctx, mad_set *set);
int mad_set_build(mad_ctx *ctx, mad_set *set, const int *slot_of_octave,
                  const int32_t *anc_coords, const int32_t *anc_octave, const double *anc_subv,
                  const int32_t *anc_index, int n_anchors, int r, int lim_main, int lim_sec);
/* The same for n_sets structures at once -- the map and the subunits of a MaD.run step, which the reference describes one
 * after the other (MaD._describe_struct, MaD.py:358-368, called per structure from get_descriptors, MaD.py:116-163): ONE k_orient grid, one
 * scan, one row expansion and ONE k_describe grid over the anchors / rows of all of them, so that small structures do not each
 * pay a launch that cannot fill 256 CUs.  Arrays of n_sets entries; slot_of_octave has 2 per set.  Each set comes out exactly as
 * mad_set_build would have made it.  Asynchronous, on the lane of sets[0]; every set's consumers wait for its own event. */
int mad_set_build_many(mad_ctx *ctx, int n_sets, mad_set *const *sets, const int *slot_of_octave,
                       const int32_t *const *anc_coords, const int32_t *const *anc_octave,
                       const double *const *anc_subv, const int32_t *const *anc_index, const int *n_anchors,
                       int r, int lim_main, int lim_sec);
/* Load rows computed earlier (descriptor cache, MaD.py:861-875): anchor = row -> anchor id. */
int mad_set_load(mad_ctx *ctx, mad_set *set, int64_t n_rows, const int32_t *row_anchor,
                 const int32_t *row_main, const double *row_R, const int16_t *dsc, int D,
                 const double *anc_subv, const int32_t *anc_index, const int32_t *anc_octave, int n_anchors);
int mad_set_size(mad_ctx *ctx, const mad_set *set, int64_t *n_rows, int32_t *n_anchors);
/* Any of the output pointers may be NULL. */
int mad_set_download(mad_ctx *ctx, const mad_set *set, int32_t *row_anchor, int32_t *row_main,
                     int32_t *row_sec, double *row_R, int16_t *dsc);

/*
 * a11 + a12 + top-k for one (subunit = hi, map = lo) pair of sets, entirely on the
 * device.  results: k x 23 rows of MaD.py:451 in the order of MaD.py:480 (fewer if
 * n_pairs < k: *n_out).  pair_index (nullable): k row-major pair ranks.
 * stats (nullable) int64[4] = {n_pairs, l_hi, l_lo, n_correlations}.
 */
int mad_match_topk(mad_ctx *ctx, const mad_set *hi, const mad_set *lo, double cc, double dist,
                   int64_t k, double *results, int64_t *pair_index, int64_t *n_out, int64_t *stats);
/*
 * The same for n subunit sets against one map set, with up to 4 matches in flight before the host waits for
 * the oldest.  results: n x k x 23; pair_index (nullable): n x k; n_out: n; stats (nullable): n x 4.
 * mad_match_fetch / _results / _used afterwards refer to the LAST match of the batch.
 */
int mad_match_topk_many(mad_ctx *ctx, int n, const mad_set *const *hi, const mad_set *lo, double cc, double dist,
                        int64_t k, double *results, int64_t *pair_index, int64_t *n_out, int64_t *stats);

/*
 * The same call in two halves, for callers that keep the device fed: _begin enqueues every match and returns
 * (the output arrays belong to the bracket until _finish returns: _finish fills them; _begin itself only does so for a
 * match whose lane it has to reuse, n > 8).  Between the two the caller may enqueue other work -- typically
 * mad_set_build of the NEXT batch into other sets: sets read by the open bracket must not be rebuilt or destroyed, and
 * no other match call may be made.  Up to three brackets may be open at once (each has its own pinned result staging);
 * _finish completes the older one.
 */
int mad_match_topk_many_begin(mad_ctx *ctx, int n, const mad_set *const *hi, const mad_set *lo, double cc, double dist,
                              int64_t k, double *results, int64_t *pair_index, int64_t *n_out, int64_t *stats);
int mad_match_topk_many_finish(mad_ctx *ctx);
/* on != 0: mad_match_topk_many(_begin) computes the score tiles of all matches of a bracket (first match of every lane) in ONE
 * GEMM grid -- fewer, fuller launches (C3: 0.151 -> 0.106 ms of device time per step) at the price of every match waiting for
 * the slowest set; off (default): one GEMM per match, each on its own lane, which overlaps better when several lanes are busy
 * (1.07 against 1.21 ms per overlapped C3 step).  Results are identical either way. */
int mad_set_batching(mad_ctx *ctx, int on);
/* Tuning values that change speed, never results.  "pose_split" (-1 auto: subunits of more than 512 anchors; 0 off; 1 on): the
 * pose search of mad_match_topk* brackets the best-scoring pairs first and, in a second launch, abandons every other pair as
 * soon as its running upper bound falls below the k-th lower bound of the first batch.  "pose_split_min": that first batch
 * holds at least this many pairs (default 4096) and at least 64 k.  "pose_mx" (0 default, 1): the bracket of hi clouds of up to 512
 * points with its coarse map on the matrix cores, four pairs per wave (k_pose_bounds_mx; measured slower, kept for comparison).  "ori_queue" / "dsc_queue": entries in use of the queues through which k_orient / k_describe hand the
 * directions their fast classifiers cannot decide to the exact one (defaults: all 512 / 768); a small value drives the kernels'
 * full-queue paths for every anchor / row -- a test hook, the results are the same. */
int mad_set_option(mad_ctx *ctx, const char *name, double value);

/*
 * Which pose-scoring kernel the most recently enqueued match used: 0 = k_pose_lds (both clouds in LDS as float64),
 * 1 = k_pose_lds32 (lo cloud as float32 offsets in LDS), 2 = k_pose (global cell list), -1 = none yet.  For tests and
 * diagnostics: the three give identical counts, so only this tells when a sizing change has demoted a workload.
 */
int mad_last_pose_kernel(mad_ctx *ctx);
/*
 * How many pairs of the most recently COMPLETED match went through the exact search.  mad_match_topk* only report the k best
 * pairs (MaD.py:480,502), so the pose search first brackets every pair's count with the occupancy bitmaps alone (lower bound =
 * points certainly within dist, upper bound = those plus the points in the uncertain shell) and searches exactly only the pairs
 * whose upper bound reaches the k-th largest lower bound; the k rows and their order are those of the full search.  Equal to
 * n_pairs when nothing could be pruned.  mad_match_fetch(counts) / mad_match_results complete the other pairs on demand.
 */
int64_t mad_last_pose_selected(mad_ctx *ctx);
/* Device buffers the context has (re)allocated since mad_init (scratch and set storage grow on demand and never shrink).  Each is a
 * hipMalloc -- and, when it replaces a smaller buffer, a hipFree that waits for every stream.  A caller that pipelines steps can
 * check that its steady state shows none (bench.py: config.device_allocations_in_timed_region). */
int64_t mad_device_allocations(mad_ctx *ctx);
/* After mad_match_topk: all pairs of that call (for MaD._match_dsc's full return value). */
int mad_match_fetch(mad_ctx *ctx, int32_t *pair_hi, int32_t *pair_lo, double *pair_score,
                    int32_t *counts, int64_t cap);
/* After mad_match_topk(hi, lo, ...): the MaD.py:451 rows of ALL pairs, row-major pair order (n_pairs x 23). */
int mad_match_results(mad_ctx *ctx, const mad_set *hi, const mad_set *lo, double *results, int64_t cap);
/* After mad_match_topk: which anchors of each set appear in a pair (uint8 flags). */
int mad_match_used(mad_ctx *ctx, uint8_t *hi_anchor_used, int32_t n_hi_anchors,
                   uint8_t *lo_anchor_used, int32_t n_lo_anchors);

/* ---- a13: structure_utils.refine_pdb (structure_utils.py:58-161) --------------- */

/*
 * Upload the density map the candidates are refined in (Dmap.grid3d, float32
 * [nx][ny][nz], origin and voxel spacing in Angstrom).  The library derives the
 * np.gradient field (structure_utils.py:80) once and keeps both.
 */
int mad_upload_density(mad_ctx *ctx, const float *grid, int nx, int ny, int nz,
                       double ox, double oy, double oz, double voxsp);
/*
 * Refine n_cand rigid placements of the same n_atoms-atom structure at once
 * (one persistent workgroup per candidate).  coords: n_cand x n_atoms x 3, updated
 * in place.  converged / last_step: per candidate, the reference's return values.
 */
int mad_refine(mad_ctx *ctx, double *coords, int n_cand, int64_t n_atoms, int n_steps,
               double max_step, double min_step, int32_t *converged, int32_t *last_step);

/* ---- a14-a15: PDB.structure_to_density (PDB.py:131-292) ------------------------ */

/*
 * atoms n x 3 (Angstrom), mass n.  Call with grid = NULL to get dims (output grid
 * shape) and origin; then with grid = float32[dims0*dims1*dims2] ([x][y][z]).
 */
int mad_structure_to_density(mad_ctx *ctx, const double *atoms, const double *mass, int64_t n,
                             double resolution, double voxsp, double isovalue, int pad,
                             int32_t dims[3], double origin[3], float *grid);

/* ---- a16: Dmap.get_CCC_with_grid (Dmap.py:153-258) ----------------------------- */

/*
 * Both grids float32 [x][y][z]; voxels below isovalue are zeroed IN PLACE in both,
 * as the reference does (Dmap.py:160-161).  *ccc = <a,b>/sqrt(<a,a><b,b>) over the
 * overlap box, 0 if the boxes do not overlap.
 */
int mad_ccc(mad_ctx *ctx, float *grid1, const int32_t dims1[3], const double origin1[3],
            float *grid2, const int32_t dims2[3], const double origin2[3],
            double voxsp, double isovalue, double *ccc);

/* ---- one subunit's pair grid sharded over GPUs by blocks of map rows (the exchange steps are the caller's:
 *      OR of the flag vectors, all-gather of the per-shard top-k; mad_amd/dist.py::sharded_match) -------------- */

/*
 * Stage B of a sharded match: correlate hi against the lo rows [lo_begin, lo_end) (MaD.py:416-424 on that block of
 * `preds`), keep the pairs on the device, return this shard's flags "anchor takes part in a pair" (one byte per
 * anchor of hi / of lo).  The OR of the flags over all shards defines the global clouds of MaD.py:427-428.
 */
int mad_match_shard_pairs(mad_ctx *ctx, const mad_set *hi, const mad_set *lo, int64_t lo_begin, int64_t lo_end, double cc,
                          uint8_t *used_hi, uint8_t *used_lo, int64_t *n_pairs);

/*
 * Stage C: score the shard's pairs against the global clouds (MaD.py:433-451) and return its k best in the order of
 * MaD.py:480 restricted to the shard: result rows [k][23], match counts, and pair_rank = hi_row * N_lo + lo_row, the
 * position of the pair in the unsharded row-major list.  *l_hi = size of the global hi cloud.  Must follow
 * mad_match_shard_pairs for the same sets on the same ctx.
 */
int mad_match_shard_topk(mad_ctx *ctx, const mad_set *hi, const mad_set *lo, const uint8_t *used_hi_all,
                         const uint8_t *used_lo_all, double dist, int64_t k, double *results, int64_t *pair_rank,
                         int32_t *counts, int64_t *n_out, int64_t *l_hi);

/*
 * Stages B and C without a host round trip (MaD.py:420-451 for one block of map rows; the loop it shards is serial over subunits,
 * MaD.py:165-190).  Everything is enqueued on the lane of `hi`; the flags and the shard's list live in DEVICE memory of the caller,
 * who orders the two exchanges of SURVEY.md 8(e) -- the OR all-reduce of the flags, the all-gather of the lists -- on that lane's
 * stream (mad_set_stream(hi)) between and behind the two calls.
 *   mad_match_shard_begin: hi against the lo rows [lo_begin, lo_end) of a lo set ASSUMED to have n_lo rows (the caller cut the blocks
 *     from that number; the device checks it).  d_flags: hi->n_anchors + lo->n_anchors bytes, hi's flags first.
 *   mad_match_shard_score: the shard's pairs against the global clouds (d_flags_all: the OR over the shards, same layout); d_out:
 *     mad_match_shard_record_doubles(k) float64 = [rows m, flags, |hi cloud|, pairs][k x 23 result rows][k counts][k pair ranks].
 *     flags != 0 (1 score-matrix capacity, 2 pair capacity, 4 n_lo was wrong, 8 selection list): m = 0, repeat the shard through
 *     mad_match_shard_pairs / mad_match_shard_topk.
 */
int64_t mad_match_shard_record_doubles(int64_t k);
int mad_match_shard_begin(mad_ctx *ctx, const mad_set *hi, const mad_set *lo, int64_t lo_begin, int64_t lo_end, int64_t n_lo,
                          double cc, uint8_t *d_flags);
int mad_match_shard_score(mad_ctx *ctx, const mad_set *hi, const mad_set *lo, const uint8_t *d_flags_all, double dist, int64_t k,
                          double *d_out);
/*
 * The records of a group's shards (n float64 in device memory: what the caller's all-gather left behind mad_match_shard_score) to
 * the host: copied into pinned memory of the library on the lane of `hi`; *ticket names the copy.  mad_match_shard_wait blocks
 * until it has arrived and copies it to `out` (host).  At most 8 copies of one lane may be pending.
 */
int mad_match_shard_collect(mad_ctx *ctx, const mad_set *hi, const double *d_all, int64_t n, int *ticket);
int mad_match_shard_wait(mad_ctx *ctx, int ticket, double *out, int64_t n);

/* ---- one structure's rows built in shares on several GPUs (stage A).  Orientation and description
 *      are independent per anchor (Orientator.py:80-108, Descriptor.py:106-116): anchor a of the structure's list goes
 *      to share a % n_shares (local position a / n_shares), every rank runs mad_set_build on its share, the shares
 *      travel as fixed-size "wire images" (one all-gather, issued by the caller: mad_amd/dist.py) and every rank
 *      assembles the full set, rows in the reference's order (anchor x main x sec, Orientator.py:90-106) -- bit for bit
 *      the set mad_set_build makes from the whole list. ------------------------------------------------------------ */

/* Bytes of a wire image with room for cap_rows rows of D counts (header + main / sec / local anchor / norm / int8 rows). */
int64_t mad_set_wire_bytes(int D, int64_t cap_rows);
/*
 * Packs the rows of a built set into a wire image.  wire_on_device != 0: `wire` is device memory (e.g. the tensor
 * handed to the all-gather) and the call is asynchronous on the set's lane (mad_set_stream); otherwise host memory
 * and synchronous.  A share with more than cap_rows rows is sent as an empty image whose header carries its row count;
 * the import then fails with MAD_ENOSPC at the first call that needs the set's size.
 */
int mad_set_export(mad_ctx *ctx, const mad_set *share, void *wire, int wire_on_device, int64_t cap_rows);
/*
 * Assembles `set` from n_shares wire images laid out back to back (share r at wires + r * mad_set_wire_bytes).  The
 * anchors are those of the WHOLE structure in the reference's order; anc_coords may be NULL (an imported set is never
 * described again).  Rfinal, inv(Rfinal), the int16 counts and the result metadata are re-derived on the importing
 * GPU with the builder's own expressions.  wires_on_device as above (asynchronous on the set's lane).
 */
int mad_set_import(mad_ctx *ctx, mad_set *set, const void *wires, int wires_on_device, int n_shares, int64_t cap_rows,
                   const int32_t *anc_coords, const int32_t *anc_octave, const double *anc_subv, const int32_t *anc_index,
                   int n_anchors);
/*
 * The lane a set is built on, its HIP stream (for callers that order a collective against the export / import
 * kernels: hipStreamWaitEvent or a torch ExternalStream), and a way to put two sets on one lane so that
 * build -> export -> all-gather -> import is one in-order stream.  mad_set_bind_lane synchronises the ctx.
 */
int mad_set_lane(mad_ctx *ctx, const mad_set *set);
void *mad_set_stream(mad_ctx *ctx, const mad_set *set);
int mad_set_bind_lane(mad_ctx *ctx, mad_set *set, int lane);

/*
 * a14-a16 for a batch of placed copies of one structure, on the device end to end: candidate c's atoms
 * (atoms + c*n*3, float64) -> simulated density at the voxel spacing of the map uploaded with
 * mad_upload_density (PDB.structure_to_density(resolution, voxsp, isovalue = density_isovalue)) ->
 * ccc[c] = Dmap.get_CCC_with_grid(grid, x0, y0, z0, isovalue = ccc_isovalue) against that map.  The uploaded
 * map is clamped on the fly, not modified.  One read-back at the end (MaD.py:613-616 per solution).
 */
int mad_density_ccc(mad_ctx *ctx, const double *atoms, const double *mass, int n_cand, int64_t n,
                    double resolution, double density_isovalue, double ccc_isovalue, double *ccc);

/*
 * MaD._refine_filtered_solutions (MaD.py:556-629) for a batch of candidate poses, on the device from the poses to the scores -- what
 * mad_refine + mad_density_ccc do with a host round trip of all coordinates in between.  The candidates may belong to several
 * structures (the subunits of a run: the reference refines them one subunit after the other, MaD.py:165-190): structure s =
 * base_atoms[first_atom[s] .. first_atom[s + 1]) (float64 xyz) with masses alongside, candidate c is a pose of structure cand_struct[c]:
 *   start_c = (atoms - hi_p[c]) @ rot[c] + lo_p[c]             (MaD.py:566-569: translate_atoms(-hi), rotate_atoms(R), translate_atoms(lo);
 *                                                               rot[c] row-major 3x3 in PDB.rotate_atoms' convention coords @ R)
 *   refine_pdb(map, start_c, n_steps, max_step, min_step)      (structure_utils.py:58-161, against the map of mad_upload_density)
 *   ccc[c] = map.get_CCC_with_grid(structure_to_density(refined_c, resolution, voxsp of the map, density_isovalue), ccc_isovalue)
 * converged / last_step: refine_pdb's return values per candidate.  coords (nullable): the refined coordinates of candidate 0, 1, ...
 * back to back; NULL leaves them on the device (15 doubles per candidate in and ~1.6 KB per candidate out cross the bus).  A
 * candidate whose refinement ends in NaN coordinates (structure_utils.py:97-98) gets ccc = NaN.
 */
int mad_dock_refine_score(mad_ctx *ctx, int n_struct, const double *base_atoms, const double *mass, const int64_t *first_atom,
                          int n_cand, const int32_t *cand_struct, const double *hi_p, const double *lo_p, const double *rot,
                          int n_steps, double max_step, double min_step, double resolution, double density_isovalue,
                          double ccc_isovalue, double *coords, int32_t *converged, int32_t *last_step, double *ccc);

/* ---- next to the path, downstream: occupancy overlap for assembly building ------------ */

/*
 * structure_utils.get_overlap(g1, g2, voxsp, isovalue) (structure_utils.py:163-259): both grids
 * (host, float32 [x][y][z], z fastest) are clamped in place (values < isovalue -> 0), the common box
 * follows from the origins (Angstrom) in voxel units with python's round(), *common = voxels of the
 * box where both grids are > 0, *n_pos1 = voxels of grid1 that are > 0.  The reference's return value
 * is common / n_pos1 (0 when n_pos1 == 0 or the boxes do not meet).
 */
int mad_grid_overlap(mad_ctx *ctx, float *grid1, const int32_t d1[3], const double o1[3], float *grid2,
                     const int32_t d2[3], const double o2[3], double voxsp, double isovalue, int64_t *common,
                     int64_t *n_pos1);

/*
 * The overlap table of MaD._build_from_single / _build_models (MaD.py:667-686, 760-783) in one call:
 * structure s = atoms[first_atom[s] .. first_atom[s+1]) (float64 xyz, masses alongside) is turned into
 * PDB.structure_to_density(resolution, voxsp, isovalue = density_isovalue) on the device, the grids stay
 * there, and overlap[i * n_struct + j] = get_overlap(grid_i, grid_j, voxsp, overlap_isovalue) for i < j
 * (0 elsewhere, as the reference leaves its table).  One read-back of the counts.
 */
int mad_overlap_matrix(mad_ctx *ctx, const double *atoms, const double *mass, const int64_t *first_atom,
                       int n_struct, double resolution, double voxsp, double density_isovalue,
                       double overlap_isovalue, double *overlap);

/* ---- next to the path, upstream: MapSpace.build_space (MapSpace.py:116-189) and the
 *      dense half of Detector.find_anchors (Detector.py:28-29) ------------------------ */

typedef struct mad_space mad_space;

int mad_space_create(mad_ctx *ctx, mad_space **out);
void mad_space_destroy(mad_ctx *ctx, mad_space *s);

/*
 * Builds the scale space of one density grid on the device.
 *   grid      host, [x][y][z] z fastest, float32 (is_f64 = 0; PDB / MRC input) or float64
 *             (is_f64 = 1; Situs input, MapSpace.py:93-96); padded with `pad` zero voxels
 *             per face (MapSpace.py:117-118).
 *   oct_mode  1 = base, 2 = upsampled, 3 = both (MapSpace.py:147-163).  List entry 0 is the
 *             upsampled octave when it exists, as in the reference.
 *   g0, g2    order-0 and order-2 Gaussian kernels of sigma_init, 2*radius+1 taps each, as
 *             scipy.ndimage computes them (the host passes numpy's values so that the
 *             weights are identical); sig2 = sigma_init^2 (MapSpace.py:171).
 *   pre       order-0 kernel of the pre-smoothing sigma (MapSpace.py:144), pre_radius = 0
 *             to skip it.
 *   lu, ev_w, ev_i   per axis a (padded length n_a): banded LU factors [5][n_a] of the
 *             not-a-knot cubic collocation matrix, and for the 2 n_a - 1 half-integer
 *             sites the 4 basis weights + first coefficient index (interp1d(kind="cubic"),
 *             MapSpace.py:206-214).  Only read when the upsampled octave is built.
 *   slot_up, slot_base   field slots that receive the gradient texels of np.gradient(
 *             gaussian_filter(grid, sigma_init)) (MapSpace.py:182-187); -1 = do not fill.
 * Filter passes reproduce scipy.ndimage's summation order and per-pass rounding; the
 * spline agrees with scipy to ~4e-16 relative (see mad_space.hip).
 */
int mad_space_build(mad_ctx *ctx, mad_space *s, const void *grid, int is_f64, int nx, int ny, int nz, int pad,
                    int oct_mode, const double *g0, const double *g2, int radius, double sig2,
                    const double *pre, int pre_radius, const double *const *lu, const double *const *ev_w,
                    const int32_t *const *ev_i, int slot_up, int slot_base);

/* n_octaves list entries; dims6 = [entry][3]; kind2[entry] = 0 upsampled / 1 base; is_f64_2[entry]. */
int mad_space_info(mad_ctx *ctx, const mad_space *s, int *n_octaves, int32_t *dims6, int32_t *kind2, int32_t *is_f64_2);

/* what: 0 = grid_list[entry], 1 = map_space[entry] (LoG), 2 = gauss_list[entry]; storage type of the entry. */
int mad_space_download(mad_ctx *ctx, const mad_space *s, int entry, int what, void *out);

/*
 * skimage.feature.peak_local_max(map_space[entry], exclude_border=border, threshold_abs=
 * threshold) as Detector.py:29 calls it: voxels equal to the maximum of their zero-extended
 * 3x3x3 neighbourhood and strictly above the threshold.  Returns linear indices
 * ((x*ny + y)*nz + z) and values in NO particular order; the host sorts (row-major, then by
 * descending value).  MAD_ENOSPC with *n_out = required capacity if cap is too small.
 */
int mad_space_peaks(mad_ctx *ctx, const mad_space *s, int entry, double threshold, int border, int64_t *lin_index,
                    double *value, int64_t cap, int64_t *n_out);

/* (2r+1)^3 LoG neighbourhoods of n voxels (zero outside), storage type of the entry: the input of
 * Detector.check_localize (Detector.py:53-123), which stays on the host. */
int mad_space_patches(mad_ctx *ctx, const mad_space *s, int entry, const int32_t *coords, int n, int r, void *out);

#ifdef __cplusplus
}
#endif
#endif /* MAD_AMD_H */
