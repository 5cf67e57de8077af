/*
 * mad_oracle.c -- CPU restatement of the MaD hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle for the HIP library in mad_amd/csrc.  It is a
 * plain, scalar, single-threaded C restatement of the arithmetic the reference
 * (LBM-EPFL/MaD, pure Python/numpy/scipy) performs on the hot path.  Nothing in
 * the product (mad_amd/) may include, link or call it: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and only as the
 * checker / the timed CPU baseline.
 *
 * Pinning: tests/test_oracle_golden.py checks every function below against
 * golden vectors generated in the build container by importing the reference
 * itself (tests/golden/make_golden.py, fixtures in tests/golden/ *.npz).
 *
 * Each function cites the reference lines it follows (paths under the
 * reference checkout, e.g. mad/Orientator.py:116-169).
 *
 * Conventions shared with include/mad_amd.h:
 *   - gradient fields are three planar float32 volumes gx,gy,gz of shape
 *     [nx][ny][nz], z fastest (the memory layout of the reference's
 *     np.moveaxis(np.gradient(...)) view, mad/MapSpace.py:187);
 *   - 3x3 matrices are row-major double[9];
 *   - EQSP bounds are double[Z][4] = theta_min phi_min theta_max phi_max,
 *     centres double[Z][2] = theta phi (mad/eqsp/eqsp.py:16-33).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAX_Z 128
#define ORC_TWO_PI 6.283185307179586476925286766559
#define ORC_PI 3.14159265358979323846

/* ------------------------------------------------------------------ */
/* small helpers                                                       */
/* ------------------------------------------------------------------ */

/* math_utils.py:15-27  euler_rod_mat(axis, angle) */
static void orc_rod(const double ax[3], double angle, double m[9]) {
    double a = cos(angle / 2.0);
    double s = sin(angle / 2.0);
    double b = -ax[0] * s, c = -ax[1] * s, d = -ax[2] * s;
    double aa = a * a, bb = b * b, cc = c * c, dd = d * d;
    double bc = b * c, ad = a * d, ac = a * c, ab = a * b, bd = b * d, cd = c * d;
    m[0] = aa + bb - cc - dd; m[1] = 2 * (bc + ad);      m[2] = 2 * (bd - ac);
    m[3] = 2 * (bc - ad);      m[4] = aa + cc - bb - dd; m[5] = 2 * (cd + ab);
    m[6] = 2 * (bd + ac);      m[7] = 2 * (cd - ab);      m[8] = aa + dd - bb - cc;
}

/* math_utils.py:5-13 unit_vector: v / sqrt(v.v); a zero vector is returned as is */
static void orc_unit(const double v[3], double o[3]) {
    double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    if (n == 0.0 || n != n) { o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; return; }
    o[0] = v[0] / n; o[1] = v[1] / n; o[2] = v[2] / n;
}

static void orc_mat3_mul(const double a[9], const double b[9], double o[9]) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            o[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}

/* general 3x3 inverse (np.linalg.inv, MaD.py:438 / Descriptor.py:132) by cofactors */
static void orc_mat3_inv(const double m[9], double o[9]) {
    double c00 = m[4] * m[8] - m[5] * m[7];
    double c01 = m[5] * m[6] - m[3] * m[8];
    double c02 = m[3] * m[7] - m[4] * m[6];
    double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    double id = 1.0 / det;
    o[0] = c00 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c01 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c02 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

static double orc_clamp1(double z) { return z > 1.0 ? 1.0 : (z < -1.0 ? -1.0 : z); }

/* Zone test of Orientator.py:328-331 / Descriptor.py:181-184: strict, on theta,
 * theta+2pi and phi.  Returns the LAST matching zone or -1. */
/* Orientator.process_df_gradient (Orientator.py:326-334) sums the weights zone by zone, so a direction inside the sliver where
 * the rounded bounds of two zones overlap (theta in (0, 6.2832 - 2 pi) matches the first AND the last zone of a belt) is counted in
 * both; the Descriptor (Descriptor.py:173-187) assigns one zone id per sample, the last match winning (orc_zone below). */
static void orc_count_zones(double th, double sth, double ph, const double *bounds, int Z, int w, int32_t *cnt) {
    for (int a = 0; a < Z; a++) {
        const double *b = bounds + 4 * a;
        int thm = (th < b[2]) && (th > b[0]);
        int sthm = (sth < b[2]) && (sth > b[0]);
        int phm = (ph < b[3]) && (ph > b[1]);
        if ((thm || sthm) && phm) cnt[a] += w;
    }
}

static int orc_zone(double th, double sth, double ph, const double *bounds, int Z) {
    int hit = -1;
    for (int a = 0; a < Z; a++) {
        const double *b = bounds + 4 * a;
        int thm = (th < b[2]) && (th > b[0]);
        int sthm = (sth < b[2]) && (sth > b[0]);
        int phm = (ph < b[3]) && (ph > b[1]);
        if ((thm || sthm) && phm) hit = a;
    }
    return hit;
}

/* ------------------------------------------------------------------ */
/* EQSP derived tables                                                 */
/* ------------------------------------------------------------------ */

/* eqsp.py:37-46: a new belt starts whenever phi_min changes.  belt_first[a] =
 * index of the first zone of the belt zone a belongs to. */
void orc_eqsp_belt_first(const double *bounds, int Z, int32_t *belt_first) {
    double prev = -1.0;
    int first = 0;
    for (int a = 0; a < Z; a++) {
        if (bounds[4 * a + 1] != prev) { first = a; prev = bounds[4 * a + 1]; }
        belt_first[a] = first;
    }
}

/* Orientator.py:198-205 (+ eqsp.py:29-31): matrix that brings the centre of
 * zone `a` onto +z; identity for a == 0 (Orientator.py:211). */
/* The two matrix tables as the reference's numpy code produces them (oracle.py builds them with the reference's own expressions:
 * numpy's dot / sin / cos are not bit-identical to a scalar C restatement, and a direction that the rotation puts exactly on a zone
 * bound is decided by that last bit).  When set, orc_to_dom_mat / orc_adj_sec_mat read the tables; the C formulas below remain
 * for callers without numpy tables and agree with them to 1e-15. */
static const double *g_tab_dom = 0, *g_tab_adj = 0;
static int g_tab_Z = 0;
void orc_set_matrix_tables(const double *dom, const double *adj, int Z) { g_tab_dom = dom; g_tab_adj = adj; g_tab_Z = Z; }

void orc_to_dom_mat(const double *centers, int a, double m[9]) {
    if (g_tab_dom && a >= 0 && a < g_tab_Z) { memcpy(m, g_tab_dom + 9 * a, 9 * sizeof(double)); return; }
    if (a == 0) { memset(m, 0, 9 * sizeof(double)); m[0] = m[4] = m[8] = 1.0; return; }
    double th = centers[2 * a], ph = centers[2 * a + 1];
    double c[3] = { sin(ph) * cos(th), sin(ph) * sin(th), cos(ph) }, u[3];
    orc_unit(c, u);
    double angle = acos(orc_clamp1(u[2]));
    double cr[3] = { u[1] * 1.0 - u[2] * 0.0, u[2] * 0.0 - u[0] * 1.0, u[0] * 0.0 - u[1] * 0.0 }, ax[3];
    orc_unit(cr, ax);
    orc_rod(ax, angle, m);
}

/* Orientator.py:253-263: z-rotation that brings the centre of zone `s` onto the
 * azimuth of the first zone of its belt. */
void orc_adj_sec_mat(const double *bounds, const double *centers, int Z, int s, double m[9]) {
    if (g_tab_adj && Z == g_tab_Z && s >= 0 && s < Z) { memcpy(m, g_tab_adj + 9 * s, 9 * sizeof(double)); return; }
    int32_t bf[ORC_MAX_Z];
    orc_eqsp_belt_first(bounds, Z, bf);
    double ftheta = -1.0 * (centers[2 * s] - centers[2 * bf[s]]);
    double ax[3] = { 0, 0, 1 };
    orc_rod(ax, ftheta, m);
}

/* ------------------------------------------------------------------ */
/* a1-a8: orientation assignment                                       */
/* ------------------------------------------------------------------ */

/* Orientator.py:336-340: quantise counts to 0..50 of their max (int32 truncation) */
static int orc_quantise(const int32_t *cnt, int n, int32_t *q) {
    int32_t mx = 0;
    for (int i = 0; i < n; i++) if (cnt[i] > mx) mx = cnt[i];
    if (mx == 0) return 0;
    for (int i = 0; i < n; i++) q[i] = (int32_t)((double)cnt[i] / (double)mx * 50.0);
    return 1;
}

/*
 * Orientator.assign_orientations (Orientator.py:68-110) for n anchors of one octave.
 *   octave 1 = base grid (unit stride), octave 0 = upsampled grid (stride 2),
 *   r = box_side (patch_size/2/... = 8 for the default patch of 16).
 * Rows come out in anchor order x main bin ascending x secondary bin ascending.
 * Returns 0, or -1 if more than `cap` rows would be produced (*n_rows = needed).
 * n_reject counts anchors refused by the border test (Orientator.py:131-135).
 */
int orc_orient(const float *gx, const float *gy, const float *gz, int nx, int ny, int nz,
               int octave, const int32_t *coords, int n, int r, int lim_main, int lim_sec,
               const double *bounds, const double *centers, int Z,
               int32_t *row_anchor, int32_t *row_main, int32_t *row_sec, double *row_R,
               int32_t *row_count, int64_t *n_rows, int64_t cap, int32_t *n_reject) {
    const int B = 2 * r + 1, nv = B * B * B;
    const int stride = (octave == 1) ? 1 : 2;
    float *box = (float *)malloc(sizeof(float) * 3 * nv);
    int32_t *w = (int32_t *)malloc(sizeof(int32_t) * nv);
    int64_t rows = 0;
    int32_t rejects = 0;
    const float two_pi_f = (float)ORC_TWO_PI;
    const float cutoff = 1e-5f;

    /* Orientator.py:38-47: sphere mask, weight 1 where |offset| <= 1.05 r */
    int32_t *mask = (int32_t *)malloc(sizeof(int32_t) * nv);
    for (int i = 0; i < B; i++) for (int j = 0; j < B; j++) for (int k = 0; k < B; k++) {
        int sq = (i - r) * (i - r) + (j - r) * (j - r) + (k - r) * (k - r);
        mask[(i * B + j) * B + k] = (sqrt((double)sq) <= r * 1.05) ? 1 : 0;
    }

    for (int a = 0; a < n; a++) {
        int x = coords[3 * a], y = coords[3 * a + 1], z = coords[3 * a + 2];
        /* step01, Orientator.py:128-135 / 149-155 */
        int xm = x - r * stride, ym = y - r * stride, zm = z - r * stride;
        int xp = x + r * stride + 1, yp = y + r * stride + 1, zp = z + r * stride + 1;
        if (xm < 0 || ym < 0 || zm < 0 || xp > nx - 1 || yp > ny - 1 || zp > nz - 1) { rejects++; continue; }
        for (int i = 0; i < B; i++) for (int j = 0; j < B; j++) for (int k = 0; k < B; k++) {
            size_t src = ((size_t)(xm + i * stride) * ny + (size_t)(ym + j * stride)) * nz + (size_t)(zm + k * stride);
            int v = (i * B + j) * B + k;
            float fx = gx[src], fy = gy[src], fz = gz[src];
            /* Orientator.py:139: float32 sqrt(sum(square)) */
            volatile float sx = fx * fx, sy = fy * fy, sz = fz * fz;
            volatile float s1 = sx + sy;
            volatile float s2 = s1 + sz;
            float magn = sqrtf(s2);
            if (magn > cutoff) { fx = fx / magn; fy = fy / magn; fz = fz / magn; } /* :142-143 */
            box[3 * v] = fx; box[3 * v + 1] = fy; box[3 * v + 2] = fz;
            w[v] = (magn < cutoff) ? 0 : mask[v];                                  /* :146-147 */
        }
        /* step02: process_df_gradient on the float32 box, Orientator.py:305-340 */
        int32_t cnt[ORC_MAX_Z], q[ORC_MAX_Z];
        memset(cnt, 0, sizeof(cnt));
        for (int v = 0; v < nv; v++) {
            if (!w[v]) continue;
            float th = (float)atan2((double)box[3 * v + 1], (double)box[3 * v]);
            if (th < 0.0f) th = th + two_pi_f;
            float sth = th + two_pi_f;
            float ph = (float)acos(orc_clamp1((double)box[3 * v + 2]));
            orc_count_zones((double)th, (double)sth, (double)ph, bounds, Z, w[v], cnt);
        }
        if (!orc_quantise(cnt, Z, q)) continue;   /* no bins > 0.8*0 -> no rows */
        /* Orientator.py:181-184 */
        int mains[ORC_MAX_Z], nmain = 0;
        for (int i = 0; i < Z; i++) if ((double)q[i] > 50 * 0.8) mains[nmain++] = i;
        if (nmain > lim_main) continue;

        for (int mi = 0; mi < nmain; mi++) {
            int mb = mains[mi];
            double dom[9];
            int32_t q1[ORC_MAX_Z];
            orc_to_dom_mat(centers, mb, dom);
            if (mb != 0) {
                /* step03: rotate the float32 box by dom (-> float64) and re-bin, :204-206, :303 */
                int32_t c1[ORC_MAX_Z];
                memset(c1, 0, sizeof(c1));
                for (int v = 0; v < nv; v++) {
                    if (!w[v]) continue;
                    double g0 = box[3 * v], g1 = box[3 * v + 1], g2 = box[3 * v + 2];
                    double rx = g0 * dom[0] + g1 * dom[1] + g2 * dom[2];
                    double ry = g0 * dom[3] + g1 * dom[4] + g2 * dom[5];
                    double rz = g0 * dom[6] + g1 * dom[7] + g2 * dom[8];
                    double th = atan2(ry, rx);
                    if (th < 0) th += ORC_TWO_PI;
                    double sth = th + ORC_TWO_PI;
                    double ph = acos(orc_clamp1(rz));
                    orc_count_zones(th, sth, ph, bounds, Z, w[v], c1);
                }
                if (!orc_quantise(c1, Z, q1)) memcpy(q1, c1, sizeof(int32_t) * Z);
            } else {
                memcpy(q1, q, sizeof(int32_t) * Z);   /* :211 no re-binning for the pole */
            }
            /* step04, Orientator.py:228-239 */
            int32_t mx = 0;
            for (int i = 1; i < Z - 1; i++) if (q1[i] > mx) mx = q1[i];
            if (mx == 0) continue;
            int secs[ORC_MAX_Z], nsec = 0;
            for (int i = 1; i < Z - 1; i++) {
                int32_t q2 = (int32_t)((double)q1[i] / (double)mx * 50.0);
                if ((double)q2 > 50 * 0.8) secs[nsec++] = i;
            }
            if (nsec > lim_sec) continue;
            for (int si = 0; si < nsec; si++) {
                if (rows < cap) {
                    double adj[9];
                    orc_adj_sec_mat(bounds, centers, Z, secs[si], adj);
                    row_anchor[rows] = a; row_main[rows] = mb; row_sec[rows] = secs[si];
                    orc_mat3_mul(adj, dom, row_R + 9 * rows);                  /* :105 */
                    if (row_count) memcpy(row_count + (size_t)Z * rows, q1, sizeof(int32_t) * Z);
                }
                rows++;
            }
        }
    }
    free(box); free(w); free(mask);
    *n_rows = rows;
    if (n_reject) *n_reject = rejects;
    return rows > cap ? -1 : 0;
}

/* ---- the same with Orientator(gw_sig != 0): a Gaussian window on the orientation histogram (Orientator.py:49-54).  The zone
 * counts are then float64 sums of the weights of the voxels in each zone: df.weight_mask[area_mask] selects them in C order
 * (all voxels of the box, weight 0 included) and np.sum adds them pairwise (first element + numpy's blocked pairwise sum of the
 * rest), which orc_np_sum restates. ---- */
static double orc_pairwise(const double *a, int n) {
    if (n < 8) { double res = 0.0; for (int i = 0; i < n; i++) res += a[i]; return res; }
    if (n <= 128) {
        double r[8];
        int i;
        for (int j = 0; j < 8; j++) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8) for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return orc_pairwise(a, n2) + orc_pairwise(a + n2, n - n2);
}
static double orc_np_sum(const double *a, int n) { return n <= 0 ? 0.0 : a[0] + orc_pairwise(a + 1, n - 1); }

static void orc_list_zones(double th, double sth, double ph, const double *bounds, int Z, double w, double *lists, int *ln, int stride) {
    for (int a = 0; a < Z; a++) {
        const double *b = bounds + 4 * a;
        int thm = (th < b[2]) && (th > b[0]);
        int sthm = (sth < b[2]) && (sth > b[0]);
        int phm = (ph < b[3]) && (ph > b[1]);
        if ((thm || sthm) && phm) lists[(size_t)a * stride + ln[a]++] = w;
    }
}

int orc_orient_gw(const float *gx, const float *gy, const float *gz, int nx, int ny, int nz,
               int octave, const int32_t *coords, int n, int r, int lim_main, int lim_sec,
               const double *bounds, const double *centers, int Z,
               int32_t *row_anchor, int32_t *row_main, int32_t *row_sec, double *row_R,
               int32_t *row_count, int64_t *n_rows, int64_t cap, int32_t *n_reject, double gw_sig) {
    const int B = 2 * r + 1, nv = B * B * B;
    const int stride = (octave == 1) ? 1 : 2;
    float *box = (float *)malloc(sizeof(float) * 3 * nv);
    double *w = (double *)malloc(sizeof(double) * nv);
    double *lists = (double *)malloc(sizeof(double) * (size_t)Z * nv);
    int ln[ORC_MAX_Z];
    int64_t rows = 0;
    int32_t rejects = 0;
    const float two_pi_f = (float)ORC_TWO_PI;
    const float cutoff = 1e-5f;

    /* Orientator.py:38-47: sphere mask, weight 1 where |offset| <= 1.05 r */
    /* :49-54 Gaussian window exp(-d^2 / (2 sigma^2)) times the sphere mask (ones when gw_sig == 0) */
    double *mask = (double *)malloc(sizeof(double) * nv);
    for (int i = 0; i < B; i++) for (int j = 0; j < B; j++) for (int k = 0; k < B; k++) {
        int sq = (i - r) * (i - r) + (j - r) * (j - r) + (k - r) * (k - r);
        double gwt = gw_sig != 0.0 ? exp(-1.0 * ((double)sq / (2.0 * (gw_sig * gw_sig)))) : 1.0;
        mask[(i * B + j) * B + k] = (sqrt((double)sq) <= r * 1.05) ? gwt : 0.0;
    }

    for (int a = 0; a < n; a++) {
        int x = coords[3 * a], y = coords[3 * a + 1], z = coords[3 * a + 2];
        /* step01, Orientator.py:128-135 / 149-155 */
        int xm = x - r * stride, ym = y - r * stride, zm = z - r * stride;
        int xp = x + r * stride + 1, yp = y + r * stride + 1, zp = z + r * stride + 1;
        if (xm < 0 || ym < 0 || zm < 0 || xp > nx - 1 || yp > ny - 1 || zp > nz - 1) { rejects++; continue; }
        for (int i = 0; i < B; i++) for (int j = 0; j < B; j++) for (int k = 0; k < B; k++) {
            size_t src = ((size_t)(xm + i * stride) * ny + (size_t)(ym + j * stride)) * nz + (size_t)(zm + k * stride);
            int v = (i * B + j) * B + k;
            float fx = gx[src], fy = gy[src], fz = gz[src];
            /* Orientator.py:139: float32 sqrt(sum(square)) */
            volatile float sx = fx * fx, sy = fy * fy, sz = fz * fz;
            volatile float s1 = sx + sy;
            volatile float s2 = s1 + sz;
            float magn = sqrtf(s2);
            if (magn > cutoff) { fx = fx / magn; fy = fy / magn; fz = fz / magn; } /* :142-143 */
            box[3 * v] = fx; box[3 * v + 1] = fy; box[3 * v + 2] = fz;
            w[v] = (magn < cutoff) ? 0.0 : mask[v];                                  /* :146-147 */
        }
        /* step02: process_df_gradient on the float32 box, Orientator.py:305-340 */
        int32_t cnt[ORC_MAX_Z], q[ORC_MAX_Z];
        memset(ln, 0, sizeof(ln));
        for (int v = 0; v < nv; v++) {
            float th = (float)atan2((double)box[3 * v + 1], (double)box[3 * v]);
            if (th < 0.0f) th = th + two_pi_f;
            float sth = th + two_pi_f;
            float ph = (float)acos(orc_clamp1((double)box[3 * v + 2]));
            orc_list_zones((double)th, (double)sth, (double)ph, bounds, Z, w[v], lists, ln, nv);
        }
        /* :334 df.ar_count is an int32 array (DensityFeature.py:50): the float64 sum is truncated when it is stored */
        for (int zz = 0; zz < Z; zz++) cnt[zz] = (int32_t)orc_np_sum(lists + (size_t)zz * nv, ln[zz]);
        if (!orc_quantise(cnt, Z, q)) continue;   /* no bins > 0.8*0 -> no rows */
        /* Orientator.py:181-184 */
        int mains[ORC_MAX_Z], nmain = 0;
        for (int i = 0; i < Z; i++) if ((double)q[i] > 50 * 0.8) mains[nmain++] = i;
        if (nmain > lim_main) continue;

        for (int mi = 0; mi < nmain; mi++) {
            int mb = mains[mi];
            double dom[9];
            int32_t q1[ORC_MAX_Z];
            orc_to_dom_mat(centers, mb, dom);
            if (mb != 0) {
                /* step03: rotate the float32 box by dom (-> float64) and re-bin, :204-206, :303 */
                int32_t c1[ORC_MAX_Z];
                memset(ln, 0, sizeof(ln));
                for (int v = 0; v < nv; v++) {
                    double g0 = box[3 * v], g1 = box[3 * v + 1], g2 = box[3 * v + 2];
                    double rx = g0 * dom[0] + g1 * dom[1] + g2 * dom[2];
                    double ry = g0 * dom[3] + g1 * dom[4] + g2 * dom[5];
                    double rz = g0 * dom[6] + g1 * dom[7] + g2 * dom[8];
                    double th = atan2(ry, rx);
                    if (th < 0) th += ORC_TWO_PI;
                    double sth = th + ORC_TWO_PI;
                    double ph = acos(orc_clamp1(rz));
                    orc_list_zones(th, sth, ph, bounds, Z, w[v], lists, ln, nv);
                }
                for (int zz = 0; zz < Z; zz++) c1[zz] = (int32_t)orc_np_sum(lists + (size_t)zz * nv, ln[zz]);
                if (!orc_quantise(c1, Z, q1)) memcpy(q1, c1, sizeof(int32_t) * Z);
            } else {
                memcpy(q1, q, sizeof(int32_t) * Z);   /* :211 no re-binning for the pole */
            }
            /* step04, Orientator.py:228-239 */
            int32_t mx = 0;
            for (int i = 1; i < Z - 1; i++) if (q1[i] > mx) mx = q1[i];
            if (mx == 0) continue;
            int secs[ORC_MAX_Z], nsec = 0;
            for (int i = 1; i < Z - 1; i++) {
                int32_t q2 = (int32_t)((double)q1[i] / (double)mx * 50.0);
                if ((double)q2 > 50 * 0.8) secs[nsec++] = i;
            }
            if (nsec > lim_sec) continue;
            for (int si = 0; si < nsec; si++) {
                if (rows < cap) {
                    double adj[9];
                    orc_adj_sec_mat(bounds, centers, Z, secs[si], adj);
                    row_anchor[rows] = a; row_main[rows] = mb; row_sec[rows] = secs[si];
                    orc_mat3_mul(adj, dom, row_R + 9 * rows);                  /* :105 */
                    if (row_count) memcpy(row_count + (size_t)Z * rows, q1, sizeof(int32_t) * Z);
                }
                rows++;
            }
        }
    }
    free(box); free(w); free(mask); free(lists);
    *n_rows = rows;
    if (n_reject) *n_reject = rejects;
    return rows > cap ? -1 : 0;
}

/* ------------------------------------------------------------------ */
/* a9-a10: descriptor generation                                       */
/* ------------------------------------------------------------------ */

/*
 * Descriptor.step06_distribute_subeqsp (Descriptor.py:123-202) for n rows of one
 * octave.  coords = integer voxel position of the row's anchor in its octave,
 * R = Rfinal.  dsc[n][64*Z] int16, sub-cube id j*16+i*4+k (Descriptor.py:44-64),
 * zone fastest.  S = 2r samples per axis (16).
 */
/* Descriptor.py:44-93: which of the dsc_size sub-regions holds lattice point (i, j, k) (indices along axes 0, 1, 2 of the
 * S^3 sample cube), in the order of the reference's sub_slices lists: the third axis runs fastest, then the first, then the
 * second (64 and 27); the 8-region list has its own order; 1 = the whole cube. */
static int orc_sub_region(int i, int j, int k, int S, int dsc_size) {
    if (dsc_size == 64) { const int q = S / 4; return (j / q) * 16 + (i / q) * 4 + (k / q); }
    if (dsc_size == 27) {
        const int a = S / 3, b = 2 * S / 3;      /* fl//3, 2*fl//3 with fl = 2*dr = S */
        const int bi = i < a ? 0 : (i < b ? 1 : 2), bj = j < a ? 0 : (j < b ? 1 : 2), bk = k < a ? 0 : (k < b ? 1 : 2);
        return bj * 9 + bi * 3 + bk;
    }
    if (dsc_size == 8) {
        const int h = S / 2;
        const int bi = i < h ? 0 : 1, bj = j < h ? 0 : 1, bk = k < h ? 0 : 1;
        return bi * 4 + bj * 2 + (1 - bk);      /* (s1,s1,s2), (s1,s1,s1), (s1,s2,s2), (s1,s2,s1), (s2,...) */
    }
    return 0;
}

int orc_describe_sized(const float *gx, const float *gy, const float *gz, int nx, int ny, int nz,
                       int octave, const int32_t *coords, const double *R, int64_t n, int r,
                       const double *bounds, int Z, int dsc_size, int16_t *dsc);

int orc_describe(const float *gx, const float *gy, const float *gz, int nx, int ny, int nz,
                 int octave, const int32_t *coords, const double *R, int64_t n, int r,
                 const double *bounds, int Z, int16_t *dsc) {
    return orc_describe_sized(gx, gy, gz, nx, ny, nz, octave, coords, R, n, r, bounds, Z, 64, dsc);
}

int orc_describe_sized(const float *gx, const float *gy, const float *gz, int nx, int ny, int nz,
                       int octave, const int32_t *coords, const double *R, int64_t n, int r,
                       const double *bounds, int Z, int dsc_size, int16_t *dsc) {
    const int S = 2 * r;
    const int D = dsc_size * Z;
    if (dsc_size != 64 && dsc_size != 27 && dsc_size != 8 && dsc_size != 1) return -2;
    const float cut_norm = 1e-12f, cut_zero = 1e-5f;
    int *zone = (int *)malloc(sizeof(int) * S * S * S);
    for (int64_t row = 0; row < n; row++) {
        const double *Rf = R + 9 * row;
        double inv[9];
        int16_t *out = dsc + (size_t)D * row;
        memset(out, 0, sizeof(int16_t) * D);
        orc_mat3_inv(Rf, inv);
        double c0 = coords[3 * row], c1 = coords[3 * row + 1], c2 = coords[3 * row + 2];
        int oob = 0;
        for (int i = 0; i < S && !oob; i++) for (int j = 0; j < S && !oob; j++) for (int k = 0; k < S; k++) {
            /* Descriptor.py:34-35 lattices; :132-133 P = L @ inv(R).T + c */
            double l0, l1, l2;
            if (octave == 0) { l0 = -2 * r + 1 + 2 * i; l1 = -2 * r + 1 + 2 * j; l2 = -2 * r + 1 + 2 * k; }
            else { l0 = -r + 0.5 + i; l1 = -r + 0.5 + j; l2 = -r + 0.5 + k; }
            double p[3];
            p[0] = (l0 * inv[0] + l1 * inv[1] + l2 * inv[2]) + c0;
            p[1] = (l0 * inv[3] + l1 * inv[4] + l2 * inv[5]) + c1;
            p[2] = (l0 * inv[6] + l1 * inv[7] + l2 * inv[8]) + c2;
            /* scipy RegularGridInterpolator(method="nearest", bounds_error=True), MapSpace.py:189 */
            int dims[3] = { nx, ny, nz }, idx[3];
            for (int d = 0; d < 3; d++) {
                if (!(p[d] >= 0.0) || !(p[d] <= (double)(dims[d] - 1))) { oob = 1; break; }
                int ii = (int)floor(p[d]);
                if (ii > dims[d] - 2) ii = dims[d] - 2;
                if (ii < 0) ii = 0;
                double nd = p[d] - (double)ii;
                idx[d] = (nd <= 0.5) ? ii : ii + 1;
            }
            if (oob) break;
            size_t src = ((size_t)idx[0] * ny + idx[1]) * nz + idx[2];
            float fx = gx[src], fy = gy[src], fz = gz[src];
            volatile float sx = fx * fx, sy = fy * fy, sz = fz * fz;
            volatile float s1 = sx + sy;
            volatile float s2 = s1 + sz;
            float magn = sqrtf(s2);                                           /* :150 */
            if (magn > cut_norm) { fx = fx / magn; fy = fy / magn; fz = fz / magn; } /* :153-154 */
            double g0 = fx, g1 = fy, g2 = fz;                                   /* :155 g @ R.T */
            double rx = g0 * Rf[0] + g1 * Rf[1] + g2 * Rf[2];
            double ry = g0 * Rf[3] + g1 * Rf[4] + g2 * Rf[5];
            double rz = g0 * Rf[6] + g1 * Rf[7] + g2 * Rf[8];
            double th = atan2(ry, rx);
            if (th < 0) th += ORC_TWO_PI;
            double sth = th + ORC_TWO_PI;
            double ph = acos(orc_clamp1(rz));
            int zn = orc_zone(th, sth, ph, bounds, Z);
            if (zn < 0) zn = 0;                                                 /* :173 default zone 0 */
            if (magn < cut_zero) zn = -1;                                       /* :190 */
            zone[(i * S + j) * S + k] = zn;
        }
        if (oob) continue;                                                      /* :142-149 zero descriptor */
        for (int i = 0; i < S; i++) for (int j = 0; j < S; j++) for (int k = 0; k < S; k++) {
            int zn = zone[(i * S + j) * S + k];
            if (zn < 0) continue;
            int sub = orc_sub_region(i, j, k, S, dsc_size);
            out[sub * Z + zn]++;
        }
    }
    free(zone);
    return 0;
}

/* ------------------------------------------------------------------ */
/* a11: correlation + threshold                                        */
/* ------------------------------------------------------------------ */

/*
 * MaD._match_dsc part 1 (MaD.py:416-424).  Rows are L2-normalised in float64
 * (zero rows stay zero), preds = hi @ lo.T, pairs = where(preds > cc) row-major.
 * scores (optional, n_hi*n_lo) receives the full matrix.
 */
int orc_correlate(const int16_t *hi, int64_t n_hi, const int16_t *lo, int64_t n_lo, int D, double cc,
                  double *scores, int32_t *pair_hi, int32_t *pair_lo, double *pair_score,
                  int64_t *n_pairs, int64_t cap) {
    double *hn = (double *)malloc(sizeof(double) * (size_t)n_hi * D);
    double *ln = (double *)malloc(sizeof(double) * (size_t)n_lo * D);
    for (int pass = 0; pass < 2; pass++) {
        const int16_t *src = pass ? lo : hi;
        double *dst = pass ? ln : hn;
        int64_t nr = pass ? n_lo : n_hi;
        for (int64_t i = 0; i < nr; i++) {
            double s = 0;
            for (int k = 0; k < D; k++) s += (double)src[i * D + k] * (double)src[i * D + k];
            double nrm = sqrt(s);
            for (int k = 0; k < D; k++) dst[i * D + k] = nrm > 0 ? (double)src[i * D + k] / nrm : (double)src[i * D + k];
        }
    }
    int64_t np_ = 0;
    for (int64_t i = 0; i < n_hi; i++) for (int64_t j = 0; j < n_lo; j++) {
        double s = 0;
        const double *a = hn + i * D, *b = ln + j * D;
        for (int k = 0; k < D; k++) s += a[k] * b[k];
        if (scores) scores[i * n_lo + j] = s;
        if (s > cc) {
            if (np_ < cap) { pair_hi[np_] = (int32_t)i; pair_lo[np_] = (int32_t)j; pair_score[np_] = s; }
            np_++;
        }
    }
    free(hn); free(ln);
    *n_pairs = np_;
    return np_ > cap ? -1 : 0;
}

/* ------------------------------------------------------------------ */
/* a12: pose scoring (repeatability)                                   */
/* ------------------------------------------------------------------ */

/*
 * MaD._match_dsc part 2 (MaD.py:433-451).  hi_cloud / lo_cloud are the unique
 * sub-voxel anchor coordinates of the rows that appear in at least one pair
 * (MaD.py:427-428; the caller does the np.unique).  For each pair:
 *   R = inv(lo.R) @ hi.R; cloud' = (hi_cloud - hi.p) @ R.T + lo.p;
 *   repeat = 100 * #{nearest lo point closer than dist} / l.
 * results[n_pairs][23] as MaD.py:451.  Brute-force nearest neighbour.
 */
int orc_pose_score(const int32_t *pair_hi, const int32_t *pair_lo, const double *pair_score, int64_t n_pairs,
                   const double *hi_p, const double *hi_R, const int32_t *hi_meta,
                   const double *lo_p, const double *lo_R, const int32_t *lo_meta,
                   const double *hi_cloud, int64_t l_hi, const double *lo_cloud, int64_t l_lo,
                   double dist, double *results, int32_t *counts) {
    for (int64_t p = 0; p < n_pairs; p++) {
        int64_t ih = pair_hi[p], il = pair_lo[p];
        double inv[9], R[9];
        orc_mat3_inv(lo_R + 9 * il, inv);
        orc_mat3_mul(inv, hi_R + 9 * ih, R);
        const double *ph = hi_p + 3 * ih, *pl = lo_p + 3 * il;
        int32_t cnt = 0;
        for (int64_t a = 0; a < l_hi; a++) {
            double d0 = hi_cloud[3 * a] - ph[0], d1 = hi_cloud[3 * a + 1] - ph[1], d2 = hi_cloud[3 * a + 2] - ph[2];
            double x = (d0 * R[0] + d1 * R[1] + d2 * R[2]) + pl[0];
            double y = (d0 * R[3] + d1 * R[4] + d2 * R[5]) + pl[1];
            double z = (d0 * R[6] + d1 * R[7] + d2 * R[8]) + pl[2];
            double best = INFINITY;
            for (int64_t b = 0; b < l_lo; b++) {
                double e0 = lo_cloud[3 * b] - x, e1 = lo_cloud[3 * b + 1] - y, e2 = lo_cloud[3 * b + 2] - z;
                double dd = e0 * e0 + e1 * e1 + e2 * e2;
                if (dd < best) best = dd;
            }
            if (sqrt(best) < dist) cnt++;
        }
        if (counts) counts[p] = cnt;
        if (results) {
            double *o = results + 23 * p;
            o[0] = pair_score[p];
            o[1] = 100.0 * (double)cnt / (double)l_hi;
            o[2] = lo_meta[3 * il]; o[3] = lo_meta[3 * il + 1]; o[4] = lo_meta[3 * il + 2];
            o[5] = hi_meta[3 * ih]; o[6] = hi_meta[3 * ih + 1]; o[7] = hi_meta[3 * ih + 2];
            o[8] = ph[0]; o[9] = ph[1]; o[10] = ph[2];
            o[11] = pl[0]; o[12] = pl[1]; o[13] = pl[2];
            memcpy(o + 14, R, sizeof(double) * 9);
        }
    }
    return 0;
}

/*
 * Order of MaD._filter_dsc_pairs' stable sort (MaD.py:480): repeatability
 * descending, ties in input (row-major pair) order.  Writes the first k indices.
 */
int orc_topk(const int32_t *counts, int64_t n, int64_t k, int64_t *order) {
    /* simple stable selection: counting sort on the integer count */
    int32_t mx = 0;
    for (int64_t i = 0; i < n; i++) if (counts[i] > mx) mx = counts[i];
    int64_t *start = (int64_t *)calloc((size_t)mx + 2, sizeof(int64_t));
    for (int64_t i = 0; i < n; i++) start[mx - counts[i] + 1]++;
    for (int32_t c = 0; c <= mx; c++) start[c + 1] += start[c];
    for (int64_t i = 0; i < n; i++) {
        int64_t pos = start[mx - counts[i]]++;
        if (pos < k) order[pos] = i;
    }
    free(start);
    return 0;
}

/* ------------------------------------------------------------------ */
/* a13: rigid-body refinement                                          */
/* ------------------------------------------------------------------ */

/* np.gradient with unit spacing on a float32 volume, one axis (numpy function_base):
 * interior (f[i+1]-f[i-1])/2, one-sided first-order differences at the ends. */
static float orc_grad1(const float *g, int nx, int ny, int nz, int x, int y, int z, int axis) {
    int n = axis == 0 ? nx : (axis == 1 ? ny : nz);
    int i = axis == 0 ? x : (axis == 1 ? y : z);
    size_t st = axis == 0 ? (size_t)ny * nz : (axis == 1 ? (size_t)nz : 1);
    size_t c = ((size_t)x * ny + y) * nz + z;
    if (i == 0) { volatile float d = g[c + st] - g[c]; return d / 1.0f; }
    if (i == n - 1) { volatile float d = g[c] - g[c - st]; return d / 1.0f; }
    volatile float d = g[c + st] - g[c - st];
    return d / 2.0f;
}

/*
 * structure_utils.refine_pdb (structure_utils.py:58-161).  grid = dmap.grid3d
 * float32 [nx][ny][nz]; origin/voxsp in Angstrom; coords double[n][3] updated in
 * place.  Returns 0; *converged, *last_step as the reference's (converged, step).
 * trace (optional) receives per executed step [trans(3), rot(9), step_size] = 13 doubles.
 */
int orc_refine(const float *grid, int nx, int ny, int nz, double ox, double oy, double oz, double vs,
               double *coords, int64_t n, int n_steps, double max_step, double min_step,
               int32_t *converged, int32_t *last_step, double *trace) {
    double *init = (double *)malloc(sizeof(double) * 3 * n);
    double *prev = (double *)malloc(sizeof(double) * 3 * n);
    memcpy(init, coords, sizeof(double) * 3 * n);
    memcpy(prev, coords, sizeof(double) * 3 * n);
    /* :66-67 centroid (np.mean) and farthest atom */
    double cen[3] = { 0, 0, 0 };
    for (int64_t i = 0; i < n; i++) { cen[0] += init[3 * i]; cen[1] += init[3 * i + 1]; cen[2] += init[3 * i + 2]; }
    cen[0] /= (double)n; cen[1] /= (double)n; cen[2] /= (double)n;
    double maxd = 0;
    for (int64_t i = 0; i < n; i++) {
        double a = init[3 * i] - cen[0], b = init[3 * i + 1] - cen[1], c = init[3 * i + 2] - cen[2];
        double d = sqrt(a * a + b * b + c * c);
        if (d > maxd) maxd = d;
    }
    /* :76-78 np.arange grid coordinates: p[i] = o + i * ((o + vs) - o) */
    double o3[3] = { ox, oy, oz }, del[3];
    int dims[3] = { nx, ny, nz };
    for (int d = 0; d < 3; d++) del[d] = (o3[d] + vs) - o3[d];
    double trans[3] = { 0, 0, 0 };
    double rot[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    double step_size = max_step;
    int batch = 0, conv = 0, step = 0;
    for (step = 0; step < n_steps; step++) {
        /* :91-96 */
        double ct[3] = { cen[0] + trans[0], cen[1] + trans[1], cen[2] + trans[2] };
        int has_nan = 0;
        for (int64_t i = 0; i < n; i++) {
            double a = init[3 * i] - cen[0], b = init[3 * i + 1] - cen[1], c = init[3 * i + 2] - cen[2];
            double x = a * rot[0] + b * rot[3] + c * rot[6];
            double y = a * rot[1] + b * rot[4] + c * rot[7];
            double z = a * rot[2] + b * rot[5] + c * rot[8];
            coords[3 * i] = x + ct[0]; coords[3 * i + 1] = y + ct[1]; coords[3 * i + 2] = z + ct[2];
            if (coords[3 * i] != coords[3 * i] || coords[3 * i + 1] != coords[3 * i + 1] || coords[3 * i + 2] != coords[3 * i + 2]) has_nan = 1;
        }
        if (has_nan) { *converged = 0; *last_step = step; free(init); free(prev); return 1; }
        double sg[3] = { 0, 0, 0 }, tq[3] = { 0, 0, 0 };
        for (int64_t i = 0; i < n; i++) {
            double p[3] = { coords[3 * i], coords[3 * i + 1], coords[3 * i + 2] };
            /* :101-103 strictly inside [o, o + (s-1) vs) */
            int inside = 1;
            for (int d = 0; d < 3; d++)
                if (!(p[d] > o3[d]) || !(p[d] < o3[d] + dims[d] * vs - vs)) inside = 0;
            if (!inside) continue;
            /* :106 trilinear RegularGridInterpolator on np.gradient(grid3d) */
            int i0[3]; double y[3];
            for (int d = 0; d < 3; d++) {
                int ii = (int)floor((p[d] - o3[d]) / vs);
                if (ii < 0) ii = 0;
                if (ii > dims[d] - 2) ii = dims[d] - 2;
                while (ii > 0 && p[d] < o3[d] + ii * del[d]) ii--;
                while (ii < dims[d] - 2 && p[d] >= o3[d] + (ii + 1) * del[d]) ii++;
                double g0 = o3[d] + ii * del[d], g1 = o3[d] + (ii + 1) * del[d];
                i0[d] = ii; y[d] = (p[d] - g0) / (g1 - g0);
            }
            double g[3] = { 0, 0, 0 };
            for (int cx = 0; cx < 2; cx++) for (int cy = 0; cy < 2; cy++) for (int cz = 0; cz < 2; cz++) {
                double wgt = 1.0;
                wgt = wgt * (cx ? y[0] : 1 - y[0]);
                wgt = wgt * (cy ? y[1] : 1 - y[1]);
                wgt = wgt * (cz ? y[2] : 1 - y[2]);
                for (int ax = 0; ax < 3; ax++)
                    g[ax] = g[ax] + (double)orc_grad1(grid, nx, ny, nz, i0[0] + cx, i0[1] + cy, i0[2] + cz, ax) * wgt;
            }
            sg[0] += g[0]; sg[1] += g[1]; sg[2] += g[2];
            /* :121-122 torque = sum cross(g, x - centre) */
            double c0 = p[0] - cen[0], c1 = p[1] - cen[1], c2 = p[2] - cen[2];
            tq[0] += g[1] * c2 - g[2] * c1;
            tq[1] += g[2] * c0 - g[0] * c2;
            tq[2] += g[0] * c1 - g[1] * c0;
        }
        if (!(step % 2)) {
            /* :111-116 translation step */
            double u[3];
            orc_unit(sg, u);
            for (int d = 0; d < 3; d++) { u[d] *= step_size; trans[d] += u[d]; }
            for (int64_t i = 0; i < n; i++) { coords[3 * i] += u[0]; coords[3 * i + 1] += u[1]; coords[3 * i + 2] += u[2]; }
        } else {
            /* :123-138 rotation step about centre + trans */
            double ax[3], sm[9], nr[9];
            orc_unit(tq, ax);
            orc_rod(ax, step_size / maxd, sm);
            double ctr[3] = { cen[0] + trans[0], cen[1] + trans[1], cen[2] + trans[2] };
            double neg[3] = { -1 * cen[0] - trans[0], -1 * cen[1] - trans[1], -1 * cen[2] - trans[2] };
            for (int64_t i = 0; i < n; i++) {
                double a = coords[3 * i] + neg[0], b = coords[3 * i + 1] + neg[1], c = coords[3 * i + 2] + neg[2];
                double x = a * sm[0] + b * sm[3] + c * sm[6];
                double y = a * sm[1] + b * sm[4] + c * sm[7];
                double z = a * sm[2] + b * sm[5] + c * sm[8];
                coords[3 * i] = x + ctr[0]; coords[3 * i + 1] = y + ctr[1]; coords[3 * i + 2] = z + ctr[2];
            }
            orc_mat3_mul(rot, sm, nr);
            memcpy(rot, nr, sizeof(rot));
        }
        /* :141-147 step halving every 4 steps */
        batch++;
        if (batch == 4) {
            double mn = 0;
            for (int64_t i = 0; i < n; i++) {
                double a = prev[3 * i] - coords[3 * i], b = prev[3 * i + 1] - coords[3 * i + 1], c = prev[3 * i + 2] - coords[3 * i + 2];
                double d = sqrt(a * a + b * b + c * c);
                if (d > mn) mn = d;
            }
            if (mn < step_size) step_size *= 0.5;
            batch = 0;
            memcpy(prev, coords, sizeof(double) * 3 * n);
        }
        if (trace) {
            double *t = trace + 13 * step;
            t[0] = trans[0]; t[1] = trans[1]; t[2] = trans[2];
            memcpy(t + 3, rot, sizeof(rot));
            t[12] = step_size;
        }
        if (step_size < min_step) { conv = 1; break; }                             /* :150-152 */
    }
    if (step == n_steps) step = n_steps - 1;   /* python's loop variable after exhaustion */
    *converged = conv; *last_step = step;
    free(init); free(prev);
    return 0;
}

/* ------------------------------------------------------------------ */
/* a14-a15: density simulation                                         */
/* ------------------------------------------------------------------ */

/*
 * PDB.interpolate_to_grid_massweighted (PDB.py:215-292).  dims_out = pxb,pyb,pzb;
 * min_out = lattice-aligned minx,miny,minz.  Call once with grid == NULL to get
 * the dimensions, then with a zeroed double[pxb*pyb*pzb] (x fastest: index
 * pxb*pyb*k + pxb*j + i, PDB.py:217-218).
 */
int orc_splat(const double *atoms, const double *mass, int64_t n, double vs, int pad,
              int32_t dims_out[3], double min_out[3], double *grid) {
    double mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (int64_t i = 0; i < n; i++) for (int d = 0; d < 3; d++) {
        if (atoms[3 * i + d] < mn[d]) mn[d] = atoms[3 * i + d];
        if (atoms[3 * i + d] > mx[d]) mx[d] = atoms[3 * i + d];
    }
    int margin = 2 + pad;
    for (int d = 0; d < 3; d++) {
        mn[d] = vs * floor(mn[d] / vs);
        mx[d] = vs * ceil(mx[d] / vs);
        dims_out[d] = (int32_t)ceil((mx[d] - mn[d]) / vs) + 2 * margin + 1;
        min_out[d] = mn[d];
    }
    if (!grid) return 0;
    int pxb = dims_out[0], pyb = dims_out[1];
    for (int64_t i = 0; i < n; i++) {
        double gxp = margin + (atoms[3 * i] - mn[0]) / vs;
        double gyp = margin + (atoms[3 * i + 1] - mn[1]) / vs;
        double gzp = margin + (atoms[3 * i + 2] - mn[2]) / vs;
        int x0 = (int)floor(gxp), y0 = (int)floor(gyp), z0 = (int)floor(gzp);
        int x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
        double a = x1 - gxp, b = y1 - gyp, c = z1 - gzp, m = mass[i];
#define IDZ(k, j, i) ((size_t)pxb * pyb * (k) + (size_t)pxb * (j) + (i))
        grid[IDZ(z0, y0, x0)] += m * a * b * c;
        grid[IDZ(z1, y0, x0)] += m * a * b * (1 - c);
        grid[IDZ(z0, y1, x0)] += m * a * (1 - b) * c;
        grid[IDZ(z0, y0, x1)] += m * (1 - a) * b * c;
        grid[IDZ(z1, y1, x0)] += m * a * (1 - b) * (1 - c);
        grid[IDZ(z0, y1, x1)] += m * (1 - a) * (1 - b) * c;
        grid[IDZ(z1, y0, x1)] += m * (1 - a) * b * (1 - c);
        grid[IDZ(z1, y1, x1)] += m * (1 - a) * (1 - b) * (1 - c);
#undef IDZ
    }
    size_t nv = (size_t)dims_out[0] * dims_out[1] * dims_out[2];
    double gm = -INFINITY;
    for (size_t i = 0; i < nv; i++) if (grid[i] > gm) gm = grid[i];
    for (size_t i = 0; i < nv; i++) grid[i] = grid[i] / gm;                       /* :290 */
    return 0;
}

/*
 * PDB.structure_to_density (PDB.py:131-163): full 3-D convolution of the splat
 * grid (x fastest as above, read as [x][y][z] through order='F') with the
 * normalised Gaussian of sigma = res/(pi*sqrt(2))/vs truncated at ceil(3 sigma);
 * float32 result / max, values < isovalue zeroed.  out is [ox][oy][oz] C order
 * (z fastest) with o* = p* + 2r.  Returns r through *r_out.
 */
int orc_blur(const double *splat, const int32_t dims[3], double resolution, double vs, double isovalue,
             float *out, int32_t *r_out) {
    double sig = resolution / (ORC_PI * sqrt(2.0)) / vs;
    int r = (int)ceil(3.0 * sig);
    *r_out = r;
    if (!out) return 0;
    int K = 2 * r + 1;
    double *h = (double *)malloc(sizeof(double) * K * K * K), hs = 0;
    for (int z = -r; z <= r; z++) for (int y = -r; y <= r; y++) for (int x = -r; x <= r; x++) {
        double v = exp(-(double)(x * x + y * y + z * z) / (2.0 * sig * sig));
        h[((z + r) * K + (y + r)) * K + (x + r)] = v; hs += v;
    }
    for (int i = 0; i < K * K * K; i++) h[i] /= hs;
    int px = dims[0], py = dims[1], pz = dims[2];
    int ox = px + 2 * r, oy = py + 2 * r, oz = pz + 2 * r;
    size_t no = (size_t)ox * oy * oz;
    double *acc = (double *)calloc(no, sizeof(double));
    /* the kernel is symmetric, so axis naming of h is immaterial */
    for (int x = 0; x < px; x++) for (int y = 0; y < py; y++) for (int z = 0; z < pz; z++) {
        double v = splat[(size_t)px * py * z + (size_t)px * y + x];
        if (v == 0.0) continue;
        for (int a = 0; a < K; a++) for (int b = 0; b < K; b++) {
            double *dst = acc + ((size_t)(x + a) * oy + (y + b)) * oz + z;
            const double *hk = h + (a * K + b) * K;
            for (int c = 0; c < K; c++) dst[c] += v * hk[c];
        }
    }
    float mxv = -INFINITY;
    for (size_t i = 0; i < no; i++) { out[i] = (float)acc[i]; if (out[i] > mxv) mxv = out[i]; }
    for (size_t i = 0; i < no; i++) { out[i] = out[i] / mxv; if (out[i] < (float)isovalue) out[i] = 0.0f; }
    free(acc); free(h);
    return 0;
}

/* ------------------------------------------------------------------ */
/* a16: cross-correlation of two grids on their overlap box            */
/* ------------------------------------------------------------------ */

static long orc_pyround(double v) { return (long)nearbyint(v); } /* python round(): half to even */

/*
 * Dmap.get_CCC_with_grid (Dmap.py:153-258).  Both grids float32 [x][y][z] C order.
 * Negative/below-isovalue voxels are zeroed IN PLACE in both (Dmap.py:160-161).
 * Un-centred normalised dot product over the overlap box, float32 accumulation
 * as np.dot on float32 vectors; returns 0 when the box is inverted (:232-234).
 */
double orc_ccc(float *g1, const int32_t d1[3], const double o1[3],
               float *g2, const int32_t d2[3], const double o2[3], double vs, double isovalue) {
    size_t n1 = (size_t)d1[0] * d1[1] * d1[2], n2 = (size_t)d2[0] * d2[1] * d2[2];
    for (size_t i = 0; i < n1; i++) if (g1[i] < (float)isovalue) g1[i] = 0;
    for (size_t i = 0; i < n2; i++) if (g2[i] < (float)isovalue) g2[i] = 0;
    long mn1[3], mn2[3], mx1[3], mx2[3];
    for (int d = 0; d < 3; d++) {
        double a = o1[d] / vs, b = o2[d] / vs;
        if (a > b) { mn1[d] = 0; mn2[d] = orc_pyround(a - b); }
        else if (a < b) { mn1[d] = orc_pyround(b - a); mn2[d] = 0; }
        else { mn1[d] = 0; mn2[d] = 0; }
        if (a + d1[d] > b + d2[d]) { mx1[d] = orc_pyround(b + d2[d] - a); mx2[d] = d2[d]; }
        else if (a + d1[d] < b + d2[d]) { mx1[d] = d1[d]; mx2[d] = orc_pyround(a + d1[d] - b); }
        else { mx1[d] = d1[d]; mx2[d] = d2[d]; }
    }
    for (int d = 0; d < 3; d++) if (mx1[d] - mn1[d] < 0) return 0.0;
    /* python slicing clamps to the array and to empty */
    long e[3];
    for (int d = 0; d < 3; d++) {
        long a0 = mn1[d] < 0 ? 0 : mn1[d], a1 = mx1[d] > d1[d] ? d1[d] : mx1[d];
        long b0 = mn2[d] < 0 ? 0 : mn2[d], b1 = mx2[d] > d2[d] ? d2[d] : mx2[d];
        long ea = a1 - a0 > 0 ? a1 - a0 : 0, eb = b1 - b0 > 0 ? b1 - b0 : 0;
        e[d] = ea < eb ? ea : eb;   /* shapes agree in every non-degenerate case */
        mn1[d] = a0; mn2[d] = b0;
    }
    double olap = 0, na = 0, nb = 0;
    for (long x = 0; x < e[0]; x++) for (long y = 0; y < e[1]; y++) for (long z = 0; z < e[2]; z++) {
        float a = g1[((size_t)(mn1[0] + x) * d1[1] + (mn1[1] + y)) * d1[2] + (mn1[2] + z)];
        float b = g2[((size_t)(mn2[0] + x) * d2[1] + (mn2[1] + y)) * d2[2] + (mn2[2] + z)];
        olap += (double)a * b; na += (double)a * a; nb += (double)b * b;
    }
    return olap / sqrt(na * nb);
}
