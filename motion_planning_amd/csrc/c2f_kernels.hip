// Reference-line tables and batched Cartesian->Frenet conversion for gfx950.
//
// Reference: Model/cartesian_frenet_conversion.py:11-70 on top of
// Model/curve.py:13-186.  cartesian_to_frenet projects a point on the NEAREST
// PRE-SAMPLED point of the reference spline (no refinement), so everything it
// returns besides the distance is a function of the sample index alone.
// mp_refline_tables therefore builds, once per scenario, the sample set and the
// arc length / heading / curvature at every sample, following the reference's
// own (quirky) evaluation rules:
//   - the sample ordinates come from Curve.calc_point_arr, which switches
//     spline segment only once a sample lies more than dt past the segment end;
//   - heading and curvature come from the scalar Curve.calc_point (get_i);
//   - arc length is the rectangle rule over end-inclusive sample sets, one
//     piece per spline segment, added up in segment order.
// mp_c2f_batch is then a nearest-sample search plus table look-ups.
//
// The whole file is compiled without floating-point contraction so that the
// operation order written here (the reference's) is the order executed.
#pragma clang fp contract(off)
#include <atomic>

#include "mp_device.h"
#include "c2f_device.h"
#include "plan_glue.h"

#define RL_SEG_CAP 128 // spline segments per reference line
#define RL_ARC_CAP 128 // samples in one rectangle-rule piece
#define RL_FT 16       // per segment: integrand values at 0, dt, 2dt, ... kept in LDS

// Model/curve.py:80-91 get_i: first segment whose running end reaches t.
__device__ __forceinline__ void curve_get_i(const double *T, int n, double t, int *ii, double *tt)
{
    double t_sum = 0.0;
    for (int i = 0; i < n; ++i) {
        t_sum += T[i];
        if (t <= t_sum) {
            *ii = i;
            *tt = t - t_sum + T[i];
            return;
        }
    }
    *ii = n - 1;
    *tt = t_sum;
}

// curve_get_i with the running sums it accumulates kept in a table: Tseq[i] =
// (((T[0] + T[1]) + ...) + T[i]), added in that order.  Positive segment lengths
// make the table ascend, and the first i with t <= Tseq[i] is found by bisection;
// otherwise by the same scan as curve_get_i.
__device__ __forceinline__ void curve_get_i_tab(const double *Tseq, const double *T, int n, bool ascending,
                                                double t, int *ii, double *tt)
{
    int i;
    if (ascending) {
        int lo = 0, hi = n; // first index with t <= Tseq[index], n if none
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (t <= Tseq[mid]) hi = mid; else lo = mid + 1;
        }
        i = lo;
    } else {
        for (i = 0; i < n; ++i)
            if (t <= Tseq[i]) break;
    }
    if (i < n) {
        *ii = i;
        *tt = t - Tseq[i] + T[i];
    } else {
        *ii = n - 1;
        *tt = Tseq[n - 1];
    }
}

// Model/curve.py:175-186 QuinticPoly.calc_arc_len for t0 <= t1; *bad is set
// when the piece has more than RL_ARC_CAP samples.  ftab (may be null): the
// integrand sqrt(1 + x'(t)^2) of this segment at t = 0, dt, 2dt, ... (RL_FT
// values).  A piece that starts at 0 samples exactly those abscissae plus its own
// end point, so it takes them from the table and evaluates only the end point;
// the sum runs in the same order either way.
__device__ double quintic_arc_len(const double *a, double t0, double t1, double dt, int *bad,
                                  const double *ftab = nullptr)
{
    int n;
    double delta;
    const int K = arange_count(t0, t1, dt, RL_ARC_CAP, &n, &delta);
    if (K < 0) { *bad = 1; return 0.0; }
    if (K == 0) return 0.0;
    auto integrand = [&](int i) {
        double d = quintic_eval(a, arange_value(i, n, t0, t1, dt, delta), 1);
        return __dsqrt_rn(1 + d * d);
    };
    double s;
    if (ftab && t0 == 0.0 && n <= RL_FT) {
        const double f_end = (K > n) ? integrand(n) : 0.0; // the appended end point
        s = np_sum_gen(K, [&](int i) { return i < n ? ftab[i] : f_end; });
    } else {
        s = np_sum_gen(K, integrand);
    }
    return 1 * dt * s;
}

struct RlDev {
    int32_t n_knots_max, m_max;
    const double *knots_x, *knots_y;
    const int32_t *n_knots;
    const double *vec, *ego_xy;
    double d_x;
    double *tab_x, *tab_y, *tab_s, *tab_theta, *tab_kappa;
    int32_t *M;
    double *origin;
    int32_t *status;
    double *inv_x, *inv_s, *inv_y, *inv_theta; // Frenet->Cartesian table, optional
    int32_t *n_inv;
};

// Everything Curve.calc_arc_len(rx_ori, x) (curve.py:93-115) shares between calls,
// in LDS: segment lengths and their running sums, spline coefficients, the
// integrand table, and arc_pre[e] = the length from the origin (segment st_i, local
// abscissa st_t) up to the start of segment e, added up in segment order.
struct ArcCtx {
    const double *T, *Tseq, *coef, *ftab, *arc_pre;
    int n, st_i;
    bool ascending;
    double st_t, t0, dt;
};

__device__ double arc_from_origin(const ArcCtx &c, double t1, int *bad)
{
    if (c.t0 > t1) return 0.0;
    int en_i;
    double en_t;
    curve_get_i_tab(c.Tseq, c.T, c.n, c.ascending, t1, &en_i, &en_t);
    if (en_t > c.T[en_i]) en_t = c.T[en_i];
    if (c.st_i == en_i) return quintic_arc_len(c.coef + 6 * c.st_i, c.st_t, en_t, c.dt, bad);
    return c.arc_pre[en_i] + quintic_arc_len(c.coef + 6 * en_i, 0, en_t, c.dt, bad, c.ftab + RL_FT * en_i);
}

// Doubles of dynamic LDS per spline segment slot (ns = n_knots_max - 1 slots): T, seg_full, Tseq,
// Scum (1 each), coef (6), ftab (RL_FT), and vv, aa, arc_pre with one slot more each.
__host__ __device__ __forceinline__ size_t rl_seg_doubles(int ns) { return (size_t)ns * (10 + RL_FT) + 3 * ((size_t)ns + 1); }

// One workgroup per scenario.  Dynamic LDS, sized by the batch's knot slots so that short
// reference lines leave room for many workgroups per CU: the per-segment arrays (see
// rl_seg_doubles), then tsum[m_max] (double) seg[m_max] (int).
__global__ void refline_tables_kernel(RlDev a)
{
    extern __shared__ double smem[];
    const int ns = min(a.n_knots_max - 1, RL_SEG_CAP); // (a line with more segments is refused below)
    double *T = smem, *coef = T + ns, *seg_full = coef + 6 * ns;
    double *vv = seg_full + ns, *aa = vv + ns + 1;
    double *ftab = aa + ns + 1;        // arc-length integrand of every segment at 0, dt, 2dt, ...
    double *Tseq = ftab + RL_FT * ns;  // running sums of T in segment order
    double *Scum = Tseq + ns;          // running end of the current segment in calc_point_arr's walk
    double *arc_pre = Scum + ns;       // arc length from the origin to the start of a segment
    __shared__ double arc0[RL_ARC_CAP];         // integrand values of the origin's piece
    __shared__ double red_c[16];
    __shared__ int red_i[16];
    __shared__ double sc_d[8]; // t_max, delta, first piece, s_end
    __shared__ int sc_i[8];    // M, n samples before the appended one, status, k_ori, ascending, parallel walk, start segment
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    double *tsum = arc_pre + ns + 1;
    int *seg = (int *)(tsum + a.m_max);
    const int npts = a.n_knots[b];
    const int n = npts - 1;
    const double *kx = a.knots_x + (size_t)b * a.n_knots_max;
    const double *ky = a.knots_y + (size_t)b * a.n_knots_max;
    const double dt = a.d_x;
    double *tab_x = a.tab_x + (size_t)b * a.m_max, *tab_y = a.tab_y + (size_t)b * a.m_max;
    double *tab_s = a.tab_s + (size_t)b * a.m_max, *tab_th = a.tab_theta + (size_t)b * a.m_max;
    double *tab_kr = a.tab_kappa + (size_t)b * a.m_max;

    if (n < 1 || n > RL_SEG_CAP || npts > a.n_knots_max) {
        if (tid == 0) { a.status[b] = MP_RL_E_CAPACITY; a.M[b] = 0; }
        return;
    }
    const double t_bios = kx[0];
    // Curve.__init__ (curve.py:22-38)
    for (int i = tid; i < n; i += nt) T[i] = kx[i + 1] - kx[i];
    __syncthreads();
    for (int i = tid; i <= n; i += nt)
        vv[i] = (i == 0 || i == n) ? a.vec[b] : 1.0 * (ky[i + 1] - ky[i - 1]) / (T[i] + T[i - 1]);
    __syncthreads();
    for (int i = tid; i <= n; i += nt)
        aa[i] = (i == 0 || i == n) ? 0.0 : 1.0 * (vv[i + 1] - vv[i - 1]) / (T[i] + T[i - 1]);
    __syncthreads();
    for (int i = tid; i < n; i += nt)
        quintic_coef(ky[i], vv[i], aa[i], ky[i + 1], vv[i + 1], aa[i + 1], T[i], coef + 6 * i);
    __syncthreads();
    // integrand table: abscissae as get_arange(0, ., dt) places them (0, 0+dt, k*((0+dt)-0))
    for (int idx = tid; idx < n * RL_FT; idx += nt) {
        const int i = idx / RL_FT, q = idx - i * RL_FT;
        const double t = arange_value(q, RL_FT, 0.0, 0.0, dt, (0.0 + dt) - 0.0);
        const double d = quintic_eval(coef + 6 * i, t, 1);
        ftab[idx] = __dsqrt_rn(1 + d * d);
    }
    // projection sample set (curve.py:120-123) and the lagged segment walk of
    // calc_point_arr (curve.py:43-66): the walk moves on by at most one segment per
    // sample, when the sample lies more than dt past the running end of the current
    // one.  With every segment longer than the sample spacing a sample never passes
    // two ends at once, so the segment of sample j is the start segment plus the
    // number of ends (+ dt) below it, which every sample finds on its own; shorter
    // segments take the sequential walk.
    if (tid == 0) {
        int st = MP_RL_OK;
        double run = 0.0;
        bool ascending = true, long_segs = true;
        for (int q = 0; q < n; ++q) {
            run += T[q];
            Tseq[q] = run;
            Scum[q] = run; // the walk's running ends when it starts in segment 0 (else redone below)
            if (!(T[q] > 0.0)) ascending = false;
            if (q >= 1 && !(T[q] > dt * (1.0 + 1e-9))) long_segs = false;
        }
        const double t_max = np_sum_gen(n, [&](int i) { return T[i]; });
        int na;
        double delta;
        int M = arange_count(0.0, t_max, dt, a.m_max, &na, &delta);
        if (M < 0) { st = MP_RL_E_CAPACITY; M = 0; }
        else if (M == 0) { st = MP_RL_E_INDEX; }
        sc_d[0] = t_max;
        sc_d[1] = delta;
        sc_i[1] = na;
        const bool parallel_walk = ascending && long_segs;
        int i0 = 0;
        if (M > 0) {
            // t_arr - t_bios of element 0 is (0 + t_bios) - t_bios
            double tt;
            curve_get_i(T, n, (0.0 + t_bios) - t_bios, &i0, &tt);
            double t_sum = np_sum_gen(i0 + 1, [&](int q) { return T[q]; });
            if (parallel_walk && i0 == 0) {
                // Scum already holds these very additions
            } else if (parallel_walk) {
                Scum[i0] = t_sum;
                for (int q = i0 + 1; q < n; ++q) { t_sum += T[q]; Scum[q] = t_sum; }
            } else {
                int i = i0;
                for (int j = 0; j < M; ++j) {
                    const double tj = (arange_value(j, na, 0.0, t_max, dt, delta) + t_bios) - t_bios;
                    if (tj > t_sum + dt) {
                        i += 1;
                        if (i < n) t_sum += T[i];
                        else { st = MP_RL_E_INDEX; M = 0; break; }
                    }
                    seg[j] = i;
                    tsum[j] = t_sum;
                }
            }
        }
        sc_i[0] = M;
        sc_i[2] = st;
        sc_i[4] = ascending;
        sc_i[5] = parallel_walk;
        sc_i[6] = i0;
    }
    __syncthreads();
    const int M = sc_i[0];
    const bool ascending = sc_i[4] != 0;
    {
        int past_end = 0;
        if (sc_i[2] == MP_RL_OK && sc_i[5]) {
            const double t_max_ = sc_d[0], delta_ = sc_d[1];
            const int na_ = sc_i[1], i0 = sc_i[6];
            for (int j = tid; j < M; j += nt) {
                const double tj = (arange_value(j, na_, 0.0, t_max_, dt, delta_) + t_bios) - t_bios;
                int i = i0;
                while (i < n && tj > Scum[i] + dt) ++i;
                if (i < n) { seg[j] = i; tsum[j] = Scum[i]; }
                else past_end = 1; // the walk would step past the last segment: IndexError
            }
        }
        past_end = __syncthreads_or(past_end);
        if (past_end && tid == 0) sc_i[2] = MP_RL_E_INDEX;
        __syncthreads();
    }
    if (sc_i[2] != MP_RL_OK) {
        if (tid == 0) { a.status[b] = sc_i[2]; a.M[b] = 0; }
        return;
    }
    const double t_max = sc_d[0], delta = sc_d[1];
    const int na = sc_i[1];
    // ordinates, heading, curvature per sample; nearest sample to the ego point
    const double ex = a.ego_xy[2 * b], ey = a.ego_xy[2 * b + 1];
    double best = MP_INF;
    int bi = 0x7fffffff;
    for (int k = tid; k < M; k += nt) {
        const double x = arange_value(k, na, 0.0, t_max, dt, delta) + t_bios;
        const int i = seg[k];
        const double y = quintic_eval(coef + 6 * i, (x - t_bios) - tsum[k] + T[i], 0);
        tab_x[k] = x;
        tab_y[k] = y;
        int ii;
        double tt;
        curve_get_i_tab(Tseq, T, n, ascending, x - t_bios, &ii, &tt);
        const double dl = quintic_eval(coef + 6 * ii, tt, 1);
        const double ddl = quintic_eval(coef + 6 * ii, tt, 2);
        tab_th[k] = atan(dl);
        // (1 + y'^2)^1.5 as w * sqrt(w): within an ulp or two of the library power, like the
        // device's own pow(), at a tenth of its instructions
        const double w = 1 + dl * dl;
        tab_kr[k] = 1.0 * fabs(ddl) / (w * __dsqrt_rn(w));
        const double dx = x - ex, dy = y - ey;
        const double d = __dsqrt_rn(dx * dx + dy * dy);
        if (d < best) { best = d; bi = k; }
    }
    group_argmin<MP_WAVE>(best, bi);
    if ((tid & (MP_WAVE - 1)) == 0) { red_c[tid / MP_WAVE] = best; red_i[tid / MP_WAVE] = bi; }
    __syncthreads();
    int bad = 0;
    if (tid == 0) {
        for (int w = 1; w < (nt + MP_WAVE - 1) / MP_WAVE; ++w) argmin_merge(best, bi, red_c[w], red_i[w]);
        sc_i[3] = bi;
    }
    __syncthreads();
    const int k_ori = sc_i[3];
    const double rx_ori = arange_value(k_ori, na, 0.0, t_max, dt, delta) + t_bios;
    // get_s pieces shared by every sample (curve.py:93-110): the origin's
    // segment from its local start to the segment end, and every full segment
    const double t0 = rx_ori - t_bios;
    int st_i;
    double st_t;
    curve_get_i_tab(Tseq, T, n, ascending, t0, &st_i, &st_t);
    for (int i = tid; i < n; i += nt) seg_full[i] = quintic_arc_len(coef + 6 * i, 0, T[i], dt, &bad, ftab + RL_FT * i);
    // the origin's segment from st_t to its end (quintic_arc_len's sum): the integrand values
    // by as many lanes, added up by one in that function's order
    int n0;
    double delta0;
    const int K0 = arange_count(st_t, T[st_i], dt, RL_ARC_CAP, &n0, &delta0);
    if (K0 < 0) bad = 1;
    for (int q = tid; q < K0; q += nt) {
        const double d = quintic_eval(coef + 6 * st_i, arange_value(q, n0, st_t, T[st_i], dt, delta0), 1);
        arc0[q] = __dsqrt_rn(1 + d * d);
    }
    __syncthreads();
    if (tid == 0) {
        // arc length up to the start of every later segment, in calc_arc_len's order:
        // the origin's segment from st_t to its end, then the full segments one by one
        double arc = K0 > 0 ? 1 * dt * np_sum_gen(K0, [&](int q) { return arc0[q]; }) : 0.0;
        for (int e = st_i + 1; e < n; ++e) {
            arc_pre[e] = arc;
            arc += seg_full[e];
        }
    }
    __syncthreads();
    ArcCtx ac{T, Tseq, coef, ftab, arc_pre, n, st_i, ascending, st_t, t0, dt};
    for (int k = tid; k < M; k += nt) {
        const double x = arange_value(k, na, 0.0, t_max, dt, delta) + t_bios;
        tab_s[k] = arc_from_origin(ac, x - t_bios, &bad);
    }
    // get_s of the last knot: the length SLMap lays its layers along (sl_map.py:52)
    // (by the last lane: the first wave carries every single-lane step above)
    const double rx_max = kx[n];
    if (tid == nt - 1)
        sc_d[3] = arc_from_origin(ac, rx_max - t_bios, &bad);
    // Frenet->Cartesian table (cfc.py:55-60, 72-86): abscissae
    // get_arange(rx_ori, rx_max, d_x) with their arc length, ordinate and heading
    int n_inv = 0;
    if (a.inv_x) {
        int ng;
        double dg;
        n_inv = arange_count(rx_ori, rx_max, dt, a.m_max, &ng, &dg);
        if (n_inv < 0) { bad = 1; n_inv = 0; }
        for (int k = tid; k < n_inv; k += nt) {
            const double x = arange_value(k, ng, rx_ori, rx_max, dt, dg);
            int ii;
            double tt;
            curve_get_i_tab(Tseq, T, n, ascending, x - t_bios, &ii, &tt);
            const size_t o = (size_t)b * a.m_max + k;
            a.inv_x[o] = x;
            a.inv_s[o] = arc_from_origin(ac, x - t_bios, &bad);
            a.inv_y[o] = quintic_eval(coef + 6 * ii, tt, 0);
            a.inv_theta[o] = atan(quintic_eval(coef + 6 * ii, tt, 1));
        }
    }
    bad = __syncthreads_or(bad);
    if (tid == 0) {
        a.status[b] = bad ? MP_RL_E_CAPACITY : MP_RL_OK;
        a.M[b] = bad ? 0 : M;
        a.origin[4 * b] = rx_ori;
        a.origin[4 * b + 1] = tab_y[k_ori];
        a.origin[4 * b + 2] = sc_d[3];
        a.origin[4 * b + 3] = rx_max;
        if (a.n_inv) a.n_inv[b] = bad ? 0 : n_inv;
    }
}

struct C2fDev {
    int32_t n_pts, m_max;
    const double *px, *py, *vx, *vy;
    const int32_t *pt_scn;
    const double *tab_x, *tab_y, *tab_s, *tab_theta, *tab_kappa;
    const int32_t *M;
    double *s, *l, *s_dt;
};

// One thread per query point (cartesian_frenet_conversion.py:32-53).
__global__ void c2f_kernel(C2fDev a)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.n_pts) return;
    const int b = a.pt_scn[p];
    const int M = a.M[b];
    const double x = a.px[p], y = a.py[p];
    const double *tx = a.tab_x + (size_t)b * a.m_max, *ty = a.tab_y + (size_t)b * a.m_max;
    if (M <= 0) {
        const double nan = __builtin_nan("");
        a.s[p] = nan;
        a.l[p] = nan;
        if (a.s_dt) a.s_dt[p] = nan;
        return;
    }
    const size_t tb = (size_t)b * a.m_max;
    double s, l;
    const int k = c2f_point(tx, ty, a.tab_s + tb, a.tab_theta + tb, M, x, y, &s, &l);
    a.s[p] = s;
    a.l[p] = l;
    if (a.s_dt) a.s_dt[p] = c2f_s_dt(a.tab_theta[tb + k], a.tab_kappa[tb + k], l, a.vx[p], a.vy[p]);
}

struct F2cDev {
    int32_t n_pts, m_max;
    const double *s, *l;
    const int32_t *pt_scn;      // scene of each point, or null: point p belongs to scene p / pts_per_scn
    int32_t pts_per_scn;
    const int32_t *n_valid;     // with pts_per_scn: points of a scene beyond n_valid[scene] give 0 (may be null)
    const double *inv_x, *inv_s, *inv_y, *inv_theta, *origin;
    const int32_t *n_inv;
    double *x, *y;
    // whole-replan call only (null otherwise): the scene's outcome and what the two spline
    // evaluations before this kernel reported
    int32_t *plan_status;
    const int32_t *prof_status, *final_status, *n_prof;
    // whole replan from world-frame scenes only (null otherwise): the ego pose of every scene
    // and where the points go in the world frame, R p + t (robot_map.py:84-91)
    const double *rot, *trans;
    double *wx, *wy;
};

// GeneralConverter.robot_to_world (general_converter.py:19-22), products accumulated as
// robot_to_world_kernel does
__device__ __forceinline__ void f2c_to_world(const F2cDev &a, const int p, const int b, double px, double py)
{
    const double *r = a.rot + 4 * (size_t)b;
    a.wx[p] = __dadd_rn(fma(r[1], py, __dmul_rn(r[0], px)), a.trans[2 * b]);
    a.wy[p] = __dadd_rn(fma(r[3], py, __dmul_rn(r[2], px)), a.trans[2 * b + 1]);
}

// One thread per point (cartesian_frenet_conversion.py:55-60 with get_rxy
// :72-86): the first table abscissa whose arc length reaches s; past the table
// the end of the line, whose ordinate/heading sit in the last slot +1.
// arc: the scene's arc-length table (null: read it from a.inv_s); ascending: the table was
// checked to never decrease, so the first entry that reaches s is found by bisection.
__device__ __forceinline__ void f2c_one(const F2cDev &a, const int p, const int b, const double *arc_tab = nullptr,
                                        const bool ascending = false)
{
    if (a.plan_status && p == b * a.pts_per_scn && a.plan_status[b] == MP_PLAN_OK) {
        // a scene whose profile or final path points were not produced must not read as
        // planned (PlannerInterface.run_step would raise IndexError from path_sampling,
        // planner_interface.py:157, curve.py:57-65)
        const int ps = a.prof_status[b], fs = a.final_status[b];
        if (ps == MP_RL_E_CAPACITY || fs == MP_RL_E_CAPACITY) a.plan_status[b] = MP_PLAN_E_ST_CAPACITY;
        else if (ps != MP_RL_OK || fs != MP_RL_OK || a.n_valid[b] != a.n_prof[b] || a.n_prof[b] <= 0)
            a.plan_status[b] = MP_PLAN_E_INDEX;
    }
    if (!a.pt_scn && a.n_valid && p - b * a.pts_per_scn >= a.n_valid[b]) {
        a.x[p] = a.y[p] = 0.0;
        if (a.wx) a.wx[p] = a.wy[p] = 0.0;
        return;
    }
    const int n = a.n_inv[b];
    const double s = a.s[p], l = a.l[p];
    const double *arc = arc_tab ? arc_tab : a.inv_s + (size_t)b * a.m_max;
    int k = n - 1; // the table ends on rx_max (get_arange appends it)
    if (ascending) {
        int lo = 0, hi = n; // first index with s <= arc[index], n if none (also for a NaN s)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (s <= arc[mid]) hi = mid; else lo = mid + 1;
        }
        if (lo < n) k = lo;
    } else {
        for (int i = 0; i < n; ++i)
            if (s <= arc[i]) { k = i; break; }
    }
    if (n <= 0) {
        a.x[p] = a.y[p] = __builtin_nan("");
        if (a.wx) a.wx[p] = a.wy[p] = __builtin_nan("");
        return;
    }
    const size_t o = (size_t)b * a.m_max + k;
    const double th = a.inv_theta[o];
    const double px = a.inv_x[o] - l * sin(th), py = a.inv_y[o] + l * cos(th);
    a.x[p] = px;
    a.y[p] = py;
    if (a.wx) f2c_to_world(a, p, b, px, py);
}

__global__ void f2c_kernel(F2cDev a)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.n_pts) return;
    f2c_one(a, p, a.pt_scn ? a.pt_scn[p] : p / a.pts_per_scn);
}

struct CvDev {
    int32_t n_knots_max, q_max, mode;
    const double *knots_t, *knots_x;
    const int32_t *n_knots;
    const double *vec;
    double dt;
    const double *query;
    const int32_t *n_query;
    double *out_t, *out_x, *out_dx;
    int32_t *n_out, *status;
};

// Spline through knots + evaluation on an increasing sample set with
// Curve.calc_point_arr's rules (curve.py:13-73).  One workgroup per curve.
//   mode 0: samples get_arange(t0, t0 + np.sum(T), dt)   (SLMap.path_sampling, sl_map.py:115-118)
//   mode 1: samples get_arange(t0, t_last, dt)           (SpeedPlanner.plan, speed_planner.py:47)
//   mode 2: samples given by the caller                  (path_sampling(curve, False, ss))
// Dynamic LDS, sized by the curve's knot slots (ns = n_knots_max - 1 segment slots) so that short
// splines leave room for many workgroups per CU: T[ns] coef[6 ns] vv[ns + 1] aa[ns + 1] Scum[ns],
// then tq[q_max] tsum[q_max] (double) seg[q_max] (int); cv_lds_bytes gives the total.
// (the body of curve_eval_kernel for curve b with npts knots; every thread of the
// workgroup calls it, and a workgroup may call it again after a barrier)
__host__ __device__ __forceinline__ int cv_seg_slots(int n_knots_max)
{
    return n_knots_max - 1 < RL_SEG_CAP ? (n_knots_max > 1 ? n_knots_max - 1 : 1) : RL_SEG_CAP;
}
__host__ __device__ __forceinline__ size_t cv_lds_bytes(int n_knots_max, int q_max)
{
    return ((size_t)10 * cv_seg_slots(n_knots_max) + 2 + 2 * (size_t)q_max) * sizeof(double) + (size_t)q_max * sizeof(int);
}

__device__ __forceinline__ void curve_eval_scene(const CvDev &a, const int b, const int npts, double *smem)
{
    __shared__ double sc_d[2];          // fill step and end of the sample set
    __shared__ int sc_i[4];             // n samples, status, start segment / samples on the grid, ascending
    const int tid = threadIdx.x, nt = blockDim.x;
    const int ns = cv_seg_slots(a.n_knots_max);
    double *T = smem, *coef = T + ns, *vv = coef + 6 * ns, *aa = vv + ns + 1;
    double *Scum = aa + ns + 1; // running end of the current segment in calc_point_arr's walk
    double *tq = Scum + ns, *tsum = tq + a.q_max;
    int *seg = (int *)(tsum + a.q_max);
    const int n = npts - 1;
    const double *kt = a.knots_t + (size_t)b * a.n_knots_max;
    const double *kx = a.knots_x + (size_t)b * a.n_knots_max;
    if (n < 1 || n > RL_SEG_CAP || npts > a.n_knots_max) {
        if (tid == 0) { a.status[b] = n < 1 ? MP_RL_E_INDEX : MP_RL_E_CAPACITY; a.n_out[b] = 0; }
        return;
    }
    const double t_bios = kt[0], dt = a.dt;
    for (int i = tid; i < n; i += nt) T[i] = kt[i + 1] - kt[i];
    __syncthreads();
    for (int i = tid; i <= n; i += nt)
        vv[i] = (i == 0 || i == n) ? a.vec[b] : 1.0 * (kx[i + 1] - kx[i - 1]) / (T[i] + T[i - 1]);
    __syncthreads();
    for (int i = tid; i <= n; i += nt)
        aa[i] = (i == 0 || i == n) ? 0.0 : 1.0 * (vv[i + 1] - vv[i - 1]) / (T[i] + T[i - 1]);
    __syncthreads();
    for (int i = tid; i < n; i += nt)
        quintic_coef(kx[i], vv[i], aa[i], kx[i + 1], vv[i + 1], aa[i + 1], T[i], coef + 6 * i);
    __syncthreads();
    // sample abscissae
    if (tid == 0) {
        int st = MP_RL_OK, m = 0, na = 0;
        double delta = 0.0, t_en = 0.0;
        if (a.mode == 2) {
            m = max(a.n_query[b], 0);
            if (m > a.q_max) { st = MP_RL_E_CAPACITY; m = 0; }
        } else {
            t_en = (a.mode == 0) ? t_bios + np_sum_gen(n, [&](int i) { return T[i]; }) : kt[n];
            m = arange_count(t_bios, t_en, dt, a.q_max, &na, &delta);
            if (m < 0) { st = MP_RL_E_CAPACITY; m = 0; }
        }
        if (st == MP_RL_OK && m == 0) st = MP_RL_E_INDEX; // t_arr[0] of an empty array
        sc_i[0] = m;
        sc_i[1] = st;
        sc_i[2] = na;
        sc_d[0] = delta;
        sc_d[1] = t_en;
    }
    __syncthreads();
    {
        const int m0 = sc_i[0], na = sc_i[2];
        const double delta = sc_d[0], t_en = sc_d[1];
        for (int j = tid; j < m0; j += nt)
            tq[j] = (a.mode == 2) ? a.query[(size_t)b * a.q_max + j] : arange_value(j, na, t_bios, t_en, dt, delta);
    }
    __syncthreads();
    // calc_point_arr's walk over the segments (curve.py:43-66): it moves on by at most one
    // segment per sample, when the sample lies more than dt past the running end of the
    // current one.  Every sample first counts the segment ends (+ dt) below it on its own;
    // if those counts never drop and never rise by more than one from sample to sample they
    // ARE the walk's segments, otherwise the walk is done sequentially.
    if (tid == 0 && sc_i[0] > 0) {
        int st = sc_i[1], m = sc_i[0], i0 = 0;
        const double first = tq[0] - t_bios;
        bool ascending = true;
        if (first < 0.0) { st = MP_RL_E_INDEX; m = 0; }
        else {
            double tt;
            curve_get_i(T, n, first, &i0, &tt);
            double t_sum = np_sum_gen(i0 + 1, [&](int q) { return T[q]; });
            Scum[i0] = t_sum;
            for (int q = i0 + 1; q < n; ++q) {
                t_sum += T[q];
                Scum[q] = t_sum;
                if (!(T[q] > 0.0)) ascending = false;
            }
        }
        sc_i[0] = m;
        sc_i[1] = st;
        sc_i[2] = i0;
        sc_i[3] = ascending;
    }
    __syncthreads();
    {
        const int m0 = sc_i[0], i0 = sc_i[2];
        int irregular = sc_i[3] ? 0 : 1;
        if (!irregular) {
            for (int j = tid; j < m0; j += nt) {
                const double tj = tq[j] - t_bios;
                int i = i0;
                while (i < n && tj > Scum[i] + dt) ++i;
                seg[j] = i;
            }
        }
        __syncthreads();
        if (!irregular) {
            for (int j = tid; j < m0; j += nt) {
                const int step = seg[j] - (j > 0 ? seg[j - 1] : i0);
                if (step < 0 || step > 1) irregular = 1;
            }
        }
        irregular = __syncthreads_or(irregular);
        if (!irregular) {
            int past_end = 0;
            for (int j = tid; j < m0; j += nt) {
                if (seg[j] < n) tsum[j] = Scum[seg[j]];
                else past_end = 1; // the walk steps past the last segment: IndexError
            }
            past_end = __syncthreads_or(past_end);
            if (past_end && tid == 0) { sc_i[0] = 0; sc_i[1] = MP_RL_E_INDEX; }
        } else if (tid == 0 && m0 > 0) {
            int i = i0, m = m0, st = sc_i[1];
            double t_sum = np_sum_gen(i + 1, [&](int q) { return T[q]; });
            for (int j = 0; j < m; ++j) {
                if (tq[j] - t_bios > t_sum + dt) {
                    i += 1;
                    if (i < n) t_sum += T[i];
                    else { st = MP_RL_E_INDEX; m = 0; break; }
                }
                seg[j] = i;
                tsum[j] = t_sum;
            }
            sc_i[0] = m;
            sc_i[1] = st;
        }
    }
    __syncthreads();
    const int m = sc_i[0];
    for (int j = tid; j < m; j += nt) {
        const int i = seg[j];
        const double tl = (tq[j] - t_bios) - tsum[j] + T[i];
        const size_t o = (size_t)b * a.q_max + j;
        a.out_t[o] = tq[j];
        a.out_x[o] = quintic_eval(coef + 6 * i, tl, 0);
        if (a.out_dx) a.out_dx[o] = quintic_eval(coef + 6 * i, tl, 1);
    }
    if (tid == 0) {
        a.n_out[b] = m;
        a.status[b] = sc_i[1];
    }
}

__global__ void curve_eval_kernel(CvDev a)
{
    extern __shared__ double smem[];
    curve_eval_scene(a, blockIdx.x, a.n_knots[blockIdx.x], smem);
}

// ---------------------------------------------------------------------------
// Stages of the whole replan that depend on one another scene by scene, one workgroup
// per scene (plan_glue.h).  A stage hands its arrays to the next one through global
// memory, as the separate kernels did; the barrier between two stages makes a
// workgroup's own writes visible to all of its threads.
// ---------------------------------------------------------------------------
__global__ void plan_path_kernel(PlanPathDev a)
{
    extern __shared__ double smem[];
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int NS = a.n_s_max;
    // outcome of the path search, read by every thread before anything is written
    int st = a.status[b];
    const int st_in = st;
    const int n = a.n_layers[b];
    if (st == MP_PLAN_OK && !a.track) {
        const int sl = a.sl_status[b];
        if (sl != MP_SL_FOUND) st = (sl == MP_SL_NOT_FOUND) ? MP_PLAN_NOT_FOUND : (sl < 0 ? sl : MP_PLAN_NOT_FOUND);
    }
    __syncthreads();
    int32_t *pidx = a.path_idx + (size_t)b * NS;
    if (a.track) {
        // the centre node of every layer, or nothing for a scene that failed earlier
        for (int s = tid; s < NS; s += nt) pidx[s] = (st_in == MP_PLAN_OK && s < n) ? a.mid_l : -1;
        if (tid == 0) {
            a.sl_status[b] = st_in == MP_PLAN_OK ? MP_SL_FOUND : MP_SL_SKIPPED;
            a.layers_done[b] = st_in == MP_PLAN_OK ? n : 0;
            a.best_cost[b] = __builtin_nan("");
        }
    }
    const int npts = st == MP_PLAN_OK ? n + 1 : 0;
    double *kt = a.knots_t + (size_t)b * (NS + 1), *kv = a.knots_v + (size_t)b * (NS + 1);
    if (st == MP_PLAN_OK) {
        const double *lay = a.s_layers + (size_t)b * NS;
        for (int s = tid; s <= n; s += nt) {
            kt[s] = s == 0 ? a.ego_sl[2 * b] : lay[s - 1];
            kv[s] = s == 0 ? a.ego_sl[2 * b + 1] : a.l_row[a.track ? a.mid_l : pidx[s - 1]];
        }
    }
    if (tid == 0) {
        a.status[b] = st;
        a.n_pk[b] = npts;
    }
    __syncthreads();
    // the path every path_dt; a scene without knots reports MP_RL_E_INDEX and no samples
    CvDev c{NS + 1, a.p_max, 0, a.knots_t, a.knots_v, a.n_pk, a.zeros, a.path_dt, nullptr, nullptr,
            a.path_s, a.path_l, nullptr, a.n_path, a.path_status};
    curve_eval_scene(c, b, npts, smem);
    __syncthreads();
    if (tid == 0) {
        int n_s = 0;
        if (st == MP_PLAN_OK) {
            const int P = a.n_path[b];
            if (a.path_status[b] != MP_RL_OK || P <= 0) st = MP_PLAN_E_ST_CAPACITY;
            else {
                const double s_end = a.path_s[(size_t)b * a.p_max + P - 1];
                const double q = floor(s_end / a.st_ds);
                if (!(q >= 0.0) || q + 1.0 > (double)a.st_n_s_max) st = MP_PLAN_E_ST_CAPACITY;
                else n_s = (int)q + 1;
            }
        }
        if (st != MP_PLAN_OK) a.n_path[b] = 0; // later stages skip the scene
        a.st_n_s[b] = n_s;
        a.status[b] = st;
    }
}

__global__ void plan_tail_kernel(PlanTailDev a)
{
    extern __shared__ double smem[];
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int n_t = a.n_t;
    int st = a.status[b];
    if (st == MP_PLAN_OK && a.reg_status[b] != MP_ST_OK) st = MP_PLAN_E_ST_CAPACITY;
    const int len = st == MP_PLAN_OK ? a.st_path_len[b] : 0;
    __syncthreads();
    if (tid == 0) {
        a.status[b] = st;
        if (st != MP_PLAN_OK) a.st_path_len[b] = 0;
    }
    if (st == MP_PLAN_OK) {
        for (int i = tid; i < n_t; i += nt) {
            double t = 0.0;
            for (int k = 0; k < i; ++k) t += a.st_dt; // t of knot i as the running sum reaches it
            const int j = a.st_path_s_ind[(size_t)b * n_t + i];
            a.sp_t[(size_t)b * n_t + i] = t;
            a.sp_s[(size_t)b * n_t + i] = i < len ? (double)j * a.st_ds : 0.0;
        }
    }
    __syncthreads();
    CvDev prof{n_t, a.q_max, 1, a.sp_t, a.sp_s, a.st_path_len, a.robot_vel, a.curve_dt, nullptr, nullptr,
               a.prof_t, a.ss, a.vv, a.n_prof, a.prof_status};
    curve_eval_scene(prof, b, len, smem);
    __syncthreads();
    CvDev fin{a.n_pk_max, a.q_max, 2, a.knots_t, a.knots_v, a.n_pk, a.zeros, a.path_dt, a.ss, a.n_prof,
              a.final_s, a.final_l, nullptr, a.n_final, a.final_status};
    curve_eval_scene(fin, b, a.n_pk[b], smem);
    __syncthreads();
    F2cDev f{0, a.m_max, a.final_s, a.final_l, nullptr, a.q_max, a.n_final, a.inv_x, a.inv_s, a.inv_y,
             a.inv_theta, nullptr, a.n_inv, a.final_x, a.final_y, a.status, a.prof_status, a.final_status,
             a.n_prof, a.rot, a.trans, a.world_x, a.world_y};
    // the scene's arc-length table once into LDS (the spline evaluations are done with it);
    // a table that never decreases -- every table of a sound reference line -- is bisected
    const int n_inv = a.n_inv[b];
    const bool staged = n_inv > 0 && n_inv <= a.m_max;
    int descends = 0;
    if (staged) {
        for (int i = tid; i < n_inv; i += nt) smem[i] = a.inv_s[(size_t)b * a.m_max + i];
        __syncthreads();
        for (int i = tid; i + 1 < n_inv; i += nt) descends |= !(smem[i] <= smem[i + 1]);
    }
    descends = __syncthreads_or(descends);
    for (int j = tid; j < a.q_max; j += nt)
        f2c_one(f, b * a.q_max + j, b, staged ? smem : nullptr, staged && !descends);
}

// ---------------------------------------------------------------------------
extern void mp_set_error(const char *fmt, ...);

static int c2f_launch_check(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        mp_set_error("%s: %s", what, hipGetErrorString(e));
        return MP_E_LAUNCH;
    }
    return MP_OK;
}

extern "C" int mp_refline_tables(int32_t B, const double *knots_x, const double *knots_y,
                                 const int32_t *n_knots, int32_t n_knots_max, const double *vec,
                                 const double *ego_xy, double d_x, int32_t m_max, double *tab_x, double *tab_y,
                                 double *tab_s, double *tab_theta, double *tab_kappa, int32_t *M, double *origin,
                                 int32_t *status, double *inv_x, double *inv_s, double *inv_y,
                                 double *inv_theta, int32_t *n_inv, void *stream)
{
    if (B <= 0 || n_knots_max < 2 || m_max <= 0 || !(d_x > 0)) { mp_set_error("bad sizes"); return MP_E_BADARG; }
    if (!knots_x || !knots_y || !n_knots || !vec || !ego_xy || !tab_x || !tab_y || !tab_s || !tab_theta ||
        !tab_kappa || !M || !origin || !status) {
        mp_set_error("null array");
        return MP_E_BADARG;
    }
    if (inv_x && (!inv_s || !inv_y || !inv_theta || !n_inv)) {
        mp_set_error("the Frenet->Cartesian table needs all five arrays");
        return MP_E_BADARG;
    }
    if (m_max > 4096) { mp_set_error("mp_refline_tables: m_max > 4096"); return MP_E_UNSUPPORTED; }
    RlDev a{n_knots_max, m_max, knots_x, knots_y, n_knots, vec, ego_xy, d_x, tab_x, tab_y, tab_s, tab_theta,
            tab_kappa, M, origin, status, inv_x, inv_s, inv_y, inv_theta, n_inv};
    const int ns = n_knots_max - 1 < RL_SEG_CAP ? n_knots_max - 1 : RL_SEG_CAP; // lines with more segments report MP_RL_E_CAPACITY
    size_t lds = rl_seg_doubles(ns) * sizeof(double) + (size_t)m_max * (sizeof(double) + sizeof(int));
    lds = (lds + 15) & ~(size_t)15;
    if (lds > 64 * 1024) { // long lines with many knot slots: beyond what a launch gets without asking
        static std::atomic<size_t> allowed{0};
        if (allowed.load(std::memory_order_relaxed) < lds) {
            if (hipFuncSetAttribute((const void *)refline_tables_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds) != hipSuccess) {
                (void)hipGetLastError();
                mp_set_error("mp_refline_tables: %zu bytes of LDS per line are not available", lds);
                return MP_E_UNSUPPORTED;
            }
            allowed.store(lds, std::memory_order_relaxed);
        }
    }
    // A large batch fills the GPU with lines: one wave per line then (its barriers cost nothing
    // and a CU holds four times as many lines); a small one wants its lines done soon.
    int threads = B >= 1024 ? 64 : (m_max <= 128 ? 128 : 256);
    if (mp_knob_rl_threads() > 0) threads = mp_knob_rl_threads(); // development aid
    hipLaunchKernelGGL(refline_tables_kernel, dim3(B), dim3(threads), lds, (hipStream_t)stream, a);
    return c2f_launch_check("mp_refline_tables");
}

extern "C" int mp_f2c_batch(int32_t n_pts, const double *s, const double *l, const int32_t *pt_scn,
                            int32_t m_max, const double *inv_x, const double *inv_s, const double *inv_y,
                            const double *inv_theta, const int32_t *n_inv, double *x, double *y, void *stream)
{
    if (n_pts <= 0 || m_max <= 0) { mp_set_error("bad sizes"); return MP_E_BADARG; }
    if (!s || !l || !pt_scn || !inv_x || !inv_s || !inv_y || !inv_theta || !n_inv || !x || !y) {
        mp_set_error("null array");
        return MP_E_BADARG;
    }
    F2cDev a{n_pts, m_max, s, l, pt_scn, 1, nullptr, inv_x, inv_s, inv_y, inv_theta, nullptr, n_inv, x, y,
             nullptr, nullptr, nullptr, nullptr};
    hipLaunchKernelGGL(f2c_kernel, dim3((n_pts + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
    return c2f_launch_check("mp_f2c_batch");
}

// Spline kernels: dynamic LDS on top of the few static words of curve_eval_scene must stay
// within the 64 KB a workgroup gets here.
#define CV_STATIC_LDS 256
static int cv_lds_check(const char *what, const char *dim, int value, size_t lds)
{
    if (lds + CV_STATIC_LDS <= 64 * 1024) return MP_OK;
    mp_set_error("%s: %s = %d needs %zu bytes of LDS per workgroup, more than the kernel can get", what, dim, value,
                 lds + CV_STATIC_LDS);
    return MP_E_UNSUPPORTED;
}

// Workgroup size of the spline stages: as many threads as there are points to evaluate, up to 256.
static int cv_threads(int n_points)
{
    if (mp_knob_cv_threads() > 0) return mp_knob_cv_threads(); // development aid
    return n_points <= 128 ? 128 : 256;
}

// The two fused stages of the whole replan (plan_glue.h).
int mp_plan_path_stage(int32_t B, const PlanPathDev &a, void *stream)
{
    size_t lds = cv_lds_bytes(a.n_s_max + 1, a.p_max);
    lds = (lds + 15) & ~(size_t)15;
    if (int rc = cv_lds_check("mp_plan_scenes (path spline)", "p_max", a.p_max, lds)) return rc;
    // (as in the tail stage: one wave per scene in a large batch)
    const int threads = (B >= 1024 && mp_knob_cv_threads() <= 0) ? MP_WAVE : cv_threads(a.p_max);
    hipLaunchKernelGGL(plan_path_kernel, dim3(B), dim3(threads), lds, (hipStream_t)stream, a);
    return c2f_launch_check("mp_plan_scenes (path spline)");
}

int mp_plan_tail_stage(int32_t B, const PlanTailDev &a, void *stream)
{
    size_t lds = cv_lds_bytes(a.n_t > a.n_pk_max ? a.n_t : a.n_pk_max, a.q_max); // both splines of the stage
    const bool table_larger = lds < (size_t)a.m_max * sizeof(double);
    if (table_larger) lds = (size_t)a.m_max * sizeof(double); // the arc-length table
    lds = (lds + 15) & ~(size_t)15;
    if (int rc = cv_lds_check("mp_plan_scenes (speed profile and path points)", table_larger ? "m_max" : "q_max",
                              table_larger ? a.m_max : a.q_max, lds))
        return rc;
    // (a few dozen points per scene and single-lane sections between them: in a large batch one
    // wave per scene, whose barriers cost nothing, does better than two)
    const int threads = (B >= 1024 && mp_knob_cv_threads() <= 0) ? MP_WAVE : cv_threads(a.q_max);
    hipLaunchKernelGGL(plan_tail_kernel, dim3(B), dim3(threads), lds, (hipStream_t)stream, a);
    return c2f_launch_check("mp_plan_scenes (speed profile and path points)");
}

extern "C" int mp_curve_eval(int32_t B, const double *knots_t, const double *knots_x, const int32_t *n_knots,
                             int32_t n_knots_max, const double *vec, double dt, int32_t mode,
                             const double *query, const int32_t *n_query, int32_t q_max, double *out_t,
                             double *out_x, double *out_dx, int32_t *n_out, int32_t *status, void *stream)
{
    if (B <= 0 || n_knots_max < 2 || q_max <= 0 || !(dt > 0) || mode < 0 || mode > 2) {
        mp_set_error("bad sizes/mode");
        return MP_E_BADARG;
    }
    if (!knots_t || !knots_x || !n_knots || !vec || !out_t || !out_x || !n_out || !status ||
        (mode == 2 && (!query || !n_query))) {
        mp_set_error("null array");
        return MP_E_BADARG;
    }
    CvDev a{n_knots_max, q_max, mode, knots_t, knots_x, n_knots, vec, dt, query, n_query, out_t, out_x, out_dx,
            n_out, status};
    size_t lds = cv_lds_bytes(n_knots_max, q_max);
    lds = (lds + 15) & ~(size_t)15;
    if (int rc = cv_lds_check("mp_curve_eval", "q_max", q_max, lds)) return rc;
    hipLaunchKernelGGL(curve_eval_kernel, dim3(B), dim3(q_max <= 128 ? 128 : 256), lds, (hipStream_t)stream, a);
    return c2f_launch_check("mp_curve_eval");
}

extern "C" int mp_c2f_batch(int32_t n_pts, const double *px, const double *py, const double *vx,
                            const double *vy, const int32_t *pt_scn, int32_t m_max, const double *tab_x,
                            const double *tab_y, const double *tab_s, const double *tab_theta,
                            const double *tab_kappa, const int32_t *M, double *s, double *l, double *s_dt,
                            void *stream)
{
    if (n_pts <= 0 || m_max <= 0) { mp_set_error("bad sizes"); return MP_E_BADARG; }
    if (!px || !py || !pt_scn || !tab_x || !tab_y || !tab_s || !tab_theta || !tab_kappa || !M || !s || !l) {
        mp_set_error("null array");
        return MP_E_BADARG;
    }
    if (s_dt && (!vx || !vy)) { mp_set_error("s_dt requested without velocities"); return MP_E_BADARG; }
    C2fDev a{n_pts, m_max, px, py, vx, vy, pt_scn, tab_x, tab_y, tab_s, tab_theta, tab_kappa, M, s, l, s_dt};
    hipLaunchKernelGGL(c2f_kernel, dim3((n_pts + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
    return c2f_launch_check("mp_c2f_batch");
}
