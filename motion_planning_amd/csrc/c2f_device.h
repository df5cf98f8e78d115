// Cartesian->Frenet of one point against a reference line's sample tables
// (Model/cartesian_frenet_conversion.py:32-53 on Model/curve.py:117-126),
// shared by every kernel that projects points so that all of them produce the
// same bits.  Include from files compiled without floating-point contraction.
#pragma once
#include "mp_device.h"

// Curve.projection + cal_dist_arr (Utils/tool.py:62-72): the first minimum of the
// distances sqrt(dx^2 + dy^2) to the M samples (tx, ty), found without a square
// root per sample; then the sign of the lateral offset from the heading at that
// sample and the arc length from the table.  Returns the sample index.
__device__ __forceinline__ int c2f_nearest(const double *tx, const double *ty, int M, double x, double y,
                                           double *dist_out)
{
    // sqrt is monotone, so the smallest distance is the root of the smallest squared
    // distance m2, and the first sample that reaches it is the first one whose squared
    // distance does not exceed `top`, the largest double with that same rounded root.
    // The abscissae ascend, so the search starts at the sample next to x and walks
    // outwards on both sides only while (tx - x)^2 alone stays within the best value
    // so far: a sample beyond that has d2 >= (tx - x)^2 > m2 and can neither win nor
    // tie (the 1e-12 margin is far above the few ulps between m2 and top).
    int i0 = 0;
    if (M > 1) {
        const double q = (x - tx[0]) / (tx[1] - tx[0]);
        i0 = q > 0.0 ? (q < (double)(M - 1) ? (int)q : M - 1) : 0;
    }
    auto dist2 = [&](int i, double *dx2) {
        const double dx = tx[i] - x, dy = ty[i] - y;
        *dx2 = dx * dx;
        return *dx2 + dy * dy;
    };
    double dx2;
    double m2 = dist2(i0, &dx2);
    int k = i0;
    for (int i = i0 - 1; i >= 0; --i) {
        const double d2 = dist2(i, &dx2);
        if (dx2 > m2 * (1.0 + 1e-12)) break;
        if (d2 <= m2) { m2 = d2; k = i; } // equal squared distances: the lower index
    }
    for (int i = i0 + 1; i < M; ++i) {
        const double d2 = dist2(i, &dx2);
        if (dx2 > m2 * (1.0 + 1e-12)) break;
        if (d2 < m2) { m2 = d2; k = i; }
    }
    const double best = __dsqrt_rn(m2);
    double top = m2;
    for (int it = 0; it < 8; ++it) {
        const double nx = nextafter(top, MP_INF);
        if (__dsqrt_rn(nx) == best) top = nx; else break;
    }
    if (top > m2) {
        for (int i = k - 1; i >= 0; --i) {
            const double d2 = dist2(i, &dx2);
            if (dx2 > top) break;
            if (d2 <= top) k = i;
        }
    }
    *dist_out = best;
    return k;
}

// The whole conversion of one point; (cth, sth) = cos and sin of the heading at the
// nearest sample.
__device__ __forceinline__ void c2f_finish(const double *tx, const double *ty, const double *ts, int k,
                                           double best, double cth, double sth, double x, double y,
                                           double *s_out, double *l_out)
{
    const double dx = x - tx[k], dy = y - ty[k];
    const double l_sign = (dy * cth - dx * sth > 0) ? 1.0 : -1.0;
    *s_out = ts[k];
    *l_out = l_sign * best;
}

__device__ __forceinline__ int c2f_point(const double *tx, const double *ty, const double *ts,
                                         const double *tth, int M, double x, double y, double *s_out,
                                         double *l_out)
{
    double best;
    const int k = c2f_nearest(tx, ty, M, x, y, &best);
    const double th = tth[k];
    c2f_finish(tx, ty, ts, k, best, cos(th), sin(th), x, y, s_out, l_out);
    return k;
}

// The same with cos / sin of every sample's heading tabulated by the caller (cos_t[k] =
// cos(tth[k]), sin_t[k] = sin(tth[k]): the same functions on the same arguments, so the
// same bits), for kernels that convert many points against one line.
__device__ __forceinline__ int c2f_point_tab(const double *tx, const double *ty, const double *ts,
                                             const double *cos_t, const double *sin_t, int M, double x,
                                             double y, double *s_out, double *l_out)
{
    double best;
    const int k = c2f_nearest(tx, ty, M, x, y, &best);
    c2f_finish(tx, ty, ts, k, best, cos_t[k], sin_t[k], x, y, s_out, l_out);
    return k;
}

// Longitudinal speed along the line (cartesian_frenet_conversion.py:44-52).
__device__ __forceinline__ double c2f_s_dt(double th, double kr, double l, double vx, double vy)
{
    const double vtheta = atan2(vy, vx);
    const double speed = __dsqrt_rn(vx * vx + vy * vy);
    return speed * cos(vtheta - th) / (1 - kr * l);
}
