// Cartesian->Frenet of one point against a reference line's sample tables
// (Model/cartesian_frenet_conversion.py:32-53 on Model/curve.py:117-126),
// shared by every kernel that projects points so that all of them produce the
// same bits.  Include from files compiled without floating-point contraction.
#pragma once
#include "mp_device.h"

// Curve.projection + cal_dist_arr: the first minimum of the distances to the M
// samples (tx, ty); then the sign of the lateral offset from the heading at that
// sample and the arc length from the table.  Returns the sample index.
__device__ __forceinline__ int c2f_point(const double *tx, const double *ty, const double *ts,
                                         const double *tth, int M, double x, double y, double *s_out,
                                         double *l_out)
{
    double best = MP_INF;
    int k = 0;
    for (int i = 0; i < M; ++i) {
        const double dx = tx[i] - x, dy = ty[i] - y;
        const double d = __dsqrt_rn(dx * dx + dy * dy);
        if (d < best) { best = d; k = i; }
    }
    const double th = tth[k];
    const double dx = x - tx[k], dy = y - ty[k];
    const double l_sign = (dy * cos(th) - dx * sin(th) > 0) ? 1.0 : -1.0;
    *s_out = ts[k];
    *l_out = l_sign * best;
    return k;
}

// Longitudinal speed along the line (cartesian_frenet_conversion.py:44-52).
__device__ __forceinline__ double c2f_s_dt(double th, double kr, double l, double vx, double vy)
{
    const double vtheta = atan2(vy, vx);
    const double speed = __dsqrt_rn(vx * vx + vy * vy);
    return speed * cos(vtheta - th) / (1 - kr * l);
}
