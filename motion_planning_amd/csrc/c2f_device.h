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

// Blocks of 2^bsh consecutive samples, at most 64 of them, for c2f_nearest_boxes.
__host__ __device__ __forceinline__ int c2f_block_shift(int M)
{
    int bsh = 3;
    while (((M + (1 << bsh) - 1) >> bsh) > 64) ++bsh;
    return bsh;
}

// The same nearest sample as c2f_nearest for callers that convert many points against one
// line and hold, per block of 2^bsh samples, the smallest and the largest ordinate
// (ylo[b], yhi[b]; the abscissae ascend, so a block's first and last sample bound them).
// A point d metres from the line makes c2f_nearest look at every sample within d of its
// abscissa, 4 d / d_x of them; here
//   1. one sample per block inside that range gives an upper bound mb of the smallest
//      squared distance,
//   2. a block is kept only if its bounding box comes within mb: the squared distance to
//      the box, built with the same roundings as a sample's, never exceeds that of a sample
//      inside (differences, squares and the sum are monotone),
//   3. the samples of the kept blocks are compared, the block of the bound first, and a
//      block is dropped if its box has fallen outside the minimum found meanwhile.
// Step 3 also keeps the runner-up n2.  The reference takes the first minimum of the rounded
// roots; if the runner-up's root differs from the minimum's, every other sample's is larger
// still and the minimum's index is the answer whatever the order of the comparisons.  If it
// does not (two samples at the same rounded distance) the point is given to c2f_nearest.
__device__ __forceinline__ int c2f_nearest_boxes(const double *tx, const double *ty, const double *ylo,
                                                 const double *yhi, int bsh, int M, double x, double y,
                                                 double *dist_out)
{
    const int bs = 1 << bsh, nb = (M + bs - 1) >> bsh;
    int i0 = 0;
    if (M > 1) {
        const double q = (x - tx[0]) / (tx[1] - tx[0]);
        i0 = q > 0.0 ? (q < (double)(M - 1) ? (int)q : M - 1) : 0;
    }
    const int b0 = i0 >> bsh;
    auto dist2 = [&](int i) {
        const double dx = tx[i] - x, dy = ty[i] - y;
        return dx * dx + dy * dy;
    };
    // squared distance to the box of block b; *gx2 = its part along x
    auto box2 = [&](int b, double *gx2) {
        const int first = b << bsh, last = min(first + bs, M) - 1;
        const double xa = tx[first] - x, xb = tx[last] - x;     // xa <= xb
        const double ya = ylo[b] - y, yb = yhi[b] - y;           // ya <= yb
        const double gx = xa > 0.0 ? xa : (xb < 0.0 ? xb : 0.0);
        const double gy = ya > 0.0 ? ya : (yb < 0.0 ? yb : 0.0);
        *gx2 = gx * gx;
        return *gx2 + gy * gy;
    };
    // 1. (a block wholly on one side of x, further away along x than the bound, ends the
    //    walk on that side: the blocks behind it are further still)
    double mb = dist2(i0);
    int bb = b0;
    int b_lo = b0, b_hi = b0; // the blocks looked at
    for (int b = b0 - 1; b >= 0; --b) {
        const double dxe = tx[(b << bsh) + bs - 1] - x;
        if (dxe < 0.0 && dxe * dxe > mb * (1.0 + 1e-12)) break;
        b_lo = b;
        const double d2 = dist2((b << bsh) + (bs >> 1));
        if (d2 < mb) { mb = d2; bb = b; }
    }
    for (int b = b0 + 1; b < nb; ++b) {
        const double dxe = tx[b << bsh] - x;
        if (dxe > 0.0 && dxe * dxe > mb * (1.0 + 1e-12)) break;
        b_hi = b;
        const double d2 = dist2(min((b << bsh) + (bs >> 1), M - 1));
        if (d2 < mb) { mb = d2; bb = b; }
    }
    // 2.
    const double thr = mb * (1.0 + 1e-12);
    unsigned long long keep = 0ull;
    for (int b = b_lo; b <= b_hi; ++b) {
        double gx2;
        if (b != bb && box2(b, &gx2) <= thr) keep |= 1ull << b;
    }
    // 3.
    double m2 = MP_INF, n2 = MP_INF;
    int k = 0;
    auto block = [&](int b) {
        const int first = b << bsh, end = min(first + bs, M);
        for (int i = first; i < end; ++i) {
            const double d2 = dist2(i);
            const double hi = d2 > m2 ? d2 : m2;
            n2 = hi < n2 ? hi : n2;
            if (d2 < m2) { m2 = d2; k = i; }
        }
    };
    block(bb);
    while (keep) {
        const int b = __ffsll((long long)keep) - 1;
        keep &= keep - 1ull;
        double gx2;
        if (box2(b, &gx2) <= m2 * (1.0 + 1e-12)) block(b);
    }
    const double best = __dsqrt_rn(m2);
    const bool tie = n2 <= m2 * (1.0 + 1e-12) && __dsqrt_rn(n2) == best;
    *dist_out = best;
    if (tie) k = c2f_nearest(tx, ty, M, x, y, dist_out);
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
// same bits), for kernels that convert many points against one line.  With block bounds
// (ylo != nullptr, see c2f_nearest_boxes) the nearest sample is found through them.
__device__ __forceinline__ int c2f_point_tab(const double *tx, const double *ty, const double *ts,
                                             const double *cos_t, const double *sin_t, int M, double x,
                                             double y, double *s_out, double *l_out,
                                             const double *ylo = nullptr, const double *yhi = nullptr, int bsh = 0)
{
    double best;
    const int k = ylo ? c2f_nearest_boxes(tx, ty, ylo, yhi, bsh, M, x, y, &best) : c2f_nearest(tx, ty, M, x, y, &best);
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
