// Device-side helpers shared by the lattice kernels (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "../../include/mplattice.h"

#define MP_WAVE 64

// Testing knobs (host side; mp_api.hip): read once from the environment, changed only by
// mp_set_knob; -1 = not set.  Workgroup-size knobs are whole waves within 64..1024.
int mp_knob_sl_narrow();
int mp_knob_sl_pair_cap();
int mp_knob_st_regions_once();
int mp_knob_st_threads();
int mp_knob_rl_threads();
int mp_knob_fe_threads();
int mp_knob_cv_threads();
int mp_knob_st_generic();
int mp_knob_order(); // 0: the *_ordered entry points launch in the caller's order
int mp_knob_sl_rows(); // predecessor rows of the fused SL kernel's compact form (tests: small values force its second launch)
#define MP_INF (__builtin_huge_val())

// ---- wave-level reductions (64 lanes) ---------------------------------------

// One DPP move of a double (both halves with the same control).  Lanes outside
// row_mask keep their own value.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(hi, lo);
}

// DPP moves whose control gives every lane a source lane (quad permutations and
// mirrors): nothing of the destination survives, so no copy of it is needed.
template <int CTRL>
__device__ __forceinline__ int dpp_all_i32(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}

template <int CTRL>
__device__ __forceinline__ double dpp_all_f64(double v)
{
    return __hiloint2double(dpp_all_i32<CTRL>(__double2hiint(v)), dpp_all_i32<CTRL>(__double2loint(v)));
}

#define MP_DPP_QUAD_1032 0xB1   // quad_perm [1,0,3,2]: lane ^ 1
#define MP_DPP_QUAD_2301 0x4E   // quad_perm [2,3,0,1]: lane ^ 2
#define MP_DPP_QUAD_3210 0x1B   // quad_perm [3,2,1,0]: lane ^ 3
#define MP_DPP_HALF_MIRROR 0x141 // lane k <-> 7-k inside each group of 8
#define MP_DPP_MIRROR 0x140      // lane k <-> 15-k inside each row of 16

// Minimum over the wave, returned in every lane.  Data-parallel-primitive
// moves inside the 16-lane rows (quad swaps, half mirror, mirror), row
// broadcasts across rows, then lane 63 is read back; no LDS traffic.
// Costs are never NaN, so v_min_f64 stands in for compare-and-select.  Every lane of
// the wave must be active (a DPP move from an inactive lane reads 0).
// A value every lane of the wave holds alike, moved to scalar registers: values read from LDS
// or produced by a wave reduction sit in vector registers even when they are the same in all
// lanes, and a kernel short of vector registers spills to memory what it could keep there.
__device__ __forceinline__ double uni_f64(double v)
{
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int uni_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ double wave_min_f64(double v)
{
    v = __builtin_fmin(dpp_all_f64<MP_DPP_QUAD_1032>(v), v);
    v = __builtin_fmin(dpp_all_f64<MP_DPP_QUAD_2301>(v), v);
    v = __builtin_fmin(dpp_all_f64<MP_DPP_HALF_MIRROR>(v), v);
    v = __builtin_fmin(dpp_all_f64<MP_DPP_MIRROR>(v), v);
    v = __builtin_fmin(dpp_f64<0x142, 0xA>(v), v); // row_bcast15 into rows 1 and 3
    v = __builtin_fmin(dpp_f64<0x143, 0xC>(v), v); // row_bcast31 into rows 2 and 3
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

// First-minimum argmin: lower cost wins, equal costs keep the lower index.
__device__ __forceinline__ void argmin_merge(double &c, int &i, double oc, int oi)
{
    bool take = (oc < c) | ((oc == c) & (oi < i));
    c = take ? oc : c;
    i = take ? oi : i;
}

// (cost, index) first-minimum over aligned groups of WIDTH lanes, result in every lane
// of the group.  Inside a 16-lane row the partners come from DPP moves (after the two
// quad steps all lanes of a quad agree, so a mirror pairs quads as well as lane ^ 4 or
// lane ^ 8 would); wider groups finish with shuffles.  Every lane of the wave must be active.
template <int WIDTH>
__device__ __forceinline__ void group_argmin(double &c, int &i)
{
    static_assert(WIDTH >= 2 && WIDTH <= MP_WAVE && (WIDTH & (WIDTH - 1)) == 0, "group width");
    argmin_merge(c, i, dpp_all_f64<MP_DPP_QUAD_1032>(c), dpp_all_i32<MP_DPP_QUAD_1032>(i));
    if (WIDTH >= 4) argmin_merge(c, i, dpp_all_f64<MP_DPP_QUAD_2301>(c), dpp_all_i32<MP_DPP_QUAD_2301>(i));
    if (WIDTH >= 8) argmin_merge(c, i, dpp_all_f64<MP_DPP_HALF_MIRROR>(c), dpp_all_i32<MP_DPP_HALF_MIRROR>(i));
    if (WIDTH >= 16) argmin_merge(c, i, dpp_all_f64<MP_DPP_MIRROR>(c), dpp_all_i32<MP_DPP_MIRROR>(i));
#pragma unroll
    for (int m = 16; m < WIDTH; m <<= 1) {
        double oc = __shfl_xor(c, m, MP_WAVE);
        int oi = __shfl_xor(i, m, MP_WAVE);
        argmin_merge(c, i, oc, oi);
    }
}

// First-minimum (cost, index) of 8 rows at once: on entry every lane holds its own
// candidate of each of the 8 rows; on return lane k holds, in (rc, ri), the wave-wide
// result of row k & 7.  The first three steps halve the rows a lane is responsible for
// (a lane keeps the rows whose index bit equals its own lane bit and hands the others to
// its partner), so the 8 rows cost about a quarter of 8 separate wave reductions.
// Every lane of the wave must be active.
__device__ __forceinline__ void wave_argmin_rows8(const double (&c)[8], const int (&i)[8], double &rc, int &ri)
{
    const int lane = threadIdx.x & (MP_WAVE - 1);
    double c4[4], c2[2];
    int i4[4], i2[2];
    {
        const bool up = lane & 1; // partner: lane ^ 1
#pragma unroll
        for (int q = 0; q < 4; ++q) { // rows 2q (kept by even lanes) and 2q+1 (kept by odd lanes)
            const double send_c = up ? c[2 * q] : c[2 * q + 1];
            const int send_i = up ? i[2 * q] : i[2 * q + 1];
            c4[q] = up ? c[2 * q + 1] : c[2 * q];
            i4[q] = up ? i[2 * q + 1] : i[2 * q];
            argmin_merge(c4[q], i4[q], dpp_all_f64<MP_DPP_QUAD_1032>(send_c), dpp_all_i32<MP_DPP_QUAD_1032>(send_i));
        }
    }
    {
        const bool up = lane & 2; // partner: lane ^ 2; c4[q] is row 2q + (lane & 1)
#pragma unroll
        for (int q = 0; q < 2; ++q) { // c4[2q] (row bit 1 clear) and c4[2q+1] (row bit 1 set)
            const double send_c = up ? c4[2 * q] : c4[2 * q + 1];
            const int send_i = up ? i4[2 * q] : i4[2 * q + 1];
            c2[q] = up ? c4[2 * q + 1] : c4[2 * q];
            i2[q] = up ? i4[2 * q + 1] : i4[2 * q];
            argmin_merge(c2[q], i2[q], dpp_all_f64<MP_DPP_QUAD_2301>(send_c), dpp_all_i32<MP_DPP_QUAD_2301>(send_i));
        }
    }
    {
        const bool up = lane & 4; // partner: lane ^ 4 = half mirror, then the quad reversed
        const double send_c = up ? c2[0] : c2[1];
        const int send_i = up ? i2[0] : i2[1];
        rc = up ? c2[1] : c2[0];
        ri = up ? i2[1] : i2[0];
        argmin_merge(rc, ri, dpp_all_f64<MP_DPP_QUAD_3210>(dpp_all_f64<MP_DPP_HALF_MIRROR>(send_c)),
                     dpp_all_i32<MP_DPP_QUAD_3210>(dpp_all_i32<MP_DPP_HALF_MIRROR>(send_i)));
    }
    // lane k now holds row k & 7 over its group of 8 lanes; the groups are merged as they are
    // lane ^ 8 = row mirror, then half mirror
    argmin_merge(rc, ri, dpp_all_f64<MP_DPP_HALF_MIRROR>(dpp_all_f64<MP_DPP_MIRROR>(rc)),
                 dpp_all_i32<MP_DPP_HALF_MIRROR>(dpp_all_i32<MP_DPP_MIRROR>(ri)));
#pragma unroll
    for (int m = 16; m < MP_WAVE; m <<= 1) {
        const double oc = __shfl_xor(rc, m, MP_WAVE);
        const int oi = __shfl_xor(ri, m, MP_WAVE);
        argmin_merge(rc, ri, oc, oi);
    }
}

// ---- NumPy-order arithmetic ------------------------------------------------

// np.sum of K copies of v, in NumPy's pairwise-summation order (K <= 128):
// the guide cost of a straight edge is d_s * sum((l - mid)**2) over K equal
// samples, and reproducing the order keeps exact ties between lanes exact.
__device__ __forceinline__ double np_sum_const(double v, int K)
{
    if (K < 8) {
        double r = 0.0;
        for (int i = 0; i < K; ++i) r = __dadd_rn(r, v);
        return r;
    }
    int blocks = K / 8; // each of the 8 accumulators receives `blocks` copies
    double r = v;
    for (int i = 1; i < blocks; ++i) r = __dadd_rn(r, v);
    double p = __dadd_rn(r, r);
    double q = __dadd_rn(p, p);
    double res = __dadd_rn(q, q);
    for (int i = blocks * 8; i < K; ++i) res = __dadd_rn(res, v);
    return res;
}

// Sample count and positions of get_arange(t_st, t_en, dt) (Utils/tool.py:27-33):
// np.arange gives n = ceil((t_en-t_st)/dt) values t_st + i*((t_st+dt)-t_st) and
// t_en is appended when the last one differs.  Returns K (0 = empty); *n_out
// receives n and *delta_out the fill step.  K > cap is reported as -1.
__device__ __forceinline__ int arange_count(double t_st, double t_en, double dt, int cap,
                                            int *n_out, double *delta_out)
{
    *n_out = 0;
    *delta_out = 0.0;
    if (__dadd_rn(t_st, dt) > t_en) return 0;
    double q = __ddiv_rn(__dsub_rn(t_en, t_st), dt);
    if (!(q <= (double)cap)) return -1;
    int n = (int)ceil(q);
    if (n < 1) return -1;
    double delta = __dsub_rn(__dadd_rn(t_st, dt), t_st);
    double last = (n == 1) ? t_st : __dadd_rn(t_st, __dmul_rn((double)(n - 1), delta));
    if (n == 2) last = __dadd_rn(t_st, dt);
    int K = n + (last != t_en ? 1 : 0);
    *n_out = n;
    *delta_out = delta;
    return K > cap ? -1 : K;
}

__device__ __forceinline__ double arange_value(int k, int n, double t_st, double t_en, double dt,
                                               double delta)
{
    if (k >= n) return t_en;
    if (k == 0) return t_st;
    if (k == 1) return __dadd_rn(t_st, dt);
    return __dadd_rn(t_st, __dmul_rn((double)k, delta));
}

// ---- order of the workgroups ----------------------------------------------------------

// A scenario's guessed weight as one of 64 bins, quarter octaves (the kernels that plan
// scenarios of very different length launch their workgroups by descending bin).
#define MP_WEIGHT_BINS 64
// A batch that fits on the chip at once (6-8 workgroups on each of 256 CUs) gains nothing
// from an order: every workgroup starts at the launch.
#define MP_ORDER_MIN_BATCH 1536
__device__ __forceinline__ int mp_weight_bin(double w)
{
    if (!(w >= 1.0)) return 0;
    const int bin = (int)(4.0f * log2f((float)(w < 1e9 ? w : 1e9))) + 1;
    return bin < MP_WEIGHT_BINS - 1 ? bin : MP_WEIGHT_BINS - 1;
}
void mp_order_launch(int B, const int32_t *key, int32_t *order, hipStream_t st);

// ---- quintic pieces (Model/curve.py) -----------------------------------------

// Model/curve.py:130-157 QuinticPoly coefficients.
__device__ __forceinline__ void quintic_coef(double xs, double vxs, double axs, double xe, double vxe, double axe, double T,
                             double *a)
{
    a[0] = xs;
    a[1] = vxs;
    a[2] = axs / 2.0;
    a[3] = 1.0 * (20 * xe - 20 * xs - (8 * vxe + 12 * vxs) * T - (3 * axs - axe) * T * T) / (2 * T * T * T);
    a[4] = 1.0 * (30 * xs - 30 * xe + (14 * vxe + 16 * vxs) * T + (3 * axs - 2 * axe) * T * T) /
           (2 * T * T * T * T);
    a[5] = 1.0 * (12 * xe - 12 * xs - (6 * vxe + 6 * vxs) * T - (axs - axe) * T * T) /
           (2 * T * T * T * T * T);
}

// Model/curve.py:159-173 calc_point (power form; powers by repeated multiply).
__device__ __forceinline__ double quintic_eval(const double *a, double t, int order)
{
    const double t2 = t * t, t3 = t2 * t;
    if (order == 0) return a[0] + a[1] * t + a[2] * t2 + a[3] * t3 + a[4] * (t2 * t2) + a[5] * (t2 * t3);
    if (order == 1) return a[1] + 2 * a[2] * t + 3 * a[3] * t2 + 4 * a[4] * t3 + 5 * a[5] * (t2 * t2);
    if (order == 2) return 2 * a[2] + 6 * a[3] * t + 12 * a[4] * t2 + 20 * a[5] * t3;
    return 6 * a[3] + 24 * a[4] * t + 60 * a[5] * t2;
}

// np.sum (pairwise order, n <= 128) over generated elements f(i).
template <class F>
__device__ __forceinline__ double np_sum_gen(int n, F f)
{
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; ++i) r += f(i);
        return r;
    }
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = f(j);
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] += f(i + j);
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += f(i);
    return res;
}

