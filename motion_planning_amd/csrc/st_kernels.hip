// ST-lattice speed DP and ST blocked-interval kernels for gfx950.
//
// Reference: Planning/DP_Speed/speed_planner.py:57-162 (DP) and
// Planning/DP_Speed/st_map.py:24-60 (blocked intervals).
//
// The DP is layer-sequential and history dependent: acceleration and jerk of
// an edge read the predecessor indices already fixed for earlier layers
// (speed_planner.py:124-136).  Every term of the edge cost is evaluated with
// the reference's own operation order and without fused multiply-adds
// (__dadd_rn/__dmul_rn/__ddiv_rn), so costs and therefore every first-minimum
// choice are bit-identical to the Python planner.  Terms that depend only on
// an integer difference (speed on j-p, acceleration on j-2p+pp, jerk on
// j-3p+3pp-ppp) are tabulated once per scenario in LDS, which removes all
// divisions from the edge loop without changing a single bit.
#include "mp_device.h"

struct StDev {
    int32_t n_t, n_s_max, r_max;
    const int32_t *n_s;
    const double *s_down, *s_up;
    const int32_t *n_reg;
    const double *cur_vel;
    double speed_lim, ds, dt, dt2, dt3; // dt2 = dt**2, dt3 = dt**3 as Python evaluates them
    double *cost;
    int32_t *pre;
    int32_t *path_s_ind, *path_len;
};

// Obstacle cost of cell s against the blocked intervals of one time layer
// (speed_planner.py:138-150), accumulated in interval order.
__device__ __forceinline__ double st_ob_cost(double s, double front, const double *down, const double *up,
                                             int R)
{
    double ob = 0.0;
    for (int k = 0; k < R; ++k) {
        const double dn = down[k], u = up[k];
        double term = 0.0;
        if (s >= dn && s <= u) {
            term = 1e8;
        } else if (s < dn && s >= __dsub_rn(dn, front)) {
            double d = __dsub_rn(__dadd_rn(front, s), dn);
            term = __dmul_rn(1e3, __dmul_rn(d, d));
        } else if (s > u && s <= __dadd_rn(u, 25.0)) {
            double d = __dsub_rn(__dadd_rn(25.0, s), u);
            term = __dmul_rn(1e3, __dmul_rn(d, d));
        }
        ob = __dadd_rn(ob, term);
    }
    return ob;
}

// One workgroup per scenario.  Dynamic LDS (n = n_s of the scenario, N = n_s_max):
//   cost_prev[N] cost_cur[N] front[N] speed[N] acc[2N] jerk2[2N] jerk3[4N]
//   down[r_max] up[r_max] last_col[n_t] | pre16[n_t][N] i16
__global__ void st_dp_kernel(StDev a)
{
    extern __shared__ double smem[];
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int N = a.n_s_max, n_t = a.n_t, R_max = a.r_max;
    const int n_s = a.n_s[b];
    double *cost_prev = smem;
    double *cost_cur = cost_prev + N;
    double *front = cost_cur + N;
    double *speed = front + N;
    double *acc = speed + N;       // index n + (n_s-1), n in [-(n_s-1), n_s-1]
    double *jerk2 = acc + 2 * N;   // same index, layer i == 2
    double *jerk3 = jerk2 + 2 * N; // index m + 2(n_s-1), layers i >= 3
    double *down = jerk3 + 4 * N;
    double *up = down + R_max;
    double *last_col = up + R_max;
    short *pre16 = (short *)(last_col + n_t);

    const double ds = a.ds, dt = a.dt, lim = a.speed_lim;
    const double v0 = a.cur_vel[b];
    const double dt2 = a.dt2, dt3 = a.dt3;
    double *cost_g = a.cost ? a.cost + (size_t)b * n_t * N : nullptr;
    int32_t *pre_g = a.pre ? a.pre + (size_t)b * n_t * N : nullptr;

    // tables (speed_planner.py:120-136, 151-160)
    for (int d = tid; d < n_s; d += nt) {
        double v = __ddiv_rn(__dmul_rn(__dmul_rn(1.0, (double)d), ds), dt);
        front[d] = __dmul_rn(4.5, v);
        double sc;
        if (v > lim) sc = __dmul_rn(__dmul_rn(100.0, __dmul_rn(v, v)), dt);
        else {
            double dv = __ddiv_rn(__dsub_rn(v, lim), lim);
            sc = __dmul_rn(__dmul_rn(-100.0, dv), dt);
        }
        speed[d] = sc;
    }
    for (int q = tid; q < 2 * n_s - 1; q += nt) {
        int n = q - (n_s - 1);
        double acv = __ddiv_rn(__dmul_rn(__dmul_rn(1.0, (double)n), ds), dt2);
        acc[q] = __dmul_rn(__dmul_rn(acv, acv), dt);
        double jk = __ddiv_rn(acv, dt);
        jerk2[q] = __dmul_rn(__dmul_rn(jk, jk), dt);
    }
    for (int q = tid; q < 4 * n_s - 3; q += nt) {
        int m = q - 2 * (n_s - 1);
        double jk = __ddiv_rn(__dmul_rn(__dmul_rn(1.0, (double)m), ds), dt3);
        jerk3[q] = __dmul_rn(__dmul_rn(jk, jk), dt);
    }
    // row 0: cost 0, pre_ind -1 (st_map.py:86-91)
    for (int j = tid; j < N; j += nt) {
        cost_prev[j] = 0.0;
        pre16[j] = -1;
        if (cost_g) cost_g[j] = 0.0;
        if (pre_g) pre_g[j] = -1;
    }
    if (tid == 0) last_col[0] = 0.0;
    __syncthreads();

    for (int i = 1; i < n_t; ++i) {
        const int R = a.n_reg[(size_t)b * n_t + i];
        for (int k = tid; k < R; k += nt) {
            down[k] = a.s_down[((size_t)b * n_t + i) * R_max + k];
            up[k] = a.s_up[((size_t)b * n_t + i) * R_max + k];
        }
        __syncthreads();
        const short *pre1 = pre16 + (size_t)(i - 1) * N;
        const short *pre2 = pre16 + (size_t)(i >= 2 ? i - 2 : 0) * N;
        // Cells are handed out in pairs (t, n_s-1-t) and the two predecessor
        // loops (j+1 iterations for cell j) run back to back in ONE loop of
        // n_s+1 iterations, so every lane of a wave does the same amount of work.
        for (int t = tid; 2 * t < n_s; t += nt) {
            const int j1 = t, j2 = n_s - 1 - t;
            const bool two = j2 != j1;
            const double s1 = __dmul_rn((double)j1, ds), s2 = __dmul_rn((double)j2, ds);
            double bestA = MP_INF, bestB = MP_INF;
            int bpA = -1, bpB = -1;
            if (i == 1) {
                // only p_j = 0 (speed_planner.py:96-100)
                for (int half = 0; half < (two ? 2 : 1); ++half) {
                    const int j = half ? j2 : j1;
                    double v = __ddiv_rn(__dmul_rn(__dmul_rn(1.0, (double)j), ds), dt);
                    double ac = __ddiv_rn(__dsub_rn(v, v0), dt);
                    double jk = __ddiv_rn(ac, dt);
                    double ob = st_ob_cost(half ? s2 : s1, front[j], down, up, R);
                    double tot = __dadd_rn(__dadd_rn(__dadd_rn(ob, speed[j]), __dmul_rn(__dmul_rn(ac, ac), dt)),
                                           __dmul_rn(__dmul_rn(jk, jk), dt));
                    if (half) { bestB = tot; bpB = 0; } else { bestA = tot; bpA = 0; }
                }
            } else {
                const int total = j1 + 1 + (two ? j2 + 1 : 0);
                for (int q = 0; q < total; ++q) {
                    const bool second = q > j1;
                    const int j = second ? j2 : j1;
                    const int p = second ? q - j1 - 1 : q;
                    const int d = j - p;
                    const int pp = pre1[p];
                    const int n = d - (p - pp);
                    double jc;
                    if (i == 2) jc = jerk2[n + (n_s - 1)];
                    else {
                        const int ppp = pre2[pp];
                        jc = jerk3[(j - 3 * p + 3 * pp - ppp) + 2 * (n_s - 1)];
                    }
                    double ob = st_ob_cost(second ? s2 : s1, front[d], down, up, R);
                    double tot = __dadd_rn(__dadd_rn(__dadd_rn(ob, speed[d]), acc[n + (n_s - 1)]), jc);
                    double c = __dadd_rn(tot, cost_prev[p]);
                    if (second) {
                        if (bpB < 0 || c < bestB) { bestB = c; bpB = p; }
                    } else {
                        if (bpA < 0 || c < bestA) { bestA = c; bpA = p; }
                    }
                }
            }
            for (int half = 0; half < (two ? 2 : 1); ++half) {
                const int j = half ? j2 : j1;
                const double best = half ? bestB : bestA;
                const int bp = half ? bpB : bpA;
                cost_cur[j] = best;
                pre16[(size_t)i * N + j] = (short)bp;
                if (cost_g) cost_g[(size_t)i * N + j] = best;
                if (pre_g) pre_g[(size_t)i * N + j] = bp;
            }
        }
        // padding cells keep the initial values
        for (int j = n_s + tid; j < N; j += nt) {
            if (cost_g) cost_g[(size_t)i * N + j] = 0.0;
            if (pre_g) pre_g[(size_t)i * N + j] = -1;
        }
        __syncthreads();
        for (int j = tid; j < n_s; j += nt) cost_prev[j] = cost_cur[j];
        if (tid == 0) last_col[i] = cost_cur[n_s - 1];
        __syncthreads();
    }
    // end state and backtrack (speed_planner.py:63-89)
    if (tid == 0) {
        double min_i = last_col[1];
        int arg_i = 1;
        for (int i = 2; i < n_t; ++i)
            if (last_col[i] < min_i) { min_i = last_col[i]; arg_i = i; }
        double min_j = cost_prev[0];
        int arg_j = 0;
        for (int j = 1; j < n_s; ++j)
            if (cost_prev[j] < min_j) { min_j = cost_prev[j]; arg_j = j; }
        int end_i, end_j;
        if (min_i <= min_j) { end_i = arg_i; end_j = n_s - 1; }
        else { end_i = n_t - 1; end_j = arg_j; }
        int32_t *out = a.path_s_ind + (size_t)b * n_t;
        for (int i = end_i + 1; i < n_t; ++i) out[i] = -1;
        int j = end_j;
        out[end_i] = end_j;
        for (int i = end_i; i >= 1; --i) {
            j = pre16[(size_t)i * N + j];
            out[i - 1] = j;
        }
        a.path_len[b] = end_i + 1;
    }
}

// ---------------------------------------------------------------------------
// ST blocked intervals
// ---------------------------------------------------------------------------

// Smallest double x with sqrt(x) >= r under round-to-nearest: then
// "sqrt(d2) < r" (the reference's test, Utils/tool.py:58-59 + st_map.py:51-52)
// is exactly "d2 < x", and the scan needs no square roots.
__device__ __forceinline__ double sqrt_threshold(double r)
{
    if (!(r > 0.0)) return 0.0; // sqrt(d2) < r never holds for r <= 0
    double x = __dmul_rn(r, r);
    for (int it = 0; it < 8 && __dsqrt_rn(x) >= r; ++it) x = nextafter(x, 0.0);
    for (int it = 0; it < 16 && __dsqrt_rn(x) < r; ++it) x = nextafter(x, MP_INF);
    return x;
}

struct RegDev {
    int32_t n_t, p_max, n_st_max, n_dy_max, r_max;
    const double *path_s, *path_l;
    const int32_t *n_path;
    const double *st_s, *st_l, *st_r;
    const int32_t *n_st;
    const double *dy_s, *dy_l, *dy_r;
    const int32_t *n_dy;
    double *s_down, *s_up;
    int32_t *n_reg, *status;
};

// One workgroup per scenario; the sampled path sits in LDS.  For every time
// layer the obstacles are visited in the reference's order (statics, then
// dynamics) in chunks of blockDim threads, one obstacle per thread scanning
// the path; hits are appended in obstacle order with a ballot prefix count.
__global__ void st_regions_kernel(RegDev a)
{
    extern __shared__ double smem[];
    __shared__ int wave_cnt[16];
    __shared__ int base_sh;
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & (MP_WAVE - 1), wave = tid / MP_WAVE, nw = nt / MP_WAVE;
    const int P = a.n_path[b], n_st = a.n_st ? a.n_st[b] : 0, n_dy = a.n_dy ? a.n_dy[b] : 0;
    double *ps = smem, *pl = smem + a.p_max;
    for (int k = tid; k < P; k += nt) {
        ps[k] = a.path_s[(size_t)b * a.p_max + k];
        pl[k] = a.path_l[(size_t)b * a.p_max + k];
    }
    __syncthreads();
    int overflow = 0;
    const int n_ob = n_st + n_dy;
    for (int i = 0; i < a.n_t; ++i) {
        if (tid == 0) base_sh = 0;
        __syncthreads();
        for (int o0 = 0; o0 < n_ob; o0 += nt) {
            const int o = o0 + tid;
            bool hit = false;
            double dn = 0.0, u = 0.0;
            if (o < n_ob) {
                double ox, oy, rad;
                if (o < n_st) {
                    ox = a.st_s[(size_t)b * a.n_st_max + o];
                    oy = a.st_l[(size_t)b * a.n_st_max + o];
                    rad = a.st_r[(size_t)b * a.n_st_max + o];
                } else {
                    const size_t q = (size_t)b * a.n_dy_max + (o - n_st);
                    ox = a.dy_s[q * a.n_t + i];
                    oy = a.dy_l[q * a.n_t + i];
                    rad = a.dy_r[q];
                }
                const double thr = sqrt_threshold(rad);
                for (int k = 0; k < P; ++k) {
                    double dx = __dsub_rn(ox, ps[k]), dy = __dsub_rn(oy, pl[k]);
                    double d2 = __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy));
                    if (d2 < thr) {
                        if (!hit) { dn = ps[k]; hit = true; }
                        u = ps[k];
                    }
                }
            }
            const unsigned long long bal = __ballot(hit);
            if (lane == 0) wave_cnt[wave] = __popcll(bal);
            __syncthreads();
            int pos = base_sh;
            for (int w = 0; w < wave; ++w) pos += wave_cnt[w];
            pos += __popcll(bal & ((1ull << lane) - 1ull));
            if (hit) {
                if (pos < a.r_max) {
                    a.s_down[((size_t)b * a.n_t + i) * a.r_max + pos] = dn;
                    a.s_up[((size_t)b * a.n_t + i) * a.r_max + pos] = u;
                } else overflow = 1;
            }
            __syncthreads();
            if (tid == 0) {
                int tot = 0;
                for (int w = 0; w < nw; ++w) tot += wave_cnt[w];
                base_sh += tot;
            }
            __syncthreads();
        }
        if (tid == 0) a.n_reg[(size_t)b * a.n_t + i] = min(base_sh, a.r_max);
        __syncthreads();
    }
    overflow = __syncthreads_or(overflow);
    if (tid == 0) a.status[b] = overflow ? MP_ST_E_CAPACITY : MP_ST_OK;
}

// ---------------------------------------------------------------------------
extern void mp_set_error(const char *fmt, ...);

static int st_launch_check(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        mp_set_error("%s: %s", what, hipGetErrorString(e));
        return MP_E_LAUNCH;
    }
    return MP_OK;
}

extern "C" int mp_st_dp(int32_t B, int32_t n_t, int32_t n_s_max, const int32_t *n_s, const double *s_down,
                        const double *s_up, const int32_t *n_reg, int32_t r_max, const double *cur_vel,
                        double speed_lim, double ds, double dt, double *cost, int32_t *pre,
                        int32_t *path_s_ind, int32_t *path_len, void *stream)
{
    if (B <= 0 || n_t < 2 || n_s_max <= 0 || r_max < 0) { mp_set_error("bad B/n_t/n_s_max/r_max"); return MP_E_BADARG; }
    if (!n_s || !n_reg || !cur_vel || !path_s_ind || !path_len || (r_max > 0 && (!s_down || !s_up))) {
        mp_set_error("null array");
        return MP_E_BADARG;
    }
    if (!(ds > 0) || !(dt > 0) || !(speed_lim > 0)) { mp_set_error("ds, dt, speed_lim must be positive"); return MP_E_BADARG; }
    if (n_t > 64 || n_s_max > 1024 || r_max > 64) {
        mp_set_error("mp_st_dp supports n_t <= 64, n_s_max <= 1024, r_max <= 64");
        return MP_E_UNSUPPORTED;
    }
    StDev a{n_t, n_s_max, r_max > 0 ? r_max : 1, n_s, s_down, s_up, n_reg, cur_vel, speed_lim, ds, dt,
            pow(dt, 2.0), pow(dt, 3.0), cost, pre, path_s_ind, path_len};
    size_t lds = ((size_t)12 * n_s_max + 2 * a.r_max + n_t) * sizeof(double) + (size_t)n_t * n_s_max * sizeof(short);
    lds = (lds + 15) & ~(size_t)15;
    if (lds > 160 * 1024) { mp_set_error("mp_st_dp: grid too large for LDS"); return MP_E_UNSUPPORTED; }
    int half = (n_s_max + 1) / 2;
    int threads = half <= 64 ? 64 : (half <= 128 ? 128 : 256);
    hipLaunchKernelGGL(st_dp_kernel, dim3(B), dim3(threads), lds, (hipStream_t)stream, a);
    return st_launch_check("mp_st_dp");
}

extern "C" int mp_st_regions(int32_t B, int32_t n_t, const double *path_s, const double *path_l,
                             const int32_t *n_path, int32_t p_max, const double *st_s, const double *st_l,
                             const double *st_r, const int32_t *n_st, int32_t n_st_max, const double *dy_s,
                             const double *dy_l, const double *dy_r, const int32_t *n_dy, int32_t n_dy_max,
                             int32_t r_max, double *s_down, double *s_up, int32_t *n_reg, int32_t *status,
                             void *stream)
{
    if (B <= 0 || n_t <= 0 || p_max <= 0 || r_max <= 0 || n_st_max < 0 || n_dy_max < 0) {
        mp_set_error("bad sizes");
        return MP_E_BADARG;
    }
    if (!path_s || !path_l || !n_path || !s_down || !s_up || !n_reg || !status ||
        (n_st_max > 0 && (!st_s || !st_l || !st_r || !n_st)) ||
        (n_dy_max > 0 && (!dy_s || !dy_l || !dy_r || !n_dy))) {
        mp_set_error("null array");
        return MP_E_BADARG;
    }
    size_t lds = (size_t)2 * p_max * sizeof(double);
    if (lds > 128 * 1024) { mp_set_error("mp_st_regions: p_max too large"); return MP_E_UNSUPPORTED; }
    RegDev a{n_t, p_max, n_st_max, n_dy_max, r_max, path_s, path_l, n_path, st_s, st_l, st_r,
             n_st_max > 0 ? n_st : nullptr, dy_s, dy_l, dy_r, n_dy_max > 0 ? n_dy : nullptr,
             s_down, s_up, n_reg, status};
    int n_ob = n_st_max + n_dy_max;
    int threads = n_ob <= 64 ? 64 : (n_ob <= 128 ? 128 : 256);
    hipLaunchKernelGGL(st_regions_kernel, dim3(B), dim3(threads), lds, (hipStream_t)stream, a);
    return st_launch_check("mp_st_regions");
}
