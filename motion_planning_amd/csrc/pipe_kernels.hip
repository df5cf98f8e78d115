// Whole replan for B scenes behind one entry point (mp_plan_scenes), gfx950.
//
// Reference: Planning/Interface/planner_interface.py:54-68, 133-159 -- the chain
// SLMap -> PathPlanner -> path_sampling -> STMap -> SpeedPlanner -> path_convert
// that PlannerInterface.run_step walks for one scene.  Here every stage is a
// kernel over all scenes and the stages hand their results to each other in
// device memory (one arena, layout below), so the host enqueues a fixed list of
// launches and never waits for a value: layer counts, obstacle counts, ST grid
// widths and failure codes stay on the device as per-scene arrays.
//
// The heavy stages are the library's own entry points (mp_refline_tables,
// mp_sl_plan, mp_st_regions, mp_st_dp); this file adds the front end,
//   plan_frontend_kernel  SLMap.create_frenet_map + add_obstacle (sl_map.py:33-112)
// and the short stages between them run as one kernel each, beside the spline
// code (plan_glue.h): the path stage (knots of the path spline, its samples, ST
// grid width) and the tail stage (knots of the speed spline, speed profile, path
// points at the profile, those points in the robot frame).
#pragma clang fp contract(off)
#include "mp_device.h"
#include "c2f_device.h"
#include "plan_glue.h"

#include <string.h>

#include <mutex>

extern void mp_set_error(const char *fmt, ...);
int mp_sl_plan_keyed(const mp_sl_batch *batch, const mp_sl_params *params, double *cost_map, int32_t *index_map,
                     int32_t *path_idx, int32_t *status, int32_t *layers_done, double *best_cost,
                     int32_t *order_ws, void *stream);
int mp_st_dp_keyed(int32_t B, int32_t n_t, int32_t n_s_max, const int32_t *n_s, const double *s_down,
                   const double *s_up, const int32_t *n_reg, int32_t r_max, const double *cur_vel,
                   double speed_lim, double ds, double dt, double *cost, int32_t *pre, int32_t *path_s_ind,
                   int32_t *path_len, int32_t *order_ws, void *stream);
int mp_st_regions_strided(int32_t B, int32_t n_t, const double *path_s, const double *path_l,
                          const int32_t *n_path, int32_t p_max, const double *st_s, const double *st_l,
                          const double *st_r, const int32_t *n_st, int32_t n_st_max, const double *dy_s,
                          const double *dy_l, const double *dy_r, const int32_t *n_dy, int32_t n_dy_max,
                          int32_t dy_fstride, int32_t r_max, double *s_down, double *s_up, int32_t *n_reg,
                          int32_t *status, void *stream, const int32_t *key_n_s, int32_t *key_out);

int mp_scene_ingest_into(int32_t B, const double *rot, const double *trans, const double *ego_vel_w,
                         const double *way_xy, const int32_t *n_way, int32_t way_max, const double *ob_pos,
                         const double *ob_vel, const double *ob_rad, const int32_t *n_ob, int32_t ob_max,
                         double ignore_dist, int32_t n_frames, double dt, double *knots_x, double *knots_y,
                         double *ego_vel_r, double *st_pos, double *st_rad, int32_t *n_st, int32_t st_max,
                         double *dy_pos, double *dy_rad, int32_t *n_dy, int32_t dy_max, int32_t *overflow,
                         void *stream);

// ---------------------------------------------------------------------------
// Arena layout: X(name, C type, MP_F64 / MP_I32, dim0, dim1, dim2, dim3)
// ---------------------------------------------------------------------------
#define NOB(d) ((d).n_st_max + (d).n_dy_max * (d).n_frames > 0 ? (d).n_st_max + (d).n_dy_max * (d).n_frames : 1)
#define ONE(n) ((n) > 0 ? (n) : 1)
#define OPT(d, n) ((d).want_maps ? (n) : 0)
#define WLD(d, n) ((d).world ? (n) : 0) // arrays of the world-frame call only
// The first PLAN_N_ZEROED arrays are cleared by one memset at the start of every call
// (result rows whose unused tails no kernel writes, and the zero inputs).
#define PLAN_N_ZEROED 13
#define PLAN_ARRAYS(X)                                                                       \
    X(zeros, double, MP_F64, d.B, 2, 1, 1)                                                   \
    X(s_layers, double, MP_F64, d.B, d.n_s_max, 1, 1)                                        \
    X(knots_t, double, MP_F64, d.B, d.n_s_max + 1, 1, 1)                                     \
    X(knots_v, double, MP_F64, d.B, d.n_s_max + 1, 1, 1)                                     \
    X(path_s, double, MP_F64, d.B, d.p_max, 1, 1)                                            \
    X(path_l, double, MP_F64, d.B, d.p_max, 1, 1)                                            \
    X(sp_t, double, MP_F64, d.B, d.n_t, 1, 1)                                                \
    X(sp_s, double, MP_F64, d.B, d.n_t, 1, 1)                                                \
    X(prof_t, double, MP_F64, d.B, d.q_max, 1, 1)                                            \
    X(ss, double, MP_F64, d.B, d.q_max, 1, 1)                                                \
    X(vv, double, MP_F64, d.B, d.q_max, 1, 1)                                                \
    X(final_s, double, MP_F64, d.B, d.q_max, 1, 1)                                           \
    X(final_l, double, MP_F64, d.B, d.q_max, 1, 1)                                           \
    X(status, int32_t, MP_I32, d.B, 1, 1, 1)                                                 \
    X(tab_x, double, MP_F64, d.B, d.m_max, 1, 1)                                             \
    X(tab_y, double, MP_F64, d.B, d.m_max, 1, 1)                                             \
    X(tab_s, double, MP_F64, d.B, d.m_max, 1, 1)                                             \
    X(tab_theta, double, MP_F64, d.B, d.m_max, 1, 1)                                         \
    X(tab_kappa, double, MP_F64, d.B, d.m_max, 1, 1)                                         \
    X(tab_m, int32_t, MP_I32, d.B, 1, 1, 1)                                                  \
    X(origin, double, MP_F64, d.B, 4, 1, 1)                                                  \
    X(rl_status, int32_t, MP_I32, d.B, 1, 1, 1)                                              \
    X(inv_x, double, MP_F64, d.B, d.m_max, 1, 1)                                             \
    X(inv_s, double, MP_F64, d.B, d.m_max, 1, 1)                                             \
    X(inv_y, double, MP_F64, d.B, d.m_max, 1, 1)                                             \
    X(inv_theta, double, MP_F64, d.B, d.m_max, 1, 1)                                         \
    X(n_inv, int32_t, MP_I32, d.B, 1, 1, 1)                                                  \
    X(ego_sl, double, MP_F64, d.B, 2, 1, 1)                                                  \
    X(robot_vel, double, MP_F64, d.B, 1, 1, 1)                                               \
    X(st_s, double, MP_F64, d.B, ONE(d.n_st_max), 1, 1)                                      \
    X(st_l, double, MP_F64, d.B, ONE(d.n_st_max), 1, 1)                                      \
    X(dy_s, double, MP_F64, d.B, ONE(d.n_dy_max), d.n_frames, 1)                             \
    X(dy_l, double, MP_F64, d.B, ONE(d.n_dy_max), d.n_frames, 1)                             \
    X(ob_s, double, MP_F64, d.B, NOB(d), 1, 1)                                               \
    X(ob_l, double, MP_F64, d.B, NOB(d), 1, 1)                                               \
    X(n_ob, int32_t, MP_I32, d.B, 1, 1, 1)                                                   \
    X(ob_dist, double, MP_F64, d.B, 1, 1, 1)                                                 \
    X(n_layers, int32_t, MP_I32, d.B, 1, 1, 1)                                               \
    X(l_row, double, MP_F64, d.n_l, 1, 1, 1)                                                 \
    X(path_idx, int32_t, MP_I32, d.B, d.n_s_max, 1, 1)                                       \
    X(sl_status, int32_t, MP_I32, d.B, 1, 1, 1)                                              \
    X(layers_done, int32_t, MP_I32, d.B, 1, 1, 1)                                            \
    X(best_cost, double, MP_F64, d.B, 1, 1, 1)                                               \
    X(cost_map, double, MP_F64, OPT(d, d.B), d.n_s_max, d.n_l, 1)                            \
    X(index_map, int32_t, MP_I32, OPT(d, d.B), d.n_s_max, d.n_l, 1)                          \
    X(n_pk, int32_t, MP_I32, d.B, 1, 1, 1)                                                   \
    X(n_path, int32_t, MP_I32, d.B, 1, 1, 1)                                                 \
    X(path_status, int32_t, MP_I32, d.B, 1, 1, 1)                                            \
    X(st_n_s, int32_t, MP_I32, d.B, 1, 1, 1)                                                 \
    X(st_down, double, MP_F64, d.B, d.n_t, d.r_max, 1)                                       \
    X(st_up, double, MP_F64, d.B, d.n_t, d.r_max, 1)                                         \
    X(st_n_reg, int32_t, MP_I32, d.B, d.n_t, 1, 1)                                           \
    X(reg_status, int32_t, MP_I32, d.B, 1, 1, 1)                                             \
    X(st_path_s_ind, int32_t, MP_I32, d.B, d.n_t, 1, 1)                                      \
    X(st_path_len, int32_t, MP_I32, d.B, 1, 1, 1)                                            \
    X(st_cost, double, MP_F64, OPT(d, d.B), d.n_t, d.st_n_s_max, 1)                          \
    X(st_pre, int32_t, MP_I32, OPT(d, d.B), d.n_t, d.st_n_s_max, 1)                          \
    X(n_prof, int32_t, MP_I32, d.B, 1, 1, 1)                                                 \
    X(prof_status, int32_t, MP_I32, d.B, 1, 1, 1)                                            \
    X(n_final, int32_t, MP_I32, d.B, 1, 1, 1)                                                \
    X(final_status, int32_t, MP_I32, d.B, 1, 1, 1)                                           \
    X(final_x, double, MP_F64, d.B, d.q_max, 1, 1)                                           \
    X(final_y, double, MP_F64, d.B, d.q_max, 1, 1)                                           \
    X(order_ws, int32_t, MP_I32, d.B, 2, 1, 1)                                               \
    X(w_knots_x, double, MP_F64, WLD(d, d.B), d.n_knots_max, 1, 1)                           \
    X(w_knots_y, double, MP_F64, WLD(d, d.B), d.n_knots_max, 1, 1)                           \
    X(w_ego_vel, double, MP_F64, WLD(d, d.B), 2, 1, 1)                                       \
    X(w_st_pos, double, MP_F64, WLD(d, d.B), ONE(d.n_st_max), 2, 1)                          \
    X(w_st_rad, double, MP_F64, WLD(d, d.B), ONE(d.n_st_max), 1, 1)                          \
    X(w_n_st, int32_t, MP_I32, WLD(d, d.B), 1, 1, 1)                                         \
    X(w_dy_pos, double, MP_F64, WLD(d, d.B), ONE(d.n_dy_max), d.n_frames, 2)                 \
    X(w_dy_rad, double, MP_F64, WLD(d, d.B), ONE(d.n_dy_max), 1, 1)                          \
    X(w_n_dy, int32_t, MP_I32, WLD(d, d.B), 1, 1, 1)                                         \
    X(w_status, int32_t, MP_I32, WLD(d, d.B), 1, 1, 1)                                       \
    X(world_x, double, MP_F64, WLD(d, d.B), d.q_max, 1, 1)                                   \
    X(world_y, double, MP_F64, WLD(d, d.B), d.q_max, 1, 1)

struct PlanPtrs {
#define X(name, ctype, dt, d0, d1, d2, d3) ctype *name;
    PLAN_ARRAYS(X)
#undef X
};

struct ArrInfo {
    const char *name;
    int32_t dtype;
    int64_t shape[4];
    int64_t offset, bytes;
};

static int plan_array_table(const mp_plan_dims &d, ArrInfo *out, int64_t *total)
{
    int n = 0;
    int64_t off = 0;
#define X(nm, ctype, dt, d0, d1, d2, d3)                                              \
    {                                                                                 \
        ArrInfo &a = out[n++];                                                        \
        a.name = #nm;                                                                 \
        a.dtype = dt;                                                                 \
        a.shape[0] = (d0); a.shape[1] = (d1); a.shape[2] = (d2); a.shape[3] = (d3);   \
        a.bytes = a.shape[0] * a.shape[1] * a.shape[2] * a.shape[3] * (int64_t)sizeof(ctype); \
        a.offset = off;                                                               \
        off += (a.bytes + 255) & ~(int64_t)255;                                       \
    }
    PLAN_ARRAYS(X)
#undef X
    *total = off;
    return n;
}

#define X(name, ctype, dt, d0, d1, d2, d3) +1
static const int PLAN_N_ARRAYS = 0 PLAN_ARRAYS(X);
#undef X

static int check_dims(const mp_plan_dims *d)
{
    if (!d) { mp_set_error("null dims"); return MP_E_BADARG; }
    if (d->B <= 0 || d->n_knots_max < 2 || d->n_st_max < 0 || d->n_dy_max < 0 || d->n_frames <= 0 ||
        d->n_l <= 0 || d->m_max <= 0 || d->n_s_max <= 0 || d->p_max <= 0 || d->n_t < 2 || d->st_n_s_max <= 0 ||
        d->r_max <= 0 || d->q_max <= 0) {
        mp_set_error("mp_plan_dims: non-positive size");
        return MP_E_BADARG;
    }
    if (d->n_t > d->n_frames && d->n_dy_max > 0) {
        mp_set_error("mp_plan_dims: n_t time layers need at least n_t predicted frames per moving obstacle");
        return MP_E_BADARG;
    }
    return MP_OK;
}

extern "C" int32_t mp_plan_array_count(void) { return PLAN_N_ARRAYS; }

extern "C" int mp_plan_array_info(const mp_plan_dims *dims, int32_t i, const char **name, int32_t *dtype,
                                  int64_t shape[4], int64_t *offset)
{
    int rc = check_dims(dims);
    if (rc) return rc;
    if (i < 0 || i >= PLAN_N_ARRAYS || !name || !dtype || !shape || !offset) {
        mp_set_error("mp_plan_array_info: bad index or null output");
        return MP_E_BADARG;
    }
    ArrInfo tab[PLAN_N_ARRAYS];
    int64_t total;
    plan_array_table(*dims, tab, &total);
    *name = tab[i].name;
    *dtype = tab[i].dtype;
    for (int k = 0; k < 4; ++k) shape[k] = tab[i].shape[k];
    *offset = tab[i].offset;
    return MP_OK;
}

extern "C" int64_t mp_plan_arena_bytes(const mp_plan_dims *dims)
{
    if (check_dims(dims)) return -1;
    ArrInfo tab[PLAN_N_ARRAYS];
    int64_t total;
    plan_array_table(*dims, tab, &total);
    return total;
}

extern "C" void mp_plan_default_params(mp_plan_params *p)
{
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->l_width = 3.5;
    p->save_width = 0.6;
    p->layer_gap = 8.0;
    p->max_map_len = 100.0;
    p->d_x = 0.5;
    p->path_dt = 0.5;
    p->sl.d_s = 0.5;
    p->sl.w_d = 0.5;
    p->sl.no_path_cost = 1e4;
    p->sl.early_exit = 1;
    p->sl.mid_l = -1;
    p->speed_lim = 25.0 * 5.0 / 18.0;
    p->st_ds = 1.0;
    p->st_dt = 0.5;
    p->path_planner = 0;
}

// ---------------------------------------------------------------------------
// glue kernels
// ---------------------------------------------------------------------------
struct FrontDev {
    mp_scene_batch in;
    mp_plan_dims d;
    PlanPtrs o;
    double l_width, layer_gap, max_map_len;
    double l_start, l_stop, l_step; // lateral samples: np.linspace(l_start, l_stop, n_l)
};

// One workgroup per scene: ego state, lattice layers, every obstacle frame to
// Frenet, lane filter and the obstacle list in the reference's order (statics,
// then each moving obstacle's frames; sl_map.py:86-109).  The line's sample
// table sits in LDS (x, y per sample).  Workgroup 0 also writes the lateral
// sample row shared by all scenes.
__global__ void plan_frontend_kernel(FrontDev a)
{
    extern __shared__ double smem[];
    __shared__ int wave_cnt[16];
    __shared__ int base_sh, nlay_sh;
    __shared__ double rmax_sh[16];
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & (MP_WAVE - 1), wave = tid / MP_WAVE, nw = nt / MP_WAVE;
    const mp_plan_dims &d = a.d;
    const int m_max = d.m_max;
    if (b == 0) {
        // np.linspace: i*step + start, the last sample set to stop (sl_map.py:74)
        for (int i = tid; i < d.n_l; i += nt)
            a.o.l_row[i] = (d.n_l > 1 && i == d.n_l - 1) ? a.l_stop : (double)i * a.l_step + a.l_start;
    }
    const int n_st = d.n_st_max > 0 ? min(max(a.in.n_st[b], 0), d.n_st_max) : 0;
    const int n_dy = d.n_dy_max > 0 ? min(max(a.in.n_dy[b], 0), d.n_dy_max) : 0;
    const int F = d.n_frames;
    const int n_slot = d.n_st_max + d.n_dy_max * F;
    const int ob_stride = NOB(d);
    const int M = a.o.tab_m[b];
    const bool line_ok = a.o.rl_status[b] == MP_RL_OK && M > 0;
    const double end_s = a.o.origin[4 * b + 2];
    // (world-frame call: a scene that kept more obstacles than its slots hold is not planned)
    const bool crowded = a.o.w_status && a.o.w_status[b] != 0;
    // A line with more samples than the tables hold whose knots alone span more than the map
    // may be long: the reference builds its converter and then raises "map too big" for it
    // (sl_map.py:57-60; the arc length is at least the span), so that is the scene's outcome,
    // whatever m_max the caller chose.
    bool too_long = false;
    if (!line_ok && a.o.rl_status[b] == MP_RL_E_CAPACITY) {
        const int nk = a.in.n_knots[b];
        const double *kx = a.in.knots_x + (size_t)b * d.n_knots_max;
        too_long = nk >= 2 && nk <= d.n_knots_max && kx[nk - 1] - kx[0] > a.max_map_len;
    }
    if (!line_ok || end_s > a.max_map_len || crowded) {
        if (tid == 0) {
            a.o.status[b] = too_long ? MP_PLAN_E_MAP_TOO_BIG : !line_ok ? MP_PLAN_E_BAD_LINE
                                     : (end_s > a.max_map_len ? MP_PLAN_E_MAP_TOO_BIG : MP_PLAN_E_CAPACITY);
            a.o.n_layers[b] = 0;
            a.o.n_ob[b] = 0;
            a.o.order_ws[b] = 0;
            a.o.ob_dist[b] = 0.0;
            a.o.ego_sl[2 * b] = a.o.ego_sl[2 * b + 1] = a.o.robot_vel[b] = __builtin_nan("");
        }
        return;
    }
    double *tx = smem, *ty = smem + m_max, *cos_t = smem + 2 * m_max, *sin_t = smem + 3 * m_max;
    double *ylo = smem + 4 * m_max, *yhi = ylo + 64; // per block of samples, see c2f_nearest_boxes
    const int bsh = c2f_block_shift(M);
    const size_t tb = (size_t)b * m_max;
    for (int k = tid; k < M; k += nt) {
        tx[k] = a.o.tab_x[tb + k];
        ty[k] = a.o.tab_y[tb + k];
        // heading of every sample: a few hundred evaluations per scene instead of one
        // per obstacle frame
        const double th = a.o.tab_theta[tb + k];
        cos_t[k] = cos(th);
        sin_t[k] = sin(th);
    }
    if (tid == 0) base_sh = 0;
    __syncthreads();
    for (int q = tid; q < ((M + (1 << bsh) - 1) >> bsh); q += nt) {
        const int first = q << bsh, end = min(first + (1 << bsh), M);
        double lo = ty[first], hi = lo;
        for (int k = first + 1; k < end; ++k) {
            lo = ty[k] < lo ? ty[k] : lo;
            hi = ty[k] > hi ? ty[k] : hi;
        }
        ylo[q] = lo;
        yhi[q] = hi;
    }
    __syncthreads();
    const double *ts = a.o.tab_s + tb, *tth = a.o.tab_theta + tb;

    if (tid == 0) {
        // ego at the robot frame's origin (sl_map.py:38-42)
        double s, l;
        const int k = c2f_point(tx, ty, ts, tth, M, 0.0, 0.0, &s, &l);
        a.o.ego_sl[2 * b] = s;
        a.o.ego_sl[2 * b + 1] = l;
        a.o.robot_vel[b] = c2f_s_dt(tth[k], a.o.tab_kappa[tb + k], l, a.in.ego_vel[2 * b], a.in.ego_vel[2 * b + 1]);
    }
    if (tid == (nt > MP_WAVE ? MP_WAVE : 0)) { // first lane of the second wave, if there is one
        // layers every layer_gap, the line's end as the last one (sl_map.py:53-66)
        double *lay = a.o.s_layers + (size_t)b * d.n_s_max;
        double ss = a.layer_gap;
        int n = 0;
        bool full = false;
        while (true) {
            if (n >= d.n_s_max) { full = true; break; }
            lay[n++] = ss;
            ss += a.layer_gap;
            if (ss + a.layer_gap / 2 >= end_s) {
                if (n >= d.n_s_max) { full = true; break; }
                lay[n++] = end_s;
                break;
            }
        }
        a.o.n_layers[b] = full ? 0 : n;
        nlay_sh = full ? 0 : n;
        a.o.status[b] = full ? MP_PLAN_E_CAPACITY : MP_PLAN_OK;
    }

    // slots no obstacle uses read 0 in the Frenet arrays
    for (int p = tid; p < n_slot; p += nt) {
        const bool is_st = p < d.n_st_max;
        const int o = is_st ? p : (p - d.n_st_max) / F;
        if (is_st ? o < n_st : o < n_dy) continue;
        if (is_st) {
            a.o.st_s[(size_t)b * ONE(d.n_st_max) + o] = 0.0;
            a.o.st_l[(size_t)b * ONE(d.n_st_max) + o] = 0.0;
        } else {
            const size_t q = (size_t)b * ONE(d.n_dy_max) * F + (p - d.n_st_max);
            a.o.dy_s[q] = 0.0;
            a.o.dy_l[q] = 0.0;
        }
    }
    // the scene's own obstacle frames, statics first (the slots in use are the first n_st of
    // the statics' and the first n_dy * F of the dynamics')
    const int n_used = n_st + n_dy * F;
    double r_loc = 0.0;
    for (int p0 = 0; p0 < n_used; p0 += nt) {
        const int pu = p0 + tid;
        const int p = pu < n_st ? pu : d.n_st_max + (pu - n_st);
        bool keep = false;
        double s = 0.0, l = 0.0;
        if (pu < n_used) {
            const bool is_st = p < d.n_st_max;
            const int o = is_st ? p : (p - d.n_st_max) / F;
            const bool used = true;
            double rad = 0.0;
            if (used) {
                double x, y;
                if (is_st) {
                    const size_t q = (size_t)b * d.n_st_max + o;
                    x = a.in.st_pos[2 * q];
                    y = a.in.st_pos[2 * q + 1];
                    rad = a.in.st_rad[q];
                } else {
                    const int f = (p - d.n_st_max) - o * F;
                    const size_t q = (size_t)b * d.n_dy_max + o;
                    x = a.in.dy_pos[2 * (q * F + f)];
                    y = a.in.dy_pos[2 * (q * F + f) + 1];
                    rad = a.in.dy_rad[q];
                }
                c2f_point_tab(tx, ty, ts, cos_t, sin_t, M, x, y, &s, &l, ylo, yhi, bsh);
                keep = fabs(l) - rad < a.l_width; // check_in_line (sl_map.py:111-112)
                if (keep) r_loc = rad > r_loc ? rad : r_loc;
            }
            if (is_st) {
                a.o.st_s[(size_t)b * ONE(d.n_st_max) + o] = s;
                a.o.st_l[(size_t)b * ONE(d.n_st_max) + o] = l;
            } else {
                const size_t q = (size_t)b * ONE(d.n_dy_max) * F + (p - d.n_st_max);
                a.o.dy_s[q] = s;
                a.o.dy_l[q] = l;
            }
        }
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) wave_cnt[wave] = __popcll(bal);
        __syncthreads();
        int pos = base_sh;
        for (int w = 0; w < wave; ++w) pos += wave_cnt[w];
        pos += __popcll(bal & ((1ull << lane) - 1ull));
        if (keep) {
            a.o.ob_s[(size_t)b * ob_stride + pos] = s;
            a.o.ob_l[(size_t)b * ob_stride + pos] = l;
        }
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int w = 0; w < nw; ++w) tot += wave_cnt[w];
            base_sh += tot;
        }
        __syncthreads();
    }
    r_loc = -wave_min_f64(-r_loc);
    if (lane == 0) rmax_sh[wave] = r_loc;
    __syncthreads();
    if (tid == 0) {
        double r = rmax_sh[0];
        for (int w = 1; w < nw; ++w) r = rmax_sh[w] > r ? rmax_sh[w] : r;
        a.o.n_ob[b] = base_sh;
        a.o.ob_dist[b] = r; // max radius of the kept entries (sl_map.py:94,106), 0 if none
        // the scene's weight for the ordered path search: obstacle entries x layers
        // (what sl_weight_narrow_kernel would compute from these two counts)
        a.o.order_ws[b] = mp_weight_bin((double)(base_sh + 1) * (double)nlay_sh);
    }
}

// ---------------------------------------------------------------------------
static int pipe_launch_check(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        mp_set_error("%s: %s", what, hipGetErrorString(e));
        return MP_E_LAUNCH;
    }
    return MP_OK;
}

static int plan_scenes_impl(const mp_scene_batch *scenes, const mp_plan_dims *dims, const mp_plan_params *params,
                            void *arena, int64_t arena_bytes, void *stream, PlanPtrs *ptrs_out,
                            hipEvent_t *stage_ev = nullptr, const mp_world_batch *world = nullptr);

extern "C" int mp_plan_scenes(const mp_scene_batch *scenes, const mp_plan_dims *dims, const mp_plan_params *params,
                              void *arena, int64_t arena_bytes, void *stream)
{
    return plan_scenes_impl(scenes, dims, params, arena, arena_bytes, stream, nullptr);
}

extern "C" int mp_plan_world_scenes(const mp_world_batch *world, const mp_plan_dims *dims,
                                    const mp_plan_params *params, void *arena, int64_t arena_bytes, void *stream)
{
    if (!world) { mp_set_error("mp_plan_world_scenes: null scenes"); return MP_E_BADARG; }
    return plan_scenes_impl(nullptr, dims, params, arena, arena_bytes, stream, nullptr, nullptr, world);
}

// stage_ev: null, or MP_PLAN_N_STAGES + 1 events; event 0 is recorded before the first
// stage and event i + 1 behind stage i (mp_plan_scenes_timed).
// world: null (robot-frame scenes in *scenes), or world-frame scenes: the ingestion stage then
// writes the robot-frame scene into the arena and the chain runs on that.
static int plan_scenes_impl(const mp_scene_batch *scenes, const mp_plan_dims *dims, const mp_plan_params *params,
                            void *arena, int64_t arena_bytes, void *stream, PlanPtrs *ptrs_out,
                            hipEvent_t *stage_ev, const mp_world_batch *world)
{
    int rc = check_dims(dims);
    if (rc) return rc;
    if ((!scenes && !world) || !params || !arena) { mp_set_error("null scenes/params/arena"); return MP_E_BADARG; }
    const mp_plan_dims d = *dims;
    if (world) {
        const mp_world_batch &w = *world;
        if (!d.world) { mp_set_error("mp_plan_world_scenes: dims.world must be 1 (the arena holds the scene)"); return MP_E_BADARG; }
        if (w.ob_max < 0 || !(w.frame_dt > 0)) { mp_set_error("mp_world_batch: bad ob_max / frame_dt"); return MP_E_BADARG; }
        if (!w.rot || !w.trans || !w.ego_vel_w || !w.way_xy || !w.n_way || !w.n_ob ||
            (w.ob_max > 0 && (!w.ob_pos || !w.ob_vel || !w.ob_rad))) {
            mp_set_error("mp_world_batch: null array");
            return MP_E_BADARG;
        }
        if (w.ob_max > 0 && (d.n_st_max <= 0 || d.n_dy_max <= 0)) {
            mp_set_error("mp_plan_world_scenes: n_st_max and n_dy_max must be positive when scenes hold obstacles");
            return MP_E_BADARG;
        }
    }
    mp_scene_batch in;
    if (scenes) in = *scenes;
    else memset(&in, 0, sizeof(in));
    if (!world && (!in.knots_x || !in.knots_y || !in.n_knots || !in.line_vec || !in.ego_vel ||
        (d.n_st_max > 0 && (!in.st_pos || !in.st_rad || !in.n_st)) ||
        (d.n_dy_max > 0 && (!in.dy_pos || !in.dy_rad || !in.n_dy)))) {
        mp_set_error("mp_scene_batch: null array");
        return MP_E_BADARG;
    }
    if (((uintptr_t)arena & 255) != 0) { mp_set_error("arena must be 256-byte aligned"); return MP_E_BADARG; }
    ArrInfo tab[PLAN_N_ARRAYS];
    int64_t total;
    plan_array_table(d, tab, &total);
    if (arena_bytes < total) {
        mp_set_error("arena too small: %lld bytes needed", (long long)total);
        return MP_E_BADARG;
    }
    if (d.n_l > 256 || (size_t)d.n_s_max * d.n_l > 65536 || d.n_t > 64 || d.st_n_s_max > 1024 || d.r_max > 256 ||
        d.m_max > 4096 || d.p_max > 4096 || d.q_max > 4096 || d.n_s_max + 1 > 128) {
        mp_set_error("mp_plan_scenes: a size exceeds what the kernels support");
        return MP_E_UNSUPPORTED;
    }
    PlanPtrs o;
    {
        int i = 0;
        char *base = (char *)arena;
#define X(name, ctype, dt, d0, d1, d2, d3) \
    o.name = tab[i].bytes ? (ctype *)(base + tab[i].offset) : nullptr; \
    ++i;
        PLAN_ARRAYS(X)
#undef X
    }
    if (ptrs_out) *ptrs_out = o;
    hipStream_t st = (hipStream_t)stream;
    const mp_plan_params &p = *params;
    const int B = d.B;
    const int mid_l = p.sl.mid_l >= 0 ? p.sl.mid_l : (int)((d.n_l - 1) / 2);
    if (mid_l >= d.n_l) { mp_set_error("mid_l out of range"); return MP_E_BADARG; }

    int stage_no = 0;
#define PLAN_STAGE_DONE()                                                                         \
    do {                                                                                          \
        if (stage_ev && hipEventRecord(stage_ev[++stage_no], st) != hipSuccess) {                 \
            mp_set_error("mp_plan_scenes_timed: hipEventRecord failed");                          \
            return MP_E_LAUNCH;                                                                   \
        }                                                                                         \
    } while (0)
    if (stage_ev && hipEventRecord(stage_ev[0], st) != hipSuccess) {
        mp_set_error("mp_plan_scenes_timed: hipEventRecord failed");
        return MP_E_LAUNCH;
    }
    // result rows whose tails nobody writes start from zero ("zeros" doubles as the ego
    // position (0, 0) and as the path spline's end slope)
    if (hipMemsetAsync(arena, 0, (size_t)tab[PLAN_N_ZEROED].offset, st) != hipSuccess) {
        mp_set_error("mp_plan_scenes: memset failed");
        return MP_E_LAUNCH;
    }
    PLAN_STAGE_DONE();

    // 0. world-frame scenes: into the robot frame, obstacles classified and filtered, the
    //    moving ones' frames rolled out (coor_trans); the chain then reads the arena's scene
    if (world) {
        const mp_world_batch &w = *world;
        rc = mp_scene_ingest_into(B, w.rot, w.trans, w.ego_vel_w, w.way_xy, w.n_way, d.n_knots_max, w.ob_pos,
                                  w.ob_vel, w.ob_rad, w.n_ob, w.ob_max, w.ignore_dist, d.n_frames, w.frame_dt,
                                  o.w_knots_x, o.w_knots_y, o.w_ego_vel, o.w_st_pos, o.w_st_rad, o.w_n_st,
                                  ONE(d.n_st_max), o.w_dy_pos, o.w_dy_rad, o.w_n_dy, ONE(d.n_dy_max), o.w_status,
                                  stream);
        if (rc) return rc;
        in.knots_x = o.w_knots_x;
        in.knots_y = o.w_knots_y;
        in.n_knots = w.n_way;
        in.line_vec = w.line_vec ? w.line_vec : o.zeros; // zeros: B x 2 cleared doubles
        in.ego_vel = o.w_ego_vel;
        in.st_pos = o.w_st_pos;
        in.st_rad = o.w_st_rad;
        in.n_st = o.w_n_st;
        in.dy_pos = o.w_dy_pos;
        in.dy_rad = o.w_dy_rad;
        in.n_dy = o.w_n_dy;
    }
    PLAN_STAGE_DONE();

    // 1. Frenet frame of every scene, the ego at the robot frame's origin
    rc = mp_refline_tables(B, in.knots_x, in.knots_y, in.n_knots, d.n_knots_max, in.line_vec, o.zeros, p.d_x,
                           d.m_max, o.tab_x, o.tab_y, o.tab_s, o.tab_theta, o.tab_kappa, o.tab_m, o.origin,
                           o.rl_status, o.inv_x, o.inv_s, o.inv_y, o.inv_theta, o.n_inv, stream);
    if (rc) return rc;
    PLAN_STAGE_DONE();

    // 2. ego state, layers, obstacle frames to Frenet, obstacle list
    {
        FrontDev f;
        f.in = in;
        f.d = d;
        f.o = o;
        f.l_width = p.l_width;
        f.layer_gap = p.layer_gap;
        f.max_map_len = p.max_map_len;
        f.l_start = 0.0 - (p.l_width - p.save_width) / 2.0; // sl_map.py:71-72
        f.l_stop = 0.0 + (p.l_width - p.save_width) / 2.0;
        f.l_step = d.n_l > 1 ? (f.l_stop - f.l_start) / (double)(d.n_l - 1) : 0.0;
        const size_t lds = ((size_t)4 * d.m_max + 128) * sizeof(double); // tables + block bounds
        if (lds > 64 * 1024) { mp_set_error("m_max too large"); return MP_E_UNSUPPORTED; }
        const int n_slot = d.n_st_max + d.n_dy_max * d.n_frames;
        int threads = n_slot <= 64 ? 64 : (n_slot <= 128 ? 128 : 256);
        if (mp_knob_fe_threads() > 0) threads = mp_knob_fe_threads(); // development aid
        hipLaunchKernelGGL(plan_frontend_kernel, dim3(B), dim3(threads), lds, st, f);
        rc = pipe_launch_check("mp_plan_scenes (front end)");
        if (rc) return rc;
    }
    PLAN_STAGE_DONE();

    const bool track = p.path_planner == 1;

    // 3. path search over the lattice
    if (!track) {
        mp_sl_batch sb;
        sb.B = B;
        sb.n_s = d.n_s_max;
        sb.n_l = d.n_l;
        sb.ob_stride = NOB(d);
        sb.s_layers = o.s_layers;
        sb.l_map = o.l_row;
        sb.l_bstride = 0; // one lateral row for every scene and layer
        sb.l_rstride = 0;
        sb.ego_sl = o.ego_sl;
        sb.ob_s = o.ob_s;
        sb.ob_l = o.ob_l;
        sb.n_ob = o.n_ob;
        sb.ob_dist = o.ob_dist;
        sb.n_layers = o.n_layers;
        mp_sl_params sp = p.sl;
        sp.mid_l = mid_l;
        // (scenes differ in obstacles and layers by an order of magnitude: heaviest first)
        rc = mp_sl_plan_keyed(&sb, &sp, o.cost_map, o.index_map, o.path_idx, o.sl_status, o.layers_done,
                              o.best_cost, o.order_ws, stream);
        if (rc) return rc;
    }
    PLAN_STAGE_DONE();
    // 4. knots of the path spline, the path every path_dt, ST grid width (one kernel)
    {
        PlanPathDev a;
        a.n_s_max = d.n_s_max; a.p_max = d.p_max; a.st_n_s_max = d.st_n_s_max; a.track = track; a.mid_l = mid_l;
        a.path_dt = p.path_dt; a.st_ds = p.st_ds;
        a.status = o.status; a.sl_status = o.sl_status; a.layers_done = o.layers_done; a.path_idx = o.path_idx;
        a.n_layers = o.n_layers; a.best_cost = o.best_cost;
        a.ego_sl = o.ego_sl; a.s_layers = o.s_layers; a.l_row = o.l_row;
        a.zeros = o.zeros; // end slope 0.0 (path_planner.py:53)
        a.knots_t = o.knots_t; a.knots_v = o.knots_v; a.n_pk = o.n_pk;
        a.path_s = o.path_s; a.path_l = o.path_l; a.n_path = o.n_path; a.path_status = o.path_status;
        a.st_n_s = o.st_n_s;
        rc = mp_plan_path_stage(B, a, stream);
        if (rc) return rc;
    }
    PLAN_STAGE_DONE();

    // 5. blocked intervals, 6. speed search
    rc = mp_st_regions_strided(B, d.n_t, o.path_s, o.path_l, o.n_path, d.p_max, o.st_s, o.st_l, in.st_rad, in.n_st,
                               d.n_st_max, o.dy_s, o.dy_l, in.dy_rad, in.n_dy, d.n_dy_max, d.n_frames, d.r_max,
                               o.st_down, o.st_up, o.st_n_reg, o.reg_status, stream, o.st_n_s, o.order_ws);
    if (rc) return rc;
    PLAN_STAGE_DONE();
    rc = mp_st_dp_keyed(B, d.n_t, d.st_n_s_max, o.st_n_s, o.st_down, o.st_up, o.st_n_reg, d.r_max, o.robot_vel,
                        p.speed_lim, p.st_ds, p.st_dt, o.st_cost, o.st_pre, o.st_path_s_ind, o.st_path_len,
                        o.order_ws, stream);
    if (rc) return rc;
    PLAN_STAGE_DONE();

    // 7. knots of the speed spline, speed profile every dt/5, path points at its arc lengths,
    //    those points back in the robot frame, final status (one kernel)
    {
        PlanTailDev a;
        a.n_t = d.n_t; a.n_pk_max = d.n_s_max + 1; a.q_max = d.q_max; a.m_max = d.m_max;
        a.st_ds = p.st_ds; a.st_dt = p.st_dt;
        a.curve_dt = p.st_dt * 1.0 / 5; // speed_planner.py:22
        a.path_dt = p.path_dt;
        a.status = o.status; a.reg_status = o.reg_status;
        a.st_path_len = o.st_path_len; a.st_path_s_ind = o.st_path_s_ind;
        a.sp_t = o.sp_t; a.sp_s = o.sp_s; a.robot_vel = o.robot_vel; a.zeros = o.zeros;
        a.knots_t = o.knots_t; a.knots_v = o.knots_v; a.n_pk = o.n_pk;
        a.prof_t = o.prof_t; a.ss = o.ss; a.vv = o.vv; a.n_prof = o.n_prof; a.prof_status = o.prof_status;
        a.final_s = o.final_s; a.final_l = o.final_l; a.n_final = o.n_final; a.final_status = o.final_status;
        a.inv_x = o.inv_x; a.inv_s = o.inv_s; a.inv_y = o.inv_y; a.inv_theta = o.inv_theta; a.n_inv = o.n_inv;
        a.final_x = o.final_x; a.final_y = o.final_y;
        a.rot = world ? world->rot : nullptr;
        a.trans = world ? world->trans : nullptr;
        a.world_x = world ? o.world_x : nullptr;
        a.world_y = world ? o.world_y : nullptr;
        rc = mp_plan_tail_stage(B, a, stream);
        if (rc) return rc;
    }
    PLAN_STAGE_DONE();
#undef PLAN_STAGE_DONE
    if (stage_ev && stage_no != MP_PLAN_N_STAGES) {
        mp_set_error("mp_plan_scenes_timed: %d stages recorded, %d expected", stage_no, MP_PLAN_N_STAGES);
        return MP_E_LAUNCH;
    }
    return MP_OK;
}

static const char *const g_stage_names[MP_PLAN_N_STAGES] = {
    "clear result rows (memset)", "scene ingestion (world to robot frame; world-frame scenes only)",
    "reference-line tables", "front end (Frenet map, obstacle list)",
    "SL path DP", "path spline (knots, samples, ST grid width)", "ST blocked intervals", "ST speed DP",
    "speed profile, path points, Frenet to Cartesian"};

extern "C" const char *mp_plan_stage_name(int32_t i)
{
    return (i >= 0 && i < MP_PLAN_N_STAGES) ? g_stage_names[i] : nullptr;
}

static int plan_scenes_timed_impl(const mp_scene_batch *scenes, const mp_world_batch *world, const mp_plan_dims *dims,
                                  const mp_plan_params *params, void *arena, int64_t arena_bytes, void *stream,
                                  float *stage_ms);

extern "C" int mp_plan_scenes_timed(const mp_scene_batch *scenes, const mp_plan_dims *dims,
                                    const mp_plan_params *params, void *arena, int64_t arena_bytes, void *stream,
                                    float *stage_ms)
{
    return plan_scenes_timed_impl(scenes, nullptr, dims, params, arena, arena_bytes, stream, stage_ms);
}

extern "C" int mp_plan_world_scenes_timed(const mp_world_batch *world, const mp_plan_dims *dims,
                                          const mp_plan_params *params, void *arena, int64_t arena_bytes,
                                          void *stream, float *stage_ms)
{
    if (!world) { mp_set_error("mp_plan_world_scenes_timed: null scenes"); return MP_E_BADARG; }
    return plan_scenes_timed_impl(nullptr, world, dims, params, arena, arena_bytes, stream, stage_ms);
}

static int plan_scenes_timed_impl(const mp_scene_batch *scenes, const mp_world_batch *world, const mp_plan_dims *dims,
                                  const mp_plan_params *params, void *arena, int64_t arena_bytes, void *stream,
                                  float *stage_ms)
{
    if (!stage_ms) { mp_set_error("mp_plan_scenes_timed: null stage_ms"); return MP_E_BADARG; }
    hipEvent_t ev[MP_PLAN_N_STAGES + 1];
    int made = 0;
    int rc = MP_OK;
    for (; made <= MP_PLAN_N_STAGES; ++made)
        if (hipEventCreate(&ev[made]) != hipSuccess) { mp_set_error("hipEventCreate failed"); rc = MP_E_LAUNCH; break; }
    if (!rc) rc = plan_scenes_impl(scenes, dims, params, arena, arena_bytes, stream, nullptr, ev, world);
    if (!rc && hipEventSynchronize(ev[MP_PLAN_N_STAGES]) != hipSuccess) {
        mp_set_error("mp_plan_scenes_timed: hipEventSynchronize failed");
        rc = MP_E_LAUNCH;
    }
    for (int i = 0; !rc && i < MP_PLAN_N_STAGES; ++i)
        if (hipEventElapsedTime(&stage_ms[i], ev[i], ev[i + 1]) != hipSuccess) {
            mp_set_error("mp_plan_scenes_timed: hipEventElapsedTime failed");
            rc = MP_E_LAUNCH;
        }
    for (int i = 0; i < made; ++i) (void)hipEventDestroy(ev[i]);
    return rc;
}

// ---------------------------------------------------------------------------
// The whole replan for callers that cannot allocate device memory.
// mp_plan_scenes_static is mp_plan_scenes with the arena (every intermediate
// array of the chain) in a fixed device array that is part of the library's code
// object; the caller passes only the scenes and the result arrays.  One array per
// GPU, shared by every call on that GPU and keeping the previous call's contents:
// the chain clears what it needs cleared (the same memset as mp_plan_scenes), and
// two calls that overlap on one GPU (two streams, two threads) race.
// ---------------------------------------------------------------------------
static __device__ __align__(256) unsigned char scratch[MP_PLAN_STATIC_BYTES];

#define PLAN_STATIC_MAX_DEVICES 64
static std::mutex g_scratch_mutex;
static void *g_scratch_ptr[PLAN_STATIC_MAX_DEVICES]; // device address of `scratch` on each GPU, looked up once

// Device address of the current GPU's copy of `scratch` (the array's name is only a
// host-side handle; every GPU holds its own copy at its own address).
static int plan_static_arena(void **out)
{
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= PLAN_STATIC_MAX_DEVICES) {
        mp_set_error("mp_plan_scenes_static: no usable current device (%s)",
                     e != hipSuccess ? hipGetErrorString(e) : "device index out of range");
        return MP_E_LAUNCH;
    }
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void *p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            mp_set_error("mp_plan_scenes_static: hipGetSymbolAddress failed (%s)", hipGetErrorString(e));
            return MP_E_LAUNCH;
        }
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return MP_OK;
}

struct PlanOutDev {
    int32_t B, q_max;
    const int32_t *status, *n_final;
    const double *final_x, *final_y, *vv;
    mp_plan_result out;
};

// Results out of the arena into the caller's arrays, one thread per profile slot.
__global__ void plan_copy_out_kernel(PlanOutDev a)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.B * a.q_max) return;
    // slots behind a scene's last point read 0 (a scene that lost its plan in the last
    // stage keeps converted points in the arena, but reports none)
    const int b = i / a.q_max, j = i - b * a.q_max;
    const bool live = a.status[b] == MP_PLAN_OK && j < a.n_final[b];
    a.out.x[i] = live ? a.final_x[i] : 0.0;
    a.out.y[i] = live ? a.final_y[i] : 0.0;
    a.out.speed[i] = live ? a.vv[i] : 0.0;
    if (i < a.B) {
        const int st = a.status[i];
        a.out.status[i] = st;
        a.out.n_points[i] = st == MP_PLAN_OK ? a.n_final[i] : 0;
    }
}

extern "C" int mp_plan_scenes_static(const mp_scene_batch *scenes, const mp_plan_dims *dims,
                                     const mp_plan_params *params, const mp_plan_result *result, void *stream)
{
    int rc = check_dims(dims);
    if (rc) return rc;
    if (!result || !result->status || !result->n_points || !result->x || !result->y || !result->speed) {
        mp_set_error("mp_plan_scenes_static: null result array");
        return MP_E_BADARG;
    }
    if (dims->want_maps) {
        mp_set_error("mp_plan_scenes_static: want_maps is not available (the tables stay in the library's scratch)");
        return MP_E_UNSUPPORTED;
    }
    const int64_t need = mp_plan_arena_bytes(dims);
    if (need < 0 || need > (int64_t)MP_PLAN_STATIC_BYTES) {
        mp_set_error("mp_plan_scenes_static: these sizes need %lld bytes of scratch, the library holds %lld; "
                     "plan fewer scenes per call or use mp_plan_scenes with a caller-owned arena",
                     (long long)need, (long long)MP_PLAN_STATIC_BYTES);
        return MP_E_UNSUPPORTED;
    }
    void *arena = nullptr;
    rc = plan_static_arena(&arena);
    if (rc) return rc;
    PlanPtrs o;
    rc = plan_scenes_impl(scenes, dims, params, arena, (int64_t)MP_PLAN_STATIC_BYTES, stream, &o);
    if (rc) return rc;
    PlanOutDev c{dims->B, dims->q_max, o.status, o.n_final, o.final_x, o.final_y, o.vv, *result};
    const int n = dims->B * dims->q_max;
    hipLaunchKernelGGL(plan_copy_out_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, c);
    return pipe_launch_check("mp_plan_scenes_static (results)");
}
