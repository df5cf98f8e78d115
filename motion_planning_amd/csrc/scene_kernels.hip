// World-frame scene ingestion and the way back for gfx950.
//
// What is computed (reference: Planning/Interface/robot_map.py:25-68 with
// Model/general_converter.py:14-26, Model/obstacle.py:36-64, Utils/tool.py:41-42
// and 58-59): every world point p becomes R^T (p - t) in the ego ("robot") frame,
// obstacles more than 1 m behind the ego are dropped, the rest are split into
// static (|vx|+|vy| < 0.001 on the WORLD velocity) and dynamic ones; a static
// obstacle is kept when it lies within ignore_dist of the ego, a dynamic one
// when any of its predicted frames does.  Frames are rolled out by repeated
// addition, pos[k+1] = pos[k] + vel*dt, as the reference does (not pos0 + k*vel*dt).
// Kept obstacles keep their input order.
//
// One wave per scene: lanes take obstacles 64 at a time, a ballot gives each
// kept obstacle its output slot.
#include "mp_device.h"

struct SceneDev {
    int32_t way_max, ob_max, n_frames;
    int32_t st_max, dy_max; // slots per scene of the kept statics / dynamics (mp_scene_ingest: ob_max each)
    int32_t *overflow;      // null, or per scene 1 when more obstacles were kept than the slots hold, else 0
    const double *rot, *trans, *ego_vel_w;
    const double *way_xy;
    const int32_t *n_way;
    const double *ob_pos, *ob_vel, *ob_rad;
    const int32_t *n_ob;
    double ignore_dist, dt;
    double *knots_x, *knots_y, *ego_vel_r;
    double *st_pos, *st_rad;
    int32_t *n_st;
    double *dy_pos, *dy_vel, *dy_rad;
    int32_t *n_dy;
    int32_t *ob_class;
};

// R^T v for the row-major 2x2 rotation r (np.dot(R.transpose(), v)).  NumPy
// hands the product to a BLAS dgemm, which on FMA-capable hosts accumulates
// a0*v0 first and fuses the second term onto it; rounding the same way keeps
// the rotated coordinates bit-identical to the reference's (its nearest-sample
// Frenet projection turns a last-bit difference into a 0.5 m jump).
__device__ __forceinline__ void rot_t(const double *r, double vx, double vy, double *ox, double *oy)
{
    *ox = fma(r[2], vy, __dmul_rn(r[0], vx));
    *oy = fma(r[3], vy, __dmul_rn(r[1], vx));
}

__global__ void scene_ingest_kernel(SceneDev a)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    const double r[4] = {a.rot[4 * b], a.rot[4 * b + 1], a.rot[4 * b + 2], a.rot[4 * b + 3]};
    const double tx = a.trans[2 * b], ty = a.trans[2 * b + 1];
    // reference line and ego velocity (robot_map.py:25-44)
    const int n_way = min(max(a.n_way[b], 0), a.way_max);
    for (int i = lane; i < a.way_max; i += MP_WAVE) {
        double x = 0.0, y = 0.0;
        if (i < n_way) {
            const double *p = a.way_xy + ((size_t)b * a.way_max + i) * 2;
            rot_t(r, __dsub_rn(p[0], tx), __dsub_rn(p[1], ty), &x, &y);
        }
        a.knots_x[(size_t)b * a.way_max + i] = x;
        a.knots_y[(size_t)b * a.way_max + i] = y;
    }
    if (lane == 0) {
        double vx, vy;
        rot_t(r, a.ego_vel_w[2 * b], a.ego_vel_w[2 * b + 1], &vx, &vy);
        a.ego_vel_r[2 * b] = vx;
        a.ego_vel_r[2 * b + 1] = vy;
    }
    // obstacles (robot_map.py:47-68)
    const int n_ob = min(max(a.n_ob[b], 0), a.ob_max); // a damaged count must not reach other scenes' slots
    const size_t ob0 = (size_t)b * a.ob_max;
    int n_st = 0, n_dy = 0; // wave-uniform running counts
    for (int base = 0; base < n_ob; base += MP_WAVE) {
        const int o = base + lane;
        int cls = -1; // -1: no obstacle in this lane
        double px = 0.0, py = 0.0, vx = 0.0, vy = 0.0;
        if (o < n_ob) {
            const double *p = a.ob_pos + (ob0 + o) * 2, *v = a.ob_vel + (ob0 + o) * 2;
            rot_t(r, __dsub_rn(p[0], tx), __dsub_rn(p[1], ty), &px, &py);
            if (px < -1.0) {
                cls = MP_OB_BEHIND;
            } else if (__dadd_rn(fabs(v[0]), fabs(v[1])) < 0.001) {
                const double d = sqrt(__dadd_rn(__dmul_rn(px, px), __dmul_rn(py, py)));
                cls = d < a.ignore_dist ? MP_OB_STATIC : MP_OB_STATIC_FAR;
            } else {
                rot_t(r, v[0], v[1], &vx, &vy);
                double qx = px, qy = py, dmin = MP_INF;
                for (int k = 0; k < a.n_frames; ++k) {
                    const double d = sqrt(__dadd_rn(__dmul_rn(qx, qx), __dmul_rn(qy, qy)));
                    dmin = d < dmin ? d : dmin;
                    qx = __dadd_rn(qx, __dmul_rn(vx, a.dt));
                    qy = __dadd_rn(qy, __dmul_rn(vy, a.dt));
                }
                cls = dmin < a.ignore_dist ? MP_OB_DYNAMIC : MP_OB_DYNAMIC_FAR;
            }
            if (a.ob_class) a.ob_class[ob0 + o] = cls;
        }
        const unsigned long long m_st = __ballot(cls == MP_OB_STATIC), m_dy = __ballot(cls == MP_OB_DYNAMIC);
        const unsigned long long below = (1ull << lane) - 1ull;
        const int k_st = n_st + __popcll(m_st & below), k_dy = n_dy + __popcll(m_dy & below);
        if (cls == MP_OB_STATIC && k_st < a.st_max) {
            const size_t slot = (size_t)b * a.st_max + k_st;
            a.st_pos[slot * 2] = px;
            a.st_pos[slot * 2 + 1] = py;
            a.st_rad[slot] = a.ob_rad[ob0 + o];
        } else if (cls == MP_OB_DYNAMIC && k_dy < a.dy_max) {
            const size_t slot = (size_t)b * a.dy_max + k_dy;
            double qx = px, qy = py;
            double *out = a.dy_pos + slot * a.n_frames * 2;
            for (int k = 0; k < a.n_frames; ++k) {
                out[2 * k] = qx;
                out[2 * k + 1] = qy;
                qx = __dadd_rn(qx, __dmul_rn(vx, a.dt));
                qy = __dadd_rn(qy, __dmul_rn(vy, a.dt));
            }
            if (a.dy_vel) {
                a.dy_vel[slot * 2] = vx;
                a.dy_vel[slot * 2 + 1] = vy;
            }
            a.dy_rad[slot] = a.ob_rad[ob0 + o];
        }
        n_st += __popcll(m_st);
        n_dy += __popcll(m_dy);
    }
    if (lane == 0) {
        a.n_st[b] = min(n_st, a.st_max);
        a.n_dy[b] = min(n_dy, a.dy_max);
        if (a.overflow) a.overflow[b] = (n_st > a.st_max || n_dy > a.dy_max) ? 1 : 0;
    }
}

// R p + t per point (GeneralConverter.robot_to_world, general_converter.py:19-22).
__global__ void robot_to_world_kernel(int n, const double *x, const double *y, const int32_t *pt_scn,
                                      const double *rot, const double *trans, double *wx, double *wy)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int b = pt_scn[i];
    const double *r = rot + 4 * (size_t)b;
    const double px = x[i], py = y[i];
    // same accumulation order as rot_t
    wx[i] = __dadd_rn(fma(r[1], py, __dmul_rn(r[0], px)), trans[2 * b]);
    wy[i] = __dadd_rn(fma(r[3], py, __dmul_rn(r[2], px)), trans[2 * b + 1]);
}

extern void mp_set_error(const char *fmt, ...);

static int scene_launch_check(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        mp_set_error("%s: %s", what, hipGetErrorString(e));
        return MP_E_LAUNCH;
    }
    return MP_OK;
}

extern "C" int mp_scene_ingest(int32_t B, const double *rot, const double *trans, const double *ego_vel_w,
                               const double *way_xy, const int32_t *n_way, int32_t way_max,
                               const double *ob_pos, const double *ob_vel, const double *ob_rad,
                               const int32_t *n_ob, int32_t ob_max, double ignore_dist, int32_t n_frames,
                               double dt, double *knots_x, double *knots_y, double *ego_vel_r, double *st_pos,
                               double *st_rad, int32_t *n_st, double *dy_pos, double *dy_vel, double *dy_rad,
                               int32_t *n_dy, int32_t *ob_class, void *stream)
{
    if (B <= 0 || way_max <= 0 || ob_max < 0 || n_frames <= 0) {
        mp_set_error("mp_scene_ingest: non-positive size");
        return MP_E_BADARG;
    }
    if (!rot || !trans || !ego_vel_w || !way_xy || !n_way || !n_ob || !knots_x || !knots_y || !ego_vel_r ||
        !n_st || !n_dy || (ob_max > 0 && (!ob_pos || !ob_vel || !ob_rad || !st_pos || !st_rad || !dy_pos || !dy_rad))) {
        mp_set_error("mp_scene_ingest: null array");
        return MP_E_BADARG;
    }
    SceneDev a{way_max, ob_max, n_frames, ob_max, ob_max, nullptr, rot, trans, ego_vel_w, way_xy, n_way, ob_pos,
               ob_vel, ob_rad, n_ob, ignore_dist, dt, knots_x, knots_y, ego_vel_r, st_pos, st_rad, n_st, dy_pos,
               dy_vel, dy_rad, n_dy, ob_class};
    hipLaunchKernelGGL(scene_ingest_kernel, dim3(B), dim3(MP_WAVE), 0, (hipStream_t)stream, a);
    return scene_launch_check("mp_scene_ingest");
}

// The same kernel for the whole replan from world-frame scenes: kept statics / dynamics go
// into rows of st_max / dy_max slots (fewer than ob_max where the caller knows a bound), and
// overflow[b] tells whether scene b kept more than fit.  Arguments were checked by the chain.
int mp_scene_ingest_into(int32_t B, const double *rot, const double *trans, const double *ego_vel_w,
                         const double *way_xy, const int32_t *n_way, int32_t way_max, const double *ob_pos,
                         const double *ob_vel, const double *ob_rad, const int32_t *n_ob, int32_t ob_max,
                         double ignore_dist, int32_t n_frames, double dt, double *knots_x, double *knots_y,
                         double *ego_vel_r, double *st_pos, double *st_rad, int32_t *n_st, int32_t st_max,
                         double *dy_pos, double *dy_rad, int32_t *n_dy, int32_t dy_max, int32_t *overflow,
                         void *stream)
{
    SceneDev a{way_max, ob_max, n_frames, st_max, dy_max, overflow, rot, trans, ego_vel_w, way_xy, n_way, ob_pos,
               ob_vel, ob_rad, n_ob, ignore_dist, dt, knots_x, knots_y, ego_vel_r, st_pos, st_rad, n_st, dy_pos,
               nullptr, dy_rad, n_dy, nullptr};
    hipLaunchKernelGGL(scene_ingest_kernel, dim3(B), dim3(MP_WAVE), 0, (hipStream_t)stream, a);
    return scene_launch_check("mp_plan_world_scenes (scene ingestion)");
}

extern "C" int mp_robot_to_world(int32_t n_pts, const double *x, const double *y, const int32_t *pt_scn,
                                 const double *rot, const double *trans, double *wx, double *wy, void *stream)
{
    if (n_pts < 0) { mp_set_error("mp_robot_to_world: negative size"); return MP_E_BADARG; }
    if (n_pts == 0) return MP_OK;
    if (!x || !y || !pt_scn || !rot || !trans || !wx || !wy) {
        mp_set_error("mp_robot_to_world: null array");
        return MP_E_BADARG;
    }
    hipLaunchKernelGGL(robot_to_world_kernel, dim3((n_pts + 255) / 256), dim3(256), 0, (hipStream_t)stream, n_pts,
                       x, y, pt_scn, rot, trans, wx, wy);
    return scene_launch_check("mp_robot_to_world");
}

// ---------------------------------------------------------------------------
// Result record of one scenario for the gather across GPUs (nothing in the
// reference corresponds to it).  Bytes, little endian:
//   [0,8) best_cost f64 | [8,10) layers_done u16 | [10] status i8 | [11] st_len u8
//   | st_path i16[n_t] | path_idx i16[n_s] | zero padding to a multiple of 8
// One thread per output byte, so stores are coalesced.
__global__ void pack_results_kernel(int B, int n_s, int n_t, int width, const int32_t *path_idx,
                                    const int32_t *status, const int32_t *layers_done, const double *best_cost,
                                    const int32_t *st_path, const int32_t *st_len, unsigned char *out)
{
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (size_t)B * width) return;
    const int b = (int)(g / width), o = (int)(g - (size_t)b * width);
    unsigned v = 0;
    if (o < 8) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(best_cost ? best_cost[b] : 0.0);
        v = (unsigned)(bits >> (8 * o));
    } else if (o < 10) {
        v = (unsigned)layers_done[b] >> (8 * (o - 8));
    } else if (o == 10) {
        v = (unsigned)status[b];
    } else if (o == 11) {
        v = (unsigned)st_len[b];
    } else if (o < 12 + 2 * n_t) {
        const int k = (o - 12) >> 1;
        v = (unsigned)st_path[(size_t)b * n_t + k] >> (8 * ((o - 12) & 1));
    } else if (o < 12 + 2 * n_t + 2 * n_s) {
        const int k = (o - 12 - 2 * n_t) >> 1;
        v = (unsigned)path_idx[(size_t)b * n_s + k] >> (8 * ((o - 12 - 2 * n_t) & 1));
    }
    out[g] = (unsigned char)(v & 0xFFu);
}

extern "C" int32_t mp_result_record_bytes(int32_t n_s, int32_t n_t)
{
    if (n_s <= 0 || n_t <= 0) return -1;
    return (12 + 2 * n_t + 2 * n_s + 7) & ~7;
}

extern "C" int mp_pack_results(int32_t B, int32_t n_s, int32_t n_t, const int32_t *path_idx, const int32_t *status,
                               const int32_t *layers_done, const double *best_cost, const int32_t *st_path,
                               const int32_t *st_len, void *records, void *stream)
{
    if (B <= 0 || n_s <= 0 || n_t <= 0) { mp_set_error("bad sizes"); return MP_E_BADARG; }
    if (!path_idx || !status || !layers_done || !st_path || !st_len || !records) {
        mp_set_error("null array");
        return MP_E_BADARG;
    }
    const int width = mp_result_record_bytes(n_s, n_t);
    const size_t total = (size_t)B * width;
    hipLaunchKernelGGL(pack_results_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       B, n_s, n_t, width, path_idx, status, layers_done, best_cost, st_path, st_len,
                       (unsigned char *)records);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { mp_set_error("mp_pack_results: %s", hipGetErrorString(e)); return MP_E_LAUNCH; }
    return MP_OK;
}
