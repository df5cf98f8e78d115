// Stages of the whole replan (pipe_kernels.hip) that run as one kernel each because they
// are short and depend on one another scene by scene: spline knots, spline evaluation and
// the scalars around them.  The kernels live beside the spline code (c2f_kernels.hip); the
// chain fills these argument blocks from its arena.
#pragma once
#include <stdint.h>

// Path stage: outcome of the path search and knots of the path spline, [ego] + chosen nodes
// (path_planner.py:40-53; track_path_planner.py:25-35 takes the centre node of every
// layer); the path every path_dt (SLMap.path_sampling, sl_map.py:115-118); width of the ST
// grid from the last sample's arc length (planner_interface.py:145, st_map.py:17).
struct PlanPathDev {
    int32_t n_s_max, p_max, st_n_s_max, track, mid_l;
    double path_dt, st_ds;
    int32_t *status, *sl_status, *layers_done, *path_idx;
    const int32_t *n_layers;
    double *best_cost;
    const double *ego_sl, *s_layers, *l_row, *zeros;
    double *knots_t, *knots_v;
    int32_t *n_pk;
    double *path_s, *path_l;
    int32_t *n_path, *path_status, *st_n_s;
};

// Tail stage: knots of the speed spline, t = 0, dt, 2dt, ... by repeated addition and
// s = index * ds (speed_planner.py:35-46); the speed profile every dt/5 (speed_planner.py:22,47);
// the path spline at the profile's arc lengths (path_sampling(curve, False, ss),
// planner_interface.py:157); those points in the robot frame
// (cartesian_frenet_conversion.py:55-60) and, for world-frame scenes, in the world frame
// (robot_map.py:84-91); and the scene's final status.
struct PlanTailDev {
    int32_t n_t, n_pk_max, q_max, m_max;
    double st_ds, st_dt, curve_dt, path_dt;
    int32_t *status;
    const int32_t *reg_status;
    int32_t *st_path_len;
    const int32_t *st_path_s_ind;
    double *sp_t, *sp_s;
    const double *robot_vel, *zeros;
    const double *knots_t, *knots_v;
    const int32_t *n_pk;
    double *prof_t, *ss, *vv;
    int32_t *n_prof, *prof_status;
    double *final_s, *final_l;
    int32_t *n_final, *final_status;
    const double *inv_x, *inv_s, *inv_y, *inv_theta;
    const int32_t *n_inv;
    double *final_x, *final_y;
    // from world-frame scenes only (null otherwise): ego poses and the path in the world frame
    const double *rot, *trans;
    double *world_x, *world_y;
};

int mp_plan_path_stage(int32_t B, const PlanPathDev &a, void *stream);
int mp_plan_tail_stage(int32_t B, const PlanTailDev &a, void *stream);
