/*
 * mplattice.h -- C ABI of libmplattice.so: batched lattice planning kernels
 * for AMD MI355X (gfx950).
 *
 * The reference planner (pure Python) has no FFI of its own; its boundary is
 * the Python class surface that Planning/Interface/planner_interface.py:133-159
 * calls.  Each entry point below names the reference method whose loop nest it
 * replaces.  A batch holds B independent scenarios; scenario b of every array
 * is the data one reference object would hold.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc / torch CUDA tensor) unless
 *     the parameter is a small by-value struct; the caller owns all buffers;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls
 *     only enqueue work, they never synchronise and never allocate;
 *   - the return value is 0 on success or a negative MP_E_* code for argument
 *     errors detected on the host (nothing was enqueued); per-scenario outcomes
 *     are written to the `status` array on the device;
 *   - all real numbers are IEEE double, indices are int32.
 *
 * Testing aids ("knobs": they select between code paths that must give
 * identical results, or tune a launch; unset = the normal choice).  Each is read
 * from the environment variable of the same name ONCE, at the library's first
 * launch that consults it, and can be changed afterwards only through
 * mp_set_knob(): MP_SL_NARROW=0 (lattices of up to 16 lanes through the
 * thread-per-node kernel), MP_SL_PAIR_CAP=n (size of the kernels' LDS pair
 * store), MP_ST_REGIONS_ONCE=0 (per-layer scans in mp_st_regions),
 * MP_ST_GENERIC=1 (mp_st_dp through the kernel that holds every relaxation path
 * instead of the instance built for the batch's interval bound r_max; =2: that
 * kernel without its per-cell interval tables, so that every layer with intervals
 * is relaxed by one thread per cell, the path grids too large for the tables take),
 * MP_ORDER=0 (the *_ordered entry points launch in the caller's order),
 * MP_ST_THREADS=n / MP_RL_THREADS=n / MP_FE_THREADS=n / MP_CV_THREADS=n (workgroup
 * size of mp_st_dp / mp_refline_tables / the front end / the two spline stages of
 * mp_plan_scenes; honoured only when a multiple of 64 within 64..1024), MP_ST_LANES=4|8 (lanes that share a
 * cell in mp_st_dp's relaxation), MP_ST_PREFIX=k (large batches of mp_st_dp plan
 * cells 0..k first and bound the rest of the grid by their result, on the kernel
 * for every path too, which plans in one sweep by default; 0: one sweep over the
 * whole grid), MP_SL_ROWS=n (predecessor rows mp_sl_plan's
 * kernel keeps in LDS before it leaves a scenario to its second launch).
 */
#ifndef MPLATTICE_H
#define MPLATTICE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MP_VERSION 124

/* host-side return codes */
#define MP_OK 0
#define MP_E_BADARG (-1)      /* null pointer / non-positive size                 */
#define MP_E_UNSUPPORTED (-2) /* size beyond what the kernel supports             */
#define MP_E_LAUNCH (-3)      /* hipLaunch failed; see mp_last_error()            */

/* per-scenario status written by the SL kernels */
#define MP_SL_FOUND 1
#define MP_SL_NOT_FOUND 0     /* path_planner.py:74-77 "Path not found!"          */
#define MP_SL_E_EMPTY_EDGE (-2) /* a layer gap shorter than d_s with obstacles present:
                                   the reference raises ValueError (min of empty) */
#define MP_SL_E_CAPACITY (-3) /* more than MP_SL_K_CAP samples on one edge        */
#define MP_SL_SKIPPED (-4)    /* the scenario's layer count is 0: nothing was planned */
/* Samples per edge the layer kernels (mp_sl_plan, mp_sl_edge_cost) hold: 64 m
 * between layers at the reference's d_s = 0.5, where SLMap's own lattices stop
 * at 12 m (sl_map.py:69-78).  A scenario with a longer edge gets
 * MP_SL_E_CAPACITY from them and nothing else of it is written;
 * mp_sl_edge_cost_nodes + mp_sl_dp price such a lattice (up to 2^20 samples an
 * edge), and the Python PathPlanner falls to that pair by itself. */
#define MP_SL_K_CAP 128

/* per-scenario status written by mp_st_regions */
#define MP_ST_OK 1
#define MP_ST_E_CAPACITY (-3) /* more blocked intervals at one time layer than r_max */

/* per-scenario status written by mp_refline_tables */
#define MP_RL_OK 1
#define MP_RL_E_INDEX (-1)    /* the reference raises IndexError (curve.py:57-65, 82-85) */
#define MP_RL_E_CAPACITY (-3) /* more than m_max samples / n_knots_max knots      */

/* Planning/DP_Path/path_planner.py:22-25 constants, overridable. */
typedef struct mp_sl_params {
    double d_s;          /* sample spacing along an edge (0.5)                    */
    double w_d;          /* guide/smooth mix (0.5)                                */
    double no_path_cost; /* 1e4: obstacle penalty = 3x, abort threshold = 10x     */
    int32_t early_exit;  /* 1: stop relaxing a scenario at the first failing layer
                            (reference behaviour); 0: relax every layer anyway    */
    int32_t mid_l;       /* guide lane index, sl_map.py:20 int((n_l-1)/2)         */
} mp_sl_params;

/* One batch of SL lattices (Planning/DP_Path/sl_map.py:50-81, 86-109).
 * Layer s of scenario b sits at arc length s_layers[b*n_s+s]; its lateral
 * samples are l_map[b*l_bstride + s*l_rstride + l] (l_rstride = 0 when every
 * layer shares one row).  Obstacles are struct-of-arrays, ob_stride slots per
 * scenario of which the first n_ob[b] are used; ob_dist[b] is the single
 * collision radius (sl_map.py:94,106).  SLMap derives the layer count from the
 * length of the reference line (sl_map.py:61-68), so scenarios of one batch may
 * use fewer than n_s layers: n_layers[b] (NULL = every scenario uses n_s) is
 * the count of scenario b, array strides stay n_s, the unused tail of every
 * output row keeps the initial values (cost 0, index -2, path -1), and a count
 * of 0 skips the scenario with status MP_SL_SKIPPED. */
typedef struct mp_sl_batch {
    int32_t B, n_s, n_l, ob_stride;
    const double *s_layers;  /* [B][n_s]                    */
    const double *l_map;
    int64_t l_bstride, l_rstride;
    const double *ego_sl;    /* [B][2] (s, l) of the ego    */
    const double *ob_s;      /* [B][ob_stride]              */
    const double *ob_l;      /* [B][ob_stride]              */
    const int32_t *n_ob;     /* [B]                         */
    const double *ob_dist;   /* [B]                         */
    const int32_t *n_layers; /* [B] or NULL                 */
} mp_sl_batch;

int mp_version(void);
const char *mp_last_error(void);
/* Sets a testing knob by the name of its environment variable; value < 0 = back
 * to "not set".  MP_E_BADARG for an unknown name or a value the knob does not
 * accept.  No counterpart in the reference. */
int mp_set_knob(const char *name, int32_t value);

/* Replaces PathPlanner.find_path (path_planner.py:66-101) with cal_cost/get_path
 * (:104-155) fused in: edge costs are produced and consumed on chip.
 * Limits: n_l <= 256, n_s*n_l <= 65536.
 * Outputs (cost_map, index_map, best_cost may be NULL):
 *   cost_map  [B][n_s][n_l]  rows not relaxed stay 0      (path_planner.py:20)
 *   index_map [B][n_s][n_l]  rows not relaxed stay -2, row 0 is -1 (:19-21,94)
 *   path_idx  [B][n_s]       chosen lane per layer, -1 when not found (:79-87)
 *   status    [B]            MP_SL_*
 *   layers_done [B]          number of layers relaxed
 *   best_cost [B]            cost of the chosen end node (NaN when not found) */
int mp_sl_plan(const mp_sl_batch *batch, const mp_sl_params *params, double *cost_map,
               int32_t *index_map, int32_t *path_idx, int32_t *status, int32_t *layers_done,
               double *best_cost, void *stream);

/* mp_sl_plan with the workgroups of the fused kernel launched heaviest scenario
 * first.  No counterpart in the reference (its planner handles one scenario at a
 * time, path_planner.py:66-88); a batch whose plans differ in length ends when
 * its last long plan does, and launched in the caller's order a long one may
 * start last.  order_ws: 2*B int32 of device scratch, contents irrelevant on
 * entry and on return (weights in the first half, the order in the second); two
 * small launches in front of the kernel fill it.  The weight is a guess from the
 * obstacles near the first layers and decides nothing but the order (batches
 * of up to 1536 scenarios, which fit on the chip at once, are launched as they
 * come and the workspace is not touched): outputs
 * are those of mp_sl_plan, bit for bit.  Lattices of up to 16 lanes, whose
 * kernel prices every edge, are weighted by obstacles x layers. */
int mp_sl_plan_ordered(const mp_sl_batch *batch, const mp_sl_params *params, double *cost_map,
                       int32_t *index_map, int32_t *path_idx, int32_t *status, int32_t *layers_done,
                       double *best_cost, int32_t *order_ws, void *stream);

/* Measurement aid, no counterpart in the reference: mp_sl_plan (maps not
 * written) by an instance of the same kernel that also counts the lattice edges
 * it actually prices -- exact pruning skips most of the n_l + (n_s-1)*n_l^2 edges
 * of a plan -- and ADDS that count, summed over the batch, to n_evaluated[0];
 * n_evaluated[1] receives the edges priced one after the other (the busiest lane
 * of every wave, layer by layer), which is what the kernel's time follows;
 * n_evaluated[2] the turns of the walk over the sources and n_evaluated[3] the
 * collision tests in them, each again for the busiest lane of every wave
 * (n_evaluated: four device uint64 the caller zeroes).  Lattices of more than 16 lanes only (the
 * narrow kernel prices every edge).  Results are identical to mp_sl_plan's. */
int mp_sl_plan_counted(const mp_sl_batch *batch, const mp_sl_params *params, int32_t *path_idx,
                       int32_t *status, int32_t *layers_done, double *best_cost, uint64_t *n_evaluated,
                       void *stream);

/* Replaces PathPlanner.cal_cost + get_path (path_planner.py:104-155) for every
 * lattice edge: edge_cost[B][n_s][n_l(to)][n_l(from)]; layer 0 holds the ego
 * edge in [..][to][0] and NaN elsewhere.  status as above (no FOUND/NOT_FOUND:
 * MP_SL_FOUND means "ok"). */
int mp_sl_edge_cost(const mp_sl_batch *batch, const mp_sl_params *params, double *edge_cost,
                    int32_t *status, void *stream);

/* PathPlanner.cal_cost + get_path (path_planner.py:104-155) for a lattice whose
 * nodes carry their own arc length: the reference reads s_map[s][l] per node
 * (:106,109), so a layer need not sit at one s (SLMap never builds such a lattice,
 * sl_map.py:69-78, but a caller may fill s_map by hand).  s_map, l_map
 * [B][n_s][n_l]; the rest as in mp_sl_batch; n_layers may be NULL.  One thread
 * per edge, every sample and every obstacle visited, no limit of 128 samples per
 * edge (up to 2^20; more reports MP_SL_E_CAPACITY): the general, slow companion
 * of mp_sl_edge_cost, with the same edge_cost layout and status; follow it with
 * mp_sl_dp.  Any n_l, n_s; B <= 65535. */
int mp_sl_edge_cost_nodes(int32_t B, int32_t n_s, int32_t n_l, const double *s_map, const double *l_map,
                          const double *ego_sl, const double *ob_s, const double *ob_l, int32_t ob_stride,
                          const int32_t *n_ob, const double *ob_dist, const int32_t *n_layers,
                          const mp_sl_params *params, double *edge_cost, int32_t *status, void *stream);

/* Replaces PathPlanner.find_path/find_path_point (path_planner.py:66-101) over
 * a materialised edge-cost tensor: layer-wise min-relaxation with first-minimum
 * predecessor, abort test, end-node argmin and backtrack.  index_map is
 * required (the backtrack reads it). Any n_l, n_s; n_layers [B] as in
 * mp_sl_batch (may be NULL). */
int mp_sl_dp(int32_t B, int32_t n_s, int32_t n_l, const int32_t *n_layers, const double *edge_cost,
             const mp_sl_params *params, double *cost_map, int32_t *index_map, int32_t *path_idx,
             int32_t *status, int32_t *layers_done, double *best_cost, void *stream);

/* Replaces SpeedPlanner.find_speed + find_speed_point + cal_cost
 * (Planning/DP_Speed/speed_planner.py:57-162).  Scenario b uses an
 * n_t x n_s[b] grid (n_s[b] <= n_s_max) with cell size ds x dt; blocked
 * intervals of time layer i are s_down/s_up[b][i][0..n_reg[b][i]).
 * Limits: n_t <= 64, n_s_max <= 1024, r_max <= 256.
 * Outputs (cost, pre may be NULL):
 *   cost [B][n_t][n_s_max], pre [B][n_t][n_s_max]   row 0 = 0 / -1 (st_map.py:86-91)
 *   path_s_ind [B][n_t]  backtracked s indices, -1 padded; path_len [B] = end_i+1 */
int mp_st_dp(int32_t B, int32_t n_t, int32_t n_s_max, const int32_t *n_s, const double *s_down,
             const double *s_up, const int32_t *n_reg, int32_t r_max, const double *cur_vel,
             double speed_lim, double ds, double dt, double *cost, int32_t *pre,
             int32_t *path_s_ind, int32_t *path_len, void *stream);

/* mp_st_dp with the workgroups launched widest grid first, for batches whose
 * scenarios differ in grid width (a plan's time follows n_s squared); order_ws
 * as in mp_sl_plan_ordered (2*B int32 of device scratch).  No counterpart in
 * the reference; outputs are those of mp_st_dp, bit for bit. */
int mp_st_dp_ordered(int32_t B, int32_t n_t, int32_t n_s_max, const int32_t *n_s, const double *s_down,
             const double *s_up, const int32_t *n_reg, int32_t r_max, const double *cur_vel,
             double speed_lim, double ds, double dt, double *cost, int32_t *pre,
             int32_t *path_s_ind, int32_t *path_len, int32_t *order_ws, void *stream);

/* Replaces STMap.add_obstacle + set_obstate (Planning/DP_Speed/st_map.py:24-60):
 * for each time layer and obstacle, the s of the first and last path point
 * closer than the obstacle radius (distance taken in Frenet (s,l) space).
 * path_s/path_l [B][p_max] (first n_path[b] used); statics st_s/st_l/st_r
 * [B][n_st_max]; dynamics dy_s/dy_l [B][n_dy_max][n_t], dy_r [B][n_dy_max].
 * Outputs s_down/s_up [B][n_t][r_max], n_reg [B][n_t], status [B] MP_ST_*. */
int mp_st_regions(int32_t B, int32_t n_t, const double *path_s, const double *path_l,
                  const int32_t *n_path, int32_t p_max, const double *st_s, const double *st_l,
                  const double *st_r, const int32_t *n_st, int32_t n_st_max, const double *dy_s,
                  const double *dy_l, const double *dy_r, const int32_t *n_dy, int32_t n_dy_max,
                  int32_t r_max, double *s_down, double *s_up, int32_t *n_reg, int32_t *status,
                  void *stream);

/* Replaces CartesianFrenetConverter.__init__ (Model/cartesian_frenet_conversion.py:11-27)
 * and everything cartesian_to_frenet derives from the reference line alone:
 * Curve.__init__ (Model/curve.py:13-38), the projection sample set
 * (curve.py:120-123, calc_point_arr :40-73), get_s per sample (calc_arc_len
 * :93-115, :175-186), heading and curvature per sample (cfc.py:37, :62-66).
 * knots_x/knots_y [B][n_knots_max], first n_knots[b] used; vec [B] end slope;
 * ego_xy [B][2] point whose projection fixes the Frenet origin; d_x = 0.5.
 * Outputs, each [B][m_max] with M[b] valid: tab_x, tab_y, tab_s, tab_theta,
 * tab_kappa; origin [B][4] = (rx_ori, ry_ori, get_s(last knot) -- the length
 * SLMap lays its layers along, sl_map.py:52 --, rx_max); status [B] MP_RL_*.
 * Optional (all five NULL to skip) Frenet->Cartesian table of get_rxy
 * (cfc.py:72-86): abscissae get_arange(rx_ori, rx_max, d_x) with their arc
 * length, ordinate and heading, inv_* [B][m_max], n_inv [B]. */
int mp_refline_tables(int32_t B, const double *knots_x, const double *knots_y,
                      const int32_t *n_knots, int32_t n_knots_max, const double *vec,
                      const double *ego_xy, double d_x, int32_t m_max, double *tab_x,
                      double *tab_y, double *tab_s, double *tab_theta, double *tab_kappa,
                      int32_t *M, double *origin, int32_t *status, double *inv_x, double *inv_s,
                      double *inv_y, double *inv_theta, int32_t *n_inv, void *stream);

/* Replaces CartesianFrenetConverter.frenet_to_cartesian + get_rxy
 * (Model/cartesian_frenet_conversion.py:55-60, 72-86) for n_pts points:
 * the first table abscissa whose arc length reaches s (the line's end if none
 * does), offset by l along the normal. */
int mp_f2c_batch(int32_t n_pts, const double *s, const double *l, const int32_t *pt_scn,
                 int32_t m_max, const double *inv_x, const double *inv_s, const double *inv_y,
                 const double *inv_theta, const int32_t *n_inv, double *x, double *y, void *stream);

/* Replaces Curve.__init__ + Curve.calc_point_arr (Model/curve.py:13-73) for the
 * planner's output splines, one curve per scenario: knots_t/knots_x
 * [B][n_knots_max] (first n_knots[b] used), end slope vec [B], spacing dt.
 *   mode 0: samples get_arange(t0, t0+sum(T), dt)  -- SLMap.path_sampling (sl_map.py:114-126)
 *   mode 1: samples get_arange(t0, t_last, dt)     -- SpeedPlanner.plan (speed_planner.py:44-49)
 *   mode 2: samples query [B][q_max], n_query [B]  -- path_sampling(curve, False, ss)
 * Outputs [B][q_max]: out_t (abscissae), out_x (value), out_dx (first
 * derivative, may be NULL); n_out [B]; status [B] MP_RL_* (MP_RL_E_INDEX where
 * the reference raises IndexError). */
int mp_curve_eval(int32_t B, const double *knots_t, const double *knots_x, const int32_t *n_knots,
                  int32_t n_knots_max, const double *vec, double dt, int32_t mode,
                  const double *query, const int32_t *n_query, int32_t q_max, double *out_t,
                  double *out_x, double *out_dx, int32_t *n_out, int32_t *status, void *stream);

/* Replaces CartesianFrenetConverter.cartesian_to_frenet
 * (Model/cartesian_frenet_conversion.py:32-53) + Curve.projection
 * (Model/curve.py:117-126) for n_pts query points; point i belongs to scenario
 * pt_scn[i] and reads that scenario's tables (as mp_refline_tables wrote them:
 * the nearest-sample search relies on tab_x ascending).  vx/vy/s_dt may all be
 * NULL (order 0). */
int mp_c2f_batch(int32_t n_pts, const double *px, const double *py, const double *vx,
                 const double *vy, const int32_t *pt_scn, int32_t m_max, const double *tab_x,
                 const double *tab_y, const double *tab_s, const double *tab_theta,
                 const double *tab_kappa, const int32_t *M, double *s, double *l, double *s_dt,
                 void *stream);

/* obstacle classes written by mp_scene_ingest */
#define MP_OB_BEHIND 0       /* more than 1 m behind the ego: dropped (robot_map.py:49-50) */
#define MP_OB_STATIC 1       /* static, within ignore_dist: kept                 */
#define MP_OB_STATIC_FAR 2   /* static, out of range: dropped                    */
#define MP_OB_DYNAMIC 3      /* moving, some predicted frame within ignore_dist: kept */
#define MP_OB_DYNAMIC_FAR 4  /* moving, never within range: dropped              */

/* Replaces RobotMap.add_robot / add_ref_line / add_obstacle
 * (Planning/Interface/robot_map.py:25-68, with GeneralConverter.world_to_robot /
 * vel_world_to_robot, Model/general_converter.py:14-26, check_static,
 * Utils/tool.py:41-42, and DynamicObstacle.get_trajectory / cal_min_dist,
 * Model/obstacle.py:46-64) for B world-frame scenes.
 * Inputs: rot [B][4] row-major ego rotation R, trans [B][2] ego position t,
 * ego_vel_w [B][2]; way_xy [B][way_max][2] lane waypoints (first n_way[b]
 * used); obstacles ob_pos / ob_vel [B][ob_max][2], ob_rad [B][ob_max]
 * (first n_ob[b] used); ignore_dist (20), n_frames (11) and dt (0.5) of the
 * prediction.
 * Outputs: knots_x / knots_y [B][way_max] (zero beyond n_way[b]); ego_vel_r
 * [B][2]; kept statics st_pos [B][ob_max][2], st_rad [B][ob_max], n_st [B];
 * kept dynamics dy_pos [B][ob_max][n_frames][2], dy_vel [B][ob_max][2] (may be
 * NULL), dy_rad [B][ob_max], n_dy [B], all in input order, slots beyond the
 * counts untouched; ob_class [B][ob_max] MP_OB_* (may be NULL). */
int mp_scene_ingest(int32_t B, const double *rot, const double *trans, const double *ego_vel_w,
                    const double *way_xy, const int32_t *n_way, int32_t way_max,
                    const double *ob_pos, const double *ob_vel, const double *ob_rad,
                    const int32_t *n_ob, int32_t ob_max, double ignore_dist, int32_t n_frames,
                    double dt, double *knots_x, double *knots_y, double *ego_vel_r, double *st_pos,
                    double *st_rad, int32_t *n_st, double *dy_pos, double *dy_vel, double *dy_rad,
                    int32_t *n_dy, int32_t *ob_class, void *stream);

/* Replaces RobotMap.path_convert / GeneralConverter.robot_to_world
 * (robot_map.py:84-91, general_converter.py:19-22) for n_pts points: point i
 * of scene pt_scn[i] becomes R p + t. */
int mp_robot_to_world(int32_t n_pts, const double *x, const double *y, const int32_t *pt_scn,
                      const double *rot, const double *trans, double *wx, double *wy, void *stream);


/* ---- result record for the gather across GPUs ------------------------------
 * No counterpart in the reference (it plans one scene in one process): ranks
 * plan disjoint blocks of scenarios and exchange one fixed-size record per
 * scenario.  Record of mp_result_record_bytes(n_s, n_t) bytes, little endian:
 *   [0,8) best_cost f64 | [8,10) layers_done u16 | [10] status i8 | [11] st_len u8
 *   | st_path i16[n_t] | path_idx i16[n_s] | zero padding to a multiple of 8
 * mp_pack_results writes records [B][record bytes] from the outputs of
 * mp_sl_plan and mp_st_dp (best_cost may be NULL: 0.0). */
int32_t mp_result_record_bytes(int32_t n_s, int32_t n_t);
int mp_pack_results(int32_t B, int32_t n_s, int32_t n_t, const int32_t *path_idx, const int32_t *status,
                    const int32_t *layers_done, const double *best_cost, const int32_t *st_path,
                    const int32_t *st_len, void *records, void *stream);

/* ---- whole replan, B scenes per call --------------------------------------
 * Replaces what PlannerInterface.run_step computes between coor_trans and the
 * conversion back to the world frame (Planning/Interface/planner_interface.py:
 * 54-68: path_plan :133-138, speed_plan :140-159, sl_map.path_convert) for B
 * robot-frame scenes, every stage on the device and nothing read back by the
 * host in between:
 *   reference-line tables (CartesianFrenetConverter.__init__) -> ego state and
 *   every obstacle frame to Frenet, lane filter, obstacle list, lattice layers
 *   (SLMap.create_frenet_map / add_obstacle, sl_map.py:33-112) -> SL path DP
 *   (PathPlanner.plan) or the centre lane (TrackPathPlanner.plan) -> path spline
 *   sampled every 0.5 m (SLMap.path_sampling) -> ST grid size and blocked
 *   intervals (STMap.__init__ / add_obstacle) -> ST speed DP and the speed
 *   profile every dt/5 (SpeedPlanner.plan) -> path points at the profile's arc
 *   lengths (path_sampling(curve, False, ss)) -> Frenet->Cartesian
 *   (SLMap.path_convert).
 * All intermediate and final arrays live in ONE caller-owned device arena whose
 * layout the library defines: mp_plan_array_info() describes array i (name,
 * element type, shape, byte offset) for the given sizes and
 * mp_plan_arena_bytes() gives the arena size.  Scene b's slice of every array is
 * what one reference object would hold; arrays are indexed by scene whether or
 * not the scene found a path. */

/* per-scene outcome, array "status" */
#define MP_PLAN_OK 1
#define MP_PLAN_NOT_FOUND 0        /* PathPlanner.plan returned (False, [])           */
#define MP_PLAN_E_EMPTY_EDGE (-2)  /* = MP_SL_E_EMPTY_EDGE                            */
#define MP_PLAN_E_CAPACITY (-3)    /* = MP_SL_E_CAPACITY, or more layers than n_s_max */
#define MP_PLAN_E_MAP_TOO_BIG (-4) /* sl_map.py:57-60 ValueError; also a line whose samples
                                      exceed m_max while its knots span more than the map
                                      may be long (its arc length is at least that span)  */
#define MP_PLAN_E_BAD_LINE (-5)    /* reference-line tables failed (MP_RL_E_*)        */
#define MP_PLAN_E_ST_CAPACITY (-6) /* ST grid wider than st_n_s_max, more blocked
                                      intervals than r_max, path longer than p_max or
                                      speed profile longer than q_max                 */
#define MP_PLAN_E_INDEX (-7)       /* the speed profile or the path points at its arc
                                      lengths could not be sampled: the reference raises
                                      IndexError (curve.py:57-65, 82-85)              */

/* Robot-frame scenes (what RobotMap holds after coor_trans, robot_map.py:25-68). */
typedef struct mp_scene_batch {
    const double *knots_x, *knots_y; /* [B][n_knots_max] reference-line waypoints     */
    const int32_t *n_knots;          /* [B]                                           */
    const double *line_vec;          /* [B] end slope of the line (0)                 */
    const double *ego_vel;           /* [B][2]                                        */
    const double *st_pos;            /* [B][n_st_max][2] static obstacles             */
    const double *st_rad;            /* [B][n_st_max]                                 */
    const int32_t *n_st;             /* [B]                                           */
    const double *dy_pos;            /* [B][n_dy_max][n_frames][2] predicted frames   */
    const double *dy_rad;            /* [B][n_dy_max]                                 */
    const int32_t *n_dy;             /* [B]                                           */
} mp_scene_batch;

/* Sizes that fix the arena layout. */
typedef struct mp_plan_dims {
    int32_t B, n_knots_max, n_st_max, n_dy_max;
    int32_t n_frames;   /* frames per moving obstacle (11, Model/obstacle.py:46-53)   */
    int32_t n_l;        /* lateral samples (5, planner_interface.py:100)              */
    int32_t m_max;      /* reference-line samples per scene (line span / 0.5 + 2)     */
    int32_t n_s_max;    /* lattice layers per scene (13 for a 100 m map)              */
    int32_t p_max;      /* path samples per scene (map length / 0.5 + 2)              */
    int32_t n_t;        /* ST time layers = ceil(plan_time / st_dt) + 1 (11)          */
    int32_t st_n_s_max; /* ST cells per time layer (map length / st_ds + 1)           */
    int32_t r_max;      /* blocked intervals per time layer                           */
    int32_t q_max;      /* speed-profile samples ((n_t - 1) * 5 + 2)                  */
    int32_t want_maps;  /* 1: keep SL cost_map / index_map and ST cost / pre tables   */
    int32_t world;      /* 1: the arena also holds the ingested robot-frame scene and the
                           final path in the world frame (mp_plan_world_scenes)       */
} mp_plan_dims;

/* Constants of the reference's planner stack; mp_plan_default_params fills in
 * the values the reference hard-codes. */
typedef struct mp_plan_params {
    double l_width, save_width; /* 3.5, 0.6 (planner_interface.py:99, robot_map.py:35) */
    double layer_gap;           /* 8: SLMap.d_s (sl_map.py:25)                         */
    double max_map_len;         /* 100 (sl_map.py:57)                                  */
    double d_x;                 /* 0.5: converter sample spacing (cfc.py:12)           */
    double path_dt;             /* 0.5: path spline sampling (path_planner.py:53)      */
    mp_sl_params sl;            /* mid_l < 0: int((n_l-1)/2)                           */
    double speed_lim;           /* 25*5/18 (planner_interface.py:151)                  */
    double st_ds, st_dt;        /* 1.0, 0.5 (st_map.py:15-16)                          */
    int32_t path_planner;       /* 0: SL lattice DP, 1: centre lane                    */
    int32_t reserved;
} mp_plan_params;

#define MP_F64 0
#define MP_I32 1

void mp_plan_default_params(mp_plan_params *params);
int32_t mp_plan_array_count(void);
/* Array i of the arena: *name (static string), *dtype MP_F64 / MP_I32, shape[4]
 * (unused trailing dimensions are 1), *offset in bytes.  Returns MP_OK or
 * MP_E_BADARG. */
int mp_plan_array_info(const mp_plan_dims *dims, int32_t i, const char **name, int32_t *dtype,
                       int64_t shape[4], int64_t *offset);
int64_t mp_plan_arena_bytes(const mp_plan_dims *dims);
/* Enqueues the whole chain on `stream`.  arena: device memory of at least
 * mp_plan_arena_bytes(dims) bytes, 256-byte aligned.  Argument errors are found
 * before anything is enqueued; a launch failure part-way (MP_E_LAUNCH) leaves
 * the earlier stages enqueued and the arena's later arrays undefined.  The call
 * can be captured in a HIP graph. */
int mp_plan_scenes(const mp_scene_batch *scenes, const mp_plan_dims *dims, const mp_plan_params *params,
                   void *arena, int64_t arena_bytes, void *stream);

/* Measurement aid, no counterpart in the reference: mp_plan_scenes with a HIP
 * event between its MP_PLAN_N_STAGES stages.  Blocks until the chain has run and
 * writes each stage's duration in milliseconds to stage_ms[MP_PLAN_N_STAGES]
 * (HOST memory); mp_plan_stage_name(i) names stage i.  Results in the arena as
 * from mp_plan_scenes.  Not for graph capture (creates and waits on events). */
#define MP_PLAN_N_STAGES 9
const char *mp_plan_stage_name(int32_t i);
int mp_plan_scenes_timed(const mp_scene_batch *scenes, const mp_plan_dims *dims, const mp_plan_params *params,
                         void *arena, int64_t arena_bytes, void *stream, float *stage_ms);

/* ---- whole replan from world-frame scenes ----------------------------------
 * Replaces PlannerInterface.run_step from its inputs to its outputs
 * (Planning/Interface/planner_interface.py:35-95): coor_trans (:98-131, the ego
 * pose as rotation R and translation t, RobotMap.add_robot / add_ref_line /
 * add_obstacle, robot_map.py:25-68) -> path_plan -> speed_plan ->
 * sl_map.path_convert -> robot_map.path_convert (robot_map.py:84-91), for B scenes
 * in one enqueue: mp_scene_ingest's kernel writes the robot-frame scene into the
 * arena, the chain of mp_plan_scenes runs on it, and its last stage stores every
 * path point in the world frame as well.
 * dims: world = 1; n_knots_max = waypoint slots per scene; n_st_max / n_dy_max =
 * capacity for KEPT static / moving obstacles (a scene that keeps more gets status
 * MP_PLAN_E_CAPACITY); ob_max obstacle slots per scene in the input.
 * Additional arrays of the arena (mp_plan_array_info): w_knots_x, w_knots_y,
 * w_ego_vel, w_st_pos, w_st_rad, w_n_st, w_dy_pos, w_dy_rad, w_n_dy (the scene as
 * mp_scene_batch describes it), w_status, and world_x / world_y [B][q_max], the
 * path of array final_x / final_y as R p + t (0 beyond n_final). */
typedef struct mp_world_batch {
    const double *rot;       /* [B][4] row-major ego rotation R                       */
    const double *trans;     /* [B][2] ego position t                                 */
    const double *ego_vel_w; /* [B][2]                                                */
    const double *way_xy;    /* [B][n_knots_max][2] lane waypoints ahead              */
    const int32_t *n_way;    /* [B]                                                   */
    const double *line_vec;  /* [B] end slope of the line, or NULL (0)                */
    const double *ob_pos;    /* [B][ob_max][2]                                        */
    const double *ob_vel;    /* [B][ob_max][2]                                        */
    const double *ob_rad;    /* [B][ob_max]                                           */
    const int32_t *n_ob;     /* [B]                                                   */
    int32_t ob_max;
    int32_t reserved;
    double ignore_dist;      /* 20 (robot_map.py:19)                                  */
    double frame_dt;         /* 0.5: spacing of the predicted frames (obstacle.py:48) */
} mp_world_batch;
int mp_plan_world_scenes(const mp_world_batch *world, const mp_plan_dims *dims, const mp_plan_params *params,
                         void *arena, int64_t arena_bytes, void *stream);
/* mp_plan_scenes_timed for world-frame scenes (the ingestion is stage 1). */
int mp_plan_world_scenes_timed(const mp_world_batch *world, const mp_plan_dims *dims,
                               const mp_plan_params *params, void *arena, int64_t arena_bytes, void *stream,
                               float *stage_ms);

/* ---- whole replan without a caller-owned arena ------------------------------
 * No counterpart in the reference.  For callers that cannot allocate device
 * memory (code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc): mp_plan_scenes with every intermediate array in a fixed
 * device array of MP_PLAN_STATIC_BYTES that belongs to the library (each GPU the
 * library runs on holds one until the process ends).  The caller passes the
 * scenes and the result arrays only; nothing is allocated or freed, by the
 * library or on its behalf.  Sizes whose arena (mp_plan_arena_bytes) exceeds
 * MP_PLAN_STATIC_BYTES, and want_maps, are refused with MP_E_UNSUPPORTED.  All
 * calls on one GPU share that GPU's array: calls that overlap on one GPU (two
 * streams, two threads) race; calls on one stream are safe back to back.
 * The first call on a GPU looks up that GPU's address of the array
 * (hipGetSymbolAddress, which may load the code object and cannot be captured):
 * make one ordinary call per GPU before capturing the entry point in a HIP graph.
 *   status   [B]         MP_PLAN_*
 *   n_points [B]         valid entries of x / y / speed (0 unless MP_PLAN_OK)
 *   x, y     [B][q_max]  path points in the robot frame (0 beyond n_points)
 *   speed    [B][q_max]  speed profile at those points, m/s (0 beyond n_points) */
#define MP_PLAN_STATIC_BYTES (32u << 20)
typedef struct mp_plan_result {
    int32_t *status, *n_points;
    double *x, *y, *speed;
} mp_plan_result;
int mp_plan_scenes_static(const mp_scene_batch *scenes, const mp_plan_dims *dims, const mp_plan_params *params,
                          const mp_plan_result *result, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MPLATTICE_H */
