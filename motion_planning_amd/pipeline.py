"""Batched full planning pipeline: what PlannerInterface.run_step does for one
scene (reference Planning/Interface/planner_interface.py:35-95, without CARLA),
for B scenes at once with every stage on the GPU:

  reference-line tables -> Cartesian->Frenet of the ego and of every obstacle
  frame -> SL lattice + obstacle filtering -> SL path DP -> path spline and
  0.5 m sampling -> ST blocked intervals -> ST speed DP -> speed profile at
  0.1 s -> path points at the profile's arc lengths -> Frenet->Cartesian.

Host work is limited to sizing decisions that need a scalar per scene (layer
count, ST grid width) and to grouping scenes by layer count.
"""
import math

import numpy as np
import torch

from . import _lib, batch as mb

PIPE_OK, PIPE_NOT_FOUND, PIPE_MAP_TOO_BIG, PIPE_BAD_LINE, PIPE_ST_CAPACITY = 1, 0, -4, -5, -6
N_FRAMES = 11  # dynamic-obstacle frames: t = 0, 0.5, ..., 5.0 (Model/obstacle.py:46-53)


def layer_arc_lengths(end_s, d_s=8.0):
    """SLMap layer placement (sl_map.py:61-66): d_s, 2 d_s, ... then end_s as the
    last layer once the next regular layer would come within d_s/2 of it."""
    out = []
    for e in end_s:
        lay, s = [], d_s
        while True:
            lay.append(s)
            s += d_s
            if s + d_s / 2 >= e:
                lay.append(float(e))
                break
        out.append(lay)
    return out


def plan_batch(knots_x, knots_y, n_knots, ego_vel, st_pos, st_rad, n_st, dy_pos, dy_rad, n_dy,
               line_vec=None, l_width=3.5, save_width=0.6, n_l=5, speed_lim=25.0 * 5.0 / 18.0, plan_time=5,
               want_maps=False, path_planner="dp"):
    """Plans B scenes given in the robot frame (ego at the origin).

    knots_x / knots_y [B, nk]; n_knots [B]; ego_vel [B, 2]; statics st_pos [B, Ns, 2],
    st_rad [B, Ns], n_st [B]; dynamics dy_pos [B, Nd, 11, 2] (predicted frames),
    dy_rad [B, Nd], n_dy [B].  path_planner "dp" runs the SL lattice DP (PathPlanner),
    "track" follows the lattice's centre lane (TrackPathPlanner, the planner the
    reference's PlannerInterface.path_plan instantiates, planner_interface.py:133-138).
    Returns a dict of tensors / lists; `status` [B] is PIPE_OK, PIPE_NOT_FOUND or a
    negative PIPE_* code."""
    if path_planner not in ("dp", "track"):
        raise ValueError("path_planner must be 'dp' or 'track'")
    dev = mb._dev()
    f64 = dict(dtype=torch.float64, device=dev)
    knots_x, knots_y = mb.to_dev(knots_x), mb.to_dev(knots_y)
    B = knots_x.shape[0]
    n_knots = mb.to_dev(n_knots, torch.int32)
    ego_vel = mb.to_dev(ego_vel)
    st_pos = mb.to_dev(st_pos).reshape(B, -1, 2)
    st_rad = mb.to_dev(st_rad).reshape(B, -1)
    n_st = mb.to_dev(n_st, torch.int32)
    Ns = st_pos.shape[1]
    dy_pos = mb.to_dev(dy_pos).reshape(B, -1, N_FRAMES, 2)
    dy_rad = mb.to_dev(dy_rad).reshape(B, -1)
    n_dy = mb.to_dev(n_dy, torch.int32)
    Nd = dy_pos.shape[1]
    line_vec = torch.zeros(B, **f64) if line_vec is None else mb.to_dev(line_vec)
    res = dict(status=torch.full((B,), PIPE_OK, dtype=torch.int32, device=dev))

    # 1. Frenet frame of every scene; 2. ego and obstacle frames into it
    tables = mb.refline_tables(knots_x, knots_y, n_knots, line_vec, torch.zeros((B, 2), **f64))
    scn = torch.arange(B, dtype=torch.int32, device=dev)
    ego_s, ego_l, ego_sdt = mb.c2f(tables, torch.zeros(B, **f64), torch.zeros(B, **f64), scn, ego_vel[:, 0],
                                   ego_vel[:, 1])
    n_pts = Ns + Nd * N_FRAMES
    if n_pts:
        pts = torch.cat([st_pos, dy_pos.reshape(B, Nd * N_FRAMES, 2)], dim=1)
        ob_s, ob_l = mb.c2f(tables, pts[:, :, 0].reshape(-1), pts[:, :, 1].reshape(-1),
                            scn.repeat_interleave(n_pts))
        ob_s, ob_l = ob_s.view(B, n_pts), ob_l.view(B, n_pts)
    else:
        ob_s = ob_l = torch.zeros((B, 0), **f64)
    rad = torch.cat([st_rad, dy_rad.repeat_interleave(N_FRAMES, dim=1)], dim=1)
    slot = torch.arange(n_pts, device=dev)[None]
    valid = torch.where(slot < Ns, slot < n_st[:, None], (slot - Ns) // N_FRAMES < n_dy[:, None])
    res.update(tables=tables, ego_sl=torch.stack([ego_s, ego_l], 1), robot_vel=ego_sdt,
               st_traj=torch.stack([ob_s[:, :Ns], ob_l[:, :Ns]], 2),
               dy_traj=torch.stack([ob_s[:, Ns:].view(B, Nd, N_FRAMES), ob_l[:, Ns:].view(B, Nd, N_FRAMES)], 3))

    # 3. lattice layers (needs one scalar per scene on the host)
    tab_status = tables.status.cpu().numpy()
    end_s = tables.origin[:, 2].cpu().numpy()
    status = np.full(B, PIPE_OK, np.int32)
    status[tab_status != _lib.MP_RL_OK] = PIPE_BAD_LINE
    status[(status == PIPE_OK) & (end_s > 100)] = PIPE_MAP_TOO_BIG   # sl_map.py:57-60
    ok = np.flatnonzero(status == PIPE_OK)
    layers = layer_arc_lengths(end_s[ok])
    n_s_of = np.zeros(B, np.int64)
    n_s_of[ok] = [len(v) for v in layers]
    half = (l_width - save_width) / 2.0
    l_row = torch.as_tensor(np.linspace(0.0 - half, 0.0 + half, num=n_l), **f64)

    # 4. obstacles that can touch the lattice: |l| - r < l_width (sl_map.py:111-112),
    #    kept slots moved to the front in their original order; one radius per scene
    keep = valid & (ob_l.abs() - rad < l_width)
    order = torch.argsort((~keep).to(torch.int8), dim=1, stable=True)
    kept_s, kept_l = torch.gather(ob_s, 1, order), torch.gather(ob_l, 1, order)
    n_keep = keep.sum(dim=1).to(torch.int32)
    ob_dist = torch.where(keep, rad, torch.zeros_like(rad)).max(dim=1).values if n_pts else torch.zeros(B, **f64)
    res.update(n_s=torch.as_tensor(n_s_of, device=dev), ob_s=kept_s, ob_l=kept_l, n_ob=n_keep, ob_dist=ob_dist)

    # 5. SL path DP per group of equal layer count
    ns_max = int(n_s_of.max()) if len(ok) else 1
    s_layers = torch.zeros((B, ns_max), **f64)
    path_idx = torch.full((B, ns_max), -1, dtype=torch.int32, device=dev)
    sl_status = torch.zeros(B, dtype=torch.int32, device=dev)
    best_cost = torch.full((B,), float("nan"), **f64)
    cost_maps, index_maps = {}, {}
    for n_s in sorted(set(n_s_of[ok].tolist())):
        ids = ok[n_s_of[ok] == n_s]
        sel = torch.as_tensor(ids, device=dev)
        lay = torch.as_tensor(np.array([layers[i] for i in np.flatnonzero(n_s_of[ok] == n_s)]), **f64)
        s_layers[sel, :n_s] = lay
        if path_planner == "track":
            # every layer's centre node, always "found" (track_path_planner.py:25-35)
            path_idx[sel, :n_s] = int((n_l - 1) / 2)
            sl_status[sel] = _lib.MP_SL_FOUND
            continue
        sb = mb.SLBatch(lay, l_row[None].expand(len(ids), n_l).contiguous(), res["ego_sl"][sel], kept_s[sel],
                        kept_l[sel], n_keep[sel], ob_dist[sel])
        out = mb.sl_plan(sb, mb.sl_params(n_l=n_l), want_maps=want_maps)
        path_idx[sel, :n_s] = out.path_idx
        sl_status[sel] = out.status
        best_cost[sel] = out.best_cost
        if want_maps:
            for k, i in enumerate(ids):
                cost_maps[int(i)] = out.cost_map[k]
                index_maps[int(i)] = out.index_map[k]
    sl_np = sl_status.cpu().numpy()
    status[(status == PIPE_OK) & (sl_np == _lib.MP_SL_NOT_FOUND)] = PIPE_NOT_FOUND
    status[(status == PIPE_OK) & (sl_np < 0)] = sl_np[(status == PIPE_OK) & (sl_np < 0)]
    res.update(s_layers=s_layers, path_idx=path_idx, best_cost=best_cost, cost_map=cost_maps, index_map=index_maps)

    found = np.flatnonzero(status == PIPE_OK)
    res["status"] = torch.as_tensor(status, device=dev)
    if len(found) == 0:
        return res
    fs = torch.as_tensor(found, device=dev)
    F = len(found)
    fscn = torch.arange(F, dtype=torch.int32, device=dev)

    # 6. path spline through the ego point and the chosen nodes, sampled every 0.5 m
    nk = torch.as_tensor(n_s_of[found] + 1, dtype=torch.int32, device=dev)
    idx = path_idx[fs].clamp(min=0).long()
    knots_t = torch.cat([res["ego_sl"][fs, :1], s_layers[fs]], dim=1)
    knots_v = torch.cat([res["ego_sl"][fs, 1:], l_row[idx]], dim=1)
    p_max = int(float(knots_t.max()) / 0.5) + 8
    zero = torch.zeros(F, **f64)
    path = mb.curve_eval(knots_t, knots_v, nk, zero, 0.5, mb.CURVE_ARC, p_max)
    n_path = path["n"]
    last = (n_path.long() - 1).clamp(min=0)
    s_end = path["t"].gather(1, last[:, None])[:, 0]

    # 7. ST grid and blocked intervals (all statics, frame i of each dynamic obstacle)
    st_n_s = (torch.floor(s_end / 1.0) + 1).to(torch.int32)       # st_map.py:17
    n_t = int(math.ceil(1.0 * plan_time / 0.5)) + 1
    r_max = max(1, min(Ns + Nd, 256))
    down, up, n_reg, reg_status = mb.st_regions(
        n_t, path["t"], path["x"], n_path, res["st_traj"][fs][:, :, 0], res["st_traj"][fs][:, :, 1], st_rad[fs],
        n_st[fs], res["dy_traj"][fs][:, :, :n_t, 0], res["dy_traj"][fs][:, :, :n_t, 1], dy_rad[fs], n_dy[fs], r_max)

    # 8. ST speed DP; 9. speed profile every 0.1 s
    st = mb.st_dp(st_n_s, down, up, n_reg, ego_sdt[fs], speed_lim, want_tables=want_maps)
    plen = st["path_len"]
    tk = 0.5 * torch.arange(n_t, **f64)[None].expand(F, n_t).contiguous()
    sk = st["path_s_ind"].clamp(min=0).to(torch.float64) * 1.0
    q_max = int(round(0.5 * (n_t - 1) / 0.1)) + 4
    prof = mb.curve_eval(tk, sk, plen, ego_sdt[fs], 0.1, mb.CURVE_KNOTS, q_max, want_dx=True)

    # 10. path points at the profile's arc lengths, then back to the robot frame
    fin = mb.curve_eval(knots_t, knots_v, nk, zero, 0.5, mb.CURVE_QUERY, q_max, query=prof["x"], n_query=prof["n"])
    xs, ys = mb.f2c(tables_select(tables, fs), fin["t"].reshape(-1), fin["x"].reshape(-1),
                    fscn.repeat_interleave(q_max))
    bad_st = (reg_status != _lib.MP_ST_OK).cpu().numpy()
    status[found[bad_st]] = PIPE_ST_CAPACITY
    res["status"] = torch.as_tensor(status, device=dev)
    res.update(found_ids=fs, path_s=path["t"], path_l=path["x"], n_path=n_path, st_n_s=st_n_s, st_down=down,
               st_up=up, st_n_reg=n_reg, st_path_s_ind=st["path_s_ind"], st_path_len=plen, st_cost=st["cost"],
               st_pre=st["pre"], ss=prof["x"], vv=prof["dx"], n_prof=prof["n"], final_s=fin["t"], final_l=fin["x"],
               final_x=xs.view(F, q_max), final_y=ys.view(F, q_max))
    return res


def tables_select(t, sel):
    """View of the reference-line tables restricted to scenes `sel` (for point
    kernels whose scene index counts within the selection)."""
    out = mb.RefLineTables.__new__(mb.RefLineTables)
    out.B, out.m_max = len(sel), t.m_max
    for k in ("tab_x", "tab_y", "tab_s", "tab_theta", "tab_kappa", "M", "origin", "status", "inv_x", "inv_s",
              "inv_y", "inv_theta", "n_inv"):
        setattr(out, k, getattr(t, k)[sel].contiguous())
    return out


def plan_world_batch(rot, trans, ego_vel_w, way_xy, n_way, ob_pos, ob_vel, ob_rad, n_ob, ignore_dist=20.0,
                     **plan_args):
    """World-frame scenes end to end: what PlannerInterface.run_step computes between
    coor_trans and the two path_convert calls (planner_interface.py:54-68) for B scenes.

    rot [B, 2, 2] ego rotation, trans [B, 2] ego position, ego_vel_w [B, 2]; way_xy
    [B, W, 2] lane waypoints ahead, n_way [B]; obstacles ob_pos / ob_vel [B, O, 2],
    ob_rad [B, O], n_ob [B], all in the world frame.  Returns plan_batch's dict plus
    `scene` (the ingested robot-frame scene) and `world_x` / `world_y` [F, q_max], the
    final path of the found scenes in the world frame."""
    sc = mb.scene_ingest(rot, trans, ego_vel_w, way_xy, n_way, ob_pos, ob_vel, ob_rad, n_ob, ignore_dist)
    res = plan_batch(sc["knots_x"], sc["knots_y"], n_way, sc["ego_vel"], sc["st_pos"], sc["st_rad"], sc["n_st"],
                     sc["dy_pos"], sc["dy_rad"], sc["n_dy"], **plan_args)
    res["scene"] = sc
    if "final_x" in res:
        fs = res["found_ids"]
        F, q_max = res["final_x"].shape
        rot_d, trans_d = mb.to_dev(rot).reshape(-1, 4)[fs], mb.to_dev(trans).reshape(-1, 2)[fs]
        scn = torch.arange(F, dtype=torch.int32, device=fs.device).repeat_interleave(q_max)
        wx, wy = mb.robot_to_world(rot_d, trans_d, res["final_x"], res["final_y"], scn)
        res.update(world_x=wx.view(F, q_max), world_y=wy.view(F, q_max))
    return res
