"""Batched full planning pipeline: what PlannerInterface.run_step does for one
scene (reference Planning/Interface/planner_interface.py:35-95, without CARLA),
for B scenes at once with every stage on the GPU:

  reference-line tables -> Cartesian->Frenet of the ego and of every obstacle
  frame -> SL lattice + obstacle filtering -> SL path DP -> path spline and
  0.5 m sampling -> ST blocked intervals -> ST speed DP -> speed profile at
  0.1 s -> path points at the profile's arc lengths -> Frenet->Cartesian.

The whole chain is one library call (mp_plan_scenes): the host enqueues a fixed
list of kernels and reads nothing back in between.  Every intermediate and
final array lives in one device arena whose layout the library reports; the
result dict holds views into it, all indexed by scene.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib, batch as mb
from ._lib import PlanDims, PlanParams, SceneBatch, check, lib

PIPE_OK, PIPE_NOT_FOUND = _lib.MP_PLAN_OK, _lib.MP_PLAN_NOT_FOUND
PIPE_MAP_TOO_BIG, PIPE_BAD_LINE, PIPE_ST_CAPACITY = (_lib.MP_PLAN_E_MAP_TOO_BIG, _lib.MP_PLAN_E_BAD_LINE,
                                                    _lib.MP_PLAN_E_ST_CAPACITY)
PIPE_INDEX = _lib.MP_PLAN_E_INDEX  # the reference raises IndexError while sampling the result
PIPE_CAPACITY = _lib.MP_PLAN_E_CAPACITY
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


def default_params(**over):
    """mp_plan_params with the reference's constants; keyword overrides by field name
    (fields of the nested SL parameters as sl_<name>)."""
    p = PlanParams()
    lib().mp_plan_default_params(C.byref(p))
    for k, v in over.items():
        if k.startswith("sl_"):
            setattr(p.sl, k[3:], v)
        else:
            setattr(p, k, v)
    return p


class PlanArena:
    """One device allocation holding every array of a mp_plan_scenes call, with a
    torch view per array (`arena.arrays[name]`).  Reusable across calls of the
    same sizes."""

    _TYPES = {_lib.MP_F64: torch.float64, _lib.MP_I32: torch.int32}

    def __init__(self, dims, device=None):
        self.dims = dims
        dev = device or mb._dev()
        n = lib().mp_plan_arena_bytes(C.byref(dims))
        if n < 0:
            raise _lib.MpError("mp_plan_arena_bytes: " + lib().mp_last_error().decode())
        self.nbytes = int(n)
        self.buf = torch.empty(self.nbytes + 256, dtype=torch.uint8, device=dev)
        shift = (-self.buf.data_ptr()) % 256
        self.base = self.buf[shift:shift + self.nbytes]
        self.arrays = {}
        name, dtype, off = C.c_char_p(), C.c_int32(), C.c_int64()
        shape = (C.c_int64 * 4)()
        for i in range(lib().mp_plan_array_count()):
            check(lib().mp_plan_array_info(C.byref(dims), i, C.byref(name), C.byref(dtype), shape, C.byref(off)),
                  "mp_plan_array_info")
            dt = self._TYPES[dtype.value]
            shp = list(shape)
            while len(shp) > 1 and shp[-1] == 1:
                shp.pop()
            count = int(np.prod(list(shape)))
            if count == 0:
                self.arrays[name.value.decode()] = None
                continue
            nb = count * (8 if dt == torch.float64 else 4)
            self.arrays[name.value.decode()] = self.base[off.value:off.value + nb].view(dt).view(shp)


M_MAX_MAP = 232   # reference-line samples every 0.5 m for a 100 m map, with a margin


def plan_dims(B, n_knots_max, n_st_max, n_dy_max, n_l=5, m_max=232, plan_time=5, params=None, want_maps=False,
              n_frames=N_FRAMES, r_max=None, world=False):
    """Arena sizes for scenes of the given shape: capacities follow from the map length
    limit (100 m) and the sampling constants.  world: the arena also holds the ingested
    scene and the world-frame path (plan_world_scenes)."""
    p = params or default_params()
    n_t = int(math.ceil(1.0 * plan_time / p.st_dt)) + 1            # st_map.py:18
    n_s_max = int(p.max_map_len / p.layer_gap) + 2
    p_max = int(p.max_map_len / p.path_dt) + 8
    st_n_s_max = int(p.max_map_len / p.st_ds) + 2
    q_max = (n_t - 1) * 5 + 4
    if r_max is None:
        r_max = max(1, min(n_st_max + n_dy_max, 256))
    return PlanDims(B, n_knots_max, n_st_max, n_dy_max, n_frames, n_l, m_max, n_s_max, p_max, n_t, st_n_s_max,
                    r_max, q_max, int(bool(want_maps)), int(bool(world)))


def plan_scenes(scenes, dims, params, arena=None):
    """Enqueues one mp_plan_scenes call.  scenes: dict of device tensors named as the
    fields of mp_scene_batch.  Returns the arena (its `arrays` are the results)."""
    arena = arena or PlanArena(dims)
    sb = SceneBatch(*[mb._p(scenes.get(n)) for n, _ in SceneBatch._fields_])
    check(lib().mp_plan_scenes(C.byref(sb), C.byref(dims), C.byref(params), C.c_void_p(arena.base.data_ptr()),
                               arena.nbytes, mb._stream()), "mp_plan_scenes")
    return arena


def plan_scenes_timed(scenes, dims, params, arena=None):
    """mp_plan_scenes with a HIP event between its stages (measurement aid; blocks until
    the chain has run).  Returns (arena, [(stage name, milliseconds), ...])."""
    arena = arena or PlanArena(dims)
    sb = SceneBatch(*[mb._p(scenes.get(n)) for n, _ in SceneBatch._fields_])
    ms = (C.c_float * _lib.MP_PLAN_N_STAGES)()
    check(lib().mp_plan_scenes_timed(C.byref(sb), C.byref(dims), C.byref(params), C.c_void_p(arena.base.data_ptr()),
                                     arena.nbytes, mb._stream(), ms), "mp_plan_scenes_timed")
    return arena, [(lib().mp_plan_stage_name(i).decode(), float(ms[i])) for i in range(_lib.MP_PLAN_N_STAGES)]


def _world_batch(world, ignore_dist, frame_dt):
    """mp_world_batch from a dict of device tensors named as its pointer fields."""
    ob_max = int(world["ob_rad"].shape[1]) if world.get("ob_rad") is not None and world["ob_rad"].dim() == 2 else 0
    ptrs = [mb._p(world.get(n)) for n, t in _lib.WorldBatch._fields_ if t is _lib._vp]
    return _lib.WorldBatch(*ptrs, ob_max, 0, float(ignore_dist), float(frame_dt))


def plan_world_scenes(world, dims, params, arena=None, ignore_dist=20.0, frame_dt=0.5):
    """Enqueues one mp_plan_world_scenes call: world-frame scenes in, world-frame paths out
    (PlannerInterface.run_step, planner_interface.py:35-95, for B scenes).  world: dict of
    device tensors rot [B, 4], trans [B, 2], ego_vel_w [B, 2], way_xy [B, W, 2], n_way [B],
    ob_pos / ob_vel [B, O, 2], ob_rad [B, O], n_ob [B] and optionally line_vec [B].  dims
    must have been made with world=True.  Returns the arena."""
    arena = arena or PlanArena(dims)
    wb = _world_batch(world, ignore_dist, frame_dt)
    check(lib().mp_plan_world_scenes(C.byref(wb), C.byref(dims), C.byref(params), C.c_void_p(arena.base.data_ptr()),
                                     arena.nbytes, mb._stream()), "mp_plan_world_scenes")
    return arena


def plan_world_scenes_timed(world, dims, params, arena=None, ignore_dist=20.0, frame_dt=0.5):
    """plan_world_scenes with a HIP event between its stages (measurement aid; blocks)."""
    arena = arena or PlanArena(dims)
    wb = _world_batch(world, ignore_dist, frame_dt)
    ms = (C.c_float * _lib.MP_PLAN_N_STAGES)()
    check(lib().mp_plan_world_scenes_timed(C.byref(wb), C.byref(dims), C.byref(params),
                                           C.c_void_p(arena.base.data_ptr()), arena.nbytes, mb._stream(), ms),
          "mp_plan_world_scenes_timed")
    return arena, [(lib().mp_plan_stage_name(i).decode(), float(ms[i])) for i in range(_lib.MP_PLAN_N_STAGES)]


def plan_scenes_static(scenes, dims, params, out=None):
    """Enqueues one mp_plan_scenes_static call: the same chain with every intermediate
    array in the library's own fixed device array, for callers that must not allocate
    (the tensors in `out` -- status, n_points [B] int32; x, y, speed [B, q_max] float64 --
    are the caller's; they are created here only when not passed in).  Raises MpError
    when the sizes need more scratch than the library holds."""
    B, q = dims.B, dims.q_max
    if out is None:
        dev = mb._dev()
        out = dict(status=torch.empty(B, dtype=torch.int32, device=dev),
                   n_points=torch.empty(B, dtype=torch.int32, device=dev),
                   x=torch.empty((B, q), dtype=torch.float64, device=dev),
                   y=torch.empty((B, q), dtype=torch.float64, device=dev),
                   speed=torch.empty((B, q), dtype=torch.float64, device=dev))
    for k, shp, dt in (("status", (B,), torch.int32), ("n_points", (B,), torch.int32), ("x", (B, q), torch.float64),
                       ("y", (B, q), torch.float64), ("speed", (B, q), torch.float64)):
        t = out[k]
        if tuple(t.shape) != shp or t.dtype != dt or not t.is_contiguous() or not t.is_cuda:
            raise ValueError("plan_scenes_static: out[%r] must be a contiguous CUDA %s tensor of shape %s" % (k, dt, shp))
    sb = SceneBatch(*[mb._p(scenes.get(n)) for n, _ in SceneBatch._fields_])
    res = _lib.PlanResult(*[mb._p(out[k]) for k, _ in _lib.PlanResult._fields_])
    check(lib().mp_plan_scenes_static(C.byref(sb), C.byref(dims), C.byref(params), C.byref(res), mb._stream()),
          "mp_plan_scenes_static")
    return out


def plan_batch(knots_x, knots_y, n_knots, ego_vel, st_pos, st_rad, n_st, dy_pos, dy_rad, n_dy,
               line_vec=None, l_width=3.5, save_width=0.6, n_l=5, speed_lim=25.0 * 5.0 / 18.0, plan_time=5,
               want_maps=False, path_planner="dp", m_max=None, arena=None):
    """Plans B scenes given in the robot frame (ego at the origin).

    knots_x / knots_y [B, nk]; n_knots [B]; ego_vel [B, 2]; statics st_pos [B, Ns, 2],
    st_rad [B, Ns], n_st [B]; dynamics dy_pos [B, Nd, 11, 2] (predicted frames),
    dy_rad [B, Nd], n_dy [B].  path_planner "dp" runs the SL lattice DP (PathPlanner),
    "track" follows the lattice's centre lane (TrackPathPlanner, the planner the
    reference's PlannerInterface.path_plan instantiates, planner_interface.py:133-138).
    m_max: reference-line samples per scene (default: from the knots' x span; for knots
    already on the device the samples of the longest line a map may hold, no read back).
    Returns a dict of tensors indexed by scene; `status` [B] is PIPE_OK,
    PIPE_NOT_FOUND or a negative PIPE_* code; the dict also carries `arena`
    (reusable through the `arena` argument for batches of the same shape)."""
    if path_planner not in ("dp", "track"):
        raise ValueError("path_planner must be 'dp' or 'track'")
    dev = mb._dev()
    f64 = dict(dtype=torch.float64, device=dev)
    if m_max is None:
        if isinstance(knots_x, torch.Tensor) and knots_x.is_cuda:
            # no device->host read: samples for the longest line a map may hold (100 m) and a
            # margin; a longer line is reported as PIPE_MAP_TOO_BIG, as the reference raises
            m_max = M_MAX_MAP
        else:
            kx = torch.as_tensor(np.asarray(knots_x))
            span = float((kx.max(dim=1).values - kx.min(dim=1).values).max()) if kx.numel() else 0.0
            m_max = int(span / 0.5) + 8
    knots_x, knots_y = mb.to_dev(knots_x), mb.to_dev(knots_y)
    B = knots_x.shape[0]

    def rows(x, *tail):
        x = mb.to_dev(x)
        return x.reshape((B, -1) + tail) if x.numel() else x.new_zeros((B, 0) + tail)

    scenes = dict(knots_x=knots_x, knots_y=knots_y, n_knots=mb.to_dev(n_knots, torch.int32),
                  ego_vel=mb.to_dev(ego_vel).reshape(B, 2),
                  st_pos=rows(st_pos, 2), st_rad=rows(st_rad), n_st=mb.to_dev(n_st, torch.int32),
                  dy_pos=rows(dy_pos, N_FRAMES, 2), dy_rad=rows(dy_rad), n_dy=mb.to_dev(n_dy, torch.int32),
                  line_vec=torch.zeros(B, **f64) if line_vec is None else mb.to_dev(line_vec))
    Ns, Nd = scenes["st_pos"].shape[1], scenes["dy_pos"].shape[1]
    params = default_params(l_width=l_width, save_width=save_width, speed_lim=speed_lim,
                            path_planner=0 if path_planner == "dp" else 1)
    dims = plan_dims(B, knots_x.shape[1], Ns, Nd, n_l=n_l, m_max=m_max, plan_time=plan_time, params=params,
                     want_maps=want_maps)
    if arena is not None and bytes(arena.dims) != bytes(dims):
        raise ValueError("arena was laid out for other sizes")
    arena = plan_scenes(scenes, dims, params, arena)
    res = dict(arena.arrays)
    res.update(arena=arena, dims=dims, scenes=scenes, n_s=res["n_layers"],
               tables=_tables_view(res, B, m_max))
    return res


def found_ids(res):
    """Scenes with a complete plan (device->host read)."""
    return torch.nonzero(res["status"] == PIPE_OK)[:, 0]


def _tables_view(arr, B, m_max):
    """The reference-line tables of the arena as a batch.RefLineTables (for the point
    kernels mb.c2f / mb.f2c)."""
    t = mb.RefLineTables.__new__(mb.RefLineTables)
    t.B, t.m_max = B, m_max
    for k in ("tab_x", "tab_y", "tab_s", "tab_theta", "tab_kappa", "origin", "inv_x", "inv_s", "inv_y", "inv_theta",
              "n_inv"):
        setattr(t, k, arr[k])
    t.M, t.status = arr["tab_m"], arr["rl_status"]
    return t


def plan_world_batch(rot, trans, ego_vel_w, way_xy, n_way, ob_pos, ob_vel, ob_rad, n_ob, ignore_dist=20.0,
                     line_vec=None, l_width=3.5, save_width=0.6, n_l=5, speed_lim=25.0 * 5.0 / 18.0, plan_time=5,
                     want_maps=False, path_planner="dp", m_max=None, arena=None, n_st_max=None, n_dy_max=None):
    """World-frame scenes end to end in ONE library call (mp_plan_world_scenes): what
    PlannerInterface.run_step computes from coor_trans to the two path_convert calls
    (planner_interface.py:35-95) for B scenes.

    rot [B, 2, 2] ego rotation, trans [B, 2] ego position, ego_vel_w [B, 2]; way_xy
    [B, W, 2] lane waypoints ahead, n_way [B]; obstacles ob_pos / ob_vel [B, O, 2],
    ob_rad [B, O], n_ob [B], all in the world frame.  n_st_max / n_dy_max: slots for the
    kept static / moving obstacles of a scene (default O each; a scene that keeps more
    reports PIPE_CAPACITY).  Returns plan_batch's dict plus `scene` (the ingested
    robot-frame scene, views into the arena) and `world_x` / `world_y` [B, q_max], the final
    path of every scene in the world frame (n_final[b] valid points, 0 behind them)."""
    if path_planner not in ("dp", "track"):
        raise ValueError("path_planner must be 'dp' or 'track'")
    rot, trans = mb.to_dev(rot).reshape(-1, 4).contiguous(), mb.to_dev(trans).reshape(-1, 2).contiguous()
    B = rot.shape[0]
    ego_vel_w = mb.to_dev(ego_vel_w).reshape(B, 2).contiguous()
    way_xy = mb.to_dev(way_xy).reshape(B, -1, 2).contiguous()
    W = way_xy.shape[1]
    n_way, n_ob = mb.to_dev(n_way, torch.int32), mb.to_dev(n_ob, torch.int32)
    def rows(x, *tail):
        x = mb.to_dev(x)
        return x.reshape((B, -1) + tail).contiguous() if x.numel() else x.new_zeros((B, 0) + tail)

    ob_pos, ob_vel, ob_rad = rows(ob_pos, 2), rows(ob_vel, 2), rows(ob_rad)
    O = ob_pos.shape[1]
    assert ob_vel.shape[1] == O and ob_rad.shape[1] == O
    if m_max is None:
        # samples of the reference line: from the waypoints' abscissae in the robot frame,
        # R^T (p - t) (one device->host read)
        d = way_xy - trans[:, None, :]
        kx = rot[:, 0:1] * d[:, :, 0] + rot[:, 2:3] * d[:, :, 1]
        used = torch.arange(W, device=kx.device)[None, :] < n_way[:, None]
        big = torch.finfo(kx.dtype).max
        span = (torch.where(used, kx, -big).max(dim=1).values - torch.where(used, kx, big).min(dim=1).values)
        m_max = int(float(span.clamp(min=0).max()) / 0.5) + 8 if kx.numel() else 8
    world = dict(rot=rot, trans=trans, ego_vel_w=ego_vel_w, way_xy=way_xy, n_way=n_way, n_ob=n_ob,
                 line_vec=None if line_vec is None else mb.to_dev(line_vec))
    if O:
        world.update(ob_pos=ob_pos, ob_vel=ob_vel, ob_rad=ob_rad)
    Ns = O if n_st_max is None else int(n_st_max)
    Nd = O if n_dy_max is None else int(n_dy_max)
    params = default_params(l_width=l_width, save_width=save_width, speed_lim=speed_lim,
                            path_planner=0 if path_planner == "dp" else 1)
    dims = plan_dims(B, W, Ns, Nd, n_l=n_l, m_max=m_max, plan_time=plan_time, params=params, want_maps=want_maps,
                     world=True)
    if arena is not None and bytes(arena.dims) != bytes(dims):
        raise ValueError("arena was laid out for other sizes")
    arena = plan_world_scenes(world, dims, params, arena, ignore_dist=ignore_dist)
    res = dict(arena.arrays)
    scene = dict(knots_x=res["w_knots_x"], knots_y=res["w_knots_y"], ego_vel=res["w_ego_vel"], st_pos=res["w_st_pos"],
                 st_rad=res["w_st_rad"], n_st=res["w_n_st"], dy_pos=res["w_dy_pos"], dy_rad=res["w_dy_rad"],
                 n_dy=res["w_n_dy"])
    res.update(arena=arena, dims=dims, world=world, scene=scene, scenes=scene, n_s=res["n_layers"],
               tables=_tables_view(res, B, m_max))
    return res
