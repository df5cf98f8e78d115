"""ctypes binding of libmplattice.so (the HIP kernels behind a C ABI).

The library is built in-tree by `make -C motion_planning_amd/csrc` (or
__graft_entry__.build()).  There is no CPU fallback: if the shared object is
missing or a GPU entry point is called without a GPU, the call raises.
Signatures follow include/mplattice.h.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MP_LIB_PATH: another build of the library (A/B timing of two builds in one session)
LIB_PATH = os.environ.get("MP_LIB_PATH") or os.path.join(_HERE, "libmplattice.so")

MP_OK = 0
MP_VERSION = 124   # MP_VERSION of include/mplattice.h this binding was written for
MP_SL_FOUND, MP_SL_NOT_FOUND, MP_SL_E_EMPTY_EDGE, MP_SL_E_CAPACITY, MP_SL_SKIPPED = 1, 0, -2, -3, -4
MP_ST_OK, MP_ST_E_CAPACITY = 1, -3
MP_RL_OK, MP_RL_E_INDEX, MP_RL_E_CAPACITY = 1, -1, -3
MP_OB_BEHIND, MP_OB_STATIC, MP_OB_STATIC_FAR, MP_OB_DYNAMIC, MP_OB_DYNAMIC_FAR = 0, 1, 2, 3, 4
MP_PLAN_OK, MP_PLAN_NOT_FOUND, MP_PLAN_E_EMPTY_EDGE, MP_PLAN_E_CAPACITY = 1, 0, -2, -3
MP_PLAN_E_MAP_TOO_BIG, MP_PLAN_E_BAD_LINE, MP_PLAN_E_ST_CAPACITY, MP_PLAN_E_INDEX = -4, -5, -6, -7
MP_F64, MP_I32 = 0, 1

_vp = C.c_void_p


class SlParams(C.Structure):
    _fields_ = [("d_s", C.c_double), ("w_d", C.c_double), ("no_path_cost", C.c_double),
                ("early_exit", C.c_int32), ("mid_l", C.c_int32)]


class SlBatch(C.Structure):
    _fields_ = [("B", C.c_int32), ("n_s", C.c_int32), ("n_l", C.c_int32), ("ob_stride", C.c_int32),
                ("s_layers", _vp), ("l_map", _vp), ("l_bstride", C.c_int64), ("l_rstride", C.c_int64),
                ("ego_sl", _vp), ("ob_s", _vp), ("ob_l", _vp), ("n_ob", _vp), ("ob_dist", _vp),
                ("n_layers", _vp)]


class SceneBatch(C.Structure):       # mp_scene_batch
    _fields_ = [(n, _vp) for n in ("knots_x", "knots_y", "n_knots", "line_vec", "ego_vel", "st_pos", "st_rad",
                                   "n_st", "dy_pos", "dy_rad", "n_dy")]


class PlanDims(C.Structure):         # mp_plan_dims
    _fields_ = [(n, C.c_int32) for n in ("B", "n_knots_max", "n_st_max", "n_dy_max", "n_frames", "n_l", "m_max",
                                         "n_s_max", "p_max", "n_t", "st_n_s_max", "r_max", "q_max", "want_maps",
                                         "world")]


class WorldBatch(C.Structure):       # mp_world_batch
    _fields_ = [(n, _vp) for n in ("rot", "trans", "ego_vel_w", "way_xy", "n_way", "line_vec", "ob_pos", "ob_vel",
                                   "ob_rad", "n_ob")] + [("ob_max", C.c_int32), ("reserved", C.c_int32),
                                                         ("ignore_dist", C.c_double), ("frame_dt", C.c_double)]


class PlanResult(C.Structure):       # mp_plan_result
    _fields_ = [(n, _vp) for n in ("status", "n_points", "x", "y", "speed")]


MP_PLAN_STATIC_BYTES = 32 << 20
MP_PLAN_N_STAGES = 9


class PlanParams(C.Structure):       # mp_plan_params
    _fields_ = [("l_width", C.c_double), ("save_width", C.c_double), ("layer_gap", C.c_double),
                ("max_map_len", C.c_double), ("d_x", C.c_double), ("path_dt", C.c_double), ("sl", SlParams),
                ("speed_lim", C.c_double), ("st_ds", C.c_double), ("st_dt", C.c_double),
                ("path_planner", C.c_int32), ("reserved", C.c_int32)]


# name -> (restype, argtypes); every symbol include/mplattice.h declares
SIGNATURES = {
    "mp_version": (C.c_int, []),
    "mp_last_error": (C.c_char_p, []),
    "mp_set_knob": (C.c_int, [C.c_char_p, C.c_int32]),
    "mp_sl_plan": (C.c_int, [C.POINTER(SlBatch), C.POINTER(SlParams)] + [_vp] * 7),
    "mp_sl_plan_ordered": (C.c_int, [C.POINTER(SlBatch), C.POINTER(SlParams)] + [_vp] * 8),
    "mp_sl_plan_counted": (C.c_int, [C.POINTER(SlBatch), C.POINTER(SlParams)] + [_vp] * 6),
    "mp_sl_edge_cost": (C.c_int, [C.POINTER(SlBatch), C.POINTER(SlParams), _vp, _vp, _vp]),
    "mp_sl_edge_cost_nodes": (C.c_int, [C.c_int32] * 3 + [_vp] * 5 + [C.c_int32] + [_vp] * 3
                              + [C.POINTER(SlParams)] + [_vp] * 3),
    "mp_sl_dp": (C.c_int, [C.c_int32] * 3 + [_vp, _vp, C.POINTER(SlParams)] + [_vp] * 7),
    "mp_st_dp": (C.c_int, [C.c_int32] * 3 + [_vp] * 4 + [C.c_int32, _vp] + [C.c_double] * 3 + [_vp] * 5),
    "mp_st_dp_ordered": (C.c_int, [C.c_int32] * 3 + [_vp] * 4 + [C.c_int32, _vp] + [C.c_double] * 3 + [_vp] * 6),
    "mp_st_regions": (C.c_int, [C.c_int32] * 2 + [_vp] * 3 + [C.c_int32] + [_vp] * 4 + [C.c_int32] + [_vp] * 4
                      + [C.c_int32] * 2 + [_vp] * 5),
    "mp_refline_tables": (C.c_int, [C.c_int32, _vp, _vp, _vp, C.c_int32, _vp, _vp, C.c_double, C.c_int32]
                          + [_vp] * 14),
    "mp_f2c_batch": (C.c_int, [C.c_int32, _vp, _vp, _vp, C.c_int32] + [_vp] * 8),
    "mp_curve_eval": (C.c_int, [C.c_int32, _vp, _vp, _vp, C.c_int32, _vp, C.c_double, C.c_int32, _vp, _vp,
                                C.c_int32] + [_vp] * 6),
    "mp_c2f_batch": (C.c_int, [C.c_int32] + [_vp] * 5 + [C.c_int32] + [_vp] * 10),
    "mp_scene_ingest": (C.c_int, [C.c_int32] + [_vp] * 5 + [C.c_int32] + [_vp] * 4 + [C.c_int32, C.c_double,
                                                                                         C.c_int32, C.c_double]
                        + [_vp] * 12),
    "mp_robot_to_world": (C.c_int, [C.c_int32] + [_vp] * 8),
    "mp_result_record_bytes": (C.c_int32, [C.c_int32, C.c_int32]),
    "mp_pack_results": (C.c_int, [C.c_int32] * 3 + [_vp] * 8),
    "mp_plan_default_params": (None, [C.POINTER(PlanParams)]),
    "mp_plan_array_count": (C.c_int32, []),
    "mp_plan_array_info": (C.c_int, [C.POINTER(PlanDims), C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_int32),
                                     C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "mp_plan_arena_bytes": (C.c_int64, [C.POINTER(PlanDims)]),
    "mp_plan_scenes_static": (C.c_int, [C.POINTER(SceneBatch), C.POINTER(PlanDims), C.POINTER(PlanParams),
                                        C.POINTER(PlanResult), _vp]),
    "mp_plan_stage_name": (C.c_char_p, [C.c_int32]),
    "mp_plan_scenes_timed": (C.c_int, [C.POINTER(SceneBatch), C.POINTER(PlanDims), C.POINTER(PlanParams), _vp,
                                       C.c_int64, _vp, C.POINTER(C.c_float)]),
    "mp_plan_scenes": (C.c_int, [C.POINTER(SceneBatch), C.POINTER(PlanDims), C.POINTER(PlanParams), _vp, C.c_int64,
                                 _vp]),
    "mp_plan_world_scenes": (C.c_int, [C.POINTER(WorldBatch), C.POINTER(PlanDims), C.POINTER(PlanParams), _vp,
                                       C.c_int64, _vp]),
    "mp_plan_world_scenes_timed": (C.c_int, [C.POINTER(WorldBatch), C.POINTER(PlanDims), C.POINTER(PlanParams), _vp,
                                             C.c_int64, _vp, C.POINTER(C.c_float)]),
}

_lib = None


def lib():
    """Load the shared object once; raise if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libmplattice.so is missing: build it with `make -C motion_planning_amd/csrc` "
                "(there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class MpError(RuntimeError):
    pass


def set_knob(name, value=-1):
    """Testing aid: select a code path / tune a launch (names in include/mplattice.h);
    value < 0 puts the knob back to "not set"."""
    check(lib().mp_set_knob(name.encode(), int(value)), "mp_set_knob")


def check(rc, what):
    if rc != MP_OK:
        raise MpError("%s failed (%d): %s" % (what, rc, lib().mp_last_error().decode()))
