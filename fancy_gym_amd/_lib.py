"""
ctypes binding of ``libmpk.so`` (include/mpk.h).  There is NO CPU fallback: if the HIP library is missing or cannot be
loaded every entry point of the package that needs it raises ``MPKLibraryError``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MPK_LIB") or os.path.join(_HERE, "libmpk.so")   # MPK_LIB: A/B builds of the library

MPK_ABI_VERSION = 4
MP_TYPES = {"promp": 0, "dmp": 1, "prodmp": 2}
PHASE_TYPES = {"linear": 0, "exp": 1}
BASIS_TYPES = {"rbf": 0, "zero_rbf": 1, "prodmp": 2}
CTRL_TYPES = {"motor": 0, "velocity": 1, "position": 2}
PLANT_TYPES = {"static": 0, "double_integrator": 1}

MPK_EINVAL, MPK_ENOTIMPL, MPK_EHIP, MPK_ERANGE, MPK_ENODEV, MPK_ECOMM = -1, -2, -3, -4, -5, -6
MPK_COMM_ID_BYTES = 128
MPK_OPT_AUTO = -1
# mp_pytorch semantics that cannot be checked here: explicit switches (include/mpk.h); first entry = default
RELATIVE_GOAL_MODES = {"before_scale": 0, "after_scale": 1}   # first = the shipped default (include/mpk.h)
GOAL_OFFSET_MODES = {"ignore": 0, "add": 1}
SINGLE_RBF_MODES = {"unit_gap": 0, "refuse": 1}
DMP_FIRST_SAMPLE_MODES = {"init": 0, "step": 1}
REWARD_TYPES = {None: 0, "none": 0, "simple_reacher": 1}       # MPK_REWARD_*
AGG_MODES = {"sum": 0, "mean": 1, "last": 2}                   # MPK_AGG_*
OPTION_KEYS = ("mapping", "bulk", "quad", "pd_quad", "write_through", "ipw", "phase", "phase_table", "phase_chunk",
               "pd_simple", "split", "lds_pad", "pipe", "flat", "phase_flat", "ring", "ring_np", "ring_ns", "ring_m", "ring_dbg", "ring_parts", "tiles_wpb", "serial_order", "ring_nc", "pd_generic", "dmp_response", "ablations", "ring_tb", "pd_helper", "phase_waves", "phase_split", "phase_pipe", "pd_pipe")


class MPKLibraryError(RuntimeError):
    """libmpk.so is missing / failed to load / reported a HIP failure."""


class mpk_config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("device", C.c_int32),
        ("mp_type", C.c_int32), ("phase_type", C.c_int32), ("basis_type", C.c_int32),
        ("num_dof", C.c_int32), ("num_basis", C.c_int32), ("num_basis_outside", C.c_int32),
        ("num_basis_zero_start", C.c_int32), ("num_basis_zero_goal", C.c_int32),
        ("learn_tau", C.c_int32), ("learn_delay", C.c_int32),
        ("auto_scale_basis", C.c_int32), ("relative_goal", C.c_int32),
        ("disable_goal", C.c_int32), ("disable_weights", C.c_int32),
        ("pre_compute_length_factor", C.c_int32),
        ("relative_goal_mode", C.c_int32), ("goal_offset_mode", C.c_int32), ("single_rbf_mode", C.c_int32),
        ("dmp_first_sample", C.c_int32), ("reserved0", C.c_int32),
        ("tau", C.c_double), ("delay", C.c_double), ("alpha_phase", C.c_double),
        ("tau_bound", C.c_double * 2), ("delay_bound", C.c_double * 2),
        ("basis_bandwidth_factor", C.c_double), ("basis_alpha", C.c_double), ("basis_dt", C.c_double),
        ("weights_scale", C.c_double), ("goal_scale", C.c_double), ("dmp_alpha", C.c_double),
        ("dt", C.c_double), ("duration", C.c_double), ("goal_offset", C.c_double),
    ]


class mpk_rollout_cfg(C.Structure):
    _fields_ = [
        ("controller_type", C.c_int32), ("plant_type", C.c_int32), ("dt", C.c_double),
        ("p_gains", C.POINTER(C.c_double)), ("d_gains", C.POINTER(C.c_double)),
        ("act_low", C.POINTER(C.c_double)), ("act_high", C.POINTER(C.c_double)),
    ]


class mpk_validity_gate(C.Structure):
    _fields_ = [
        ("pos_low", C.POINTER(C.c_double)), ("pos_high", C.POINTER(C.c_double)),
        ("check_tau_delay", C.c_int32), ("reserved0", C.c_int32),
        ("tau_bound", C.c_double * 2), ("delay_bound", C.c_double * 2),
        ("raw_params", C.c_void_p), ("valid", C.c_void_p), ("penalty", C.c_void_p),
    ]


class mpk_replan_state(C.Structure):
    _fields_ = [
        ("traj_steps", C.c_void_p), ("plan_steps", C.c_void_p), ("done", C.c_void_p), ("seg_len", C.c_void_p),
        ("done_out", C.c_void_p), ("cond_pos", C.c_void_p), ("cond_vel", C.c_void_p),
        ("every", C.c_int32), ("max_planning_times", C.c_int32), ("horizon", C.c_int32), ("reserved0", C.c_int32),
    ]


_vp, _i32, _dbl = C.c_void_p, C.c_int32, C.c_double

# name -> (restype, argtypes); one row per symbol declared in include/mpk.h
SIGNATURES = {
    "mpk_last_error": (C.c_char_p, []),
    "mpk_abi_version": (C.c_int, []),
    "mpk_source_hash": (C.c_char_p, []),
    "mpk_device_count": (C.c_int, []),
    "mpk_create": (C.c_int, [C.POINTER(mpk_config), C.POINTER(_vp)]),
    "mpk_destroy": (None, [_vp]),
    "mpk_num_params": (C.c_int, [_vp]),
    "mpk_num_steps": (C.c_int, [_vp]),
    "mpk_num_dof": (C.c_int, [_vp]),
    "mpk_params_bounds": (C.c_int, [_vp, _vp, _vp]),
    "mpk_set_duration": (C.c_int, [_vp, _dbl, _dbl]),
    "mpk_set_option": (C.c_int, [_vp, C.c_char_p, C.c_int64]),
    "mpk_get_option": (C.c_int, [_vp, C.c_char_p, C.POINTER(C.c_int64)]),
    "mpk_times": (C.c_int, [_vp, _vp]),
    "mpk_trajectory": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _dbl, _vp, _vp, _i32, _vp]),
    "mpk_trajectory_actions": (C.c_int, [_vp, _vp, _vp, _vp, _dbl, C.POINTER(mpk_rollout_cfg), _vp, _vp,
                                         _vp, _vp, _vp, _i32, _vp]),
    "mpk_trajectory_rollout": (C.c_int, [_vp, _vp, _vp, _vp, _dbl, C.POINTER(mpk_rollout_cfg), _vp, _vp, _vp,
                                         _vp, _vp, _vp, _i32, _vp]),
    "mpk_episode_reset": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "mpk_replan_step": (C.c_int, [_vp, _vp, _vp, _vp, _dbl, C.POINTER(mpk_rollout_cfg), _vp, _vp,
                                  C.POINTER(mpk_replan_state), _vp, _vp, _vp, _i32, _vp]),
    "mpk_episode_return": (C.c_int, [_vp, _vp, _vp, _vp, _dbl, C.POINTER(mpk_rollout_cfg), _vp, _vp, C.POINTER(mpk_replan_state),
                                     _vp, _vp, _i32, _vp, _vp, _i32, _i32, _vp, _i32, _vp]),
    "mpk_replan_step_gated": (C.c_int, [_vp, _vp, _vp, _vp, _dbl, C.POINTER(mpk_rollout_cfg), _vp, _vp,
                                        C.POINTER(mpk_replan_state), C.POINTER(mpk_validity_gate), _vp, _vp, _vp, _i32, _vp]),
    "mpk_episode_return_gated": (C.c_int, [_vp, _vp, _vp, _vp, _dbl, C.POINTER(mpk_rollout_cfg), _vp, _vp, C.POINTER(mpk_replan_state),
                                           C.POINTER(mpk_validity_gate), _vp, _vp, _i32, _vp, _vp, _i32, _i32, _vp, _i32, _vp]),
    "mpk_reward_aggregate": (C.c_int, [_vp, _vp, _vp, _i32, _vp, _i32, _i32, _vp]),
    "mpk_pd_rollout": (C.c_int, [_vp, C.POINTER(mpk_rollout_cfg), _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "mpk_condition_gather": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "mpk_unpin_tables": (C.c_int, [_vp]),
    "mpk_check_range": (C.c_int, [_vp, _vp]),
    "mpk_poll_fault": (C.c_int, [_vp]),
    "mpk_gate_flags": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "mpk_reacher_rollout": (C.c_int, [_vp, C.POINTER(mpk_rollout_cfg), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp,
                                      _vp, _i32, _i32, _vp]),
    "mpk_replan_advance": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "mpk_traj_validity": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _i32, _vp]),
    "mpk_traj_validity_penalty": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "mpk_prodmp_tables": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mpk_prodmp_indices": (C.c_int, [_vp, _dbl, _vp, _vp, _vp]),
    "mpk_scaled_basis": (C.c_int, [_vp, _vp, _i32, _vp, _vp]),
    "mpk_selftest_division": (C.c_int, [_vp, C.c_float, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64), _vp]),
    "mpk_host_prodmp_tables": (C.c_int, [C.POINTER(mpk_config), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mpk_host_rbf": (C.c_int, [C.POINTER(mpk_config), _vp, _vp]),
    "mpk_host_times": (C.c_int, [_dbl, _dbl, _vp, _i32]),
    "mpk_host_num_params": (C.c_int, [C.POINTER(mpk_config)]),
    "mpk_comm_unique_id": (C.c_int, [_vp]),
    "mpk_comm_create": (C.c_int, [_vp, _i32, _i32, _i32, C.POINTER(_vp)]),
    "mpk_comm_rank": (C.c_int, [_vp]),
    "mpk_comm_world": (C.c_int, [_vp]),
    "mpk_allgather": (C.c_int, [_vp, _vp, _vp, C.c_int64, _vp]),
    "mpk_comm_destroy": (None, [_vp]),
    "mpk_last_kernel": (C.c_char_p, [_vp]),
}

_lib: Optional[C.CDLL] = None

# the files libmpk.so is built from, in the order mpk_source_hash() is defined over (include/mpk.h)
_ROOT = os.path.dirname(_HERE)
KERNEL_UNITS = ("mpk_traj_family.hip", "mpk_traj_ring.hip", "mpk_episode.hip", "mpk_traj_launch.hip", "mpk_traj_wide.hip", "mpk_traj_phase.hip",
                "mpk_phase_fused.hip", "mpk_rollout.hip", "mpk_misc.hip")          # translation units of the device code (mpk_traj_family.hip: once per MP type)
KERNEL_HEADERS = ("mpk_dev.h", "mpk_tile.h", "mpk_traj_tiles.h", "mpk_traj_stream.h", "mpk_traj_flat.h", "mpk_traj_ring.h", "mpk_traj_quad.h",
                  "mpk_traj_pipe.h", "mpk_reward.h", "mpk_phase.h", "mpk_trace_reader.h")
SOURCE_FILES = (os.path.join(_ROOT, "include", "mpk.h"), os.path.join(_HERE, "csrc", "mpk_internal.h"),
                os.path.join(_HERE, "csrc", "mpk_host.cpp")) + \
    tuple(os.path.join(_HERE, "csrc", f) for f in KERNEL_HEADERS + KERNEL_UNITS)


def stamp(extra_flags: str = "") -> Optional[str]:
    """what a build stamps into the library: the source hash, plus a tag of the extra compile flags of an A/B build (such a
    library is then NOT the checked-out sources for load() / _stale(): it needs MPK_LIB to be used)"""
    import hashlib
    h = source_hash()
    if h is None or not extra_flags.strip():
        return h
    return h[:48] + "f1a9" + hashlib.sha256(extra_flags.strip().encode()).hexdigest()[:12]


def source_hash() -> Optional[str]:
    """sha256 over the checked-out sources (None when the tree carries no sources, e.g. a binary-only install)"""
    import hashlib
    h = hashlib.sha256()
    for path in SOURCE_FILES:
        if not os.path.exists(path):
            return None
        h.update(os.path.basename(path).encode() + b"\n")
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def embedded_source_hash(path: str = None) -> Optional[str]:
    """the hash stamped into a built library, read from the FILE (no dlopen: a stale library must not get loaded)"""
    import re
    path = path or LIB_PATH
    try:
        with open(path, "rb") as f:
            m = re.search(rb"MPK_SOURCE_HASH=([0-9a-f]{64}|unstamped)", f.read())
    except OSError:
        return None
    return m.group(1).decode() if m else None


def load() -> C.CDLL:
    """Load libmpk.so (once) and bind every declared symbol; raise MPKLibraryError if that is impossible."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MPKLibraryError(
            f"{LIB_PATH} not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(hipcc --offload-arch=gfx950). fancy_gym_amd has no CPU fallback.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover - depends on the machine
        raise MPKLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise MPKLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.mpk_abi_version() != MPK_ABI_VERSION:
        raise MPKLibraryError("libmpk.so ABI version mismatch")
    # the loaded binary must be the checked-out sources (MPK_LIB names a deliberate A/B build: exempt)
    want, got = source_hash(), (lib.mpk_source_hash() or b"").decode()
    if want is not None and "MPK_LIB" not in os.environ and got != want:
        raise MPKLibraryError(
            f"{LIB_PATH} was built from other sources (stamped {got[:12]}..., checked out {want[:12]}...). Rebuild it with "
            f"`python -c 'import __graft_entry__ as g; g.build()'`.")
    _lib = lib
    return lib


def last_error() -> str:
    msg = load().mpk_last_error()
    return msg.decode() if msg else ""


def check(rc: int) -> int:
    """Map a negative return code to the exception class the reference raises for the same condition."""
    if rc >= 0:
        return rc
    msg = last_error()
    if rc == MPK_EINVAL:
        raise ValueError(msg)
    if rc == MPK_ENOTIMPL:
        raise NotImplementedError(msg)
    if rc == MPK_ERANGE:
        raise RuntimeError(msg)
    raise MPKLibraryError(f"libmpk error {rc}: {msg}")


def set_option(key: str, value: int = MPK_OPT_AUTO, handle=None) -> None:
    """mpk_set_option: kernel-selection override (include/mpk.h); handle None = the process-wide default"""
    check(load().mpk_set_option(handle, key.encode(), int(value)))


def get_option(key: str, handle=None) -> int:
    v = C.c_int64()
    check(load().mpk_get_option(handle, key.encode(), C.byref(v)))
    return int(v.value)


def reset_options(handle=None) -> None:
    for k in OPTION_KEYS:
        set_option(k, MPK_OPT_AUTO, handle)
