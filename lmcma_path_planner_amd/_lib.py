"""ctypes loader for liblmcma_hip.so (HIP kernels + C ABI + host optimiser).

The library is the product: there is no Python or CPU fallback.  Importing this
module raises if the shared object has not been built (``make -C
lmcma_path_planner_amd/csrc`` or ``__graft_entry__.build()``).
"""
import ctypes
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LMCMA_HIP_LIB", os.path.join(_HERE, "lib", "liblmcma_hip.so"))

c_double_p = ctypes.POINTER(ctypes.c_double)
c_float_p = ctypes.POINTER(ctypes.c_float)
c_int32_p = ctypes.POINTER(ctypes.c_int32)
c_int64_p = ctypes.POINTER(ctypes.c_int64)
c_uint8_p = ctypes.POINTER(ctypes.c_uint8)


class HipConfig(ctypes.Structure):
    """lmcma_hip_config of include/lmcma_hip.h."""
    _fields_ = [
        ("struct_size", ctypes.c_int32),
        ("device", ctypes.c_int32),
        ("dims", ctypes.c_int32),
        ("waypoints", ctypes.c_int32),
        ("n_params", ctypes.c_int32),
        ("lambda_total", ctypes.c_int32),
        ("lambda_local", ctypes.c_int32),
        ("first_candidate", ctypes.c_int32),
        ("m", ctypes.c_int32),
        ("mu", ctypes.c_int32),
        ("z_source", ctypes.c_int32),
        ("cost_source", ctypes.c_int32),
        ("pad_mode", ctypes.c_int32),
        ("queries", ctypes.c_int32),
        ("seed", ctypes.c_uint64),
        ("w_len", ctypes.c_double),
        ("w_grid", ctypes.c_double),
        ("w_smooth", ctypes.c_double),
        ("start", ctypes.c_double * 3),
        ("goal", ctypes.c_double * 3),
        ("stream", ctypes.c_void_p),
    ]


class OptSummary(ctypes.Structure):
    """lmcma_hip_opt_summary of include/lmcma_hip.h."""
    _fields_ = [
        ("itr", ctypes.c_int32),
        ("iterator_sz", ctypes.c_int32),
        ("counteval", ctypes.c_int64),
        ("best_f", ctypes.c_double),
        ("sigma", ctypes.c_double),
        ("s", ctypes.c_double),
        ("done", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
    ]


class OptInfo(ctypes.Structure):
    """lmcma_opt_info of include/lmcma_opt.h."""
    _fields_ = [
        ("n", ctypes.c_int32),
        ("lambda_", ctypes.c_int32),
        ("m", ctypes.c_int32),
        ("generation", ctypes.c_int32),
        ("counteval", ctypes.c_int64),
        ("best_f", ctypes.c_double),
        ("sigma", ctypes.c_double),
    ]


class RosMarker(ctypes.Structure):
    """lmcma_ros_marker of include/lmcma_ros.h."""
    _fields_ = [
        ("frame_id", ctypes.c_char * 32),
        ("ns", ctypes.c_char * 32),
        ("id", ctypes.c_int32),
        ("type", ctypes.c_int32),
        ("action", ctypes.c_int32),
        ("position", ctypes.c_double * 3),
        ("orientation", ctypes.c_double * 4),
        ("scale", ctypes.c_double * 3),
        ("color", ctypes.c_float * 4),
        ("n_points", ctypes.c_int32),
        ("points", c_double_p),
    ]


class RosPointCloud2(ctypes.Structure):
    """lmcma_ros_pointcloud2 of include/lmcma_ros.h."""
    _fields_ = [
        ("frame_id", ctypes.c_char * 32),
        ("height", ctypes.c_uint32),
        ("width", ctypes.c_uint32),
        ("field_name", (ctypes.c_char * 4) * 3),
        ("field_offset", ctypes.c_uint32 * 3),
        ("field_datatype", ctypes.c_uint8 * 3),
        ("field_count", ctypes.c_uint32 * 3),
        ("is_bigendian", ctypes.c_uint8),
        ("point_step", ctypes.c_uint32),
        ("row_step", ctypes.c_uint32),
        ("is_dense", ctypes.c_uint8),
        ("data_bytes", ctypes.c_size_t),
        ("data", c_uint8_p),
    ]


class RosOctomap(ctypes.Structure):
    """lmcma_ros_octomap of include/lmcma_ros.h."""
    _fields_ = [
        ("frame_id", ctypes.c_char * 32),
        ("binary", ctypes.c_uint8),
        ("id", ctypes.c_char * 16),
        ("resolution", ctypes.c_double),
        ("data_bytes", ctypes.c_size_t),
        ("data", ctypes.POINTER(ctypes.c_int8)),
    ]


# every symbol include/lmcma_hip.h, lmcma_opt.h and lmcma_ros.h declare: (restype, argtypes)
_vp = ctypes.c_void_p
_i = ctypes.c_int
SIGNATURES = {
    # lmcma_hip.h
    "lmcma_hip_last_error": (ctypes.c_char_p, []),
    "lmcma_hip_version": (_i, []),
    "lmcma_hip_create": (_i, [ctypes.POINTER(HipConfig), ctypes.POINTER(_vp)]),
    "lmcma_hip_destroy": (_i, [_vp]),
    "lmcma_hip_set_grid": (_i, [_vp, _vp, _i, _i, _i, _i, ctypes.c_double]),
    "lmcma_hip_set_bounds": (_i, [_vp, c_double_p, c_double_p]),
    "lmcma_hip_set_grid_from_occupancy": (_i, [_vp, ctypes.c_void_p, _i, _i, _i, ctypes.c_double, ctypes.c_void_p]),
    "lmcma_hip_set_cov_factor": (_i, [_vp, c_double_p]),
    "lmcma_hip_set_state": (_i, [_vp, c_double_p, ctypes.c_double, c_double_p, c_double_p, c_double_p, _i,
                                 ctypes.c_double]),
    "lmcma_hip_sample": (_i, [_vp, ctypes.c_uint32, c_double_p]),
    "lmcma_hip_evaluate": (_i, [_vp]),
    "lmcma_hip_set_costs": (_i, [_vp, _i, _i, c_double_p]),
    "lmcma_hip_select": (_i, [_vp, c_float_p, c_int32_p]),
    "lmcma_hip_rank_noalloc": (_i, [_vp, _i, _vp, _vp, _vp]),
    "lmcma_hip_recombine": (_i, [_vp, c_double_p, c_double_p]),
    "lmcma_hip_get_rows": (_i, [_vp, _i, _i, c_double_p]),
    "lmcma_hip_get_costs": (_i, [_vp, _i, _i, c_double_p]),
    "lmcma_hip_get_terms": (_i, [_vp, _i, _i, c_double_p]),
    "lmcma_hip_score_rows": (_i, [_vp, c_float_p, _i, c_float_p, c_float_p]),
    "lmcma_hip_opt_init": (_i, [_vp, c_double_p, ctypes.c_double]),
    "lmcma_hip_opt_reset": (_i, [_vp]),
    "lmcma_hip_opt_set_query": (_i, [_vp, _i, c_double_p, ctypes.c_double]),
    "lmcma_hip_query_select": (_i, [_vp, _i]),
    "lmcma_hip_queries": (_i, [_vp]),
    "lmcma_hip_run": (_i, [_vp, _i]),
    "lmcma_hip_phase_local": (_i, [_vp]),
    "lmcma_hip_phase_local_range": (_i, [_vp, _i, _i]),
    "lmcma_hip_phase_rank": (_i, [_vp]),
    "lmcma_hip_phase_update": (_i, [_vp, _i]),
    "lmcma_hip_comm_unique_id": (_i, [ctypes.c_char_p]),
    "lmcma_hip_comm_init": (_i, [_vp, ctypes.c_char_p, _i, _i]),
    "lmcma_hip_comm_attach": (_i, [_vp, _vp, _i, _i]),
    "lmcma_hip_comm_destroy": (_i, [_vp]),
    "lmcma_hip_run_sharded": (_i, [_vp, _i]),
    "lmcma_hip_peer_handle": (_i, [_vp, ctypes.c_char_p]),
    "lmcma_hip_peer_connect": (_i, [_vp, ctypes.c_char_p, _i, _i]),
    "lmcma_hip_peer_enable": (_i, [_vp, _i]),
    "lmcma_hip_peer_disconnect": (_i, [_vp]),
    "lmcma_hip_peer_status": (_i, [_vp, ctypes.POINTER(_i), ctypes.POINTER(_i)]),
    "lmcma_hip_run_sharded_group": (_i, [ctypes.POINTER(_vp), _i, _i]),
    "lmcma_hip_opt_get": (_i, [_vp, ctypes.POINTER(OptSummary), c_double_p]),
    "lmcma_hip_opt_get_vectors": (_i, [_vp, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p,
                                       c_int32_p]),
    "lmcma_hip_buffer": (_i, [_vp, _i, ctypes.POINTER(_vp), ctypes.POINTER(ctypes.c_size_t)]),
    "lmcma_hip_row_stride": (_i, [_vp]),
    "lmcma_hip_synchronize": (_i, [_vp]),
    "lmcma_hip_stream_mark": (_i, [_vp, _i]),
    "lmcma_hip_stream_elapsed": (_i, [_vp, c_double_p]),
    "lmcma_hip_timing_enable": (_i, [_vp, _i]),
    "lmcma_hip_timing_get": (_i, [_vp, c_double_p, c_int64_p]),
    "lmcma_hip_timing_event_overhead": (_i, [_vp, _i, c_double_p]),
    # lmcma_ros.h
    "lmcma_ros_path_marker": (_i, [c_double_p, _i, _i, c_double_p, c_double_p, ctypes.c_double, ctypes.c_double,
                                   c_double_p, _i, ctypes.POINTER(RosMarker)]),
    "lmcma_ros_path_marker_delete": (_i, [ctypes.POINTER(RosMarker)]),
    "lmcma_ros_grid_pointcloud2": (_i, [_vp, _i, _i, _i, ctypes.c_double, c_double_p, _vp, ctypes.c_size_t,
                                        ctypes.POINTER(RosPointCloud2)]),
    "lmcma_ros_grid_octomap_binary": (_i, [_vp, _i, _i, _i, ctypes.c_double, ctypes.POINTER(ctypes.c_int), _vp,
                                           ctypes.c_size_t, ctypes.POINTER(RosOctomap)]),
    # lmcma_opt.h
    "lmcma_opt_last_error": (ctypes.c_char_p, []),
    "lmcma_opt_create": (_i, [c_double_p, c_double_p, c_double_p, c_double_p, _i, _i, ctypes.c_double, _i, _i,
                              ctypes.POINTER(_vp)]),
    "lmcma_opt_create_sep": (_i, [c_double_p, c_double_p, c_double_p, c_double_p, _i, _i, ctypes.c_double, _i, _i,
                                  ctypes.POINTER(_vp)]),
    "lmcma_opt_create_chol": (_i, [c_double_p, c_double_p, c_double_p, c_double_p, _i, _i, ctypes.c_double, _i, _i,
                                   ctypes.POINTER(_vp)]),
    "lmcma_opt_destroy": (_i, [_vp]),
    "lmcma_opt_set_options": (_i, [_vp, _i, _i, _i, _i]),
    "lmcma_opt_set_shard": (_i, [_vp, _i, _i, ctypes.c_char_p]),
    "lmcma_opt_set_problem": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, ctypes.c_double, c_double_p, c_double_p,
                                   ctypes.c_double, ctypes.c_double, ctypes.c_double, _i]),
    "lmcma_opt_init": (_i, [_vp, _i]),
    "lmcma_opt_ask": (_i, [_vp, c_double_p, _i]),
    "lmcma_opt_tell": (_i, [_vp, c_double_p, _i]),
    "lmcma_opt_done": (_i, [_vp]),
    "lmcma_opt_step": (_i, [_vp, _i]),
    "lmcma_opt_get_info": (_i, [_vp, ctypes.POINTER(OptInfo)]),
    "lmcma_opt_get_mean": (_i, [_vp, c_double_p]),
    "lmcma_opt_get_population": (_i, [_vp, c_double_p]),
    "lmcma_opt_get_last_costs": (_i, [_vp, c_double_p]),
    "lmcma_opt_get_diagonal": (_i, [_vp, c_double_p]),
    "lmcma_opt_get_factor": (_i, [_vp, c_double_p]),
    "lmcma_host_gauss_stream": (_i, [_i, _i, c_double_p]),
    "lmcma_host_uniform_stream": (_i, [_i, _i, c_double_p]),
    "lmcma_host_covariance": (_i, [_i, _i, c_double_p]),
    "lmcma_host_differentiation_matrix": (_i, [_i, _i, ctypes.c_double, c_double_p, _i]),
    "lmcma_host_cholesky": (_i, [c_double_p, c_double_p, _i]),
    "lmcma_host_apply_cov_l": (_i, [c_double_p, c_double_p, _i]),
    "lmcma_host_sort": (_i, [_i, c_double_p, c_int32_p]),
    "lmcma_grid_two_bar": (_i, [_i, _i, c_uint8_p]),
    "lmcma_grid_random_boxes": (_i, [_i, _i, _i, ctypes.c_uint64, ctypes.c_double, c_double_p, c_double_p,
                                     c_uint8_p]),
    "lmcma_grid_edt": (_i, [c_uint8_p, _i, _i, _i, c_float_p]),
    "lmcma_grid_8ssedt": (_i, [c_uint8_p, _i, _i, c_int32_p, c_double_p]),
    "lmcma_grid_state_cost": (_i, [c_float_p, ctypes.c_long, ctypes.c_double, c_float_p]),
    "lmcma_grid_read_binvox": (_i, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int), c_double_p, c_double_p, c_uint8_p,
                                    ctypes.c_long]),
    "lmcma_grid_read_bmp": (_i, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                 c_uint8_p, ctypes.c_long]),
}

_lib = None


def load():
    """Load liblmcma_hip.so once and attach the prototypes.  Raises if missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "liblmcma_hip.so is not built (%s): run `make -C lmcma_path_planner_amd/csrc` "
            "or __graft_entry__.build(); there is no CPU fallback" % LIB_PATH)
    # The PyTorch ROCm wheel ships its own copies of the HIP and HSA runtimes under other
    # library names.  Loaded after this library's runtime they start a second runtime in the
    # process, which finds no GPU ("No HIP GPUs are available"); loaded first, their symbols are
    # global and this library's runtime binds to them, so both see the same device.  A process
    # that will use torch next to this library (torch views of the library's buffers, RCCL
    # through torch.distributed) therefore needs torch loaded first.
    if os.environ.get("LMCMA_HIP_NO_TORCH_PRELOAD") is None and "torch" not in sys.modules:
        if importlib.util.find_spec("torch") is not None:
            import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header and library disagree
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


class LmcmaError(RuntimeError):
    pass


def check_hip(rc, what):
    if rc != 0:
        raise LmcmaError("%s failed (%d): %s" % (what, rc, load().lmcma_hip_last_error().decode()))


def check_opt(rc, what):
    if rc < 0:
        raise LmcmaError("%s failed: %s" % (what, load().lmcma_opt_last_error().decode()))
