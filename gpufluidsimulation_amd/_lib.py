"""ctypes view of the C-ABI (include/bimocq_gpu.h, include/bimocq_solver.h).

There is no CPU fallback: if the HIP library is missing this raises, and every operator
latches FL_ERR_NO_DEVICE when no gfx950 device is visible.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
HIP_SO = os.path.join(HERE, "libbimocq_hip.so")
HOST_SO = os.path.join(HERE, "libbimocq_host.so")

VP = C.c_void_p          # device pointers travel as plain addresses
c_f, c_i, c_b, c_d = C.c_float, C.c_int, C.c_bool, C.c_double

_G = [c_f, c_i, c_i, c_i]            # h, ni, nj, nk

HIP_SIGS = {
    # 1. reference operators
    "gpu_solve_forward": (None, [VP] * 6 + _G + [c_f, c_f]),
    "gpu_solve_backwardDMC": (None, [VP] * 9 + _G + [c_f]),
    "gpu_advect_velocity": (None, [VP] * 9 + _G + [c_b]),
    "gpu_advect_vel_double": (None, [VP] * 12 + _G + [c_b, c_f]),
    "gpu_advect_field": (None, [VP] * 5 + _G + [c_b]),
    "gpu_advect_field_double": (None, [VP] * 8 + _G + [c_b, c_f]),
    "gpu_advect_vel_double_global": (None, [VP] * 12 + _G + [c_b, c_f]),
    "gpu_advect_field_double_global": (None, [VP] * 8 + _G + [c_b, c_f]),
    "gpu_accumulate_velocity": (None, [VP] * 9 + _G + [c_b, c_f]),
    "gpu_accumulate_field": (None, [VP] * 5 + _G + [c_b, c_f]),
    "gpu_estimate_distortion": (None, [VP] * 7 + _G),
    "gpu_add": (None, [VP, VP, c_f, c_i]),
    "gpu_compensate_velocity": (None, [VP] * 15 + _G + [c_b]),
    "gpu_compensate_field": (None, [VP] * 9 + _G + [c_b]),
    "gpu_semilag": (None, [VP] * 5 + [c_i, c_i, c_i] + _G + [c_f, c_f]),
    "gpu_emit_smoke": (None, [VP] * 5 + _G + [c_f] * 7),
    "gpu_add_buoyancy": (None, [VP] * 3 + [c_i, c_i, c_i, c_f, c_f, c_f]),
    "gpu_diffuse_field": (None, [VP] * 3 + [c_i, c_i, c_i, c_i, c_f]),
    "gpu_add_field": (None, [VP, VP, VP, c_f, c_i]),
    "gpu_projection_jacobi": (None, [VP] * 7 + [c_i, c_i, c_i, c_i, c_f, c_f, c_f]),
    "gpu_clamp_extrema": (None, [VP] * 5 + [c_i] * 6 + [c_f] * 5),
    "gpu_mad": (None, [VP, VP, VP, c_f, c_f, c_i]),
    "gpu_conjugate_gradient": (None, [VP] * 8 + [c_i, c_i, c_i, c_i, c_f]),
    "gpu_multi_grid_conjugate_gradient": (None, [VP] * 11 + [c_i, c_i, c_d]),
    # 2. runtime
    "fl_context_create": (VP, [c_i]),
    "fl_context_make_current": (None, [VP]),
    "fl_context_current": (VP, []),
    "fl_context_destroy": (None, [VP]),
    "fl_init": (c_i, [c_i]),
    "fl_shutdown": (None, []),
    "fl_shutdown_all": (None, []),
    "fl_malloc": (VP, [C.c_size_t]),
    "fl_free": (None, [VP]),
    "fl_memset": (None, [VP, c_i, C.c_size_t]),
    "fl_memcpy_h2d": (None, [VP, VP, C.c_size_t]),
    "fl_memcpy_d2h": (None, [VP, VP, C.c_size_t]),
    "fl_memcpy_d2d": (None, [VP, VP, C.c_size_t]),
    "fl_sync": (None, []),
    "fl_aux_begin": (None, []),
    "fl_aux_end": (None, []),
    "fl_aux_join": (None, []),
    "fl_malloc_host": (VP, [C.c_size_t]),
    "fl_free_host": (None, [VP]),
    "fl_download_begin": (VP, [VP, VP, C.c_size_t]),
    "fl_download_wait": (c_i, [VP]),
    "fl_event_create": (VP, []),
    "fl_event_record": (None, [VP]),
    "fl_event_elapsed_ms": (c_f, [VP, VP]),
    "fl_event_destroy": (None, [VP]),
    "fl_last_error": (c_i, []),
    "fl_last_error_string": (C.c_char_p, []),
    "fl_clear_error": (None, []),
    "fl_compute_stream": (VP, []),
    "fl_set_option": (None, [c_i, c_i]),
    "fl_get_option": (c_i, [c_i]),
    "fl_jacobi_profile": (None, [C.POINTER(c_d), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "fl_jacobi_kernel_name": (C.c_char_p, []),
    "fl_mg_smooth_kernel_name": (C.c_char_p, []),
    "fl_mg_fused_launches": (C.c_longlong, []),
    # 3. additive
    "gpu_init_maps": (None, [VP, VP, VP] + _G),
    "gpu_maps_quarter_safe": (c_i, [VP, VP, VP] + _G),
    "fl_map_guard_reset": (None, [c_i]),
    "fl_map_guard_read": (None, [C.POINTER(c_i)]),
    "fl_nonfinite_seen": (c_i, [c_i]),
    "gpu_max_abs3": (c_f, [VP, VP, VP, c_i, c_i, c_i]),
    "gpu_divergence": (None, [VP] * 4 + [c_i, c_i, c_i, c_f]),
    "gpu_jacobi_sweeps": (c_i, [VP, VP, VP, c_i, c_i, c_i, c_i, c_f, c_f]),
    "gpu_gradient": (None, [VP] * 4 + [c_i, c_i, c_i, c_f]),
    "fl_comm_selftest": (c_i, []),
    "gpu_diffuse_sweeps": (c_i, [VP, VP, VP, c_i, c_i, c_i, c_i, c_f]),
    "gpu_max_field": (c_f, [VP, C.c_size_t]),
    "gpu_max_field_owned": (c_f, [VP, c_i, c_i, c_i]),
    "gpu_map_travel_z": (None, [VP, VP, c_f, c_i, c_i, c_i, C.POINTER(c_f)]),
    "gpu_smoothing_jacobi": (None, [VP, VP, VP, c_d, c_d, c_i, c_i, c_i, c_i]),
    "gpu_mgcg_slab_supported": (c_i, [c_i] * 8),
    "gpu_multi_grid_conjugate_gradient_slab": (None, [VP, VP, VP, VP] + [c_i] * 7 + [c_d]),
    "gpu_gradient_delta": (None, [VP] * 7 + [c_i, c_i, c_i, c_f]),
    "gpu_jacobi_sweep_range": (None, [VP, VP, VP, c_i, c_i, c_i, c_i, c_i, c_f, c_f]),
    "gpu_jacobi_sweep_pair_ranges": (c_i, [VP, VP, VP, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f]),
    "gpu_jacobi_sweep_triple_ranges": (c_i, [VP, VP, VP, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f]),
    "gpu_residual_norms": (None, [VP, VP, c_i, c_i, c_i, C.POINTER(c_d), C.POINTER(c_f)]),
    "gpu_clamp_extrema_box": (None, [VP, VP, c_i, c_i, c_i]),
    "gpu_clamp_extrema_box_w": (None, [VP, VP, c_i, c_i, c_i]),
    "gpu_compensate_error_velocity": (None, [VP] * 12 + _G + [c_b]),
    "gpu_compensate_error_field": (None, [VP] * 6 + _G + [c_b]),
    "gpu_advect_field2": (None, [VP] * 7 + _G + [c_b]),
    "gpu_compensate_error_field2": (None, [VP] * 9 + _G + [c_b]),
    "gpu_accumulate_field2": (None, [VP, VP, c_f, VP, VP, c_f] + [VP] * 3 + _G + [c_b]),
    "gpu_accumulate_velocity2": (None, [VP] * 3 + [c_f] + [VP] * 3 + [c_f] + [VP] * 6 + _G + [c_b]),
    "gpu_accumulate_component": (None, [VP, c_f, VP, c_f, VP, VP, VP, VP] + _G + [c_i, c_b]),
    "gpu_accumulate_velocity_identity": (None, [VP] * 9 + _G + [c_b, c_f]),
    "fl_report_error": (None, [c_i, C.c_char_p]),
    # 4. multi-GPU
    "fl_set_slab": (None, [c_i, c_i, c_i, c_i, c_i]),
    "fl_set_plane_window": (c_i, [c_i, c_i]),
    "fl_comm_set_null": (None, [c_i, c_i]),
    "fl_comm_unique_id": (c_i, [VP]),
    "fl_comm_init": (c_i, [VP, c_i, c_i]),
    "fl_comm_destroy": (None, []),
    "fl_comm_count": (c_i, []),
    "fl_comm_check": (c_i, [c_i]),
    "fl_comm_rank": (c_i, []),
    "fl_comm_size": (c_i, []),
    "fl_halo_exchange": (None, [c_i, C.POINTER(VP), C.POINTER(C.c_size_t), C.POINTER(c_i), c_i, c_i, c_i, c_i]),
    "fl_halo_wait": (None, []),
    "fl_comm_stats": (None, [VP, c_i]),
    "fl_comm_profile": (None, [C.POINTER(c_d), C.POINTER(C.c_longlong), c_i]),
    "fl_comm_rccl_version": (c_i, []),
    "fl_comm_set_custom": (None, [c_i, c_i, VP, VP]),
    # wall sheets (reference-faithful DMC border on z-slab ranks)
    "fl_box_pack": (None, [VP, c_i, c_i, c_i, c_i, VP, c_i, VP]),
    "fl_box_unpack": (None, [VP, c_i, c_i, c_i, c_i, VP, c_i, VP]),
    "fl_box_copy": (None, [VP, c_i, c_i, c_i, c_i, VP, c_i, c_i, VP, c_i]),
    "fl_p2p_exchange": (None, [c_i, VP, VP, VP, VP, VP]),
    "fl_p2p_exchange_begin": (None, [c_i, VP, VP, VP, VP, VP]),
    "fl_comm_set_custom_p2p": (None, [VP]),
    "gpu_accumulate_wall_fixup": (None, [VP, c_i, c_i, VP, VP, VP, VP, VP, c_f, c_i, c_i, c_i, c_i, c_f, VP, c_i, VP, c_i, VP, c_i]),
}

FL_OK, FL_ERR_NO_DEVICE, FL_ERR_HIP, FL_ERR_BAD_ARGUMENT, FL_ERR_UNSUPPORTED, FL_ERR_COMM = range(6)
FL_OPT_RESIDUAL_STRIDE, FL_OPT_SKIP_UNIT_BLEND, FL_OPT_JACOBI_VARIANT = 1, 2, 3
FL_OPT_PROFILE_JACOBI, FL_OPT_JACOBI_KCHUNK, FL_OPT_JACOBI_ROWS, FL_OPT_STRUCTURED_MAPS = 4, 5, 6, 7
FL_OPT_JACOBI_FUSE, FL_OPT_JACOBI_KCHUNK2, FL_OPT_MGCG_GRAPH, FL_OPT_FAST_LERP = 8, 9, 10, 11
FL_OPT_FUSED_HOUSEKEEPING = 12
FL_OPT_MAP_QUARTER_FP32 = 13
FL_OPT_MGCG_TILE = 14
FL_OPT_PROFILE_COMM = 15
FL_OPT_RESERVE_CUS = 16
FL_OPT_MGCG_BOTTOM = 17
FL_OPT_MGCG_FUSE = 20
FL_OPT_FIELD_WINDOW = 18
FL_OPT_COMM_CHECK = 19


class BimocqLibraryMissing(RuntimeError):
    pass


_hip = None


def hip_lib():
    """Load libbimocq_hip.so (the HIP kernels + runtime).  Raises if it has not been built."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_SO):
            raise BimocqLibraryMissing(
                f"{HIP_SO} not found: build it with `make` (or __graft_entry__.build()); "
                "there is no CPU fallback for the product path")
        lib = C.CDLL(HIP_SO)
        for name, (res, args) in HIP_SIGS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _hip = lib
        # The library registers the same teardown with the C runtime's atexit; interpreter shutdown comes first, while
        # every Python-side owner of device memory is still alive and the HIP runtime is whole (include/bimocq_gpu.h:
        # fl_shutdown_all).
        import atexit
        atexit.register(lib.fl_shutdown_all)
    return _hip


class BimocqError(RuntimeError):
    pass


def check(lib=None):
    """Raise if the library has latched an error (and clear it)."""
    lib = lib or hip_lib()
    code = lib.fl_last_error()
    if code != FL_OK:
        text = lib.fl_last_error_string().decode(errors="replace")
        lib.fl_clear_error()
        raise BimocqError(f"bimocq error {code}: {text}")
