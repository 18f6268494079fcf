"""The C-ABI library builds, loads and exports every symbol include/*.h declares."""
import ctypes
import os
import subprocess

from volsurfs_amd import _lib


def test_library_builds_and_exports_all_declared_symbols():
    _lib.build()
    assert os.path.exists(_lib.LIB_PATH)
    cdll = ctypes.CDLL(_lib.LIB_PATH)
    names = _lib.declared_symbols()
    assert "vsa_composite_dense_fwd" in names
    for n in names:
        assert hasattr(cdll, n), f"{n} declared in include/volsurfs_hip.h but not exported"
    assert cdll.vsa_version() >= 1


def test_exported_symbols_are_c_linkage():
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True,
                         text=True, check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    for n in _lib.declared_symbols():
        assert n in exported


def test_argument_errors_are_reported_not_ignored():
    """Every entry point validates its arguments before touching the device and returns a
    negative VSA_ERR_* status (SURVEY §8b "Errors": never continue silently); checked here
    without a GPU, through the paths that return before any launch."""
    import pytest
    L = _lib.lib()
    null = ctypes.c_void_p(0)
    ERR_ARG, ERR_UNSUPPORTED = -1, -2
    hdr = open(_lib.HEADER_PATH).read()
    assert "#define VSA_ERR_ARG (-1)" in hdr and "#define VSA_ERR_UNSUPPORTED (-2)" in hdr
    assert L.vsa_composite_dense_fwd(null, null, null, 0, null, null, null, null, null, null, 10, 5, 0, null) == ERR_ARG
    assert L.vsa_composite_dense_bwd_l1(null, null, null, 0, null, null, ctypes.c_float(1.0), null, null, 10, 5, 0, null) == ERR_ARG
    assert L.vsa_composite_dense_fwd_bwd_l1(null, null, null, 0, null, ctypes.c_float(1.0), null, null, null, 10, 5, 0, null) == ERR_ARG
    assert L.vsa_trace(null, null, null, 1, 10, null, null, 10, ctypes.c_float(0), null, null, null, null) == ERR_ARG
    roots = (ctypes.c_int32 * 1)(0)
    assert L.vsa_trace(null, null, roots, 1, 99, null, null, 10, ctypes.c_float(0), null, null, null, null) == ERR_UNSUPPORTED   # tree deeper than the stack
    assert L.vsa_trace(null, null, roots, 1, 10, null, null, 0, ctypes.c_float(0), null, null, null, null) == 0                 # empty batch is fine
    assert L.vsa_trace(null, null, roots, 17, 10, null, null, 10, ctypes.c_float(0), null, null, null, null) == ERR_ARG        # > VSA_MAX_SHELLS
    assert L.vsa_nt_encode_fwd(null, null, null, null, null, null) == ERR_ARG
    assert L.vsa_nt_mlp_bwd(null, null, null, null, null, null, null, ctypes.c_float(1.0), null) == ERR_ARG
    assert L.vsa_nt_shade_fwd(null, null, null, null, null, null, null, null, 10, null, null, null, null, null, null) == ERR_ARG
    assert L.vsa_grid_encode_fwd(null, null, null, 10, null, null) == ERR_ARG
    assert L.vsa_sh_encode(null, 10, 7, null, null) == ERR_ARG                                   # degree > 4
    assert L.vsa_packed_sum_over_rays(null, null, null, null, 10, 1, null) == ERR_ARG
    u64 = ctypes.c_uint64(1)
    assert L.vsa_camera_rays(null, null, 4, 4, 1, 0, u64, u64, null, null, null, null) == ERR_ARG
    assert L.vsa_camera_rays(null, null, 0, 4, 1, 0, u64, u64, null, null, null, null) == 0      # no pixels
    assert L.vsa_camera_rays(null, null, 4, 4, 0, 0, u64, u64, null, null, null, null) == ERR_ARG  # R < 1
    f1 = ctypes.c_float(1.0)
    assert L.vsa_occ_grid_points(null, 8, 24, f1, f1, f1, 1, 0, u64, u64, null, null) == ERR_ARG    # 24 is not a power of two
    assert L.vsa_occ_grid_points(null, 0, 16, f1, f1, f1, 1, 0, u64, u64, null, null) == 0
    assert L.vsa_occ_update_values(null, null, 4, ctypes.c_float(1.5), null, null) == ERR_ARG       # decay > 1
    assert L.vsa_occ_check(null, 4, 16, f1, f1, f1, null, null, null, null, null, null) == ERR_ARG
    assert L.vsa_sample_fg_occupied(null, null, null, null, f1, 1, 8, 0, u64, u64, 16, f1, f1, f1, null, null,
                                    null, null, null, null, null, null, 4, null) == ERR_ARG
    assert L.vsa_tile_order(null, null, 12, 16, 3, 0, null) == ERR_ARG                              # 12 % 8
    assert L.vsa_tile_order(null, null, 16, 16, 3, 0, null) == ERR_ARG                              # null arrays
    assert L.vsa_reel_next_rays_batch(null, null, null, null, 0, 4, 4, 8, 1, 0, u64, u64, null, null, null,
                                      null, null, null, null) == ERR_ARG                             # no cameras
    assert L.vsa_count_hits(null, ctypes.c_longlong(10), null, null, null) == ERR_ARG
    assert L.vsa_l1_mean(null, null, ctypes.c_longlong(0), null, null, null) == ERR_ARG
    assert 0 < L.vsa_reduce_scratch_bytes() <= 1 << 16
    assert L.vsa_legacy_hit_prep(null, null, null, null, null, null, null, 10, 10, null, null, null, null) == ERR_ARG
    assert L.vsa_legacy_hit_prep(null, null, null, null, null, null, null, 0, 10, null, null, null, null) == 0          # no hits
    assert L.vsa_legacy_shade_out_fwd(null, 2, null, 0, 0, null, null, null, null, 0, 0, 5, 1, null, null, null, null, null, null,
                                      null) == ERR_ARG                                                              # < 3 colour channels
    assert L.vsa_legacy_shade_out_bwd(null, null, null, null, null, null, null, 10, 5, 0, null, 3, null, 0, null) == ERR_ARG
    # round-3 entry points
    assert L.vsa_nt_encode_mlp_fwd(null, null, null, null, null, null, null, null, null) == ERR_ARG
    assert L.vsa_packed_composite_fwd(null, null, null, null, null, null, 10, null) == ERR_ARG
    assert L.vsa_packed_composite_fwd(null, null, null, null, null, null, -1, null) == ERR_ARG
    assert L.vsa_packed_composite_bwd(null, null, null, null, null, null, null, null, 10, 1, null) == ERR_ARG
    fr = (ctypes.c_float * 6)(0, 0, 0, 1, 1, 1)
    assert L.vsa_nt_compact_frame(null, null, null, null, null, null, null, null) == ERR_ARG
    assert L.vsa_nt_rebalance(null, null) == ERR_ARG
    assert L.vsa_nt_balance_bytes() == 4 * (6 * 1025 + 6 * 1024 + 12 + 6 * 1024)
    ll = ctypes.c_longlong
    assert L.vsa_trace_q_fb(null, null, roots, fr, 1, 10, null, null, 10, ctypes.c_float(0), null, null, null,
                            null, ll(1 << 20), 0, null) == ERR_ARG                                        # null arrays
    assert L.vsa_trace_q_fb(null, null, roots, fr, 1, 10, null, null, 10, ctypes.c_float(0), null, null, null,
                            null, ll(1 << 20), 3, null) == ERR_ARG                                        # phase
    assert L.vsa_trace_q_fb(null, null, roots, fr, 1, 48, null, null, 10, ctypes.c_float(0), null, null, null,
                            null, ll(1 << 20), 0, null) == ERR_UNSUPPORTED
    assert L.vsa_trace_q_fb(null, null, roots, fr, 1, 10, null, null, 0, ctypes.c_float(0), null, null, null,
                            null, ll(0), 1, null) == 0
    assert L.vsa_trace_feedback_bytes(-1, 1) < 0
    assert L.vsa_trace_feedback_bytes(640000, 5) == 2 * ((16 + 50000 + 12 * (50000 // 8 + 64) + 255) // 256 * 256) + 256
    # round-5 entry points
    assert L.vsa_bvh_refit(null, null, 3) == ERR_ARG
    one = (ctypes.c_int32 * 1)(1)
    assert L.vsa_nt_encode_bwd_phased(null, null, null, f1, null, null, null, 1, one, null, null, null, 0, null) == ERR_ARG
    assert L.vsa_dp_flags_create(0, null) == ERR_ARG and L.vsa_dp_flags_destroy(null) == 0
    assert L.vsa_dp_flags_read(null, null) == ERR_ARG
    assert L.vsa_dp_signal(null, 0, null, 1, null) == ERR_ARG
    assert L.vsa_dp_stream_wait(null, 0, 1, 0, null) == ERR_ARG
    # round-2 entry points
    assert L.vsa_intersect_primitive(null, null, 10, 0, f1, null, null, null, null, null, null) == ERR_ARG
    assert L.vsa_intersect_primitive(null, null, 10, 2, f1, null, null, null, null, null, null) == ERR_ARG     # kind
    assert L.vsa_intersect_primitive(null, null, 0, 1, f1, null, null, null, null, null, null) == 0
    assert L.vsa_permuto_encode_fwd(null, null, null, null, 10, null, 48, null) == ERR_ARG
    assert L.vsa_permuto_encode_bwd(null, null, null, null, 48, 10, null, null) == ERR_ARG
    assert L.vsa_mlp_bwd(null, null, 0, 10, null, 0, null, null, null, null, null, null, 0, null, null) == ERR_ARG
    assert L.vsa_grid_encode_bwd_sliced(null, null, null, 10, null, null, null) == ERR_ARG
    assert L.vsa_grid_encode_bwd_binned(null, null, null, 10, null, null, null) == ERR_ARG
    assert L.vsa_grid_encode_bwd_binned_workspace(null, 10, null) == ERR_ARG
    h = ctypes.c_void_p()
    assert L.vsa_bvh_build(null, null, 0, 0, 4, ctypes.byref(h)) != 0
    # the Python layer turns any non-zero status into an exception
    with pytest.raises(_lib.VolsurfsHipError):
        _lib.call("vsa_sh_encode", None, 10, 7, None, None)


def test_import_volsurfs_resolves_to_the_mirror():
    """src/PyBridge.cxx:19 names the extension `volsurfs`; utils/background.py:3,5 and
    params/cmd_params.py:2,8-9 import it under that name and locate the repo root from it."""
    import importlib
    import inspect
    volsurfs = importlib.import_module("volsurfs")
    from volsurfs import OccupancyGrid, RaySampler, RaySamplesPacked, VolumeRendering
    from volsurfs_amd import volsurfs as mirror
    assert VolumeRendering is mirror.VolumeRendering and RaySampler is mirror.RaySampler
    assert OccupancyGrid is mirror.OccupancyGrid and RaySamplesPacked is mirror.RaySamplesPacked
    root = os.path.dirname(os.path.abspath(volsurfs.__file__))
    assert os.path.isdir(os.path.join(root, "volsurfs_amd"))          # module sits at the repo root
    # every method the pybind module exports exists with the same name (PyBridge.cxx:33-138)
    for cls, names in ((VolumeRendering, ["cumprod_one_minus_alpha_to_transmittance", "integrate_with_weights_1d",
                                          "integrate_with_weights_3d", "sdf2alpha", "sum_over_rays",
                                          "median_depth_over_rays", "cumsum_over_rays", "compute_cdf",
                                          "importance_sample", "combine_ray_samples_packets",
                                          "cumprod_one_minus_alpha_to_transmittance_backward",
                                          "integrate_with_weights_1d_backward",
                                          "integrate_with_weights_3d_backward", "sum_over_rays_backward"]),
                       (RaySampler, ["compute_samples_fg", "compute_samples_fg_in_grid_occupied_regions",
                                     "compute_samples_bg", "init_with_one_sample_per_ray", "contract_samples",
                                     "uncontract_samples"])):
        for n in names:
            assert callable(inspect.getattr_static(cls, n).__func__), n
    assert VolumeRendering.bug_compat is True      # default = what the reference computes


def test_every_declared_function_gets_its_prototype_from_the_header():
    """_lib.declared_prototypes: argument and return types parsed from include/volsurfs_hip.h and set on the loaded
    library (restype / argtypes), so that ctypes converts scalars to the declared width and refuses a call with the
    wrong number or kind of arguments instead of passing garbage (VERDICT r5 missing #5)."""
    import pytest
    protos = _lib.declared_prototypes()
    assert sorted(protos) == _lib.declared_symbols()
    res, args = protos["vsa_trace_q_fb"]
    assert res is ctypes.c_int and len(args) == 17 and args[14] is ctypes.c_longlong and args[9] is ctypes.c_float
    assert protos["vsa_trace_feedback_bytes"] == (ctypes.c_longlong, [ctypes.c_int, ctypes.c_int])
    assert protos["vsa_nt_balance_bytes"] == (ctypes.c_longlong, [])
    L = _lib.lib()
    for name, (res, args) in protos.items():
        fn = getattr(L, name)
        assert fn.restype is res and list(fn.argtypes) == args, name
    # a 64-bit count arrives as 64 bits, a short call and a float where an int belongs are refused
    assert L.vsa_trace_feedback_bytes(1 << 20, 5) > 0
    with pytest.raises(TypeError):
        L.vsa_trace_feedback_bytes(5)
    with pytest.raises(ctypes.ArgumentError):
        L.vsa_trace_feedback_bytes(1.5, 5)
    assert _lib.call("vsa_trace_coop_config", 16, 24, 4096) == 0
    with pytest.raises(_lib.VolsurfsHipError):
        _lib.call("vsa_trace_coop_config", 0, 24, 4096)
