"""ctypes binding of libpanogrf_hip.so (include/panogrf_hip.h).  Fails loudly when absent."""
import ctypes as C
import os

# torch ships its own libamdhip64.so.7; importing it FIRST makes libpanogrf_hip.so bind to that one
# runtime.  Loaded the other way round the process ends up with two HIP runtimes and launches from
# this library fail with "no ROCm-capable device is detected".
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))


def library_path():
    return os.environ.get("PANOGRF_HIP_LIB", os.path.join(_HERE, "libpanogrf_hip.so"))


class PgrfError(RuntimeError):
    pass


class PgrfFrame(C.Structure):
    _fields_ = [("rfn", C.c_int), ("H", C.c_int), ("W", C.c_int), ("hf", C.c_int), ("wf", C.c_int),
                ("hr", C.c_int), ("wr", C.c_int),
                ("imgs", C.c_void_p), ("img_feats", C.c_void_p), ("ray_feats", C.c_void_p),
                ("w2c", C.c_void_p), ("ref_depth_range", C.c_void_p),
                ("que_c2w", C.c_float * 12), ("que_depth_range", C.c_float * 2),
                ("dataset", C.c_int), ("perspective", C.c_int), ("que_kinv", C.c_float * 9)]


# PGRF_DATASET_*: cfg['dataset_name'] of the reference -> spherical convention of network/spt_utils.py
DATASETS = {'m3d': 0, 'replica_test': 1, 'residential': 2, 'CoffeeArea': 3}


class PgrfPassOut(C.Structure):
    _fields_ = [("pixel_colors", C.c_void_p), ("render_depth", C.c_void_p), ("hit_prob", C.c_void_p),
                ("colors", C.c_void_p), ("density", C.c_void_p), ("nan_flag", C.c_void_p), ("render_uncert", C.c_void_p)]


class PgrfPassGrad(C.Structure):
    _fields_ = [("pixel_colors", C.c_void_p), ("render_depth", C.c_void_p), ("hit_prob", C.c_void_p),
                ("colors", C.c_void_p), ("density", C.c_void_p)]


class PgrfPassDebug(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("px", "depth_v", "ddiff", "vis", "hit", "ray_feats", "rgb", "img_feats",
                                          "neuray", "base_x", "geo", "attn")]


class PgrfRenderCfg(C.Structure):
    _fields_ = [("min_depth", C.c_float), ("max_depth", C.c_float), ("use_disp", C.c_int), ("dn", C.c_int),
                ("fdn", C.c_int), ("hierarchical", C.c_int), ("fine_use_all", C.c_int), ("c2f_all", C.c_int),
                ("use_vis_coarse", C.c_int), ("use_vis_fine", C.c_int), ("precision", C.c_int)]


ABI_VERSION = 5      # PGRF_ABI_VERSION of include/panogrf_hip.h (struct layouts below)

# every symbol include/panogrf_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "pgrf_last_error": (C.c_char_p, []),
    "pgrf_abi_version": (C.c_int, []),
    "pgrf_pack_map": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "pgrf_blob_floats": (C.c_int, []),
    "pgrf_raw_count": (C.c_int, []),
    "pgrf_raw_name": (C.c_char_p, [C.c_int]),
    "pgrf_pack_weights": (C.c_int, [C.POINTER(_P), C.c_int, C.c_int, C.c_int, _P]),
    "pgrf_pack_weights_device": (C.c_int, [C.POINTER(_P), C.c_int, C.c_int, C.c_int, _P, _P]),
    "pgrf_pack_map_selftest": (C.c_int, [C.c_int, C.c_int]),
    "pgrf_sample_coarse_depth": (C.c_int, [C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, _P, _P]),
    "pgrf_sample_fine_depth": (C.c_int, [_P, _P, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "pgrf_sample_coarse_depth_train": (C.c_int, [C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "pgrf_sample_fine_depth_train": (C.c_int, [_P, _P, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "pgrf_render_pass_workspace_bytes": (C.c_size_t, [C.POINTER(PgrfFrame)]),
    "pgrf_render_pass": (C.c_int, [C.POINTER(PgrfFrame), _P, C.c_int, _P, _P, C.c_int, C.c_int,
                                   C.POINTER(PgrfPassOut), C.POINTER(PgrfPassDebug), _P, C.c_size_t, _P]),
    "pgrf_merge_c2f": (C.c_int, [_P, _P, _P, C.c_int, _P, _P, _P, C.c_int, C.c_int, C.POINTER(PgrfPassOut), _P]),
    "pgrf_merge_c2f_backward": (C.c_int, [_P, _P, _P, C.c_int, _P, _P, _P, C.c_int, C.c_int, C.POINTER(PgrfPassGrad), _P, _P, _P, _P, _P]),
    "pgrf_conv3d_upbrick_packed_floats": (C.c_size_t, [C.c_int, C.c_int]),
    "pgrf_pack_conv3d_upbrick": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "pgrf_conv3d_upbrick": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_float, _P, C.c_int, C.c_int, _P]),
    "pgrf_conv3d_small16_packed_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "pgrf_pack_conv3d_small16": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "pgrf_conv3d_small16": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _P, C.c_int, C.c_int, _P]),
    "pgrf_conv3d_upfused_packed_floats": (C.c_size_t, [C.c_int, C.c_int]),
    "pgrf_pack_conv3d_upfused": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "pgrf_conv3d_upfused": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_float, _P, C.c_int, C.c_int, _P]),
    "pgrf_comm_unique_id": (C.c_int, [_P]),
    "pgrf_comm_init": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(_P)]),
    "pgrf_allgather_image": (C.c_int, [_P, _P, _P, C.c_size_t, _P]),
    "pgrf_comm_destroy": (C.c_int, [_P]),
    "pgrf_comm_rccl_version": (C.c_int, [C.POINTER(C.c_int)]),
    "pgrf_render_pass_backward_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "pgrf_render_pass_backward": (C.c_int, [C.POINTER(PgrfFrame), C.POINTER(_P), C.c_int, C.c_int, _P, _P, C.c_int, C.c_int,
                                            C.POINTER(PgrfPassGrad), C.POINTER(_P), _P, _P, C.POINTER(PgrfPassOut), _P,
                                            C.c_size_t, _P]),
    "pgrf_render_pass_backward_phase": (C.c_int, [C.POINTER(PgrfFrame), C.POINTER(_P), C.c_int, C.c_int, _P, _P, C.c_int, C.c_int,
                                                  C.POINTER(PgrfPassGrad), C.POINTER(_P), _P, _P, C.POINTER(PgrfPassOut), _P,
                                                  C.c_size_t, _P, C.c_int]),
    "pgrf_depth_hypotheses": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int, C.c_float, C.c_float, _P,
                                        C.c_int, _P, _P]),
    "pgrf_cost_volume_erp": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.POINTER(C.c_int), C.c_int, C.c_float, C.c_int, C.c_int, _P, _P]),
    "pgrf_conv3d_packed_floats": (C.c_size_t, [C.c_int, C.c_int]),
    "pgrf_pack_conv3d": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "pgrf_conv3d_wrap": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, _P, C.c_int,
                                   C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _P, C.c_int, C.c_int, _P]),
    "pgrf_conv3d_wino_packed_floats": (C.c_size_t, [C.c_int, C.c_int]),
    "pgrf_pack_conv3d_wino": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "pgrf_conv3d_wino_supported": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "pgrf_conv3d_wino_preferred": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "pgrf_conv3d_wino": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_int, C.c_float, _P, C.c_int, C.c_int, _P]),
    "pgrf_conv3d_wino16_packed_floats": (C.c_size_t, [C.c_int, C.c_int]),
    "pgrf_pack_conv3d_wino16": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "pgrf_conv3d_wino16_supported": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "pgrf_conv3d_wino16_preferred": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "pgrf_conv3d_wino16": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_int, C.c_float, _P, C.c_int, C.c_int, _P]),
    "pgrf_conv3d_wino16_ksplit": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "pgrf_conv3d_wino16_splitk": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int,
                                          C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, _P, C.c_size_t, _P]),
    "pgrf_conv3d_wino16_pool": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int,
                                        C.c_int, C.c_int, C.c_int, C.c_float, _P, C.c_int, C.c_int, _P, C.c_int, C.c_int, _P]),
    "pgrf_conv3d_small_packed_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "pgrf_pack_conv3d_small": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "pgrf_conv3d_small": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _P, C.c_int, C.c_int, _P]),
    "pgrf_conv3d_cout1_gather_up": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                              _P, C.c_int, C.c_int, _P]),
    "pgrf_conv3d_cout1_project": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "pgrf_conv3d_cout1_gather": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_float, _P, C.c_int, C.c_int, _P]),
    "pgrf_avgpool3d_2": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "pgrf_upsample3d_2": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "pgrf_render_workspace_bytes": (C.c_size_t, [C.POINTER(PgrfFrame), C.POINTER(PgrfRenderCfg), C.c_int]),
    "pgrf_render_rays": (C.c_int, [C.POINTER(PgrfFrame), _P, _P, C.POINTER(PgrfRenderCfg), _P, C.c_int,
                                   C.POINTER(PgrfPassOut), C.POINTER(PgrfPassOut), _P, _P, C.c_size_t, _P]),
    "pgrf_conv2d_packed_floats": (C.c_size_t, [C.c_int, C.c_int]),
    "pgrf_pack_conv2d": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "pgrf_conv2d": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int,
                              C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, _P]),
    "pgrf_conv2d_direct": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, _P, C.c_int, C.c_int, _P]),
    "pgrf_maxpool3s2_wrap": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, _P]),
    "pgrf_instnorm_chunks": (C.c_int, [C.c_int]),
    "pgrf_instnorm": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_float, C.c_int, _P, C.c_int,
                                C.c_int, _P, C.c_int, C.c_int, _P, _P]),
    "pgrf_resize_bilinear": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_int, _P]),
    "pgrf_equirec_to_cube": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_longlong, C.c_longlong, C.c_longlong, C.c_longlong,
                                       _P, _P, C.c_int, C.c_int, _P, _P, _P, _P, C.c_longlong, C.c_longlong, C.c_longlong,
                                       C.c_longlong, _P]),
    "pgrf_render_rays_train": (C.c_int, [C.POINTER(PgrfFrame), _P, _P, C.POINTER(PgrfRenderCfg), _P, C.c_int, _P, _P,
                                         C.POINTER(PgrfPassOut), C.POINTER(PgrfPassOut), _P, _P, _P, C.c_size_t, _P]),
}


class _Lib:
    """Lazy handle: loading is deferred to first use so that importing the package on a box
    without the built library gives a clear error at the call site, not an ImportError maze."""

    def __init__(self):
        self._dll = None

    def _load(self):
        if self._dll is None:
            path = library_path()
            if not os.path.exists(path):
                raise PgrfError(
                    f"{path} not found: build it with `make -C panogrf_amd/csrc -j8` (or `python -c "
                    f"'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
            dll = C.CDLL(path)
            for name, (res, args) in SYMBOLS.items():
                fn = getattr(dll, name)   # AttributeError if the .so lacks a declared symbol
                fn.restype = res
                fn.argtypes = args
            ver = dll.pgrf_abi_version()
            if ver < 0 and os.environ.get("PGRF_ALLOW_ABLATION_LIB") != "1":
                raise PgrfError(f"{path} was built from experiment objects (ablation / timing switches, pgrf_abi_version() = {ver}): "
                                f"not the product library.  Profiling tools set PGRF_ALLOW_ABLATION_LIB=1 to load it on purpose.")
            if abs(ver) != ABI_VERSION:
                raise PgrfError(f"{path} has ABI version {ver}, this package binds version {ABI_VERSION}: rebuild with "
                                f"`make -C panogrf_amd/csrc -j8`")
            self._dll = dll
        return self._dll

    def __getattr__(self, name):
        return getattr(self._load(), name)

    def check(self, rc):
        if rc != 0:
            raise PgrfError(f"libpanogrf_hip error {rc}: {self.pgrf_last_error().decode()}")


lib = _Lib()
