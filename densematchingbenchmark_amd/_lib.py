"""ctypes binding of libdmb_hip.so -- the only way the Python host layer reaches the HIP kernels.

PyTorch is plumbing here: it owns device memory (``tensor.data_ptr()``) and the current HIP stream
(``torch.cuda.current_stream().cuda_stream``); everything that computes lives behind the C ABI of
``include/dmb_hip.h``.  There is NO fallback: if the library is missing or a tensor is not a CUDA/HIP
tensor, the call raises.
"""
import ctypes
import os
import re

import torch  # imported first on purpose: the library then binds to the HIP runtime torch already loaded

_PKG = os.path.dirname(os.path.abspath(__file__))
# DMB_LIB=dev (development scripts only): the build with the kernel-variant switches, lib/libdmb_hip_dev.so (build.py dev=True)
DEV_BUILD = os.environ.get("DMB_LIB", "").startswith("dev")    # "dev" or "dev_<tag>" (a build-time experiment, build.py)
LIB_PATH = os.path.join(_PKG, "lib", "libdmb_hip_%s.so" % os.environ["DMB_LIB"] if DEV_BUILD else "libdmb_hip.so")
DECONV3D_WORKSPACE_BYTES = 2048   # include/dmb_hip.h: DMB_DECONV3D_WORKSPACE_BYTES
CONV_SINGLE_CHAIN = 0x100         # include/dmb_hip.h: DMB_CONV_SINGLE_CHAIN
HEADER_PATH = os.path.join(os.path.dirname(_PKG), "include", "dmb_hip.h")

_c_int, _c_float, _c_void_p, _c_ll = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_longlong
_P = _c_void_p  # device pointer
_HI = ctypes.POINTER(ctypes.c_int)  # host int array
_HF = ctypes.POINTER(ctypes.c_float)  # host float array

ABI_VERSION = 8

# name -> (restype, argtypes)
SIGNATURES = {
    "dmb_abi_version": (_c_int, []),
    "dmb_last_error": (ctypes.c_char_p, []),
    "dmb_build_id": (ctypes.c_char_p, []),
    "dmb_cat_fms_f32": (_c_int, [_P, _P, _P, _c_int, _c_int, _c_int, _c_int, _c_int, _HI, _P]),
    "dmb_dif_fms_f32": (_c_int, [_P, _P, _P, _c_int, _c_int, _c_int, _c_int, _c_int, _HI, _P]),
    "dmb_gwc_fms_f32": (_c_int, [_P, _P, _P, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _HI, _c_int, _c_int, _P]),
    "dmb_fast_cat_fms_f32": (_c_int, [_P, _P, _P, _P] + [_c_int] * 6 + [_P]),
    "dmb_fast_dif_fms_f32": (_c_int, [_P, _P, _P, _P] + [_c_int] * 7 + [_c_float, _P]),
    "dmb_fast_fms_bwd_f32": (_c_int, [_P] * 9 + [_c_int] * 7 + [_c_float, _P]),
    "dmb_spn_gaterecurrent2d_f32": (_c_int, [_P] * 5 + [_c_int] * 6 + [_P]),
    "dmb_spn_gaterecurrent2d_bwd_f32": (_c_int, [_P] * 10 + [_c_int] * 6 + [_P]),
    "dmb_conv2d_k3_multi_f32": (_c_int, [_c_int, _P, _P, _P, _P, _P, _c_int, _c_int, _c_int, _c_int, _P]),
    "dmb_cat_fms_into_f32": (_c_int, [_P, _P, _P, _c_int, _c_int, _c_int, _c_int, _c_int, _HI, _c_int, _c_int, _P]),
    "dmb_correlation1d_f32": (_c_int, [_P, _P, _P, _c_int, _c_int, _c_int, _c_int, _c_int, _c_float, _P]),
    "dmb_copy_window_f32": (_c_int, [_P, _P, _c_ll, _c_int, _c_int, _c_int, _P]),
    "dmb_catconv_pack_weights_f32": (_c_int, [_P, _P, _c_int, _c_int, _c_int, _c_int, _P]),
    "dmb_catconv_finalize_f32": (_c_int, [_P] * 8 + [_c_int] * 8 + [_P]),
    "dmb_catconv_combine_f32": (_c_int, [_P] * 9 + [_c_int] * 7 + [_P]),
    "dmb_conv3d_packed_floats": (_c_ll, [_c_int, _c_int]),
    "dmb_deconv3d_packed_floats": (_c_ll, [_c_int, _c_int]),
    "dmb_conv3d_pack_weights_f32": (_c_int, [_P, _P, _c_int, _c_int, _P]),
    "dmb_deconv3d_pack_weights_f32": (_c_int, [_P, _P, _c_int, _c_int, _P]),
    "dmb_conv3d_k3_f32": (_c_int, [_P, _P, _P, _P, _P, _P] + [_c_int] * 8 + [_P]),
    "dmb_conv3d_k3_c1_f32": (_c_int, [_P, _P, _c_float, _P, _P] + [_c_int] * 6 + [_P]),
    "dmb_deconv3d_k3s2_f32": (_c_int, [_P, _P, _P, _P, _P, _P] + [_c_int] * 8 + [_P, _P]),
    "dmb_zero_columns_f32": (_c_int, [_P, _c_ll, _c_int, _c_int, _P]),
    "dmb_trilinear_ac_f32": (_c_int, [_P, _P] + [_c_int] * 7 + [_P]),
    "dmb_deconv3d_k8s4_c1_f32": (_c_int, [_P, _P, _P] + [_c_int] * 4 + [_P]),
    "dmb_deconv3d_k8s4_c1_soft_argmin_f32": (_c_int, [_P, _P, _P, _P] + [_c_int] * 4 + [_c_float, _HF, _P]),
    "dmb_soft_argmin_f32": (_c_int, [_P, _P, _c_int, _c_int, _c_int, _c_int, _c_float, _c_int, _HF, _P]),
    "dmb_soft_argmin_sampled_f32": (_c_int, [_P, _P, _P, _c_int, _c_int, _c_int, _c_int, _c_float, _c_int, _P]),
    "dmb_local_soft_argmin_f32": (_c_int, [_P, _P, _P] + [_c_int] * 8 + [_c_float, _P]),
    "dmb_trilinear_soft_argmin_f32": (_c_int, [_P, _P] + [_c_int] * 7 + [_c_float, _HF, _P]),
    "dmb_trilinear_ac_soft_argmin_f32": (_c_int, [_P, _P, _P] + [_c_int] * 7 + [_c_float, _HF, _P]),
    "dmb_conf_head_packed_floats": (_c_ll, [_c_int, _c_int]),
    "dmb_conf_head_pack_weights_f32": (_c_int, [_P, _P, _c_int, _c_int, _P]),
    "dmb_conf_head_f32": (_c_int, [_P, _P, _P, _P, _P, _P] + [_c_int] * 5 + [_P]),
    "dmb_conf_gather_f32": (_c_int, [_P, _P, _P] + [_c_int] * 4 + [_P]),
    "dmb_conf_phase_conv2d_f32": (_c_int, [_P] * 6 + [_c_int] * 5 + [_P]),
    "dmb_conf_ring_f32": (_c_int, [_P] * 6 + [_c_int] * 5 + [_P]),
    "dmb_conv2d_packed_floats": (_c_ll, [_c_int, _c_int, _c_int]),
    "dmb_conv2d_pack_weights_f32": (_c_int, [_P, _P, _c_int, _c_int, _c_int, _P]),
    "dmb_conv2d_f32": (_c_int, [_P, _P, _P, _P, _P, _P] + [_c_int] * 12 + [_P]),
    "dmb_avgpool2d_f32": (_c_int, [_P, _P] + [_c_int] * 7 + [_P]),
    "dmb_bilinear_ac_f32": (_c_int, [_P, _P] + [_c_int] * 8 + [_P]),
    "dmb_bilinear_scale_f32": (_c_int, [_P, _P] + [_c_int] * 6 + [_c_float] + [_c_int] * 2 + [_P]),
    "dmb_epe_accum_f64": (_c_int, [_P, _P, _P, _P] + [_c_int] * 5 + [_c_float, _c_float, _P]),
    "dmb_epe_accum_multi_f64": (_c_int, [_c_int, _P, _P, _P, _P] + [_c_int] * 5 + [_c_float, _c_float, _P]),
    "dmb_conv3d_x6_packed_bytes": (_c_ll, [_c_int, _c_int]),
    "dmb_conv3d_x6_pack_weights_f32": (_c_int, [_P, _P, _c_int, _c_int, _P]),
    "dmb_conv3d_k3_x6_f32": (_c_int, [_P, _P, _P, _P, _P, _P] + [_c_int] * 7 + [_P]),
    "dmb_conv3d_pack_dgrad_weights_f32": (_c_int, [_P, _P, _c_int, _c_int, _P]),
    "dmb_conv3d_wgrad_workspace_floats": (_c_ll, [_c_int, _c_int]),
    "dmb_conv3d_k3_wgrad_f32": (_c_int, [_P, _P, _P, _P] + [_c_int] * 6 + [_P]),
    "dmb_conv3d_k3s2_wgrad_f32": (_c_int, [_P, _P, _P, _P] + [_c_int] * 9 + [_P]),
    "dmb_conv2d_wgrad_workspace_floats": (_c_ll, [_c_int, _c_int]),
    "dmb_conv2d_wgrad_f32": (_c_int, [_P, _P, _P, _P] + [_c_int] * 7 + [_P]),
    "dmb_channel_dot_f32": (_c_int, [_P, _P, _P, _P, _c_int, _c_int, _c_ll, _P]),
    "dmb_conv3d_pack_weights_multi_f32": (_c_int, [_P, _c_int, _P]),
    "dmb_cat_first_wgrad_maps_f32": (_c_int, [_P, _P, _P, _c_int, _c_int, _c_int, _c_int, _c_int, _P]),
    "dmb_conv3d_k3_bnstats_partials": (_c_ll, [_c_int] * 6),
    "dmb_conv3d_k3_bnstats_f32": (_c_int, [_P] * 4 + [_c_int] * 6 + [_P]),
    "dmb_bn_train_act_f32": (_c_int, [_P, _P, _c_int] + [_P] * 5 + [_c_float, _c_float] + [_P] * 6 + [_c_int, _c_int, _c_ll, _c_int, _P]),
    "dmb_bn_workspace_doubles": (_c_ll, [_c_int, _c_ll]),
    "dmb_bn_train_stats_f32": (_c_int, [_P, _P, _P, _P, _P, _c_float, _c_float, _P, _P, _P, _P, _P, _c_int, _c_int, _c_ll, _P]),
    "dmb_bn_act_f32": (_c_int, [_P, _P, _P, _P, _P, _c_int, _c_int, _c_ll, _c_int, _P]),
    "dmb_bn_train_fwd_f32": (_c_int, [_P] * 6 + [_c_float, _c_float] + [_P] * 7 + [_c_int, _c_int, _c_ll, _c_int, _P]),
    "dmb_bn_act_bwd_f32": (_c_int, [_P] * 13 + [_c_int, _c_int, _c_ll, _c_int, _c_int, _P]),
    "dmb_cat_fms_bwd_f32": (_c_int, [_P, _P, _P, _c_int, _c_int, _c_int, _c_int, _c_int, _HI, _P]),
    "dmb_dif_fms_bwd_f32": (_c_int, [_P, _P, _P, _c_int, _c_int, _c_int, _c_int, _c_int, _HI, _P]),
    "dmb_soft_argmin_bwd_f32": (_c_int, [_P, _P, _P, _P, _c_int, _c_int, _c_int, _c_int, _c_float, _HF, _P]),
    "dmb_trilinear_ac_soft_argmin_bwd_f32": (_c_int, [_P, _P, _P, _P, _P, _P] + [_c_int] * 7 + [_c_float, _HF, _P]),
    "dmb_trilinear_ac_bwd_f32": (_c_int, [_P, _P, _P] + [_c_int] * 7 + [_P]),
    "dmb_avgpool2d_bwd_f32": (_c_int, [_P, _P] + [_c_int] * 5 + [_P]),
    "dmb_bilinear_ac_bwd_f32": (_c_int, [_P, _P] + [_c_int] * 6 + [_P]),
    "dmb_bilinear_scale_bwd_f32": (_c_int, [_P, _P] + [_c_int] * 6 + [_c_float, _P]),
    "dmb_deconv3d_k8s4_bwd_workspace_doubles": (_c_ll, []),
    "dmb_deconv3d_k8s4_c1_bwd_f32": (_c_int, [_P] * 6 + [_c_int] * 4 + [_P]),
    "dmb_loss_workspace_doubles": (_c_ll, [_c_ll]),
    "dmb_stereo_focal_loss_fwd_f32": (_c_int, [_P, _P, _P, _c_float, _HF, _P, _P, _P] + [_c_int] * 4 + [_c_float] * 5 + [_P]),
    "dmb_stereo_focal_loss_bwd_f32": (_c_int, [_P, _P, _P, _c_float, _HF, _P, _P, _P, _c_float, _P, _P] + [_c_int] * 4
                                      + [_c_float] * 5 + [_P]),
    "dmb_map_loss_fwd_f32": (_c_int, [_P, _P, _P, _P, _c_ll, _c_float, _c_float, _c_int, _P]),
    "dmb_stereo_pad_normalize_f32": (_c_int, [_P, _P] + [_c_int] * 11 + [_HF, _HF, _P]),
    "dmb_stereo_pad_normalize_u8": (_c_int, [_P, _P] + [_c_int] * 11 + [_HF, _HF, _P]),
    "dmb_map_loss_bwd_f32": (_c_int, [_P, _P, _P, _P, _c_float, _P, _c_ll, _c_float, _c_float, _c_int, _P]),
}


def header_symbols():
    """Every entry point declared in include/dmb_hip.h (used by the CPU-side export test)."""
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dmb_[a-z0-9_]+)\s*\(", text)))


class DmbLibraryError(RuntimeError):
    pass


_lib = None


def load():
    """Load libdmb_hip.so (once).  Raises DmbLibraryError -- never falls back to another implementation."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DmbLibraryError(
            "libdmb_hip.so is not built (%s). Run `python -m densematchingbenchmark_amd.build` "
            "(or __graft_entry__.build()); there is no CPU/PyTorch fallback for this path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.dmb_abi_version() != ABI_VERSION:
        raise DmbLibraryError("%s reports ABI version %d, this binding is written for %d: rebuild it (python -m "
                              "densematchingbenchmark_amd.build)" % (LIB_PATH, lib.dmb_abi_version(), ABI_VERSION))
    tagged = DEV_BUILD and os.environ["DMB_LIB"] != "dev"     # (a build-time experiment: its defines are not known here)
    if not tagged:
        from . import build
        want, got = build.sources_digest(dev=DEV_BUILD), (lib.dmb_build_id() or b"").decode()
        if want != got:
            raise DmbLibraryError("%s was built from other sources than the ones next to it (build id %s..., sources %s...): "
                                  "rebuild it (python -m densematchingbenchmark_amd.build)" % (LIB_PATH, got[:12], want[:12]))
    if DEV_BUILD:
        lib.dmb_dev_set_option.restype = None
        lib.dmb_dev_set_option.argtypes = [_c_int, _c_int]
    _lib = lib
    return lib


_shim = None
_shim_state = "untried"      # "untried" | "loaded" | the reason it is not in use


def shim():
    """The thin torch extension over the same C ABI (csrc/torch_shim.cpp -> lib/_dmb_torch_shim.so: unwraps tensors, takes the
    current HIP stream from c10, raises DmbLibraryError on a non-zero code) for the entry points on the per-pair launch path, or
    None -- then the ctypes table above binds them, as it binds every other entry point: the SAME functions of the SAME library,
    only more interpreter time per launch (profiles/r06_binding_overhead.log).  Refused (-> None) unless it was built from the
    csrc/torch_shim.cpp / dmb_hip.h / torch next to it and linked against the library that is loaded.  DMB_SHIM=0: never;
    DMB_SHIM=require: raise instead of returning None."""
    global _shim, _shim_state
    if _shim_state != "untried":
        return _shim
    mode = os.environ.get("DMB_SHIM", "auto")
    try:
        if mode == "0":
            raise DmbLibraryError("disabled by DMB_SHIM=0")
        if DEV_BUILD:
            raise DmbLibraryError("the development library is bound through ctypes only")
        lib = load()
        path = os.path.join(_PKG, "lib", "_dmb_torch_shim.so")
        if not os.path.exists(path):
            raise DmbLibraryError("%s is not built (python -m densematchingbenchmark_amd.build --shim)" % path)
        import importlib.util
        from . import build
        spec = importlib.util.spec_from_file_location("_dmb_torch_shim", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        if mod.build_id() != build.shim_digest():
            raise DmbLibraryError("lib/_dmb_torch_shim.so was not built from the csrc/torch_shim.cpp, dmb_hip.h and torch next to it")
        if mod.library_build_id() != lib.dmb_build_id().decode() or mod.abi_version() != ABI_VERSION:
            raise DmbLibraryError("lib/_dmb_torch_shim.so is bound to another libdmb_hip.so than the one loaded")
        mod.set_error_class(DmbLibraryError)
        _shim, _shim_state = mod, "loaded"
    except Exception as e:  # noqa: BLE001  (any failure to bring the shim up leaves the ctypes binding of the same C ABI)
        _shim, _shim_state = None, "%s: %s" % (type(e).__name__, e)
        if mode == "require":
            raise
    return _shim


def shim_state():
    shim()
    return _shim_state


def check(code, what):
    if code != 0:
        msg = load().dmb_last_error()
        raise DmbLibraryError("%s failed with code %d (%s)" % (what, code, msg.decode() if msg else ""))


def dev_ptr(t, name="tensor", allow_none=False):
    """Device pointer of a contiguous FP32 HIP tensor; raises for anything else (no silent CPU path)."""
    if t is None:
        if allow_none:
            return None
        raise DmbLibraryError("%s is None" % name)
    if not isinstance(t, torch.Tensor):
        raise DmbLibraryError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise DmbLibraryError(
            "%s lives on %s: the dmb HIP path only runs on a GPU (cuda/hip) tensor and has no CPU fallback" % (name, t.device))
    if t.dtype != torch.float32 and t.dtype != torch.float64 and t.dtype != torch.int64:
        raise DmbLibraryError("%s has dtype %s; FP32 expected" % (name, t.dtype))
    if not t.is_contiguous():
        raise DmbLibraryError("%s must be contiguous" % name)
    if t.device.index != torch.cuda.current_device():   # every operand, not only the one the stream is taken from
        raise DmbLibraryError("%s lives on cuda:%d but the current device is cuda:%d" % (name, t.device.index, torch.cuda.current_device()))
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    """The caller's current HIP stream on ``device``.  Kernels launch on the calling thread's CURRENT device: a tensor that
    lives on another one is refused here instead of failing inside the launch (or worse, on a foreign stream)."""
    if device is not None and getattr(device, "index", None) is not None and device.index != torch.cuda.current_device():
        raise DmbLibraryError("operand on cuda:%d but the current device is cuda:%d: wrap the call in torch.cuda.device(%d) "
                              "(one process per GPU sets it once)" % (device.index, torch.cuda.current_device(), device.index))
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def host_ints(values):
    arr = (ctypes.c_int * len(values))(*[int(v) for v in values])
    return arr


def host_floats(values):
    arr = (ctypes.c_float * len(values))(*[float(v) for v in values])
    return arr
