"""ctypes binding of libgrlhip.so (C ABI: include/grlhip.h).

The library is built in-tree by ``__graft_entry__.build()`` /
``generativerl_preview_amd/csrc/build.sh``.  There is no fallback: if the shared object is
missing, or a compute entry point is called without a HIP device, this raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GRLHIP_LIB: developer override used by the kernel A/B scripts (scripts/ab_*.sh)
LIB_PATH = os.environ.get("GRLHIP_LIB") or os.path.join(_HERE, "libgrlhip.so")

MAX_HIDDEN = 8
F32, BF16 = 0, 1
NET_RESNET, NET_CONCAT_MLP, NET_QNET = 0, 1, 2
COEF_STRIDE = 12


class NetDesc(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("x_dim", ctypes.c_int32), ("out_dim", ctypes.c_int32),
                ("cond_dim", ctypes.c_int32), ("t_dim", ctypes.c_int32), ("cond_hidden", ctypes.c_int32),
                ("t_cond_hidden", ctypes.c_int32), ("n_hidden", ctypes.c_int32),
                ("hidden", ctypes.c_int32 * MAX_HIDDEN)]


class GrlHipError(RuntimeError):
    pass


_lib = None

_vp, _i32, _i64, _sz, _f = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_size_t, ctypes.c_float

# name -> (restype, argtypes); kept in one table so tests can check it against include/grlhip.h
PROTOTYPES = {
    "grlhip_last_error": (ctypes.c_char_p, []),
    "grlhip_device_available": (ctypes.c_int, []),
    "grlhip_plan_create": (ctypes.c_int, [ctypes.POINTER(NetDesc), ctypes.POINTER(_vp)]),
    "grlhip_plan_destroy": (None, [_vp]),
    "grlhip_plan_num_params": (ctypes.c_int, [_vp]),
    "grlhip_plan_param_info": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_char_p, _sz,
                                              ctypes.POINTER(_i32), ctypes.POINTER(_i32)]),
    "grlhip_plan_set_fourier_grad": (ctypes.c_int, [_vp, ctypes.c_int]),
    "grlhip_plan_set_backward_split": (ctypes.c_int, [_vp, ctypes.c_int]),
    "grlhip_plan_schedule_check": (ctypes.c_int, [_vp, ctypes.c_int]),
    "grlhip_plan_program_info": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.POINTER(_i32), ctypes.POINTER(_i32)]),
    "grlhip_bf16_fused_kernel": (ctypes.c_int, [_vp, _vp, ctypes.c_char_p, ctypes.c_size_t]),
    "grlhip_packed_bytes": (_sz, [_vp, ctypes.c_int]),
    "grlhip_pack_weights": (ctypes.c_int, [_vp, ctypes.POINTER(_vp), ctypes.c_int, ctypes.c_int, _vp, _vp]),
    "grlhip_workspace_bytes": (_sz, [_vp, _vp, _i64, ctypes.c_int]),
    "grlhip_net_forward": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, _vp, _vp, _vp, _i64, _vp, _sz, _vp]),
    "grlhip_energy_grad": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, _vp, _vp, _f, _vp, _i64, _vp, _sz, _vp]),
    "grlhip_sample_ode": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _vp, _vp,
                                         _vp, _sz, _vp]),
    "grlhip_sample_fixed": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp,
                                           _vp, _vp, _vp, _vp, _sz, _vp]),
    "grlhip_static_scratch_bytes": (_sz, []),
    "grlhip_sample_ode_static": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp]),
    "grlhip_sample_sde": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _vp,
                                         _vp, _vp, _sz, _vp]),
    "grlhip_sample_stages": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _vp, _vp,
                                            _vp, _sz, _vp]),
    "grlhip_loss_workspace_bytes": (_sz, [_vp, _i64, ctypes.c_int]),
    "grlhip_grad_numel": (_i64, [_vp]),
    "grlhip_grad_offset": (_i64, [_vp, ctypes.c_int]),
    "grlhip_loss_fwd_bwd": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                           _i64, _vp, _sz, _vp]),
    "grlhip_net_backward": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _sz, _vp]),
    "grlhip_sample_ode_backward": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp,
                                                  _vp, _sz, _vp]),
    "grlhip_sample_stages_backward": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp,
                                                  _vp, _sz, _vp]),
    "grlhip_adam_step": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                        ctypes.c_float, ctypes.c_float, _i64, _vp]),
    "grlhip_adam_step_counted": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                                ctypes.c_float, ctypes.c_float, _vp, ctypes.c_int, _vp]),
    "grlhip_cep_loss_fwd_bwd": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp,
                                               _i64, _vp, _sz, _vp]),
}


def lib():
    """The loaded library; raises GrlHipError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GrlHipError(
                "libgrlhip.so is missing (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or generativerl_preview_amd/csrc/build.sh -- there is no CPU fallback" % LIB_PATH)
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def check(rc):
    if rc != 0:
        raise GrlHipError("grlhip error %d: %s" % (rc, lib().grlhip_last_error().decode()))


def device_available():
    return bool(lib().grlhip_device_available())
