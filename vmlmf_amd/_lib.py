"""ctypes binding of the C ABI in include/vmlmf_hip.h (libvmlmf_hip.so, built in-tree by csrc/Makefile).

There is no CPU fallback: if the library is missing, or a call fails, this raises.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# VMLMF_LIB: an instrumented build of the same library (tools/microbench), never a different implementation
LIB_PATH = os.environ.get("VMLMF_LIB") or os.path.join(_HERE, "lib", "libvmlmf_hip.so")
CSRC = os.path.join(_HERE, "csrc")

MAX_G = 2
NKERNELS = 13
HEAD_MAX_CLASSES = 32
MAX_TENSORS = 48
V1_CELL, V2_GROUP_CELL, V3_LM, V4_LM_GROUP, V5_LMF_CELL, V6_GROUP_NOVM = 1, 2, 3, 4, 5, 6
ABI_VERSION = 13
GUARD_WORDS, GUARD_GO, GUARD_SKIPPED = 72, 64, 66
ADAM_FIRST, ADAM_LAST = 1, 2
DT_F32, DT_BF16 = 0, 1
DTYPES = {"f32": 0, "fp32": 0, "float32": 0, "bf16": 1, "bfloat16": 1}
E_BADARG, E_SHAPE, E_UNSUPPORTED, E_WORKSPACE, E_COMM, E_PROTOCOL = -1, -2, -3, -4, -5, -6
SUM, AVG = 0, 1
COMM_ID_BYTES = 128
P2P_HANDLE_BYTES = 64
P2P_MAX_RANKS = 8

_fp = ctypes.POINTER(ctypes.c_float)


class Desc(ctypes.Structure):
    _fields_ = [("variant", ctypes.c_int32), ("B", ctypes.c_int32), ("T", ctypes.c_int32),
                ("I", ctypes.c_int32), ("H", ctypes.c_int32), ("w_rank", ctypes.c_int32),
                ("g", ctypes.c_int32), ("u_ranks", ctypes.c_int32 * MAX_G),
                ("time_major", ctypes.c_int32), ("training", ctypes.c_int32), ("dtype", ctypes.c_int32)]


class Params(ctypes.Structure):
    _fields_ = [("dia_x", ctypes.c_void_p), ("dia_h", ctypes.c_void_p), ("u_x", ctypes.c_void_p),
                ("v_x", ctypes.c_void_p), ("b_x", ctypes.c_void_p), ("b_h", ctypes.c_void_p),
                ("u_h", ctypes.c_void_p * MAX_G), ("v_h", ctypes.c_void_p * MAX_G),
                ("w_gate", ctypes.c_void_p * 4), ("u_gate", ctypes.c_void_p * 4), ("b_gate", ctypes.c_void_p * 4)]


class Head(ctypes.Structure):
    _fields_ = [("classes", ctypes.c_int32), ("weight", ctypes.c_void_p), ("bias", ctypes.c_void_p),
                ("logits", ctypes.c_void_p), ("dlogits", ctypes.c_void_p), ("dweight", ctypes.c_void_p),
                ("dbias", ctypes.c_void_p)]


class Ce(ctypes.Structure):
    """vmlmf_ce (ABI 10): the cross-entropy criterion riding on the classifier's logits in the forward launch."""
    _fields_ = [("target", ctypes.c_void_p), ("ignore_index", ctypes.c_int64), ("loss", ctypes.c_void_p), ("nvalid", ctypes.c_void_p),
                ("lse", ctypes.c_void_p), ("dlogits_unit", ctypes.c_void_p), ("ticket", ctypes.c_void_p)]


class Dropout(ctypes.Structure):
    """vmlmf_dropout (ABI 11): dropout of a layer's output inside the layer's launches."""
    _fields_ = [("p", ctypes.c_float), ("site", ctypes.c_int32), ("state", ctypes.c_void_p), ("y_dropped", ctypes.c_void_p)]


class Extra(ctypes.Structure):
    _fields_ = [("packed", ctypes.c_void_p), ("head", ctypes.POINTER(Head)), ("ce", ctypes.POINTER(Ce)), ("drop", ctypes.POINTER(Dropout))]


class StackLayer(ctypes.Structure):
    """vmlmf_stack_layer (ABI 7): one layer of a stack run by the wavefront launches."""
    _fields_ = [("desc", Desc), ("params", ctypes.POINTER(Params)), ("h0", ctypes.c_void_p), ("c0", ctypes.c_void_p),
                ("y", ctypes.c_void_p), ("hT", ctypes.c_void_p), ("cT", ctypes.c_void_p), ("reserve", ctypes.c_void_p),
                ("dhT", ctypes.c_void_p), ("dcT", ctypes.c_void_p), ("dh0", ctypes.c_void_p), ("dc0", ctypes.c_void_p),
                ("grads", ctypes.POINTER(Params)), ("drop", ctypes.POINTER(Dropout))]


STACK_MAX = 4


class Sizes(ctypes.Structure):
    _fields_ = [("workspace_bytes", ctypes.c_size_t), ("reserve_bytes", ctypes.c_size_t),
                ("rows_per_wg", ctypes.c_int32), ("threads_per_wg", ctypes.c_int32),
                ("workgroups", ctypes.c_int32), ("kx", ctypes.c_int32), ("kh", ctypes.c_int32)]


class TensorList(ctypes.Structure):
    _fields_ = [("param", ctypes.c_void_p * MAX_TENSORS), ("grad", ctypes.c_void_p * MAX_TENSORS),
                ("numel", ctypes.c_int64 * MAX_TENSORS), ("state_offset", ctypes.c_int64 * MAX_TENSORS),
                ("step_index", ctypes.c_int32 * MAX_TENSORS), ("count", ctypes.c_int32)]


# every symbol include/vmlmf_hip.h declares: (restype, argtypes)
_vp, _sz, _i = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
SYMBOLS = {
    "vmlmf_abi_version": (_i, []),
    "vmlmf_build_info": (ctypes.c_char_p, []),
    "vmlmf_last_error": (ctypes.c_char_p, []),
    "vmlmf_tune": (_i, [ctypes.c_char_p, _i]),
    "vmlmf_check_status": (_i, []),
    "vmlmf_stack_query": (_i, [_i, _vp, _vp, _vp]),
    "vmlmf_stack_dropout_fused": (_i, [_i, _vp]),
    "vmlmf_stack_forward": (_i, [_i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "vmlmf_stack_backward": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "vmlmf_query": (_i, [ctypes.POINTER(Desc), ctypes.POINTER(Sizes)]),
    "vmlmf_seq_forward": (_i, [ctypes.POINTER(Desc), ctypes.POINTER(Params), _vp, _vp, _vp, _vp, _vp, _vp,
                               _vp, _vp, _sz, _vp]),
    "vmlmf_pack_bytes": (_i, [ctypes.POINTER(Desc), ctypes.POINTER(_sz)]),
    "vmlmf_pack_params": (_i, [ctypes.POINTER(Desc), ctypes.POINTER(Params), _vp, _vp]),
    "vmlmf_seq_forward_packed": (_i, [ctypes.POINTER(Desc), ctypes.POINTER(Params), _vp, _vp, _vp, _vp, _vp, _vp,
                                      _vp, _vp, _sz, _vp, _vp]),
    "vmlmf_seq_backward_packed": (_i, [ctypes.POINTER(Desc), ctypes.POINTER(Params), _vp, _vp, _vp, _vp, _vp, _vp,
                                       _vp, _vp, _vp, _vp, _vp, ctypes.POINTER(Params), _vp, _sz, _vp, _vp]),
    "vmlmf_tune_generation": (_i, []),
    "vmlmf_tune_get": (_i, [ctypes.c_char_p, ctypes.POINTER(_i)]),
    "vmlmf_seq_forward_ex": (_i, [ctypes.POINTER(Desc), ctypes.POINTER(Params), _vp, _vp, _vp, _vp, _vp, _vp,
                                  _vp, _vp, _sz, _vp, ctypes.POINTER(Extra)]),
    "vmlmf_seq_backward_ex": (_i, [ctypes.POINTER(Desc), ctypes.POINTER(Params), _vp, _vp, _vp, _vp, _vp, _vp,
                                   _vp, _vp, _vp, _vp, _vp, ctypes.POINTER(Params), _vp, _sz, _vp, ctypes.POINTER(Extra)]),
    "vmlmf_seq_backward": (_i, [ctypes.POINTER(Desc), ctypes.POINTER(Params), _vp, _vp, _vp, _vp, _vp, _vp,
                                _vp, _vp, _vp, _vp, _vp, ctypes.POINTER(Params), _vp, _sz, _vp]),
    "vmlmf_head_forward": (_i, [_i, _i, _i, _vp, ctypes.c_longlong, _vp, _vp, _vp, _vp]),
    "vmlmf_head_backward": (_i, [_i, _i, _i, _vp, ctypes.c_longlong, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vmlmf_ce_forward": (_i, [_i, _i, _vp, _vp, ctypes.c_int64, _vp, _vp, _vp, _vp, _vp]),
    "vmlmf_ce_backward": (_i, [_i, _i, _vp, _vp, ctypes.c_int64, _vp, _vp, _vp, _vp, _vp]),
    "vmlmf_nll_forward": (_i, [_i, _i, _vp, _vp, ctypes.c_float, _vp, _vp, _vp, _vp]),
    "vmlmf_nll_backward": (_i, [_i, _i, _vp, _vp, ctypes.c_float, _vp, _vp, _vp, _vp]),
    "vmlmf_nll_grad_scratch_floats": (_sz, [_i, _i]),
    "vmlmf_nll_forward_grad": (_i, [_i, _i, _vp, _vp, _vp, ctypes.c_float, _vp, _vp, _vp, _vp, _vp]),
    "vmlmf_embed_backward_scratch_bytes": (_sz, [_i, _i]),
    "vmlmf_embed_backward": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "vmlmf_dropout_fused": (_i, [ctypes.POINTER(Desc)]),
    "vmlmf_dropout_advance": (_i, [_vp, _vp, _vp]),
    "vmlmf_dropout_apply": (_i, [ctypes.c_int64, _i, _vp, _vp, ctypes.c_float, _vp, _i, _vp]),
    "vmlmf_dropout_factors": (_i, [ctypes.POINTER(Desc), ctypes.c_int64, _i, ctypes.c_float, _vp, _i, _vp, _vp]),
    "vmlmf_embed_dropout_forward": (_i, [_i, _i, _i, _vp, _vp, _vp, ctypes.c_float, _vp, _i, _vp]),
    "vmlmf_embed_dropout_backward": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _sz, ctypes.c_float, _vp, _i, _vp]),
    "vmlmf_transpose": (_i, [_i, _i, _vp, _vp, _vp]),
    "vmlmf_adam_step": (_i, [ctypes.POINTER(TensorList), _vp, _vp, _vp, ctypes.c_float, ctypes.c_float,
                             ctypes.c_float, ctypes.c_float, ctypes.c_float, _vp]),
    "vmlmf_adam_step_guarded": (_i, [ctypes.POINTER(TensorList), _vp, _vp, _vp, ctypes.c_float, ctypes.c_float,
                                     ctypes.c_float, ctypes.c_float, ctypes.c_float, _vp, _vp]),
    "vmlmf_adam_step_ex": (_i, [ctypes.POINTER(TensorList), _vp, _vp, _vp, ctypes.c_float, ctypes.c_float,
                                ctypes.c_float, ctypes.c_float, ctypes.c_float, _vp, _i, _vp]),
    "vmlmf_sgd_clip_step": (_i, [ctypes.POINTER(TensorList), ctypes.c_float, ctypes.c_float, _vp, _vp, _vp]),
    "vmlmf_comm_unique_id": (_i, [_vp]),
    "vmlmf_comm_init": (_i, [ctypes.POINTER(_vp), _i, _i, _vp]),
    "vmlmf_comm_count": (_i, [_vp, ctypes.POINTER(_i)]),
    "vmlmf_comm_destroy": (_i, [_vp]),
    "vmlmf_flat_allreduce": (_i, [_vp, _sz, _i, _vp, _vp]),
    "vmlmf_flat_allreduce_group": (_i, [_i, ctypes.POINTER(_vp), ctypes.POINTER(_sz), _i, _vp, _vp]),
    "vmlmf_p2p_create": (_i, [ctypes.POINTER(_vp), _i, _i, _sz, _vp]),
    "vmlmf_p2p_connect": (_i, [_vp, _vp]),
    "vmlmf_p2p_allreduce": (_i, [_vp, _vp, _sz, _i, _vp]),
    "vmlmf_p2p_destroy": (_i, [_vp]),
    "vmlmf_profile_enable": (_i, [_i]),
    "vmlmf_profile_read": (_i, [_fp, ctypes.POINTER(ctypes.c_int32), _i]),
    "vmlmf_kernel_name": (ctypes.c_char_p, [_i]),
}

_lib = None


class VmlmfError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"vmlmf_hip error {code}: {msg}")
        self.code = code


def build(force=False, jobs=8):
    """Compile every HIP source for gfx950 into vmlmf_amd/lib/libvmlmf_hip.so (hipcc cross-compiles)."""
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", CSRC, f"-j{jobs}"], check=True, stdout=subprocess.DEVNULL)
    return LIB_PATH


def build_torch_binding():
    """g++ csrc/torch_binding.cpp -> vmlmf_amd/lib/libvmlmf_torch.so (TORCH_LIBRARY "vmlmf": C++ autograd functions over the C ABI)."""
    subprocess.run(["make", "-C", CSRC, "torch"], check=True, stdout=subprocess.DEVNULL)
    return os.path.join(_HERE, "lib", "libvmlmf_torch.so")


def lib():
    """The loaded shared library.  Raises if it has not been built: there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `make -C {CSRC}` (or __graft_entry__.build()). "
                "vmlmf_amd has no CPU / PyTorch fallback for the hot path.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)  # AttributeError if the export is missing
            fn.restype, fn.argtypes = res, args
        if handle.vmlmf_abi_version() != ABI_VERSION:
            raise RuntimeError("libvmlmf_hip.so ABI version mismatch: rebuild")
        _lib = handle
        if ranks_share_a_device() and "VMLMF_WRIDE" not in os.environ:
            # the riding weight-gradient workers wait for row workgroups of their own launch; when several processes queue
            # launches on ONE device a resident worker can starve the rows it waits for: such jobs
            # start on the stand-alone weight-gradient kernel instead of finding out through a failed step
            handle.vmlmf_tune(b"wride", 0)
            import sys
            print("vmlmf_amd: LOCAL_WORLD_SIZE says several local ranks share a device - the weight-gradient workers that ride on the "
                  "backward launch are switched off (stand-alone kernel; VMLMF_WRIDE=1 keeps them)", file=sys.stderr)
    return _lib


def ranks_share_a_device():
    """True when the launcher's environment says more LOCAL ranks than visible devices (torchrun: LOCAL_WORLD_SIZE; WORLD_SIZE
    counts the ranks of every node and is not looked at).  A launcher that masks one device per rank (HIP / ROCR / CUDA_VISIBLE_DEVICES
    naming a single device) gives every rank a device of its own: not shared."""
    try:
        local = int(os.environ.get("LOCAL_WORLD_SIZE") or 1)
    except ValueError:
        return False
    if local <= 1:
        return False
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None and len([d for d in v.split(",") if d.strip()]) == 1:
            return False
    import torch
    return local > max(torch.cuda.device_count(), 1)


def tune(key, value):
    """Kernel-selection switches of the library (include/vmlmf_hip.h: vmlmf_tune).  Cached descriptors are dropped."""
    check(lib().vmlmf_tune(key.encode(), int(value)))
    from . import functional
    functional._DESC_CACHE.clear()


def check_status():
    """Raises VmlmfError(E_PROTOCOL) if a launch that already ran on the current device gave up a bounded wait for another
    workgroup (its results are NaN); call after a synchronisation to learn about the launches before it."""
    check(lib().vmlmf_check_status())


def check(rc):
    if rc != 0:
        raise VmlmfError(rc, lib().vmlmf_last_error().decode())


def make_desc(variant, B, T, I, H, w_rank, u_ranks, g=1, time_major=False, training=True, dtype=0):
    d = Desc()
    d.variant, d.B, d.T, d.I, d.H, d.w_rank = variant, B, T, I, H, w_rank
    d.g = g
    ur = list(u_ranks) if isinstance(u_ranks, (list, tuple)) else [u_ranks]
    for i in range(MAX_G):
        d.u_ranks[i] = int(ur[i]) if i < len(ur) else 0
    d.time_major = 1 if time_major else 0
    d.training = 1 if training else 0
    d.dtype = DTYPES[dtype] if isinstance(dtype, str) else int(dtype)
    return d


def tune_get(key):
    """Current value of a vmlmf_tune switch ("wride": 0 while the riding workers are off or have tripped)."""
    v = ctypes.c_int(0)
    check(lib().vmlmf_tune_get(key.encode(), ctypes.byref(v)))
    return v.value


def query(desc):
    s = Sizes()
    check(lib().vmlmf_query(ctypes.byref(desc), ctypes.byref(s)))
    return s


# ---- cheap host-side access to torch's current HIP stream / device ------------------------------------------------
# torch.cuda.current_stream(dev).cuda_stream and `with torch.cuda.device(dev)` cost ~7 us and ~5 us of Python per use;
# a layer's forward + backward makes six library calls, and eager mode is host-bound at the headline shape.  The raw
# accessors below are what those wrappers call underneath; if a torch build lacks them the public API is used.
def raw_stream(dev):
    """ctypes handle of torch's current stream on `dev` (a torch.device with an index)."""
    import torch
    try:
        return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(dev.index))
    except (AttributeError, TypeError):
        return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


class on_device:
    """`with on_device(dev):` -- make `dev` the current HIP device for the library call, switching only when it is not
    already (the common case costs one integer compare)."""
    __slots__ = ("idx", "prev")

    def __init__(self, dev):
        self.idx = dev.index
        self.prev = -1

    def __enter__(self):
        import torch
        cur = torch.cuda.current_device()
        if cur != self.idx:
            torch.cuda.set_device(self.idx)
            self.prev = cur
        return self

    def __exit__(self, *exc):
        if self.prev >= 0:
            import torch
            torch.cuda.set_device(self.prev)
        return False
