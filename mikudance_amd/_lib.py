"""ctypes binding of libmdance_hip.so (the C ABI declared in include/mdance_hip.h).

There is NO fallback: if the shared library is missing the first op call raises.  Build it with
`python -c "import __graft_entry__ as g; g.build()"` or `make -C mikudance_amd/csrc`.
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_long, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmdance_hip.so")

P = c_void_p
SIGNATURES = {
    "md_version": (c_int, []),
    "md_last_error": (c_char_p, []),
    "md_gemm_f16": (c_int, [P, c_int, P, P, c_int, c_int, c_int, c_int, P, P, c_int, P, c_int, c_int, c_int, c_int, P]),
    "md_conv3x3_nhwc_f16": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_int, P, c_int,
                                    c_int, c_int, P]),
    "md_conv3x3_pad_nhwc_f16": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_int, P,
                                        c_int, c_int, c_int, P]),
    "md_conv_nhwc_f16": (c_int, [P, c_int, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_int, P,
                                 c_int, c_int, c_int, P]),
    "md_set_cu_limit": (c_int, [c_int]),
    "md_gemm_plan": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "md_conv3x3_plan": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "md_softmax_rows_f16": (c_int, [P, c_int, c_int, c_int, c_float, P]),
    "md_groupnorm_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "md_groupnorm_nhwc_f16": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_float, c_int, P, c_size_t, P]),
    "md_groupnorm_ld_nhwc_f16": (c_int, [P, c_int, P, P, P, c_int, c_int, c_int, c_int, c_float, c_int, P, c_size_t, P]),
    "md_instnorm_spade_ld_f16": (c_int, [P, c_int, P, P, c_int, c_int, c_int, c_float, P]),
    "md_layernorm_f16": (c_int, [P, P, P, P, P, P, c_int, c_int, c_float, c_int, c_int, c_int, c_int, P]),
    "md_gemm_ln_plan": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "md_gemm_ln_f16": (c_int, [P, c_int, P, P, P, c_int, c_int, c_int, c_int, c_float, P, c_int, c_int, c_int, P]),
    "md_groupnorm_table_f16": (c_int, [P, c_int, P, P, c_int, c_int, c_int, c_int, c_float, P, P, c_size_t, P]),
    "md_gemm_affine_plan": (c_int, [c_int, c_int, c_int, c_int]),
    "md_gemm_affine_f16": (c_int, [P, c_int, P, c_int, P, P, c_int, c_int, c_int, c_int, P, P]),
    "md_instnorm_spade_f16": (c_int, [P, P, P, c_int, c_int, c_int, c_float, P]),
    "md_attention_fwd_f16": (c_int, [P, c_int, P, c_int, P, c_int, P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int,
                                     c_float, P]),
    "md_temporal_attention_fwd_f16": (c_int, [P, c_int, P, c_int, P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int,
                                              c_float, P]),
    "md_pack_nhwc_f16": (c_int, [P, c_int, P, c_int, c_int, c_long, c_long, c_long, c_long, c_long, c_int, c_int, c_int,
                                 c_int, c_int, c_int, c_int, P]),
    "md_unpack_nhwc_f16": (c_int, [P, c_int, P, c_int, c_int, c_int, c_long, c_long, c_long, c_long, c_long, c_int, c_int,
                                   c_int, P]),
    "md_concat_channels_f16": (c_int, [P, c_int, P, c_int, P, c_long, P]),
    "md_window_accumulate": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "md_cfg_ddim_step": (c_int, [P, P, P, c_int, c_int, c_int, c_float, c_float, c_float, P]),
    "md_cfg_ddim_step_eta": (c_int, [P, P, P, P, c_int, c_int, c_int, c_float, c_float, c_float, c_float, P]),
}

_lib = None


class MdanceHipError(RuntimeError):
    pass


def load():
    """Load (once) and type the library.  Raises if it has not been built: the product path never falls back."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MdanceHipError(
                f"{LIB_PATH} is missing: the HIP extension has not been built (run __graft_entry__.build()). "
                "mikudance_amd has no CPU fallback.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


class Profiler:
    """Optional per-launch HIP-event timing (bench.py): each C-ABI call is bracketed by two events recorded on the
    stream the kernel is enqueued on (torch's current stream), tagged with its algorithmic FLOPs / bytes."""

    def __init__(self):
        self.enabled = False
        self.records = []

    def start(self):
        self.records = []
        self.enabled = True

    def stop(self):
        self.enabled = False

    def summary(self):
        """label -> dict(count, ms, flops, bytes); call after torch.cuda.synchronize()."""
        out = {}
        for label, flops, nbytes, e0, e1 in self.records:
            d = out.setdefault(label, dict(count=0, ms=0.0, flops=0.0, bytes=0.0))
            d["count"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += flops
            d["bytes"] += nbytes
        return out


PROFILER = Profiler()


def call(name, *args, meta=None):
    lib = load()
    if PROFILER.enabled and meta is not None:
        import torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(lib, name)(*args)
        e1.record()
        PROFILER.records.append((meta[0], meta[1], meta[2], e0, e1))
    else:
        rc = getattr(lib, name)(*args)
    if rc != 0:
        raise MdanceHipError(f"{name} failed ({rc}): {lib.md_last_error().decode()}")
