"""ctypes binding of libbde2vid.so (C ABI declared in include/bde2vid.h).

The library is built in-tree by `make` / `__graft_entry__.build()`.  There is no
fallback: if it is missing, `lib()` raises and nothing computes.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('BDE_LIB_PATH') or os.path.join(_HERE, 'libbde2vid.so')   # override: A/B benchmarking of two builds

MAX_LEVELS = 8
MAX_FRAMES = 8
ABI_VERSION = 4          # include/bde2vid.h; lib() refuses a library built from another revision of the header
ERR_RANGE = -5           # BDE_ERR_RANGE


class BdeConfig(C.Structure):
    _fields_ = [('num_bins', C.c_int32), ('basechannels', C.c_int32), ('num_encoders', C.c_int32),
                ('ks', C.c_int32), ('num_heads', C.c_int32), ('frame_num', C.c_int32),
                ('q_idx', C.c_int32), ('activation', C.c_int32),
                ('depths', C.c_int32 * MAX_LEVELS), ('buffer_index', C.c_int32 * MAX_FRAMES),
                ('recurrent_type', C.c_int32), ('use_rc', C.c_int32), ('skip_concat', C.c_int32),
                ('norm', C.c_int32), ('num_res_blocks', C.c_int32)]


_P = C.c_void_p
_I = C.c_int32
_L = C.c_int64
_PP = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); mirrors include/bde2vid.h one to one
SIGNATURES = {
    'bde_last_error': (C.c_char_p, []),
    'bde_abi_version': (_I, []),
    'bde_create': (_I, [C.POINTER(BdeConfig), _PP]),
    'bde_destroy': (None, [_P]),
    'bde_load_weight': (_I, [_P, C.c_char_p, _P, C.POINTER(_L), _I]),
    'bde_finalize_weights': (_I, [_P]),
    'bde_packed_numel': (_L, [_P]),
    'bde_packed_ptr': (_P, [_P]),
    'bde_alloc_packed': (_I, [_P]),
    'bde_forward': (_I, [_P, _PP, _I, _I, _I, _I, _PP, _P]),
    'bde_get_intermediate': (_I, [_P, C.c_char_p, _P, _L, _P]),
    'bde_split_begin': (_I, [_P, _PP, _I, _I, _I, _I, _P]),
    'bde_split_sweep': (_I, [_P, _I, _I, _P]),
    'bde_split_attend': (_I, [_P, _I, _P]),
    'bde_split_decode': (_I, [_P, _PP, _P]),
    'bde_split_buffer': (_I, [_P, C.c_char_p, _I, _I, C.POINTER(C.c_void_p), C.POINTER(_L)]),
    'bde_set_tuning': (_I, [_P, C.c_char_p, _L]),
    'bde_wait_outputs': (_I, [_P, _P]),
    'bde_get_info': (_I, [_P, C.c_char_p, C.POINTER(_L)]),
    'bde_debug_occupancy': (_I, [C.c_char_p]),
    'bde_debug_conv_shape': (_I, [_I, _I, _I, _I, _I, C.POINTER(C.c_int32)]),
    'bde_debug_split': (C.c_float, [C.POINTER(C.c_float), C.c_int64, _I, C.c_float, C.POINTER(C.c_uint16)]),
    'bde_debug_token_stamps': (_I, [_P, C.POINTER(_L), _I]),
    'bde_profile_reset': (_I, [_P, _I]),
    'bde_profile_names': (_I, [_P, C.c_char_p, _L]),
    'bde_profile_get': (_I, [_P, C.c_char_p, C.POINTER(C.c_double), C.POINTER(_L)]),
    'bde_voxelize': (_I, [_P, _P, _P, _P, _L, _I, _I, _I, _P, _P, _P]),
    'bde_voxelize_batch': (_I, [_P, _P, _P, _P, _P, _I, _L, _I, _I, _I, _P, _P, _P]),
    'bde_voxelize_events': (_I, [_P, _P, _P, _P, _P, _I, _L, _I, _I, _I, _P, _P, _P]),
    'bde_voxelize_event_ranges': (_I, [_P, _P, _P, _P, _L, _P, _P, _I, _L, _I, _I, _I, _P, _P, _P]),
    'bde_voxel_method': (_I, [_I]),
    'bde_find_ts_index': (_I, [_P, _L, _P, _I, _P, _P]),
    'bde_metric_mse': (_I, [_P, _P, _L, _I, _P, _P, _P]),
    'bde_metric_ssim': (_I, [_P, _P, _I, _I, _I, C.c_double, _P, _P, _P]),
    'bde_metric_scratch_doubles': (_I, [_I]),
    'bde_op_head': (_I, [_P, _P, _I, _I, _I, _P, _P]),
    'bde_op_recurrent_conv': (_I, [_P, _I, _I, _P, _I, _I, _I, _I, _P, _P, _P]),
    'bde_op_encoder_conv': (_I, [_P, _I, _I, _P, _I, _I, _I, _P, _P]),
    'bde_op_gate_conv': (_I, [_P, _I, _P, _I, _I, _I, _P, _P]),
    'bde_op_decoder': (_I, [_P, _I, _P, _P, _I, _I, _I, _P, _P]),
    'bde_op_pred': (_I, [_P, _P, _P, _I, _I, _I, _P, _P]),
    'bde_op_dframe_attention': (_I, [_P, _I, _PP, _I, _I, _I, _I, _I, _P, _P]),
}

_lib = None


def lib():
    """Load the shared library once; raise loudly when it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} is missing: build it with `make` (or __graft_entry__.build()). '
                'bde2vid_amd has no CPU/PyTorch fallback.')
        # PyTorch-ROCm ships its own libamdhip64; it must be the one HIP runtime of the process
        # (device pointers and streams cross between torch and this library), so make sure it is
        # mapped before libbde2vid.so resolves its libamdhip64.so.N dependency.
        import torch  # noqa: F401
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)      # AttributeError if the ABI and the binding drift apart
            fn.restype = res
            fn.argtypes = args
        # the .so is built ahead of time and travels beside the sources: a stale one may keep every symbol and still
        # disagree on a signature or on the layout of bde_config
        got = int(handle.bde_abi_version())
        if got != ABI_VERSION and not os.environ.get('BDE_LIB_ANY_ABI'):      # (A/B timing against an older build: tools/gpu.sh ab)
            raise RuntimeError(f'{LIB_PATH} reports ABI version {got}, this binding is written for {ABI_VERSION}: '
                               'rebuild the library (`make`)')
        _lib = handle
    return _lib


class RangeError(RuntimeError):
    """BDE_ERR_RANGE: an activation left the range of the two-term operand format and "sb_auto" is 0."""


def check(status: int):
    if status != 0:
        msg = lib().bde_last_error()
        text = f'libbde2vid error {status}: {msg.decode("utf-8", "replace") if msg else "?"}'
        raise (RangeError if status == ERR_RANGE else RuntimeError)(text)


def make_config(cfg) -> BdeConfig:
    """bde2vid_amd.config.GeneratorConfig -> C struct."""
    cfg.validate()
    c = BdeConfig()
    c.num_bins, c.basechannels, c.num_encoders, c.ks = cfg.num_bins, cfg.basechannels, cfg.num_encoders, cfg.ks
    c.num_heads, c.frame_num, c.q_idx = cfg.num_heads, cfg.frame_num, cfg.q_idx
    c.activation = 1 if cfg.activation == 'Sigmoid' else 0
    if cfg.num_encoders > MAX_LEVELS or cfg.frame_num > MAX_FRAMES:
        raise ValueError('too many levels / buffer frames for the C ABI')
    for i, d in enumerate(cfg.depths):
        c.depths[i] = int(d)
    for i, b in enumerate(cfg.buffer_index):
        c.buffer_index[i] = int(b)
    c.recurrent_type = 1 if cfg.recurrent_block_type == 'convgru' else 0
    c.use_rc = 1 if cfg.useRC else 0
    c.skip_concat = 1 if cfg.skip_type == 'concat' else 0
    c.norm = cfg.norm_kind
    c.num_res_blocks = int(cfg.num_res_blocks)
    return c
