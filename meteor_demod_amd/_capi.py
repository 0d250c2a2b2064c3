"""ctypes binding of include/meteor_demod_amd.h (the C-ABI shared library).

Loading fails loudly if the library has not been built: there is no Python or
CPU implementation of the demodulator to fall back to.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

LIB_DIR = Path(__file__).resolve().parent / "lib"
LIB_PATH = LIB_DIR / "libmeteor_demod_amd.so"
import os as _os
if _os.environ.get("MDEMOD_LIB_PATH"):        # experiments only: load an alternative build of the same library
    LIB_PATH = Path(_os.environ["MDEMOD_LIB_PATH"])

MDEMOD_OK = 0
MDEMOD_ERR_PARAM = -1
MDEMOD_ERR_NOMEM = -2
MDEMOD_ERR_HIP = -3
MDEMOD_ERR_OVERFLOW = -4
MDEMOD_ERR_RANGE = -5
MDEMOD_MAX_LOCK_EVENTS = 32
MDEMOD_FLAG_KERNEL_MASK, MDEMOD_FLAG_LAT_OFF, MDEMOD_FLAG_LAT_ON, MDEMOD_FLAG_V2_PACKED, MDEMOD_FLAG_NO_CLOCK_JUMP = 0x3, 0x4, 0x8, 0x10, 0x20


def variant_flags_from_env() -> int:
    """`mdemod_params.reserved` for the test and bench harness: the library itself reads no environment variable, this wrapper
    does, so that one suite can run every kernel variant.  MDEMOD_KERNEL=v1|v3, MDEMOD_LAT=0|1, MDEMOD_NO_CLOCK_JUMP=1."""
    f = {"v1": 1, "v3": 3}.get(_os.environ.get("MDEMOD_KERNEL", ""), 0)
    lat = _os.environ.get("MDEMOD_LAT", "")
    if lat == "0":
        f |= MDEMOD_FLAG_LAT_OFF
    elif lat not in ("", "-1"):
        f |= MDEMOD_FLAG_LAT_ON
    if _os.environ.get("MDEMOD_NO_CLOCK_JUMP", "") not in ("", "0"):
        f |= MDEMOD_FLAG_NO_CLOCK_JUMP
    return f


class MdemodParams(C.Structure):
    _fields_ = [
        ("pll_bw", C.c_float), ("sym_bw", C.c_float),
        ("samplerate", C.c_int32), ("symrate", C.c_int32),
        ("interp_factor", C.c_int32), ("rrc_order", C.c_int32),
        ("oqpsk", C.c_int32), ("freq_max", C.c_float),
        ("bps", C.c_int32), ("device", C.c_int32),
        ("n_streams", C.c_uint32), ("reserved", C.c_uint32),
    ]


class MdemodStatus(C.Structure):
    _fields_ = [
        ("n_samples", C.c_uint64), ("n_symbols", C.c_uint64),
        ("first_lock_symbol", C.c_int64),
        ("symbols_this_call", C.c_uint32), ("lock_events_this_call", C.c_uint32),
        ("pll_freq", C.c_float), ("omega", C.c_float), ("gain", C.c_float),
        ("locked", C.c_int32), ("locked_once", C.c_int32), ("overflow", C.c_int32),
    ]


class MdemodLockEvent(C.Structure):
    _fields_ = [("symbol", C.c_uint64), ("locked", C.c_int32), ("pad", C.c_int32)]


class MdemodStreamState(C.Structure):
    _fields_ = [
        ("agc_gain", C.c_float), ("agc_bias_re", C.c_float), ("agc_bias_im", C.c_float),
        ("pll_phase", C.c_float), ("pll_freq", C.c_float), ("pll_err", C.c_float),
        ("pll_locked", C.c_int32), ("pll_locked_once", C.c_int32), ("pll_updown", C.c_int32),
        ("t_phase", C.c_float), ("t_freq", C.c_float), ("t_prev", C.c_float),
        ("t_dual_state", C.c_int32), ("oqpsk_inphase", C.c_float),
        ("n_samples", C.c_uint64), ("n_symbols", C.c_uint64),
        ("first_lock_symbol", C.c_int64),
    ]


class MdemodRecordingOpts(C.Structure):
    _fields_ = [("tile_samples", C.c_uint32), ("acquire_samples", C.c_uint32), ("frame_samples", C.c_uint32),
                ("settle_samples", C.c_uint32), ("pilot_block", C.c_uint32), ("pilot_margin_symbols", C.c_uint32),
                ("max_pilot_samples", C.c_uint64), ("match_symbols", C.c_uint32), ("repair", C.c_int32),
                ("carrier_seed", C.c_uint32), ("clock_seed", C.c_uint32), ("debug", C.c_int32), ("debug_tile", C.c_int32)]


class MdemodRecordingReport(C.Structure):
    _fields_ = [("n_symbols", C.c_uint64), ("pilot_samples", C.c_uint64), ("pilot_symbols", C.c_uint64),
                ("exact_symbols", C.c_uint64), ("first_lock_symbol", C.c_int64), ("samples_demodulated", C.c_uint64),
                ("n_tiles", C.c_uint32), ("tile_samples", C.c_uint32), ("weak_seams", C.c_uint32), ("seam_fixes", C.c_uint32),
                ("pilot_locked", C.c_int32), ("weak_carrier_tiles", C.c_uint32),
                ("pilot_seconds", C.c_double), ("tiles_seconds", C.c_double),
                ("frame_misses", C.c_uint32), ("repaired_tiles", C.c_uint32), ("rotation_jumps", C.c_uint32),
                ("frame_residual_rms", C.c_float), ("odd_tiles_kept", C.c_uint32), ("weak_clock_tiles", C.c_uint32)]


# name -> (restype, argtypes); this table is also what the symbol-export test walks.
_P = C.POINTER
SIGNATURES = {
    "mdemod_abi_version": (C.c_uint32, []),
    "mdemod_last_error": (C.c_char_p, []),
    "mdemod_strerror": (C.c_char_p, [C.c_int]),
    "mdemod_init_device": (C.c_int, [C.c_int]),
    "mdemod_create": (C.c_int, [_P(MdemodParams), _P(C.c_void_p)]),
    "mdemod_destroy": (None, [C.c_void_p]),
    "mdemod_reset": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mdemod_max_symbols": (C.c_uint64, [C.c_void_p, C.c_uint64]),
    "mdemod_process_device_uniform": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32,
                                                C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]),
    "mdemod_process_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]),
    "mdemod_process_host": (C.c_int, [C.c_void_p, _P(C.c_void_p), _P(C.c_uint32),
                                      _P(C.c_void_p), _P(C.c_uint32), _P(C.c_uint32)]),
    "mdemod_pin_host_buffer": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "mdemod_unpin_host_buffer": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mdemod_get_status": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, _P(MdemodStatus), C.c_void_p]),
    "mdemod_get_lock_events": (C.c_int, [C.c_void_p, C.c_uint32, _P(MdemodLockEvent), C.c_uint32,
                                         _P(C.c_uint32), C.c_void_p]),
    "mdemod_get_state": (C.c_int, [C.c_void_p, C.c_uint32, _P(MdemodStreamState), C.c_void_p]),
    "mdemod_set_state": (C.c_int, [C.c_void_p, C.c_uint32, _P(MdemodStreamState), C.c_void_p]),
    "mdemod_set_state_all": (C.c_int, [C.c_void_p, _P(MdemodStreamState), C.c_void_p]),
    "mdemod_rotate_carrier": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mdemod_set_carrier_seeds": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mdemod_set_gain_seeds": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mdemod_set_clock_seeds": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mdemod_get_states": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, _P(MdemodStreamState), C.c_void_p]),
    "mdemod_copy_state": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mdemod_nominal_pitch": (C.c_uint64, [C.c_void_p, C.c_uint64]),
    "mdemod_compact_soft": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p]),
    "mdemod_fanin_peer": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]),
    "mdemod_carrier_window_samples": (C.c_uint32, [C.c_void_p, C.c_uint32]),
    "mdemod_estimate_carrier": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint32,
                                          C.c_void_p, C.c_void_p, C.c_void_p]),
    "mdemod_estimate_carrier_chirp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                                C.c_void_p, C.c_void_p, C.c_void_p]),
    "mdemod_estimate_clock": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                        C.c_void_p, C.c_void_p, C.c_void_p]),
    "mdemod_kernel_name": (C.c_char_p, [C.c_void_p]),
    "mdemod_plan_kernel": (C.c_int, [C.c_void_p, C.c_char_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "mdemod_device_count": (C.c_int, []),
    "mdemod_demodulate_recording_host": (C.c_int, [_P(MdemodParams), _P(MdemodRecordingOpts), C.c_void_p, C.c_uint64,
                                                   C.c_void_p, C.c_uint64, _P(MdemodRecordingReport)]),
    "mdemod_recording_default_opts": (None, [_P(MdemodRecordingOpts)]),
    "mdemod_demodulate_recording": (C.c_int, [_P(MdemodParams), _P(MdemodRecordingOpts), C.c_void_p, C.c_uint64,
                                              C.c_void_p, C.c_uint64, _P(MdemodRecordingReport), C.c_void_p]),
    "mdemod_history_len": (C.c_uint32, [C.c_void_p]),
    "mdemod_get_history": (C.c_int, [C.c_void_p, C.c_uint32, _P(C.c_float), C.c_void_p]),
    "mdemod_set_history": (C.c_int, [C.c_void_p, C.c_uint32, _P(C.c_float), C.c_void_p]),
    "mdemod_derive_tables": (C.c_int, [_P(MdemodParams), _P(C.c_float), C.c_uint32,
                                       _P(C.c_float), _P(C.c_float)]),
    "mdemod_get_rrc_table": (C.c_int, [C.c_void_p, _P(C.c_float), C.c_uint32]),
    "mdemod_get_loop_constants": (C.c_int, [C.c_void_p, _P(C.c_float)]),
    "mdemod_get_tanh_lut": (C.c_int, [C.c_void_p, _P(C.c_float)]),
    "mdemod_selftest_sincos": (C.c_int, [C.c_void_p, _P(C.c_float), C.c_uint32, _P(C.c_float), _P(C.c_float)]),
    "mdemod_selftest_hypot": (C.c_int, [C.c_void_p, _P(C.c_float), C.c_uint32, _P(C.c_float)]),
    "mdemod_selftest_turncode": (C.c_int, [C.c_void_p, _P(C.c_uint64), _P(C.c_uint64)]),
    "mdemod_selftest_sinlut": (C.c_int, [C.c_void_p, _P(C.c_uint64), _P(C.c_uint64)]),
    "mdemod_selftest_cabsf": (C.c_int, [C.c_void_p, C.c_uint64, _P(C.c_uint64), _P(C.c_uint64)]),
}

_lib = None


class MdemodError(RuntimeError):
    def __init__(self, code: int, what: str):
        self.code = code
        self.detail = last_error()          # mdemod_last_error(): this thread's failing call, read before anything else calls in
        super().__init__(f"{what}: error {code} ({strerror(code)})" + (f": {self.detail}" if self.detail else ""))


def hip_runtime_first() -> None:
    """One process, one HIP runtime.  PyTorch-ROCm ships its own libamdhip64; the library here is linked against the system's
    (/opt/rocm).  Whichever is loaded first serves both (same soname) - but when the system's comes first and torch then brings
    its own, torch's runtime holds the device and HIP calls of this library fail with "no ROCm-capable device is detected"
    (seen when a CPU-only test had loaded the library before the first GPU test).  So: torch first, whenever it is there."""
    try:
        import torch  # noqa: F401
    except ImportError:
        pass


def lib() -> C.CDLL:
    """The loaded C-ABI library; raises if the HIP extension is not built."""
    global _lib
    if _lib is None:
        hip_runtime_first()
        if not LIB_PATH.exists():
            raise RuntimeError(
                f"{LIB_PATH} is missing: build the HIP extension first "
                "(python -m meteor_demod_amd.build). There is no CPU fallback.")
        handle = C.CDLL(str(LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            if _os.environ.get("MDEMOD_LIB_PATH") and not hasattr(handle, name):
                continue                                   # an older build of the library on A/B duty (tools/ab4.py): newer entries are simply absent
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def strerror(code: int) -> str:
    return lib().mdemod_strerror(code).decode()


def last_error() -> str:
    """``mdemod_last_error``: what the calling thread's most recent failing entry had to say beyond its code ("" if nothing)."""
    fn = getattr(lib(), "mdemod_last_error", None)
    return (fn() or b"").decode(errors="replace") if fn is not None else ""


def check(code: int, what: str) -> int:
    if code < 0:
        raise MdemodError(code, what)
    return code
