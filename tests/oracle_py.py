"""ctypes wrapper around oracle/liblrpt_oracle.so and the oracle/_ref tools.

TEST INFRASTRUCTURE: imported only from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this.
"""
from __future__ import annotations

import ctypes as C
import subprocess
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
ORACLE_DIR = ROOT / "oracle"
ORACLE_SO = ORACLE_DIR / "liblrpt_oracle.so"
REF_HARNESS = ORACLE_DIR / "_ref" / "ref_harness"
REF_BINARY = ORACLE_DIR / "_ref" / "meteor_demod_ref"
REF_HARNESS_SHIPPED = ORACLE_DIR / "_ref" / "ref_harness_shipped"     # the reference's Release flags: timing only

TRACE_DTYPE = np.dtype([("sample_index", "<u8"), ("re", "<f4"), ("im", "<f4"), ("pll_freq", "<f4"),
                        ("omega", "<f4"), ("gain", "<f4"), ("locked", "<i4")])
assert TRACE_DTYPE.itemsize == 32


class OrcParams(C.Structure):
    _fields_ = [("pll_bw", C.c_float), ("sym_bw", C.c_float), ("samplerate", C.c_int), ("symrate", C.c_int),
                ("interp", C.c_int), ("rrc_order", C.c_int), ("oqpsk", C.c_int), ("freq_max", C.c_float)]


class OrcCf(C.Structure):
    _fields_ = [("re", C.c_float), ("im", C.c_float)]


class OrcConsts(C.Structure):
    _fields_ = [("interp", C.c_int), ("taps", C.c_int), ("oqpsk", C.c_int),
                ("pll_alpha", C.c_float), ("pll_beta", C.c_float), ("pll_fmax", C.c_float),
                ("t_alpha", C.c_float), ("t_beta", C.c_float), ("t_center", C.c_float), ("t_maxdev", C.c_float),
                ("osf", C.c_float), ("tanh_lut", C.c_float * 32), ("coeffs", C.POINTER(C.c_float))]


class OrcState(C.Structure):
    _fields_ = [("hist", C.POINTER(OrcCf)), ("hidx", C.c_int), ("gain", C.c_float), ("bias", OrcCf),
                ("pll_phase", C.c_float), ("pll_freq", C.c_float), ("pll_err", C.c_float),
                ("locked", C.c_int), ("locked_once", C.c_int), ("updown", C.c_int),
                ("t_phase", C.c_float), ("t_freq", C.c_float), ("t_prev", C.c_float),
                ("dual_state", C.c_int), ("inphase", C.c_float),
                ("n_samples", C.c_uint64), ("n_symbols", C.c_uint64), ("first_lock_symbol", C.c_int64)]


class OrcStream(C.Structure):
    _fields_ = [("c", OrcConsts), ("s", OrcState)]


class OrcLockEvent(C.Structure):
    _fields_ = [("symbol", C.c_uint64), ("locked", C.c_int32)]


_lib = None


def build() -> None:
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    subprocess.run(["make", "-s", "-C", str(ORACLE_DIR)], check=True)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not ORACLE_SO.exists():
            build()
        h = C.CDLL(str(ORACLE_SO))
        h.orc_stream_new.restype = C.POINTER(OrcStream)
        h.orc_stream_new.argtypes = [C.POINTER(OrcParams)]
        h.orc_stream_delete.argtypes = [C.POINTER(OrcStream)]
        h.orc_run.restype = C.c_long
        h.orc_run.argtypes = [C.POINTER(OrcStream), C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t,
                              C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        h.orc_file_model.restype = C.c_long
        h.orc_file_model.argtypes = [C.POINTER(OrcStream), C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t]
        h.orc_fast_sin.restype = C.c_float
        h.orc_fast_sin.argtypes = [C.c_float]
        h.orc_fast_cos.restype = C.c_float
        h.orc_fast_cos.argtypes = [C.c_float]
        h.orc_fast_sin_code.restype = C.c_float
        h.orc_fast_sin_code.argtypes = [C.c_int16]
        h.orc_quantise.restype = C.c_int8
        h.orc_quantise.argtypes = [C.c_float]
        _lib = h
    return _lib


def params_from_cfg(cfg) -> OrcParams:
    """cfg: meteor_demod_amd.DemodConfig (same eight demod_init arguments)."""
    return OrcParams(cfg.pll_bw, cfg.sym_bw, int(cfg.samplerate), int(cfg.symrate), int(cfg.interp_factor),
                     int(cfg.rrc_order), int(bool(cfg.oqpsk)), cfg.freq_max)


class OracleStream:
    """One stream of the CPU restatement (state persists across run() calls)."""

    def __init__(self, cfg):
        self.cfg = cfg
        self._p = lib().orc_stream_new(C.byref(params_from_cfg(cfg)))
        if not self._p:
            raise RuntimeError("orc_stream_new failed")

    def __del__(self):
        if getattr(self, "_p", None):
            try:
                lib().orc_stream_delete(self._p)
            except TypeError:          # interpreter shutdown: module globals are already gone
                pass
            self._p = None

    @property
    def state(self) -> OrcState:
        return self._p.contents.s

    @property
    def consts(self) -> OrcConsts:
        return self._p.contents.c

    def rrc_table(self) -> np.ndarray:
        c = self.consts
        return np.ctypeslib.as_array(c.coeffs, shape=(c.interp * c.taps,)).copy().reshape(c.interp, c.taps)

    def history(self) -> np.ndarray:
        """Last `taps` samples oldest-first as float [taps, 2]."""
        c, s = self.consts, self.state
        h = np.array([(s.hist[i].re, s.hist[i].im) for i in range(c.taps)], dtype=np.float32)
        return np.roll(h, -s.hidx, axis=0)

    def run(self, iq: np.ndarray, want_trace: bool = False):
        """Returns (soft int8 [m,2], trace structured array or None, lock events list)."""
        iq = np.ascontiguousarray(iq)
        fmt = {np.dtype(np.uint8): 8, np.dtype(np.int16): 16, np.dtype(np.float32): 32}[iq.dtype]
        n = iq.size // 2
        cap = n + 16
        soft = np.empty((cap, 2), dtype=np.int8)
        trace = np.empty(cap, dtype=TRACE_DTYPE) if want_trace else None
        events = (OrcLockEvent * 256)()
        nev = C.c_size_t()
        m = lib().orc_run(self._p, iq.ctypes.data, n, fmt, soft.ctypes.data, cap,
                          trace.ctypes.data if want_trace else None, events, 256, C.byref(nev))
        if m < 0:
            raise RuntimeError("orc_run failed")
        ev = [(int(events[i].symbol), int(events[i].locked)) for i in range(min(nev.value, 256))]
        return soft[:m].copy(), (trace[:m].copy() if want_trace else None), ev

    def file_model(self, data: bytes, fmt: int) -> bytes:
        out = np.empty(len(data) + 4096, dtype=np.uint8)
        buf = np.frombuffer(data, dtype=np.uint8)
        m = lib().orc_file_model(self._p, buf.ctypes.data, len(data), fmt, out.ctypes.data, out.size)
        if m < 0:
            raise RuntimeError("orc_file_model failed (overflow or ring_idx > 512 at EOF)")
        return out[:m].tobytes()


def oracle_demod(cfg, iq: np.ndarray, want_trace: bool = False):
    return OracleStream(cfg).run(iq, want_trace)


# ---- the real reference (only where oracle/_ref was built) -------------------------

def have_ref() -> bool:
    return REF_HARNESS.exists()


def _ref_args(cfg) -> list[str]:
    return ["oqpsk" if cfg.oqpsk else "qpsk", str(int(cfg.samplerate)), str(int(cfg.symrate)),
            str(int(cfg.interp_factor)), str(int(cfg.rrc_order)), repr(float(cfg.pll_bw)),
            repr(float(cfg.freq_max)), str(int(cfg.bps))]


def ref_demod(cfg, iq: np.ndarray, want_trace: bool = False):
    """Run the reference itself (one process per stream) through oracle/_ref/ref_harness."""
    with tempfile.TemporaryDirectory() as td:
        inp, out, tr = Path(td) / "in.raw", Path(td) / "out.s", Path(td) / "out.trace"
        np.ascontiguousarray(iq).tofile(inp)
        cmd = [str(REF_HARNESS), "run", *_ref_args(cfg), str(inp), str(out)] + ([str(tr)] if want_trace else [])
        subprocess.run(cmd, check=True, capture_output=True)
        soft = np.fromfile(out, dtype=np.int8).reshape(-1, 2)
        trace = np.fromfile(tr, dtype=TRACE_DTYPE) if want_trace else None
    return soft, trace


def ref_time(cfg, iq: np.ndarray) -> tuple[float, int]:
    """(seconds, samples) of the reference's own per-sample loop on this host, one thread."""
    with tempfile.TemporaryDirectory() as td:
        inp = Path(td) / "in.raw"
        np.ascontiguousarray(iq).tofile(inp)
        r = subprocess.run([str(REF_HARNESS), "time", *_ref_args(cfg), str(inp)], check=True, capture_output=True, text=True)
    dt, n = r.stdout.split()[:2]
    return float(dt), int(n)


def ref_rrc(cfg) -> np.ndarray:
    with tempfile.TemporaryDirectory() as td:
        out = Path(td) / "rrc.f32"
        subprocess.run([str(REF_HARNESS), "rrc", str(int(cfg.samplerate)), str(int(cfg.symrate)),
                        str(int(cfg.interp_factor)), str(int(cfg.rrc_order)), str(out)], check=True)
        return np.fromfile(out, dtype=np.float32).reshape(cfg.interp_factor, cfg.taps)


def ref_sincos(x: np.ndarray):
    with tempfile.TemporaryDirectory() as td:
        inp, out = Path(td) / "x.f32", Path(td) / "y.f32"
        np.ascontiguousarray(x, dtype=np.float32).tofile(inp)
        subprocess.run([str(REF_HARNESS), "sin", str(inp), str(out)], check=True)
        y = np.fromfile(out, dtype=np.float32)
    return y[:x.size], y[x.size:]
