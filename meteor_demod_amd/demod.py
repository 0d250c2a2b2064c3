"""Host-side mirror of the reference's demod interface over the C-ABI.

The reference configures its single global demodulator with
``demod_init(pll_bw, sym_bw, samplerate, symrate, interp_factor, rrc_order,
oqpsk, freq_max)`` (demod.h:29) and then feeds it one sample per call
(``demod_qpsk`` / ``demod_oqpsk``, demod.h:42,50).  :class:`Demodulator` takes
the same eight arguments with the same meaning and demodulates *blocks* for
``n_streams`` independent streams; state persists between calls like the
reference's globals do.

PyTorch is used only for device memory and streams.  All arithmetic happens in
the HIP kernels behind ``libmeteor_demod_amd.so``; if that library (or a GPU)
is missing every call raises — there is no CPU path here.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Sequence

import numpy as np

from . import _capi
from ._capi import (MdemodLockEvent, MdemodParams, MdemodStatus, MdemodStreamState, check)

# demod.h:8-15
RRC_ALPHA = 0.6
SYM_RATE = 72000
RRC_ORDER = 32
INTERP_FACTOR = 5
SYM_BW = 0.00005
PLL_BW = 1.0

_NP_DTYPE = {8: np.uint8, 16: np.int16, 32: np.float32}


def scale_freq_max(freq_delta_hz: float, symrate: float) -> float:
    """main.c:136 — ``-d`` in Hz to rad/symbol; negative stays negative (default)."""
    # float * double / float evaluates in double, then narrows to float
    return float(np.float32(float(np.float32(freq_delta_hz)) * (2 * math.pi) / float(np.float32(symrate))))


@dataclass
class DemodConfig:
    """The eight demod_init arguments (demod.h:17-29) + input format."""
    samplerate: int
    pll_bw: float = PLL_BW
    sym_bw: float = SYM_BW
    symrate: int = SYM_RATE
    interp_factor: int = INTERP_FACTOR
    rrc_order: int = RRC_ORDER
    oqpsk: bool = False
    freq_max: float = -1.0
    bps: int = 16

    def to_c(self, n_streams: int = 1, device: int = 0) -> MdemodParams:
        return MdemodParams(self.pll_bw, self.sym_bw, int(self.samplerate), int(self.symrate),
                            int(self.interp_factor), int(self.rrc_order), int(bool(self.oqpsk)),
                            self.freq_max, int(self.bps), int(device), int(n_streams), _capi.variant_flags_from_env())

    @property
    def taps(self) -> int:
        return 2 * self.rrc_order + 1


def derive_tables(cfg: DemodConfig):
    """Init-time tables (host only, no GPU needed): (rrc[interp, taps], consts dict, tanh_lut[32])."""
    lib = _capi.lib()
    p = cfg.to_c()
    n = cfg.interp_factor * cfg.taps
    rrc = np.empty(n, dtype=np.float32)
    consts = np.empty(8, dtype=np.float32)
    lut = np.empty(32, dtype=np.float32)
    check(lib.mdemod_derive_tables(C.byref(p), rrc.ctypes.data_as(C.POINTER(C.c_float)), n,
                                   consts.ctypes.data_as(C.POINTER(C.c_float)),
                                   lut.ctypes.data_as(C.POINTER(C.c_float))), "mdemod_derive_tables")
    names = ["pll_alpha", "pll_beta", "pll_fmax", "t_alpha", "t_beta", "t_center", "t_maxdev", "osf"]
    return rrc.reshape(cfg.interp_factor, cfg.taps), dict(zip(names, consts)), lut


class Demodulator:
    """``demod_init`` + ``demod_qpsk``/``demod_oqpsk`` for a batch of streams on one GPU."""

    def __init__(self, cfg: DemodConfig, n_streams: int = 1, device: int = 0):
        self.cfg = cfg
        self.n_streams = int(n_streams)
        self.device = int(device)
        self._lib = _capi.lib()
        self._ctx = C.c_void_p()
        p = cfg.to_c(n_streams, device)
        check(self._lib.mdemod_create(C.byref(p), C.byref(self._ctx)), "mdemod_create")

    # -- lifecycle ---------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_ctx", None):
            self._lib.mdemod_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def reset(self) -> None:
        check(self._lib.mdemod_reset(self._ctx, self._stream()), "mdemod_reset")

    # -- helpers -------------------------------------------------------------
    def _stream(self) -> C.c_void_p:
        import torch
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def max_symbols(self, n_samples: int) -> int:
        return int(self._lib.mdemod_max_symbols(self._ctx, int(n_samples)))

    def _check_iq(self, iq) -> None:
        """The kernels read raw device memory: a wrong dtype or device is an out-of-bounds read, not an exception."""
        import torch
        want = {8: torch.uint8, 16: torch.int16, 32: torch.float32}[self.cfg.bps]
        if not iq.is_cuda or iq.device.index != self.device:
            raise ValueError(f"iq lives on {iq.device}, this context on cuda:{self.device}")
        if iq.dtype != want:
            raise ValueError(f"iq dtype {iq.dtype} does not match bps={self.cfg.bps} ({want})")

    def _check_soft(self, soft) -> None:
        import torch
        if (not soft.is_cuda or soft.device.index != self.device or soft.dtype != torch.int8 or soft.dim() != 3
                or soft.shape[0] != self.n_streams or soft.shape[2] != 2 or not soft.is_contiguous()):
            raise ValueError(f"soft must be a contiguous int8 [n_streams={self.n_streams}, cap, 2] tensor on cuda:{self.device}")

    def nominal_pitch(self, n_samples: int) -> int:
        """Row pitch in symbols for what n_samples produce at the nominal rate (+1 %), a multiple of 8."""
        return int(self._lib.mdemod_nominal_pitch(self._ctx, int(n_samples)))

    def compact(self, soft, pitch: int, out=None):
        """Rows of the last call's symbols at ``pitch`` symbols instead of the hard-bound capacity of ``soft``."""
        import torch
        self._check_soft(soft)
        if pitch % 8 or soft.shape[1] % 8:
            raise ValueError("pitches must be multiples of 8 symbols")
        if out is None:
            out = torch.empty((self.n_streams, pitch, 2), dtype=torch.int8, device=soft.device)
        check(self._lib.mdemod_compact_soft(self._ctx, C.c_void_p(soft.data_ptr()), soft.shape[1], C.c_void_p(out.data_ptr()),
                                            int(pitch), self._stream()), "mdemod_compact_soft")
        return out

    def fanin_peer(self, soft, pitch: int, dst_soft, first_row: int, dst_counts=None) -> None:
        """``mdemod_fanin_peer``: this context's last symbols, compacted to ``pitch``, written by THIS GPU into rows
        ``first_row ...`` of ``dst_soft`` ([rows, pitch, 2] int8, on any GPU this one has peer access to - or its own) and the
        symbol counts into ``dst_counts[first_row ...]`` (uint32 / int32 tensor on the same device as ``dst_soft``).  The fan-in
        of a host with several GPUs in one process; asynchronous on this device's current stream."""
        import torch
        self._check_soft(soft)
        if pitch % 8 or soft.shape[1] % 8:
            raise ValueError("pitches must be multiples of 8 symbols")
        if (not dst_soft.is_cuda or dst_soft.dtype != torch.int8 or dst_soft.dim() != 3 or dst_soft.shape[1] != pitch or dst_soft.shape[2] != 2
                or not dst_soft.is_contiguous() or first_row < 0 or first_row + self.n_streams > dst_soft.shape[0]):
            raise ValueError(f"dst_soft must be a contiguous int8 [rows >= {first_row + self.n_streams}, {pitch}, 2] tensor on a GPU")
        if dst_counts is not None and (dst_counts.device != dst_soft.device or dst_counts.element_size() != 4 or dst_counts.dim() != 1
                                       or dst_counts.numel() < first_row + self.n_streams or not dst_counts.is_contiguous()):
            raise ValueError("dst_counts: a contiguous 32-bit tensor with a word per row, on dst_soft's device")
        check(self._lib.mdemod_fanin_peer(self._ctx, C.c_void_p(soft.data_ptr()), soft.shape[1], int(dst_soft.device.index),
                                          C.c_void_p(dst_soft.data_ptr()), int(pitch), int(first_row),
                                          C.c_void_p(dst_counts.data_ptr()) if dst_counts is not None else None, self._stream()),
              "mdemod_fanin_peer")

    # -- hot path -------------------------------------------------------------
    def process(self, iq, n_samples: int | None = None, soft=None):
        """Demodulate one block per stream from a device tensor.

        ``iq``: torch tensor on this device, shape [n_streams, n, 2] (dtype
        matching ``bps``), contiguous in the last two dims; stream stride is
        taken from the tensor.  Returns ``soft`` ([n_streams, cap, 2] int8).
        Asynchronous on the current torch stream; per-stream symbol counts are
        in :meth:`status`.
        """
        import torch
        self._check_iq(iq)
        if iq.dim() != 3 or iq.shape[0] != self.n_streams or iq.shape[2] != 2 or iq.stride(2) != 1 or iq.stride(1) != 2:
            raise ValueError(f"iq must be [n_streams={self.n_streams}, n, 2] with contiguous samples, got {tuple(iq.shape)} strides {iq.stride()}")
        n = int(iq.shape[1] if n_samples is None else n_samples)
        if n < 0 or n > iq.shape[1]:
            raise ValueError(f"n_samples={n} outside the block of {iq.shape[1]} samples")
        cap = self.max_symbols(n)
        if soft is None:
            soft = torch.empty((self.n_streams, cap, 2), dtype=torch.int8, device=iq.device)
        self._check_soft(soft)
        check(self._lib.mdemod_process_device_uniform(
            self._ctx, C.c_void_p(iq.data_ptr()), iq.stride(0) // 2, n,
            C.c_void_p(soft.data_ptr()), soft.shape[1], min(cap, soft.shape[1]), self._stream()),
            "mdemod_process_device_uniform")
        return soft

    def process_ragged(self, iq_flat, offsets, counts, soft):
        """Ragged batch: ``iq_flat`` [total, 2] device tensor, ``offsets`` (uint64 as int64) and
        ``counts`` (int32/uint32) device tensors of n_streams entries, ``soft`` [n_streams, cap, 2]."""
        self._check_iq(iq_flat)
        if iq_flat.dim() != 2 or iq_flat.shape[1] != 2 or not iq_flat.is_contiguous():
            raise ValueError("iq_flat must be a contiguous [total, 2] tensor")
        for name, t, size in (("offsets", offsets, 8), ("counts", counts, 4)):
            if (t.device != iq_flat.device or t.numel() != self.n_streams or t.element_size() != size or not t.is_contiguous()
                    or t.is_floating_point()):
                raise ValueError(f"{name} must be a contiguous {size * 8}-bit integer tensor of {self.n_streams} entries on {iq_flat.device}")
        self._check_soft(soft)
        check(self._lib.mdemod_process_device(
            self._ctx, C.c_void_p(iq_flat.data_ptr()), C.c_void_p(offsets.data_ptr()),
            C.c_void_p(counts.data_ptr()), C.c_void_p(soft.data_ptr()), soft.shape[1], soft.shape[1],
            self._stream()), "mdemod_process_device")
        return soft

    def process_host(self, blocks: Sequence[np.ndarray]) -> list[np.ndarray]:
        """Host convenience path (PCIe inclusive): one numpy [n_s, 2] block per stream in,
        one int8 [m_s, 2] array of soft symbols per stream out."""
        assert len(blocks) == self.n_streams
        dt = _NP_DTYPE[self.cfg.bps]
        blocks = [np.ascontiguousarray(b, dtype=dt).reshape(-1, 2) for b in blocks]
        ns = self.n_streams
        iq_ptrs = (C.c_void_p * ns)(*[b.ctypes.data for b in blocks])
        counts = (C.c_uint32 * ns)(*[b.shape[0] for b in blocks])
        caps = [self.max_symbols(b.shape[0]) for b in blocks]
        outs = [np.empty((c, 2), dtype=np.int8) for c in caps]
        soft_ptrs = (C.c_void_p * ns)(*[o.ctypes.data for o in outs])
        soft_caps = (C.c_uint32 * ns)(*caps)
        produced = (C.c_uint32 * ns)()
        check(self._lib.mdemod_process_host(self._ctx, iq_ptrs, counts, soft_ptrs, soft_caps, produced),
              "mdemod_process_host")
        return [o[:produced[i]] for i, o in enumerate(outs)]

    def pin_host(self, array: np.ndarray) -> None:
        """``mdemod_pin_host_buffer``: batches that lie inside ``array`` (rows of equal length, one stride apart) are copied
        straight from its pages by :meth:`process_host`.  Keep ``array`` alive until :meth:`unpin_host` or :meth:`close`."""
        if not array.flags["C_CONTIGUOUS"]:
            raise ValueError("pin_host wants one contiguous allocation (base pointer + nbytes is what gets pinned)")
        check(self._lib.mdemod_pin_host_buffer(self._ctx, C.c_void_p(array.ctypes.data), array.nbytes), "mdemod_pin_host_buffer")

    def unpin_host(self, array: np.ndarray) -> None:
        check(self._lib.mdemod_unpin_host_buffer(self._ctx, C.c_void_p(array.ctypes.data)), "mdemod_unpin_host_buffer")

    # -- status / state ---------------------------------------------------------
    def status(self, first: int = 0, count: int | None = None) -> list[MdemodStatus]:
        count = self.n_streams - first if count is None else count
        arr = (MdemodStatus * count)()
        check(self._lib.mdemod_get_status(self._ctx, first, count, arr, self._stream()), "mdemod_get_status")
        return list(arr)

    def status_array(self, first: int = 0, count: int | None = None) -> np.ndarray:
        """Same snapshot as :meth:`status` as one numpy structured array (cheap for 10^5 streams)."""
        count = self.n_streams - first if count is None else count
        arr = (MdemodStatus * max(count, 1))()
        check(self._lib.mdemod_get_status(self._ctx, first, count, arr, self._stream()), "mdemod_get_status")
        return np.ctypeslib.as_array(arr)[:count].copy()

    def symbol_counts(self):
        """Symbols each stream produced in the last call: int64 CPU tensor [n_streams] (synchronises)."""
        import torch
        return torch.from_numpy(self.status_array()["symbols_this_call"].astype(np.int64))

    def lock_events(self, stream: int) -> list[tuple[int, int]]:
        arr = (MdemodLockEvent * _capi.MDEMOD_MAX_LOCK_EVENTS)()
        n = C.c_uint32()
        check(self._lib.mdemod_get_lock_events(self._ctx, stream, arr, len(arr), C.byref(n), self._stream()),
              "mdemod_get_lock_events")
        return [(int(arr[i].symbol), int(arr[i].locked)) for i in range(min(n.value, len(arr)))]

    def get_state(self, stream: int) -> MdemodStreamState:
        st = MdemodStreamState()
        check(self._lib.mdemod_get_state(self._ctx, stream, C.byref(st), self._stream()), "mdemod_get_state")
        return st

    def set_state(self, stream: int, st: MdemodStreamState) -> None:
        check(self._lib.mdemod_set_state(self._ctx, stream, C.byref(st), self._stream()), "mdemod_set_state")

    def set_state_all(self, st: MdemodStreamState) -> None:
        """Every stream := ``st`` (loop state + counters), filter history zeroed: the seed of overlapped tiles."""
        check(self._lib.mdemod_set_state_all(self._ctx, C.byref(st), self._stream()), "mdemod_set_state_all")

    def rotate_carrier(self, quarter_turns) -> None:
        """pll phase of stream s += quarter_turns[s] * pi/2 (int32 device tensor of n_streams entries)."""
        assert quarter_turns.numel() == self.n_streams and quarter_turns.element_size() == 4
        check(self._lib.mdemod_rotate_carrier(self._ctx, C.c_void_p(quarter_turns.data_ptr()), self._stream()),
              "mdemod_rotate_carrier")

    def set_carrier_seeds(self, freq, updown) -> None:
        """pll frequency (rad/symbol, float32 device tensor) and sweep direction (+1/-1, int32 device tensor) per stream."""
        assert freq.numel() == self.n_streams and updown.numel() == self.n_streams
        assert freq.element_size() == 4 and updown.element_size() == 4 and freq.is_contiguous() and updown.is_contiguous()
        check(self._lib.mdemod_set_carrier_seeds(self._ctx, C.c_void_p(freq.data_ptr()), C.c_void_p(updown.data_ptr()),
                                                 self._stream()), "mdemod_set_carrier_seeds")

    def set_clock_seeds(self, t_freq) -> None:
        """Symbol-clock frequency per stream (rad per interpolated sample, float32 device tensor)."""
        assert t_freq.numel() == self.n_streams and t_freq.element_size() == 4 and t_freq.is_contiguous()
        check(self._lib.mdemod_set_clock_seeds(self._ctx, C.c_void_p(t_freq.data_ptr()), self._stream()), "mdemod_set_clock_seeds")

    def get_states(self, first: int = 0, count: int | None = None) -> list[MdemodStreamState]:
        count = self.n_streams - first if count is None else count
        arr = (MdemodStreamState * max(count, 1))()
        check(self._lib.mdemod_get_states(self._ctx, first, count, arr, self._stream()), "mdemod_get_states")
        return list(arr)[:count]

    def copy_state_from(self, other: "Demodulator") -> None:
        check(self._lib.mdemod_copy_state(self._ctx, other._ctx, self._stream()), "mdemod_copy_state")

    def set_gain_seeds(self, gain) -> None:
        """AGC gain per stream (float32 device tensor)."""
        assert gain.numel() == self.n_streams and gain.element_size() == 4 and gain.is_contiguous()
        check(self._lib.mdemod_set_gain_seeds(self._ctx, C.c_void_p(gain.data_ptr()), self._stream()), "mdemod_set_gain_seeds")

    @property
    def kernel_name(self) -> str:
        return self._lib.mdemod_kernel_name(self._ctx).decode()

    def history_len(self) -> int:
        return int(self._lib.mdemod_history_len(self._ctx))

    def get_history(self, stream: int) -> np.ndarray:
        h = np.empty((self.history_len(), 2), dtype=np.float32)
        check(self._lib.mdemod_get_history(self._ctx, stream, h.ctypes.data_as(C.POINTER(C.c_float)),
                                           self._stream()), "mdemod_get_history")
        return h

    def set_history(self, stream: int, h: np.ndarray) -> None:
        h = np.ascontiguousarray(h, dtype=np.float32).reshape(self.history_len(), 2)
        check(self._lib.mdemod_set_history(self._ctx, stream, h.ctypes.data_as(C.POINTER(C.c_float)),
                                           self._stream()), "mdemod_set_history")

    # -- tables / self tests ------------------------------------------------------
    def rrc_table(self) -> np.ndarray:
        n = self.cfg.interp_factor * self.cfg.taps
        out = np.empty(n, dtype=np.float32)
        check(self._lib.mdemod_get_rrc_table(self._ctx, out.ctypes.data_as(C.POINTER(C.c_float)), n),
              "mdemod_get_rrc_table")
        return out.reshape(self.cfg.interp_factor, self.cfg.taps)

    def selftest_sincos(self, x: np.ndarray):
        x = np.ascontiguousarray(x, dtype=np.float32)
        s = np.empty_like(x)
        c = np.empty_like(x)
        fp = C.POINTER(C.c_float)
        check(self._lib.mdemod_selftest_sincos(self._ctx, x.ctypes.data_as(fp), x.size,
                                               s.ctypes.data_as(fp), c.ctypes.data_as(fp)), "selftest_sincos")
        return s, c

    def selftest_turncode(self) -> tuple[int, int]:
        """(floats checked, mismatches) of the division-free fast_sin turn code, exhaustive on the GPU."""
        n, bad = C.c_uint64(), C.c_uint64()
        check(self._lib.mdemod_selftest_turncode(self._ctx, C.byref(n), C.byref(bad)), "selftest_turncode")
        return n.value, bad.value

    def selftest_cabsf(self, pairs: int = 1 << 32) -> tuple[int, int]:
        """(mismatches, lanes that took the exact fallback) of the short cabsf against the correctly rounded one on the device."""
        bad, fb = C.c_uint64(), C.c_uint64()
        check(self._lib.mdemod_selftest_cabsf(self._ctx, int(pairs), C.byref(bad), C.byref(fb)), "selftest_cabsf")
        return bad.value, fb.value

    def selftest_sinlut(self) -> tuple[int, int]:
        """(turn codes checked, mismatches) of fast_sin's parabola read from the table in LDS against sincos.c's integer arithmetic."""
        n, bad = C.c_uint64(), C.c_uint64()
        check(self._lib.mdemod_selftest_sinlut(self._ctx, C.byref(n), C.byref(bad)), "selftest_sinlut")
        return n.value, bad.value

    def selftest_hypot(self, xy: np.ndarray) -> np.ndarray:
        xy = np.ascontiguousarray(xy, dtype=np.float32).reshape(-1, 2)
        out = np.empty(xy.shape[0], dtype=np.float32)
        fp = C.POINTER(C.c_float)
        check(self._lib.mdemod_selftest_hypot(self._ctx, xy.ctypes.data_as(fp), xy.shape[0],
                                              out.ctypes.data_as(fp)), "selftest_hypot")
        return out
