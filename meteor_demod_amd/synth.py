"""Deterministic synthetic LRPT recordings (csrc/synth_core.h) — test/bench input only.

The same per-sample function runs on the host and on gfx950 and is bit-identical
between the two, so a huge device-resident buffer can be spot-checked tile by
tile against a CPU-generated copy.  Signal model: random (O)QPSK
symbols, RRC alpha=0.6, symbol-clock error, carrier offset + phase, AWGN at a
given Es/N0, DC offset, quantised to u8 / s16 / f32.
"""
from __future__ import annotations

import ctypes as C
import math
from pathlib import Path

import numpy as np

LIB_PATH = Path(__file__).resolve().parent / "lib" / "libmdemod_synth.so"

SPAN, OS, TRIG = 6, 256, 1024
PULSE_LEN = 2 * SPAN * OS + 2


class SynthTables(C.Structure):
    _fields_ = [("pulse", C.c_double * PULSE_LEN),
                ("cos_hi", C.c_double * TRIG), ("sin_hi", C.c_double * TRIG),
                ("cos_lo", C.c_double * TRIG), ("sin_lo", C.c_double * TRIG)]


class SynthStream(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("sym_step", C.c_uint64), ("sym_phase0", C.c_uint64),
                ("car_step", C.c_uint32), ("car_phase0", C.c_uint32),
                ("amp", C.c_double), ("noise_scale", C.c_double),
                ("dc_i", C.c_double), ("dc_q", C.c_double),
                ("oqpsk", C.c_int32), ("fmt", C.c_int32), ("car_ramp48", C.c_int64), ("clk_ramp64", C.c_int64)]


_lib = None
_tables = None
_NP = {8: np.uint8, 16: np.int16, 32: np.float32}


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise RuntimeError(f"{LIB_PATH} missing: run python -m meteor_demod_amd.build")
        from ._capi import hip_runtime_first
        hip_runtime_first()
        h = C.CDLL(str(LIB_PATH))
        h.mdemod_synth_tables_size.restype = C.c_size_t
        h.mdemod_synth_stream_size.restype = C.c_size_t
        h.mdemod_synth_tables_init.argtypes = [C.POINTER(SynthTables), C.c_double]
        h.mdemod_synth_tables_init.restype = None
        h.mdemod_synth_host.argtypes = [C.POINTER(SynthTables), C.POINTER(SynthStream), C.c_uint64, C.c_uint64, C.c_void_p]
        h.mdemod_synth_host.restype = None
        h.mdemod_synth_device.argtypes = [C.POINTER(SynthTables), C.POINTER(SynthStream), C.c_uint32, C.c_uint64,
                                          C.c_uint64, C.c_void_p, C.c_uint64, C.c_int]
        h.mdemod_synth_device.restype = C.c_int
        h.mdemod_synth_truth_probe.argtypes = [C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_int]
        h.mdemod_synth_truth_probe.restype = C.c_int
        h.mdemod_synth_truth_count.argtypes = [C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int]
        h.mdemod_synth_truth_count.restype = C.c_int
        assert h.mdemod_synth_tables_size() == C.sizeof(SynthTables)
        assert h.mdemod_synth_stream_size() == C.sizeof(SynthStream)
        _lib = h
    return _lib


def tables() -> SynthTables:
    global _tables
    if _tables is None:
        t = SynthTables()
        lib().mdemod_synth_tables_init(C.byref(t), 0.6)
        _tables = t
    return _tables


def make_stream(seed: int, samplerate: float, symrate: float, *, f0_hz: float = 1200.0,
                phase0_rad: float = 0.7, clock_ppm: float = 0.0, esn0_db: float = 12.0,
                rms: float = 6000.0, dc=(30.0, -20.0), oqpsk: bool = False, fmt: int = 16,
                sym_phase0: float = 0.25, doppler_hz_per_s: float = 0.0, clock_ppm_per_s: float = 0.0) -> SynthStream:
    """Stream descriptor: `rms` is the complex RMS amplitude in LSB of the output format.  A satellite pass moves the carrier
    (`doppler_hz_per_s`) and the symbol clock (`clock_ppm_per_s`) together: ppm/s = Hz/s divided by the RF frequency in MHz."""
    sps = samplerate / (symrate * (1.0 + clock_ppm * 1e-6))
    sym_step = int(round((1.0 / sps) * 2.0 ** 32))
    car_step = int(round((f0_hz / samplerate) * 2.0 ** 32)) & 0xFFFFFFFF
    car_phase0 = int(round((phase0_rad / (2 * math.pi)) * 2.0 ** 32)) & 0xFFFFFFFF
    # unit-energy RRC + +-1 rails: each rail has power 1/1 per symbol period -> complex power 2
    amp = rms / math.sqrt(2.0)
    esn0 = 10.0 ** (esn0_db / 10.0)
    sigma = rms * math.sqrt(sps / (2.0 * esn0))          # per-component noise std
    ih_std = math.sqrt(8.0 * (65536.0 ** 2 - 1.0) / 12.0)  # std of the 8-uniform integer sum
    return SynthStream(seed & (2 ** 64 - 1), sym_step, int(sym_phase0 * 2 ** 32) + (SPAN << 32),
                       car_step, car_phase0, amp, sigma / ih_std, dc[0], dc[1], int(oqpsk), fmt,
                       int(round(doppler_hz_per_s / samplerate ** 2 * 2.0 ** 48)),
                       int(round(clock_ppm_per_s * 1e-6 / samplerate * (1.0 / sps) * 2.0 ** 64)))


def generate_host(st: SynthStream, count: int, n0: int = 0) -> np.ndarray:
    """[count, 2] array in the stream's format, generated on the CPU."""
    out = np.empty((count, 2), dtype=_NP[st.fmt])
    lib().mdemod_synth_host(C.byref(tables()), C.byref(st), n0, count, out.ctypes.data)
    return out


def generate_device(streams, count: int, out=None, n0: int = 0, device: int = 0):
    """torch tensor [n_streams, count, 2] generated on the GPU (all streams same format)."""
    import torch
    fmt = streams[0].fmt
    tdt = {8: torch.uint8, 16: torch.int16, 32: torch.float32}[fmt]
    ns = len(streams)
    if out is None:
        out = torch.empty((ns, count, 2), dtype=tdt, device=f"cuda:{device}")
    assert out.is_contiguous() or out.stride(1) == 2
    arr = (SynthStream * ns)(*streams)
    rc = lib().mdemod_synth_device(C.byref(tables()), arr, ns, n0, count, C.c_void_p(out.data_ptr()),
                                   out.stride(0) // 2, device)
    if rc:
        raise RuntimeError(f"mdemod_synth_device failed ({rc})")
    return out


# ---- truth check of a demodulated recording (tests / bench) ------------------------------------------------------------------

def _probe(st: SynthStream, soft, m0: int, count: int, lag_min: int = -40, n_lags: int = 81, device: int = 0):
    """Best (tx rail, lag, inverted, agreement) for the received I rail and for the received Q rail over symbols [m0, m0 + count)."""
    out = np.zeros((2, 2, n_lags), dtype=np.uint32)
    rc = lib().mdemod_synth_truth_probe(st.seed, C.c_void_p(soft.data_ptr()), m0, count, lag_min, n_lags, out.ctypes.data, device)
    if rc:
        raise RuntimeError(f"mdemod_synth_truth_probe failed ({rc})")
    hyp = []
    for r in range(2):
        dev = np.abs(out[r].astype(np.int64) - count // 2)
        t, l = np.unravel_index(int(dev.argmax()), dev.shape)
        agree = int(out[r, t, l])
        inv = agree < count // 2
        hyp.append((int(t), int(lag_min + l), int(inv), (count - agree if inv else agree) / count))
    return hyp


def best_pairing_agreement(st: SynthStream, soft, m0: int, count: int, device: int = 0) -> float:
    """Share of the hard decisions of symbols [m0, m0 + count) that agree with the transmitted symbols under the best pairing
    of rails, lags and signs (the worse of the two received rails): 0.5 = no relation to the signal."""
    hyp = _probe(st, soft, m0, count, device=device)
    return float(min(hyp[0][3], hyp[1][3]))


def truth_check(st: SynthStream, soft, first_symbol: int = 0, block: int = 65536, device: int = 0, max_probes: int = 256) -> dict:
    """Hard decisions of the WHOLE device tensor `soft` (int8 [m, 2], the output of a demodulation of stream `st`) against the
    symbols the generator transmitted, from `first_symbol` on (skip what was demodulated before the PLL's lock).  The pairing
    (which transmitted rail each received rail carries, at which lag, with which sign = the PLL's quarter-turn ambiguity) is found
    on the first block that HAS one (both rails agree with some transmitted rail on more than 95 % of the block) and held; a block
    whose error rate jumps above 2 % is probed again: another pairing from there on is a rotation change or a symbol slip -
    counted, and the check carries on with it; no pairing at all (a fade, an unlocked stretch) - the blocks are counted as
    unresolved until one fits again.  At Es/N0 = 12 dB a correct QPSK demodulation has a rail error rate of Q(sqrt(Es/N0)) ~
    3.4e-5: the transmitted symbols are not the demodulator's fault."""
    m = int(soft.shape[0])
    assert soft.is_cuda and soft.dtype.itemsize == 1 and soft.is_contiguous()
    nb = (m + block - 1) // block
    size = np.full(nb, block, dtype=np.int64)
    size[-1] = m - (nb - 1) * block
    err = np.zeros((2, nb), dtype=np.int64)
    state = np.zeros(nb, dtype=np.int8)            # 0: not compared, 1: compared, 2: unresolved (no pairing fits)
    changes, first_pairing, probes = [], None, 0
    b = (first_symbol + block - 1) // block
    hyp = None
    while b < nb and probes < max_probes:
        if hyp is None:
            new = _probe(st, soft, b * block, int(size[b]), device=device)
            probes += 1
            if min(new[0][3], new[1][3]) < 0.95:
                state[b] = 2
                b += 1
                continue
            if first_pairing is None:
                first_pairing = new
            elif [x[:3] for x in new] != [x[:3] for x in last]:
                changes.append({"block": int(b), "symbol": int(b * block), "rx_i": new[0][:3], "rx_q": new[1][:3]})
            hyp = new
        last = hyp
        h6 = (C.c_int32 * 6)(hyp[0][0], hyp[0][1], hyp[0][2], hyp[1][0], hyp[1][1], hyp[1][2])
        e = np.zeros((2, nb), dtype=np.uint32)
        rc = lib().mdemod_synth_truth_count(st.seed, C.c_void_p(soft.data_ptr()), m, block, h6, e.ctypes.data, device)
        if rc:
            raise RuntimeError(f"mdemod_synth_truth_count failed ({rc})")
        bad = np.flatnonzero(e[:, b:].max(axis=0) > 0.02 * size[b:])
        stop = b + int(bad[0]) if len(bad) else nb
        err[:, b:stop] = e[:, b:stop]
        state[b:stop] = 1
        b = stop
        hyp = None                                   # the block at `stop` is probed afresh
        if b < nb:
            # a block that is bad under the pairing it is probed to have: the damage is inside it
            new = _probe(st, soft, b * block, int(size[b]), device=device)
            probes += 1
            if [x[:3] for x in new] == [x[:3] for x in last] and min(new[0][3], new[1][3]) >= 0.95:
                err[:, b] = e[:, b]
                state[b] = 1
                hyp = last
                b += 1
    cmp_mask = state == 1
    compared = int(size[cmp_mask].sum())
    total = int(err[:, cmp_mask].sum())
    first_cmp = int(np.argmax(cmp_mask)) if cmp_mask.any() else 0
    res = {"symbols_compared": compared, "first_symbol_compared": first_cmp * block, "rail_decisions_wrong": total,
           "rail_error_rate": total / (2 * compared) if compared else None, "pairing_changes": len(changes), "changes": changes[:8],
           "unresolved_blocks": int((state == 2).sum()), "blocks_not_reached": int((state[(first_symbol + block - 1) // block:] == 0).sum()),
           "worst_block_error_rate": float((err[:, cmp_mask] / size[cmp_mask]).max()) if compared else None, "block_symbols": block}
    if first_pairing is not None:
        res["pairing_rx_i"] = {"tx_rail": "IQ"[first_pairing[0][0]], "lag": first_pairing[0][1], "inverted": bool(first_pairing[0][2])}
        res["pairing_rx_q"] = {"tx_rail": "IQ"[first_pairing[1][0]], "lag": first_pairing[1][1], "inverted": bool(first_pairing[1][2])}
    return res
