"""One long recording on many lanes (BASELINE north star: "the IQ stream tiled into overlapping
blocks so each block runs the serial PLL/Gardner recurrences").

The scheme lives in the library (``mdemod_demodulate_recording``, csrc/recording.hip, NOTEBOOK.md 3.1):
serial pilot -> per-tile carrier / gain seeds -> acquire -> integrators back on their seeds -> frame of
every tile by dead reckoning of the NCO phase -> settle -> body -> seam check and repair.  This module is
the ctypes wrapper around it plus the helper the tests and the bench use to compare a result with the
serial run.  (A Python twin of the round-1 stitcher used to live here; it was removed in round 2 - the
library is the one implementation, tested against the serial oracle.)
"""
from __future__ import annotations

import os as _os

import numpy as np

AUTO = 0xFFFFFFFF


def rotate_symbols(sym, quarter_turns):
    """(I + jQ) * j**k for int8 pairs [..., 2]; ``quarter_turns`` broadcasts over the leading dims.
    Exact: soft symbols are clamped to +-127 (main.c:305), so negation never overflows."""
    import torch
    k = torch.as_tensor(quarter_turns, device=sym.device) & 3
    while k.dim() < sym.dim() - 1:
        k = k.unsqueeze(-1)
    i, q = sym[..., 0], sym[..., 1]
    ri = torch.where(k == 0, i, torch.where(k == 1, -q, torch.where(k == 2, -i, q)))
    rq = torch.where(k == 0, q, torch.where(k == 1, i, torch.where(k == 2, -q, -i)))
    return torch.stack((ri, rq), dim=-1)


def estimate_carrier_native(cfg, iq, starts, window_samples: int, device: int = 0, chirp=None):
    """``mdemod_estimate_carrier[_chirp]`` on a device tensor [n, 2]: (freq [T] float32 rad per NCO step at the middle of each
    window, quality [T], window length actually used).  ``chirp``: optional carrier slope per window, rad per NCO step per sample."""
    import ctypes as C
    import torch
    from . import _capi
    lib = _capi.lib()
    assert iq.is_cuda and iq.dim() == 2 and iq.shape[1] == 2 and iq.is_contiguous()
    p = cfg.to_c(1, device)
    st = torch.as_tensor(np.ascontiguousarray(starts, dtype=np.int64), device=iq.device)
    T = int(st.numel())
    freq = torch.empty(T, dtype=torch.float32, device=iq.device)
    qual = torch.empty(T, dtype=torch.float32, device=iq.device)
    ch = None if chirp is None else torch.as_tensor(np.ascontiguousarray(chirp, dtype=np.float32), device=iq.device)
    stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    _capi.check(lib.mdemod_estimate_carrier_chirp(C.byref(p), C.c_void_p(iq.data_ptr()), int(iq.shape[0]), C.c_void_p(st.data_ptr()),
                                                  C.c_void_p(ch.data_ptr()) if ch is not None else None, T,
                                                  int(window_samples), C.c_void_p(freq.data_ptr()), C.c_void_p(qual.data_ptr()), stream),
                "mdemod_estimate_carrier_chirp")
    used = int(lib.mdemod_carrier_window_samples(C.byref(p), int(window_samples)))
    torch.cuda.current_stream(device).synchronize()
    return freq, qual, used


def estimate_clock_native(cfg, iq, starts, window_samples: int, device: int = 0, carrier=None, chirp=None):
    """``mdemod_estimate_clock`` on a device tensor [n, 2]: (t_freq [T] float32 rad per interpolated step, quality [T]).
    ``carrier`` (OQPSK): each window's carrier in rad per NCO step, ``chirp``: its slope per sample."""
    import ctypes as C
    import torch
    from . import _capi
    lib = _capi.lib()
    assert iq.is_cuda and iq.dim() == 2 and iq.shape[1] == 2 and iq.is_contiguous()
    p = cfg.to_c(1, device)
    st = torch.as_tensor(np.ascontiguousarray(starts, dtype=np.int64), device=iq.device)
    T = int(st.numel())
    tf = torch.empty(T, dtype=torch.float32, device=iq.device)
    qual = torch.empty(T, dtype=torch.float32, device=iq.device)
    dev = lambda a: None if a is None else torch.as_tensor(np.ascontiguousarray(a.cpu().numpy() if hasattr(a, "cpu") else a, dtype=np.float32), device=iq.device)
    ca, ch = dev(carrier), dev(chirp)
    stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    _capi.check(lib.mdemod_estimate_clock(C.byref(p), C.c_void_p(iq.data_ptr()), int(iq.shape[0]), C.c_void_p(st.data_ptr()),
                                          C.c_void_p(ca.data_ptr()) if ca is not None else None,
                                          C.c_void_p(ch.data_ptr()) if ch is not None else None, T, int(window_samples),
                                          C.c_void_p(tf.data_ptr()), C.c_void_p(qual.data_ptr()), stream), "mdemod_estimate_clock")
    torch.cuda.current_stream(device).synchronize()
    return tf, qual


def demodulate_recording_native(cfg, iq, tile_samples: int = 0, acquire_samples: int = AUTO, frame_samples: int = AUTO,
                                settle_samples: int = AUTO, repair: bool = True, pilot_block: int = 65536,
                                pilot_margin_symbols: int = AUTO, max_pilot_samples: int = 0xFFFFFFFFFFFFFFFF, match_symbols: int = 192,
                                device: int = 0, carrier_seed: str = "spectrum", soft_capacity: int = 0, clock_seed: str = "spectrum", soft=None):
    """``mdemod_demodulate_recording`` on a device tensor [n, 2]: returns (soft int8 [m, 2] device tensor, report).
    Lengths in samples; 0 / AUTO take the library's defaults (see include/meteor_demod_amd.h).
    ``soft_capacity`` (symbols; 0 = nominal rate + 5 % + 65536) is what the callee checks its output against; ``soft``: the
    caller's own [capacity, 2] int8 device tensor to write into (the bounds tests put canaries around it)."""
    import ctypes as C
    import torch
    from . import _capi
    lib = _capi.lib()
    if not (iq.is_cuda and iq.dim() == 2 and iq.shape[1] == 2 and iq.is_contiguous()):
        raise ValueError("iq must be a contiguous [n, 2] device tensor")
    want = {8: torch.uint8, 16: torch.int16, 32: torch.float32}[cfg.bps]
    if iq.dtype != want or iq.device.index != device:
        raise ValueError(f"iq is {iq.dtype} on {iq.device}; expected {want} on cuda:{device}")
    if carrier_seed not in ("pilot", "spectrum") or clock_seed not in ("pilot", "spectrum"):
        raise ValueError("carrier_seed and clock_seed are 'pilot' or 'spectrum'")
    opts = _capi.MdemodRecordingOpts(int(tile_samples), int(acquire_samples), int(frame_samples), int(settle_samples),
                                     int(pilot_block), int(pilot_margin_symbols), int(max_pilot_samples), int(match_symbols),
                                     int(bool(repair)), 1 if carrier_seed == "spectrum" else 0, 0 if clock_seed == "spectrum" else 1,
                                     int(_os.environ.get("MDEMOD_RECORDING_DEBUG", "0") or 0), int(_os.environ.get("MDEMOD_RECORDING_TRACE_TILE", "-1")))
    n = int(iq.shape[0])
    p = cfg.to_c(1, device)
    cap = int(soft_capacity) or int(n * cfg.symrate / cfg.samplerate * 1.05) + 65536   # stitched output: nominal rate + slack (checked by the callee)
    if soft is None:
        soft = torch.empty((cap, 2), dtype=torch.int8, device=iq.device)
    elif not (soft.is_cuda and soft.dtype == torch.int8 and soft.dim() == 2 and soft.shape[1] == 2 and soft.is_contiguous() and soft.device == iq.device):
        raise ValueError("soft must be a contiguous [capacity, 2] int8 tensor on the input's device")
    else:
        cap = int(soft.shape[0])
    rep = _capi.MdemodRecordingReport()
    stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    _capi.check(lib.mdemod_demodulate_recording(C.byref(p), C.byref(opts), C.c_void_p(iq.data_ptr()), n,
                                                C.c_void_p(soft.data_ptr()), cap, C.byref(rep), stream),
                "mdemod_demodulate_recording")
    return soft[: rep.n_symbols], rep


# ---- evaluation helper (tests / bench) --------------------------------------------------------

def agreement(stitched: np.ndarray, serial: np.ndarray, window: int = 4096) -> dict:
    """Fraction of symbols within +-1 LSB of the serial demodulation, overall and per window.  Both are
    int8 [m, 2]; a length difference is reported, the common prefix is compared."""
    m = min(len(stitched), len(serial))
    d = np.abs(stitched[:m].astype(np.int16) - serial[:m].astype(np.int16)).max(axis=1)
    ok = d <= 1
    sign_ok = ((stitched[:m] >= 0) == (serial[:m] >= 0)).all(axis=1)
    wins = [float(ok[i:i + window].mean()) for i in range(0, m, window)]
    return {"len_stitched": int(len(stitched)), "len_serial": int(len(serial)), "within_1lsb": float(ok.mean()) if m else 1.0,
            "hard_decisions_equal": float(sign_ok.mean()) if m else 1.0, "worst_window": min(wins) if wins else 1.0,
            "windows": wins}


def converged_pair_runs(cfg, iq, copies: int = 31, seed: int = 1, block: int = 1 << 20, at_fraction=0.25):
    """The serial run of ``iq`` [n, 2] (device) and ``copies`` perturbed twins of it as bit-exact streams of the library, all reading
    the same samples: twin c takes the serial run's own state at ``at_fraction`` (one number, or one per twin) of the recording with the
    symbol-clock word moved by +-0.3 .. 3 ppm and runs on.  Returns (soft [copies + 1, cap, 2] int8 on the device, symbols per stream,
    symbol index of the perturbation: an int, or an array per twin when ``at_fraction`` was one)."""
    import torch
    from .demod import Demodulator
    n = int(iq.shape[0])
    per_twin = not np.isscalar(at_fraction)
    fr = np.asarray(at_fraction if per_twin else [at_fraction] * copies, dtype=np.float64)
    assert len(fr) == copies
    # perturbations happen at block boundaries (one common boundary when all twins share the instant, as r05's yardstick does)
    Ks = np.array([int(n * f) for f in fr]) if not per_twin else np.array([max(block, int(n * f) // block * block) for f in fr])
    S = copies + 1
    rng = np.random.default_rng(seed)
    cap_total = int(n * cfg.symrate / cfg.samplerate * 1.02) + 4096
    out = torch.zeros((S, cap_total, 2), dtype=torch.int8, device=iq.device)
    fill = np.zeros(S, dtype=np.int64)
    p0 = np.zeros(copies, dtype=np.int64)
    cuts = sorted(set(int(k) for k in Ks if 0 < k < n))
    with Demodulator(cfg, S, device=iq.device.index or 0) as d:
        pos = 0
        while pos < n:
            nxt = next((c for c in cuts if c > pos), n)
            m = min(block, n - pos, nxt - pos)
            soft = d.process(iq[pos:pos + m].unsqueeze(0).expand(S, m, 2))      # every stream reads the same samples
            torch.cuda.synchronize(iq.device)
            cnt = d.status_array()["symbols_this_call"].astype(np.int64)
            for s in range(S):
                out[s, int(fill[s]):int(fill[s]) + int(cnt[s])] = soft[s, : int(cnt[s])]
            fill += cnt
            pos += m
            for c in np.flatnonzero(Ks == pos):
                s = int(c) + 1
                # (a twin perturbed later than the others has run as an exact copy of the serial run until now)
                p0[c] = int(fill[0])
                st = d.get_state(s)
                ppm = float(rng.uniform(0.3, 3.0)) * (1 if rng.integers(0, 2) else -1)
                st.t_freq = float(np.float32(st.t_freq * (1.0 + ppm * 1e-6)))
                d.set_state(s, st)
    return out, fill, (p0 if per_twin else int(p0[0]))


def converged_pair_yardstick(cfg, iq, copies: int = 31, seed: int = 1, window: int = 4096, skip: int = 60000, block: int = 1 << 20) -> dict:
    """The yardstick for a tiled run, with enough windows behind it (round 5; r04 compared a tail of 112 in 20 515 windows with
    "1 of 1 908"): two CONVERGED runs of the reference on the same samples while they are apart - ``copies`` times over.  The copies
    take the serial run's own state at a quarter of the recording with the symbol-clock word moved by +-0.3 .. 3 ppm (inside what
    timing.c:80-86 can hold; pulled in within a few loop time constants) and run on; every one of them is a bit-exact stream of the
    library (all read the same device buffer ``iq`` [n, 2]), i.e. what the reference itself would produce from that state.  Each copy is
    compared with the unperturbed run from ``skip`` symbols after the perturbation up to the last symbol on which they differ; the
    windows of all copies are pooled.  Returns within_1lsb, the share of windows below 0.99, worst window, 1 % and 0.1 % quantiles."""
    import torch
    copies_ = copies
    out, fill, p0 = converged_pair_runs(cfg, iq, copies=copies, seed=seed, block=block)
    S = copies_ + 1
    ref = out[0].to(torch.int16)
    wins, met, n_ok, n_all = [], 0, 0.0, 0
    # the same by time since the comparison starts, and for the stretch before two runs meet again: is a pair that is about to
    # meet closer than one that is not?  (a tile never meets the serial run: it is emitted for 8 000 .. 40 000 symbols)
    edges = [0, 25_000, 50_000, 100_000, 200_000, 400_000, 800_000, 1_600_000, 1 << 62]
    bin_ok, bin_n, bin_low, bin_w = np.zeros(len(edges) - 1), np.zeros(len(edges) - 1), np.zeros(len(edges) - 1), np.zeros(len(edges) - 1)
    last_ok = last_n = 0.0
    for s in range(1, S):
        m = int(min(fill[0], fill[s]))
        dd = (out[s, :m].to(torch.int16) - ref[:m]).abs().amax(dim=1)
        nz = torch.nonzero(dd > 0)
        last = int(nz[-1]) + 1 if nz.numel() else 0
        met += 1 if last < m - window else 0
        a, b = p0 + skip, last
        if b - a < window:
            continue
        ok = (dd[a:b] <= 1).to(torch.float32)
        k = (b - a) // window
        w_s = ok[: k * window].view(k, window).mean(dim=1).cpu().numpy()
        wins.append(w_s)
        n_ok += float(ok.sum()); n_all += int(ok.numel())
        for j in range(len(edges) - 1):
            lo, hi = edges[j], min(edges[j + 1], b - a)
            if hi <= lo:
                break
            bin_ok[j] += float(ok[lo:hi].sum()); bin_n[j] += hi - lo
            wj = w_s[lo // window: hi // window]
            bin_low[j] += float((wj < 0.99).sum()); bin_w[j] += len(wj)
        if last < m - window and b - a > 400_000:                 # the 200 000 symbols before the two runs met again
            last_ok += float(ok[-200_000:].sum()); last_n += 200_000
    w = np.concatenate(wins) if wins else np.array([])
    return {"copies": copies, "copies_that_met_the_serial_run_again": met, "symbols_compared_while_apart": n_all,
            "within_1lsb": round(n_ok / max(n_all, 1), 5), "windows": int(len(w)),
            "windows_below_0.99": int((w < 0.99).sum()), "share_below_0.99": round(float((w < 0.99).mean()), 5) if len(w) else None,
            "worst_window_4096": round(float(w.min()), 4) if len(w) else None,
            "window_p01": round(float(np.quantile(w, 0.01)), 4) if len(w) else None,
            "window_p001": round(float(np.quantile(w, 0.001)), 4) if len(w) else None,
            "by_symbols_since_the_comparison_starts": [{"from": int(edges[j]), "to": (int(edges[j + 1]) if edges[j + 1] < (1 << 60) else None),
                                                        "within_1lsb": round(bin_ok[j] / bin_n[j], 5), "share_below_0.99": round(bin_low[j] / max(bin_w[j], 1), 5),
                                                        "windows": int(bin_w[j])} for j in range(len(edges) - 1) if bin_n[j] > 0],
            "last_200000_symbols_before_meeting_again": round(last_ok / last_n, 5) if last_n else None}


def tiled_vs_twins(cfg, iq, soft, exact_symbols: int, copies: int = 31, seed: int = 3, window: int = 4096, skip: int = 100_000) -> dict:
    """A tiled run's windows next to what converged twins of the serial run do IN THE SAME WINDOWS of the same recording (round 5).

    Where two converged runs of the reference disagree is mostly a property of the SIGNAL (r05, tools/tail_vs_pairs.py: in the windows
    where a tiled run falls below 0.99 a third of the twins' windows do too, against 0.16 % elsewhere), so a tail measured on one
    stretch of a recording says little about another: the yardstick has to see the same windows.  The twins (``converged_pair_runs``)
    are perturbed at instants spread over the recording, each is compared from ``skip`` symbols after its perturbation until it has
    met the serial run again; ``soft`` [m, 2] is the tiled output (device), ``exact_symbols`` its exact prefix (not compared)."""
    import torch
    out, fill, p0s = converged_pair_runs(cfg, iq, copies=copies, seed=seed, at_fraction=[0.02 + 0.9 * c / copies for c in range(copies)])
    ref = out[0].to(torch.int16)
    m = int(min(int(fill.min()), int(soft.shape[0])))
    k = m // window

    def windows(x):
        d = (x[: k * window].to(torch.int16) - ref[: k * window]).abs().amax(dim=1)
        return (d <= 1).to(torch.float32).view(k, window).mean(dim=1).cpu().numpy(), (d > 0).view(k, window).any(dim=1).cpu().numpy()

    tile_w, _ = windows(soft[:m])
    first = int(exact_symbols) // window + 1
    twin_w = np.full((copies, k), np.nan)
    for c in range(copies):
        w, differs = windows(out[c + 1, :m])
        last = int(np.flatnonzero(differs)[-1]) + 1 if differs.any() else 0
        a = max(first, (int(p0s[c]) + skip) // window + 1)
        if last > a:
            twin_w[c, a:last] = w[a:last]
    valid = np.isfinite(twin_w).sum(axis=0) >= 1
    valid[:first] = False
    idx = np.flatnonzero(valid)
    if not len(idx):
        return {"windows_compared": 0}
    t, tw = tile_w[idx], twin_w[:, idx]
    fin = np.isfinite(tw)
    twin_mean, twin_min = np.nanmean(tw, axis=0), np.nanmin(tw, axis=0)
    low = t < 0.99
    return {"copies": copies, "windows_compared": int(len(idx)), "twins_apart_per_window": round(float(fin.sum(axis=0).mean()), 2),
            "tiled": {"within_1lsb": round(float(t.mean()), 5), "share_below_0.99": round(float(low.mean()), 5), "worst_window": round(float(t.min()), 4),
                      "window_p01": round(float(np.quantile(t, 0.01)), 4)},
            "twins_same_windows": {"within_1lsb": round(float(tw[fin].mean()), 5), "share_below_0.99": round(float((tw[fin] < 0.99).mean()), 5),
                                   "worst_window": round(float(tw[fin].min()), 4), "window_p01": round(float(np.quantile(tw[fin], 0.01)), 4)},
            "where_the_tiled_run_is_below_0.99": {"windows": int(low.sum()),
                                                 "twins_mean_there": round(float(twin_mean[low].mean()), 5) if low.any() else None,
                                                 "twins_mean_elsewhere": round(float(twin_mean[~low].mean()), 5),
                                                 "share_of_twin_windows_below_0.99_there": round(float((tw[:, low][fin[:, low]] < 0.99).mean()), 4) if low.any() else None,
                                                 "some_twin_below_0.99_there_too": round(float((twin_min[low] < 0.99).mean()), 4) if low.any() else None},
            "correlation_of_tiled_window_with_twins_mean": round(float(np.corrcoef(t, twin_mean)[0, 1]), 3) if len(idx) > 2 else None}
