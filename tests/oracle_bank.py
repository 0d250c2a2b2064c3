"""A bank of oracle streams with the method surface meteor_demod_amd.recording needs.

TEST INFRASTRUCTURE (like oracle_py.py): lets the CPU tests drive the stitcher of
meteor_demod_amd/recording.py with the oracle as the tile engine, and lets the GPU
tests require that the HIP bank and the oracle bank produce the same stitched bytes.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

import oracle_py as O

_FIELDS = ["gain", "pll_phase", "pll_freq", "pll_err", "locked", "locked_once", "updown", "t_phase", "t_freq",
           "t_prev", "dual_state", "inphase", "n_samples", "n_symbols", "first_lock_symbol"]


class Snapshot:
    """Loop state of one oracle stream (the attribute names the stitcher reads match MdemodStreamState)."""

    def __init__(self, s: O.OrcState):
        for f in _FIELDS:
            setattr(self, f, getattr(s, f))
        self.bias = (s.bias.re, s.bias.im)

    @property
    def pll_locked(self):
        return self.locked

    @property
    def agc_gain(self):
        return self.gain

    def apply(self, s: O.OrcState) -> None:
        for f in _FIELDS:
            setattr(s, f, getattr(self, f))
        s.bias.re, s.bias.im = self.bias


class OracleBank:
    def __init__(self, cfg, n_streams: int):
        self.cfg = cfg
        self.n_streams = n_streams
        self.streams = [O.OracleStream(cfg) for _ in range(n_streams)]
        self._counts = np.zeros(n_streams, dtype=np.int64)

    def close(self) -> None:
        self.streams = []

    def reset(self) -> None:
        self.streams = [O.OracleStream(self.cfg) for _ in range(self.n_streams)]

    def max_symbols(self, n_samples: int) -> int:
        return int(n_samples) + 16

    def symbol_counts(self):
        return torch.from_numpy(self._counts.copy())

    # -- processing ----------------------------------------------------------------------------
    def process(self, iq):
        """iq [n_streams, n, 2] tensor -> soft [n_streams, cap, 2] int8 tensor."""
        x = iq.cpu().numpy()
        cap = self.max_symbols(x.shape[1])
        soft = torch.zeros((self.n_streams, cap, 2), dtype=torch.int8)
        for i, st in enumerate(self.streams):
            out = st.run(x[i])[0]
            soft[i, : len(out)] = torch.from_numpy(out)
            self._counts[i] = len(out)
        return soft

    def process_ragged(self, iq_flat, offsets, counts, soft):
        x = iq_flat.cpu().numpy()
        offs, cnts = offsets.cpu().numpy(), counts.cpu().numpy()
        for i, st in enumerate(self.streams):
            out = st.run(x[int(offs[i]): int(offs[i]) + int(cnts[i])])[0] if cnts[i] else np.zeros((0, 2), np.int8)
            soft[i, : len(out)] = torch.from_numpy(out)
            self._counts[i] = len(out)
        return soft

    # -- state ---------------------------------------------------------------------------------
    def get_state(self, i: int) -> Snapshot:
        return Snapshot(self.streams[i].state)

    def set_state(self, i: int, snap: Snapshot) -> None:
        snap.apply(self.streams[i]._p.contents.s)

    def get_history(self, i: int) -> np.ndarray:
        return self.streams[i].history()

    def set_history(self, i: int, h: np.ndarray) -> None:
        s = self.streams[i]._p.contents.s
        taps = self.streams[i].consts.taps
        assert h.shape == (taps, 2)
        for k in range(taps):
            s.hist[k].re, s.hist[k].im = float(h[k, 0]), float(h[k, 1])
        s.hidx = 0

    def set_state_all(self, snap: Snapshot) -> None:
        """Every stream := snap, history zeroed (mdemod_set_state_all)."""
        for st in self.streams:
            s = st._p.contents.s
            snap.apply(s)
            for k in range(st.consts.taps):
                s.hist[k].re = s.hist[k].im = 0.0
            s.hidx = 0

    def set_carrier_seeds(self, freq, updown) -> None:
        f, u = freq.cpu().numpy(), updown.cpu().numpy()
        for i, st in enumerate(self.streams):
            s = st._p.contents.s
            s.pll_freq = float(np.float32(f[i]))
            s.updown = 1 if int(u[i]) > 0 else -1

    def set_clock_seeds(self, t_freq) -> None:
        f = t_freq.cpu().numpy()
        for i, st in enumerate(self.streams):
            st._p.contents.s.t_freq = float(np.float32(f[i]))

    def set_gain_seeds(self, gain) -> None:
        g = gain.cpu().numpy()
        for i, st in enumerate(self.streams):
            st._p.contents.s.gain = max(0.0, float(np.float32(g[i])))

    def rotate_carrier(self, quarter_turns) -> None:
        """phase += k*pi/2 wrapped like pll.c:113, in double then narrowed (mdemod_rotate_carrier)."""
        q = quarter_turns.cpu().numpy()
        for i, st in enumerate(self.streams):
            k = int(q[i]) & 3
            if k:
                s = st._p.contents.s
                s.pll_phase = float(np.float32(math.fmod(float(s.pll_phase) + k * (math.pi / 2), 2 * math.pi)))
                if self.cfg.oqpsk and (k & 1):             # rails swap: the symbol clock moves by half a symbol
                    pi_f = np.float32(math.pi)
                    last_q, pend_i = float(s.t_prev), float(s.inphase)
                    if s.dual_state == 1:
                        s.t_phase = float(np.float32(s.t_phase) + pi_f); s.dual_state = 2
                        s.inphase = -last_q if k == 1 else last_q
                    else:
                        s.t_phase = float(np.float32(s.t_phase) - pi_f); s.dual_state = 1
                        s.t_prev = pend_i if k == 1 else -pend_i
