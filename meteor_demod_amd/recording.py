"""One long recording as overlapped tiles (BASELINE north star: "the IQ stream tiled
into overlapping blocks so each block runs the serial PLL/Gardner recurrences").

The reference demodulates a recording as ONE serial recurrence (main.c:303-316):
tiles of it cannot be made bit-identical to that run.  What can be
made exact is every piece of the following scheme, because each piece is an
ordinary stream with a defined initial state, and streams are bit-exact:

  pilot    The head of the recording runs as one stream from the reference's
           power-on state until the PLL has locked and stayed locked for
           ``pilot_margin_symbols``.  Its symbols ARE the reference's symbols
           (first-lock index included, so the lock gate of main.c:312 opens on the
           same chunk).  The carrier loop of pll.c needs ~1.5e5 symbols after lock
           before its frequency estimate stops moving and a seed taken earlier
           leaves a static phase lag in the tiles; with the default tile length the
           two passes age every tile's state by ~4e4 symbols themselves, so 2e4
           symbols of margin cost only ~0.5 % of +-1 LSB agreement (98.6 % instead
           of 99.1 %, measured) and save 0.3 s: one lane runs ~1.4 MS/s and the
           pilot is the latency of a single recording.
  pass 1   Every tile starts ``pre`` samples early from the pilot's end state
           (converged AGC / carrier frequency / symbol clock), warm-up symbols
           are dropped, the body's symbols are kept.
  rotation A QPSK Costas loop locks on any of four constellation rotations.  Each
           tile's rotation relative to its predecessor is measured on the samples
           both demodulated (the predecessor's tail = the tile's warm-up), prefix
           summed, and undone.  The same comparison detects the one-symbol
           duplicate / gap that appears when two tiles place the symbol that
           straddles their seam on different sides of it.
  pass 2   (``refine=True``) Tile i+1 is demodulated again, this time as the exact
           continuation of tile i's pass-1 end state (history included), turned
           into the pilot's rotation.  No warm-up is needed and all tiles now run
           in the rotation the serial reference runs in, which matters because the
           timing detector reads only the Q rail (timing.c:65-66).

All arithmetic on samples happens in the HIP kernels; this module only plans
offsets, compares int8 tails and concatenates.  It is written against a small
"bank" interface so that the CPU tests can drive exactly the same code with the
oracle as the tile engine (tests/test_recording_cpu.py).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np


# ---- planning (pure host logic) ------------------------------------------------------------

@dataclass
class TilePlan:
    n_samples: int
    pilot_end: int                 # samples [0, pilot_end) belong to the pilot
    starts: np.ndarray             # int64 [T]  first body sample of each tile
    lens: np.ndarray               # int64 [T]  body samples of each tile
    pres: np.ndarray               # int64 [T]  warm-up samples actually available (<= pre)

    @property
    def n_tiles(self) -> int:
        return int(self.starts.shape[0])


def plan_tiles(n_samples: int, pilot_end: int, tile_samples: int, pre_samples: int) -> TilePlan:
    """Tiles of ``tile_samples`` cover [pilot_end, n_samples); the last one may be short."""
    assert tile_samples > 0 and pre_samples >= 0 and 0 <= pilot_end <= n_samples
    starts = np.arange(pilot_end, n_samples, tile_samples, dtype=np.int64)
    lens = np.minimum(tile_samples, n_samples - starts).astype(np.int64)
    pres = np.minimum(pre_samples, starts).astype(np.int64)
    return TilePlan(n_samples, pilot_end, starts, lens, pres)


def default_tiling(cfg, tile_samples: int = 0, pre_samples: int = -1):
    """Tile and warm-up lengths when the caller gives none (0 / -1): 20 536 and 5 129 (OQPSK: 10 258) SYMBOLS worth of samples, i.e. 65 600 and
    16 384 samples at the reference's 72 k symbols in 230 kS/s, scaled with the samples per symbol so that a 1 MS/s
    recording gets tiles of the same duration in symbols.  The tile is kept off powers of two (lanes read at base + l * tile:
    a power-of-two stride puts a wave's lanes on the same L2 sets).  Same rule in csrc/recording.hip."""
    osf = cfg.samplerate / cfg.symrate
    if tile_samples <= 0:
        tile_samples = max(4096, int(20536 * osf) // 64 * 64)
        if tile_samples & (tile_samples - 1) == 0:
            tile_samples += 64
    if pre_samples < 0:
        # OQPSK: twice the warm-up.  Its carrier loop has half the bandwidth (demod.c:8-15) and may still be re-locking when a
        # 5 129-symbol warm-up ends; a tile that changes rotation after its start was measured throws every later tile a
        # quarter turn off (profiles/r01_rotation_jump_cases.md: both soak cases vanish with the longer warm-up)
        pre_samples = int((10258 if cfg.oqpsk else 5129) * osf)
    return int(tile_samples), int(pre_samples)


# ---- int8 symbol helpers (torch tensors, any device) -------------------------------------------

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


def gather_tails(soft, counts, k):
    """Last ``k`` symbols of each row of ``soft`` [T, cap, 2] given per-row ``counts`` -> [T, k, 2] (int32),
    plus a validity mask [T, k] (False where the row has fewer than k symbols)."""
    import torch
    T = soft.shape[0]
    idx = counts.view(T, 1).to(torch.int64) - k + torch.arange(k, device=soft.device).view(1, k)
    ok = idx >= 0
    g = torch.gather(soft, 1, idx.clamp(min=0).unsqueeze(-1).expand(T, k, 2)).to(torch.int32)
    return g * ok.unsqueeze(-1), ok


def match_tails(a_tail, b_tail):
    """Best (shift, rotation) aligning tail B onto tail A.

    ``a_tail``, ``b_tail``: [T, K+1, 2] int32, the last K+1 symbols that two demodulations produced
    for the same stretch of samples.  Returns (shift [T], rot [T], score [T], energy [T]) where
    ``b * j**rot`` matches ``a`` and shift is
       0  both end on the same symbol,
      +1  A ends one symbol later than B (A holds a symbol B does not have yet),
      -1  B ends one symbol later than A.
    """
    import torch
    K = a_tail.shape[1] - 1
    best = None
    for shift, (asl, bsl) in ((0, (slice(1, K + 1), slice(1, K + 1))),
                              (1, (slice(0, K), slice(1, K + 1))),
                              (-1, (slice(1, K + 1), slice(0, K)))):
        a, b = a_tail[:, asl], b_tail[:, bsl]
        ai, aq, bi, bq = a[..., 0], a[..., 1], b[..., 0], b[..., 1]
        re = (ai * bi + aq * bq).sum(1)            # Re sum a * conj(b)
        im = (aq * bi - ai * bq).sum(1)            # Im sum a * conj(b)
        # a ~ b * j**r  <=>  sum a conj(b) ~ |b|^2 j**r
        scores = torch.stack((re, im, -re, -im), dim=1)          # r = 0, 1, 2, 3
        sc, r = scores.max(dim=1)
        cand = (sc, torch.full_like(r, shift), r)
        if best is None:
            best = cand
        else:
            take = cand[0] > best[0]
            best = tuple(torch.where(take, c, b0) for c, b0 in zip(cand, best))
    energy = (a_tail[:, 1:].to(torch.int64) ** 2).sum((1, 2))
    return best[1], best[2], best[0], energy


def gather_heads(soft, counts, k):
    """First ``k`` symbols of each row -> [T, k, 2] int32 (zeros where the row is shorter)."""
    import torch
    T = soft.shape[0]
    kk = min(k, soft.shape[1])
    g = soft[:, :kk].to(torch.int32)
    if kk < k:
        g = torch.cat((g, torch.zeros((T, k - kk, 2), dtype=torch.int32, device=soft.device)), dim=1)
    ok = torch.arange(k, device=soft.device).view(1, k) < counts.view(T, 1)
    return g * ok.unsqueeze(-1)


def match_heads(a_head, b_head):
    """Like :func:`match_tails` for two demodulations that START on the same sample ([T, K+1, 2] int32 each):
    shift 0: both start with the same symbol; +1: A starts one symbol earlier (A[1] is B[0]); -1: B starts one earlier."""
    import torch
    K = a_head.shape[1] - 1
    best = None
    for shift, (asl, bsl) in ((0, (slice(0, K), slice(0, K))), (1, (slice(1, K + 1), slice(0, K))), (-1, (slice(0, K), slice(1, K + 1)))):
        a, b = a_head[:, asl], b_head[:, bsl]
        ai, aq, bi, bq = a[..., 0], a[..., 1], b[..., 0], b[..., 1]
        re = (ai * bi + aq * bq).sum(1)
        im = (aq * bi - ai * bq).sum(1)
        sc, r = torch.stack((re, im, -re, -im), dim=1).max(dim=1)
        cand = (sc, torch.full_like(r, shift), r)
        best = cand if best is None else tuple(torch.where(cand[0] > best[0], c, b0) for c, b0 in zip(cand, best))
    energy = (a_head[:, :K].to(torch.int64) ** 2).sum((1, 2))
    return best[1], best[2], best[0], energy


def match_rails(a_tail, b_tail):
    """OQPSK: rotation of tail B against tail A when the two rails may be paired differently.

    The I and Q rails of an OQPSK symbol come from firings half a symbol apart (demod.c:66-76).  A tile locked 90
    degrees off has the rails swapped AND paired one symbol apart (at +90 its pair k is (-Q_k, I_k+1) of the reference),
    so the rails are correlated separately: 4 rotations x 3 shifts per rail, tails [T, K+2, 2] aligned at their ends.
    Returns (rot [T], score [T], energy [T])."""
    import torch
    K = a_tail.shape[1] - 2
    a = a_tail[:, 1:K + 1]                                      # K symbols, one symbol of margin on both sides
    bI, bQ = b_tail[..., 0], b_tail[..., 1]
    maps = ((bI, bQ), (-bQ, bI), (-bI, -bQ), (bQ, -bI))        # rails of b * j**r, the convention of match_tails()
    best = None
    for r, (mI, mQ) in enumerate(maps):
        sI = torch.stack([(a[..., 0] * mI[:, 1 + d:K + 1 + d]).sum(1) for d in (-1, 0, 1)], dim=1).max(dim=1)[0]
        sQ = torch.stack([(a[..., 1] * mQ[:, 1 + d:K + 1 + d]).sum(1) for d in (-1, 0, 1)], dim=1).max(dim=1)[0]
        sc = sI + sQ
        cand = (sc, torch.full_like(sc, r))
        best = cand if best is None else tuple(torch.where(cand[0] > best[0], c, b0) for c, b0 in zip(cand, best))
    energy = (a.to(torch.int64) ** 2).sum((1, 2))
    return best[1], best[0], energy


def carrier_estimates(iq, starts, nfft, samplerate, symrate, fmax_rad=0.33, nco_steps_per_symbol=1, return_power=False):
    """Feed-forward carrier estimate of each tile: the 4th power of the samples has a spectral line at 4x the carrier
    offset whatever the data, for QPSK and for RRC-shaped OQPSK alike (``nco_steps_per_symbol`` = 2 for OQPSK, whose NCO
    advances at both rails' firings, pll.c:77,93, so its frequency word is rad per half symbol) (the reference finds the carrier by sweeping its PLL at 1e-6 rad/symbol per symbol, pll.c:125,
    which is what a tile that starts far from the pilot's estimate has no time for).  ``nfft`` samples from each start,
    batched FFT, peak within +-4*fmax with parabolic interpolation.  Returns (freq [T] float32 rad/symbol at the MIDDLE of the
    window, peak-to-mean ratio [T]).  torch.fft on whatever device ``iq`` lives on; estimation only, never symbols.
    The native entry computes the same estimate in one kernel (``carrier_line_kernel``, csrc/recording.hip): z^4 summed in
    groups of 4..16 samples before a 16384-point FFT in LDS - same bin width, same band, no FFT library (its run-time
    kernel compilation cost 1.6 s per process)."""
    import torch
    T = int(starts.shape[0])
    dev = iq.device
    n = iq.shape[0]
    kmax = int(4 * fmax_rad * symrate / (2 * np.pi) / samplerate * nfft) + 2          # bins of 4 * fmax
    win = torch.hann_window(nfft, periodic=False, device=dev, dtype=torch.float32)
    freq = torch.zeros(T, dtype=torch.float32, device=dev)
    quality = torch.zeros(T, dtype=torch.float32, device=dev)
    power = torch.zeros(T, dtype=torch.float32, device=dev)
    step = max(1, (1 << 28) // (nfft * 8))                                           # <= 256 MiB of complex64 per batch
    ar = torch.arange(nfft, device=dev)
    for t0 in range(0, T, step):
        t1 = min(T, t0 + step)
        st = torch.as_tensor(np.minimum(np.asarray(starts[t0:t1], dtype=np.int64), max(0, n - nfft)), device=dev)
        idx = (st.view(-1, 1) + ar.view(1, -1)).clamp(max=n - 1)
        x = iq[idx]                                                                   # [b, nfft, 2]
        z = torch.complex(x[..., 0].to(torch.float32), x[..., 1].to(torch.float32))
        z = z - z.mean(dim=1, keepdim=True)
        power[t0:t1] = (z.real * z.real + z.imag * z.imag).mean(dim=1)
        z = z / (z.abs().mean(dim=1, keepdim=True) + 1e-20)
        z4 = (z * z) * (z * z) * win
        sp = torch.fft.fft(z4, dim=1).abs()
        cand = torch.cat((sp[:, -kmax:], sp[:, : kmax + 1]), dim=1)                    # bins -kmax .. +kmax
        pk = cand[:, 1:-1].argmax(dim=1) + 1
        a, b, c = (cand.gather(1, (pk + d).view(-1, 1)).squeeze(1) for d in (-1, 0, 1))
        delta = 0.5 * (a - c) / (a - 2 * b + c - 1e-20)
        k = (pk - kmax).to(torch.float32) + delta
        freq[t0:t1] = (k * (samplerate / nfft / 4.0) * (2 * np.pi / (symrate * nco_steps_per_symbol))).to(torch.float32)
        quality[t0:t1] = b / (cand.mean(dim=1) + 1e-20)
    return (freq, quality, power) if return_power else (freq, quality)


def window_power(iq, start: int, length: int, count: int, last_length: int = -1):
    """Sample power (mean |z - mean|^2) of ``count`` consecutive windows of ``length`` samples from ``start`` (the last one
    ``last_length`` long if given): the pilot's blocks, the tiles' bodies.  float64 numpy array."""
    import torch
    out = np.zeros(count, dtype=np.float64)
    if count == 0:
        return out
    full = count if last_length < 0 or last_length == length else count - 1
    step = max(1, (1 << 26) // max(1, length))
    for c0 in range(0, full, step):
        c1 = min(full, c0 + step)
        x = iq[start + c0 * length: start + c1 * length].reshape(c1 - c0, length, 2).to(torch.float32)
        x = x - x.mean(dim=1, keepdim=True)
        out[c0:c1] = (x * x).sum(dim=2).mean(dim=1).double().cpu().numpy()
    if full < count and last_length > 0:
        x = iq[start + full * length: start + full * length + last_length].to(torch.float32)
        x = x - x.mean(dim=0, keepdim=True)
        out[full] = float((x * x).sum(dim=1).mean())
    return out


def _agc_step(g: float, c: float, power: float, nsym: float) -> float:
    """One window of the reference's AGC (agc.c:13-25: gain += 1e-4 * (190 - gain * |y|) per symbol) in closed form:
    towards g* = 190 / E|y| = c / sqrt(power) at the relative rate 1e-4 * 190 / g* per symbol."""
    gstar = c / float(np.sqrt(max(power, 1e-30)))
    return gstar + (g - gstar) * float(np.exp(-min(50.0, 1e-4 * 190.0 / max(gstar, 1e-30) * nsym)))


def fit_agc_calibration(gains, powers, nsyms) -> float:
    """The constant c in g* = c / sqrt(sample power), fitted on the pilot: ``gains[j]`` is the pilot's gain after block j,
    ``powers[j]`` / ``nsyms[j]`` the block's sample power and symbol count.  c is chosen so that the closed-form recursion
    started from gains[j0] reproduces gains[-1] (bisection; the recursion is monotone in c).  With a fast AGC (s16-scale
    input) this is simply gain * sqrt(power) of the last block; with a slow one (float input around +-1: time constant of
    seconds) it removes the lag the pilot's gain has whenever the amplitude is moving at the hand-over."""
    J = len(gains) - 1
    c0 = float(gains[J]) * float(np.sqrt(max(powers[J], 1e-30)))
    j0 = max(0, J - 8)
    if J == j0 or not np.isfinite(c0) or c0 <= 0:
        return c0

    def model(c):
        g = float(gains[j0])
        for j in range(j0 + 1, J + 1):
            g = _agc_step(g, c, float(powers[j]), float(nsyms[j]))
        return g
    lo, hi = c0 / 8, c0 * 8
    if not (model(lo) <= gains[J] <= model(hi)):
        return c0
    for _ in range(50):
        mid = 0.5 * (lo + hi)
        if model(mid) < gains[J]:
            lo = mid
        else:
            hi = mid
    return 0.5 * (lo + hi)


def agc_trajectory(gain0: float, c: float, power_tiles, symbols_per_tile):
    """AGC gain of the serial run at the start of every tile: the closed-form recursion of :func:`_agc_step` from the
    pilot's last gain over the tiles' sample powers.  With s16-scale input every tile simply gets its own equilibrium
    (and its warm-up converges anyway); with float input around +-1 the recursion follows the serial run's lag."""
    g = float(gain0)
    out = np.empty(len(power_tiles), dtype=np.float32)
    for i, (p, n) in enumerate(zip(power_tiles, symbols_per_tile)):
        out[i] = g
        g = _agc_step(g, c, float(p), float(n))
    return out


def agc_settle_symbols(gain: float) -> float:
    """Symbols the reference's AGC needs to settle at this gain: its step is ABSOLUTE, gain += 1e-4 * (190 - |x|)
    (agc.c:13-25), so the relative speed is 1e-4 * 190 / gain per symbol: instant for s16-scale input (gain ~ 0.03), but
    ~35 000 symbols per time constant for float input around +-1 (gain ~ 650) - and while the gain is still small the PLL's
    lock flag is already true (its error scales with the amplitude).  Six time constants; the pilot does not hand over
    earlier, or every tile would start from a gain the serial run has long left behind."""
    return 6.0 * float(gain) / (1e-4 * 190.0)


SEED_MIN_QUALITY = 8.0      # spectral line / mean of the searched band: noise alone gives 3-4, a 12 dB signal 40-50


def fill_weak_estimates(freq, quality, fallback: float, min_quality: float = SEED_MIN_QUALITY):
    """Tiles whose 4th-power spectrum shows no clear line (fade, interference: the peak is then a noise bin anywhere in
    +-fmax) take the estimate interpolated between their nearest good neighbours over the tile index; with no good tile
    at all, ``fallback`` (the pilot's frequency).  Returns a float32 tensor like ``freq``."""
    import torch
    f = freq.detach().cpu().numpy().astype(np.float64)
    good = quality.detach().cpu().numpy() >= min_quality
    if good.all():
        return freq
    if not good.any():
        return torch.full_like(freq, float(fallback))
    idx = np.arange(len(f))
    f = np.interp(idx, idx[good], f[good])
    return torch.as_tensor(f.astype(np.float32), device=freq.device)


# ---- result ---------------------------------------------------------------------------------

@dataclass
class StitchReport:
    n_tiles: int = 0
    pilot_samples: int = 0
    pilot_symbols: int = 0
    pilot_locked: bool = False
    first_lock_symbol: int = -1
    rotations: list = field(default_factory=list)          # absolute quarter turns per tile (pass 1)
    seam_shifts: list = field(default_factory=list)        # per seam: -1 / 0 / +1
    weak_seams: int = 0                                    # seams whose correlation was too weak to trust
    refine_rotations: list = field(default_factory=list)   # pass 2: residual rotation per tile (0 expected)
    samples_demodulated: int = 0                           # total kernel work incl. warm-up and pass 2
    carrier_seeds: list = field(default_factory=list)      # carrier_seed='spectrum': rad/symbol given to every tile
    gain_seeds: list = field(default_factory=list)         # ... and the AGC gain


@dataclass
class StitchedRecording:
    soft: object                   # int8 [m, 2] tensor: pilot symbols ++ tile symbols
    tile_first_symbol: np.ndarray  # index into soft of each tile's first symbol
    plan: TilePlan
    report: StitchReport


# ---- the stitcher -------------------------------------------------------------------------------

class RecordingDemodulator:
    """Demodulate ONE recording with many lanes.

    ``bank_factory(cfg, n_streams)`` must return an object with the :class:`Demodulator` methods used here
    (``reset, process, process_ragged, max_symbols, symbol_counts, get_state, set_state, get_history,
    set_history, set_state_all, rotate_carrier, close``).  The default is the HIP :class:`Demodulator`.

    ``carrier_seed``: "pilot" (default HERE) starts every tile from the pilot's carrier estimate, which makes the result a
    pure function of the tile engine: the HIP bank and the oracle bank then give the same bytes, which is what the tests
    pin.  "spectrum" gives every tile its own estimate (Doppler; the default of the C entry and of the CLI's ``--tiled``):
    the FFT's rounding depends on the device, so two engines agree only statistically.
    """

    def __init__(self, cfg, tile_samples: int = 0, pre_samples: int = -1, refine: bool = True,
                 pilot_block: int = 65536, pilot_margin_symbols: int = 20000, max_pilot_samples: int = 1 << 22,
                 match_symbols: int = 192, device: int = 0, bank_factory=None, post_samples: int = 4096,
                 carrier_seed: str = "pilot", acquire_symbols: int = 0):
        self.acquire_symbols = int(acquire_symbols)
        if cfg.oqpsk and not refine:
            # the I and Q rails of OQPSK come from different firings (demod.c:66-76): a 90 degree lock offset is
            # entangled with the half-symbol state of the symbol clock, which rotate_symbols() cannot undo on the
            # output; only the second pass (which turns the STATE, mdemod_rotate_carrier) handles it.
            raise NotImplementedError("overlapped tiles of an OQPSK recording need refine=True")
        if carrier_seed not in ("pilot", "spectrum"):
            raise NotImplementedError("carrier_seed is 'pilot' or 'spectrum'")
        self.carrier_seed = carrier_seed
        self.cfg = cfg
        self.post_samples = int(post_samples)
        tile_samples, pre_samples = default_tiling(cfg, tile_samples, pre_samples)
        self.tile_samples = int(tile_samples)
        self.pre_samples = int(pre_samples)
        self.refine = bool(refine)
        self.pilot_block = int(pilot_block)
        self.pilot_margin_symbols = int(pilot_margin_symbols)
        self.max_pilot_samples = int(max_pilot_samples)
        self.match_symbols = int(match_symbols)
        self.device = device
        if bank_factory is None:
            from .demod import Demodulator
            bank_factory = lambda c, n: Demodulator(c, n, device=device)
        self._factory = bank_factory

    # -- pilot ---------------------------------------------------------------------------------
    def _run_pilot(self, iq, rep: StitchReport):
        """Serial head: blocks of ``pilot_block`` until locked for ``pilot_margin_symbols`` (or the cap)."""
        import torch
        n = iq.shape[0]
        pilot = self._factory(self.cfg, 1)
        parts, pos, locked_at = [], 0, None
        self._pilot_blocks = []                       # (samples, gain after, symbols after) per block: AGC calibration
        while pos < n:
            b = min(self.pilot_block, n - pos)
            soft = pilot.process(iq[pos:pos + b].unsqueeze(0))
            m = int(pilot.symbol_counts()[0])
            parts.append(soft[0, :m].clone())
            pos += b
            st = pilot.get_state(0)
            self._pilot_blocks.append((b, float(st.agc_gain), int(st.n_symbols)))
            if st.pll_locked and locked_at is None:
                locked_at = st.n_symbols
            if not st.pll_locked:
                locked_at = None
            if (locked_at is not None and st.n_symbols - locked_at >= self.pilot_margin_symbols
                    and st.n_symbols >= agc_settle_symbols(st.agc_gain)):
                break
            if pos >= self.max_pilot_samples:
                break
        st = pilot.get_state(0)
        rep.pilot_samples, rep.pilot_symbols = pos, int(st.n_symbols)
        rep.pilot_locked, rep.first_lock_symbol = bool(st.pll_locked), int(st.first_lock_symbol)
        rep.samples_demodulated += pos
        soft = torch.cat(parts) if parts else torch.empty((0, 2), dtype=torch.int8, device=iq.device)
        return pilot, soft, pos

    # -- main entry ------------------------------------------------------------------------------
    def demodulate(self, iq) -> StitchedRecording:
        """``iq``: [n, 2] tensor of the recording (device tensor for the HIP bank)."""
        import torch
        assert iq.dim() == 2 and iq.shape[1] == 2
        rep = StitchReport()
        pilot, pilot_soft, pilot_end = self._run_pilot(iq, rep)
        plan = plan_tiles(int(iq.shape[0]), pilot_end, self.tile_samples, self.pre_samples)
        T = plan.n_tiles
        rep.n_tiles = T
        if T == 0:
            pilot.close()
            return StitchedRecording(pilot_soft, np.zeros(0, np.int64), plan, rep)

        dev = iq.device
        K = self.match_symbols
        seed = pilot.get_state(0)
        seed_hist = pilot.get_history(0)
        bank = self._factory(self.cfg, T)
        i64 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.int64), device=dev)
        i32 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.int32), device=dev)

        # ---- pass 1: warm-up (dropped) then body, from the pilot's end state -----------------
        bank.set_state_all(seed)
        if self.carrier_seed == "spectrum":
            # Doppler: every tile starts from ITS OWN carrier estimate (4th-power spectrum of its warm-up and the samples
            # after it), moved to the first warm-up sample with the local slope, sweep direction = sign of the slope
            nfft = 1 << int(np.floor(np.log2(max(4096, min(self.tile_samples + self.pre_samples, 1 << 17)))))
            w0 = plan.starts - plan.pres
            fmid, qual = carrier_estimates(iq, w0, nfft, self.cfg.samplerate, self.cfg.symrate,
                                           nco_steps_per_symbol=2 if self.cfg.oqpsk else 1)
            # AGC gain seeds: calibrate g* = c / sqrt(power) on the pilot's last blocks, then follow the tiles' powers
            blocks = self._pilot_blocks[-10:]
            nb = len(blocks)
            b_len = blocks[0][0]
            first = pilot_end - sum(b for b, _, _ in blocks)
            p_blocks = window_power(iq, first, b_len, nb, blocks[-1][0])
            sym_after = [s_ for _, _, s_ in blocks]
            prev = self._pilot_blocks[-nb - 1][2] if len(self._pilot_blocks) > nb else 0
            n_blocks = np.diff(np.asarray([prev] + sym_after, dtype=np.float64))
            c_agc = fit_agc_calibration([g_ for _, g_, _ in blocks], p_blocks, n_blocks)
            p_tiles = window_power(iq, int(plan.starts[0]), int(plan.lens[0]), T, int(plan.lens[-1]))
            gains = agc_trajectory(float(seed.agc_gain), c_agc, p_tiles, plan.lens * (self.cfg.symrate / self.cfg.samplerate))
            fmid = fill_weak_estimates(fmid, qual, float(seed.pll_freq))
            dt_sym = self.tile_samples * self.cfg.symrate / self.cfg.samplerate            # symbols between tile starts
            slope = torch.zeros_like(fmid)
            if T > 2:
                slope[1:-1] = (fmid[2:] - fmid[:-2]) / (2 * dt_sym)
                slope[0], slope[-1] = slope[1], slope[-2]
            # the estimate belongs to the middle of the window actually used (clamped at the end of the recording)
            wstart = np.minimum(w0, max(0, int(iq.shape[0]) - nfft))
            back = torch.as_tensor((wstart + nfft / 2 - w0).astype(np.float32), device=fmid.device)
            f0 = fmid - slope * back * (self.cfg.symrate / self.cfg.samplerate)
            fmax = float(bank.carrier_fmax()) if hasattr(bank, "carrier_fmax") else 0.3
            f0 = f0.clamp(-fmax, fmax)
            bank.set_carrier_seeds(f0.to(torch.float32).contiguous(), torch.where(slope >= 0, 1, -1).to(torch.int32).contiguous())
            rep.carrier_seeds = f0.cpu().tolist()
            bank.set_gain_seeds(torch.as_tensor(gains, dtype=torch.float32, device=dev).contiguous())
            rep.gain_seeds = gains.tolist()
        cap_pre = max(1, bank.max_symbols(int(plan.pres.max())))
        cap = bank.max_symbols(int(plan.lens.max()))
        soft_pre = torch.zeros((T, cap_pre, 2), dtype=torch.int8, device=dev)
        soft1 = torch.zeros((T, cap, 2), dtype=torch.int8, device=dev)
        acq = np.minimum(int(self.acquire_symbols * self.cfg.samplerate / self.cfg.symrate), plan.pres)
        if self.acquire_symbols > 0:
            # two-stage warm-up: acquire carrier phase / symbol clock / gain, then put the two loop INTEGRATORS back on their seeds
            # (the acquisition transient kicks them and they need 8-16 k symbols to come back: tools/seed_convergence.py)
            soft_acq = torch.zeros((T, max(1, bank.max_symbols(int(acq.max()))), 2), dtype=torch.int8, device=dev)
            bank.process_ragged(iq, i64(plan.starts - plan.pres), i32(acq), soft_acq)
            if self.carrier_seed == "spectrum":
                bank.set_carrier_seeds(f0.to(torch.float32).contiguous(), torch.where(slope >= 0, 1, -1).to(torch.int32).contiguous())
            else:
                bank.set_carrier_seeds(torch.full((T,), float(seed.pll_freq), dtype=torch.float32), torch.full((T,), int(seed.pll_updown if hasattr(seed, "pll_updown") else seed.updown), dtype=torch.int32))
            bank.set_clock_seeds(torch.full((T,), float(seed.t_freq), dtype=torch.float32))
            bank.process_ragged(iq, i64(plan.starts - plan.pres + acq), i32(plan.pres - acq), soft_pre)
        else:
            bank.process_ragged(iq, i64(plan.starts - plan.pres), i32(plan.pres), soft_pre)
        cnt_pre = bank.symbol_counts().to(dev)
        bank.process_ragged(iq, i64(plan.starts), i32(plan.lens), soft1)
        cnt1 = bank.symbol_counts().to(dev)
        rep.samples_demodulated += int(plan.pres.sum() + plan.lens.sum())

        if self.cfg.oqpsk:
            return self._finish_oqpsk(iq, rep, plan, pilot, bank, pilot_soft, seed, seed_hist, soft_pre, cnt_pre, soft1, cnt1, cap)

        # ---- rotation + seam of every tile against its predecessor (tile 0: against the pilot) ----
        prev_tail, _ = gather_tails(soft1[:-1], cnt1[:-1], K + 1) if T > 1 else (torch.zeros((0, K + 1, 2), dtype=torch.int32, device=dev), None)
        ptail, _ = gather_tails(pilot_soft.unsqueeze(0), torch.tensor([pilot_soft.shape[0]], device=dev), K + 1)
        a_tail = torch.cat((ptail, prev_tail))
        b_tail, _ = gather_tails(soft_pre, cnt_pre, K + 1)
        shift, rot, score, energy = match_tails(a_tail, b_tail)
        weak = score * 2 < energy                                  # less than half of a perfect match
        weak |= torch.as_tensor(plan.pres == 0, device=dev)        # no overlap at all
        shift = torch.where(weak, torch.zeros_like(shift), shift)
        rot = torch.where(weak, torch.zeros_like(rot), rot)
        R = torch.cumsum(rot, 0) & 3                               # absolute rotation of each tile
        rep.rotations = R.cpu().tolist()
        rep.seam_shifts = shift.cpu().tolist()
        rep.weak_seams = int(weak.sum())

        if not self.refine:
            body = rotate_symbols(soft1, R)
            pre_last, _ = gather_tails(soft_pre, cnt_pre, 1)
            head_sym = rotate_symbols(pre_last.to(torch.int8), R)[:, 0]           # used where shift == -1
            out = self._assemble(pilot_soft, body, cnt1, shift, head_sym)
            if int(shift[0]) == 1 and rep.pilot_symbols:
                rep.pilot_symbols -= 1             # the pilot's last symbol was a duplicate of tile 0's first one
            self._gate_from_tiles(rep, bank, out[1], int(seed.n_symbols) + cnt_pre.cpu().numpy(), np.arange(T), cnt1.cpu().numpy())
            pilot.close(); bank.close()
            return StitchedRecording(out[0], out[1], plan, rep)

        # ---- pass 2: tile i+1 := exact continuation of tile i's pass-1 end state, in rotation 0 ----
        bank.rotate_carrier(((4 - R) & 3).to(torch.int32))
        # stream T-1 is free in pass 2: it takes over from the pilot and runs tile 0 (exact continuation)
        bank.set_state(T - 1, seed)
        bank.set_history(T - 1, seed_hist)
        starts2 = np.concatenate((plan.starts[1:], plan.starts[:1]))
        lens2 = np.concatenate((plan.lens[1:], plan.lens[:1]))
        soft2 = torch.zeros((T, cap, 2), dtype=torch.int8, device=dev)
        bank.process_ragged(iq, i64(starts2), i32(lens2), soft2)
        cnt2s = bank.symbol_counts().to(dev)
        rep.samples_demodulated += int(plan.lens.sum())
        # back to tile order: tile 0 came from stream T-1, tile i from stream i-1
        order = torch.cat((torch.tensor([T - 1], device=dev), torch.arange(T - 1, device=dev)))
        soft2, cnt2 = soft2[order], cnt2s[order]

        # seam i|i+1: tile i+1 continued from tile i's PASS-1 trajectory; what is emitted for tile i is its
        # PASS-2 body.  Compare the two tails of tile i (same samples) for a one-symbol disagreement.
        a2, _ = gather_tails(soft2, cnt2, K + 1)
        b1, _ = gather_tails(rotate_symbols(soft1, R), cnt1, K + 1)
        shift2, rot2, score2, energy2 = match_tails(a2, b1)
        weak2 = score2 * 2 < energy2
        shift2 = torch.where(weak2, torch.zeros_like(shift2), shift2)
        rep.refine_rotations = torch.where(weak2, torch.zeros_like(rot2), rot2).cpu().tolist()
        rep.weak_seams += int(weak2[:-1].sum())
        # express as "shift of tile i+1 against its predecessor" like pass 1: +1 => predecessor (A = pass-2
        # tile i) holds an extra symbol, -1 => the reference tail (B = pass-1 tile i) holds one more.
        seam = torch.cat((torch.zeros(1, dtype=shift2.dtype, device=dev), shift2[:-1]))
        b1_last = rotate_symbols(gather_tails(soft1, cnt1, 1)[0].to(torch.int8), R)[:, 0]
        head_sym = torch.cat((torch.zeros((1, 2), dtype=torch.int8, device=dev), b1_last[:-1]))
        rep.seam_shifts = seam.cpu().tolist()
        out = self._assemble(pilot_soft, soft2, cnt2, seam, head_sym)
        n_start = np.concatenate(([0], (cnt_pre + cnt1).cpu().numpy()[:-1])) + int(seed.n_symbols)
        self._gate_from_tiles(rep, bank, out[1], n_start, (np.arange(T) - 1) % T, cnt2.cpu().numpy())
        pilot.close(); bank.close()
        return StitchedRecording(out[0], out[1], plan, rep)

    # -- OQPSK: rotation from per-rail correlation, second pass with a look-ahead into the next tile ---------------
    def _finish_oqpsk(self, iq, rep, plan, pilot, bank, pilot_soft, seed, seed_hist, soft_pre, cnt_pre, soft1, cnt1, cap):
        import torch
        dev, T, K = iq.device, plan.n_tiles, self.match_symbols
        i64 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.int64), device=dev)
        i32 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.int32), device=dev)
        prev_tail, _ = gather_tails(soft1[:-1], cnt1[:-1], K + 2) if T > 1 else (torch.zeros((0, K + 2, 2), dtype=torch.int32, device=dev), None)
        ptail, _ = gather_tails(pilot_soft.unsqueeze(0), torch.tensor([pilot_soft.shape[0]], device=dev), K + 2)
        b_tail, _ = gather_tails(soft_pre, cnt_pre, K + 2)
        rot, score, energy = match_rails(torch.cat((ptail, prev_tail)), b_tail)
        weak = (score * 2 < energy) | torch.as_tensor(plan.pres == 0, device=dev)
        rot = torch.where(weak, torch.zeros_like(rot), rot)
        R = torch.cumsum(rot, 0) & 3
        rep.rotations = R.cpu().tolist()
        rep.weak_seams = int(weak.sum())

        # pass 2: stream i continues from its pass-1 end state, turned into the pilot's rotation (carrier AND
        # half-symbol pairing), with tile i+1 and then `post` samples of tile i+2 for the seam check
        bank.rotate_carrier(((4 - R) & 3).to(torch.int32))
        bank.set_state(T - 1, seed)
        bank.set_history(T - 1, seed_hist)
        starts2 = np.concatenate((plan.starts[1:], plan.starts[:1]))
        lens2 = np.concatenate((plan.lens[1:], plan.lens[:1]))
        ends2 = starts2 + lens2
        post2 = np.minimum(self.post_samples, plan.n_samples - ends2)
        soft2 = torch.zeros((T, cap, 2), dtype=torch.int8, device=dev)
        bank.process_ragged(iq, i64(starts2), i32(lens2), soft2)
        cnt2s = bank.symbol_counts().to(dev)
        cap_post = max(1, bank.max_symbols(int(post2.max())))
        soft_post = torch.zeros((T, cap_post, 2), dtype=torch.int8, device=dev)
        bank.process_ragged(iq, i64(ends2), i32(post2), soft_post)
        cnt_posts = bank.symbol_counts().to(dev)
        rep.samples_demodulated += int(lens2.sum() + post2.sum())
        order = torch.cat((torch.tensor([T - 1], device=dev), torch.arange(T - 1, device=dev)))
        soft2, cnt2, soft_post, cnt_post = soft2[order], cnt2s[order], soft_post[order], cnt_posts[order]

        # seam i | i+1: tile i's look-ahead and tile i+1's body start on the same sample
        seam = torch.zeros(T, dtype=torch.int64, device=dev)
        head_sym = torch.zeros((T, 2), dtype=torch.int8, device=dev)
        if T > 1:
            a = gather_heads(soft_post[:-1], cnt_post[:-1], K + 1)
            b = gather_heads(soft2[1:], cnt2[1:], K + 1)
            sh, r2, sc2, en2 = match_heads(a, b)
            weak2 = sc2 * 2 < en2
            sh = torch.where(weak2, torch.zeros_like(sh), sh)
            rep.refine_rotations = [0] + torch.where(weak2, torch.zeros_like(r2), r2).cpu().tolist()
            rep.weak_seams += int(weak2.sum())
            # +1: tile i fired a symbol right after the boundary that tile i+1 does not have: insert it in front of tile
            # i+1; -1: tile i+1 starts with a symbol tile i already emitted before the boundary: drop tile i's last one
            # (the same repair _assemble() applies, with the signs of its tail convention)
            seam[1:] = -sh
            head_sym[1:] = soft_post[:-1, 0]
        rep.seam_shifts = seam.cpu().tolist()
        out = self._assemble(pilot_soft, soft2, cnt2, seam, head_sym)
        n_start = np.concatenate(([0], (cnt_pre + cnt1).cpu().numpy()[:-1])) + int(seed.n_symbols)
        self._gate_from_tiles(rep, bank, out[1], n_start, (np.arange(T) - 1) % T, cnt2.cpu().numpy())
        pilot.close(); bank.close()
        return StitchedRecording(out[0], out[1], plan, rep)

    @staticmethod
    def _gate_from_tiles(rep, bank, first_out, n_start, body_stream, keep):
        """The pilot never locked (a recording that starts before the signal does): the lock gate (main.c:308-315) opens at
        the first tile whose stream reports a first lock - inside its emitted body, or before it (then the whole body
        counts).  ``n_start[i]``: the stream's symbol count when tile i's emitted body began.  Approximate to the tiles'
        own acquisition, which is faster than the serial sweep (same rule as csrc/recording.hip)."""
        if rep.first_lock_symbol >= 0:
            return
        for i in range(len(first_out)):
            fl = int(bank.get_state(int(body_stream[i])).first_lock_symbol)
            if fl < 0:
                continue
            inside = max(0, fl - int(n_start[i]))
            rep.first_lock_symbol = int(first_out[i]) + min(inside, int(keep[i]))
            return

    # -- concatenation with seam fixes -------------------------------------------------------------
    @staticmethod
    def _assemble(pilot_soft, body, cnt, shift, head_sym):
        """pilot ++ tiles.  shift[i] = +1: the predecessor of tile i ends with a symbol tile i also emits
        (drop it from the predecessor); -1: the symbol before tile i's first one is missing (insert
        ``head_sym[i]``)."""
        import torch
        T, cap = body.shape[0], body.shape[1]
        dev = body.device
        cnt = cnt.to(torch.int64)
        drop_prev = (shift == 1).to(torch.int64)
        add_head = (shift == -1).to(torch.int64)
        n_pilot = pilot_soft.shape[0] - int(drop_prev[0])
        keep = cnt - torch.cat((drop_prev[1:], torch.zeros(1, dtype=torch.int64, device=dev)))
        keep = keep.clamp(min=0)
        per_tile = keep + add_head
        first = n_pilot + torch.cumsum(per_tile, 0) - per_tile
        total = int(n_pilot + per_tile.sum())
        out = torch.empty((total, 2), dtype=torch.int8, device=dev)
        out[:n_pilot] = pilot_soft[:n_pilot]
        hs = torch.nonzero(add_head).flatten()
        if hs.numel():
            out[first[hs]] = head_sym[hs]
        col = torch.arange(cap, device=dev).view(1, cap)
        mask = col < keep.view(T, 1)
        dst = (first + add_head).view(T, 1) + col
        out[dst[mask]] = body[mask]
        return out, first.cpu().numpy()


# ---- the same scheme inside the library (csrc/recording.hip) --------------------------------------

def estimate_carrier_native(cfg, iq, starts, window_samples: int, device: int = 0):
    """``mdemod_estimate_carrier`` on a device tensor [n, 2]: (freq [T] float32 rad per NCO step at the middle of each
    window, quality [T], window length actually used)."""
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
    stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    _capi.check(lib.mdemod_estimate_carrier(C.byref(p), C.c_void_p(iq.data_ptr()), int(iq.shape[0]), C.c_void_p(st.data_ptr()), T,
                                            int(window_samples), C.c_void_p(freq.data_ptr()), C.c_void_p(qual.data_ptr()), stream),
                "mdemod_estimate_carrier")
    used = int(lib.mdemod_carrier_window_samples(C.byref(p), int(window_samples)))
    torch.cuda.current_stream(device).synchronize()
    return freq, qual, used


def demodulate_recording_native(cfg, iq, tile_samples: int = 0, pre_samples: int = -1, refine: bool = True,
                                pilot_block: int = 65536, pilot_margin_symbols: int = 20000,
                                max_pilot_samples: int = 1 << 22, match_symbols: int = 192, device: int = 0,
                                carrier_seed: str = "pilot", soft_capacity: int = 0):
    """``mdemod_demodulate_recording`` on a device tensor [n, 2]: returns (soft int8 [m, 2] device tensor, report).
    ``soft_capacity`` (symbols; 0 = nominal rate + 5 % + 65536) is what the callee checks its output against."""
    import ctypes as C
    import torch
    from . import _capi
    lib = _capi.lib()
    assert iq.is_cuda and iq.dim() == 2 and iq.shape[1] == 2 and iq.is_contiguous()
    tile_samples, pre_samples = default_tiling(cfg, tile_samples, pre_samples)
    opts = _capi.MdemodRecordingOpts(tile_samples, pre_samples, pilot_block, pilot_margin_symbols, max_pilot_samples,
                                     match_symbols, int(refine), 1 if carrier_seed == "spectrum" else 0, 0)
    n = int(iq.shape[0])
    p = cfg.to_c(1, device)
    cap = int(soft_capacity) or int(n * cfg.symrate / cfg.samplerate * 1.05) + 65536   # stitched output: nominal rate + slack (checked by the callee)
    soft = torch.empty((cap, 2), dtype=torch.int8, device=iq.device)
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
