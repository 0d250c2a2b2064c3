"""CPU experiment (oracle only): how fast does a tile started from a seed state converge onto the serial run, and which
loop variable is the slow one?  Reproduces SURVEY §7 H2's table with this repo's synthetic signal.

Usage: seed_convergence.py [c1|c3|c4] [log2 samples]
"""
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
import oracle_py as O
from oracle_bank import Snapshot
from meteor_demod_amd import DemodConfig, synth

tag = sys.argv[1] if len(sys.argv) > 1 else "c1"
n = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 24)
if tag == "c1":
    cfg = DemodConfig(samplerate=230000)
elif tag == "c3":
    cfg = DemodConfig(samplerate=230000, symrate=80000, oqpsk=True)
else:
    cfg = DemodConfig(samplerate=1000000, rrc_order=64, interp_factor=8)
st = synth.make_stream(1000, cfg.samplerate, cfg.symrate, oqpsk=cfg.oqpsk, f0_hz=1200.0, clock_ppm=-3.5)
iq = synth.generate_host(st, n)

# serial run in blocks with state snapshots
blk = 1 << 16
ser = O.OracleStream(cfg)
soft_parts, tr_parts, snaps = [], [], {}
for p in range(0, n, blk):
    snaps[p] = (Snapshot(ser.state), ser.history())
    s, t, _ = ser.run(iq[p:p + blk], want_trace=True)
    t["sample_index"] += 0
    soft_parts.append(s); tr_parts.append(t)
soft = np.concatenate(soft_parts); tr = np.concatenate(tr_parts)
fl = ser.state.first_lock_symbol
print(f"{tag}: {n} samples, {len(soft)} symbols, first lock {fl}")
sidx = tr["sample_index"].astype(np.int64)


def run_tile(seed_pos, start, length, hist=False, tweak=None):
    t = O.OracleStream(cfg)
    snap, h = snaps[seed_pos]
    snap.apply(t._p.contents.s)
    if tweak:
        tweak(t._p.contents.s)
    s, trc, _ = t.run(iq[start:start + length], want_trace=True)
    trc = trc.copy(); trc["sample_index"] = trc["sample_index"] - snap.n_samples + start
    return s, trc


def compare(trc, s, label, windows=((0, 100), (300, 1000), (1000, 2000), (2000, 4000), (4000, 8000), (8000, 16000), (16000, 32000), (32000, 64000), (64000, 128000))):
    # align on the sample index of the firing
    pos = np.searchsorted(sidx, trc["sample_index"].astype(np.int64))
    pos = np.clip(pos, 0, len(sidx) - 1)
    same = sidx[pos] == trc["sample_index"]
    out = []
    for a, b in windows:
        if a >= len(trc):
            break
        sl = slice(a, min(b, len(trc)))
        ok = same[sl]
        p = pos[sl][ok]
        d = np.abs(s[sl][ok].astype(np.int16) - soft[p].astype(np.int16)).max(axis=1)
        # rotation-agnostic best over 4 rotations is not needed: report as is
        w1 = float((d <= 1).mean()) if len(d) else 0.0
        df = float(np.median(np.abs(trc["pll_freq"][sl][ok] - tr["pll_freq"][p]))) if ok.any() else 0
        dom = float(np.median(np.abs(trc["omega"][sl][ok] - tr["omega"][p]))) if ok.any() else 0
        dg = float(np.median(np.abs(trc["gain"][sl][ok] / tr["gain"][p] - 1))) if ok.any() else 0
        dre = float(np.median(np.hypot(trc["re"][sl][ok] - tr["re"][p], trc["im"][sl][ok] - tr["im"][p]))) if ok.any() else 0
        out.append(f"[{a},{b}): same-sample {ok.mean():.4f} within1 {w1:.4f} med|dy| {dre:.3f} d_pllf {df:.2e} d_omega {dom:.2e} d_gain {dg:.1e}")
    print(label)
    for o in out:
        print("   ", o)


osf = cfg.samplerate / cfg.symrate
# pilot end as the stitcher picks it: lock + 20000 symbols, rounded up to a block
pilot_sym = fl + 20000
pilot_pos = int(np.ceil(sidx[pilot_sym] / blk) * blk)
late_pos = (n // 2 // blk) * blk
for seed_pos, name in ((pilot_pos, "seed = pilot end (lock + 20k symbols)"), (late_pos, "seed = serial state at n/2")):
    for start in (late_pos + 7 * blk + 12345, ):
        length = min(n - start, int(130000 * osf))
        s, trc = run_tile(seed_pos, start, length)
        compare(trc, s, f"{name} (sample {seed_pos}), tile starts at sample {start}")
# which variable is the slow one: seed at n/2, replace single fields by the pilot-end values
snapP = snaps[pilot_pos][0]
for field in ("pll_freq", "t_freq", "gain"):
    def tw(sx, field=field):
        setattr(sx, field, getattr(snapP, field))
    start = late_pos + 7 * blk + 12345
    s, trc = run_tile(late_pos, start, min(n - start, int(130000 * osf)), tweak=tw)
    compare(trc, s, f"seed = serial at n/2 but {field} from the pilot end")
print("serial pll_freq at pilot end / n/2 / later:", snapP.pll_freq, snaps[late_pos][0].pll_freq, tr['pll_freq'][-1])
print("serial t_freq  at pilot end / n/2 / later:", snapP.t_freq, snaps[late_pos][0].t_freq, tr['omega'][-1])
w = 20000
pf = tr["pll_freq"][fl:]; om = tr["omega"][fl:]
print("pll_freq per 20k-symbol window after lock:", [f"{pf[i:i+w].mean():.6f}" for i in range(0, min(len(pf), 400000), w)])
print("omega    per 20k-symbol window after lock:", [f"{om[i:i+w].mean():.8f}" for i in range(0, min(len(om), 400000), w)])
print("std of pll_freq / omega over the second half:", pf[len(pf)//2:].std(), om[len(om)//2:].std(), "omega center", om.mean())

# ---- two-stage warm-up: acquire (phase, timing, gain), then put the two integrators back on their seeds -------------------
print("\n== two-stage warm-up ==")
def run_two_stage(seed_pos, start, length, n_a_samples, pll_freq_seed, t_freq_seed=None):
    t = O.OracleStream(cfg)
    snap, h = snaps[seed_pos]
    snap.apply(t._p.contents.s)
    sx = t._p.contents.s
    sx.pll_freq = float(np.float32(pll_freq_seed))
    tf = sx.t_freq if t_freq_seed is None else t_freq_seed
    parts = []
    s0, tr0, _ = t.run(iq[start:start + n_a_samples], want_trace=True)
    sx.pll_freq = float(np.float32(pll_freq_seed)); sx.t_freq = tf
    s1, tr1, _ = t.run(iq[start + n_a_samples:start + length], want_trace=True)
    s = np.concatenate((s0, s1)); trc = np.concatenate((tr0, tr1))
    trc["sample_index"] = trc["sample_index"] - snap.n_samples + start
    return s, trc

start = late_pos + 7 * blk + 12345
true_f = float(tr["pll_freq"][np.searchsorted(sidx, start)])
for n_a_sym in (0, 1000, 2000, 4000):
    for ferr in (0.0, 2e-5):
        s, trc = run_two_stage(pilot_pos, start, min(n - start, int(70000 * osf)), int(n_a_sym * osf), true_f + ferr)
        compare(trc, s, f"pilot-end seed, carrier seed = serial's local value {ferr:+.0e}, integrators re-seeded after {n_a_sym} symbols")
