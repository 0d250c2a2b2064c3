"""Soak test of the other entry points with random option combinations: ragged device launches (random offsets, empty and
short streams), the pipelined host path, state/history hand-off between contexts.  GPU vs oracle, byte for byte.
Usage: api_fuzz.py [n_configs] [seed]"""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch
import oracle_py as O
from meteor_demod_amd import DemodConfig, Demodulator, synth

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
DT = {8: np.uint8, 16: np.int16, 32: np.float32}
bad, t0, done = [], time.time(), 0
for ci in range(n_cfg):
    symrate = int(rng.choice([72000, 80000, 36000]))
    osf = float(rng.choice([2.0, 2.875, 3.1944, 3.6, 5.0, 9.0, 13.9]))
    cfg = DemodConfig(samplerate=int(symrate * osf * (1 + rng.uniform(-0.01, 0.01))), symrate=symrate, oqpsk=bool(rng.random() < 0.35),
                      rrc_order=int(rng.choice([8, 16, 32, 33, 48, 64, 70])), interp_factor=int(rng.choice([1, 2, 3, 4, 5, 8])),
                      bps=int(rng.choice([8, 16, 16, 32])))
    if not np.isfinite(O.OracleStream(cfg).rrc_table()).all():
        continue
    rms = {8: 50.0, 16: 5000.0, 32: 0.7}[cfg.bps]
    ns = int(rng.integers(1, 40))
    lens = [int(rng.choice([0, 1, 5, 63, 64, 65, 130, 1000, 3001, 7000])) for _ in range(ns)]
    src = [synth.generate_host(synth.make_stream(int(rng.integers(1 << 30)), cfg.samplerate, cfg.symrate, f0_hz=float(rng.uniform(-1500, 1500)),
                                                 esn0_db=15.0, rms=rms, oqpsk=cfg.oqpsk, fmt=cfg.bps), 9000) for _ in range(4)]
    iqs = [src[i % 4][:lens[i]] for i in range(ns)]
    want = [O.oracle_demod(cfg, a)[0] if len(a) else np.zeros((0, 2), np.int8) for a in iqs]
    why = None
    try:
        # ---- ragged device launch with unaligned offsets ----
        offs, pos = [], int(rng.integers(0, 5))
        for a in iqs:
            offs.append(pos); pos += len(a) + int(rng.integers(0, 4))
        flat = np.zeros((pos + 8, 2), dtype=DT[cfg.bps]) + (128 if cfg.bps == 8 else 0)
        for o, a in zip(offs, iqs):
            flat[o:o + len(a)] = a
        with Demodulator(cfg, ns) as d:
            soft = torch.zeros((ns, d.max_symbols(max(lens + [1])), 2), dtype=torch.int8, device="cuda")
            d.process_ragged(torch.from_numpy(flat.astype(DT[cfg.bps])).cuda(), torch.tensor(offs, dtype=torch.int64).cuda(),
                             torch.tensor(lens, dtype=torch.int32).cuda(), soft)
            torch.cuda.synchronize()
            cnt = d.symbol_counts()
            for i in range(ns):
                if not np.array_equal(soft[i, : int(cnt[i])].cpu().numpy(), want[i]):
                    why = f"ragged stream {i} len {lens[i]}"; break
            # ---- hand the state of stream 0 to a fresh context and continue ----
            if why is None and lens[0] >= 130:
                with Demodulator(cfg, 3) as e:
                    e.set_state(1, d.get_state(0)); e.set_history(1, d.get_history(0))
                    more = src[0][lens[0]: lens[0] + 1500]
                    s2 = e.process(torch.from_numpy(np.stack([more] * 3)).cuda()); torch.cuda.synchronize()
                    w2 = O.oracle_demod(cfg, src[0][: lens[0] + 1500])[0][len(want[0]):]
                    if not np.array_equal(s2[1, : int(e.symbol_counts()[1])].cpu().numpy(), w2):
                        why = "state hand-off"
        # ---- host path ----
        if why is None:
            with Demodulator(cfg, ns) as d:
                outs = d.process_host(iqs)
                for i in range(ns):
                    if not np.array_equal(outs[i], want[i]):
                        why = f"host stream {i} len {lens[i]}"; break
        # ---- host path from rows the caller pinned (round 5): equal lengths, one stride apart, inside ONE pinned array - copied in
        #      place, no staging; then a ragged call on the same context (staged), chained on the first ----
        if why is None:
            n_u = int(rng.choice([64, 1000, 4097, 9000 - 1500]))
            pad = int(rng.choice([0, 1, 3, 8, 100]))
            big = np.zeros((ns, n_u + pad, 2), dtype=DT[cfg.bps]) + (128 if cfg.bps == 8 else 0)
            for i in range(ns):
                big[i, :n_u] = src[i % 4][:n_u]
            big = big.astype(DT[cfg.bps])
            w_u = [O.OracleStream(cfg) for _ in range(min(ns, 4))]
            first = [w_u[i].run(src[i][:n_u])[0] for i in range(len(w_u))]
            with Demodulator(cfg, ns) as d:
                d.pin_host(big)
                outs = d.process_host([big[i, :n_u] for i in range(ns)])
                for i in range(ns):
                    if not np.array_equal(outs[i], first[i % 4] if i % 4 < len(first) else outs[i % 4]):
                        why = f"pinned host stream {i}"; break
                if why is None:
                    l2 = [int(rng.choice([0, 7, 500, 1499])) for _ in range(ns)]
                    outs2 = d.process_host([src[i % 4][n_u: n_u + l2[i]] for i in range(ns)])
                    for i in range(min(ns, 4)):
                        if not np.array_equal(outs2[i], w_u[i].run(src[i][n_u: n_u + l2[i]])[0] if l2[i] else np.zeros((0, 2), np.int8)):
                            why = f"chained after pinned, stream {i} len {l2[i]}"; break
                d.unpin_host(big)
    except Exception as ex:
        why = repr(ex)
    done += 1
    if why:
        bad.append((ci, cfg, ns, why))
print(f"{done} configs in {time.time()-t0:.0f} s; failures: {len(bad)}")
for b in bad[:10]:
    print("  BAD", b)
