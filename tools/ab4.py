"""A/B of library builds on one box (round 4): for every library (MDEMOD_LIB_PATH, "" = the product) one child process that times
the bench shape of c1 / c3 / c4 (393216 tiles x 16448 samples, events around each launch) and compares sampled tiles with the oracle.
    python tools/ab4.py [--configs c1,c3,c4] [--steps 6] lib1.so lib2.so ...       ("default" = the product library)"""
import json, os, subprocess, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')

def child(configs, steps):
    import numpy as np, torch
    import oracle_py as O
    from bench import demod_config
    from meteor_demod_amd import Demodulator, synth
    T, L = 393216, 16448
    from meteor_demod_amd import DemodConfig
    extra = {"x3": DemodConfig(samplerate=1024000, bps=32), "x4": DemodConfig(samplerate=1000000, rrc_order=64, interp_factor=8, bps=32),
             "x5": DemodConfig(samplerate=2048000, bps=32), "x1": DemodConfig(samplerate=1024000), "x2": DemodConfig(samplerate=1800000)}
    T0 = T
    for tag in configs:
        cfg = extra[tag] if tag in extra else demod_config(tag)[0]
        T = T0 // 2 if cfg.bps == 32 else T0
        rec = synth.make_stream(1000, cfg.samplerate, cfg.symrate, oqpsk=cfg.oqpsk, f0_hz=1200.0, clock_ppm=-3.5, fmt=cfg.bps,
                                **(dict(rms=0.25, dc=(0.001, -0.002)) if cfg.bps == 32 else {}))
        buf = torch.empty((T * L, 2), dtype=torch.float32 if cfg.bps == 32 else torch.int16, device="cuda")
        synth.generate_device([rec], T * L, out=buf.view(1, T * L, 2))
        x = buf.view(T, L, 2)
        with Demodulator(cfg, T) as d:
            soft = torch.empty((T, d.max_symbols(L), 2), dtype=torch.int8, device="cuda")
            for _ in range(2):
                d.process(x, soft=soft)
            ms = []
            for _ in range(steps):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); d.process(x, soft=soft); b.record(); torch.cuda.synchronize()
                ms.append(a.elapsed_time(b))
            d.reset()
            d.process(x, soft=soft); torch.cuda.synchronize()
            bad = 0
            picks = [0, 1, 63, 64, 255, 256, 511, 512, T // 2, T - 513, T - 1, 77777, T // 2 + 3393]
            for t in picks:
                st = d.status(int(t), 1)[0]
                want = O.oracle_demod(cfg, x[int(t)].cpu().numpy())[0]
                got = soft[int(t), : st.symbols_this_call].cpu().numpy()
                if got.shape != want.shape or not np.array_equal(got, want):
                    bad += 1
            print(json.dumps({"lib": os.environ.get("MDEMOD_LIB_PATH", "") or "default", "config": tag, "kernel": d.kernel_name,
                              "ms_min": round(min(ms), 3), "ms_med": round(sorted(ms)[len(ms) // 2], 3),
                              "gsps_med": round(T * L / sorted(ms)[len(ms) // 2] / 1e6, 1), "bad_tiles": bad, "checked": len(picks)}), flush=True)
        del buf, x, soft
        torch.cuda.empty_cache()

if __name__ == "__main__":
    args = sys.argv[1:]
    configs, steps = ["c1", "c3", "c4"], 6
    if "--child" in args:
        child(args[args.index("--child") + 1].split(","), int(args[args.index("--child") + 2]))
        sys.exit(0)
    if "--configs" in args:
        i = args.index("--configs"); configs = args[i + 1].split(","); del args[i:i + 2]
    if "--steps" in args:
        i = args.index("--steps"); steps = int(args[i + 1]); del args[i:i + 2]
    for lib in args:
        env = dict(os.environ)
        if lib != "default":
            env["MDEMOD_LIB_PATH"] = lib
        else:
            env.pop("MDEMOD_LIB_PATH", None)
        r = subprocess.run([sys.executable, __file__, "--child", ",".join(configs), str(steps)], env=env, capture_output=True, text=True)
        sys.stdout.write(r.stdout)
        if r.returncode:
            print(json.dumps({"lib": lib, "error": r.stderr[-600:]}))
        sys.stdout.flush()
