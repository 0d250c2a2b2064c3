"""Builds the native pieces IN-TREE with hipcc for gfx950.

    meteor_demod_amd/lib/libmeteor_demod_amd.so   the product: HIP kernels + C-ABI
    meteor_demod_amd/lib/libmdemod_synth.so       synthetic IQ generator (host + device)
    meteor_demod_amd/lib/meteor_demod_amd         the C host CLI (drop-in for the reference's main.c)

The oracle (test infrastructure) is built separately by oracle/Makefile; this
module never touches it.  `python -m meteor_demod_amd.build` rebuilds everything.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
LIB = PKG / "lib"
ARCH = "gfx950"

# -ffp-contract=off is part of the numerical specification: the
# reference's output is only reproducible without FMA contraction.
COMMON = ["-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall", "-Wno-unused-function",
          f"--offload-arch={ARCH}", f"-I{ROOT / 'include'}"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


def _stale(target: Path, deps: list[Path]) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(d.stat().st_mtime > t for d in deps)


def _run(cmd: list[str]) -> None:
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        sys.stderr.write(proc.stdout + proc.stderr)
        raise RuntimeError("build failed: " + " ".join(cmd))


ROT_FLAGS = ["-fno-slp-vectorize", "-Wno-inline-asm", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]
ROTP_FLAGS = ["-fno-slp-vectorize", "-Wno-inline-asm"]


def check_rot_partition() -> list[str]:
    """Compiles demod_kernel_rot.hip to assembly and returns the compiler-generated instructions (outside the #ASMSTART/#ASMEND
    blocks) that touch a VGPR at or above ROTWIN_LIMIT - there must be none (see the kernel's header comment)."""
    import re
    import tempfile
    limit = int(re.search(r"#define ROTWIN_LIMIT (\d+)", (CSRC / "rotwin_asm.h").read_text()).group(1))
    alimit = int(re.search(r"#define ROTWIN_AB (\d+)", (CSRC / "rotwin_asm.h").read_text()).group(1))
    with tempfile.TemporaryDirectory() as td:
        out = Path(td) / "rot.s"
        _run([_hipcc(), *COMMON, *ROT_FLAGS, "-x", "hip", "--offload-device-only", "-S", str(CSRC / "demod_kernel_rot.hip"), "-o", str(out)])
        bad, inasm = [], False
        for n, line in enumerate(out.read_text().split("\n"), 1):
            if "#ASMSTART" in line:
                inasm = True
            elif "#ASMEND" in line:
                inasm = False
            elif not inasm:
                t = line.strip()
                if not t or t[0] in ";." or t.endswith(":"):
                    continue
                regs = [int(m.group(1)) for m in re.finditer(r"\bv(\d+)\b", t)] + [int(m.group(2)) for m in re.finditer(r"\bv\[(\d+):(\d+)\]", t)]
                aregs = [int(m.group(1)) for m in re.finditer(r"\ba(\d+)\b", t)] + [int(m.group(2)) for m in re.finditer(r"\ba\[(\d+):(\d+)\]", t)]
                if any(r >= limit for r in regs) or any(r >= alimit for r in aregs):      # (the AccVGPR half of the hybrid window)
                    bad.append(f"{n}: {t}")
    return bad


def build(force: bool = False, verbose: bool = False) -> dict[str, Path]:
    LIB.mkdir(exist_ok=True)
    hipcc = _hipcc()
    headers = sorted(CSRC.glob("*.h")) + [ROOT / "include" / "meteor_demod_amd.h"]
    out: dict[str, Path] = {}

    # --- product library ---------------------------------------------------
    # (source, object stem, extra flags).  The register-window kernel file is compiled twice: the std geometry with scalar
    # f32 ops (packed v_pk_* are slower on gfx950) and the max-ILP machine scheduler (measured +1.7 % on configs[1]); the
    # wide geometry with the default scheduler (max-ILP makes its u8 variants spill inside the main loop).
    rw = CSRC / "demod_kernel_rw.hip"
    units = [(CSRC / "demod_kernel.hip", "demod_kernel", []),
             (rw, "demod_kernel_rw_std", ["-fno-slp-vectorize", "-DMDEMOD_RW_PART=1", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]),
             (rw, "demod_kernel_rw_wide", ["-fno-slp-vectorize", "-DMDEMOD_RW_PART=2"]),
             (CSRC / "demod_kernel_rot.hip", "demod_kernel_rot", ROT_FLAGS),
             (CSRC / "demod_kernel_rotp.hip", "demod_kernel_rotp", ROTP_FLAGS),
             (CSRC / "demod_kernel_gat.hip", "demod_kernel_gat", ["-fno-slp-vectorize"]),
             (CSRC / "demod_kernel_lat.hip", "demod_kernel_lat", ["-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]),
             (CSRC / "demod_aux.hip", "demod_aux", []), (CSRC / "recording.hip", "recording", []),
             (CSRC / "demod_api.cpp", "demod_api", []), (CSRC / "host_pipe.cpp", "host_pipe", []),
             (CSRC / "demod_host.cpp", "demod_host", [])]
    # the assembly of the rotating register window is generated (csrc/gen_rotwin_asm.py -> csrc/rotwin_asm.h)
    for gen, inc in ((CSRC / "gen_rotwin_asm.py", CSRC / "rotwin_asm.h"), (CSRC / "gen_rotpk_asm.py", CSRC / "rotpk_asm.h")):
        if force or _stale(inc, [gen]):
            text = subprocess.run([sys.executable, str(gen)], capture_output=True, text=True, check=True).stdout
            inc.write_text(text)
            headers = sorted(CSRC.glob("*.h")) + [ROOT / "include" / "meteor_demod_amd.h"]
    objs = []
    for src, stem, extra in units:
        obj = LIB / (stem + ".o")
        if force or _stale(obj, [src] + headers):
            if verbose:
                print("hipcc", src.name, "->", obj.name, flush=True)
            _run([hipcc, *COMMON, *extra, "-x", "hip", "-c", str(src), "-o", str(obj)])
        objs.append(obj)
    so = LIB / "libmeteor_demod_amd.so"
    if force or _stale(so, objs):
        _run([hipcc, "-shared", "-fPIC", "-pthread", f"--offload-arch={ARCH}", "-o", str(so), *map(str, objs)])
    out["lib"] = so

    # --- synthetic signal generator -----------------------------------------
    synth = LIB / "libmdemod_synth.so"
    if force or _stale(synth, [CSRC / "synth.hip", CSRC / "synth_core.h"]):
        if verbose:
            print("hipcc synth.hip", flush=True)
        _run([hipcc, *COMMON, "-shared", str(CSRC / "synth.hip"), "-o", str(synth)])
    out["synth"] = synth

    # --- C host CLI -----------------------------------------------------------
    host_src = ROOT / "host" / "meteor_demod_amd.c"
    if host_src.exists():
        exe = LIB / "meteor_demod_amd"
        tui_src = [ROOT / "host" / "tui.c", ROOT / "host" / "tui.h"]
        if force or _stale(exe, [host_src, ROOT / "include" / "meteor_demod_amd.h", so, *tui_src]):
            cc = shutil.which("gcc") or shutil.which("cc")
            # the full-screen display needs ncurses (as the reference's ENABLE_TUI does); linked statically so that the binary
            # that travels to another box of this image needs nothing beyond libc there
            tui = []
            if Path("/usr/include/curses.h").exists():
                tui = ["-DMDEMOD_TUI", str(tui_src[0]), "-l:libncurses.a", "-l:libtinfo.a"]
            _run([cc, "-std=gnu11", "-O2", "-Wall", "-pthread", f"-I{ROOT / 'include'}", f"-I{ROOT / 'host'}", str(host_src), *tui[:2], "-o", str(exe),
                  f"-L{LIB}", "-lmeteor_demod_amd", f"-Wl,-rpath,$ORIGIN", "-lm", *tui[2:]])
        out["cli"] = exe
    return out


if __name__ == "__main__":
    for k, v in build(force="--force" in sys.argv, verbose=True).items():
        print(f"{k}: {v}")
