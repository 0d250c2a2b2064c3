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


def _asm_limits() -> dict:
    """First VGPR (and AccVGPR) the compiler may not touch, per kernel-name pattern, from the generated assembly headers."""
    import re
    rot = (CSRC / "rotwin_asm.h").read_text()
    pk = (CSRC / "rotpk_asm.h").read_text()
    lim = {"demod_kernel_rot": (int(re.search(r"#define ROTWIN_LIMIT (\d+)", rot).group(1)), int(re.search(r"#define ROTWIN_AB (\d+)", rot).group(1)))}
    for m in re.finditer(r"#define ROTPK_(\w+?)_(\d+)_LIMIT (\d+)", pk):
        lim[f"demod_kernel_rotp_{m.group(1)}_{m.group(2)}_"] = (int(m.group(3)), 0)
    return lim


def scan_asm_partition(asm_text: str, limits: dict, scratch_free: tuple = ()) -> list[str]:
    """The v3 kernels are only correct while hipcc keeps out of the registers their inline assembly owns (the FIR windows live
    there ACROSS asm statements; `amdgpu_num_vgpr` is only a hint).  For every kernel of `asm_text` (hipcc -S output) whose name
    matches a key of `limits`: every compiler-generated instruction (outside #ASMSTART/#ASMEND) that names a VGPR >= limit or an
    AccVGPR >= the AccVGPR limit is a violation; so are scratch accesses inside the main loop, a scratch segment at all for kernels
    matching `scratch_free`, and a kernel that matches no limit although it contains inline assembly."""
    import re
    bad: list[str] = []
    lines = asm_text.split("\n")
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z\w+):", lines[i])
        if not m or "demod_kernel_rot" not in m.group(1):
            i += 1
            continue
        name = m.group(1)
        end = next(j for j in range(i, len(lines)) if lines[j].startswith(".Lfunc_end"))
        body = lines[i:end]
        key = max((k for k in limits if k in name), key=len, default=None)
        if key is None:
            if any("#ASMSTART" in b for b in body):
                bad.append(f"{name}: inline assembly but no register limit known for this kernel")
            i = end
            continue
        limit, alimit = limits[key]
        inasm, depth_loop = False, False
        for n, line in enumerate(body, i + 1):
            if "#ASMSTART" in line:
                inasm = True
            elif "#ASMEND" in line:
                inasm = False
            elif not inasm:
                t = line.strip()
                if "Loop Header: Depth=1" in line:
                    depth_loop = True
                if not t or t[0] in ";." or t.endswith(":"):
                    continue
                t = t.split(";")[0]
                regs = [int(x.group(1)) for x in re.finditer(r"\bv(\d+)\b", t)] + [int(x.group(2)) for x in re.finditer(r"\bv\[(\d+):(\d+)\]", t)]
                aregs = [int(x.group(1)) for x in re.finditer(r"\ba(\d+)\b", t)] + [int(x.group(2)) for x in re.finditer(r"\ba\[(\d+):(\d+)\]", t)]
                if any(r >= limit for r in regs) or (alimit and any(r >= alimit for r in aregs)):
                    bad.append(f"{name}:{n}: {t.strip()}  (limit v{limit}" + (f", a{alimit})" if alimit else ")"))
        # scratch: none inside the demodulator's main loop (= the depth-1 loop with the most instructions), none at all where asked
        loops = []
        for j, line in enumerate(body):
            if "Loop Header: Depth=1" in line:
                k = j
                while k > 0 and not re.match(r"^\.LBB\d+_\d+:", body[k]):
                    k -= 1
                lab = body[k].split(":")[0]
                last = max((q for q, t in enumerate(body) if re.search(r"s_c?branch\S*\s+" + re.escape(lab) + r"\b", t)), default=k)
                loops.append((last - k, k, last))
        if loops:
            _, k, last = max(loops)
            for q in range(k, last + 1):
                if body[q].strip().startswith("scratch_"):
                    bad.append(f"{name}:{i + q + 1}: scratch access inside the main loop: {body[q].strip()}")
        meta = lines[end:end + 80]
        scratch = next((int(re.search(r"(\d+)", x.split(":")[1]).group(1)) for x in meta if "ScratchSize" in x), 0)
        if scratch and any(k in name for k in scratch_free):
            bad.append(f"{name}: {scratch} bytes of scratch (spills) in a kernel that must have none")
        i = end
    return bad


# kernels that must not spill at all (the BASELINE configs[1] / [2] instances and their generic siblings)
SCRATCH_FREE = ("demod_kernel_rotILi",)


def check_rot_partition(files: dict | None = None) -> list[str]:
    """Register-partition check of EVERY assembly-owning kernel file (demod_kernel_rot.hip: std + hybrid windows;
    demod_kernel_rotp.hip: wide / mid / far packed windows - demod_kernel_gat.hip has no assembly of its own).  `files` maps a
    source stem to the hipcc -S output build() kept next to the objects; without it both files are compiled to assembly here."""
    import tempfile
    limits = _asm_limits()
    bad: list[str] = []
    with tempfile.TemporaryDirectory() as td:
        for stem, flags in (("demod_kernel_rot", ROT_FLAGS), ("demod_kernel_rotp", ROTP_FLAGS)):
            out = (files or {}).get(stem)
            if out is None or not Path(out).exists():
                out = Path(td) / (stem + ".s")
                _run([_hipcc(), *COMMON, *flags, "-x", "hip", "--offload-device-only", "-S", str(CSRC / (stem + ".hip")), "-o", str(out)])
            bad += scan_asm_partition(Path(out).read_text(), limits, SCRATCH_FREE)
    return bad


def build(force: bool = False, verbose: bool = False) -> dict[str, Path]:
    LIB.mkdir(exist_ok=True)
    hipcc = _hipcc()
    headers = sorted(CSRC.glob("*.h")) + [ROOT / "include" / "meteor_demod_amd.h"]
    out: dict[str, Path] = {}

    # --- product library ---------------------------------------------------
    # (source, object stem, extra flags).  The std rotating-window file takes the max-ILP machine scheduler (measured +1.7 % on
    # configs[1]); the packed windows the default one (max-ILP makes their u8 variants spill inside the main loop).
    units = [(CSRC / "demod_kernel.hip", "demod_kernel", []),
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
    asm_units = {"demod_kernel_rot": LIB / "demod_kernel_rot.gfx950.s", "demod_kernel_rotp": LIB / "demod_kernel_rotp.gfx950.s"}
    checked = LIB / "asm_partition.ok"
    for src, stem, extra in units:
        obj = LIB / (stem + ".o")
        if force or _stale(obj, [src] + headers) or (stem in asm_units and not asm_units[stem].exists()):
            if verbose:
                print("hipcc", src.name, "->", obj.name, flush=True)
            if stem in asm_units:
                # the device assembly is a by-product of the same compilation (-save-temps): the register-partition check below
                # reads what was actually built
                import tempfile
                with tempfile.TemporaryDirectory(dir=str(LIB)) as td:
                    tmp_obj = Path(td) / (stem + ".o")
                    _run([hipcc, *COMMON, *extra, "-save-temps=obj", "-x", "hip", "-c", str(src), "-o", str(tmp_obj)])
                    dev_s = next(Path(td).glob("*gfx950*.s"))
                    shutil.copy(dev_s, asm_units[stem])
                    shutil.move(str(tmp_obj), str(obj))
                checked.unlink(missing_ok=True)
            else:
                _run([hipcc, *COMMON, *extra, "-x", "hip", "-c", str(src), "-o", str(obj)])
        objs.append(obj)
    # the assembly-owning kernels are only correct while hipcc stays out of the assembly's registers: a violation fails the BUILD
    if force or not checked.exists():
        bad = check_rot_partition(asm_units)
        if bad:
            for stem in asm_units:
                (LIB / (stem + ".o")).unlink(missing_ok=True)          # nothing links against a kernel that would corrupt its window
            raise RuntimeError("register partition of the v3 kernels violated (compiler code in the assembly's registers, or scratch "
                               "where there must be none):\n  " + "\n  ".join(bad[:20]))
        checked.write_text("ok\n")
    for stale in LIB.glob("demod_kernel_rw_*.o"):            # objects of the v2 kernel (retired in round 4) from an older tree
        stale.unlink()
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
