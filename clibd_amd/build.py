"""Build libclibd_hip.so (gfx950) in-tree with hipcc.

    python -m clibd_amd.build [--force] [--report]

hipcc cross-compiles without a GPU, so this runs on the CPU-only authoring box; the built .so is
git-ignored but travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import argparse
import concurrent.futures as cf
import os
import re
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
INCLUDE = HERE.parent / "include"
LIB = HERE / "libclibd_hip.so"
OBJ = CSRC / "build"
ARCH = "gfx950"
SOURCES = ["capi", "gemm", "gemm256", "gemm256_tn", "layernorm", "attention", "lora", "elementwise", "loss", "topk", "paramgrad"]


def csrc_hash() -> str:
    """sha256 (16 hex digits) over the kernel sources and headers: ties a measurement file to the code it measured."""
    import hashlib

    h = hashlib.sha256()
    for f in sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.h")) + [INCLUDE / "clibd_hip.h"]):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _deps() -> list[Path]:
    return [p for p in CSRC.glob("*.h")] + [INCLUDE / "clibd_hip.h"]


def _stale(target: Path, srcs: list[Path]) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(s.stat().st_mtime > t for s in srcs)


def _flags(name: str) -> list[str]:
    """Every flag that changes the generated code of unit `name` (hashed into its stamp: ADVICE r4 — a changed --offload-arch or
    optimisation flag must not leave old objects 'current')."""
    f = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]
    if name == "capi":
        f.append(f'-DCLIBD_CSRC_HASH="{csrc_hash()}"')   # clibd_build_hash(): checked by _lib.load() against the sources
    if os.environ.get("CLIBD_GEMM_DIAG") == "1":
        f.append("-DCLIBD_GEMM_DIAG")  # tools/gemm_stamps.py: s_memtime stamps + start-up skew knob (never in the product build)
    return f


_HIPCC_VERSION = None


def _hipcc_version() -> str:
    global _HIPCC_VERSION
    if _HIPCC_VERSION is None:
        try:
            _HIPCC_VERSION = subprocess.run([_hipcc(), "--version"], capture_output=True, text=True, timeout=60).stdout.strip()
        except Exception:
            _HIPCC_VERSION = "unknown"
    return _HIPCC_VERSION


def _link_stamp() -> str:
    return f"{csrc_hash()} {ARCH}"


def _compile_one(name: str, report: bool) -> tuple[str, str]:
    src = CSRC / f"{name}.hip"
    obj = OBJ / f"{name}.o"
    cmd = [_hipcc()] + _flags(name) + ["-c", str(src), "-o", str(obj)]
    if report:
        cmd.append("-Rpass-analysis=kernel-resource-usage")
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
    return name, r.stderr


def _summarise(stderr: str) -> list[str]:
    rows, cur = [], {}
    for line in stderr.splitlines():
        m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|SGPRs Spill|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]):\s*(\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            if cur:
                rows.append(cur)
            cur = {"name": v}
        else:
            cur[k] = v
    if cur:
        rows.append(cur)
    out = []
    for r in rows:
        out.append(
            f"{r['name'][:70]:70s} vgpr={r.get('VGPRs','?'):>4} agpr={r.get('AGPRs','?'):>3} spill={r.get('VGPRs Spill','?'):>3} "
            f"scratch={r.get('ScratchSize [bytes/lane]','?'):>4} occ={r.get('Occupancy [waves/SIMD]','?')} lds={r.get('LDS Size [bytes/block]','?')}"
        )
    return out


def _unit_hash(name: str) -> str:
    """Content hash of everything one object file is compiled from: its .hip, every shared header, the flags that change code."""
    import hashlib

    h = hashlib.sha256()
    for f in [CSRC / f"{name}.hip"] + sorted(_deps()):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    h.update("\0".join(_flags(name)).encode())   # arch, optimisation level, -D switches (capi: the csrc hash it embeds)
    h.update(_hipcc_version().encode())
    return h.hexdigest()[:16]


def build(force: bool = False, report: bool = False, verbose: bool = True) -> Path:
    """Recompile every unit whose CONTENT hash (source + headers) differs from the one its object was built from.

    mtimes are not consulted: a copied / rsync'd / restored tree keeps them while the text changes, and the
    load-time check of `clibd_build_hash()` must never accept a library whose kernel objects predate the sources.  A unit's
    stamp is written only after its object exists; the library's stamp (`capi.hash`, = csrc_hash()) only after the link.
    """
    OBJ.mkdir(parents=True, exist_ok=True)

    def unit_stamp(n: str) -> Path:
        return OBJ / f"{n}.unit_hash"

    def unit_current(n: str) -> bool:
        st = unit_stamp(n)
        return (OBJ / f"{n}.o").exists() and st.exists() and st.read_text().strip() == _unit_hash(n)

    todo = [n for n in SOURCES if force or report or not unit_current(n)]
    stamp = OBJ / "capi.hash"
    if todo:
        for n in todo:
            unit_stamp(n).unlink(missing_ok=True)
        stamp.unlink(missing_ok=True)
        if verbose:
            print(f"[clibd_amd.build] hipcc --offload-arch={ARCH}: {', '.join(todo)}", flush=True)
        with cf.ThreadPoolExecutor(max_workers=min(4, len(todo))) as ex:
            for name, err in ex.map(lambda n: _compile_one(n, report), todo):
                unit_stamp(name).write_text(_unit_hash(name))
                if report:
                    print(f"== {name}")
                    print("\n".join(_summarise(err)))
    objs = [OBJ / f"{n}.o" for n in SOURCES]
    linked_ok = LIB.exists() and stamp.exists() and stamp.read_text().strip() == _link_stamp()
    if force or todo or not linked_ok or _stale(LIB, objs):
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(LIB)] + [str(o) for o in objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
        stamp.write_text(_link_stamp())
        if verbose:
            print(f"[clibd_amd.build] linked {LIB}", flush=True)
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--report", action="store_true", help="print per-kernel register/LDS usage")
    ap.add_argument("--diag", action="store_true", help="diagnostic build of gemm256 (stamps + skew knob); forces a rebuild")
    a = ap.parse_args()
    if a.diag:
        os.environ["CLIBD_GEMM_DIAG"] = "1"
        a.force = True
    build(force=a.force, report=a.report)
    sys.exit(0)
