"""Static check of the gemm256 ISA (runs on the CPU-only box: hipcc -S).

The epilogue's bias operands are inline-asm loads awaited by an inline-asm s_waitcnt that carries them as in/out operands
(gemm256.hip, "EPI_OPERAND_LOAD" / "EPI_OPERAND_WAIT").  hipcc does not know that the registers are written asynchronously: a
register copy or spill it schedules between a load and its wait would read stale data.  This script fails if, in any
instantiation, an instruction between an operand load and the operand wait touches the load's destination registers, or if a
kernel has loads without a wait behind them.

    python tools/check_gemm256_isa.py          # exit code 0 = clean
"""
import re, subprocess, sys, tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def regs_of(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def check(isa: str):
    problems, kernels = [], 0
    name, pending = None, {}
    for ln, line in enumerate(isa.splitlines(), 1):
        m = re.match(r"^(_ZN5clibd22gemm256_bf16_nt_kernel\S+):", line)
        if m:
            name, pending = m.group(1), {}
            continue
        if name is None:
            continue
        code = line.split(";")[0].strip()
        if "EPI_OPERAND_LOAD" in line:
            dst = code.split()[1].rstrip(",")
            pending[ln] = regs_of(dst)
            continue
        if "EPI_OPERAND_WAIT" in line:
            if pending:
                kernels += 1
            pending = {}
            continue
        if "s_endpgm" in code:
            if pending:
                problems.append(f"{name}: operand load at line {min(pending)} has no wait before s_endpgm")
            name = None
            continue
        if pending and code and not code.endswith(":") and not code.startswith("."):
            used = set()
            for tok in re.findall(r"v\[\d+:\d+\]|v\d+", code):
                used |= regs_of(tok)
            for l0, regs in pending.items():
                if used & regs:
                    problems.append(f"{name}: line {ln} `{code}` touches v{sorted(used & regs)} of the operand load at line {l0} before its wait")
    return problems, kernels


def main():
    with tempfile.TemporaryDirectory() as d:
        out = Path(d) / "gemm256.s"
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-Wno-unused-value",
                            "-o", str(out), str(ROOT / "clibd_amd/csrc/gemm256.hip")], capture_output=True, text=True)
        if r.returncode != 0:
            print(r.stderr)
            return 2
        problems, kernels = check(out.read_text())
    for p in problems:
        print(p)
    print(f"[check_gemm256_isa] {kernels} load->wait regions checked, {len(problems)} problem(s)")
    return 1 if problems or kernels == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
