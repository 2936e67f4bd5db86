"""Static check of the gemm256 ISA (runs on the CPU-only box: hipcc -S); any other kernel source that marks such loads is checked too.

gemm256's epilogue bias operands are inline-asm loads awaited by an inline-asm s_waitcnt that carries them as in/out operands
("EPI_OPERAND_LOAD" / "EPI_OPERAND_WAIT" markers in the asm text).  hipcc does not know that the registers are written asynchronously: a
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
    """Control-flow aware: from every marked load, every path (fall-through and branch targets) is followed until a marked wait; an
    instruction on the way that names one of the load's destination registers, or a path that reaches s_endpgm, is a problem."""
    problems, regions = [], 0
    for m in re.finditer(r"^(_ZN5clibd[^\s:]+):[^\n]*\n(.*?s_endpgm)", isa, re.S | re.M):
        name, body = m.group(1), m.group(2)
        lines = body.split("\n")
        code = [l.split(";")[0].strip() for l in lines]
        labels = {c[:-1]: i for i, c in enumerate(code) if c.endswith(":")}
        starts = []   # (first line after a group of consecutive marked loads, registers)
        i = 0
        while i < len(lines):
            if "EPI_OPERAND_LOAD" in lines[i]:
                regs = set()
                while i < len(lines) and "EPI_OPERAND_LOAD" in lines[i]:
                    regs |= regs_of(code[i].split()[1].rstrip(","))
                    i += 1
                starts.append((i, regs))
            else:
                i += 1
        for start, regs in starts:
            regions += 1
            seen, stack = set(), [start]
            while stack:
                j = stack.pop()
                while j < len(lines) and j not in seen:
                    seen.add(j)
                    c = code[j]
                    if "EPI_OPERAND_WAIT" in lines[j]:
                        break
                    if not c or c.endswith(":") or c.startswith("."):
                        j += 1
                        continue
                    if c.startswith("s_endpgm"):
                        problems.append(f"{name}: a path from the operand load before line {start} reaches s_endpgm without its wait")
                        break
                    used = set()
                    for tok in re.findall(r"v\[\d+:\d+\]|v\d+", c):
                        used |= regs_of(tok)
                    if used & regs:
                        problems.append(f"{name}: line {j} `{c}` touches v{sorted(used & regs)} of the operand load before line {start} ahead of its wait")
                        break
                    if c.startswith("s_branch"):
                        j = labels[c.split()[1]]
                        continue
                    if c.startswith("s_cbranch"):
                        stack.append(labels[c.split()[1]])
                    j += 1
    return problems, regions


def main():
    total_p, total_k = [], 0
    for src in ("gemm256.hip",):
        with tempfile.TemporaryDirectory() as d:
            out = Path(d) / "k.s"
            r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-Wno-unused-value",
                                "-o", str(out), str(ROOT / "clibd_amd/csrc" / src)], capture_output=True, text=True)
            if r.returncode != 0:
                print(r.stderr)
                return 2
            problems, kernels = check(out.read_text())
        print(f"[check_gemm256_isa] {src}: {kernels} load->wait regions checked, {len(problems)} problem(s)")
        if kernels == 0:
            problems.append(f"{src}: no marked region found")
        total_p += problems
        total_k += kernels
    for p in total_p:
        print(p)
    return 1 if total_p else 0


if __name__ == "__main__":
    sys.exit(main())
