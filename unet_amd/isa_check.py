"""Static check of the generated gfx950 ISA of conv_igemm.hip.

The 16x16x4 conv kernel issues its main-loop global loads from inline asm and waits for them with its own
``s_waitcnt vmcnt(0)`` (see the comment above ``gld_b`` in csrc/conv_igemm.hip).  Between such a load and that wait the
destination registers are "in flight": the compiler believes they already hold their values, so a register copy, spill
or reuse scheduled into that window would silently read or destroy garbage.  The analysis tracks the AGE of every outstanding load (how many asm loads
were issued after it), so a partial wait ``s_waitcnt vmcnt(N)`` -- the kernel's own or one the compiler places for its loads --
retires exactly the registers the hardware guarantees.  This module compiles the file to assembly
and proves, per kernel, that no instruction between an asm load group and the following asm wait touches a destination
register of an outstanding asm load.

``python -m unet_amd.isa_check`` prints a summary; tests/test_isa_cpu.py runs it.
"""
from __future__ import annotations

import re
import subprocess
import tempfile
from pathlib import Path
from typing import Dict, List, Set, Tuple

from .build import CSRC, FLAGS, HIPCC

_REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def _regs(text: str) -> Set[int]:
    out: Set[int] = set()
    for m in _REG.finditer(text):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def compile_to_asm(src: Path) -> str:
    with tempfile.TemporaryDirectory() as td:
        out = Path(td) / "k.s"
        flags = [f for f in FLAGS if f not in ("-fPIC",)]
        cmd = [HIPCC, *flags, "-S", "--cuda-device-only", "-o", str(out), str(src)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc -S failed:\n{r.stderr[-4000:]}")
        return out.read_text()


def _split_functions(asm: str):
    """Yields (name, [(line_no, text, in_asm)]) for every function of the assembly text."""
    name, body, in_asm = None, [], False
    for ln, line in enumerate(asm.splitlines(), 1):
        t = line.strip()
        if t.startswith(".type") and "@function" in t:
            if name is not None:
                yield name, body
            name, body, in_asm = t.split()[1].split(",")[0], [], False
            continue
        if name is None:
            continue
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith(";") or (t.startswith(".") and not t.split(";")[0].strip().endswith(":")):
            continue
        body.append((ln, t.split(";")[0].strip(), in_asm))
    if name is not None:
        yield name, body


def _check_function(name: str, body) -> Tuple[int, List[str]]:
    """Forward may-dataflow of the set of registers written by outstanding asm loads over the function's CFG."""
    # basic blocks
    blocks: List[List[Tuple[int, str, bool]]] = [[]]
    labels: Dict[str, int] = {}
    for ln, code, in_asm in body:
        if code.endswith(":"):
            if blocks[-1]:
                blocks.append([])
            labels[code[:-1]] = len(blocks) - 1
            continue
        blocks[-1].append((ln, code, in_asm))
        if code.startswith(("s_branch", "s_cbranch", "s_endpgm")):
            blocks.append([])
    succ: List[List[int]] = []
    for i, blk in enumerate(blocks):
        out: List[int] = []
        last = blk[-1][1] if blk else ""
        if last.startswith("s_branch"):
            out = [labels[last.split()[1]]]
        elif last.startswith("s_endpgm"):
            out = []
        else:
            if last.startswith("s_cbranch"):
                out.append(labels[last.split()[1]])
            if i + 1 < len(blocks):
                out.append(i + 1)
        succ.append(out)

    nloads = 0
    bad: List[str] = []

    # State: {register: age}, age = number of asm loads issued AFTER the load that writes the register, minimised over all paths
    # (the youngest the load can be).  Loads return in issue order, so `s_waitcnt vmcnt(N)` retires every register of age >= N.
    # Vector-memory instructions the compiler issues itself count as younger entries of the same counter (see _VMEM below).
    _VMCNT = re.compile(r"vmcnt\((\d+)\)")
    _VMEM = ("global_load", "global_store", "global_atomic", "buffer_load", "buffer_store", "buffer_atomic", "scratch_load", "scratch_store",
             "flat_load", "flat_store")

    def transfer(i: int, inflight: Dict[int, int], report: bool) -> Dict[int, int]:
        nonlocal nloads
        cur = dict(inflight)
        for ln, code, in_asm in blocks[i]:
            if in_asm and code.startswith(("global_load", "buffer_load")):
                ops = code.split(None, 1)[1].split(",")
                dst = _regs(ops[0])
                if report:
                    nloads += 1
                    if _regs(",".join(ops[1:])) & cur.keys():
                        bad.append(f"{name}:{ln}: load address uses an in-flight register: {code}")
                    if dst & cur.keys():
                        bad.append(f"{name}:{ln}: load overwrites an in-flight register: {code}")
                cur = {r: a + 1 for r, a in cur.items()}
                for r in dst:
                    cur[r] = 0
                continue
            if code.startswith("s_waitcnt"):
                m = _VMCNT.search(code)
                if m:                 # the kernel's own wait asm, or a compiler-placed one
                    keep = int(m.group(1))
                    cur = {r: a for r, a in cur.items() if a < keep}
                continue
            if in_asm:
                continue
            if code.startswith(_VMEM):
                # a vector-memory instruction the compiler issues itself (epilogue loads / stores, spills) is one more YOUNGER entry of the
                # same in-order counter: it ages every outstanding asm load (a counted wait behind N such instructions retires the loads)
                if cur and report:
                    hit = _regs(code) & cur.keys()
                    if hit:
                        bad.append(f"{name}:{ln}: touches in-flight v{sorted(hit)[0]}: {code}")
                cur = {r: a + 1 for r, a in cur.items()}
                continue
            if code.startswith("s_endpgm"):
                if cur and report:
                    bad.append(f"{name}:{ln}: kernel ends with asm loads in flight")
                continue
            if cur and report:
                hit = _regs(code) & cur.keys()
                if hit:
                    bad.append(f"{name}:{ln}: touches in-flight v{sorted(hit)[0]}: {code}")
        return cur

    ins: List[Dict[int, int]] = [dict() for _ in blocks]
    work = list(range(len(blocks)))
    while work:
        i = work.pop()
        out = transfer(i, ins[i], False)
        for j in succ[i]:
            changed = False
            for r, a in out.items():
                if ins[j].get(r, 1 << 30) > a:
                    ins[j][r] = a
                    changed = True
            if changed:
                work.append(j)
    for i in range(len(blocks)):
        transfer(i, ins[i], True)
    return nloads, bad


def check_asm(asm: str) -> Tuple[Dict[str, int], List[str]]:
    """Returns ({kernel: number of asm loads checked}, [violations])."""
    kernels: Dict[str, int] = {}
    bad: List[str] = []
    for name, body in _split_functions(asm):
        n, b = _check_function(name, body)
        if n:
            kernels[name] = n
        bad += b
    return kernels, bad


SOURCES = ("conv_igemm.hip", "conv_bf16.hip")     # every file whose kernels issue asm loads


def main() -> int:
    rc = 0
    for src in SOURCES:
        kernels, bad = check_asm(compile_to_asm(CSRC / src))
        print(f"{src}: {len(kernels)} kernels with asm loads, {sum(kernels.values())} loads checked, {len(bad)} violations")
        for b in bad[:40]:
            print("  ", b)
        rc |= 1 if bad else 0
    return rc


if __name__ == "__main__":
    raise SystemExit(main())
