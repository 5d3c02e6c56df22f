#!/usr/bin/env python3
"""ISA audit for the loads that csrc/pbl_gemm_big.hip issues from inline asm (LIST-mode producers: request_stage / wait_set).

Those loads are invisible to hipcc's vmcnt bookkeeping on purpose ("GEMM regime"): the registers they
write are "in flight" until the counted `s_waitcnt vmcnt(N)` that the source pairs them with, and the only thing that keeps
the program correct is that NOTHING reads or writes such a register in between -- no compiler-inserted copy, no reuse as a
temporary.  This script compiles the file to gfx950 assembly and checks exactly that on the LIST kernels:

  * a register written by a `global_load_*` inside an asm block is in flight;
  * the set requested with `w` counted waits behind it lands at counted wait number w + 2 (requests run two stages ahead);
    of the prologue's requests the first half (stage 0) lands at the first counted wait, the second half (stage 1) at the second;
  * an asm block that waits `vmcnt(0)` (the rare paths that load and wait in one block) or a compiler `s_waitcnt vmcnt(0)`
    lands everything;
  * any other instruction that names an in-flight register is reported.

The stage loop is walked twice so that the wrap-around (odd stage -> even stage of the next iteration) is covered.
Usage: python tools/audit_asm_loads.py        (exit code 1 and a listing when something is found)
"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "pb_llm_amd", "csrc", "pbl_gemm_big.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def kernel_bodies(asm):
    """{mangled name: lines} of the pbl_gemm_kernel<*, true> (LIST) instantiations"""
    out = {}
    for m in re.finditer(r"^(_ZN\S*pbl_gemm_kernelILb[01]ELb1E\S*):.*?\n(.*?)\.end_amdhsa_kernel", asm, re.S | re.M):
        out[m.group(1)] = m.group(2).split("\n")
    return out


def audit(lines):
    # the producers' stage loop: from the first counted wait to the last one, walked twice
    counted = [i for i, l in enumerate(lines) if re.search(r"s_waitcnt vmcnt\((5|10)\)", l) and i and "#ASMSTART" in lines[i - 1]]
    if len(counted) < 2:
        return ["no counted waits found: the LIST producers are not what this script knows"], 0
    first, last = counted[0], counted[-1]
    # prologue: asm loads in front of the first counted wait
    in_asm, pro = False, []
    for l in lines[:first]:
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
        elif t.startswith(";;#ASMEND"):
            in_asm = False
        elif in_asm and t.startswith("global_load"):
            pro.append(regs(t.split(None, 1)[1].split(",")[0]))
    inflight = {}                                   # register -> counted-wait number at which it lands
    for k, dst in enumerate(pro):
        for r in dst:
            inflight[r] = 1 if k < len(pro) // 2 else 2
    problems, waits, nloads = [], 0, len(pro)
    # the loop ends behind the last counted wait's stage: walk to the next s_barrier after `last`
    end = next((i for i in range(last, len(lines)) if "s_barrier" in lines[i]), len(lines) - 1)
    body = lines[first - 1:end + 1]              # (from the ASMSTART marker of the first counted wait)
    for rep in range(2):
        in_asm = False
        for off, l in enumerate(body):
            t = l.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not t or t[0] in ";.":
                continue
            ins = t.split()[0]
            if in_asm and ins.startswith("global_load"):
                ops = t[len(ins):].split(",")
                dst, src = regs(ops[0]), regs(",".join(ops[1:]))
                bad = sorted(r for r in dst | src if r in inflight)
                if bad:
                    problems.append((first + off, t, bad))
                for r in dst:
                    inflight[r] = waits + 2
                nloads += 1
                continue
            if ins == "s_waitcnt" and "vmcnt(0)" in t:
                inflight.clear()
                continue
            if in_asm and ins == "s_waitcnt" and re.search(r"vmcnt\((5|10)\)", t):
                waits += 1
                for r in [r for r, w in inflight.items() if w <= waits]:
                    del inflight[r]
                continue
            bad = sorted(r for r in regs(t) if r in inflight)
            if bad:
                problems.append((first + off, t, bad))
    return problems, nloads


def main():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.check_call([HIPCC, "-std=c++17", "-O3", "--offload-arch=gfx950", "--cuda-device-only", "-S", SRC, "-o", out],
                              stderr=subprocess.DEVNULL)
        asm = open(out).read()
    bodies = kernel_bodies(asm)
    if not bodies:
        print("no LIST kernels found in the assembly")
        return 1
    rc = 0
    for name, lines in sorted(bodies.items()):
        problems, nloads = audit(lines)
        print(f"{name[:60]}...: {nloads} asm loads walked, {len(problems)} touches of in-flight registers")
        for p in problems[:20]:
            print("   ", p)
            rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main())
