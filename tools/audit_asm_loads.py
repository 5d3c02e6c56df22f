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


# ---- round 4: the expanding waves of csrc/pbl_gemm_img.hip ---------------------------------------------------------------------
# Their slot requests are asm loads in the saddr form (`global_load_dwordx4 v[a:b], vOFF, s[c:d]`), waited for by ONE counted
# `s_waitcnt vmcnt(10 | 12)` per 64-column step.  A set requested in step q is guaranteed to have landed at the counted wait that
# closes step q + 2 (the third counted wait behind the request: in-order return, see wait_all in the source) and is used in step
# q + 4; until it has landed nothing may name its registers.  Walked: prologue, the unrolled 4-step loop body twice, the
# remainder steps.
IMG_SRC = os.path.join(REPO, "pb_llm_amd", "csrc", "pbl_gemm_img.hip")


def img_kernel_bodies(asm):
    out = {}
    # (the LDS-staged instantiations <OM, KT, XF = false>: this audit models THEIR x pieces and the `vmcnt(10)` invariant.  The round-6
    # XF instantiations -- B fragments by plain loads with `vmcnt(6)`, fixed five-load requests with `vmcnt(15)` -- are covered by the
    # generic audit_waits, whose rule V counts exactly those ages; tests/test_round5_cpu.py mutates both constants)
    for m in re.finditer(r"^(_ZN\S*pbl_gemm_img_kernelILi[0-9]ELb[01]ELb0E\S*):.*?\n(.*?)\.end_amdhsa_kernel", asm, re.S | re.M):
        out[m.group(1)] = m.group(2).split("\n")
    return out


def audit_img(lines):
    """forward dataflow over the kernel's control-flow graph.  State: {register: age}, age = vector-memory operations issued since
    the slot request that writes the register (every x piece `buffer_load ... lds` is one operation, every request block AT LEAST
    one: a slot has 1 - 5 vectors and the loads a small slot skips only make later ones land EARLIER).  `s_waitcnt vmcnt(10)`
    lands everything of age >= 10 (in-order return), `vmcnt(0)` everything.  Joins keep the youngest age (conservative).
    Reported: any instruction that names a register still in flight."""
    is_req = lambda t: re.match(r"global_load_dwordx[24]\s+v\[\d+:\d+\],\s*v\d+,\s*s\[\d+:\d+\]", t)     # noqa: E731
    CAP = 14
    # instructions with their asm-block membership
    ins, in_a, blk_id = [], False, 0
    label_at = {}
    for l in lines:
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
            in_a, blk_id = True, blk_id + 1
            continue
        if t.startswith(";;#ASMEND"):
            in_a = False
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            label_at[m.group(1)] = len(ins)
            continue
        if not t or t[0] in ";." or t.startswith("s_nop") and False:
            continue
        if re.match(r"^[a-z_]+[a-z0-9_]*\b", t):
            ins.append((t, blk_id if in_a else 0))
    n = len(ins)
    if not any(a and is_req(t) for t, a in ins):
        return ["no slot requests found: the image kernel is not what this script knows"], 0
    # local numeric labels inside asm blocks ("1:") never reach here as instructions; branches to them ("1f") are intra-block skips
    succ = [[] for _ in range(n)]
    for i, (t, a) in enumerate(ins):
        op = t.split()[0]
        m = re.search(r"(\.LBB\d+_\d+)", t)
        if op == "s_endpgm":
            continue
        if op == "s_branch" and m:
            succ[i].append(label_at[m.group(1)])
            continue
        if op.startswith("s_cbranch") and m:
            succ[i].append(label_at[m.group(1)])
        if i + 1 < n:
            succ[i].append(i + 1)
    state = [None] * n
    state[0] = {}
    work = [0]
    nreq = sum(1 for t, a in ins if a and is_req(t))

    def step(i, st):
        t, a = ins[i]
        op = t.split()[0]
        st = dict(st)
        if a and is_req(t):
            first_of_block = not (i and ins[i - 1][1] == a and any(is_req(ins[k][0]) for k in range(i - 1, -1, -1) if ins[k][1] == a))
            if first_of_block:
                st = {r: min(CAP, g + 1) for r, g in st.items()}
            for r in regs(t[len(op):].split(",")[0]):
                st[r] = 0
            return st
        if op.startswith("buffer_load") and " lds" in t:
            st = {r: min(CAP, g + 1) for r, g in st.items()}
            st["x pieces of this step"] = 0                     # (pseudo register: the youngest x piece issued since the last barrier)
            return st
        if op == "s_waitcnt" and "vmcnt(0)" in t:
            return {}
        mw = re.search(r"vmcnt\((\d+)\)", t) if (a and op == "s_waitcnt") else None
        if mw:                                                  # the step's ONE counted wait (vmcnt(10) in the shipped kernel)
            return {r: g for r, g in st.items() if g < int(mw.group(1))}
        if op == "s_barrier":
            # the barrier publishes the x pieces issued BEFORE the previous barrier: they must have landed by now (the race the
            # config-3 test found with a wait that was too lax); this step's pieces become "the previous step's"
            cur = st.pop("x pieces of this step", None)
            if cur is not None:
                st["x pieces of the previous step (in flight at the barrier that publishes them)"] = cur
            return st
        return st

    while work:
        i = work.pop()
        out = step(i, state[i])
        for j in succ[i]:
            if state[j] is None:
                state[j] = dict(out)
                work.append(j)
            else:
                new = dict(state[j])
                changed = False
                for r, g in out.items():
                    if r not in new or g < new[r]:
                        new[r] = g
                        changed = True
                if changed:
                    state[j] = new
                    work.append(j)
    problems = []
    for i, (t, a) in enumerate(ins):
        if state[i] is None:
            continue
        op = t.split()[0]
        if a and (is_req(t) or (op == "s_waitcnt")):
            if is_req(t):
                bad = sorted(r for r in regs(t[len(op):].split(",")[1]) if r in state[i])      # (the offset register; the destination is re-requested)
                if bad:
                    problems.append((i, t, bad))
            continue
        if a and op in ("s_cmp_lt_u32", "s_cbranch_scc1"):
            continue
        if op == "s_barrier":
            key = "x pieces of the previous step (in flight at the barrier that publishes them)"
            if key in state[i]:
                problems.append((i, t, [key, f"age {state[i][key]}"]))
            continue
        bad = sorted(r for r in regs(t) if r in state[i])
        if bad:
            problems.append((i, t, bad))
    return problems, nreq


# ---- round 5: a generic soundness check of the vector-memory / LDS waits, for the small-batch kernel and every other kernel ------
# pbl_sb_img_kernel keeps a ring of slot register sets in flight with plain loads (the compiler counts vmcnt: the source shapes the
# loop so that every path issues the same number of loads) and hands the x tile over at a workgroup barrier (the staging wave's
# LDS-DMA pieces, the working waves' fragment reads).  Whoever does the counting, the invariants are the same and can be checked on
# the ISA by forward dataflow over the control-flow graph:
#   V  no instruction names a VGPR whose vector-memory load may still be in flight.  State {register: age}, age = vector-memory
#      operations (loads, stores, atomics, LDS-DMA pieces: one vmcnt each, returned in order) issued since the load that writes it;
#      `s_waitcnt vmcnt(N)` lands everything of age >= N.  A NEW load into a register in flight is fine (in-order return).
#   D  no LDS-DMA piece (`buffer_load ... lds`) is in flight at an s_barrier: the barrier publishes the tile it writes.
#   L  no LDS operation of this wave is in flight at an s_barrier (`s_waitcnt lgkmcnt(0)` in front of it): a fragment read that has
#      not returned may see the NEXT tile, a tile store that has not landed is not there for the others.
# Joins keep the youngest age (conservative).  `mutate` rewrites one wait of the listing before the walk -- how the tests prove the
# audit is not vacuous: the vmcnt(13) build of the image kernel and a one-too-lax wait in the small-batch kernel must both be found.
VMEM = re.compile(r"^(global|buffer|flat|scratch)_(load|store|atomic)")


def parse_cfg(lines):
    ins, label_at = [], {}
    for l in lines:
        t = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            label_at[m.group(1)] = len(ins)
            continue
        if not t or t[0] in ";." or re.match(r"^\d+:", t):
            continue
        if re.match(r"^[a-z_]+[a-z0-9_]*\b", t):
            ins.append(t.split(";")[0].strip())
    succ = [[] for _ in ins]
    for i, t in enumerate(ins):
        op = t.split()[0]
        m = re.search(r"(\.LBB\d+_\d+)", t)
        if op == "s_endpgm":
            continue
        if op == "s_branch" and m:
            succ[i].append(label_at[m.group(1)])
            continue
        if op.startswith("s_cbranch") and m:
            succ[i].append(label_at[m.group(1)])
        if i + 1 < len(ins):
            succ[i].append(i + 1)
    return ins, succ


def audit_waits(lines, cap=40, dma_rule=True):
    """-> (findings, number of vector-memory loads walked).  A finding: (instruction index, text, what).  dma_rule=False: rule D off
    (pbl_gemm_img_kernel keeps the CURRENT step's pieces in flight across the barrier by design: audit_img checks its own rule)"""
    ins, succ = parse_cfg(lines)
    n = len(ins)
    DMA, LDS = "LDS-DMA piece", "LDS operation"

    def step(i, st):
        t = ins[i]
        op = t.split()[0]
        if VMEM.match(op):
            st = {r: (min(cap, g + 1) if (r != LDS and not isinstance(r, tuple)) else g) for r, g in st.items()}
            if "_load" in op and " lds" in t:
                st[DMA] = 0
            elif "_load" in op or ("_atomic" in op and " glc" in t or " sc0" in t and "_atomic" in op):
                for r in regs(t[len(op):].split(",")[0]):
                    st[r] = 0
            return st
        if op.startswith("ds_"):
            # rule F: LDS operations return in order among themselves; a ds_read's destination is in flight until a
            # `s_waitcnt lgkmcnt(N)` with N <= (LDS operations issued since).  Scalar loads share the counter and return out of
            # order, which only makes a wait stricter than this model: a read with `a` younger LDS operations outstanding needs
            # a + 1 <= N outstanding events.  Keys ("l", register).
            st = {r: ((min(cap, g + 1)) if isinstance(r, tuple) else g) for r, g in st.items()}
            st[LDS] = 0
            if op.startswith("ds_read") or op.startswith("ds_load"):
                for r in regs(t[len(op):].split(",")[0]):
                    st[("l", r)] = 0
            return st
        if op == "s_waitcnt":
            mv = re.search(r"vmcnt\((\d+)\)", t)
            if mv:
                k = int(mv.group(1))
                st = {r: g for r, g in st.items() if r == LDS or isinstance(r, tuple) or g < k}
            ml = re.search(r"lgkmcnt\((\d+)\)", t)
            if ml:
                k = int(ml.group(1))
                st = {r: g for r, g in st.items() if not (isinstance(r, tuple) and g >= k) and not (r == LDS and k == 0)}
            return st
        return st

    state = [None] * n
    state[0] = {}
    work = [0]
    while work:
        i = work.pop()
        out = step(i, state[i])
        for j in succ[i]:
            if state[j] is None:
                state[j] = dict(out)
                work.append(j)
            else:
                new, changed = dict(state[j]), False
                for r, g in out.items():
                    if r not in new or g < new[r]:
                        new[r] = g
                        changed = True
                if changed:
                    state[j] = new
                    work.append(j)
    findings, nloads = [], 0
    for i, t in enumerate(ins):
        if state[i] is None:
            continue
        op = t.split()[0]
        st = state[i]
        if op == "s_barrier":
            for key in ((DMA, LDS) if dma_rule else (LDS,)):
                if key in st:
                    findings.append((i, t, f"{key} in flight at the barrier"))
            continue
        if op == "s_waitcnt":
            continue
        named_v = named_l = regs(t)
        if VMEM.match(op) and "_load" in op and " lds" not in t:
            nloads += 1
            named_v = regs(",".join(t[len(op):].split(",")[1:]))                 # a NEW vector-memory load into a register whose load is in flight
        elif op.startswith("ds_read") or op.startswith("ds_load"):               # is fine (in-order return), likewise a new ds_read over a ds_read;
            named_l = regs(",".join(t[len(op):].split(",")[1:]))                 # ACROSS the two pipes it is a write-after-write hazard
        bad = sorted(r for r in named_v if r in st)
        badl = sorted(r for r in named_l if ("l", r) in st)
        if bad:
            findings.append((i, t, f"touches v{bad} in flight"))
        if badl:
            findings.append((i, t, f"touches v{badl} whose ds_read is in flight"))
    return findings, nloads


def kernel_bodies_named(asm, pattern):
    out = {}
    for m in re.finditer(r"^(_Z\S*" + pattern + r"\S*):.*?\n(.*?)\.end_amdhsa_kernel", asm, re.S | re.M):
        out[m.group(1)] = m.group(2).split("\n")
    return out


_ASM_CACHE = {}


def compile_asm(src):
    if src not in _ASM_CACHE:
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "k.s")
            subprocess.check_call([HIPCC, "-std=c++17", "-O3", "--offload-arch=gfx950", "--cuda-device-only", "-S", src, "-o", out],
                                  stderr=subprocess.DEVNULL)
            _ASM_CACHE[src] = open(out).read()
    return _ASM_CACHE[src]


def mutate_wait(lines, which, delta=1, counter="vmcnt", only_in_loop=True):
    """a copy of the listing with the `which`-th counted wait (counter(N), N >= 1 when `delta` > 0) made `delta` laxer; None when
    there are fewer such waits"""
    idx = [i for i, l in enumerate(lines) if re.search(r"s_waitcnt.*" + counter + r"\((\d+)\)", l) and
           (delta < 0 or int(re.search(counter + r"\((\d+)\)", l).group(1)) >= 1)]
    if which >= len(idx):
        return None
    out = list(lines)
    i = idx[which]
    out[i] = re.sub(counter + r"\((\d+)\)", lambda m: f"{counter}({int(m.group(1)) + delta})", out[i], count=1)
    return out


def main_waits(src=None, pattern="pbl_sb_img_kernel", verbose=True, dma_rule=True):
    """audit_waits over every kernel of `src` whose mangled name contains `pattern`; 0 = clean"""
    asm = compile_asm(src or IMG_SRC)
    bodies = kernel_bodies_named(asm, pattern)
    if not bodies:
        print(f"no kernel matching {pattern} in the assembly")
        return 1
    rc = 0
    for name, lines in sorted(bodies.items()):
        findings, nloads = audit_waits(lines, dma_rule=dma_rule)
        if verbose:
            print(f"{name[:70]}...: {nloads} vector-memory loads walked, {len(findings)} findings")
        for f in findings[:20]:
            print("   ", f)
            rc = 1
    return rc


def main_img():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.check_call([HIPCC, "-std=c++17", "-O3", "--offload-arch=gfx950", "--cuda-device-only", "-S", IMG_SRC, "-o", out],
                              stderr=subprocess.DEVNULL)
        asm = open(out).read()
    bodies = img_kernel_bodies(asm)
    if not bodies:
        print("no image kernels found in the assembly")
        return 1
    rc = 0
    for name, lines in sorted(bodies.items()):
        problems, nloads = audit_img(lines)
        print(f"{name[:70]}...: {nloads} slot requests walked, {len(problems)} touches of in-flight registers")
        for p in problems[:20]:
            print("   ", p)
            rc = 1
    return rc


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
    sys.exit(main() | main_img() | main_waits() | main_waits(pattern="pbl_gemm_img_kernel", dma_rule=False))
