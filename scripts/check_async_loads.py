#!/usr/bin/env python3
"""Static check of the inline-asm tile loads of k_bt_gemm_x6 / k_bt_dw_x6 / k_bt_gemm_bf16_big (csrc/engine_batch.hip.inc).

The loads are issued by `asm volatile("global_load_dwordx4 ...")` and waited for by an explicit `s_waitcnt vmcnt(N)`,
so the compiler does not know their destination registers are in flight.  If it places a copy (or any other use) of
such a register between the load and the wait that covers it, the copy reads stale data.  This script scans the device
assembly of the built source and fails if any instruction between an asm load and the next asm wait touches a
destination register of a load still in flight.

usage: check_async_loads.py [grlhip.s]     (without an argument the assembly is produced with hipcc -S)
"""
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "generativerl_preview_amd", "csrc")
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def check_kernel(name, lines):
    """Walk of the control-flow graph (both arms of every conditional branch; state = program counter + the loads in
    flight, oldest first).  An asm load marks its destination in flight; an asm `s_waitcnt vmcnt(N)` retires all but
    the N youngest; any other instruction that names a register of a load in flight is reported."""
    ins = []       # (kind, text, payload)
    label_at = {}
    in_asm = False
    for raw in lines:
        l = raw.strip()
        if l.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if l.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not l or l.startswith(";"):
            continue
        if l.startswith(".LBB") and ":" in l:
            label_at[l.split(":")[0]] = len(ins)
            continue
        if l.startswith("."):
            continue
        l = l.split(";")[0].strip()
        if in_asm and l.startswith("global_load_dwordx4"):
            ins.append(("load", l, frozenset(regs_of(l.split(",")[0]))))
        elif in_asm and l.startswith("s_waitcnt vmcnt("):
            ins.append(("wait", l, int(l[len("s_waitcnt vmcnt("):].split(")")[0])))
        elif l.startswith("s_branch"):
            ins.append(("jump", l, l.split()[1]))
        elif l.startswith("s_cbranch"):
            ins.append(("cjump", l, l.split()[1]))
        elif l.startswith("s_endpgm"):
            ins.append(("end", l, None))
        else:
            ins.append(("op", l, frozenset(regs_of(l))))
    n_loads = sum(1 for k, _, _ in ins if k == "load")
    n_waits = sum(1 for k, _, _ in ins if k == "wait")
    bad = {}
    seen = set()
    work = [(0, ())]
    while work:
        pc, flight = work.pop()
        while pc < len(ins):
            if (pc, flight) in seen:
                break
            seen.add((pc, flight))
            kind, text, arg = ins[pc]
            if kind == "load":
                flight = (flight + (pc,))[-32:]
            elif kind == "wait":
                flight = flight[len(flight) - arg:] if arg and len(flight) > arg else (() if arg == 0 else flight)
            elif kind == "jump":
                pc = label_at[arg]
                continue
            elif kind == "cjump":
                work.append((label_at[arg], flight))
            elif kind == "end":
                break
            else:
                for at in flight:
                    hit = arg & ins[at][2]
                    if hit:
                        bad[(pc, at)] = (pc, text, at, sorted(hit))
            pc += 1
    return n_loads, n_waits, list(bad.values())


def main():
    if len(sys.argv) > 1:
        text = open(sys.argv[1]).read()
    else:
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "grlhip.s")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-mllvm",
                                   "-pragma-unroll-threshold=1048576", "-S", "--cuda-device-only", "-o", out,
                                   os.path.join(CSRC, "grlhip.hip")])
            text = open(out).read()
    failed = False
    seen = 0
    for m in re.finditer(r"^(_ZN6grlhip2bt(?:12k_bt_gemm_x6|17k_bt_gemm_x6_pair|10k_bt_dw_x6|12k_bt_dw_many|18k_bt_gemm_bf16_big)\w+):[^\n]*\n", text, re.M):
        name = m.group(1)
        end = text.index(".end_amdhsa_kernel", m.end())
        body = text[m.end():end]
        cut = body.find(".Lfunc_end")
        lines = body[:cut if cut > 0 else len(body)].split("\n")
        loads, waits, bad = check_kernel(name, lines)
        seen += 1
        print(f"{name}: {loads} asm loads, {waits} asm waits, {len(bad)} early uses")
        for ln, l, at, hit in bad[:10]:
            print(f"   line {ln}: `{l}` touches v{hit} loaded at line {at} and not yet waited for")
        failed |= bool(bad) or loads == 0
    if seen == 0:
        print("no k_bt_gemm_x6 / k_bt_dw_x6 kernel found")
        failed = True
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
