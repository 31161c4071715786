#!/usr/bin/env python3
"""Audit of the gfx950 assembly for the hidden-load discipline of tcgnn_device.hip.

A load issued from inline asm is invisible to hipcc's waitcnt bookkeeping: nothing may read (or
copy, or overwrite) its destination VGPRs until OUR s_waitcnt has retired it.  This script walks
every kernel's listing in program order and reports any instruction that touches a VGPR with such
a load in flight (global_load_* / ds_read_* inside ;;#ASMSTART .. ;;#ASMEND) before the covering
s_waitcnt vmcnt(0) / lgkmcnt(0), or a counted lgkmcnt(N) inside an asm statement (all but the newest N LDS reads have
landed: the flat streams' two-register-set tile walk).  Straight-line approximation: labels are treated as fall-through,
which is exact for the loops in this file (every back-edge is preceded by a full wait).
usage: audit_hidden_loads.py file.s [kernel-substring]
"""
import re, sys

def regs(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(3) is not None: out.add(int(m.group(3)))
        else: out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out

def audit(path, filt=""):
    kernel, in_asm, vm, lgkm, bad, skip_to = None, False, {}, {}, 0, None
    for ln, line in enumerate(open(path), 1):
        s = line.strip()
        m = re.match(r"^(_Z\w+):", s)
        if m: kernel, vm, lgkm = m.group(1), {}, {}; continue
        if kernel is None or filt not in kernel: continue
        # the timing-only instantiations of the flat-stream kernel (template argument DBG = true: TCGNN_LDS_DBG switches as scalar
        # branches around every step) are not straight-line code: the approximation below does not apply, and they never run in the
        # product path
        if re.match(r"_Z20spmm_lds_flat_kernelILi\dELi\dELi\dELb1E", kernel): continue
        if s.startswith(";;#ASMSTART"): in_asm = True; skip_to = None; continue
        if s.startswith(";;#ASMEND"): in_asm = False; skip_to = None; continue
        # an asm statement with its own branch (lds_flat_step_if: "s_cbranch 1f ... s_branch 2f / 1: wait for everything / 2:"): the
        # path that ISSUES loads is the one to follow - the other leaves nothing in flight - so what lies behind the unconditional
        # s_branch is skipped up to its target label
        if in_asm and skip_to is not None:
            if re.match(r"^%s:" % re.escape(skip_to), s) or (skip_to[-1:] in "fb" and re.match(r"^%s:" % re.escape(skip_to[:-1]), s)): skip_to = None
            continue
        if in_asm and s.split()[0] == "s_branch": skip_to = s.split()[1]; continue
        if not s or s.startswith((";", ".")): continue
        op = s.split()[0]
        if op == "s_endpgm": kernel = None; continue
        if op == "s_waitcnt":
            if "vmcnt(0)" in s: vm = {}
            m2 = re.search(r"lgkmcnt\((\d+)\)", s)
            if m2:   # counted wait: LDS reads return in order, so all but the newest N instructions have landed
                keep = int(m2.group(1))
                if keep == 0: lgkm = {}
                elif in_asm:
                    issued = sorted(set(lgkm.values()))
                    alive = set(issued[-keep:]) if keep < len(issued) else set(issued)
                    lgkm = {r: l for r, l in lgkm.items() if l in alive}
                # (a counted wait the COMPILER emitted counts its own loads, not ours: retires nothing here)
            continue
        ops = s[len(op):].split(";")[0]
        parts = [p.strip() for p in ops.split(",")]
        touched = regs(ops)
        hit = [r for r in touched if r in vm or r in lgkm]
        is_hidden_load = in_asm and (op.startswith("global_load") or op.startswith("ds_read"))
        if hit and not (is_hidden_load and set(hit) <= regs(parts[0]) and not (regs(",".join(parts[1:])) & set(hit))):
            print("%s:%d  %s\n    touches v%s with a hidden load in flight (issued at line %s)" % (kernel[:60], ln, s, sorted(hit), [vm.get(r, lgkm.get(r)) for r in sorted(hit)]))
            bad += 1
        if is_hidden_load:
            (vm if op.startswith("global_load") else lgkm).update({r: ln for r in regs(parts[0])})
    return bad

if __name__ == "__main__":
    n = audit(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
    print("violations:", n)
    sys.exit(1 if n else 0)
