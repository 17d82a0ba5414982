#!/usr/bin/env python3
"""Audit of the hand-placed LDS reads of kernels_mfma16.h in the compiled ISA.

The corpus fragments are fetched by `asm volatile` ds_read_b128 statements hipcc does not count: between such a read and
the explicit `s_waitcnt lgkmcnt(N)` that covers it, nothing may read the destination registers (an MFMA would multiply
stale data, a compiler-made copy would carry it on).  This walks every mfma16_topk_kernel instantiation of the device
assembly (`make -C theoremsearch_amd/csrc asm`), keeps the queue of outstanding LDS / scalar-memory operations in issue
order (LDS returns in order; `lgkmcnt(N)` retires all but the youngest N), and reports any instruction that reads a
vector register an outstanding ds_read_b128 is still to write.  The walk is linear in the text and runs twice, so that
what a loop leaves in flight at its end is seen by its head.

Second check, same walk: the MFMAs are text inside `asm volatile` too, so hipcc's hazard recognizer does not see them: a
vector instruction of the compiler's (a copy, a v_accvgpr_read that brings a parked query fragment back) that writes an MFMA
operand within two instructions in front of that MFMA gets no wait states - on the GPU those k-steps came out wrong
(round 5, the first k-split build).  Reported as "operand written right in front of its MFMA".

    python tools/audit_ring.py theoremsearch_amd/csrc/build/asm/launch_mfma16-hip-amdgcn-amd-amdhsa-gfx950.s
"""
import re
import sys

REG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")


def regs(op):
    out = set()
    for m in REG.finditer(op):
        if m.group(1):
            out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def audit(name, lines):
    fifo = []          # outstanding lgkm operations: set of destination registers (empty for the others)
    bad = []
    recent = []        # destination registers of the last two instructions, if they were the compiler's vector instructions
    for rnd in range(2):
        for no, ln in lines:
            t = ln.split(";")[0].strip()
            if not t or t.endswith(":") or t.startswith("."):
                continue
            op, _, rest = t.partition(" ")
            ops = [o.strip() for o in rest.split(",")] if rest else []
            if op.startswith("v_mfma") and rnd == 1:
                if set().union(*[regs(o) for o in ops[1:]]) & (set().union(*recent) if recent else set()):
                    bad.append((no, "operand written right in front of its MFMA: " + t))
            valu = op.startswith("v_") and not op.startswith(("v_mfma", "v_cmp", "v_cmpx", "v_readfirstlane", "v_readlane"))
            recent = (recent + [regs(ops[0]) if valu and ops else set()])[-2:]
            if op == "s_waitcnt":
                m = re.search(r"lgkmcnt\((\d+)\)", t)
                if m:
                    keep = int(m.group(1))
                    while len(fifo) > keep:
                        fifo.pop(0)
                continue
            pending = set().union(*fifo) if fifo else set()
            if op.startswith("ds_read"):
                src = set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
                if src & pending and rnd == 1:
                    bad.append((no, t))
                fifo.append(regs(ops[0]))
                continue
            if op.startswith(("ds_", "s_load", "s_buffer_load", "s_memtime", "s_memrealtime")):
                if set().union(*[regs(o) for o in ops]) & pending and rnd == 1:
                    bad.append((no, t))
                fifo.append(set())
                continue
            if not pending:
                continue
            # sources: every operand but the destination (MFMA: A, B and C; stores and DMA: all of them)
            has_dst = op.startswith(("v_", "global_load_dword", "buffer_load_dword")) and not op.startswith(("global_load_lds", "v_cmp", "v_cmpx"))
            if op.startswith(("global_load_lds", "buffer_load_dwordx4")) and "lds" in t:
                has_dst = False
            srcs = ops[1:] if has_dst else ops
            used = set().union(*[regs(o) for o in srcs]) if srcs else set()
            if used & pending and rnd == 1:
                bad.append((no, t))
    return bad


def main(path):
    text = open(path).read().splitlines()
    kernels, cur, name = {}, None, None
    for i, ln in enumerate(text, 1):
        m = re.match(r"^(_ZN2ts18mfma16_topk_kernel\w+):", ln)
        if m:
            name, cur = m.group(1), []
            kernels[name] = cur
            continue
        if cur is not None:
            cur.append((i, ln))
            if "s_endpgm" in ln:
                cur = None
    total = 0
    for name, lines in kernels.items():
        bad = audit(name, lines)
        reads = sum(1 for _, l in lines if "ds_read_b128" in l)
        print(f"{name}: {reads} ds_read_b128, {len(bad)} reads of a register still in flight")
        for no, t in bad[:12]:
            print(f"    line {no}: {t}")
        total += len(bad)
    print("kernels audited:", len(kernels), "violations:", total)
    return 1 if total or not kernels else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
