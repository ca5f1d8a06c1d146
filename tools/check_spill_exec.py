#!/usr/bin/env python3
"""Lint the gfx950 ISA of the step kernels for a register-allocator hazard seen with ROCm 7.2's LLVM.

When a VGPR is spilled (to an AGPR with v_accvgpr_write, or to scratch) at the head of a control-flow join
block, the spill can be placed *before* the `s_or_b64 exec, exec, s[..]` that re-enables the lanes parked
by the preceding divergent loop / branch.  VALU and scratch stores honour exec, so the parked lanes never
save (or reload) the value and later read garbage (r01: the Newton kernel lost its qpos address).
v_writelane (SGPR spills) ignores exec and is harmless there, and so is a spill bracketed by `s_or_saveexec_b64 sX, -1` ...
`s_mov_b64 exec, sX` (whole-wave mode: how the VGPR that carries spilled SGPRs is itself saved and restored).

usage: check_spill_exec.py file.s     (hipcc -S --cuda-device-only output); exit code 1 when a hazard is found
"""
import re
import sys


def scan(path):
    hazards = []
    kernel = None
    pending = []  # exec-sensitive spill instructions seen since the last label, before any exec restore
    in_head = False
    wwm = None    # scalar register pair holding the saved exec while exec is forced to all ones
    for ln, line in enumerate(open(path), 1):
        s = line.strip()
        m = re.match(r"^(_Z\w+):", s)
        if m:
            kernel = m.group(1)
        if re.match(r"^\.?[\w$.]+:", s):  # any label starts a block
            pending, in_head, wwm = [], True, None
            continue
        mw = re.match(r"s_or_saveexec_b64\s+(s\[\d+:\d+\]),\s*-1", s)
        if mw:
            wwm = mw.group(1)
            continue
        if wwm and re.match(r"s_mov_b64\s+exec,\s*" + re.escape(wwm), s):
            wwm = None
            continue
        if not in_head or not s or s.startswith(";"):
            continue
        op = s.split()[0]
        if op in ("s_or_b64", "s_or_b32") and re.match(r"s_or_b(64|32)\s+exec(_lo)?,\s*exec", s):
            for h in pending:
                hazards.append((kernel, h[0], h[1], ln))
            pending = []
            continue
        if op in ("v_accvgpr_write_b32", "v_accvgpr_read_b32") or (
                op.startswith(("scratch_", "buffer_")) and ("Spill" in line or "Reload" in line)):
            if not wwm:
                pending.append((ln, s))
            continue
        # anything that is not scalar bookkeeping ends the block head
        if op.startswith(("s_", "v_writelane", "v_readlane")) and not op.startswith(
                ("s_cbranch", "s_branch", "s_setpc", "s_endpgm", "s_and_saveexec", "s_mov_b64 exec")):
            continue
        in_head = False
        pending = []
    return hazards


if __name__ == "__main__":
    bad = scan(sys.argv[1])
    for k, ln, ins, at in bad:
        print(f"{sys.argv[1]}:{ln}: exec-sensitive spill `{ins}` before exec restore at line {at}  [{k}]")
    print(f"{len(bad)} hazard(s)")
    sys.exit(1 if bad else 0)
