#!/usr/bin/env python3
"""isa_mix.py file.s [kernel-substring] -- instruction mix of a kernel in hipcc's assembly output (static counts)."""
import re, sys, collections
txt = open(sys.argv[1]).read()
sub = sys.argv[2] if len(sys.argv) > 2 else ""
# split into functions
funcs = re.split(r"\n(?=[_A-Za-z0-9$.]+:\s*;? *@)", txt)
for f in funcs:
    head = f.split("\n", 1)[0]
    if sub not in head or "@" not in head or "kernel" not in head:
        continue
    c = collections.Counter()
    n = 0
    for line in f.split("\n"):
        line = line.strip()
        if not line or line.startswith((";", ".", "//")) or line.endswith(":"):
            continue
        op = line.split()[0]
        if not re.match(r"^[a-z_0-9]+$", op):
            continue
        n += 1
        if re.match(r"v_(fma|mul|add|fmac|max|min|rcp|rsq|sqrt|div|cmp|cmpx|cndmask|ldexp|trig|frexp|fract|floor|rndne|cvt).*f64", op): c["valu_f64"] += 1
        elif op.startswith("v_accvgpr"): c["accvgpr_mov"] += 1
        elif op.startswith("v_mfma"): c["mfma"] += 1
        elif op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): c["lane_xfer"] += 1
        elif op.startswith("v_"): c["valu_other"] += 1
        elif op.startswith("scratch_"): c["scratch"] += 1
        elif op.startswith(("global_", "flat_", "buffer_")): c["vmem"] += 1
        elif op.startswith("ds_"): c["lds"] += 1
        elif op.startswith("s_load"): c["s_load"] += 1
        elif op.startswith("s_waitcnt"): c["s_waitcnt"] += 1
        elif op.startswith("s_nop"): c["s_nop"] += 1
        elif op.startswith("s_"): c["salu_other"] += 1
        else: c["other"] += 1
    print(head[:100])
    print("  total", n, dict(c))
