#!/usr/bin/env python3
"""Audit of the compiler's output (CPU only): for every kernel of a gfx950 assembly file, the loops that contain matrix instructions or LDS-DMA / buffer
loads AND spill traffic (v_readlane / v_writelane = spilled SGPRs coming back, scratch_ / buffer_*_dword ... offen spills) or waterfall loops.
usage: hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only x.hip -o x.s && python scripts/asm_loop_audit.py x.s"""
import re, sys, subprocess
lines = open(sys.argv[1]).read().split("\n")
kern, start = None, 0
kernels = []
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\w+):\s*(;.*)?$", l)
    if m:
        if kern: kernels.append((kern, start, i))
        kern, start = m.group(1), i
    if l.strip().startswith("s_endpgm") and kern:
        kernels.append((kern, start, i)); kern = None
def demangle(n):
    try: return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    except Exception: return n
for name, a, b in kernels:
    body = lines[a:b]
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m: labels[m.group(1)] = i
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l) or re.search(r"s_branch (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    rep = []
    for s, e in loops:
        seg = body[s:e + 1]
        mf = sum("v_mfma" in x for x in seg); sp = sum(("v_readlane_b32" in x or "v_writelane_b32" in x) for x in seg)
        wf = sum("s_and_saveexec_b64 vcc, vcc" in x for x in seg); sc = sum(x.strip().startswith("scratch_") for x in seg)
        if (mf or any("buffer_load" in x for x in seg)) and (sp or wf or sc):
            rep.append((e - s, mf, sp, wf, sc))
    if rep:
        print(demangle(name)[:150])
        for n, mf, sp, wf, sc in sorted(set(rep)):
            print(f"    loop of {n:5d} lines: {mf:4d} mfma, {sp:3d} readlane/writelane, {wf:2d} waterfall, {sc:2d} scratch ops")
