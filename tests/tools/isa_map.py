#!/usr/bin/env python
"""Map of a kernel's ISA (`hipcc -S --cuda-device-only` output): where, counted in matrix instructions, the compiler placed its
vmcnt waits, barriers and scratch (spill) accesses.  Usage: isa_map.py file.s [kernel-name substring]"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2] if len(sys.argv) > 2 else ""
starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
ends = [i for i, l in enumerate(lines) if ".amdhsa_kernel " in l]
for a in starts:
    name = lines[a].split(":")[0]
    if pat not in name:
        continue
    b = min([e for e in ends if e > a] or [len(lines)])
    mf, out, vm, ld = 0, [], 0, 0
    for l in lines[a:b]:
        if "v_mfma" in l:
            mf += 1
        elif "scratch_" in l:
            out.append(f"{mf}:{'S' if 'store' in l else 'L'}")
        elif "s_waitcnt" in l and "vmcnt" in l:
            out.append(f"{mf}:{l.strip().split(None, 1)[1].replace(' ', '')}")
        elif "s_barrier" in l:
            out.append(f"{mf}:BAR")
        elif "global_load" in l:
            ld += 1
    print(name[-60:], "mfma", mf, "global loads", ld, "lines", b - a)
    print(" ".join(out))
