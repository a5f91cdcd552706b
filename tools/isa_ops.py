#!/usr/bin/env python3
"""Compressed instruction-class trace of one kernel in a hipcc .s file (waits, barriers, MFMA, LDS, DMA, branches).
usage: isa_ops.py file.s <substring of the mangled kernel name> [start_line_count]"""
import re
import sys

s = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(s) if l.startswith("_Z") and key in l.split(":")[0])
end = next(i for i in range(start, len(s)) if ".end_amdhsa_kernel" in s[i] or s[i].startswith("\t.section") and i > start + 5)
pat = re.compile(r"\s*(s_waitcnt[^;]*|s_barrier|v_mfma\S*|ds_read\S*|ds_write\S*|global_load_lds\S*|global_store\S*|global_load\S*|"
                 r"s_cbranch\S*\s+\S+|s_branch\s+\S+|\.LBB\S+:|s_setprio \d|buffer_\S+|scratch_\S+|v_accvgpr\S*|s_endpgm)")
out, prev, cnt = [], None, 0
for l in s[start:end]:
    m = pat.match(l)
    if not m:
        continue
    o = m.group(1).strip()
    k = o if o.startswith(("s_waitcnt", ".LBB", "s_cbranch", "s_branch", "s_setprio")) else o.split()[0]
    if k == prev:
        cnt += 1
    else:
        if prev:
            out.append(f"{prev} x{cnt}" if cnt > 1 else prev)
        prev, cnt = k, 1
out.append(f"{prev} x{cnt}" if cnt > 1 else prev)
print("\n".join(out))
