#!/usr/bin/env python3
"""Cuts one kernel out of `hipcc -S --cuda-device-only` output and prints an instruction census (usage: isa_extract.py file.s
<substring of the mangled name> [out.s]).  Used to look for spills (v_readlane/v_writelane/scratch_*) and lone waits in hot loops."""
import sys
from collections import Counter

src, key = open(sys.argv[1]).read().split("\n"), sys.argv[2]
start = next(i for i, l in enumerate(src) if l.startswith("_ZN") and key in l.split(":")[0] and l.rstrip().split(";")[0].rstrip().endswith(":"))
end = next(i for i in range(start, len(src)) if src[i].startswith(".Lfunc_end"))
body = src[start:end]
ins = [l.strip() for l in body if l.startswith("\t") and not l.strip().startswith((".", ";"))]
c = Counter(i.split()[0] for i in ins)
groups = {"valu": sum(v for k, v in c.items() if k.startswith("v_")), "salu": sum(v for k, v in c.items() if k.startswith("s_")),
          "vmem": sum(v for k, v in c.items() if k.startswith(("global_", "buffer_", "flat_"))), "lds": sum(v for k, v in c.items() if k.startswith("ds_")),
          "scratch": sum(v for k, v in c.items() if k.startswith("scratch_")), "readlane": c["v_readlane_b32"], "writelane": c["v_writelane_b32"],
          "waitcnt": c["s_waitcnt"], "branches": sum(v for k, v in c.items() if k.startswith(("s_cbranch", "s_branch")))}
print(src[start].split(":")[0][:60], len(ins), groups)
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write("\n".join(body))
