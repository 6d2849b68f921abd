"""Developer tool: instruction mix of one kernel in a `hipcc -S --cuda-device-only` listing.
usage: isa_mix.py listing.s <substring of the mangled kernel name> [...]"""
import collections, re, sys
lines = open(sys.argv[1]).read().split("\n")
for pat in sys.argv[2:]:
    start = [i for i, l in enumerate(lines) if l.startswith("_Z") and pat in l.split(":")[0] and ":" in l]
    for s in start[:1]:
        c = collections.Counter(); n = 0
        for l in lines[s + 1:]:
            if l.startswith(".Lfunc_end"): break
            t = l.strip().split()
            if not t or t[0].startswith((";", ".")) or t[0].endswith(":"): continue
            op = t[0]; n += 1
            c[op] += 1
        cls = collections.Counter()
        for op, k in c.items():
            key = ("v_pk" if op.startswith("v_pk") else "v_div*" if op.startswith("v_div") else "transc" if re.match(r"v_(rcp|rsq|sqrt|sin|cos|exp|log)", op)
                   else "dpp/perm" if "dpp" in op or "permute" in op or "readlane" in op or "bpermute" in op else "valu" if op.startswith("v_") else "lds" if op.startswith("ds_")
                   else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "salu" if op.startswith("s_") else "other")
            cls[key] += k
        print(lines[s][:110]); print("  total", n, dict(cls))
        print("  top:", ", ".join(f"{op} {k}" for op, k in c.most_common(28)))
