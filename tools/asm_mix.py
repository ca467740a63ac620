"""Instruction mix of a kernel's epilogue (everything after the last v_mfma) from `hipcc -S` output.
usage: asm_mix.py file.s <substring of mangled kernel name> [...more substrings]"""
import collections
import re
import sys

src = open(sys.argv[1]).read().splitlines()
want = sys.argv[2:]
starts = [(i, l.split(":")[0]) for i, l in enumerate(src) if re.match(r"^_Z\w+:", l)]
for j, (i, name) in enumerate(starts):
    if not all(w in name for w in want):
        continue
    end = starts[j + 1][0] if j + 1 < len(starts) else len(src)
    body = []
    for l in src[i + 1:end]:
        t = l.strip()
        if t.startswith("s_endpgm"):
            break
        if t and not t.startswith((";", ".")) and not t.endswith(":"):
            body.append(t)
    last = max(k for k, l in enumerate(body) if l.startswith("v_mfma"))
    epi = body[last + 1:]
    c = collections.Counter(l.split()[0] for l in epi)
    print(name[:110], "| total", len(body), "epilogue", len(epi))
    print("   ", ", ".join(f"{k} {v}" for k, v in c.most_common(45)))
