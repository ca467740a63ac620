"""Scratch (spill) accesses, MFMAs and AGPR moves INSIDE the loops of a kernel (a loop = a backward branch spanning > 100 lines of ISA).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only [-D...] -S file.hip -o /tmp/file.s && python tools/loop_spills.py /tmp/file.s <name-substring>
"""
import re
import sys


def main():
    txt = open(sys.argv[1]).read().split("\n")
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    starts = [i for i, l in enumerate(txt) if re.match(r"^_Z\w+:", l)]
    for a, b in zip(starts, starts[1:] + [len(txt)]):
        if want not in txt[a]:
            continue
        body = txt[a:b]
        labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        print(txt[a].split(":")[0], len(body), "lines;", sum("scratch_" in l for l in body), "scratch accesses in all")
        for i, l in enumerate(body):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and 100 < i - labels[m.group(1)]:
                seg = body[labels[m.group(1)]:i]
                cnt = {k: sum(k in x for x in seg) for k in ("v_mfma", "ds_read", "scratch_store", "scratch_load", "v_accvgpr", "global_load", "s_barrier")}
                print(f"  loop lines {labels[m.group(1)]} .. {i}: {cnt}")


if __name__ == "__main__":
    main()
