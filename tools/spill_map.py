"""Where a kernel's scratch spills sit: counts scratch loads / stores, MFMAs and AGPR moves between consecutive s_barrier instructions.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S file.hip -o /tmp/file.s && python tools/spill_map.py /tmp/file.s [name-substring]
"""
import re
import sys
from collections import Counter


def main():
    txt = open(sys.argv[1]).read().split("\n")
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    starts = [i for i, l in enumerate(txt) if re.match(r"^_Z\w+:", l)]
    for a, b in zip(starts, starts[1:] + [len(txt)]):
        name = txt[a].split(":")[0]
        if want not in name:
            continue
        nb, c = 0, Counter()
        for l in txt[a:b]:
            if l.strip().startswith(";"):
                continue
            if "s_barrier" in l:
                nb += 1
            for key in ("scratch_store", "scratch_load", "v_mfma", "v_accvgpr", "ds_read", "global_load", "global_store"):
                if key in l:
                    c[(nb, key)] += 1
        print(name, b - a, "lines")
        for k in range(nb + 1):
            row = {key: c[(k, key)] for key in ("v_mfma", "ds_read", "scratch_store", "scratch_load", "v_accvgpr", "global_load", "global_store") if c[(k, key)]}
            print(f"  after barrier {k:3d}: {row}")


if __name__ == "__main__":
    main()
