"""Per-kernel VGPR / scratch (spill) usage of the HIP sources: hipcc -Rpass-analysis=kernel-resource-usage, condensed."""
import re
import subprocess
import sys
from pathlib import Path

CSRC = Path(__file__).resolve().parent.parent / "skyrim_amd" / "csrc"


def main():
    files = sys.argv[1:] or ["ops_attn", "ops_mlp", "ops_updown", "ops_embed_recover", "attention"]
    for f in files:
        r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", f"{f}.hip", "-o", "/dev/null",
                            "-Rpass-analysis=kernel-resource-usage"], cwd=CSRC, capture_output=True, text=True)
        name = None
        rec = {}
        for line in r.stderr.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
                name = re.sub(r"skp::|\(skp::.*$|void ", "", name)
                rec = {}
            for key in ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"):
                m = re.search(re.escape(key) + r": (\d+)", line)
                if m and name:
                    rec[key] = int(m.group(1))
                    if key.startswith("LDS"):
                        flag = "  <<< SPILL" if rec.get("ScratchSize [bytes/lane]", 0) else ""
                        print(f"{f:18s} v{rec.get('VGPRs', 0):3d} a{rec.get('AGPRs', 0):3d} scratch {rec.get('ScratchSize [bytes/lane]', 0):4d} "
                              f"occ {rec.get('Occupancy [waves/SIMD]', 0)}  {name[:150]}{flag}")


if __name__ == "__main__":
    main()
