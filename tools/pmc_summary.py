"""Per-kernel summary of the rocprofv3 counter passes written by tools/pmc_collect.sh.

    python tools/pmc_summary.py gpurun_out/pmc_<tag> profiles/r04_<tag>_pmc.json [--steps N] [--per-step K | --bench-json FILE] [--stamp S]

--per-step K (or --bench-json: a bench.py output line, K = the sum of its stages' launches_per_step): only the LAST steps x K dispatches of
every pass are counted, i.e. the bench's own steps.  Without it the one-time kernels of load_params (GraphCast: embedders and prepared edge
terms over 5 M rows, with the same kernel names as the step's) are summed into the per-step totals -- the reason round 2 and round 3 reported
235 and 278 GB per step for unchanged step kernels: their load paths differed.

For every kernel (demangled name, template arguments shortened) over all its dispatches: calls, average duration under the
counters, and per dispatch
  mfma_busy_pct     SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)          share of CU-busy time the matrix pipe of a SIMD is busy
  mfma_clk_per_inst SQ_VALU_MFMA_BUSY_CYCLES / SQ_INSTS_MFMA                          (16 for v_mfma_f32_16x16x32_f16: calibration)
  wait_pct / stall_pct / issue_pct   SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (disjoint, guide "PMC slots")
  lds_active_pct    SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES, bank conflicts as a share of LDS-active cycles
  hbm_GB, hbm_GBps  FETCH_SIZE (KiB, doubled: gfx950 tallies 128-B read requests at 64 B -- MI355X_MICROARCH.md "HBM") + WRITE_SIZE (KiB)
The clock under the counters is GRBM_GUI_ACTIVE / 8 XCDs / duration."""
import csv
import json
import re
import subprocess
import sys
from collections import defaultdict
from pathlib import Path


def short(name: str) -> str:
    if name.startswith("_Z"):
        # GNU c++filt does not know the _Float16 / __bf16 manglings (DF16_, DF16b): demangle with "half" (Dh) standing in for both
        tagged = name.replace("DF16b", "Dh").replace("DF16_", "Dh")
        out = subprocess.run(["c++filt", tagged], capture_output=True, text=True).stdout.strip()
        if out and not out.startswith("_Z"):
            name = out.replace("__fp16", "f16" if "DF16_" in name else "bf16").replace("half", "f16" if "DF16_" in name else "bf16")
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("skp::", "").replace("void ", "")
    name = re.sub(r"TileCfg<(\d+), (\d+), \d+, \d+, \d+>", r"\1x\2", name)
    name = re.sub(r"(APlanes|AConcatPlanes|EpGelu|EpQKV|EpStorePlanes|SinkResidual|SinkStore)<[^<>]*>", r"\1", name)
    name = re.sub(r"DmaArgs<.*$|GemmArgs<.*$|MlpArgs<.*$|ProjArgs<.*$", "", name)
    name = re.sub(r"\s+", " ", name).strip()
    return name[:160]


def load(d: Path):
    f = next(iter(sorted(d.glob("**/*counter_collection.csv"))), None)
    if f is None:
        return []
    return list(csv.DictReader(open(f)))


def launches_per_step(bench_json: str) -> int:
    line = [l for l in open(bench_json).read().strip().splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    return sum(int(v["launches_per_step"]) for v in d["roofline"]["stages"].values())


def main(prefix, out, *rest):
    steps = int(rest[rest.index("--steps") + 1]) if "--steps" in rest else 3
    per_step = int(rest[rest.index("--per-step") + 1]) if "--per-step" in rest else (launches_per_step(rest[rest.index("--bench-json") + 1]) if "--bench-json" in rest else 0)
    per = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(int)
    dur = defaultdict(float)
    cache = {}
    for pas in ("sq1", "sq2", "fetch", "write"):
        rows = load(Path(f"{prefix}_{pas}"))
        seen = set()
        for r in rows:
            if r["Kernel_Name"] not in cache:
                cache[r["Kernel_Name"]] = short(r["Kernel_Name"])
        keep = lambda k: not ("prep_" in k or "at::" in k or "rocclr" in k or "split_planes" in k)          # noqa: E731
        if per_step:                                     # the bench's own steps: the last steps x K dispatches of the engine's kernels
            ids = sorted({int(r["Dispatch_Id"]) for r in rows if keep(cache[r["Kernel_Name"]])})
            first = ids[-steps * per_step] if len(ids) >= steps * per_step else (ids[0] if ids else 0)
            rows = [r for r in rows if int(r["Dispatch_Id"]) >= first]
        for r in rows:
            kn = r["Kernel_Name"]
            k = cache[kn]
            if not keep(k):
                continue
            per[k][r["Counter_Name"] + "@" + pas] += float(r["Counter_Value"])
            if pas == "sq1" and (kn, r["Dispatch_Id"]) not in seen:
                seen.add((kn, r["Dispatch_Id"]))
                calls[k] += 1
                dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    res = {}
    for k in sorted(calls, key=lambda k: -dur[k]):
        n, c = calls[k], per[k]
        g = lambda name, pas="sq1": c.get(name + "@" + pas, 0.0) / n          # noqa: E731
        busy_cu, wave = g("SQ_BUSY_CU_CYCLES"), g("SQ_WAVE_CYCLES")
        busy2 = g("GRBM_GUI_ACTIVE", "sq2")
        e = {"calls": n, "avg_us": round(dur[k] / n, 2),
             "clock_GHz": round(g("GRBM_GUI_ACTIVE") / 8 / (dur[k] / n * 1e3), 3) if dur[k] else None,      # the counter sums the 8 XCDs
             "mfma_busy_pct": round(100 * g("SQ_VALU_MFMA_BUSY_CYCLES") / (4 * busy_cu), 2) if busy_cu else None,
             "mfma_clk_per_inst": round(g("SQ_VALU_MFMA_BUSY_CYCLES") / g("SQ_INSTS_MFMA"), 2) if g("SQ_INSTS_MFMA") else None,
             "wait_pct": round(100 * g("SQ_WAIT_ANY") / wave, 1) if wave else None,
             "stall_pct": round(100 * g("SQ_WAIT_INST_ANY") / wave, 1) if wave else None,
             "issue_pct": round(100 * g("SQ_ACTIVE_INST_ANY") / wave, 1) if wave else None,
             "lds_stall_pct": round(100 * g("SQ_WAIT_INST_LDS") / wave, 1) if wave else None,
             "lds_active_per_gui": round(g("SQ_LDS_IDX_ACTIVE", "sq2") / busy2, 3) if busy2 else None,
             "lds_conflict_pct": round(100 * g("SQ_LDS_BANK_CONFLICT", "sq2") / g("SQ_LDS_IDX_ACTIVE", "sq2"), 2) if g("SQ_LDS_IDX_ACTIVE", "sq2") else None,
             "fetch_GB": round(2 * 1024 * g("FETCH_SIZE", "fetch") / 1e9, 4), "write_GB": round(1024 * g("WRITE_SIZE", "write") / 1e9, 4)}
        e["hbm_GB"] = round(e["fetch_GB"] + e["write_GB"], 4)
        e["hbm_GBps_under_pmc"] = round(e["hbm_GB"] / (e["avg_us"] * 1e-6), 1) if e["avg_us"] else None
        e["raw_per_dispatch"] = {kk: round(v / n, 1) for kk, v in sorted(c.items())}
        res[k] = e
    total = {"hbm_GB_per_step": round(sum(e["hbm_GB"] * e["calls"] for e in res.values()) / steps, 2),
             "write_GB_per_step": round(sum(e["write_GB"] * e["calls"] for e in res.values()) / steps, 2),
             "kernel_ms_per_step_under_pmc": round(sum(e["avg_us"] * e["calls"] for e in res.values()) / steps / 1e3, 2), "steps": steps,
             "launches_per_step": per_step or None, "scope": "the bench's own steps" if per_step else "every dispatch of the process (load-time kernels included)"}
    stamp = rest[rest.index("--stamp") + 1] if "--stamp" in rest else None      # e.g. sha256 of the profiled .so, so that bench.py can say what was profiled
    json.dump({"total": total, "stamp": stamp, "kernels": res}, open(out, "w"), indent=1)
    print(json.dumps(total))
    for k, e in res.items():
        print(f"{e['avg_us']:9.1f} us x{e['calls']:4d}  mfma {e['mfma_busy_pct']}%  wait {e['wait_pct']} stall {e['stall_pct']} issue {e['issue_pct']}  "
              f"lds {e['lds_active_per_gui']} confl {e['lds_conflict_pct']}%  hbm {e['hbm_GB']} GB ({e['write_GB']} W)  {k[:90]}")


if __name__ == "__main__":
    main(*sys.argv[1:])
