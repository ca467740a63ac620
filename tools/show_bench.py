import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "ms/step", round(d["ms_per_step"], 2), "steps/s", round(d["value"], 2), "parity", d.get("parity", {}).get("max_rel_err"))
        for k, v in d["roofline"]["stages"].items():
            print(f"    {k:11s} {v['ms_per_launch']:8.4f} ms x{v['launches_per_step']:2d} = {v['ms_per_launch']*v['launches_per_step']:7.3f}  {v['tflops']:6.1f} TF  {v['alg_GBps']:7.1f} GB/s")
    except Exception as e:
        print(f, "ERR", e, open(f).read()[-800:])
