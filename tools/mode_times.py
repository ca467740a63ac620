import sys, time, torch
sys.path.insert(0, '/root/repo')
from skyrim_amd.pangu.engine import PanguEngine
from skyrim_amd.pangu.spec import PanguGeometry, init_synthetic, synthetic_state
g = PanguGeometry(721, 1440); p = init_synthetic(g, 0); x0 = synthetic_state(g, 0)
for m in sys.argv[1:]:
    eng = PanguEngine(g, m); eng.load_params(p); x = x0.to(eng.device)
    for _ in range(2): eng.step(x, x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(6): eng.step(x, x)
    torch.cuda.synchronize(); print(m, 'ms/step %.2f' % ((time.perf_counter() - t0) / 6 * 1e3), flush=True)
    del eng; torch.cuda.empty_cache()
