"""GPU diagnostic: run every stage of the HIP engine against the CPU oracle on a toy grid and print
per-stage / per-buffer error metrics (development aid; the parity tests proper live in tests/)."""
import sys
import time
import traceback
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import pangu_oracle as O  # noqa: E402
from skyrim_amd.pangu.engine import PanguEngine  # noqa: E402
from skyrim_amd.pangu.spec import PanguGeometry, init_synthetic, synthetic_state  # noqa: E402


def err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def report(name, a, b):
    e = err(a, b)
    flag = "" if e < 5e-3 else "   <<<<<< BAD"
    print(f"  {name:34s} rel-max-err {e:.3e}  (ref max {b.abs().max().item():.3e}){flag}", flush=True)
    return e


def planes(buf, n, npl, dtype):
    """sum of hi(+lo) planes of a 16-bit buffer -> fp32 [n]"""
    t = buf.view(dtype)
    per = t.numel() // npl
    out = t[:n].float()
    if npl == 2:
        out = out + t[per:per + n].float()
    return out


def unblock(flat, rows, cols):
    """[rows/16][cols/32][16][32] blocked planes -> [rows][cols]"""
    return flat[:rows * cols].reshape(rows // 16, cols // 32, 16, 32).permute(0, 2, 1, 3).reshape(rows, cols)


def block_reference(p, x, res, heads, roll):
    """oracle block with intermediates (q,k,v per window/head, attention output, hidden, result)"""
    Z, H, W = res
    C = x.shape[-1]
    wz, wh, ww = O.WINDOW
    xs = x.reshape(Z, H, W, C)
    Hp, top, bot = O._centre_pad(H, wh)
    xs = F.pad(xs, (0, 0, 0, 0, top, bot))
    mask = None
    if roll:
        xs = torch.roll(xs, shifts=(-1, -3, -6), dims=(0, 1, 2))
        mask = O.shifted_window_mask(Z, Hp, W, xs.dtype)
    nZ, nH, nW = Z // wz, Hp // wh, W // ww
    xw = xs.reshape(nZ, wz, nH, wh, nW, ww, C).permute(0, 2, 4, 1, 3, 5, 6).reshape(nZ * nH, nW, 144, C)
    qkv = F.linear(xw, p["attn.qkv.weight"], p["attn.qkv.bias"]).reshape(nZ * nH, nW, 144, 3, heads, 32).permute(3, 0, 1, 4, 2, 5)
    q, k, v = qkv[0] * 32 ** -0.5, qkv[1], qkv[2]
    att = q @ k.transpose(-1, -2)
    idx = O.position_index().reshape(-1)
    bias = p["attn.bias_table"][idx].reshape(144, 144, nZ * nH, heads).permute(2, 3, 0, 1)
    att = att + bias[:, None]
    if mask is not None:
        att = att + mask[:, :, None]
    att = torch.softmax(att, -1)
    ao = (att @ v).permute(0, 1, 3, 2, 4).reshape(nZ * nH * nW * 144, C)
    y = O.earth_block(p, x, res, heads, roll)
    # hidden
    xw2 = F.linear(ao, p["attn.proj.weight"], p["attn.proj.bias"]).reshape(nZ, nH, nW, wz, wh, ww, C)
    xw2 = xw2.permute(0, 3, 1, 4, 2, 5, 6).reshape(Z, Hp, W, C)
    if roll:
        xw2 = torch.roll(xw2, shifts=(1, 3, 6), dims=(0, 1, 2))
    xa = xw2[:, top:top + H].reshape(-1, C)
    xa = x + F.layer_norm(xa, (C,), p["norm1.weight"], p["norm1.bias"], 1e-5)
    hid = F.gelu(F.linear(xa, p["mlp.fc1.weight"], p["mlp.fc1.bias"]))
    return dict(q=q.reshape(-1), k=k.reshape(-1), vt=v.transpose(-1, -2).reshape(-1), ao=ao, hid=hid, y=y, bias=bias, mask=mask)


def run(prec, g, params, x):
    print(f"=== precision {prec}  grid {g.n_lat}x{g.n_lon}", flush=True)
    taps = {}
    t0 = time.time()
    y_ref = O.forward(params, x, taps=taps)
    print(f"  oracle step {time.time() - t0:.2f}s", flush=True)
    eng = PanguEngine(g, prec)
    eng.load_params(params)
    dev = eng.device
    npl = 2
    t16 = torch.bfloat16 if prec.startswith("bf16x3") else torch.float16
    hnpl, ht16 = (1, torch.float16) if prec.endswith("h") else (npl, t16)

    xd = x.to(dev)
    # window tables
    for r, layer in ((0, 1), (1, 2)):
        for roll in (0, 1):
            idx = eng.debug_buffer(f"widx{r}{roll}", torch.int32).cpu()
            Z, H, W = g.res(layer)
            Hp, top = g.padded_lat(layer), g.pad_top(layer)
            tok = torch.arange(Z * H * W).reshape(Z, H, W)
            t = F.pad(tok + 1, (0, 0, top, Hp - H - top)) - 1
            if roll:
                t = torch.roll(t, shifts=(-1, -3, -6), dims=(0, 1, 2))
            t = t.reshape(Z // 2, 2, Hp // 6, 6, W // 12, 12).permute(0, 2, 4, 1, 3, 5).reshape(-1)
            print(f"  widx{r}{roll} exact: {bool((idx[:t.numel()] == t.int()).all())}")
    e = eng.patch_embed(xd)
    report("patch_embed", e, taps["embed"])
    for layer, i, xin in ((1, 0, taps["embed"]), (1, 1, taps["layer1.block0"]), (2, 0, taps["down"]), (2, 1, None)):
        bp = O._block_params(params, layer, i)
        if xin is None:
            xin = O.earth_block(O._block_params(params, 2, 0), taps["down"], g.res(2), O.HEADS[1], False)
        ref = block_reference(bp, xin, g.res(layer), O.HEADS[layer - 1], i % 2 == 1)
        yb = eng.block(layer, i, xin.to(dev))
        print(f" block layer{layer}.block{i}:")
        nq = ref["q"].numel()
        # expanded bias check
        blk = {(1, 0): 0, (1, 1): 1, (2, 0): 2, (2, 1): 3}[(layer, i)]
        be = eng.debug_buffer(f"bias_exp{blk}", torch.float16).float().cpu()
        types, heads = ref["bias"].shape[:2]
        full = ref["bias"] + (ref["mask"][:, 0][:, None] if ref["mask"] is not None else 0)
        be = be.reshape(types, heads, 9, 2304)
        lane = torch.arange(64)
        q_of = (torch.arange(9)[:, None] * 16 + (lane & 15)[None, :])                      # [qf][lane]
        k_pair = 32 * torch.arange(4)[:, None, None] + 8 * (lane >> 4)[None, :, None] + torch.arange(8)[None, None, :]   # [kb][lane][8]
        k_last = 128 + 4 * (lane >> 4)[:, None] + torch.arange(4)[None, :]                 # [lane][4]
        exp_pair = full[:, :, q_of[:, None, :, None].expand(9, 4, 64, 8), k_pair[None].expand(9, 4, 64, 8)].reshape(types, heads, 9, 2048)
        exp_last = full[:, :, q_of[:, :, None].expand(9, 64, 4), k_last[None].expand(9, 64, 4)].reshape(types, heads, 9, 256)
        report("bias_exp", be, torch.cat([exp_pair, exp_last], dim=-1))
        for name in ("q", "k", "vt"):
            got = planes(eng.debug_buffer(name, torch.uint8), nq, 1, torch.float16)
            report(name, got, ref[name])
        report("ao", unblock(planes(eng.debug_buffer("ao", torch.uint8), ref["ao"].numel(), npl, t16), *ref["ao"].shape), ref["ao"])
        report("hid", unblock(planes(eng.debug_buffer("hid", torch.uint8)[:ref["hid"].numel() * 2 * hnpl] if hnpl == 1 else eng.debug_buffer("hid", torch.uint8), ref["hid"].numel(), hnpl, ht16), *ref["hid"].shape), ref["hid"])
        report("block out", yb, ref["y"])
    d = eng.downsample(taps["layer1.block1"].to(dev))
    report("downsample", d, taps["down"])
    u = eng.upsample(taps["layer3"].to(dev))
    report("upsample", u, taps["up"])
    r = eng.patch_recover(taps["layer1.block1"].to(dev), taps["layer4"].to(dev))
    report("patch_recover", r, y_ref)
    torch.cuda.synchronize()
    t0 = time.time()
    y = eng.step(xd)
    torch.cuda.synchronize()
    print(f"  engine step (first call) {1e3 * (time.time() - t0):.1f} ms")
    pc = O.per_channel_rel_err(y.cpu(), y_ref)
    print(f"  STEP per-channel rel err: max {pc.max().item():.3e} (ch {pc.argmax().item()}), median {pc.median().item():.3e}")
    # in-place + 4-step rollout
    xs, xr = xd.clone(), x
    for _ in range(4):
        eng.step(xs, xs)
        xr = O.forward(params, xr)
    pc = O.per_channel_rel_err(xs.cpu(), xr)
    print(f"  4-STEP rollout per-channel rel err: max {pc.max().item():.3e}")
    for _ in range(3):
        eng.step(xd)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(10):
        eng.step(xd)
    torch.cuda.synchronize()
    print(f"  engine step {1e3 * (time.time() - t0) / 10:.2f} ms")


if __name__ == "__main__":
    nlat, nlon = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (49, 192)
    g = PanguGeometry(nlat, nlon)
    params = init_synthetic(g, 0)
    x = synthetic_state(g, 0)
    for prec in (sys.argv[3:] or ["bf16x3", "f16x3q"]):
        try:
            run(prec, g, params, x)
        except Exception:
            traceback.print_exc()
