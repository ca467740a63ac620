"""Shapes, parameter layout and synthetic inputs of the GraphCast step (operational 0.25-degree, 13-level model).

The reference loads it as ``earth2mip.networks.graphcast.load_time_loop_operational(registry.get_model("e2mip://graphcast"))``
(/root/reference/skyrim/core/models/graphcast.py:51-54) and steps it through ``stepper.initialize / stepper.step``
(:102-118) on DeepMind's JAX implementation.  Neither package nor checkpoint is available here: the network below
follows the paper (Lam et al. 2023, Methods) -- encoder / 16-step processor / decoder of interaction networks over the
icosahedral multi-mesh, latent size 512, one-hidden-layer swish MLPs with LayerNorm, residual updates, the output a
normalised residual of the latest state.  Channel order of the state = the reference's CHANNELS (graphcast.py:17-26).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F

_LEVELS = [50, 100, 150, 200, 250, 300, 400, 500, 600, 700, 850, 925, 1000]
CHANNELS = [f"{v}{l}" for v in "zqtuvw" for l in _LEVELS] + ["u10m", "v10m", "t2m", "msl", "tp06"]     # graphcast.py:17-26
N_FORCING = 15       # toa solar radiation at t-1, t, t+1 and (sin, cos) of day / year progress at the three times
N_STATIC = 2         # surface geopotential, land-sea mask


@dataclass(frozen=True)
class GraphcastConfig:
    n_lat: int = 721
    n_lon: int = 1440
    splits: int = 6              # icosahedron refinements (M6: 40 962 mesh nodes)
    latent: int = 512
    steps: int = 16              # processor message-passing steps
    n_vars: int = 83

    @property
    def grid_in(self):           # features of a grid node: two time levels, forcings, static fields, (cos lat, sin lon, cos lon)
        return 2 * self.n_vars + N_FORCING + N_STATIC + 3


def mlp_names(cfg: GraphcastConfig) -> list[tuple[str, int, int, bool]]:
    """(name, d_in, d_out, layer_norm) of every MLP, in forward order."""
    L = cfg.latent
    out = [("embed.grid", cfg.grid_in, L, True), ("embed.mesh", 3, L, True), ("embed.g2m_edge", 4, L, True),
           ("embed.mesh_edge", 4, L, True), ("embed.m2g_edge", 4, L, True),
           ("g2m.edge", 3 * L, L, True), ("g2m.mesh_node", 2 * L, L, True), ("g2m.grid_node", L, L, True)]
    for i in range(cfg.steps):
        out += [(f"proc.{i}.edge", 3 * L, L, True), (f"proc.{i}.node", 2 * L, L, True)]
    out += [("m2g.edge", 3 * L, L, True), ("m2g.grid_node", 2 * L, L, True), ("out", L, cfg.n_vars, False)]
    return out


def param_spec(cfg: GraphcastConfig) -> list[tuple[str, tuple]]:
    spec = [("norm.mean", (cfg.n_vars,)), ("norm.std", (cfg.n_vars,)), ("norm.diff_std", (cfg.n_vars,)), ("static", (N_STATIC, cfg.n_lat, cfg.n_lon))]
    for name, d_in, d_out, ln in mlp_names(cfg):
        spec += [(name + ".fc1.weight", (cfg.latent, d_in)), (name + ".fc1.bias", (cfg.latent,)),
                 (name + ".fc2.weight", (d_out, cfg.latent)), (name + ".fc2.bias", (d_out,))]
        if ln:
            spec += [(name + ".ln.weight", (d_out,)), (name + ".ln.bias", (d_out,))]
    return spec


def channel_stats(cfg: GraphcastConfig):
    mean = torch.linspace(-20.0, 5.0e4, cfg.n_vars)           # geopotential-sized means: the fp16 split must see normalised values
    std = torch.linspace(0.5, 3.0e3, cfg.n_vars)
    return mean, std, 0.1 * std


def init_synthetic(cfg: GraphcastConfig, seed: int = 0) -> dict:
    gen = torch.Generator().manual_seed(seed)
    mean, std, dstd = channel_stats(cfg)
    out = {}
    for name, shape in param_spec(cfg):
        if name == "norm.mean":
            t = mean
        elif name == "norm.std":
            t = std
        elif name == "norm.diff_std":
            t = dstd
        elif name == "static":
            t = torch.rand(shape, generator=gen)
        elif name.endswith("ln.weight"):
            t = 1.0 + 0.05 * torch.randn(shape, generator=gen)
        elif name.endswith(".bias"):
            t = 0.02 * torch.randn(shape, generator=gen)
        else:
            t = torch.randn(shape, generator=gen) * math.sqrt(1.0 / shape[-1])
        out[name] = t.float().contiguous()
    return out


def synthetic_states(cfg: GraphcastConfig, seed: int = 0):
    """Two consecutive (n_vars, n_lat, n_lon) fp32 states (t-1, t): smooth noise around the channel means, t = t-1 + small change."""
    gen = torch.Generator().manual_seed(2000 + seed)
    mean, std, dstd = channel_stats(cfg)

    def smooth(z):
        z = F.avg_pool2d(F.pad(z[None], (4, 4, 0, 0), mode="circular"), (1, 9), stride=1)[0]
        return F.avg_pool2d(F.pad(z[None], (0, 0, 4, 4), mode="replicate"), (9, 1), stride=1)[0] * 9.0

    z0 = smooth(torch.randn(cfg.n_vars, cfg.n_lat, cfg.n_lon, generator=gen))
    z1 = smooth(torch.randn(cfg.n_vars, cfg.n_lat, cfg.n_lon, generator=gen))
    x0 = mean[:, None, None] + std[:, None, None] * z0
    x1 = x0 + dstd[:, None, None] * z1
    return x0.float().contiguous(), x1.float().contiguous()


def forcings(cfg: GraphcastConfig, hours_since_epoch: float) -> torch.Tensor:
    """(N_FORCING, n_lat, n_lon): a closed-form top-of-atmosphere insolation proxy max(0, cos zenith) and the (sin, cos) day / year
    progress at t-6h, t, t+6h.  Deterministic in time: the engine and the oracle are handed the same tensor."""
    lat = torch.deg2rad(torch.linspace(90.0, -90.0, cfg.n_lat, dtype=torch.float64))[:, None]
    lon = torch.deg2rad(torch.arange(cfg.n_lon, dtype=torch.float64) * (360.0 / cfg.n_lon))[None, :]
    out = []
    for dt in (-6.0, 0.0, 6.0):
        h = hours_since_epoch + dt
        day, year = (h / 24.0) % 1.0, (h / (24.0 * 365.25)) % 1.0
        decl = -0.409 * math.cos(2 * math.pi * (year + 10.0 / 365.25))
        hour_angle = 2 * math.pi * day + lon - math.pi
        cosz = torch.sin(lat) * math.sin(decl) + torch.cos(lat) * math.cos(decl) * torch.cos(hour_angle)
        out.append(cosz.clamp_min(0.0))
        for v in (math.sin(2 * math.pi * day), math.cos(2 * math.pi * day), math.sin(2 * math.pi * year), math.cos(2 * math.pi * year)):
            out.append(torch.full((cfg.n_lat, cfg.n_lon), v, dtype=torch.float64))
    return torch.stack(out).float().contiguous()


def flops_per_stage(cfg: GraphcastConfig, n_grid: int, n_mesh: int, e_mesh: int, e_g2m: int, e_m2g: int, executed: bool = False) -> dict:
    """Dense FLOPs of the input-dependent MLPs (2 d_in L + 2 L d_out per row), per stage of the step (bench.py's stage names).
    ``executed=False``: the network as published -- every edge MLP on its concatenated 3L-wide row.  ``executed=True``: what the engine runs
    for the same result, the first Linear of every edge MLP taken apart by distributivity (engine.py: node terms once per node,
    input-independent edge terms once per model)."""
    L = cfg.latent
    lin = lambda rows, d_in, d_out: 2.0 * rows * d_in * d_out  # noqa: E731
    mlp = lambda rows, d_in, d_out: 2.0 * rows * (d_in * L + L * d_out)  # noqa: E731
    if executed:
        st = {"embed": mlp(n_grid, cfg.grid_in, L),
              # grid->mesh edges: sender term per grid node + second Linear per edge; encoder node updates
              "encoder": lin(n_grid, L, L) + lin(e_g2m, L, L) + mlp(n_mesh, 2 * L, L) + mlp(n_grid, L, L),
              "processor": cfg.steps * (lin(e_mesh, L, L) + lin(n_mesh, L, 2 * L) + lin(e_mesh, L, L) + mlp(n_mesh, 2 * L, L)),
              "decoder": lin(n_mesh, L, L) + lin(n_grid, L, L) + lin(e_m2g, L, L) + mlp(n_grid, 2 * L, L),
              "output": mlp(n_grid, L, cfg.n_vars)}
    else:
        st = {"embed": mlp(n_grid, cfg.grid_in, L),
              "encoder": mlp(e_g2m, 3 * L, L) + mlp(n_mesh, 2 * L, L) + mlp(n_grid, L, L),
              "processor": cfg.steps * (mlp(e_mesh, 3 * L, L) + mlp(n_mesh, 2 * L, L)),
              "decoder": mlp(e_m2g, 3 * L, L) + mlp(n_grid, 2 * L, L),
              "output": mlp(n_grid, L, cfg.n_vars)}
    return {k: float(v) for k, v in st.items()}


def flops_per_step(cfg: GraphcastConfig, n_grid: int, n_mesh: int, e_mesh: int, e_g2m: int, e_m2g: int) -> float:
    """Dense FLOPs of one step of the network as published."""
    return sum(flops_per_stage(cfg, n_grid, n_mesh, e_mesh, e_g2m, e_m2g).values())


def alg_bytes_per_step(cfg: GraphcastConfig, n_grid: int, n_mesh: int, e_mesh: int, e_g2m: int, e_m2g: int, latent_bytes: int = 4, edge_bytes: int | None = None) -> dict:
    """ALGORITHMIC HBM bytes of one step, per stage: every tensor of the network's data flow read once where it is consumed and written
    once where it is produced, at the engine's storage type (fp32 latents: ``latent_bytes`` = 4), with nothing materialised that the
    model does not define -- no concatenated edge rows, no hidden activations, no edge latents of the encoder / decoder beyond their
    (input-independent, prepared) first-Linear terms, the receiver sums folded into their producers.  Weights (35 M parameters) are
    L2-resident and not counted.  This is the floor the measured traffic (rocprofv3 FETCH_SIZE + WRITE_SIZE) is held against."""
    L, V = cfg.latent, cfg.n_vars
    row = L * latent_bytes
    erow = L * (edge_bytes or latent_bytes)            # the fused engine keeps edge latents / prepared edge terms as ONE fp16 plane (edge_bytes = 2)
    grid, mesh = n_grid * row, n_mesh * row
    st = {
        "embed": n_grid * cfg.grid_in * 4 + grid,                                      # features in, grid latents out
        # grid->mesh: prepared edge terms + sender latents in, mesh aggregate out; mesh-node update (v_m0, agg in, v_m out); grid-node update in place
        "encoder": e_g2m * erow + grid + mesh + 3 * mesh + 2 * grid,
        # per layer: edge latents read + written (residual update), node latents read + written, aggregate out + in
        "processor": cfg.steps * (2 * e_mesh * erow + 4 * mesh),
        # mesh->grid: prepared edge terms + mesh and grid latents in, grid aggregate out; grid-node update (v_g, agg in, v_g out)
        "decoder": e_m2g * erow + mesh + grid + grid + 3 * grid,
        "output": grid + 2 * n_grid * V * 4,                                           # latents in, x(t) in, x(t + 6 h) out
    }
    st["total"] = float(sum(st.values()))
    return {k: float(v) for k, v in st.items()}


def flops_per_step_executed(cfg: GraphcastConfig, n_grid: int, n_mesh: int, e_mesh: int, e_g2m: int, e_m2g: int) -> float:
    """FLOPs the engine actually executes for the same result (``flops_per_stage(..., executed=True)``)."""
    return sum(flops_per_stage(cfg, n_grid, n_mesh, e_mesh, e_g2m, e_m2g, executed=True).values())
