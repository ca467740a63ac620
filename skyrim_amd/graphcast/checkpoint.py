"""DeepMind GraphCast parameters -> the engine's parameter slots.

The reference loads ``e2mip://graphcast`` (/root/reference/skyrim/core/models/graphcast.py:51-54): earth2mip wraps deepmind/graphcast,
whose checkpoint is an ``.npz`` of haiku parameters ``{"<module path>/<name>": array}`` -- three ``DeepTypedGraphNet``s (grid2mesh_gnn,
mesh_gnn, mesh2grid_gnn), every MLP a ``<...>_mlp/~/linear_0``, ``linear_1`` pair (``w`` stored [in, out], ``b``) followed by
``<...>_layer_norm`` (``scale``, ``offset``).  Not available in this environment: the module paths below are the public structure of
graphcast.py / deep_typed_graph_net.py as published, UNVERIFIED against a real file.  ``convert`` refuses anything partial.

What differs from the engine's slots, and is handled here:
  * haiku Linear weights are [in, out] -> transposed;
  * the mesh-node embedder of grid2mesh sees [zeros(grid feature width) | 3 structural features]: only the last 3 input rows matter;
  * mesh2grid also updates its mesh nodes (processor_nodes_0_mesh_nodes_mlp): computed by nothing downstream, dropped;
  * the order of the per-grid-node input features and of the output variables is fixed by deepmind's dataset stacking, not by the
    network: ``in_perm`` / ``out_perm`` (index arrays into the checkpoint's order) reorder them to the reference's CHANNELS order --
    identity if omitted, which is only right for weights trained in this order.
"""
from __future__ import annotations

import numpy as np
import torch

from .spec import GraphcastConfig, param_spec

_NB = "~_networks_builder"


def module_of(slot: str) -> tuple[str, str] | None:
    """(haiku MLP module prefix, layer-norm module) of an engine MLP name."""
    table = {
        "embed.grid": ("grid2mesh_gnn", "encoder_nodes_grid_nodes"), "embed.mesh": ("grid2mesh_gnn", "encoder_nodes_mesh_nodes"),
        "embed.g2m_edge": ("grid2mesh_gnn", "encoder_edges_grid2mesh"), "embed.mesh_edge": ("mesh_gnn", "encoder_edges_mesh"),
        "embed.m2g_edge": ("mesh2grid_gnn", "encoder_edges_mesh2grid"),
        "g2m.edge": ("grid2mesh_gnn", "processor_edges_0_grid2mesh"), "g2m.mesh_node": ("grid2mesh_gnn", "processor_nodes_0_mesh_nodes"),
        "g2m.grid_node": ("grid2mesh_gnn", "processor_nodes_0_grid_nodes"),
        "m2g.edge": ("mesh2grid_gnn", "processor_edges_0_mesh2grid"), "m2g.grid_node": ("mesh2grid_gnn", "processor_nodes_0_grid_nodes"),
        "out": ("mesh2grid_gnn", "decoder_nodes_grid_nodes"),
    }
    if slot in table:
        return table[slot]
    if slot.startswith("proc."):
        _, i, kind = slot.split(".")
        return ("mesh_gnn", f"processor_edges_{i}_mesh" if kind == "edge" else f"processor_nodes_{i}_mesh_nodes")
    return None


def haiku_keys(mlp: str) -> dict:
    gnn, stem = module_of(mlp)
    base = f"{gnn}/{_NB}/{stem}"
    return {"fc1.weight": f"{base}_mlp/~/linear_0/w", "fc1.bias": f"{base}_mlp/~/linear_0/b", "fc2.weight": f"{base}_mlp/~/linear_1/w",
            "fc2.bias": f"{base}_mlp/~/linear_1/b", "ln.weight": f"{base}_layer_norm/scale", "ln.bias": f"{base}_layer_norm/offset"}


def convert(params: dict, cfg: GraphcastConfig, mean, std, diff_std, static, in_perm=None, out_perm=None) -> dict:
    """``params``: flat {haiku path: array}.  mean / std / diff_std: per-variable normalisation of the state and of the predicted
    increment in the engine's channel order; ``static``: (2, n_lat, n_lon) surface geopotential and land-sea mask, normalised."""
    want = dict(param_spec(cfg))
    out = {"norm.mean": torch.as_tensor(np.asarray(mean), dtype=torch.float32), "norm.std": torch.as_tensor(np.asarray(std), dtype=torch.float32),
           "norm.diff_std": torch.as_tensor(np.asarray(diff_std), dtype=torch.float32), "static": torch.as_tensor(np.asarray(static), dtype=torch.float32)}
    used = set()
    mlps = sorted({s.rsplit(".", 2)[0] for s in want if s.endswith(".fc1.weight")})
    for mlp in mlps:
        for part, key in haiku_keys(mlp).items():
            slot = f"{mlp}.{part}"
            if slot not in want:
                continue                                 # the output MLP has no LayerNorm
            if key not in params:
                raise KeyError(f"{slot}: checkpoint has no {key!r}")
            a = torch.as_tensor(np.asarray(params[key]), dtype=torch.float32)
            used.add(key)
            if part.endswith("weight") and a.dim() == 2:
                a = a.T                                  # haiku [in, out] -> [out, in]
            if slot == "embed.mesh.fc1.weight" and a.shape[1] != want[slot][1]:
                a = a[:, -want[slot][1]:]                # [zeros | structural]: the zero-padded input rows carry no signal
            if slot == "embed.grid.fc1.weight" and in_perm is not None:
                a = a[:, torch.as_tensor(np.asarray(in_perm), dtype=torch.long)]
            if mlp == "out" and part.startswith("fc2") and out_perm is not None:
                a = a[torch.as_tensor(np.asarray(out_perm), dtype=torch.long)]
            if tuple(a.shape) != tuple(want[slot]):
                raise ValueError(f"{key} -> {slot}: {tuple(a.shape)}, slot wants {tuple(want[slot])}")
            out[slot] = a.contiguous()
    missing = [s for s in want if s not in out]
    dropped = [k for k in params if k not in used]
    allowed = [k for k in dropped if "processor_nodes_0_mesh_nodes" in k and k.startswith("mesh2grid_gnn")]
    if missing or len(allowed) != len(dropped):
        extra = [k for k in dropped if k not in allowed]
        raise ValueError(f"checkpoint does not match the slot table: {len(missing)} slots unfilled (first: {missing[:4]}), "
                         f"{len(extra)} parameters unplaced (first: {extra[:4]})")
    return out


def normalise_key(k: str) -> str:
    """deepmind's ``checkpoint.dump`` flattens the nested haiku dict with ':' between the levels -- ``params:<module path>:<name>``
    (e.g. ``params:mesh_gnn/~_networks_builder/encoder_edges_mesh_mlp/~/linear_0:w``) -- while a plain ``hk.data_structures`` flattening
    joins module and parameter with '/'.  Both forms are accepted: the leading ``params:`` goes, the LAST ':' becomes '/'."""
    k = k.replace("params:", "", 1)
    head, sep, tail = k.rpartition(":")
    return f"{head}/{tail}" if sep else k


def load(npz_path, cfg: GraphcastConfig, **kw) -> dict:
    z = np.load(npz_path, allow_pickle=False)
    params = {normalise_key(k): z[k] for k in z.files if not k.startswith(("model_config", "task_config", "description", "license"))}
    return convert(params, cfg, **kw)
