"""FourCastNet v2-small checkpoint -> the engine's parameter slots.

The reference obtains the weights through ``earth2mip.networks.fcnv2_sm.load(registry.get_model("e2mip://fcnv2_sm"))``
(/root/reference/skyrim/core/models/fourcastnet_v2.py:36-37): a ``weights.tar`` holding ``{"model_state": state_dict}`` of modulus'
legacy ``SphericalFourierNeuralOperatorNet`` (keys prefixed ``module.`` by DistributedDataParallel), plus ``global_means.npy`` /
``global_stds.npy`` of shape (1, 73, 1, 1).  Neither the package nor the file exists in this environment, so the key names below are
the public module structure of sfnonet.py as published (encoder / decoder = Sequential(conv1x1, act, conv1x1), blocks[i] =
{norm0, filter.filter (dhconv weight as [in, out, l, 2]), inner_skip, norm1, mlp.fwd = Sequential(conv1x1, act, conv1x1)},
``pos_embed``) -- UNVERIFIED against the real file; ``convert`` therefore refuses to return a partial or shape-mismatched set and
reports every key it could not place, so that a naming difference surfaces as an error, never as a silently random layer.
"""
from __future__ import annotations

import re

import numpy as np
import torch

from .spec import SfnoConfig, param_spec

# checkpoint key (after stripping "module." / "model.") -> slot; {i} = block index
RULES = [
    (r"encoder\.0\.weight", "encoder.fc1.weight"), (r"encoder\.0\.bias", "encoder.fc1.bias"), (r"encoder\.2\.weight", "encoder.fc2.weight"),
    (r"pos_embed", "pos_embed"),
    (r"blocks\.(\d+)\.norm0\.weight", "blocks.{0}.norm0.weight"), (r"blocks\.(\d+)\.norm0\.bias", "blocks.{0}.norm0.bias"),
    (r"blocks\.(\d+)\.filter\.filter\.weight", "blocks.{0}.filter.weight"), (r"blocks\.(\d+)\.filter\.weight", "blocks.{0}.filter.weight"),
    (r"blocks\.(\d+)\.inner_skip\.weight", "blocks.{0}.inner_skip.weight"), (r"blocks\.(\d+)\.inner_skip\.bias", "blocks.{0}.inner_skip.bias"),
    (r"blocks\.(\d+)\.norm1\.weight", "blocks.{0}.norm1.weight"), (r"blocks\.(\d+)\.norm1\.bias", "blocks.{0}.norm1.bias"),
    (r"blocks\.(\d+)\.mlp\.fwd\.0\.weight", "blocks.{0}.mlp.fc1.weight"), (r"blocks\.(\d+)\.mlp\.fwd\.0\.bias", "blocks.{0}.mlp.fc1.bias"),
    (r"blocks\.(\d+)\.mlp\.fwd\.2\.weight", "blocks.{0}.mlp.fc2.weight"), (r"blocks\.(\d+)\.mlp\.fwd\.2\.bias", "blocks.{0}.mlp.fc2.bias"),
    (r"decoder\.0\.weight", "decoder.fc1.weight"), (r"decoder\.0\.bias", "decoder.fc1.bias"), (r"decoder\.2\.weight", "decoder.fc2.weight"),
]


def slot_of(key: str) -> str | None:
    key = re.sub(r"^(module\.|model\.)+", "", key)
    for pat, slot in RULES:
        m = re.fullmatch(pat, key)
        if m:
            return slot.format(*m.groups())
    return None


def convert(state_dict: dict, cfg: SfnoConfig, means, stds) -> dict:
    """``state_dict``: the checkpoint's ``model_state`` (tensors or arrays); ``means`` / ``stds``: the (1, C, 1, 1) normalisation arrays.
    -> {slot: float32 tensor} for every slot of ``param_spec(cfg)``; 1x1-convolution weights lose their trailing (1, 1), complex
    dhconv weights stored as complex tensors are viewed as (..., 2)."""
    want = dict(param_spec(cfg))
    out, unplaced = {}, []
    for key, val in state_dict.items():
        slot = slot_of(key)
        if slot is None:
            unplaced.append(key)
            continue
        t = torch.as_tensor(np.asarray(val)) if not torch.is_tensor(val) else val
        if t.is_complex():
            t = torch.view_as_real(t)
        t = t.float()
        shape = want.get(slot)
        if shape is None:
            unplaced.append(key)
            continue
        if t.dim() == len(shape) + 2 and tuple(t.shape[-2:]) == (1, 1):
            t = t[..., 0, 0]                                   # conv1x1 [out, in, 1, 1] -> [out, in]
        if slot == "pos_embed" and t.dim() == 4:
            t = t[0]
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"{key} -> {slot}: checkpoint shape {tuple(t.shape)}, slot wants {tuple(shape)} (different hyper-parameters? "
                             f"embed_dim / num_layers / scale_factor are configuration: SfnoConfig)")
        out[slot] = t.contiguous()
    out["norm.mean"] = torch.as_tensor(np.asarray(means), dtype=torch.float32).reshape(-1)
    out["norm.std"] = torch.as_tensor(np.asarray(stds), dtype=torch.float32).reshape(-1)
    missing = [s for s in want if s not in out]
    if missing or unplaced:
        raise ValueError(f"checkpoint does not match the slot table: {len(missing)} slots unfilled (first: {missing[:5]}), "
                         f"{len(unplaced)} keys unplaced (first: {unplaced[:5]})")
    for s, shape in want.items():
        if tuple(out[s].shape) != tuple(shape):
            raise ValueError(f"{s}: {tuple(out[s].shape)} != {tuple(shape)}")
    return out


def load(path, cfg: SfnoConfig, means_path, stds_path) -> dict:
    ck = torch.load(path, map_location="cpu")
    sd = ck.get("model_state", ck) if isinstance(ck, dict) else ck
    return convert(sd, cfg, np.load(means_path), np.load(stds_path))
