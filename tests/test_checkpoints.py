"""CPU tests of the checkpoint key mappers: a state dict written in the PUBLISHED naming of the source packages maps
onto every parameter slot exactly once, with the layout changes each format needs; anything partial or mis-shaped is an error."""
import numpy as np
import pytest
import torch

from skyrim_amd.graphcast import checkpoint as GC
from skyrim_amd.graphcast.spec import GraphcastConfig
from skyrim_amd.graphcast.spec import init_synthetic as gc_init
from skyrim_amd.graphcast.spec import param_spec as gc_spec
from skyrim_amd.sfno import checkpoint as SC
from skyrim_amd.sfno.spec import SfnoConfig
from skyrim_amd.sfno.spec import init_synthetic as sfno_init


def test_sfno_state_dict_in_modulus_naming_round_trips():
    cfg = SfnoConfig(n_lat=33, n_lon=64, in_chans=5, out_chans=5, embed_dim=16, num_layers=3, scale_factor=2)
    p = sfno_init(cfg, 0)
    sd = {"module.encoder.0.weight": p["encoder.fc1.weight"][:, :, None, None], "module.encoder.0.bias": p["encoder.fc1.bias"],
          "module.encoder.2.weight": p["encoder.fc2.weight"][:, :, None, None], "module.pos_embed": p["pos_embed"][None],
          "module.decoder.0.weight": p["decoder.fc1.weight"][:, :, None, None], "module.decoder.0.bias": p["decoder.fc1.bias"],
          "module.decoder.2.weight": p["decoder.fc2.weight"][:, :, None, None]}
    for i in range(cfg.num_layers):
        b = f"module.blocks.{i}."
        sd.update({b + "norm0.weight": p[f"blocks.{i}.norm0.weight"], b + "norm0.bias": p[f"blocks.{i}.norm0.bias"],
                   b + "filter.filter.weight": torch.view_as_complex(p[f"blocks.{i}.filter.weight"].contiguous()),
                   b + "inner_skip.weight": p[f"blocks.{i}.inner_skip.weight"][:, :, None, None], b + "inner_skip.bias": p[f"blocks.{i}.inner_skip.bias"],
                   b + "norm1.weight": p[f"blocks.{i}.norm1.weight"], b + "norm1.bias": p[f"blocks.{i}.norm1.bias"],
                   b + "mlp.fwd.0.weight": p[f"blocks.{i}.mlp.fc1.weight"][:, :, None, None], b + "mlp.fwd.0.bias": p[f"blocks.{i}.mlp.fc1.bias"],
                   b + "mlp.fwd.2.weight": p[f"blocks.{i}.mlp.fc2.weight"][:, :, None, None], b + "mlp.fwd.2.bias": p[f"blocks.{i}.mlp.fc2.bias"]})
    got = SC.convert(sd, cfg, p["norm.mean"].reshape(1, -1, 1, 1).numpy(), p["norm.std"].reshape(1, -1, 1, 1).numpy())
    assert set(got) == set(p) and all(torch.equal(got[k], p[k]) for k in p)
    with pytest.raises(ValueError, match="unplaced"):
        SC.convert(dict(sd, **{"module.blocks.0.outer_skip.weight": torch.zeros(2)}), cfg, p["norm.mean"], p["norm.std"])
    with pytest.raises(ValueError, match="unfilled"):
        SC.convert({k: v for k, v in sd.items() if "norm1.bias" not in k}, cfg, p["norm.mean"], p["norm.std"])
    with pytest.raises(ValueError, match="hyper-parameters"):
        SC.convert(sd, SfnoConfig(n_lat=33, n_lon=64, in_chans=5, out_chans=5, embed_dim=24, num_layers=3, scale_factor=2), p["norm.mean"], p["norm.std"])


def test_graphcast_haiku_params_round_trip():
    cfg = GraphcastConfig(n_lat=9, n_lon=16, splits=1, latent=16, steps=2, n_vars=5)
    p = gc_init(cfg, 0)
    hk = {}
    for mlp in sorted({s.rsplit(".", 2)[0] for s, _ in gc_spec(cfg) if s.endswith(".fc1.weight")}):
        for part, key in GC.haiku_keys(mlp).items():
            slot = f"{mlp}.{part}"
            if slot in p:
                hk[key] = (p[slot].T if (part.endswith("weight") and p[slot].dim() == 2) else p[slot]).numpy()
    # deepmind's mesh-node embedder sees [zeros(grid feature width) | 3 structural features]; mesh2grid also updates its mesh nodes
    hk[GC.haiku_keys("embed.mesh")["fc1.weight"]] = np.concatenate([np.ones((cfg.grid_in - 3, cfg.latent), np.float32), p["embed.mesh.fc1.weight"].T.numpy()])
    for part in ("linear_0/w", "linear_0/b", "linear_1/w", "linear_1/b"):
        hk[f"mesh2grid_gnn/~_networks_builder/processor_nodes_0_mesh_nodes_mlp/~/{part}"] = np.zeros((2, 2), np.float32)
    assert GC.haiku_keys("proc.1.edge")["fc1.weight"] == "mesh_gnn/~_networks_builder/processor_edges_1_mesh_mlp/~/linear_0/w"
    kw = dict(mean=p["norm.mean"], std=p["norm.std"], diff_std=p["norm.diff_std"], static=p["static"])
    got = GC.convert(hk, cfg, **kw)
    assert set(got) == set(p) and all(torch.equal(got[k], p[k]) for k in p)
    perm = np.arange(cfg.grid_in)[::-1].copy()
    assert torch.equal(GC.convert(hk, cfg, in_perm=perm, **kw)["embed.grid.fc1.weight"], p["embed.grid.fc1.weight"][:, torch.from_numpy(perm)])
    with pytest.raises(KeyError):
        GC.convert({k: v for k, v in hk.items() if "processor_edges_1_mesh_layer_norm/scale" not in k}, cfg, **kw)
    with pytest.raises(ValueError, match="unplaced"):
        GC.convert(dict(hk, **{"mesh_gnn/~_networks_builder/surprise/w": np.zeros(3, np.float32)}), cfg, **kw)


def test_graphcast_npz_file_in_the_colon_key_form_of_checkpoint_dump(tmp_path):
    """``GC.load`` on a file keyed ``params:<module path>:<name>`` (the ':'-flattened form of deepmind's checkpoint.dump) and on the
    '/'-joined form: both must reach the same slots (a loader that only strips 'params:' misses every key of the real file)."""
    cfg = GraphcastConfig(n_lat=9, n_lon=16, splits=1, latent=16, steps=2, n_vars=5)
    p = gc_init(cfg, 0)
    hk = {}
    for mlp in sorted({s.rsplit(".", 2)[0] for s, _ in gc_spec(cfg) if s.endswith(".fc1.weight")}):
        for part, key in GC.haiku_keys(mlp).items():
            slot = f"{mlp}.{part}"
            if slot in p:
                hk[key] = (p[slot].T if (part.endswith("weight") and p[slot].dim() == 2) else p[slot]).numpy()
    hk[GC.haiku_keys("embed.mesh")["fc1.weight"]] = np.concatenate([np.zeros((cfg.grid_in - 3, cfg.latent), np.float32), p["embed.mesh.fc1.weight"].T.numpy()])
    kw = dict(mean=p["norm.mean"], std=p["norm.std"], diff_std=p["norm.diff_std"], static=p["static"])
    colon = {"params:" + ":".join(k.rsplit("/", 1)): v for k, v in hk.items()}
    assert "params:mesh_gnn/~_networks_builder/processor_edges_1_mesh_mlp/~/linear_0:w" in colon
    assert GC.normalise_key("params:a/~/linear_0:w") == "a/~/linear_0/w" and GC.normalise_key("a/~/linear_0/w") == "a/~/linear_0/w"
    for name, keys in (("colon.npz", colon), ("slash.npz", {"params:" + k: v for k, v in hk.items()})):
        np.savez(tmp_path / name, model_config=np.zeros(1), **keys)
        got = GC.load(tmp_path / name, cfg, **kw)
        assert set(got) == set(p) and all(torch.equal(got[k], p[k]) for k in p), name
