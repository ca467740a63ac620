"""``run_basic_inference`` -- the reference's hot loop (/root/reference/skyrim/core/models/utils.py:10-49),
driving a TimeLoop and labelling its output."""
from __future__ import annotations

from datetime import datetime
from typing import Any

import numpy as np
import torch

from ...datasource import get_initial_condition_for_model
from ...labeled import DataArray, open_dataarray


def run_basic_inference(model, n: int, data_source: Any, time: datetime, x=None):
    """Run a basic inference: returns DataArray(time = n + 1, channel, lat, lon); entry 0 is the state at ``time``."""
    if x is None:
        x = get_initial_condition_for_model(model, data_source, time)     # comes with the batch dimension
    else:
        if isinstance(x, str):
            x = open_dataarray(x)
        x = torch.as_tensor(np.asarray(x.values[-model.n_history_levels:]), dtype=torch.float32).to(model.device)
        x = x.unsqueeze(0)

    arrays, times = [], []
    for k, (time, output, _) in enumerate(model(time, x)):
        # output: (B, len(out_channel_names), len(lat), len(lon)); the D2H copy is the sync point
        arrays.append(output.cpu().numpy().squeeze())
        times.append(time)
        if k == n:
            break

    stacked = np.stack(arrays)
    coords = dict(time=times, channel=model.out_channel_names, lat=np.asarray(model.grid.lat), lon=np.asarray(model.grid.lon))
    return DataArray(stacked, dims=["time", "channel", "lat", "lon"], coords=coords)


def perturb_initial_conditions(initial_conditions: DataArray, channel, lat, lon, value):
    """Set one (channel, nearest lat/lon) cell (reference utils.py:70-92); lon < 0 wraps by +360."""
    if lon < 0:
        lon += 360
    lats, lons = initial_conditions._coords["lat"], initial_conditions._coords["lon"]
    i, j = int(np.abs(lats - lat).argmin()), int(np.abs(lons - lon).argmin())
    c = initial_conditions._coords["channel"].tolist().index(channel)
    ax = initial_conditions.dims
    sl = [slice(None)] * len(ax)
    sl[ax.index("channel")], sl[ax.index("lat")], sl[ax.index("lon")] = c, i, j
    initial_conditions.values[tuple(sl)] = value
    return initial_conditions
