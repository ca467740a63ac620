"""``run_basic_inference`` -- the reference's hot loop (/root/reference/skyrim/core/models/utils.py:10-49),
driving a TimeLoop and labelling its output."""
from __future__ import annotations

import os
from datetime import datetime
from typing import Any

import numpy as np
import torch

from ...datasource import get_initial_condition_for_model
from ...labeled import DataArray, open_dataarray


_PINNED_LIMIT = 24 << 30     # bytes of page-locked host memory run_basic_inference may take for its result


class ResidentState:
    """The device copy of the state(s) a TimeLoop produced last, tied to the host array they were delivered in.

    The reference feeds every prediction back through the host: ``rollout`` hands ``pred`` to ``predict_one_step``, which uploads
    ``pred.values[-n_history_levels:]`` again (/root/reference/skyrim/core/models/base.py:131-132, utils.py:24-31: 286 MB from pageable
    memory per step).  When the array that comes back IS the one this model delivered -- same ndarray object, still read-only -- the
    states are already in HBM and the upload is skipped.  The delivered array is marked read-only for exactly that reason: an in-place
    edit would go unnoticed; ``DataArray.copy()`` (or ``perturb_initial_conditions``, which copies on demand) gives a writable array, and
    a copy takes the ordinary upload path."""

    def __init__(self, host: np.ndarray, states: list, image=None):
        import weakref
        host.flags.writeable = False
        self._host = weakref.ref(host)
        self.states = states                       # the last n_history_levels yielded tensors, (B, C, lat, lon) each, oldest first
        # the newest state's bytes in file order, if the prediction was delivered with its big-endian image (deliver.py): the next step of
        # a saving rollout starts its own image with them instead of copying the state it starts from to the host a second time
        self.image_tail = image.tail(1) if image is not None else None

    def tensor_for(self, x, n_hist: int):
        """(1, n_hist, C, lat, lon) device tensor if ``x`` carries the array these states were delivered in, else None."""
        vals = getattr(x, "_values", None)         # (the array object, not its contents: a DataArray's ``values`` would wait for the copy)
        if vals is None:
            vals = getattr(x, "values", None)
        if vals is None or vals is not self._host() or vals.flags.writeable or len(self.states) < n_hist:
            return None
        return torch.stack([t[0] if t.dim() == 4 else t for t in self.states[-n_hist:]], dim=0).unsqueeze(0)


def run_basic_inference(model, n: int, data_source: Any, time: datetime, x=None, deliver: str | None = None, defer_check: bool = False):
    """Run a basic inference: returns DataArray(time = n + 1, channel, lat, lon); entry 0 is the state at ``time``.
    ``deliver`` (skyrim_amd/deliver.py; ``rollout`` sets it for the steps of a saving rollout): "be" = bring only the big-endian image
    of the states to the host (the file's bytes; native ``values`` are filled from it on first read), "both" = native and image, "skip" =
    nothing at all (an intermediate step of a rollout that saves nothing: the next step reads the state from HBM, nobody reads ``values``).
    ``defer_check`` (``rollout``): the loop's pending non-finite check of the LAST state is not waited for here; its flag travels to the
    host with the state and whoever reads the numbers -- ``values``, the netCDF writer -- gets the FloatingPointError instead."""
    counters = model.__dict__.setdefault("io_counters", {"state_uploads": 0, "resident_hits": 0}) if hasattr(model, "__dict__") else {}
    resident, head = getattr(model, "_resident_state", None), None
    if hasattr(model, "__dict__"):
        model._resident_state = None               # a run that raises must not leave a stale entry armed (it is set again on success)
    if x is None:
        x = get_initial_condition_for_model(model, data_source, time)     # comes with the batch dimension
        counters["state_uploads"] = counters.get("state_uploads", 0) + 1
    else:
        if isinstance(x, (str, os.PathLike)):
            x = open_dataarray(os.fspath(x))
        dev = resident.tensor_for(x, model.n_history_levels) if resident is not None else None
        if dev is not None:
            x = dev                                                        # fed straight back: the states never left HBM
            counters["resident_hits"] = counters.get("resident_hits", 0) + 1
            head = resident.image_tail if deliver in ("be", "both") else None
            if hasattr(model, "__dict__"):
                model._state_is_own_output = True                          # (a TimeLoop that range-checks initial conditions skips its own output)
        else:
            x = torch.as_tensor(np.asarray(x.values[-model.n_history_levels:]), dtype=torch.float32).to(model.device)
            x = x.unsqueeze(0)
            counters["state_uploads"] = counters.get("state_uploads", 0) + 1

    # The reference copies every yielded state to the host synchronously (`.cpu().numpy()`, utils.py:36): 286 MB per
    # step through pageable memory, the sync point of its loop.  Here the n + 1 states land in ONE pinned host buffer
    # through a copy stream, so the D2H of step k overlaps the forward of step k + 1; the result is the same array.
    loop = model(time, x)
    try:
        return _drain(model, loop, n, time, deliver, head, defer_check)
    finally:
        if hasattr(loop, "close"):
            loop.close()        # lets the TimeLoop flush its deferred checks (FiniteGuard: the last yielded state) -- may raise; also runs
                                # when the loop body itself raised, so that a generator is never left to the garbage collector


def _drain(model, loop, n: int, time, deliver=None, head=None, defer_check=False):
    """``head`` (deliver.ImagePart): the big-endian bytes of the state the loop starts from, already on the host (the previous prediction's
    image) -- the loop's first yield echoes that state, so its image is borrowed instead of swapped and copied again."""
    from ... import deliver as D
    times, stacked, arrays, side, last, be, k0 = [], None, [], None, [], None, 0
    for k, (time, output, _) in enumerate(loop):
        out = output.squeeze(0) if output.dim() == 4 and output.shape[0] == 1 else output
        if out.is_cuda:
            if stacked is None:
                pin = (n + 1) * out.numel() * 4 <= _PINNED_LIMIT
                shape = (n + 1,) + tuple(out.shape)
                side = torch.cuda.Stream(out.device)
                if deliver == "skip":
                    pass                                               # an intermediate state nobody will look at: nothing crosses to the host
                elif deliver in ("be", "both") and pin and out.dtype == torch.float32 and D.enabled():
                    D.load_library()                                   # (a missing library is an error here, not a silent host swap)
                    k0 = 1 if head is not None and n >= 1 and head.array.shape == (1,) + tuple(out.shape) else 0
                    be = torch.empty((n + 1 - k0,) + tuple(out.shape), dtype=torch.float32, pin_memory=True)
                else:
                    deliver = None
                # "be": no copy lands in the native array -- plain pageable memory that stays untouched (no page behind it) unless
                # somebody reads ``values``; one pinned block per prediction in flight, as without the image
                stacked = torch.from_numpy(np.empty(shape, dtype=np.float32)) if deliver in ("be", "skip") else torch.empty(shape, dtype=torch.float32, pin_memory=pin)
            side.wait_stream(torch.cuda.current_stream(out.device))
            with torch.cuda.stream(side):
                if deliver not in ("be", "skip"):
                    stacked[k].copy_(out, non_blocking=True)
                if be is not None and k >= k0:
                    # the bytes of the file: swapped in HBM on the copy stream (0.1 ms beside the next step's kernels), then the same
                    # device-to-host copy as the native one
                    src = out.contiguous()
                    swapped = torch.empty_like(src)
                    D.bswap32(src, swapped, side)
                    be[k - k0].copy_(swapped, non_blocking=True)
                    del src, swapped
            out.record_stream(side)
            last = (last + [output if output.dim() == 4 else output.unsqueeze(0)])[-model.n_history_levels:]
        else:
            arrays.append(out.detach().cpu().numpy())
        times.append(time)
        if k == n:
            break
    verdict = None
    if defer_check and stacked is not None and deliver == "skip" and hasattr(model, "take_pending_check"):
        model.take_pending_check()      # this state reaches nobody: the check of the rollout's last step answers for it (non-finite values never leave the network again)
    elif defer_check and stacked is not None and (be is not None or stacked.is_pinned()) and hasattr(model, "take_pending_check"):
        pend = model.take_pending_check()
        if pend is not None:    # the flag goes to the host behind the states, on the same stream: read after the same event
            flag, step, hint = pend
            host_flag = torch.empty((), dtype=torch.bool, pin_memory=True)
            side.wait_stream(torch.cuda.current_stream(flag.device))
            with torch.cuda.stream(side):
                host_flag.copy_(flag, non_blocking=True)
            flag.record_stream(side)

            def verdict(host_flag=host_flag, step=step, hint=hint):
                if not bool(host_flag.item()):
                    raise FloatingPointError(f"non-finite values in the state after step {step}: {hint}")
    if hasattr(loop, "close"):
        loop.close()            # flush BEFORE the result is built: a non-finite last state must not be delivered (deferred: nothing pending)
    ready = image = None
    if stacked is not None:
        # The last state's copy is still in flight on the side stream.  The array is handed over now and the DataArray waits for the
        # copy's event the first time its numbers are read (labeled.DataArray.values): ``rollout`` feeds the prediction straight back
        # (ResidentState: no read), so the 286 MB copy of step k runs under step k + 1 instead of between the two; the save thread and
        # any other reader wait first.  Pageable results (beyond _PINNED_LIMIT) were copied synchronously by torch: nothing to wait for.
        done = torch.cuda.Event()
        done.record(side)
        if verdict is None:
            landed = done.synchronize
        else:
            def landed(ev=done, verdict=verdict):     # every reader of the numbers -- values, the image's bytes -- passes here first
                ev.synchronize()
                verdict()
        pinned = stacked.is_pinned()
        keep = stacked                              # the tensor owns the pinned block: it must outlive the numpy view
        stacked = stacked[:len(times)].numpy()
        if be is not None:
            image = D.BigEndianImage(be[:len(times) - k0].numpy().view(">f4"), of=stacked, wait=landed, keep=be, head=[head] if k0 else ())
        if deliver == "skip":
            def ready():
                raise RuntimeError("this state was produced inside a rollout that saves nothing and hands only its last prediction out: it "
                                   "never left the GPU (use predict_one_step / forecast to look at every step)")
        elif deliver == "be":
            # nothing was copied into ``stacked``: whoever reads ``values`` gets them from the image, swapped back on the host (a second,
            # writable view of the block -- ``stacked`` itself goes read-only in ResidentState)
            ready = lambda img=image, dst=keep[:len(times)].numpy(), _keep=keep: img.fill_native(dst)    # noqa: E731
        elif pinned:
            ready = lambda _keep=keep: landed()    # noqa: E731
        else:
            done.synchronize()
        if hasattr(model, "__dict__"):
            model._resident_state = ResidentState(stacked, last, image)
    else:
        stacked = np.stack(arrays)
    coords = dict(time=times, channel=model.out_channel_names, lat=np.asarray(model.grid.lat), lon=np.asarray(model.grid.lon))
    return DataArray(stacked, dims=["time", "channel", "lat", "lon"], coords=coords, ready=ready, image=image)


def estimate_pressure_hpa(elevation_m):
    """Standard-atmosphere pressure in hPa at ``elevation_m`` metres (reference utils.py:52-67): the barometric formula of the troposphere,
    p = p0 (1 - L h / T0)^(g M / (R L)) with the ISA constants."""
    p0, lapse, t0 = 101325.0, 0.0065, 288.15          # Pa, K / m, K
    g, molar, gas = 9.80665, 0.0289644, 8.31447       # m / s^2, kg / mol, J / (mol K)
    return p0 * (1.0 - lapse * elevation_m / t0) ** (g * molar / (gas * lapse)) / 100.0


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
    if not initial_conditions.values.flags.writeable:      # a prediction as delivered by run_basic_inference (ResidentState): edit a copy
        initial_conditions.values = initial_conditions.values.copy()
    initial_conditions.values[tuple(sl)] = value
    return initial_conditions
