"""``forecast`` command line -- /root/reference/skyrim/forecast.py:59-147 (same options; entry point
``forecast = skyrim.forecast:main`` in the reference's pyproject.toml:45-46).  The Modal flag of the reference
(--modal, a hosted NVIDIA service) is accepted and rejected with a clear message."""
from __future__ import annotations

from datetime import datetime, timedelta
from pathlib import Path

import click

from .common import AVAILABLE_MODELS

yesterday = (datetime.now() - timedelta(days=1)).date().isoformat().replace("-", "")


def run_forecast(model_name: str, date: str, time: str, lead_time: int, list_models: bool, initial_conditions: str,
                 output_dir: str, filter_vars: str):
    from .core import Skyrim
    if list_models:
        print("Available models:", Skyrim.list_available_models())
        return []
    model = Skyrim(model_name, ic_source=initial_conditions)
    pred, output_paths = model.predict(
        date=date, time=time, lead_time=lead_time, save=True,
        save_config={"output_dir": output_dir or str(Path.cwd() / "outputs"),
                     "filter_vars": (filter_vars.split(",") if bool(filter_vars) else [])})
    return output_paths


@click.command()
@click.option("--model_name", "-m", type=click.Choice(AVAILABLE_MODELS, case_sensitive=False), default="pangu", help="Select model")
@click.option("--date", "-d", type=str, default=yesterday, help="YYYYMMDD")
@click.option("--time", "-t", type=str, default="0000", help="HHMM")
@click.option("--lead_time", "-l", type=int, default=6, help="Lead time in hours, int 0-24")
@click.option("--list_models", "-lm", is_flag=True, help="List all available models and exit")
@click.option("--initial_conditions", "-ic", type=click.Choice(["cds", "ifs", "gfs"], case_sensitive=False), default="gfs",
              help="Initial conditions provider.")
@click.option("--output_dir", "-o", type=str, default="", help="Output directory (local path)")
@click.option("--filter_vars", "-f", type=str, default="", help="Filter variables such as t2m (temperature) before saving forecasts.")
@click.option("--modal", "-mo", is_flag=True, help="(reference only) run on Modal -- not available in this build")
def main(model_name, date, time, lead_time, list_models, initial_conditions, output_dir, filter_vars, modal):
    if modal:
        raise click.UsageError("--modal runs the reference on a hosted A100 service; this build runs on the local MI355X")
    paths = run_forecast(model_name, date, time, lead_time, list_models, initial_conditions, output_dir, filter_vars)
    for p in paths:
        click.echo(p)
    return paths


if __name__ == "__main__":
    main()
