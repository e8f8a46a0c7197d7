"""TOML configuration front -- mirror of ``rscm.config`` for the two-layer model
(python/rscm/config/loader.py:27-128, python/rscm/config/builder.py:19-108)."""
from __future__ import annotations

import logging
from pathlib import Path
from typing import Any, Dict, Union

import numpy as np

try:  # Python >= 3.11
    import tomllib as _toml
except ModuleNotFoundError:  # the image's 3.10
    import tomli as _toml

from .core import ModelBuilder, TimeAxis
from .two_layer import TwoLayerBuilder

logger = logging.getLogger(__name__)

KNOWN_TOP_LEVEL = {"schema", "time", "components", "inputs", "outputs", "model", "initial_values"}


def deep_merge(base: Dict[str, Any], override: Dict[str, Any]) -> Dict[str, Any]:
    """Nested dicts merge recursively; lists and scalars are replaced (loader.py:27-60)."""
    result = base.copy()
    for key, value in override.items():
        if key in result and isinstance(result[key], dict) and isinstance(value, dict):
            result[key] = deep_merge(result[key], value)
        else:
            result[key] = value
    return result


def load_config(path: Union[str, Path]) -> Dict[str, Any]:
    with Path(path).open("rb") as f:
        config = _toml.load(f)
    unknown = sorted(set(config) - KNOWN_TOP_LEVEL)
    if unknown:
        logger.warning("Unknown configuration keys in %s: %s. These will be ignored.", path,
                       ", ".join(unknown))
    return config


def load_config_layers(*paths: Union[str, Path]) -> Dict[str, Any]:
    if not paths:
        return {}
    result = load_config(paths[0])
    for p in paths[1:]:
        result = deep_merge(result, load_config(p))
    return result


def two_layer_builder(config: Dict[str, Any]) -> ModelBuilder:
    """builder.py:50-108: ``time_points = np.arange(start, end + 1)`` -> TimeAxis.from_values;
    TwoLayer from ``components.climate.parameters``; Ts, Td start at 0 unless
    ``[initial_values]`` says otherwise.  Like the reference builder this attaches NO forcing --
    add one with ``with_exogenous_variable`` (without it ``run()`` yields NaN, as there)."""
    params = config.get("components", {}).get("climate", {}).get("parameters", {})
    t = config.get("time", {})
    start, end = t.get("start", 1750), t.get("end", 2100)
    axis = TimeAxis.from_values(np.arange(start, end + 1, dtype=float))
    init = {"Surface Temperature": 0.0, "Deep Ocean Temperature": 0.0}
    init.update({k: float(v) for k, v in config.get("initial_values", {}).items()})
    return (ModelBuilder().with_time_axis(axis)
            .with_rust_component(TwoLayerBuilder.from_parameters(params).build())
            .with_initial_values(init))


def build_model(config: Dict[str, Any]):
    model_type = config.get("model", {}).get("type", "")
    if model_type == "two-layer":
        return two_layer_builder(config).build()
    raise ValueError(f"Unknown model type: {model_type!r}")
