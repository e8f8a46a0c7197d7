"""Model (de)serialisation and graph rendering for the host-side mirror.

``model.to_toml()`` / ``Model.from_toml(text)`` mirror the reference's ``Model::to_toml`` / ``from_toml``
(python/rscm/_lib/core/__init__.pyi:563-628; crates/rscm-core/src/model/runtime.rs:270-300): the text holds
the description of the model -- time axis, components with their parameters, schema, exogenous series,
initial values -- and the state needed to continue from the current step (``checkpoint()``: stored rows,
look-back rows, internal component states).  The layout is this package's own (the reference
serialises its Rust structs through serde/typetag; that layout is not reproduced); it is plain TOML,
read back with ``tomli``.  Meant for models of a few members -- a checkpoint of a large ensemble
belongs in ``core.save_checkpoint`` (``.npz``).

``model.as_dot()`` renders the component graph (nodes in registration order, one labelled edge per
variable that flows from a producer to a consumer) as Graphviz DOT, like the reference's debugging aid
(runtime.rs:529-544).
"""
from __future__ import annotations

import math
from typing import Dict, List

import numpy as np


# --------------------------------------------------------------------------------- TOML writer
def _scalar(v) -> str:
    if isinstance(v, (bool, np.bool_)):
        return "true" if v else "false"
    if isinstance(v, (int, np.integer)):
        return str(int(v))
    if isinstance(v, (float, np.floating)):
        x = float(v)
        if math.isnan(x):
            return "nan"
        if math.isinf(x):
            return "inf" if x > 0 else "-inf"
        r = repr(x)
        return r if any(c in r for c in ".en") else r + ".0"
    if isinstance(v, str):
        return '"' + v.replace("\\", "\\\\").replace('"', '\\"') + '"'
    raise TypeError(f"cannot write {type(v).__name__} to TOML")


def _value(v) -> str:
    if isinstance(v, np.ndarray):
        v = v.tolist()
    if isinstance(v, (list, tuple)):
        return "[" + ", ".join(_value(x) for x in v) + "]"
    return _scalar(v)


def _key(k: str) -> str:
    return k if k and all(c.isalnum() or c in "_-" for c in k) else _scalar(str(k))


def dumps(obj: Dict[str, object]) -> str:
    """A nested dict of scalars / strings / (nested) lists of numbers / lists of dicts as TOML."""
    lines: List[str] = []

    def is_table_array(v) -> bool:
        return isinstance(v, (list, tuple)) and len(v) > 0 and all(isinstance(x, dict) for x in v)

    def emit(table: Dict[str, object], path: List[str]) -> None:
        for k, v in table.items():
            if v is None or isinstance(v, dict) or is_table_array(v):
                continue
            lines.append(f"{_key(k)} = {_value(v)}")
        for k, v in table.items():
            if isinstance(v, dict):
                lines.append("")
                lines.append("[" + ".".join(_key(p) for p in path + [k]) + "]")
                emit(v, path + [k])
            elif is_table_array(v):
                for item in v:
                    lines.append("")
                    lines.append("[[" + ".".join(_key(p) for p in path + [k]) + "]]")
                    emit(item, path + [k])

    emit(obj, [])
    return "\n".join(lines) + "\n"


def loads(text: str) -> Dict[str, object]:
    import tomli
    return tomli.loads(text)


# ----------------------------------------------------------------------- model <-> description
def component_registry() -> Dict[str, type]:
    from . import components, magicc, two_layer
    from .core import Component
    out: Dict[str, type] = {}
    for mod in (two_layer, components, magicc):
        for obj in vars(mod).values():
            if isinstance(obj, type) and issubclass(obj, Component) and obj is not Component:
                out[obj.type_name] = obj
    return out


def describe(builder, model) -> Dict[str, object]:
    """The builder's description plus the model's checkpoint as a TOML-ready dict."""
    from .core import GraphModel
    doc: Dict[str, object] = {
        "model": {"format": "rscm_amd-model-1", "graph": isinstance(model, GraphModel), "n_members": int(model.n_members),
                  "time_index": int(model.time_index), "execution_order": getattr(model, "_execution_order", "reference"),
                  "device": int(builder._device)},
        "time_axis": {"bounds": builder._axis.bounds()},
        "components": [{"type": c.type_name, **({"step_size": float(c.step_size)} if hasattr(c, "step_size") else {}),
                        "parameters": {k: float(v) for k, v in c.parameters.items()}} for c in builder._components],
        "initial_values": dict(builder._initial),
        "exogenous": [],
    }
    for name in builder._exogenous.names():
        ts = builder._exogenous.get_timeseries_by_name(name)
        doc["exogenous"].append({"name": name, "units": ts.units, "interpolation": ts.interpolation_strategy.name,
                                 "bounds": ts.time_axis.bounds(), "values": ts.values()})
    if builder._schema is not None:
        s = builder._schema
        doc["schema"] = {"variables": [{"name": n, "unit": v.unit, "grid": v.grid_type.name} for n, v in s.variables.items()],
                         "aggregates": [{"name": n, "unit": u, "operation": op, "contributors": list(c),
                                         **({"weights": list(w)} if w is not None else {})}
                                        for n, (u, op, c, w) in s.aggregates.items()]}
    if builder._grid_weights:
        doc["grid_weights"] = {k.name: list(v) for k, v in builder._grid_weights.items()}
    doc["state"] = _plain(model.checkpoint())
    return doc


def _plain(x):
    if isinstance(x, dict):
        return {str(k): _plain(v) for k, v in x.items() if v is not None}
    if isinstance(x, np.ndarray):
        return x.tolist()
    if isinstance(x, (np.integer,)):
        return int(x)
    if isinstance(x, (np.floating,)):
        return float(x)
    return x


def rebuild(doc: Dict[str, object]):
    """ModelBuilder + restored model from a description written by ``describe``."""
    from . import core
    if doc.get("model", {}).get("format") != "rscm_amd-model-1":
        raise ValueError("not a model written by rscm_amd (missing or unknown [model] format)")
    registry = component_registry()
    b = core.ModelBuilder().with_time_axis(core.TimeAxis.from_bounds(np.array(doc["time_axis"]["bounds"], dtype=np.float64)))
    b.with_device(int(doc["model"].get("device", 0)))
    if "schema" in doc:
        schema = core.VariableSchema()
        for v in doc["schema"].get("variables", []):
            schema.add_variable(v["name"], v["unit"], core.GridType[v["grid"]])
        for a in doc["schema"].get("aggregates", []):
            schema.add_aggregate(a["name"], a["unit"], a["operation"], a["contributors"], a.get("weights"))
        b.with_schema(schema)
    for name, w in doc.get("grid_weights", {}).items():
        b.with_grid_weights(core.GridType[name], w)
    for c in doc.get("components", []):
        if c["type"] not in registry:
            raise ValueError(f"unknown component type {c['type']!r}")
        comp = registry[c["type"]](dict(c.get("parameters", {})))
        if "step_size" in c:
            comp.step_size = float(c["step_size"])
        b.with_rust_component(comp)
    b.with_initial_values(doc.get("initial_values", {}))
    for e in doc.get("exogenous", []):
        ts = core.Timeseries(np.array(e["values"], dtype=np.float64), core.TimeAxis.from_bounds(np.array(e["bounds"], dtype=np.float64)),
                             e.get("units", ""), core.InterpolationStrategy[e["interpolation"]])
        b.with_exogenous_variable(e["name"], ts)
    model = b.build(n_members=int(doc["model"]["n_members"]), execution_order=doc["model"].get("execution_order", "reference"))
    model.restore(_arrays(doc["state"]))
    return model


_ARRAY_KEYS = ("bounds", "params", "internal")


def _arrays(ck):
    """Checkpoint dict read back from TOML: lists become the arrays ``restore`` expects."""
    if "ensembles" in ck:  # a GraphModel checkpoint
        out = dict(ck)
        out["ensembles"] = {k: _arrays(v) for k, v in ck["ensembles"].items()}
        return out
    out = dict(ck)
    for k in _ARRAY_KEYS:
        if k in out:
            out[k] = np.array(out[k], dtype=np.float64)
    out.setdefault("internal", None)
    out["state"] = {k: np.array(v, dtype=np.float64) for k, v in ck.get("state", {}).items()}
    out["history"] = {k: np.array(v, dtype=np.float64) for k, v in ck.get("history", {}).items()}
    return out


# ------------------------------------------------------------------------------------- DOT
def as_dot(builder) -> str:
    """Graphviz DOT of the component graph: node 0 is the root ModelBuilder::build starts the
    traversal from, then the components in registration order, then the schema aggregates."""
    aggregates = builder._schema.aggregates if builder._schema else {}
    names = ["<root>"] + [c.type_name for c in builder._components] + [f"Aggregator:{a}" for a in aggregates]
    producer: Dict[str, int] = {}
    edges: List[tuple] = []
    for i, comp in enumerate(builder._components, start=1):
        fed = False
        for name, _, kind in comp.definitions:
            if kind in ("Input", "State") and name in producer and producer[name] != i:
                edges.append((producer[name], i, name))
                fed = True
            elif kind == "Input" and name in aggregates:
                edges.append((len(builder._components) + 1 + list(aggregates).index(name), i, name))
                fed = True
        if not fed:
            edges.append((0, i, ""))
        for name, _, kind in comp.definitions:
            if kind in ("Output", "State"):
                producer[name] = i
    for k, (agg, (_, _, contributors, _)) in enumerate(aggregates.items()):
        node = len(builder._components) + 1 + k
        fed = False
        for c in contributors:
            if c in producer:
                edges.append((producer[c], node, c))
                fed = True
        if not fed:
            edges.append((0, node, ""))
    lines = ["digraph {"]
    for i, n in enumerate(names):
        lines.append(f"    {i} [ label = {_scalar(n)} ]")
    for a, b_, label in edges:
        lines.append(f"    {a} -> {b_} [ label = {_scalar(label)} ]")
    lines.append("}")
    return "\n".join(lines) + "\n"


# ------------------------------------------------------------------------------- debug_info
def debug_info(builder, model, fmt: str = "plain") -> str:
    """Model::debug_info of the reference (crates/rscm-core/src/python/model.rs:207-218): the execution
    order with, per component, its inputs tagged with their source (exo / upstream / own_state), its
    states and outputs, grid tags for FourBox variables.  ``fmt``: "plain", "rich" (ANSI colours) or
    "json" ({"components": [{"order", "name", "inputs", "states", "outputs"}]})."""
    import json
    if fmt not in ("plain", "rich", "json"):
        raise ValueError(f"Unknown format '{fmt}'. Expected 'rich', 'plain', or 'json'.")
    from .core import GridType
    aggregates = builder._schema.aggregates if builder._schema else {}
    order = list(getattr(model, "_order", None) or builder._graph_order(aggregates))
    sources = model.variable_sources()
    tag = {"Exogenous": "exo", "UpstreamOutput": "upstream", "OwnState": "own_state"}
    by_name = {c.type_name: c for c in builder._components}
    fourbox = {name for c in builder._components for name, _, _ in c.definitions
               if builder._schema is not None and builder._schema.grid_types.get(name) == GridType.FourBox}
    comps = []
    for k, node in enumerate(order):
        entry = {"order": k, "name": node, "inputs": [], "states": [], "outputs": []}
        if node in by_name:
            for name, unit, kind in by_name[node].definitions:
                item = {"name": name, "unit": unit, "grid": "FourBox" if name in fourbox else "Scalar"}
                if kind == "Input":
                    entry["inputs"].append({**item, "source": tag.get(sources.get((name, node), "Exogenous"), "exo")})
                elif kind == "State":
                    entry["states"].append(item)
                else:
                    entry["outputs"].append(item)
        elif node.startswith("Aggregator:"):
            agg = aggregates[node[len("Aggregator:"):]]
            entry["operation"] = agg.operation_type
            entry["inputs"] = [{"name": c, "unit": agg.unit, "grid": "Scalar", "source": "upstream"} for c in agg.contributors]
            entry["outputs"] = [{"name": agg.name, "unit": agg.unit, "grid": "Scalar"}]
        elif node.startswith("Transform:"):
            var = node[len("Transform:"):]
            entry["inputs"] = [{"name": var, "unit": "", "grid": "FourBox", "source": "upstream"}]
            entry["outputs"] = [{"name": var, "unit": "", "grid": "Scalar"}]
        comps.append(entry)
    if fmt == "json":
        return json.dumps({"time_index": int(model.time_index), "n_members": int(model.n_members), "components": comps})
    colour = (lambda code, s: f"\x1b[{code}m{s}\x1b[0m") if fmt == "rich" else (lambda code, s: s)
    lines = [f"Model at time index {model.time_index} ({model.n_members} member(s)); execution order:"]
    for e in comps:
        lines.append(f"[{e['order']}] {e['name']}" + (f" ({e['operation']})" if "operation" in e else ""))
        for i in e["inputs"]:
            grid = colour(33, " [FourBox]") if i["grid"] == "FourBox" else ""
            lines.append("    " + colour(32, "<-") + f" {i['name']} ({i['source']}){grid}")
        for s_ in e["states"]:
            grid = colour(33, " [FourBox]") if s_["grid"] == "FourBox" else ""
            lines.append("    " + colour(35, "<>") + f" {s_['name']}{grid}")
        for o in e["outputs"]:
            grid = colour(33, " [FourBox]") if o["grid"] == "FourBox" else ""
            lines.append("    " + colour(34, "->") + f" {o['name']}{grid}")
    return "\n".join(lines) + "\n"
