#!/usr/bin/env python3
"""Writes tests/golden/reference_python_api.json: the public names of the reference's Python package with their parameter
names and default values (text of the default expression), read from /root/reference/nvmolkit/*.py with `ast` — an
inventory of the interface the drop-in has to offer, not source text.  Run in the build container (the reference is not on
the GPU box):  python tests/golden/make_api_fixture.py"""
import ast
import json
from pathlib import Path

REF = Path("/root/reference/nvmolkit")


def params(fn: ast.FunctionDef, drop_self: bool):
    a = fn.args
    pos = a.posonlyargs + a.args
    defaults = [None] * (len(pos) - len(a.defaults)) + [ast.unparse(d) for d in a.defaults]
    out = [{"name": p.arg, "default": d, "kind": "positional"} for p, d in zip(pos, defaults)]
    if drop_self and out:
        out = out[1:]
    out += [{"name": p.arg, "default": None if d is None else ast.unparse(d), "kind": "keyword"} for p, d in zip(a.kwonlyargs, a.kw_defaults)]
    return out


def is_overload(fn):
    return any((isinstance(d, ast.Name) and d.id == "overload") or (isinstance(d, ast.Attribute) and d.attr == "overload")
               for d in fn.decorator_list)


api = {}
for path in sorted(REF.glob("*.py")):
    tree = ast.parse(path.read_text())
    mod = {}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and not node.name.startswith("_") and not is_overload(node):
            mod[node.name] = {"type": "function", "params": params(node, False)}
        elif isinstance(node, ast.ClassDef) and not node.name.startswith("_"):
            methods = {}
            for m in node.body:
                if isinstance(m, ast.FunctionDef) and (not m.name.startswith("_") or m.name == "__init__") and not is_overload(m):
                    static = any(isinstance(d, ast.Name) and d.id == "staticmethod" for d in m.decorator_list)
                    prop = any(isinstance(d, ast.Name) and d.id == "property" for d in m.decorator_list) or any(
                        isinstance(d, ast.Attribute) and d.attr == "setter" for d in m.decorator_list)
                    methods[m.name] = {"property": True} if prop else {"params": params(m, not static)}
            mod[node.name] = {"type": "class", "bases": [ast.unparse(b) for b in node.bases], "methods": methods}
    if mod:
        api[path.name[:-3]] = mod
out = Path(__file__).with_name("reference_python_api.json")
out.write_text(json.dumps(api, indent=1, sort_keys=True) + "\n")
print(out, sum(len(m) for m in api.values()), "public names in", len(api), "modules")
