#!/usr/bin/env python3
"""Walk the reference tree and record every call into the `qserve_backend` extension modules and into
`flash_attn_varlen_func`: file:line, module, function, number of positional arguments, keyword names.

Runs only where /root/reference exists (the authoring container).  Output: tests/golden/callsites.json, which
tests/test_callsites.py binds against the mirror's Python signatures (inspect.signature) - on any machine.
Nothing of the reference is copied: only the SHAPE of its calls is recorded.
"""
import ast
import json
import os
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "callsites.json")
BACKEND_MODULES = {"qgemm_w4a8_per_chn", "qgemm_w4a8_per_group", "qgemm_w8a8", "fused_attention", "fused_kernels",
                   "layernorm_ops", "activation_ops"}


def dotted(node):
    parts = []
    while isinstance(node, ast.Attribute):
        parts.append(node.attr)
        node = node.value
    if isinstance(node, ast.Name):
        parts.append(node.id)
        return list(reversed(parts))
    return None


def scan(path, rel):
    tree = ast.parse(open(path).read(), filename=path)
    aliases = {}          # local name -> backend module
    flash_names = set()
    for n in ast.walk(tree):
        if isinstance(n, ast.ImportFrom) and n.module == "qserve_backend":
            for a in n.names:
                if a.name in BACKEND_MODULES:
                    aliases[a.asname or a.name] = a.name
        elif isinstance(n, ast.Import):
            for a in n.names:
                if a.name.startswith("qserve_backend.") and a.name.split(".")[1] in BACKEND_MODULES and a.asname:
                    aliases[a.asname] = a.name.split(".")[1]
        elif isinstance(n, ast.ImportFrom) and n.module and n.module.startswith("flash_attn"):
            for a in n.names:
                if a.name == "flash_attn_varlen_func":
                    flash_names.add(a.asname or a.name)
    sites = []
    for n in ast.walk(tree):
        if not isinstance(n, ast.Call):
            continue
        d = dotted(n.func)
        if not d:
            continue
        mod = fn = None
        if len(d) == 3 and d[0] == "qserve_backend" and d[1] in BACKEND_MODULES:
            mod, fn = d[1], d[2]
        elif len(d) == 2 and d[0] in aliases:
            mod, fn = aliases[d[0]], d[1]
        elif len(d) == 1 and d[0] in flash_names:
            mod, fn = "flash_attn.flash_attn_interface", d[0]
        if mod is None:
            continue
        if any(isinstance(a, ast.Starred) for a in n.args) or any(k.arg is None for k in n.keywords):
            continue
        sites.append(dict(site=f"{rel}:{n.lineno}", module=mod, function=fn, positional=len(n.args),
                          keywords=sorted(k.arg for k in n.keywords)))
    return sites


def main():
    sites = []
    for base in ("qserve",):
        for dp, _, fs in os.walk(os.path.join(REF, base)):
            for f in sorted(fs):
                if f.endswith(".py"):
                    p = os.path.join(dp, f)
                    sites += scan(p, os.path.relpath(p, REF))
    sites.sort(key=lambda s: (s["site"].split(":")[0], int(s["site"].split(":")[1])))
    json.dump(dict(reference="mit-han-lab/qserve", sites=sites), open(OUT, "w"), indent=1)
    print(f"{len(sites)} call sites -> {OUT}")


if __name__ == "__main__":
    main()
