"""Build-container script (needs /root/reference; NOT run on the GPU box): walks the reference's training and evaluation scripts
with `ast` and writes the NAMES they use on the drop-in surfaces -- attributes / methods read on the model object, keyword
arguments passed to render() / render_motion(), keys read from the dict render() returns, names imported from the two drop-in
modules -- to tests/golden/api_surface.json.  The fixture is data (a list of identifiers), not source; tests/test_api_surface.py
checks that this package exposes every one of them.

    python tests/golden/make_api_surface.py [/root/reference]
"""
import ast
import json
import os
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "api_surface.json")

MODEL_NAMES = {"gaussians", "pc"}                    # what the scripts call the GaussianModel instance
RENDER_FUNCS = {"render", "render_motion"}
PKG_NAMES = {"render_pkg", "pkg", "results", "rendering"}
DROPIN_MODULES = {"gaussian_renderer", "diff_gaussian_rasterization"}


def walk(path, lo, hi):
    tree = ast.parse(open(path).read())
    model_attrs, model_calls, render_kwargs, pkg_keys, imports = set(), set(), {}, set(), set()
    pkg_vars = set(PKG_NAMES)
    for node in ast.walk(tree):
        if isinstance(node, ast.ImportFrom) and node.module in DROPIN_MODULES:
            imports.update(f"{node.module}.{a.name}" for a in node.names)
        if not hasattr(node, "lineno") or not (lo <= node.lineno <= hi):
            continue
        if isinstance(node, ast.Assign) and isinstance(node.value, ast.Call) and isinstance(node.value.func, ast.Name) \
                and node.value.func.id in RENDER_FUNCS:
            for t in node.targets:
                if isinstance(t, ast.Name):
                    pkg_vars.add(t.id)
    for node in ast.walk(tree):
        if not hasattr(node, "lineno") or not (lo <= node.lineno <= hi):
            continue
        if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id in MODEL_NAMES:
            model_attrs.add(node.attr)
        if isinstance(node, ast.Call):
            f = node.func
            if isinstance(f, ast.Attribute) and isinstance(f.value, ast.Name) and f.value.id in MODEL_NAMES:
                model_calls.add(f.attr)
            if isinstance(f, ast.Name) and f.id in RENDER_FUNCS:
                render_kwargs.setdefault(f.id, set()).update(k.arg for k in node.keywords if k.arg)
                render_kwargs[f.id].add(f"#positional={len(node.args)}")
            if isinstance(f, ast.Name) and f.id in ("GaussianRasterizationSettings", "rasterizer"):
                render_kwargs.setdefault(f.id, set()).update(k.arg for k in node.keywords if k.arg)
            if isinstance(f, ast.Name) and f.id in MODEL_NAMES:               # gaussians(time, it, return_weights=...)
                render_kwargs.setdefault("__call__", set()).update(k.arg for k in node.keywords if k.arg)
        if isinstance(node, ast.Subscript):
            v = node.value
            key = node.slice
            if isinstance(key, ast.Constant) and isinstance(key.value, str):
                if isinstance(v, ast.Name) and v.id in pkg_vars:
                    pkg_keys.add(key.value)
                if isinstance(v, ast.Call) and isinstance(v.func, ast.Name) and v.func.id in RENDER_FUNCS:
                    pkg_keys.add(key.value)
    return dict(model_attributes=sorted(model_attrs), model_methods_called=sorted(model_calls),
                render_keywords={k: sorted(v) for k, v in sorted(render_kwargs.items())}, render_pkg_keys=sorted(pkg_keys),
                imports=sorted(imports))


surface = {
    "_generated_by": "tests/golden/make_api_surface.py (ast walk; identifiers only)",
    "train.py:36-201": walk(os.path.join(REF, "train.py"), 36, 201),
    "eval.py:35-258": walk(os.path.join(REF, "eval.py"), 35, 258),
    "gaussian_renderer/__init__.py:18-191": walk(os.path.join(REF, "gaussian_renderer", "__init__.py"), 18, 191),
}
json.dump(surface, open(OUT, "w"), indent=1, sort_keys=True)
print(json.dumps(surface, indent=1))
