"""bench.py and the tools only run on the GPU box: catch undefined names (a typo in a rarely taken branch would
otherwise surface in the driver's round-end run) with a small scope check here on the CPU."""
import ast
import builtins
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ["bench.py", "__graft_entry__.py"] + [os.path.join("tools", f) for f in sorted(os.listdir(os.path.join(ROOT, "tools"))) if f.endswith(".py")]


def _bound_names(node):
    """names bound directly in this scope (not in nested functions / classes)"""
    out = set()

    def visit(n, top):
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            out.add(n.name)
            if not top:
                return
        if isinstance(n, ast.Lambda) and not top:
            return
        if isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            out.add(n.id)
        elif isinstance(n, (ast.Import, ast.ImportFrom)):
            for a in n.names:
                out.add((a.asname or a.name).split(".")[0])
        elif isinstance(n, ast.ExceptHandler) and n.name:
            out.add(n.name)
        elif isinstance(n, (ast.Global, ast.Nonlocal)):
            out.update(n.names)
        elif isinstance(n, ast.arg):
            out.add(n.arg)
        for ch in ast.iter_child_nodes(n):
            visit(ch, False)
    visit(node, True)
    return out


def _check(node, env, errors, fname):
    scope = env | _bound_names(node)
    for ch in ast.walk(node) if False else ast.iter_child_nodes(node):
        _walk(ch, scope, errors, fname)


def _walk(n, scope, errors, fname):
    if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda)):
        _check(n, scope, errors, fname)
        return
    if isinstance(n, ast.ClassDef):
        _check(n, scope, errors, fname)
        return
    if isinstance(n, (ast.ListComp, ast.SetComp, ast.DictComp, ast.GeneratorExp)):
        inner = set(scope)
        for g in n.generators:
            for t in ast.walk(g.target):
                if isinstance(t, ast.Name):
                    inner.add(t.id)
        for ch in ast.iter_child_nodes(n):
            _walk(ch, inner, errors, fname)
        return
    if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in scope:
        errors.append("%s:%d: name %r is not defined in any enclosing scope" % (fname, n.lineno, n.id))
    for ch in ast.iter_child_nodes(n):
        _walk(ch, scope, errors, fname)


@pytest.mark.parametrize("rel", FILES)
def test_no_undefined_names(rel):
    src = open(os.path.join(ROOT, rel)).read()
    tree = ast.parse(src, rel)
    errors = []
    _check(tree, set(dir(builtins)) | {"__file__", "__name__", "__doc__"}, errors, rel)
    assert not errors, "\n".join(errors)
