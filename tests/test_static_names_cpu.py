"""Names used in the package, the bench and the tools must be defined somewhere: most of this code only runs on a GPU box,
so a misspelt or un-imported name would otherwise show up there for the first time.  A small scope-aware check over the
syntax trees (no third-party linter in this image)."""
import ast
import builtins
import glob
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Scope:
    def __init__(self, parent=None, is_class=False):
        self.parent, self.names, self.is_class = parent, set(), is_class


def _bind_targets(node, scope):
    for n in ast.walk(node):
        if isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            scope.names.add(n.id)


def _collect(node, scope):
    """First pass over one scope's body: everything it binds (anywhere in the body, order ignored)."""
    for child in ast.iter_child_nodes(node):
        if isinstance(child, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            scope.names.add(child.name)
            for d in child.decorator_list:
                _collect(d, scope)
            continue                                  # its body is its own scope
        if isinstance(child, ast.Lambda):
            continue
        if isinstance(child, (ast.Import, ast.ImportFrom)):
            for a in child.names:
                scope.names.add((a.asname or a.name).split(".")[0])
        elif isinstance(child, ast.Name) and isinstance(child.ctx, (ast.Store, ast.Del)):
            scope.names.add(child.id)
        elif isinstance(child, ast.ExceptHandler) and child.name:
            scope.names.add(child.name)
        elif isinstance(child, (ast.Global, ast.Nonlocal)):
            scope.names.update(child.names)
        elif isinstance(child, (ast.ListComp, ast.SetComp, ast.DictComp, ast.GeneratorExp)):
            for g in child.generators:               # comprehension variables: treated as bound in the enclosing scope
                _bind_targets(g.target, scope)
        _collect(child, scope)


def _visible(name, scope):
    s = scope
    first = True
    while s is not None:
        if name in s.names and (first or not s.is_class):
            return True
        s, first = s.parent, False
    return hasattr(builtins, name)


def _check(node, scope, problems, fname):
    for child in ast.iter_child_nodes(node):
        if isinstance(child, (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda)):
            inner = Scope(scope)
            a = child.args
            for arg in a.posonlyargs + a.args + a.kwonlyargs + ([a.vararg] if a.vararg else []) + ([a.kwarg] if a.kwarg else []):
                inner.names.add(arg.arg)
            for d in a.defaults + [d for d in a.kw_defaults if d is not None]:
                _check_expr(d, scope, problems, fname)
            if isinstance(child, ast.Lambda):
                _collect(child, inner)
                _check(child, inner, problems, fname)
            else:
                for st in child.body:
                    holder = ast.Module(body=[st], type_ignores=[])
                    _collect(holder, inner)
                for st in child.body:
                    _check(ast.Module(body=[st], type_ignores=[]), inner, problems, fname)
            continue
        if isinstance(child, ast.ClassDef):
            inner = Scope(scope, is_class=True)
            for st in child.body:
                _collect(ast.Module(body=[st], type_ignores=[]), inner)
            for b in child.bases:
                _check_expr(b, scope, problems, fname)
            for st in child.body:
                _check(ast.Module(body=[st], type_ignores=[]), inner, problems, fname)
            continue
        if isinstance(child, ast.Name) and isinstance(child.ctx, ast.Load) and not _visible(child.id, scope):
            problems.append(f"{fname}:{child.lineno}: undefined name '{child.id}'")
        _check(child, scope, problems, fname)


def _check_expr(expr, scope, problems, fname):
    _check(ast.Module(body=[ast.Expr(value=expr)], type_ignores=[]), scope, problems, fname)


def undefined_names(path):
    tree = ast.parse(open(path).read(), filename=path)
    top = Scope()
    top.names.update({"__file__", "__name__", "__doc__", "__builtins__", "__spec__", "__package__"})
    _collect(tree, top)
    problems = []
    _check(tree, top, problems, os.path.relpath(path, ROOT))
    return problems


def test_the_checker_finds_a_missing_import(tmp_path):
    f = tmp_path / "m.py"
    f.write_text("import sys\n\ndef g(a):\n    b = a + 1\n    return os.path.join(str(b), sys.argv[0], [q for q in range(3)], undefined_thing)\n")
    found = undefined_names(str(f))
    assert any("'os'" in p for p in found) and any("'undefined_thing'" in p for p in found) and len(found) == 2


def test_no_undefined_names_in_the_tree():
    files = sorted(glob.glob(os.path.join(ROOT, "cytvdn_amd", "*.py")) + glob.glob(os.path.join(ROOT, "tools", "*.py")) +
                   glob.glob(os.path.join(ROOT, "tools", "ubench", "*.py")) + glob.glob(os.path.join(ROOT, "oracle", "*.py")) +
                   glob.glob(os.path.join(ROOT, "tests", "*.py")) +
                   [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")])
    assert len(files) > 40
    problems = [p for f in files for p in undefined_names(f)]
    assert not problems, "\n".join(problems)
