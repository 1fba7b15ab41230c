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


def test_attributes_used_on_the_binding_module_exist():
    """Every `_lib.<name>` the package, the bench, the tools and the tests mention is something cytvdn_amd/_lib.py defines
    (functions, constants, ctypes mirrors): a typo in GPU-only code would otherwise wait for a GPU box to show."""
    from cytvdn_amd import _lib
    files = sorted(glob.glob(os.path.join(ROOT, "cytvdn_amd", "*.py")) + glob.glob(os.path.join(ROOT, "tools", "*.py")) +
                   glob.glob(os.path.join(ROOT, "tests", "*.py")) + [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")])
    missing = []
    for path in files:
        tree = ast.parse(open(path).read(), filename=path)
        for node in ast.walk(tree):
            if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id == "_lib" \
                    and isinstance(node.ctx, ast.Load) and not hasattr(_lib, node.attr):
                missing.append(f"{os.path.relpath(path, ROOT)}:{node.lineno}: _lib.{node.attr}")
    assert not missing, "\n".join(missing)


def test_fields_set_on_the_argument_structs_exist():
    """`a.<field> = ...` on an IterArgs / RunArgs / ManyArgs instance silently creates a Python attribute when the field
    name is wrong (ctypes structures accept it) -- and the library then sees zero.  Keyword arguments of the
    constructors and attribute stores on names bound to them are checked against the declared fields."""
    from cytvdn_amd import _lib
    structs = {"IterArgs": _lib.IterArgs, "RunArgs": _lib.RunArgs, "ManyArgs": _lib.ManyArgs, "PlanOut": _lib.PlanOut}
    fields = {k: {f[0] for f in v._fields_} for k, v in structs.items()}
    files = sorted(glob.glob(os.path.join(ROOT, "cytvdn_amd", "*.py")) + glob.glob(os.path.join(ROOT, "tools", "*.py")) +
                   glob.glob(os.path.join(ROOT, "tests", "*.py")) + [os.path.join(ROOT, "bench.py")])
    bad = []
    for path in files:
        tree = ast.parse(open(path).read(), filename=path)
        for fn in [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)]:
            bindings = []                                 # (line, variable, struct kind or None) in this function
            for node in ast.walk(fn):
                if isinstance(node, ast.Assign):
                    kind = None
                    if isinstance(node.value, ast.Call):
                        f = node.value.func
                        kind = f.attr if isinstance(f, ast.Attribute) else (f.id if isinstance(f, ast.Name) else None)
                        if kind in structs:
                            for kw in node.value.keywords:
                                if kw.arg and kw.arg not in fields[kind]:
                                    bad.append(f"{os.path.relpath(path, ROOT)}:{node.lineno}: {kind}({kw.arg}=...)")
                        else:
                            kind = None
                    if isinstance(node.value, ast.Attribute) and node.value.attr == "_args":
                        kind = "IterArgs"                  # `A = self._args`: the engines' persistent argument block
                    for tgt in node.targets:
                        if isinstance(tgt, ast.Name):
                            bindings.append((node.lineno, tgt.id, kind))
            for node in ast.walk(fn):
                if isinstance(node, ast.Attribute) and isinstance(node.ctx, ast.Store) and isinstance(node.value, ast.Name):
                    before = [b for b in bindings if b[1] == node.value.id and b[0] <= node.lineno]
                    kind = max(before)[2] if before else None      # the binding in force: the latest one above this line
                    if kind and node.attr not in fields[kind]:
                        bad.append(f"{os.path.relpath(path, ROOT)}:{node.lineno}: {kind}.{node.attr} = ...")
    assert not bad, "\n".join(bad)
