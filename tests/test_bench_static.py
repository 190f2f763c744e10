"""bench.py is what the driver runs at round end, also with --gpus N: names it uses must exist on every path (a NameError in the
N > 1 branch shows only on a multi-GPU node; round 3 had one).  The package's modules are checked the same way.  Static check: every name loaded in a function is bound somewhere in that function,
in an enclosing function, at module level, or is a builtin."""
import ast
import builtins
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _walk_scope(node):
    """the nodes of one scope: does not descend into nested function / class bodies (their names are theirs)"""
    todo = list(node.body) if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef)) else list(ast.iter_child_nodes(node))
    while todo:
        n = todo.pop()
        yield n
        if not isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef, ast.Lambda)):
            todo.extend(ast.iter_child_nodes(n))


def _bound_names(node):
    names = set()
    if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef)):
        a = node.args
        for arg in a.posonlyargs + a.args + a.kwonlyargs + ([a.vararg] if a.vararg else []) + ([a.kwarg] if a.kwarg else []):
            names.add(arg.arg)
    for n in _walk_scope(node):
        if isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            names.add(n.id)
        elif isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            names.add(n.name)
        elif isinstance(n, ast.arg):
            names.add(n.arg)
        elif isinstance(n, (ast.Import, ast.ImportFrom)):
            for a in n.names:
                names.add((a.asname or a.name).split(".")[0])
        elif isinstance(n, ast.ExceptHandler) and n.name:
            names.add(n.name)
        elif isinstance(n, (ast.Global, ast.Nonlocal)):
            names.update(n.names)
    return names


def _sources():
    import glob
    return ["bench.py", "__graft_entry__.py"] + sorted(os.path.relpath(p, REPO) for p in glob.glob(os.path.join(REPO, "flid_amd", "**", "*.py"), recursive=True))


def test_bench_has_no_undefined_names():
    for fn in _sources():
        tree = ast.parse(open(os.path.join(REPO, fn)).read())
        module_names = _bound_names(tree) | set(dir(builtins)) | {"__file__", "__name__", "__class__"}
        bad = []

        def visit(func, outer):
            scope = outer | _bound_names(func)
            for n in _walk_scope(func):
                if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in scope:
                    bad.append((fn, func.name, n.id, n.lineno))
                elif isinstance(n, ast.Lambda):                   # a lambda's body: its parameters + the enclosing scope
                    inner = scope | {a.arg for a in n.args.args + n.args.kwonlyargs} | \
                        {m.id for m in ast.walk(n.body) if isinstance(m, ast.Name) and isinstance(m.ctx, ast.Store)}
                    for m in ast.walk(n.body):
                        if isinstance(m, ast.Name) and isinstance(m.ctx, ast.Load) and m.id not in inner:
                            bad.append((fn, func.name, m.id, m.lineno))
            for child in _walk_scope(func):
                if isinstance(child, (ast.FunctionDef, ast.AsyncFunctionDef)):
                    visit(child, scope)

        def walk_top(body, outer):
            for top in body:
                if isinstance(top, (ast.FunctionDef, ast.AsyncFunctionDef)):
                    visit(top, outer)
                elif isinstance(top, ast.ClassDef):              # methods see the module's names (not the class body's)
                    walk_top(top.body, outer)

        walk_top(tree.body, module_names)
        assert not bad, bad
