"""Undefined-name check of the package and the tests without third-party linters (there is no network in the image):
compiles every file and reports names a function body loads that no enclosing scope, global, builtin or import provides."""
import ast
import builtins
import sys


def check(path):
    src = open(path).read()
    tree = ast.parse(src, path)
    bad = []
    mod_names = set(dir(builtins)) | {"__file__", "__name__", "__doc__"}
    for n in ast.walk(tree):
        if isinstance(n, (ast.Import, ast.ImportFrom)):
            for a in n.names:
                mod_names.add((a.asname or a.name).split(".")[0])
        elif isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            mod_names.add(n.name)
    for n in tree.body:
        for t in ast.walk(n):
            if isinstance(t, ast.Name) and isinstance(t.ctx, (ast.Store, ast.Del)):
                mod_names.add(t.id)

    def scope_names(fn):
        names = set()
        a = fn.args
        for arg in a.posonlyargs + a.args + a.kwonlyargs:
            names.add(arg.arg)
        if a.vararg:
            names.add(a.vararg.arg)
        if a.kwarg:
            names.add(a.kwarg.arg)
        for t in ast.walk(fn):
            if isinstance(t, ast.Name) and isinstance(t.ctx, (ast.Store, ast.Del)):
                names.add(t.id)
            elif isinstance(t, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
                names.add(t.name)
            elif isinstance(t, (ast.Import, ast.ImportFrom)):
                for al in t.names:
                    names.add((al.asname or al.name).split(".")[0])
            elif isinstance(t, ast.ExceptHandler) and t.name:
                names.add(t.name)
            elif isinstance(t, (ast.Global, ast.Nonlocal)):
                names.update(t.names)
        return names

    def visit(node, env):
        for ch in ast.iter_child_nodes(node):
            if isinstance(ch, (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda)):
                inner = env | (scope_names(ch) if not isinstance(ch, ast.Lambda) else {a.arg for a in ch.args.args + ch.args.kwonlyargs} | ({ch.args.vararg.arg} if ch.args.vararg else set()) | ({ch.args.kwarg.arg} if ch.args.kwarg else set()))
                visit(ch, inner)
            elif isinstance(ch, ast.ClassDef):
                cls = set()
                for t in ch.body:
                    for u in ast.walk(t):
                        if isinstance(u, ast.Name) and isinstance(u.ctx, ast.Store):
                            cls.add(u.id)
                        elif isinstance(u, (ast.FunctionDef, ast.ClassDef)):
                            cls.add(u.name)
                visit(ch, env | cls)
            elif isinstance(ch, (ast.ListComp, ast.SetComp, ast.DictComp, ast.GeneratorExp)):
                comp = set()
                for g in ch.generators:
                    for u in ast.walk(g.target):
                        if isinstance(u, ast.Name):
                            comp.add(u.id)
                visit(ch, env | comp)
            else:
                if isinstance(ch, ast.Name) and isinstance(ch.ctx, ast.Load) and ch.id not in env:
                    bad.append((ch.lineno, ch.id))
                visit(ch, env)

    visit(tree, mod_names)
    return bad


if __name__ == "__main__":
    rc = 0
    for p in sys.argv[1:]:
        for ln, name in check(p):
            print("%s:%d: undefined name %r" % (p, ln, name))
            rc = 1
    sys.exit(rc)
