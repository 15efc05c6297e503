"""CPU: static undefined-name check of every Python file of the repo.  Most of the product and of the -m gpu tests only executes on
the GPU box; a misspelt name there would surface as a NameError at round end.  For every function, the names the compiler treats
as global references (symtable) must be bound at module level or be builtins."""
import ast
import builtins
import glob
import os
import symtable

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def module_names(tree):
    names=set()
    for node in ast.walk(tree):
        if isinstance(node,(ast.FunctionDef,ast.AsyncFunctionDef,ast.ClassDef)): names.add(node.name)
        elif isinstance(node,ast.Import):
            for a in node.names: names.add((a.asname or a.name).split('.')[0])
        elif isinstance(node,ast.ImportFrom):
            for a in node.names: names.add(a.asname or a.name)
        elif isinstance(node,(ast.Assign,ast.AugAssign,ast.AnnAssign)):
            tg = node.targets if isinstance(node,ast.Assign) else [node.target]
            for t in tg:
                for n in ast.walk(t):
                    if isinstance(n,ast.Name): names.add(n.id)
        elif isinstance(node,(ast.For,ast.comprehension)):
            for n in ast.walk(node.target):
                if isinstance(n,ast.Name): names.add(n.id)
        elif isinstance(node,ast.With):
            for it in node.items:
                if it.optional_vars is not None:
                    for n in ast.walk(it.optional_vars):
                        if isinstance(n,ast.Name): names.add(n.id)
        elif isinstance(node, ast.ExceptHandler) and node.name: names.add(node.name)
        elif isinstance(node, ast.Global):
            names.update(node.names)
    return names
def check(path):
    src=open(path).read()
    tree=ast.parse(src)
    # module-level (global) names: conservative = any binding anywhere at module scope
    top=set()
    for node in tree.body:
        top |= module_names(ast.Module(body=[node], type_ignores=[])) if not isinstance(node,(ast.FunctionDef,ast.ClassDef,ast.AsyncFunctionDef)) else {node.name}
    # also names declared `global` inside functions
    for node in ast.walk(tree):
        if isinstance(node, ast.Global): top.update(node.names)
    bi=set(dir(builtins))|{'__file__','__name__','__doc__'}
    st=symtable.symtable(src,path,'exec')
    bad=[]
    def walk(t):
        for c in t.get_children():
            if c.get_type() in ('function',):
                for s in c.get_symbols():
                    if s.is_global() and s.is_referenced() and not s.is_assigned():
                        if s.get_name() not in top and s.get_name() not in bi:
                            bad.append((c.get_name(), c.get_lineno(), s.get_name()))
            walk(c)
    walk(st)
    # module-level references
    return bad


def test_no_undefined_global_names_anywhere():
    files = []
    for pat in ("tests/*.py", "flash_diffusion_amd/*.py", "oracle/*.py", "scripts/*.py", "bench.py", "__graft_entry__.py"):
        files += sorted(glob.glob(os.path.join(ROOT, pat)))
    assert len(files) > 40
    bad = [(os.path.relpath(f, ROOT),) + b for f in files for b in check(f)]
    assert not bad, bad
