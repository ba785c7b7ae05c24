"""Every name a function of the package uses resolves to a local, an enclosing scope, a module global or a builtin.
The GPU paths cannot run in the CPU suite, so a slip like a deleted closure variable (a NameError at run time on the GPU
box only) is caught here instead."""
import builtins
import glob
import os
import symtable

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _unresolved(path):
    src = open(path).read()
    top = symtable.symtable(src, path, "exec")
    module_names = set(top.get_identifiers())
    bad = []

    def walk(tab):
        for sym in tab.get_symbols():
            if sym.is_global() and not sym.is_declared_global():
                n = sym.get_name()
                if n not in module_names and not hasattr(builtins, n):
                    bad.append((n, tab.get_name(), tab.get_lineno()))
        for ch in tab.get_children():
            walk(ch)

    walk(top)
    return bad


def test_no_unresolved_names_in_package_and_bench():
    files = sorted(glob.glob(os.path.join(ROOT, "blackjax_amd", "*.py"))) + [os.path.join(ROOT, "bench.py"),
                                                                              os.path.join(ROOT, "__graft_entry__.py")]
    problems = {os.path.relpath(f, ROOT): _unresolved(f) for f in files}
    problems = {f: b for f, b in problems.items() if b}
    assert not problems, problems
