#!/usr/bin/env python3
"""Regenerate tools/README.md: one table row per script (first sentence of its docstring / leading comment)."""
import ast, glob, os, re
ROOT = os.path.dirname(os.path.abspath(__file__))


def first_sentence(t):
    t = " ".join(t.split())
    return t[:240] + ("..." if len(t) > 240 else "")


def doc_of(path):
    src = open(path, errors="replace").read()
    if path.endswith(".py"):
        try:
            d = ast.get_docstring(ast.parse(src)) or ""
        except SyntaxError:
            d = ""
        return first_sentence(d.split("\n\n")[0].split(":\n")[0]) if d else ""
    lines = [l for l in src.splitlines()[:12]]
    com = [re.sub(r"^(#|//)\s?", "", l) for l in lines if l.startswith(("#", "//")) and not l.startswith("#!")]
    return first_sentence(" ".join(com).split(". ")[0]) if com else ""


rows = []
for pat in ("*.py", "gpu/*.sh", "probe/*.hip", "probe/*.py"):
    for f in sorted(glob.glob(os.path.join(ROOT, pat))):
        rel = "tools/" + os.path.relpath(f, ROOT)
        if rel.endswith("make_index.py"):
            continue
        rows.append("| `%s` | %s |" % (rel, doc_of(f).replace("|", "/")))
head = """# tools/ index

Micro-benchmarks, profiling helpers and hardware probes behind the numbers in `DESIGN.md`.  None of this is imported by the
product (`ofq_amd/`); one test (`tests/test_graph_gpu.py`) runs `tools/two_rank_trace.py` as a subprocess; every script runs on the GPU box from the repository root (`gpurun -- 'python tools/x.py'`).
The `OFQ_*` environment switches some of them set select between kernels that give the same results (A/B switches, test hooks).
(This file: `python tools/make_index.py`.)

| script | what it does |
|---|---|
"""
open(os.path.join(ROOT, "README.md"), "w").write(head + "\n".join(rows) + "\n")
print(len(rows), "entries")
