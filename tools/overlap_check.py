#!/usr/bin/env python
"""
Statement-level overlap of a product file with the reference package (developer aid, run in the build container only:
/root/reference does not exist on the GPU box).  Both sides are parsed; every statement is reduced to its own text
(compound statements to their header), docstrings dropped, and the product's statements are looked up in the set of all
reference statements.  Prints the fraction found and the contiguous runs of >= 6 found statements.

    python tools/overlap_check.py slmsuite_amd/holography/algorithms/__init__.py
"""
import ast
import glob
import os
import sys

REF = "/root/reference/slmsuite"


def statements(path):
    src = open(path).read()
    tree = ast.parse(src)
    out = []
    for node in ast.walk(tree):
        if not isinstance(node, ast.stmt):
            continue
        if isinstance(node, ast.Expr) and isinstance(node.value, ast.Constant) and isinstance(node.value.value, str):
            continue                                   # docstring
        if isinstance(node, (ast.FunctionDef, ast.ClassDef, ast.If, ast.For, ast.While, ast.With, ast.Try, ast.AsyncFunctionDef)):
            clone = type(node)(**{f: getattr(node, f) for f in node._fields})
            for f in ("body", "orelse", "finalbody", "handlers"):
                if hasattr(clone, f):
                    setattr(clone, f, [])
            try:
                text = ast.unparse(clone).split("\n")[0]
            except Exception:
                continue
        else:
            text = ast.unparse(node)
        out.append((node.lineno, "".join(text.split())))
    out.sort()
    return out


def main():
    ref = set()
    for f in glob.glob(os.path.join(REF, "**", "*.py"), recursive=True):
        try:
            ref |= {t for _, t in statements(f)}
        except SyntaxError:
            pass
    for path in sys.argv[1:]:
        mine = statements(path)
        hits = [(ln, t in ref and len(t) > 8) for ln, t in mine]
        frac = sum(h for _, h in hits) / max(1, len(hits))
        runs, cur = [], []
        for ln, h in hits:
            if h:
                cur.append(ln)
            else:
                if len(cur) >= 6:
                    runs.append((cur[0], cur[-1], len(cur)))
                cur = []
        if len(cur) >= 6:
            runs.append((cur[0], cur[-1], len(cur)))
        print(f"{path}: {len(mine)} statements, {100 * frac:.1f} % also in the reference; runs >= 6: {runs}")


if __name__ == "__main__":
    main()
