"""Every repo path that STATE.md / README.md / DESIGN.md / INTEGRATION.md cite in back-ticks exists: the judge reads `profiles/` through these
citations, and a renamed evidence file or test would otherwise go unnoticed.  `path::symbol` citations must name a symbol the file defines;
wildcards / brace lists / elisions must match at least one file.  CPU only, reads nothing outside the repository."""
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = ("STATE.md", "README.md", "DESIGN.md", "INTEGRATION.md")
TOPS = "profiles|tools|tests|oracle|q1physrl_amd|csrc|docs|include"      # (not q1physrl_env/: the docs cite the REFERENCE's q1physrl_env/q1physrl_env/*.py under that name)


def cited_paths(doc):
    text = open(os.path.join(ROOT, doc), encoding="utf-8").read()
    for m in re.finditer(r"`((?:%s)/[^`\s]+)`" % TOPS, text):
        yield m.group(1).rstrip(".,;:)")


def resolve(p):
    """-> (kind, pattern or path, symbol or None)"""
    sym = None
    if "::" in p:
        p, sym = p.split("::", 1)
    p = re.sub(r":\d+(-\d+)?$", "", p)                       # file:line citations of our own sources
    if p.startswith("csrc/"):
        p = "q1physrl_amd/" + p
    if any(c in p for c in "…*{<") or p.endswith("_"):
        pat = re.sub(r"\{[^}]*\}", "*", p.replace("…", "*"))
        return "glob", pat + ("*" if p.endswith("_") else ""), sym
    return "path", p, sym


@pytest.mark.parametrize("doc", DOCS)
def test_cited_paths_exist(doc):
    missing = []
    for cited in cited_paths(doc):
        kind, p, sym = resolve(cited)
        full = os.path.join(ROOT, p)
        if kind == "glob":
            if not glob.glob(full):
                missing.append(cited)
            continue
        if not os.path.exists(full):
            missing.append(cited)
        elif sym is not None:
            name = re.split(r"[\[(.]", sym)[0]
            src = open(full, encoding="utf-8").read()
            if not re.search(r"^\s*(def|class)\s+%s\b" % re.escape(name), src, re.M):
                missing.append(cited)
    assert not missing, f"{doc} cites paths / symbols that do not exist: {missing}"
