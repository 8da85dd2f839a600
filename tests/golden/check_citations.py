"""Build-container check (the reference checkout does not travel): every `path/file.py:LINE[-LINE]` citation of the
reference in the headers, docs and package docstrings points at an existing file with at least that many lines.

    python -m pytest tests/golden/check_citations.py -q        (or: python tests/golden/check_citations.py)
"""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
PAT = re.compile(r"((?:Classification|DDPM|SD)/[\w\-/\.]+?\.(?:py|yml|yaml|md|txt)):(\d+)(?:-(\d+))?")


def _sources():
    yield os.path.join(ROOT, "include", "salun.h")
    for name in ("DESIGN.md", "INTEGRATION.md", "README.md"):
        yield os.path.join(ROOT, name)
    for base in ("unlearn_saliency_amd", "oracle"):
        for d, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".c")):
                    yield os.path.join(d, f)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
def test_reference_citations_resolve():
    line_counts = {}
    bad, seen = [], 0
    for src in _sources():
        text = open(src, errors="ignore").read()
        for m in PAT.finditer(text):
            rel, lo, hi = m.group(1), int(m.group(2)), int(m.group(3) or m.group(2))
            path = os.path.join(REF, rel)
            if rel not in line_counts:
                line_counts[rel] = sum(1 for _ in open(path, errors="ignore")) if os.path.isfile(path) else -1
            n = line_counts[rel]
            seen += 1
            if n < 0:
                bad.append(f"{os.path.relpath(src, ROOT)}: {rel} does not exist")
            elif not (1 <= lo <= hi <= n):
                bad.append(f"{os.path.relpath(src, ROOT)}: {rel}:{lo}-{hi} outside 1..{n}")
    assert seen > 50, "citation pattern found too few references"
    assert not bad, "\n".join(bad[:20])


if __name__ == "__main__":
    test_reference_citations_resolve()
    print("citations ok")
