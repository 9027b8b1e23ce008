#!/usr/bin/env python3
"""INTEGRATION.md section B quotes mb-istft-vits_amd/reference_binding.py verbatim: re-insert the file's current text
between the section's ```python fence and its closing fence.  usage: python scripts/sync_integration_md.py [--check]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
md_path = os.path.join(ROOT, "INTEGRATION.md")
src = open(os.path.join(ROOT, "mb-istft-vits_amd", "reference_binding.py")).read()
md = open(md_path).read()
head = md.index("## B. Keeping the reference's `models.py`")
a = md.index("```python\n", head) + len("```python\n")
b = md.index("\n```\n", a)
new = md[:a] + src.rstrip("\n") + md[b:]
if "--check" in sys.argv:
    sys.exit(0 if new == md else 1)
open(md_path, "w").write(new)
print("INTEGRATION.md section B: %d bytes of reference_binding.py" % len(src))
