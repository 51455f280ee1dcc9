"""Not a test: rewrites the code blocks of INTEGRATION.md that show source files verbatim (between `<!-- BEGIN path -->` and
`<!-- END path -->`) from the files themselves.  `--check`: exit 1 if a block differs from its file.  python tools/sync_integration.py"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(ROOT, "INTEGRATION.md")
text = open(path).read()


def block(m):
    src = open(os.path.join(ROOT, m.group(1))).read().rstrip("\n")
    return "<!-- BEGIN %s -->\n```cpp\n%s\n```\n<!-- END %s -->" % (m.group(1), src, m.group(1))


new = re.sub(r"<!-- BEGIN (\S+) -->\n.*?<!-- END \1 -->", block, text, flags=re.S)
if "--check" in sys.argv:
    sys.exit(0 if new == text else 1)
if new != text:
    open(path, "w").write(new)
    print("INTEGRATION.md updated")
