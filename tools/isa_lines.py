"""Developer tool (CPU): static instructions per source line of one kernel, from a `hipcc -S -gline-tables-only` listing.
`python tools/isa_lines.py listing.s solve_kernelIfLi6 [top]` -- where the code of a long kernel goes before any GPU time is spent on it."""
import collections
import re
import sys


def main():
    text = open(sys.argv[1]).read()
    key = sys.argv[2]
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    m = re.search(r"^(_Z\w*%s\w*):" % re.escape(key), text, re.M)
    body = text[m.end():text.index(".size\t" + m.group(1), m.end())]
    files = dict(re.findall(r'\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', text)) or dict(re.findall(r'\.file\s+(\d+)\s+"([^"]+)"', text))
    per = collections.Counter()
    valu = collections.Counter()
    loc = ("?", 0)
    for l in body.split("\n"):
        mm = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if mm:
            loc = (mm.group(1), int(mm.group(2)))
            continue
        t = l.strip()
        if not l.startswith("\t") or not t or t.startswith((".", ";")):
            continue
        per[loc] += 1
        if t.startswith("v_"):
            valu[loc] += 1
    total = sum(per.values())
    print(f"{key}: {total} instructions, {sum(valu.values())} VALU")
    for (f, ln), c in per.most_common(top):
        print(f"  {c:6d} ({valu[(f, ln)]:5d} VALU)  {files.get(f, f).split('/')[-1]}:{ln}")


if __name__ == "__main__":
    main()
