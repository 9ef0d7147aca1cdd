#!/usr/bin/env python3
"""Reads a PAG_WALK_TRACE=1 log (the last walk in it): cost per classification of the first-round segment jobs by the time they began."""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
st = [i for i, ln in enumerate(lines) if ln.startswith("[trace] walks")]
body = lines[st[-1]:]
post, b = {}, collections.defaultdict(lambda: [0, 0.0, 0, 0])
for ln in body:
    m = re.match(r"\[trace\] post t=([\d.]+) ctg (\d+) (seg|chain) (-?\d+) ring (\d+) mode (\d+) init (\d+)", ln)
    if m:
        post.setdefault((int(m.group(2)), m.group(3), int(m.group(4))), int(m.group(6)))
    m = re.match(r"\[trace\] done t=([\d.]+) ctg (\d+) (seg|chain) (-?\d+) dev ([\d.]+)\.\.([\d.]+) len (\d+) classify (\d+)", ln)
    if m and m.group(3) == "seg" and post.get((int(m.group(2)), "seg", int(m.group(4)))) == 1:
        s = b[int(float(m.group(5)) // 10) * 10]
        s[0] += 1
        s[1] += float(m.group(6)) - float(m.group(5))
        s[2] += int(m.group(8))
print("us per classification by begin time:", " ".join(f"{t}:{1000 * s[1] / max(1, s[2]):.2f}({s[0]})" for t, s in sorted(b.items())))
