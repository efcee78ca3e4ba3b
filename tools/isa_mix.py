"""Instruction mix of the kernels in a hipcc -S --cuda-device-only listing: python tools/isa_mix.py file.s <substring of the kernel name> [top]"""
import re
import sys
from collections import Counter

text = open(sys.argv[1]).read()
pat = sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)\n\s*s_endpgm", text, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if pat not in name:
        continue
    c = Counter()
    for line in body.split("\n"):
        line = line.strip()
        if not line or line[0] in ".;/" or line.endswith(":"):
            continue
        c[line.split()[0]] += 1
    print(name, "total", sum(c.values()))
    for k, v in c.most_common(top):
        print("   ", k, v)
