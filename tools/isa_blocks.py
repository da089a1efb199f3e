"""Histogram of the instructions of one basic block of a hipcc -S listing: python tools/isa_blocks.py file.s .LBB16_171"""
import re, sys, collections
s = open(sys.argv[1]).read()
lab = sys.argv[2]
i = s.index('\n' + lab + ':')
m = re.search(r'\n\.LBB\d+_\d+:|\n\.Lfunc_end', s[i + 5:])
b = s[i:i + 5 + m.start()]
c = collections.Counter(l.split()[0] for l in b.split('\n')[1:] if l.strip() and not l.strip().startswith(';'))
for k, v in c.most_common(25):
    print(v, k)
if len(sys.argv) > 3:
    print(b)
