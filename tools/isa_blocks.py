"""Basic blocks of one kernel in a hipcc -S listing with their instruction mix (VALU / SALU / LDS / VMEM / scratch), largest first.
Usage: python tools/isa_blocks.py listing.s kernel-name-substring [min_instructions]"""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
minins = int(sys.argv[3]) if len(sys.argv) > 3 else 40
m = [x for x in re.finditer(r'^(\S+):\s*; @', s, re.M) if pat in x.group(1)]
if not m: sys.exit("kernel not found")
i = m[0].start(); j = s.find('.Lfunc_end', i)
blocks, cur, name = [], [], 'entry'
for l in s[i:j].splitlines()[1:]:
    t = l.split(';')[0].strip()
    if not t or t.startswith('.') and not t.endswith(':'): continue
    if t.endswith(':'):
        blocks.append((name, cur)); cur, name = [], t[:-1]; continue
    cur.append(t)
blocks.append((name, cur))
def cls(op):
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_'): return 'salu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith('scratch_'): return 'scratch'
    if op.startswith(('global_', 'flat_', 'buffer_')): return 'vmem'
    return 'other'
print("kernel instructions:", sum(len(b) for _, b in blocks))
for name, b in sorted(blocks, key=lambda x: -len(x[1])):
    if len(b) < minins: break
    c = {}
    for t in b: c[cls(t.split()[0])] = c.get(cls(t.split()[0]), 0) + 1
    print(f"{name:16s} {len(b):5d}  " + "  ".join(f"{k} {v}" for k, v in sorted(c.items())))
