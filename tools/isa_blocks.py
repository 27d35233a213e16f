#!/usr/bin/env python3
"""Instruction mix per basic block of one kernel in a hipcc -S listing.  usage: tools/isa_blocks.py file.s <mangled-name-prefix> [min_instrs]"""
import re
import sys
lines = open(sys.argv[1]).read().split('\n')
name = sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 12
start = next(i for i, l in enumerate(lines) if l.startswith(name) and l.split(';')[0].strip().endswith(':'))
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
blocks, cur = [], ('entry', [])
for ln in lines[start:end]:
    m = re.match(r'^(\.LBB\d+_\d+):', ln)
    if m:
        blocks.append(cur)
        cur = (m.group(1), [])
    elif ln.startswith('\t') and not ln.startswith('\t.') and not ln.strip().startswith(';'):
        cur[1].append(ln.strip())
blocks.append(cur)
tv = 0
for nm, ins in blocks:
    valu = [i for i in ins if i.startswith('v_')]
    tv += len(valu)
    if len(ins) < minn:
        continue
    print(f"{nm:10s} n={len(ins):4d} valu={len(valu):4d} trans={sum(i.startswith(('v_rcp', 'v_rsq', 'v_sqrt')) for i in valu):2d} "
          f"dpp={sum('dpp' in i for i in valu):2d} mov={sum(i.startswith(('v_mov', 'v_accvgpr')) for i in valu):3d} "
          f"cnd={sum(i.startswith('v_cndmask') for i in valu):3d} lds={sum(i.startswith('ds_') for i in ins):2d} "
          f"glob={sum(i.startswith('global_') for i in ins):2d} scratch={sum(i.startswith('scratch_') for i in ins):2d} "
          f"salu={sum(i.startswith('s_') for i in ins):3d} wait={sum(i.startswith('s_waitcnt') for i in ins):2d}")
print("total valu", tv, "blocks", len(blocks))
