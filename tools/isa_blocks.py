#!/usr/bin/env python3
"""usage: isa_blocks.py file.s <kernel-name-substring> [min_lds_or_mfma_ops]
Lists the basic blocks of one kernel of a hipcc -S listing that hold LDS / MFMA work, with their
instruction mix (VALU / SALU / LDS / VMEM counts): what an inner loop really costs per trip."""
import re
import sys
from collections import Counter


def blocks_of(lines, key):
    starts = [i for i, l in enumerate(lines) if l.startswith('_ZN') and key in l and ':' in l.split(';')[0]]
    i0 = starts[0]
    end = next(i for i in range(i0, len(lines)) if lines[i].startswith('.Lfunc_end'))
    blocks, cur, name = [], [], 'entry'
    for l in lines[i0 + 1:end]:
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            blocks.append((name, cur))
            cur, name = [], m.group(1)
        else:
            t = l.strip()
            if t and not t.startswith(';') and not t.startswith('.'):
                cur.append(t)
    blocks.append((name, cur))
    return blocks


def main():
    lines = open(sys.argv[1]).read().split('\n')
    key = sys.argv[2]
    minops = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    bl = blocks_of(lines, key)
    print(key, 'blocks', len(bl), 'instructions', sum(len(b) for _, b in bl))
    for n, b in bl:
        c = Counter(x.split()[0] for x in b)
        lds = sum(k for i, k in c.items() if i.startswith('ds_'))
        mf = sum(k for i, k in c.items() if i.startswith('v_mfma'))
        if lds + mf >= minops:
            v = sum(k for i, k in c.items() if i.startswith('v_') and not i.startswith('v_mfma'))
            sa = sum(k for i, k in c.items() if i.startswith('s_'))
            vm = sum(k for i, k in c.items() if i.startswith(('global_', 'buffer_', 'flat_')))
            print(f'  {n}: {len(b)} instr  valu {v}  mfma {mf}  lds {lds}  salu {sa}  vmem {vm}')
            print('     ', sorted(c.items(), key=lambda kv: -kv[1])[:48])


if __name__ == '__main__':
    main()
