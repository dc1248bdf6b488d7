"""Developer tool: instruction mix per basic block of one kernel in a hipcc -S listing.
    python tools/isa_mix.py file.s <substring of the mangled kernel name> [min block size]"""
import re
import sys


def classify(ins):
    op = ins.split()[0]
    if op.startswith('v_mfma'):
        return 'mfma'
    if op.startswith('v_'):
        if any(op.startswith(x) for x in ('v_exp', 'v_log', 'v_rcp', 'v_rsq', 'v_sqrt', 'v_sin', 'v_cos')):
            return 'trans'
        if 'dpp' in ins or op.startswith('v_permlane') or op.startswith('v_readlane') or op.startswith('v_readfirstlane'):
            return 'xlane'
        return 'valu'
    if op.startswith('ds_bpermute') or op.startswith('ds_swizzle') or op.startswith('ds_permute'):
        return 'ds_perm'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'vmem'
    if op.startswith('s_waitcnt'):
        return 'wait'
    if op.startswith('s_barrier'):
        return 'barrier'
    if op.startswith('s_'):
        return 'salu'
    return 'other'


def main():
    path, key = sys.argv[1], sys.argv[2]
    minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    L = open(path).read().split('\n')
    start = [i for i, l in enumerate(L) if key in l and l.rstrip().split(' ')[0].endswith(':') and l.startswith('_Z')][0]
    end = [i for i in range(start, len(L)) if L[i].startswith('.Lfunc_end')][0]
    blocks, cur = [], ('entry', [])
    for l in L[start:end]:
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            blocks.append(cur)
            cur = (m.group(1), [])
        elif l.startswith('\t') and not l.startswith('\t.') and not l.strip().startswith(';'):
            cur[1].append(l.strip())
    blocks.append(cur)
    tot = {}
    for name, ins in blocks:
        c = {}
        for i in ins:
            k = classify(i)
            c[k] = c.get(k, 0) + 1
            tot[k] = tot.get(k, 0) + 1
        if len(ins) >= minsz:
            print(f'{name:12s} {len(ins):5d} ', ' '.join(f'{k}={v}' for k, v in sorted(c.items())))
    print('total', tot)


if __name__ == '__main__':
    main()
