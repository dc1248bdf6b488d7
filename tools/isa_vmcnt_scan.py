"""Developer tool: for every loop of every kernel in hipcc -S listings, count global stores, global loads, MFMAs and
`s_waitcnt vmcnt` waits, and print the loops that both store and drain the counter (`vmcnt(0)`).  On gfx9 loads and stores share
one in-order counter: a drain — or any wait for a load — inside a loop that also stores waits for the stores' acknowledgement.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip -S --cuda-device-only csrc/<file>.hip -o <file>.s
    python tools/isa_vmcnt_scan.py <file>.s ..."""
import re, sys
# for every kernel: inside loops (between a "Loop Header" label and the backward branch), report vmcnt(0) waits and whether the loop has global stores
for f in sys.argv[1:]:
    L = open(f).read().split('\n')
    i = 0
    kern = None
    while i < len(L):
        l = L[i]
        if l.startswith('_Z') and l.rstrip().split(' ')[0].endswith(':'):
            kern = l.split(':')[0]
            # collect function body
            j = i
            while j < len(L) and not L[j].startswith('.Lfunc_end'):
                j += 1
            body = L[i:j]
            # find loops: label lines with 'Loop Header', and 'in Loop: Header=' membership
            hdrs = {}
            cur_blk = None
            blk_loop = {}
            for k, b in enumerate(body):
                m = re.match(r'^(\.LBB\d+_\d+):\s*;\s*(.*)$', b)
                if m:
                    cur_blk = m.group(1)
                    info = m.group(2)
                    if 'Loop Header' in info:
                        hdrs[cur_blk] = dict(stores=0, loads=0, vm0=0, vmn=0, mfma=0, depth=info)
                        blk_loop[cur_blk] = cur_blk
                    mm = re.search(r'in Loop: Header=(BB\d+_\d+)', info)
                    if mm:
                        blk_loop[cur_blk] = '.L' + mm.group(1)
                    continue
                if re.match(r'^(\.LBB\d+_\d+):', b):
                    cur_blk = b.split(':')[0]
                    continue
                lp = blk_loop.get(cur_blk)
                if lp in hdrs:
                    t = b.strip()
                    if t.startswith('global_store') or t.startswith('buffer_store'): hdrs[lp]['stores'] += 1
                    if t.startswith('global_load') or t.startswith('buffer_load'): hdrs[lp]['loads'] += 1
                    if t.startswith('v_mfma'): hdrs[lp]['mfma'] += 1
                    if t.startswith('s_waitcnt') and 'vmcnt(0)' in t: hdrs[lp]['vm0'] += 1
                    elif t.startswith('s_waitcnt') and 'vmcnt' in t: hdrs[lp]['vmn'] += 1
            for h, d in hdrs.items():
                if d['stores'] and d['vm0']:
                    print(f"{f:14s} {kern[:70]:70s} {h:12s} stores={d['stores']:3d} loads={d['loads']:3d} mfma={d['mfma']:4d} vmcnt(0)={d['vm0']} vmcnt(n)={d['vmn']}")
            i = j
        i += 1
