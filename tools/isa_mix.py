#!/usr/bin/env python3
"""Instruction mix of the hot loop of a kernel in an llvm-objdump -d listing.

usage: isa_mix.py listing.s [kernel-substring]
Finds the longest backward branch inside the (first matching) kernel and counts the instructions of that loop body
by issue class.  Cost model (tools/ubench/valu_rate.hip, op_rate.hip on MI355X): plain wave64 VALU = 1 unit,
v_pk_*_f32 = 2 units, transcendental = 4 units.
"""
import re, sys, collections

def main():
    path = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ''
    lines = open(path).read().split('\n')
    start = None
    for i, l in enumerate(lines):
        m = re.match(r'^([0-9a-f]+) <(.*)>:$', l)
        if m:
            if start is not None: end = i; break
            if pat in l: start = i
    else:
        end = len(lines)
    body = []
    for l in lines[start + 1:end]:
        m = re.match(r'^\s+(\S+)\s+(.*?)//\s*([0-9A-Fa-f]+):', l)
        if m: body.append((int(m.group(3), 16), m.group(1), m.group(2)))
    addr_idx = {a: i for i, (a, _, _) in enumerate(body)}
    best = None
    for i, (a, op, args) in enumerate(body):
        if op.startswith('s_cbranch') or op == 's_branch':
            off = int(args.split()[0])
            if off >= 32768:
                tgt = a + 4 + (off - 65536) * 4
                if tgt in addr_idx:
                    span = i - addr_idx[tgt]
                    if best is None or span > best[0]: best = (span, addr_idx[tgt], i)
    span, lo, hi = best
    cnt = collections.Counter(); ops = collections.Counter()
    for a, op, args in body[lo:hi + 1]:
        ops[op] += 1
        if op.startswith('v_pk_') and op.endswith('_f32'): cnt['valu_pk_f32'] += 1
        elif op.startswith(('v_exp', 'v_log', 'v_rcp', 'v_rsq', 'v_sqrt', 'v_sin', 'v_cos')): cnt['valu_trans'] += 1
        elif op.startswith('v_mfma'): cnt['mfma'] += 1
        elif op.startswith('v_'): cnt['valu_plain'] += 1
        elif op.startswith('ds_'): cnt['lds'] += 1
        elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): cnt['vmem'] += 1
        elif op.startswith('s_waitcnt'): cnt['waitcnt'] += 1
        elif op.startswith('s_'): cnt['salu'] += 1
        else: cnt['other'] += 1
    print(f'loop body: {span + 1} instructions')
    for k, v in sorted(cnt.items(), key=lambda kv: -kv[1]): print(f'  {k:14s} {v}')
    units = cnt['valu_plain'] + 2 * cnt['valu_pk_f32'] + 4 * cnt['valu_trans']
    print(f'  VALU issue units (plain=1, pk=2, trans=4): {units}')
    print('top ops:', ', '.join(f'{o}:{n}' for o, n in ops.most_common(28)))

main()
