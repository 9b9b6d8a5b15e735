#!/usr/bin/env python
"""tools/isa_walk.py cq_lines.s [decisions.json] — walks ONE path through a line-annotated gfx950 disassembly
(llvm-objdump -d -l of a -gline-tables-only build, one kernel) and histograms the instructions on it by source line.

The path starts at the kernel entry; at every conditional branch the walker takes the direction named in the decisions
file ({"<hex address>": "taken" | "fall"}), else a default (exec-mask skips fall through = the guarded block runs; exec
loop back-edges are not taken; uniform branches: prompt in the report as UNDECIDED and fall through).  Used to read the
quiet NS = 1 iteration of the streaming kernel: which source lines the ~430 VALU instructions per quad
come from."""
import collections
import json
import re
import sys

ins_re = re.compile(r'^\t(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):')
tgt_re = re.compile(r'<[^>]*\+0x([0-9a-f]+)>')


def parse(path):
    ins, line, base = [], None, None
    for ln in open(path):
        if ln.startswith('; /'):
            line = ln[2:].strip()
            line = re.sub(r'^.*/csrc/\./', '', line)
            line = re.sub(r'^/opt/rocm[^ ]*/include/', 'rocm:', line)
            continue
        m = re.match(r'^([0-9a-f]+) <', ln)
        if m:
            base = int(m.group(1), 16)
            continue
        m = ins_re.match(ln)
        if m:
            addr = int(m.group(3), 16)
            t = tgt_re.search(ln)
            ins.append({'op': m.group(1), 'args': m.group(2), 'addr': addr, 'line': line,
                        'target': base + int(t.group(1), 16) if (t and m.group(1).startswith(('s_cbranch', 's_branch'))) else None})
    return ins


def kind(op):
    if op.startswith('v_'):
        return 'valu'
    if op.startswith(('buffer_', 'global_', 'flat_', 'scratch_')):
        return 'vmem'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith('s_load') or op.startswith('s_buffer_load'):
        return 'smem'
    if op.startswith('s_'):
        return 'salu'
    return 'other'


def main():
    ins = parse(sys.argv[1])
    dec = json.load(open(sys.argv[2])) if len(sys.argv) > 2 else {}
    start = dec.get('start')
    stop = set(int(a, 16) for a in dec.get('stop', []))
    by_addr = {i['addr']: k for k, i in enumerate(ins)}
    pc = by_addr[int(start, 16)] if start else 0
    hist = collections.defaultdict(lambda: collections.Counter())
    total = collections.Counter()
    seen_back = collections.Counter()
    ops = collections.Counter()
    log = []
    steps = 0
    undecided = 0
    trace = open(dec['trace'], 'w') if dec.get('trace') else None
    while pc < len(ins) and steps < 200000:
        i = ins[pc]
        steps += 1
        if i['addr'] in stop and steps > 1:
            log.append(f"stop at {i['addr']:x}")
            break
        k = kind(i['op'])
        hist[i['line']][k] += 1
        if trace is not None:
            trace.write(f"{i['addr']:x}\t{i['line']}\t{i['op']} {i['args']}\n")
        ops[i['op'].replace('_e32','').replace('_e64','')] += 1
        total[k] += 1
        op = i['op']
        if op == 's_endpgm':
            break
        if op == 's_branch':
            pc = by_addr[i['target']]
            continue
        if op.startswith('s_cbranch'):
            key = f"{i['addr']:x}"
            back = i['target'] <= i['addr']
            if key in dec:
                d = dec[key]
                if isinstance(d, list):                 # e.g. ["taken", "fall"]: successive visits
                    d = d[min(seen_back[key], len(d) - 1)]
                seen_back[key] += 1
                take = d == 'taken'
                how = 'given'
            elif 'exec' in op:
                take = False
                how = 'default-exec'
            else:
                take = False
                how = 'UNDECIDED'
            nxt = ins[pc + 1]
            tg = ins[by_addr[i['target']]]
            if how != 'default-exec' or dec.get('verbose'):
                log.append(f"{key} {op:<18} {how:<12} {'TAKEN' if take else 'fall '} at {i['line']}  | fall-> {nxt['line']}  | target {i['target']:x}{' (back)' if back else ''} -> {tg['line']}")
            if how == 'UNDECIDED':
                undecided += 1
                if undecided >= int(dec.get('max_undecided', 12)):
                    break
            pc = by_addr[i['target']] if take else pc + 1
            continue
        pc += 1
    print('\n'.join(log))
    print('\nTOTAL', dict(total))
    print('OPS', ', '.join(f'{k} {v}' for k, v in ops.most_common(int(dec.get('top_ops', 60)))))
    rows = sorted(hist.items(), key=lambda kv: -sum(kv[1].values()))
    for line, c in rows[:int(dec.get('top', 70))]:
        print(f"{sum(c.values()):5d}  valu {c['valu']:4d} salu {c['salu']:4d} vmem {c['vmem']:3d} lds {c['lds']:3d}  {line}")


if __name__ == '__main__':
    main()
