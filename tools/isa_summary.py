"""Build helper: one-line picture of a kernel's instruction stream out of hipcc -S output.
usage: isa_summary.py FILE.s KERNEL_SUBSTRING   (M mfma, r ds_read, W ds_write, D buffer/global load, T store, L/S scratch load/store,
B branch, | barrier, w s_waitcnt, z s_sleep, v other vector, s other scalar)"""
import sys
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = [i for i, l in enumerate(lines) if l.startswith('_Z') and key in l and l.split(';')[0].rstrip().endswith(':')][0]
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('.Lfunc_end'))
body = lines[start:end]
out = []
for l in body:
    t = l.strip().split(' ')[0] if l.strip() else ''
    if not t or t.startswith('.') or t.startswith(';') or t.endswith(':'):
        if t.endswith(':') and t.startswith('.LBB'):
            out.append('\n' + t + ' ')
        continue
    c = ('M' if t.startswith('v_mfma') else 'L' if t.startswith('scratch_load') else 'S' if t.startswith('scratch_store') else
         'r' if t.startswith('ds_read') or t.startswith('ds_load') else 'W' if t.startswith('ds_write') or t.startswith('ds_store') else
         'D' if t.startswith('buffer_load') or t.startswith('global_load') else 'T' if t.startswith('buffer_store') or t.startswith('global_store') else
         'B' if t.startswith('s_cbranch') or t.startswith('s_branch') else '|' if t.startswith('s_barrier') else 'w' if t.startswith('s_waitcnt') else
         'z' if t.startswith('s_sleep') else 'A' if 'atomic' in t else 'v' if t.startswith('v_') else 's')
    out.append(c)
print(''.join(out))
