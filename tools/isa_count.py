"""Static instruction-class counts per kernel of a hipcc -S (gfx950) listing: tools/isa_count.py file.s"""
import re, sys, collections
lines = open(sys.argv[1]).read().split('\n')
cur = None
data = {}
for line in lines:
    m = re.match(r'^(_Z\w+):', line)
    if m:
        cur = m.group(1); data[cur] = collections.Counter(); continue
    if cur is None:
        continue
    t = line.strip()
    if t.startswith('s_endpgm'):
        cur = None; continue
    if not t or t[0] in ';.' or t.endswith(':'):
        continue
    data[cur][t.split()[0]] += 1
TRANS = ('v_rcp_f32', 'v_rsq_f32', 'v_exp_f32', 'v_log_f32', 'v_sqrt_f32', 'v_sin_f32', 'v_cos_f32')
for name, c in data.items():
    g = collections.Counter()
    for op, n in c.items():
        if op.startswith('v_pk_'): g['v_pk'] += n
        elif op.startswith(TRANS): g['trans'] += n
        elif op.startswith('v_mfma'): g['mfma'] += n
        elif op.startswith('v_'): g['v_other'] += n
        elif op.startswith('ds_'): g['ds'] += n
        elif op.startswith('s_'): g['s'] += n
        elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): g['mem'] += n
        else: g[op] += n
    print(name[:70], sum(c.values()), dict(g))
    if len(sys.argv) > 2:
        print('   ', c.most_common(int(sys.argv[2])))
