"""Per-round kernel-time table of one DP layer from a rocprofv3 --kernel-trace CSV (bench.py --parts 3)."""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    n = n.replace('void ', '').replace('cpk::', '')
    m = re.match(r'(k_\w+)(<[^(]*>)?', n)
    if not m: return n[:24]
    name = m.group(1)
    if name == 'k_lpass_own' and m.group(2) and m.group(2).rstrip('>').endswith('true'): name = 'k_lpass_gap'
    return name
names = [short(r['Kernel_Name']) for r in rows]
idx = [i for i, n in enumerate(names) if n.startswith('k_combine')]
L = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # which layer: the one between the L-th and (L+1)-th combine
lo, hi = idx[L] + 1, idx[L + 1] + 1
rounds, cur = [], collections.OrderedDict()
for r, n in zip(rows[lo:hi], names[lo:hi]):
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if n.startswith('k_rpass') and any(k.startswith('k_setup_short') for k in cur):
        rounds.append(cur); cur = collections.OrderedDict()
    cur[n] = cur.get(n, 0) + d
rounds.append(cur)
keys = ['k_rpass_small', 'k_rpass_wave', 'k_setup_short', 'k_scan_reduce', 'k_scan_blocksums', 'k_scan_apply', 'k_own_map', 'k_lpass_own', 'k_lpass_gap',
        'k_gap_finish', 'k_gap_seg', 'k_gap_merge', 'k_fix_own_lane', 'k_fix_own', 'k_tile_t0', 'k_lpass', 'k_span_short', 'k_open', 'k_fix']
print('rd ' + ' '.join(f'{k[2:10]:>8}' for k in keys) + '    total')
tot = collections.Counter()
for i, c in enumerate(rounds):
    print(f'{i:2d} ' + ' '.join(f'{c.get(k, 0):8.0f}' for k in keys) + f' {sum(c.values()):8.0f}')
    for k, v in c.items(): tot[k] += v
print('sum', round(sum(tot.values())), [(k, round(v)) for k, v in tot.most_common()])
print('layer wall us', (int(rows[hi - 1]['End_Timestamp']) - int(rows[lo]['Start_Timestamp'])) / 1e3)
