import csv, glob, sys, collections
d, log = sys.argv[1], sys.argv[2]
names = [l.split()[1:] for l in open(log) if l.startswith('VARIANTS')][0]
f = glob.glob(f'{d}/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'conv3x3_ws' in r['Kernel_Name'] or 'conv3x3_tile_f16x3' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
acc = collections.defaultdict(list)
for i, r in enumerate(rows):
    v = names[(i // 2) % len(names)]
    acc[(v, 'fwd' if i % 2 == 0 else 'dgrad')].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for v in names:
    f_, d_ = acc[(v, 'fwd')], acc[(v, 'dgrad')]
    print(f'{v:14s} fwd {sum(f_[2:]) / len(f_[2:]):7.1f} us   dgrad {sum(d_[2:]) / len(d_[2:]):7.1f} us   (rounds 3..{len(f_)})')
