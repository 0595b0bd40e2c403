"""Per-kernel register / LDS / scratch usage from a hipcc --save-temps assembly file (.s): python tools/kernel_resources.py file.s [filter]"""
import re, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for blk in txt.split('  - .agpr_count:')[1:]:
    f = dict(re.findall(r'\.(\w+):\s+(\S+)', blk))
    if flt in f.get('name', ''):
        print(f"{f.get('name','?')[:90]:90s} vgpr {f.get('vgpr_count')} spill {f.get('vgpr_spill_count')} sgpr {f.get('sgpr_count')} "
              f"lds {f.get('group_segment_fixed_size')} scratch {f.get('private_segment_fixed_size')}")
