"""Registers, scratch and LDS of every kernel in a gfx950 assembly listing (hipcc ... -save-temps=obj -> *-gfx950.s, or
`llvm-objdump`-free: reads the amdhsa.kernels metadata).  Usage: python tools/kernel_resources.py file.s [name filter ...]"""
import re
import subprocess
import sys


def main(path, *filters):
    s = open(path).read()
    md = s[s.index('amdhsa.kernels:'):]
    for k in md.split('  - .agpr_count:')[1:]:
        name = re.search(r'\.name:\s+(\S+)', k).group(1)
        if filters and not any(f in name for f in filters):
            continue
        g = lambda key: int(re.search(r'\.%s:\s+(\d+)' % key, k).group(1))
        try:
            dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
        except OSError:
            dn = name
        print("%-100s vgpr %3d agpr %3d sgpr %3d scratch %5d lds %6d spill %d" % (dn[:100], g('vgpr_count'), int(k.split('\n')[0]), g('sgpr_count'),
              g('private_segment_fixed_size'), g('group_segment_fixed_size'), g('vgpr_spill_count')))


if __name__ == "__main__":
    main(*sys.argv[1:])
