#!/usr/bin/env python3
"""Development tool (CPU): register / LDS / scratch figures of every kernel in a --save-temps .s file.
  python tools/dev/kernel_regs.py build/isa/<unit>-hip-amdgcn-amd-amdhsa-gfx950.s [name-substring]"""
import re
import sys

s = open(sys.argv[1]).read()
md = s[s.index("amdhsa.kernels:"):]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in md.split("  - .agpr_count")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if flt not in name:
        continue
    g = lambda k: re.search(r"\." + k + r":\s+(\d+)", blk).group(1)  # noqa: E731
    print(f"{name[:90]:90s} vgpr {g('vgpr_count'):>3} sgpr {g('sgpr_count'):>3} sgpr_spill {g('sgpr_spill_count'):>3} "
          f"vgpr_spill {g('vgpr_spill_count'):>2} scratch {g('private_segment_fixed_size'):>3} lds {g('group_segment_fixed_size'):>6}")
