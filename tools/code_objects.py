#!/usr/bin/env python3
"""Kernel resource figures straight from a BUILT library (CPU only): every gfx950 code object embedded in the .so's
`.hip_fatbin` section is unbundled and its AMDGPU metadata note read with llvm-readelf.

  python tools/code_objects.py [path/to/libcpmppi.so] [name-substring]

`kernels(path)` -> list of dicts {name, vgpr_count, agpr_count, sgpr_count, sgpr_spill_count, vgpr_spill_count,
private_segment_fixed_size (scratch bytes per lane), group_segment_fixed_size (static LDS), unit (index of the code object)}.
Used by tests/test_abi_and_host.py to keep scratch out of the rollout kernels (round 1 lost 4-15 % to a 28-byte slot)."""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM_BIN = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
INT_KEYS = ("vgpr_count", "agpr_count", "sgpr_count", "sgpr_spill_count", "vgpr_spill_count", "private_segment_fixed_size",
            "group_segment_fixed_size", "kernarg_segment_size")


def code_objects(so_path, target="gfx950"):
    """The device ELF images for `target` inside so_path's .hip_fatbin (one per translation unit)."""
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([os.path.join(LLVM_BIN, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", so_path,
                        os.path.join(tmp, "discard.so")], check=True)
        data = open(fat, "rb").read()
    out, pos = [], 0
    while True:
        i = data.find(MAGIC, pos)
        if i < 0:
            return out
        n = struct.unpack_from("<Q", data, i + len(MAGIC))[0]
        off = i + len(MAGIC) + 8
        for _ in range(n):
            o, size, tl = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + tl].decode()
            off += tl
            if target in triple and size:
                out.append(data[i + o:i + o + size])
        pos = i + 1


def kernels(so_path, target="gfx950"):
    res = []
    for unit, elf in enumerate(code_objects(so_path, target)):
        with tempfile.NamedTemporaryFile(suffix=".elf") as f:
            f.write(elf)
            f.flush()
            txt = subprocess.run([os.path.join(LLVM_BIN, "llvm-readelf"), "--notes", f.name], check=True,
                                 capture_output=True, text=True).stdout
        if "amdhsa.kernels:" not in txt:
            continue
        for blk in re.split(r"\n  - \.agpr_count:", txt[txt.index("amdhsa.kernels:"):])[1:]:
            blk = ".agpr_count:" + blk
            k = {"unit": unit, "name": re.search(r"\.name:\s+(\S+)", blk).group(1)}
            for key in INT_KEYS:
                m = re.search(r"\." + key + r":\s+(\d+)", blk)
                k[key] = int(m.group(1)) if m else None
            res.append(k)
    return res


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] else os.path.join(ROOT, "cartpolesimulation_amd", "libcpmppi.so")
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for k in kernels(path):
        if flt in k["name"]:
            print(f"u{k['unit']} {k['name'][:100]:100s} vgpr {k['vgpr_count']:>3} agpr {k['agpr_count']:>3} sgpr {k['sgpr_count']:>3} "
                  f"sgpr_spill {k['sgpr_spill_count']:>3} vgpr_spill {k['vgpr_spill_count']:>2} "
                  f"scratch {k['private_segment_fixed_size']:>3} lds {k['group_segment_fixed_size']:>6}")
