#!/usr/bin/env python3
"""Development tool (CPU): the gfx950 ISA of ONE kernel of a built library, plus a static census of its instruction stream.

  python tools/dev/isa.py [--lib path] [--kernel '0,1,2,2,1,0'] [--dump out.s]

--kernel: the template arguments of rollout_cost_kernel (COST, FAST, NOISE, R, VARIANT, INTEG).  Prints, per basic block
(label to label), the number of VALU / SALU / memory / branch instructions, so that the per-substep and per-control-step
instruction counts can be read without a GPU (SQ_INSTS_VALU of a launch = sum over blocks of count x executions)."""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import code_objects as CO  # noqa: E402


def disassemble(lib, mangled):
    for elf in CO.code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".elf") as f:
            f.write(elf)
            f.flush()
            r = subprocess.run([os.path.join(CO.LLVM_BIN, "llvm-objdump"), "-d", "--no-show-raw-insn", f"--disassemble-symbols={mangled}",
                                f.name], capture_output=True, text=True)
            if mangled + ">:" in r.stdout:
                return r.stdout
    raise SystemExit(f"kernel {mangled} not found in {lib}")


def census(text):
    """Basic blocks by branch targets: -> list of dicts (start offset, counts, the branch that ends the block and its target)."""
    ins = []                                   # (offset, opcode, target offset or None)
    base = None
    for line in text.splitlines():
        m = re.match(r"^\s+([a-z_0-9]+)\b.*//\s*([0-9A-Fa-f]+):", line)
        if not m:
            continue
        off = int(m.group(2), 16)
        if base is None:
            base = off
        t = re.search(r"\+0x([0-9a-f]+)>\s*$", line)
        ins.append((off - base, m.group(1), int(t.group(1), 16) if (t and m.group(1).startswith(("s_cbranch", "s_branch"))) else None))
    leaders = {0}
    for k, (off, op, tgt) in enumerate(ins):
        if tgt is not None:
            leaders.add(tgt)
            if k + 1 < len(ins):
                leaders.add(ins[k + 1][0])
    blocks, cur = [], None
    for off, op, tgt in ins:
        if off in leaders:
            cur = {"label": f"+0x{off:x}", "start": off, "valu": 0, "pk": 0, "trans": 0, "salu": 0, "mem": 0, "lds": 0, "branch": 0,
                   "other": 0, "n": 0, "exit": ""}
            blocks.append(cur)
        cur["n"] += 1
        if op.startswith("v_"):
            cur["valu"] += 1
            cur["pk"] += op.startswith("v_pk_")
            cur["trans"] += bool(re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_", op))
        elif op.startswith("s_cbranch") or op == "s_branch":
            cur["branch"] += 1
            cur["exit"] = f"{op} -> +0x{tgt:x}" + ("  (BACK EDGE)" if tgt <= off else "")
        elif op.startswith(("s_load", "global_", "buffer_", "flat_", "scratch_")):
            cur["mem"] += 1
        elif op.startswith("ds_"):
            cur["lds"] += 1
        elif op.startswith("s_"):
            cur["salu"] += 1
        else:
            cur["other"] += 1
    return blocks


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.environ.get("CPMPPI_LIB") or os.path.join(ROOT, "cartpolesimulation_amd", "libcpmppi.so"))
    ap.add_argument("--kernel", default="0,1,2,2,1,0")
    ap.add_argument("--dump", default=None)
    ap.add_argument("--min", type=int, default=8, help="blocks with fewer instructions are summed into one line")
    a = ap.parse_args()
    c, fast, noise, r, v, integ = (int(x) for x in a.kernel.split(","))
    mangled = f"_ZN8cpmppi_k19rollout_cost_kernelILi{c}ELb{fast}ELi{noise}ELi{r}ELi{v}ELi{integ}EEEvN6cpmppi6ParamsENS_8StepPtrsE"
    text = disassemble(a.lib, mangled)
    if a.dump:
        open(a.dump, "w").write(text)
    bl = census(text)
    tot = {k: sum(b[k] for b in bl) for k in ("n", "valu", "pk", "trans", "salu", "mem", "lds", "branch")}
    print(f"{mangled}: {len(bl)} blocks, {tot}")
    small = {k: 0 for k in tot}
    for b in bl:
        if b["n"] < a.min:
            for k in small:
                small[k] += b[k]
            continue
        print(f"  {b['label']:>9s} n {b['n']:4d}  valu {b['valu']:4d} (pk {b['pk']:3d}, trans {b['trans']:2d})  salu {b['salu']:3d}  mem {b['mem']:2d}  "
              f"lds {b['lds']:2d}  {b['exit']}")
    print(f"  (blocks under {a.min} instructions together: {small})")
