#!/usr/bin/env python3
"""What the loops of a gfx950 kernel are made of, from its disassembly:
    tools/kernel_resources.sh obj.o name-regex out.s && tools/loop_census.py out.s name-regex
For every backward branch of every matching kernel: the instruction count of the loop body [target, branch] and how many
of them are scratch loads / stores (VGPR spills), v_readlane / v_writelane (SGPR spills), v_accvgpr moves (AGPR
copies), MFMAs, LDS, global / flat memory, other VALU and SALU.  Nested loops are listed each with their own body."""
import re
import sys

CLASSES = (("scratch_load", r"^scratch_load"), ("scratch_store", r"^scratch_store"), ("readlane", r"^v_readlane"),
           ("writelane", r"^v_writelane"), ("accvgpr", r"^v_accvgpr"), ("mfma", r"^v_mfma"), ("lds", r"^ds_"),
           ("global", r"^(global_|flat_|buffer_)"), ("barrier", r"^s_barrier"), ("waitcnt", r"^s_waitcnt"),
           ("valu", r"^v_"), ("salu", r"^s_"))


def census(lines):
    out = dict.fromkeys([c for c, _ in CLASSES] + ["total"], 0)
    for _, ins in lines:
        out["total"] += 1
        for name, pat in CLASSES:
            if re.match(pat, ins):
                out[name] += 1
                break
    return out


def main():
    path, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else ".")
    kernels, cur = {}, None
    for ln in open(path):
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
        if m:
            cur = kernels.setdefault(m.group(1), [])
            continue
        m = re.match(r"^\s+(\S.*?)\s*//\s*([0-9A-Fa-f]+):", ln)
        if m and cur is not None:
            cur.append((int(m.group(2), 16), m.group(1).strip()))
    for name, ins in kernels.items():
        if not re.search(pat, name) or not ins:
            continue
        print(f"== {name}: {len(ins)} instructions")
        whole = census(ins)
        print("   whole kernel: " + " ".join(f"{k}={v}" for k, v in whole.items() if v))
        addr_index = {a: i for i, (a, _) in enumerate(ins)}
        for i, (addr, text) in enumerate(ins):
            m = re.match(r"^s_cbranch\S*\s+(\d+)|^s_branch\s+(\d+)", text)
            if not m:
                continue
            # the disassembler prints the simm16 word offset; the comment form "<name+0x...>" is not always there
            off = int(m.group(1) or m.group(2))
            if off >= 0x8000:
                off -= 0x10000
            tgt = addr + 4 + 4 * off
            if tgt <= addr and tgt in addr_index:
                body = ins[addr_index[tgt]:i + 1]
                c = census(body)
                print(f"   loop {tgt:#x}..{addr:#x}: " + " ".join(f"{k}={v}" for k, v in c.items() if v))


if __name__ == "__main__":
    main()
