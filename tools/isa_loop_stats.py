"""Static view of the persistent GEMM kernel's K loop in the gfx950 code object: instruction counts per class for the two load phases
and the two MFMA segments of one K tile.  The in-kernel clock measurements (profiles/r03_gemm_where_the_cycles_go.md) say the load phase
is instruction-issue time, so these counts are the off-GPU proxy for a schedule change (they need no GPU: hipcc cross-compiles).

    python tools/isa_loop_stats.py [--obj imagefolder_amd/csrc/_build/xq_gemm.o] [--match 'gemm_pring_kernelILi0ELi0ELi0ELb0E']

A K loop is recognised as four consecutive s_barrier with 16 v_mfma between the 1st / 2nd and the 3rd / 4th and none between the 2nd / 3rd,
closed by a backward branch behind the 4th; its first load phase is what lies between the branch target and the 1st barrier.
"""
import argparse
import os
import re
import shutil
import subprocess
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
CLASSES = [("mfma", r"^v_mfma"), ("valu", r"^v_"), ("ds_read", r"^ds_read"), ("ds_write", r"^ds_write"), ("lds_dma", r"^global_load_lds"),
           ("vmem", r"^(global|buffer|scratch|flat)_"), ("smem", r"^s_(load|memtime|buffer_load)"), ("waitcnt", r"^s_waitcnt"), ("nop", r"^s_nop"),
           ("branch", r"^s_(cbranch|branch)"), ("salu", r"^s_")]


def disassemble(obj):
    tmp = tempfile.mkdtemp()
    local = os.path.join(tmp, os.path.basename(obj))
    shutil.copy(obj, local)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, capture_output=True, cwd=tmp)
    co = [f for f in os.listdir(tmp) if f.endswith("gfx950")]
    assert co, "no gfx950 code object in " + obj
    return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", os.path.join(tmp, co[0])], check=True, capture_output=True, text=True).stdout


def kernels(text):
    cur, out = None, {}
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        if cur is None:
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):", line)
        if m:
            out[cur].append((int(m.group(3), 16), m.group(1), m.group(2)))
    return out


def classify(op):
    for name, pat in CLASSES:
        if re.match(pat, op):
            return name
    return "other"


def count(ins):
    c = {}
    for _, op, _ in ins:
        k = classify(op)
        c[k] = c.get(k, 0) + 1
    return c


def find_loop(ins):
    bars = [i for i, (_, op, _) in enumerate(ins) if op == "s_barrier"]
    nm = lambda a, b: sum(1 for _, op, _ in ins[a:b] if op.startswith("v_mfma"))
    for j in range(len(bars) - 3):
        b0, b1, b2, b3 = bars[j:j + 4]
        if nm(b0, b1) == 16 and nm(b1, b2) == 0 and nm(b2, b3) == 16:
            # backward branch behind b3
            for k in range(b3 + 1, min(b3 + 600, len(ins))):
                addr, op, args = ins[k]
                if op.startswith("s_cbranch") or op == "s_branch":
                    off = int(args.split()[0])
                    if off > 32767:
                        off -= 65536
                    tgt = addr + 4 + 4 * off
                    if tgt < ins[b0][0]:
                        head = next(i for i, (a, _, _) in enumerate(ins) if a >= tgt)
                        return head, b0, b1, b2, b3, k
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--obj", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "imagefolder_amd", "csrc", "_build", "xq_gemm.o"))
    ap.add_argument("--match", default="gemm_pring_kernelILi", help="substring of the mangled kernel names to report")
    ap.add_argument("--dump", default=None, help="substring: print the non-MFMA instructions of that kernel's loop")
    a = ap.parse_args()
    ks = kernels(disassemble(a.obj))
    cols = ["valu", "salu", "ds_read", "lds_dma", "waitcnt", "nop", "branch", "smem", "vmem", "ds_write", "other"]
    print(f"{'kernel <AK, BK, ACT, SUMS>':34s}{'segment':>16s}" + "".join(f"{c:>9s}" for c in cols) + f"{'total':>8s}")
    for name, ins in ks.items():
        if a.match not in name:
            continue
        m = re.search(r"gemm_pring_kernelILi(\d+)ELi(\d+)ELi(\d+)ELb(\d+)E", name)
        tag = "<" + ", ".join(m.groups()) + ">" if m else name[:30]
        lp = find_loop(ins)
        if lp is None:
            print(f"{tag:34s}   no two-phase K loop found")
            continue
        head, b0, b1, b2, b3, br = lp
        segs = [("load phase A", ins[head:b0]), ("MFMA segment A", ins[b0 + 1:b1]), ("load phase B", ins[b1 + 1:b2]), ("MFMA segment B", ins[b2 + 1:b3]),
                ("loop tail", ins[b3 + 1:br + 1])]
        tot = {}
        for sname, si in segs:
            c = count(si)
            n = sum(v for k, v in c.items() if k != "mfma")
            print(f"{tag:34s}{sname:>16s}" + "".join(f"{c.get(k, 0):9d}" for k in cols) + f"{n:8d}")
            for k, v in c.items():
                tot[k] = tot.get(k, 0) + v
        n = sum(v for k, v in tot.items() if k != "mfma")
        print(f"{tag:34s}{'K tile':>16s}" + "".join(f"{tot.get(k, 0):9d}" for k in cols) + f"{n:8d}")
        if a.dump and a.dump in name:
            for addr, op, args in ins[head:br + 1]:
                if not op.startswith("v_mfma"):
                    print(f"      {addr:08x}  {op} {args}")


if __name__ == "__main__":
    main()
