"""Static instruction mix of one kernel in a `hipcc -S --cuda-device-only` listing (VALU / SALU / branch / LDS / VMEM / waits),
whole body and per basic-block label, to compare two versions of a kernel before going to the GPU.
    python tools/count_isa.py file.s <mangled-name-substring> [--blocks]"""
import re
import sys


def kind(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_nop")):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    lines = open(sys.argv[1]).read().split("\n")
    flt, blocks = sys.argv[2], "--blocks" in sys.argv
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z\w+):", lines[i])
        if not (m and flt in m.group(1)):
            i += 1
            continue
        name, total, per_block, label = m.group(1), {}, [], "entry"
        cur = {}
        i += 1
        while i < len(lines) and not lines[i].startswith(".Lfunc_end"):
            ln = lines[i].strip()
            i += 1
            if not ln or ln.startswith((";", "//")):
                continue
            if ln.startswith("."):
                if ln.endswith(":") or re.match(r"^\.LBB\w+:", ln):
                    per_block.append((label, cur))
                    label, cur = ln.split(":")[0], {}
                continue
            k = kind(ln.split()[0])
            total[k] = total.get(k, 0) + 1
            cur[k] = cur.get(k, 0) + 1
        per_block.append((label, cur))
        print(name[-70:], dict(sorted(total.items())))
        if blocks:
            for label, c in per_block:
                if sum(c.values()) >= 8:
                    print("   %-14s %s" % (label, dict(sorted(c.items()))))


if __name__ == "__main__":
    main()
