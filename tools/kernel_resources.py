"""Per-kernel register / LDS / scratch use from a hipcc -S listing (the .amdhsa metadata at its end).
    hipcc --offload-arch=gfx950 ... -S --cuda-device-only file.hip -o file.s
    python tools/kernel_resources.py file.s [name-filter]"""
import re
import sys


def main():
    text = open(sys.argv[1]).read()
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    blocks = text.split("  - .agpr_count:")[1:]
    for blk in blocks:
        name = re.search(r"\.name:\s+(\S+)", blk)
        if not name or flt not in name.group(1):
            continue

        def field(key):
            m = re.search(r"\." + key + r":\s+(\d+)", blk)
            return m.group(1) if m else "?"
        print("%-110s vgpr %s spill %s scratch %s sgpr %s lds %s" % (
            name.group(1)[-110:], field("vgpr_count"), field("vgpr_spill_count"),
            field("private_segment_fixed_size"), field("sgpr_count"), field("group_segment_fixed_size")))


if __name__ == "__main__":
    main()
