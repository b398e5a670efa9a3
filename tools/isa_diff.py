"""Compare the gfx950 ISA of kernels between two `hipcc -S --cuda-device-only` outputs (a check that
a source refactoring left a tuned kernel's code untouched).  usage: isa_diff.py old.s new.s [filter]
A kernel of old.s is matched to the kernel of new.s whose mangled name is equal, or equal after
`--map OLD=NEW` substring substitutions."""
import re
import sys


def kernels(path):
    text = open(path).read()
    out = {}
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\s*s_endpgm", text, re.S | re.M):
        body = [ln.split(";")[0].rstrip() for ln in m.group(2).split("\n")]
        out[m.group(1)] = [ln for ln in body if ln.strip() and not ln.strip().startswith(".")]
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--map")]
    maps = [a.split("=", 1) for a in sys.argv[1:] if a.startswith("--map") for a in [a[6:]]]
    old, new = kernels(args[0]), kernels(args[1])
    flt = args[2] if len(args) > 2 else ""
    for name, body in old.items():
        if flt not in name:
            continue
        other = name
        for a, b in maps:
            other = other.replace(a, b)
        if other not in new:
            print(f"{name[:90]}: no counterpart")
            continue
        nb = new[other]
        diff = sum(1 for x, y in zip(body, nb) if x != y) + abs(len(body) - len(nb))
        print(f"{name[:90]}: {len(body)} / {len(nb)} instructions, "
              + ("IDENTICAL" if body == nb else f"{diff} lines differ"))


if __name__ == "__main__":
    main()
