#!/usr/bin/env python3
"""oracle/ref/cu2cpp.py — TEST INFRASTRUCTURE ONLY.  Build-time filter used by oracle/ref/Makefile: reads one reference .cu file
where it lies and writes it to stdout with every kernel launch

    kernel<targs> <<< grid, block [, shmem [, stream]] >>> (args);

rewritten as a call of the host block emulator (oracle/ref/emu.h)

    emu::named("kernel<targs>"), emu::launch(grid, block [, shmem [, stream]], [&]{ kernel<targs>(args); });

Nothing else is touched (g++ cannot parse the <<< >>> launch syntax; the kernel bodies are compiled as written).  The output goes
to a temporary directory that the Makefile removes after compiling; no reference text is kept in the repository."""
import re
import sys


def match_paren(s, i):
    """index just after the parenthesis group that opens at s[i] == '('"""
    depth = 0
    while i < len(s):
        if s[i] == "(":
            depth += 1
        elif s[i] == ")":
            depth -= 1
            if depth == 0:
                return i + 1
        i += 1
    raise ValueError("unbalanced parentheses")


def main():
    src = open(sys.argv[1], encoding="latin-1").read()
    out = []
    pos = 0
    pat = re.compile(r"([A-Za-z_][A-Za-z_0-9:]*(?:\s*<[^<>;(){}]*>)?)\s*<\s?<\s?<(.*?)>\s?>\s?>\s*\(", re.S)
    while True:
        m = pat.search(src, pos)
        if not m:
            out.append(src[pos:])
            break
        out.append(src[pos:m.start()])
        end = match_paren(src, m.end() - 1)
        args = src[m.end():end - 1]
        out.append("emu::named(\"%s\"), emu::launch(%s, [&]{ %s(%s); })" % (m.group(1).strip().replace('"', ""), m.group(2).strip(), m.group(1).strip(), args))
        pos = end
    sys.stdout.write('#include "emu.h"\n' + "".join(out))


if __name__ == "__main__":
    main()
