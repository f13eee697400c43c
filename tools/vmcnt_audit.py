#!/usr/bin/env python3
"""For every kernel in a hipcc -S listing: the vector-memory waits inside its loops.

A `s_waitcnt vmcnt(0)` in a main loop that also stores means the wave waits for the acknowledgement of its own output stores
before it may use the next unit's samples (vmcnt retires loads and stores together, in issue order) -- the pattern the counted
buffer stores of ss_wave.h remove.  Usage: vmcnt_audit.py file.s [name-filter]"""
import re
import sys


def main():
    path = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    name, body = None, []
    kernels = []
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, body = m.group(1), []
            continue
        if name is not None:
            body.append(line)
            if "s_endpgm" in line:
                kernels.append((name, body))
                name = None
    for name, body in kernels:
        if flt and flt not in name:
            continue
        in_loop = [("Loop" in l and "Header" in l) or "in Loop" in l for l in body]
        # a line is "inside a loop" if the nearest preceding label comment says so
        inside, cur = [], False
        for l in body:
            if l.startswith(".LBB") or l.startswith("; %bb"):
                cur = "Loop" in l
            inside.append(cur)
        waits, stores, loads = [], 0, 0
        for l, ins in zip(body, inside):
            if not ins:
                continue
            t = l.strip()
            m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", t)
            if m:
                waits.append(int(m.group(1)))
            elif t.startswith(("global_store", "buffer_store", "flat_store")):
                stores += 1
            elif t.startswith(("global_load", "buffer_load", "flat_load")):
                loads += 1
        print(f"{name[:110]}\n    in loops: loads {loads} stores {stores} vmcnt waits {waits}")


if __name__ == "__main__":
    main()
