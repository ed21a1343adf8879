#!/usr/bin/env python
"""Post-mortem of a GPU memory access fault in a replayed graph: python tools/r6/postmortem.py <snapshot pickle> <log with the fault line>.
Names (1) the allocator segment the faulting address belongs to / borders on, (2) every allocation of the recorded trace that ends
at or spans the address (with the Python frames inside this repository), (3) segment_free / segment_unmap events covering it."""
import pickle
import re
import sys


def frames_of(ev, n=6):
    fr = [f for f in ev.get("frames", []) if "/ofasys_amd/" in f.get("filename", "") or "bench.py" in f.get("filename", "") or "/tools/" in f.get("filename", "")]
    return " <- ".join(f"{f['filename'].split('/')[-1]}:{f['line']}({f['name']})" for f in fr[:n])


def main():
    snap = pickle.load(open(sys.argv[1], "rb"))
    log = open(sys.argv[2]).read()
    m = re.search(r"on address (0x[0-9a-f]+)", log)
    if not m:
        print("no fault line in the log")
        return
    F = int(m.group(1), 16)
    print(f"fault address {F:#x}; private pool of the step graphs: {snap.get('pool')}")
    for seg in snap["segments"]:
        a, e = seg["address"], seg["address"] + seg["total_size"]
        if a - (4 << 20) <= F <= e + (4 << 20):
            rel = "CONTAINS" if a <= F < e else ("ends AT the fault address" if e == F else ("ends %d B before" % (F - e) if e < F else "starts %d B after" % (a - F)))
            print(f"segment [{a:#x}, {e:#x}) {seg['total_size']} B pool {tuple(seg.get('segment_pool_id', (0, 0)))} type {seg.get('segment_type')} {rel}")
            if a <= F <= e:
                off = a
                for b in seg["blocks"]:
                    if off + b["size"] >= F - (1 << 20):
                        print(f"    block [{off:#x}, {off + b['size']:#x}) {b['size']} B {b['state']} requested {b.get('requested_size')}  {frames_of(b)}")
                    off += b["size"]
    n = 0
    for dev, trace in enumerate(snap["device_traces"]):
        print(f"device {dev}: {len(trace)} trace events")
        for i, ev in enumerate(trace):
            a, sz = ev.get("addr", 0), ev.get("size", 0)
            act = ev.get("action", "")
            if act.startswith("segment"):
                if a - (2 << 20) <= F <= a + sz + (2 << 20):
                    print(f"  [{i}] {act} [{a:#x}, {a + sz:#x}) {sz} B  {frames_of(ev)}")
            elif act == "alloc":
                if a <= F <= a + sz or (a + sz <= F and F - (a + sz) < 4096):
                    n += 1
                    print(f"  [{i}] alloc [{a:#x}, {a + sz:#x}) {sz} B ends {F - (a + sz)} B before the fault address  {frames_of(ev)}")
    print(f"{n} allocation(s) adjacent to the fault address")


if __name__ == "__main__":
    main()
