"""Where do a kernel's spills land?  Reads the ISA hipcc emits for one source file (hipcc ... --cuda-device-only -S),
finds every loop of every kernel (a backward branch to a label) and counts, inside each loop body, the scratch loads / stores
(VGPR spill traffic: each reload is a memory round trip AND a vmcnt wait), v_readlane / v_writelane (SGPR spills parked in
VGPR lanes: VALU instructions, no memory), LDS and FP64 instructions.  Usage:
    python bench/isa_loop_spills.py file.s [kernel-name-substring] [min loop length]"""
import re
import sys

src = open(sys.argv[1]).read().split("\n")
want = sys.argv[2] if len(sys.argv) > 2 else ""
min_len = int(sys.argv[3]) if len(sys.argv) > 3 else 60
starts = [i for i, l in enumerate(src) if re.match(r"^_Z\w+:\s*(;.*)?$", l)]
for si, a in enumerate(starts):
    name = src[a].split(":")[0]
    if want not in name:
        continue
    b = next(i for i in range(a, len(src)) if "s_endpgm" in src[i])
    body = src[a:b + 1]
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB[0-9_]+):", l)] if m}
    loops = set()
    for i, l in enumerate(body):
        m = re.search(r"s_c?branch\w*\s+(\.LBB[0-9_]+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.add((labels[m.group(1)], i))
    cnt = lambda lo, hi, pat: sum(1 for l in body[lo:hi + 1] if re.search(pat, l))
    print(f"{name}: {len(body)} lines; whole kernel: scratch_load {cnt(0, len(body) - 1, 'scratch_load')} scratch_store "
          f"{cnt(0, len(body) - 1, 'scratch_store')} v_readlane {cnt(0, len(body) - 1, 'v_readlane')} v_writelane {cnt(0, len(body) - 1, 'v_writelane')}")
    # innermost view: report each loop once, outer loops that merely contain inner ones are listed too (their own counts include the inner bodies)
    for lo, hi in sorted(loops):
        if hi - lo < min_len:
            continue
        print(f"  loop lines {lo + 1:6d}-{hi + 1:6d} ({hi - lo:5d} instr): scratch_load {cnt(lo, hi, 'scratch_load'):3d} scratch_store "
              f"{cnt(lo, hi, 'scratch_store'):3d} | readlane {cnt(lo, hi, 'v_readlane'):3d} writelane {cnt(lo, hi, 'v_writelane'):3d} | ds "
              f"{cnt(lo, hi, r'ds_(read|write)'):3d} f64 {cnt(lo, hi, '_f64'):4d} global {cnt(lo, hi, 'global_'):3d} s_waitcnt {cnt(lo, hi, 's_waitcnt'):3d}")
