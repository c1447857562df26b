# development aid (this container): the per-step instruction count of k_xdrop_slice from the device assembly.
# usage: python tools/xdrop_isa.py [bella.s]   (without a file: hipcc --cuda-device-only -S of bella_amd/csrc/bella_hip.hip, ~90 s)
# Prints the basic blocks of the kernel's innermost step loop in layout order with their VALU counts split by issue rate
# (profiles/r05_valu_rates.txt: add/sub/and/or/xor/lshr/mov issue every ~2.4 cycles per SIMD, everything else every ~4.15), marks the
# blocks a step only enters on a rare event -- the region behind an `s_cbranch_execz` that holds the result store (X-drop / end of
# Phase 4), the rebase (v_pk_sub_i16), the deferred clamp (only v_pk_min/max) or a sequence-end move -- and sums the rest: the
# instructions EVERY anti-diagonal step executes.
import collections, os, re, subprocess, sys, tempfile

FULL = ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_mov_b32", "v_not_b32", "v_add_co_u32", "v_sub_co_u32")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 else None
if path is None:
    path = os.path.join(tempfile.mkdtemp(), "bella.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-result", "-Wno-pass-failed",
                           "--cuda-device-only", "-S", "-o", path, os.path.join(root, "bella_amd", "csrc", "bella_hip.hip")])
lines = open(path).read().split("\n")
s = next(i for i, l in enumerate(lines) if l.startswith("_ZN5bella13k_xdrop_sliceENS_14XdropSliceArgsE:"))
e = next(i for i in range(s, len(lines)) if lines[i].startswith(".Lfunc_end"))
# the kernel as one instruction list with the labels in place
flat = []                                              # (label or None, instruction text or None, loop depth of the enclosing block)
depth = 0
body = lines[s + 1:e]
for n, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?", l)
    if m:
        cm = m.group(2) or ""
        for l2 in body[n + 1:n + 4]:                      # (a loop header's annotation runs over several comment lines)
            if l2.strip().startswith(";") and "Loop Header: Depth=" in l2: cm = l2
            if not l2.strip().startswith(";"): break
        d = re.search(r"Depth=(\d+)", cm)
        depth = int(d.group(1)) if d else 0
        flat.append((m.group(1), None, depth)); continue
    t = l.strip()
    if not t or t[0] in ";.": continue
    flat.append((None, t, depth))
dmax = max(d for _, _, d in flat)
lo = min(i for i, x in enumerate(flat) if x[2] == dmax)
hi = max(i for i, x in enumerate(flat) if x[2] == dmax) + 1
labelpos = {x[0]: i for i, x in enumerate(flat) if x[0]}
def rate(op):
    base = op.replace("_e32", "").replace("_e64", "")
    return "full" if base in FULL else "half"
report = []
def walk(a, b, level):
    """instructions of flat[a:b] that every step executes: a region behind `s_cbranch_execz L` (skipped when no lane is in it) is left out
    when its OWN instructions (nested regions aside) hold a result store, a load (sequence-end refill), the rebase or only the clamp"""
    own, common = [], collections.Counter()
    i = a
    while i < b:
        lab, t, _ = flat[i]
        if t is None: i += 1; continue
        op = t.split()[0]
        own.append(op)
        if op == "s_cbranch_execz" and t.split()[-1] in labelpos and i < labelpos[t.split()[-1]] <= b:
            j = labelpos[t.split()[-1]]
            sub_own, sub_common = walk(i + 1, j, level + 1)
            valu = [o for o in sub_own if o.startswith("v_")]
            only_clamp = len(valu) >= 8 and all(o.startswith(("v_pk_min_i16", "v_pk_max_i16", "v_max_i16")) for o in valu) and sum(o.startswith("v_pk_min_i16") for o in valu) >= 8
            rare = any(o.startswith(("global_store", "global_load")) for o in sub_own) or sum(o.startswith("v_pk_sub_i16") for o in sub_own) >= 8 or only_clamp
            # the sequence-end region (Phase 4 and the step that enters it): no store of its own, but its one nested region is the
            # result store of the 28th Phase-4 step
            if not rare and len(valu) < 60 and report and report[-1][0] == level + 1 and report[-1][4]: rare = True
            report.append((level, flat[i + 1][0] or "(after %s)" % op, len(valu), "RARE" if rare else "every step", any(o.startswith("global_store") for o in sub_own) and len(valu) <= 10))
            if not rare: common.update(sub_common)
            i = j
            continue
        if op.startswith("v_"): common[rate(op)] += 1
        elif op.startswith("s_"): common["salu"] += 1
        i += 1
    return own, common
_, common = walk(lo, hi, 0)
print("k_xdrop_slice: innermost loop (depth %d), %d instructions in all" % (dmax, sum(1 for x in flat[lo:hi] if x[1])))
for level, name, nv, what, _ in report:
    print("  %sregion behind s_cbranch_execz at %-12s own VALU %3d  %s" % ("  " * level, name, nv, what))
h, f = common["half"], common["full"]
print("every step: %d VALU = %d at ~4.15 cycles + %d at ~2.4 cycles = %.0f VALU issue cycles per wavefront-step; %d SALU" % (h + f, h, f, 4.15 * h + 2.4 * f, common["salu"]))
# the outer loop (one trip per sixteen steps): the wave-uniform exit test and the two sequence windows' checkpoints
outer = [x[1].split()[0] for i, x in enumerate(flat) if x[1] and x[2] == dmax - 1]
ov = [o for o in outer if o.startswith("v_")]
print("outer loop (once per 16 steps): %d instructions, %d VALU = %.1f VALU per step if all of it ran (its reload branches run for the lanes that consumed bases)" % (len(outer), len(ov), len(ov) / 16.0))
