#!/usr/bin/env python3
"""What sits in the loops of a kernel besides its arithmetic: per kernel and loop depth, the instructions that are bookkeeping rather than
work -- scratch accesses (spills), selects, 64-bit integer VALU operations (compares, address arithmetic), AGPR copies, lane-spilled
SGPR reloads, scalar multiplies -- and the instruction classes (FP64 / LDS / other vector / scalar / memory) of every depth.

Round 4's gains in the split lock-step kernel (+5.6 %: 64-bit VALU compares and pointer casts per key request, an in-loop spill), the duo
kernel (-2.9 %: 310 selects per CMUX step) and the keyswitch kernels came from reading exactly these numbers.

  python tools/isa_scan.py [file.hip] [kernel-name regex] [-- extra hipcc flags]
  python tools/isa_scan.py redsec_amd/csrc/rs_bootstrap.hip 'wgs_kernel.*Li8E' -- -DRS_BS_PART=2 -mllvm -amdgpu-sched-strategy=max-memory-clause

Compiles the file to gfx950 assembly with the build's common flags (the loop depths are LLVM's block comments; blocks laid out of line can
carry a shallower depth than the loop they belong to: read the table together with the source)."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BOOK = re.compile(r"^(scratch_\w+|v_cndmask\w*|v_cmp_\w+_[iu]64\w*|v_lshl_add_u64|v_mad_u64_u32|v_add_co_u32\w*|v_addc_co_u32\w*|v_accvgpr\w+|v_readlane_b32|"
                  r"v_writelane_b32|v_readfirstlane_b32|v_mul_lo_u32|v_mul_hi_u32|s_mul_i32|s_mul_hi_u32|s_cmp_lg_u64|flat_\w+)")


def classes(op):
    if op.startswith("v_") and "f64" in op:
        return "fp64"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "scratch_", "flat_", "buffer_")):
        return "mem"
    if op.startswith("s_waitcnt"):
        return "wait"
    return "salu" if op.startswith("s_") else "other"


def main():
    argv = sys.argv[1:]
    extra = []
    if "--" in argv:
        k = argv.index("--")
        argv, extra = argv[:k], argv[k + 1:]
    src = argv[0] if argv else os.path.join(ROOT, "redsec_amd/csrc/rs_bootstrap.hip")
    pat = re.compile(argv[1] if len(argv) > 1 else ".")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        cmd = ["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
               "-I" + os.path.join(ROOT, "redsec_amd/csrc"), "--cuda-device-only", "-S", src, "-o", out] + extra
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    for s in starts:
        name = lines[s].split(":")[0]
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("rs::", "")
        if not pat.search(name) and not pat.search(dem):
            continue
        e = next(i for i in range(s, len(lines)) if lines[i].startswith(".Lfunc_end"))
        depth, cls, book = 0, collections.defaultdict(collections.Counter), collections.defaultdict(collections.Counter)
        for l in lines[s:e]:
            if l.startswith(".LBB"):
                m = re.search(r"Depth=(\d+)", l)
                depth = int(m.group(1)) if m else 0
                continue
            t = l.strip()
            if not t or t[0] in ";." or t.endswith(":"):
                continue
            op = t.split()[0]
            cls[depth][classes(op)] += 1
            m = BOOK.match(op)
            if m:
                book[depth][m.group(1)] += 1
        print(dem[:110])
        for d in sorted(cls):
            print("   depth %d  %s" % (d, "  ".join("%s %d" % (k, cls[d][k]) for k in ("fp64", "lds", "valu", "salu", "wait", "mem") if cls[d][k])))
            if book[d]:
                print("            " + "  ".join("%s %d" % kv for kv in book[d].most_common()))


if __name__ == "__main__":
    main()
