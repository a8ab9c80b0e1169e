"""Register, scratch and LDS budgets of the hot kernels, read from the code objects inside the BUILT libredsec_hip.so.

The kernels of this path live at the edge of the register file (256 VGPRs at two waves per SIMD) and of the 160 KB of LDS, and
round 4 showed twice what falls off that edge: XCD-cohort state held in registers put a spill (and with it a wait for every
outstanding key load) into each CMUX step of the split lock-step kernel, and a first form of the keyswitch kernels compiled to
1.3-7.5 KB of scratch per lane. Neither changes a single output word, so no parity test sees it. This test reads the kernel
descriptors the compiler wrote (llvm-readelf --notes on the gfx950 code objects) and holds them to the budgets the design
rests on: occupancy-defining VGPR counts, scratch bytes per lane, LDS bytes per workgroup. No GPU needed."""
import glob
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def _kernels():
    lib = os.path.join(ROOT, "redsec_amd", "libredsec_hip.so")
    if not (os.path.exists(lib) and os.path.exists(os.path.join(LLVM, "llvm-objdump"))):
        pytest.skip("library not built or ROCm LLVM tools absent")
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        copy = os.path.join(tmp, "l.so")
        shutil.copy(lib, copy)                      # --offloading writes the bundles next to its input
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", copy], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for co in glob.glob(copy + ".*gfx950"):
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
            for blk in notes.split("\n  - ")[1:]:
                m = re.search(r"\.name:\s+(\S+)", blk)
                if not m:
                    continue
                num = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1))
                out[m.group(1)] = dict(vgpr=num("vgpr_count"), scratch=num("private_segment_fixed_size"), lds=num("group_segment_fixed_size"))
    return out


# (mangled-name pattern, what it is, max VGPRs incl. AGPRs, max scratch bytes per lane, max LDS bytes per workgroup)
BUDGETS = [
    (r"22blind_rotate_wg_kernelINS_5XfFft.*Li8EEEv", "lock-step kernel (headline): two waves per SIMD, one workgroup per CU", 256, 128, 163840),
    (r"23blind_rotate_wgs_kernel.*Li8EEEv", "split lock-step kernel: no scratch access inside the CMUX loop since the lean key requests (88 B outside)", 256, 96, 163840),
    (r"23blind_rotate_duo_kernel", "duo kernel (80 B with the last stage group's twiddles kept in registers: all outside the CMUX loop)", 256, 96, 163840),
    (r"24blind_rotate_duos_kernel", "split duo kernel", 256, 48, 163840),
    (r"25blind_rotate_coop8_kernel", "coop8: eight waves of one ciphertext", 256, 16, 163840),
    (r"32blind_rotate_coop8_listed_kernel", "coop8 with the listed step (l < 4): a key row stays in registers across two barriers -- no scratch", 256, 0, 163840),
    (r"19blind_rotate_kernelINS_5XfNtt.*Li8EEEv", "exact-NTT per-wave kernel: 8 waves fill the LDS", 256, 128, 163840),
    (r"23gen_blind_rotate_kernelILi1[0-3]E", "general ring kernels: pinned to two waves per SIMD", 256, 96, 163840),
    (r"27keyswitch_tiled_comb_kernel", "keyswitch, combined digits: two workgroups per CU", 240, 0, 81920),
    (r"22keyswitch_tiled_kernel", "keyswitch, one lookup per digit: at least two workgroups per CU", 224, 0, 54613),
]


def test_hot_kernels_stay_inside_their_register_scratch_and_lds_budgets():
    ks = _kernels()
    assert len(ks) > 60, "expected the whole library's kernels in the code objects, found %d" % len(ks)
    for pat, what, vgpr, scratch, lds in BUDGETS:
        hit = {n: k for n, k in ks.items() if re.search(pat, n)}
        assert hit, "no kernel matches %s (%s)" % (pat, what)
        for n, k in hit.items():
            assert k["vgpr"] <= vgpr and k["scratch"] <= scratch and k["lds"] <= lds, (what, n, k, dict(vgpr=vgpr, scratch=scratch, lds=lds))
