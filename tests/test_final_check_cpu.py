"""An untested final tree must be impossible (the tree the driver tested had never passed its own GPU suite --
the last code commit came after the builder's last suite run). tools/final_check.sh runs the `-m gpu` suite through gpurun and
commits its log under profiles/rNN/ with the commit and a hash of every code file in the header; this test recomputes the hash
and fails when the code differs from what the NEWEST log ran on, or when that run was not green. Docs, profiles and tools may
follow the log; code (redsec_amd/, include/, oracle/, tests/, bench.py, __graft_entry__.py) may not."""
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import tree_hash  # noqa: E402


def _header(path):
    h = {}
    for line in open(path, errors="replace"):
        if not line.startswith("# "):
            break
        k, _, v = line[2:].partition(":")
        h[k.strip()] = v.strip()
    return h


def _logs():
    logs = glob.glob(os.path.join(ROOT, "profiles", "r*", "gpu_suite_final_*.log"))
    return sorted(logs, key=lambda p: (_header(p).get("date", ""), p))


def test_tree_hash_sees_code_and_nothing_else(tmp_path):
    files = tree_hash.code_files()
    assert "bench.py" in files and "redsec_amd/csrc/rs_bootstrap.hip" in files and "tests/test_gpu_parity.py" in files
    assert "include/redsec_hip.h" in files and "oracle/redsec_oracle.c" in files and "redsec_amd/host/layers.cpp" in files
    assert not any(f.endswith((".so", ".stamp", ".md", ".pyc")) or f.startswith(("profiles/", "tools/", "gpurun_out/")) for f in files)
    # sensitive to one changed byte, to a new file and to a rename; blind to documentation
    root = tmp_path / "t"
    (root / "redsec_amd").mkdir(parents=True)
    (root / "redsec_amd" / "a.py").write_text("x = 1\n")
    (root / "bench.py").write_text("pass\n")
    h0 = tree_hash.tree_hash(str(root))
    (root / "DESIGN.md").write_text("words\n")
    assert tree_hash.tree_hash(str(root)) == h0
    (root / "redsec_amd" / "a.py").write_text("x = 2\n")
    h1 = tree_hash.tree_hash(str(root))
    assert h1 != h0
    (root / "redsec_amd" / "b.hip").write_text("")
    h2 = tree_hash.tree_hash(str(root))
    assert h2 != h1
    os.rename(root / "redsec_amd" / "b.hip", root / "redsec_amd" / "c.hip")
    assert tree_hash.tree_hash(str(root)) != h2


def test_the_newest_gpu_suite_log_is_green_and_ran_on_this_code():
    logs = _logs()
    assert logs, "no profiles/rNN/gpu_suite_final_*.log: run tools/final_check.sh"
    newest = logs[-1]
    h = _header(newest)
    assert re.fullmatch(r"[0-9a-f]{40}", h.get("head", "")), newest
    assert h.get("tree_sha256") == h.get("tree_sha256_on_box"), "the box ran another tree than the one hashed: " + newest
    m = re.search(r"(\d+) passed", h.get("result", ""))
    assert m and int(m.group(1)) >= 150 and "failed" not in h["result"] and "error" not in h["result"], (newest, h.get("result"))
    now = tree_hash.tree_hash()
    assert h["tree_sha256"] == now, ("code changed after the last GPU suite run (%s, head %s): run tools/final_check.sh again"
                                     % (os.path.relpath(newest, ROOT), h["head"][:10]))
