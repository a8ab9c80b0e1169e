"""One sha256 over every source file a GPU suite run exercises (product, headers, oracle, tests, bench.py, build entry).

tools/final_check.sh writes it into the header of the suite log it commits under profiles/rNN/; tests/test_final_check_cpu.py
recomputes it and fails when the tree's code differs from the newest log's -- documentation, profiles and tools may change after
the last GPU run, code may not ("make an untested final tree impossible").

usage: python tools/tree_hash.py            -> prints the hash
       python tools/tree_hash.py --list     -> prints `sha256  path` per file, then the hash
"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# directories walked recursively, and the file types that count as code there
CODE_DIRS = ("redsec_amd", "include", "oracle", "tests")
CODE_EXT = (".py", ".c", ".cpp", ".h", ".hip", ".sh", ".dat", ".csv", ".json", ".npz", ".npy", ".txt")
CODE_NAMES = ("Makefile",)
CODE_FILES = ("bench.py", "__graft_entry__.py")
SKIP_DIRS = ("__pycache__", "_ref", ".pytest_cache", ".hypothesis")


def code_files(root=ROOT):
    out = [f for f in CODE_FILES if os.path.exists(os.path.join(root, f))]
    for d in CODE_DIRS:
        for base, dirs, files in os.walk(os.path.join(root, d)):
            dirs[:] = sorted(x for x in dirs if x not in SKIP_DIRS)
            for f in sorted(files):
                if f.endswith(CODE_EXT) or f in CODE_NAMES:
                    out.append(os.path.relpath(os.path.join(base, f), root))
    return sorted(out)


def tree_hash(root=ROOT, listing=None):
    h = hashlib.sha256()
    for rel in code_files(root):
        d = hashlib.sha256(open(os.path.join(root, rel), "rb").read()).hexdigest()
        if listing is not None:
            listing.append((d, rel))
        h.update(("%s  %s\n" % (d, rel)).encode())
    return h.hexdigest()


if __name__ == "__main__":
    rows = []
    digest = tree_hash(listing=rows)
    if "--list" in sys.argv:
        for d, rel in rows:
            print(d, rel)
    print(digest)
