#!/usr/bin/env python3
"""Per-kernel digest of the gfx950 code objects inside a built libredsec_hip.so: the instruction stream of every function
(llvm-objdump -d, addresses and encodings stripped; branch operands are relative offsets already) hashed per symbol.
Two builds whose digests agree run the same instructions -- how a source clean-up (pruned experiment switches) is shown to
leave the default build untouched.
usage: tools/codeobj_digest.py [lib.so] > digest.json ; tools/codeobj_digest.py --diff a.json b.json"""
import glob, hashlib, json, os, re, shutil, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"


def digest(lib):
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        copy = os.path.join(tmp, "l.so")
        shutil.copy(lib, copy)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", copy], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for co in sorted(glob.glob(copy + ".*gfx950")):
            txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", "--no-leading-addr", co], capture_output=True, text=True, check=True).stdout
            sym, body = None, []
            for line in txt.splitlines():
                m = re.match(r"^(?:[0-9a-f]+ )?<(.+)>:$", line)
                if m:
                    if sym:
                        out[sym] = {"n": len(body), "sha": hashlib.sha256("\n".join(body).encode()).hexdigest()[:16]}
                    sym, body = m.group(1), []
                    continue
                if sym and line.strip():
                    ins = line.split("//")[0].strip()
                    body.append(ins)
            if sym:
                out[sym] = {"n": len(body), "sha": hashlib.sha256("\n".join(body).encode()).hexdigest()[:16]}
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--diff":
        a, b = json.load(open(sys.argv[2])), json.load(open(sys.argv[3]))
        same = [k for k in a if k in b and a[k] == b[k]]
        diff = [k for k in a if k in b and a[k] != b[k]]
        print("kernels/functions: %d before, %d after; identical %d; changed %d; only before %d; only after %d"
              % (len(a), len(b), len(same), len(diff), len(set(a) - set(b)), len(set(b) - set(a))))
        for k in diff:
            print("  changed:", k[:150], a[k]["n"], "->", b[k]["n"], "instructions")
        for k in sorted(set(a) - set(b)):
            print("  only before:", k[:150])
        for k in sorted(set(b) - set(a)):
            print("  only after:", k[:150])
        sys.exit(1 if diff or set(a) != set(b) else 0)
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "redsec_amd", "libredsec_hip.so")
    print(json.dumps(digest(lib), indent=0, sort_keys=True))
