#!/bin/bash
OUT=gpurun_out/r03_gen5; mkdir -p $OUT
for v in gen_dbg_noasm gen_dbg_near2 gen_dbg_near2; do
  REDSEC_HIP_LIB=$PWD/variants/lib_$v.so timeout -k 10 200 python tools/r03/r03_dbg_large.py ${v}_$RANDOM 2>&1 | grep -v amdgpu.ids | tee -a $OUT/dbg2.txt
done
python - <<'PY' | tee -a gpurun_out/r03_gen5/dbg2.txt
import glob, numpy as np
fs = sorted(glob.glob("gpurun_out/r03_gen5/out_*.npy"))
ref = [f for f in fs if "noasm" in f][0]
r = np.load(ref)
for f in fs:
    o = np.load(f)
    d = np.argwhere(o != r)
    print(f, "differing words:", len(d), "first:", d[:3].tolist(), "rows:", sorted(set(d[:,0].tolist())))
PY
