#!/bin/bash
# Round-3 first GPU call: (1) what a fresh box charges a python process (first `import torch`, smoke()), (2) LDS issue
# costs beside an FP64 stream, (3) phase shares of the workgroup blind rotation (-DRS_STAMPS build), (4) bench baseline.
# Everything lands under gpurun_out/r03_probe1/.
set -o pipefail
OUT=gpurun_out/r03_probe1
mkdir -p $OUT
t0=$(date +%s.%N)
stamp() { echo "$(echo "$(date +%s.%N) - $t0" | bc) $*" | tee -a $OUT/timeline.txt; }
stamp start
python -c "import time; t=time.time(); import ctypes; ctypes.CDLL('libamdhip64.so'); print('dlopen libamdhip64', round(time.time()-t,2))" 2>&1 | tee -a $OUT/timeline.txt
stamp hip_loaded
python -c "import time; t=time.time(); import numpy; print('import numpy', round(time.time()-t,2))" 2>&1 | tee -a $OUT/timeline.txt
python -c "import time; t=time.time(); import torch; print('first import torch', round(time.time()-t,2)); t=time.time(); torch.zeros(1).cuda(); print('first cuda tensor', round(time.time()-t,2))" 2>&1 | tee -a $OUT/timeline.txt
stamp torch_first
python -c "import time; t=time.time(); import torch; print('second import torch', round(time.time()-t,2))" 2>&1 | tee -a $OUT/timeline.txt
stamp torch_second
python -c "
import time; t=time.time()
import __graft_entry__ as g
g.smoke(); print('smoke()', round(time.time()-t,2))" 2>&1 | tail -3 | tee -a $OUT/timeline.txt
stamp smoke_done
tools/lds_issue_bench > $OUT/lds_issue.jsonl 2> $OUT/lds_issue.err || echo "lds_issue_bench failed"
stamp lds_issue_done
REDSEC_HIP_LIB=$PWD/variants/lib_stamps.so timeout -k 10 300 python tools/stamp_profile.py default128 16384 > $OUT/stamps_default128.json 2> $OUT/stamps_default128.err || echo "stamps d128 failed"
REDSEC_HIP_LIB=$PWD/variants/lib_stamps.so timeout -k 10 300 python tools/stamp_profile.py redsec_small_v2 16384 > $OUT/stamps_redsec.json 2> $OUT/stamps_redsec.err || echo "stamps redsec failed"
stamp stamps_done
timeout -k 10 400 python bench.py --steps 5 --warmup 1 > $OUT/bench_base.json 2> $OUT/bench_base.err || echo "bench failed"
stamp bench_done
tail -c 600 $OUT/bench_base.json
cat $OUT/timeline.txt
