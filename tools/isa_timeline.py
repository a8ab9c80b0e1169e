"""Compressed timeline of a kernel's ISA: LDS instructions, waits and barriers with the number of vector
instructions between them -- the view that exposed the in-order LDS drains.

  hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -Iinclude -Iredsec_amd/csrc -S --cuda-device-only \
        -o /tmp/k.s redsec_amd/csrc/rs_bootstrap.hip
  awk '/^_ZN2rs22blind_rotate_wg_kernel/{f=1} f{print} f&&/s_endpgm/{exit}' /tmp/k.s > /tmp/one.s
  python tools/isa_timeline.py /tmp/one.s

V<n> n vector instructions; r/R ds_read_b64/b128; W/w ds_write2_b64/ds_write_b64; q/Q ds_read_b32/read2st64_b32;
X ds_write2st64_b32; G global_load_lds; <Ln> s_waitcnt lgkmcnt(n); <Vn> vmcnt(n); |BAR| s_barrier.
"""
import re,sys
lines=open(sys.argv[1]).read().split('\n')
out=[]; v=0
def flush():
    global v
    if v: out.append("V%d"%v); v=0
for i,l in enumerate(lines):
    t=l.strip()
    if not t or t.startswith(';') or t.startswith('.'):
        if t.startswith('.LBB'): flush(); out.append("\n[%d %s]"%(i+1,t.split(':')[0]))
        continue
    op=t.split()[0]
    if op.startswith('v_'): v+=1
    elif op.startswith('ds_'):
        flush(); out.append({'ds_read_b64':'r','ds_read_b128':'R','ds_write2_b64':'W','ds_write_b64':'w','ds_read_b32':'q','ds_read2st64_b32':'Q','ds_write2st64_b32':'X','ds_read_u16':'u','ds_write_b32':'x'}.get(op,op))
    elif op=='s_waitcnt':
        flush(); m=re.search(r'lgkmcnt\((\d+)\)',t); m2=re.search(r'vmcnt\((\d+)\)',t)
        out.append("<%s%s>"%("L"+m.group(1) if m else "", "V"+m2.group(1) if m2 else ""))
    elif op=='s_barrier': flush(); out.append("|BAR|")
    elif op.startswith('global_load_lds'): flush(); out.append("G")
    elif op.startswith('s_cbranch') or op=='s_branch': flush(); out.append("(br)")
flush()
print(" ".join(out))
