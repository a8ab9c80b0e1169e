"""What the GPU box's host gives this process (CPU share) and how the CPU oracle scales with threads.
  python tools/cpu_probe.py [gates]"""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 2 and sys.argv[1] == "--child":
    import numpy as np, oracle_lib as ol
    gates = int(sys.argv[2])
    ks = ol.KeySet(ol.params("default128"), seed=1)
    ctx = ol.Ctx(ks); ctx.set_fft(True)
    a = ks.encrypt(np.full(gates, 1 << 29, np.int32), 2.0 ** -15, seed=3)
    b = ks.encrypt(np.full(gates, -(1 << 29), np.int32), 2.0 ** -15, seed=4)
    ctx.gate_batch("NAND", a[:8], b[:8])
    t = time.time(); ctx.gate_batch("NAND", a, b); dt = time.time() - t
    print("threads %s: %d gates in %.2f s = %.1f bootstraps/s" % (os.environ.get("OMP_NUM_THREADS"), gates, dt, gates / dt), flush=True)
    sys.exit(0)
print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "-")
print(subprocess.run("lscpu | grep -E 'Model name|^CPU\\(s\\)|Thread|Socket|NUMA node\\(s\\)'", shell=True, capture_output=True, text=True).stdout)
gates = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for th in (8, 16, 32, 64, 128):
    env = dict(os.environ, OMP_NUM_THREADS=str(th))
    subprocess.run([sys.executable, __file__, "--child", str(gates)], env=env)
