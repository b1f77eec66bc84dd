"""debug aid: every fuzz configuration of the large-table-paths variant in its own process, device buffers poisoned"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = ("import sys; sys.path.insert(0, %r); import tests.test_gpu_fuzz as f; f._run_configuration(int(sys.argv[1])); "
        "print('PASS')") % root
env = dict(os.environ, NVSM_POISON="1", NVSM_LAZY_MIN_MB="0", NVSM_ENTRY_WALK_MIN="0")
if len(sys.argv) > 1 and sys.argv[1] == "plain":
    env.pop("NVSM_LAZY_MIN_MB"); env.pop("NVSM_ENTRY_WALK_MIN")
seeds = range(0, 120, 3) if not (len(sys.argv) > 1 and sys.argv[1] == "plain") else range(120)
for seed in seeds:
    r = subprocess.run([sys.executable, "-c", code, str(seed)], env=env, capture_output=True, text=True, cwd=root)
    ok = "PASS" in r.stdout
    if not ok:
        tail = [l for l in (r.stdout + r.stderr).splitlines() if "File \"/usr" not in l][-6:]
        print(seed, "FAIL rc", r.returncode, " | ".join(tail)[:600], flush=True)
print("sweep done", flush=True)
