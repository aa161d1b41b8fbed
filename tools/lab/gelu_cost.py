"""Upper bound of what a cheaper GELU could buy: the C2 step with every exact-erf GELU replaced by ReLU (wrong outputs, timing only)."""
import sys, os, subprocess, json
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = '''
import sys, runpy
sys.path.insert(0, %r)
from videoglamm_amd import ops
if %d: ops.ACT_GELU = ops.ACT_RELU
sys.argv = ["bench.py", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-quality", "--no-roofline"]
runpy.run_path(%r, run_name="__main__")
'''
for r in range(2):
    for relu in (0, 1):
        out = subprocess.run([sys.executable, "-c", code % (root, relu, os.path.join(root, "bench.py"))], capture_output=True, text=True).stdout
        d = json.loads(out.strip().splitlines()[-1])
        print("GELU -> ReLU" if relu else "exact GELU  ", d["ms_per_step"], flush=True)
