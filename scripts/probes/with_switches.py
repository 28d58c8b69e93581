"""python scripts/probes/with_switches.py name=value[,name=value] (script.py | -m module) [arguments]: run a script or module of this
repository with run-time switches of variantformer_amd.runtime set for the whole process (context-variable default of the main
thread), e.g.   with_switches.py overlap_cre_stream=1 -m pytest tests -m gpu -q"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from variantformer_amd import runtime
kw = {k: bool(int(v)) for k, v in (kv.split("=") for kv in sys.argv[1].split(","))}
runtime.set_for_this_context(**kw)
print(f"[with_switches] {runtime.switches()}", file=sys.stderr, flush=True)
if sys.argv[2] == "-m":
    mod = sys.argv[3]
    sys.argv = [mod] + sys.argv[4:]
    runpy.run_module(mod, run_name="__main__", alter_sys=True)
else:
    script = sys.argv[2]
    sys.argv = [script] + sys.argv[3:]
    runpy.run_path(script, run_name="__main__")
