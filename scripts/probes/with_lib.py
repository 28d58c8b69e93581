"""python scripts/probes/with_lib.py path/to/libvf_*.so script.py [arguments]: run a script of this repository on an alternative
build of the library (scripts/probes/build_probe_libs.py); the script's own _lib.load() then returns the one loaded here."""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from variantformer_amd import _lib
_lib.load(os.path.abspath(sys.argv[1]))
print(f"[with_lib] {sys.argv[1]}", file=sys.stderr, flush=True)
script = sys.argv[2]
sys.argv = [script] + sys.argv[3:]
runpy.run_path(script, run_name="__main__")
