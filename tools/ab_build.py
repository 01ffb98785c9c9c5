"""Developer A/B builds of the library: python tools/ab_build.py NAME [-DFLAG ...]  ->  tools/_build/libmjmpc_NAME.so
(product flags + the extra defines); run a timing tool against it with MJMPC_AMD_LIB=tools/_build/libmjmpc_NAME.so."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mjmpc_amd import build as B
name = sys.argv[1]
lib = os.path.join(ROOT, "tools", "_build", "libmjmpc_%s.so" % name)
os.makedirs(os.path.dirname(lib), exist_ok=True)
B.build(extra_flags=sys.argv[2:], lib=lib)
print(lib)
