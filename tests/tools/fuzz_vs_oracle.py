"""python tests/tools/fuzz_vs_oracle.py [seed] [scenes] [light_tiles 0/1] -- random small scenes, the HIP path against the CPU oracle (tests/fuzz.py);
prints every scene and the summary the GPU suite records in parity_rNN.json."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from tests import fuzz
res = fuzz.run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 24,
               light_tiles=(bool(int(sys.argv[3])) if len(sys.argv) > 3 else None))
print("failures", res["misses"])
print(json.dumps(res))
