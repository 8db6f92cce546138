"""python -m sedumi_amd.driver problem.mat [more.mat ...] -- solve SeDuMi problem files (At | A, b, c, K) on the MI355X without MATLAB:
one JSON line per problem (iterations, c'x, b'y, feasratio, seconds)."""
import json
import sys
import time

from .loop import load_mat, solve

for path in sys.argv[1:]:
    At, b, c, K = load_mat(path)
    t0 = time.time()
    r = solve(At, b, c, K)
    print(json.dumps({"problem": path, "iter": int(r["iter"]), "cx": float(r["cx"]), "by": float(r["by"]), "feasratio": float(r["feasratio"]),
                      "STOP": int(r["STOP"]), "seconds": time.time() - t0}), flush=True)
