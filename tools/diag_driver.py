"""tools/diag_driver.py <name> [width:growth ...] -- whole solves of an example with the resident plan at the given super-block
widths / growth bounds next to the run with the reference hot path: per iteration the step lengths and CG counts of both runs, the
growth of the explicit inverses and how many super-blocks were substituted instead."""
import sys, os
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_driver as td
from driver import sedumi_loop as sl

name = sys.argv[1]
cfgs = [tuple(float(x) for x in a.split(":")) for a in sys.argv[2:]] or [(0, 1e4)]


class Hot(sl.PlanHot):
    def __init__(self, width, growth):
        super().__init__(0)
        self.width, self.growth, self.stats = int(width), growth, []

    def factor(self, S, d, DAt, L, pars):
        from sedumi_amd.plan import Plan
        if self.plan is None:
            self.plan = Plan(self.device)
            self.plan.set_solve_width(self.width)
            self.plan.set_chol(S["L"], S["ADA"])
            self.plan.set_growth_max(self.growth)
            K = S["K"]
            self.plan.set_ada(S["A"], S["Ablkjc"], K, S["DAt"]["q"] if K["q"].size else None)
        out = super().factor(S, d, DAt, L, pars)
        self.stats.append(self.plan.solve_stats())
        return out


ref = td.reference_run(name)
for width, growth in cfgs:
    hot = Hot(width, growth)
    r = td.run(name, hot)
    print(f"== {name} width {int(width)} growth_max {growth:g}: iter {r['iter']} vs {ref['iter']}  cx {r['cx']:.10g} vs {ref['cx']:.10g}")
    for i, (a, b) in enumerate(zip(r["rows"], ref["rows"])):
        st = hot.stats[min(i, len(hot.stats) - 1)]
        print(f"  it {i+1:2d} tP {a['tP']:.4f}/{b['tP']:.4f} tD {a['tD']:.4f}/{b['tD']:.4f} kcg {a['kcg1']},{a['kcg2']}/{b['kcg1']},{b['kcg2']} "
              f"gap {a['gap']:.3e}/{b['gap']:.3e} blocks {st[0]} bad {st[1]} growth {st[2]:.2e}")
