"""Wall clock of the reference's system-level flow (src/batt_models/battgp_full.py:41-125: build 1 + 8 cell models,
predict each at the reference operating point on the 300-point grid, delete it) on synthetic BattData.
    python tools/system_probe.py [N per cell ...]        (BGP_STREAMS=k: k GPs in flight on the one GPU; unset = the
driver's automatic choice, 1 = the reference's sequential loop)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from battgp_amd.battgp_full import BattGP_Full  # noqa: E402
from battgp_amd.synthetic import SyntheticBattData  # noqa: E402

for n in [int(a) for a in sys.argv[1:]] or [1000, 4000, 16000]:
    for rep in range(2):
        bd = SyntheticBattData(f"sys{n}_{rep}", n_cells=8, seed=rep)
        t0 = time.perf_counter()
        k = int(os.environ["BGP_STREAMS"]) if "BGP_STREAMS" in os.environ else None
        sysmodel = BattGP_Full(bd, max_training_data=n, device=0, in_flight=k)
        used = sysmodel._in_flight([sysmodel.packmodel, *sysmodel.cellmodels], 300) if k is None else k
        t1 = time.perf_counter()
        res = sysmodel.predict_cell_r0_op(save=False)
        t2 = time.perf_counter()
    print(json.dumps({"n_per_cell": n, "gps": 9, "in_flight": used, "build_ms": (t1 - t0) * 1e3, "predict_all_ms": (t2 - t1) * 1e3,
                      "per_gp_ms": (t2 - t1) * 1e3 / 9, "columns": len(res.df.columns)}), flush=True)
