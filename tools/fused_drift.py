"""How far apart in z do the fused RL workgroups that share an XCD run?  (diagnostic)

Needs a library built with `make EXTRA=-DLSR_FUSED_PROBE_TIME`: every workgroup then records the
100 MHz wall clock at its start, at its 40th and 120th plane and at its end.  Prints, per dispatch
round and XCD, the spread of those times over the workgroups that run side by side -- one plane
step takes ~3 us, and an XCD's L2 holds less than two plane steps of its 32 workgroups' traffic, so
halo lines are shared only between workgroups that are within about a step of each other.
"""
import ctypes
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    import bench
    from shrimpy_amd import _lib
    from shrimpy_amd.deconvolve import RichardsonLucyPlan
    from shrimpy_amd.deskew import get_deskewed_data_shape

    dev = torch.device("cuda", 0)
    out_shape, _ = get_deskewed_data_shape(bench.WORKLOADS["config2"][1], **bench.DESKEW)
    plan = RichardsonLucyPlan(out_shape, None, dev, psf_factors=bench.gaussian_factors())
    y = plan.new_padded_input()
    y.view.copy_(torch.rand(out_shape, device=dev) + 0.5)
    lib = _lib.load()
    n_wg = 4096
    probe = torch.zeros(4 * n_wg, dtype=torch.int64, device=dev)
    lib.lsr_debug_set_fused_probe.argtypes = [ctypes.c_void_p]
    lib.lsr_debug_set_fused_probe.restype = None
    lib.lsr_debug_set_fused_probe(ctypes.c_void_p(probe.data_ptr()))
    plan(y, iterations=3)
    torch.cuda.synchronize()
    t = probe.cpu().numpy().reshape(n_wg, 4).astype(np.float64)
    used = np.nonzero(t[:, 0])[0]
    n = int(used.max()) + 1
    t = t[:n]
    t0 = t[:, 0].min()
    t = (t - t0) / 100.0  # microseconds
    cus = 256
    print(json.dumps({"workgroups": n, "launch_us": float(t[:, 3].max())}))
    for rnd in range((n + cus - 1) // cus):
        ids = np.arange(rnd * cus, min((rnd + 1) * cus, n))
        rows = []
        for xcd in range(8):
            g = ids[ids % 8 == xcd]
            if len(g) == 0:
                continue
            rows.append([np.ptp(t[g, k]) for k in range(4)])
        rows = np.array(rows)
        print(json.dumps({"round": rnd, "workgroups": len(ids),
                          "spread_us_start_p40_p120_end_mean_over_xcds": [round(float(v), 2) for v in rows.mean(0)],
                          "max_over_xcds": [round(float(v), 2) for v in rows.max(0)],
                          "us_per_plane_p40_to_p120": round(float(np.mean(t[ids, 2] - t[ids, 1]) / 80.0), 3)}))
    # neighbours in the tile order inside an XCD: |dt| at plane 120 between consecutive workgroups
    for rnd in range(min(4, n // cus)):
        d = []
        for xcd in range(8):
            g = np.arange(rnd * cus + xcd, (rnd + 1) * cus, 8)
            d.extend(np.abs(np.diff(t[g, 2])))
        print(json.dumps({"round": rnd, "neighbour_abs_dt_us_at_p120": {"median": round(float(np.median(d)), 2),
                          "p90": round(float(np.percentile(d, 90)), 2), "max": round(float(np.max(d)), 2)}}))


if __name__ == "__main__":
    main()
