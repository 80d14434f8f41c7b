#!/bin/bash
# Rehearsal of the sharded CLI on a one-GPU box: two ranks share the card (gloo carries the barrier),
# each reconstructs its share of a small plate; the output is compared with a single-rank run.
#   tools/cli_two_ranks.sh <workdir under gpurun_out>
set -e
R=$GRAFT_REPO_ROOT
W=$R/gpurun_out/$1
rm -rf $W && mkdir -p $W
cd $R
python - <<PY
import numpy as np, yaml
from shrimpy_amd.io.omezarr import open_ome_zarr
rng = np.random.default_rng(1)
with open_ome_zarr("$W/raw.zarr", layout="hcs", mode="w", channel_names=["LS"], prefer_iohub=False) as plate:
    for i in range(5):
        pos = plate.create_position("A", str(i + 1), "0")
        arr = pos.create_zeros("0", shape=(2, 1, 96, 24, 70), dtype="uint16", scale=(1, 1, 0.15, 0.1133, 0.1133))
        for t in range(2):
            arr.write_volume(t, 0, rng.integers(90, 900, (96, 24, 70)).astype(np.uint16))
open("$W/recon.yml", "w").write(yaml.safe_dump(dict(
    deskew=dict(pixel_size_um=0.1133, ls_angle_deg=30.0, scan_step_um=0.15, keep_overhang=False, average_n_slices=3),
    deconvolution=dict(iterations=5, gaussian_shape_zyx=[5, 5, 5], gaussian_sigma_zyx=[1.2, 1.0, 1.0]))))
PY
timeout -k 10 200 python -m shrimpy_amd reconstruct -i $W/raw.zarr -c $W/recon.yml -o $W/one.zarr
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 \
    -m shrimpy_amd reconstruct -i $W/raw.zarr -c $W/recon.yml -o $W/two.zarr
python - <<PY
import numpy as np
from shrimpy_amd.io.omezarr import open_ome_zarr
a, b = open_ome_zarr("$W/one.zarr", prefer_iohub=False), open_ome_zarr("$W/two.zarr", prefer_iohub=False)
n = 0
for (ka, pa), (kb, pb) in zip(a.positions(), b.positions()):
    assert ka == kb
    for t in range(2):
        va, vb = pa["0"].read_volume(t, 0), pb["0"].read_volume(t, 0)
        assert va.shape == vb.shape and np.array_equal(va, vb) and float(va.max()) > 0, (ka, t)
        n += 1
print("two ranks == one rank on", n, "volumes")
PY
