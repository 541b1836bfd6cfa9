"""Regenerates tests/golden/*.npz: small seeded problems with the oracle's outputs.

The reference cannot be built or imported in this image (Eigen3 / Pinocchio absent), and it
ships no golden vectors, so these fixtures do NOT pin the oracle to the reference binary -- the oracle's
header keeps saying "parity unpinned".  They pin (a) the oracle against silent regressions and (b) the HIP
path against a committed answer that does not need the oracle at test time.

  python tests/golden/make_golden.py        (from the repo root; needs only numpy + gcc)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402
from robotoc_amd import problems as pr  # noqa: E402
from robotoc_amd.grid import discretize, anymal_trot_sequence  # noqa: E402
from robotoc_amd.types import Records, anymal_dims  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def anymal_short():
    """ANYmal trot cut to N=12 (one lift, one impact, one switching-constraint grid)."""
    dims = anymal_dims()
    cs = anymal_trot_sequence(t0=0.05, swing=0.08, double_support=0.05, cycles=1)
    return dims, discretize(12, 12 * 0.02, 0.0, cs)


def main():
    dims, grids = anymal_short()
    L = orc.layout(dims)
    batch = 1
    kkt = pr.make_kkt_batch(L, grids, batch, mode="factory")
    dx0 = pr.make_dx0(L, batch)
    ric = Records(L, "ric").zeros(batch, len(grids))
    d = Records(L, "dir").zeros(batch, len(grids))
    st = orc.riccati_sweep_batch(L, grids, kkt.copy(), ric, d, dx0=dx0)
    np.savez_compressed(os.path.join(HERE, "anymal_trot_n12_riccati.npz"), kkt=kkt, dx0=dx0, ric=ric, dir=d,
                        grid=np.array([[g.type, g.sto, g.sto_next, g.switching_constraint, g.dimf, g.dims,
                                        g.num_grids_in_phase, g.time_stage] for g in grids]),
                        grid_dt=np.array([g.dt for g in grids]))
    batch = 2
    dims2, grids2, meta = pr.config_iiwa14()
    L2 = orc.layout(dims2)
    kkt2 = Records(L2, "kkt").zeros(batch, len(grids2))
    for b in range(batch):
        pr.fill_unconstr_instance(L2, len(grids2), kkt2[b], np.random.default_rng(pr.BASE_SEED + b))
    dx02 = pr.make_dx0(L2, batch)
    ric2 = Records(L2, "ric").zeros(batch, len(grids2))
    d2 = Records(L2, "dir").zeros(batch, len(grids2))
    orc.unconstr_sweep_batch(L2, len(grids2), meta["dt"], kkt2.copy(), ric2, d2, dx0=dx02)
    np.savez_compressed(os.path.join(HERE, "iiwa14_unconstr_riccati.npz"), kkt=kkt2, dx0=dx02, ric=ric2, dir=d2,
                        dt=meta["dt"])
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))


if __name__ == "__main__":
    main()
