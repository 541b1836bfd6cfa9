import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robotoc_amd import capi, problems as pr
from robotoc_amd.types import BUF_KKT, BUF_DX0
for name, fn in (("jump_sto", pr.config_anymal_jump_sto), ("trot", pr.config_anymal_trot)):
    d, g, _ = fn()
    c = capi.Context(d, len(g), 1, 0); c.set_grid(g)
    c.upload(BUF_KKT, pr.make_kkt_batch(c.L, g, 1)); c.upload(BUF_DX0, pr.make_dx0(c.L, 1))
    c.time_phase(4, 2); ser = (c.time_phase(0, 10), c.time_phase(1, 10))
    c.set_backward_scan(True); c.time_phase(4, 2); sc = (c.time_phase(0, 10), c.time_phase(1, 10))
    print(name, "serial bwd/fwd ms", ser, "scan bwd/fwd ms", sc)
