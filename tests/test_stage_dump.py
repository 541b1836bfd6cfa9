"""Stage dump / replay format: numpy writer/reader round trip on the CPU; on the GPU a
dump written in numpy is loaded by the C library and swept, and a dump saved by the C library is read
back in numpy -- both bit for bit."""
import numpy as np
import pytest

from robotoc_amd import problems as pr
from robotoc_amd.replay import read_dump, write_dump
from robotoc_amd.types import BUF_CDD, BUF_CON, BUF_CONE, BUF_DIR, BUF_DX0, BUF_KKT, BUF_RIC, joint_limit_rows, Layout


def _problem(batch=3):
    import ctypes as C
    from robotoc_amd import capi
    dims, grids, _ = pr.config_anymal_trot()
    L = Layout()
    capi.lib().rtoc_layout_for_dims(C.byref(dims), C.byref(L))
    kkt, cdd = pr.make_precondense_batch(L, grids, batch)
    return dims, grids, L, dict(kkt=kkt, cdd=cdd, con=pr.make_constraint_batch(L, grids, batch),
                                cone=pr.make_cone_batch(L, grids, batch, 4), dx0=pr.make_dx0(L, batch))


def test_numpy_round_trip(tmp_path):
    dims, grids, L, d = _problem()
    rows = joint_limit_rows(dims)
    path = tmp_path / "stage.rtocdump"
    write_dump(path, dims, grids, 3, {BUF_KKT: d["kkt"], BUF_CDD: d["cdd"], BUF_CON: d["con"], BUF_CONE: d["cone"],
                                      BUF_DX0: d["dx0"]}, rows=rows, cone_contacts=4, cone_dim=3)
    m = read_dump(path)
    assert (m["dims"].nv, m["dims"].nu, m["dims"].nc_max) == (dims.nv, dims.nu, dims.nc_max)
    assert m["batch"] == 3 and len(m["grids"]) == len(grids) and len(m["rows"]) == len(rows)
    assert all(a.type == b.type and a.dimf == b.dimf and a.dt == b.dt for a, b in zip(m["grids"], grids))
    assert all((a.var, a.index, a.sign, a.level) == (b.var, b.index, b.sign, b.level) for a, b in zip(m["rows"], rows))
    assert (m["cone_contacts"], m["cone_dim"]) == (4, 3)
    for b, key in ((BUF_KKT, "kkt"), (BUF_CDD, "cdd"), (BUF_CON, "con"), (BUF_CONE, "cone"), (BUF_DX0, "dx0")):
        assert np.array_equal(m["buffers"][b], d[key].reshape(-1))
    with open(path, "ab") as f:
        f.write(b"x")
    with pytest.raises(ValueError):
        read_dump(path)


@pytest.mark.gpu
def test_c_library_replays_a_numpy_dump_and_saves_it_back(tmp_path):
    from robotoc_amd import capi
    dims, grids, L, d = _problem()
    rows = joint_limit_rows(dims)
    src = tmp_path / "recorded.rtocdump"
    write_dump(src, dims, grids, 3, {BUF_KKT: d["kkt"], BUF_CDD: d["cdd"], BUF_CON: d["con"], BUF_CONE: d["cone"],
                                     BUF_DX0: d["dx0"]}, rows=rows, cone_contacts=4, cone_dim=3)
    # reference run: the same data uploaded by hand
    ref = capi.Context(dims, len(grids), 3, 0)
    rep = capi.Context.from_stage_dump(src, 0)
    try:
        ref.set_grid(grids)
        ref.set_constraint_rows(rows)
        ref.set_friction_cones(4, 3)
        for b, key in ((BUF_KKT, "kkt"), (BUF_CDD, "cdd"), (BUF_CON, "con"), (BUF_CONE, "cone"), (BUF_DX0, "dx0")):
            ref.upload(b, d[key])
        out = []
        for ctx in (ref, rep):
            ctx.condense()
            ctx.riccati_sweep()
            ctx.expand(0.995)
            assert (ctx.status() == 0).all()
            out.append((ctx.download_records(BUF_RIC, "ric"), ctx.download_records(BUF_DIR, "dir"),
                        ctx.download_records(BUF_CON, "con")))
        for a, b in zip(out[0], out[1]):
            assert np.array_equal(a, b)
        # and back: the replayed context dumps its state, numpy reads the same numbers
        dst = tmp_path / "after.rtocdump"
        rep.save_stage_dump(dst, [BUF_KKT, BUF_RIC, BUF_DIR])
        m = read_dump(dst)
        assert sorted(m["buffers"]) == [BUF_KKT, BUF_RIC, BUF_DIR]
        assert np.array_equal(m["buffers"][BUF_RIC], out[1][0].reshape(-1))
        assert np.array_equal(m["buffers"][BUF_DIR], out[1][1].reshape(-1))
    finally:
        ref.close()
        rep.close()


@pytest.mark.gpu
def test_malformed_dump_is_rejected(tmp_path):
    from robotoc_amd import capi
    bad = tmp_path / "bad.rtocdump"
    import ctypes as C
    h = C.c_void_p()
    bad.write_bytes(b"RTOCDMP1" + b"\0" * 300)  # right magic, nonsense header
    assert capi.lib().rtoc_load_stage_dump(str(bad).encode(), 0, C.byref(h)) == -7  # RTOC_ERR_IO
    assert capi.lib().rtoc_load_stage_dump(str(tmp_path / "missing").encode(), 0, C.byref(h)) == -7
