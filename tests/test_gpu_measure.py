"""GPU: the library's measurement entries (bench.py's shader_mhz / this-box ceiling): the in-kernel clock stamps change no
result, survive graph-replayed steps, and read a plausible clock; the packed-FMA stream lands in a plausible range."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_clock_stamps_change_no_bit_and_read_a_plausible_clock(nbx, oracle):
    n = 65536
    b = oracle.round_inputs_to_f32(oracle.generate(7, n, 3))
    with nbx.Context(n, 3) as ctx:
        ctx.upload(b)
        ctx.compute_accel()
        plain = ctx.forces(oracle.G)
        with pytest.raises(nbx.NbxError):          # nothing stamped has run yet
            ctx.shader_clock()
        ctx.enable_clock_stamps(True)
        ctx.compute_accel()
        stamped = ctx.forces(oracle.G)
        clk = ctx.shader_clock()
        assert np.array_equal(plain, stamped)
        assert 500.0 < clk["min_mhz"] <= clk["median_mhz"] <= clk["max_mhz"] < 2600.0, clk
        assert clk["workgroups"] >= n // 2048
        # graph-replayed steps carry the stamps too (the captured step is re-captured when stamping toggles)
        ctx.step(1.0, 6, oracle.G)
        clk2 = ctx.shader_clock()
        assert 500.0 < clk2["median_mhz"] < 2600.0, clk2
        ctx.enable_clock_stamps(False)
        got = b.copy()
        ctx.download(got)
    ref = b.copy()
    with nbx.Context(n, 3) as ctx:
        ctx.upload(ref)
        ctx.step(1.0, 6, oracle.G)
        ctx.download(ref)
    assert np.array_equal(got, ref), "stepping with stamps on must move the bodies exactly as without"


def test_variants_without_stamps_say_so(nbx, oracle):
    b = oracle.round_inputs_to_f32(oracle.generate(8, 4096, 3))
    with nbx.Context(4096, 3) as ctx:
        ctx.upload(b)
        ctx.set_tuning(0, nbx.variants().index("strict_f64_t4"))
        with pytest.raises(nbx.NbxError):
            ctx.enable_clock_stamps(True)


def test_valu_ceiling_is_plausible(nbx):
    tf, mhz = nbx.package.capi.measure_valu_ceiling(0, 30.0)
    assert 60.0 < tf < 160.0, tf       # 157.3 = 2.4 GHz x 256 CUs x 64 lane pairs x 4 flop; a dense stream holds less than 2.4 GHz
    assert 1000.0 < mhz < 2600.0, mhz
    with pytest.raises(nbx.NbxError):
        nbx.package.capi.measure_valu_ceiling(0, 0.0)
