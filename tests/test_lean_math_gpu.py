"""GPU: accuracy of the float64 primitives AS THE DEVICE EXECUTES THEM (csrc/lean_math.h through uavenv_lean_math_eval: v_rcp_f64 /
v_rsq_f64 seeds, FMA contraction), against x87 long double on the host, over the operand ranges the env kernels feed them.
tests/test_lean_math.py measures the same source on the host, where lm_div / lm_rsqrt run stand-ins; this closes that gap."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _eval(op, a, b=None):
    import torch

    from drl_uav_cellularnet_amd import _capi

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    lib = _capi.load()
    ta = torch.as_tensor(a, dtype=torch.float64).cuda()
    tb = None if b is None else torch.as_tensor(b, dtype=torch.float64).cuda()
    o0, o1 = torch.empty_like(ta), torch.empty_like(ta)
    _capi.check(lib.uavenv_lean_math_eval(op, ta.data_ptr(), None if tb is None else tb.data_ptr(), o0.data_ptr(), o1.data_ptr(),
                                          ta.numel(), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    return o0.cpu().numpy(), o1.cpu().numpy()


def _ulp_err(got, ref_ld):
    ref = ref_ld.astype(np.float64)
    _, e = np.frexp(ref)
    ulp = np.ldexp(1.0, e - 53)
    err = np.abs((got.astype(np.longdouble) - ref_ld)).astype(np.float64) / ulp
    return float(err.max()), float(err.mean())


N = 1_000_000
LD = np.longdouble


def test_device_division_and_rsqrt():
    if np.finfo(LD).nmant < 60:
        pytest.skip("no extended-precision long double on this host")
    rs = np.random.RandomState(1)
    # lm_div call sites: f / (2 + f) in the logs (divisor in [1.70, 2.42)), power / (noise + interference)
    f = rs.uniform(-0.2929, 0.4143, N)
    got, _ = _eval(0, f, 2.0 + f)
    worst, mean = _ulp_err(got, f.astype(LD) / (2.0 + f).astype(LD))
    assert worst <= 1.0 and mean < 0.3, (worst, mean)
    num = 10.0 ** rs.uniform(-20, -3, N)
    den = 7.943282347242822e-16 + 10.0 ** rs.uniform(-20, -3, N)
    got, _ = _eval(0, num, den)
    worst, mean = _ulp_err(got, num.astype(LD) / den.astype(LD))
    assert worst <= 1.0 and mean < 0.3, (worst, mean)
    # lm_rsqrt call sites: squared distances (25 .. 25 * 2 * 200^2 m^2), Box-Muller t = -2 ln(1-u) in (0, 74]
    for x in (25.0 * rs.randint(1, 80000, N).astype(np.float64), rs.uniform(1e-12, 74.0, N)):
        got, _ = _eval(1, x)
        worst, mean = _ulp_err(got, 1.0 / np.sqrt(x.astype(LD)))
        assert worst <= 1.5 and mean < 0.4, (worst, mean)


def test_device_log_exp2_sincospi():
    if np.finfo(LD).nmant < 60:
        pytest.skip("no extended-precision long double on this host")
    rs = np.random.RandomState(2)
    u = (rs.randint(0, 2 ** 53, N, dtype=np.int64).astype(np.float64)) * 2.0 ** -53
    for x in (1.0 - u, 10.0 ** rs.uniform(-20, 13, N), rs.uniform(0.9, 1.1, N)):        # Box-Muller argument, SINR ratio, near 1
        got, _ = _eval(2, x)
        worst, mean = _ulp_err(got, np.log(x.astype(LD)))
        assert worst < 1.0 and mean < 0.3, (worst, mean)
    for x in (rs.uniform(-6.0, 6.0, N), rs.uniform(-60.0, 60.0, N)):                   # c_exp * fading, and a wide range
        got, _ = _eval(3, x)
        worst, mean = _ulp_err(got, np.exp2(x.astype(LD)))
        assert worst < 1.0 and mean < 0.3, (worst, mean)
    x = rs.uniform(0.0, 2.0, N)
    s, c = _eval(4, x)
    # reference with an EXACT argument reduction (q = rint(2x), r = x - q/2 is exact in float64), so that the error is relative
    # also next to the zeros of the functions: sin(pi x), cos(pi x) from sin / cos of pi r, |r| <= 1/4, rotated by the quadrant
    pi = LD("3.14159265358979323846264338327950288")
    q = np.rint(2.0 * x)
    r = (x - 0.5 * q).astype(LD)
    sr, cr = np.sin(pi * r), np.cos(pi * r)
    k = q.astype(np.int64) & 3
    ref_s = np.where(k == 0, sr, np.where(k == 1, cr, np.where(k == 2, -sr, -cr)))
    ref_c = np.where(k == 0, cr, np.where(k == 1, -sr, np.where(k == 2, -cr, sr)))
    ok = np.abs(ref_s.astype(np.float64)) > 1e-300
    ws, ms = _ulp_err(s[ok], ref_s[ok])
    ok = np.abs(ref_c.astype(np.float64)) > 1e-300
    wc, mc = _ulp_err(c[ok], ref_c[ok])
    assert ws < 2.0 and wc < 2.0 and ms < 0.4 and mc < 0.4, (ws, wc, ms, mc)
